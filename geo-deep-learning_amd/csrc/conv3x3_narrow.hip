// Direct 3x3 / stride-1 / pad-1 convolution for NARROW-INPUT layers on large maps: C in {8, 16, 32} input channels, dense
// NHWC input, any number of output channels in 32-wide slices (blockIdx.y) -- UNet++'s 512^2 / 256^2 decoder stages and
// their data gradients (e.g. 32 -> 320 channels), the 16 -> classes head; smp DecoderBlock / SegmentationHead, reference
// call site tasks_with_models/segmentation_unetplus.py:126-131.
//
// These layers are HBM-bound (32 + 32 bytes per pixel at C = N = 16 against 4.6 kFLOP), but the implicit-GEMM tiles treat
// them as GEMMs with K = 9 C: every filter tap re-stages the activation tile (9 x the bytes through LDS-DMA) and a
// 256-pixel tile lives for three K-steps, so prologue / epilogue dominate -- 0.8-1.0 ms per layer at batch 32 against
// ~0.15 ms of memory time (tools/log_conv_plans.py).  Here a block owns 4 rows x 64 columns of one image:
//   * the (4+2) x (64+2) pixel window is staged ONCE by LDS-DMA (out-of-image pixels arrive as hardware zeros = the
//     convolution's zero padding), 6 / 13 / 25 KiB for C = 8 / 16 / 32;
//   * the block's 32-output-channel slice of the filter ([32][9 C] bf16) sits in registers as MFMA operand fragments;
//   * a wave computes 32 consecutive pixels of one row per unit: per k16 group one ds_read_b128 of the window (the tap is
//     a byte offset) and one v_mfma_f32_32x32x16_bf16, 5 / 9 / 18 groups per unit, then the common epilogue
//     (bias, folded BN, ReLU, residual, bf16 / f32 output).
// K order inside the registers is the weight matrix's own (r, s, c); bit-exactness with the implicit-GEMM kernels is not
// claimed (the accumulation order over K differs), results agree to f32 rounding of the same bf16 products.
#include "conv_gemm_common.h"

using namespace gdlconv;

namespace {

constexpr int TH = 4, TW = 64, WIN_W = TW + 2;

template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_narrow_kernel(const KArgs k) {
  constexpr int NG = (9 * CIN + 15) / 16;                   // k16 groups
  constexpr int CPP = CIN / 8;                              // 16-byte chunks per pixel
  constexpr int CHUNKS = (TH + 2) * WIN_W * CPP;
  constexpr int NP = (CHUNKS + 63) / 64;                    // 1 KiB DMA pieces
  constexpr unsigned kOob = 0x80000000u;
  __shared__ __attribute__((aligned(16))) unsigned char win[NP * 1024];

  const gdl_conv_args& a = k.a;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles_x = a.W / TW, tiles_y = a.H / TH;
  const int b = blockIdx.x / (tiles_x * tiles_y);
  const int t = blockIdx.x - b * (tiles_x * tiles_y);
  const int y0 = (t / tiles_x) * TH, x0 = (t % tiles_x) * TW;
  const srd_t srd_a = make_srd(a.in, k.in_span);
  const unsigned lds_base =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)win);

  // ---- stage the window: chunk q = (pixel q / CPP of the flattened window, 16-byte piece q % CPP)
  for (int i = wave; i < NP; i += 4) {
    const int q = i * 64 + lane;
    const int pix = q / CPP, sub = q - pix * CPP;
    const int r = pix / WIN_W, c = pix - r * WIN_W;
    const int gy = y0 - 1 + r, gx = x0 - 1 + c;
    const bool ok = q < CHUNKS && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    const unsigned v = ok ? (unsigned)((((int64_t)b * a.H + gy) * a.W + gx) * CIN * 2 + sub * 16) : kOob;
    dma16_buf(v, srd_a, 0u, lds_base + i * 1024);
  }

  // ---- the filter as MFMA fragments: lane (n = lane & 31, half = lane >> 5) holds k = 16 g + 8 half .. + 7 of row n
  const int frow = lane & 31, fhalf = lane >> 5;
  const int n0 = blockIdx.y * 32;
  bf16x8_t wf[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int kidx = g * 16 + fhalf * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (n0 + frow < a.N && kidx < 9 * CIN) v = *(const uint4*)((const uint16_t*)a.w + (int64_t)(n0 + frow) * a.w_sN + kidx);
    wf[g] = __builtin_bit_cast(bf16x8_t, v);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int unit = wave + 4 * u, urow = unit >> 1, ucol = (unit & 1) * 32;
    f32x16_t acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    const unsigned char* p0 = win + ((urow * WIN_W + ucol + frow) * CIN) * 2;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      // this lane's 8 K elements of group g: tap = k / CIN, channels k % CIN .. + 7 (the two halves may sit in two taps)
      const int k0 = g * 16, k1 = g * 16 + 8;
      const int t0 = k0 / CIN, c0 = k0 % CIN, t1 = k1 / CIN, c1 = k1 % CIN;
      const int off0 = (((t0 / 3) * WIN_W + (t0 % 3)) * CIN + c0) * 2;
      const int off1 = t1 < 9 ? (((t1 / 3) * WIN_W + (t1 % 3)) * CIN + c1) * 2 : off0;   // beyond the filter: weights are zero
      const uint4 v = *(const uint4*)(p0 + (fhalf ? off1 : off0));
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[g], __builtin_bit_cast(bf16x8_t, v), acc[0][0], 0, 0, 0);
    }
    const int m0 = ((b * a.H + y0 + urow) * a.W) + x0 + ucol;
    conv_epilogue<1, 1, false>(k, acc, m0, n0, 0, 0, lane, 0);
  }
}

}  // namespace

namespace gdlconv {

bool conv3x3_narrow_applicable(const gdl_conv_args& a) {
  return a.dtype == GDL_BF16 && a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1 && a.nz == 1 && a.Ho == a.H &&
         a.Wo == a.W && (a.C == 8 || a.C == 16 || a.C == 32) && a.N <= 32 * 65535 && a.in_sW == a.C &&
         a.in_sH == (int64_t)a.W * a.C && a.in_sB == (int64_t)a.H * a.W * a.C && a.W % TW == 0 && a.H % TH == 0 &&
         a.w_sN % 8 == 0 && !a.aux_out && a.act != GDL_ACT_MUL_GELU_GRAD &&
         (int64_t)a.B * (a.H / TH) * (a.W / TW) < (1ll << 31);
}

int conv3x3_narrow_launch(const KArgs& k, hipStream_t stream) {
  const gdl_conv_args& a = k.a;
  dim3 grid((unsigned)((int64_t)a.B * (a.H / TH) * (a.W / TW)), (unsigned)((a.N + 31) / 32)), block(256);
  if (a.C == 8) hipLaunchKernelGGL(conv3x3_narrow_kernel<8>, grid, block, 0, stream, k);
  else if (a.C == 16) hipLaunchKernelGGL(conv3x3_narrow_kernel<16>, grid, block, 0, stream, k);
  else hipLaunchKernelGGL(conv3x3_narrow_kernel<32>, grid, block, 0, stream, k);
  GDL_CHECK_LAUNCH("gdl_conv_gemm(3x3 narrow)");
  return GDL_OK;
}

}  // namespace gdlconv
