// 3x3 / stride-1 implicit-GEMM convolution with SHARED activation staging (256x256 tile, 8 waves, ping-pong loaders).
//
// In conv_gemm.hip every K-step (one filter tap, one 64-channel chunk) stages its own 256-row activation tile, but
// the three taps s = 0,1,2 of one filter row read the SAME pixels shifted by one: tile(s) = rows [m0-1+s, m0+255+s]
// of the flattened pixel stream.  This kernel stages ONE 258-row tile per (chunk, filter row) and lets the three
// K-steps fetch their fragments at row offsets 0 / +1 / +2.  An LDS-DMA piece blocks the issuing wave for ~100
// cycles and that -- not bandwidth -- bounds the ping-pong kernel, so 36 + 3*32 = 132 pieces per three K-steps instead
// of 3*64 = 192 is a direct cut of the critical path.
//
// What the flattening breaks, and how it is repaired: a staged row is just "pixel q of the NHWC stream", so for an
// output pixel on an image edge the shifted row is a real pixel of the neighbouring image row / image instead of the
// zero padding.  Every lane therefore carries, per 32-row tile it feeds to the MFMA, a 9-bit mask of the taps that are
// inside the image for ITS output row, and zeroes the activation fragment (4 v_cndmask) where the tap is outside.
// Rows before the tensor start / after its end are zero-filled by the buffer descriptor.
//
// Requirements (checked on the host): R = S = 3, stride 1, pad 1, pixel-dense input (in_sH = W*in_sW,
// in_sB = H*in_sH), no batching over z.  K order (chunk, r, s) -- the same as conv_gemm.hip's default.
#include "conv_gemm_common.h"

using namespace gdlconv;

namespace {

template <typename T>
__global__ __launch_bounds__(512) void conv3x3_sf_kernel(const KArgs k) {
  constexpr int ES = TileTraits<T>::ES;
  constexpr int BKE = TileTraits<T>::BKE;
  constexpr int TM = 4, TN = 2, WARPS_N = 4;
  constexpr int BM = 256, BN = 256;
  constexpr int A_PIECES = 36;                       // 288 staged rows >= BM + 2, a multiple of 3 slots x 4 loaders
  constexpr int A_BYTES = A_PIECES * 1024, B_BYTES = BN * 128;
  constexpr unsigned kOob = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // A0 | A1 | B0 | B1

  const gdl_conv_args& a = k.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  const int lwave = wave & 3, half = wave >> 2;      // waves w and w+4 share a SIMD and alternate as loaders

  const int lid = xcd_remap(blockIdx.x, k.tiles_m * k.tiles_n);
  int tile_m, tile_n;
  tile_order(k, lid, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const srd_t srd_a = make_srd(a.in, k.in_span);
  const srd_t srd_b = make_srd(a.w, k.w_span);
  const unsigned lds_base =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
  const int W = a.W, HW = a.H * a.W;
  const int n_macro = k.kc * 3;                      // (chunk, filter row) pairs
  const int KT = n_macro * 3;

  // ---- DMA geometry
  const int lrow = lane >> 3, lslot = lane & 7;
  unsigned b_voff[8];                                // loader wave: 8 weight pieces per K-step (rows (i*4+lwave)*8..)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = (i * 4 + lwave) * 8 + lrow;
    const int chunk = lslot ^ ((r >> 1) & 7);
    const int n = n0 + r;
    b_voff[i] = n < a.N ? (unsigned)((n * a.w_sN + chunk * (16 / ES)) * ES) : kOob;
  }
  // activation piece p (0..35) = staged rows 8p..8p+7 <-> pixels q0 + 8p + lrow, q0 = m0 - 1 + (r - 1) * W.
  // slot `part` (0..2) of a macro step carries pieces part*12 + lwave*3 + {0,1,2} of the loader half.
  auto issue_a = [&](int g, int part) {
    const int cc = g / 3, r = g - 3 * cc;
    const int q0 = m0 - 1 + (r - 1) * W;
    const unsigned lds = lds_base + (g & 1) * A_BYTES;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int p = part * 12 + lwave * 3 + i;
      const int j = p * 8 + lrow;
      const int chunk = lslot ^ ((j >> 1) & 7);
      const int q = q0 + j;
      const unsigned v = (unsigned)q < (unsigned)k.M ? (unsigned)((q * a.in_sW + cc * BKE + chunk * (16 / ES)) * ES)
                                                     : kOob;
      dma16_buf(v, srd_a, 0u, lds + p * 1024);
    }
  };
  auto issue_b = [&](int t) {                        // weights of K-step t = (cc*3 + r)*3 + s
    const int g = t / 3, s = t - 3 * g, cc = g / 3, r = g - 3 * cc;
    const unsigned wk = (unsigned)(((r * 3 + s) * a.C + cc * BKE) * ES);
    const unsigned lds = lds_base + 2 * A_BYTES + (t & 1) * B_BYTES + lwave * 1024;
#pragma unroll
    for (int i = 0; i < 8; ++i) dma16_buf(b_voff[i], srd_b, wk, lds + i * 4 * 1024);
  };
  // DMA "slot" after the barrier that ends K-step t: weights of step t+2 and one third of the activations of the
  // macro step after next ((t+1)/3 + 1), issued by loader half (t & 1)
  auto slot = [&](int t) {
    if (half != (t & 1) || k.dbg == 1) return;
    if (t + 2 < KT) issue_b(t + 2);
    const int u = t + 1, g = u / 3 + 1;
    if (g < n_macro) issue_a(g, u - 3 * (u / 3));
  };

  // ---- per-lane tap validity of the output rows this lane feeds as MFMA operand (row = lane & 31 of tile i)
  const int frow = lane & 31, fhalf = lane >> 5;
  unsigned fmask[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + (wm * TM + i) * 32 + frow;
    unsigned mask = 0;
    if (m < k.M) {
      const int rem = m % HW, oy = rem / W, ox = rem - oy * W;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s)
          if ((unsigned)(oy + r - 1) < (unsigned)a.H && (unsigned)(ox + s - 1) < (unsigned)W) mask |= 1u << (r * 3 + s);
    }
    fmask[i] = mask;
  }

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addressing: output row `row` of the block tile <-> staged row row + s (staged row 0 = pixel m0 - 1 of the
  // filter row's stream).  (row >> 1) & 7 is unchanged by + 32*i, so one swizzle per s serves the four tiles.
  int a_off[3], a_swz[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int row = wm * TM * 32 + frow + s;
    a_off[s] = row * 128;
    a_swz[s] = (row >> 1) & 7;
  }
  const int b_swz = (frow >> 1) & 7;
  const int b_off = (wn * TN * 32 + frow) * 128;

  uint4 fa[2][TM], fb[2][TN];
  auto fetch = [&](int g, int t, int s, int kk, int buf) {     // s is a compile-time constant at every call site
    const unsigned char* sa = smem + (g & 1) * A_BYTES + a_off[s] + (((2 * kk + fhalf) ^ a_swz[s]) << 4);
    const unsigned char* sb = smem + 2 * A_BYTES + (t & 1) * B_BYTES + b_off + (((2 * kk + fhalf) ^ b_swz) << 4);
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[buf][i] = *(const uint4*)(sa + i * 32 * 128);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[buf][j] = *(const uint4*)(sb + j * 32 * 128);
  };
  auto mfmas = [&](int buf, unsigned bit) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (!(fmask[i] & bit)) fa[buf][i] = make_uint4(0, 0, 0, 0);    // tap outside the image for this lane's row
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (ES == 2) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(bf16x8_t, fb[buf][j]), __builtin_bit_cast(bf16x8_t, fa[buf][i]), acc[i][j], 0, 0, 0);
        } else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fb[buf][j].x), __uint_as_float(fa[buf][i].x), acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fb[buf][j].y), __uint_as_float(fa[buf][i].y), acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fb[buf][j].z), __uint_as_float(fa[buf][i].z), acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fb[buf][j].w), __uint_as_float(fa[buf][i].w), acc[i][j], 0, 0, 0);
        }
      }
    }
  };

  const unsigned long long t0c = k.probe ? __builtin_readcyclecounter() : 0;
  const unsigned long long t0r = k.probe ? __builtin_amdgcn_s_memrealtime() : 0;
  // ---- prologue: weights of step 0 and the whole first activation tile, then the "slot -1" work
  if (half == 0) { issue_b(0); issue_a(0, 0); } else { issue_a(0, 1); issue_a(0, 2); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  slot(-1);           // t = -1: half 1 loads the weights of step 1 and the first third of macro step 1
  // all scalar (kernel-argument) loads are complete here: tell the waitcnt inserter, so that inside the loop it can
  // wait for the OLDER fragment reads only (lgkmcnt(6)) instead of draining every LDS read before the first MFMAs
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
  fetch(0, 0, 0, 0, 0);

  // ---- main loop over macro steps g = (chunk, filter row); the three taps s are unrolled (static fragment offsets)
  for (int g = 0; g < n_macro; ++g) {
    const int r = g % 3;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int t = g * 3 + s;
      const unsigned bit = 1u << (r * 3 + s);
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        fetch(g, t, s, kk + 1, (kk + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(kk & 1, bit);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (t + 1 < KT) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA for step t+1 (weights / activations) landed
        __syncthreads();                                   // ... everyone's; the stages of step t are free
        slot(t);
        if (s < 2) fetch(g, t + 1, s + 1, 0, 0);
        else fetch(g + 1, t + 1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfmas(1, bit);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (k.probe && tid == 0 && blockIdx.x < 2048) {
    k.probe[2 * blockIdx.x] = __builtin_readcyclecounter() - t0c;
    k.probe[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
  }
  __syncthreads();   // every wave is past its last fragment read: the stages become the epilogue's transpose buffers
  conv_epilogue<TM, TN, false>(k, acc, m0, n0, wm, wn, lane, 0, smem + wave * 8192, smem + 8 * 8192 + wave * 1024);
  if (k.probe && tid == 0 && blockIdx.x < 2048) {
    k.probe[4096 + blockIdx.x] = __builtin_readcyclecounter() - t0c;
    k.probe[8192 + 2 * blockIdx.x] = t0r;                                   // block timeline (100 MHz ticks)
    k.probe[8192 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
}

}  // namespace

namespace gdlconv {

// Can this call run on the shared-staging kernel?
bool conv3x3_sf_applicable(const gdl_conv_args& a) {
  return a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1 && a.nz == 1 && a.Ho == a.H && a.Wo == a.W &&
         a.in_sH == (int64_t)a.W * a.in_sW && a.in_sB == (int64_t)a.H * a.in_sH && a.N % 256 == 0 && !a.aux_out &&
         a.act != GDL_ACT_MUL_GELU_GRAD;
}

int conv3x3_sf_launch(const KArgs& k, hipStream_t stream) {
  KArgs kk = k;
  kk.tiles_m = (k.M + 255) / 256;
  kk.tiles_n = (k.a.N + 255) / 256;
  kk.n_group = conv_n_group(k.a, 256, 256, 32);
  const size_t lds = 2 * 36 * 1024 + 2 * 256 * 128;
  dim3 grid(kk.tiles_m * kk.tiles_n), block(512);
  if (k.a.dtype == GDL_BF16) {
    GDL_SET_MAX_LDS_ONCE(conv3x3_sf_kernel<bf16_tag>, lds);
    hipLaunchKernelGGL(conv3x3_sf_kernel<bf16_tag>, grid, block, lds, stream, kk);
  } else {
    GDL_SET_MAX_LDS_ONCE(conv3x3_sf_kernel<float>, lds);
    hipLaunchKernelGGL(conv3x3_sf_kernel<float>, grid, block, lds, stream, kk);
  }
  GDL_CHECK_LAUNCH("gdl_conv_gemm(3x3 shared staging)");
  return GDL_OK;
}

}  // namespace gdlconv
