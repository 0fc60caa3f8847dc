// Convolution weight gradient on MFMA (gfx950):
//   dw[n, tap*C + c] = sum_p dy[p, n] * x[p @ tap, c]           (reduction over pixels p)
//
// As a GEMM the reduction dimension (pixels) is the SLOW dimension of both NHWC operands, while
// MFMA fragments want 8 (bf16) / 4 (f32) consecutive k per lane.  Each thread therefore loads an
// 8x8 (bf16) or 4x4 (f32) [pixel][channel] block with 16-byte loads, transposes it in registers
// (v_perm_b32 pairs for bf16, pure register renaming for f32) and writes [channel][pixel] rows
// with ds_write_b128 into the SAME swizzled 128-byte-row LDS tile format as the forward kernel,
// so the fragment fetch + MFMA core is identical.  One block = one (n-tile, tap, c-tile, split);
// split-K partials go to a workspace and are reduced deterministically.
#include "gdl_common.h"

namespace {

struct WArgs {
  gdl_wgrad_args a;
  int64_t P;          // B*Ho*Wo
  int ctiles;         // C / 128
  int splits;
  int64_t p_per_split;  // multiple of the K-step
  int x_dense, dy_dense;
};

template <typename T> struct WT;
template <> struct WT<float> { static constexpr int ES = 4, BKP = 32, BLK = 4; };      // 4x4 blocks
template <> struct WT<bf16_tag> { static constexpr int ES = 2, BKP = 64, BLK = 8; };   // 8x8 blocks

__device__ __forceinline__ void transpose8x8_b16(uint4 (&r)[8]) {
  // rows r[i] = 8 halfwords a[i][0..7]; result r[j] = a[0..7][j]
  uint32_t t[4][8];  // t[i][j] = (a[2i][j], a[2i+1][j])
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t x[4] = {r[2 * i].x, r[2 * i].y, r[2 * i].z, r[2 * i].w};
    const uint32_t y[4] = {r[2 * i + 1].x, r[2 * i + 1].y, r[2 * i + 1].z, r[2 * i + 1].w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      t[i][2 * k] = __builtin_amdgcn_perm(y[k], x[k], 0x05040100u);      // (x.lo, y.lo)
      t[i][2 * k + 1] = __builtin_amdgcn_perm(y[k], x[k], 0x07060302u);  // (x.hi, y.hi)
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = make_uint4(t[0][j], t[1][j], t[2][j], t[3][j]);
}

__device__ __forceinline__ void transpose4x4_b32(uint4 (&r)[4]) {
  const uint4 a = r[0], b = r[1], c = r[2], d = r[3];
  r[0] = make_uint4(a.x, b.x, c.x, d.x);
  r[1] = make_uint4(a.y, b.y, c.y, d.y);
  r[2] = make_uint4(a.z, b.z, c.z, d.z);
  r[3] = make_uint4(a.w, b.w, c.w, d.w);
}

// tile = 128 (n) x 128 (c), 4 waves as 2x2, each wave 2x2 MFMA tiles of 32x32
template <typename T>
__global__ __launch_bounds__(256) void wgrad_kernel(const WArgs k) {
  constexpr int ES = WT<T>::ES, BKP = WT<T>::BKP, BLK = WT<T>::BLK;
  constexpr int BM = 128, BN = 128, TM = 2, TN = 2;
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const gdl_wgrad_args& a = k.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int n0 = blockIdx.x * BM;
  const int tap = blockIdx.y / k.ctiles, c0 = (blockIdx.y % k.ctiles) * BN;
  const int tap_r = tap / a.S, tap_s = tap % a.S;
  const int split = blockIdx.z;
  const int64_t p_begin = (int64_t)split * k.p_per_split;
  int64_t p_end = p_begin + k.p_per_split;
  if (p_end > k.P) p_end = k.P;
  const int HoWo = a.Ho * a.Wo;

  // ---- staging roles ----
  // bf16: threads 0..127 stage dy (A), 128..255 stage x (B); one 8x8 block each.
  // f32 : every thread stages one 4x4 block of A and one of B.
  constexpr int GROUPS = 128 / BLK;  // channel groups across the tile: 16 (bf16) / 32 (f32)
  const int bt = (ES == 2) ? (tid & 127) : tid;
  const int cg = bt % GROUPS;        // channel group (BLK channels = 16 bytes)
  const int pg = bt / GROUPS;        // pixel group (BLK pixels) 0..7
  const bool do_a = (ES == 4) || tid < 128;
  const bool do_b = (ES == 4) || tid >= 128;

  uint4 ra[BLK], rb[BLK];

  auto fetch = [&](int64_t pbase) {
#pragma unroll
    for (int i = 0; i < BLK; ++i) {
      const int64_t p = pbase + pg * BLK + i;
      const bool pv = p < p_end;
      int b = 0, oy = 0, ox = 0;
      const int64_t pp = pv ? p : 0;
      if (!(k.x_dense && k.dy_dense)) {
        b = (int)(pp / HoWo);
        const int rem = (int)(pp - (int64_t)b * HoWo);
        oy = rem / a.Wo; ox = rem - oy * a.Wo;
      }
      if (do_a) {
        const int n = n0 + cg * BLK;
        const int64_t off = k.dy_dense ? pp * a.dy_sW
                                       : (int64_t)b * a.dy_sB + (int64_t)oy * a.dy_sH + (int64_t)ox * a.dy_sW;
        ra[i] = (pv && n < a.N) ? *(const uint4*)((const unsigned char*)a.dy + (off + n) * ES)
                                : make_uint4(0, 0, 0, 0);
      }
      if (do_b) {
        const int c = c0 + cg * BLK;
        bool ok = pv && c < a.C;
        int64_t off;
        if (k.x_dense) {
          off = pp * a.in_sW;
        } else {
          const int iy = oy * a.stride + tap_r - a.pad, ix = ox * a.stride + tap_s - a.pad;
          ok = ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
          off = (int64_t)b * a.in_sB + (int64_t)iy * a.in_sH + (int64_t)ix * a.in_sW;
        }
        rb[i] = ok ? *(const uint4*)((const unsigned char*)a.in + (off + c) * ES) : make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto stash = [&](int stage) {
    unsigned char* sa = smem + stage * STAGE_BYTES;
    unsigned char* sb = sa + BM * 128;
    if (do_a) {
      if constexpr (ES == 2) transpose8x8_b16(ra); else transpose4x4_b32(ra);
#pragma unroll
      for (int j = 0; j < BLK; ++j) {
        const int r = cg * BLK + j;  // channel row of the tile
        *(uint4*)(sa + r * 128 + ((pg ^ ((r >> 1) & 7)) << 4)) = ra[j];
      }
    }
    if (do_b) {
      if constexpr (ES == 2) transpose8x8_b16(rb); else transpose4x4_b32(rb);
#pragma unroll
      for (int j = 0; j < BLK; ++j) {
        const int r = cg * BLK + j;
        *(uint4*)(sb + r * 128 + ((pg ^ ((r >> 1) & 7)) << 4)) = rb[j];
      }
    }
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5, fswz = (frow >> 1) & 7;
  const int a_lds0 = (wm * TM * 32 + frow) * 128;
  const int b_lds0 = BM * 128 + (wn * TN * 32 + frow) * 128;

  const int KT = (int)((p_end - p_begin + BKP - 1) / BKP);
  if (KT > 0) {
    fetch(p_begin);
    stash(0);
  }
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const bool more = kt + 1 < KT;
    if (more) fetch(p_begin + (int64_t)(kt + 1) * BKP);
    const unsigned char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int coff = (((2 * kk + fhalf) ^ fswz) << 4);
      uint4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *(const uint4*)(st + a_lds0 + i * 32 * 128 + coff);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *(const uint4*)(st + b_lds0 + j * 32 * 128 + coff);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (ES == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8_t, fa[i]), __builtin_bit_cast(bf16x8_t, fb[j]), acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].x), __uint_as_float(fb[j].x), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].y), __uint_as_float(fb[j].y), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].z), __uint_as_float(fb[j].z), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].w), __uint_as_float(fb[j].w), acc[i][j], 0, 0, 0);
          }
        }
    }
    if (more) stash((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: rows = n, cols = c ----
  float* dst;
  int64_t ld;
  if (k.splits > 1) {
    ld = (int64_t)a.R * a.S * a.C;
    dst = a.workspace + (int64_t)split * a.N * ld;
  } else {
    ld = a.dw_sN;
    dst = a.dw;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
      if (n >= a.N) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int c = c0 + (wn * TN + j) * 32 + frow;
        if (c >= a.C) continue;
        float* q = dst + (int64_t)n * ld + (int64_t)tap * a.C + c;
        const float v = acc[i][j][r];
        *q = (k.splits == 1 && a.accumulate) ? *q + v : v;
      }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int splits, int N,
                                                           int64_t K, float* dw, int64_t dw_sN, int accumulate) {
  const int64_t total = (int64_t)N * K;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int sp = 0; sp < splits; ++sp) s += ws[(int64_t)sp * total + i];
    const int64_t n = i / K, kk = i - n * K;
    float* q = dw + n * dw_sN + kk;
    *q = accumulate ? *q + s : s;
  }
}

int choose_splits(const gdl_wgrad_args& a, int64_t P, int bkp) {
  const int64_t tiles = (int64_t)((a.N + 127) / 128) * a.R * a.S * ((a.C + 127) / 128);
  int64_t want = (1024 + tiles - 1) / tiles;           // aim for ~1024 blocks
  const int64_t max_by_k = P / (8 * bkp);              // keep >= 8 K-steps per split
  if (want > max_by_k) want = max_by_k;
  if (want > 64) want = 64;
  if (want < 1) want = 1;
  return (int)want;
}

}  // namespace

extern "C" int64_t gdl_conv_wgrad_workspace(const gdl_wgrad_args* ap) {
  if (!ap) return 0;
  const gdl_wgrad_args& a = *ap;
  const int64_t P = (int64_t)a.B * a.Ho * a.Wo;
  const int bkp = a.dtype == GDL_BF16 ? 64 : 32;
  const int splits = choose_splits(a, P, bkp);
  if (splits <= 1) return 0;
  return (int64_t)splits * a.N * a.R * a.S * a.C * (int64_t)sizeof(float);
}

extern "C" int gdl_conv_wgrad(const gdl_wgrad_args* ap, gdl_stream_t stream) {
  GDL_CHECK_ARG(ap != nullptr, "gdl_conv_wgrad: null args");
  const gdl_wgrad_args& a = *ap;
  GDL_CHECK_ARG(a.dtype == GDL_F32 || a.dtype == GDL_BF16, "gdl_conv_wgrad: bad dtype");
  GDL_CHECK_ARG(a.in && a.dy && a.dw, "gdl_conv_wgrad: null tensor");
  const int es = (int)gdl_elem_size(a.dtype), al = 16 / es;
  GDL_CHECK_ARG(a.C % al == 0 && a.N % al == 0, "gdl_conv_wgrad: C and N must be multiples of %d", al);
  GDL_CHECK_ARG(a.in_sB % al == 0 && a.in_sH % al == 0 && a.in_sW % al == 0 && a.dy_sB % al == 0 &&
                    a.dy_sH % al == 0 && a.dy_sW % al == 0, "gdl_conv_wgrad: strides must keep 16-byte alignment");
  GDL_CHECK_ARG(((uintptr_t)a.in % 16 == 0) && ((uintptr_t)a.dy % 16 == 0), "gdl_conv_wgrad: pointers must be 16-byte aligned");
  WArgs k;
  k.a = a;
  k.P = (int64_t)a.B * a.Ho * a.Wo;
  k.ctiles = (a.C + 127) / 128;
  const int bkp = a.dtype == GDL_BF16 ? 64 : 32;
  k.splits = choose_splits(a, k.P, bkp);
  if (k.splits > 1) {
    const int64_t need = (int64_t)k.splits * a.N * a.R * a.S * a.C * (int64_t)sizeof(float);
    GDL_CHECK_ARG(a.workspace && a.workspace_bytes >= need, "gdl_conv_wgrad: workspace too small (%lld needed)", (long long)need);
  }
  int64_t per = (k.P + k.splits - 1) / k.splits;
  per = (per + bkp - 1) / bkp * bkp;
  k.p_per_split = per;
  k.x_dense = (a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0 && a.H == a.Ho && a.W == a.Wo &&
               a.in_sH == (int64_t)a.W * a.in_sW && a.in_sB == (int64_t)a.H * a.in_sH);
  k.dy_dense = (a.dy_sH == (int64_t)a.Wo * a.dy_sW && a.dy_sB == (int64_t)a.Ho * a.dy_sH);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((a.N + 127) / 128, a.R * a.S * k.ctiles, k.splits);
  const size_t lds = 2 * (128 + 128) * 128;
  if (a.dtype == GDL_BF16) hipLaunchKernelGGL(wgrad_kernel<bf16_tag>, grid, dim3(256), lds, s, k);
  else hipLaunchKernelGGL(wgrad_kernel<float>, grid, dim3(256), lds, s, k);
  if (k.splits > 1) {
    const int64_t K = (int64_t)a.R * a.S * a.C;
    int64_t g = ((int64_t)a.N * K + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)g), dim3(256), 0, s, a.workspace, k.splits, a.N, K, a.dw,
                       a.dw_sN, a.accumulate);
  }
  GDL_CHECK_LAUNCH("gdl_conv_wgrad");
  return GDL_OK;
}
