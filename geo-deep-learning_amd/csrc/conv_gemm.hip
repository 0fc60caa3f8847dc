// Implicit-GEMM convolution / linear for gfx950 (CDNA4) on MFMA.
//
//   out[m, n] = epi( alpha * sum_{tap, c} in[pixel(m) + tap, c] * w[n, tap*C + c] )
//
// * NHWC activations, [N][R*S*C] weights: both operands are K-contiguous, so an operand
//   tile is ROWS x 128 bytes of K (64 bf16 or 32 f32) -- im2col never exists in memory;
//   the 3x3 halo/zero padding is resolved per 16-byte chunk while staging.
// * Staging: global -> VGPR (16 B/lane, issued BEFORE the MFMA phase of the current tile)
//   -> LDS (written AFTER it), two LDS stages, one barrier per K-step.
// * LDS tile rows are 128 B; 16-byte chunk c of row r lives at chunk (c ^ ((r>>1)&7)), which
//   makes the 4 x 16-lane groups of a ds_read_b128 fragment fetch conflict-free.
// * MFMA: v_mfma_f32_32x32x16_bf16 (bf16) or v_mfma_f32_32x32x2_f32 (exact f32 -- the parity
//   path); a wave owns TM x TN tiles of 32x32, f32 accumulators stay in registers.
// * K order inside a 32-byte chunk pair is permuted identically for both operands (legal for
//   a reduction), so both dtypes fetch fragments with the same ds_read_b128.
// * blockIdx -> tile mapping is XCD-aware: the blocks of one XCD walk neighbouring N tiles of
//   the same M rows so the activation rows are re-read from that XCD's L2.
#include "gdl_common.h"

namespace {

struct KArgs {
  gdl_conv_args a;
  int M;        // B*Ho*Wo
  int kc;       // C / BKE
  int KT;       // R*S*kc
  int tiles_m, tiles_n;
  int in_dense, out_dense, res_dense;
};

template <typename T> struct TileTraits;
template <> struct TileTraits<float> { static constexpr int ES = 4; static constexpr int BKE = 32; };
template <> struct TileTraits<bf16_tag> { static constexpr int ES = 2; static constexpr int BKE = 64; };

__device__ __forceinline__ int xcd_remap(int id, int n) {
  // bijective "XCD-major" remap (cdna guide T1): hardware places block id on XCD id % 8.
  const int q = n >> 3, r = n & 7, xcd = id & 7, idx = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <typename T, int WARPS_M, int WARPS_N, int TM, int TN>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N) void conv_gemm_kernel(const KArgs k) {
  constexpr int ES = TileTraits<T>::ES;
  constexpr int BKE = TileTraits<T>::BKE;
  constexpr int BM = WARPS_M * TM * 32, BN = WARPS_N * TN * 32;
  constexpr int NT = 64 * WARPS_M * WARPS_N;
  constexpr int CA = BM * 8 / NT, CB = BN * 8 / NT;  // 16-B chunks per thread per tile
  constexpr int ROWSTEP = NT / 8;
  static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile/threads mismatch");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // layout: [stage][A: BM*128 | B: BN*128]
  constexpr int STAGE_BYTES = (BM + BN) * 128;

  const gdl_conv_args& a = k.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;

  const int lid = xcd_remap(blockIdx.x, k.tiles_m * k.tiles_n);
  const int tile_n = lid % k.tiles_n, tile_m = lid / k.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.y;
  const int z0 = z / a.nz_inner, z1 = z % a.nz_inner;
  const unsigned char* in_base =
      (const unsigned char*)a.in + (z0 * a.in_sZ0 + z1 * a.in_sZ1) * ES;
  const unsigned char* w_base = (const unsigned char*)a.w + (z0 * a.w_sZ0 + z1 * a.w_sZ1) * ES;
  const int64_t out_zoff = z0 * a.out_sZ0 + z1 * a.out_sZ1;

  // ---- per-thread staging geometry ----
  const int cchunk = tid & 7;        // which 16-B chunk of the 128-B row
  const int crow = tid >> 3;         // first row handled
  int64_t a_off[CA];                 // element offset of (iy0, ix0, c=0) for the row
  int a_iy0[CA], a_ix0[CA];
  bool a_ok[CA];
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int m = m0 + crow + i * ROWSTEP;
    a_ok[i] = m < k.M;
    const int mm = a_ok[i] ? m : 0;
    int b, oy, ox;
    if (k.in_dense) { b = 0; oy = 0; ox = mm; }
    else { b = mm / HoWo; const int rem = mm - b * HoWo; oy = rem / a.Wo; ox = rem - oy * a.Wo; }
    if (k.in_dense) {  // 1x1, stride 1, dense rows: offset is linear in m, nothing to clip
      a_iy0[i] = 0; a_ix0[i] = 0;
      a_off[i] = (int64_t)mm * a.in_sW;
    } else {
      a_iy0[i] = oy * a.stride - a.pad;
      a_ix0[i] = ox * a.stride - a.pad;
      a_off[i] = (int64_t)b * a.in_sB + (int64_t)a_iy0[i] * a.in_sH + (int64_t)a_ix0[i] * a.in_sW;
    }
  }
  int64_t b_off[CB];
  bool b_ok[CB];
#pragma unroll
  for (int i = 0; i < CB; ++i) {
    const int n = n0 + crow + i * ROWSTEP;
    b_ok[i] = n < a.N;
    b_off[i] = (int64_t)(b_ok[i] ? n : 0) * a.w_sN;
  }

  uint4 ra[CA], rb[CB];
  int tap_r = 0, tap_s = 0, cc = 0;  // position of the NEXT tile to fetch

  auto fetch = [&]() {
    const int64_t tap_off = (int64_t)tap_r * a.in_sH + (int64_t)tap_s * a.in_sW +
                            (int64_t)cc * BKE + cchunk * (16 / ES);
    const int64_t wk = ((int64_t)(tap_r * a.S + tap_s) * a.C) + (int64_t)cc * BKE + cchunk * (16 / ES);
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const bool ok = a_ok[i] && (unsigned)(a_iy0[i] + tap_r) < (unsigned)a.H &&
                      (unsigned)(a_ix0[i] + tap_s) < (unsigned)a.W;
      ra[i] = ok ? *(const uint4*)(in_base + (a_off[i] + tap_off) * ES) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < CB; ++i)
      rb[i] = b_ok[i] ? *(const uint4*)(w_base + (b_off[i] + wk) * ES) : make_uint4(0, 0, 0, 0);
    if (++cc == k.kc) { cc = 0; if (++tap_s == a.S) { tap_s = 0; ++tap_r; } }
  };
  auto stash = [&](int stage) {
    unsigned char* sa = smem + stage * STAGE_BYTES;
    unsigned char* sb = sa + BM * 128;
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int r = crow + i * ROWSTEP;
      *(uint4*)(sa + r * 128 + ((cchunk ^ ((r >> 1) & 7)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int r = crow + i * ROWSTEP;
      *(uint4*)(sb + r * 128 + ((cchunk ^ ((r >> 1) & 7)) << 4)) = rb[i];
    }
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5;
  const int fswz = (frow >> 1) & 7;
  const int a_lds0 = (wm * TM * 32 + frow) * 128;
  const int b_lds0 = BM * 128 + (wn * TN * 32 + frow) * 128;

  fetch();
  stash(0);
  __syncthreads();

  for (int kt = 0; kt < k.KT; ++kt) {
    const bool more = kt + 1 < k.KT;
    if (more) fetch();
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
                __builtin_bit_cast(bf16x8_t, fa[i]), __builtin_bit_cast(bf16x8_t, fb[j]), acc[i][j],
                0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].x),
                                                              __uint_as_float(fb[j].x), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].y),
                                                              __uint_as_float(fb[j].y), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].z),
                                                              __uint_as_float(fb[j].z), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].w),
                                                              __uint_as_float(fb[j].w), acc[i][j], 0, 0, 0);
          }
        }
    }
    if (more) stash((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds column n = lane&31 of each tile, rows (r&3)+8(r>>2)+4*fhalf ----
  float e_bias[TN], e_scale[TN], e_shift[TN];
  bool n_ok[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + frow;
    n_ok[j] = n < a.N;
    const int nn = n_ok[j] ? n : 0;
    e_bias[j] = a.bias ? a.bias[nn] : 0.f;
    e_scale[j] = a.scale ? a.scale[nn] : 1.f;
    e_shift[j] = a.shift ? a.shift[nn] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
      if (m >= k.M) continue;
      int64_t ooff, roff = 0;
      float bscale = 1.f;
      if (k.out_dense && k.res_dense && !a.batch_scale) {
        ooff = (int64_t)m * a.out_sW;
        roff = (int64_t)m * a.res_sW;
      } else {
        const int b = m / HoWo, rem = m - b * HoWo, oy = rem / a.Wo, ox = rem - oy * a.Wo;
        ooff = (int64_t)b * a.out_sB + (int64_t)oy * a.out_sH + (int64_t)ox * a.out_sW;
        roff = (int64_t)b * a.res_sB + (int64_t)oy * a.res_sH + (int64_t)ox * a.res_sW;
        if (a.batch_scale) bscale = a.batch_scale[b];
      }
      ooff += out_zoff;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (!n_ok[j]) continue;
        const int n = n0 + (wn * TN + j) * 32 + frow;
        float v = acc[i][j][r] * a.alpha + e_bias[j];
        if (a.scale) v = v * e_scale[j] + e_shift[j];
        if (a.act == GDL_ACT_RELU) v = fmaxf(v, 0.f);
        else if (a.act == GDL_ACT_GELU) v = gelu_erf(v);
        v *= bscale;
        if (a.resid) v += load_as_f32(a.resid, roff + n, a.resid_dtype);
        store_from_f32(a.out, ooff + n, v, a.out_dtype);
      }
    }
  }
}

template <typename T, int WARPS_M, int WARPS_N, int TM, int TN>
int launch(const KArgs& k, hipStream_t stream) {
  constexpr int BM = WARPS_M * TM * 32, BN = WARPS_N * TN * 32;
  KArgs kk = k;
  kk.tiles_m = (k.M + BM - 1) / BM;
  kk.tiles_n = (k.a.N + BN - 1) / BN;
  const size_t lds = 2 * (BM + BN) * 128;
  auto kern = conv_gemm_kernel<T, WARPS_M, WARPS_N, TM, TN>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dim3 grid(kk.tiles_m * kk.tiles_n, k.a.nz), block(64 * WARPS_M * WARPS_N);
  hipLaunchKernelGGL(kern, grid, block, lds, stream, kk);
  GDL_CHECK_LAUNCH("gdl_conv_gemm");
  return GDL_OK;
}

}  // namespace

extern "C" int gdl_conv_gemm(const gdl_conv_args* ap, gdl_stream_t stream) {
  GDL_CHECK_ARG(ap != nullptr, "gdl_conv_gemm: null args");
  const gdl_conv_args& a = *ap;
  GDL_CHECK_ARG(a.dtype == GDL_F32 || a.dtype == GDL_BF16, "gdl_conv_gemm: bad dtype %d", a.dtype);
  GDL_CHECK_ARG(a.out_dtype == GDL_F32 || a.out_dtype == GDL_BF16, "gdl_conv_gemm: bad out_dtype");
  const int es = (int)gdl_elem_size(a.dtype);
  const int bke = 128 / es, al = 16 / es;
  GDL_CHECK_ARG(a.in && a.w && a.out, "gdl_conv_gemm: null tensor");
  GDL_CHECK_ARG(a.B > 0 && a.H > 0 && a.W > 0 && a.C > 0 && a.N > 0 && a.Ho > 0 && a.Wo > 0,
                "gdl_conv_gemm: non-positive dims");
  GDL_CHECK_ARG(a.R > 0 && a.S > 0 && a.stride > 0 && a.pad >= 0, "gdl_conv_gemm: bad filter");
  GDL_CHECK_ARG(a.C % bke == 0, "gdl_conv_gemm: C=%d must be a multiple of %d", a.C, bke);
  GDL_CHECK_ARG(a.in_sB % al == 0 && a.in_sH % al == 0 && a.in_sW % al == 0 && a.w_sN % al == 0 &&
                    a.in_sZ0 % al == 0 && a.in_sZ1 % al == 0 && a.w_sZ0 % al == 0 && a.w_sZ1 % al == 0,
                "gdl_conv_gemm: strides must keep 16-byte alignment");
  GDL_CHECK_ARG(((uintptr_t)a.in % 16 == 0) && ((uintptr_t)a.w % 16 == 0),
                "gdl_conv_gemm: operand pointers must be 16-byte aligned");
  GDL_CHECK_ARG(a.nz >= 1 && a.nz_inner >= 1, "gdl_conv_gemm: nz/nz_inner must be >= 1");
  GDL_CHECK_ARG((int64_t)a.B * a.Ho * a.Wo < (1ll << 31), "gdl_conv_gemm: M too large");
  KArgs k;
  k.a = a;
  k.M = a.B * a.Ho * a.Wo;
  k.kc = a.C / bke;
  k.KT = a.R * a.S * k.kc;
  // "dense" = the (b,oy,ox) -> offset map is linear in m, so no divisions are needed
  k.in_dense = (a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0 && a.H == a.Ho && a.W == a.Wo &&
                a.in_sH == (int64_t)a.W * a.in_sW && a.in_sB == (int64_t)a.H * a.in_sH);
  k.out_dense = (a.out_sH == (int64_t)a.Wo * a.out_sW && a.out_sB == (int64_t)a.Ho * a.out_sH);
  k.res_dense = !a.resid || (a.res_sH == (int64_t)a.Wo * a.res_sW && a.res_sB == (int64_t)a.Ho * a.res_sH);
  hipStream_t s = (hipStream_t)stream;
  const int variant = gdl_conv_gemm_plan(ap, nullptr);
  if (a.dtype == GDL_BF16) {
    if (variant == 1) return launch<bf16_tag, 2, 2, 2, 2>(k, s);
    return launch<bf16_tag, 2, 2, 1, 1>(k, s);
  }
  if (variant == 1) return launch<float, 2, 2, 2, 2>(k, s);
  return launch<float, 2, 2, 1, 1>(k, s);
}

// Tile selection: 128x128 tiles (variant 1) when there is enough work to fill 256 CUs, else 64x64
// (variant 0).  Also reports the ALGORITHMIC flops of the call (2*M*N*K, no padding counted).
extern "C" int gdl_conv_gemm_plan(const gdl_conv_args* ap, int64_t* flops) {
  if (!ap) return -1;
  const gdl_conv_args& a = *ap;
  const int64_t M = (int64_t)a.B * a.Ho * a.Wo;
  if (flops) *flops = 2 * M * a.N * ((int64_t)a.R * a.S * a.C) * a.nz;
  const int64_t big_tiles = ((M + 127) / 128) * ((a.N + 127) / 128) * a.nz;
  return (big_tiles >= 256 && a.N >= 128) ? 1 : 0;
}
