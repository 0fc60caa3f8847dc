// Implicit-GEMM convolution / linear for gfx950 (CDNA4) on MFMA.
//
//   out[m, n] = epi( alpha * sum_{tap, c} in[pixel(m) + tap, c] * w[n, tap*C + c] )
//
// * NHWC activations, [N][R*S*C] weights: both operands are K-contiguous, so an operand
//   tile is ROWS x 128 bytes of K (64 bf16 or 32 f32) -- im2col never exists in memory.
// * Staging is direct global -> LDS DMA (global_load_lds_dwordx4, 1 KiB = 8 tile rows per
//   wave-instruction, no VGPR round trip, no ds_write).  The 3x3 halo / zero padding / M and N
//   tails need no predication: a lane whose 16-byte chunk is out of range simply points its
//   (per-lane) SOURCE address at a zero page.
// * The DMA writes LDS lane-linearly, so the bank swizzle lives on the source side: LDS slot s
//   of row r receives source chunk s ^ ((r>>1)&7), and fragment reads apply the same involution
//   (ds_read_b128 of chunk c at slot c ^ ((r>>1)&7)): the 4 x 16-lane groups of a fragment fetch
//   hit 16 distinct 16-byte slots -> conflict-free.
// * Two LDS stages; per K-step: wait own DMA (vmcnt 0) -> barrier -> issue next tile's DMA ->
//   MFMA phase of the current tile (DMA in flight underneath).  One barrier per K-step.
// * MFMA: v_mfma_f32_32x32x16_bf16 (bf16) or v_mfma_f32_32x32x2_f32 (exact f32 -- the parity
//   path); a wave owns TM x TN tiles of 32x32, f32 accumulators stay in registers.  K order
//   inside a 32-byte chunk pair is permuted identically for both operands (legal for a
//   reduction), so both dtypes fetch fragments with the same ds_read_b128.
// * blockIdx -> tile mapping is XCD-aware: the blocks of one XCD walk neighbouring N tiles of
//   the same M rows so the activation rows are re-read from that XCD's L2.
#include <type_traits>

#include "gdl_common.h"

namespace {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ uint4 g_zero_page[8];  // 128 bytes of zeros: source of every out-of-range chunk

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

// EXTRA = the training-only epilogue features (aux_out store of the pre-activation, GDL_ACT_MUL_GELU_GRAD);
// they live in separate instantiations so the inference / frozen-encoder kernels keep their register budget.
template <typename T, int WARPS_M, int WARPS_N, int TM, int TN, bool EXTRA>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N) void conv_gemm_kernel(const KArgs k) {
  constexpr int ES = TileTraits<T>::ES;
  constexpr int BKE = TileTraits<T>::BKE;
  constexpr int BM = WARPS_M * TM * 32, BN = WARPS_N * TN * 32;
  constexpr int NW = WARPS_M * WARPS_N;
  constexpr int CA = BM / (8 * NW), CB = BN / (8 * NW);  // DMA instructions per wave per tile
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile/waves mismatch");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE_BYTES = (BM + BN) * 128;  // [A: BM rows | B: BN rows] x 128 B

  const gdl_conv_args& a = k.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;

  const int lid = xcd_remap(blockIdx.x, k.tiles_m * k.tiles_n);
  const int tile_n = lid % k.tiles_n, tile_m = lid / k.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.y;
  const int z0 = z / a.nz_inner, z1 = z % a.nz_inner;
  const unsigned char* in_base = (const unsigned char*)a.in + (z0 * a.in_sZ0 + z1 * a.in_sZ1) * ES;
  const unsigned char* w_base = (const unsigned char*)a.w + (z0 * a.w_sZ0 + z1 * a.w_sZ1) * ES;
  const int64_t out_zoff = z0 * a.out_sZ0 + z1 * a.out_sZ1;
  const unsigned char* zero = (const unsigned char*)g_zero_page;

  // ---- DMA geometry: wave w, instruction i covers tile rows (i*NW + w)*8 .. +7; lane l writes
  //      LDS slot (l & 7) of row +(l >> 3) and therefore fetches source chunk slot ^ swz(row).
  const int lrow = lane >> 3, lslot = lane & 7;
  int64_t a_off[CA];   // byte offset of (iy0, ix0, chunk) for the row, tap (0,0), cc = 0
  int a_iy0[CA], a_ix0[CA];
  bool a_ok[CA];
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int r = (i * NW + wave) * 8 + lrow;
    const int chunk = lslot ^ ((r >> 1) & 7);
    const int m = m0 + r;
    a_ok[i] = m < k.M;
    const int mm = a_ok[i] ? m : 0;
    int64_t off;
    if (k.in_dense) {  // 1x1, stride 1, dense rows: offset is linear in m, nothing to clip
      a_iy0[i] = 0; a_ix0[i] = 0;
      off = (int64_t)mm * a.in_sW;
    } else {
      const int b = mm / HoWo, rem = mm - b * HoWo, oy = rem / a.Wo, ox = rem - oy * a.Wo;
      a_iy0[i] = oy * a.stride - a.pad;
      a_ix0[i] = ox * a.stride - a.pad;
      off = (int64_t)b * a.in_sB + (int64_t)a_iy0[i] * a.in_sH + (int64_t)a_ix0[i] * a.in_sW;
    }
    a_off[i] = (off + chunk * (16 / ES)) * ES;
  }
  const unsigned char* b_ptr[CB];  // weight row + chunk (k = 0); nullptr-free: tails use zero page
  bool b_ok[CB];
#pragma unroll
  for (int i = 0; i < CB; ++i) {
    const int r = (i * NW + wave) * 8 + lrow;
    const int chunk = lslot ^ ((r >> 1) & 7);
    const int n = n0 + r;
    b_ok[i] = n < a.N;
    b_ptr[i] = w_base + ((int64_t)(b_ok[i] ? n : 0) * a.w_sN + chunk * (16 / ES)) * ES;
  }

  int tap_r = 0, tap_s = 0, cc = 0;  // position of the NEXT tile to fetch
  int64_t wk = 0;                    // its byte offset along the weight rows

  auto issue = [&](int stage) {
    unsigned char* sa = smem + stage * STAGE_BYTES;
    unsigned char* sb = sa + BM * 128;
    const int64_t tap_off = ((int64_t)tap_r * a.in_sH + (int64_t)tap_s * a.in_sW + (int64_t)cc * BKE) * ES;
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const bool ok = a_ok[i] && (unsigned)(a_iy0[i] + tap_r) < (unsigned)a.H &&
                      (unsigned)(a_ix0[i] + tap_s) < (unsigned)a.W;
      const unsigned char* src = ok ? in_base + a_off[i] + tap_off : zero;
      dma16_to_lds(src, sa + (i * NW + wave) * 1024);
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const unsigned char* src = b_ok[i] ? b_ptr[i] + wk : zero;
      dma16_to_lds(src, sb + (i * NW + wave) * 1024);
    }
    wk += 128;
    if (++cc == k.kc) { cc = 0; if (++tap_s == a.S) { tap_s = 0; ++tap_r; } }
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

  issue(0);
  for (int kt = 0; kt < k.KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA for tile kt has landed
    __syncthreads();                                   // ... and everyone's; everyone left tile kt-1
    if (kt + 1 < k.KT) issue((kt + 1) & 1);            // DMA of tile kt+1 flies under the MFMAs
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
  // Phase 1 (per 32-row slab i): apply the per-column epilogue in registers and park the slab in a
  // wave-private LDS region as [32 rows][TN*32 cols] in the OUTPUT dtype.  Phase 2: every lane
  // picks up 16 contiguous bytes of one row, applies the per-row terms (DropPath scale, residual)
  // and issues ONE coalesced 16-byte store -- instead of 16*TN scattered 2/4-byte stores.
  __syncthreads();  // all waves are done with the operand stages: LDS is free for the transposes
  auto run = [&](auto oes_c) {
    constexpr int OES = decltype(oes_c)::value;        // output element size
    constexpr int COLS = TN * 32;
    constexpr int ROWB = COLS * OES + 16;              // +16 B pad: the two half-waves hit different banks
    constexpr int CPR = COLS * OES / 16;               // 16-byte chunks per row
    constexpr int PER = 16 / OES;                      // elements per chunk
    constexpr int PASSES = 32 * CPR / 64;
    unsigned char* reg = smem + wave * (32 * ROWB);
    const int prow = lane / CPR, pchunk = lane % CPR;
    const bool vec_ok = ((uintptr_t)a.out % 16 == 0) && ((uintptr_t)a.aux_out % 16 == 0) && (a.out_sW % PER == 0) && (a.out_sH % PER == 0) &&
                        (a.out_sB % PER == 0) && (out_zoff % PER == 0);
    const bool mulgrad = EXTRA && a.act == GDL_ACT_MUL_GELU_GRAD;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      // pass 0 (only with aux_out): acc*alpha + bias, before scale/shift/act; pass 1: the final output
#pragma unroll
      for (int pass = EXTRA ? 0 : 1; pass < 2; ++pass) {
        const bool is_aux = EXTRA && pass == 0;
        if (is_aux && !a.aux_out) continue;
        unsigned char* dst_base = (unsigned char*)(is_aux ? a.aux_out : a.out);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[i][j][r] * a.alpha + e_bias[j];
            if (!is_aux) {
              if (a.scale) v = v * e_scale[j] + e_shift[j];
              if (a.act == GDL_ACT_RELU) v = fmaxf(v, 0.f);
              else if (a.act == GDL_ACT_GELU) v = gelu_erf(v);
            }
            const int row = (r & 3) + 8 * (r >> 2) + 4 * fhalf;
            unsigned char* q = reg + row * ROWB + (j * 32 + frow) * OES;
            if constexpr (OES == 2) *(uint16_t*)q = f32_to_bf16(v);
            else *(float*)q = v;
          }
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          const int row = ps * (64 / CPR) + prow;
          const int m = m0 + (wm * TM + i) * 32 + row;
          const int n = n0 + wn * COLS + pchunk * PER;
          if (m >= k.M || n >= a.N) continue;
          const uint4 raw = *(const uint4*)(reg + row * ROWB + pchunk * 16);
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
          ooff += out_zoff + n;
          roff += n;
          const bool full = n + PER <= a.N;
          const bool plain = is_aux || (!a.resid && !a.batch_scale);
          if (full && vec_ok && plain) {
            *(uint4*)(dst_base + ooff * OES) = raw;
            continue;
          }
          float v[PER];
          if constexpr (OES == 2) {
            const uint32_t w4[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(w4[e] << 16); v[2 * e + 1] = __uint_as_float(w4[e] & 0xffff0000u); }
          } else {
            v[0] = __uint_as_float(raw.x); v[1] = __uint_as_float(raw.y); v[2] = __uint_as_float(raw.z); v[3] = __uint_as_float(raw.w);
          }
          if (!plain) {
#pragma unroll
            for (int e = 0; e < PER; ++e) {
              v[e] *= bscale;
              if (a.resid && n + e < a.N) {
                const float rv = load_as_f32(a.resid, roff + e, a.resid_dtype);
                v[e] = mulgrad ? v[e] * gelu_erf_grad(rv) : v[e] + rv;
              }
              if (a.act == GDL_ACT_RESID_RELU) v[e] = fmaxf(v[e], 0.f);
            }
          }
          if (full && vec_ok) {
            if constexpr (OES == 2)
              *(uint4*)(dst_base + ooff * 2) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]),
                                                          pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
            else
              *(float4*)((float*)dst_base + ooff) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int e = 0; e < PER; ++e)
              if (n + e < a.N) store_from_f32(dst_base, ooff + e, v[e], a.out_dtype);
          }
        }
      }
    }
  };
  if (a.out_dtype == GDL_BF16) run(std::integral_constant<int, 2>{});
  else run(std::integral_constant<int, 4>{});
}

template <typename T, int WARPS_M, int WARPS_N, int TM, int TN, bool EXTRA>
int launch_x(const KArgs& k, hipStream_t stream) {
  constexpr int BM = WARPS_M * TM * 32, BN = WARPS_N * TN * 32;
  KArgs kk = k;
  kk.tiles_m = (k.M + BM - 1) / BM;
  kk.tiles_n = (k.a.N + BN - 1) / BN;
  const size_t lds = 2 * (BM + BN) * 128;
  auto kern = conv_gemm_kernel<T, WARPS_M, WARPS_N, TM, TN, EXTRA>;
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

template <typename T, int WARPS_M, int WARPS_N, int TM, int TN>
int launch(const KArgs& k, hipStream_t stream) {
  if (k.a.aux_out || k.a.act == GDL_ACT_MUL_GELU_GRAD) return launch_x<T, WARPS_M, WARPS_N, TM, TN, true>(k, stream);
  return launch_x<T, WARPS_M, WARPS_N, TM, TN, false>(k, stream);
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
  GDL_CHECK_ARG(a.act >= GDL_ACT_NONE && a.act <= GDL_ACT_RESID_RELU, "gdl_conv_gemm: bad act %d", a.act);
  GDL_CHECK_ARG(a.act != GDL_ACT_RESID_RELU || a.resid, "gdl_conv_gemm: GDL_ACT_RESID_RELU needs `resid`");
  GDL_CHECK_ARG(a.act != GDL_ACT_MUL_GELU_GRAD || (a.resid && !a.aux_out),
                "gdl_conv_gemm: GDL_ACT_MUL_GELU_GRAD takes the pre-activation tensor in `resid`");
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
    if (variant == 2) return launch_x<bf16_tag, 2, 4, 4, 2, false>(k, s);
    if (variant == 1) return launch<bf16_tag, 2, 2, 2, 2>(k, s);
    return launch<bf16_tag, 2, 2, 1, 1>(k, s);
  }
  if (variant == 2) return launch_x<float, 2, 4, 4, 2, false>(k, s);
  if (variant == 1) return launch<float, 2, 2, 2, 2>(k, s);
  return launch<float, 2, 2, 1, 1>(k, s);
}

// Tile selection by available parallelism (256 CUs): 256x256 tiles / 8 waves (variant 2) when
// they still give >= 512 blocks, 128x128 / 4 waves (variant 1) when that gives >= 256 blocks,
// else 64x64 (variant 0).  Also reports the ALGORITHMIC flops of the call (2*M*N*K, no padding).
static int g_forced_variant = -1;
extern "C" void gdl_debug_force_conv_variant(int v) { g_forced_variant = v; }  // tuning hook (-1 = auto)

extern "C" int gdl_conv_gemm_plan(const gdl_conv_args* ap, int64_t* flops) {
  if (!ap) return -1;
  const gdl_conv_args& a = *ap;
  const int64_t M = (int64_t)a.B * a.Ho * a.Wo;
  if (flops) *flops = 2 * M * a.N * ((int64_t)a.R * a.S * a.C) * a.nz;
  if (g_forced_variant >= 0 && !(g_forced_variant == 2 && (a.aux_out || a.act == GDL_ACT_MUL_GELU_GRAD)))
    return g_forced_variant;
  const int64_t t256 = ((M + 255) / 256) * ((a.N + 255) / 256) * a.nz;
  const int64_t t128 = ((M + 127) / 128) * ((a.N + 127) / 128) * a.nz;
  const int64_t ksteps = (int64_t)a.R * a.S * a.C * (int64_t)gdl_elem_size(a.dtype) / 128;
  // measured (tools/bench_conv.py): 256^2 tiles win whenever K is deep, even at ~1 block per CU;
  // for shallow K (ViT linears, 12 K-steps) the 128^2 tile's shorter prologue/epilogue wins
  // the training-only epilogue (aux_out / GELU-grad) does not fit the 256^2 tile's register budget
  const bool extra = a.aux_out != nullptr || a.act == GDL_ACT_MUL_GELU_GRAD;
  if (!extra && a.N % 256 == 0 && (t256 >= 512 || (t256 >= 256 && ksteps >= 32))) return 2;
  if (t128 >= 256 && a.N >= 128) return 1;
  return 0;
}
