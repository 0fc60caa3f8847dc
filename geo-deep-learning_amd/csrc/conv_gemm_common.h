// Shared pieces of the implicit-GEMM kernels (conv_gemm.hip, conv3x3_sf.hip): launch arguments, tile traits, the
// XCD-aware tile order and the register epilogue.
#pragma once
#include <type_traits>

#include "gdl_common.h"

namespace gdlconv {

struct KArgs {
  gdl_conv_args a;
  int M;        // B*Ho*Wo
  int kc;       // ceil(C / BKE)
  int c_tail;   // channels in the last K chunk when C is not a multiple of BKE (else 0): the CT kernels zero-fill
  int KT;       // R*S*kc
  int tiles_m, tiles_n;
  int n_group;  // tile order: N tiles are walked in groups of n_group (0 = all): [group][tile_m][tile_n in group], see tile_order()
  int in_dense, out_dense, res_dense;
  unsigned in_span, w_span;   // bytes addressed from the (z-offset) operand base: buffer num_records
  unsigned out_span;          // conv_gemm_w4p.hip only: the same for the output (stores of rows past M are range-checked away)
  int tap_inner;              // K order: 1 = channel chunk outer / filter tap inner (default), 0 = tap outer
  int dbg;                    // tuning experiments only: 1 = no DMA after the first tile, 2 = no MFMA
  int epi_v2;                 // A/B hook (gdl_debug_set_conv_epilogue): 0 = the round-2 epilogue (conv_epilogue_rows)
  unsigned long long* probe;  // tuning only (12288 u64): per block {shader cycles, 100 MHz ticks} of the K loop, K loop + epilogue cycles, {start, end} ticks
};

template <typename T> struct TileTraits;
template <> struct TileTraits<float> { static constexpr int ES = 4; static constexpr int BKE = 32; };
template <> struct TileTraits<bf16_tag> { static constexpr int ES = 2; static constexpr int BKE = 64; };

// the transcendental activations are kept out of line: the epilogue is unrolled 32-fold (static accumulator
// indexing) and would otherwise exceed the unroller's size budget
__device__ __noinline__ float4 gelu4(float4 v) {
  return make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
}
__device__ __noinline__ float4 mul_gelu_grad4(float4 v, float4 u) {
  return make_float4(v.x * gelu_erf_grad(u.x), v.y * gelu_erf_grad(u.y), v.z * gelu_erf_grad(u.z),
                     v.w * gelu_erf_grad(u.w));
}

// bf16 compute path: erf by Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, far below the bf16 rounding of the result);
// one v_rcp + one v_exp instead of the ~50-instruction branchy erff.  The exact-f32 (parity) path keeps erff.
struct GeluTerms { float half_pe, e; };   // 0.5 * poly(t) * exp(-x^2/2) and exp(-x^2/2)
__device__ __forceinline__ GeluTerms gelu_terms_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);      // the polynomial's coefficients carry the factor 0.5
  p = fmaf(p, t, 0.5f * 1.421413741f);
  p = fmaf(p, t, 0.5f * -0.284496736f);
  p = fmaf(p, t, 0.5f * 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * z * z);
  return {p * t * e, e};
}
__device__ __forceinline__ float gelu_fast(float x) {
  const float h = gelu_terms_fast(x).half_pe;          // 0.5 * erfc(|x|/sqrt2)
  // x * (1 - h) for x >= 0, x * h below: max(x, 0) - |x| * h, one v_max + one v_fma (12 full-rate + 2 quarter-rate instructions
  // per element in all; this function is 12 k of the 50 k cycles of a ViT fc1 tile)
  return __builtin_fmaf(-fabsf(x), h, fmaxf(x, 0.f));
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
  const GeluTerms g = gelu_terms_fast(x);
  return (x >= 0.f ? 1.0f - g.half_pe : g.half_pe) + x * 0.3989422804014327f * g.e;
}
__device__ __forceinline__ float4 gelu4_fast(float4 v) {
  return make_float4(gelu_fast(v.x), gelu_fast(v.y), gelu_fast(v.z), gelu_fast(v.w));
}
__device__ __forceinline__ float4 mul_gelu_grad4_fast(float4 v, float4 u) {
  return make_float4(v.x * gelu_grad_fast(u.x), v.y * gelu_grad_fast(u.y), v.z * gelu_grad_fast(u.z),
                     v.w * gelu_grad_fast(u.w));
}

__device__ __forceinline__ int xcd_remap(int id, int n) {
  // bijective "XCD-major" remap (cdna guide T1): hardware places block id on XCD id % 8.
  const int q = n >> 3, r = n & 7, xcd = id & 7, idx = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}


// Tile order.  lid walks the N tiles of one M row innermost (the blocks of an XCD then share the activation rows through its
// L2) -- but only n_group of them: when the whole weight matrix does not fit the 4 MiB L2 beside the streaming activations
// (ViT fc1: 3072 x 768 bf16 = 4.7 MB), every M row would pull every weight panel in again from the Infinity Cache.  With
// groups, an XCD works through ALL M rows against one L2-resident slice of the weights before it moves to the next slice;
// the activations are then read once per group instead of once.
__device__ __forceinline__ void tile_order(const KArgs& k, int lid, int& tile_m, int& tile_n) {
  const int ng = k.n_group;
  if (ng <= 0 || ng >= k.tiles_n) { tile_n = lid % k.tiles_n; tile_m = lid / k.tiles_n; return; }
  const int per_group = k.tiles_m * ng;
  const int g = lid / per_group, idx = lid - g * per_group;
  const int width = min(ng, k.tiles_n - g * ng);       // the last group may be narrower
  tile_m = idx / width;
  tile_n = g * ng + idx - tile_m * width;
}
// host side: N tiles per group for bm x bn tiles of which an XCD runs `conc` at a time (0 = no grouping)
int conv_n_group(const gdl_conv_args& a, int bm, int bn, int conc);

// 3x3 shared-staging kernel (conv3x3_sf.hip)
bool conv3x3_sf_applicable(const gdl_conv_args& a);
int conv3x3_sf_launch(const KArgs& k, hipStream_t stream);
// dual-resident 256 x 128 tile, two workgroups per CU (conv_gemm_dual.hip): short-K bf16 layers
bool conv_gemm_dual_applicable(const gdl_conv_args& a);
int conv_gemm_dual_launch(const KArgs& k, hipStream_t stream);
// 256 x 256 tile, one wave per SIMD, every non-MFMA instruction in an MFMA shadow (conv_gemm_w4.hip)
bool conv_gemm_w4_applicable(const gdl_conv_args& a);
int conv_gemm_w4_launch(const KArgs& k, hipStream_t stream);
// persistent 256 x 256 ping-pong tile for dense 1x1 bf16 layers: next tile's first stage in flight under the epilogue (conv_gemm_persist.hip)
bool conv_gemm_persist_applicable(const gdl_conv_args& a);
int conv_gemm_persist_launch(const KArgs& k, hipStream_t stream);
// persistent 256 x 256 tile, one wave per SIMD, finished tile parked in registers and stored from the next tile's MFMA shadows
// (conv_gemm_w4p.hip): dense 1x1 bf16 layers with bf16 outputs and at least 12 K-steps
bool conv_gemm_w4p_applicable(const gdl_conv_args& a);
int conv_gemm_w4p_launch(const KArgs& k, hipStream_t stream);
// direct 3x3 kernel for C in {8,16,32} on large dense maps, outputs in 32-channel slices (conv3x3_narrow.hip)
bool conv3x3_narrow_applicable(const gdl_conv_args& a);
int conv3x3_narrow_launch(const KArgs& k, hipStream_t stream);


// Coalesced epilogue through a wave-private LDS transpose (see conv_epilogue).  Returns false (nothing done) when the
// call does not qualify; the decision depends on kernel arguments only, i.e. it is uniform over the workgroup.
// `lds` = 8 KiB transpose buffer, `ldc` = 768 B for the wave's per-channel constants (bias | scale | shift of its 64
// channels): they are fetched from global memory ONCE per tile, every pass re-reads them from the LDS.
template <int TM, bool EXTRA>
__device__ __forceinline__ bool conv_epilogue_rows(const KArgs& k, f32x16_t (&acc)[TM][2], int m0, int n0, int wm,
                                                   int wn, int lane, int64_t out_zoff, unsigned char* lds,
                                                   unsigned char* ldc) {
  const gdl_conv_args& a = k.a;
  const bool out_bf16 = a.out_dtype == GDL_BF16;
  const bool mulgrad = EXTRA && a.act == GDL_ACT_MUL_GELU_GRAD;
  const bool plain = !a.resid && !a.batch_scale;
  const bool fast = a.dtype == GDL_BF16;                 // bf16 compute path: fast erf
  const int HoWo = a.Ho * a.Wo;
  const bool vec_ok = ((uintptr_t)a.out % 16 == 0) && ((uintptr_t)a.aux_out % 16 == 0) && (a.out_sW % 8 == 0) &&
                      (out_zoff % 8 == 0) && (a.N % 16 == 0);
  const bool res_ok = !a.resid || (((uintptr_t)a.resid % 16 == 0) && (a.res_sW % 4 == 0));
  const bool vecs_ok = ((uintptr_t)a.bias % 16 == 0) && ((uintptr_t)a.scale % 16 == 0) && ((uintptr_t)a.shift % 16 == 0);
  if (!(k.out_dense && k.res_dense && vec_ok && res_ok && vecs_ok && (!a.batch_scale || HoWo >= 32))) return false;
  const int frow = lane & 31, fhalf = lane >> 5;
  const int n_w = n0 + wn * 64;                          // first channel of the wave's 64-channel strip
  const bool packed = plain && out_bf16;                 // LDS holds bf16 rows of 128 B; otherwise f32 rows of 256 B
  // ---- per-channel constants of the strip -> LDS (lanes 0..15: one float4 of each array)
  if (lane < 16) {
    const int n = n_w + 4 * lane;
    const bool in = n < a.N;                             // N % 16 == 0: a group of 4 is all in or all out
    // (explicit branches: a `cond ? *p : constant` select makes hipcc pick between POINTERS, one of them to a scratch copy)
    float4 vb = make_float4(0.f, 0.f, 0.f, 0.f), vs = make_float4(1.f, 1.f, 1.f, 1.f), vh = vb;
    if (a.bias && in) vb = *(const float4*)(a.bias + n);
    if (a.scale && in) vs = *(const float4*)(a.scale + n);
    if (a.scale && a.shift && in) vh = *(const float4*)(a.shift + n);
    *(float4*)(ldc + lane * 16) = vb;
    if (a.scale) {
      *(float4*)(ldc + 256 + lane * 16) = vs;
      *(float4*)(ldc + 512 + lane * 16) = vh;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  // per-channel part of one 4-channel group at strip offset c: v = act(scale * (alpha * acc + bias) + shift),
  // pre = alpha * acc + bias
  auto channel_terms = [&](const f32x16_t& cacc, int g, int c, float (&v)[4], float (&pre)[4]) {
    const float4 b4 = *(const float4*)(ldc + c * 4);
    const float b[4] = {b4.x, b4.y, b4.z, b4.w};
    float s[4] = {1.f, 1.f, 1.f, 1.f}, h[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.scale) {
      const float4 s4 = *(const float4*)(ldc + 256 + c * 4), h4 = *(const float4*)(ldc + 512 + c * 4);
      s[0] = s4.x; s[1] = s4.y; s[2] = s4.z; s[3] = s4.w;
      h[0] = h4.x; h[1] = h4.y; h[2] = h4.z; h[3] = h4.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = cacc[4 * g + e] * a.alpha + b[e];
      pre[e] = x;
      if (a.scale) x = x * s[e] + h[e];
      if (a.act == GDL_ACT_RELU) x = fmaxf(x, 0.f);
      v[e] = x;
    }
    if (a.act == GDL_ACT_GELU) {
      const float4 t = fast ? gelu4_fast(make_float4(v[0], v[1], v[2], v[3])) : gelu4(make_float4(v[0], v[1], v[2], v[3]));
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
  };
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int mp = m0 + (wm * TM + i) * 32;              // first row of this pass (wave-uniform)
    if (mp >= k.M) break;
    // ---- (0) per-row operands of this pass, requested before anything is stored: the residual as coalesced row
    // segments (lane = (row (lane >> 4) + 4t, 16-byte slot lane & 15)) and the (at most two) DropPath scales
    float rv[8][4];
    float rs0 = 1.f, rs1 = 1.f;
    int rem0 = 0;
    if (!packed && !plain) {
      if (a.batch_scale) {
        const int b0 = mp / HoWo;                        // HoWo >= 32: a pass spans at most two samples
        rem0 = mp - b0 * HoWo;
        rs0 = a.batch_scale[b0];
        rs1 = b0 + 1 < a.B ? a.batch_scale[b0 + 1] : 0.f;
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int m = mp + (lane >> 4) + 4 * t, n = n_w + 4 * (lane & 15);
        rv[t][0] = rv[t][1] = rv[t][2] = rv[t][3] = 0.f;
        if (a.resid && m < k.M && n < a.N) {
          const int64_t ro = (int64_t)m * a.res_sW + n;
          if (a.resid_dtype == GDL_BF16) {
            const uint2 u = *(const uint2*)((const uint16_t*)a.resid + ro);
            rv[t][0] = __uint_as_float(u.x << 16); rv[t][1] = __uint_as_float(u.x & 0xffff0000u);
            rv[t][2] = __uint_as_float(u.y << 16); rv[t][3] = __uint_as_float(u.y & 0xffff0000u);
          } else {
            const float4 u = *(const float4*)((const float*)a.resid + ro);
            rv[t][0] = u.x; rv[t][1] = u.y; rv[t][2] = u.z; rv[t][3] = u.w;
          }
        }
      }
    }
    // ---- (1) MFMA layout -> LDS.  `which` = 0: the output values, 1: the pre-activation copy (aux_out)
    auto to_lds = [&](int which) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (packed) {
#pragma unroll
          for (int p2 = 0; p2 < 2; ++p2) {
            float va[4], vb[4], pa[4], pb[4];
            channel_terms(acc[i][j], 2 * p2, 32 * j + 16 * p2 + 4 * fhalf, va, pa);
            channel_terms(acc[i][j], 2 * p2 + 1, 32 * j + 16 * p2 + 8 + 4 * fhalf, vb, pb);
            const float (&xa)[4] = which ? pa : va;
            const float (&xb)[4] = which ? pb : vb;
            unsigned a0 = pack_bf16x2(xa[0], xa[1]), a1 = pack_bf16x2(xa[2], xa[3]);
            unsigned b0 = pack_bf16x2(xb[0], xb[1]), b1 = pack_bf16x2(xb[2], xb[3]);
            // lane pair (l, l+32) trades one 8-byte piece: low lanes end with channels [16p2, 16p2+8), high lanes
            // with [16p2+8, 16p2+16) of their row
            const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            const int slot = 4 * j + 2 * p2 + fhalf;     // 16-byte slot of the 128-byte row
            *(uint4*)(lds + frow * 128 + ((slot ^ (frow & 7)) << 4)) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
          }
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float v[4], pre[4];
            channel_terms(acc[i][j], g, 32 * j + 8 * g + 4 * fhalf, v, pre);
            const float (&x)[4] = which ? pre : v;
            const int slot = 8 * j + 2 * g + fhalf;      // 16-byte slot of the 256-byte row
            *(float4*)(lds + frow * 256 + ((slot ^ (frow & 7)) << 4)) = make_float4(x[0], x[1], x[2], x[3]);
          }
        }
      }
    };
    // ---- (2) LDS -> full row segments in global memory
    auto from_lds = [&](void* base, bool row_terms) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's LDS writes are done (wave-private region)
      __builtin_amdgcn_wave_barrier();
      if (packed) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int rr = (lane >> 3) + 8 * t, slot = lane & 7;
          const uint4 v = *(const uint4*)(lds + rr * 128 + ((slot ^ (rr & 7)) << 4));
          const int m = mp + rr, n = n_w + 8 * slot;
          if (m < k.M && n < a.N) *(uint4*)((uint16_t*)base + (int64_t)m * a.out_sW + out_zoff + n) = v;
        }
      } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int rr = (lane >> 4) + 4 * t, slot = lane & 15;
          const float4 q = *(const float4*)(lds + rr * 256 + ((slot ^ (rr & 7)) << 4));
          float v[4] = {q.x, q.y, q.z, q.w};
          const int m = mp + rr, n = n_w + 4 * slot;
          if (m >= k.M || n >= a.N) continue;
          if (row_terms && !plain) {
            if (a.batch_scale) {
              const float rs = rem0 + rr >= HoWo ? rs1 : rs0;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= rs;
            }
            if (EXTRA && mulgrad) {
              const float4 u = make_float4(rv[t][0], rv[t][1], rv[t][2], rv[t][3]);
              const float4 x = fast ? mul_gelu_grad4_fast(make_float4(v[0], v[1], v[2], v[3]), u)
                                    : mul_gelu_grad4(make_float4(v[0], v[1], v[2], v[3]), u);
              v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float x = v[e] + rv[t][e];
                if (a.act == GDL_ACT_RESID_RELU) x = fmaxf(x, 0.f);
                v[e] = x;
              }
            }
          }
          const int64_t off = (int64_t)m * a.out_sW + out_zoff + n;
          if (out_bf16) *(uint2*)((uint16_t*)base + off) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
          else *(float4*)((float*)base + off) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    };
    if (EXTRA && a.aux_out) {
      to_lds(1);
      from_lds(a.aux_out, false);
    }
    to_lds(0);
    from_lds(a.out, true);
  }
  return true;
}

// Round 3: the same coalesced epilogue with ALL element-wise work moved behind the transpose.  In the store layout a lane
// always owns the same CH consecutive channels (8 for bf16 outputs, 4 for f32), so the per-channel constants are CH registers
// loaded once per tile instead of LDS reads per accumulator group; the MFMA-layout side is a plain dump of the raw f32
// accumulators (eight ds_write_b128 per pass, no VALU); the residual rows of pass i + 1 are requested before pass i is
// stored (they cost one exposed memory latency per PASS before: 57 k cycles per proj tile against 36 k for its K loop).
// Per element the arithmetic and its order are exactly conv_epilogue_rows' (x = alpha * acc + bias; x * scale + shift;
// ReLU / GELU; * DropPath scale; + residual; ReLU) -- results are bit-identical, tests/test_hip_ops.py holds it to that.
// Not for the training-only extras (aux_out, GELU-gradient multiply): those keep conv_epilogue_rows.

// EF selects how the element-wise terms are decided: 0 = at run time from the arguments (any combination); otherwise the
// combination is a COMPILE-TIME constant -- bit 0 set, bit 1 = per-channel scale/shift, bit 2 = DropPath scale, bits 3-4 =
// activation (0 none, 1 ReLU, 2 GELU) -- and the caller guarantees the arguments match it.  With run-time flags hipcc turns
// every `if (a.scale)` / `if (a.act == ...)` into v_cndmask selects per ELEMENT (about 75 VALU instructions per 8 outputs of
// which a bias-only layer needs 12), and a wave's epilogue is a serial chain of such groups: measured 10.8 k cycles per 256^2
// qkv tile of which 12.2 k remained with the global stores removed (profiles/r03b_gemm_epilogue_parts.txt).  Arithmetic and
// its order are the same in every instantiation: results are bit-identical to EF = 0.
#ifndef GDL_EPI_MASK
#define GDL_EPI_MASK 15
#endif
#define GDL_EPI_ON(bit) ((GDL_EPI_MASK & (bit)) != 0)
constexpr int EPI_RUNTIME = 0, EPI_FIXED = 1, EPI_SCALE = 2, EPI_DROPPATH = 4, EPI_RELU = 8, EPI_GELU = 16;
// EPI_STATS (bf16 outputs, 8-channel layout): the wave also writes the sum and the sum of squares of its (bf16-rounded) outputs
// per channel -- one partial row [2][N] per 32 * TM output pixels in a.stats_partial: train-mode BatchNorm statistics
// without a pass over the convolution's output (gdl_conv_gemm_stats_rows says when)
constexpr int EPI_STATS = 32;

template <int TM, int CH, bool RESID, int EF = EPI_RUNTIME>
__device__ __forceinline__ void conv_epilogue_rows2_impl(const KArgs& k, f32x16_t (&acc)[TM][2], int m0, int n0, int wm, int wn,
                                                        int lane, int64_t out_zoff, unsigned char* lds) {
  // the two fused multiply-adds of the element-wise chain are written out; nothing else may be contracted: with compile-time
  // terms "DropPath scale, then + residual" are adjacent and would fuse, with run-time terms (a select in between) they do not
#pragma clang fp contract(off)
  constexpr int NT = CH == 8 ? 4 : 8;                    // row groups per pass: a pass is 32 rows x 64 channels
  constexpr int LPR = 64 / CH;                           // lanes per row
  const gdl_conv_args& a = k.a;
  constexpr bool RT = EF == EPI_RUNTIME;
  const bool has_scale = RT ? a.scale != nullptr : (EF & EPI_SCALE) != 0;
  const bool has_bs = RT ? a.batch_scale != nullptr : (EF & EPI_DROPPATH) != 0;
  const int act = RT ? a.act : (EF & EPI_RELU) ? GDL_ACT_RELU : (EF & EPI_GELU) ? GDL_ACT_GELU : GDL_ACT_NONE;
  const bool plain = !RESID && !has_bs;
  const bool fast = a.dtype == GDL_BF16;
  const int HoWo = a.Ho * a.Wo;
  const int frow = lane & 31, fhalf = lane >> 5;
  const int srow = lane / LPR, slot = lane % LPR;        // store layout: row srow + (64 / LPR) t, channels CH * slot ..
  // FULL wave tiles only (the caller checks the wave's rows and channels against M and N): no per-lane range conditions, so every
  // load and store below is unconditional straight-line code and hipcc's waitcnt pass counts them exactly (with `continue`
  // / exec-masked memory operations in the loop it fell back to vmcnt(0) in front of every store: the residual prefetch
  // was serialised again)
  const int n_l = n0 + wn * 64 + CH * slot;
  float cb[CH], cs[CH], chh[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) { cb[e] = 0.f; cs[e] = 1.f; chh[e] = 0.f; }
  {
#pragma unroll
    for (int q = 0; q < CH / 4; ++q) {
      if (a.bias) { const float4 t = *(const float4*)(a.bias + n_l + 4 * q); cb[4 * q] = t.x; cb[4 * q + 1] = t.y; cb[4 * q + 2] = t.z; cb[4 * q + 3] = t.w; }
      if (has_scale) { const float4 t = *(const float4*)(a.scale + n_l + 4 * q); cs[4 * q] = t.x; cs[4 * q + 1] = t.y; cs[4 * q + 2] = t.z; cs[4 * q + 3] = t.w; }
      if (has_scale && a.shift) { const float4 t = *(const float4*)(a.shift + n_l + 4 * q); chh[4 * q] = t.x; chh[4 * q + 1] = t.y; chh[4 * q + 2] = t.z; chh[4 * q + 3] = t.w; }
    }
  }
  // residual rows of one pass: NT pieces of CH channels (f32 or bf16 in memory) as f32 registers
  // residual rows are kept RAW (as loaded: four words per row group = 4 f32, or 8 bf16 for the 8-channel layout) and
  // converted at their use, so that nothing between the request and the use needs the data (the request is a prefetch).
  // vmcnt counts loads and stores in ONE in-order queue: a wait for loads that were issued behind stores also waits for
  // those stores' acknowledgements (measured: 60 k cycles per proj tile with the loads at the start of every pass).
  // (the residual has the OUTPUT's dtype here -- f32 stream + f32 residual, bf16 map + bf16 residual -- so a row group is
  // always one 16-byte load; other combinations keep conv_epilogue_rows)
  constexpr int RW = 4;
  // addresses = a buffer descriptor on the wave tile's first element (uniform) + ONE per-lane 32-bit offset + a uniform
  // (scalar) row offset per access: written as per-lane 64-bit `m * stride + n` products -- or as 64-bit pointer sums --
  // the compiler precomputed the 32 + 32 addresses of a tile ahead of the passes (128 registers) and spilled
  constexpr int OES = CH == 8 ? 2 : 4;                   // bytes per output element; the residual has the output's dtype here
  const int m_w = m0 + wm * TM * 32;                     // first row of the wave tile (uniform)
  const __amdgpu_buffer_rsrc_t res_rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const unsigned char*)a.resid + ((int64_t)m_w * a.res_sW + n0 + wn * 64) * OES), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((unsigned char*)a.out + ((int64_t)m_w * a.out_sW + out_zoff + n0 + wn * 64) * OES), 0, 0x7fffffff, 0x00020000);
  const unsigned res_lo = (unsigned)((srow * (int)a.res_sW + CH * slot) * OES);
  const unsigned out_lo = (unsigned)((srow * (int)a.out_sW + CH * slot) * OES);
  typedef __attribute__((ext_vector_type(4))) unsigned epi_u4;
  auto load_resid_t = [&](int mp, int t, uint32_t (&rw)[NT][RW]) {
    const unsigned ru = (unsigned)((mp - m_w + (64 / LPR) * t) * (int)a.res_sW * OES);   // uniform
    const epi_u4 u = __builtin_amdgcn_raw_buffer_load_b128(res_rs, res_lo, ru, 0);
    rw[t][0] = u.x; rw[t][1] = u.y; rw[t][2] = u.z; rw[t][3] = u.w;
  };
  auto resid_value = [&](const uint32_t (&w)[RW], int e) -> float {
    if constexpr (CH == 8) return __uint_as_float((e & 1) ? (w[e >> 1] & 0xffff0000u) : (w[e >> 1] << 16));
    else return __uint_as_float(w[e]);
  };
  // DropPath scales, requested FIRST: these are vector loads (the compiler cannot prove the array read-only), and behind the
  // residual prefetch in vmcnt's in-order queue the wait for them would drain the prefetch.  HoWo >= 256 (checked by the
  // caller): the tile's (at most 256) rows span at most two samples.
  float rs0 = 1.f, rs1 = 1.f;
  int rem0 = 0;                                          // row m belongs to sample b0 + 1 when m - m0 + rem0 >= HoWo
  if (has_bs) {
    const int b0 = m0 / HoWo;
    rem0 = m0 - b0 * HoWo;
    rs0 = a.batch_scale[b0];
    rs1 = b0 + 1 < a.B ? a.batch_scale[b0 + 1] : 0.f;
  }
  // ONE buffer: row group t of the next pass is requested right after row group t of this pass has been consumed and BEFORE
  // its store is issued, so a wait for it never has to cover a store issued behind it
  float st1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, st2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // EPI_STATS
  uint32_t rva[NT][RW];
  if constexpr (RESID) {
#pragma unroll
    for (int t = 0; t < NT; ++t) load_resid_t(m0 + wm * TM * 32, t, rva);
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int mp = m0 + (wm * TM + i) * 32;              // first row of this pass (wave-uniform)
    // ---- (1) raw accumulators, MFMA layout -> LDS (f32 rows of 256 B, 16-byte slots swizzled by row)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int sl = 8 * j + 2 * g + fhalf;
        *(float4*)(lds + frow * 256 + ((sl ^ (frow & 7)) << 4)) =
            make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- (2) LDS -> registers in the store layout -> element-wise terms -> full row segments in global memory
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int rr = srow + (64 / LPR) * t;
      float v[CH];
#pragma unroll
      for (int q = 0; q < CH / 4; ++q) {
        const int sl = (CH / 4) * slot + q;
        const float4 x = *(const float4*)(lds + rr * 256 + ((sl ^ (rr & 7)) << 4));
        v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
      }
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        float x = __builtin_fmaf(v[e], a.alpha, cb[e]);
        if (has_scale) x = __builtin_fmaf(x, cs[e], chh[e]);
        if (act == GDL_ACT_RELU) x = fmaxf(x, 0.f);
        v[e] = x;
      }
      if (act == GDL_ACT_GELU) {
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
          const float4 x4 = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
          const float4 y4 = fast ? gelu4_fast(x4) : gelu4(x4);
          v[4 * q] = y4.x; v[4 * q + 1] = y4.y; v[4 * q + 2] = y4.z; v[4 * q + 3] = y4.w;
        }
      }
      if (!plain) {
        if (has_bs) {
          const float rs = rem0 + (wm * TM + i) * 32 + rr >= HoWo ? rs1 : rs0;
#pragma unroll
          for (int e = 0; e < CH; ++e) v[e] *= rs;
        }
        if constexpr (RESID) {
#pragma unroll
          for (int e = 0; e < CH; ++e) {
            float x = v[e] + resid_value(rva[t], e);
            if (act == GDL_ACT_RESID_RELU) x = fmaxf(x, 0.f);
            v[e] = x;
          }
          if (i + 1 < TM) load_resid_t(mp + 32, t, rva);
        }
      }
      const unsigned ou = (unsigned)((i * 32 + (64 / LPR) * t) * (int)a.out_sW * OES);    // uniform (scalar): rows i*32 + step*t of the wave tile
      epi_u4 o;
      if constexpr (CH == 8) {
        o = epi_u4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
      } else {
        o = epi_u4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
      }
      if constexpr ((EF & EPI_STATS) != 0 && CH == 8) {
        const uint32_t ow[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = __uint_as_float(ow[e] << 16), hi = __uint_as_float(ow[e] & 0xffff0000u);
          st1[2 * e] += lo; st2[2 * e] = __builtin_fmaf(lo, lo, st2[2 * e]);
          st1[2 * e + 1] += hi; st2[2 * e + 1] = __builtin_fmaf(hi, hi, st2[2 * e + 1]);
        }
      }
      // the row offset goes into the VECTOR offset (one v_add), not into soffset: hipcc (ROCm 7.2) assumes that a buffer store
      // with a register soffset cannot have its data registers overwritten too early and pads nothing, but gfx950 does need
      // the wait states of a >64-bit store -- with `..., out_lo, ou` the VALU instruction behind the store clobbered the
      // fourth data dword of some lanes (tools/debug/epi_diff.py: wrong values in columns 4k+3 of the t = 1 row groups)
      __builtin_amdgcn_raw_buffer_store_b128(o, out_rs, out_lo + ou, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  if constexpr ((EF & EPI_STATS) != 0 && CH == 8) {
    // lanes with equal (lane & 7) hold the same 8 channels (rows differ): fold lane bits 3, 4, 5 in a fixed order, then lanes
    // 0-7 write the wave's 64 channels of its partial row
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o2 = 8; o2 < 64; o2 <<= 1) { st1[e] += __shfl_xor(st1[e], o2, 64); st2[e] += __shfl_xor(st2[e], o2, 64); }
    }
    if (lane < 8) {
      const int64_t row = (int64_t)(m0 + wm * TM * 32) / (TM * 32);
      float* dst = a.stats_partial + (row * 2) * a.N + n0 + wn * 64 + lane * 8;
      *(float4*)dst = make_float4(st1[0], st1[1], st1[2], st1[3]);
      *(float4*)(dst + 4) = make_float4(st1[4], st1[5], st1[6], st1[7]);
      *(float4*)(dst + a.N) = make_float4(st2[0], st2[1], st2[2], st2[3]);
      *(float4*)(dst + a.N + 4) = make_float4(st2[4], st2[5], st2[6], st2[7]);
    }
  }
}

template <int TM>
__device__ __forceinline__ bool conv_epilogue_rows2(const KArgs& k, f32x16_t (&acc)[TM][2], int m0, int n0, int wm, int wn, int lane,
                                                   int64_t out_zoff, unsigned char* lds) {
  const gdl_conv_args& a = k.a;
  const bool out_bf16 = a.out_dtype == GDL_BF16;
  const int HoWo = a.Ho * a.Wo;
  const bool vec_ok = ((uintptr_t)a.out % 16 == 0) && (a.out_sW % 8 == 0) && (out_zoff % 8 == 0) && (a.N % 16 == 0);
  const bool res_ok = !a.resid || (((uintptr_t)a.resid % 16 == 0) && (a.res_sW % 8 == 0) && (a.resid_dtype == a.out_dtype));
  const bool vecs_ok = ((uintptr_t)a.bias % 16 == 0) && ((uintptr_t)a.scale % 16 == 0) && ((uintptr_t)a.shift % 16 == 0);
  if (!k.epi_v2 || a.aux_out || a.act == GDL_ACT_MUL_GELU_GRAD ||
      !(k.out_dense && k.res_dense && vec_ok && res_ok && vecs_ok && (!a.batch_scale || HoWo >= 256)))
    return false;
  // full WAVE tiles only (see the implementation): the wave's TM x 32 rows and 64 channels are all in range.  The decision
  // is per wave -- both epilogues only touch wave-private LDS
  if (m0 + (wm + 1) * TM * 32 > k.M || n0 + (wn + 1) * 64 > a.N) return false;
  // the combinations the three models' hot layers use get compile-time element-wise terms (k.epi_v2 == 2 keeps the
  // run-time form for A/B runs); everything else takes the run-time form
  const bool sc = a.scale != nullptr, bs = a.batch_scale != nullptr, rs = a.resid != nullptr;
  if (a.stats_partial) {           // (the host only sets it for bias-only bf16 calls made of whole wave tiles: gdl_conv_gemm_stats_rows)
    conv_epilogue_rows2_impl<TM, 8, false, EPI_FIXED | EPI_STATS>(k, acc, m0, n0, wm, wn, lane, out_zoff, lds);
    return true;
  }
  if (k.epi_v2 != 2) {
    if (GDL_EPI_ON(1) && !rs && !bs && !sc && a.act == GDL_ACT_NONE) {                       // bias only: qkv, laterals, tap products, data gradients
      if (out_bf16) conv_epilogue_rows2_impl<TM, 8, false, EPI_FIXED>(k, acc, m0, n0, wm, wn, lane, out_zoff, lds);
      else conv_epilogue_rows2_impl<TM, 4, false, EPI_FIXED>(k, acc, m0, n0, wm, wn, lane, out_zoff, lds);
      return true;
    }
    if (GDL_EPI_ON(2) && !rs && !bs && !sc && a.act == GDL_ACT_GELU && out_bf16) {           // MLP fc1
      conv_epilogue_rows2_impl<TM, 8, false, EPI_FIXED | EPI_GELU>(k, acc, m0, n0, wm, wn, lane, out_zoff, lds);
      return true;
    }
    if (GDL_EPI_ON(4) && !rs && !bs && sc && a.act == GDL_ACT_RELU && out_bf16) {            // ConvModule with folded BatchNorm (eval)
      conv_epilogue_rows2_impl<TM, 8, false, EPI_FIXED | EPI_SCALE | EPI_RELU>(k, acc, m0, n0, wm, wn, lane, out_zoff, lds);
      return true;
    }
    if (GDL_EPI_ON(8) && rs && sc && a.act == GDL_ACT_NONE && !out_bf16) {                   // proj / fc2: LayerScale (x DropPath) + f32 residual stream
      if (bs) conv_epilogue_rows2_impl<TM, 4, true, EPI_FIXED | EPI_SCALE | EPI_DROPPATH>(k, acc, m0, n0, wm, wn, lane, out_zoff, lds);
      else conv_epilogue_rows2_impl<TM, 4, true, EPI_FIXED | EPI_SCALE>(k, acc, m0, n0, wm, wn, lane, out_zoff, lds);
      return true;
    }
  }
  if (a.resid) {
    if (out_bf16) conv_epilogue_rows2_impl<TM, 8, true>(k, acc, m0, n0, wm, wn, lane, out_zoff, lds);
    else conv_epilogue_rows2_impl<TM, 4, true>(k, acc, m0, n0, wm, wn, lane, out_zoff, lds);
  } else {
    if (out_bf16) conv_epilogue_rows2_impl<TM, 8, false>(k, acc, m0, n0, wm, wn, lane, out_zoff, lds);
    else conv_epilogue_rows2_impl<TM, 4, false>(k, acc, m0, n0, wm, wn, lane, out_zoff, lds);
  }
  return true;
}

// ---- epilogue.  Call with the accumulators of MFMAs that ran with SWAPPED operands (D = W_tile x X_tile^T).
// `lds_wave` / `lds_consts`: 8 KiB + 768 B of LDS private to this wave and free to overwrite (the caller has passed a
// workgroup barrier after the last fragment read of the K loop), or nullptr.  With it, and for pixel-dense outputs, the tile leaves
// through a wave-private LDS transpose: 32 rows x 64 channels per pass are written in the MFMA layout (16-byte pieces,
// slot ^ (row & 7): conflict-free), read back with 8 (bf16) / 16 (f32) consecutive lanes per row and stored as FULL
// 128 / 256-byte row segments -- instead of 64 scattered 16-byte pieces per store instruction, which cost 120-250
// cycles per instruction in the memory pipeline and doubled the bytes written to HBM (partial lines; PMC, round 2).
// Per-channel terms (bias, scale/shift, activation) are applied before the transpose, per-row terms (DropPath scale,
// residual, GELU-gradient multiply) after it, where the residual is read as coalesced row segments too.
template <int TM, int TN, bool EXTRA>
__device__ __forceinline__ void conv_epilogue(const KArgs& k, f32x16_t (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                              int lane, int64_t out_zoff, unsigned char* lds_wave = nullptr,
                                              unsigned char* lds_consts = nullptr) {
  const gdl_conv_args& a = k.a;
  const int frow = lane & 31, fhalf = lane >> 5;
  const int HoWo = a.Ho * a.Wo;
  if constexpr (TN == 2) {
    if (lds_wave && conv_epilogue_rows2<TM>(k, acc, m0, n0, wm, wn, lane, out_zoff, lds_wave)) return;
    if (lds_wave && conv_epilogue_rows<TM, EXTRA>(k, acc, m0, n0, wm, wn, lane, out_zoff, lds_wave, lds_consts)) return;
  }
  // ---- epilogue.  The MFMAs ran with swapped operands (D = W_tile x X_tile^T), so a lane owns ONE output row
  // m = lane&31 of each 32x32 tile and its 16 accumulators are 4 groups of 4 CONSECUTIVE channels
  // n = 8g + 4*(lane>>5) + e: per-channel terms are float4 loads, per-row terms (DropPath scale, residual) are
  // per lane, and results leave as 8/16-byte pieces straight from registers -- no LDS round trip, no barrier.
  // bf16: the two lanes that hold adjacent 8-byte pieces of a row trade one piece (v_permlane32_swap) so that each
  // stores 16 contiguous bytes.
  const bool out_bf16 = a.out_dtype == GDL_BF16;
  const int oes = out_bf16 ? 2 : 4;
  const bool mulgrad = EXTRA && a.act == GDL_ACT_MUL_GELU_GRAD;
  const bool plain = !a.resid && !a.batch_scale;
  const bool vec_ok = ((uintptr_t)a.out % 16 == 0) && ((uintptr_t)a.aux_out % 16 == 0) && (a.out_sW % 8 == 0) &&
                      (a.out_sH % 8 == 0) && (a.out_sB % 8 == 0) && (out_zoff % 8 == 0) && (a.N % 16 == 0);
  const bool res_vec = !a.resid || (((uintptr_t)a.resid % 16 == 0) && (a.res_sW % 4 == 0) && (a.res_sH % 4 == 0) &&
                                    (a.res_sB % 4 == 0));
  int64_t row_o[TM], row_r[TM];
  float row_s[TM];
  bool m_ok[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + (wm * TM + i) * 32 + frow;
    m_ok[i] = m < k.M;
    row_s[i] = 1.f;
    if (k.out_dense && k.res_dense && !a.batch_scale) {
      row_o[i] = (int64_t)m * a.out_sW + out_zoff;
      row_r[i] = (int64_t)m * a.res_sW;
    } else {
      const int mm = m_ok[i] ? m : 0;
      const int b = mm / HoWo, rem = mm - b * HoWo, oy = rem / a.Wo, ox = rem - oy * a.Wo;
      row_o[i] = (int64_t)b * a.out_sB + (int64_t)oy * a.out_sH + (int64_t)ox * a.out_sW + out_zoff;
      row_r[i] = (int64_t)b * a.res_sB + (int64_t)oy * a.res_sH + (int64_t)ox * a.res_sW;
      if (a.batch_scale) row_s[i] = a.batch_scale[b];
    }
  }
  auto load4 = [&](const float* p, int n, float dflt, float (&o)[4]) {
    if (!p) { o[0] = o[1] = o[2] = o[3] = dflt; return; }
    if (n + 3 < a.N && ((uintptr_t)p % 16 == 0)) {
      const float4 t = *(const float4*)(p + n);
      o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = n + e < a.N ? p[n + e] : dflt;
    }
  };
  // values of one 4-channel group of row i after the whole epilogue; `pre` gets alpha*acc + bias
  auto finish = [&](int i, int j, int g, int n, const float (&bias4)[4], const float (&sc4)[4], const float (&sh4)[4],
                    float (&v)[4], float (&pre)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = acc[i][j][4 * g + e] * a.alpha + bias4[e];
      pre[e] = x;
      if (a.scale) x = x * sc4[e] + sh4[e];
      if (a.act == GDL_ACT_RELU) x = fmaxf(x, 0.f);
      v[e] = x;
    }
    if (a.act == GDL_ACT_GELU) {
      const float4 t = gelu4(make_float4(v[0], v[1], v[2], v[3]));
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    if (!plain && m_ok[i]) {
      float rv[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.resid) {
        const int64_t ro = row_r[i] + n;
        if (res_vec && n + 3 < a.N) {
          if (a.resid_dtype == GDL_BF16) {
            const uint2 t = *(const uint2*)((const uint16_t*)a.resid + ro);
            rv[0] = __uint_as_float(t.x << 16); rv[1] = __uint_as_float(t.x & 0xffff0000u);
            rv[2] = __uint_as_float(t.y << 16); rv[3] = __uint_as_float(t.y & 0xffff0000u);
          } else {
            const float4 t = *(const float4*)((const float*)a.resid + ro);
            rv[0] = t.x; rv[1] = t.y; rv[2] = t.z; rv[3] = t.w;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < a.N) rv[e] = load_as_f32(a.resid, ro + e, a.resid_dtype);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= row_s[i];
      if (EXTRA && mulgrad) {
        const float4 t = mul_gelu_grad4(make_float4(v[0], v[1], v[2], v[3]), make_float4(rv[0], rv[1], rv[2], rv[3]));
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = v[e] + rv[e];
          if (a.act == GDL_ACT_RESID_RELU) x = fmaxf(x, 0.f);
          v[e] = x;
        }
      }
    }
  };
  auto store4 = [&](void* base, int64_t off, int n, const float (&v)[4]) {   // 4 channels of one row, tail-safe
    if (vec_ok && n + 3 < a.N) {
      if (out_bf16) *(uint2*)((uint16_t*)base + off) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      else *(float4*)((float*)base + off) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < a.N) store_from_f32(base, off + e, v[e], a.out_dtype);
    }
  };
  (void)oes;
  auto tile_col = [&](auto jc) {   // static j (the unroller's size budget does not cover the 8-wave variant's body)
    constexpr int j = decltype(jc)::value;
    const int nt = n0 + (wn * TN + j) * 32;
    if (out_bf16 && vec_ok) {
      if (nt < a.N) {
      // pairs of groups (2p, 2p+1): after the lane-pair swap the low half-wave stores channels [16p, 16p+8) of its
      // row and the high half-wave channels [16p+8, 16p+16)
#pragma unroll
      for (int p2 = 0; p2 < 2; ++p2) {
        if (nt + 16 * p2 >= a.N) continue;            // N % 32 == 16: the tile's upper 16 channels do not exist
        const int na = nt + 16 * p2 + 4 * fhalf, nb = na + 8;
        float ba[4], sa[4], ha[4], bb[4], sb[4], hb[4];
        load4(a.bias, na, 0.f, ba); load4(a.scale, na, 1.f, sa); load4(a.shift, na, 0.f, ha);
        load4(a.bias, nb, 0.f, bb); load4(a.scale, nb, 1.f, sb); load4(a.shift, nb, 0.f, hb);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          float va[4], vb[4], pa[4], pb[4];
          finish(i, j, 2 * p2, na, ba, sa, ha, va, pa);
          finish(i, j, 2 * p2 + 1, nb, bb, sb, hb, vb, pb);
          const int ncol = nt + 16 * p2 + 8 * fhalf;
          auto emit = [&](void* base, const float (&xa)[4], const float (&xb)[4]) {
            unsigned a0 = pack_bf16x2(xa[0], xa[1]), a1 = pack_bf16x2(xa[2], xa[3]);
            unsigned b0 = pack_bf16x2(xb[0], xb[1]), b1 = pack_bf16x2(xb[2], xb[3]);
            // upper half of (a0,a1) <-> lower half of (b0,b1): low lanes end with (lowA, highA), high with (lowB, highB)
            const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            if (m_ok[i]) *(uint4*)((uint16_t*)base + row_o[i] + ncol) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
          };
          if (EXTRA && a.aux_out) emit(a.aux_out, pa, pb);
          emit(a.out, va, vb);
        }
      }
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nt + 8 * g + 4 * fhalf;
        if (n < a.N) {
          float b4[4], s4[4], h4[4];
          load4(a.bias, n, 0.f, b4); load4(a.scale, n, 1.f, s4); load4(a.shift, n, 0.f, h4);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            float v[4], pre[4];
            finish(i, j, g, n, b4, s4, h4, v, pre);
            if (m_ok[i]) {
              if (EXTRA && a.aux_out) store4(a.aux_out, row_o[i] + n, n, pre);
              store4(a.out, row_o[i] + n, n, v);
            }
          }
        }
      }
    }
  };
  tile_col(std::integral_constant<int, 0>{});
  if constexpr (TN > 1) tile_col(std::integral_constant<int, 1>{});
  static_assert(TN <= 2, "epilogue is written for TN <= 2");
}

}  // namespace gdlconv
