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
  int in_dense, out_dense, res_dense;
  unsigned in_span, w_span;   // bytes addressed from the (z-offset) operand base: buffer num_records
  int tap_inner;              // K order: 1 = channel chunk outer / filter tap inner (default), 0 = tap outer
  int dbg;                    // tuning experiments only: 1 = no DMA after the first tile, 2 = no MFMA
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

__device__ __forceinline__ int xcd_remap(int id, int n) {
  // bijective "XCD-major" remap (cdna guide T1): hardware places block id on XCD id % 8.
  const int q = n >> 3, r = n & 7, xcd = id & 7, idx = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}


// 3x3 shared-staging kernel (conv3x3_sf.hip)
bool conv3x3_sf_applicable(const gdl_conv_args& a);
int conv3x3_sf_launch(const KArgs& k, hipStream_t stream);

// ---- epilogue.  Call with the accumulators of MFMAs that ran with SWAPPED operands (D = W_tile x X_tile^T).
template <int TM, int TN, bool EXTRA>
__device__ __forceinline__ void conv_epilogue(const KArgs& k, f32x16_t (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                              int lane, int64_t out_zoff) {
  const gdl_conv_args& a = k.a;
  const int frow = lane & 31, fhalf = lane >> 5;
  const int HoWo = a.Ho * a.Wo;
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
