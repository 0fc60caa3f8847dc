// Register-window walker shared by the depthwise 3x3 kernels (forward: dwconv.hip, fused backward:
// transformer_bwd.hip).  A thread owns 4 channels of one row segment y, x0..x1-1 of an NHWC tensor and keeps the
// 3x3 input window in registers: each output pixel costs three new pixel loads (the column x+1) instead of nine, and
// the column after that is already in flight as raw (unconverted) words while the current pixel is computed.
#pragma once
#include <type_traits>

#include "gdl_common.h"

namespace gdldw {

// a pixel's 4 channels as they sit in memory
template <typename T> struct Px;
template <> struct Px<float> {
  using raw = float4;
  static __device__ __forceinline__ raw zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  static __device__ __forceinline__ raw ld(const void* p, int64_t off) { return *(const float4*)((const float*)p + off); }
  static __device__ __forceinline__ void cvt(const raw& v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
  static __device__ __forceinline__ void st(void* p, int64_t off, const float (&o)[4]) {
    *(float4*)((float*)p + off) = make_float4(o[0], o[1], o[2], o[3]);
  }
};
template <> struct Px<uint16_t> {
  using raw = uint2;
  static __device__ __forceinline__ raw zero() { return make_uint2(0u, 0u); }
  static __device__ __forceinline__ raw ld(const void* p, int64_t off) { return *(const uint2*)((const uint16_t*)p + off); }
  static __device__ __forceinline__ void cvt(const raw& v, float (&o)[4]) {
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void st(void* p, int64_t off, const float (&o)[4]) {
    *(uint2*)((uint16_t*)p + off) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
  }
};
template <> struct Px<bf16_tag> : Px<uint16_t> {};

// Row segment descriptor: pix = (b*H + y)*W is the pixel index of (b, y, 0).
struct Seg { int64_t pix; int y, x0, x1; };

// items = B * nseg * H with y fastest, so neighbouring items (waves of one block) are neighbouring rows of the same
// x range and share their halo rows through the CU's L1.
__device__ __forceinline__ Seg seg_of(int64_t item, int H, int W, int seglen, int nseg) {
  const int y = (int)(item % H);
  const int64_t t = item / H;
  const int sg = (int)(t % nseg);
  const int64_t b = t / nseg;
  Seg s;
  s.y = y; s.x0 = sg * seglen; s.x1 = s.x0 + seglen < W ? s.x0 + seglen : W; s.pix = (b * H + y) * W;
  return s;
}
inline int seg_len(int W) { return W <= 32 ? W : 32; }

// Calls f(x, L, M, R) for x = x0..x1-1 where L/M/R[r][j] are the input columns x-1, x, x+1 (rows y-1, y, y+1; zero
// outside the image) of the thread's 4 channels starting at c.
template <typename T, typename F>
__device__ __forceinline__ void walk(const void* __restrict__ u, const Seg& sg, int H, int W, int C, int c, F&& f) {
  using P = Px<T>;
  using raw_t = typename P::raw;
  float col[3][3][4];
  raw_t nxt[3];
  const bool rv0 = sg.y > 0, rv2 = sg.y + 1 < H;
  const int64_t base = sg.pix * C + c;
  const int64_t rowb = (int64_t)W * C;
  auto fetch = [&](int xx) {
    const bool in = (unsigned)xx < (unsigned)W && xx <= sg.x1;      // x1 is the last halo column this segment needs
    const int64_t o = base + (int64_t)xx * C;
    nxt[0] = (in && rv0) ? P::ld(u, o - rowb) : P::zero();
    nxt[1] = in ? P::ld(u, o) : P::zero();
    nxt[2] = (in && rv2) ? P::ld(u, o + rowb) : P::zero();
  };
  auto land = [&](auto slot) {
    constexpr int S = decltype(slot)::value;
#pragma unroll
    for (int r = 0; r < 3; ++r) P::cvt(nxt[r], col[S][r]);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  fetch(sg.x0 - 1); land(I0{});
  fetch(sg.x0);     land(I1{});
  fetch(sg.x0 + 1);
  for (int x = sg.x0; x < sg.x1; x += 3) {
    land(I2{}); fetch(x + 2);
    f(x, col[0], col[1], col[2]);
    if (x + 1 < sg.x1) {
      land(I0{}); fetch(x + 3);
      f(x + 1, col[1], col[2], col[0]);
    }
    if (x + 2 < sg.x1) {
      land(I1{}); fetch(x + 4);
      f(x + 2, col[2], col[0], col[1]);
    }
  }
}

// per-channel tap weights + bias of the thread's 4 channels
struct Taps {
  float w[9][4], b[4];
  __device__ __forceinline__ void load(const float* __restrict__ w9, const float* __restrict__ bias, int C, int c) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float4 v = *(const float4*)(w9 + t * C + c);
      w[t][0] = v.x; w[t][1] = v.y; w[t][2] = v.z; w[t][3] = v.w;
    }
    const float4 bb = *(const float4*)(bias + c);
    b[0] = bb.x; b[1] = bb.y; b[2] = bb.z; b[3] = bb.w;
  }
  // bias + sum over the window; the tap order (r outer, s inner) matches the per-pixel kernel it replaces
  __device__ __forceinline__ void apply(const float (&L)[3][4], const float (&M)[3][4], const float (&R)[3][4],
                                        float (&acc)[4]) const {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = b[j];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] += L[r][j] * w[r * 3 + 0][j];
        acc[j] += M[r][j] * w[r * 3 + 1][j];
        acc[j] += R[r][j] * w[r * 3 + 2][j];
      }
  }
};

}  // namespace gdldw
