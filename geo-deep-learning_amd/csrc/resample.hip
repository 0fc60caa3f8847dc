// Bilinear resize (align_corners=False) and adaptive average pooling on NHWC tensors.
// HBM-bound: one thread per (pixel, 4-channel vector); f32 arithmetic; gather formulation in
// both directions (no atomics -> deterministic backward).
#include "gdl_common.h"
#include <type_traits>

namespace {

template <typename T> struct V4;
template <> struct V4<float> {
  typedef float elem;
  static __device__ __forceinline__ void ld(const void* p, int64_t off, float (&o)[4]) {
    const float4 v = *(const float4*)((const float*)p + off);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  static __device__ __forceinline__ void st(void* p, int64_t off, const float (&o)[4]) {
    *(float4*)((float*)p + off) = make_float4(o[0], o[1], o[2], o[3]);
  }
};
template <> struct V4<uint16_t> {
  typedef uint16_t elem;
  static __device__ __forceinline__ void ld(const void* p, int64_t off, float (&o)[4]) {
    const uint2 v = *(const uint2*)((const uint16_t*)p + off);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void st(void* p, int64_t off, const float (&o)[4]) {
    *(uint2*)((uint16_t*)p + off) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
  }
};

// 8 x bf16 = one 16-byte access: the bf16 -> bf16 resamples run 8 channels per thread (half the index math per byte)
struct V8 {
  typedef uint16_t elem;
  static __device__ __forceinline__ void ld(const void* p, int64_t off, float (&o)[8]) {
    const uint4 v = *(const uint4*)((const uint16_t*)p + off);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[2 * e] = __uint_as_float(w[e] << 16); o[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void st(void* p, int64_t off, const float (&o)[8]) {
    *(uint4*)((uint16_t*)p + off) = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]),
                                               pack_bf16x2(o[6], o[7]));
  }
};

// torch area_pixel_compute_source_index(align_corners=False) + index/lambda (UpSample.h)
__device__ __forceinline__ void src_index(float ratio, int dst, int in_size, int& i0, int& i1, float& l1) {
  float s = ratio * ((float)dst + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const void* __restrict__ in, int B, int Hi,
                                                           int Wi, int C, int64_t isB, int64_t isH,
                                                           int64_t isW, void* out, int Ho, int Wo,
                                                           int64_t osB, int64_t osH, int64_t osW,
                                                           int accumulate) {
  const int cv = C / 4;
  const int64_t total = (int64_t)B * Ho * Wo * cv;
  const float ry = (float)Hi / (float)Ho, rx = (float)Wi / (float)Wo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cv) * 4;
    int64_t t = i / cv;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const int64_t base = (int64_t)b * isB + c;
    float a[4], bb[4], cc[4], d[4], o[4];
    if (Hi == Ho && Wi == Wo) {  // identity resize == strided copy / cast / accumulate
      const int64_t ooff = (int64_t)b * osB + (int64_t)oy * osH + (int64_t)ox * osW + c;
      V4<TI>::ld(in, base + oy * isH + ox * isW, a);
      if (accumulate) {
        V4<TO>::ld(out, ooff, o);
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] += o[j];
      }
      V4<TO>::st(out, ooff, a);
      continue;
    }
    int y0, y1, x0, x1; float ly, lx;
    src_index(ry, oy, Hi, y0, y1, ly);
    src_index(rx, ox, Wi, x0, x1, lx);
    V4<TI>::ld(in, base + y0 * isH + x0 * isW, a);
    V4<TI>::ld(in, base + y0 * isH + x1 * isW, bb);
    V4<TI>::ld(in, base + y1 * isH + x0 * isW, cc);
    V4<TI>::ld(in, base + y1 * isH + x1 * isW, d);
    const int64_t ooff = (int64_t)b * osB + (int64_t)oy * osH + (int64_t)ox * osW + c;
    if (accumulate) V4<TO>::ld(out, ooff, o);
    else { o[0] = o[1] = o[2] = o[3] = 0.f; }
    const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] += hy * (hx * a[j] + lx * bb[j]) + ly * (hx * cc[j] + lx * d[j]);
    V4<TO>::st(out, ooff, o);
  }
}

// out = sum_k bilinear(src_k -> Ho x Wo), up to three dense NHWC sources of different sizes with the same channel count,
// ONE write of the output.  Used where a 1x1 convolution over a concat of upsampled pyramid levels is evaluated per
// level at the level's own resolution (a 1x1 convolution and a bilinear resize commute), SegFormer's linear_fuse:
// the upsampled partial results are summed here instead of being written one by one.  Source maps are small (L2-resident).
struct SumSrc { const void* p[3]; int H[3], W[3]; };
// grid = (ceil(Wo * C/VEC / 256), B * Ho): one block per output-row segment, so the batch / row decomposition and every
// source's row pair + vertical weight are block-uniform; a thread only derives its column pair.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bilinear_sum_kernel(SumSrc src, int nsrc, int B, int C, void* out, int Ho, int Wo) {
  const int cv = C / VEC;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= Wo * cv) return;
  const int ox = j / cv, c = (j - ox * cv) * VEC;
  const int b = blockIdx.y / Ho, oy = blockIdx.y - b * Ho;
  float o[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) o[e] = 0.f;
  for (int k = 0; k < nsrc; ++k) {
    const int Hi = src.H[k], Wi = src.W[k];
    int y0, y1, x0, x1; float ly, lx;
    src_index((float)Hi / (float)Ho, oy, Hi, y0, y1, ly);
    src_index((float)Wi / (float)Wo, ox, Wi, x0, x1, lx);
    const int64_t r0 = ((int64_t)b * Hi + y0) * Wi * C + c, r1 = ((int64_t)b * Hi + y1) * Wi * C + c;
    float a[VEC], bb[VEC], cc[VEC], d[VEC];
    if constexpr (VEC == 8) {
      V8::ld(src.p[k], r0 + (int64_t)x0 * C, a);
      V8::ld(src.p[k], r0 + (int64_t)x1 * C, bb);
      V8::ld(src.p[k], r1 + (int64_t)x0 * C, cc);
      V8::ld(src.p[k], r1 + (int64_t)x1 * C, d);
    } else {
      V4<T>::ld(src.p[k], r0 + (int64_t)x0 * C, a);
      V4<T>::ld(src.p[k], r0 + (int64_t)x1 * C, bb);
      V4<T>::ld(src.p[k], r1 + (int64_t)x0 * C, cc);
      V4<T>::ld(src.p[k], r1 + (int64_t)x1 * C, d);
    }
    const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] += hy * (hx * a[e] + lx * bb[e]) + ly * (hx * cc[e] + lx * d[e]);
  }
  const int64_t ooff = (((int64_t)b * Ho + oy) * Wo + ox) * C + c;
  if constexpr (VEC == 8) V8::st(out, ooff, o);
  else V4<T>::st(out, ooff, o);
}

// Strided NHWC copy with dtype conversion (the compute-dtype cast of a channel slice, densifying a strided gradient):
// one thread per (pixel, 4-channel vector); rows of pixels are walked per block so that no 64-bit division runs per element.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void copy_cast_kernel(const void* __restrict__ in, int W, int C, int64_t isB, int64_t isH,
                                                        int64_t isW, void* out, int H, int64_t osB, int64_t osH, int64_t osW) {
  const int cv = C / 4;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= W * cv) return;
  const int x = j / cv, c = (j - x * cv) * 4;
  const int b = blockIdx.y / H, y = blockIdx.y - b * H;
  float v[4];
  V4<TI>::ld(in, (int64_t)b * isB + (int64_t)y * isH + (int64_t)x * isW + c, v);
  V4<TO>::st(out, (int64_t)b * osB + (int64_t)y * osH + (int64_t)x * osW + c, v);
}
// bf16 -> bf16 with 16-byte accesses
__global__ __launch_bounds__(256) void copy8_kernel(const void* __restrict__ in, int W, int C, int64_t isB, int64_t isH,
                                                    int64_t isW, void* out, int H, int64_t osB, int64_t osH, int64_t osW) {
  const int cv = C / 8;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= W * cv) return;
  const int x = j / cv, c = (j - x * cv) * 8;
  const int b = blockIdx.y / H, y = blockIdx.y - b * H;
  const uint4 v = *(const uint4*)((const uint16_t*)in + (int64_t)b * isB + (int64_t)y * isH + (int64_t)x * isW + c);
  *(uint4*)((uint16_t*)out + (int64_t)b * osB + (int64_t)y * osH + (int64_t)x * osW + c) = v;
}

// Backward (gather): din[iy,ix] (+)= sum over outputs whose taps touch (iy,ix).
// Candidate outputs: src in (i-1, i+1)  =>  dst in ((i-0.5)/ratio - 0.5, (i+1.5)/ratio - 0.5).
template <typename TO_, typename TI_>  // TO_ = dtype of dout, TI_ = dtype of din
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const void* __restrict__ dout, int B, int Ho,
                                                           int Wo, int C, int64_t osB, int64_t osH,
                                                           int64_t osW, void* din, int Hi, int Wi,
                                                           int64_t isB, int64_t isH, int64_t isW,
                                                           int accumulate) {
  const int cv = C / 4;
  const int64_t total = (int64_t)B * Hi * Wi * cv;
  const float ry = (float)Hi / (float)Ho, rx = (float)Wi / (float)Wo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cv) * 4;
    int64_t t = i / cv;
    const int ix = (int)(t % Wi); t /= Wi;
    const int iy = (int)(t % Hi);
    const int b = (int)(t / Hi);
    int oy_lo = (int)floorf(((float)iy - 0.5f) / ry - 0.5f) - 1, oy_hi = (int)ceilf(((float)iy + 1.5f) / ry - 0.5f) + 1;
    int ox_lo = (int)floorf(((float)ix - 0.5f) / rx - 0.5f) - 1, ox_hi = (int)ceilf(((float)ix + 1.5f) / rx - 0.5f) + 1;
    oy_lo = oy_lo < 0 ? 0 : oy_lo; ox_lo = ox_lo < 0 ? 0 : ox_lo;
    oy_hi = oy_hi > Ho - 1 ? Ho - 1 : oy_hi; ox_hi = ox_hi > Wo - 1 ? Wo - 1 : ox_hi;
    float acc[4] = {0, 0, 0, 0};
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1; float ly;
      src_index(ry, oy, Hi, y0, y1, ly);
      const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1; float lx;
        src_index(rx, ox, Wi, x0, x1, lx);
        const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
        if (wx == 0.f) continue;
        float g[4];
        V4<TO_>::ld(dout, (int64_t)b * osB + (int64_t)oy * osH + (int64_t)ox * osW + c, g);
        const float w = wy * wx;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += w * g[j];
      }
    }
    const int64_t ioff = (int64_t)b * isB + (int64_t)iy * isH + (int64_t)ix * isW + c;
    if (accumulate) {
      float o[4];
      V4<TI_>::ld(din, ioff, o);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += o[j];
    }
    V4<TI_>::st(din, ioff, acc);
  }
}

// ---- bf16 -> bf16, 8 channels per thread (same arithmetic as the generic kernels above)
__global__ __launch_bounds__(256) void bilinear_fwd8_kernel(const void* __restrict__ in, int B, int Hi, int Wi, int C,
                                                            int64_t isB, int64_t isH, int64_t isW, void* out, int Ho,
                                                            int Wo, int64_t osB, int64_t osH, int64_t osW,
                                                            int accumulate) {
  const int cv = C / 8;
  const int64_t total = (int64_t)B * Ho * Wo * cv;
  const float ry = (float)Hi / (float)Ho, rx = (float)Wi / (float)Wo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cv) * 8;
    int64_t t = i / cv;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const int64_t base = (int64_t)b * isB + c;
    const int64_t ooff = (int64_t)b * osB + (int64_t)oy * osH + (int64_t)ox * osW + c;
    float a[8], bb[8], cc[8], d[8], o[8];
    if (Hi == Ho && Wi == Wo) {
      V8::ld(in, base + oy * isH + ox * isW, a);
      if (accumulate) {
        V8::ld(out, ooff, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += o[j];
      }
      V8::st(out, ooff, a);
      continue;
    }
    int y0, y1, x0, x1; float ly, lx;
    src_index(ry, oy, Hi, y0, y1, ly);
    src_index(rx, ox, Wi, x0, x1, lx);
    V8::ld(in, base + y0 * isH + x0 * isW, a);
    V8::ld(in, base + y0 * isH + x1 * isW, bb);
    V8::ld(in, base + y1 * isH + x0 * isW, cc);
    V8::ld(in, base + y1 * isH + x1 * isW, d);
    if (accumulate) V8::ld(out, ooff, o);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = 0.f;
    }
    const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] += hy * (hx * a[j] + lx * bb[j]) + ly * (hx * cc[j] + lx * d[j]);
    V8::st(out, ooff, o);
  }
}

__global__ __launch_bounds__(256) void bilinear_bwd8_kernel(const void* __restrict__ dout, int B, int Ho, int Wo, int C,
                                                            int64_t osB, int64_t osH, int64_t osW, void* din, int Hi,
                                                            int Wi, int64_t isB, int64_t isH, int64_t isW,
                                                            int accumulate) {
  const int cv = C / 8;
  const int64_t total = (int64_t)B * Hi * Wi * cv;
  const float ry = (float)Hi / (float)Ho, rx = (float)Wi / (float)Wo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cv) * 8;
    int64_t t = i / cv;
    const int ix = (int)(t % Wi); t /= Wi;
    const int iy = (int)(t % Hi);
    const int b = (int)(t / Hi);
    int oy_lo = (int)floorf(((float)iy - 0.5f) / ry - 0.5f) - 1, oy_hi = (int)ceilf(((float)iy + 1.5f) / ry - 0.5f) + 1;
    int ox_lo = (int)floorf(((float)ix - 0.5f) / rx - 0.5f) - 1, ox_hi = (int)ceilf(((float)ix + 1.5f) / rx - 0.5f) + 1;
    oy_lo = oy_lo < 0 ? 0 : oy_lo; ox_lo = ox_lo < 0 ? 0 : ox_lo;
    oy_hi = oy_hi > Ho - 1 ? Ho - 1 : oy_hi; ox_hi = ox_hi > Wo - 1 ? Wo - 1 : ox_hi;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1; float ly;
      src_index(ry, oy, Hi, y0, y1, ly);
      const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1; float lx;
        src_index(rx, ox, Wi, x0, x1, lx);
        const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
        if (wx == 0.f) continue;
        float g[8];
        V8::ld(dout, (int64_t)b * osB + (int64_t)oy * osH + (int64_t)ox * osW + c, g);
        const float w = wy * wx;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += w * g[j];
      }
    }
    const int64_t ioff = (int64_t)b * isB + (int64_t)iy * isH + (int64_t)ix * isW + c;
    if (accumulate) {
      float o[8];
      V8::ld(din, ioff, o);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += o[j];
    }
    V8::st(din, ioff, acc);
  }
}

// Large upsampling factors (UperNet's pyramid pooling: 1x1 .. 6x6 maps resized to 18 x 18, factors 3 .. 18): an input pixel
// gathers from up to (2 f + 2)^2 output pixels, and the flat kernel above walks that window with ONE thread per 8 channels --
// 324 dependent 16-byte loads for the 1x1 branch, 1024 threads on the whole chip: 112 us at batch 4, 162 us at batch 32.
// Here a block owns one input pixel: 32 lanes cover 256 channels, the 8 lane groups split the window rows, LDS adds the
// eight partial sums in a fixed order.
__global__ __launch_bounds__(256) void bilinear_bwd8_window_kernel(const void* __restrict__ dout, int Ho, int Wo, int C,
                                                                   int64_t osB, int64_t osH, int64_t osW, void* din, int Hi,
                                                                   int Wi, int64_t isB, int64_t isH, int64_t isW, int accumulate) {
  __shared__ float red[8][32][8];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = (blockIdx.x * 32 + cl) * 8;
  int t = blockIdx.y;
  const int ix = t % Wi; t /= Wi;
  const int iy = t % Hi;
  const int b = t / Hi;
  const float ry = (float)Hi / (float)Ho, rx = (float)Wi / (float)Wo;
  int oy_lo = (int)floorf(((float)iy - 0.5f) / ry - 0.5f) - 1, oy_hi = (int)ceilf(((float)iy + 1.5f) / ry - 0.5f) + 1;
  int ox_lo = (int)floorf(((float)ix - 0.5f) / rx - 0.5f) - 1, ox_hi = (int)ceilf(((float)ix + 1.5f) / rx - 0.5f) + 1;
  oy_lo = oy_lo < 0 ? 0 : oy_lo; ox_lo = ox_lo < 0 ? 0 : ox_lo;
  oy_hi = oy_hi > Ho - 1 ? Ho - 1 : oy_hi; ox_hi = ox_hi > Wo - 1 ? Wo - 1 : ox_hi;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (c < C) {
    for (int oy = oy_lo + sl; oy <= oy_hi; oy += 8) {
      int y0, y1; float ly;
      src_index(ry, oy, Hi, y0, y1, ly);
      const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1; float lx;
        src_index(rx, ox, Wi, x0, x1, lx);
        const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
        if (wx == 0.f) continue;
        float g[8];
        V8::ld(dout, (int64_t)b * osB + (int64_t)oy * osH + (int64_t)ox * osW + c, g);
        const float w = wy * wx;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += w * g[j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[sl][cl][j] = acc[j];
  __syncthreads();
  if (sl != 0 || c >= C) return;
#pragma unroll
  for (int k = 1; k < 8; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += red[k][cl][j];
  const int64_t ioff = (int64_t)b * isB + (int64_t)iy * isH + (int64_t)ix * isW + c;
  if (accumulate) {
    float o[8];
    V8::ld(din, ioff, o);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += o[j];
  }
  V8::st(din, ioff, acc);
}

// ---- row-structured versions of the two 8-channel kernels: blockIdx.y = (batch, row), so the row decomposition and the
// vertical source index are per-block scalars and a thread divides once (the flat kernels above spend ~100 VALU
// instructions of index arithmetic per 16 bytes and ran at 2.3 TB/s; DOFA's resamples are 5.6 % of the train step).
// Same arithmetic, same summation order -> bit-identical results.
__global__ __launch_bounds__(256) void bilinear_fwd8_rows_kernel(const void* __restrict__ in, int Hi, int Wi, int C,
                                                                 int64_t isB, int64_t isH, int64_t isW, void* out,
                                                                 int Ho, int Wo, int64_t osB, int64_t osH, int64_t osW,
                                                                 int accumulate) {
  const int cv = C / 8;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= Wo * cv) return;
  const int b = blockIdx.y / Ho, oy = blockIdx.y - b * Ho;
  const int ox = j / cv, c = (j - ox * cv) * 8;
  const float ry = (float)Hi / (float)Ho, rx = (float)Wi / (float)Wo;
  const int64_t base = (int64_t)b * isB + c;
  const int64_t ooff = (int64_t)b * osB + (int64_t)oy * osH + (int64_t)ox * osW + c;
  float a[8], bb[8], cc[8], d[8], o[8];
  if (Hi == Ho && Wi == Wo) {
    V8::ld(in, base + oy * isH + ox * isW, a);
    if (accumulate) {
      V8::ld(out, ooff, o);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += o[e];
    }
    V8::st(out, ooff, a);
    return;
  }
  int y0, y1, x0, x1; float ly, lx;
  src_index(ry, oy, Hi, y0, y1, ly);
  src_index(rx, ox, Wi, x0, x1, lx);
  V8::ld(in, base + y0 * isH + x0 * isW, a);
  V8::ld(in, base + y0 * isH + x1 * isW, bb);
  V8::ld(in, base + y1 * isH + x0 * isW, cc);
  V8::ld(in, base + y1 * isH + x1 * isW, d);
  if (accumulate) V8::ld(out, ooff, o);
  else {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
  }
  const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] += hy * (hx * a[e] + lx * bb[e]) + ly * (hx * cc[e] + lx * d[e]);
  V8::st(out, ooff, o);
}

// Exact integer upsampling factors (F = 2, 4: the FPN top-down path, the pyramid levels of fpn_bottleneck's concat): a thread owns
// the GAP between four neighbouring input pixels -- the F x F output pixels whose two source rows and two source columns are
// those four -- loads them once and writes F x F outputs (the row kernel above loads four 16-byte pieces per 16 bytes stored:
// 2.65 TB/s on a write-bound op).  Gaps -1 and Hi - 1 (Wi - 1) are the clamped borders with F / 2 rows (columns).  Same
// source indices, weights and expression per output as bilinear_fwd8_rows_kernel: bit-identical results.
template <int F>
__global__ __launch_bounds__(256) void bilinear_fwd8_gap_kernel(const void* __restrict__ in, int Hi, int Wi, int C,
                                                                int64_t isB, int64_t isH, int64_t isW, void* out,
                                                                int64_t osB, int64_t osH, int64_t osW, const void* acc_src) {
  // acc_src: nullptr, `out` itself (out += result) or another map with out's strides (out = that + result: the FPN top-down add
  // without first copying the lateral into the output)
  const int cv = C / 8;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= (Wi + 1) * cv) return;
  const int Ho = F * Hi, Wo = F * Wi;
  const int b = blockIdx.y / (Hi + 1), gy = blockIdx.y - b * (Hi + 1);     // gap gy: output rows F gy - F/2 .. F gy + F/2 - 1
  const int gx = j / cv, c = (j - gx * cv) * 8;
  const float ry = (float)Hi / (float)Ho, rx = (float)Wi / (float)Wo;
  const int oy0 = F * gy - F / 2 < 0 ? 0 : F * gy - F / 2, ox0 = F * gx - F / 2 < 0 ? 0 : F * gx - F / 2;
  int y0, y1, x0, x1; float l;
  src_index(ry, oy0, Hi, y0, y1, l);
  src_index(rx, ox0, Wi, x0, x1, l);
  const int64_t base = (int64_t)b * isB + c;
  float a[8], bb[8], cc[8], d[8];
  V8::ld(in, base + y0 * isH + x0 * isW, a);
  V8::ld(in, base + y0 * isH + x1 * isW, bb);
  V8::ld(in, base + y1 * isH + x0 * isW, cc);
  V8::ld(in, base + y1 * isH + x1 * isW, d);
#pragma unroll
  for (int jy = 0; jy < F; ++jy) {
    const int oy = F * gy - F / 2 + jy;
    if (oy < 0 || oy >= Ho) continue;
    int t0, t1; float ly;
    src_index(ry, oy, Hi, t0, t1, ly);
    const float hy = 1.f - ly;
#pragma unroll
    for (int jx = 0; jx < F; ++jx) {
      const int ox = F * gx - F / 2 + jx;
      if (ox < 0 || ox >= Wo) continue;
      float lx;
      src_index(rx, ox, Wi, t0, t1, lx);
      const float hx = 1.f - lx;
      const int64_t ooff = (int64_t)b * osB + (int64_t)oy * osH + (int64_t)ox * osW + c;
      float o[8];
      if (acc_src) V8::ld(acc_src, ooff, o);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += hy * (hx * a[e] + lx * bb[e]) + ly * (hx * cc[e] + lx * d[e]);
      V8::st(out, ooff, o);
    }
  }
}

// NX = upper bound of the horizontal output window of one input pixel (2 * ceil(Wo / Wi) + 4); the window's weights are
// computed once per thread instead of once per (output row, output column)
template <int NX>
__global__ __launch_bounds__(256) void bilinear_bwd8_rows_kernel(const void* __restrict__ dout, int Ho, int Wo, int C,
                                                                 int64_t osB, int64_t osH, int64_t osW, void* din,
                                                                 int Hi, int Wi, int64_t isB, int64_t isH, int64_t isW,
                                                                 int accumulate) {
  const int cv = C / 8;
  // Neighbouring input rows gather from overlapping dout rows (a x4 resize: 8 dout rows per input row, 4 apart).  Dealt
  // out in launch order, consecutive blocks land on different XCDs and every L2 fetches its own copy (PMC: 1.30 GB fetched
  // per launch for 0.34 GB of dout); with the XCD-major order one XCD walks a contiguous band of rows.
  const int nblk = gridDim.x * gridDim.y;
  const int id = blockIdx.y * gridDim.x + blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = id & 7, idx = id >> 3;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int brow = lid / gridDim.x, bx = lid - brow * gridDim.x;
  const int j = bx * 256 + threadIdx.x;
  if (j >= Wi * cv) return;
  const int b = brow / Hi, iy = brow - b * Hi;
  const int ix = j / cv, c = (j - ix * cv) * 8;
  const float ry = (float)Hi / (float)Ho, rx = (float)Wi / (float)Wo;
  int oy_lo = (int)floorf(((float)iy - 0.5f) / ry - 0.5f) - 1, oy_hi = (int)ceilf(((float)iy + 1.5f) / ry - 0.5f) + 1;
  int ox_lo = (int)floorf(((float)ix - 0.5f) / rx - 0.5f) - 1, ox_hi = (int)ceilf(((float)ix + 1.5f) / rx - 0.5f) + 1;
  oy_lo = oy_lo < 0 ? 0 : oy_lo; ox_lo = ox_lo < 0 ? 0 : ox_lo;
  oy_hi = oy_hi > Ho - 1 ? Ho - 1 : oy_hi; ox_hi = ox_hi > Wo - 1 ? Wo - 1 : ox_hi;
  float wx[NX];
#pragma unroll
  for (int k = 0; k < NX; ++k) {
    const int ox = ox_lo + k;
    wx[k] = 0.f;
    if (ox <= ox_hi) {
      int x0, x1; float lx;
      src_index(rx, ox, Wi, x0, x1, lx);
      wx[k] = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
    }
  }
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const int64_t col = (int64_t)b * osB + (int64_t)ox_lo * osW + c;
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    int y0, y1; float ly;
    src_index(ry, oy, Hi, y0, y1, ly);
    const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
    if (wy == 0.f) continue;                               // uniform over the block
    const int64_t rowoff = col + (int64_t)oy * osH;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      if (wx[k] == 0.f) continue;
      float g[8];
      V8::ld(dout, rowoff + (int64_t)k * osW, g);
      const float w = wy * wx[k];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += w * g[e];
    }
  }
  const int64_t ioff = (int64_t)b * isB + (int64_t)iy * isH + (int64_t)ix * isW + c;
  if (accumulate) {
    float o[8];
    V8::ld(din, ioff, o);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += o[e];
  }
  V8::st(din, ioff, acc);
}

// Exact integer factors (F = 2, 4): a thread WALKS DOWN a column of input pixels.  The F gradient rows between input rows g - 1
// and g feed both of them (weights 1 - ly and ly): they are loaded once and added to two running accumulators, instead of once
// per input row (the row kernel above fetches 4 F^2 pieces per input pixel, this one 2 F^2 (1 + 1 / RSEG)).  A block column is cut
// into segments of RSEG input rows for parallelism; the gap rows on a segment border are read by both neighbours.  Every input
// pixel receives its terms in the row kernel's order (gradient rows ascending, window columns ascending, zero weights skipped)
// with the same products: bit-identical results.
template <int F, int RSEG>
__global__ __launch_bounds__(256) void bilinear_bwd8_walk_kernel(const void* __restrict__ dout, int C, int64_t osB, int64_t osH,
                                                                 int64_t osW, void* din, int Hi, int Wi, int64_t isB, int64_t isH,
                                                                 int64_t isW, int accumulate) {
  const int cv = C / 8, Ho = F * Hi, Wo = F * Wi;
  // XCD-major block order (as in the row kernel): horizontally neighbouring blocks read overlapping gradient columns (a window
  // of 2 F columns every F) and vertically neighbouring segments share their border rows -- one XCD walks a contiguous range
  const int nblk = gridDim.x * gridDim.y;
  const int id = blockIdx.y * gridDim.x + blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = id & 7, idx = id >> 3;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int brow = lid / gridDim.x, bx = lid - brow * gridDim.x;
  const int j = bx * 256 + threadIdx.x;
  if (j >= Wi * cv) return;
  const int nseg = (Hi + RSEG - 1) / RSEG;
  const int b = brow / nseg, seg = brow - b * nseg;
  const int iy0 = seg * RSEG, iy1 = iy0 + RSEG < Hi ? iy0 + RSEG : Hi;
  const int ix = j / cv, c = (j - ix * cv) * 8;
  const float ry = (float)Hi / (float)Ho, rx = (float)Wi / (float)Wo;
  // the window columns with a non-zero weight are exactly F ix - F/2 .. F ix + 3F/2 - 1 (inside the image); a column outside is
  // fetched from the clamped address -- UNCONDITIONAL loads, so that the 2 F pieces of a gradient row are in flight together
  // (inside `if (wx[k] == 0.f) continue` every load waited for its predecessor's multiply-adds) -- and its term is not added
  constexpr int NW = 2 * F;
  float wx[NW];
  int64_t coff[NW];
  unsigned okmask = 0;
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    const int ox = F * ix - F / 2 + k;
    const bool ok = ox >= 0 && ox < Wo;
    const int oxc = ox < 0 ? 0 : (ox > Wo - 1 ? Wo - 1 : ox);
    int x0, x1; float lx;
    src_index(rx, oxc, Wi, x0, x1, lx);
    wx[k] = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
    coff[k] = (int64_t)oxc * osW;
    okmask |= (ok ? 1u : 0u) << k;
  }
  const int64_t col = (int64_t)b * osB + c;
  float up[8], dn[8];        // accumulators of input rows g - 1 and g while gap g is walked
#pragma unroll
  for (int e = 0; e < 8; ++e) up[e] = dn[e] = 0.f;
  for (int g = iy0; g <= iy1; ++g) {                       // gap g: gradient rows F g - F/2 .. F g + F/2 - 1 (source rows g - 1, g)
#pragma unroll
    for (int jy = 0; jy < F; ++jy) {
      const int oy = F * g - F / 2 + jy;
      if (oy < 0 || oy >= Ho) continue;                    // (uniform)
      int y0, y1; float ly;
      src_index(ry, oy, Hi, y0, y1, ly);
      // weights of this gradient row for input rows g - 1 and g, as the row kernel forms them for each of the two
      const float w_up = g - 1 >= iy0 ? (y0 == g - 1 ? 1.f - ly : 0.f) + (y1 == g - 1 ? ly : 0.f) : 0.f;
      const float w_dn = g < iy1 ? (y0 == g ? 1.f - ly : 0.f) + (y1 == g ? ly : 0.f) : 0.f;
      if (w_up == 0.f && w_dn == 0.f) continue;            // (uniform)
      const int64_t rowoff = col + (int64_t)oy * osH;
      float v[NW][8];
#pragma unroll
      for (int k = 0; k < NW; ++k) V8::ld(dout, rowoff + coff[k], v[k]);
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        const bool ok = (okmask >> k) & 1u;
        if (w_up != 0.f) {                                 // (uniform)
          const float w = w_up * wx[k];
#pragma unroll
          for (int e = 0; e < 8; ++e) up[e] = ok ? up[e] + w * v[k][e] : up[e];
        }
        if (w_dn != 0.f) {
          const float w = w_dn * wx[k];
#pragma unroll
          for (int e = 0; e < 8; ++e) dn[e] = ok ? dn[e] + w * v[k][e] : dn[e];
        }
      }
    }
    if (g - 1 >= iy0) {                                    // input row g - 1 is complete
      const int64_t ioff = (int64_t)b * isB + (int64_t)(g - 1) * isH + (int64_t)ix * isW + c;
      if (accumulate) {
        float o[8];
        V8::ld(din, ioff, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) up[e] += o[e];
      }
      V8::st(din, ioff, up);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { up[e] = dn[e]; dn[e] = 0.f; }
  }
}

// Backward of  y = conv3x3(pad 1)(bilinear_resize(x))  WITHOUT the high-resolution data gradient.
//   y[p] = sum_t W_t (S_t U x)[p]      U = the resize (spatial), S_t = shift by tap t with zero fill (spatial), W_t = channel mix
// U and S_t act on pixels, W_t on channels, so they commute:  dx = sum_t W_t^T G_t  and  dW_t = sum_q G_t[q] (x) x[q]  with
//   G_t = U^T S_t^T dy                  -- nine LOW-resolution maps per output channel.
// Both gradients become GEMMs over the low-resolution pixels (1/16 of the MACs of the full-resolution data / weight
// gradient for a x4 resize).  This kernel is the gather G_t[q] = sum_{p'} U[p', q] dy[p' - t] (p' and p' - t inside the
// image), written as [B, Hi, Wi, 9 * N] with tap block 8 - t (the order of gdl_pack_dgrad's flipped taps, so that the
// existing data-gradient operand applies as a 1x1 convolution).  One block per low-resolution row segment; the
// vertical weights are block-uniform.  f32 accumulation, fixed order, no atomics.
// WXL: the window's horizontal weights live in LDS ([NX][thread]) and the column loop is a runtime loop -- for wide windows
// (x8 resize: NX = 20) the fully unrolled loop with register-held weights spilled.
template <typename V, int VEC, int NX, bool WXL = false>
__global__ __launch_bounds__(256) void resize_conv3x3_bwd_gather_kernel(const void* __restrict__ dy, int Ho, int Wo, int N,
                                                                        void* g, int Hi, int Wi) {
  __shared__ float wxs[WXL ? NX + 4 : 1][WXL ? 256 : 1];
  const int cv = N / VEC;
  const int nblk = gridDim.x * gridDim.y;
  const int id = blockIdx.y * gridDim.x + blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = id & 7, idx = id >> 3;      // XCD-major: neighbouring rows share an L2
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int brow = lid / gridDim.x, bx = lid - brow * gridDim.x;
  const int j0 = bx * 256 + threadIdx.x;
  if (j0 >= Wi * cv) return;
  const int b = brow / Hi, iy = brow - b * Hi;
  const int ix = j0 / cv, c = (j0 - ix * cv) * VEC;
  const float ry = (float)Hi / (float)Ho, rx = (float)Wi / (float)Wo;
  int oy_lo = (int)floorf(((float)iy - 0.5f) / ry - 0.5f) - 1, oy_hi = (int)ceilf(((float)iy + 1.5f) / ry - 0.5f) + 1;
  int ox_lo = (int)floorf(((float)ix - 0.5f) / rx - 0.5f) - 1, ox_hi = (int)ceilf(((float)ix + 1.5f) / rx - 0.5f) + 1;
  oy_lo = oy_lo < 0 ? 0 : oy_lo; ox_lo = ox_lo < 0 ? 0 : ox_lo;
  oy_hi = oy_hi > Ho - 1 ? Ho - 1 : oy_hi; ox_hi = ox_hi > Wo - 1 ? Wo - 1 : ox_hi;
  // horizontal weights of the window columns ox_lo + k (zero beyond the window)
  float wx[WXL ? 1 : NX];
  if constexpr (WXL) {
    // rows 0,1 and NX+2,NX+3 of the LDS table are zero guards: column j reads entries j, j+1, j+2 = window k = j-2, j-1, j
    wxs[0][threadIdx.x] = 0.f; wxs[1][threadIdx.x] = 0.f; wxs[NX + 2][threadIdx.x] = 0.f; wxs[NX + 3][threadIdx.x] = 0.f;
    for (int k = 0; k < NX; ++k) {
      const int ox = ox_lo + k;
      float w = 0.f;
      if (ox <= ox_hi) {
        int x0, x1; float lx;
        src_index(rx, ox, Wi, x0, x1, lx);
        w = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
      }
      wxs[k + 2][threadIdx.x] = w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int ox = ox_lo + k;
      wx[k] = 0.f;
      if (ox <= ox_hi) {
        int x0, x1; float lx;
        src_index(rx, ox, Wi, x0, x1, lx);
        wx[k] = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
      }
    }
  }
  auto wyf = [&](int oy) -> float {            // vertical weight U_y[oy, iy]; 0 outside the image / the window
    if (oy < oy_lo || oy > oy_hi) return 0.f;
    int y0, y1; float ly;
    src_index(ry, oy, Hi, y0, y1, ly);
    return (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
  };
  float acc[9][VEC];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[t][e] = 0.f;
  const int sy_lo = oy_lo > 0 ? oy_lo - 1 : 0, sy_hi = oy_hi < Ho - 1 ? oy_hi + 1 : Ho - 1;
  for (int sy = sy_lo; sy <= sy_hi; ++sy) {
    // source pixel p = (sy, sx) feeds tap (r, s) through p' = p + (r - 1, s - 1)
    const float wr[3] = {wyf(sy - 1), wyf(sy), wyf(sy + 1)};
    if (wr[0] == 0.f && wr[1] == 0.f && wr[2] == 0.f) continue;            // uniform over the block
    const int64_t rowoff = (((int64_t)b * Ho + sy) * Wo) * N + c;
    auto column = [&](int sx, float w0, float w1, float w2) {
      if (sx < 0 || sx >= Wo || (w0 == 0.f && w1 == 0.f && w2 == 0.f)) return;
      float v[VEC];
      V::ld(dy, rowoff + (int64_t)sx * N, v);
      const float wc[3] = {w0, w1, w2};
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        float gx[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) gx[e] = wc[s3] * v[e];
#pragma unroll
        for (int r3 = 0; r3 < 3; ++r3)
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[r3 * 3 + s3][e] += wr[r3] * gx[e];
      }
    };
    // p'_x = sx + s - 1 = ox_lo + (j + s - 2): window column index k = j + s - 2
    if constexpr (WXL) {
#pragma unroll 2
      for (int j = 0; j < NX + 2; ++j) column(ox_lo - 1 + j, wxs[j][threadIdx.x], wxs[j + 1][threadIdx.x], wxs[j + 2][threadIdx.x]);
    } else {
#pragma unroll
      for (int j = 0; j < NX + 2; ++j) {
        const float w0 = (j - 2 >= 0 && j - 2 < NX) ? wx[j - 2 < 0 ? 0 : (j - 2 >= NX ? NX - 1 : j - 2)] : 0.f;
        const float w1 = (j - 1 >= 0 && j - 1 < NX) ? wx[j - 1 < 0 ? 0 : (j - 1 >= NX ? NX - 1 : j - 1)] : 0.f;
        const float w2 = (j < NX) ? wx[j >= NX ? NX - 1 : j] : 0.f;
        column(ox_lo - 1 + j, w0, w1, w2);
      }
    }
  }
  const int64_t obase = ((((int64_t)b * Hi + iy) * Wi) + ix) * (9 * (int64_t)N) + c;
#pragma unroll
  for (int t = 0; t < 9; ++t) V::st(g, obase + (int64_t)(8 - t) * N, acc[t]);
}

// The same gather in two separable passes (U = U_y (x) U_x and S_t = S_r (x) S_s, so U^T S_t^T = (U_y^T S_r^T) (x) (U_x^T S_s^T)):
//   pass 1 (rows):    R_r[b, iy, ox] = sum_sy U_y[sy + r - 1, iy] dy[b, sy, ox]          three maps [3][B, Hi, Wo, N]
//   pass 2 (columns): G_(r,s)[b, iy, ix] = sum_sx U_x[sx + s - 1, ix] R_r[b, iy, sx]     nine maps, tap block 8 - (3 r + s)
// One load feeds 3 FMAs per channel in each pass instead of 12 per load over a (rows x columns) window: the single-pass kernel
// is VALU-bound at resize factors of 4 and 8 (1.35 ms for the 1 GB gradient of the neck's x4 level against 0.3 ms of
// memory time); the price is the intermediate (3 / factor of dy's size, written and read once, in dy's dtype).
template <typename V, int VEC>
__global__ __launch_bounds__(256) void resize_conv3x3_bwd_rows_kernel(const void* __restrict__ dy, int Ho, int Wo, int N,
                                                                      void* r3, int Hi, int64_t plane) {
  const int cv = N / VEC;
  const int nblk = gridDim.x * gridDim.y;
  const int id = blockIdx.y * gridDim.x + blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = id & 7, idx = id >> 3;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int brow = lid / gridDim.x, bx = lid - brow * gridDim.x;
  const int j0 = bx * 256 + threadIdx.x;
  if (j0 >= Wo * cv) return;
  const int b = brow / Hi, iy = brow - b * Hi;
  const int ox = j0 / cv, c = (j0 - ox * cv) * VEC;
  const float ry = (float)Hi / (float)Ho;
  int oy_lo = (int)floorf(((float)iy - 0.5f) / ry - 0.5f) - 1, oy_hi = (int)ceilf(((float)iy + 1.5f) / ry - 0.5f) + 1;
  oy_lo = oy_lo < 0 ? 0 : oy_lo;
  oy_hi = oy_hi > Ho - 1 ? Ho - 1 : oy_hi;
  auto wyf = [&](int oy) -> float {
    if (oy < oy_lo || oy > oy_hi) return 0.f;
    int y0, y1; float ly;
    src_index(ry, oy, Hi, y0, y1, ly);
    return (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
  };
  float acc[3][VEC];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[t][e] = 0.f;
  const int sy_lo = oy_lo > 0 ? oy_lo - 1 : 0, sy_hi = oy_hi < Ho - 1 ? oy_hi + 1 : Ho - 1;
  for (int sy = sy_lo; sy <= sy_hi; ++sy) {
    const float wr[3] = {wyf(sy - 1), wyf(sy), wyf(sy + 1)};
    if (wr[0] == 0.f && wr[1] == 0.f && wr[2] == 0.f) continue;
    float v[VEC];
    V::ld(dy, (((int64_t)b * Ho + sy) * Wo + ox) * N + c, v);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[t][e] += wr[t] * v[e];
  }
  const int64_t o = (((int64_t)b * Hi + iy) * Wo + ox) * N + c;
#pragma unroll
  for (int t = 0; t < 3; ++t) V::st(r3, o + t * plane, acc[t]);
}

template <typename V, int VEC, int NX, bool WXL>
__global__ __launch_bounds__(256) void resize_conv3x3_bwd_cols_kernel(const void* __restrict__ r3, int Wo, int N, void* g, int Hi,
                                                                      int Wi, int64_t plane) {
  __shared__ float wxs[WXL ? NX + 4 : 1][WXL ? 256 : 1];
  const int cv = N / VEC;
  const int j0 = blockIdx.x * 256 + threadIdx.x;
  if (j0 >= Wi * cv) return;
  const int brow = blockIdx.y;                         // = b * Hi + iy
  const int ix = j0 / cv, c = (j0 - ix * cv) * VEC;
  const float rx = (float)Wi / (float)Wo;
  int ox_lo = (int)floorf(((float)ix - 0.5f) / rx - 0.5f) - 1, ox_hi = (int)ceilf(((float)ix + 1.5f) / rx - 0.5f) + 1;
  ox_lo = ox_lo < 0 ? 0 : ox_lo;
  ox_hi = ox_hi > Wo - 1 ? Wo - 1 : ox_hi;
  auto wxf = [&](int k) -> float {
    const int ox = ox_lo + k;
    if (ox > ox_hi) return 0.f;
    int x0, x1; float lx;
    src_index(rx, ox, Wi, x0, x1, lx);
    return (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
  };
  float wx[WXL ? 1 : NX];
  if constexpr (WXL) {
    wxs[0][threadIdx.x] = 0.f; wxs[1][threadIdx.x] = 0.f; wxs[NX + 2][threadIdx.x] = 0.f; wxs[NX + 3][threadIdx.x] = 0.f;
    for (int k = 0; k < NX; ++k) wxs[k + 2][threadIdx.x] = wxf(k);
  } else {
#pragma unroll
    for (int k = 0; k < NX; ++k) wx[k] = wxf(k);
  }
  float acc[9][VEC];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[t][e] = 0.f;
  const int64_t rowoff = (int64_t)brow * Wo * N + c;
  auto column = [&](int sx, float w0, float w1, float w2) {
    if (sx < 0 || sx >= Wo || (w0 == 0.f && w1 == 0.f && w2 == 0.f)) return;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float v[VEC];
      V::ld(r3, rowoff + r * plane + (int64_t)sx * N, v);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        acc[r * 3 + 0][e] += w0 * v[e];
        acc[r * 3 + 1][e] += w1 * v[e];
        acc[r * 3 + 2][e] += w2 * v[e];
      }
    }
  };
  if constexpr (WXL) {
#pragma unroll 2
    for (int j = 0; j < NX + 2; ++j) column(ox_lo - 1 + j, wxs[j][threadIdx.x], wxs[j + 1][threadIdx.x], wxs[j + 2][threadIdx.x]);
  } else {
#pragma unroll
    for (int j = 0; j < NX + 2; ++j) {
      const float w0 = (j - 2 >= 0 && j - 2 < NX) ? wx[j - 2 < 0 ? 0 : (j - 2 >= NX ? NX - 1 : j - 2)] : 0.f;
      const float w1 = (j - 1 >= 0 && j - 1 < NX) ? wx[j - 1 < 0 ? 0 : (j - 1 >= NX ? NX - 1 : j - 1)] : 0.f;
      const float w2 = (j < NX) ? wx[j >= NX ? NX - 1 : j] : 0.f;
      column(ox_lo - 1 + j, w0, w1, w2);
    }
  }
  const int64_t obase = ((int64_t)brow * Wi + ix) * (9 * (int64_t)N) + c;
#pragma unroll
  for (int t = 0; t < 9; ++t) V::st(g, obase + (int64_t)(8 - t) * N, acc[t]);
}

// FORWARD of  y = conv3x3(pad 1)(bilinear_resize(x))  at LOW resolution (the transpose of the gather above).
//   y = sum_t W_t (S_t U x) = sum_t S_t U (W_t x)        (U, S_t act on pixels, W_t on channels: they commute)
// so the nine tap products z_t = W_t x are ONE 1x1 convolution over the low-resolution pixels with 9 N output channels
// (1 / factor^2 of the MACs of the convolution on the upsampled map), and what is left is this kernel:
//   y[b, oy, ox, n] = sum_{r,s} [0 <= oy+r-1 < Ho][0 <= ox+s-1 < Wo]  bilinear(z_(r,s))[oy + r - 1, ox + s - 1]
// z dense [B, Hi, Wi, 9 N], tap block t = 3 r + s.  Up to three sources of different (integer) factors are summed into one
// output (UperNet's fpn_bottleneck over its upsampled levels); the upsampled maps and the concat buffer never exist.
// A thread owns RUN consecutive output pixels of one row x VEC channels.  For a source of factor F the run is RUN / F cells
// (one cell = the F outputs that share a low-resolution column triple): per cell the nine taps are first combined
// vertically into h[s][column] (54 vector loads: 3 filter rows x 2 source rows x 3 taps x 3 columns), then every output of
// the cell takes 9 multiply-adds from h -- 54/F + 9 multiply-adds per output element instead of 36 for pixel-by-pixel
// sampling.  f32 accumulation in a fixed order, no atomics.
struct TapSrc { const void* p[3]; int H[3], W[3], F[3]; };

template <typename V, int VEC, int RUN, int F>
__device__ __forceinline__ void tapsum_source(float (&acc)[RUN][VEC], const void* __restrict__ z, int Hi, int Wi, int N, int b,
                                              int oy, int Ho, int ox0, int Wo, int c) {
  constexpr int CELLS = RUN / F;
  const float ry = (float)Hi / (float)Ho, rx = (float)Wi / (float)Wo;
  const int pix = 9 * N;                                           // elements per low-resolution pixel
  const char* zb = (const char*)z + (int64_t)b * Hi * Wi * pix * (int64_t)sizeof(typename V::elem);   // this image (uniform)
#pragma unroll 1
  for (int cell = 0; cell < CELLS; ++cell) {
    const int ix = ox0 / F + cell;
    float wx[F + 2][3];                            // wx[pos][k]: weight of source column ix - 1 + k for position ix*F - 1 + pos
#pragma unroll
    for (int pos = 0; pos < F + 2; ++pos) {
      const int px = ix * F - 1 + pos;
      int x0 = -9, x1 = -9; float lx = 0.f;
      const bool in = px >= 0 && px < Wo;
      if (in) src_index(rx, px, Wi, x0, x1, lx);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int col = ix - 1 + k;
        wx[pos][k] = in ? ((x0 == col ? 1.f - lx : 0.f) + (x1 == col ? lx : 0.f)) : 0.f;
      }
    }
    int coff[3];                                   // out-of-image columns are clamped: their weights are zero
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int col = ix - 1 + k;
      col = col < 0 ? 0 : (col > Wi - 1 ? Wi - 1 : col);
      coff[k] = col * pix + c;
    }
    float h[3][3][VEC];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) h[s][k][e] = 0.f;
#pragma unroll 1
    for (int r = 0; r < 3; ++r) {                  // filter row r reads output row oy + r - 1 of the (virtual) upsampled map
      const int py = oy + r - 1;
      if (py < 0 || py >= Ho) continue;            // the convolution's zero padding (uniform over the block)
      int y0, y1; float ly;
      src_index(ry, py, Hi, y0, y1, ly);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const float w = half ? ly : 1.f - ly;
        if (w == 0.f) continue;                    // uniform
        const int rowoff = (half ? y1 : y0) * Wi * pix + 3 * r * N;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            float v[VEC];
            V::ld(zb, rowoff + coff[kx] + s * N, v);
#pragma unroll
            for (int e = 0; e < VEC; ++e) h[s][kx][e] += w * v[e];
          }
        }
      }
    }
    float o[F][VEC];
#pragma unroll
    for (int j = 0; j < F; ++j) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[j][e] = 0.f;
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int e = 0; e < VEC; ++e) o[j][e] += wx[j + s][kx] * h[s][kx][e];
    }
#pragma unroll
    for (int cc = 0; cc < CELLS; ++cc) {
      if (cell == cc) {                            // uniform: keeps the accumulator indices compile-time constants
#pragma unroll
        for (int j = 0; j < F; ++j)
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[cc * F + j][e] += o[j][e];
      }
    }
  }
}

// grid = (ceil(Wo / RUN * N / VEC / 256), B * Ho); addvec (f32 [N], may be null) is added to every output (conv bias, or
// the folded BatchNorm shift in eval mode), then the optional ReLU.
template <typename V, int VEC, int RUN>
__global__ __launch_bounds__(256) void resize_conv3x3_fwd_sum_kernel(TapSrc src, int nsrc, int N, void* out, int Ho, int Wo,
                                                                     const float* __restrict__ addvec, int relu) {
  const int cv = N / VEC;
  const int nblk = gridDim.x * gridDim.y;
  const int id = blockIdx.y * gridDim.x + blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = id & 7, idx = id >> 3;      // XCD-major: neighbouring rows share an L2
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int brow = lid / gridDim.x, bx = lid - brow * gridDim.x;
  const int j0 = bx * 256 + threadIdx.x;
  if (j0 >= (Wo / RUN) * cv) return;
  const int b = brow / Ho, oy = brow - b * Ho;
  const int xr = j0 / cv, c = (j0 - xr * cv) * VEC;
  const int ox0 = xr * RUN;
  float acc[RUN][VEC];
#pragma unroll
  for (int p = 0; p < RUN; ++p)
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[p][e] = 0.f;
  for (int k = 0; k < nsrc; ++k) {
    const int F = src.F[k];
    if constexpr (RUN >= 8) { if (F == 8) tapsum_source<V, VEC, RUN, 8>(acc, src.p[k], src.H[k], src.W[k], N, b, oy, Ho, ox0, Wo, c); }
    if constexpr (RUN >= 4) { if (F == 4) tapsum_source<V, VEC, RUN, 4>(acc, src.p[k], src.H[k], src.W[k], N, b, oy, Ho, ox0, Wo, c); }
    if (F == 2) tapsum_source<V, VEC, RUN, 2>(acc, src.p[k], src.H[k], src.W[k], N, b, oy, Ho, ox0, Wo, c);
  }
  float add[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) add[e] = addvec ? addvec[c + e] : 0.f;
  const int64_t obase = (((int64_t)b * Ho + oy) * Wo + ox0) * N + c;
#pragma unroll
  for (int p = 0; p < RUN; ++p) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      acc[p][e] += add[e];
      if (relu) acc[p][e] = fmaxf(acc[p][e], 0.f);
    }
    V::st(out, obase + (int64_t)p * N, acc[p]);
  }
}

// ---- the same sum on the matrix cores (bf16, N % 64 == 0): the production form.
// The pixel-by-pixel kernel above reads every z vector ~24 times through L1/L2 (13.5 x the output bytes: measured 1.5 ms for the
// neck's x4 level against 0.3 ms of HBM time).  Per 4 x 4 output patch the sum is a small dense product that is THE SAME for
// every channel:   out[p, n] = sum_k A[p, k] z[k, n],   k = (filter tap (r, s), window pixel (wy, dx)),
// where the window is the 3 x 3 (factor 2: 4 x 4) low-resolution pixels the patch's 6 x 6 tap positions interpolate from and
// A[p, k] = Uy[oy + r - 1, wy] * Ux[ox + s - 1, dx] (zero outside the output = the convolution's padding).  A factorises, so k
// is laid out as a = (r, wy) outer, b = (s, dx) inner padded to 16: K = 16 * 3 WIN, five or six v_mfma_f32_16x16x32_bf16
// steps per 16 channels.  The weights are bilinear fractions (k/4, k/8, k/16 and their pairwise products): exact in bf16.
// A block = 4 output rows x 4 PXB columns x 64 channels of one image: it stages the low-resolution window (rows of 64
// channels = 128 B, every (pixel, tap) one row) by LDS-DMA as it lies in memory, builds the two 1-D weight tables in LDS,
// and each wave runs its patches: z fragments (rows = channels) through ds_read_b64_tr_b16, weight fragments (columns =
// pixels) from the tables.  Sources are staged one after the other into the same LDS and accumulate in registers; the
// result leaves through an LDS transpose as whole 128-byte lines.  f32 accumulation, fixed order.
typedef __attribute__((ext_vector_type(4))) short tm_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short tm_s16x8_t;
typedef __attribute__((address_space(3))) tm_s16x4_t* tm_lds_s16x4_ptr;
constexpr unsigned kTmOob = 0x80000000u;
__device__ __forceinline__ int tm_swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

struct TapMArgs {
  const uint16_t* z[3];
  int H[3], W[3], LF[3];          // low-resolution size and log2(factor) of every source
  int nsrc, N, Ho, Wo, B;
  uint16_t* out;
  const float* addvec;
  int relu;
  int nslots, slot_off[3];        // version 2: byte offsets of the window slots (two for one source, else one per source)
  int gx, nblk;                   // version 2 (persistent): patches per output row block, patches in all
};

// the two 1-D weight tables of source k for the block at (oy0, oxb0): TYs [4][12] (+ padding to 64 floats), TXs [COLS][16]
template <int PXB, int LF>
__device__ __forceinline__ void tapm_tables(const TapMArgs& a, int k, float* TYs, float* TXs, int oy0, int oxb0, int tid) {
  constexpr int WIN = LF == 1 ? 4 : 3, lf = LF;
  constexpr int COLS = 4 * PXB, NA = 3 * WIN;
  const int Hi = a.H[k], Wi = a.W[k];
  const int wr0 = (oy0 >> lf) - 1;
  const float ry = (float)Hi / (float)a.Ho, rx = (float)Wi / (float)a.Wo;
  if (tid < 48) {
    const int py = tid / 12, ai = tid - py * 12;
    float w = 0.f;
    if (ai < NA) {
      const int r = ai / WIN, wy = ai - r * WIN, pos = oy0 + py + r - 1;
      if (pos >= 0 && pos < a.Ho) {
        int y0, y1; float ly;
        src_index(ry, pos, Hi, y0, y1, ly);
        const int row = wr0 + wy;
        w = (y0 == row ? 1.f - ly : 0.f) + (y1 == row ? ly : 0.f);
      }
    }
    TYs[tid] = w;
  }
  for (int i = tid; i < COLS * 16; i += 256) {
    const int pc = i >> 4, bi = i & 15;
    float w = 0.f;
    if (bi < NA) {
      const int s3 = bi / WIN, dx = bi - s3 * WIN, pos = oxb0 + pc + s3 - 1;
      if (pos >= 0 && pos < a.Wo) {
        int x0, x1; float lx;
        src_index(rx, pos, Wi, x0, x1, lx);
        const int col = ((oxb0 + (pc & ~3)) >> lf) - 1 + dx;          // the PATCH's window starts at its own first cell - 1
        w = (x0 == col ? 1.f - lx : 0.f) + (x1 == col ? lx : 0.f);
      }
    }
    TXs[i] = w;
  }
}

// stage the window of source k, channels c0 .. c0 + 63: LDS row R = (wyi * WC + wci) * 9 + tap, 128 bytes; no wait
template <int PXB, int LF>
__device__ __forceinline__ void tapm_issue(const TapMArgs& a, int k, unsigned lds_stage, int b, int oy0, int oxb0, int c0, int lane,
                                           int wave) {
  constexpr int WIN = LF == 1 ? 4 : 3, lf = LF;
  constexpr int COLS = 4 * PXB;
  const int Hi = a.H[k], Wi = a.W[k], N = a.N;
  constexpr int WC = ((COLS - 4) >> lf) + WIN;        // window columns of the block (compile-time: the divisions below are cheap)
  const int wr0 = (oy0 >> lf) - 1, wcb0 = (oxb0 >> lf) - 1;
  const srd_t srd = make_srd(a.z[k] + (int64_t)b * Hi * Wi * 9 * N, (unsigned)((int64_t)Hi * Wi * 9 * N * 2));
  constexpr int nrows = WIN * WC * 9, npieces = (nrows + 7) >> 3;
  for (int piece = wave; piece < npieces; piece += 4) {
    const int R = piece * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ tm_swz(R);
    const int pc = R / 9, t = R - pc * 9;
    const int wyi = pc / WC, wci = pc - wyi * WC;
    const int row = wr0 + wyi, col = wcb0 + wci;
    const bool ok = R < nrows && row >= 0 && row < Hi && col >= 0 && col < Wi;
    const unsigned v = ok ? (unsigned)((((row * Wi + col) * 9 + t) * N + c0) * 2 + chunk * 16) : kTmOob;
    dma16_buf(v, srd, 0u, lds_stage + piece * 1024);
  }
}

// the wave's patches of source k from a staged window: acc[j2][nt] += z-fragments x weight-fragments
template <int PXB, int LF>
__device__ __forceinline__ void tapm_mfma(const TapMArgs& a, int k, f32x4_t (&acc)[PXB / 4][4], const float* TYs, const float* TXs,
                                          const unsigned char* stage, int lane, int wave) {
  constexpr int WIN = LF == 1 ? 4 : 3, lf = LF;
  constexpr int COLS = 4 * PXB, NA = 3 * WIN, NKS = (NA + 1) / 2;
  constexpr int WC = ((COLS - 4) >> lf) + WIN;
  int lane_mfma = lane;
  asm volatile("" : "+v"(lane_mfma));                 // keeps the fragment address arithmetic below this point (register budget)
  const int L = lane_mfma & 15, g = lane_mfma >> 4;
#pragma unroll
  for (int j2 = 0; j2 < PXB / 4; ++j2) {
    const int j = wave + 4 * j2;
    const int wcj = (4 * j) >> lf;                                    // the patch's window start inside the block's window
    // weight fragments (B operand: column = pixel L of the patch, k = 16 a + b)
    float txv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) txv[e] = TXs[(4 * j + (L & 3)) * 16 + 8 * (g & 1) + e];
    bf16x8_t wf[NKS];
    int raddr[NKS][2];                                                // LDS byte offset of this lane's 8-byte piece for channel tile 0
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int ai = 2 * ks + (g >> 1);
      const float ty = TYs[(L >> 2) * 12 + ai];                       // zero for ai >= NA (table padding)
      uint32_t pk[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) pk[e] = pack_bf16x2(ty * txv[2 * e], ty * txv[2 * e + 1]);
      wf[ks] = __builtin_bit_cast(bf16x8_t, make_uint4(pk[0], pk[1], pk[2], pk[3]));
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int bq = 8 * (g & 1) + 4 * h + (L >> 2);
        const bool ok = ai < NA && bq < NA;
        const int r = ai / WIN, wy = ai - r * WIN, s3 = bq / WIN, dx = bq - s3 * WIN;
        const int R = ok ? ((wy * WC + wcj + dx) * 9 + 3 * r + s3) : 0;   // padding slots: any staged row, their weight is zero
        // 16-byte slot = (channel >> 3) ^ swizzle(R); channel = 16 nt + 4 (L & 3): nt only flips slot bits 1-2 -> one XOR per tile
        raddr[ks][h] = R * 128 + (((((L & 3) >> 1) ^ tm_swz(R))) << 4) + (L & 1) * 8;
      }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const tm_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tm_lds_s16x4_ptr)(stage + (raddr[ks][0] ^ (nt << 5))));
        const tm_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tm_lds_s16x4_ptr)(stage + (raddr[ks][1] ^ (nt << 5))));
        const tm_s16x8_t zv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        acc[j2][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, zv), wf[ks], acc[j2][nt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);          // one channel tile's fragment reads in flight at a time (register budget)
    }
  }
}

// accumulators -> (+ addvec, ReLU) -> bf16 -> LDS tile [4 rows][COLS pixels] x 128 B, 16-byte slots swizzled by pixel
template <int PXB>
__device__ __forceinline__ void tapm_to_tile(const TapMArgs& a, f32x4_t (&acc)[PXB / 4][4], unsigned char* tile, int c0, int lane, int wave) {
  constexpr int COLS = 4 * PXB;
  const int L = lane & 15, g = lane >> 4;
#pragma unroll
  for (int j2 = 0; j2 < PXB / 4; ++j2) {
    const int pp = (L >> 2) * COLS + 4 * (wave + 4 * j2) + (L & 3);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int ch = 16 * nt + 4 * g;
      float v[4];
      const float4 av = a.addvec ? *(const float4*)(a.addvec + c0 + ch) : make_float4(0.f, 0.f, 0.f, 0.f);   // (c0 + ch) % 4 == 0
      const float add[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = acc[j2][nt][i] + add[i];
        if (a.relu) v[i] = fmaxf(v[i], 0.f);
      }
      *(uint2*)(tile + pp * 128 + ((((ch >> 3) ^ (pp & 7))) << 4) + (ch & 7) * 2) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    }
  }
}

// grid = (ceil(Wo / (4 PXB)) * N / 64, B * ceil(Ho / 4)); dynamic LDS = 256 + 256 PXB + max over the sources of the window bytes
template <int PXB>
__global__ __launch_bounds__(256, PXB == 8 ? 3 : 4) void resize_conv3x3_fwd_sum_mfma_kernel(const TapMArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int COLS = 4 * PXB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nchunks = a.N >> 6, nprow = (a.Ho + 3) >> 2;
  const int nblk = gridDim.x * gridDim.y;
  const int id = blockIdx.y * gridDim.x + blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = id & 7, idx = id >> 3;      // XCD-major: neighbouring patch rows share an L2
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int brow = lid / gridDim.x, bx = lid - brow * gridDim.x;
  const int b = brow / nprow, oy0 = (brow - b * nprow) * 4;
  const int cb = bx / nchunks, c0 = (bx - cb * nchunks) * 64, oxb0 = cb * COLS;
  float* TYs = (float*)smem;
  float* TXs = TYs + 64;
  unsigned char* stage = smem + 256 + COLS * 64;
  const unsigned lds_stage = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)stage);
  f32x4_t acc[PXB / 4][4];
#pragma unroll
  for (int j2 = 0; j2 < PXB / 4; ++j2)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[j2][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < a.nsrc; ++k) {
    if (k > 0) __syncthreads();                       // the previous source's fragment reads are done: tables and window are rewritten
    if (a.LF[k] == 1) { tapm_tables<PXB, 1>(a, k, TYs, TXs, oy0, oxb0, tid); tapm_issue<PXB, 1>(a, k, lds_stage, b, oy0, oxb0, c0, lane, wave); }
    else if (a.LF[k] == 2) { tapm_tables<PXB, 2>(a, k, TYs, TXs, oy0, oxb0, tid); tapm_issue<PXB, 2>(a, k, lds_stage, b, oy0, oxb0, c0, lane, wave); }
    else { tapm_tables<PXB, 3>(a, k, TYs, TXs, oy0, oxb0, tid); tapm_issue<PXB, 3>(a, k, lds_stage, b, oy0, oxb0, c0, lane, wave); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (a.LF[k] == 1) tapm_mfma<PXB, 1>(a, k, acc, TYs, TXs, stage, lane, wave);
    else if (a.LF[k] == 2) tapm_mfma<PXB, 2>(a, k, acc, TYs, TXs, stage, lane, wave);
    else tapm_mfma<PXB, 3>(a, k, acc, TYs, TXs, stage, lane, wave);
  }
  __syncthreads();
  // ---- epilogue through an LDS transpose (the window's memory is free now)
  unsigned char* tile = stage;
  tapm_to_tile<PXB>(a, acc, tile, c0, lane, wave);
  __syncthreads();
  for (int i = tid; i < 4 * COLS * 8; i += 256) {
    const int pp = i >> 3, chunk = i & 7;
    const int py = pp / COLS, oy = oy0 + py, ox = oxb0 + pp - py * COLS;
    if (oy < a.Ho && ox < a.Wo) {
      const uint4 v = *(const uint4*)(tile + pp * 128 + ((chunk ^ (pp & 7)) << 4));
      *(uint4*)(a.out + (((int64_t)b * a.Ho + oy) * a.Wo + ox) * a.N + c0 + chunk * 8) = v;
    }
  }
}

// ---- version 2: one block owns its 4 x 4 PXB output pixels for ALL channels and walks the 64-channel chunks (and, per chunk,
// the sources) through LDS window slots: the window of step i + 1 is in flight while step i runs on the matrix cores.
// Everything that does not depend on the channel is computed ONCE per block and kept in registers: the weight fragments
// and fragment addresses of the wave's patches, and the per-lane DMA offsets (the chunk enters as the buffer instruction's
// scalar offset).  PMC of version 1 on the neck's x4 level: 1400 VALU instructions per wave and 64-channel block -- runtime
// divisions in the DMA addressing, the table and fragment set-up -- against 40 MFMAs: VALU-bound at 2.2 TB/s.
// With `stats` the per-channel sum and sum of squares of the (bf16-rounded) outputs of the block are written as one partial
// row each ([block][2][N] f32, the layout bn_stats_final reduces): train-mode BatchNorm needs no statistics pass.
// grid = (ceil(Wo / (4 PXB)), B * ceil(Ho / 4)); dynamic LDS = NSRC * (256 + 256 PXB) tables + 512 PXB tile bytes + the window
// slots: one source -> two slots used alternately; several sources -> one slot per source (step (chunk, k) reads slot k while
// the next step's window lands in another one)
// LF0 (single source only): the factor as a template parameter -- the fragment count per channel tile, the DMA piece count and
// the window slot then need no run-time branches (0 = read a.LF[k] at run time)
template <int PXB, int NSRC, int LF0 = 0>
__global__ __launch_bounds__(256, NSRC == 1 ? 3 : 1) void resize_conv3x3_fwd_sum_mfma2_kernel(const TapMArgs a, float* stats) {
  static_assert(LF0 == 0 || NSRC == 1, "compile-time factor: one source");
  auto lf_of = [&](int k) { return LF0 ? LF0 : a.LF[k]; };
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int COLS = 4 * PXB, TBL = 256 + COLS * 64, NPW = PXB / 4;
  constexpr int MAXP = PXB == 8 ? 9 : 12;                 // DMA pieces per wave and window (factor 2 needs 4 x 16 blocks)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nchunks = a.N >> 6, nprow = (a.Ho + 3) >> 2;
  const int nblk = a.nblk;
  unsigned char* tile = smem + NSRC * TBL;
  unsigned char* ring = tile + 4 * COLS * 128;
  const unsigned lds_ring = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)ring);
  const int L = lane & 15, g = lane >> 4;
  bf16x8_t wf[NSRC][NPW][6];                              // weight fragments (B operand: column = pixel L of the patch)
  int raddr[NSRC][NPW][6][2];                             // LDS byte offsets of the z fragment pieces, channel tile 0
  unsigned doff[NSRC][MAXP];                              // DMA source offsets of this lane's pieces (channel 0), for the patch at (set_wr0, set_wc0)
  srd_t srd[NSRC];
  // PERSISTENT (round 4): a workgroup walks patches vid = blockIdx.x, + gridDim.x, ...  Everything channel-independent -- the
  // two weight tables, the weight fragments, the fragment and DMA offsets: ~6000 VALU instructions per wave with their runtime
  // divisions, as much as 26 of the 12 channel steps of a 768-channel patch (PMC: 366 M VALU instructions per launch of which
  // 115 M in the channel loop) -- is the SAME for every patch whose windows lie inside the image and whose rows have the same
  // interpolation phase: there only the buffer descriptor moves (by the distance between the patch positions), and the set-up
  // runs again only for patches that touch the image border or change phase (key < 0 / key differs).
  int set_key = -2, set_wr0[NSRC], set_wc0[NSRC];
#pragma unroll
  for (int k = 0; k < NSRC; ++k) { set_wr0[k] = 0; set_wc0[k] = 0; }
  for (int vid = blockIdx.x; vid < nblk; vid += gridDim.x) {
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = vid & 7, idx = vid >> 3;      // XCD-major: neighbouring patch rows share an L2
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int brow = lid / a.gx, cb = lid - brow * a.gx;
  const int b = brow / nprow, oy0 = (brow - b * nprow) * 4;
  const int oxb0 = cb * COLS;
  const bool full = oy0 + 4 <= a.Ho && oxb0 + COLS <= a.Wo;
  int key = full ? 0 : -1;
#pragma unroll
  for (int k = 0; k < NSRC; ++k) {
    const int lf = lf_of(k), win = lf == 1 ? 4 : 3, WC = ((COLS - 4) >> lf) + win;
    const int wr0 = (oy0 >> lf) - 1, wc0 = (oxb0 >> lf) - 1;
    if (wr0 < 0 || wc0 < 0 || wr0 + win > a.H[k] || wc0 + WC > a.W[k]) key = -1;
    else if (key >= 0) key = key * 8 + (oy0 & ((1 << lf) - 1));      // row phase (4-row patches: only factor 8 has two); columns: 16 | oxb0
  }
  if (key < 0 || key != set_key) {
  set_key = key;
  if (vid != (int)blockIdx.x) __syncthreads();            // the previous patch's fragment set-up / statistics may still read the tables
  // (the set-up derives everything from an opaque copy of the thread id: otherwise its lane-only subexpressions are hoisted in
  // front of the patch loop and stay live -- spilled -- across the channel loop)
  int tid_s = tid;
  asm volatile("" : "+v"(tid_s));
  const int lane = tid_s & 63, L = lane & 15, g = lane >> 4;
#pragma unroll
  for (int k = 0; k < NSRC; ++k) {
    float* TYs = (float*)(smem + k * TBL);
    if (lf_of(k) == 1) tapm_tables<PXB, 1>(a, k, TYs, TYs + 64, oy0, oxb0, tid_s);
    else if (lf_of(k) == 2) tapm_tables<PXB, 2>(a, k, TYs, TYs + 64, oy0, oxb0, tid_s);
    else tapm_tables<PXB, 3>(a, k, TYs, TYs + 64, oy0, oxb0, tid_s);
  }
  __syncthreads();
  // ---- channel-independent state of every source
#pragma unroll
  for (int k = 0; k < NSRC; ++k) {
    const int lf = lf_of(k), Hi = a.H[k], Wi = a.W[k];
    const bool w4 = lf == 1;
    const int win = w4 ? 4 : 3, na = 3 * win;
    const int WC = ((COLS - 4) >> lf) + win;
    const float* TYs = (const float*)(smem + k * TBL);
    const float* TXs = TYs + 64;
#pragma unroll
    for (int j2 = 0; j2 < NPW; ++j2) {
      const int j = wave + 4 * j2;
      const int wcj = (4 * j) >> lf;
      float txv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) txv[e] = TXs[(4 * j + (L & 3)) * 16 + 8 * (g & 1) + e];
#pragma unroll
      for (int ks = 0; ks < 6; ++ks) {
        const int ai = 2 * ks + (g >> 1);
        const float ty = TYs[(L >> 2) * 12 + ai];         // zero for ai >= na (table padding)
        uint32_t pk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = pack_bf16x2(ty * txv[2 * e], ty * txv[2 * e + 1]);
        wf[k][j2][ks] = __builtin_bit_cast(bf16x8_t, make_uint4(pk[0], pk[1], pk[2], pk[3]));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int bq = 8 * (g & 1) + 4 * h + (L >> 2);
          const bool ok = ai < na && bq < na;
          const int r = w4 ? ai >> 2 : ai / 3, wy = ai - r * win, s3 = w4 ? bq >> 2 : bq / 3, dx = bq - s3 * win;
          const int R = ok ? ((wy * WC + wcj + dx) * 9 + 3 * r + s3) : 0;
          raddr[k][j2][ks][h] = R * 128 + (((((L & 3) >> 1) ^ tm_swz(R))) << 4) + (L & 1) * 8;
        }
      }
    }
    const int wr0 = (oy0 >> lf) - 1, wcb0 = (oxb0 >> lf) - 1;
    set_wr0[k] = wr0; set_wc0[k] = wcb0;
    const int nrows = win * WC * 9;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int piece = wave + 4 * i;
      const int R = piece * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ tm_swz(R);
      const int pc = R / 9, t = R - pc * 9;
      const int wyi = pc / WC, wci = pc - wyi * WC;
      const int row = wr0 + wyi, col = wcb0 + wci;
      const bool ok = R < nrows && row >= 0 && row < Hi && col >= 0 && col < Wi;
      doff[k][i] = ok ? (unsigned)((((row * Wi + col) * 9 + t) * a.N) * 2 + chunk * 16) : kTmOob;
    }
  }
  }
  // the image's buffer descriptor, moved from the set-up patch to this one (interior patches only: every piece stays inside)
#pragma unroll
  for (int k = 0; k < NSRC; ++k) {
    const int lf = lf_of(k), Hi = a.H[k], Wi = a.W[k];
    const int64_t shift = ((int64_t)(((oy0 >> lf) - 1) - set_wr0[k]) * Wi + (((oxb0 >> lf) - 1) - set_wc0[k])) * 9 * a.N;
    srd[k] = make_srd(a.z[k] + (int64_t)b * Hi * Wi * 9 * a.N + shift, (unsigned)((int64_t)Hi * Wi * 9 * a.N * 2));
  }
  auto issue = [&](int it) {
    const int c = it / NSRC, k = it - c * NSRC;
    const unsigned dst = lds_ring + (NSRC == 1 ? (it & 1) * a.slot_off[1] : a.slot_off[it % a.nslots]);   // one source: two slots, slot_off[0] = 0
#pragma unroll
    for (int kk = 0; kk < NSRC; ++kk) {
      if (kk != k) continue;                               // (static source index for the register arrays)
      const int lf = lf_of(kk), win = lf == 1 ? 4 : 3;
      const int npieces = (win * (((COLS - 4) >> lf) + win) * 9 + 7) >> 3;
#pragma unroll
      for (int i = 0; i < MAXP; ++i) {
        if (wave + 4 * i < npieces) dma16_buf(doff[kk][i], srd[kk], (unsigned)(c * 128), dst + (wave + 4 * i) * 1024);
      }
    }
  };
  f32x4_t acc[NPW][4];
#pragma unroll
  for (int j2 = 0; j2 < NPW; ++j2)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[j2][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int nit = nchunks * NSRC;
  issue(0);
  for (int it = 0; it < nit; ++it) {
    const int c = it / NSRC, k = it - c * NSRC;
    // this wave's pieces of step `it` have landed.  They were issued BEFORE the previous step's output stores, and vmcnt
    // retires in order: inside the image (every thread stores exactly PXB / 2 pieces per chunk) the wait leaves those stores
    // in flight instead of exposing their acknowledgement
    if (full && it > 0 && (it - 1) % NSRC == NSRC - 1) {
      if (stats && wave < 2) {                             // (waves 0 and 1 also wrote one statistics row each)
        if constexpr (PXB == 8) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      } else {
        if constexpr (PXB == 8) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                                      // ... everyone's; nobody reads the other slot or the tile any more
    if (it + 1 < nit) issue(it + 1);
    const unsigned char* stage = ring + (NSRC == 1 ? (it & 1) * a.slot_off[1] : a.slot_off[it % a.nslots]);
#pragma unroll
    for (int kk = 0; kk < NSRC; ++kk) {
      if (kk != k) continue;
      const int nks = lf_of(kk) == 1 ? 6 : 5;
#pragma unroll
      for (int j2 = 0; j2 < NPW; ++j2) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
          for (int ks = 0; ks < 6; ++ks) {
            if (ks < nks) {
              const tm_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tm_lds_s16x4_ptr)(stage + (raddr[kk][j2][ks][0] ^ (nt << 5))));
              const tm_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tm_lds_s16x4_ptr)(stage + (raddr[kk][j2][ks][1] ^ (nt << 5))));
              const tm_s16x8_t zv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
              acc[j2][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, zv), wf[kk][j2][ks], acc[j2][nt], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (k != NSRC - 1) continue;
    // ---- the chunk is complete: transpose through the tile, store whole 128-byte lines, optional statistics
    const int c0 = c * 64;
    tapm_to_tile<PXB>(a, acc, tile, c0, lane, wave);
#pragma unroll
    for (int j2 = 0; j2 < NPW; ++j2)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[j2][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
    for (int u = 0; u < (4 * COLS * 8) / 256; ++u) {
      const int i = tid + 256 * u;
      const int pp = i >> 3, chunk = i & 7;                // chunk = tid & 7 for every u: a thread always owns the same 8 channels
      const int py = pp / COLS, oy = oy0 + py, ox = oxb0 + pp - py * COLS;
      if (full || (oy < a.Ho && ox < a.Wo)) {               // `full` is uniform: then every thread stores, and the wait above counts on it
        const uint4 v = *(const uint4*)(tile + pp * 128 + ((chunk ^ (pp & 7)) << 4));
        *(uint4*)(a.out + (((int64_t)b * a.Ho + oy) * a.Wo + ox) * a.N + c0 + chunk * 8) = v;
        if (stats) {
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
            s1[2 * e] += lo; s2[2 * e] += lo * lo;
            s1[2 * e + 1] += hi; s2[2 * e + 1] += hi * hi;
          }
        }
      }
    }
    if (stats) {
      // lanes with equal (lane & 7) own the same channels: fold lane bits 3, 4, 5, then the four waves through the (by then
      // idle) tile memory, in a fixed order
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
      }
      __syncthreads();                                    // every thread has read its pixels from the tile
      float* red = (float*)tile;                          // [wave][sum | sum of squares][64 channels]
      if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[(wave * 2) * 64 + lane * 8 + e] = s1[e]; red[(wave * 2 + 1) * 64 + lane * 8 + e] = s2[e]; }
      }
      __syncthreads();
      if (tid < 128) {
        const int which = tid >> 6, ch = tid & 63;
        const float t = ((red[(0 * 2 + which) * 64 + ch] + red[(1 * 2 + which) * 64 + ch]) + red[(2 * 2 + which) * 64 + ch]) +
                        red[(3 * 2 + which) * 64 + ch];
        stats[((int64_t)lid * 2 + which) * a.N + c0 + ch] = t;
      }
    }
  }
  __syncthreads();                                        // the next patch's first window overwrites slot 0, its statistics the tile
  }
}

// ---- version 3 (round 6): sources of factor 2 / 4 (one source, Ho % 4 == 0) or 2 + 4 + 8 (UperNet's fpn_bottleneck, Ho % 8 == 0)
// -- a ROLLING window of low-resolution rows.
// Versions 1 and 2 stage the whole 3 (4) x WC pixel window of every 4 x 16 patch: each tap-product pixel travels L2 -> LDS 4.5 times
// (factor 4) and, because the patches that share it run at different times, 2.4 times from memory (PMC: 1383 MB fetched per
// launch for 574 MB of tap products at batch 32) -- the kernel moved its over-fetched bytes at 4.5 TB/s and was memory-bound on them.
// Here a workgroup owns a COLUMN = (image, 64-channel chunk, strip of 16 output columns) and walks it top to bottom, four
// output rows per step: the window rows live in a ring of LDS row slots and only the new low-resolution rows of a later step
// are fetched (D steps ahead, by LDS-DMA).  The strips of one (image, chunk) are walked by neighbouring workgroups of ONE XCD
// at the same pace, so the one or two window columns two strips share come out of that L2.
// What makes one set-up per workgroup enough (a strip's columns all look alike):
//   * rows outside the image are CLAMPED in the DMA address instead of re-weighted: torch's clamp of the source index is the
//     same as bilinear weights on an edge-replicated map ((1 - l) z[0] + l z[0] = z[0], exact in f32), so the row weights
//     of every step are the interior ones;
//   * the convolution's zero padding touches the first and the last step only (tap row -1 of output row 0, tap row Ho of
//     output row Ho - 1): those two steps use weight fragments with these products zeroed (kept beside the interior ones).
// The K dimension is PACKED: k = (tap row, window row) * NA + (tap column, window column), NA = 3 WIN, K = NA^2 = 81 (144 for
// factor 2) in 3 (5) steps of 32 -- versions 1 / 2 pad the inner index to 16 (5 and 6 steps).  Compute was the bound of the
// first form of this kernel (fpn_bottleneck: 809 us of matrix-core + LDS work against 608 us for all its memory traffic).
// A step's output leaves through the LDS tile at the TOP of the next step and has the whole step to be acknowledged.
// D = steps the row fetches run ahead.  One step ahead leaves a workgroup one memory latency per step (factor 4, four workgroups
// per CU: 4.1 us per step and workgroup, 3.1 TB/s); D > 1: the wait at the top of a step leaves the later fetches -- and the
// stores between them; every wave issues the same number in a strip that lies inside the image -- in flight (vmcnt retires in
// order).
// STATS (one source): per-channel sum and sum of squares of the bf16-rounded outputs on the matrix cores, straight from the
// tile: with the tile fragment F (K = 32 pixels x 16 channels) as BOTH operands the diagonal of F^T F is the sum of squares,
// ones^T F the sum; the accumulators run over the whole column, so a launch writes B * strips partial rows (version 2: one per
// patch = 36 x as many, reduced by shuffles and two extra barriers per step: 1510 us with statistics against 1082 us without
// at batch 64).
// Three sources: one ring per source and one set of accumulators.  Factor 8: a step of four output rows is half a cell -- two
// row phases with their own weight fragments, a new window row every other step.  One workgroup per CU (the three rings are
// 96 + 35 + 20 KiB at D = 2); pieces a wave does not have, and the factor-8 source's idle steps, are fetches whose lanes are all
// out of range (zeros into a spare KiB, no memory traffic), so one vmcnt immediate serves every wave and step.
template <int LF, int D, int NW = 4>                       // NW: waves of the workgroup (4: a wave owns a patch; 8: a patch and half of its channel tiles)
struct RollSrc {
  static constexpr int WIN = LF == 1 ? 4 : 3, NPH = LF == 3 ? 2 : 1;
  static constexpr int NA = 3 * WIN, KR = NA * NA, NKS = (KR + 31) / 32;
  static constexpr int WC = (12 >> LF) + WIN, ROWS = WC * 9, PIECES = (ROWS + 7) / 8, SLOT = PIECES * 1024;
  static constexpr int RPS = LF == 1 ? 2 : 1;              // new window rows of a step that fetches
  static constexpr int NSLOT = WIN + (LF == 3 ? (D + 1) / 2 : D * RPS);
  static constexpr int MAXP = (PIECES + NW - 1) / NW, NDMA = RPS * MAXP, BYTES = NSLOT * SLOT;
  static constexpr int NMW = (NKS + 3) / 4;
  bf16x8_t wf[NPH][NKS];                                   // weight fragments (B operand: column = pixel of the patch)
  unsigned mtop[NMW], mbot[NMW];                           // bit 8 (ks % 4) + e of word ks / 4: element e of K-step ks is zero in the first / last step
  unsigned araddr[NKS][2];                                 // ring-relative byte offset of the lane's two fragment pieces, window row 0 of the step in slot 0
  unsigned doff[MAXP];
  srd_t srd;
  unsigned lds;                                            // LDS address of the ring
  int Hi;
  int64_t row_bytes;

  // the weight tables live in `tab` (>= 128 + 256 floats) during the call; two barriers inside
  __device__ __forceinline__ void setup(float* tab, unsigned lds_ring, int Hi_, int Wi, int Wo, int N, int oxb0, int tid, int wave) {
    const int lane = tid & 63, L = lane & 15, g = lane >> 4;
    Hi = Hi_;
    lds = lds_ring;
    row_bytes = (int64_t)Wi * 9 * N * 2;
    float* TY = tab;                                       // [NPH][4][12] (64 floats per phase)
    float* TX = tab + 128;                                 // [16][16]
    if (tid < 48 * NPH) {
      const int ph = tid / 48, q = tid - ph * 48, py = q / 12, ai = q - py * 12;
      float w = 0.f;
      if (ai < NA) {
        const int r = ai / WIN, wy = ai - r * WIN;
        const float s = ((float)(4 * ph + py + r - 1) + 0.5f) * (1.f / (float)(1 << LF)) - 0.5f;   // relative to the cell, not clamped
        const float fl = floorf(s);
        const int y0 = (int)fl, rr = wy - 1;
        const float ly = s - fl;
        w = (y0 == rr ? 1.f - ly : 0.f) + (y0 + 1 == rr ? ly : 0.f);
      }
      TY[ph * 64 + q] = w;
    }
    if (tid < 256) {
      const int pc = tid >> 4, bi = tid & 15;
      float w = 0.f;
      if (bi < NA) {
        const int s3 = bi / WIN, dx = bi - s3 * WIN, pos = oxb0 + pc + s3 - 1;
        if (pos >= 0 && pos < Wo) {
          int x0, x1; float lx;
          src_index((float)Wi / (float)Wo, pos, Wi, x0, x1, lx);
          const int col = ((oxb0 + (pc & ~3)) >> LF) - 1 + dx;
          w = (x0 == col ? 1.f - lx : 0.f) + (x1 == col ? lx : 0.f);
        }
      }
      TX[tid] = w;
    }
    __syncthreads();
    const int pw = wave & 3;                               // the wave's patch
    const int wcj = (4 * pw) >> LF, wcb0 = (oxb0 >> LF) - 1;
    const int py = L >> 2, px = 4 * pw + (L & 3);
#pragma unroll
    for (int q = 0; q < NMW; ++q) { mtop[q] = 0u; mbot[q] = 0u; }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      float wv[NPH][8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 32 * ks + 8 * g + e;
        const bool ok = k < KR;
        const int ai = ok ? k / NA : 0, bi = ok ? k - ai * NA : 0;
        const float tx = TX[px * 16 + bi];
        const int r = ai / WIN;
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) wv[ph][e] = ok ? TY[ph * 64 + py * 12 + ai] * tx : 0.f;
        if (ok && py == 0 && r == 0) mtop[ks >> 2] |= 1u << (8 * (ks & 3) + e);      // tap row -1 of output row 0
        if (ok && py == 3 && r == 2) mbot[ks >> 2] |= 1u << (8 * (ks & 3) + e);      // tap row Ho of output row Ho - 1
      }
#pragma unroll
      for (int ph = 0; ph < NPH; ++ph)
        wf[ph][ks] = __builtin_bit_cast(bf16x8_t, make_uint4(pack_bf16x2(wv[ph][0], wv[ph][1]), pack_bf16x2(wv[ph][2], wv[ph][3]),
                                                             pack_bf16x2(wv[ph][4], wv[ph][5]), pack_bf16x2(wv[ph][6], wv[ph][7])));
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = 32 * ks + 8 * g + 4 * h + (L >> 2);  // the fragment row this lane supplies (ds_read_b64_tr_b16: four rows x four channel quads per 16 lanes)
        const bool ok = k < KR;
        const int ai = ok ? k / NA : 0, bi = ok ? k - ai * NA : 0;
        const int r = ai / WIN, wy = ai - r * WIN, s3 = bi / WIN, dx = bi - s3 * WIN;
        const int R = ok ? ((wcj + dx) * 9 + 3 * r + s3) : 0;          // padding: any staged row, its weight is zero
        araddr[ks][h] = (unsigned)(wy * SLOT + R * 128 + (((((L & 3) >> 1) ^ tm_swz(R))) << 4) + (L & 1) * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int piece = wave + NW * i;
      const int R = piece * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ tm_swz(R);
      const int pc = R / 9, t = R - pc * 9;
      const int col = wcb0 + pc;
      const bool ok = piece < PIECES && R < ROWS && col >= 0 && col < Wi;
      doff[i] = ok ? (unsigned)((((col * 9 + t) * N) * 2) + chunk * 16) : kTmOob;
    }
    __syncthreads();
  }

  __device__ __forceinline__ void set_image(const uint16_t* z, int b) {
    srd = make_srd(z + (int64_t)b * Hi * (row_bytes / 2), (unsigned)((int64_t)Hi * row_bytes));
  }
  __device__ __forceinline__ void issue_row(int cc, int y, int wave, unsigned lds_dummy, bool pad) const {
    const int yc = y < 0 ? 0 : (y > Hi - 1 ? Hi - 1 : y);
    const int slot = (y + 1) % NSLOT;
    const unsigned soff = (unsigned)((int64_t)yc * row_bytes) + (unsigned)(cc * 128);
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      if (wave + NW * i < PIECES) dma16_buf(doff[i], srd, soff, lds + slot * SLOT + (wave + NW * i) * 1024);
      else if (pad) dma16_buf(kTmOob, srd, 0u, lds_dummy);
    }
  }
  // the window rows step s adds (step 0: the whole window); pad: always NDMA instructions per wave
  __device__ __forceinline__ void issue_step(int cc, int s, int wave, unsigned lds_dummy, bool pad) const {
    if (s == 0) {
#pragma unroll
      for (int y = 0; y < WIN; ++y) issue_row(cc, y - 1, wave, lds_dummy, false);
    } else if (LF == 1) {
      issue_row(cc, 2 * s + 1, wave, lds_dummy, pad);
      issue_row(cc, 2 * s + 2, wave, lds_dummy, pad);
    } else if (LF == 2) {
      issue_row(cc, s + 1, wave, lds_dummy, pad);
    } else {
      if ((s & 1) == 0) issue_row(cc, (s >> 1) + 1, wave, lds_dummy, pad);
      else if (pad) {
#pragma unroll
        for (int i = 0; i < MAXP; ++i) dma16_buf(kTmOob, srd, 0u, lds_dummy);
      }
    }
  }
  // step i of the walk: acc += this source's share, for the NTN channel tiles nt0 .. of the wave's patch.  FIRST: the accumulators
  // are not initialised yet and start from addv.  ring = generic pointer to the ring.  PD = K-steps the fragment reads run ahead
  // of the matrix cores.
  template <bool FIRST, int PD, int NTN>
  __device__ __forceinline__ void compute(const unsigned char* ring, int i, bool top, bool bot, int nt0, f32x4_t (&acc)[NTN],
                                          const f32x4_t (&addv)[NTN]) const {
    const int cell0 = LF == 1 ? 2 * i : (LF == 2 ? i : (i >> 1));
    unsigned s0b = (unsigned)((cell0 % NSLOT) * SLOT);     // byte offset of the slot of the step's first window row
    asm volatile("" : "+s"(s0b));                          // (opaque: no unrolling over the ring's period with every address kept)
    auto run = [&](auto which, auto phase) {
      constexpr int PH = decltype(phase)::value, WH = decltype(which)::value;   // WH: 0 interior, 1 first step, 2 last step
      tm_s16x8_t zb[PD + 1][NTN];
      auto load = [&](int ks, tm_s16x8_t (&zf)[NTN]) {
        unsigned ad[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const unsigned v = araddr[ks][h] + s0b;          // (window row + first slot) mod NSLOT: the offset stays below 2 BYTES
          const unsigned t = v - (unsigned)BYTES;
          ad[h] = v < t ? v : t;                           // unsigned min: t wrapped around when v < BYTES
        }
#pragma unroll
        for (int j = 0; j < NTN; ++j) {
          const unsigned xm = (unsigned)((nt0 + j) << 5);
          const tm_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tm_lds_s16x4_ptr)(ring + (ad[0] ^ xm)));
          const tm_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tm_lds_s16x4_ptr)(ring + (ad[1] ^ xm)));
          zf[j] = tm_s16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      };
#pragma unroll
      for (int ks = 0; ks < PD; ++ks)
        if (ks < NKS) load(ks, zb[ks % (PD + 1)]);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        if (ks + PD < NKS) load(ks + PD, zb[(ks + PD) % (PD + 1)]);
        bf16x8_t w = wf[PH][ks];
        if constexpr (WH != 0) {                           // the zero padding: two steps of a column pay ~25 instructions per K-step here
          const unsigned bits = ((WH == 1 ? mtop[ks >> 2] : mbot[ks >> 2]) >> (8 * (ks & 3))) & 0xffu;
          const uint4 wz = __builtin_bit_cast(uint4, w);
          unsigned wr[4] = {wz.x, wz.y, wz.z, wz.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned m2 = (bits >> (2 * q)) & 3u;
            wr[q] &= ~(((m2 & 1u) * 0xffffu) | ((m2 >> 1) * 0xffff0000u));
          }
          w = __builtin_bit_cast(bf16x8_t, make_uint4(wr[0], wr[1], wr[2], wr[3]));
        }
#pragma unroll
        for (int j = 0; j < NTN; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, zb[ks % (PD + 1)][j]), w, (FIRST && ks == 0) ? addv[j] : acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // (the walk has at least two steps: the first one is never the last)
    if (top) run(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
    else if (bot) run(std::integral_constant<int, 2>{}, std::integral_constant<int, NPH - 1>{});
    else if (NPH == 2 && (i & 1)) run(std::integral_constant<int, 0>{}, std::integral_constant<int, NPH - 1>{});
    else run(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  }
};

struct TapRArgs {
  const uint16_t* z;
  uint16_t* out;
  const float* addvec;
  float* stats;
  int Hi, Wi, N, Ho, Wo, B, relu;
  int nstrips, G, ncols, nchunks;   // strips per image row, column groups per XCD, columns = B * chunks, chunks = N / 64
};

// accumulators of a step -> (ReLU) -> bf16 -> the LDS tile [4 rows][16 pixels] x 128 B, 16-byte slots swizzled by pixel
template <int NTN>
__device__ __forceinline__ void roll_to_tile(unsigned char* tile, const f32x4_t (&acc)[NTN], int relu, int oxb0, int Wo, int lane, int pw, int nt0) {
  int t = lane;
  asm volatile("" : "+v"(t));
  const int l = t & 15, gq = t >> 4;
  const int pp = (l >> 2) * 16 + 4 * pw + (l & 3);
  const bool pxvalid = oxb0 + 4 * pw + (l & 3) < Wo;
#pragma unroll
  for (int j = 0; j < NTN; ++j) {
    const int ch = 16 * (nt0 + j) + 4 * gq;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = acc[j][e];
      if (relu) v[e] = fmaxf(v[e], 0.f);
      v[e] = pxvalid ? v[e] : 0.f;                         // columns beyond the image: zeros (not stored; the statistics count them as nothing)
    }
    *(uint2*)(tile + pp * 128 + ((((ch >> 3) ^ (pp & 7))) << 4) + (ch & 7) * 2) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
  }
}

template <int LF, bool STATS, int D>
__global__ __launch_bounds__(256, LF == 2 ? 3 : 2) void resize_conv3x3_fwd_sum_roll_kernel(const TapRArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using S = RollSrc<LF, D>;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = lane & 15, g = lane >> 4;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int grp = idx / a.nstrips, strip = idx - grp * a.nstrips;
  const int oxb0 = strip * 16;
  unsigned char* tile = smem;                              // [4 rows][16 pixels] x 128 B; the two weight tables during the set-up
  const unsigned char* ring = smem + 4 * 16 * 128;
  S s;
  s.setup((float*)tile, (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)(smem + 4 * 16 * 128)),
          a.Hi, a.Wi, a.Wo, a.N, oxb0, tid, wave);
  const int nsteps = a.Ho >> 2;
  const bool fullstrip = oxb0 + 16 <= a.Wo;                // every thread stores its two pieces of every tile
  const int npw = (S::PIECES - wave + 3) / 4;              // DMA instructions of this wave per window row
  const int gg = xcd * a.G + grp, gstride = 8 * a.G;
  bool primed = false;
  for (int col = gg; col < a.ncols; col += gstride) {
    const int b = col / a.nchunks, c = col - b * a.nchunks;
    if (!primed) {
      s.set_image(a.z, b);
#pragma unroll
      for (int q = 0; q < D; ++q)
        if (q < nsteps) s.issue_step(c, q, wave, 0u, false);
    }
    // the accumulators start from the per-channel addend (kept in registers for the column: no ordinary load -- whose wait the
    // compiler would place without knowing of the fetches in flight -- inside the step loop)
    f32x4_t addv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const float4 v = a.addvec ? *(const float4*)(a.addvec + c * 64 + 16 * nt + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
      addv[nt] = f32x4_t{v.x, v.y, v.z, v.w};
    }
    f32x4_t sd1 = {0.f, 0.f, 0.f, 0.f}, sd2 = {0.f, 0.f, 0.f, 0.f};
    uint16_t* obase = a.out + (int64_t)b * a.Ho * a.Wo * a.N + c * 64;
    auto flush = [&](int step) {                           // the tile of `step` -> global memory (+ statistics)
      int t = tid;
      asm volatile("" : "+v"(t));                          // (addresses from an opaque thread id: not kept in registers across the steps)
      uint16_t* orow = obase + (int64_t)step * 4 * a.Wo * a.N;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = t + 256 * u, pp = i >> 3, chunk = i & 7;
        const uint4 v = *(const uint4*)(tile + pp * 128 + ((chunk ^ (pp & 7)) << 4));
        if (oxb0 + (pp & 15) < a.Wo) *(uint4*)(orow + ((int64_t)(pp >> 4) * a.Wo + oxb0 + (pp & 15)) * a.N + chunk * 8) = v;
      }
      if constexpr (STATS) {                               // wave w: channels 16 w .. 16 w + 15 of the chunk, all 64 pixels
        const int l = t & 15, gq = (t >> 4) & 3;
        const tm_s16x8_t one8 = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const int p0 = 32 * k2 + 8 * gq + (l >> 2), p1 = p0 + 4;
          const int base = (l & 1) * 8, s16 = 2 * wave + ((l & 3) >> 1);
          const tm_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tm_lds_s16x4_ptr)(tile + p0 * 128 + ((s16 ^ (p0 & 7)) << 4) + base));
          const tm_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tm_lds_s16x4_ptr)(tile + p1 * 128 + ((s16 ^ (p1 & 7)) << 4) + base));
          const tm_s16x8_t zv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          const bf16x8_t f = __builtin_bit_cast(bf16x8_t, zv);
          sd1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, one8), f, sd1, 0, 0, 0);
          sd2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, f, sd2, 0, 0, 0);
        }
      }
    };
#pragma unroll 1
    for (int i = 0; i < nsteps; ++i) {
      // (lgkmcnt(0) in every wait: hipcc drops its own LDS wait in front of the barrier below -- behind these asm statements the
      // last ds_write of the previous step's tile was still in flight when other waves read the tile: one patch row of the last
      // channel tile wrong in 1 of 4 launches)
      // this wave's pieces of step i's rows have landed.  Steady state (the fetch of step i was followed by two stores, then
      // D - 1 times by a fetch and two stores): those may stay in flight
      if (D > 1 && fullstrip && i > D && i + D <= nsteps) {
        if (npw == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 + (D - 1) * (2 * S::RPS + 2)) : "memory");
        else if (npw == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 + (D - 1) * (1 * S::RPS + 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __syncthreads();                                    // ... everyone's; step i - 1 is computed and its tile written
      if (i + D < nsteps) s.issue_step(c, i + D, wave, 0u, false);
      if (i > 0) flush(i - 1);
      f32x4_t acc[4];
      s.template compute<true, 1, 4>(ring, i, i == 0, i == nsteps - 1, 0, acc, addv);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();                                    // everyone has flushed the previous tile
      roll_to_tile<4>(tile, acc, a.relu, oxb0, a.Wo, lane, wave, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                      // the last tile is written; nobody reads the ring any more
    const int ncol = col + gstride;
    primed = ncol < a.ncols;
    if (primed) {                                          // the next column's first rows travel under this column's tail
      const int nb = ncol / a.nchunks, nc = ncol - nb * a.nchunks;
      s.set_image(a.z, nb);
#pragma unroll
      for (int q = 0; q < D; ++q)
        if (q < nsteps) s.issue_step(nc, q, wave, 0u, false);
    }
    flush(nsteps - 1);
    if constexpr (STATS) {
      const int64_t prow = (int64_t)b * a.nstrips + strip;
      float* srow = a.stats + prow * 2 * a.N + c * 64 + 16 * wave + L;
      if (g == 0) srow[0] = sd1[0];
      if (g == (L >> 2)) {
        const int e = L & 3;
        srow[a.N] = e == 0 ? sd2[0] : (e == 1 ? sd2[1] : (e == 2 ? sd2[2] : sd2[3]));
      }
    }
  }
}

struct TapR3Args {
  const uint16_t* z[3];
  int H[3], W[3];
  uint16_t* out;
  const float* addvec;
  int N, Ho, Wo, B, relu;
  int nstrips, nstreams, ncols, nchunks, nblocks;
  int dbg;                          // tuning probes (gdl_debug_set_tapsum_roll >= 16): 16 = no matrix-core work, 32 = no row fetches after the first window, 64 = no stores
};

// 512 threads: wave w works on patch w & 3 and the channel tiles 2 (w >> 2), 2 (w >> 2) + 1 -- two waves per SIMD, so that one's LDS
// round trips and address arithmetic run under the other's matrix-core work (with 256 threads the waves were issuing 60 % of
// their cycles and parked the rest: 720 us of compute against 608 us for all memory traffic)
template <int D>
__global__ __launch_bounds__(512, 1) void resize_conv3x3_fwd_sum_roll3_kernel(const TapR3Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NW = 8, NTN = 2;
  using SA = RollSrc<1, D, NW>;
  using SB = RollSrc<2, D, NW>;
  using SC = RollSrc<3, D, NW>;
  constexpr int NDMA = SA::NDMA + SB::NDMA + SC::NDMA;     // per wave and steady-state step
  constexpr int OA = 4 * 16 * 128, OB = OA + SA::BYTES, OC = OB + SB::BYTES, OD = OC + SC::BYTES;   // rings and the spare KiB behind the tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pw = wave & 3, nt0 = 2 * (wave >> 2);
  // workgroup -> (stream of columns, strip): the workgroups of one XCD take consecutive pairs, so the strips of a stream sit on one XCD
  // (a stream that straddles two XCDs fetches the two window columns at that seam twice)
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  int pair = idx;
  for (int x = 0; x < xcd; ++x) pair += (a.nblocks - x + 7) >> 3;
  const int stream = pair / a.nstrips, strip = pair - stream * a.nstrips;
  const int oxb0 = strip * 16;
  unsigned char* tile = smem;
  auto lds_of = [](unsigned char* p) { return (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)p); };
  const unsigned lds_dummy = lds_of(smem + OD);
  SA sa; SB sb; SC sc;
  sa.setup((float*)tile, lds_of(smem + OA), a.H[0], a.W[0], a.Wo, a.N, oxb0, tid, wave);
  sb.setup((float*)tile, lds_of(smem + OB), a.H[1], a.W[1], a.Wo, a.N, oxb0, tid, wave);
  sc.setup((float*)tile, lds_of(smem + OC), a.H[2], a.W[2], a.Wo, a.N, oxb0, tid, wave);
  const int nsteps = a.Ho >> 2;
  const bool fullstrip = oxb0 + 16 <= a.Wo;
  const int g = lane >> 4;
  bool primed = false;
  for (int col = stream; col < a.ncols; col += a.nstreams) {
    const int b = col / a.nchunks, c = col - b * a.nchunks;
    if (!primed) {
      sa.set_image(a.z[0], b); sb.set_image(a.z[1], b); sc.set_image(a.z[2], b);
#pragma unroll
      for (int s = 0; s < D; ++s)
        if (s < nsteps) { sa.issue_step(c, s, wave, lds_dummy, false); sb.issue_step(c, s, wave, lds_dummy, false); sc.issue_step(c, s, wave, lds_dummy, false); }
    }
    f32x4_t addv[NTN];
#pragma unroll
    for (int j = 0; j < NTN; ++j) {
      const float4 v = a.addvec ? *(const float4*)(a.addvec + c * 64 + 16 * (nt0 + j) + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
      addv[j] = f32x4_t{v.x, v.y, v.z, v.w};
    }
    uint16_t* obase = a.out + (int64_t)b * a.Ho * a.Wo * a.N + c * 64;
    auto flush = [&](int step) {                           // 512 threads x 16 bytes = the tile
      int t = tid;
      asm volatile("" : "+v"(t));
      uint16_t* orow = obase + (int64_t)step * 4 * a.Wo * a.N;
      const int pp = t >> 3, chunk = t & 7;
      const uint4 v = *(const uint4*)(tile + pp * 128 + ((chunk ^ (pp & 7)) << 4));
      if (oxb0 + (pp & 15) < a.Wo) *(uint4*)(orow + ((int64_t)(pp >> 4) * a.Wo + oxb0 + (pp & 15)) * a.N + chunk * 8) = v;
    };
#pragma unroll 1
    for (int i = 0; i < nsteps; ++i) {
      // (lgkmcnt(0) with every wait: see the one-source kernel.)  Steady state: the fetch of step i was followed by one store, then
      // D - 1 times by a fetch (NDMA instructions in every wave) and a store
      if (D > 1 && fullstrip && i > D && i + D <= nsteps) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(1 + (D - 1) * (NDMA + 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();
      if (i + D < nsteps && !(a.dbg & 32)) {
        sa.issue_step(c, i + D, wave, lds_dummy, true); sb.issue_step(c, i + D, wave, lds_dummy, true); sc.issue_step(c, i + D, wave, lds_dummy, true);
      }
      if (i > 0 && !(a.dbg & 64)) flush(i - 1);
      f32x4_t acc[NTN];
      const bool top = i == 0, bot = i == nsteps - 1;
      if (a.dbg & 16) {
#pragma unroll
        for (int j = 0; j < NTN; ++j) acc[j] = addv[j];
      } else {
        sa.template compute<true, 1, NTN>(smem + OA, i, top, bot, nt0, acc, addv);
        sb.template compute<false, 1, NTN>(smem + OB, i, top, bot, nt0, acc, addv);
        sc.template compute<false, 1, NTN>(smem + OC, i, top, bot, nt0, acc, addv);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();
      roll_to_tile<NTN>(tile, acc, a.relu, oxb0, a.Wo, lane, pw, nt0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const int ncol = col + a.nstreams;
    primed = ncol < a.ncols;
    if (primed) {
      const int nb = ncol / a.nchunks, nc = ncol - nb * a.nchunks;
      sa.set_image(a.z[0], nb); sb.set_image(a.z[1], nb); sc.set_image(a.z[2], nb);
#pragma unroll
      for (int s = 0; s < D; ++s)
        if (s < nsteps) { sa.issue_step(nc, s, wave, lds_dummy, false); sb.issue_step(nc, s, wave, lds_dummy, false); sc.issue_step(nc, s, wave, lds_dummy, false); }
    }
    flush(nsteps - 1);
  }
}

// ---- the BACKWARD gather on the matrix cores (bf16, N % 64 == 0, factors 2 and 4): G_t = U^T S_t^T dy as ONE pass.
// For a low-resolution pixel q the nine maps are a small dense product that is the same for every channel:
//   G[q, t, n] = sum_p W[p, t] dy[p, n],   p = the (2F + 2)^2 output pixels whose tap positions interpolate from q,
//   W[p, (r, s)] = Uy[py + r - 1, qy] Ux[px + s - 1, qx]   (zero when the position lies outside the output),
// k = 16 * (row of the window) + (column of the window): three (factor 2) or five (factor 4) v_mfma_f32_16x16x32_bf16 steps per
// 16 channels, nine of the sixteen output columns used.  The two-pass kernels above move dy once and an intermediate of 3 / F
// of its size twice (0.93 ms for the neck's x4 level); here dy is staged once per 64 channels into LDS by DMA, the weights are
// products of two small tables, and -- as in the forward kernel (version 2) -- a block walks all channel chunks with the next
// window in flight and everything channel-independent in registers.  One block = four neighbouring pixels of one
// low-resolution row (one per wave).  Output layout = the two-pass kernels': [B, Hi, Wi, 9 N], tap block 8 - t.
//
// BN = true (round 5): the gradient that is gathered is the BatchNorm(+ReLU) backward of the incoming one, formed on the way
// into the LDS -- dy = ga rs (dz [bn(x) > 0] - mean(dz') - xhat mean(dz' xhat)) per element from dz and the convolution output
// x, with four per-channel constants (bn_bwd_coef_kernel).  The separate bn_bwd_dx pass (read dz, read x, write dy: 3 GB at
// the neck's x4 level) and this kernel's read of dy are replaced by one read of dz and x here.  The LDS-DMA cannot transform,
// so this variant stages through registers: a lane always fetches source chunk (lane & 7) -- its eight channels and their
// constants stay in registers for a whole channel step -- and the bank swizzle moves to the LDS side of the write.
struct GatherMArgs {
  const uint16_t* dy;
  uint16_t* g;
  int B, Ho, Wo, N, Hi, Wi;
  const uint16_t* x;      // BN: the convolution output the statistics were taken from, same layout as dy
  const float* coef;      // BN: [N / 8][4][8] = {a, thr, c2, c3} per channel (bn_bwd_coef_kernel)
};

// dy = a * (x a > thr ? dz : 0) - (c2 x + c3):  a = gamma rstd, thr = a mean - beta (-inf without ReLU),
// c2 = a rstd k2, c3 = a k1 - c2 mean with k1 = sum(dz') / P, k2 = sum(dz' xhat) / P   (bn_bwd_dx of norm.hip, refactored)
__global__ __launch_bounds__(256) void bn_bwd_coef_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                          int relu, const float* __restrict__ dgamma_sum,
                                                          const float* __restrict__ dbeta_sum, double inv_p, int N, float* __restrict__ coef) {
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch >= N) return;
  const double rs = 1.0 / sqrt((double)var[ch] + (double)eps), a = (double)gamma[ch] * rs;
  const double k1 = (double)dbeta_sum[ch] * inv_p, k2 = (double)dgamma_sum[ch] * inv_p;
  const double c2 = a * k2 * rs, c3 = a * k1 - c2 * (double)mean[ch];
  float* o = coef + (ch >> 3) * 32 + (ch & 7);
  o[0] = (float)a;
  o[8] = relu ? (float)(a * (double)mean[ch] - (double)beta[ch]) : -INFINITY;
  o[16] = (float)c2;
  o[24] = (float)c3;
}

template <int LF, bool BN = false>
__global__ __launch_bounds__(256, 2) void resize_conv3x3_bwd_gather_mfma_kernel(const GatherMArgs a) {
  constexpr int F = 1 << LF, WINB = 2 * F + 2, NKS = WINB / 2, QB = 4;
  constexpr int WCOLS = F * (QB - 1) + WINB;                         // window columns of the block
  constexpr int NROWS = WINB * WCOLS, NPIECES = (NROWS + 7) / 8, MAXP = (NPIECES + 3) / 4;
  constexpr int SLOT = NPIECES * 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* TYb = (float*)smem;                                         // [3][16]      weight of window row for filter row r
  float* TXb = TYb + 48;                                             // [QB][3][16]  the same for columns, per pixel of the block
  unsigned char* tile = smem + 1024;                                 // [QB * 9 rows][128 B]
  unsigned char* ring = tile + 5 * 1024;
  const unsigned lds_ring = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)ring);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = gridDim.x * gridDim.y;
  const int id = blockIdx.y * gridDim.x + blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = id & 7, idx = id >> 3;      // XCD-major: neighbouring rows share an L2
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int brow = lid / gridDim.x, cb = lid - brow * gridDim.x;
  const int b = brow / a.Hi, qy = brow - b * a.Hi;
  const int qxb0 = cb * QB;
  const int py0 = F * qy - (F / 2 + 1), pxb0 = F * qxb0 - (F / 2 + 1);
  const float ry = (float)a.Hi / (float)a.Ho, rx = (float)a.Wi / (float)a.Wo;
  if (tid < 48) {
    const int r = tid >> 4, wl = tid & 15;
    float w = 0.f;
    const int pos = py0 + wl + r - 1;
    if (wl < WINB && pos >= 0 && pos < a.Ho && py0 + wl >= 0 && py0 + wl < a.Ho) {
      int y0, y1; float ly;
      src_index(ry, pos, a.Hi, y0, y1, ly);
      w = (y0 == qy ? 1.f - ly : 0.f) + (y1 == qy ? ly : 0.f);
    }
    TYb[tid] = w;
  }
  if (tid >= 64 && tid < 64 + QB * 48) {
    const int i = tid - 64, j = i / 48, r3 = (i - j * 48) >> 4, wl = i & 15;
    const int qx = qxb0 + j, px = pxb0 + F * j + wl, pos = px + r3 - 1;
    float w = 0.f;
    if (wl < WINB && qx < a.Wi && pos >= 0 && pos < a.Wo && px >= 0 && px < a.Wo) {
      int x0, x1; float lx;
      src_index(rx, pos, a.Wi, x0, x1, lx);
      w = (x0 == qx ? 1.f - lx : 0.f) + (x1 == qx ? lx : 0.f);
    }
    TXb[i] = w;
  }
  __syncthreads();
  // ---- channel-independent state: weight fragments (B operand: column = tap), fragment addresses, DMA offsets
  const int L = lane & 15, g = lane >> 4;
  const int j = wave;                                                // this wave's pixel of the block
  bf16x8_t wf[NKS];
  int raddr[NKS][2];
  {
    const int r = L / 3, s3 = L - 3 * r;
    float txv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) txv[e] = L < 9 ? TXb[(j * 3 + s3) * 16 + 8 * (g & 1) + e] : 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int pyl = 2 * ks + (g >> 1);
      const float ty = L < 9 ? TYb[r * 16 + pyl] : 0.f;
      uint32_t pk[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) pk[e] = pack_bf16x2(ty * txv[2 * e], ty * txv[2 * e + 1]);
      wf[ks] = __builtin_bit_cast(bf16x8_t, make_uint4(pk[0], pk[1], pk[2], pk[3]));
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pxl = 8 * (g & 1) + 4 * h + (L >> 2);
        const bool ok = pyl < WINB && pxl < WINB;
        const int R = ok ? pyl * WCOLS + F * j + pxl : 0;            // padding slots: any staged row, their weight is zero
        raddr[ks][h] = R * 128 + (((((L & 3) >> 1) ^ tm_swz(R))) << 4) + (L & 1) * 8;
      }
    }
  }
  unsigned doff[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int R = (wave + 4 * i) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ tm_swz(R);
    const int wy = R / WCOLS, wc = R - wy * WCOLS;
    const int py = py0 + wy, px = pxb0 + wc;
    const bool ok = R < NROWS && py >= 0 && py < a.Ho && px >= 0 && px < a.Wo;
    doff[i] = ok ? (unsigned)(((py * a.Wo + px) * a.N) * 2 + chunk * 16) : kTmOob;
  }
  const srd_t srd = make_srd(a.dy + (int64_t)b * a.Ho * a.Wo * a.N, (unsigned)((int64_t)a.Ho * a.Wo * a.N * 2));
  auto issue = [&](int c) {
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      if (wave + 4 * i < NPIECES) dma16_buf(doff[i], srd, (unsigned)(c * 128), lds_ring + (c & 1) * SLOT + (wave + 4 * i) * 1024);
    }
  };
  // ---- BN: register staging.  Lane = (row lane >> 3 of the piece, source chunk lane & 7)
  typedef __attribute__((ext_vector_type(4))) unsigned gb_u4;
  constexpr int MAXPB = BN ? MAXP : 1;
  unsigned boff[MAXPB], bdst[MAXPB];
  gb_u4 rdz[MAXPB], rxv[MAXPB];
  float ca[8], cthr[8], cc2[8], cc3[8];
  __amdgpu_buffer_rsrc_t rs_dz, rs_x;
  if constexpr (BN) {
    const unsigned img_bytes = (unsigned)((int64_t)a.Ho * a.Wo * a.N * 2);
    rs_dz = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dy + (int64_t)b * a.Ho * a.Wo * a.N), 0, img_bytes, 0x00020000);
    rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (int64_t)b * a.Ho * a.Wo * a.N), 0, img_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int R = (wave + 4 * i) * 8 + (lane >> 3);
      const int wy = R / WCOLS, wc = R - wy * WCOLS;
      const int py = py0 + wy, px = pxb0 + wc;
      const bool ok = R < NROWS && py >= 0 && py < a.Ho && px >= 0 && px < a.Wo;
      boff[i] = ok ? (unsigned)(((py * a.Wo + px) * a.N) * 2 + (lane & 7) * 16) : kTmOob;
      bdst[i] = (unsigned)((wave + 4 * i) * 1024 + (lane >> 3) * 128 + (((lane & 7) ^ tm_swz(R)) << 4));
    }
  }
  auto bn_load = [&](int c) {
    if constexpr (BN) {
#pragma unroll
      for (int i = 0; i < MAXP; ++i) {
        if (wave + 4 * i < NPIECES) {
          rdz[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_dz, boff[i], c * 128, 0);
          rxv[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, boff[i], c * 128, 0);
        }
      }
      const float4* cp = (const float4*)(a.coef + ((int64_t)c * 8 + (lane & 7)) * 32);
      const float4 t0 = cp[0], t1 = cp[1], t2 = cp[2], t3 = cp[3], t4 = cp[4], t5 = cp[5], t6 = cp[6], t7 = cp[7];
      ca[0] = t0.x; ca[1] = t0.y; ca[2] = t0.z; ca[3] = t0.w; ca[4] = t1.x; ca[5] = t1.y; ca[6] = t1.z; ca[7] = t1.w;
      cthr[0] = t2.x; cthr[1] = t2.y; cthr[2] = t2.z; cthr[3] = t2.w; cthr[4] = t3.x; cthr[5] = t3.y; cthr[6] = t3.z; cthr[7] = t3.w;
      cc2[0] = t4.x; cc2[1] = t4.y; cc2[2] = t4.z; cc2[3] = t4.w; cc2[4] = t5.x; cc2[5] = t5.y; cc2[6] = t5.z; cc2[7] = t5.w;
      cc3[0] = t6.x; cc3[1] = t6.y; cc3[2] = t6.z; cc3[3] = t6.w; cc3[4] = t7.x; cc3[5] = t7.y; cc3[6] = t7.z; cc3[7] = t7.w;
    }
  };
  auto bn_store = [&](int c) {
    if constexpr (BN) {
#pragma unroll
      for (int i = 0; i < MAXP; ++i) {
        if (wave + 4 * i < NPIECES) {
          const unsigned dzw[4] = {rdz[i].x, rdz[i].y, rdz[i].z, rdz[i].w}, xw[4] = {rxv[i].x, rxv[i].y, rxv[i].z, rxv[i].w};
          uint32_t pk[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x0 = __uint_as_float(xw[e] << 16), x1 = __uint_as_float(xw[e] & 0xffff0000u);
            const float g0 = __uint_as_float(dzw[e] << 16), g1 = __uint_as_float(dzw[e] & 0xffff0000u);
            const float m0 = x0 * ca[2 * e] > cthr[2 * e] ? g0 : 0.f, m1 = x1 * ca[2 * e + 1] > cthr[2 * e + 1] ? g1 : 0.f;
            const float o0 = fmaf(ca[2 * e], m0, -fmaf(cc2[2 * e], x0, cc3[2 * e]));
            const float o1 = fmaf(ca[2 * e + 1], m1, -fmaf(cc2[2 * e + 1], x1, cc3[2 * e + 1]));
            pk[e] = boff[i] == kTmOob ? 0u : pack_bf16x2(o0, o1);      // out-of-image / padding rows stay zeros, as the DMA leaves them
          }
          *(uint4*)(ring + (c & 1) * SLOT + bdst[i]) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
      }
    }
  };
  const int nchunks = a.N >> 6;
  if constexpr (BN) { bn_load(0); bn_store(0); }
  else issue(0);
  for (int c = 0; c < nchunks; ++c) {
    if constexpr (!BN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                 // chunk c landed; nobody reads the other slot or the tile
    if (c + 1 < nchunks) {
      if constexpr (BN) bn_load(c + 1);
      else issue(c + 1);
    }
    const unsigned char* stage = ring + (c & 1) * SLOT;
    f32x4_t acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      acc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const tm_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tm_lds_s16x4_ptr)(stage + (raddr[ks][0] ^ (nt << 5))));
        const tm_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tm_lds_s16x4_ptr)(stage + (raddr[ks][1] ^ (nt << 5))));
        const tm_s16x8_t zv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, zv), wf[ks], acc[nt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // D[n][t]: lane = (tap L, channels 4 g ..): tile row = pixel * 9 + (8 - tap)
    if (L < 9) {
      const int row = j * 9 + (8 - L);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int ch = 16 * nt + 4 * g;
        *(uint2*)(tile + row * 128 + ((((ch >> 3) ^ (row & 7))) << 4) + (ch & 7) * 2) =
            make_uint2(pack_bf16x2(acc[nt][0], acc[nt][1]), pack_bf16x2(acc[nt][2], acc[nt][3]));
      }
    }
    __syncthreads();
    for (int i = tid; i < QB * 9 * 8; i += 256) {
      const int row = i >> 3, chunk = i & 7;
      const int jq = row / 9, tb = row - jq * 9, qx = qxb0 + jq;
      if (qx < a.Wi) {
        const uint4 v = *(const uint4*)(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
        *(uint4*)(a.g + ((((int64_t)b * a.Hi + qy) * a.Wi + qx) * 9 + tb) * a.N + c * 64 + chunk * 8) = v;
      }
    }
    if (BN && c + 1 < nchunks) bn_store(c + 1);       // into the slot last read in iteration c - 1 (two barriers ago)
  }
}

// nn.AdaptiveAvgPool2d: bin i covers [floor(i*In/S), ceil((i+1)*In/S))
__device__ __forceinline__ void pool_bin(int i, int in, int s, int& lo, int& hi) {
  lo = (i * in) / s;
  hi = ((i + 1) * in + s - 1) / s;
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const void* __restrict__ in, int B, int Hi, int Wi,
                                                          int C, int64_t isB, int64_t isH, int64_t isW,
                                                          void* out, int S) {
  const int cv = C / 4;
  const int64_t total = (int64_t)B * S * S * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cv) * 4;
    int64_t t = i / cv;
    const int ox = (int)(t % S); t /= S;
    const int oy = (int)(t % S);
    const int b = (int)(t / S);
    int y0, y1, x0, x1;
    pool_bin(oy, Hi, S, y0, y1);
    pool_bin(ox, Wi, S, x0, x1);
    float acc[4] = {0, 0, 0, 0};
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x) {
        float v[4];
        V4<TI>::ld(in, (int64_t)b * isB + y * isH + x * isW + c, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += v[j];
      }
    const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] *= inv;
    V4<TO>::st(out, (((int64_t)b * S + oy) * S + ox) * C + c, acc);
  }
}

// Large bins (pyramid pooling of an 18 x 18 map into 1 .. 3 bins per side: 36 .. 324 pixels each): the kernel above sums a bin
// with one thread per 4 channels -- 35 us whatever the batch.  Here 32 lanes cover 128 channels and the block's 8 lane groups
// split the bin's rows; LDS adds the partial sums in a fixed order.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void avgpool_fwd_split_kernel(const void* __restrict__ in, int Hi, int Wi, int C, int64_t isB,
                                                                int64_t isH, int64_t isW, void* out, int S) {
  __shared__ float red[8][32][4];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = (blockIdx.x * 32 + cl) * 4;
  int t = blockIdx.y;
  const int ox = t % S; t /= S;
  const int oy = t % S;
  const int b = t / S;
  int y0, y1, x0, x1;
  pool_bin(oy, Hi, S, y0, y1);
  pool_bin(ox, Wi, S, x0, x1);
  float acc[4] = {0, 0, 0, 0};
  if (c < C) {
    for (int y = y0 + sl; y < y1; y += 8)
      for (int x = x0; x < x1; ++x) {
        float v[4];
        V4<TI>::ld(in, (int64_t)b * isB + y * isH + x * isW + c, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += v[j];
      }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[sl][cl][j] = acc[j];
  __syncthreads();
  if (sl != 0 || c >= C) return;
#pragma unroll
  for (int k = 1; k < 8; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += red[k][cl][j];
  const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] *= inv;
  V4<TO>::st(out, (((int64_t)b * S + oy) * S + ox) * C + c, acc);
}

template <typename TO_, typename TI_>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const void* __restrict__ dout, int B, int S, int C,
                                                          void* din, int Hi, int Wi, int64_t isB,
                                                          int64_t isH, int64_t isW, int accumulate) {
  const int cv = C / 4;
  const int64_t total = (int64_t)B * Hi * Wi * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cv) * 4;
    int64_t t = i / cv;
    const int x = (int)(t % Wi); t /= Wi;
    const int y = (int)(t % Hi);
    const int b = (int)(t / Hi);
    float acc[4] = {0, 0, 0, 0};
    for (int oy = 0; oy < S; ++oy) {
      int y0, y1;
      pool_bin(oy, Hi, S, y0, y1);
      if (y < y0 || y >= y1) continue;
      for (int ox = 0; ox < S; ++ox) {
        int x0, x1;
        pool_bin(ox, Wi, S, x0, x1);
        if (x < x0 || x >= x1) continue;
        float g[4];
        V4<TO_>::ld(dout, (((int64_t)b * S + oy) * S + ox) * C + c, g);
        const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += g[j] * inv;
      }
    }
    const int64_t ioff = (int64_t)b * isB + (int64_t)y * isH + (int64_t)x * isW + c;
    if (accumulate) {
      float o[4];
      V4<TI_>::ld(din, ioff, o);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += o[j];
    }
    V4<TI_>::st(din, ioff, acc);
  }
}

inline unsigned grid_for(int64_t total) {
  int64_t g = (total + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

#define DISPATCH2(KERN, DA, DB, ...)                                                                 \
  do {                                                                                               \
    if ((DA) == GDL_BF16 && (DB) == GDL_BF16) hipLaunchKernelGGL((KERN<uint16_t, uint16_t>), __VA_ARGS__); \
    else if ((DA) == GDL_BF16) hipLaunchKernelGGL((KERN<uint16_t, float>), __VA_ARGS__);             \
    else if ((DB) == GDL_BF16) hipLaunchKernelGGL((KERN<float, uint16_t>), __VA_ARGS__);             \
    else hipLaunchKernelGGL((KERN<float, float>), __VA_ARGS__);                                      \
  } while (0)

int g_flat_resample = 0;   // A/B hook: 1 = the flat-index kernels

inline bool vec8_ok(const void* a, const void* b, int C, int64_t s0, int64_t s1, int64_t s2, int64_t s3, int64_t s4,
                    int64_t s5) {
  return C % 8 == 0 && ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0) && s0 % 8 == 0 && s1 % 8 == 0 &&
         s2 % 8 == 0 && s3 % 8 == 0 && s4 % 8 == 0 && s5 % 8 == 0;
}

int g_gather_mfma = 1;     // A/B hook (gdl_debug_set_gather_mfma): 0 = the two-pass / single-pass VALU gathers of the backward
int g_tapsum_mfma = 1;     // A/B hook (gdl_debug_set_tapsum_mfma): 0 = the pixel-by-pixel kernel; 1 = MFMA, version chosen by shape;
                           // 5 = version 2 (4 x 16 pixel blocks, all channels, pipelined); 2 / 4 = version 1 (32- / 16-column blocks x 64 channels)
int g_tapsum_vec = 0;      // A/B hook (gdl_debug_set_tapsum_vec): 4 = 8-byte bf16 vectors per thread instead of 16-byte ones

}  // namespace

extern "C" void gdl_debug_set_tapsum_vec(int vec) { g_tapsum_vec = vec; }
extern "C" void gdl_debug_set_tapsum_mfma(int mode) { g_tapsum_mfma = mode; }
extern "C" void gdl_debug_set_gather_mfma(int on) { g_gather_mfma = on; }

// matrix-core form of the backward gather; returns false when the shape does not qualify (the caller runs the VALU kernels)
static bool gather_mfma_ok(int dtype, int B, int Ho, int Wo, int N, int Hi, int Wi) {
  if (!g_gather_mfma || dtype != GDL_BF16 || N % 64 != 0 || Hi <= 0 || Wi <= 0) return false;
  const int f = Ho / Hi;
  if (!((f == 2 || f == 4) && Hi * f == Ho && Wi * f == Wo)) return false;
  return (int64_t)B * Hi <= 65535 && (int64_t)Ho * Wo * N * 2 < 0x7ffffff0ll;
}
extern "C" int gdl_resize_conv3x3_bwd_gather_one_pass(int dtype, int B, int Ho, int Wo, int N, int Hi, int Wi) {
  return gather_mfma_ok(dtype, B, Ho, Wo, N, Hi, Wi) ? 1 : 0;
}

static bool gather_mfma_launch(const void* dy, int dtype, int B, int Ho, int Wo, int N, void* g, int Hi, int Wi, hipStream_t s,
                               const void* bn_x = nullptr, const float* bn_coef = nullptr) {
  if (!gather_mfma_ok(dtype, B, Ho, Wo, N, Hi, Wi) || (uintptr_t)dy % 16 || (uintptr_t)g % 16 || (uintptr_t)bn_x % 16) return false;
  const int f = Ho / Hi;
  GatherMArgs a = {(const uint16_t*)dy, (uint16_t*)g, B, Ho, Wo, N, Hi, Wi, (const uint16_t*)bn_x, bn_coef};
  const dim3 grid((unsigned)((Wi + 3) / 4), (unsigned)(B * Hi));
  if (f == 2) {
    constexpr int pieces = (6 * (2 * 3 + 6) + 7) / 8;
    const size_t lds = 6 * 1024 + 2 * pieces * 1024;
    if (bn_x) {
      GDL_SET_MAX_LDS_ONCE((resize_conv3x3_bwd_gather_mfma_kernel<1, true>), 160 * 1024);
      hipLaunchKernelGGL((resize_conv3x3_bwd_gather_mfma_kernel<1, true>), grid, dim3(256), lds, s, a);
    } else {
      GDL_SET_MAX_LDS_ONCE(resize_conv3x3_bwd_gather_mfma_kernel<1>, 160 * 1024);
      hipLaunchKernelGGL(resize_conv3x3_bwd_gather_mfma_kernel<1>, grid, dim3(256), lds, s, a);
    }
  } else {
    constexpr int pieces = (10 * (4 * 3 + 10) + 7) / 8;
    const size_t lds = 6 * 1024 + 2 * pieces * 1024;
    if (bn_x) {
      GDL_SET_MAX_LDS_ONCE((resize_conv3x3_bwd_gather_mfma_kernel<2, true>), 160 * 1024);
      hipLaunchKernelGGL((resize_conv3x3_bwd_gather_mfma_kernel<2, true>), grid, dim3(256), lds, s, a);
    } else {
      GDL_SET_MAX_LDS_ONCE(resize_conv3x3_bwd_gather_mfma_kernel<2>, 160 * 1024);
      hipLaunchKernelGGL(resize_conv3x3_bwd_gather_mfma_kernel<2>, grid, dim3(256), lds, s, a);
    }
  }
  return true;
}

// The gather of gdl_resize_conv3x3_bwd_gather applied to the BatchNorm(+ReLU) backward of dz, without that gradient ever being
// written: see resize_conv3x3_bwd_gather_mfma_kernel<LF, true>.  Only where gdl_resize_conv3x3_bwd_gather_one_pass() says 1.
extern "C" int gdl_resize_conv3x3_bwd_gather_bn(const void* dz, const void* x, int dtype, int B, int Ho, int Wo, int N, void* g, int Hi,
                                                int Wi, const float* mean, const float* var, const float* gamma, const float* beta,
                                                float eps, int relu, const float* dgamma_sum, const float* dbeta_sum, int64_t P_total,
                                                float* coef_ws, gdl_stream_t stream) {
  GDL_CHECK_ARG(dz && x && g && mean && var && gamma && beta && dgamma_sum && dbeta_sum && coef_ws,
                "gdl_resize_conv3x3_bwd_gather_bn: null pointer");
  GDL_CHECK_ARG(B > 0 && Ho > 0 && Wo > 0 && Hi > 0 && Wi > 0 && N > 0 && P_total > 0, "gdl_resize_conv3x3_bwd_gather_bn: bad dims");
  GDL_CHECK_ARG(gather_mfma_ok(dtype, B, Ho, Wo, N, Hi, Wi) && (uintptr_t)dz % 16 == 0 && (uintptr_t)x % 16 == 0 &&
                    (uintptr_t)g % 16 == 0 && (uintptr_t)coef_ws % 16 == 0,
                "gdl_resize_conv3x3_bwd_gather_bn: needs the one-pass matrix-core form (bf16, N %% 64 == 0, factor 2 or 4, "
                "gdl_resize_conv3x3_bwd_gather_one_pass() == 1) and 16-byte aligned pointers");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, mean, var, gamma, beta, eps, relu, dgamma_sum,
                     dbeta_sum, 1.0 / (double)P_total, N, coef_ws);
  const bool ok = gather_mfma_launch(dz, dtype, B, Ho, Wo, N, g, Hi, Wi, s, x, coef_ws);
  GDL_CHECK_ARG(ok, "gdl_resize_conv3x3_bwd_gather_bn: launch refused");
  GDL_CHECK_LAUNCH("gdl_resize_conv3x3_bwd_gather_bn");
  return GDL_OK;
}

// shared launcher; stats != nullptr: per-block partial sums of the outputs ([rows][2][N] f32, rows = gdl_resize_conv3x3_fwd_sum_bn_rows)
static int g_tapsum_persist = 1;
extern "C" void gdl_debug_set_tapsum_persist(int on) { g_tapsum_persist = on; }  // A/B hook: 0 = one workgroup per patch (round 3)
static int tapsum_num_cus() {
  static int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    return cus;
  }();
  return n;
}
static int g_tapsum_roll = 1;
constexpr int kRollDepth4 = 3;    // factor 4: steps the row fetches of version 3 run ahead (ring of 3 + depth slots of 7 KiB, three workgroups per CU)
extern "C" void gdl_debug_set_tapsum_roll(int on) { g_tapsum_roll = on; }  // A/B hook: 0 = versions 1 / 2 where version 3 (rolling row window) applies
// stat_rows (with stats): the number of partial rows the launch wrote (<= gdl_resize_conv3x3_fwd_sum_bn_rows)
static int tapsum_launch(const void* const* zs, const int* hs, const int* ws, int nsrc, int dtype, int B, int N, void* out, int Ho, int Wo,
                         const float* addvec, int relu, float* stats, hipStream_t st, int64_t* stat_rows = nullptr) {
  GDL_CHECK_ARG(zs && hs && ws && out && nsrc >= 1 && nsrc <= 3, "gdl_resize_conv3x3_fwd_sum: 1..3 sources");
  GDL_CHECK_ARG(dtype == GDL_F32 || dtype == GDL_BF16, "gdl_resize_conv3x3_fwd_sum: bad dtype");
  GDL_CHECK_ARG(B > 0 && Ho > 0 && Wo > 0 && N > 0, "gdl_resize_conv3x3_fwd_sum: bad sizes");
  TapSrc s;
  int run = 2;
  for (int k = 0; k < 3; ++k) {
    s.p[k] = nullptr; s.H[k] = 1; s.W[k] = 1; s.F[k] = 0;
    if (k >= nsrc) continue;
    GDL_CHECK_ARG(zs[k] && hs[k] > 0 && ws[k] > 0 && (uintptr_t)zs[k] % 16 == 0, "gdl_resize_conv3x3_fwd_sum: bad source %d", k);
    const int f = Ho / hs[k];
    GDL_CHECK_ARG((f == 2 || f == 4 || f == 8) && hs[k] * f == Ho && ws[k] * f == Wo,
                  "gdl_resize_conv3x3_fwd_sum: source %d: the resize factor must be 2, 4 or 8 in both directions", k);
    s.p[k] = zs[k]; s.H[k] = hs[k]; s.W[k] = ws[k]; s.F[k] = f;
    run = f > run ? f : run;
  }
  GDL_CHECK_ARG(Wo % run == 0, "gdl_resize_conv3x3_fwd_sum: Wo must be a multiple of the largest factor");
  bool img_ok = (uintptr_t)out % 16 == 0;
  for (int k = 0; k < nsrc; ++k) img_ok = img_ok && (int64_t)s.H[k] * s.W[k] * 9 * N * 2 < 0x7ffffff0ll;   // 32-bit buffer offsets per image
  const bool mfma_ok = dtype == GDL_BF16 && N % 64 == 0 && img_ok && (int64_t)B * ((Ho + 3) / 4) <= 65535 && (uintptr_t)addvec % 16 == 0;
  GDL_CHECK_ARG(!stats || mfma_ok, "gdl_resize_conv3x3_fwd_sum_bn: needs bf16, N %% 64 == 0, 16-byte aligned output and per-channel "
                "addend (bias), sources below 2 GiB per image, B * ceil(Ho / 4) <= 65535");
  if (mfma_ok && (g_tapsum_mfma || stats)) {
    TapMArgs m;
    int lfmin = 3;
    for (int k = 0; k < 3; ++k) {
      m.z[k] = (const uint16_t*)s.p[k]; m.H[k] = s.H[k]; m.W[k] = s.W[k];
      m.LF[k] = s.F[k] == 8 ? 3 : (s.F[k] == 4 ? 2 : 1);
      if (k < nsrc && m.LF[k] < lfmin) lfmin = m.LF[k];
    }
    m.nsrc = nsrc; m.N = N; m.Ho = Ho; m.Wo = Wo; m.B = B; m.out = (uint16_t*)out; m.addvec = addvec; m.relu = relu;
    m.nslots = 0; m.slot_off[0] = m.slot_off[1] = m.slot_off[2] = 0;
    if (stat_rows) *stat_rows = (int64_t)B * ((Ho + 3) / 4) * ((Wo + 15) / 16);
    // version 3 for the three upsampled levels of fpn_bottleneck: factors 2, 4, 8 in that order, whole cells of the coarsest one
    if (g_tapsum_roll && !stats && nsrc == 3 && m.LF[0] == 1 && m.LF[1] == 2 && m.LF[2] == 3 && Ho % 8 == 0 && g_tapsum_mfma == 1) {
      constexpr int D3 = 2;
      TapR3Args r;
      for (int k = 0; k < 3; ++k) { r.z[k] = m.z[k]; r.H[k] = m.H[k]; r.W[k] = m.W[k]; }
      r.out = m.out; r.addvec = addvec; r.N = N; r.Ho = Ho; r.Wo = Wo; r.B = B; r.relu = relu;
      r.nstrips = (Wo + 15) / 16; r.nchunks = N / 64; r.ncols = B * r.nchunks;
      int streams = tapsum_num_cus() / r.nstrips;            // one workgroup per CU
      if (streams > r.ncols) streams = r.ncols;
      if (streams < 1) streams = 1;
      r.nstreams = streams; r.nblocks = streams * r.nstrips;
      r.dbg = g_tapsum_roll >= 16 ? g_tapsum_roll : 0;
      const size_t lds = 4 * 16 * 128 + RollSrc<1, D3, 8>::BYTES + RollSrc<2, D3, 8>::BYTES + RollSrc<3, D3, 8>::BYTES + 1024;
      static_assert(4 * 16 * 128 + RollSrc<1, D3, 8>::BYTES + RollSrc<2, D3, 8>::BYTES + RollSrc<3, D3, 8>::BYTES + 1024 <= 160 * 1024, "LDS budget");
      GDL_SET_MAX_LDS_ONCE((resize_conv3x3_fwd_sum_roll3_kernel<D3>), 160 * 1024);
      hipLaunchKernelGGL((resize_conv3x3_fwd_sum_roll3_kernel<D3>), dim3((unsigned)r.nblocks), dim3(512), lds, st, r);
      GDL_CHECK_LAUNCH("gdl_resize_conv3x3_fwd_sum");
      return GDL_OK;
    }
    // version 3 (rolling row window): one source of factor 2 or 4 whose output rows come in whole patches
    if (g_tapsum_roll && nsrc == 1 && (m.LF[0] == 1 || m.LF[0] == 2) && Ho % 4 == 0 && Ho >= 8 && (g_tapsum_mfma == 1 || stats)) {
      TapRArgs r;
      r.z = m.z[0]; r.out = m.out; r.addvec = addvec; r.stats = stats;
      r.Hi = m.H[0]; r.Wi = m.W[0]; r.N = N; r.Ho = Ho; r.Wo = Wo; r.B = B; r.relu = relu;
      r.nstrips = (Wo + 15) / 16; r.nchunks = N / 64; r.ncols = B * r.nchunks;
      const int occ = m.LF[0] == 2 ? 3 : 2, slots = tapsum_num_cus() / 8 * occ;         // resident workgroups per XCD (launch bounds)
      int G = slots / r.nstrips;
      if (G > (r.ncols + 7) / 8) G = (r.ncols + 7) / 8;
      if (G < 1) G = 1;
      r.G = G;
      const dim3 grid((unsigned)(8 * G * r.nstrips));
      const size_t lds = 4 * 16 * 128 + (size_t)(m.LF[0] == 2 ? (3 + (g_tapsum_roll == 2 ? 1 : kRollDepth4)) * 7 : 6 * 12) * 1024;
#define TAPR(LF_, ST_, D_) do { GDL_SET_MAX_LDS_ONCE((resize_conv3x3_fwd_sum_roll_kernel<LF_, ST_, D_>), 160 * 1024);                    \
    hipLaunchKernelGGL((resize_conv3x3_fwd_sum_roll_kernel<LF_, ST_, D_>), grid, dim3(256), lds, st, r); } while (0)
      if (m.LF[0] == 2 && g_tapsum_roll == 2) { if (stats) TAPR(2, true, 1); else TAPR(2, false, 1); }      // (A/B: one step ahead)
      else if (m.LF[0] == 2) { if (stats) TAPR(2, true, kRollDepth4); else TAPR(2, false, kRollDepth4); }
      else { if (stats) TAPR(1, true, 1); else TAPR(1, false, 1); }
#undef TAPR
      if (stat_rows) *stat_rows = (int64_t)B * r.nstrips;
      GDL_CHECK_LAUNCH("gdl_resize_conv3x3_fwd_sum");
      return GDL_OK;
    }
    // the staging loop writes whole 1 KiB pieces (8 rows): round a window up to that
    auto window_bytes_lf = [&](int cols, int lf) { const int win = lf == 1 ? 4 : 3; return (win * (((cols - 4) >> lf) + win) * 9 * 128 + 1023) / 1024 * 1024; };
    // version 2 (all channels of a 4 x 16 pixel block per workgroup, windows double-buffered) where three blocks fit a CU: ONE
    // source of factor 4 or 8 (measured on the neck's x4 level: 442 us against 768; with the 46 KiB windows of a factor-2
    // source, or three sources, one block per CU remains and version 1 is the faster one: 284 vs 315 us, 693 vs 1148 us);
    // always for the statistics variant
    const bool v2_pays = nsrc == 1 && lfmin >= 2;
    if ((g_tapsum_mfma == 1 && v2_pays) || g_tapsum_mfma == 5 || stats) {
      const int pxb = 4, cols = 4 * pxb;   // (4 x 32 blocks: the cached fragments of two patches per wave do not fit the register budget)
      int ring = 0;
      m.nslots = nsrc == 1 ? 2 : nsrc;
      for (int j = 0; j < m.nslots; ++j) { m.slot_off[j] = ring; ring += window_bytes_lf(cols, m.LF[nsrc == 1 ? 0 : j]); }
      const size_t lds = (size_t)nsrc * (256 + cols * 64) + 4 * cols * 128 + ring;
      GDL_CHECK_ARG(lds <= 160 * 1024 || !stats, "gdl_resize_conv3x3_fwd_sum_bn: LDS budget exceeded");
      m.gx = (Wo + cols - 1) / cols;
      m.nblk = m.gx * B * ((Ho + 3) / 4);
      const int resident = tapsum_num_cus() * (nsrc == 1 ? 3 : 1);       // workgroups a launch keeps resident (launch bounds)
      const dim3 grid((unsigned)(g_tapsum_persist && m.nblk > resident ? resident : m.nblk));
      if (lds > 160 * 1024) {
        // (does not happen for 1..3 sources of factors 2 / 4 / 8; version 1 below would take over)
      } else
#define TAPM2(PXB_, NSRC_, LF_) do { GDL_SET_MAX_LDS_ONCE((resize_conv3x3_fwd_sum_mfma2_kernel<PXB_, NSRC_, LF_>), 160 * 1024);                 \
    hipLaunchKernelGGL((resize_conv3x3_fwd_sum_mfma2_kernel<PXB_, NSRC_, LF_>), grid, dim3(256), lds, st, m, stats); } while (0)
      if (nsrc == 1 && m.LF[0] == 2) TAPM2(4, 1, 2); else if (nsrc == 1 && m.LF[0] == 3) TAPM2(4, 1, 3); else if (nsrc == 1) TAPM2(4, 1, 1);
      else if (nsrc == 2) TAPM2(4, 2, 0); else TAPM2(4, 3, 0);
#undef TAPM2
      if (lds <= 160 * 1024) {
        GDL_CHECK_LAUNCH("gdl_resize_conv3x3_fwd_sum");
        return GDL_OK;
      }
    }
    // version 1: one block = 4 rows x 16 or 32 columns x 64 channels; 32 columns when every window stays under 48 KiB
    auto window_bytes = [&](int cols) { return window_bytes_lf(cols, lfmin); };
    const int pxb = (window_bytes(32) <= 48 * 1024 && g_tapsum_mfma != 4) ? 8 : 4;
    const int cols = 4 * pxb;
    const int tile_bytes = 4 * cols * 128;
    const size_t lds = 256 + cols * 64 + (size_t)(window_bytes(cols) > tile_bytes ? window_bytes(cols) : tile_bytes);
    const dim3 grid((unsigned)(((Wo + cols - 1) / cols) * (N / 64)), (unsigned)(B * ((Ho + 3) / 4)));
    if (pxb == 8) {
      GDL_SET_MAX_LDS_ONCE(resize_conv3x3_fwd_sum_mfma_kernel<8>, 160 * 1024);
      hipLaunchKernelGGL(resize_conv3x3_fwd_sum_mfma_kernel<8>, grid, dim3(256), lds, st, m);
    } else {
      GDL_SET_MAX_LDS_ONCE(resize_conv3x3_fwd_sum_mfma_kernel<4>, 160 * 1024);
      hipLaunchKernelGGL(resize_conv3x3_fwd_sum_mfma_kernel<4>, grid, dim3(256), lds, st, m);
    }
    GDL_CHECK_LAUNCH("gdl_resize_conv3x3_fwd_sum");
    return GDL_OK;
  }
  int vec = dtype == GDL_BF16 ? 8 : 4;
  // 16-byte vectors with 8-pixel runs need more than 256 registers: 8-byte vectors there (and wherever N % 8 != 0)
  if (dtype == GDL_BF16 && (g_tapsum_vec == 4 || N % 8 != 0 || (run == 8 && g_tapsum_vec != 8))) vec = 4;
  GDL_CHECK_ARG(N % vec == 0 && (uintptr_t)out % 16 == 0, "gdl_resize_conv3x3_fwd_sum: N must be a multiple of 4, out 16-byte aligned");
  GDL_CHECK_ARG((int64_t)B * Ho <= 65535, "gdl_resize_conv3x3_fwd_sum: B * Ho must fit one grid dimension");
  const dim3 grid((unsigned)(((Wo / run) * (N / vec) + 255) / 256), (unsigned)(B * Ho));
#define TAPSUM(V, VEC, RUN) hipLaunchKernelGGL((resize_conv3x3_fwd_sum_kernel<V, VEC, RUN>), grid, dim3(256), 0, st, s, nsrc, N, out, \
                                               Ho, Wo, addvec, relu)
#define TAPSUM_RUN(V, VEC) do { if (run == 8) TAPSUM(V, VEC, 8); else if (run == 4) TAPSUM(V, VEC, 4); else TAPSUM(V, VEC, 2); } while (0)
  if (dtype == GDL_F32) TAPSUM_RUN(V4<float>, 4);
  else if (vec == 8) TAPSUM_RUN(V8, 8);
  else TAPSUM_RUN(V4<uint16_t>, 4);
#undef TAPSUM_RUN
#undef TAPSUM
  GDL_CHECK_LAUNCH("gdl_resize_conv3x3_fwd_sum");
  return GDL_OK;
}

// ---- the same sum for ANY resize ratio (round 5): out[b, oy, ox, n] (+)= sum_(r,s) bilinear(z_(r,s))[oy + r - 1, ox + s - 1].
// The kernels above are built around an integer factor (a cell of F output columns shares its source columns); DOFA-large's
// top FPN level is int(73 * 0.5) = 36 pixels wide against 292 (models/utils.py:106-110: x 8.11), which sent the whole
// `fpn_bottleneck` to the unfused path (concat buffer + a K = 9 * 4096 convolution at full resolution).  This is the plain
// gather form -- per output vector 9 taps x 4 corners = 36 loads, f32 accumulation in a fixed order -- for the ONE level of a
// pyramid that needs it: the source is small (it is the level being upsampled ~8 x) and stays in L2.
template <typename V, int VEC>
__global__ __launch_bounds__(256) void resize_conv3x3_fwd_sum_any_kernel(const void* __restrict__ z, int Hi, int Wi, int N, void* out,
                                                                         int Ho, int Wo, int accumulate,
                                                                         const float* __restrict__ addvec, int relu) {
  const int cv = N / VEC;
  const int j0 = blockIdx.x * 256 + threadIdx.x;
  if (j0 >= Wo * cv) return;
  const int b = blockIdx.y / Ho, oy = blockIdx.y - b * Ho;
  const int ox = j0 / cv, c = (j0 - ox * cv) * VEC;
  const float ry = (float)Hi / (float)Ho, rx = (float)Wi / (float)Wo;
  const int64_t pix = 9ll * N;
  const int64_t zb = (int64_t)b * Hi * Wi * pix + c;
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  int xs0[3], xs1[3]; float lxs[3]; bool xin[3];
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) {
    const int px = ox + s3 - 1;
    xin[s3] = px >= 0 && px < Wo;
    xs0[s3] = xs1[s3] = 0; lxs[s3] = 0.f;
    if (xin[s3]) src_index(rx, px, Wi, xs0[s3], xs1[s3], lxs[s3]);
  }
#pragma unroll 1
  for (int r = 0; r < 3; ++r) {
    const int py = oy + r - 1;
    if (py < 0 || py >= Ho) continue;              // the convolution's zero padding (uniform over the block)
    int y0, y1; float ly;
    src_index(ry, py, Hi, y0, y1, ly);
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
      if (!xin[s3]) continue;
      const int64_t t = (int64_t)(3 * r + s3) * N;
      float v00[VEC], v01[VEC], v10[VEC], v11[VEC];
      V::ld(z, zb + ((int64_t)y0 * Wi + xs0[s3]) * pix + t, v00);
      V::ld(z, zb + ((int64_t)y0 * Wi + xs1[s3]) * pix + t, v01);
      V::ld(z, zb + ((int64_t)y1 * Wi + xs0[s3]) * pix + t, v10);
      V::ld(z, zb + ((int64_t)y1 * Wi + xs1[s3]) * pix + t, v11);
      const float lx = lxs[s3];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float top = v00[e] + lx * (v01[e] - v00[e]), bot = v10[e] + lx * (v11[e] - v10[e]);
        acc[e] += top + ly * (bot - top);
      }
    }
  }
  const int64_t o = (((int64_t)b * Ho + oy) * Wo + ox) * N + c;
  if (accumulate) {
    float prev[VEC];
    V::ld(out, o, prev);
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] += prev[e];
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    if (addvec) acc[e] += addvec[c + e];
    if (relu) acc[e] = fmaxf(acc[e], 0.f);
  }
  V::st(out, o, acc);
}

extern "C" int gdl_resize_conv3x3_fwd_sum_any(const void* z, int hs, int ws, int dtype, int B, int N, void* out, int Ho, int Wo,
                                              int accumulate, const float* addvec, int relu, gdl_stream_t stream) {
  GDL_CHECK_ARG(z && out && hs > 0 && ws > 0 && B > 0 && N > 0 && Ho > 0 && Wo > 0, "gdl_resize_conv3x3_fwd_sum_any: bad args");
  GDL_CHECK_ARG(dtype == GDL_F32 || dtype == GDL_BF16, "gdl_resize_conv3x3_fwd_sum_any: bad dtype");
  const int vec = dtype == GDL_BF16 ? 8 : 4;
  GDL_CHECK_ARG(N % vec == 0 && (uintptr_t)z % 16 == 0 && (uintptr_t)out % 16 == 0,
                "gdl_resize_conv3x3_fwd_sum_any: N must be a multiple of the 16-byte vector, pointers 16-byte aligned");
  GDL_CHECK_ARG((int64_t)B * Ho <= 65535, "gdl_resize_conv3x3_fwd_sum_any: B * Ho must fit one grid dimension");
  const dim3 grid((unsigned)(((int64_t)Wo * (N / vec) + 255) / 256), (unsigned)(B * Ho));
  if (dtype == GDL_BF16)
    hipLaunchKernelGGL((resize_conv3x3_fwd_sum_any_kernel<V8, 8>), grid, dim3(256), 0, (hipStream_t)stream, z, hs, ws, N, out, Ho, Wo,
                       accumulate, addvec, relu);
  else
    hipLaunchKernelGGL((resize_conv3x3_fwd_sum_any_kernel<V4<float>, 4>), grid, dim3(256), 0, (hipStream_t)stream, z, hs, ws, N, out, Ho,
                       Wo, accumulate, addvec, relu);
  GDL_CHECK_LAUNCH("gdl_resize_conv3x3_fwd_sum_any");
  return GDL_OK;
}

extern "C" int gdl_resize_conv3x3_fwd_sum(const void* const* zs, const int* hs, const int* ws, int nsrc, int dtype, int B, int N,
                                          void* out, int Ho, int Wo, const float* addvec, int relu, gdl_stream_t stream) {
  return tapsum_launch(zs, hs, ws, nsrc, dtype, B, N, out, Ho, Wo, addvec, relu, nullptr, (hipStream_t)stream);
}

extern "C" int64_t gdl_resize_conv3x3_fwd_sum_bn_rows(int B, int Ho, int Wo) {
  if (B <= 0 || Ho <= 0 || Wo <= 0) return 0;
  return (int64_t)B * ((Ho + 3) / 4) * ((Wo + 15) / 16);
}

extern "C" int gdl_resize_conv3x3_fwd_sum_bn(const void* const* zs, const int* hs, const int* ws, int nsrc, int dtype, int B, int N,
                                             void* out, int Ho, int Wo, const float* addvec, float* workspace, int64_t ws_bytes,
                                             float* mean, float* var, float* running_mean, float* running_var, float momentum,
                                             gdl_stream_t stream) {
  GDL_CHECK_ARG(workspace && mean && var, "gdl_resize_conv3x3_fwd_sum_bn: null pointer");
  const int64_t rows = gdl_resize_conv3x3_fwd_sum_bn_rows(B, Ho, Wo);
  GDL_CHECK_ARG(ws_bytes >= rows * 2 * N * (int64_t)sizeof(float), "gdl_resize_conv3x3_fwd_sum_bn: workspace too small");
  int64_t written = rows;
  const int st = tapsum_launch(zs, hs, ws, nsrc, dtype, B, N, out, Ho, Wo, addvec, 0, workspace, (hipStream_t)stream, &written);
  if (st != GDL_OK) return st;
  return gdl_bn_stats_finalize(workspace, (int)written, N, (int64_t)B * Ho * Wo, mean, var, running_mean, running_var, momentum, stream);
}

extern "C" void gdl_debug_set_flat_resample(int on) { g_flat_resample = on; }

static int bilinear_fwd_impl(const void* in, int in_dtype, int B, int Hi, int Wi, int C, int64_t isB,
                             int64_t isH, int64_t isW, void* out, int out_dtype, int Ho, int Wo,
                             int64_t osB, int64_t osH, int64_t osW, int accumulate, const void* base,
                             gdl_stream_t stream) {
  GDL_CHECK_ARG(in && out, "gdl_bilinear_fwd: null pointer");
  GDL_CHECK_ARG(C % 4 == 0 && isB % 4 == 0 && isH % 4 == 0 && isW % 4 == 0 && osB % 4 == 0 && osH % 4 == 0 &&
                    osW % 4 == 0, "gdl_bilinear_fwd: C and strides must be multiples of 4");
  if (in_dtype == GDL_BF16 && out_dtype == GDL_BF16 && vec8_ok(in, out, C, isB, isH, isW, osB, osH, osW)) {
    const int64_t total8 = (int64_t)B * Ho * Wo * (C / 8);
    const int fac = (Hi > 1 && Wi > 1 && Ho % Hi == 0 && Wo % Wi == 0 && Ho / Hi == Wo / Wi) ? Ho / Hi : 0;
    const bool gap = (fac == 2 || fac == 4) && (int64_t)B * (Hi + 1) <= 65535 && !g_flat_resample;
    if (base && !(gap && (uintptr_t)base % 16 == 0)) return GDL_ERR_UNSUPPORTED;      // (the caller copies + accumulates)
    if (gap) {
      const dim3 grid((unsigned)(((Wi + 1) * (C / 8) + 255) / 256), (unsigned)(B * (Hi + 1)));
      const void* acc_src = base ? base : (accumulate ? out : nullptr);
      if (fac == 2) hipLaunchKernelGGL(bilinear_fwd8_gap_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, in, Hi, Wi, C, isB, isH, isW, out, osB, osH, osW, acc_src);
      else hipLaunchKernelGGL(bilinear_fwd8_gap_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, in, Hi, Wi, C, isB, isH, isW, out, osB, osH, osW, acc_src);
    } else if ((int64_t)B * Ho <= 65535 && g_flat_resample != 1)
      hipLaunchKernelGGL(bilinear_fwd8_rows_kernel, dim3((unsigned)((Wo * (C / 8) + 255) / 256), (unsigned)(B * Ho)), dim3(256), 0,
                         (hipStream_t)stream, in, Hi, Wi, C, isB, isH, isW, out, Ho, Wo, osB, osH, osW, accumulate);
    else
    hipLaunchKernelGGL(bilinear_fwd8_kernel, dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream, in, B, Hi, Wi, C,
                       isB, isH, isW, out, Ho, Wo, osB, osH, osW, accumulate);
    GDL_CHECK_LAUNCH("gdl_bilinear_fwd");
    return GDL_OK;
  }
  if (base) return GDL_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)B * Ho * Wo * (C / 4);
  DISPATCH2(bilinear_fwd_kernel, in_dtype, out_dtype, dim3(grid_for(total)), dim3(256), 0,
            (hipStream_t)stream, in, B, Hi, Wi, C, isB, isH, isW, out, Ho, Wo, osB, osH, osW, accumulate);
  GDL_CHECK_LAUNCH("gdl_bilinear_fwd");
  return GDL_OK;
}

extern "C" int gdl_bilinear_fwd(const void* in, int in_dtype, int B, int Hi, int Wi, int C, int64_t isB,
                                int64_t isH, int64_t isW, void* out, int out_dtype, int Ho, int Wo,
                                int64_t osB, int64_t osH, int64_t osW, int accumulate,
                                gdl_stream_t stream) {
  return bilinear_fwd_impl(in, in_dtype, B, Hi, Wi, C, isB, isH, isW, out, out_dtype, Ho, Wo, osB, osH, osW, accumulate, nullptr, stream);
}

extern "C" int gdl_bilinear_fwd_add(const void* in, int in_dtype, int B, int Hi, int Wi, int C, int64_t isB,
                                    int64_t isH, int64_t isW, const void* base, void* out, int out_dtype, int Ho, int Wo,
                                    int64_t osB, int64_t osH, int64_t osW, gdl_stream_t stream) {
  GDL_CHECK_ARG(in && base && out, "gdl_bilinear_fwd_add: null pointer");
  // one pass where the gap kernel applies (bf16, 16-byte vectors, factor 2 / 4); otherwise the lateral is copied and the resized
  // map accumulated into it -- same values either way (out = base, then += the interpolation in f32, rounded once)
  const int st = bilinear_fwd_impl(in, in_dtype, B, Hi, Wi, C, isB, isH, isW, out, out_dtype, Ho, Wo, osB, osH, osW, 0, base, stream);
  if (st != GDL_ERR_UNSUPPORTED) return st;
  const int st2 = gdl_copy_cast(base, out_dtype, B, Ho, Wo, C, osB, osH, osW, out, out_dtype, osB, osH, osW, stream);
  if (st2 != GDL_OK) return st2;
  return bilinear_fwd_impl(in, in_dtype, B, Hi, Wi, C, isB, isH, isW, out, out_dtype, Ho, Wo, osB, osH, osW, 1, nullptr, stream);
}

extern "C" int gdl_copy_cast(const void* in, int in_dtype, int B, int H, int W, int C, int64_t isB, int64_t isH, int64_t isW,
                             void* out, int out_dtype, int64_t osB, int64_t osH, int64_t osW, gdl_stream_t stream) {
  GDL_CHECK_ARG(in && out && B > 0 && H > 0 && W > 0 && C > 0, "gdl_copy_cast: bad args");
  GDL_CHECK_ARG((in_dtype == GDL_F32 || in_dtype == GDL_BF16) && (out_dtype == GDL_F32 || out_dtype == GDL_BF16), "gdl_copy_cast: bad dtype");
  GDL_CHECK_ARG(C % 4 == 0 && isB % 4 == 0 && isH % 4 == 0 && isW % 4 == 0 && osB % 4 == 0 && osH % 4 == 0 && osW % 4 == 0 &&
                    (uintptr_t)in % 8 == 0 && (uintptr_t)out % 8 == 0,
                "gdl_copy_cast: C, strides and pointers must keep 4-channel alignment");
  GDL_CHECK_ARG((int64_t)B * H <= 65535 * 1ll, "gdl_copy_cast: B * H must fit one grid dimension");
  if (in_dtype == GDL_BF16 && out_dtype == GDL_BF16 && vec8_ok(in, out, C, isB, isH, isW, osB, osH, osW)) {
    hipLaunchKernelGGL(copy8_kernel, dim3((unsigned)((W * (C / 8) + 255) / 256), (unsigned)(B * H)), dim3(256), 0,
                       (hipStream_t)stream, in, W, C, isB, isH, isW, out, H, osB, osH, osW);
  } else {
    const dim3 grid((unsigned)((W * (C / 4) + 255) / 256), (unsigned)(B * H));
    DISPATCH2(copy_cast_kernel, in_dtype, out_dtype, grid, dim3(256), 0, (hipStream_t)stream, in, W, C, isB, isH, isW, out, H, osB,
              osH, osW);
  }
  GDL_CHECK_LAUNCH("gdl_copy_cast");
  return GDL_OK;
}

extern "C" int gdl_bilinear_sum_fwd(const void* const* srcs, const int* hs, const int* ws, int nsrc, int dtype, int B, int C,
                                    void* out, int Ho, int Wo, gdl_stream_t stream) {
  GDL_CHECK_ARG(srcs && hs && ws && out && nsrc >= 1 && nsrc <= 3, "gdl_bilinear_sum_fwd: 1..3 sources");
  GDL_CHECK_ARG(dtype == GDL_F32 || dtype == GDL_BF16, "gdl_bilinear_sum_fwd: bad dtype");
  GDL_CHECK_ARG(B > 0 && Ho > 0 && Wo > 0 && C > 0 && C % (dtype == GDL_BF16 ? 8 : 4) == 0,
                "gdl_bilinear_sum_fwd: C must be a multiple of the 16-byte vector");
  SumSrc s;
  for (int k = 0; k < 3; ++k) {
    s.p[k] = k < nsrc ? srcs[k] : nullptr;
    s.H[k] = k < nsrc ? hs[k] : 1;
    s.W[k] = k < nsrc ? ws[k] : 1;
    GDL_CHECK_ARG(k >= nsrc || (srcs[k] && hs[k] > 0 && ws[k] > 0 && (uintptr_t)srcs[k] % 16 == 0), "gdl_bilinear_sum_fwd: bad source %d", k);
  }
  GDL_CHECK_ARG((uintptr_t)out % 16 == 0, "gdl_bilinear_sum_fwd: out must be 16-byte aligned");
  const int vec = dtype == GDL_BF16 ? 8 : 4;
  GDL_CHECK_ARG((int64_t)B * Ho <= 65535, "gdl_bilinear_sum_fwd: B * Ho must fit one grid dimension");
  const dim3 grid((unsigned)((Wo * (C / vec) + 255) / 256), (unsigned)(B * Ho));
  if (dtype == GDL_BF16)
    hipLaunchKernelGGL((bilinear_sum_kernel<uint16_t, 8>), grid, dim3(256), 0, (hipStream_t)stream, s, nsrc, B, C, out, Ho, Wo);
  else
    hipLaunchKernelGGL((bilinear_sum_kernel<float, 4>), grid, dim3(256), 0, (hipStream_t)stream, s, nsrc, B, C, out, Ho, Wo);
  GDL_CHECK_LAUNCH("gdl_bilinear_sum_fwd");
  return GDL_OK;
}

extern "C" int gdl_bilinear_bwd(const void* dout, int dout_dtype, int B, int Ho, int Wo, int C, int64_t osB,
                                int64_t osH, int64_t osW, void* din, int din_dtype, int Hi, int Wi,
                                int64_t isB, int64_t isH, int64_t isW, int accumulate, gdl_stream_t stream) {
  GDL_CHECK_ARG(dout && din, "gdl_bilinear_bwd: null pointer");
  GDL_CHECK_ARG(C % 4 == 0 && isB % 4 == 0 && isH % 4 == 0 && isW % 4 == 0 && osB % 4 == 0 && osH % 4 == 0 &&
                    osW % 4 == 0, "gdl_bilinear_bwd: C and strides must be multiples of 4");
  if (dout_dtype == GDL_BF16 && din_dtype == GDL_BF16 && vec8_ok(dout, din, C, isB, isH, isW, osB, osH, osW)) {
    const int64_t total8 = (int64_t)B * Hi * Wi * (C / 8);
    const int nx = 2 * ((Wo + Wi - 1) / Wi) + 4;            // horizontal output window of one input pixel, upper bound
    const dim3 grid_rows((unsigned)((Wi * (C / 8) + 255) / 256), (unsigned)(B * Hi));
#define BWD_ROWS(NX) hipLaunchKernelGGL(bilinear_bwd8_rows_kernel<NX>, grid_rows, dim3(256), 0, (hipStream_t)stream, dout, Ho, \
                                        Wo, C, osB, osH, osW, din, Hi, Wi, isB, isH, isW, accumulate)
    const int fac = (Hi > 1 && Wi > 1 && Ho % Hi == 0 && Wo % Wi == 0 && Ho / Hi == Wo / Wi) ? Ho / Hi : 0;
    if ((fac == 2 || fac == 4) && (int64_t)B * ((Hi + 7) / 8) <= 65535 && !g_flat_resample) {
      const dim3 grid((unsigned)((Wi * (C / 8) + 255) / 256), (unsigned)(B * ((Hi + 7) / 8)));
      if (fac == 2) hipLaunchKernelGGL((bilinear_bwd8_walk_kernel<2, 8>), grid, dim3(256), 0, (hipStream_t)stream, dout, C, osB, osH, osW, din, Hi, Wi, isB, isH, isW, accumulate);
      else hipLaunchKernelGGL((bilinear_bwd8_walk_kernel<4, 8>), grid, dim3(256), 0, (hipStream_t)stream, dout, C, osB, osH, osW, din, Hi, Wi, isB, isH, isW, accumulate);
    } else if ((int64_t)B * Hi <= 65535 && nx <= 20 && g_flat_resample != 1) {
      if (nx <= 8) BWD_ROWS(8); else if (nx <= 12) BWD_ROWS(12); else BWD_ROWS(20);
    } else if (nx > 20 && (int64_t)B * Hi * Wi <= 65535 && !g_flat_resample) {
      hipLaunchKernelGGL(bilinear_bwd8_window_kernel, dim3((unsigned)((C / 8 + 31) / 32), (unsigned)(B * Hi * Wi)), dim3(256), 0,
                         (hipStream_t)stream, dout, Ho, Wo, C, osB, osH, osW, din, Hi, Wi, isB, isH, isW, accumulate);
    } else
#undef BWD_ROWS
    hipLaunchKernelGGL(bilinear_bwd8_kernel, dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream, dout, B, Ho, Wo,
                       C, osB, osH, osW, din, Hi, Wi, isB, isH, isW, accumulate);
    GDL_CHECK_LAUNCH("gdl_bilinear_bwd");
    return GDL_OK;
  }
  const int64_t total = (int64_t)B * Hi * Wi * (C / 4);
  DISPATCH2(bilinear_bwd_kernel, dout_dtype, din_dtype, dim3(grid_for(total)), dim3(256), 0,
            (hipStream_t)stream, dout, B, Ho, Wo, C, osB, osH, osW, din, Hi, Wi, isB, isH, isW, accumulate);
  GDL_CHECK_LAUNCH("gdl_bilinear_bwd");
  return GDL_OK;
}

extern "C" int gdl_resize_conv3x3_bwd_gather(const void* dy, int dtype, int B, int Ho, int Wo, int N, void* g, int Hi, int Wi,
                                             gdl_stream_t stream) {
  GDL_CHECK_ARG(dy && g && B > 0 && Ho > 0 && Wo > 0 && Hi > 0 && Wi > 0, "gdl_resize_conv3x3_bwd_gather: bad args");
  GDL_CHECK_ARG(dtype == GDL_F32 || dtype == GDL_BF16, "gdl_resize_conv3x3_bwd_gather: bad dtype");
  const int vec = dtype == GDL_BF16 ? 8 : 4;
  GDL_CHECK_ARG(N % vec == 0 && (uintptr_t)dy % 16 == 0 && (uintptr_t)g % 16 == 0,
                "gdl_resize_conv3x3_bwd_gather: N must be a multiple of the 16-byte vector, pointers 16-byte aligned");
  GDL_CHECK_ARG((int64_t)B * Hi <= 65535, "gdl_resize_conv3x3_bwd_gather: B * Hi must fit one grid dimension");
  const int nx = 2 * ((Wo + Wi - 1) / Wi) + 4;
  GDL_CHECK_ARG(nx <= 24, "gdl_resize_conv3x3_bwd_gather: resize factors above 10 are not instantiated");
  const dim3 grid((unsigned)((Wi * (N / vec) + 255) / 256), (unsigned)(B * Hi));
  hipStream_t s = (hipStream_t)stream;
  if (gather_mfma_launch(dy, dtype, B, Ho, Wo, N, g, Hi, Wi, s)) {
    GDL_CHECK_LAUNCH("gdl_resize_conv3x3_bwd_gather");
    return GDL_OK;
  }
#define GATHER(V, VEC, NX, WXL) hipLaunchKernelGGL((resize_conv3x3_bwd_gather_kernel<V, VEC, NX, WXL>), grid, dim3(256), 0, s, dy, Ho, Wo, N, g, Hi, Wi)
  if (dtype == GDL_BF16) {
    if (nx <= 8) GATHER(V8, 8, 8, false); else if (nx <= 12) GATHER(V8, 8, 12, false); else if (nx <= 20) GATHER(V8, 8, 20, true);
    else GATHER(V8, 8, 24, true);                  // non-integer factors just above 8 (DOFA-large: 36 -> 292)
  } else {
    if (nx <= 8) GATHER(V4<float>, 4, 8, false); else if (nx <= 12) GATHER(V4<float>, 4, 12, false);
    else if (nx <= 20) GATHER(V4<float>, 4, 20, true); else GATHER(V4<float>, 4, 24, true);
  }
#undef GATHER
  GDL_CHECK_LAUNCH("gdl_resize_conv3x3_bwd_gather");
  return GDL_OK;
}

extern "C" int64_t gdl_resize_conv3x3_bwd_gather_workspace(int dtype, int B, int Wo, int N, int Hi) {
  if (B <= 0 || Wo <= 0 || N <= 0 || Hi <= 0) return 0;
  return 3ll * B * Hi * Wo * N * (dtype == GDL_BF16 ? 2 : 4);
}

// two-pass form of gdl_resize_conv3x3_bwd_gather; ws = gdl_resize_conv3x3_bwd_gather_workspace() bytes (16-byte aligned)
extern "C" int gdl_resize_conv3x3_bwd_gather2(const void* dy, int dtype, int B, int Ho, int Wo, int N, void* g, int Hi, int Wi,
                                              void* ws, int64_t ws_bytes, gdl_stream_t stream) {
  GDL_CHECK_ARG(dy && g && ws && B > 0 && Ho > 0 && Wo > 0 && Hi > 0 && Wi > 0, "gdl_resize_conv3x3_bwd_gather2: bad args");
  GDL_CHECK_ARG(dtype == GDL_F32 || dtype == GDL_BF16, "gdl_resize_conv3x3_bwd_gather2: bad dtype");
  const int vec = dtype == GDL_BF16 ? 8 : 4;
  GDL_CHECK_ARG(N % vec == 0 && (uintptr_t)dy % 16 == 0 && (uintptr_t)g % 16 == 0 && (uintptr_t)ws % 16 == 0,
                "gdl_resize_conv3x3_bwd_gather2: N must be a multiple of the 16-byte vector, pointers 16-byte aligned");
  GDL_CHECK_ARG(ws_bytes >= gdl_resize_conv3x3_bwd_gather_workspace(dtype, B, Wo, N, Hi), "gdl_resize_conv3x3_bwd_gather2: workspace too small");
  GDL_CHECK_ARG((int64_t)B * Hi <= 65535, "gdl_resize_conv3x3_bwd_gather2: B * Hi must fit one grid dimension");
  const int nx = 2 * ((Wo + Wi - 1) / Wi) + 4;
  GDL_CHECK_ARG(nx <= 24, "gdl_resize_conv3x3_bwd_gather2: resize factors above 10 are not instantiated");
  const int64_t plane = (int64_t)B * Hi * Wo * N;
  hipStream_t s = (hipStream_t)stream;
  if (gather_mfma_launch(dy, dtype, B, Ho, Wo, N, g, Hi, Wi, s)) {      // one pass on the matrix cores: the workspace stays unused
    GDL_CHECK_LAUNCH("gdl_resize_conv3x3_bwd_gather2");
    return GDL_OK;
  }
  const dim3 grid1((unsigned)((Wo * (N / vec) + 255) / 256), (unsigned)(B * Hi));
  const dim3 grid2((unsigned)((Wi * (N / vec) + 255) / 256), (unsigned)(B * Hi));
#define COLS(V, VEC, NX, WXL) hipLaunchKernelGGL((resize_conv3x3_bwd_cols_kernel<V, VEC, NX, WXL>), grid2, dim3(256), 0, s, ws, Wo, N, g, Hi, Wi, plane)
  if (dtype == GDL_BF16) {
    hipLaunchKernelGGL((resize_conv3x3_bwd_rows_kernel<V8, 8>), grid1, dim3(256), 0, s, dy, Ho, Wo, N, ws, Hi, plane);
    if (nx <= 8) COLS(V8, 8, 8, false); else if (nx <= 12) COLS(V8, 8, 12, false); else if (nx <= 20) COLS(V8, 8, 20, true);
    else COLS(V8, 8, 24, true);
  } else {
    hipLaunchKernelGGL((resize_conv3x3_bwd_rows_kernel<V4<float>, 4>), grid1, dim3(256), 0, s, dy, Ho, Wo, N, ws, Hi, plane);
    if (nx <= 8) COLS(V4<float>, 4, 8, false); else if (nx <= 12) COLS(V4<float>, 4, 12, false);
    else if (nx <= 20) COLS(V4<float>, 4, 20, true); else COLS(V4<float>, 4, 24, true);
  }
#undef COLS
  GDL_CHECK_LAUNCH("gdl_resize_conv3x3_bwd_gather2");
  return GDL_OK;
}

extern "C" int gdl_adaptive_avgpool_fwd(const void* in, int dtype, int B, int Hi, int Wi, int C, int64_t isB,
                                        int64_t isH, int64_t isW, void* out, int out_dtype, int So,
                                        gdl_stream_t stream) {
  GDL_CHECK_ARG(in && out && So > 0, "gdl_adaptive_avgpool_fwd: bad args");
  GDL_CHECK_ARG(C % 4 == 0 && isB % 4 == 0 && isH % 4 == 0 && isW % 4 == 0, "gdl_adaptive_avgpool_fwd: C/strides % 4");
  const int64_t total = (int64_t)B * So * So * (C / 4);
  if ((Hi / So) * (Wi / So) >= 32 && (int64_t)B * So * So <= 65535 && !g_flat_resample) {      // large bins: rows split over 8 lane groups
    const dim3 grid((unsigned)((C / 4 + 31) / 32), (unsigned)(B * So * So));
    DISPATCH2(avgpool_fwd_split_kernel, dtype, out_dtype, grid, dim3(256), 0, (hipStream_t)stream, in, Hi, Wi, C, isB, isH, isW, out, So);
    GDL_CHECK_LAUNCH("gdl_adaptive_avgpool_fwd");
    return GDL_OK;
  }
  DISPATCH2(avgpool_fwd_kernel, dtype, out_dtype, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in,
            B, Hi, Wi, C, isB, isH, isW, out, So);
  GDL_CHECK_LAUNCH("gdl_adaptive_avgpool_fwd");
  return GDL_OK;
}

extern "C" int gdl_adaptive_avgpool_bwd(const void* dout, int dtype, int B, int So, int C, void* din,
                                        int din_dtype, int Hi, int Wi, int64_t isB, int64_t isH, int64_t isW,
                                        int accumulate, gdl_stream_t stream) {
  GDL_CHECK_ARG(dout && din && So > 0, "gdl_adaptive_avgpool_bwd: bad args");
  GDL_CHECK_ARG(C % 4 == 0 && isB % 4 == 0 && isH % 4 == 0 && isW % 4 == 0, "gdl_adaptive_avgpool_bwd: C/strides % 4");
  const int64_t total = (int64_t)B * Hi * Wi * (C / 4);
  DISPATCH2(avgpool_bwd_kernel, dtype, din_dtype, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
            dout, B, So, C, din, Hi, Wi, isB, isH, isW, accumulate);
  GDL_CHECK_LAUNCH("gdl_adaptive_avgpool_bwd");
  return GDL_OK;
}
