// Depthwise 3x3 convolution (stride 1, pad 1) + bias (+ exact-erf GELU) on NHWC tensors: the DWConv
// inside SegFormer's Mix-FFN (mix_transformer.py:533-546, :56-63).  HBM-bound: one thread per
// (pixel, 4-channel vector); the 9 neighbour reads of adjacent pixels are served by L1/L2.
#include "gdl_common.h"

namespace {

template <typename T> struct Ld4;
template <> struct Ld4<float> {
  static __device__ __forceinline__ void ld(const void* p, int64_t off, float (&o)[4]) {
    const float4 v = *(const float4*)((const float*)p + off);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  static __device__ __forceinline__ void st(void* p, int64_t off, const float (&o)[4]) {
    *(float4*)((float*)p + off) = make_float4(o[0], o[1], o[2], o[3]);
  }
};
template <> struct Ld4<uint16_t> {
  static __device__ __forceinline__ void ld(const void* p, int64_t off, float (&o)[4]) {
    const uint2 v = *(const uint2*)((const uint16_t*)p + off);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void st(void* p, int64_t off, const float (&o)[4]) {
    *(uint2*)((uint16_t*)p + off) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
  }
};

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const void* __restrict__ in, int B, int H, int W, int C,
                                                        const float* __restrict__ w9, const float* __restrict__ bias,
                                                        int gelu, void* out) {
  const int cv = C / 4;
  const int64_t total = (int64_t)B * H * W * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cv) * 4;
    int64_t t = i / cv;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    float acc[4];
    {
      const float4 bb = *(const float4*)(bias + c);
      acc[0] = bb.x; acc[1] = bb.y; acc[2] = bb.z; acc[3] = bb.w;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int yy = y + r - 1;
      if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int xx = x + s - 1;
        if ((unsigned)xx >= (unsigned)W) continue;
        float v[4];
        Ld4<TI>::ld(in, (((int64_t)b * H + yy) * W + xx) * C + c, v);
        const float4 ww = *(const float4*)(w9 + (r * 3 + s) * C + c);
        acc[0] += v[0] * ww.x; acc[1] += v[1] * ww.y; acc[2] += v[2] * ww.z; acc[3] += v[3] * ww.w;
      }
    }
    if (gelu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = gelu_erf(acc[j]);
    }
    Ld4<TO>::st(out, (((int64_t)b * H + y) * W + x) * C + c, acc);
  }
}

}  // namespace

extern "C" int gdl_dwconv3x3(const void* in, int dtype, int B, int H, int W, int C, const float* w9, const float* bias,
                             int gelu, void* out, int out_dtype, gdl_stream_t stream) {
  GDL_CHECK_ARG(in && w9 && bias && out, "gdl_dwconv3x3: null pointer");
  GDL_CHECK_ARG(C % 4 == 0 && B > 0 && H > 0 && W > 0, "gdl_dwconv3x3: C must be a multiple of 4");
  const int64_t total = (int64_t)B * H * W * (C / 4);
  int64_t g = (total + 255) / 256;
  if (g > 32768) g = 32768;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)g), block(256);
  if (dtype == GDL_BF16 && out_dtype == GDL_BF16) hipLaunchKernelGGL((dwconv3x3_kernel<uint16_t, uint16_t>), grid, block, 0, s, in, B, H, W, C, w9, bias, gelu, out);
  else if (dtype == GDL_BF16) hipLaunchKernelGGL((dwconv3x3_kernel<uint16_t, float>), grid, block, 0, s, in, B, H, W, C, w9, bias, gelu, out);
  else if (out_dtype == GDL_BF16) hipLaunchKernelGGL((dwconv3x3_kernel<float, uint16_t>), grid, block, 0, s, in, B, H, W, C, w9, bias, gelu, out);
  else hipLaunchKernelGGL((dwconv3x3_kernel<float, float>), grid, block, 0, s, in, B, H, W, C, w9, bias, gelu, out);
  GDL_CHECK_LAUNCH("gdl_dwconv3x3");
  return GDL_OK;
}
