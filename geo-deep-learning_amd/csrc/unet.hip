// HBM-bound pieces of the ResNet-encoder / UNet++ path (smp.UnetPlusPlus, reference call site
// tasks_with_models/segmentation_unetplus.py:126-131): 3x3/s2 max-pool, nearest x2 up-sampling written straight
// into the dense-skip concat buffer, residual add + ReLU.  NHWC, 8 channels (16 bytes bf16) / 4 channels (f32)
// per thread, strided tensors so concat slices are read / written in place.
#include "gdl_common.h"

namespace {

template <typename T> struct Vec;   // VEC channels per thread = one 16-byte access
template <> struct Vec<bf16_tag> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const void* p, int64_t i, float (&v)[8]) {
    const uint4 r = *(const uint4*)((const uint16_t*)p + i);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(w[e] << 16); v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void store(void* p, int64_t i, const float (&v)[8]) {
    *(uint4*)((uint16_t*)p + i) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                             pack_bf16x2(v[6], v[7]));
  }
};
template <> struct Vec<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const void* p, int64_t i, float (&v)[4]) {
    const float4 r = *(const float4*)((const float*)p + i);
    v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
  }
  static __device__ __forceinline__ void store(void* p, int64_t i, const float (&v)[4]) {
    *(float4*)((float*)p + i) = make_float4(v[0], v[1], v[2], v[3]);
  }
};

struct Strides { int64_t sB, sH, sW; };

// ---------------------------------------------------------------- max-pool 3x3 / stride 2 / pad 1
// F.max_pool2d semantics: padding is -inf, the window is scanned (kh, kw) and a later element only wins
// with a strict '>' -- so the FIRST maximum of the window receives the gradient.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const void* __restrict__ in, int B, int H, int W, int C,
                                                          Strides is, void* out, int Ho, int Wo, Strides os) {
  constexpr int V = Vec<T>::N;
  const int cv = C / V;
  const int64_t total = (int64_t)B * Ho * Wo * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cv) * V;
    int64_t t = i / cv;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float m[V];
#pragma unroll
    for (int e = 0; e < V; ++e) m[e] = -INFINITY;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int y = oy * 2 - 1 + r;
      if ((unsigned)y >= (unsigned)H) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int x = ox * 2 - 1 + s;
        if ((unsigned)x >= (unsigned)W) continue;
        float v[V];
        Vec<T>::load(in, (int64_t)b * is.sB + (int64_t)y * is.sH + (int64_t)x * is.sW + c, v);
#pragma unroll
        for (int e = 0; e < V; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
      }
    }
    Vec<T>::store(out, (int64_t)b * os.sB + (int64_t)oy * os.sH + (int64_t)ox * os.sW + c, m);
  }
}

// gather form: din[y,x] = sum over the (<= 4) windows containing (y,x) whose first maximum is (y,x).
// A block owns 16 x 8 input pixels x 8 channel vectors of one image.  Phase 1: the first-maximum position of each of the
// 9 x 5 windows that touch those pixels (one byte per channel, kept in LDS) -- every window is scanned ONCE per block
// instead of once per pixel it contains (the per-pixel form re-scanned up to four windows of nine loads each: 1.45 ms for the
// ResNet stem's [32,256,256,64] map).  Phase 2: each pixel adds the dout of the windows that elected it, rows then columns
// ascending -- a fixed order, no atomics.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const void* __restrict__ in, int B, int H, int W, int C,
                                                          Strides is, const void* __restrict__ dout, int Ho, int Wo,
                                                          Strides ds, void* din, Strides gs, int tiles_x) {
  constexpr int V = Vec<T>::N;
  __shared__ unsigned char widx[9][5][8][V];
  const int cv = C / V;
  const int b = blockIdx.y, cv0 = blockIdx.z * 8;
  const int Y0 = (blockIdx.x / tiles_x) * 16, X0 = (blockIdx.x % tiles_x) * 8;
  const int oy0 = Y0 >> 1, ox0 = X0 >> 1;
  for (int item = threadIdx.x; item < 9 * 5 * 8; item += 256) {
    const int cvi = item & 7, wx = (item >> 3) % 5, wy = item / 40;
    const int oy = oy0 + wy, ox = ox0 + wx, c = (cv0 + cvi) * V;
    unsigned char pos[V];
#pragma unroll
    for (int e = 0; e < V; ++e) pos[e] = 15;
    if (oy < Ho && ox < Wo && cv0 + cvi < cv) {
      float m[V];
#pragma unroll
      for (int e = 0; e < V; ++e) m[e] = -INFINITY;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int yy = oy * 2 - 1 + r;
        if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int xx = ox * 2 - 1 + q;
          if ((unsigned)xx >= (unsigned)W) continue;
          float v[V];
          Vec<T>::load(in, (int64_t)b * is.sB + (int64_t)yy * is.sH + (int64_t)xx * is.sW + c, v);
#pragma unroll
          for (int e = 0; e < V; ++e)
            if (v[e] > m[e]) { m[e] = v[e]; pos[e] = (unsigned char)(r * 3 + q); }   // strict '>': the first maximum wins
        }
      }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) widx[wy][wx][cvi][e] = pos[e];
  }
  __syncthreads();
  for (int item = threadIdx.x; item < 16 * 8 * 8; item += 256) {
    const int cvi = item & 7, px = (item >> 3) & 7, py = item >> 6;
    const int y = Y0 + py, x = X0 + px, c = (cv0 + cvi) * V;
    if (y >= H || x >= W || cv0 + cvi >= cv) continue;
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    // window oy covers rows 2*oy-1 .. 2*oy+1: an even row lies in one window, an odd row in two
    for (int oy = y / 2; oy <= (y + 1) / 2; ++oy) {
      if (oy >= Ho) continue;
      const int ry = y - (oy * 2 - 1);
      for (int ox = x / 2; ox <= (x + 1) / 2; ++ox) {
        if (ox >= Wo) continue;
        const unsigned char code = (unsigned char)(ry * 3 + (x - (ox * 2 - 1)));
        float g[V];
        Vec<T>::load(dout, (int64_t)b * ds.sB + (int64_t)oy * ds.sH + (int64_t)ox * ds.sW + c, g);
        const unsigned char* w = widx[oy - oy0][ox - ox0][cvi];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += w[e] == code ? g[e] : 0.f;
      }
    }
    Vec<T>::store(din, (int64_t)b * gs.sB + (int64_t)y * gs.sH + (int64_t)x * gs.sW + c, acc);
  }
}

// ---------------------------------------------------------------- nearest x2
template <typename T>
__global__ __launch_bounds__(256) void nearest2x_fwd_kernel(const void* __restrict__ in, int B, int H, int W, int C,
                                                            Strides is, void* out, Strides os) {
  constexpr int V = Vec<T>::N;
  const int cv = C / V;
  const int64_t total = (int64_t)B * H * 2 * W * 2 * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cv) * V;
    int64_t t = i / cv;
    const int ox = (int)(t % (2 * W)); t /= 2 * W;
    const int oy = (int)(t % (2 * H));
    const int b = (int)(t / (2 * H));
    float v[V];
    Vec<T>::load(in, (int64_t)b * is.sB + (int64_t)(oy >> 1) * is.sH + (int64_t)(ox >> 1) * is.sW + c, v);
    Vec<T>::store(out, (int64_t)b * os.sB + (int64_t)oy * os.sH + (int64_t)ox * os.sW + c, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void nearest2x_bwd_kernel(const void* __restrict__ dout, int B, int H, int W, int C,
                                                            Strides ds, void* din, Strides gs) {
  constexpr int V = Vec<T>::N;
  const int cv = C / V;
  const int64_t total = (int64_t)B * H * W * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cv) * V;
    int64_t t = i / cv;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float v[V];
        Vec<T>::load(dout, (int64_t)b * ds.sB + (int64_t)(2 * y + r) * ds.sH + (int64_t)(2 * x + s) * ds.sW + c, v);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += v[e];
      }
    Vec<T>::store(din, (int64_t)b * gs.sB + (int64_t)y * gs.sH + (int64_t)x * gs.sW + c, acc);
  }
}

// ---------------------------------------------------------------- residual add + ReLU (dense tensors)
template <typename T>
__global__ __launch_bounds__(256) void add_relu_kernel(const void* __restrict__ a, const void* __restrict__ b, void* out,
                                                       int64_t nvec) {
  constexpr int V = Vec<T>::N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    float x[V], y[V];
    Vec<T>::load(a, i * V, x);
    Vec<T>::load(b, i * V, y);
#pragma unroll
    for (int e = 0; e < V; ++e) x[e] = fmaxf(x[e] + y[e], 0.f);
    Vec<T>::store(out, i * V, x);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const void* __restrict__ y, const void* __restrict__ dy, void* dx,
                                                       int64_t nvec) {
  constexpr int V = Vec<T>::N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    float o[V], g[V];
    Vec<T>::load(y, i * V, o);
    Vec<T>::load(dy, i * V, g);
#pragma unroll
    for (int e = 0; e < V; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
    Vec<T>::store(dx, i * V, g);
  }
}

// [P][C] (any C) -> [P][Cpad] in another dtype, channels >= C zero-filled
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void pad_channels_kernel(const void* __restrict__ in, int64_t P, int C, void* out,
                                                           int Cpad) {
  const int64_t total = P * Cpad;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i / Cpad;
    const int c = (int)(i - p * Cpad);
    ElemIO<TO>::store(out, i, c < C ? ElemIO<TI>::load(in, p * C + c) : 0.f);
  }
}

inline unsigned blocks_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 262144) b = 262144;
  if (b < 1) b = 1;
  return (unsigned)b;
}

inline bool aligned(const void* p, int dtype, int64_t a, int64_t b, int64_t c, int C) {
  const int v = dtype == GDL_BF16 ? 8 : 4;
  return ((uintptr_t)p % 16 == 0) && a % v == 0 && b % v == 0 && c % v == 0 && C % v == 0;
}

}  // namespace

extern "C" int gdl_maxpool3x3s2_fwd(const void* in, int dtype, int B, int H, int W, int C, int64_t in_sB, int64_t in_sH,
                                    int64_t in_sW, void* out, int64_t out_sB, int64_t out_sH, int64_t out_sW,
                                    gdl_stream_t stream) {
  GDL_CHECK_ARG(in && out && B > 0 && H > 0 && W > 0, "gdl_maxpool3x3s2_fwd: bad args");
  GDL_CHECK_ARG(aligned(in, dtype, in_sB, in_sH, in_sW, C) && aligned(out, dtype, out_sB, out_sH, out_sW, C),
                "gdl_maxpool3x3s2_fwd: channels / strides / pointers must keep 16-byte alignment");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const Strides is{in_sB, in_sH, in_sW}, os{out_sB, out_sH, out_sW};
  hipStream_t s = (hipStream_t)stream;
  if (dtype == GDL_BF16) {
    hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_tag>, dim3(blocks_for((int64_t)B * Ho * Wo * (C / 8))), dim3(256), 0, s, in, B, H, W, C, is, out, Ho, Wo, os);
  } else {
    hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(blocks_for((int64_t)B * Ho * Wo * (C / 4))), dim3(256), 0, s, in, B, H, W, C, is, out, Ho, Wo, os);
  }
  GDL_CHECK_LAUNCH("gdl_maxpool3x3s2_fwd");
  return GDL_OK;
}

extern "C" int gdl_maxpool3x3s2_bwd(const void* in, const void* dout, void* din, int dtype, int B, int H, int W, int C,
                                    int64_t in_sB, int64_t in_sH, int64_t in_sW, int64_t d_sB, int64_t d_sH,
                                    int64_t d_sW, int64_t g_sB, int64_t g_sH, int64_t g_sW, gdl_stream_t stream) {
  GDL_CHECK_ARG(in && dout && din && B > 0 && H > 0 && W > 0, "gdl_maxpool3x3s2_bwd: bad args");
  GDL_CHECK_ARG(aligned(in, dtype, in_sB, in_sH, in_sW, C) && aligned(dout, dtype, d_sB, d_sH, d_sW, C) &&
                    aligned(din, dtype, g_sB, g_sH, g_sW, C),
                "gdl_maxpool3x3s2_bwd: channels / strides / pointers must keep 16-byte alignment");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const Strides is{in_sB, in_sH, in_sW}, ds{d_sB, d_sH, d_sW}, gs{g_sB, g_sH, g_sW};
  hipStream_t s = (hipStream_t)stream;
  GDL_CHECK_ARG(B <= 65535, "gdl_maxpool3x3s2_bwd: batch too large for one launch");
  const int tiles_x = (W + 7) / 8, tiles_y = (H + 15) / 16;
  if (dtype == GDL_BF16) {
    hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_tag>, dim3(tiles_x * tiles_y, B, (C / 8 + 7) / 8), dim3(256), 0, s, in, B, H, W, C, is, dout, Ho, Wo, ds, din, gs, tiles_x);
  } else {
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(tiles_x * tiles_y, B, (C / 4 + 7) / 8), dim3(256), 0, s, in, B, H, W, C, is, dout, Ho, Wo, ds, din, gs, tiles_x);
  }
  GDL_CHECK_LAUNCH("gdl_maxpool3x3s2_bwd");
  return GDL_OK;
}

extern "C" int gdl_nearest2x_fwd(const void* in, int dtype, int B, int H, int W, int C, int64_t in_sB, int64_t in_sH,
                                 int64_t in_sW, void* out, int64_t out_sB, int64_t out_sH, int64_t out_sW,
                                 gdl_stream_t stream) {
  GDL_CHECK_ARG(in && out && B > 0 && H > 0 && W > 0, "gdl_nearest2x_fwd: bad args");
  GDL_CHECK_ARG(aligned(in, dtype, in_sB, in_sH, in_sW, C) && aligned(out, dtype, out_sB, out_sH, out_sW, C),
                "gdl_nearest2x_fwd: channels / strides / pointers must keep 16-byte alignment");
  const Strides is{in_sB, in_sH, in_sW}, os{out_sB, out_sH, out_sW};
  hipStream_t s = (hipStream_t)stream;
  if (dtype == GDL_BF16) {
    hipLaunchKernelGGL(nearest2x_fwd_kernel<bf16_tag>, dim3(blocks_for((int64_t)B * H * W * 4 * (C / 8))), dim3(256), 0, s, in, B, H, W, C, is, out, os);
  } else {
    hipLaunchKernelGGL(nearest2x_fwd_kernel<float>, dim3(blocks_for((int64_t)B * H * W * 4 * (C / 4))), dim3(256), 0, s, in, B, H, W, C, is, out, os);
  }
  GDL_CHECK_LAUNCH("gdl_nearest2x_fwd");
  return GDL_OK;
}

extern "C" int gdl_nearest2x_bwd(const void* dout, int dtype, int B, int H, int W, int C, int64_t d_sB, int64_t d_sH,
                                 int64_t d_sW, void* din, int64_t g_sB, int64_t g_sH, int64_t g_sW,
                                 gdl_stream_t stream) {
  GDL_CHECK_ARG(dout && din && B > 0 && H > 0 && W > 0, "gdl_nearest2x_bwd: bad args");
  GDL_CHECK_ARG(aligned(dout, dtype, d_sB, d_sH, d_sW, C) && aligned(din, dtype, g_sB, g_sH, g_sW, C),
                "gdl_nearest2x_bwd: channels / strides / pointers must keep 16-byte alignment");
  const Strides ds{d_sB, d_sH, d_sW}, gs{g_sB, g_sH, g_sW};
  hipStream_t s = (hipStream_t)stream;
  if (dtype == GDL_BF16) {
    hipLaunchKernelGGL(nearest2x_bwd_kernel<bf16_tag>, dim3(blocks_for((int64_t)B * H * W * (C / 8))), dim3(256), 0, s, dout, B, H, W, C, ds, din, gs);
  } else {
    hipLaunchKernelGGL(nearest2x_bwd_kernel<float>, dim3(blocks_for((int64_t)B * H * W * (C / 4))), dim3(256), 0, s, dout, B, H, W, C, ds, din, gs);
  }
  GDL_CHECK_LAUNCH("gdl_nearest2x_bwd");
  return GDL_OK;
}

extern "C" int gdl_add_relu(const void* a, const void* b, void* out, int dtype, int64_t n, gdl_stream_t stream) {
  const int v = dtype == GDL_BF16 ? 8 : 4;
  GDL_CHECK_ARG(a && b && out && n % v == 0, "gdl_add_relu: n must be a multiple of %d", v);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == GDL_BF16) hipLaunchKernelGGL(add_relu_kernel<bf16_tag>, dim3(blocks_for(n / v)), dim3(256), 0, s, a, b, out, n / v);
  else hipLaunchKernelGGL(add_relu_kernel<float>, dim3(blocks_for(n / v)), dim3(256), 0, s, a, b, out, n / v);
  GDL_CHECK_LAUNCH("gdl_add_relu");
  return GDL_OK;
}

extern "C" int gdl_relu_bwd(const void* y, const void* dy, void* dx, int dtype, int64_t n, gdl_stream_t stream) {
  const int v = dtype == GDL_BF16 ? 8 : 4;
  GDL_CHECK_ARG(y && dy && dx && n % v == 0, "gdl_relu_bwd: n must be a multiple of %d", v);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == GDL_BF16) hipLaunchKernelGGL(relu_bwd_kernel<bf16_tag>, dim3(blocks_for(n / v)), dim3(256), 0, s, y, dy, dx, n / v);
  else hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(blocks_for(n / v)), dim3(256), 0, s, y, dy, dx, n / v);
  GDL_CHECK_LAUNCH("gdl_relu_bwd");
  return GDL_OK;
}

extern "C" int gdl_pad_channels(const void* in, int in_dtype, int64_t P, int C, void* out, int out_dtype, int Cpad,
                                gdl_stream_t stream) {
  GDL_CHECK_ARG(in && out && C > 0 && Cpad >= C, "gdl_pad_channels: bad args");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(blocks_for(P * Cpad));
#define PC(TI, TO) hipLaunchKernelGGL((pad_channels_kernel<TI, TO>), grid, dim3(256), 0, s, in, P, C, out, Cpad)
  if (in_dtype == GDL_BF16) { if (out_dtype == GDL_BF16) PC(bf16_tag, bf16_tag); else PC(bf16_tag, float); }
  else { if (out_dtype == GDL_BF16) PC(float, bf16_tag); else PC(float, float); }
#undef PC
  GDL_CHECK_LAUNCH("gdl_pad_channels");
  return GDL_OK;
}
