// Helpers for the fused "bilinear x4 upsample -> 3x3 convolution" of the neck's finest level
// (multilevel_neck.py:157-158: resize(scale_factor=4, bilinear, align_corners=False) followed by the 3x3 ConvModule).
//
// The composition is linear, and on the x4 grid it decomposes exactly: output pixel (4y + py, 4x + px) is a 3x3 (often
// 2x3 / 3x2 / 2x2) convolution of the LOW-RES map around (y, x) with phase-specific weights
//     Wc[py][px][n][sy][sx][c] = sum_{dy,dx} A[py][dy][sy] * A[px][dx][sx] * w[n][dy][dx][c],
// A[p][d][s] = coefficient of low-res row y + s - 1 in upsampled row 4y + p + d - 1 (bilinear weights 0.375/0.625 and
// 0.125/0.875, index clamping = replicate padding of the low-res map).  Phases 0 and 3 touch two low-res rows, phases 1
// and 2 three: 6.25 taps on average instead of 9 -> 31 % fewer MACs than convolving the upsampled map, and the
// [B,4H,4W,C] intermediate (1 GB at batch 32) is never written or read.  The convolution's zero padding at the border
// of the UPSAMPLED map only concerns the outermost output rows / columns: they are recomputed exactly by four 1x3 line
// convolutions (host side, gdlhip/nn.py) that overwrite them.
//
// Kernels here: NHWC border padding (replicate / zero) and the weight combination.  The phase convolutions themselves
// run on the implicit-GEMM kernel (conv_gemm.hip) with strided outputs.
#include "gdl_common.h"

namespace {

// out[b][y][x][:] = in[b][y - ph][x - pw][:]  (replicate: indices clamped; zero: zeros outside); 16-byte vectors
__global__ __launch_bounds__(256) void pad_nhwc_kernel(const uint4* __restrict__ in, int H, int W, int cv, int64_t isB,
                                                       int64_t isH, int64_t isW, uint4* __restrict__ out, int ph, int pw,
                                                       int zero_mode, int64_t total) {
  const int Ho = H + 2 * ph, Wo = W + 2 * pw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cv);
    int64_t t = i / cv;
    const int x = (int)(t % Wo); t /= Wo;
    const int y = (int)(t % Ho);
    const int64_t b = t / Ho;
    int sy = y - ph, sx = x - pw;
    const bool inside = (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W;
    sy = sy < 0 ? 0 : (sy > H - 1 ? H - 1 : sy);
    sx = sx < 0 ? 0 : (sx > W - 1 ? W - 1 : sx);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (inside || !zero_mode) v = in[b * isB + (int64_t)sy * isH + (int64_t)sx * isW + c];
    out[i] = v;
  }
}

// bilinear x4, align_corners=False: upsampled index 4y + p reads low-res rows y-1, y, y+1 with these weights
__device__ __forceinline__ float alpha4(int p, int t) {   // t = 0,1,2 <-> low-res offset -1, 0, +1
  const float a[4][3] = {{0.375f, 0.625f, 0.f}, {0.125f, 0.875f, 0.f}, {0.f, 0.875f, 0.125f}, {0.f, 0.625f, 0.375f}};
  return a[p][t];
}
// A[p][d][s]: coefficient of low-res offset s-1 in upsampled row 4y + p + (d-1)
__device__ __forceinline__ float coefA(int p, int d, int s) {
  const int q = p + d - 1;                       // -1 .. 4
  const int yo = q < 0 ? -1 : (q > 3 ? 1 : 0);   // low-res row shift of the neighbouring upsampled row
  const int pp = q - 4 * yo;
  const int t = s - yo;                          // s-1 = yo + (t-1)
  return (t < 0 || t > 2) ? 0.f : alpha4(pp, t);
}

template <typename T>
__global__ __launch_bounds__(256) void subpix4_weights_kernel(const float* __restrict__ w, int N, int C, void* g22,
                                                              void* g23, void* g32, void* g33, void* lines) {
  // one thread per (n, c): loads the nine taps once, emits all 100 phase taps + the four border-line matrices
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)N * C) return;
  const int n = (int)(i / C), c = (int)(i - (int64_t)n * C);
  float wt[3][3];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) wt[dy][dx] = w[(int64_t)n * 9 * C + (dy * 3 + dx) * C + c];
#pragma unroll
  for (int py = 0; py < 4; ++py) {
    const int R = (py == 0 || py == 3) ? 2 : 3, sy0 = py == 3 ? 1 : 0, iy = (py == 0 || py == 1) ? 0 : 1;
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      const int S = (px == 0 || px == 3) ? 2 : 3, sx0 = px == 3 ? 1 : 0, ix = (px == 0 || px == 1) ? 0 : 1;
      void* dst = R == 2 ? (S == 2 ? g22 : g23) : (S == 2 ? g32 : g33);
      const int64_t base = ((int64_t)(iy * 2 + ix) * N + n) * (R * S * C) + c;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        if (r >= R) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          if (s >= S) continue;
          float acc = 0.f;
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) acc += coefA(py, dy, sy0 + r) * coefA(px, dx, sx0 + s) * wt[dy][dx];
          ElemIO<T>::store(dst, base + (int64_t)(r * S + s) * C, acc);
        }
      }
    }
  }
  // Border lines of the upsampled map.  The convolution zero-pads THERE, and upsampled rows 0 / 1 (and 4H-2 / 4H-1) are
  // both the horizontally upsampled first (last) low-res row, so output row 0 is a 1x3 convolution of that line with
  // w[dy=0] + w[dy=+1] (row 4H-1: w[-1] + w[0]); columns likewise.  lines = [top | bottom | left | right][N][3][C]
  const int64_t ls = (int64_t)N * 3 * C, lb = (int64_t)n * 3 * C + c;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    ElemIO<T>::store(lines, 0 * ls + lb + t * C, wt[1][t] + wt[2][t]);
    ElemIO<T>::store(lines, 1 * ls + lb + t * C, wt[0][t] + wt[1][t]);
    ElemIO<T>::store(lines, 2 * ls + lb + t * C, wt[t][1] + wt[t][2]);
    ElemIO<T>::store(lines, 3 * ls + lb + t * C, wt[t][0] + wt[t][1]);
  }
}


}  // namespace


extern "C" int gdl_pad_nhwc(const void* in, int dtype, int B, int H, int W, int C, int64_t in_sB, int64_t in_sH,
                            int64_t in_sW, void* out, int pad_h, int pad_w, int zero_mode, gdl_stream_t stream) {
  GDL_CHECK_ARG(in && out && B > 0 && H > 0 && W > 0 && pad_h >= 0 && pad_w >= 0, "gdl_pad_nhwc: bad args");
  const int al = dtype == GDL_BF16 ? 8 : 4;
  GDL_CHECK_ARG(C % al == 0 && in_sB % al == 0 && in_sH % al == 0 && in_sW % al == 0 && ((uintptr_t)in % 16 == 0) &&
                    ((uintptr_t)out % 16 == 0), "gdl_pad_nhwc: 16-byte alignment of channels / strides / pointers");
  const int cv = C / al;
  const int64_t total = (int64_t)B * (H + 2 * pad_h) * (W + 2 * pad_w) * cv;
  int64_t g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(pad_nhwc_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const uint4*)in, H, W, cv,
                     in_sB / al, in_sH / al, in_sW / al, (uint4*)out, pad_h, pad_w, zero_mode, total);
  GDL_CHECK_LAUNCH("gdl_pad_nhwc");
  return GDL_OK;
}

extern "C" int gdl_subpix4_weights(const float* w, int N, int C, int out_dtype, void* g22, void* g23, void* g32,
                                   void* g33, void* lines, gdl_stream_t stream) {
  GDL_CHECK_ARG(w && g22 && g23 && g32 && g33 && lines && N > 0 && C > 0, "gdl_subpix4_weights: bad args");
  const unsigned grid = (unsigned)(((int64_t)N * C + 255) / 256);
  if (out_dtype == GDL_BF16)
    hipLaunchKernelGGL(subpix4_weights_kernel<bf16_tag>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, N, C, g22, g23, g32, g33, lines);
  else
    hipLaunchKernelGGL(subpix4_weights_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, N, C, g22, g23, g32, g33, lines);
  GDL_CHECK_LAUNCH("gdl_subpix4_weights");
  return GDL_OK;
}
