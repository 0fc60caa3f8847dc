// Depthwise 3x3 convolution (stride 1, pad 1) + bias (+ exact-erf GELU) on NHWC tensors: the DWConv
// inside SegFormer's Mix-FFN (mix_transformer.py:533-546, :56-63).  HBM-bound: a thread owns 4 channels of a row
// segment and slides the 3x3 window along x in registers (dwconv_walk.h).
#include "dwconv_walk.h"

namespace {

// thread -> (row segment, 4-channel vector), channel vectors fastest; blocks are dealt to the XCDs in contiguous
// ranges (block b runs on XCD b % 8) so that rows sharing halo lines meet in one L2.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const void* __restrict__ in, int H, int W, int C,
                                                        const float* __restrict__ w9, const float* __restrict__ bias,
                                                        int gelu, void* __restrict__ out, int seglen, int nseg,
                                                        int64_t total) {
  const int cv = C / 4;
  const int64_t nb8 = gridDim.x / 8;
  const int64_t blk = (int64_t)(blockIdx.x % 8) * nb8 + blockIdx.x / 8;
  const int64_t i = blk * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % cv) * 4;
  const gdldw::Seg sg = gdldw::seg_of(i / cv, H, W, seglen, nseg);
  gdldw::Taps tp;
  tp.load(w9, bias, C, c);
  gdldw::walk<TI>(in, sg, H, W, C, c, [&](int x, const float (&L)[3][4], const float (&M)[3][4], const float (&R)[3][4]) {
    float acc[4];
    tp.apply(L, M, R, acc);
    if (gelu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = gelu_erf(acc[j]);
    }
    gdldw::Px<TO>::st(out, (sg.pix + x) * C + c, acc);
  });
}

}  // namespace

extern "C" int gdl_dwconv3x3(const void* in, int dtype, int B, int H, int W, int C, const float* w9, const float* bias,
                             int gelu, void* out, int out_dtype, gdl_stream_t stream) {
  GDL_CHECK_ARG(in && w9 && bias && out, "gdl_dwconv3x3: null pointer");
  GDL_CHECK_ARG(C % 4 == 0 && B > 0 && H > 0 && W > 0, "gdl_dwconv3x3: C must be a multiple of 4");
  const int seglen = gdldw::seg_len(W), nseg = (W + seglen - 1) / seglen;
  const int64_t total = (int64_t)B * nseg * H * (C / 4);
  const int64_t nb = (((total + 255) / 256 + 7) / 8) * 8;
  GDL_CHECK_ARG(nb < (int64_t)1 << 31, "gdl_dwconv3x3: tensor too large");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)nb), block(256);
#define DWL(TI, TO) hipLaunchKernelGGL((dwconv3x3_kernel<TI, TO>), grid, block, 0, s, in, H, W, C, w9, bias, gelu, out, seglen, nseg, total)
  if (dtype == GDL_BF16 && out_dtype == GDL_BF16) DWL(uint16_t, uint16_t);
  else if (dtype == GDL_BF16) DWL(uint16_t, float);
  else if (out_dtype == GDL_BF16) DWL(float, uint16_t);
  else DWL(float, float);
#undef DWL
  GDL_CHECK_LAUNCH("gdl_dwconv3x3");
  return GDL_OK;
}
