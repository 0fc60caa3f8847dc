// GPU-side input augmentation: the reference's kornia pipeline
// (tasks_with_models/segmentation_dofa.py:91-121: h/v flip, rot90, RandomResizedCrop zoom-in / zoom-out, one of them
// per batch via random_apply=1, each with p = 0.5 per sample) runs on the main-process CPU before the transfer; here
// it is ONE HBM-bound kernel after the H2D copy, fused with the /255 + standardise step when the tile arrives raw.
// Per sample the host supplies {kind, k, y0, x0, h, w}: image taps are bilinear (F.interpolate / kornia
// align_corners=False on the crop), masks nearest; flips and quarter turns are exact index maps.
#include "gdl_common.h"

namespace {

enum { AUG_NONE = 0, AUG_HFLIP = 1, AUG_VFLIP = 2, AUG_ROT90 = 3, AUG_CROP = 4 };

template <typename TI>
__device__ __forceinline__ float tap(const TI* p, int64_t i, bool norm, float m, float s) {
  const float v = (float)p[i];
  return norm ? (v / 255.0f - m) / s : v;
}

template <typename TI>
__global__ __launch_bounds__(256) void augment_kernel(const TI* __restrict__ img, float* __restrict__ out,
                                                      const int64_t* __restrict__ mask, int64_t* __restrict__ out_mask,
                                                      int B, int C, int H, int W, const float* __restrict__ mean,
                                                      const float* __restrict__ stdv, const float* __restrict__ prm) {
  const int64_t HW = (int64_t)H * W;
  const int64_t total = (int64_t)B * (C + (mask ? 1 : 0)) * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % W);
    int64_t t = i / W;
    const int y = (int)(t % H);
    t /= H;
    const int planes = C + (mask ? 1 : 0);
    const int c = (int)(t % planes), b = (int)(t / planes);
    const float* q = prm + (int64_t)b * 8;
    const int kind = (int)q[0];
    const bool is_mask = c == C;
    int sy = y, sx = x;
    if (kind == AUG_HFLIP) sx = W - 1 - x;
    else if (kind == AUG_VFLIP) sy = H - 1 - y;
    else if (kind == AUG_ROT90) {          // torch.rot90(k) on (H, W): square tiles only (checked by the host)
      const int k = (int)q[1] & 3;
      if (k == 1) { sy = x; sx = W - 1 - y; }
      else if (k == 2) { sy = H - 1 - y; sx = W - 1 - x; }
      else if (k == 3) { sy = H - 1 - x; sx = y; }
    }
    if (kind != AUG_CROP) {
      if (is_mask) out_mask[(int64_t)b * HW + (int64_t)y * W + x] = mask[(int64_t)b * HW + (int64_t)sy * W + sx];
      else {
        const bool norm = mean != nullptr;
        out[((int64_t)b * C + c) * HW + (int64_t)y * W + x] =
            tap(img, ((int64_t)b * C + c) * HW + (int64_t)sy * W + sx, norm, norm ? mean[c] : 0.f, norm ? stdv[c] : 1.f);
      }
      continue;
    }
    // crop [y0, y0+h) x [x0, x0+w) resized to H x W
    const int y0 = (int)q[2], x0 = (int)q[3], h = (int)q[4], w = (int)q[5];
    if (is_mask) {                         // nearest: src = floor(dst * in/out)
      int my = (int)floorf((float)y * ((float)h / (float)H)), mx = (int)floorf((float)x * ((float)w / (float)W));
      my = my < h - 1 ? my : h - 1;
      mx = mx < w - 1 ? mx : w - 1;
      out_mask[(int64_t)b * HW + (int64_t)y * W + x] = mask[(int64_t)b * HW + (int64_t)(y0 + my) * W + (x0 + mx)];
      continue;
    }
    float fy = ((float)y + 0.5f) * ((float)h / (float)H) - 0.5f, fx = ((float)x + 0.5f) * ((float)w / (float)W) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int ya = (int)fy, xa = (int)fx;
    const int yb = ya + (ya < h - 1 ? 1 : 0), xb = xa + (xa < w - 1 ? 1 : 0);
    const float ly = fy - (float)ya, lx = fx - (float)xa;
    const bool norm = mean != nullptr;
    const float m = norm ? mean[c] : 0.f, s = norm ? stdv[c] : 1.f;
    const int64_t base = ((int64_t)b * C + c) * HW;
    const float v00 = tap(img, base + (int64_t)(y0 + ya) * W + (x0 + xa), norm, m, s);
    const float v01 = tap(img, base + (int64_t)(y0 + ya) * W + (x0 + xb), norm, m, s);
    const float v10 = tap(img, base + (int64_t)(y0 + yb) * W + (x0 + xa), norm, m, s);
    const float v11 = tap(img, base + (int64_t)(y0 + yb) * W + (x0 + xb), norm, m, s);
    // same association as ATen's upsample_bilinear2d: (1-ly)*((1-lx)*v00 + lx*v01) + ly*((1-lx)*v10 + lx*v11)
    out[base + (int64_t)y * W + x] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  }
}

}  // namespace

extern "C" int gdl_augment(const void* img, int kind, float* out, const int64_t* mask, int64_t* out_mask, int B, int C,
                           int H, int W, const float* mean, const float* stdv, const float* params,
                           gdl_stream_t stream) {
  GDL_CHECK_ARG(img && out && params && B > 0 && C > 0 && H > 0 && W > 0, "gdl_augment: bad args");
  GDL_CHECK_ARG((mask == nullptr) == (out_mask == nullptr), "gdl_augment: mask and out_mask go together");
  GDL_CHECK_ARG(kind >= GDL_RAW_U8 && kind <= GDL_RAW_F32, "gdl_augment: bad sample kind %d", kind);
  GDL_CHECK_ARG((mean == nullptr) == (stdv == nullptr), "gdl_augment: mean and std go together");
  const int64_t total = (int64_t)B * (C + (mask ? 1 : 0)) * H * W;
  int64_t g = (total + 255) / 256;
  if (g > 262144) g = 262144;
  hipStream_t s = (hipStream_t)stream;
#define AUG(T) hipLaunchKernelGGL(augment_kernel<T>, dim3((unsigned)g), dim3(256), 0, s, (const T*)img, out, mask, out_mask, B, C, H, W, mean, stdv, params)
  if (kind == GDL_RAW_U8) AUG(uint8_t);
  else if (kind == GDL_RAW_U16) AUG(uint16_t);
  else if (kind == GDL_RAW_I16) AUG(int16_t);
  else AUG(float);
#undef AUG
  GDL_CHECK_LAUNCH("gdl_augment");
  return GDL_OK;
}
