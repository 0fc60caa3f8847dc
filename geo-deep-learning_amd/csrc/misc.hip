// Small HBM-bound kernels around the GEMMs: V transpose, row softmax, DOFA patchify / dynamic
// kernel packing, casts, broadcast adds, u8 normalise, classifier tail, Dice loss, Adam.
#include <type_traits>

#include <atomic>

#include "gdl_common.h"

namespace {

inline unsigned grid_for(int64_t total, int per_block = 256) {
  int64_t g = (total + per_block - 1) / per_block;
  return (unsigned)(g < 1 ? 1 : (g > 32768 ? 32768 : g));
}

// ------------------------------------------------------------------ V^T for attention
// V rows (token n, head h at v + b*v_sB + n*v_sN + h*hd) -> vt [B,H,hd,Npad] (keys >= N zero).
// 32x32 LDS tile transpose.
template <typename T>
__global__ __launch_bounds__(256) void v_transpose_kernel(const T* __restrict__ v, int N, int H, int hd,
                                                          int64_t v_sB, int64_t v_sN, T* __restrict__ vt, int Npad) {
  __shared__ T tile[32][33];
  const int bh = blockIdx.z, b = bh / H, h = bh % H;
  const int n0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t rowstride = v_sN;
  const T* src = v + (int64_t)b * v_sB + (int64_t)h * hd;
  for (int j = ty; j < 32; j += 8) {
    const int n = n0 + j, d = d0 + tx;
    tile[j][tx] = (n < N && d < hd) ? src[(int64_t)n * rowstride + d] : (T)0;
  }
  __syncthreads();
  T* dst = vt + ((int64_t)bh * hd) * Npad;
  for (int j = ty; j < 32; j += 8) {
    const int d = d0 + j, n = n0 + tx;
    if (d < hd && n < Npad) dst[(int64_t)d * Npad + n] = tile[tx][j];
  }
}

// ------------------------------------------------------------------ row softmax
// one wave per row; row cached in registers when it fits (n_cols <= 64*MAXV)
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const void* __restrict__ in, void* out, int64_t rows,
                                                           int n_valid, int n_cols) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t base = row * n_cols;
  float v[MAXV];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = i * 64 + lane;
    v[i] = c < n_valid ? ElemIO<T>::load(in, base + c) : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
  mx = wave_max(mx);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    v[i] = (i * 64 + lane) < n_valid ? expf(v[i] - mx) : 0.f;
    s += v[i];
  }
  const float inv = 1.f / wave_sum(s);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = i * 64 + lane;
    if (c < n_cols) ElemIO<T>::store(out, base + c, v[i] * inv);
  }
}

// ------------------------------------------------------------------ DOFA patch embed helpers
// im2col for conv2d(kernel P, stride, padding pad) on NCHW f32 -> [B*Gh*Gw][Kpad]; only used where the
// input has too few channels for the implicit-GEMM kernel (C = 3..10 bands: DOFA patch embed, MiT stem)
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, int B, int C, int H, int W,
                                                       int P, int stride, int pad, int Gh, int Gw, void* cols, int Kpad) {
  const int64_t total = (int64_t)B * Gh * Gw * Kpad;
  const int K = C * P * P;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int kk = (int)(i % Kpad);
    int64_t t = i / Kpad;
    const int gx = (int)(t % Gw); t /= Gw;
    const int gy = (int)(t % Gh);
    const int b = (int)(t / Gh);
    float v = 0.f;
    if (kk < K) {
      const int s = kk % P, r = (kk / P) % P, c = kk / (P * P);
      const int y = gy * stride + r - pad, x = gx * stride + s - pad;
      if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
        v = img[(((int64_t)b * C + c) * H + y) * W + x];
    }
    ElemIO<T>::store(cols, i, v);
  }
}

// Same, four consecutive k per thread (Kpad % 4 == 0): ONE index decomposition per 4 outputs, (s, r, c) advanced
// incrementally, one 8- or 16-byte store.  The element-per-thread form spent its time in integer divisions (1.4 ms for the
// 671 MB of the ResNet 7x7 / stride-2 stem at batch 32).
template <typename T>
__global__ __launch_bounds__(256) void patchify4_kernel(const float* __restrict__ img, int B, int C, int H, int W,
                                                        int P, int stride, int pad, int Gh, int Gw, void* cols, int Kpad) {
  const int kq = Kpad >> 2;
  const int64_t total = (int64_t)B * Gh * Gw * kq;
  const int K = C * P * P;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int kk = (int)(i % kq) * 4;
    int64_t t = i / kq;
    const int gx = (int)(t % Gw); t /= Gw;
    const int gy = (int)(t % Gh);
    const int b = (int)(t / Gh);
    int s = kk % P, r = (kk / P) % P, c = kk / (P * P);
    const int y0 = gy * stride - pad, x0 = gx * stride - pad;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int y = y0 + r, x = x0 + s;
      v[e] = (kk + e < K && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                 ? img[(((int64_t)b * C + c) * H + y) * W + x] : 0.f;
      if (++s == P) { s = 0; if (++r == P) { r = 0; ++c; } }
    }
    if constexpr (std::is_same<T, float>::value)
      *(float4*)((float*)cols + i * 4) = make_float4(v[0], v[1], v[2], v[3]);
    else
      *(uint2*)((uint16_t*)cols + i * 4) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
  }
}

// generated kernel G [C][P*P][D] f32 (dofa_v2.py:157-166) -> conv weight [D][Kpad] (k = c*P*P + rs), * scaler
template <typename T>
__global__ __launch_bounds__(256) void dofa_pack_kernel(const float* __restrict__ g, int C, int PP, int D,
                                                        float scaler, void* out, int Kpad) {
  const int64_t total = (int64_t)D * Kpad;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int kk = (int)(i % Kpad), d = (int)(i / Kpad);
    float v = 0.f;
    if (kk < C * PP) v = g[(int64_t)kk * D + d] * scaler;
    ElemIO<T>::store(out, i, v);
  }
}

// backward of the pack: dG[kk][d] = scaler * dW[d][kk] for kk < C*P*P (f32)
__global__ __launch_bounds__(256) void dofa_unpack_grad_kernel(const float* __restrict__ dw, int CPP, int D, float scaler,
                                                               int Kpad, float* __restrict__ dg) {
  const int64_t total = (int64_t)CPP * D;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int d = (int)(i % D), kk = (int)(i / D);
    dg[i] = dw[(int64_t)d * Kpad + kk] * scaler;
  }
}

// ------------------------------------------------------------------ elementwise
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const void* __restrict__ in, void* out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    ElemIO<TO>::store(out, i, ElemIO<TI>::load(in, i));
}

__global__ __launch_bounds__(256) void scale_f32_kernel(const float* __restrict__ in, float* out, int64_t n, float s) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = in[i] * s;
}

// out[r,:] = a[(r % a_rows),:] + (b ? b[(r % b_rows),:] : 0)
__global__ __launch_bounds__(256) void add_rows_kernel(const float* __restrict__ a, int64_t a_rows, int64_t a_stride,
                                                       const float* __restrict__ b, int64_t b_rows, int64_t b_stride,
                                                       float* out, int64_t out_stride, int64_t rows, int D) {
  const int64_t total = rows * D;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / D;
    const int c = (int)(i - r * D);
    float v = a[(r % a_rows) * a_stride + c];
    if (b) v += b[(r % b_rows) * b_stride + c];
    out[r * out_stride + c] = v;
  }
}

__global__ __launch_bounds__(256) void normalize_u8_kernel(const uint8_t* __restrict__ in, float* out, int C,
                                                           int64_t HW, int64_t total4, const float* __restrict__ mean,
                                                           const float* __restrict__ stdv) {
  // 4 pixels per thread (HW % 4 == 0): (x/255 - mean)/std exactly as utils/tensors.py:10-35
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(((i * 4) / HW) % C);
    const uchar4 u = *(const uchar4*)(in + i * 4);
    const float m = mean[c], s = stdv[c];
    float4 o;
    o.x = ((float)u.x / 255.0f - m) / s; o.y = ((float)u.y / 255.0f - m) / s;
    o.z = ((float)u.z / 255.0f - m) / s; o.w = ((float)u.w / 255.0f - m) / s;
    *(float4*)(out + i * 4) = o;
  }
}

// same arithmetic for any raw sample dtype the reference's `.float()` accepts (uint8 / uint16 / int16 / f32)
template <typename TI>
__global__ __launch_bounds__(256) void normalize_raw_kernel(const TI* __restrict__ in, float* out, int C, int64_t HW,
                                                            int64_t total, const float* __restrict__ mean,
                                                            const float* __restrict__ stdv) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)((i / HW) % C);
    out[i] = ((float)in[i] / 255.0f - mean[c]) / stdv[c];
  }
}

// ------------------------------------------------------------------ IoU counts (torchmetrics MeanIoU update)
// per sample b and class k: intersection |pred==k & target==k|, |pred==k|, |target==k| as exact 64-bit counts.
// LDS histogram per block, integer atomics only (deterministic).
__global__ __launch_bounds__(256) void iou_counts_kernel(const int64_t* __restrict__ pred,
                                                         const int64_t* __restrict__ target, int64_t P, int K,
                                                         unsigned long long* __restrict__ counts) {
  __shared__ unsigned h[3 * 64];
  for (int i = threadIdx.x; i < 3 * K; i += 256) h[i] = 0;
  __syncthreads();
  const int b = blockIdx.y;
  const int64_t per = (P + gridDim.x - 1) / gridDim.x;
  const int64_t p0 = per * blockIdx.x, p1 = p0 + per < P ? p0 + per : P;
  for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) {
    const int64_t a = pred[(int64_t)b * P + p], t = target[(int64_t)b * P + p];
    const bool av = a >= 0 && a < K, tv = t >= 0 && t < K;
    if (av) atomicAdd(&h[K + (int)a], 1u);
    if (tv) atomicAdd(&h[2 * K + (int)t], 1u);
    if (av && a == t) atomicAdd(&h[(int)a], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * K; i += 256)
    if (h[i]) atomicAdd(&counts[((int64_t)b * 3 + i / K) * K + i % K], (unsigned long long)h[i]);
}

// ------------------------------------------------------------------ classifier tail
// 1x1 conv to K<=16 classes: one wave per pixel, 4 channels per lane per step.
template <typename T, int K>
__global__ __launch_bounds__(256) void head_1x1_kernel(const void* __restrict__ feat, int64_t P, int C, int64_t f_sP,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       const float* __restrict__ chan_scale, int64_t pix_per_img,
                                                       float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= P) return;
  float acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = 0.f;
  const float* cs = chan_scale ? chan_scale + (p / pix_per_img) * C : nullptr;
  for (int c = lane * 4; c < C; c += 256) {
    float v[4];
    if constexpr (sizeof(T) == 4) {
      const float4 t = *(const float4*)((const float*)feat + p * f_sP + c);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
      const uint2 t = *(const uint2*)((const uint16_t*)feat + p * f_sP + c);
      v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
      v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    }
    if (cs) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] *= cs[c + j];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float4 ww = *(const float4*)(w + (int64_t)k * C + c);
      acc[k] += (v[0] * ww.x + v[1] * ww.y) + (v[2] * ww.z + v[3] * ww.w);
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float s = wave_sum(acc[k]);
    if (lane == 0) out[p * K + k] = s + (bias ? bias[k] : 0.f);
  }
}

// The same head as a skinny MFMA GEMM (round 5): D[class][pixel] = W[class][c] . feat[pixel][c] with v_mfma_f32_16x16x32_bf16.
// The wave-per-pixel kernel above runs at 1.3 TB/s on the decoder output (340 MB, 267 us at batch 32): every pixel re-reads the
// K x C weights through the L1 (10 bytes of L1 traffic per byte of features), spends ~100 VALU instructions on five 64-lane
// butterflies, and a wave lives for one 8-byte load per lane.  Here a wave owns 16-pixel tiles (16 x C bf16, contiguous in HBM):
//   * NKS coalesced 16-byte loads per lane fetch the tile (the NEXT tile's are issued before this tile's MFMAs: 8 KB in flight per
//     wave), a wave-private LDS slot re-shapes it into B fragments (chunk index XOR row: conflict-free both ways);
//   * the f32 weights live in registers as bf16 hi + lo fragments (w = hi + lo to 2^-17: f32-grade logits), two MFMAs per 32 channels;
//   * lane (pixel j, group g) ends with classes 4g .. 4g+3 of its pixel: 4-float stores.
// Dense bf16 features with C = 32 NKS and no Dropout2d scale; everything else takes the kernel above.
template <int NKS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void head_1x1_mfma_kernel(const uint16_t* __restrict__ feat, int64_t P, const float* __restrict__ w,
                                                            const float* __restrict__ bias, int K, float* __restrict__ out) {
  constexpr int C = NKS * 32, ROW = C * 2, TILE = 16 * ROW;
  extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned char* st = hsm + wv * TILE;
  const int j = lane & 15, g = lane >> 4;
  bf16x8_t wh[NKS], wl[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x0 = j < K ? w[(int64_t)j * C + 32 * ks + 8 * g + 2 * e] : 0.f;
      const float x1 = j < K ? w[(int64_t)j * C + 32 * ks + 8 * g + 2 * e + 1] : 0.f;
      hi[e] = pack_bf16x2(x0, x1);
      lo[e] = pack_bf16x2(x0 - __uint_as_float(hi[e] << 16), x1 - __uint_as_float(hi[e] & 0xffff0000u));
    }
    wh[ks] = __builtin_bit_cast(bf16x8_t, make_uint4(hi[0], hi[1], hi[2], hi[3]));
    wl[ks] = __builtin_bit_cast(bf16x8_t, make_uint4(lo[0], lo[1], lo[2], lo[3]));
  }
  float bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = (bias && 4 * g + r < K) ? bias[4 * g + r] : 0.f;
  const int64_t ntiles = (P + 15) / 16, stride = (int64_t)gridDim.x * 4;
  uint4 rg[NKS];
  auto gload = [&](int64_t tile) {
    const unsigned char* base = (const unsigned char*)feat + tile * TILE;
#pragma unroll
    for (int u = 0; u < NKS; ++u) {
      const int off = u * 1024 + lane * 16;
      rg[u] = tile * 16 + off / ROW < P ? *(const uint4*)(base + off) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  int64_t t = (int64_t)blockIdx.x * 4 + wv;
  if (t < ntiles) gload(t);
  for (; t < ntiles; t += stride) {
#pragma unroll
    for (int u = 0; u < NKS; ++u) {
      const int off = u * 1024 + lane * 16, row = off / ROW, c = (off % ROW) >> 4;
      *(uint4*)(st + row * ROW + ((c ^ (row & 15)) << 4)) = rg[u];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (wave-private slot: the LDS queue is in order, no barrier needed)
    if (t + stride < ntiles) gload(t + stride);
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const bf16x8_t fb = __builtin_bit_cast(bf16x8_t, *(const uint4*)(st + j * ROW + (((4 * ks + g) ^ j) << 4)));
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks], fb, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ks], fb, acc, 0, 0, 0);
    }
    const int64_t p = t * 16 + j;
    if (p < P) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * g + r < K) out[p * K + 4 * g + r] = acc[r] + bv[r];
    }
    asm volatile("" ::: "memory");                          // the next tile's LDS writes stay behind this tile's fragment reads
  }
}

// The same for C = 256 NW channels and / or a Dropout2d channel scale (SegFormer's 768-wide decoder, segformer_mlp.py:64-65,120-125; the
// DOFA heads in training when the map's pixel count is a multiple of 16).  The NW waves of a workgroup each take a 256-channel slice
// of the SAME 16-pixel tile (their 128 weight registers are that slice's hi / lo fragments), the partial 16 x 16 products meet in the
// LDS and wave 0 adds them in a fixed order.  A workgroup walks a CONTIGUOUS range of tiles, so the image index changes a handful of
// times: with a channel scale the fragments are those of w * scale[image] (the scale folded into the weights in f32 before the
// hi / lo split -- the features stay the bf16 values in memory), rebuilt when the image changes.  pix_per_img % 16 == 0 then (no tile
// straddles two images).
template <int NW, bool SCALE>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(SCALE ? 2 : 3, SCALE ? 2 : 3)))   // (SCALE: rebuilding the fragments
void head_1x1_mfma_wide_kernel(                                                 //  mid-loop spills 55 registers at three waves per SIMD)
const uint16_t* __restrict__ feat, int64_t P, const float* __restrict__ w, const float* __restrict__ bias,
                               const float* __restrict__ chan_scale, int64_t pix_per_img, int K, float* __restrict__ out) {
  constexpr int NKS = 8, C = 256 * NW, ROW = 512, TILE = 16 * ROW;
  extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* st = hsm + wv * TILE;
  f32x4_t* red = (f32x4_t*)(hsm + NW * TILE);      // [2][NW][64]
  const int j = lane & 15, g = lane >> 4;
  bf16x8_t wh[NKS], wl[NKS];
  auto load_w = [&](const float* cs) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      uint32_t hi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = wv * 256 + 32 * ks + 8 * g + 2 * e;
        float x0 = j < K ? w[(int64_t)j * C + c] : 0.f, x1 = j < K ? w[(int64_t)j * C + c + 1] : 0.f;
        if (SCALE) { x0 *= cs[c]; x1 *= cs[c + 1]; }
        hi[e] = pack_bf16x2(x0, x1);
        lo[e] = pack_bf16x2(x0 - __uint_as_float(hi[e] << 16), x1 - __uint_as_float(hi[e] & 0xffff0000u));
      }
      wh[ks] = __builtin_bit_cast(bf16x8_t, make_uint4(hi[0], hi[1], hi[2], hi[3]));
      wl[ks] = __builtin_bit_cast(bf16x8_t, make_uint4(lo[0], lo[1], lo[2], lo[3]));
    }
  };
  float bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = (bias && 4 * g + r < K) ? bias[4 * g + r] : 0.f;
  const int64_t ntiles = (P + 15) / 16;
  const int64_t tpb = (ntiles + gridDim.x - 1) / gridDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * tpb, t1 = t0 + tpb < ntiles ? t0 + tpb : ntiles;
  uint4 rg[NKS];
  auto gload = [&](int64_t tile) {
#pragma unroll
    for (int u = 0; u < NKS; ++u) {
      const int64_t row = tile * 16 + 2 * u + (lane >> 5);
      rg[u] = row < P ? *(const uint4*)(feat + row * C + wv * 256 + (lane & 31) * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  if (!SCALE) load_w(nullptr);
  int64_t img = -1;
  int par = 0;
  if (t0 < t1) gload(t0);
  for (int64_t t = t0; t < t1; ++t) {
    if (SCALE) {
      const int64_t b = (t * 16) / pix_per_img;
      if (b != img) { img = b; load_w(chan_scale + b * C); }
    }
#pragma unroll
    for (int u = 0; u < NKS; ++u) {
      const int row = 2 * u + (lane >> 5), c = lane & 31;
      *(uint4*)(st + row * ROW + ((c ^ (row & 15)) << 4)) = rg[u];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (t + 1 < t1) gload(t + 1);
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const bf16x8_t fb = __builtin_bit_cast(bf16x8_t, *(const uint4*)(st + j * ROW + (((4 * ks + g) ^ j) << 4)));
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks], fb, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ks], fb, acc, 0, 0, 0);
    }
    if (NW > 1) {
      red[(par * NW + wv) * 64 + lane] = acc;
      __syncthreads();          // one barrier per tile: the two parities of `red` alternate, wave 0 is past its reads of a parity
                                // before anyone passes the NEXT barrier and writes that parity again
      if (wv == 0) {
#pragma unroll
        for (int q = 1; q < NW; ++q) {
          const f32x4_t o = red[(par * NW + q) * 64 + lane];
          acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2]; acc[3] += o[3];
        }
      }
      par ^= 1;
    }
    const int64_t p = t * 16 + j;
    if (wv == 0 && p < P) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * g + r < K) out[p * K + 4 * g + r] = acc[r] + bv[r];
    }
    asm volatile("" ::: "memory");
  }
}

// backward of the 1x1 head wrt features and weights
//  dfeat[p,c] = sum_k dlog[p,k] * w[k,c] (* chan_scale) ; dw[k,c] = sum_p dlog[p,k]*feat[p,c]*cs ; db[k] = sum_p dlog[p,k]
template <typename T, int K>
__global__ __launch_bounds__(256) void head_1x1_bwd_feat_kernel(const float* __restrict__ dlog, int64_t P, int C,
                                                                const float* __restrict__ w,
                                                                const float* __restrict__ chan_scale,
                                                                int64_t pix_per_img, void* dfeat, int64_t d_sP) {
  const int cv = C / 4;
  const int64_t total = P * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i / cv;
    const int c = (int)(i - p * cv) * 4;
    float g[K];
#pragma unroll
    for (int k = 0; k < K; ++k) g[k] = dlog[p * K + k];
    float o[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float4 ww = *(const float4*)(w + (int64_t)k * C + c);
      o[0] += g[k] * ww.x; o[1] += g[k] * ww.y; o[2] += g[k] * ww.z; o[3] += g[k] * ww.w;
    }
    if (chan_scale) {
      const float* cs = chan_scale + (p / pix_per_img) * C + c;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] *= cs[j];
    }
    if constexpr (sizeof(T) == 4) *(float4*)((float*)dfeat + p * d_sP + c) = make_float4(o[0], o[1], o[2], o[3]);
    else *(uint2*)((uint16_t*)dfeat + p * d_sP + c) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
  }
}

// The same for dense bf16 gradients with the weights in registers (round 5): a lane owns EIGHT fixed channels (one 16-byte store) and
// walks pixels; the kernel above re-reads its K float4 weight vectors from the L1 for every 8 bytes it stores (164 us for 340 MB).
// Per-element arithmetic: the same products added in the same order, as explicit fused multiply-adds.  256 % (C / 8) == 0, no
// Dropout2d scale.
template <int K>
__global__ __launch_bounds__(256) void head_1x1_bwd_feat8_kernel(const float* __restrict__ dlog, int64_t P, int C,
                                                                 const float* __restrict__ w, uint16_t* __restrict__ dfeat) {
  const int cv = C >> 3, q = threadIdx.x % cv, ro = threadIdx.x / cv, rpb = 256 / cv;
  float wr[K][8];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float4 a = *(const float4*)(w + (int64_t)k * C + q * 8), b = *(const float4*)(w + (int64_t)k * C + q * 8 + 4);
    wr[k][0] = a.x; wr[k][1] = a.y; wr[k][2] = a.z; wr[k][3] = a.w; wr[k][4] = b.x; wr[k][5] = b.y; wr[k][6] = b.z; wr[k][7] = b.w;
  }
  const int64_t step = (int64_t)gridDim.x * rpb;
  auto one = [&](int64_t p, const float (&gk)[K]) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = fmaf(gk[k], wr[k][e], o[e]);
    }
    *(uint4*)(dfeat + p * C + q * 8) = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
  };
  int64_t p = (int64_t)blockIdx.x * rpb + ro;
  for (; p + 3 * step < P; p += 4 * step) {       // four pixels' gradients requested before the first is used
    float gk[4][K];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int k = 0; k < K; ++k) gk[u][k] = dlog[(p + u * step) * K + k];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) one(p + u * step, gk[u]);
  }
  for (; p < P; p += step) {
    float gk[K];
#pragma unroll
    for (int k = 0; k < K; ++k) gk[k] = dlog[p * K + k];
    one(p, gk);
  }
}

// dw partials: grid (C/256, nsplit); each lane 4 channels; ws[split][K][C]; db via extra row K
template <typename T, int K>
__global__ __launch_bounds__(256) void head_1x1_bwd_w_partial(const void* __restrict__ feat, const float* __restrict__ dlog,
                                                              int64_t P, int C, int64_t f_sP,
                                                              const float* __restrict__ chan_scale,
                                                              int64_t pix_per_img, float* __restrict__ ws) {
  __shared__ float red[4][K][256];
  __shared__ float redb[4][K];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const int nsplit = gridDim.y;
  const int64_t per = (P + nsplit - 1) / nsplit;
  const int64_t p0 = per * blockIdx.y, p1 = p0 + per < P ? p0 + per : P;
  float acc[K][4];
  float accb[K];
#pragma unroll
  for (int k = 0; k < K; ++k) { accb[k] = 0.f; acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.f; }
  auto load = [&](int64_t p, float (&v)[4]) {
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (c < C) {
      if constexpr (sizeof(T) == 4) {
        const float4 t = *(const float4*)((const float*)feat + p * f_sP + c);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else {
        const uint2 t = *(const uint2*)((const uint16_t*)feat + p * f_sP + c);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
      }
      if (chan_scale) {
        const float* cs = chan_scale + (p / pix_per_img) * C + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= cs[j];
      }
    }
  };
  auto add = [&](int64_t p, const float (&v)[4]) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float g = dlog[p * K + k];
      accb[k] += g;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[k][j] += g * v[j];
    }
  };
  int64_t p = p0 + wv;
  for (; p + 12 < p1; p += 16) {      // four pixel rows in flight per wave (one at a time ran at 2 TB/s), same summation order
    // (round 5: eight rows in flight need 134 registers -- three waves per SIMD -- and run 1.6 x SLOWER; a forward kernel that
    //  walks 16-pixel runs per wave with four rows in flight loses the same way: 243 us against 133 -- profiles/r05n_*)
    float v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) load(p + 4 * u, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) add(p + 4 * u, v[u]);
  }
  for (; p < p1; p += 4) {
    float v[4];
    load(p, v);
    add(p, v);
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wv][k][lane * 4 + j] = acc[k][j];
    if (lane == 0) redb[wv][k] = accb[k];
  }
  __syncthreads();
  const int t = threadIdx.x, cc = blockIdx.x * 256 + t;
  float* wsb = ws + (int64_t)blockIdx.y * (K + 1) * C;
  if (cc < C) {
#pragma unroll
    for (int k = 0; k < K; ++k) wsb[(int64_t)k * C + cc] = (red[0][k][t] + red[1][k][t]) + (red[2][k][t] + red[3][k][t]);
  }
  if (blockIdx.x == 0 && t < K) wsb[(int64_t)K * C + t] = (redb[0][t] + redb[1][t]) + (redb[2][t] + redb[3][t]);
}

// dw partials for dense bf16 features (round 5): the shape of bn_bwd_partial8 -- a workgroup owns whole pixel rows, a lane EIGHT fixed
// channels (16-byte loads), four rows requested before the first is used (4 KB in flight per wave; the kernel above keeps 2 KB and
// pays one L2 round trip for the gradients after every batch of feature loads: 1.5-1.9 TB/s).  256 % (C / 8) == 0, K <= 8.
template <int K>
__global__ __launch_bounds__(256) void head_1x1_bwd_w_partial8(const uint16_t* __restrict__ feat, const float* __restrict__ dlog,
                                                               int64_t P, int C, float* __restrict__ ws) {
  __shared__ float red[256][9];             // [thread][channel | bias term] (+1: bank spread)
  const int G = C >> 3, R = 256 / G;        // lanes per pixel, pixel rows per block step
  const int t = threadIdx.x, g = t % G, prow = t / G;
  const int nsplit = gridDim.x;
  const int64_t per = (P + nsplit - 1) / nsplit;
  const int64_t p0 = per * blockIdx.x, p1 = p0 + per < P ? p0 + per : P;
  float acc[K][8], accb[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    accb[k] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
  }
  auto add = [&](const uint4& xv, const float (&gk)[K]) {
    float v[8];
    v[0] = __uint_as_float(xv.x << 16); v[1] = __uint_as_float(xv.x & 0xffff0000u);
    v[2] = __uint_as_float(xv.y << 16); v[3] = __uint_as_float(xv.y & 0xffff0000u);
    v[4] = __uint_as_float(xv.z << 16); v[5] = __uint_as_float(xv.z & 0xffff0000u);
    v[6] = __uint_as_float(xv.w << 16); v[7] = __uint_as_float(xv.w & 0xffff0000u);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      accb[k] += gk[k];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[k][j] = fmaf(gk[k], v[j], acc[k][j]);
    }
  };
  int64_t p = p0 + prow;
  for (; p + 3 * R < p1; p += 4 * R) {
    uint4 xv[4];
    float gk[4][K];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      xv[u] = *(const uint4*)(feat + (p + (int64_t)u * R) * C + 8 * g);
#pragma unroll
      for (int k = 0; k < K; ++k) gk[u][k] = dlog[(p + (int64_t)u * R) * K + k];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) add(xv[u], gk[u]);
  }
  for (; p < p1; p += R) {
    float gk[K];
#pragma unroll
    for (int k = 0; k < K; ++k) gk[k] = dlog[p * K + k];
    add(*(const uint4*)(feat + p * C + 8 * g), gk);
  }
  float* wsb = ws + (int64_t)blockIdx.x * (K + 1) * C;
#pragma unroll
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red[t][j] = acc[k][j];
    red[t][8] = accb[k];
    __syncthreads();
    for (int c = t; c < C; c += 256) {
      float ss = 0.f;
      for (int r = 0; r < R; ++r) ss += red[r * G + (c >> 3)][c & 7];
      wsb[(int64_t)k * C + c] = ss;
    }
    if (t == 0) {
      float sb = 0.f;
      for (int r = 0; r < R; ++r) sb += red[r * G][8];
      wsb[(int64_t)K * C + k] = sb;
    }
    __syncthreads();
  }
}

// General forms of the two kernels above: TPB = 256 or 192 threads (C / 8 lanes per pixel must divide it: 96 lanes for SegFormer's 768
// channels) and an optional Dropout2d channel scale, held per lane for the image it is working in (a workgroup walks a contiguous
// pixel range: the eight scale values are re-read a handful of times).  Arithmetic as in the 4-channel kernels: the scale multiplies
// the finished feature-gradient sum / the feature value before it enters the weight-gradient sum.
template <int K, int TPB, bool SCALE>
__global__ __launch_bounds__(TPB) void head_1x1_bwd_feat8g_kernel(const float* __restrict__ dlog, int64_t P, int C, const float* __restrict__ w,
                                                                  const float* __restrict__ chan_scale, int64_t pix_per_img,
                                                                  uint16_t* __restrict__ dfeat) {
  const int G = C >> 3, q = threadIdx.x % G, ro = threadIdx.x / G, R = TPB / G;
  float wr[K][8];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float4 a = *(const float4*)(w + (int64_t)k * C + q * 8), b = *(const float4*)(w + (int64_t)k * C + q * 8 + 4);
    wr[k][0] = a.x; wr[k][1] = a.y; wr[k][2] = a.z; wr[k][3] = a.w; wr[k][4] = b.x; wr[k][5] = b.y; wr[k][6] = b.z; wr[k][7] = b.w;
  }
  const int64_t per = (P + gridDim.x - 1) / gridDim.x;
  const int64_t p0 = per * blockIdx.x, p1 = p0 + per < P ? p0 + per : P;
  float csr[8];
  int64_t bc = -1;
  auto one = [&](int64_t p, const float (&gk)[K]) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = fmaf(gk[k], wr[k][e], o[e]);
    }
    if (SCALE) {
      const int64_t b = p / pix_per_img;
      if (b != bc) {
        bc = b;
        const float4 a = *(const float4*)(chan_scale + b * C + q * 8), c4 = *(const float4*)(chan_scale + b * C + q * 8 + 4);
        csr[0] = a.x; csr[1] = a.y; csr[2] = a.z; csr[3] = a.w; csr[4] = c4.x; csr[5] = c4.y; csr[6] = c4.z; csr[7] = c4.w;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] *= csr[e];
    }
    *(uint4*)(dfeat + p * C + q * 8) = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
  };
  if (ro >= R) return;
  int64_t p = p0 + ro;
  for (; p + 3 * R < p1; p += 4 * R) {
    float gk[4][K];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int k = 0; k < K; ++k) gk[u][k] = dlog[(p + (int64_t)u * R) * K + k];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) one(p + (int64_t)u * R, gk[u]);
  }
  for (; p < p1; p += R) {
    float gk[K];
#pragma unroll
    for (int k = 0; k < K; ++k) gk[k] = dlog[p * K + k];
    one(p, gk);
  }
}

template <int K, int TPB, bool SCALE>
__global__ __launch_bounds__(TPB) void head_1x1_bwd_w_partial8g(const uint16_t* __restrict__ feat, const float* __restrict__ dlog, int64_t P,
                                                                int C, const float* __restrict__ chan_scale, int64_t pix_per_img,
                                                                float* __restrict__ ws) {
  __shared__ float red[TPB][9];
  const int G = C >> 3, R = TPB / G;
  const int t = threadIdx.x, g = t % G, prow = t / G;
  const int nsplit = gridDim.x;
  const int64_t per = (P + nsplit - 1) / nsplit;
  const int64_t p0 = per * blockIdx.x, p1 = p0 + per < P ? p0 + per : P;
  float acc[K][8], accb[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    accb[k] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
  }
  float csr[8];
  int64_t bc = -1;
  auto add = [&](int64_t p, const uint4& xv, const float (&gk)[K]) {
    float v[8];
    v[0] = __uint_as_float(xv.x << 16); v[1] = __uint_as_float(xv.x & 0xffff0000u);
    v[2] = __uint_as_float(xv.y << 16); v[3] = __uint_as_float(xv.y & 0xffff0000u);
    v[4] = __uint_as_float(xv.z << 16); v[5] = __uint_as_float(xv.z & 0xffff0000u);
    v[6] = __uint_as_float(xv.w << 16); v[7] = __uint_as_float(xv.w & 0xffff0000u);
    if (SCALE) {
      const int64_t b = p / pix_per_img;
      if (b != bc) {
        bc = b;
        const float4 a = *(const float4*)(chan_scale + b * C + g * 8), c4 = *(const float4*)(chan_scale + b * C + g * 8 + 4);
        csr[0] = a.x; csr[1] = a.y; csr[2] = a.z; csr[3] = a.w; csr[4] = c4.x; csr[5] = c4.y; csr[6] = c4.z; csr[7] = c4.w;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= csr[j];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      accb[k] += gk[k];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[k][j] = fmaf(gk[k], v[j], acc[k][j]);
    }
  };
  if (prow < R) {
    int64_t p = p0 + prow;
    for (; p + 3 * R < p1; p += 4 * R) {
      uint4 xv[4];
      float gk[4][K];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        xv[u] = *(const uint4*)(feat + (p + (int64_t)u * R) * C + 8 * g);
#pragma unroll
        for (int k = 0; k < K; ++k) gk[u][k] = dlog[(p + (int64_t)u * R) * K + k];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) add(p + (int64_t)u * R, xv[u], gk[u]);
    }
    for (; p < p1; p += R) {
      float gk[K];
#pragma unroll
      for (int k = 0; k < K; ++k) gk[k] = dlog[p * K + k];
      add(p, *(const uint4*)(feat + p * C + 8 * g), gk);
    }
  }
  float* wsb = ws + (int64_t)blockIdx.x * (K + 1) * C;
#pragma unroll
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red[t][j] = acc[k][j];
    red[t][8] = accb[k];
    __syncthreads();
    for (int c = t; c < C; c += TPB) {
      float ss = 0.f;
      for (int r = 0; r < R; ++r) ss += red[r * G + (c >> 3)][c & 7];
      wsb[(int64_t)k * C + c] = ss;
    }
    if (t == 0) {
      float sb = 0.f;
      for (int r = 0; r < R; ++r) sb += red[r * G][8];
      wsb[(int64_t)K * C + k] = sb;
    }
    __syncthreads();
  }
}

template <int K>
__global__ __launch_bounds__(1024) void head_1x1_bwd_w_final(const float* __restrict__ ws, int nsplit, int C,
                                                             float* __restrict__ dw, float* __restrict__ db) {
  // outputs 0..K*C-1 = dw, K*C..K*C+K-1 = db; block = 8 outputs x 128 split groups (round 5: up to 2048 partial rows)
  __shared__ double part[128][8];
  const int cl = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const int i = blockIdx.x * 8 + cl;
  const int total = K * C + (db ? K : 0);
  double s = 0;
  if (i < total) {
    const int64_t off = i < K * C ? i : (int64_t)K * C + (i - K * C);
    s = ordered_sum8<double>(grp, nsplit, 128, [&](int sp) { return ws[(int64_t)sp * (K + 1) * C + off]; });
  }
  part[grp][cl] = s;
  __syncthreads();
  if (grp < 8) {          // 128 -> 8 -> 1, fixed order
    s = 0;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += part[grp * 16 + g][cl];
  }
  __syncthreads();
  if (grp < 8) part[grp][cl] = s;
  __syncthreads();
  if (grp != 0 || i >= total) return;
  s = 0;
#pragma unroll
  for (int g = 0; g < 8; ++g) s += part[g][cl];
  if (i < K * C) dw[i] = (float)s;
  else db[i - K * C] = (float)s;
}

__device__ __forceinline__ void src_index2(float ratio, int dst, int in_size, int& i0, int& i1, float& l1) {
  float s = ratio * ((float)dst + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
}

// NHWC [B,Hi,Wi,K] f32 -> NCHW [B,K,Ho,Wo] f32, bilinear align_corners=False
template <int K>
__global__ __launch_bounds__(256) void upsample_logits_kernel(const float* __restrict__ in, int B, int Hi, int Wi,
                                                              float* __restrict__ out, int Ho, int Wo) {
  const int64_t total = (int64_t)B * Ho * Wo;
  const float ry = (float)Hi / (float)Ho, rx = (float)Wi / (float)Wo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % Wo);
    const int64_t t = i / Wo;
    const int oy = (int)(t % Ho), b = (int)(t / Ho);
    int y0, y1, x0, x1; float ly, lx;
    src_index2(ry, oy, Hi, y0, y1, ly);
    src_index2(rx, ox, Wi, x0, x1, lx);
    const float* p00 = in + (((int64_t)b * Hi + y0) * Wi + x0) * K;
    const float* p01 = in + (((int64_t)b * Hi + y0) * Wi + x1) * K;
    const float* p10 = in + (((int64_t)b * Hi + y1) * Wi + x0) * K;
    const float* p11 = in + (((int64_t)b * Hi + y1) * Wi + x1) * K;
    const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
    for (int k = 0; k < K; ++k)
      out[(((int64_t)b * K + k) * Ho + oy) * Wo + ox] = hy * (hx * p00[k] + lx * p01[k]) + ly * (hx * p10[k] + lx * p11[k]);
  }
}

// Backward of the logit upsample, separable (bilinear weights factor as wy*wx), gather form:
//   pass 1: tmp[b,k,iy,ox] = sum_oy wy(oy->iy) * dout[b,k,oy,ox]      (coalesced along ox)
//   pass 2: din[b,iy,ix,k] = sum_ox wx(ox->ix) * tmp[b,k,iy,ox]
__device__ __forceinline__ void cand_range(int i, float ratio, int out_size, int& lo, int& hi) {
  lo = (int)floorf(((float)i - 0.5f) / ratio - 0.5f) - 1;
  hi = (int)ceilf(((float)i + 1.5f) / ratio - 0.5f) + 1;
  lo = lo < 0 ? 0 : lo;
  hi = hi > out_size - 1 ? out_size - 1 : hi;
}

__global__ __launch_bounds__(256) void upsample_logits_bwd_pass1(const float* __restrict__ dout, int BK, int Ho, int Wo,
                                                                 float* __restrict__ tmp, int Hi) {
  const int64_t total = (int64_t)BK * Hi * Wo;
  const float ry = (float)Hi / (float)Ho;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % Wo);
    const int64_t t = i / Wo;
    const int iy = (int)(t % Hi);
    const int64_t bk = t / Hi;
    int lo, hi;
    cand_range(iy, ry, Ho, lo, hi);
    float acc = 0.f;
    for (int oy = lo; oy <= hi; ++oy) {
      int y0, y1; float ly;
      src_index2(ry, oy, Hi, y0, y1, ly);
      const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
      if (wy != 0.f) acc += wy * dout[(bk * Ho + oy) * Wo + ox];
    }
    tmp[i] = acc;
  }
}

template <int K>
__global__ __launch_bounds__(256) void upsample_logits_bwd_pass2(const float* __restrict__ tmp, int B, int Wo,
                                                                 float* __restrict__ din, int Hi, int Wi) {
  const int64_t total = (int64_t)B * Hi * Wi * K;
  const float rx = (float)Wi / (float)Wo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % K);
    int64_t t = i / K;
    const int ix = (int)(t % Wi); t /= Wi;
    const int iy = (int)(t % Hi);
    const int b = (int)(t / Hi);
    int lo, hi;
    cand_range(ix, rx, Wo, lo, hi);
    const float* row = tmp + (((int64_t)b * K + k) * Hi + iy) * Wo;
    float acc = 0.f;
    for (int ox = lo; ox <= hi; ++ox) {
      int x0, x1; float lx;
      src_index2(rx, ox, Wi, x0, x1, lx);
      const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
      if (wx != 0.f) acc += wx * row[ox];
    }
    din[i] = acc;
  }
}

// softmax(dim=1).argmax(dim=1): first index of the maximal f32 softmax value (torch semantics)
template <int K>
__global__ __launch_bounds__(256) void softmax_argmax_kernel(const float* __restrict__ logits, int B, int64_t HW,
                                                             int64_t* __restrict__ mask) {
  const int64_t total = (int64_t)B * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / HW, p = i - b * HW;
    float x[K], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) { x[k] = logits[(b * K + k) * HW + p]; mx = fmaxf(mx, x[k]); }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { x[k] = expf(x[k] - mx); s += x[k]; }
    int best = 0; float bv = x[0] / s;
#pragma unroll
    for (int k = 1; k < K; ++k) { const float v = x[k] / s; if (v > bv) { bv = v; best = k; } }
    mask[i] = best;
  }
}

// class probabilities of the exported inference model: softmax over the class dim (K > 1) or sigmoid (K == 1)
template <int K>
__global__ __launch_bounds__(256) void class_probs_kernel(const float* __restrict__ logits, int B, int64_t HW,
                                                          float* __restrict__ probs) {
  const int64_t total = (int64_t)B * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / HW, p = i - b * HW;
    if constexpr (K == 1) {
      probs[i] = 1.0f / (1.0f + expf(-logits[i]));
    } else {
      float x[K], mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < K; ++k) { x[k] = logits[(b * K + k) * HW + p]; mx = fmaxf(mx, x[k]); }
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) { x[k] = expf(x[k] - mx); s += x[k]; }
#pragma unroll
      for (int k = 0; k < K; ++k) probs[(b * K + k) * HW + p] = x[k] / s;
    }
  }
}

// ------------------------------------------------------------------ Dice loss (smp multiclass)
// pass 1: per-block partial sums of I_c = sum p_c*[y==c], S_c = sum p_c, N_c = count(y==c)
template <int K>
__global__ __launch_bounds__(256) void dice_partial_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                           int B, int64_t HW, float* __restrict__ ws) {
  __shared__ float red[4][3 * K];
  const int64_t total = (int64_t)B * HW;
  float I[K], S[K], Nc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) I[k] = S[k] = Nc[k] = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / HW, p = i - b * HW;
    float x[K], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) { x[k] = logits[(b * K + k) * HW + p]; mx = fmaxf(mx, x[k]); }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { x[k] = expf(x[k] - mx); s += x[k]; }
    const float inv = 1.f / s;
    const int y = (int)target[i];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float pk = x[k] * inv;
      S[k] += pk;
      if (y == k) { I[k] += pk; Nc[k] += 1.f; }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float a = wave_sum(I[k]), bsum = wave_sum(S[k]), c = wave_sum(Nc[k]);
    if (lane == 0) { red[wv][k] = a; red[wv][K + k] = bsum; red[wv][2 * K + k] = c; }
  }
  __syncthreads();
  if (threadIdx.x < 3 * K)
    ws[(int64_t)blockIdx.x * 3 * K + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

template <int K>
__global__ __launch_bounds__(256) void dice_final_kernel(const float* __restrict__ ws, int nblk, float eps,
                                                         float* __restrict__ sums, float* __restrict__ loss) {
  __shared__ double part[4][64];
  __shared__ double tot[64];
  const int t = threadIdx.x, v = t & 63, grp = t >> 6;   // 4 groups of 64 value slots (3K <= 48)
  static_assert(3 * K <= 64, "dice_final_kernel: at most 21 classes");
  double s = 0;
  if (v < 3 * K) s = ordered_sum8<double>(grp, nblk, 4, [&](int i) { return ws[(int64_t)i * 3 * K + v]; });   // same order, 8 loads in flight
  part[grp][v] = s;
  __syncthreads();
  if (t < 3 * K) {
    double r = 0;
    for (int g = 0; g < 4; ++g) r += part[g][t];
    tot[t] = r;
    sums[t] = (float)r;
  }
  __syncthreads();
  if (t == 0) {
    double l = 0;
    for (int k = 0; k < K; ++k) {
      const double I = tot[k], card = tot[K + k] + tot[2 * K + k];
      const double dice = 2.0 * I / (card > eps ? card : eps);
      if (tot[2 * K + k] > 0) l += 1.0 - dice;
    }
    loss[0] = (float)(l / K);
  }
}

// pass 2: dL/dlogits; sums = [I | S | N] as produced above
template <int K>
__global__ __launch_bounds__(256) void dice_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                       int B, int64_t HW, const float* __restrict__ sums, float eps,
                                                       const float* __restrict__ upstream, float grad_scale,
                                                       float* __restrict__ dlogits, int accumulate) {
  float ca[K], cb[K];  // dL/dp_c = ca[c]*[y==c] + cb[c]
  const float up = (upstream ? upstream[0] : 1.f) * grad_scale;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float I = sums[k], card = sums[K + k] + sums[2 * K + k];
    const bool on = sums[2 * K + k] > 0.f && card > eps;
    ca[k] = on ? -2.f / (K * card) * up : 0.f;
    cb[k] = on ? 2.f * I / (K * card * card) * up : 0.f;
  }
  const int64_t total = (int64_t)B * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / HW, p = i - b * HW;
    float x[K], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) { x[k] = logits[(b * K + k) * HW + p]; mx = fmaxf(mx, x[k]); }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { x[k] = expf(x[k] - mx); s += x[k]; }
    const float inv = 1.f / s;
    const int y = (int)target[i];
    float g[K], dot = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      x[k] *= inv;
      g[k] = cb[k] + (y == k ? ca[k] : 0.f);
      dot += x[k] * g[k];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int64_t o = (b * K + k) * HW + p;
      const float v = x[k] * (g[k] - dot);
      dlogits[o] = accumulate ? dlogits[o] + v : v;
    }
  }
}

// ------------------------------------------------------------------ Dice loss straight from the LOW-resolution logits (round 5)
// The training step only needs the loss and its gradient, not the [B, K, 512, 512] f32 logits: per head the tail was
// upsample (write 168 MB) -> dice partial (read 235 MB) -> dice backward (read 235, write 168) -> transposed upsample in two passes
// (read 168 ...), 350 us per head and two heads per step (dofa.py:89-105, segmentation_dofa.py:226-229).  Both kernels below
// evaluate the bilinear logit of a full-resolution pixel on the fly from the [B, Hi, Wi, K] f32 map the 1x1 head wrote (13 MB at
// batch 32: L2 / MALL resident), with the SAME expression as upsample_logits_kernel:
//   forward  -- the three per-class sums of dice_partial_kernel, same workgroup count and pixel order (the same partial sums);
//   backward -- one thread per LOW-resolution logit vector gathers wy * wx * dL/dlogit over the full-resolution pixels that
//               interpolate from it (softmax and Dice coefficients recomputed there): d(low) in one pass, f32, fixed order.
template <int K>
__device__ __forceinline__ void bilinear_logits(const float* __restrict__ in, int b, int Hi, int Wi, int y0, int y1, int x0, int x1,
                                                float ly, float lx, float (&x)[K]) {
  const float* p00 = in + (((int64_t)b * Hi + y0) * Wi + x0) * K;
  const float* p01 = in + (((int64_t)b * Hi + y0) * Wi + x1) * K;
  const float* p10 = in + (((int64_t)b * Hi + y1) * Wi + x0) * K;
  const float* p11 = in + (((int64_t)b * Hi + y1) * Wi + x1) * K;
  const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
  for (int k = 0; k < K; ++k) x[k] = hy * (hx * p00[k] + lx * p01[k]) + ly * (hx * p10[k] + lx * p11[k]);
}

// softmax(dim=1).argmax(dim=1) of the resized logits straight from the head's low-resolution map (validation / test / inference:
// `outputs.out.softmax(dim=1).argmax(dim=1)`, segmentation_dofa.py:278-281): the bilinear logits of a pixel with the expression of
// upsample_logits_kernel, then softmax_argmax_kernel's decision -- the same mask, bit for bit, without the [B, K, H, W] f32 tensor
// (335 MB written and read again at batch 64).
template <int K>
__global__ __launch_bounds__(256) void upsample_argmax_kernel(const float* __restrict__ low, int B, int Hi, int Wi, int Ho, int Wo,
                                                              int64_t* __restrict__ mask) {
  const int64_t total = (int64_t)B * Ho * Wo;
  const float ry = (float)Hi / (float)Ho, rx = (float)Wi / (float)Wo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % Wo);
    const int64_t t = i / Wo;
    const int oy = (int)(t % Ho), b = (int)(t / Ho);
    int y0, y1, x0, x1; float ly, lx;
    src_index2(ry, oy, Hi, y0, y1, ly);
    src_index2(rx, ox, Wi, x0, x1, lx);
    float x[K], mx = -INFINITY;
    bilinear_logits<K>(low, b, Hi, Wi, y0, y1, x0, x1, ly, lx, x);
#pragma unroll
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, x[k]);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { x[k] = expf(x[k] - mx); s += x[k]; }
    int best = 0; float bv = x[0] / s;
#pragma unroll
    for (int k = 1; k < K; ++k) { const float v = x[k] / s; if (v > bv) { bv = v; best = k; } }
    mask[i] = best;
  }
}

template <int K>
__global__ __launch_bounds__(256) void dice_lowres_partial_kernel(const float* __restrict__ low, const int64_t* __restrict__ target,
                                                                  int B, int Hi, int Wi, int Ho, int Wo, float* __restrict__ ws) {
  __shared__ float red[4][3 * K];
  const int64_t total = (int64_t)B * Ho * Wo;
  const float ry = (float)Hi / (float)Ho, rx = (float)Wi / (float)Wo;
  float I[K], S[K], Nc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) I[k] = S[k] = Nc[k] = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % Wo);
    const int64_t t = i / Wo;
    const int oy = (int)(t % Ho), b = (int)(t / Ho);
    int y0, y1, x0, x1; float ly, lx;
    src_index2(ry, oy, Hi, y0, y1, ly);
    src_index2(rx, ox, Wi, x0, x1, lx);
    float x[K], mx = -INFINITY;
    bilinear_logits<K>(low, b, Hi, Wi, y0, y1, x0, x1, ly, lx, x);
#pragma unroll
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, x[k]);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { x[k] = expf(x[k] - mx); s += x[k]; }
    const float inv = 1.f / s;
    const int y = (int)target[i];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float pk = x[k] * inv;
      S[k] += pk;
      if (y == k) { I[k] += pk; Nc[k] += 1.f; }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float a = wave_sum(I[k]), bsum = wave_sum(S[k]), c = wave_sum(Nc[k]);
    if (lane == 0) { red[wv][k] = a; red[wv][K + k] = bsum; red[wv][2 * K + k] = c; }
  }
  __syncthreads();
  if (threadIdx.x < 3 * K)
    ws[(int64_t)blockIdx.x * 3 * K + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

constexpr int DICE_LOWRES_MAX_FACTOR = 64;   // (every loop below is a run-time loop; the bound only keeps the K > 8 gather kernel's
                                             // window -- (2 * factor + 4)^2 softmax evaluations per low-resolution logit -- finite)

template <int K>
__global__ __launch_bounds__(256) void dice_lowres_bwd_kernel(const float* __restrict__ low, const int64_t* __restrict__ target, int B,
                                                              int Hi, int Wi, int Ho, int Wo, const float* __restrict__ sums,
                                                              float eps, const float* __restrict__ upstream, float grad_scale,
                                                              float* __restrict__ dlow) {
  float ca[K], cb[K];  // dL/dp_c = ca[c]*[y==c] + cb[c]   (dice_bwd_kernel)
  const float up = (upstream ? upstream[0] : 1.f) * grad_scale;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float I = sums[k], card = sums[K + k] + sums[2 * K + k];
    const bool on = sums[2 * K + k] > 0.f && card > eps;
    ca[k] = on ? -2.f / (K * card) * up : 0.f;
    cb[k] = on ? 2.f * I / (K * card * card) * up : 0.f;
  }
  const int64_t total = (int64_t)B * Hi * Wi;
  const float ry = (float)Hi / (float)Ho, rx = (float)Wi / (float)Wo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ix = (int)(i % Wi);
    const int64_t t = i / Wi;
    const int iy = (int)(t % Hi), b = (int)(t / Hi);
    int ylo, yhi, xlo, xhi;
    cand_range(iy, ry, Ho, ylo, yhi);
    cand_range(ix, rx, Wo, xlo, xhi);
    float acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
      int y0, y1; float ly;
      src_index2(ry, oy, Hi, y0, y1, ly);
      const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
      if (wy == 0.f) continue;
      const int64_t trow = ((int64_t)b * Ho + oy) * Wo;
      // (runtime loop, indices recomputed per column: holding the window's columns in registers -- 4 x 16 values -- and unrolling
      // cost 208 registers = two waves per SIMD for a kernel that lives on L1 / L2 latency)
#pragma unroll 1
      for (int ox = xlo; ox <= xhi; ++ox) {
        int x0, x1; float lx;
        src_index2(rx, ox, Wi, x0, x1, lx);
        const float w = wy * ((x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f));
        if (w == 0.f) continue;
        float x[K], mx = -INFINITY;
        bilinear_logits<K>(low, b, Hi, Wi, y0, y1, x0, x1, ly, lx, x);
#pragma unroll
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, x[k]);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) { x[k] = expf(x[k] - mx); s += x[k]; }
        const float inv = 1.f / s;
        const int y = (int)target[trow + ox];
        float g[K], dot = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          x[k] *= inv;
          g[k] = cb[k] + (y == k ? ca[k] : 0.f);
          dot += x[k] * g[k];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] += w * (x[k] * (g[k] - dot));
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) dlow[i * K + k] = acc[k];
  }
}

// The same gradient with every full-resolution pixel's softmax evaluated ONCE (the gather kernel above evaluates it once per
// low-resolution logit that interpolates into it: four times, 272 us at batch 32 -- no faster than the three launches it replaced).
// A workgroup owns a DT_H x DT_W tile of full-resolution pixels:
//   1. dL/dlogit of its pixels -> LDS (K x 2048 floats);
//   2. transposed bilinear, rows: tmp[k][iy][c] = sum over the tile's rows of wy(row -> iy) dl[k][row][c] for the low-resolution rows
//      the tile touches (LDS);
//   3. columns: part[k][iy][ix] = sum_c wx(c -> ix) tmp[k][iy][c] -> the tile's partial patch in the workspace.
// dice_lowres_bwd_reduce_kernel then adds, per low-resolution logit, the patches of the (at most four) tiles that touch it, in a
// fixed order.  No atomics, f32, deterministic.  K <= 8 (LDS); more classes take the gather kernel.
constexpr int DT_H = 32, DT_W = 64, DT_MAXN = 36;     // tile; bound on the low-resolution rows / columns one tile side can touch
constexpr int DT_T = 1024;    // threads per tile: two pixels each.  With 256 (eight pixels each, LDS allowing two workgroups per CU =
                              // two waves per SIMD) phase 1 was a chain of eight L2 round trips per thread: 215 us at batch 32

struct DiceTile {
  const float* low; const int64_t* target; const float* sums; const float* upstream; float* ws; float* dlow;
  int B, Hi, Wi, Ho, Wo, tiles_y, tiles_x, ny_max, nx_max;
  float eps, grad_scale;
};

// low-resolution index range [lo, hi] that the full-resolution positions [p0, p1] interpolate from
__device__ __forceinline__ void touched_range(float ratio, int p0, int p1, int in_size, int& lo, int& hi) {
  int a0, a1, b0, b1; float l;
  src_index2(ratio, p0, in_size, a0, a1, l);
  src_index2(ratio, p1, in_size, b0, b1, l);
  lo = a0; hi = b1;
}

template <int K>
__global__ __launch_bounds__(DT_T) void dice_lowres_bwd_tile_kernel(const DiceTile a) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  float* dl = dsm;                                   // [K][DT_H][DT_W]
  float* tmp = dsm + K * DT_H * DT_W;                // [K][ny_max][DT_W + 1]  (+1: phase 3's threads differ in j at equal c)
  float* wyt = tmp + K * a.ny_max * (DT_W + 1);      // [ny_max][DT_H]  weight of tile row r for low-resolution row iy_lo + j
  float* wxt = wyt + a.ny_max * DT_H;                // [nx_max][DT_W]  the same for columns
  const int tid = threadIdx.x;
  const int tx = blockIdx.x % a.tiles_x, ty = (blockIdx.x / a.tiles_x) % a.tiles_y, b = blockIdx.x / (a.tiles_x * a.tiles_y);
  const int oy0 = ty * DT_H, ox0 = tx * DT_W;
  const int rows = a.Ho - oy0 < DT_H ? a.Ho - oy0 : DT_H, cols = a.Wo - ox0 < DT_W ? a.Wo - ox0 : DT_W;
  const float ry = (float)a.Hi / (float)a.Ho, rx = (float)a.Wi / (float)a.Wo;
  float ca[K], cb[K];
  const float up = (a.upstream ? a.upstream[0] : 1.f) * a.grad_scale;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float I = a.sums[k], card = a.sums[K + k] + a.sums[2 * K + k];
    const bool on = a.sums[2 * K + k] > 0.f && card > a.eps;
    ca[k] = on ? -2.f / (K * card) * up : 0.f;
    cb[k] = on ? 2.f * I / (K * card * card) * up : 0.f;
  }
  // ---- 1. dL/dlogit of the tile (zeros outside the image)
  for (int i = tid; i < DT_H * DT_W; i += DT_T) {
    const int r = i / DT_W, c = i - r * DT_W;
    float v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = 0.f;
    if (r < rows && c < cols) {
      const int oy = oy0 + r, ox = ox0 + c;
      int y0, y1, x0, x1; float ly, lx;
      src_index2(ry, oy, a.Hi, y0, y1, ly);
      src_index2(rx, ox, a.Wi, x0, x1, lx);
      float x[K], mx = -INFINITY;
      bilinear_logits<K>(a.low, b, a.Hi, a.Wi, y0, y1, x0, x1, ly, lx, x);
#pragma unroll
      for (int k = 0; k < K; ++k) mx = fmaxf(mx, x[k]);
      float sden = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) { x[k] = expf(x[k] - mx); sden += x[k]; }
      const float inv = 1.f / sden;
      const int y = (int)a.target[((int64_t)b * a.Ho + oy) * a.Wo + ox];
      float g[K], dot = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        x[k] *= inv;
        g[k] = cb[k] + (y == k ? ca[k] : 0.f);
        dot += x[k] * g[k];
      }
#pragma unroll
      for (int k = 0; k < K; ++k) v[k] = x[k] * (g[k] - dot);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) dl[(k * DT_H + r) * DT_W + c] = v[k];
  }
  // the low-resolution rows iy_lo .. iy_hi / columns ix_lo .. ix_hi this tile touches, and the two 1-D weight tables
  int iy_lo, iy_hi, ix_lo, ix_hi;
  touched_range(ry, oy0, oy0 + rows - 1, a.Hi, iy_lo, iy_hi);
  touched_range(rx, ox0, ox0 + cols - 1, a.Wi, ix_lo, ix_hi);
  const int ny = iy_hi - iy_lo + 1, nx = ix_hi - ix_lo + 1;
  for (int i = tid; i < ny * DT_H; i += DT_T) {
    const int j = i / DT_H, r = i - j * DT_H;
    float wv = 0.f;
    if (r < rows) {
      int y0, y1; float ly;
      src_index2(ry, oy0 + r, a.Hi, y0, y1, ly);
      wv = (y0 == iy_lo + j ? 1.f - ly : 0.f) + (y1 == iy_lo + j ? ly : 0.f);
    }
    wyt[i] = wv;
  }
  for (int i = tid; i < nx * DT_W; i += DT_T) {
    const int q = i / DT_W, c = i - q * DT_W;
    float wv = 0.f;
    if (c < cols) {
      int x0, x1; float lx;
      src_index2(rx, ox0 + c, a.Wi, x0, x1, lx);
      wv = (x0 == ix_lo + q ? 1.f - lx : 0.f) + (x1 == ix_lo + q ? lx : 0.f);
    }
    wxt[i] = wv;
  }
  __syncthreads();
  // ---- 2. rows
  for (int i = tid; i < ny * DT_W; i += DT_T) {
    const int j = i / DT_W, c = i - j * DT_W;
    float acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.f;
    // (only the tile rows that can interpolate from low-resolution row iy_lo + j: 2 / ratio + 3 of the 32 -- same terms, same order)
    int r_lo, r_hi;
    cand_range(iy_lo + j, ry, a.Ho, r_lo, r_hi);
    r_lo = r_lo - oy0 < 0 ? 0 : r_lo - oy0;
    r_hi = r_hi - oy0 > rows - 1 ? rows - 1 : r_hi - oy0;
    for (int r = r_lo; r <= r_hi; ++r) {
      const float wy = wyt[j * DT_H + r];
      if (wy != 0.f) {
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] += wy * dl[(k * DT_H + r) * DT_W + c];
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) tmp[(k * a.ny_max + j) * (DT_W + 1) + c] = acc[k];
  }
  __syncthreads();
  // ---- 3. columns -> the tile's partial patch [ny_max][nx_max][K] in the workspace (entries beyond ny / nx are never read)
  float* patch = a.ws + (int64_t)blockIdx.x * a.ny_max * a.nx_max * K;
  for (int i = tid; i < ny * nx; i += DT_T) {
    const int j = i / nx, q = i - j * nx;
    float acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.f;
    int c_lo, c_hi;
    cand_range(ix_lo + q, rx, a.Wo, c_lo, c_hi);
    c_lo = c_lo - ox0 < 0 ? 0 : c_lo - ox0;
    c_hi = c_hi - ox0 > cols - 1 ? cols - 1 : c_hi - ox0;
    for (int c = c_lo; c <= c_hi; ++c) {
      const float wx = wxt[q * DT_W + c];
      if (wx != 0.f) {
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] += wx * tmp[(k * a.ny_max + j) * (DT_W + 1) + c];
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) patch[(j * a.nx_max + q) * K + k] = acc[k];
  }
}

template <int K>
__global__ __launch_bounds__(256) void dice_lowres_bwd_reduce_kernel(const DiceTile a) {
  const int64_t total = (int64_t)a.B * a.Hi * a.Wi;
  const float ry = (float)a.Hi / (float)a.Ho, rx = (float)a.Wi / (float)a.Wo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ix = (int)(i % a.Wi);
    const int64_t t = i / a.Wi;
    const int iy = (int)(t % a.Hi), b = (int)(t / a.Hi);
    int ylo, yhi, xlo, xhi;
    cand_range(iy, ry, a.Ho, ylo, yhi);              // full-resolution rows / columns that can interpolate from (iy, ix)
    cand_range(ix, rx, a.Wo, xlo, xhi);
    float acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.f;
    for (int ty = ylo / DT_H; ty <= yhi / DT_H; ++ty) {
      const int oy0 = ty * DT_H, rows = a.Ho - oy0 < DT_H ? a.Ho - oy0 : DT_H;
      int iy_lo, iy_hi;
      touched_range(ry, oy0, oy0 + rows - 1, a.Hi, iy_lo, iy_hi);
      if (iy < iy_lo || iy > iy_hi) continue;
      for (int tx = xlo / DT_W; tx <= xhi / DT_W; ++tx) {
        const int ox0 = tx * DT_W, cols = a.Wo - ox0 < DT_W ? a.Wo - ox0 : DT_W;
        int ix_lo, ix_hi;
        touched_range(rx, ox0, ox0 + cols - 1, a.Wi, ix_lo, ix_hi);
        if (ix < ix_lo || ix > ix_hi) continue;
        const float* patch = a.ws + ((int64_t)(b * a.tiles_y + ty) * a.tiles_x + tx) * a.ny_max * a.nx_max * K;
        const float* src = patch + ((iy - iy_lo) * a.nx_max + (ix - ix_lo)) * K;
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] += src[k];
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) a.dlow[i * K + k] = acc[k];
  }
}

// ------------------------------------------------------------------ Dice loss (smp binary)
// smp DiceLoss(mode="binary") (configs/unetplus_config_RGB.yaml: num_classes 1): p = exp(logsigmoid(x)), one class,
// sums over dims (batch, pixels); the target is used as a 0/1 weight.  Partials have the multiclass layout with K = 1
// ([I | S | N]) so dice_final_kernel<1> finishes them (loss * [sum y > 0], mean over the single class).
__global__ __launch_bounds__(256) void dice_binary_partial_kernel(const float* __restrict__ logits,
                                                                  const int64_t* __restrict__ target, int64_t total,
                                                                  float* __restrict__ ws) {
  __shared__ float red[4][3];
  float I = 0.f, S = 0.f, Nc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float x = logits[i];
    // exp(logsigmoid(x)) with logsigmoid(x) = min(x, 0) - log1p(exp(-|x|)), as torch computes it
    const float p = expf(fminf(x, 0.f) - log1pf(expf(-fabsf(x))));
    const float y = (float)target[i];
    I += p * y; S += p; Nc += y;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float a = wave_sum(I), b = wave_sum(S), c = wave_sum(Nc);
  if (lane == 0) { red[wv][0] = a; red[wv][1] = b; red[wv][2] = c; }
  __syncthreads();
  if (threadIdx.x < 3)
    ws[(int64_t)blockIdx.x * 3 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void dice_binary_bwd_kernel(const float* __restrict__ logits,
                                                              const int64_t* __restrict__ target, int64_t total,
                                                              const float* __restrict__ sums, float eps,
                                                              const float* __restrict__ upstream, float grad_scale,
                                                              float* __restrict__ dlogits, int accumulate) {
  const float up = (upstream ? upstream[0] : 1.f) * grad_scale;
  const float I = sums[0], card = sums[1] + sums[2];
  const bool on = sums[2] > 0.f && card > eps;
  const float ca = on ? -2.f / card * up : 0.f;              // dL/dp = ca * y + cb
  const float cb = on ? 2.f * I / (card * card) * up : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float x = logits[i];
    const float p = expf(fminf(x, 0.f) - log1pf(expf(-fabsf(x))));
    const float v = (cb + ca * (float)target[i]) * p * (1.f - p);
    dlogits[i] = accumulate ? dlogits[i] + v : v;
  }
}

// ------------------------------------------------------------------ optimizer
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += x[i] * x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}

__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float max_norm, float* __restrict__ coef) {
  const float norm = sqrtf(sumsq[0]);
  const float c = max_norm / (norm + 1e-6f);   // torch.nn.utils.clip_grad_norm_
  coef[0] = c < 1.f ? c : 1.f;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float lr, float b1, float b2,
                                                   float eps, float wd, float bc1, float bc2,
                                                   const float* __restrict__ clip_coef) {
  const float cc = clip_coef ? clip_coef[0] : 1.f;
  const float step = lr / bc1, rs = 1.f / sqrtf(bc2);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float gi = g[i] * cc;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = pi - step * mi / (sqrtf(vi) * rs + eps);
  }
}

// x[o, r, :] *= s[o]   (DropPath: o = sample; Dropout2d on NHWC handled via chan_scale elsewhere)
template <typename T>
__global__ __launch_bounds__(256) void scale_outer_kernel(void* x, const float* __restrict__ s, int64_t outer, int64_t inner) {
  const int64_t total = outer * inner;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    ElemIO<T>::store(x, i, ElemIO<T>::load(x, i) * s[i / inner]);
}

// position_embedding (dofa_v2.py:9-35): out[m, :D/2] = sin(pos*omega), out[m, D/2:] = cos(pos*omega).
// The frequency table omega[D/2] (a constant of the module) comes from the host so that the only
// difference to the reference is the sin/cos implementation (angles reach ~2000 rad).
__global__ void sincos_embed_kernel(const float* __restrict__ pos, const float* __restrict__ omega_tab,
                                    int M, int D, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = D / 2;
  if (i >= M * half) return;
  const int m = i / half, d = i - m * half;
  const float v = pos[m] * omega_tab[d];
  out[(int64_t)m * D + d] = sinf(v);
  out[(int64_t)m * D + half + d] = cosf(v);
}

// eval-mode BatchNorm folded into the conv epilogue: scale = g*rsqrt(var+eps), shift = b - mean*scale
__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mean, const float* __restrict__ var, float eps, int C,
                               float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float s = gamma[c] / sqrtf(var[c] + eps);
  scale[c] = s;
  shift[c] = beta[c] - mean[c] * s;
}

// forward weights [N][T][C] (T = R*S taps) -> data-gradient weights [C][T][N] with the taps flipped
// (t' = T-1-t): dx = conv(dy, w_dgrad, pad = R-1-pad).  32x32 LDS tile transpose per tap.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void pack_dgrad_kernel(const void* __restrict__ w, int N, int T, int Cc,
                                                         void* out) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z, n0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int n = n0 + j, c = c0 + tx;
    tile[j][tx] = (n < N && c < Cc) ? ElemIO<TI>::load(w, ((int64_t)n * T + t) * Cc + c) : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, n = n0 + tx;
    if (c < Cc && n < N) ElemIO<TO>::store(out, ((int64_t)c * T + (T - 1 - t)) * N + n, tile[tx][j]);
  }
}

}  // namespace

// ============================================================================ C ABI
extern "C" int gdl_sincos_embed(const float* pos, const float* omega, int M, int D, float* out, gdl_stream_t stream) {
  GDL_CHECK_ARG(pos && omega && out && D % 2 == 0 && M > 0, "gdl_sincos_embed: bad args");
  const int total = M * (D / 2);
  hipLaunchKernelGGL(sincos_embed_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, pos, omega, M, D, out);
  GDL_CHECK_LAUNCH("gdl_sincos_embed");
  return GDL_OK;
}

extern "C" int gdl_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                           int C, float* scale, float* shift, gdl_stream_t stream) {
  GDL_CHECK_ARG(gamma && beta && mean && var && scale && shift, "gdl_bn_fold: null pointer");
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, var, eps, C, scale, shift);
  GDL_CHECK_LAUNCH("gdl_bn_fold");
  return GDL_OK;
}

extern "C" int gdl_pack_dgrad(const void* w, int w_dtype, int N, int T, int C, void* out, int out_dtype,
                              gdl_stream_t stream) {
  GDL_CHECK_ARG(w && out && N > 0 && T > 0 && C > 0, "gdl_pack_dgrad: bad args");
  dim3 grid((C + 31) / 32, (N + 31) / 32, T);
  hipStream_t s = (hipStream_t)stream;
  if (w_dtype == GDL_F32 && out_dtype == GDL_BF16) hipLaunchKernelGGL((pack_dgrad_kernel<float, bf16_tag>), grid, dim3(256), 0, s, w, N, T, C, out);
  else if (w_dtype == GDL_F32) hipLaunchKernelGGL((pack_dgrad_kernel<float, float>), grid, dim3(256), 0, s, w, N, T, C, out);
  else if (out_dtype == GDL_BF16) hipLaunchKernelGGL((pack_dgrad_kernel<bf16_tag, bf16_tag>), grid, dim3(256), 0, s, w, N, T, C, out);
  else hipLaunchKernelGGL((pack_dgrad_kernel<bf16_tag, float>), grid, dim3(256), 0, s, w, N, T, C, out);
  GDL_CHECK_LAUNCH("gdl_pack_dgrad");
  return GDL_OK;
}

extern "C" int gdl_v_transpose(const void* v, int dtype, int B, int N, int H, int hd, int64_t v_sB, int64_t v_sN,
                               void* vt, int Npad, gdl_stream_t stream) {
  GDL_CHECK_ARG(v && vt && Npad >= N, "gdl_v_transpose: bad args");
  dim3 grid((Npad + 31) / 32, (hd + 31) / 32, B * H);
  if (dtype == GDL_BF16)
    hipLaunchKernelGGL(v_transpose_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)v, N, H, hd, v_sB, v_sN, (uint16_t*)vt, Npad);
  else
    hipLaunchKernelGGL(v_transpose_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)v, N, H, hd, v_sB, v_sN, (float*)vt, Npad);
  GDL_CHECK_LAUNCH("gdl_v_transpose");
  return GDL_OK;
}

extern "C" int gdl_softmax_rows(const void* in, void* out, int dtype, int64_t rows, int n_valid, int n_cols,
                                gdl_stream_t stream) {
  GDL_CHECK_ARG(in && out && n_valid > 0 && n_valid <= n_cols, "gdl_softmax_rows: bad args");
  GDL_CHECK_ARG(n_cols <= 64 * 96, "gdl_softmax_rows: n_cols=%d too large (max 6144)", n_cols);
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((rows + 3) / 4);
#define SM(T, V) hipLaunchKernelGGL((softmax_rows_kernel<T, V>), dim3(grid), dim3(256), 0, s, in, out, rows, n_valid, n_cols)
  const int v = (n_cols + 63) / 64;
  if (dtype == GDL_BF16) {
    if (v <= 4) SM(bf16_tag, 4); else if (v <= 24) SM(bf16_tag, 24); else SM(bf16_tag, 96);
  } else {
    if (v <= 4) SM(float, 4); else if (v <= 24) SM(float, 24); else SM(float, 96);
  }
#undef SM
  GDL_CHECK_LAUNCH("gdl_softmax_rows");
  return GDL_OK;
}

extern "C" int gdl_patchify(const float* img, int B, int C, int H, int W, int P, int stride, int pad, int Gh, int Gw,
                            void* cols, int out_dtype, int Kpad, gdl_stream_t stream) {
  GDL_CHECK_ARG(img && cols && Kpad >= C * P * P, "gdl_patchify: bad args");
  const int64_t total = (int64_t)B * Gh * Gw * Kpad;
  if (Kpad % 4 == 0 && (uintptr_t)cols % 16 == 0) {
    if (out_dtype == GDL_BF16)
      hipLaunchKernelGGL(patchify4_kernel<bf16_tag>, dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream, img, B, C, H, W, P, stride, pad, Gh, Gw, cols, Kpad);
    else
      hipLaunchKernelGGL(patchify4_kernel<float>, dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream, img, B, C, H, W, P, stride, pad, Gh, Gw, cols, Kpad);
  } else if (out_dtype == GDL_BF16)
    hipLaunchKernelGGL(patchify_kernel<bf16_tag>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, img, B, C, H, W, P, stride, pad, Gh, Gw, cols, Kpad);
  else
    hipLaunchKernelGGL(patchify_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, img, B, C, H, W, P, stride, pad, Gh, Gw, cols, Kpad);
  GDL_CHECK_LAUNCH("gdl_patchify");
  return GDL_OK;
}

extern "C" int gdl_dofa_pack_kernel(const float* g, int C, int PP, int D, float scaler, void* out, int out_dtype, int Kpad,
                                    gdl_stream_t stream) {
  GDL_CHECK_ARG(g && out && Kpad >= C * PP, "gdl_dofa_pack_kernel: bad args");
  const int64_t total = (int64_t)D * Kpad;
  if (out_dtype == GDL_BF16)
    hipLaunchKernelGGL(dofa_pack_kernel<bf16_tag>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, g, C, PP, D, scaler, out, Kpad);
  else
    hipLaunchKernelGGL(dofa_pack_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, g, C, PP, D, scaler, out, Kpad);
  GDL_CHECK_LAUNCH("gdl_dofa_pack_kernel");
  return GDL_OK;
}

extern "C" int gdl_dofa_unpack_grad(const float* dw, int C, int PP, int D, float scaler, int Kpad, float* dg,
                                    gdl_stream_t stream) {
  GDL_CHECK_ARG(dw && dg && Kpad >= C * PP, "gdl_dofa_unpack_grad: bad args");
  const int64_t total = (int64_t)C * PP * D;
  hipLaunchKernelGGL(dofa_unpack_grad_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dw, C * PP, D, scaler, Kpad, dg);
  GDL_CHECK_LAUNCH("gdl_dofa_unpack_grad");
  return GDL_OK;
}

extern "C" int gdl_cast(const void* in, int in_dtype, void* out, int out_dtype, int64_t n, gdl_stream_t stream) {
  GDL_CHECK_ARG(in && out, "gdl_cast: null pointer");
  if (n <= 0) return GDL_OK;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(grid_for(n));
  if (in_dtype == GDL_F32 && out_dtype == GDL_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_tag>), grid, dim3(256), 0, s, in, out, n);
  else if (in_dtype == GDL_BF16 && out_dtype == GDL_F32) hipLaunchKernelGGL((cast_kernel<bf16_tag, float>), grid, dim3(256), 0, s, in, out, n);
  else if (in_dtype == GDL_F32) hipLaunchKernelGGL((cast_kernel<float, float>), grid, dim3(256), 0, s, in, out, n);
  else hipLaunchKernelGGL((cast_kernel<bf16_tag, bf16_tag>), grid, dim3(256), 0, s, in, out, n);
  GDL_CHECK_LAUNCH("gdl_cast");
  return GDL_OK;
}

extern "C" int gdl_scale_f32(const float* in, float* out, int64_t n, float s, gdl_stream_t stream) {
  GDL_CHECK_ARG(in && out, "gdl_scale_f32: null pointer");
  if (n <= 0) return GDL_OK;
  hipLaunchKernelGGL(scale_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, in, out, n, s);
  GDL_CHECK_LAUNCH("gdl_scale_f32");
  return GDL_OK;
}

extern "C" int gdl_add_rows(const float* a, int64_t a_rows, int64_t a_stride, const float* b, int64_t b_rows,
                            int64_t b_stride, float* out, int64_t out_stride, int64_t rows, int D,
                            gdl_stream_t stream) {
  GDL_CHECK_ARG(a && out && a_rows > 0 && (!b || b_rows > 0), "gdl_add_rows: bad args");
  hipLaunchKernelGGL(add_rows_kernel, dim3(grid_for(rows * D)), dim3(256), 0, (hipStream_t)stream, a, a_rows, a_stride, b,
                     b_rows, b_stride, out, out_stride, rows, D);
  GDL_CHECK_LAUNCH("gdl_add_rows");
  return GDL_OK;
}

extern "C" int gdl_normalize_u8(const uint8_t* in, float* out, int B, int C, int64_t HW, const float* mean,
                                const float* stdv, gdl_stream_t stream) {
  GDL_CHECK_ARG(in && out && mean && stdv, "gdl_normalize_u8: null pointer");
  GDL_CHECK_ARG(HW % 4 == 0, "gdl_normalize_u8: H*W must be a multiple of 4");
  const int64_t total4 = (int64_t)B * C * HW / 4;
  hipLaunchKernelGGL(normalize_u8_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, in, out, C, HW, total4, mean, stdv);
  GDL_CHECK_LAUNCH("gdl_normalize_u8");
  return GDL_OK;
}

extern "C" int gdl_normalize_raw(const void* in, int kind, float* out, int B, int C, int64_t HW, const float* mean,
                                 const float* stdv, gdl_stream_t stream) {
  GDL_CHECK_ARG(in && out && mean && stdv, "gdl_normalize_raw: null pointer");
  GDL_CHECK_ARG(kind >= GDL_RAW_U8 && kind <= GDL_RAW_F32, "gdl_normalize_raw: bad sample kind %d", kind);
  if (kind == GDL_RAW_U8 && HW % 4 == 0 && (uintptr_t)in % 4 == 0)
    return gdl_normalize_u8((const uint8_t*)in, out, B, C, HW, mean, stdv, stream);
  const int64_t total = (int64_t)B * C * HW;
  hipStream_t s = (hipStream_t)stream;
#define NR(T) hipLaunchKernelGGL(normalize_raw_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, (const T*)in, out, C, HW, total, mean, stdv)
  if (kind == GDL_RAW_U8) NR(uint8_t);
  else if (kind == GDL_RAW_U16) NR(uint16_t);
  else if (kind == GDL_RAW_I16) NR(int16_t);
  else NR(float);
#undef NR
  GDL_CHECK_LAUNCH("gdl_normalize_raw");
  return GDL_OK;
}

extern "C" int gdl_scale_outer(void* x, int dtype, const float* s, int64_t outer, int64_t inner, gdl_stream_t stream) {
  GDL_CHECK_ARG(x && s, "gdl_scale_outer: null pointer");
  const int64_t total = outer * inner;
  if (total <= 0) return GDL_OK;
  if (dtype == GDL_BF16) hipLaunchKernelGGL(scale_outer_kernel<bf16_tag>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, s, outer, inner);
  else hipLaunchKernelGGL(scale_outer_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, s, outer, inner);
  GDL_CHECK_LAUNCH("gdl_scale_outer");
  return GDL_OK;
}

#define K_SWITCH(K, ...)                                                     \
  switch (K) {                                                               \
    case 1: { constexpr int KK = 1; __VA_ARGS__; } break;                    \
    case 2: { constexpr int KK = 2; __VA_ARGS__; } break;                    \
    case 3: { constexpr int KK = 3; __VA_ARGS__; } break;                    \
    case 4: { constexpr int KK = 4; __VA_ARGS__; } break;                    \
    case 5: { constexpr int KK = 5; __VA_ARGS__; } break;                    \
    case 6: { constexpr int KK = 6; __VA_ARGS__; } break;                    \
    case 7: { constexpr int KK = 7; __VA_ARGS__; } break;                    \
    case 8: { constexpr int KK = 8; __VA_ARGS__; } break;                    \
    case 9: { constexpr int KK = 9; __VA_ARGS__; } break;                    \
    case 10: { constexpr int KK = 10; __VA_ARGS__; } break;                  \
    case 11: { constexpr int KK = 11; __VA_ARGS__; } break;                  \
    case 12: { constexpr int KK = 12; __VA_ARGS__; } break;                  \
    case 13: { constexpr int KK = 13; __VA_ARGS__; } break;                  \
    case 14: { constexpr int KK = 14; __VA_ARGS__; } break;                  \
    case 15: { constexpr int KK = 15; __VA_ARGS__; } break;                  \
    case 16: { constexpr int KK = 16; __VA_ARGS__; } break;                  \
    default: gdl_set_error("num classes K=%d unsupported (1..16)", K); return GDL_ERR_UNSUPPORTED; \
  }

static int gdl_num_cus() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    return cus;
  }();
  return n;
}

// A/B hook: 0 = the round-4 head kernels (wave per pixel forward, L1-resident weights in the feature gradient) for every shape
static std::atomic<int> g_head_mfma{1};
extern "C" void gdl_debug_set_head_mfma(int on) { g_head_mfma = on; }

extern "C" int gdl_head_1x1(const void* feat, int dtype, int64_t P, int C, int64_t f_sP, const float* w,
                            const float* bias, const float* chan_scale, int64_t pix_per_img, float* out, int K,
                            gdl_stream_t stream) {
  GDL_CHECK_ARG(feat && w && out && C % 4 == 0 && f_sP % 4 == 0 && pix_per_img > 0, "gdl_head_1x1: bad args");
  hipStream_t s = (hipStream_t)stream;
  if (g_head_mfma.load(std::memory_order_relaxed) && dtype == GDL_BF16 && !chan_scale && f_sP == C && (C == 128 || C == 256) && K >= 1 &&
      K <= 16 && P >= 1024 && (uintptr_t)feat % 16 == 0) {
    const int64_t ntiles = (P + 15) / 16;
    const int cap = 3 * gdl_num_cus();     // all workgroups resident at once (155 registers: three waves per SIMD), each wave walks its share
    const unsigned blocks = (unsigned)((ntiles + 3) / 4 < cap ? (ntiles + 3) / 4 : cap);
    if (C == 256) hipLaunchKernelGGL((head_1x1_mfma_kernel<8>), dim3(blocks), dim3(256), 4 * 16 * 512, s, (const uint16_t*)feat, P, w, bias, K, out);
    else hipLaunchKernelGGL((head_1x1_mfma_kernel<4>), dim3(blocks), dim3(256), 4 * 16 * 256, s, (const uint16_t*)feat, P, w, bias, K, out);
    GDL_CHECK_LAUNCH("gdl_head_1x1(mfma)");
    return GDL_OK;
  }
  if (g_head_mfma.load(std::memory_order_relaxed) && dtype == GDL_BF16 && f_sP == C && C % 256 == 0 && C <= 1024 && K >= 1 && K <= 16 &&
      P >= 1024 && (uintptr_t)feat % 16 == 0 && (!chan_scale || pix_per_img % 16 == 0)) {
    // 256 NW channels and / or a Dropout2d scale: NW waves per tile, contiguous tile ranges
    const int NW = C / 256;
    const int64_t ntiles = (P + 15) / 16;
    const int64_t cap = (int64_t)((chan_scale ? 8 : 12) / NW) * gdl_num_cus();      // every workgroup resident: 2 / 3 waves per SIMD
    const unsigned blocks = (unsigned)(ntiles < cap ? ntiles : cap);
    const size_t lds = (size_t)NW * (16 * 512 + 2 * 64 * 16);
    const uint16_t* f = (const uint16_t*)feat;
#define GDL_HEAD_WIDE(NWV, SC) hipLaunchKernelGGL((head_1x1_mfma_wide_kernel<NWV, SC>), dim3(blocks), dim3(64 * NWV), lds, s, f, P, w, bias, chan_scale, pix_per_img, K, out)
    switch (NW * 2 + (chan_scale ? 1 : 0)) {
      case 2: GDL_HEAD_WIDE(1, false); break;
      case 3: GDL_HEAD_WIDE(1, true); break;
      case 4: GDL_HEAD_WIDE(2, false); break;
      case 5: GDL_HEAD_WIDE(2, true); break;
      case 6: GDL_HEAD_WIDE(3, false); break;
      case 7: GDL_HEAD_WIDE(3, true); break;
      case 8: GDL_HEAD_WIDE(4, false); break;
      default: GDL_HEAD_WIDE(4, true); break;
    }
#undef GDL_HEAD_WIDE
    GDL_CHECK_LAUNCH("gdl_head_1x1(mfma, wide)");
    return GDL_OK;
  }
  const unsigned grid = (unsigned)((P + 3) / 4);
  K_SWITCH(K, if (dtype == GDL_BF16) hipLaunchKernelGGL((head_1x1_kernel<uint16_t, KK>), dim3(grid), dim3(256), 0, s, feat, P, C, f_sP, w, bias, chan_scale, pix_per_img, out);
              else hipLaunchKernelGGL((head_1x1_kernel<float, KK>), dim3(grid), dim3(256), 0, s, feat, P, C, f_sP, w, bias, chan_scale, pix_per_img, out));
  GDL_CHECK_LAUNCH("gdl_head_1x1");
  return GDL_OK;
}

// partial rows of the weight gradient: one workgroup each.  Round 5: up to 2048 (eight workgroups per CU) -- with 512 the main head's
// 340 MB were read by two workgroups per CU, 16 KB in flight per CU: 1.5 TB/s
static int head_bwd_nsplit(int64_t P) {
  int64_t ns = P / 256; if (ns < 1) ns = 1; if (ns > 2048) ns = 2048;
  return (int)ns;
}

extern "C" int64_t gdl_head_1x1_bwd_workspace(int64_t P, int C, int K) {
  return (int64_t)head_bwd_nsplit(P) * (K + 1) * (int64_t)C * sizeof(float);
}

extern "C" int gdl_head_1x1_bwd(const void* feat, int dtype, const float* dlog, int64_t P, int C, int64_t f_sP,
                                const float* w, const float* chan_scale, int64_t pix_per_img, void* dfeat,
                                int64_t d_sP, float* dw, float* db, int K, float* ws, int64_t ws_bytes,
                                gdl_stream_t stream) {
  GDL_CHECK_ARG(feat && dlog && w && dw && ws && C % 4 == 0 && f_sP % 4 == 0, "gdl_head_1x1_bwd: bad args");
  GDL_CHECK_ARG(ws_bytes >= gdl_head_1x1_bwd_workspace(P, C, K), "gdl_head_1x1_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int nsplit = head_bwd_nsplit(P);
  dim3 gridw((C + 255) / 256, nsplit);
  const int64_t total = P * (C / 4);
  const bool feat8 = g_head_mfma.load(std::memory_order_relaxed) && dtype == GDL_BF16 && dfeat && !chan_scale && d_sP == C && C % 8 == 0 &&
                     C / 8 <= 256 && 256 % (C / 8) == 0 && K <= 8 && (uintptr_t)dfeat % 16 == 0 && P >= 1024 &&
                     (uintptr_t)w % 16 == 0;      // (the kernel reads the weights with float4 loads: a view at an odd offset takes the general path)
  // general wide forms (a Dropout2d scale, or C / 8 lanes per pixel that only divide 192 threads: SegFormer's 768 channels)
  const int G8 = C % 8 == 0 ? C / 8 : 0;
  const int tpb = G8 == 0 ? 0 : (G8 <= 256 && 256 % G8 == 0) ? 256 : (G8 <= 192 && 192 % G8 == 0) ? 192 : 0;
  const bool gen_ok = g_head_mfma.load(std::memory_order_relaxed) && dtype == GDL_BF16 && tpb != 0 && K <= 8 && P >= 1024 &&
                      (uintptr_t)chan_scale % 16 == 0 && (uintptr_t)w % 16 == 0 && (chan_scale || tpb == 192);
  const bool feat8g = gen_ok && dfeat && !feat8 && d_sP == C && (uintptr_t)dfeat % 16 == 0;
  const bool w8g = gen_ok && f_sP == C && (uintptr_t)feat % 16 == 0;
  K_SWITCH(K,
    if (dtype == GDL_BF16) {
      if constexpr (KK <= 8) {
        if (feat8g) {
          const dim3 gg(4 * gdl_num_cus());
          if (tpb == 256) hipLaunchKernelGGL((head_1x1_bwd_feat8g_kernel<KK, 256, true>), gg, dim3(256), 0, s, dlog, P, C, w, chan_scale, pix_per_img, (uint16_t*)dfeat);
          else if (chan_scale) hipLaunchKernelGGL((head_1x1_bwd_feat8g_kernel<KK, 192, true>), gg, dim3(192), 0, s, dlog, P, C, w, chan_scale, pix_per_img, (uint16_t*)dfeat);
          else hipLaunchKernelGGL((head_1x1_bwd_feat8g_kernel<KK, 192, false>), gg, dim3(192), 0, s, dlog, P, C, w, chan_scale, pix_per_img, (uint16_t*)dfeat);
        }
        if (w8g) {
          if (tpb == 256) hipLaunchKernelGGL((head_1x1_bwd_w_partial8g<KK, 256, true>), dim3(nsplit), dim3(256), 0, s, (const uint16_t*)feat, dlog, P, C, chan_scale, pix_per_img, ws);
          else if (chan_scale) hipLaunchKernelGGL((head_1x1_bwd_w_partial8g<KK, 192, true>), dim3(nsplit), dim3(192), 0, s, (const uint16_t*)feat, dlog, P, C, chan_scale, pix_per_img, ws);
          else hipLaunchKernelGGL((head_1x1_bwd_w_partial8g<KK, 192, false>), dim3(nsplit), dim3(192), 0, s, (const uint16_t*)feat, dlog, P, C, chan_scale, pix_per_img, ws);
        }
        if (feat8) hipLaunchKernelGGL((head_1x1_bwd_feat8_kernel<KK>), dim3(4 * gdl_num_cus()), dim3(256), 0, s, dlog, P, C, w, (uint16_t*)dfeat);   // (all resident at once: 74 registers allow six waves per SIMD, 8 per CU left a third of the grid for a second round)
      }
      if (dfeat && !feat8 && !feat8g) hipLaunchKernelGGL((head_1x1_bwd_feat_kernel<uint16_t, KK>), dim3(grid_for(total)), dim3(256), 0, s, dlog, P, C, w, chan_scale, pix_per_img, dfeat, d_sP);
      bool w8 = false;
      if constexpr (KK <= 8) {
        w8 = g_head_mfma.load(std::memory_order_relaxed) && !chan_scale && f_sP == C && C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0 &&
             (uintptr_t)feat % 16 == 0 && P >= 1024;
        if (w8) hipLaunchKernelGGL((head_1x1_bwd_w_partial8<KK>), dim3(nsplit), dim3(256), 0, s, (const uint16_t*)feat, dlog, P, C, ws);
      }
      if (!w8 && !w8g) hipLaunchKernelGGL((head_1x1_bwd_w_partial<uint16_t, KK>), gridw, dim3(256), 0, s, feat, dlog, P, C, f_sP, chan_scale, pix_per_img, ws);
    } else {
      if (dfeat) hipLaunchKernelGGL((head_1x1_bwd_feat_kernel<float, KK>), dim3(grid_for(total)), dim3(256), 0, s, dlog, P, C, w, chan_scale, pix_per_img, dfeat, d_sP);
      hipLaunchKernelGGL((head_1x1_bwd_w_partial<float, KK>), gridw, dim3(256), 0, s, feat, dlog, P, C, f_sP, chan_scale, pix_per_img, ws);
    }
    hipLaunchKernelGGL((head_1x1_bwd_w_final<KK>), dim3((KK * C + KK + 7) / 8), dim3(1024), 0, s, ws, nsplit, C, dw, db));
  GDL_CHECK_LAUNCH("gdl_head_1x1_bwd");
  return GDL_OK;
}

extern "C" int gdl_upsample_logits(const float* in, int B, int Hi, int Wi, int K, float* out, int Ho, int Wo,
                                   gdl_stream_t stream) {
  GDL_CHECK_ARG(in && out, "gdl_upsample_logits: null pointer");
  const int64_t total = (int64_t)B * Ho * Wo;
  K_SWITCH(K, hipLaunchKernelGGL((upsample_logits_kernel<KK>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, B, Hi, Wi, out, Ho, Wo));
  GDL_CHECK_LAUNCH("gdl_upsample_logits");
  return GDL_OK;
}

extern "C" int64_t gdl_upsample_logits_bwd_workspace(int B, int K, int Hi, int Wo) {
  return (int64_t)B * K * Hi * Wo * (int64_t)sizeof(float);
}

extern "C" int gdl_upsample_logits_bwd(const float* dout, int B, int Ho, int Wo, int K, float* din, int Hi, int Wi,
                                       float* ws, int64_t ws_bytes, gdl_stream_t stream) {
  GDL_CHECK_ARG(dout && din && ws, "gdl_upsample_logits_bwd: null pointer");
  GDL_CHECK_ARG(ws_bytes >= gdl_upsample_logits_bwd_workspace(B, K, Hi, Wo), "gdl_upsample_logits_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(upsample_logits_bwd_pass1, dim3(grid_for((int64_t)B * K * Hi * Wo)), dim3(256), 0, s, dout, B * K, Ho, Wo, ws, Hi);
  K_SWITCH(K, hipLaunchKernelGGL((upsample_logits_bwd_pass2<KK>), dim3(grid_for((int64_t)B * Hi * Wi * K)), dim3(256), 0, s, ws, B, Wo, din, Hi, Wi));
  GDL_CHECK_LAUNCH("gdl_upsample_logits_bwd");
  return GDL_OK;
}

extern "C" int gdl_softmax_argmax(const float* logits, int B, int K, int64_t HW, int64_t* mask, gdl_stream_t stream) {
  GDL_CHECK_ARG(logits && mask, "gdl_softmax_argmax: null pointer");
  const int64_t total = (int64_t)B * HW;
  K_SWITCH(K, hipLaunchKernelGGL((softmax_argmax_kernel<KK>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, logits, B, HW, mask));
  GDL_CHECK_LAUNCH("gdl_softmax_argmax");
  return GDL_OK;
}

extern "C" int gdl_upsample_argmax(const float* low, int B, int Hi, int Wi, int K, int64_t* mask, int Ho, int Wo, gdl_stream_t stream) {
  GDL_CHECK_ARG(low && mask && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && K >= 2, "gdl_upsample_argmax: bad args");
  const int64_t total = (int64_t)B * Ho * Wo;
  K_SWITCH(K, hipLaunchKernelGGL((upsample_argmax_kernel<KK>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, low, B, Hi, Wi, Ho, Wo, mask));
  GDL_CHECK_LAUNCH("gdl_upsample_argmax");
  return GDL_OK;
}

extern "C" int gdl_class_probs(const float* logits, int B, int K, int64_t HW, float* probs, gdl_stream_t stream) {
  GDL_CHECK_ARG(logits && probs, "gdl_class_probs: null pointer");
  const int64_t total = (int64_t)B * HW;
  K_SWITCH(K, hipLaunchKernelGGL((class_probs_kernel<KK>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, logits, B, HW, probs));
  GDL_CHECK_LAUNCH("gdl_class_probs");
  return GDL_OK;
}

static int dice_blocks(int64_t total) {
  int64_t g = (total + 2047) / 2048;
  return (int)(g < 1 ? 1 : (g > 512 ? 512 : g));
}

extern "C" int64_t gdl_dice_loss_workspace(int B, int K, int64_t HW) {
  return (int64_t)dice_blocks((int64_t)B * HW) * 3 * K * sizeof(float);
}

extern "C" int gdl_dice_loss_fwd(const float* logits, const int64_t* target, int B, int K, int64_t HW, float eps,
                                 float* sums, float* loss, float* ws, int64_t ws_bytes, gdl_stream_t stream) {
  GDL_CHECK_ARG(logits && target && sums && loss && ws, "gdl_dice_loss_fwd: null pointer");
  GDL_CHECK_ARG(ws_bytes >= gdl_dice_loss_workspace(B, K, HW), "gdl_dice_loss_fwd: workspace too small");
  const int nblk = dice_blocks((int64_t)B * HW);
  hipStream_t s = (hipStream_t)stream;
  K_SWITCH(K, hipLaunchKernelGGL((dice_partial_kernel<KK>), dim3(nblk), dim3(256), 0, s, logits, target, B, HW, ws);
              hipLaunchKernelGGL((dice_final_kernel<KK>), dim3(1), dim3(256), 0, s, ws, nblk, eps, sums, loss));
  GDL_CHECK_LAUNCH("gdl_dice_loss_fwd");
  return GDL_OK;
}

// (four times the workgroups of dice_partial_kernel: the scattered 4-byte loads of the on-the-fly bilinear logit are a chain of L2
// round trips per pixel, and 512 workgroups = two waves per SIMD do not cover it)
static int dice_lowres_blocks(int64_t total) {
  int64_t g = (total + 1023) / 1024;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}
extern "C" int64_t gdl_dice_loss_lowres_workspace(int B, int K, int Ho, int Wo) {
  return (int64_t)dice_lowres_blocks((int64_t)B * Ho * Wo) * 3 * K * sizeof(float);
}

// Dice loss (multiclass) of bilinear(low -> [Ho, Wo]) against target [B, Ho, Wo] WITHOUT the full-resolution logits: low = the
// [B, Hi, Wi, K] f32 map gdl_head_1x1 writes.  sums / loss as gdl_dice_loss_fwd; workspace of gdl_dice_loss_lowres_workspace(B, K, Ho,
// Wo) bytes.  Upsampling factors up to 64 per direction.
extern "C" int gdl_dice_loss_lowres_fwd(const float* low, const int64_t* target, int B, int K, int Hi, int Wi, int Ho, int Wo, float eps,
                                        float* sums, float* loss, float* ws, int64_t ws_bytes, gdl_stream_t stream) {
  GDL_CHECK_ARG(low && target && sums && loss && ws, "gdl_dice_loss_lowres_fwd: null pointer");
  GDL_CHECK_ARG(B > 0 && Hi > 0 && Wi > 0 && Ho >= Hi && Wo >= Wi, "gdl_dice_loss_lowres_fwd: bad sizes (an upsample is expected)");
  GDL_CHECK_ARG((Ho + Hi - 1) / Hi <= DICE_LOWRES_MAX_FACTOR && (Wo + Wi - 1) / Wi <= DICE_LOWRES_MAX_FACTOR,
                "gdl_dice_loss_lowres_fwd: upsampling factors above 64 are not supported");
  GDL_CHECK_ARG(ws_bytes >= gdl_dice_loss_lowres_workspace(B, K, Ho, Wo), "gdl_dice_loss_lowres_fwd: workspace too small");
  const int nblk = dice_lowres_blocks((int64_t)B * Ho * Wo);
  hipStream_t s = (hipStream_t)stream;
  K_SWITCH(K, hipLaunchKernelGGL((dice_lowres_partial_kernel<KK>), dim3(nblk), dim3(256), 0, s, low, target, B, Hi, Wi, Ho, Wo, ws);
              hipLaunchKernelGGL((dice_final_kernel<KK>), dim3(1), dim3(256), 0, s, ws, nblk, eps, sums, loss));
  GDL_CHECK_LAUNCH("gdl_dice_loss_lowres_fwd");
  return GDL_OK;
}

static std::atomic<int> g_dice_tiled{1};
extern "C" void gdl_debug_set_dice_lowres_tiled(int on) { g_dice_tiled = on; }   // A/B hook: 0 = the gather kernel for every class count

static bool dice_tile_dims(int K, int Hi, int Wi, int Ho, int Wo, int& ny_max, int& nx_max) {
  // low-resolution rows / columns one tile side can touch: DT * ratio + 2 (an upper bound for ratios <= 1)
  ny_max = (int)((int64_t)DT_H * Hi / Ho) + 3;
  nx_max = (int)((int64_t)DT_W * Wi / Wo) + 3;
  return g_dice_tiled && K <= 8 && ny_max <= DT_MAXN && nx_max <= DT_MAXN + DT_MAXN;
}

// bytes of scratch gdl_dice_loss_lowres_bwd needs (0: none -- the gather kernel)
extern "C" int64_t gdl_dice_loss_lowres_bwd_workspace(int B, int K, int Hi, int Wi, int Ho, int Wo) {
  int ny, nx;
  if (B <= 0 || Hi <= 0 || Wi <= 0 || Ho < Hi || Wo < Wi || !dice_tile_dims(K, Hi, Wi, Ho, Wo, ny, nx)) return 0;
  const int64_t tiles = (int64_t)B * ((Ho + DT_H - 1) / DT_H) * ((Wo + DT_W - 1) / DT_W);
  return tiles * ny * nx * K * (int64_t)sizeof(float);
}

// d loss / d low [B, Hi, Wi, K] (f32, overwritten) from the sums of the forward; upstream (device scalar, may be null) * grad_scale
// multiplies the gradient.  ws: gdl_dice_loss_lowres_bwd_workspace() bytes (may be null when that is 0).
extern "C" int gdl_dice_loss_lowres_bwd(const float* low, const int64_t* target, int B, int K, int Hi, int Wi, int Ho, int Wo, float eps,
                                        const float* sums, const float* upstream, float grad_scale, float* dlow, float* ws,
                                        int64_t ws_bytes, gdl_stream_t stream) {
  GDL_CHECK_ARG(low && target && sums && dlow, "gdl_dice_loss_lowres_bwd: null pointer");
  GDL_CHECK_ARG(B > 0 && Hi > 0 && Wi > 0 && Ho >= Hi && Wo >= Wi, "gdl_dice_loss_lowres_bwd: bad sizes");
  GDL_CHECK_ARG((Ho + Hi - 1) / Hi <= DICE_LOWRES_MAX_FACTOR && (Wo + Wi - 1) / Wi <= DICE_LOWRES_MAX_FACTOR,
                "gdl_dice_loss_lowres_bwd: upsampling factors above 64 are not supported");
  {
    int ny, nx;
    const int64_t need = gdl_dice_loss_lowres_bwd_workspace(B, K, Hi, Wi, Ho, Wo);
    if (need > 0 && ws && ws_bytes >= need && dice_tile_dims(K, Hi, Wi, Ho, Wo, ny, nx)) {
      DiceTile a;
      a.low = low; a.target = target; a.sums = sums; a.upstream = upstream; a.ws = ws; a.dlow = dlow;
      a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ho = Ho; a.Wo = Wo;
      a.tiles_y = (Ho + DT_H - 1) / DT_H; a.tiles_x = (Wo + DT_W - 1) / DT_W; a.ny_max = ny; a.nx_max = nx;
      a.eps = eps; a.grad_scale = grad_scale;
      const unsigned tiles = (unsigned)(B * a.tiles_y * a.tiles_x);
      const int64_t total = (int64_t)B * Hi * Wi;
      hipStream_t st = (hipStream_t)stream;
      K_SWITCH(K, if (KK <= 8) {
                    const size_t lds = ((size_t)KK * DT_H * DT_W + (size_t)KK * ny * (DT_W + 1) + (size_t)ny * DT_H + (size_t)nx * DT_W) * sizeof(float);
                    GDL_SET_MAX_LDS_ONCE((dice_lowres_bwd_tile_kernel<(KK <= 8 ? KK : 8)>), 160 * 1024);
                    hipLaunchKernelGGL((dice_lowres_bwd_tile_kernel<(KK <= 8 ? KK : 8)>), dim3(tiles), dim3(DT_T), lds, st, a);
                    hipLaunchKernelGGL((dice_lowres_bwd_reduce_kernel<(KK <= 8 ? KK : 8)>), dim3(grid_for(total)), dim3(256), 0, st, a);
                  });
      GDL_CHECK_LAUNCH("gdl_dice_loss_lowres_bwd");
      return GDL_OK;
    }
  }
  const int64_t total = (int64_t)B * Hi * Wi;
  K_SWITCH(K, hipLaunchKernelGGL((dice_lowres_bwd_kernel<KK>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, low,
                                 target, B, Hi, Wi, Ho, Wo, sums, eps, upstream, grad_scale, dlow));
  GDL_CHECK_LAUNCH("gdl_dice_loss_lowres_bwd");
  return GDL_OK;
}

extern "C" int gdl_dice_loss_bwd(const float* logits, const int64_t* target, int B, int K, int64_t HW, float eps,
                                 const float* sums, const float* upstream, float grad_scale, float* dlogits,
                                 int accumulate, gdl_stream_t stream) {
  GDL_CHECK_ARG(logits && target && sums && dlogits, "gdl_dice_loss_bwd: null pointer");
  const int64_t total = (int64_t)B * HW;
  K_SWITCH(K, hipLaunchKernelGGL((dice_bwd_kernel<KK>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, logits, target, B, HW, sums, eps, upstream, grad_scale, dlogits, accumulate));
  GDL_CHECK_LAUNCH("gdl_dice_loss_bwd");
  return GDL_OK;
}

extern "C" int gdl_dice_binary_loss_fwd(const float* logits, const int64_t* target, int64_t total, float eps,
                                        float* sums, float* loss, float* ws, int64_t ws_bytes, gdl_stream_t stream) {
  GDL_CHECK_ARG(logits && target && sums && loss && ws, "gdl_dice_binary_loss_fwd: null pointer");
  const int nblk = dice_blocks(total);
  GDL_CHECK_ARG(ws_bytes >= (int64_t)nblk * 3 * (int64_t)sizeof(float), "gdl_dice_binary_loss_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(dice_binary_partial_kernel, dim3(nblk), dim3(256), 0, s, logits, target, total, ws);
  hipLaunchKernelGGL((dice_final_kernel<1>), dim3(1), dim3(256), 0, s, ws, nblk, eps, sums, loss);
  GDL_CHECK_LAUNCH("gdl_dice_binary_loss_fwd");
  return GDL_OK;
}

extern "C" int gdl_dice_binary_loss_bwd(const float* logits, const int64_t* target, int64_t total, float eps,
                                        const float* sums, const float* upstream, float grad_scale, float* dlogits,
                                        int accumulate, gdl_stream_t stream) {
  GDL_CHECK_ARG(logits && target && sums && dlogits, "gdl_dice_binary_loss_bwd: null pointer");
  hipLaunchKernelGGL(dice_binary_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, logits, target,
                     total, sums, eps, upstream, grad_scale, dlogits, accumulate);
  GDL_CHECK_LAUNCH("gdl_dice_binary_loss_bwd");
  return GDL_OK;
}

extern "C" int gdl_sumsq(const float* x, int64_t n, float* out_accum, gdl_stream_t stream) {
  GDL_CHECK_ARG(x && out_accum, "gdl_sumsq: null pointer");
  if (n <= 0) return GDL_OK;
  int64_t g = (n + 4095) / 4096; if (g > 1024) g = 1024;
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, n, out_accum);
  GDL_CHECK_LAUNCH("gdl_sumsq");
  return GDL_OK;
}

extern "C" int gdl_clip_coef(const float* sumsq, float max_norm, float* coef, gdl_stream_t stream) {
  GDL_CHECK_ARG(sumsq && coef, "gdl_clip_coef: null pointer");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm, coef);
  GDL_CHECK_LAUNCH("gdl_clip_coef");
  return GDL_OK;
}

extern "C" int gdl_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, float bc1, float bc2,
                             const float* clip_coef, gdl_stream_t stream) {
  GDL_CHECK_ARG(p && g && m && v, "gdl_adam_step: null pointer");
  if (n <= 0) return GDL_OK;
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2,
                     eps, weight_decay, bc1, bc2, clip_coef);
  GDL_CHECK_LAUNCH("gdl_adam_step");
  return GDL_OK;
}

extern "C" int gdl_iou_counts(const int64_t* pred, const int64_t* target, int B, int64_t P, int K, int64_t* counts,
                              gdl_stream_t stream) {
  GDL_CHECK_ARG(pred && target && counts && B > 0 && P > 0, "gdl_iou_counts: bad args");
  GDL_CHECK_ARG(K > 0 && K <= 64, "gdl_iou_counts: 1..64 classes");
  GDL_CHECK_ARG(P / 64 < (1ll << 32), "gdl_iou_counts: sample too large");
  hipStream_t s = (hipStream_t)stream;
  (void)hipMemsetAsync(counts, 0, (size_t)B * 3 * K * sizeof(int64_t), s);
  int64_t chunks = (P + 65535) / 65536;
  if (chunks > 1024) chunks = 1024;
  hipLaunchKernelGGL(iou_counts_kernel, dim3((unsigned)chunks, B), dim3(256), 0, s, pred, target, P, K,
                     (unsigned long long*)counts);
  GDL_CHECK_LAUNCH("gdl_iou_counts");
  return GDL_OK;
}
