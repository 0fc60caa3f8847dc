// LayerNorm / BatchNorm kernels (HBM-bound; wave-shuffle reductions, 16-byte accesses).
#include "gdl_common.h"

namespace {

// ---------------------------------------------------------------- LayerNorm forward
// One wave per row, row kept in registers (D <= 64*4*VPL), f32 statistics.
template <int VPL>  // float4 vectors per lane
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x,
                                                            int64_t x_stride,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, void* y,
                                                            int y_dtype, int64_t rows, int D,
                                                            float eps) {
  const int lane = threadIdx.x & 63;
  // narrow rows (D <= 128): 16 or 32 lanes hold a row, a wave normalises 4 or 2 rows (rows_per_block = 4 * nsub)
  const int lpp = (VPL == 1 && D <= 128) ? (D <= 64 ? 16 : 32) : 64;
  const int nsub = 64 / lpp, cl = lane % lpp;
  const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * nsub + lane / lpp;
  if (row >= rows) return;
  auto group_sum = [&](float t) {
    for (int o = lpp >> 1; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    return t;
  };
  const float* xr = x + row * x_stride;
  float4 v[VPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (i * 64 + cl) * 4;
    v[i] = c < D ? *(const float4*)(xr + c) : make_float4(0, 0, 0, 0);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = group_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (i * 64 + cl) * 4;
    if (c < D) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = rsqrtf(group_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (i * 64 + cl) * 4;
    if (c >= D) continue;
    const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
    const float o0 = (v[i].x - mean) * rstd * g.x + b.x, o1 = (v[i].y - mean) * rstd * g.y + b.y;
    const float o2 = (v[i].z - mean) * rstd * g.z + b.z, o3 = (v[i].w - mean) * rstd * g.w + b.w;
    if (y_dtype == GDL_BF16) {
      *(uint2*)((uint16_t*)y + row * D + c) = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
    } else {
      *(float4*)((float*)y + row * D + c) = make_float4(o0, o1, o2, o3);
    }
  }
}

// ---------------------------------------------------------------- BatchNorm (NHWC, [P][C])
// A wave covers 256 channels (4 per lane); blockDim = 256 = 4 pixel lanes x 64.
// grid = (ceil(C/256), nsplit).  Partial sums -> workspace[split][2][C].
template <typename T>
__device__ __forceinline__ void load4(const void* p, int64_t off, float (&o)[4]) {
  if constexpr (sizeof(T) == 4) {
    const float4 v = *(const float4*)((const float*)p + off);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  } else {
    const uint2 v = *(const uint2*)((const uint16_t*)p + off);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  }
}
template <typename T>
__device__ __forceinline__ void store4(void* p, int64_t off, const float (&o)[4]) {
  if constexpr (sizeof(T) == 4) {
    *(float4*)((float*)p + off) = make_float4(o[0], o[1], o[2], o[3]);
  } else {
    *(uint2*)((uint16_t*)p + off) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
  }
}

// Lane -> (4-channel group, pixel sub-index).  Wide layers: the 64 lanes of a wave cover 256 channels of ONE pixel.
// Narrow layers (C <= 128: ResNet / UNet++ 16- to 128-channel maps): 4 .. 32 lanes cover a pixel and the wave
// handles 16 .. 2 pixels per step, so no lane idles.  A wave then strides 4 * nsub pixels.
struct LaneMap { int c, sub, nsub, lpp; };
__device__ __forceinline__ LaneMap lane_map(int lane, int C, int block_x) {
  const int lpp = C > 128 ? 64 : (C > 64 ? 32 : (C > 32 ? 16 : (C > 16 ? 8 : 4)));
  LaneMap m;
  m.lpp = lpp; m.nsub = 64 / lpp; m.sub = lane / lpp; m.c = block_x * 256 + (lane % lpp) * 4;
  return m;
}
// index into a per-wave [64 lanes][4] scratch row of the value lane-group `sub` holds for channel t of the block
__device__ __forceinline__ int lane_slot(const LaneMap& m, int sub, int t) { return (sub * m.lpp + (t >> 2)) * 4 + (t & 3); }

template <typename T>
__global__ __launch_bounds__(256) void bn_stats_partial(const void* __restrict__ x, int64_t P, int C,
                                                        int64_t x_sP, float* __restrict__ ws) {
  __shared__ float red[2][4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const LaneMap lm = lane_map(lane, C, blockIdx.x);
  const int c = lm.c, st = 4 * lm.nsub;
  const int nsplit = gridDim.y;
  const int64_t per = (P + nsplit - 1) / nsplit;
  const int64_t p0 = per * blockIdx.y, p1 = p0 + per < P ? p0 + per : P;
  float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (c < C) {
    int64_t p = p0 + w * lm.nsub + lm.sub;
    for (; p + 3 * st < p1; p += 4 * st) {      // four independent loads in flight per lane
      float v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) load4<T>(x, (p + st * u) * x_sP + c, v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[j] += v[u][j]; q[j] += v[u][j] * v[u][j]; }
    }
    for (; p < p1; p += st) {
      float v[4];
      load4<T>(x, p * x_sP + c, v);
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[j] += v[j]; q[j] += v[j] * v[j]; }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[0][w][lane * 4 + j] = s[j]; red[1][w][lane * 4 + j] = q[j]; }
  __syncthreads();
  const int t = threadIdx.x;
  const int cc = blockIdx.x * 256 + t;
  if (cc < C) {
    float ss = 0.f, qq = 0.f;
    for (int sub = 0; sub < lm.nsub; ++sub) {
      const int i = lane_slot(lm, sub, t);
      ss += (red[0][0][i] + red[0][1][i]) + (red[0][2][i] + red[0][3][i]);
      qq += (red[1][0][i] + red[1][1][i]) + (red[1][2][i] + red[1][3][i]);
    }
    ws[((int64_t)blockIdx.y * 2 + 0) * C + cc] = ss;
    ws[((int64_t)blockIdx.y * 2 + 1) * C + cc] = qq;
  }
}

// Two column sums over the nsplit partial rows at once (s: row 2i, q: row 2i + 1), each added in row order i = begin, begin + step, ...
// with EIGHT rows of both in flight (the pattern of ordered_sum8)
__device__ __forceinline__ void ordered_sum8_pair(const float* __restrict__ ws, int begin, int nsplit, int step, int C, int c, double& s,
                                                  double& q) {
  s = 0; q = 0;
  int i = begin;
  for (; i + 7 * step < nsplit; i += 8 * step) {
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = ws[((int64_t)(i + u * step) * 2) * C + c];
      b[u] = ws[((int64_t)(i + u * step) * 2 + 1) * C + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { s += a[u]; q += b[u]; }
  }
  for (; i < nsplit; i += step) { s += ws[((int64_t)i * 2) * C + c]; q += ws[((int64_t)i * 2 + 1) * C + c]; }
}

constexpr int BN_FINAL_T = 1024, BN_FINAL_G = BN_FINAL_T / 4;     // threads; split groups per channel

// block = 4 channels x 256 split groups (round 5; was 64: up to 2600 partial rows meant 40 rows per thread in five dependent batches
// for the sum and five more for the squares -- 14.5 us per BatchNorm layer, 21 layers).  Both sums' loads go out together, the groups
// meet in the LDS in a fixed order (256 -> 16 -> 1).
__device__ __forceinline__ bool bn_final_reduce(double (&ps)[BN_FINAL_G][4], double (&pq)[BN_FINAL_G][4], int cl, int grp, double& s,
                                                double& q) {
  ps[grp][cl] = s; pq[grp][cl] = q;
  __syncthreads();
  if (grp < 16) {
    s = 0; q = 0;
#pragma unroll
    for (int g = 0; g < 16; ++g) { s += ps[grp * 16 + g][cl]; q += pq[grp * 16 + g][cl]; }
  }
  __syncthreads();
  if (grp < 16) { ps[grp][cl] = s; pq[grp][cl] = q; }
  __syncthreads();
  if (grp != 0) return false;
  s = 0; q = 0;
#pragma unroll
  for (int g = 0; g < 16; ++g) { s += ps[g][cl]; q += pq[g][cl]; }
  return true;
}

__global__ __launch_bounds__(BN_FINAL_T) void bn_stats_final(const float* __restrict__ ws, int nsplit, int C, int64_t P,
                                                             float* __restrict__ mean, float* __restrict__ var, float* running_mean,
                                                             float* running_var, float momentum) {
  __shared__ double ps[BN_FINAL_G][4], pq[BN_FINAL_G][4];
  const int cl = threadIdx.x & 3, grp = threadIdx.x >> 2;
  const int c = blockIdx.x * 4 + cl;
  double s = 0, q = 0;
  if (c < C) ordered_sum8_pair(ws, grp, nsplit, BN_FINAL_G, C, c, s, q);
  if (!bn_final_reduce(ps, pq, cl, grp, s, q) || c >= C) return;
  const double m = s / (double)P;
  double v = q / (double)P - m * m;
  v = v > 0 ? v : 0;
  mean[c] = (float)m;
  var[c] = (float)v;
  if (running_mean) {  // nn.BatchNorm2d: unbiased variance for the running estimate
    const double unb = P > 1 ? v * (double)P / (double)(P - 1) : v;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const void* __restrict__ x, void* y, int64_t P,
                                                       int C, int64_t x_sP, int64_t y_sP,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ var,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps,
                                                       int relu) {
  // lane owns 4 fixed channels (params live in registers); grid = (C/256, pixel splits)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const LaneMap lm = lane_map(lane, C, blockIdx.x);
  const int c = lm.c;
  if (c >= C) return;
  float mu[4], sc[4], be[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { mu[j] = mean[c + j]; sc[j] = rsqrtf(var[c + j] + eps) * gamma[c + j]; be[j] = beta[c + j]; }
  const int64_t per = (P + gridDim.y - 1) / gridDim.y;
  const int64_t p0 = per * blockIdx.y, p1 = p0 + per < P ? p0 + per : P;
  for (int64_t p = p0 + w * lm.nsub + lm.sub; p < p1; p += 4 * lm.nsub) {
    float v[4];
    load4<T>(x, p * x_sP + c, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float o = (v[j] - mu[j]) * sc[j] + be[j];
      v[j] = relu ? fmaxf(o, 0.f) : o;
    }
    store4<T>(y, p * y_sP + c, v);
  }
}

// backward pass 1: dbeta = sum g, dgamma = sum g*xhat with g = dy * [bn(x) > 0]
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_partial(const void* __restrict__ x,
                                                      const void* __restrict__ dy, int64_t P, int C,
                                                      int64_t x_sP, int64_t dy_sP,
                                                      const float* __restrict__ mean,
                                                      const float* __restrict__ var,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, int relu,
                                                      float* __restrict__ ws) {
  __shared__ float red[2][4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const LaneMap lm = lane_map(lane, C, blockIdx.x);
  const int c = lm.c, st = 4 * lm.nsub;
  const int nsplit = gridDim.y;
  const int64_t per = (P + nsplit - 1) / nsplit;
  const int64_t p0 = per * blockIdx.y, p1 = p0 + per < P ? p0 + per : P;
  float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (c < C) {
    float mu[4], rs[4], ga[4], be[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mu[j] = mean[c + j]; rs[j] = rsqrtf(var[c + j] + eps); ga[j] = gamma[c + j]; be[j] = beta[c + j];
    }
    int64_t p = p0 + w * lm.nsub + lm.sub;
    for (; p + st < p1; p += 2 * st) {          // two pixel rows (four loads) in flight per lane
      float v[2][4], g[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) { load4<T>(x, (p + st * u) * x_sP + c, v[u]); load4<T>(dy, (p + st * u) * dy_sP + c, g[u]); }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (v[u][j] - mu[j]) * rs[j];
          const float gg = (relu && !(xh * ga[j] + be[j] > 0.f)) ? 0.f : g[u][j];
          s[j] += gg; q[j] += gg * xh;
        }
    }
    for (; p < p1; p += st) {
      float v[4], g[4];
      load4<T>(x, p * x_sP + c, v);
      load4<T>(dy, p * dy_sP + c, g);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = (v[j] - mu[j]) * rs[j];
        const float gg = (relu && !(xh * ga[j] + be[j] > 0.f)) ? 0.f : g[j];
        s[j] += gg; q[j] += gg * xh;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[0][w][lane * 4 + j] = s[j]; red[1][w][lane * 4 + j] = q[j]; }
  __syncthreads();
  const int t = threadIdx.x, cc = blockIdx.x * 256 + t;
  if (cc < C) {
    float ss = 0.f, qq = 0.f;
    for (int sub = 0; sub < lm.nsub; ++sub) {
      const int i = lane_slot(lm, sub, t);
      ss += (red[0][0][i] + red[0][1][i]) + (red[0][2][i] + red[0][3][i]);
      qq += (red[1][0][i] + red[1][1][i]) + (red[1][2][i] + red[1][3][i]);
    }
    ws[((int64_t)blockIdx.y * 2 + 0) * C + cc] = ss;
    ws[((int64_t)blockIdx.y * 2 + 1) * C + cc] = qq;
  }
}

// bf16, dense rows, C % 8 == 0 and C / 8 a divisor of the block size: the same sums with 16-BYTE loads -- thread t owns the
// eight channels 8 (t % G) of pixel row t / G (G = C / 8 lanes cover one pixel), so a block reads whole contiguous pixel rows
// and a lane's channels never change; four row groups in flight.  The 8-byte / 256-channel mapping above reached 3.4 TB/s on
// the decoder's maps (2 x 1 GB for the neck x4 level); this one 5.0-5.9 TB/s (tools/bench_bn.py).  The element-wise passes
// (apply, dx) gain nothing from the same mapping -- they are bound by their writes at ~5.0 TB/s either way -- and neither do the
// forward statistics, whose 8-byte kernel already keeps four loads in flight (4.4-5.5 TB/s both ways; measured, not kept).
__device__ __forceinline__ void unpack8(const uint4& v, float (&o)[8]) {
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
  o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}

template <int TPB>
__global__ __launch_bounds__(TPB) void bn_bwd_partial8(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, int64_t P,
                                                       int C, const float* __restrict__ mean, const float* __restrict__ var,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                       int relu, float* __restrict__ ws) {
  __shared__ float red[2][TPB][9];          // [sum | sum * xhat][thread][channel] (+1: bank spread)
  const int G = C >> 3, R = TPB / G;        // lanes per pixel, pixel rows per block step
  const int t = threadIdx.x, g = t % G, prow = t / G;
  const int nsplit = gridDim.x;
  const int64_t per = (P + nsplit - 1) / nsplit;
  const int64_t p0 = per * blockIdx.x, p1 = p0 + per < P ? p0 + per : P;
  float s[8], q[8], sc[8], sh[8], mu[8], rs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = 8 * g + j;
    s[j] = 0.f; q[j] = 0.f;
    mu[j] = mean[c]; rs[j] = rsqrtf(var[c] + eps);
    sc[j] = gamma[c]; sh[j] = beta[c];
  }
  auto add = [&](const uint4& xv, const uint4& gv) {
    float v[8], d[8];
    unpack8(xv, v); unpack8(gv, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (v[j] - mu[j]) * rs[j];
      const float gg = (relu && !(xh * sc[j] + sh[j] > 0.f)) ? 0.f : d[j];
      s[j] += gg; q[j] += gg * xh;
    }
  };
  if (prow < R) {
    int64_t p = p0 + prow;
    for (; p + 3 * R < p1; p += 4 * R) {
      uint4 xv[4], gv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        xv[u] = *(const uint4*)(x + (p + (int64_t)u * R) * C + 8 * g);
        gv[u] = *(const uint4*)(dy + (p + (int64_t)u * R) * C + 8 * g);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) add(xv[u], gv[u]);
    }
    for (; p < p1; p += R) add(*(const uint4*)(x + p * C + 8 * g), *(const uint4*)(dy + p * C + 8 * g));
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[0][t][j] = s[j]; red[1][t][j] = q[j]; }
  __syncthreads();
  for (int c = t; c < C; c += TPB) {
    float ss = 0.f, qq = 0.f;
    for (int r = 0; r < R; ++r) { ss += red[0][r * G + (c >> 3)][c & 7]; qq += red[1][r * G + (c >> 3)][c & 7]; }
    ws[((int64_t)blockIdx.x * 2 + 0) * C + c] = ss;
    ws[((int64_t)blockIdx.x * 2 + 1) * C + c] = qq;
  }
}

__global__ __launch_bounds__(BN_FINAL_T) void bn_bwd_final(const float* __restrict__ ws, int nsplit, int C, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta) {
  __shared__ double ps[BN_FINAL_G][4], pq[BN_FINAL_G][4];
  const int cl = threadIdx.x & 3, grp = threadIdx.x >> 2;
  const int c = blockIdx.x * 4 + cl;
  double s = 0, q = 0;
  if (c < C) ordered_sum8_pair(ws, grp, nsplit, BN_FINAL_G, C, c, s, q);
  if (!bn_final_reduce(ps, pq, cl, grp, s, q) || c >= C) return;
  dbeta[c] = (float)s;
  dgamma[c] = (float)q;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_dx(const void* __restrict__ x, const void* __restrict__ dy,
                                                 void* dx, int64_t P, int64_t P_total, int C, int64_t x_sP, int64_t dy_sP,
                                                 int64_t dx_sP, const float* __restrict__ mean,
                                                 const float* __restrict__ var,
                                                 const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float eps, int relu,
                                                 const float* __restrict__ dgamma,
                                                 const float* __restrict__ dbeta, const float* __restrict__ total_dev) {
  // SyncBatchNorm: the sums are global and so is the pixel count, which only exists on the device (ranks with ragged batches)
  const float invP = total_dev ? 1.0f / total_dev[0] : 1.0f / (float)P_total;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const LaneMap lm = lane_map(lane, C, blockIdx.x);
  const int c = lm.c;
  if (c >= C) return;
  float mu[4], rs[4], ga[4], be[4], k1[4], k2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mu[j] = mean[c + j]; rs[j] = rsqrtf(var[c + j] + eps); ga[j] = gamma[c + j]; be[j] = beta[c + j];
    k1[j] = dbeta[c + j] * invP; k2[j] = dgamma[c + j] * invP;
  }
  const int64_t per = (P + gridDim.y - 1) / gridDim.y;
  const int64_t p0 = per * blockIdx.y, p1 = p0 + per < P ? p0 + per : P;
  for (int64_t p = p0 + w * lm.nsub + lm.sub; p < p1; p += 4 * lm.nsub) {
    float v[4], g[4];
    load4<T>(x, p * x_sP + c, v);
    load4<T>(dy, p * dy_sP + c, g);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (v[j] - mu[j]) * rs[j];
      const float gg = (relu && !(xh * ga[j] + be[j] > 0.f)) ? 0.f : g[j];
      v[j] = ga[j] * rs[j] * (gg - k1[j] - xh * k2[j]);
    }
    store4<T>(dx, p * dx_sP + c, v);
  }
}

// ---------------------------------------------------------------- small maps: the whole BatchNorm in ONE launch per direction
// Round 5.  At the reference's own per-GPU batch of 4 (configs/dofa_config_RGB.yaml:85) fourteen of the 21 train-mode
// ConvModules of DOFA + UperNet have at most 36 x 36 x 4 = 5184 pixels: their BatchNorm is five launches forward (partial
// statistics, final, normalise) and backward (partial sums, final, dx) of 5-10 us each, shorter than the gap between two
// dependent launches -- in eager mode and inside a hipGraph alike.  Here one workgroup owns FOUR channels over ALL pixels:
// pass 1 accumulates the sums (f32 per thread, f64 across the block, like the multi-block kernels), the statistics are
// finished in LDS, pass 2 re-reads its 8 B / pixel column (just read: L2) and writes the result.  grid = C / 4 workgroups of
// 1024 threads, a thread strides 1024 pixels with four loads in flight in BOTH passes (the first version -- 256 threads, an
// un-unrolled second pass -- was one dependent L2 round trip per pixel and LOST to the three short launches inside a hipGraph:
// 450 vs 461 tiles/s at batch 4, profiles/r05c_*).
constexpr int BNS_T = 1024;      // threads per workgroup of the small-map kernels (16 waves: 5 pixels per thread at 36 x 36 x 4)

// block-wide sum of 8 values per thread (4 channels x {first, second} sum): f32 butterflies inside each wave (what the
// multi-workgroup kernels do inside a workgroup), then the 16 wave results in f64 through LDS in a fixed order -- deterministic.
// Result valid in every thread.
__device__ __forceinline__ void bns_block_sum(const float (&s)[4], const float (&q)[4], double (&out)[8], float (*red)[8]) {
  float v[8] = {s[0], s[1], s[2], s[3], q[0], q[1], q[2], q[3]};
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[j] += __shfl_xor(v[j], o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red[w][j] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    double t = 0;
#pragma unroll 1
    for (int k = 0; k < BNS_T / 64; ++k) t += (double)red[k][j];
    out[j] = t;
  }
}

template <typename T>
__global__ __launch_bounds__(BNS_T) void bn_small_fwd_kernel(const void* __restrict__ x, void* y, int64_t P, int C, int64_t x_sP,
                                                             int64_t y_sP, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, int relu,
                                                             float* __restrict__ mean, float* __restrict__ var, float* running_mean,
                                                             float* running_var, float momentum) {
  __shared__ float red[BNS_T / 64][8];
  const int t = threadIdx.x, c = blockIdx.x * 4;
  float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  int64_t p = t;
  for (; p + 3 * BNS_T < P; p += 4 * BNS_T) {
    float v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) load4<T>(x, (p + BNS_T * u) * x_sP + c, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[j] += v[u][j]; q[j] += v[u][j] * v[u][j]; }
  }
  for (; p < P; p += BNS_T) {
    float v[4];
    load4<T>(x, p * x_sP + c, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) { s[j] += v[j]; q[j] += v[j] * v[j]; }
  }
  double acc[8];
  bns_block_sum(s, q, acc, red);
  float mu[4], sc[4], be[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const double m = acc[j] / (double)P;
    double v = acc[4 + j] / (double)P - m * m;
    v = v > 0 ? v : 0;
    if (t == 0) {
      mean[c + j] = (float)m;
      var[c + j] = (float)v;
      if (running_mean) {
        const double unb = P > 1 ? v * (double)P / (double)(P - 1) : v;
        running_mean[c + j] = (float)((1.0 - momentum) * running_mean[c + j] + momentum * m);
        running_var[c + j] = (float)((1.0 - momentum) * running_var[c + j] + momentum * unb);
      }
    }
    mu[j] = (float)m;
    sc[j] = rsqrtf((float)v + eps) * gamma[c + j];   // the same f32 expressions as bn_apply_kernel: identical outputs
    be[j] = beta[c + j];
  }
  p = t;
  for (; p + 3 * BNS_T < P; p += 4 * BNS_T) {          // pass 2: the column was just read (L2); four loads in flight again
    float v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) load4<T>(x, (p + BNS_T * u) * x_sP + c, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float o = (v[u][j] - mu[j]) * sc[j] + be[j];
        v[u][j] = relu ? fmaxf(o, 0.f) : o;
      }
      store4<T>(y, (p + BNS_T * u) * y_sP + c, v[u]);
    }
  }
  for (; p < P; p += BNS_T) {
    float v[4];
    load4<T>(x, p * x_sP + c, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float o = (v[j] - mu[j]) * sc[j] + be[j];
      v[j] = relu ? fmaxf(o, 0.f) : o;
    }
    store4<T>(y, p * y_sP + c, v);
  }
}

// backward: dbeta = sum g', dgamma = sum g' xhat (written out: they are the parameter gradients), then
// dx = gamma rstd (g' - dbeta / P - xhat dgamma / P), g' = dy [bn(x) > 0]; same per-element expressions as bn_bwd_partial / bn_bwd_dx
template <typename T>
__global__ __launch_bounds__(BNS_T) void bn_small_bwd_kernel(const void* __restrict__ x, const void* __restrict__ dy, void* dx, int64_t P,
                                                             int C, int64_t x_sP, int64_t dy_sP, int64_t dx_sP,
                                                             const float* __restrict__ mean, const float* __restrict__ var,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                             int relu, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[BNS_T / 64][8];
  const int t = threadIdx.x, c = blockIdx.x * 4;
  float mu[4], rs[4], ga[4], be[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { mu[j] = mean[c + j]; rs[j] = rsqrtf(var[c + j] + eps); ga[j] = gamma[c + j]; be[j] = beta[c + j]; }
  float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  int64_t p = t;
  for (; p + BNS_T < P; p += 2 * BNS_T) {
    float v[2][4], g[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) { load4<T>(x, (p + BNS_T * u) * x_sP + c, v[u]); load4<T>(dy, (p + BNS_T * u) * dy_sP + c, g[u]); }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = (v[u][j] - mu[j]) * rs[j];
        const float gg = (relu && !(xh * ga[j] + be[j] > 0.f)) ? 0.f : g[u][j];
        s[j] += gg; q[j] += gg * xh;
      }
  }
  for (; p < P; p += BNS_T) {
    float v[4], g[4];
    load4<T>(x, p * x_sP + c, v);
    load4<T>(dy, p * dy_sP + c, g);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (v[j] - mu[j]) * rs[j];
      const float gg = (relu && !(xh * ga[j] + be[j] > 0.f)) ? 0.f : g[j];
      s[j] += gg; q[j] += gg * xh;
    }
  }
  double acc[8];
  bns_block_sum(s, q, acc, red);
  float k1[4], k2[4];
  const float invP = 1.0f / (float)P;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float db = (float)acc[j], dg = (float)acc[4 + j];
    if (t == 0) { dbeta[c + j] = db; dgamma[c + j] = dg; }
    k1[j] = db * invP;
    k2[j] = dg * invP;
  }
  p = t;
  for (; p + BNS_T < P; p += 2 * BNS_T) {
    float v[2][4], g[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) { load4<T>(x, (p + BNS_T * u) * x_sP + c, v[u]); load4<T>(dy, (p + BNS_T * u) * dy_sP + c, g[u]); }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = (v[u][j] - mu[j]) * rs[j];
        const float gg = (relu && !(xh * ga[j] + be[j] > 0.f)) ? 0.f : g[u][j];
        v[u][j] = ga[j] * rs[j] * (gg - k1[j] - xh * k2[j]);
      }
      store4<T>(dx, (p + BNS_T * u) * dx_sP + c, v[u]);
    }
  }
  for (; p < P; p += BNS_T) {
    float v[4], g[4];
    load4<T>(x, p * x_sP + c, v);
    load4<T>(dy, p * dy_sP + c, g);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (v[j] - mu[j]) * rs[j];
      const float gg = (relu && !(xh * ga[j] + be[j] > 0.f)) ? 0.f : g[j];
      v[j] = ga[j] * rs[j] * (gg - k1[j] - xh * k2[j]);
    }
    store4<T>(dx, p * dx_sP + c, v);
  }
}

// ---------------------------------------------------------------- SyncBatchNorm message (round 5)
// One rank's statistics as the count-weighted message that is summed over the ranks: [n mean | n E[x^2] | n] (2 C + 1 floats),
// and the global statistics back out of the summed message (+ the running-estimate update with the unbiased variance and the
// global count, which never leaves the device).  Replaces ~5 + ~10 one-line torch kernels per layer: with 21 train-mode
// BatchNorms that was ~300 launches of a DDP + SyncBatchNorm step (configs/dofa_config_RGB.yaml:5-13).
__global__ __launch_bounds__(256) void syncbn_pack_kernel(const float* __restrict__ mean, const float* __restrict__ var, float n, int C,
                                                          float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) {
    const float m = mean[c];
    out[c] = m * n;
    out[C + c] = (var[c] + m * m) * n;
  }
  if (c == 0) out[2 * C] = n;
}

__global__ __launch_bounds__(256) void syncbn_unpack_kernel(const float* __restrict__ packed, int C, float* __restrict__ mean,
                                                            float* __restrict__ var, float* running_mean, float* running_var,
                                                            float momentum) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float total = packed[2 * C];
  const float m = packed[c] / total;
  float v = packed[C + c] / total - m * m;
  v = v > 0.f ? v : 0.f;
  mean[c] = m;
  var[c] = v;
  if (running_mean) {
    const float denom = total - 1.f > 1.f ? total - 1.f : 1.f;
    running_mean[c] = running_mean[c] * (1.f - momentum) + m * momentum;
    running_var[c] = running_var[c] * (1.f - momentum) + v * (momentum * total / denom);
  }
}

// pixel splits of the elementwise BN kernels: ~4096 blocks in total, >= 16 pixels per wave
int bn_ew_splits(int64_t P, int C) {
  const int cg = (C + 255) / 256;
  int64_t n = 4096 / cg;
  if (n > P / 64) n = P / 64;
  if (n < 1) n = 1;
  return (int)n;
}

int g_bn_wide = 1;     // A/B hook (gdl_debug_set_bn_wide): 0 = the 8-byte / 256-channel mapping for every shape

int bn_nsplit(int64_t P, int C) {
  // ~2048 blocks in total (8 per CU): these reductions are HBM-bound and need many waves in flight
  const int cg = (C + 255) / 256;
  int64_t n = P / 64;
  if (n < 1) n = 1;
  if (n > 2048 / cg) n = 2048 / cg;
  return (int)n;
}

}  // namespace

extern "C" void gdl_debug_set_bn_wide(int on) { g_bn_wide = on; }

extern "C" int gdl_layernorm_fwd(const float* x, int64_t x_stride, const float* gamma,
                                 const float* beta, void* y, int y_dtype, int64_t rows, int D, float eps,
                                 gdl_stream_t stream) {
  GDL_CHECK_ARG(x && gamma && beta && y, "gdl_layernorm_fwd: null pointer");
  GDL_CHECK_ARG(D > 0 && D % 4 == 0 && D <= 2048, "gdl_layernorm_fwd: D=%d must be a multiple of 4, <= 2048", D);
  GDL_CHECK_ARG(x_stride % 4 == 0, "gdl_layernorm_fwd: x_stride must be a multiple of 4");
  if (rows <= 0) return GDL_OK;
  hipStream_t s = (hipStream_t)stream;
  const int rpb = 4 * (D <= 64 ? 4 : (D <= 128 ? 2 : 1));   // rows per block: narrow rows share a wave
  const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
  const int vpl = (D + 255) / 256;
#define LN_LAUNCH(V) hipLaunchKernelGGL(layernorm_fwd_kernel<V>, dim3(grid), dim3(256), 0, s, x, x_stride, gamma, beta, y, y_dtype, rows, D, eps)
  if (vpl <= 1) LN_LAUNCH(1);
  else if (vpl <= 2) LN_LAUNCH(2);
  else if (vpl <= 4) LN_LAUNCH(4);
  else LN_LAUNCH(8);
#undef LN_LAUNCH
  GDL_CHECK_LAUNCH("gdl_layernorm_fwd");
  return GDL_OK;
}

extern "C" int64_t gdl_bn_stats_workspace(int64_t P, int C) {
  return (int64_t)bn_nsplit(P, C) * 2 * C * sizeof(float);
}

// One-launch train-mode BatchNorm(+ReLU) for small maps (see bn_small_fwd_kernel): statistics, running-estimate update and the
// normalised output.  y may alias x.  mean / var [C] are outputs (saved for the backward).
extern "C" int gdl_bn_small_fwd(const void* x, void* y, int dtype, int64_t P, int C, int64_t x_sP, int64_t y_sP, const float* gamma,
                                const float* beta, float eps, int relu, float* mean, float* var, float* running_mean,
                                float* running_var, float momentum, gdl_stream_t stream) {
  GDL_CHECK_ARG(x && y && gamma && beta && mean && var, "gdl_bn_small_fwd: null pointer");
  GDL_CHECK_ARG(dtype == GDL_F32 || dtype == GDL_BF16, "gdl_bn_small_fwd: bad dtype");
  GDL_CHECK_ARG(C % 4 == 0 && x_sP % 4 == 0 && y_sP % 4 == 0 && P > 0, "gdl_bn_small_fwd: C and strides must be multiples of 4");
  if (dtype == GDL_BF16)
    hipLaunchKernelGGL(bn_small_fwd_kernel<uint16_t>, dim3(C / 4), dim3(BNS_T), 0, (hipStream_t)stream, x, y, P, C, x_sP, y_sP, gamma, beta,
                       eps, relu, mean, var, running_mean, running_var, momentum);
  else
    hipLaunchKernelGGL(bn_small_fwd_kernel<float>, dim3(C / 4), dim3(BNS_T), 0, (hipStream_t)stream, x, y, P, C, x_sP, y_sP, gamma, beta,
                       eps, relu, mean, var, running_mean, running_var, momentum);
  GDL_CHECK_LAUNCH("gdl_bn_small_fwd");
  return GDL_OK;
}

// One-launch backward of the same: dgamma / dbeta [C] (f32 outputs) and dx (may alias x or dy).  Single-process statistics only
// (under SyncBatchNorm the sums cross the ranks between the two passes: gdl_bn_bwd_reduce + gdl_bn_bwd_dx).
extern "C" int gdl_bn_small_bwd(const void* x, const void* dy, void* dx, int dtype, int64_t P, int C, int64_t x_sP, int64_t dy_sP,
                                int64_t dx_sP, const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                                int relu, float* dgamma, float* dbeta, gdl_stream_t stream) {
  GDL_CHECK_ARG(x && dy && dx && mean && var && gamma && beta && dgamma && dbeta, "gdl_bn_small_bwd: null pointer");
  GDL_CHECK_ARG(dtype == GDL_F32 || dtype == GDL_BF16, "gdl_bn_small_bwd: bad dtype");
  GDL_CHECK_ARG(C % 4 == 0 && x_sP % 4 == 0 && dy_sP % 4 == 0 && dx_sP % 4 == 0 && P > 0, "gdl_bn_small_bwd: C and strides must be multiples of 4");
  if (dtype == GDL_BF16)
    hipLaunchKernelGGL(bn_small_bwd_kernel<uint16_t>, dim3(C / 4), dim3(BNS_T), 0, (hipStream_t)stream, x, dy, dx, P, C, x_sP, dy_sP, dx_sP,
                       mean, var, gamma, beta, eps, relu, dgamma, dbeta);
  else
    hipLaunchKernelGGL(bn_small_bwd_kernel<float>, dim3(C / 4), dim3(BNS_T), 0, (hipStream_t)stream, x, dy, dx, P, C, x_sP, dy_sP, dx_sP,
                       mean, var, gamma, beta, eps, relu, dgamma, dbeta);
  GDL_CHECK_LAUNCH("gdl_bn_small_bwd");
  return GDL_OK;
}

extern "C" int gdl_syncbn_pack(const float* mean, const float* var, double count, int C, float* out, gdl_stream_t stream) {
  GDL_CHECK_ARG(mean && var && out && C > 0 && count > 0, "gdl_syncbn_pack: bad arguments");
  hipLaunchKernelGGL(syncbn_pack_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean, var, (float)count, C, out);
  GDL_CHECK_LAUNCH("gdl_syncbn_pack");
  return GDL_OK;
}

extern "C" int gdl_syncbn_unpack(const float* packed, int C, float* mean, float* var, float* running_mean, float* running_var,
                                 float momentum, gdl_stream_t stream) {
  GDL_CHECK_ARG(packed && mean && var && C > 0 && (!running_mean == !running_var), "gdl_syncbn_unpack: bad arguments");
  hipLaunchKernelGGL(syncbn_unpack_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, packed, C, mean, var, running_mean,
                     running_var, momentum);
  GDL_CHECK_LAUNCH("gdl_syncbn_unpack");
  return GDL_OK;
}

extern "C" int gdl_bn_stats(const void* x, int dtype, int64_t P, int C, int64_t x_sP, float* mean,
                            float* var, float* running_mean, float* running_var, float momentum, float* ws,
                            int64_t ws_bytes, gdl_stream_t stream) {
  GDL_CHECK_ARG(x && mean && var && ws, "gdl_bn_stats: null pointer");
  GDL_CHECK_ARG(C % 4 == 0 && x_sP % 4 == 0 && P > 0, "gdl_bn_stats: C and stride must be multiples of 4");
  GDL_CHECK_ARG(ws_bytes >= gdl_bn_stats_workspace(P, C), "gdl_bn_stats: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int nsplit = bn_nsplit(P, C);
  dim3 grid((C + 255) / 256, nsplit);
  if (dtype == GDL_BF16) hipLaunchKernelGGL(bn_stats_partial<uint16_t>, grid, dim3(256), 0, s, x, P, C, x_sP, ws);
  else hipLaunchKernelGGL(bn_stats_partial<float>, grid, dim3(256), 0, s, x, P, C, x_sP, ws);
  hipLaunchKernelGGL(bn_stats_final, dim3((C + 3) / 4), dim3(BN_FINAL_T), 0, s, ws, nsplit, C, P, mean, var, running_mean, running_var, momentum);
  GDL_CHECK_LAUNCH("gdl_bn_stats");
  return GDL_OK;
}

// final reduction of per-block partial sums [nsplit][2][C] written by another kernel (gdl_resize_conv3x3_fwd_sum_bn): the
// second half of gdl_bn_stats
extern "C" int gdl_bn_stats_finalize(const float* ws, int nsplit, int C, int64_t P, float* mean, float* var, float* running_mean,
                                     float* running_var, float momentum, gdl_stream_t stream) {
  GDL_CHECK_ARG(ws && mean && var && nsplit > 0 && C > 0 && P > 0, "gdl_bn_stats_finalize: bad arguments");
  hipLaunchKernelGGL(bn_stats_final, dim3((C + 3) / 4), dim3(BN_FINAL_T), 0, (hipStream_t)stream, ws, nsplit, C, P, mean, var, running_mean,
                     running_var, momentum);
  GDL_CHECK_LAUNCH("gdl_bn_stats_finalize");
  return GDL_OK;
}

extern "C" int gdl_bn_apply(const void* x, void* y, int dtype, int64_t P, int C, int64_t x_sP, int64_t y_sP,
                            const float* mean, const float* var, const float* gamma, const float* beta,
                            float eps, int relu, gdl_stream_t stream) {
  GDL_CHECK_ARG(x && y && mean && var && gamma && beta, "gdl_bn_apply: null pointer");
  GDL_CHECK_ARG(C % 4 == 0 && x_sP % 4 == 0 && y_sP % 4 == 0, "gdl_bn_apply: C/strides must be multiples of 4");
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = P * (C / 4);
  (void)total;
  const dim3 grid((C + 255) / 256, bn_ew_splits(P, C));
  if (dtype == GDL_BF16)
    hipLaunchKernelGGL(bn_apply_kernel<uint16_t>, grid, dim3(256), 0, s, x, y, P, C, x_sP, y_sP, mean, var, gamma, beta, eps, relu);
  else
    hipLaunchKernelGGL(bn_apply_kernel<float>, grid, dim3(256), 0, s, x, y, P, C, x_sP, y_sP, mean, var, gamma, beta, eps, relu);
  GDL_CHECK_LAUNCH("gdl_bn_apply");
  return GDL_OK;
}

extern "C" int gdl_bn_bwd_reduce(const void* x, const void* dy, int dtype, int64_t P, int C, int64_t x_sP,
                                 int64_t dy_sP, const float* mean, const float* var, const float* gamma,
                                 const float* beta, float eps, int relu, float* dgamma, float* dbeta, float* ws,
                                 int64_t ws_bytes, gdl_stream_t stream) {
  GDL_CHECK_ARG(x && dy && mean && var && gamma && beta && dgamma && dbeta && ws, "gdl_bn_bwd_reduce: null pointer");
  GDL_CHECK_ARG(C % 4 == 0 && x_sP % 4 == 0 && dy_sP % 4 == 0, "gdl_bn_bwd_reduce: C/strides % 4");
  GDL_CHECK_ARG(ws_bytes >= gdl_bn_stats_workspace(P, C), "gdl_bn_bwd_reduce: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  int nsplit = bn_nsplit(P, C);
  dim3 grid((C + 255) / 256, nsplit);
  const int G = C / 8;
  const bool wide = g_bn_wide && dtype == GDL_BF16 && C % 8 == 0 && x_sP == C && dy_sP == C && (uintptr_t)x % 16 == 0 && (uintptr_t)dy % 16 == 0 &&
                    (256 % G == 0 || 192 % G == 0) && P >= 4096;
  if (wide) {     // one block = whole pixel rows; as many blocks as the workspace has rows (<= 2048, ~8 per CU)
    if (256 % G == 0)
      hipLaunchKernelGGL(bn_bwd_partial8<256>, dim3(nsplit), dim3(256), 0, s, (const uint16_t*)x, (const uint16_t*)dy, P, C, mean, var, gamma, beta, eps, relu, ws);
    else
      hipLaunchKernelGGL(bn_bwd_partial8<192>, dim3(nsplit), dim3(192), 0, s, (const uint16_t*)x, (const uint16_t*)dy, P, C, mean, var, gamma, beta, eps, relu, ws);
  } else if (dtype == GDL_BF16)
    hipLaunchKernelGGL(bn_bwd_partial<uint16_t>, grid, dim3(256), 0, s, x, dy, P, C, x_sP, dy_sP, mean, var, gamma, beta, eps, relu, ws);
  else
    hipLaunchKernelGGL(bn_bwd_partial<float>, grid, dim3(256), 0, s, x, dy, P, C, x_sP, dy_sP, mean, var, gamma, beta, eps, relu, ws);
  hipLaunchKernelGGL(bn_bwd_final, dim3((C + 3) / 4), dim3(BN_FINAL_T), 0, s, ws, nsplit, C, dgamma, dbeta);
  GDL_CHECK_LAUNCH("gdl_bn_bwd_reduce");
  return GDL_OK;
}

extern "C" int gdl_bn_bwd_dx(const void* x, const void* dy, void* dx, int dtype, int64_t P, int C, int64_t x_sP,
                             int64_t dy_sP, int64_t dx_sP, const float* mean, const float* var,
                             const float* gamma, const float* beta, float eps, int relu, const float* dgamma_sum,
                             const float* dbeta_sum, int64_t P_total, gdl_stream_t stream) {
  GDL_CHECK_ARG(x && dy && dx && mean && var && gamma && beta && dgamma_sum && dbeta_sum, "gdl_bn_bwd_dx: null pointer");
  GDL_CHECK_ARG(C % 4 == 0 && x_sP % 4 == 0 && dy_sP % 4 == 0 && dx_sP % 4 == 0 && P_total > 0, "gdl_bn_bwd_dx: C/strides % 4");
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = P * (C / 4);
  (void)total;
  const dim3 g2((C + 255) / 256, bn_ew_splits(P, C));
  if (dtype == GDL_BF16)
    hipLaunchKernelGGL(bn_bwd_dx<uint16_t>, g2, dim3(256), 0, s, x, dy, dx, P, P_total, C, x_sP, dy_sP, dx_sP, mean, var, gamma, beta, eps, relu, dgamma_sum, dbeta_sum, (const float*)nullptr);
  else
    hipLaunchKernelGGL(bn_bwd_dx<float>, g2, dim3(256), 0, s, x, dy, dx, P, P_total, C, x_sP, dy_sP, dx_sP, mean, var, gamma, beta, eps, relu, dgamma_sum, dbeta_sum, (const float*)nullptr);
  GDL_CHECK_LAUNCH("gdl_bn_bwd_dx");
  return GDL_OK;
}

// gdl_bn_bwd_dx for SyncBatchNorm: dgamma_sum / dbeta_sum are the all-reduced (global) sums and the global pixel count is read
// from device memory (total_count: one f32, the count entry of the forward's all-reduced message).
extern "C" int gdl_bn_bwd_dx_sync(const void* x, const void* dy, void* dx, int dtype, int64_t P, int C, int64_t x_sP, int64_t dy_sP,
                                  int64_t dx_sP, const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                                  int relu, const float* dgamma_sum, const float* dbeta_sum, const float* total_count,
                                  gdl_stream_t stream) {
  GDL_CHECK_ARG(x && dy && dx && mean && var && gamma && beta && dgamma_sum && dbeta_sum && total_count, "gdl_bn_bwd_dx_sync: null pointer");
  GDL_CHECK_ARG(dtype == GDL_F32 || dtype == GDL_BF16, "gdl_bn_bwd_dx_sync: bad dtype");
  GDL_CHECK_ARG(C % 4 == 0 && x_sP % 4 == 0 && dy_sP % 4 == 0 && dx_sP % 4 == 0 && P > 0, "gdl_bn_bwd_dx_sync: C/strides % 4");
  hipStream_t s = (hipStream_t)stream;
  const dim3 g2((C + 255) / 256, bn_ew_splits(P, C));
  if (dtype == GDL_BF16)
    hipLaunchKernelGGL(bn_bwd_dx<uint16_t>, g2, dim3(256), 0, s, x, dy, dx, P, (int64_t)1, C, x_sP, dy_sP, dx_sP, mean, var, gamma, beta, eps, relu, dgamma_sum, dbeta_sum, total_count);
  else
    hipLaunchKernelGGL(bn_bwd_dx<float>, g2, dim3(256), 0, s, x, dy, dx, P, (int64_t)1, C, x_sP, dy_sP, dx_sP, mean, var, gamma, beta, eps, relu, dgamma_sum, dbeta_sum, total_count);
  GDL_CHECK_LAUNCH("gdl_bn_bwd_dx_sync");
  return GDL_OK;
}
