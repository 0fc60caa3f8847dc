// Multi-tensor optimizer kernels: ONE launch covers every parameter tensor of the step.
// The host passes a device table of chunks; row = {param, grad, exp_avg, exp_avg_sq, count, shadow} with the
// pointers already offset to the chunk (<= 65536 elements each), all f32, dense.  shadow (0 = none) = a bf16 copy of the
// parameter in the same element order (the GEMM operand the next forward reads): the update writes it too, so the step has
// no per-parameter cast launches (28 per DOFA + UperNet step).
#include "gdl_common.h"

namespace {

constexpr int ROW = 6;

__global__ __launch_bounds__(256) void multi_sumsq_kernel(const int64_t* __restrict__ table, float* __restrict__ acc) {
  __shared__ float red[4];
  const int64_t* row = table + (int64_t)blockIdx.x * ROW;
  const float* g = (const float*)row[1];
  const int n = (int)row[4];
  float s = 0.f;
  const int n4 = n >> 2;
  for (int i = threadIdx.x; i < n4; i += 256) {
    const float4 v = ((const float4*)g)[i];
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, (red[0] + red[1]) + (red[2] + red[3]));
}

__global__ __launch_bounds__(256) void multi_adam_kernel(const int64_t* __restrict__ table, float lr, float b1, float b2,
                                                         float eps, float wd, float bc1, float bc2,
                                                         const float* __restrict__ clip_coef) {
  const int64_t* row = table + (int64_t)blockIdx.x * ROW;
  float* p = (float*)row[0];
  const float* g = (const float*)row[1];
  float* m = (float*)row[2];
  float* v = (float*)row[3];
  const int n = (int)row[4];
  uint16_t* sh = (uint16_t*)row[5];
  const float cc = clip_coef ? clip_coef[0] : 1.f;
  const float step = lr / bc1, rs = 1.f / sqrtf(bc2);
  for (int i = threadIdx.x; i < n; i += 256) {
    float gi = g[i] * cc;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float pn = pi - step * mi / (sqrtf(vi) * rs + eps);
    p[i] = pn;
    if (sh) sh[i] = f32_to_bf16(pn);
  }
}

// hipGraph-capturable form: the step count and the hyper-parameters live in DEVICE memory, so that a captured optimizer step
// replays with the right bias corrections and a learning rate the host may rewrite between replays.
// state = {step, lr, beta1, beta2, eps, weight_decay, bc1, bc2} (f32)
__global__ void adam_tick_kernel(float* __restrict__ st, double b1, double b2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float step = st[0] + 1.f;
    st[0] = step;
    // in double from the optimizer's own (double) betas, like the host form's `1 - beta ** step`: the f32 results are the
    // same bits, so a captured step and an eager step update identically (an f32 powf differs in the last bit, 1e-7 of a
    // parameter after one step -- which a random-init network with train-mode BatchNorm amplifies to O(lr) within two steps)
    st[6] = (float)(1.0 - pow(b1, (double)step));
    st[7] = (float)(1.0 - pow(b2, (double)step));
  }
}

__global__ __launch_bounds__(256) void multi_adam_dev_kernel(const int64_t* __restrict__ table, const float* __restrict__ st,
                                                             const float* __restrict__ clip_coef) {
  const int64_t* row = table + (int64_t)blockIdx.x * ROW;
  float* p = (float*)row[0];
  const float* g = (const float*)row[1];
  float* m = (float*)row[2];
  float* v = (float*)row[3];
  const int n = (int)row[4];
  uint16_t* sh = (uint16_t*)row[5];
  const float lr = st[1], b1 = st[2], b2 = st[3], eps = st[4], wd = st[5], bc1 = st[6], bc2 = st[7];
  const float cc = clip_coef ? clip_coef[0] : 1.f;
  const float step = lr / bc1, rs = 1.f / sqrtf(bc2);
  for (int i = threadIdx.x; i < n; i += 256) {
    float gi = g[i] * cc;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float pn = pi - step * mi / (sqrtf(vi) * rs + eps);
    p[i] = pn;
    if (sh) sh[i] = f32_to_bf16(pn);
  }
}

// Derived GEMM operands of the conv parameters, rebuilt from the freshly updated f32 parameters in ONE launch (the fused optimizer
// calls it right behind gdl_multi_adam): before, every trainable 3x3 layer cost 2-3 launches per step between the update and its
// next use -- a strided copy + a cast for the tap-major / channel-slice operands, a transpose for the data-gradient operand: 39
// launches of 5-12 us per DOFA training step, 3 % of the step at the reference's per-GPU batch 4.
// table rows (int64 x 10): {src f32 [N][T][C], dst bf16, N, T, C, c0, Cs, mode, first tile, tiles along c}; a tile is 32 output
// channels x 32 input channels of one tap.  mode 0: dst[n][t][c - c0]   (channel slice c0 .. c0 + Cs of a 3x3 parameter)
//                                        mode 1: dst[t][n][c - c0]   (tap-major: the nine tap products as one 1x1 convolution)
//                                        mode 2: dst[c][T-1-t][n]    (data gradient: transposed, taps flipped = gdl_pack_dgrad)
// Values are rounded exactly like gdl_cast / gdl_pack_dgrad round them: results are bit-identical to the separate launches.
__global__ __launch_bounds__(256) void multi_repack_kernel(const int64_t* __restrict__ table, int rows) {
  __shared__ float tile[32][33];
  const int64_t b = blockIdx.x;
  int r = 0;
  while (r + 1 < rows && table[(r + 1) * 10 + 8] <= b) ++r;     // (uniform: scalar loads; rows <= a few dozen)
  const int64_t* row = table + r * 10;
  const float* src = (const float*)row[0];
  uint16_t* dst = (uint16_t*)row[1];
  const int N = (int)row[2], T = (int)row[3], C = (int)row[4], c0 = (int)row[5], Cs = (int)row[6], mode = (int)row[7];
  const int tiles_c = (int)row[9], tiles_n = (N + 31) / 32;
  int local = (int)(b - row[8]);
  const int t = local / (tiles_n * tiles_c);
  local -= t * tiles_n * tiles_c;
  const int n0 = (local / tiles_c) * 32, cs0 = (local % tiles_c) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  if (mode != 2) {
    const int cs = cs0 + tx;
    for (int j = ty; j < 32; j += 8) {
      const int n = n0 + j;
      if (n < N && cs < Cs) {
        const float v = src[((int64_t)n * T + t) * C + c0 + cs];
        const int64_t o = mode == 0 ? ((int64_t)n * T + t) * Cs + cs : ((int64_t)t * N + n) * Cs + cs;
        dst[o] = f32_to_bf16(v);
      }
    }
    return;
  }
  for (int j = ty; j < 32; j += 8) {
    const int n = n0 + j, cs = cs0 + tx;
    tile[j][tx] = (n < N && cs < Cs) ? src[((int64_t)n * T + t) * C + c0 + cs] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int cs = cs0 + j, n = n0 + tx;
    if (cs < Cs && n < N) dst[((int64_t)cs * T + (T - 1 - t)) * N + n] = f32_to_bf16(tile[tx][j]);
  }
}

}  // namespace

extern "C" int gdl_multi_repack(const int64_t* table, int rows, int64_t total_tiles, gdl_stream_t stream) {
  GDL_CHECK_ARG(table && rows >= 0 && total_tiles >= 0 && total_tiles < (1ll << 31), "gdl_multi_repack: bad args");
  if (rows == 0 || total_tiles == 0) return GDL_OK;
  hipLaunchKernelGGL(multi_repack_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, table, rows);
  GDL_CHECK_LAUNCH("gdl_multi_repack");
  return GDL_OK;
}

extern "C" int gdl_adam_tick(float* state, double beta1, double beta2, gdl_stream_t stream) {
  GDL_CHECK_ARG(state, "gdl_adam_tick: null state");
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, beta1, beta2);
  GDL_CHECK_LAUNCH("gdl_adam_tick");
  return GDL_OK;
}

extern "C" int gdl_multi_adam_dev(const int64_t* table, int nchunks, const float* state, const float* clip_coef, gdl_stream_t stream) {
  GDL_CHECK_ARG(table && state && nchunks >= 0, "gdl_multi_adam_dev: bad args");
  if (nchunks == 0) return GDL_OK;
  hipLaunchKernelGGL(multi_adam_dev_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, table, state, clip_coef);
  GDL_CHECK_LAUNCH("gdl_multi_adam_dev");
  return GDL_OK;
}

extern "C" int gdl_multi_sumsq(const int64_t* table, int nchunks, float* acc, gdl_stream_t stream) {
  GDL_CHECK_ARG(table && acc && nchunks >= 0, "gdl_multi_sumsq: bad args");
  if (nchunks == 0) return GDL_OK;
  hipLaunchKernelGGL(multi_sumsq_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, table, acc);
  GDL_CHECK_LAUNCH("gdl_multi_sumsq");
  return GDL_OK;
}

extern "C" int gdl_multi_adam(const int64_t* table, int nchunks, float lr, float beta1, float beta2, float eps,
                              float weight_decay, float bc1, float bc2, const float* clip_coef, gdl_stream_t stream) {
  GDL_CHECK_ARG(table && nchunks >= 0, "gdl_multi_adam: bad args");
  if (nchunks == 0) return GDL_OK;
  hipLaunchKernelGGL(multi_adam_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, table, lr, beta1, beta2, eps,
                     weight_decay, bc1, bc2, clip_coef);
  GDL_CHECK_LAUNCH("gdl_multi_adam");
  return GDL_OK;
}
