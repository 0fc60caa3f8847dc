// Backward kernels of the transformer blocks (timm ViT Block / SegFormer MiT Block): all HBM-bound,
// f32 arithmetic, wave-shuffle row reductions; column reductions go through per-block partials in a
// workspace + a deterministic (double) final pass -- no float atomics anywhere.
#include "dwconv_walk.h"

namespace {

// out[i] (+)= sum_blk ws[blk*total + i]; 32 columns x 8 block-groups per 256-thread block
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ ws, int nblk, int64_t stride,
                                                              int total, float* __restrict__ out, int accumulate) {
  // block = 8 columns x 32 partial groups
  __shared__ double part[32][8];
  const int cl = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const int i = blockIdx.x * 8 + cl;
  double s = 0;
  if (i < total) s = ordered_sum8<double>(grp, nblk, 32, [&](int b) { return ws[(int64_t)b * stride + i]; });
  part[grp][cl] = s;
  __syncthreads();
  if (grp != 0 || i >= total) return;
  s = 0;
#pragma unroll
  for (int g = 0; g < 32; ++g) s += part[g][cl];
  out[i] = accumulate ? out[i] + (float)s : (float)s;
}

inline void launch_reduce(const float* ws, int nblk, int64_t stride, int total, float* out, int accumulate,
                          hipStream_t s) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((total + 7) / 8), dim3(256), 0, s, ws, nblk, stride, total, out,
                     accumulate);
}

inline int row_blocks(int64_t rows) {
  int64_t n = rows / 16;
  if (n < 1) n = 1;
  if (n > 512) n = 512;
  return (int)n;
}

// ---------------------------------------------------------------- LayerNorm backward
// One wave per row (row kept in registers), a block's 4 waves stride over the rows of its slice.
//   xhat = (x - mu) * rstd ; g = dy * gamma ; dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) (+ dres)
//   dgamma = sum_rows dy * xhat ; dbeta = sum_rows dy      -> partials ws[blk][2][D]
template <typename TDY, int VPL>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, int64_t x_stride,
                                                            const void* __restrict__ dy, const float* __restrict__ gamma,
                                                            const float* __restrict__ dres, int64_t dres_stride,
                                                            float* __restrict__ dx, int64_t dx_stride, int64_t rows, int D,
                                                            float eps, float* __restrict__ ws) {
  __shared__ float red[2][4][64 * 4 * VPL];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // narrow rows (D <= 128, MiT stages 1-2): 16 or 32 lanes hold a row and a wave normalises 4 or 2 rows at once
  const int lpp = (VPL == 1 && D <= 128) ? (D <= 64 ? 16 : 32) : 64;
  const int nsub = 64 / lpp, sub = lane / lpp, cl = lane % lpp;
  auto group_sum = [&](float v) {
    for (int o = lpp >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  float4 pg[VPL], pb[VPL], gm[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    pg[i] = make_float4(0, 0, 0, 0); pb[i] = make_float4(0, 0, 0, 0);
    const int c = (i * 64 + cl) * 4;
    gm[i] = c < D ? *(const float4*)(gamma + c) : make_float4(0, 0, 0, 0);
  }
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = per * blockIdx.x, r1 = r0 + per < rows ? r0 + per : rows;
  for (int64_t row = r0 + w * nsub + sub; row < r1; row += 4 * nsub) {
    const float* xr = x + row * x_stride;
    float4 v[VPL], d[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (i * 64 + cl) * 4;
      if (c < D) {
        v[i] = *(const float4*)(xr + c);
        if constexpr (sizeof(TDY) == 4) {
          d[i] = *(const float4*)((const float*)dy + row * D + c);
        } else {
          const uint2 t = *(const uint2*)((const uint16_t*)dy + row * D + c);
          d[i] = make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u),
                             __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u));
        }
      } else {
        v[i] = make_float4(0, 0, 0, 0); d[i] = make_float4(0, 0, 0, 0);
      }
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = group_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (i * 64 + cl) * 4;
      if (c < D) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
      }
    }
    const float rstd = rsqrtf(group_sum(q) / (float)D + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;   // xhat
      pg[i].x += d[i].x * v[i].x; pg[i].y += d[i].y * v[i].y; pg[i].z += d[i].z * v[i].z; pg[i].w += d[i].w * v[i].w;
      pb[i].x += d[i].x; pb[i].y += d[i].y; pb[i].z += d[i].z; pb[i].w += d[i].w;
      d[i].x *= gm[i].x; d[i].y *= gm[i].y; d[i].z *= gm[i].z; d[i].w *= gm[i].w;  // g
      sg += (d[i].x + d[i].y) + (d[i].z + d[i].w);
      sgx += (d[i].x * v[i].x + d[i].y * v[i].y) + (d[i].z * v[i].z + d[i].w * v[i].w);
    }
    const float mg = group_sum(sg) / (float)D, mgx = group_sum(sgx) / (float)D;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (i * 64 + cl) * 4;
      if (c >= D) continue;
      float4 o;
      o.x = rstd * (d[i].x - mg - v[i].x * mgx); o.y = rstd * (d[i].y - mg - v[i].y * mgx);
      o.z = rstd * (d[i].z - mg - v[i].z * mgx); o.w = rstd * (d[i].w - mg - v[i].w * mgx);
      if (dres) {
        const float4 r = *(const float4*)(dres + row * dres_stride + c);
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
      *(float4*)(dx + row * dx_stride + c) = o;
    }
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    *(float4*)&red[0][w][(i * 64 + lane) * 4] = pg[i];
    *(float4*)&red[1][w][(i * 64 + lane) * 4] = pb[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    float sg = 0.f, sb = 0.f;
    for (int q = 0; q < nsub; ++q) {
      const int i = nsub == 1 ? c : (q * lpp + (c >> 2)) * 4 + (c & 3);
      sg += (red[0][0][i] + red[0][1][i]) + (red[0][2][i] + red[0][3][i]);
      sb += (red[1][0][i] + red[1][1][i]) + (red[1][2][i] + red[1][3][i]);
    }
    ws[((int64_t)blockIdx.x * 2 + 0) * D + c] = sg;
    ws[((int64_t)blockIdx.x * 2 + 1) * D + c] = sb;
  }
}

// Lane -> (4-channel group, row sub-index) for the column reductions below: 256 channels of one row per wave, or, on
// narrow tensors (C <= 128), 4 or 2 rows side by side so that no lane idles.
struct ColMap { int c, sub, nsub, lpp; };
__device__ __forceinline__ ColMap col_map(int lane, int C, int block_x) {
  const int lpp = C > 128 ? 64 : (C > 64 ? 32 : (C > 32 ? 16 : (C > 16 ? 8 : 4)));
  ColMap m;
  m.lpp = lpp; m.nsub = 64 / lpp; m.sub = lane / lpp; m.c = block_x * 256 + (lane % lpp) * 4;
  return m;
}
// sum over the 4 waves and the nsub row sub-indices of channel t's partials in red[4][256] (lane-major, 4 per lane)
__device__ __forceinline__ float col_total(const float (&red)[4][256], const ColMap& m, int t) {
  float s = 0.f;
  for (int q = 0; q < m.nsub; ++q) {
    const int i = (q * m.lpp + (t >> 2)) * 4 + (t & 3);
    s += (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  }
  return s;
}

// ---------------------------------------------------------------- column sums (bias gradients)
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const void* __restrict__ x, int64_t rows, int C,
                                                             int64_t x_stride, float* __restrict__ ws) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const ColMap cm = col_map(lane, C, blockIdx.x);
  const int c = cm.c;
  const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = per * blockIdx.y, r1 = r0 + per < rows ? r0 + per : rows;
  float s[4] = {0, 0, 0, 0};
  if (c < C)
    for (int64_t r = r0 + w * cm.nsub + cm.sub; r < r1; r += 4 * cm.nsub) {
      if constexpr (sizeof(T) == 4) {
        const float4 v = *(const float4*)((const float*)x + r * x_stride + c);
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
      } else {
        const uint2 t = *(const uint2*)((const uint16_t*)x + r * x_stride + c);
        s[0] += __uint_as_float(t.x << 16); s[1] += __uint_as_float(t.x & 0xffff0000u);
        s[2] += __uint_as_float(t.y << 16); s[3] += __uint_as_float(t.y & 0xffff0000u);
      }
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[w][lane * 4 + j] = s[j];
  __syncthreads();
  const int t = threadIdx.x, cc = blockIdx.x * 256 + t;
  if (cc < C) ws[(int64_t)blockIdx.y * C + cc] = col_total(red, cm, t);
}

// ---------------------------------------------------------------- LayerScale / DropPath backward
// forward: y = x + s[b] * gamma[c] * z   (z = branch output before LayerScale)
// dz[b,n,c] = g * s[b] * gamma[c]  (compute dtype) ; dgamma[c] = sum g * s[b] * z  -> partials ws[blk][C]
template <typename TZ, typename TD>
__global__ __launch_bounds__(256) void layerscale_bwd_kernel(const float* __restrict__ g, const void* __restrict__ z,
                                                             const float* __restrict__ gamma, const float* __restrict__ s,
                                                             int64_t rows, int64_t rows_per_batch, int C, void* dz,
                                                             float* __restrict__ ws) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const ColMap cm = col_map(lane, C, blockIdx.x);
  const int c = cm.c;
  const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = per * blockIdx.y, r1 = r0 + per < rows ? r0 + per : rows;
  float acc[4] = {0, 0, 0, 0};
  if (c < C) {
    float gm[4] = {1.f, 1.f, 1.f, 1.f};
    if (gamma) { const float4 t = *(const float4*)(gamma + c); gm[0] = t.x; gm[1] = t.y; gm[2] = t.z; gm[3] = t.w; }
    for (int64_t r = r0 + w * cm.nsub + cm.sub; r < r1; r += 4 * cm.nsub) {
      const float sb = s ? s[r / rows_per_batch] : 1.f;
      const float4 gv = *(const float4*)(g + r * C + c);
      const float gg[4] = {gv.x * sb, gv.y * sb, gv.z * sb, gv.w * sb};
      if (gamma && z) {
        float zv[4];
        gdldw::Px<TZ>::cvt(gdldw::Px<TZ>::ld(z, r * C + c), zv);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += gg[j] * zv[j];
      }
      const float o[4] = {gg[0] * gm[0], gg[1] * gm[1], gg[2] * gm[2], gg[3] * gm[3]};
      gdldw::Px<TD>::st(dz, r * C + c, o);
    }
  }
  if (ws) {
#pragma unroll
    for (int j = 0; j < 4; ++j) red[w][lane * 4 + j] = acc[j];
    __syncthreads();
    const int t = threadIdx.x, cc = blockIdx.x * 256 + t;
    if (cc < C) ws[(int64_t)blockIdx.y * C + cc] = col_total(red, cm, t);
  }
}

// ---------------------------------------------------------------- softmax backward (rows)
// dS = P * (dP - sum_k dP*P) * scale over the first n_valid columns; pad columns -> 0.  In place on dP ok.
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const void* __restrict__ p, const void* __restrict__ dp,
                                                               void* ds, int64_t rows, int n_valid, int n_cols,
                                                               float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t base = row * n_cols;
  float pv[MAXV], dv[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = i * 64 + lane;
    if (c < n_valid) { pv[i] = ElemIO<T>::load(p, base + c); dv[i] = ElemIO<T>::load(dp, base + c); }
    else { pv[i] = 0.f; dv[i] = 0.f; }
    s += pv[i] * dv[i];
  }
  s = wave_sum(s);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = i * 64 + lane;
    if (c < n_cols) ElemIO<T>::store(ds, base + c, pv[i] * (dv[i] - s) * scale);
  }
}

// ---------------------------------------------------------------- depthwise 3x3 (+GELU) backward
// forward: pre = dw3x3(u) + b ; y = gelu(pre).  One pass over u and dy with the 3x3 window of u in registers
// (dwconv_walk.h):  dpre = dy * gelu'(pre) [pre recomputed], written once, and in the same step
//   dw9[t][c] += dpre[p,c] * u[p + tap t, c] ; db[c] += dpre[p,c]      -> partials ws[split][10][C]
// Narrow tensors (C <= 128) put 4 or 2 row segments side by side in a wave.
template <typename T>
__global__ __launch_bounds__(256) void dwconv_gelu_bwd_kernel(const void* __restrict__ u, const void* __restrict__ dy,
                                                              int H, int W, int C, const float* __restrict__ w9,
                                                              const float* __restrict__ bias, void* __restrict__ dpre,
                                                              float* __restrict__ ws, int seglen, int nseg,
                                                              int64_t items) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lpp = C > 128 ? 64 : (C > 64 ? 32 : 16), nsub = 64 / lpp, sub = lane / lpp;
  const int c = blockIdx.x * 256 + (lane % lpp) * 4;
  const int64_t per = (items + gridDim.y - 1) / gridDim.y;
  const int64_t i0 = per * blockIdx.y, i1 = i0 + per < items ? i0 + per : items;
  float acc[10][4];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f;
  if (c < C) {
    gdldw::Taps tp;
    tp.load(w9, bias, C, c);
    for (int64_t item = i0 + wv * nsub + sub; item < i1; item += 4 * nsub) {
      const gdldw::Seg sg = gdldw::seg_of(item, H, W, seglen, nseg);
      gdldw::walk<T>(u, sg, H, W, C, c, [&](int x, const float (&L)[3][4], const float (&M)[3][4], const float (&R)[3][4]) {
        const int64_t off = (sg.pix + x) * C + c;
        const typename gdldw::Px<T>::raw graw = gdldw::Px<T>::ld(dy, off);
        float pre[4], g[4], d[4];
        tp.apply(L, M, R, pre);
        gdldw::Px<T>::cvt(graw, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = g[j] * gelu_erf_grad(pre[j]);
        gdldw::Px<T>::st(dpre, off, d);
        if (sizeof(typename gdldw::Px<T>::raw) == 8) {       // the weight gradient sees dpre as stored (bf16)
          d[0] = __uint_as_float(pack_bf16x2(0.f, d[0]) & 0xffff0000u); d[1] = __uint_as_float(pack_bf16x2(0.f, d[1]) & 0xffff0000u);
          d[2] = __uint_as_float(pack_bf16x2(0.f, d[2]) & 0xffff0000u); d[3] = __uint_as_float(pack_bf16x2(0.f, d[3]) & 0xffff0000u);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[r * 3 + 0][j] += d[j] * L[r][j];
            acc[r * 3 + 1][j] += d[j] * M[r][j];
            acc[r * 3 + 2][j] += d[j] * R[r][j];
          }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[9][j] += d[j];
      });
    }
  }
  float* wsb = ws + (int64_t)blockIdx.y * 10 * C;
  for (int t = 0; t < 10; ++t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wv][lane * 4 + j] = acc[t][j];
    __syncthreads();
    const int tt = threadIdx.x, cc = blockIdx.x * 256 + tt;
    if (cc < C) {
      float sum = 0.f;
      for (int sb = 0; sb < nsub; ++sb) {
        const int i = (sb * lpp + (tt >> 2)) * 4 + (tt & 3);
        sum += (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
      }
      wsb[(int64_t)t * C + cc] = sum;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- col2im (data gradient of strided convs)
// cols[(b,oy,ox)][(r,s,c)] = dy . W   (a plain GEMM);   dx[b,y,x,c] = sum of the taps that touch (y,x)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void col2im_kernel(const void* __restrict__ cols, int B, int Ho, int Wo, int R, int S,
                                                     int C, int stride, int pad, int H, int W, void* dx, int64_t sB,
                                                     int64_t sH, int64_t sW) {
  const int cv = C / 4;
  const int64_t total = (int64_t)B * H * W * cv;
  const int64_t ldc = (int64_t)R * S * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cv) * 4;
    int64_t t = i / cv;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    float acc[4] = {0, 0, 0, 0};
    for (int r = 0; r < R; ++r) {
      const int ty = y + pad - r;
      if (ty < 0 || ty % stride) continue;
      const int oy = ty / stride;
      if (oy >= Ho) continue;
      for (int q = 0; q < S; ++q) {
        const int tx = x + pad - q;
        if (tx < 0 || tx % stride) continue;
        const int ox = tx / stride;
        if (ox >= Wo) continue;
        const int64_t off = (((int64_t)b * Ho + oy) * Wo + ox) * ldc + (int64_t)(r * S + q) * C + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += ElemIO<TI>::load(cols, off + j);
      }
    }
    const int64_t o = (int64_t)b * sB + (int64_t)y * sH + (int64_t)x * sW + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) ElemIO<TO>::store(dx, o + j, acc[j]);
  }
}

}  // namespace

// ================================================================================ C ABI
extern "C" int64_t gdl_colreduce_workspace(int64_t rows, int C, int planes) {
  return (int64_t)row_blocks(rows) * planes * C * (int64_t)sizeof(float);
}

extern "C" int gdl_layernorm_bwd(const float* x, int64_t x_stride, const void* dy, int dy_dtype, const float* gamma,
                                 const float* dres, int64_t dres_stride, float* dx, int64_t dx_stride, int64_t rows,
                                 int D, float eps, float* dgamma, float* dbeta, int accumulate_params, float* ws,
                                 int64_t ws_bytes, gdl_stream_t stream) {
  GDL_CHECK_ARG(x && dy && gamma && dx && dgamma && dbeta && ws, "gdl_layernorm_bwd: null pointer");
  GDL_CHECK_ARG(D > 0 && D % 4 == 0 && D <= 1024 && x_stride % 4 == 0 && dx_stride % 4 == 0 && dres_stride % 4 == 0,
                "gdl_layernorm_bwd: D must be a multiple of 4 and <= 1024, strides multiples of 4 (D=%d)", D);
  GDL_CHECK_ARG(ws_bytes >= gdl_colreduce_workspace(rows, D, 2), "gdl_layernorm_bwd: workspace too small");
  if (rows <= 0) return GDL_OK;
  hipStream_t s = (hipStream_t)stream;
  const int nblk = row_blocks(rows);
  const int vpl = (D + 255) / 256;
#define LNB(T, V)                                                                                                   \
  hipLaunchKernelGGL((layernorm_bwd_kernel<T, V>), dim3(nblk), dim3(256), 0, s, x, x_stride, dy, gamma, dres,       \
                     dres_stride, dx, dx_stride, rows, D, eps, ws)
  if (dy_dtype == GDL_BF16) {
    if (vpl <= 1) LNB(uint16_t, 1); else if (vpl <= 2) LNB(uint16_t, 2); else LNB(uint16_t, 4);
  } else {
    if (vpl <= 1) LNB(float, 1); else if (vpl <= 2) LNB(float, 2); else LNB(float, 4);
  }
#undef LNB
  launch_reduce(ws, nblk, 2 * (int64_t)D, D, dgamma, accumulate_params, s);
  launch_reduce(ws + D, nblk, 2 * (int64_t)D, D, dbeta, accumulate_params, s);
  GDL_CHECK_LAUNCH("gdl_layernorm_bwd");
  return GDL_OK;
}

extern "C" int gdl_colsum(const void* x, int dtype, int64_t rows, int C, int64_t x_stride, float* out, int accumulate,
                          float* ws, int64_t ws_bytes, gdl_stream_t stream) {
  GDL_CHECK_ARG(x && out && ws, "gdl_colsum: null pointer");
  GDL_CHECK_ARG(C > 0 && C % 4 == 0 && x_stride % 4 == 0, "gdl_colsum: C and row stride must be multiples of 4");
  GDL_CHECK_ARG(ws_bytes >= gdl_colreduce_workspace(rows, C, 1), "gdl_colsum: workspace too small");
  if (rows <= 0) return GDL_OK;
  hipStream_t s = (hipStream_t)stream;
  const int nblk = row_blocks(rows);
  const dim3 grid((C + 255) / 256, nblk);
  if (dtype == GDL_BF16) hipLaunchKernelGGL(colsum_partial_kernel<uint16_t>, grid, dim3(256), 0, s, x, rows, C, x_stride, ws);
  else hipLaunchKernelGGL(colsum_partial_kernel<float>, grid, dim3(256), 0, s, x, rows, C, x_stride, ws);
  launch_reduce(ws, nblk, C, C, out, accumulate, s);
  GDL_CHECK_LAUNCH("gdl_colsum");
  return GDL_OK;
}

extern "C" int gdl_layerscale_bwd(const float* g, const void* z, int z_dtype, const float* gamma,
                                  const float* batch_scale, int64_t rows, int64_t rows_per_batch, int C, void* dz,
                                  int dz_dtype, float* dgamma, int accumulate, float* ws, int64_t ws_bytes,
                                  gdl_stream_t stream) {
  GDL_CHECK_ARG(g && dz, "gdl_layerscale_bwd: null pointer");
  GDL_CHECK_ARG(C > 0 && C % 4 == 0 && rows_per_batch > 0, "gdl_layerscale_bwd: C must be a multiple of 4");
  GDL_CHECK_ARG(!gamma || (z && dgamma && ws && ws_bytes >= gdl_colreduce_workspace(rows, C, 1)),
                "gdl_layerscale_bwd: gamma needs z, dgamma and a workspace");
  if (rows <= 0) return GDL_OK;
  hipStream_t s = (hipStream_t)stream;
  const int nblk = row_blocks(rows);
  const dim3 grid((C + 255) / 256, nblk);
  float* wsp = gamma ? ws : nullptr;
#define LSB(TZ, TD) hipLaunchKernelGGL((layerscale_bwd_kernel<TZ, TD>), grid, dim3(256), 0, s, g, z, gamma, batch_scale, rows, rows_per_batch, C, dz, wsp)
  if (z_dtype == GDL_BF16) { if (dz_dtype == GDL_BF16) LSB(bf16_tag, bf16_tag); else LSB(bf16_tag, float); }
  else { if (dz_dtype == GDL_BF16) LSB(float, bf16_tag); else LSB(float, float); }
#undef LSB
  if (gamma) launch_reduce(ws, nblk, C, C, dgamma, accumulate, s);
  GDL_CHECK_LAUNCH("gdl_layerscale_bwd");
  return GDL_OK;
}

extern "C" int gdl_softmax_bwd_rows(const void* p, const void* dp, void* ds, int dtype, int64_t rows, int n_valid,
                                    int n_cols, float scale, gdl_stream_t stream) {
  GDL_CHECK_ARG(p && dp && ds, "gdl_softmax_bwd_rows: null pointer");
  GDL_CHECK_ARG(n_valid > 0 && n_valid <= n_cols && n_cols <= 64 * 32, "gdl_softmax_bwd_rows: n_cols must be <= 2048");
  if (rows <= 0) return GDL_OK;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((rows + 3) / 4));
  const int mv = (n_cols + 63) / 64;
#define SMB(T, V) hipLaunchKernelGGL((softmax_bwd_rows_kernel<T, V>), grid, dim3(256), 0, s, p, dp, ds, rows, n_valid, n_cols, scale)
  if (dtype == GDL_BF16) {
    if (mv <= 4) SMB(bf16_tag, 4); else if (mv <= 8) SMB(bf16_tag, 8); else if (mv <= 16) SMB(bf16_tag, 16); else SMB(bf16_tag, 32);
  } else {
    if (mv <= 4) SMB(float, 4); else if (mv <= 8) SMB(float, 8); else if (mv <= 16) SMB(float, 16); else SMB(float, 32);
  }
#undef SMB
  GDL_CHECK_LAUNCH("gdl_softmax_bwd_rows");
  return GDL_OK;
}

extern "C" int gdl_dwconv3x3_gelu_bwd(const void* u, const void* dy, int dtype, int B, int H, int W, int C,
                                      const float* w9, const float* bias, void* dpre, float* dw9, float* dbias,
                                      int accumulate, float* ws, int64_t ws_bytes, gdl_stream_t stream) {
  GDL_CHECK_ARG(u && dy && w9 && bias && dpre && dw9 && dbias && ws, "gdl_dwconv3x3_gelu_bwd: null pointer");
  GDL_CHECK_ARG(C > 0 && C % 4 == 0, "gdl_dwconv3x3_gelu_bwd: C must be a multiple of 4");
  const int64_t P = (int64_t)B * H * W;
  GDL_CHECK_ARG(ws_bytes >= gdl_colreduce_workspace(P, C, 10), "gdl_dwconv3x3_gelu_bwd: workspace too small");
  if (P <= 0) return GDL_OK;
  hipStream_t s = (hipStream_t)stream;
  const int seglen = gdldw::seg_len(W), nseg = (W + seglen - 1) / seglen;
  const int64_t items = (int64_t)B * nseg * H;
  const int nsub = C > 128 ? 1 : (C > 64 ? 2 : 4);
  int nblk = row_blocks(P);                                   // the workspace holds row_blocks(P) partials
  if (nblk > (items + 4 * nsub - 1) / (4 * nsub)) nblk = (int)((items + 4 * nsub - 1) / (4 * nsub));
  const dim3 grid((C + 255) / 256, nblk);
  if (dtype == GDL_BF16)
    hipLaunchKernelGGL(dwconv_gelu_bwd_kernel<bf16_tag>, grid, dim3(256), 0, s, u, dy, H, W, C, w9, bias, dpre, ws, seglen, nseg, items);
  else
    hipLaunchKernelGGL(dwconv_gelu_bwd_kernel<float>, grid, dim3(256), 0, s, u, dy, H, W, C, w9, bias, dpre, ws, seglen, nseg, items);
  launch_reduce(ws, nblk, 10 * (int64_t)C, 9 * C, dw9, accumulate, s);
  launch_reduce(ws + 9 * (int64_t)C, nblk, 10 * (int64_t)C, C, dbias, accumulate, s);
  GDL_CHECK_LAUNCH("gdl_dwconv3x3_gelu_bwd");
  return GDL_OK;
}

extern "C" int gdl_col2im(const void* cols, int dtype, int B, int Ho, int Wo, int R, int S, int C, int stride, int pad,
                          int H, int W, void* dx, int dx_dtype, int64_t dx_sB, int64_t dx_sH, int64_t dx_sW,
                          gdl_stream_t stream) {
  GDL_CHECK_ARG(cols && dx, "gdl_col2im: null pointer");
  GDL_CHECK_ARG(C > 0 && C % 4 == 0 && stride > 0 && pad >= 0 && R > 0 && S > 0, "gdl_col2im: bad geometry (C %% 4)");
  const int64_t total = (int64_t)B * H * W * (C / 4);
  if (total <= 0) return GDL_OK;
  hipStream_t s = (hipStream_t)stream;
  const unsigned nb = (unsigned)((total + 255) / 256 < 262144 ? (total + 255) / 256 : 262144);
#define C2I(TI, TO) hipLaunchKernelGGL((col2im_kernel<TI, TO>), dim3(nb), dim3(256), 0, s, cols, B, Ho, Wo, R, S, C, stride, pad, H, W, dx, dx_sB, dx_sH, dx_sW)
  if (dtype == GDL_BF16) { if (dx_dtype == GDL_BF16) C2I(bf16_tag, bf16_tag); else C2I(bf16_tag, float); }
  else { if (dx_dtype == GDL_BF16) C2I(float, bf16_tag); else C2I(float, float); }
#undef C2I
  GDL_CHECK_LAUNCH("gdl_col2im");
  return GDL_OK;
}
