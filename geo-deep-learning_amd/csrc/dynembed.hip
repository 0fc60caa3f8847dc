// Channel-adaptive patch embedding of the SegFormer "dynamic" encoder (mix_transformer.py:762-859, DynamicChannelEmbed):
// every input band goes through ONE shared 7x7/stride-4 convolution (patchify + GEMM, done by the caller), is scaled
// by a per-band weight vector generated from the band's sinusoidal position code, and the bands are pooled per pixel
// with a softmax attention over bands.
//
//   gdl_chan_weights_fwd/bwd : the input-independent part, [C x 128] matrices, one block
//       hid = relu(pos W0^T + b0) ; cw = tanh(hid W2^T + b2)          (weight_gen, :781-786)
//       hb  = pos W1b^T + b1                                          (the position half of channel_attention[0])
//   gdl_chan_pool_fwd/bwd    : per output pixel (one wave per pixel, lane = embedding element)
//       xw[c] = conv[c] * cw[c] ; h[c] = relu(W1a xw[c] + hb[c]) ; s[c] = w2 . h[c] + b2s
//       a = softmax_c(s) ; agg = sum_c a[c] xw[c]                     (:823-853)
//     backward recomputes xw / h, writes dconv and per-wave partial sums of the parameter gradients into a workspace;
//     a second kernel adds the partials in a fixed order (no atomics).  d b2s == 0 identically (softmax is shift
//     invariant), so it is not produced.
// All f32; HBM-bound on the conv tensor ([B*C, P, E], read once forward, read + written once backward).
#include "gdl_common.h"

namespace {

constexpr int kMaxC = 16;     // bands
constexpr int kMaxE = 64;     // embedding width (MiT-B0: 32, B1..B5: 64)
constexpr int kMaxH1 = 32;    // hidden width of the band attention = E / 2

__global__ __launch_bounds__(256) void chan_weights_fwd_kernel(const float* __restrict__ pos, int C, int PD, int HD, int E,
                                                               int H1, const float* __restrict__ W0,
                                                               const float* __restrict__ b0, const float* __restrict__ W2,
                                                               const float* __restrict__ b2, const float* __restrict__ W1b,
                                                               int64_t w1b_ld, const float* __restrict__ b1,
                                                               float* __restrict__ hid, float* __restrict__ cw,
                                                               float* __restrict__ hb) {
  const int t = threadIdx.x;
  for (int i = t; i < C * HD; i += 256) {
    const int c = i / HD, h = i - c * HD;
    float s = b0[h];
    for (int k = 0; k < PD; ++k) s += pos[c * PD + k] * W0[h * PD + k];
    hid[i] = s > 0.f ? s : 0.f;
  }
  for (int i = t; i < C * H1; i += 256) {
    const int c = i / H1, j = i - c * H1;
    float s = b1[j];
    for (int k = 0; k < PD; ++k) s += pos[c * PD + k] * W1b[j * w1b_ld + k];
    hb[i] = s;
  }
  __threadfence_block();   // hid is read back below by other threads of this (single) block
  __syncthreads();
  for (int i = t; i < C * E; i += 256) {
    const int c = i / E, e = i - c * E;
    float s = b2[e];
    for (int k = 0; k < HD; ++k) s += hid[c * HD + k] * W2[e * HD + k];
    cw[i] = tanhf(s);
  }
}

// dhid lives in LDS: C * HD <= 16 * 128 floats
__global__ __launch_bounds__(256) void chan_weights_bwd_kernel(const float* __restrict__ pos, int C, int PD, int HD, int E,
                                                               int H1, const float* __restrict__ W2,
                                                               const float* __restrict__ hid, const float* __restrict__ cw,
                                                               const float* __restrict__ dcw, const float* __restrict__ dhb,
                                                               float* __restrict__ dW0, float* __restrict__ db0,
                                                               float* __restrict__ dW2, float* __restrict__ db2,
                                                               float* __restrict__ dW1b, float* __restrict__ db1) {
  __shared__ float dz2[kMaxC * kMaxE];
  __shared__ float dhid[kMaxC * 128];
  const int t = threadIdx.x;
  for (int i = t; i < C * E; i += 256) dz2[i] = dcw[i] * (1.f - cw[i] * cw[i]);
  __syncthreads();
  for (int i = t; i < E * HD; i += 256) {
    const int e = i / HD, h = i - e * HD;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += dz2[c * E + e] * hid[c * HD + h];
    dW2[i] = s;
  }
  for (int e = t; e < E; e += 256) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += dz2[c * E + e];
    db2[e] = s;
  }
  for (int i = t; i < C * HD; i += 256) {
    const int c = i / HD, h = i - c * HD;
    float s = 0.f;
    for (int e = 0; e < E; ++e) s += dz2[c * E + e] * W2[e * HD + h];
    dhid[i] = hid[i] > 0.f ? s : 0.f;
  }
  __syncthreads();
  for (int i = t; i < HD * PD; i += 256) {
    const int h = i / PD, k = i - h * PD;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += dhid[c * HD + h] * pos[c * PD + k];
    dW0[i] = s;
  }
  for (int h = t; h < HD; h += 256) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += dhid[c * HD + h];
    db0[h] = s;
  }
  for (int i = t; i < H1 * PD; i += 256) {
    const int j = i / PD, k = i - j * PD;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += dhb[c * H1 + j] * pos[c * PD + k];
    dW1b[i] = s;
  }
  for (int j = t; j < H1; j += 256) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += dhb[c * H1 + j];
    db1[j] = s;
  }
}

// Shared by forward and backward: the wave's pixel -> xw (registers, lane = e), h and s through LDS.
//   lds_x [wave][C][E]   xw of the wave's pixel
//   lds_s [wave][C]      attention logits
// Roles: "lane e" (e = lane, active if e < E) and "lane (j, sub)" (j = lane % H1, sub = lane / H1, nsub = 64 / H1): the
// second role owns hidden unit j for the bands sub, sub + nsub, ...
struct PoolArgs {
  const float* conv;   // [B][C][P][E]
  const float* cw;     // [C][E]
  const float* w1a;    // [H1][ld] first E columns
  int64_t w1a_ld;
  const float* hb;     // [C][H1]
  const float* w2;     // [H1]
  float b2s;
  int B, C, P, E, H1;
};

template <bool BWD>
__global__ __launch_bounds__(256) void chan_pool_kernel(const PoolArgs k, float* __restrict__ agg, float* __restrict__ attn,
                                                        const float* __restrict__ dagg, float* __restrict__ dconv,
                                                        float* __restrict__ ws, int64_t ws_ld) {
  __shared__ float lds_x[4][kMaxC * kMaxE];
  __shared__ float lds_s[4][kMaxC];
  __shared__ float lds_dh[4][kMaxC * kMaxH1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = k.C, E = k.E, H1 = k.H1;
  const int e = lane, j = lane % H1, sub = lane / H1, nsub = 64 / H1;
  const bool e_on = e < E;
  // role (j, sub): row j of W1a and w2[j];  role e: column e of W1a (backward) and cw[.][e]
  float w1row[kMaxE];
#pragma unroll
  for (int q = 0; q < kMaxE; ++q) w1row[q] = q < E ? k.w1a[j * k.w1a_ld + q] : 0.f;
  const float w2j = k.w2[j];
  float cwv[kMaxC];
#pragma unroll
  for (int c = 0; c < kMaxC; ++c) cwv[c] = (c < C && e_on) ? k.cw[c * E + e] : 0.f;
  float w1col[BWD ? kMaxH1 : 1];
  float acc_w1[BWD ? kMaxH1 : 1], acc_cw[BWD ? kMaxC : 1], acc_hb[BWD ? kMaxC : 1], acc_w2 = 0.f;
  if constexpr (BWD) {
#pragma unroll
    for (int q = 0; q < kMaxH1; ++q) { w1col[q] = (q < H1 && e_on) ? k.w1a[q * k.w1a_ld + e] : 0.f; acc_w1[q] = 0.f; }
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) { acc_cw[c] = 0.f; acc_hb[c] = 0.f; }
  }
  const int64_t npix = (int64_t)k.B * k.P;
  for (int64_t base = (int64_t)blockIdx.x * 4; base < npix; base += (int64_t)gridDim.x * 4) {
    const int64_t pix = base + wave;
    const bool on = pix < npix;
    const int64_t b = on ? pix / k.P : 0, p = on ? pix - b * k.P : 0;
    float cv[kMaxC], xw[kMaxC];
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) {
      cv[c] = (on && c < C && e_on) ? k.conv[((b * C + c) * k.P + p) * E + e] : 0.f;
      xw[c] = cv[c] * cwv[c];
      if (c < C && e_on) lds_x[wave][c * E + e] = xw[c];
    }
    __syncthreads();
    // h[c][j] for this lane's bands, logits s[c]
    float hval[kMaxC];   // slot q <-> band sub + q * nsub
#pragma unroll
    for (int q = 0; q < kMaxC; ++q) {
      const int c = sub + q * nsub;
      float d = 0.f;
      if (c < C) {
        d = k.hb[c * H1 + j];
#pragma unroll
        for (int qe = 0; qe < kMaxE; ++qe)
          if (qe < E) d += w1row[qe] * lds_x[wave][c * E + qe];
        d = d > 0.f ? d : 0.f;
      }
      hval[q] = d;
      float sv = d * w2j;
      for (int o = 1; o < H1; o <<= 1) sv += __shfl_xor(sv, o, 64);
      if (c < C && j == 0) lds_s[wave][c] = sv + k.b2s;
    }
    __syncthreads();
    float a[kMaxC], mx = -3.0e38f, den = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
      if (c < C) mx = fmaxf(mx, lds_s[wave][c]);
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) { a[c] = c < C ? expf(lds_s[wave][c] - mx) : 0.f; den += a[c]; }
    const float inv = 1.f / den;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) a[c] *= inv;
    if constexpr (!BWD) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < kMaxC; ++c) s += a[c] * xw[c];
      if (on && e_on) agg[pix * E + e] = s;
      if (on && lane < C) {
        float av = 0.f;
#pragma unroll
        for (int c = 0; c < kMaxC; ++c) av = lane == c ? a[c] : av;
        attn[pix * C + lane] = av;
      }
    } else {
      const float dg = (on && e_on) ? dagg[pix * E + e] : 0.f;
      float da[kMaxC], dot = 0.f;
#pragma unroll
      for (int c = 0; c < kMaxC; ++c) {
        da[c] = c < C ? wave_sum(dg * xw[c]) : 0.f;
        dot += a[c] * da[c];
      }
      float ds[kMaxC];
#pragma unroll
      for (int c = 0; c < kMaxC; ++c) ds[c] = a[c] * (da[c] - dot);
      // role (j, sub): dh[c][j], its parameter gradients
#pragma unroll
      for (int q = 0; q < kMaxC; ++q) {
        const int c = sub + q * nsub;
        if (c < C) {
          float dsc = 0.f;
#pragma unroll
          for (int cc = 0; cc < kMaxC; ++cc) dsc = cc == c ? ds[cc] : dsc;
          const float dh = hval[q] > 0.f ? dsc * w2j : 0.f;
          lds_dh[wave][c * H1 + j] = dh;
          acc_hb[q] += dh;
          acc_w2 += dsc * hval[q];
        }
      }
      __syncthreads();
      // role e: dxw, dconv, d cw, d W1a
#pragma unroll
      for (int c = 0; c < kMaxC; ++c) {
        if (c < C) {
          float dx = a[c] * dg;
#pragma unroll
          for (int q = 0; q < kMaxH1; ++q)
            if (q < H1) {
              const float dh = lds_dh[wave][c * H1 + q];
              dx += w1col[q] * dh;
              acc_w1[q] += dh * xw[c];
            }
          acc_cw[c] += dx * cv[c];
          if (on && e_on) dconv[((b * C + c) * k.P + p) * E + e] = dx * cwv[c];
        }
      }
    }
    __syncthreads();   // lds_x / lds_s / lds_dh are rewritten by the next pixel
  }
  if constexpr (BWD) {
    // partial row of this wave: [dW1a H1*E | dhb C*H1 | dw2 H1 | dcw C*E]
    float* row = ws + ((int64_t)blockIdx.x * 4 + wave) * ws_ld;
    if (e_on) {
#pragma unroll
      for (int q = 0; q < kMaxH1; ++q)
        if (q < H1) row[q * E + e] = acc_w1[q];
#pragma unroll
      for (int c = 0; c < kMaxC; ++c)
        if (c < C) row[H1 * E + C * H1 + H1 + c * E + e] = acc_cw[c];
    }
#pragma unroll
    for (int q = 0; q < kMaxC; ++q) {
      const int c = sub + q * nsub;
      if (c < C) row[H1 * E + c * H1 + j] = acc_hb[q];
    }
    for (int o = H1; o < 64; o <<= 1) acc_w2 += __shfl_xor(acc_w2, o, 64);
    if (sub == 0) row[H1 * E + C * H1 + j] = acc_w2;
  }
}

// out[i] = sum_rows ws[row][i], rows added in index order by 8 interleaved groups
__global__ __launch_bounds__(256) void chan_pool_reduce_kernel(const float* __restrict__ ws, int nrows, int64_t ld, int total,
                                                               float* __restrict__ out) {
  __shared__ float part[8][32];
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + col;
  float s = 0.f;
  if (i < total)
    for (int r = grp; r < nrows; r += 8) s += ws[(int64_t)r * ld + i];
  part[grp][col] = s;
  __syncthreads();
  if (grp == 0 && i < total)
    out[i] = ((part[0][col] + part[1][col]) + (part[2][col] + part[3][col])) +
             ((part[4][col] + part[5][col]) + (part[6][col] + part[7][col]));
}

int pool_blocks(int64_t npix) {
  int64_t n = (npix + 3) / 4;
  if (n > 1024) n = 1024;
  if (n < 1) n = 1;
  return (int)n;
}
int64_t pool_row(int C, int E, int H1) { return (int64_t)H1 * E + (int64_t)C * H1 + H1 + (int64_t)C * E; }

bool dims_ok(int C, int E, int H1) {
  return C >= 1 && C <= kMaxC && (E == 32 || E == 64) && (H1 == 16 || H1 == 32) && H1 <= E;
}

}  // namespace

extern "C" int gdl_chan_weights_fwd(const float* pos, int C, int PD, int HD, int E, int H1, const float* W0,
                                    const float* b0, const float* W2, const float* b2, const float* W1b, int64_t w1b_ld,
                                    const float* b1, float* hid, float* cw, float* hb, gdl_stream_t stream) {
  GDL_CHECK_ARG(pos && W0 && b0 && W2 && b2 && W1b && b1 && hid && cw && hb, "gdl_chan_weights_fwd: null pointer");
  GDL_CHECK_ARG(dims_ok(C, E, H1) && PD > 0 && HD > 0 && HD <= 128,
                "gdl_chan_weights_fwd: need 1 <= C <= 16, E in {32, 64}, H1 in {16, 32}, hidden <= 128 (C=%d E=%d)", C, E);
  hipLaunchKernelGGL(chan_weights_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pos, C, PD, HD, E, H1, W0, b0,
                     W2, b2, W1b, w1b_ld, b1, hid, cw, hb);
  GDL_CHECK_LAUNCH("gdl_chan_weights_fwd");
  return GDL_OK;
}

extern "C" int gdl_chan_weights_bwd(const float* pos, int C, int PD, int HD, int E, int H1, const float* W2,
                                    const float* hid, const float* cw, const float* dcw, const float* dhb, float* dW0,
                                    float* db0, float* dW2, float* db2, float* dW1b, float* db1, gdl_stream_t stream) {
  GDL_CHECK_ARG(pos && W2 && hid && cw && dcw && dhb && dW0 && db0 && dW2 && db2 && dW1b && db1,
                "gdl_chan_weights_bwd: null pointer");
  GDL_CHECK_ARG(dims_ok(C, E, H1) && PD > 0 && HD > 0 && HD <= 128, "gdl_chan_weights_bwd: unsupported sizes (C=%d E=%d)", C, E);
  hipLaunchKernelGGL(chan_weights_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pos, C, PD, HD, E, H1, W2, hid,
                     cw, dcw, dhb, dW0, db0, dW2, db2, dW1b, db1);
  GDL_CHECK_LAUNCH("gdl_chan_weights_bwd");
  return GDL_OK;
}

extern "C" int gdl_chan_pool_fwd(const float* conv, int B, int C, int64_t P, int E, int H1, const float* cw,
                                 const float* w1a, int64_t w1a_ld, const float* hb, const float* w2, float b2s,
                                 float* agg, float* attn, gdl_stream_t stream) {
  GDL_CHECK_ARG(conv && cw && w1a && hb && w2 && agg && attn, "gdl_chan_pool_fwd: null pointer");
  GDL_CHECK_ARG(dims_ok(C, E, H1) && P < ((int64_t)1 << 31), "gdl_chan_pool_fwd: unsupported sizes (C=%d E=%d H1=%d)", C, E, H1);
  if ((int64_t)B * P <= 0) return GDL_OK;
  PoolArgs k{conv, cw, w1a, w1a_ld, hb, w2, b2s, B, C, (int)P, E, H1};
  hipLaunchKernelGGL(chan_pool_kernel<false>, dim3(pool_blocks((int64_t)B * P)), dim3(256), 0, (hipStream_t)stream, k, agg,
                     attn, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (int64_t)0);
  GDL_CHECK_LAUNCH("gdl_chan_pool_fwd");
  return GDL_OK;
}

extern "C" int64_t gdl_chan_pool_workspace(int B, int C, int64_t P, int E, int H1) {
  return (int64_t)pool_blocks((int64_t)B * P) * 4 * pool_row(C, E, H1) * (int64_t)sizeof(float);
}

// grads = [dW1a H1*E | dhb C*H1 | dw2 H1 | dcw C*E] (f32, overwritten)
extern "C" int gdl_chan_pool_bwd(const float* conv, int B, int C, int64_t P, int E, int H1, const float* cw,
                                 const float* w1a, int64_t w1a_ld, const float* hb, const float* w2, float b2s,
                                 const float* dagg, float* dconv, float* grads, float* ws, int64_t ws_bytes,
                                 gdl_stream_t stream) {
  GDL_CHECK_ARG(conv && cw && w1a && hb && w2 && dagg && dconv && grads && ws, "gdl_chan_pool_bwd: null pointer");
  GDL_CHECK_ARG(dims_ok(C, E, H1) && P < ((int64_t)1 << 31), "gdl_chan_pool_bwd: unsupported sizes (C=%d E=%d H1=%d)", C, E, H1);
  GDL_CHECK_ARG(ws_bytes >= gdl_chan_pool_workspace(B, C, P, E, H1), "gdl_chan_pool_bwd: workspace too small");
  if ((int64_t)B * P <= 0) return GDL_OK;
  hipStream_t s = (hipStream_t)stream;
  PoolArgs k{conv, cw, w1a, w1a_ld, hb, w2, b2s, B, C, (int)P, E, H1};
  const int nb = pool_blocks((int64_t)B * P);
  const int64_t ld = pool_row(C, E, H1);
  hipLaunchKernelGGL(chan_pool_kernel<true>, dim3(nb), dim3(256), 0, s, k, (float*)nullptr, (float*)nullptr, dagg, dconv, ws, ld);
  hipLaunchKernelGGL(chan_pool_reduce_kernel, dim3((unsigned)((ld + 31) / 32)), dim3(256), 0, s, ws, nb * 4, ld, (int)ld, grads);
  GDL_CHECK_LAUNCH("gdl_chan_pool_bwd");
  return GDL_OK;
}
