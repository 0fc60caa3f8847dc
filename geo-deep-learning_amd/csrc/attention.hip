// Fused (flash-style) attention forward for the DOFA ViT blocks: bf16, head_dim 64, f32 softmax.
//
//   O[b, q, h*64 + d] = sum_k softmax_k( scale * Q[q]·K[k] ) V[k, d]
//
// Work split: block = 4 waves x 32 queries; KV tile = 64 keys, staged by global -> LDS DMA
// (two stages, one barrier per tile) in the same swizzled 128-byte-row format as the GEMM.
// Q and K/V may live in different tensors with different lengths (SegFormer's spatial-reduction
// attention, mix_transformer.py:120-157) or in one packed qkv tensor (timm Attention).
// Both MFMAs are issued "transposed" so that every softmax quantity of a query lives in ONE lane
// (plus its partner lane+32), i.e. no cross-lane traffic besides one xor-32 shuffle per tile:
//   S^T[key, q] = K · Q^T        A = K rows (LDS),  B = Q (registers, loaded once)
//   O^T[d,  q] = V^T · P^T       A = V^T rows (LDS), B = P^T = the S^T accumulator itself
// The k-index order of the second MFMA is permuted to match the accumulator layout of the first
// (legal for a reduction), so P never leaves registers and is only converted f32 -> bf16.
#include "gdl_common.h"

namespace {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ uint4 g_attn_zero_page[8];

struct FArgs {
  const uint16_t* q;    // query rows: q + b*q_sB + n*q_sN + h*64
  const uint16_t* k;    // key rows:   k + b*k_sB + n*k_sN + h*64
  int64_t q_sB, q_sN, k_sB, k_sN;
  const uint16_t* vt;   // [B, H, 64, Npad]
  uint16_t* o;          // [B, Nq, H*64]
  int B, H, Nq, N, Npad;  // N = number of keys
  float scale_log2e;    // scale * log2(e)
};

__global__ __launch_bounds__(256) void flash_fwd_kernel(const FArgs f) {
  constexpr int HD = 64, KV = 64;
  constexpr int STAGE_BYTES = 2 * KV * 128;  // K tile [64 keys][128 B] + V^T tile [64 d][128 B]
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / f.H, h = bh % f.H;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int frow = lane & 31, fhalf = lane >> 5, fswz = (frow >> 1) & 7;
  const int64_t row_stride = f.k_sN;  // elements between key tokens
  const uint16_t* qbase = f.q + (int64_t)b * f.q_sB + (int64_t)h * HD;
  const uint16_t* kbase = f.k + (int64_t)b * f.k_sB + (int64_t)h * HD;
  const uint16_t* vtbase = f.vt + (int64_t)bh * HD * f.Npad;

  // ---- Q fragments (B operand of S^T): lane (query = frow, half) holds d = 16*kk + 8*half + 0..7
  uint4 qf[4];
  {
    const int q = q0 + frow;
    const bool ok = q < f.Nq;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      qf[kk] = ok ? *(const uint4*)(qbase + (int64_t)q * f.q_sN + kk * 16 + fhalf * 8) : make_uint4(0, 0, 0, 0);
  }

  // ---- staging by DMA (global_load_lds_dwordx4): per tile 8 KiB of K rows + 8 KiB of V^T rows =
  //      16 wave-instructions of 1 KiB (8 rows); wave w issues row groups w and w+4 of each.  The LDS
  //      image is lane-linear, so the swizzle is applied to the per-lane SOURCE chunk; keys >= N read
  //      a zero page (V^T is zero-padded in memory already).
  const int uwave = __builtin_amdgcn_readfirstlane(wave);
  const int lrow = lane >> 3, lslot = lane & 7;
  const unsigned char* zero = (const unsigned char*)g_attn_zero_page;
  auto issue = [&](int stage, int kv0) {
    unsigned char* sk = smem + stage * STAGE_BYTES;
    unsigned char* sv = sk + KV * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rg = uwave + 4 * i;
      const int r = rg * 8 + lrow;
      const int chunk = lslot ^ ((r >> 1) & 7);
      const int key = kv0 + r;
      const unsigned char* srck = key < f.N ? (const unsigned char*)(kbase + (int64_t)key * row_stride + chunk * 8) : zero;
      dma16_to_lds(srck, sk + rg * 1024);
      const unsigned char* srcv = (const unsigned char*)(vtbase + (int64_t)r * f.Npad + kv0 + chunk * 8);
      dma16_to_lds(srcv, sv + rg * 1024);
    }
  };

  f32x16_t ot[2];  // O^T accumulators: d tile dt, rows d = 32*dt + (r&3)+8*(r>>2)+4*half, col = query
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;  // running max (log2 domain, shared by the lane pair), own partial sum

  const int ntiles = (f.N + KV - 1) / KV;
  issue(0, 0);
  for (int t = 0; t < ntiles; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own DMA of tile t landed
    __syncthreads();                                   // everyone's did; everyone left tile t-1
    if (t + 1 < ntiles) issue((t + 1) & 1, (t + 1) * KV);
    const unsigned char* sk = smem + (t & 1) * STAGE_BYTES;
    const unsigned char* sv = sk + KV * 128;

    // ---- S^T = K · Q^T : two 32-key row tiles
    f32x16_t st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const uint4 ka = *(const uint4*)(sk + (kt * 32 + frow) * 128 + (((2 * kk + fhalf) ^ fswz) << 4));
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ka),
                                                         __builtin_bit_cast(bf16x8_t, qf[kk]), st[kt], 0, 0, 0);
      }
    }
    // ---- online softmax for query = frow (this lane holds 32 of the 64 keys, partner the rest)
    const int kv0 = t * KV;
    if (t == ntiles - 1) {  // only the last tile can contain keys >= N (wave-uniform branch)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
          st[kt][r] = key < f.N ? st[kt][r] : -INFINITY;
        }
    }
    float mx = -INFINITY;   // running max is tracked on the RAW scores (scale > 0 commutes with max)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);          // finite: every tile has >= 1 valid key
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * f.scale_log2e);  // first tile: exp2(-inf) = 0
    m_run = m_new;
    const float mc = m_new * f.scale_log2e;
    float lsum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(st[kt][r], f.scale_log2e, -mc));  // one FMA + v_exp_f32
        st[kt][r] = p;
        lsum += p;
      }
    l_run = l_run * alpha + lsum;
    if (!__all(alpha == 1.f)) {  // no running max moved in this wave: skip the O rescale
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[dt][r] *= alpha;
    }

    // ---- O^T += V^T · P^T ; MFMA step (kt, s): B = P regs [kt][8s..8s+7] of this lane,
    //      A lane (d, half') = V^T[d][keys 32kt + 16s + 4half' + {0..3}, +8 + {0..3}]
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8_t pb;
#pragma unroll
        for (int e = 0; e < 8; ++e) pb[e] = (__bf16)st[kt][8 * s + e];
        const int c0 = 4 * kt + 2 * s;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const unsigned char* rowp = sv + (dt * 32 + frow) * 128 + fhalf * 8;
          const uint2 lo = *(const uint2*)(rowp + ((c0 ^ fswz) << 4));
          const uint2 hi = *(const uint2*)(rowp + (((c0 + 1) ^ fswz) << 4));
          const uint4 va = make_uint4(lo.x, lo.y, hi.x, hi.y);
          ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, va), pb, ot[dt], 0, 0, 0);
        }
      }

  }

  // ---- finalize: l = own + partner; O = O^T / l ; lane holds query frow, 32 of the 64 d's
  const float l = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l;
  const int q = q0 + frow;
  if (q < f.Nq) {
    uint16_t* orow = f.o + ((int64_t)b * f.Nq + q) * ((int64_t)f.H * HD) + (int64_t)h * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * fhalf;
        const uint2 v = make_uint2(pack_bf16x2(ot[dt][4 * g] * inv, ot[dt][4 * g + 1] * inv),
                                   pack_bf16x2(ot[dt][4 * g + 2] * inv, ot[dt][4 * g + 3] * inv));
        *(uint2*)(orow + d) = v;
      }
  }
}

}  // namespace

extern "C" int gdl_flash_attn_fwd(const void* q, int64_t q_sB, int64_t q_sN, const void* k, int64_t k_sB, int64_t k_sN,
                                  const void* vt, void* o, int B, int H, int Nq, int Nkv, int Npad, float scale,
                                  gdl_stream_t stream) {
  GDL_CHECK_ARG(q && k && vt && o, "gdl_flash_attn_fwd: null pointer");
  GDL_CHECK_ARG(B > 0 && H > 0 && Nq > 0 && Nkv > 0 && Npad >= Nkv && Npad % 64 == 0, "gdl_flash_attn_fwd: bad dims (Npad %% 64)");
  GDL_CHECK_ARG(((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)vt % 16 == 0) && ((uintptr_t)o % 8 == 0) &&
                    q_sB % 8 == 0 && q_sN % 8 == 0 && k_sB % 8 == 0 && k_sN % 8 == 0,
                "gdl_flash_attn_fwd: pointers / strides must keep 16-byte alignment");
  FArgs f;
  f.q = (const uint16_t*)q; f.k = (const uint16_t*)k;
  f.q_sB = q_sB; f.q_sN = q_sN; f.k_sB = k_sB; f.k_sN = k_sN;
  f.vt = (const uint16_t*)vt;
  f.o = (uint16_t*)o;
  f.B = B; f.H = H; f.Nq = Nq; f.N = Nkv; f.Npad = Npad;
  f.scale_log2e = scale * 1.4426950408889634f;
  dim3 grid((Nq + 127) / 128, B * H);
  hipLaunchKernelGGL(flash_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, f);
  GDL_CHECK_LAUNCH("gdl_flash_attn_fwd");
  return GDL_OK;
}
