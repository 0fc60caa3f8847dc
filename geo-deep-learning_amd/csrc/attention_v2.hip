// Fused attention for the ViT / MiT blocks, second generation: bf16, head_dim 64, f32 softmax.
//
//   forward   O = softmax(scale Q K^T) V, plus LSE[b,h,q] = log sum_k exp(scale q.k)      (flash_fwd2_kernel)
//   backward  dQ, dK, dV with the probabilities RECOMPUTED tile by tile from LSE -- nothing of size Nq x Nkv ever
//             exists in HBM (the first generation materialised P and dS as [B,H,Nq,Npad]).   Two kernels, no atomics:
//             flash_bwd_dq_kernel (one block per 128 queries, loops over key tiles) and flash_bwd_dkv_kernel (one block
//             per 128 keys, loops over query tiles); each recomputes S and dP for its own tiles.  Deterministic.
//
// Differences from attention.hip's kernel: V (and, in the backward, K / Q / dO) are consumed TRANSPOSED straight out
// of the row-major [token][64] tile in LDS through ds_read_b64_tr_b16 -- the separate V^T pass over HBM is gone; a wave
// owns 64 queries (two 32-row MFMA tiles), so every K / V fragment read from the LDS feeds two MFMAs; operands are
// staged through buffer descriptors (out-of-range tokens arrive as hardware zeros).
//
// MFMA conventions (v_mfma_f32_32x32x16_bf16, D[i][j] += sum_k A[i][k] B[k][j]): lane l supplies A[i = l&31][8 k's of
// half l>>5] and B[8 k's of half l>>5][j = l&31]; it receives column j = l&31, rows (r&3) + 8(r>>2) + 4(l>>5).  All
// products are arranged so that the softmax statistics of a row live in one lane (+ its partner l^32), and every
// second GEMM takes the first one's ACCUMULATOR as its B operand (k order permuted identically on the A side).
//
// LDS tiles are [64 tokens][128 B] with the 16-byte chunk index XORed by tr_swz(token) -- conflict-free both for
// ds_read_b128 row fragments and for ds_read_b64_tr_b16 transposed fragments.
#include "gdl_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

constexpr unsigned kOob = 0x80000000u;
constexpr int TILE_BYTES = 64 * 128;   // one [64 tokens][64 channels] bf16 tile

__device__ __forceinline__ int tr_swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

// blockIdx.x -> (token block, batch*head).  The blocks of ONE (batch, head) read the same K / V (or Q / dO) rows: dealt
// out in launch order they land on all eight XCDs and each XCD's L2 fetches every head's operands again (PMC, round 2:
// 1.1 GB fetched per forward launch for 0.19 GB of q/k/v, L2 hit rate 42 %).  When the head count is a multiple of 8,
// XCD x (= block id % 8) owns heads x, x+8, ... and walks their token blocks consecutively.
__device__ __forceinline__ void decode_block(int nblk, int BH, int& blk, int& bh) {
  const int id = blockIdx.x;
  if ((BH & 7) == 0) {
    const int j = id >> 3;
    bh = (id & 7) + 8 * (j / nblk);
    blk = j % nblk;
  } else {
    bh = id / nblk;
    blk = id % nblk;
  }
}

struct Attn2Args {
  const uint16_t *q, *k, *v, *o, *dout;        // token rows: base + b*sB + n*sN + h*64 (bf16)
  uint16_t *out, *dq, *dk, *dv;
  int64_t q_sB, q_sN, k_sB, k_sN, v_sB, v_sN, o_sB, o_sN, do_sB, do_sN;
  int64_t dq_sB, dq_sN, dk_sB, dk_sN, dv_sB, dv_sN;
  float* lse;                                   // [B, H, Nq]  natural-log sum-exp of the scaled scores
  float* dvec;                                  // [B, H, Nq]  rowsum(dO * O)
  int B, H, Nq, N;                              // N = number of keys
  float scale, scale_log2e;
  int qsplit;                                   // dK/dV kernel: query range cut into qsplit parts (blockIdx.y)
  float* dkv_part;                              // qsplit > 1: f32 partials [qsplit][B*H][N][dK 64 | dV 64]
};

// One wave stages pieces of a [64 tokens][64 ch] tile: piece p (0..7) = token rows 8p..8p+7.
__device__ __forceinline__ void stage_piece(const srd_t& srd, int64_t tok_stride_bytes, int tok0, int ntok, int piece,
                                            unsigned lds_tile, int lane) {
  const int r = piece * 8 + (lane >> 3);
  const int chunk = (lane & 7) ^ tr_swz(r);
  const int tok = tok0 + r;
  const unsigned v = tok < ntok ? (unsigned)(tok * tok_stride_bytes + chunk * 16) : kOob;
  dma16_buf(v, srd, 0u, lds_tile + piece * 1024);
}

// A-operand fragment of 32 token rows (row = lane & 31 of row tile `rt`), k = channels 16kk + 8*half .. +7
__device__ __forceinline__ bf16x8_t row_frag(const unsigned char* tile, int rt, int kk, int lane) {
  const int row = rt * 32 + (lane & 31);
  const uint4 v = *(const uint4*)(tile + row * 128 + (((2 * kk + (lane >> 5)) ^ tr_swz(row)) << 4));
  return __builtin_bit_cast(bf16x8_t, v);
}

// A-operand fragment of the TRANSPOSED tile: row i = channel 32*ct + (lane & 31), k = tokens
// 32*tt + 16*s + 4*half + {0,1,2,3, 8,9,10,11} -- the token order of accumulator registers 8s .. 8s+7 of a
// 32x32 tile whose rows are tokens (so that accumulator can be the B operand as it is).
__device__ __forceinline__ bf16x8_t col_frag(const unsigned char* tile, int ct, int tt, int s, int lane) {
  const int g = lane >> 4, l = lane & 15;
  const int ch = 32 * ct + 16 * (g & 1) + 4 * (l & 3);
  const int t0 = 32 * tt + 16 * s + 4 * (g >> 1) + (l >> 2);
  const unsigned char* p0 = tile + t0 * 128 + (((ch >> 3) ^ tr_swz(t0)) << 4) + (ch & 7) * 2;
  const int t1 = t0 + 8;
  const unsigned char* p1 = tile + t1 * 128 + (((ch >> 3) ^ tr_swz(t1)) << 4) + (ch & 7) * 2;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p0);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p1);
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ bf16x8_t acc_as_b(const f32x16_t& a, int s) {
  bf16x8_t r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (__bf16)a[8 * s + e];
  return r;
}

__device__ __forceinline__ f32x16_t zero16() {
  f32x16_t z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// ------------------------------------------------------------------------------------------------ forward
// block = NW waves x 64 queries (NW = 4; 2 for query ranges of at most 128 rows); KV tile = 64 keys (K tile + V tile =
// 16 KiB per stage, two stages).
template <int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void flash_fwd2_kernel(const Attn2Args f) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];   // stage s: K at 2s, V at 2s+1
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int blk, bh;
  decode_block((f.Nq + 64 * NW - 1) / (64 * NW), f.B * f.H, blk, bh);
  const int b = bh / f.H, h = bh % f.H;
  const int q0 = blk * (64 * NW) + wave * 64;
  const bool active = q0 < f.Nq;                                   // wave-uniform; idle waves still stage and sync
  const int frow = lane & 31, fhalf = lane >> 5;
  const uint16_t* qbase = f.q + (int64_t)b * f.q_sB + (int64_t)h * 64;
  const srd_t srd_k = make_srd(f.k + (int64_t)b * f.k_sB + (int64_t)h * 64, (unsigned)(((int64_t)f.N - 1) * f.k_sN * 2 + 128));
  const srd_t srd_v = make_srd(f.v + (int64_t)b * f.v_sB + (int64_t)h * 64, (unsigned)(((int64_t)f.N - 1) * f.v_sN * 2 + 128));
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);

  bf16x8_t qf[2][4];   // B operand of S^T: lane (query frow of tile qt, half) holds channels 16kk + 8 half ..
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = q0 + qt * 32 + frow;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q < f.Nq) v = *(const uint4*)(qbase + (int64_t)q * f.q_sN + kk * 16 + fhalf * 8);
      qf[qt][kk] = __builtin_bit_cast(bf16x8_t, v);
    }
  }
  auto issue = [&](int stage, int kv0) {   // 8 + 8 pieces per tile pair, dealt round-robin to the NW waves
#pragma unroll
    for (int i = 0; i < (8 + NW - 1) / NW; ++i) {
      const int p = wave + NW * i;
      if (p < 8) {
        stage_piece(srd_k, f.k_sN * 2, kv0, f.N, p, lds_base + (2 * stage) * TILE_BYTES, lane);
        stage_piece(srd_v, f.v_sN * 2, kv0, f.N, p, lds_base + (2 * stage + 1) * TILE_BYTES, lane);
      }
    }
  };
  f32x16_t ot[2][2];   // O^T accumulators [query tile][channel tile]: rows = channels, column = query
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) ot[qt][ct] = zero16();
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

  const int ntiles = (f.N + 63) / 64;
  issue(0, 0);
  for (int t = 0; t < ntiles; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < ntiles) issue((t + 1) & 1, (t + 1) * 64);
    if (!active) continue;
    const unsigned char* sk = smem + (2 * (t & 1)) * TILE_BYTES;
    const unsigned char* sv = sk + TILE_BYTES;
    // ---- S^T[key, query] = K . Q^T for two key row tiles x two query tiles
    f32x16_t st[2][2];   // [query tile][key tile]
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) st[qt][kt] = zero16();
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8_t ka = row_frag(sk, kt, kk, lane);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) st[qt][kt] = MFMA(ka, qf[qt][kk], st[qt][kt]);
      }
    // ---- online softmax per query (lane + partner lane^32 hold its 64 scores of this tile)
    const int kv0 = t * 64;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      if (t == ntiles - 1) {   // only the last tile can hold keys >= N
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kv0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
            st[qt][kt][r] = key < f.N ? st[qt][kt][r] : -INFINITY;
          }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[qt][kt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[qt], mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_new) * f.scale_log2e);
      m_run[qt] = m_new;
      const float mc = m_new * f.scale_log2e;
      float lsum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(st[qt][kt][r], f.scale_log2e, -mc));
          st[qt][kt][r] = p;
          lsum += p;
        }
      l_run[qt] = l_run[qt] * alpha + lsum;
      if (!__all(alpha == 1.f)) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int r = 0; r < 16; ++r) ot[qt][ct][r] *= alpha;
      }
    }
    // ---- O^T[ch, query] += V^T . P^T : A = transposed V fragments, B = the probabilities in registers
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8_t pb0 = acc_as_b(st[0][kt], s), pb1 = acc_as_b(st[1][kt], s);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const bf16x8_t va = col_frag(sv, ct, kt, s, lane);
          ot[0][ct] = MFMA(va, pb0, ot[0][ct]);
          ot[1][ct] = MFMA(va, pb1, ot[1][ct]);
        }
      }
  }
  if (!active) return;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const float l = l_run[qt] + __shfl_xor(l_run[qt], 32, 64);
    const float inv = 1.f / l;
    const int q = q0 + qt * 32 + frow;
    if (q >= f.Nq) continue;
    if (f.lse && fhalf == 0)
      f.lse[(int64_t)bh * f.Nq + q] = (m_run[qt] * f.scale_log2e + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
    uint16_t* orow = f.out + (int64_t)b * f.o_sB + (int64_t)q * f.o_sN + (int64_t)h * 64;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = ct * 32 + 8 * g + 4 * fhalf;
        *(uint2*)(orow + d) = make_uint2(pack_bf16x2(ot[qt][ct][4 * g] * inv, ot[qt][ct][4 * g + 1] * inv),
                                         pack_bf16x2(ot[qt][ct][4 * g + 2] * inv, ot[qt][ct][4 * g + 3] * inv));
      }
  }
}

// ------------------------------------------------------------------------------------------------ forward, round 3
// Same tiles, operands and arithmetic as flash_fwd2_kernel (bit-identical results with defer = 0), but a wave owns ONE
// 32-query tile and four waves share a SIMD (<= 128 registers).  Why: at head dim 64 the softmax costs more VALU cycles
// (max, scale, exp2, row sum, bf16 pack: ~230 issue slots per 32 x 64 scores, exp2 at quarter rate) than its two GEMMs cost
// matrix cycles (16 MFMAs), and the chip holds ~1.7 GHz under this load -- the kernel is VALU-bound, and what was missing
// with two 64-query waves per SIMD was overlap: each wave runs S -> softmax -> PV as one dependent chain behind a per-tile
// barrier, so the matrix pipe idled while both waves did softmax and the VALU idled while both waited for MFMAs or LDS
// (PMC, round 2: MFMA busy 28 %, VALU busy 54 %).  Four shorter chains per SIMD interleave: 278 -> 240 us at batch 32
// (N = 1297, 12 heads), 60 -> 47 us at batch 4.  Costs: every K / V fragment read from the LDS feeds one MFMA instead of
// two (LDS read traffic doubles, still < 50 % of the LDS rate here) and blocks of 128 queries stage K / V twice as often.
// Tried and measured, not kept: staggering the two query tiles of a 64-query wave (S0 S1 | softmax0 | PV0 | softmax1 |
// PV1, fragments held in registers: 278 -> 268 us); v_pk_fma_f32 / v_pk_add_f32 for the exponent arguments and row sums
// (20 % fewer VALU instructions, no time: packed f32 issues at half rate); eight-wave blocks of 256 queries (244 us).
// Round 6, measured and not kept (N = 1297, batch 64, 445-468 us): blocks of three waves (96 queries: 47 padded query rows per head
// instead of 111) 489-508 us -- five blocks per CU stage K / V 27 % more often; the all-padding second half of the last key tile
// (17 of 64 keys) skipped behind a uniform branch: 480-490 us -- the second copy of the tile body costs 64 bytes of scratch at the
// 128-register budget of four waves per SIMD.
// `defer` > 0: the running maximum is only raised (and O rescaled) when some row's tile maximum exceeds it by more than
// `defer` in the exp2 domain -- P stays below 2^defer; with the usual slowly growing maxima most tiles skip the rescale.
template <int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW == 2 ? 2 : 4, 4))) void flash_fwd3_kernel(const Attn2Args f, const float defer) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];   // stage s: K at 2s, V at 2s+1
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int blk, bh;
  decode_block((f.Nq + 32 * NW - 1) / (32 * NW), f.B * f.H, blk, bh);
  const int b = bh / f.H, h = bh % f.H;
  const int q0 = blk * (32 * NW) + wave * 32;
  const bool active = q0 < f.Nq;                                   // wave-uniform; idle waves still stage and sync
  const int frow = lane & 31, fhalf = lane >> 5;
  const uint16_t* qbase = f.q + (int64_t)b * f.q_sB + (int64_t)h * 64;
  const srd_t srd_k = make_srd(f.k + (int64_t)b * f.k_sB + (int64_t)h * 64, (unsigned)(((int64_t)f.N - 1) * f.k_sN * 2 + 128));
  const srd_t srd_v = make_srd(f.v + (int64_t)b * f.v_sB + (int64_t)h * 64, (unsigned)(((int64_t)f.N - 1) * f.v_sN * 2 + 128));
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);

  bf16x8_t qf[4];   // B operand of S^T: lane (query frow, half) holds channels 16kk + 8 half ..
  {
    const int q = q0 + frow;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q < f.Nq) v = *(const uint4*)(qbase + (int64_t)q * f.q_sN + kk * 16 + fhalf * 8);
      qf[kk] = __builtin_bit_cast(bf16x8_t, v);
    }
  }
  auto issue = [&](int stage, int kv0) {   // 8 + 8 pieces per tile pair, dealt round-robin to the NW waves
#pragma unroll
    for (int i = 0; i < (8 + NW - 1) / NW; ++i) {
      const int p = wave + NW * i;
      if (p < 8) {
        stage_piece(srd_k, f.k_sN * 2, kv0, f.N, p, lds_base + (2 * stage) * TILE_BYTES, lane);
        stage_piece(srd_v, f.v_sN * 2, kv0, f.N, p, lds_base + (2 * stage + 1) * TILE_BYTES, lane);
      }
    }
  };
  f32x16_t ot[2] = {zero16(), zero16()};   // O^T accumulators [channel tile]: rows = channels, column = query
  float m_run = -INFINITY, l_run = 0.f;
  const float sl2 = f.scale_log2e;
  const int ntiles = (f.N + 63) / 64;
  issue(0, 0);
  for (int t = 0; t < ntiles; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < ntiles) issue((t + 1) & 1, (t + 1) * 64);
    if (!active) continue;
    const unsigned char* sk = smem + (2 * (t & 1)) * TILE_BYTES;
    const unsigned char* sv = sk + TILE_BYTES;
    // ---- S^T[key, query] = K . Q^T for two key row tiles
    f32x16_t st[2] = {zero16(), zero16()};
    // (the two key row tiles interleaved: consecutive MFMAs never share an accumulator; same sums as tile-by-tile)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) st[kt] = MFMA(row_frag(sk, kt, kk, lane), qf[kk], st[kt]);
    // ---- online softmax per query (lane + partner lane^32 hold its 64 scores of this tile)
    if (t == ntiles - 1) {   // only the last tile can hold keys >= N
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * 64 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
          st[kt][r] = key < f.N ? st[kt][r] : -INFINITY;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float m_new = fmaxf(m_run, mx);
    if (defer > 0.f && __all((m_new - m_run) * sl2 <= defer)) m_new = m_run;   // (-inf start: never deferred)
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sl2);
    m_run = m_new;
    const float mc = m_new * sl2;
    float lsum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(st[kt][r], sl2, -mc));
        st[kt][r] = p;
        lsum += p;
      }
    l_run = l_run * alpha + lsum;
    if (!__all(alpha == 1.f)) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[ct][r] *= alpha;
    }
    // ---- O^T[ch, query] += V^T . P^T : A = transposed V fragments, B = the probabilities in registers
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const bf16x8_t pb = acc_as_b(st[kt], sl);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) ot[ct] = MFMA(col_frag(sv, ct, kt, sl, lane), pb, ot[ct]);
      }
  }
  if (!active) return;
  const float l = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l;
  const int q = q0 + frow;
  if (q >= f.Nq) return;
  if (f.lse && fhalf == 0)
    f.lse[(int64_t)bh * f.Nq + q] = (m_run * sl2 + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
  uint16_t* orow = f.out + (int64_t)b * f.o_sB + (int64_t)q * f.o_sN + (int64_t)h * 64;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = ct * 32 + 8 * g + 4 * fhalf;
      *(uint2*)(orow + d) = make_uint2(pack_bf16x2(ot[ct][4 * g] * inv, ot[ct][4 * g + 1] * inv),
                                       pack_bf16x2(ot[ct][4 * g + 2] * inv, ot[ct][4 * g + 3] * inv));
    }
}

// Round 4 tried to take work off the vector ALU (the round-3 analysis called this kernel VALU-bound): softmax scale folded into
// Q, the running reference subtracted by a fifth MFMA step (a "ones" K column against a Q column holding -m), no per-tile
// maximum at all (the reference only moves when a row sum leaves [2^-20, 2^20]).  Counters (profiles/r04c_pmc_attention_*):
// VALU instructions 67.8 M -> 49.8 M per launch (-26 %), MFMAs +12.5 %, and the kernel got SLOWER (226 -> 240 us): the time a
// wave spends parked at s_waitcnt / barriers went from 39 % to 56 % of its cycles.  A wave's tile is one dependent chain
// (K fragment reads -> S MFMAs -> exp2 -> V fragment reads -> PV MFMAs) and what the other three waves of the SIMD offer while
// it waits is exactly that VALU work; with less of it the LDS latencies are exposed.  Requesting all fragments of a phase ahead
// of its MFMAs needs 32 more registers than the 128 that four waves per SIMD allow (spills: 410 us).  The experiment was
// removed again; what stayed is the interleaved order of the two S accumulation chains below.

// ------------------------------------------------------------------------------------------------ forward, round 5
// The round-3 kernel with the two GEMMs of a wave SOFTWARE-PIPELINED across key tiles: S of tile t+1 is issued to the matrix
// pipe BEFORE the softmax of tile t, so a wave's own exp2 / max / row-sum VALU work runs under its own MFMAs instead of behind
// them (round-4 counters: a 32-query wave spent 39 % of its cycles parked on the chain K fragments -> S -> softmax -> V fragments
// -> PV; only the other waves of the SIMD filled those gaps).  Same tiles, same arithmetic, same order of every sum as
// flash_fwd3_kernel -- bit-identical results.  What changes:
//   * two S accumulator sets (current / next, roles swapped every tile: the loop is unrolled by two, no register copies);
//   * K runs one tile ahead of V in the LDS: at the top of iteration t the block has K[t+1] and V[t]; it then requests K[t+2]
//     into the slot of K[t] and V[t+1] into the slot of V[t-1] (both last read in iteration t-1, which the barrier closes).
//     Still two K and two V stages = 32 KiB;
//   * QLDS = false: Q fragments in registers, three waves per SIMD (the second S set costs 32 registers: 168 available);
//     QLDS = true : Q fragments re-read from a wave-private 4 KiB LDS tile each iteration, four waves per SIMD (128 registers).
template <int NW, bool QLDS>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(3, 3)))
void flash_fwd4_kernel(const Attn2Args f, const float defer) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES + (QLDS ? NW * 4096 : 16)];   // K stages 0,1 | V stages 2,3 | Q
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int blk, bh;
  decode_block((f.Nq + 32 * NW - 1) / (32 * NW), f.B * f.H, blk, bh);
  const int b = bh / f.H, h = bh % f.H;
  const int q0 = blk * (32 * NW) + wave * 32;
  const bool active = q0 < f.Nq;                                   // wave-uniform; idle waves still stage and sync
  const int frow = lane & 31, fhalf = lane >> 5;
  const uint16_t* qbase = f.q + (int64_t)b * f.q_sB + (int64_t)h * 64;
  const srd_t srd_k = make_srd(f.k + (int64_t)b * f.k_sB + (int64_t)h * 64, (unsigned)(((int64_t)f.N - 1) * f.k_sN * 2 + 128));
  const srd_t srd_v = make_srd(f.v + (int64_t)b * f.v_sB + (int64_t)h * 64, (unsigned)(((int64_t)f.N - 1) * f.v_sN * 2 + 128));
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
  unsigned char* qtile = smem + 4 * TILE_BYTES + (QLDS ? wave * 4096 : 0);

  bf16x8_t qf[4];   // B operand of S^T: lane (query frow, half) holds channels 16kk + 8 half ..
  {
    const int q = q0 + frow;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q < f.Nq) v = *(const uint4*)(qbase + (int64_t)q * f.q_sN + kk * 16 + fhalf * 8);
      if (QLDS) *(uint4*)(qtile + frow * 128 + (((2 * kk + fhalf) ^ tr_swz(frow)) << 4)) = v;
      else qf[kk] = __builtin_bit_cast(bf16x8_t, v);
    }
  }
  auto issue_k = [&](int stage, int kv0) {   // 8 pieces per tile, dealt round-robin to the NW waves
#pragma unroll
    for (int i = 0; i < (8 + NW - 1) / NW; ++i) {
      const int p = wave + NW * i;
      if (p < 8) stage_piece(srd_k, f.k_sN * 2, kv0, f.N, p, lds_base + stage * TILE_BYTES, lane);
    }
  };
  auto issue_v = [&](int stage, int kv0) {
#pragma unroll
    for (int i = 0; i < (8 + NW - 1) / NW; ++i) {
      const int p = wave + NW * i;
      if (p < 8) stage_piece(srd_v, f.v_sN * 2, kv0, f.N, p, lds_base + (2 + stage) * TILE_BYTES, lane);
    }
  };
  // S^T[key, query] = K . Q^T for the two key row tiles of one staged K tile (interleaved: consecutive MFMAs never share
  // an accumulator; same sums as tile-by-tile)
  auto scores = [&](f32x16_t (&st)[2], int kstage) {
    const unsigned char* sk = smem + kstage * TILE_BYTES;
    st[0] = zero16(); st[1] = zero16();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const bf16x8_t qb = QLDS ? row_frag(qtile, 0, kk, lane) : qf[kk];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) st[kt] = MFMA(row_frag(sk, kt, kk, lane), qb, st[kt]);
    }
  };
  f32x16_t ot[2] = {zero16(), zero16()};   // O^T accumulators [channel tile]: rows = channels, column = query
  float m_run = -INFINITY, l_run = 0.f;
  const float sl2 = f.scale_log2e;
  const int ntiles = (f.N + 63) / 64;
  f32x16_t sa[2], sb[2];

  // one key tile: `cur` holds S of tile t (issued one iteration ago), `nxt` receives S of tile t+1 first
  auto tile = [&](int t, f32x16_t (&cur)[2], f32x16_t (&nxt)[2]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                         // K[t+1] and V[t] are in the LDS; every wave is done with K[t] and V[t-1]
    if (t + 2 < ntiles) issue_k(t & 1, (t + 2) * 64);
    if (t + 1 < ntiles) issue_v((t + 1) & 1, (t + 1) * 64);
    if (!active) return;
    if (t + 1 < ntiles) scores(nxt, (t + 1) & 1);
    const unsigned char* sv = smem + (2 + (t & 1)) * TILE_BYTES;
    // ---- online softmax per query (lane + partner lane^32 hold its 64 scores of this tile)
    if (t == ntiles - 1) {   // only the last tile can hold keys >= N
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * 64 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
          cur[kt][r] = key < f.N ? cur[kt][r] : -INFINITY;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, cur[kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float m_new = fmaxf(m_run, mx);
    if (defer > 0.f && __all((m_new - m_run) * sl2 <= defer)) m_new = m_run;   // (-inf start: never deferred)
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sl2);
    m_run = m_new;
    const float mc = m_new * sl2;
    float lsum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(cur[kt][r], sl2, -mc));
        cur[kt][r] = p;
        lsum += p;
      }
    l_run = l_run * alpha + lsum;
    if (!__all(alpha == 1.f)) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[ct][r] *= alpha;
    }
    // ---- O^T[ch, query] += V^T . P^T : A = transposed V fragments, B = the probabilities in registers
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const bf16x8_t pb = acc_as_b(cur[kt], sl);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) ot[ct] = MFMA(col_frag(sv, ct, kt, sl, lane), pb, ot[ct]);
      }
  };

  issue_k(0, 0);
  issue_v(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (ntiles > 1) issue_k(1, 64);
  if (active) scores(sa, 0);
  for (int t = 0; t < ntiles; t += 2) {
    tile(t, sa, sb);
    if (t + 1 < ntiles) tile(t + 1, sb, sa);
  }
  if (!active) return;
  const float l = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l;
  const int q = q0 + frow;
  if (q >= f.Nq) return;
  if (f.lse && fhalf == 0)
    f.lse[(int64_t)bh * f.Nq + q] = (m_run * sl2 + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
  uint16_t* orow = f.out + (int64_t)b * f.o_sB + (int64_t)q * f.o_sN + (int64_t)h * 64;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = ct * 32 + 8 * g + 4 * fhalf;
      *(uint2*)(orow + d) = make_uint2(pack_bf16x2(ot[ct][4 * g] * inv, ot[ct][4 * g + 1] * inv),
                                       pack_bf16x2(ot[ct][4 * g + 2] * inv, ot[ct][4 * g + 3] * inv));
    }
}

// ------------------------------------------------------------------------------------------------ backward, part 0
// dvec[b,h,q] = sum_d dO[q,d] * O[q,d]: one wave per (b, q) row, lane = channel within a head
__global__ __launch_bounds__(256) void flash_bwd_dot_kernel(const Attn2Args f) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)f.B * f.Nq) return;
  const int b = (int)(row / f.Nq), q = (int)(row - (int64_t)b * f.Nq);
  const uint16_t* po = f.o + (int64_t)b * f.o_sB + (int64_t)q * f.o_sN;
  const uint16_t* pd = f.dout + (int64_t)b * f.do_sB + (int64_t)q * f.do_sN;
  for (int h = 0; h < f.H; ++h) {
    const float s = wave_sum(bf16_to_f32(po[h * 64 + lane]) * bf16_to_f32(pd[h * 64 + lane]));
    if (lane == 0) f.dvec[((int64_t)b * f.H + h) * f.Nq + q] = s;
  }
}

// ------------------------------------------------------------------------------------------------ backward, dQ
// block = 4 waves x 32 queries; loops over key tiles of 64 (K tile + V tile per stage).  Per tile and wave:
//   S^T  = K . Q^T            P^T  = exp(scale S^T - LSE_q)
//   dP^T = V . dO^T           dS^T = P^T (dP^T - dvec_q) scale
//   dQ^T[ch, q] += K^T . dS^T (A = transposed K fragments, B = dS^T in registers)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void flash_bwd_dq_kernel(const Attn2Args f) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int blk, bh;
  decode_block((f.Nq + 127) / 128, f.B * f.H, blk, bh);
  const int b = bh / f.H, h = bh % f.H;
  const int q0 = blk * 128 + wave * 32;
  const bool active = q0 < f.Nq;
  const int frow = lane & 31, fhalf = lane >> 5;
  const srd_t srd_k = make_srd(f.k + (int64_t)b * f.k_sB + (int64_t)h * 64, (unsigned)(((int64_t)f.N - 1) * f.k_sN * 2 + 128));
  const srd_t srd_v = make_srd(f.v + (int64_t)b * f.v_sB + (int64_t)h * 64, (unsigned)(((int64_t)f.N - 1) * f.v_sN * 2 + 128));
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
  const int q = q0 + frow;
  const bool qok = q < f.Nq;
  bf16x8_t qf[4], dof[4];
  {
    const uint16_t* qp = f.q + (int64_t)b * f.q_sB + (int64_t)q * f.q_sN + (int64_t)h * 64;
    const uint16_t* dp = f.dout + (int64_t)b * f.do_sB + (int64_t)q * f.do_sN + (int64_t)h * 64;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      uint4 a = make_uint4(0, 0, 0, 0), c = a;
      if (qok) { a = *(const uint4*)(qp + kk * 16 + fhalf * 8); c = *(const uint4*)(dp + kk * 16 + fhalf * 8); }
      qf[kk] = __builtin_bit_cast(bf16x8_t, a);
      dof[kk] = __builtin_bit_cast(bf16x8_t, c);
    }
  }
  const float lse2 = qok ? f.lse[(int64_t)bh * f.Nq + q] * 1.4426950408889634f : 0.f;   // log2 domain
  const float dv_q = qok ? f.dvec[(int64_t)bh * f.Nq + q] : 0.f;
  auto issue = [&](int stage, int kv0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      stage_piece(srd_k, f.k_sN * 2, kv0, f.N, 2 * wave + i, lds_base + (2 * stage) * TILE_BYTES, lane);
      stage_piece(srd_v, f.v_sN * 2, kv0, f.N, 2 * wave + i, lds_base + (2 * stage + 1) * TILE_BYTES, lane);
    }
  };
  f32x16_t dqt[2] = {zero16(), zero16()};   // dQ^T: rows = channels (tile ct), column = query
  const int ntiles = (f.N + 63) / 64;
  issue(0, 0);
  for (int t = 0; t < ntiles; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < ntiles) issue((t + 1) & 1, (t + 1) * 64);
    if (!active) continue;
    const unsigned char* sk = smem + (2 * (t & 1)) * TILE_BYTES;
    const unsigned char* sv = sk + TILE_BYTES;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      f32x16_t st = zero16(), dp = zero16();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        st = MFMA(row_frag(sk, kt, kk, lane), qf[kk], st);
        dp = MFMA(row_frag(sv, kt, kk, lane), dof[kk], dp);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = t * 64 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
        const float p = key < f.N ? __builtin_amdgcn_exp2f(fmaf(st[r], f.scale_log2e, -lse2)) : 0.f;
        st[r] = p * (dp[r] - dv_q) * f.scale;                      // dS^T
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8_t dsb = acc_as_b(st, s);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) dqt[ct] = MFMA(col_frag(sk, ct, kt, s, lane), dsb, dqt[ct]);
      }
    }
  }
  if (!active || !qok) return;
  uint16_t* row = f.dq + (int64_t)b * f.dq_sB + (int64_t)q * f.dq_sN + (int64_t)h * 64;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = ct * 32 + 8 * g + 4 * fhalf;
      *(uint2*)(row + d) = make_uint2(pack_bf16x2(dqt[ct][4 * g], dqt[ct][4 * g + 1]), pack_bf16x2(dqt[ct][4 * g + 2], dqt[ct][4 * g + 3]));
    }
}

// ------------------------------------------------------------------------------------------------ backward, dK / dV
// block = 4 waves x 32 keys; loops over query tiles of 64 (Q tile + dO tile per stage, LSE / dvec of the tile beside
// them).  Per tile and wave (lane = key column, registers = queries):
//   S   = Q . K^T   (A = Q rows, B = K registers)          P  = exp(scale S - LSE_q)
//   dP  = dO . V^T  (A = dO rows, B = V registers)         dS = P (dP - dvec_q) scale
//   dV^T[ch, key] += dO^T . P     dK^T[ch, key] += Q^T . dS   (A = transposed dO / Q fragments, B = P / dS registers)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void flash_bwd_dkv_kernel(const Attn2Args f) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES + 2 * 512];   // + per stage: LSE[64] | dvec[64]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int blk, bh;
  decode_block((f.N + 127) / 128, f.B * f.H, blk, bh);
  const int b = bh / f.H, h = bh % f.H;
  const int k0 = blk * 128 + wave * 32;
  const bool active = k0 < f.N;
  const int frow = lane & 31, fhalf = lane >> 5;
  const srd_t srd_q = make_srd(f.q + (int64_t)b * f.q_sB + (int64_t)h * 64, (unsigned)(((int64_t)f.Nq - 1) * f.q_sN * 2 + 128));
  const srd_t srd_do = make_srd(f.dout + (int64_t)b * f.do_sB + (int64_t)h * 64, (unsigned)(((int64_t)f.Nq - 1) * f.do_sN * 2 + 128));
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
  const int key = k0 + frow;
  const bool kok = key < f.N;
  bf16x8_t kf[4], vf[4];
  {
    const uint16_t* kp = f.k + (int64_t)b * f.k_sB + (int64_t)key * f.k_sN + (int64_t)h * 64;
    const uint16_t* vp = f.v + (int64_t)b * f.v_sB + (int64_t)key * f.v_sN + (int64_t)h * 64;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      uint4 a = make_uint4(0, 0, 0, 0), c = a;
      if (kok) { a = *(const uint4*)(kp + kk * 16 + fhalf * 8); c = *(const uint4*)(vp + kk * 16 + fhalf * 8); }
      kf[kk] = __builtin_bit_cast(bf16x8_t, a);
      vf[kk] = __builtin_bit_cast(bf16x8_t, c);
    }
  }
  float* stats = (float*)(smem + 4 * TILE_BYTES);                // [stage][2][64]
  auto issue = [&](int stage, int qs) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      stage_piece(srd_q, f.q_sN * 2, qs, f.Nq, 2 * wave + i, lds_base + (2 * stage) * TILE_BYTES, lane);
      stage_piece(srd_do, f.do_sN * 2, qs, f.Nq, 2 * wave + i, lds_base + (2 * stage + 1) * TILE_BYTES, lane);
    }
    if (wave == 0) {                                             // 64 LSE (log2 domain) + 64 dvec values of the tile
      const int qq = qs + lane;
      const bool ok = qq < f.Nq;
      stats[stage * 128 + lane] = ok ? f.lse[(int64_t)bh * f.Nq + qq] * 1.4426950408889634f : 0.f;
      stats[stage * 128 + 64 + lane] = ok ? f.dvec[(int64_t)bh * f.Nq + qq] : 0.f;
    }
  };
  f32x16_t dvt[2] = {zero16(), zero16()}, dkt[2] = {zero16(), zero16()};   // rows = channels (tile ct), column = key
  // query tiles [t_lo, t_hi) of this block: with few keys (MiT's spatial reduction: 256 keys for up to 16384 queries) the
  // key blocks alone are 2 x B x H workgroups, so the query range is cut into blockIdx.y parts whose f32 partial sums a
  // second kernel adds in a fixed order
  const int ntiles_all = (f.Nq + 63) / 64;
  const int per = (ntiles_all + f.qsplit - 1) / f.qsplit;
  const int t_lo = blockIdx.y * per, t_hi = t_lo + per < ntiles_all ? t_lo + per : ntiles_all;
  if (t_lo < t_hi) issue(t_lo & 1, t_lo * 64);
  for (int t = t_lo; t < t_hi; ++t) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < t_hi) issue((t + 1) & 1, (t + 1) * 64);
    if (!active) continue;
    const unsigned char* sq = smem + (2 * (t & 1)) * TILE_BYTES;
    const unsigned char* sd = sq + TILE_BYTES;
    const float* st_lse = stats + (t & 1) * 128;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      f32x16_t s = zero16(), dp = zero16();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        s = MFMA(row_frag(sq, qt, kk, lane), kf[kk], s);           // rows = queries of row tile qt, column = key
        dp = MFMA(row_frag(sd, qt, kk, lane), vf[kk], dp);
      }
      f32x16_t p;
#pragma unroll
      for (int g = 0; g < 4; ++g) {                                // registers 4g..4g+3 = queries qt*32 + 8g + 4 half + e
        const int qi = qt * 32 + 8 * g + 4 * fhalf;
        const float4 l4 = *(const float4*)(st_lse + qi), d4 = *(const float4*)(st_lse + 64 + qi);
        const float le[4] = {l4.x, l4.y, l4.z, l4.w}, de[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const bool ok = kok && (t * 64 + qi + e) < f.Nq;
          const float pv = ok ? __builtin_amdgcn_exp2f(fmaf(s[r], f.scale_log2e, -le[e])) : 0.f;
          p[r] = pv;
          s[r] = pv * (dp[r] - de[e]) * f.scale;                   // dS
        }
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8_t pb = acc_as_b(p, ss), dsb = acc_as_b(s, ss);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          dvt[ct] = MFMA(col_frag(sd, ct, qt, ss, lane), pb, dvt[ct]);
          dkt[ct] = MFMA(col_frag(sq, ct, qt, ss, lane), dsb, dkt[ct]);
        }
      }
    }
  }
  if (!active || !kok) return;
  if (f.qsplit > 1) {
    float* part = f.dkv_part + (((int64_t)blockIdx.y * f.B * f.H + bh) * f.N + key) * 128;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = ct * 32 + 8 * g + 4 * fhalf;
        *(float4*)(part + d) = make_float4(dkt[ct][4 * g], dkt[ct][4 * g + 1], dkt[ct][4 * g + 2], dkt[ct][4 * g + 3]);
        *(float4*)(part + 64 + d) = make_float4(dvt[ct][4 * g], dvt[ct][4 * g + 1], dvt[ct][4 * g + 2], dvt[ct][4 * g + 3]);
      }
    return;
  }
  uint16_t* rk = f.dk + (int64_t)b * f.dk_sB + (int64_t)key * f.dk_sN + (int64_t)h * 64;
  uint16_t* rv = f.dv + (int64_t)b * f.dv_sB + (int64_t)key * f.dv_sN + (int64_t)h * 64;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = ct * 32 + 8 * g + 4 * fhalf;
      *(uint2*)(rk + d) = make_uint2(pack_bf16x2(dkt[ct][4 * g], dkt[ct][4 * g + 1]), pack_bf16x2(dkt[ct][4 * g + 2], dkt[ct][4 * g + 3]));
      *(uint2*)(rv + d) = make_uint2(pack_bf16x2(dvt[ct][4 * g], dvt[ct][4 * g + 1]), pack_bf16x2(dvt[ct][4 * g + 2], dvt[ct][4 * g + 3]));
    }
}

// dK / dV = sum over the query parts, parts added in index order; one thread per (batch*head, key, 4 channels of dK|dV)
__global__ __launch_bounds__(256) void flash_bwd_dkv_reduce_kernel(const Attn2Args f) {
  const int64_t total = (int64_t)f.B * f.H * f.N * 32;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int d4 = (int)(i & 31), key = (int)((i >> 5) % f.N), bh = (int)((i >> 5) / f.N);
  const int64_t plane = (int64_t)f.B * f.H * f.N * 128;
  const float* p = f.dkv_part + ((int64_t)bh * f.N + key) * 128 + d4 * 4;
  float4 a = *(const float4*)p;
  for (int s = 1; s < f.qsplit; ++s) {
    const float4 v = *(const float4*)(p + s * plane);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  const int b = bh / f.H, h = bh % f.H;
  const bool is_v = d4 >= 16;
  const int d = (d4 & 15) * 4;
  uint16_t* dst = is_v ? f.dv + (int64_t)b * f.dv_sB + (int64_t)key * f.dv_sN + (int64_t)h * 64 + d
                       : f.dk + (int64_t)b * f.dk_sB + (int64_t)key * f.dk_sN + (int64_t)h * 64 + d;
  *(uint2*)dst = make_uint2(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w));
}

// query parts of the dK/dV kernel: enough workgroups for ~2 per CU, at least four 64-query tiles per part
int dkv_qsplit(int B, int H, int Nq, int Nkv) {
  const int64_t base = (int64_t)((Nkv + 127) / 128) * B * H;
  const int ntiles = (Nq + 63) / 64;
  int64_t s = (512 + base - 1) / base;
  if (s > ntiles / 4) s = ntiles / 4;
  if (s > 16) s = 16;
  return s < 1 ? 1 : (int)s;
}

bool strides_ok(int64_t a, int64_t b) { return a % 8 == 0 && b % 8 == 0; }

constexpr int kFlashFwdDefault = 3;
int g_flash_fwd = kFlashFwdDefault;   // A/B hook (gdl_debug_set_flash_fwd): 2 = the round-2 kernel (64-query waves), 3 = 32-query waves,
                                      // 4 / 5 = round 5, key tiles software-pipelined (Q in registers / in the LDS); < 0 = the default
float g_flash_defer = 6.f;    // deferred running maximum (see flash_fwd3_kernel); 0 = exact online softmax, bit-identical to 2

}  // namespace

extern "C" void gdl_debug_set_flash_fwd(int version, float defer) { g_flash_fwd = version < 0 ? kFlashFwdDefault : version; g_flash_defer = defer; }

extern "C" int gdl_flash_attn_fwd2(const void* q, int64_t q_sB, int64_t q_sN, const void* k, int64_t k_sB, int64_t k_sN,
                                   const void* v, int64_t v_sB, int64_t v_sN, void* o, int64_t o_sB, int64_t o_sN,
                                   float* lse, int B, int H, int Nq, int Nkv, float scale, gdl_stream_t stream) {
  GDL_CHECK_ARG(q && k && v && o, "gdl_flash_attn_fwd2: null pointer");
  GDL_CHECK_ARG(B > 0 && H > 0 && Nq > 0 && Nkv > 0, "gdl_flash_attn_fwd2: bad dims");
  GDL_CHECK_ARG(((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) && ((uintptr_t)o % 8 == 0) &&
                    strides_ok(q_sB, q_sN) && strides_ok(k_sB, k_sN) && strides_ok(v_sB, v_sN) && o_sB % 4 == 0 && o_sN % 4 == 0,
                "gdl_flash_attn_fwd2: pointers / strides must keep 16-byte alignment");
  GDL_CHECK_ARG(((int64_t)Nkv - 1) * k_sN * 2 + 128 < 0x7fffffffll && ((int64_t)Nkv - 1) * v_sN * 2 + 128 < 0x7fffffffll,
                "gdl_flash_attn_fwd2: one (batch, head) key / value slab spans more than 2 GiB");
  Attn2Args f = {};
  f.q = (const uint16_t*)q; f.k = (const uint16_t*)k; f.v = (const uint16_t*)v; f.out = (uint16_t*)o; f.lse = lse;
  f.q_sB = q_sB; f.q_sN = q_sN; f.k_sB = k_sB; f.k_sN = k_sN; f.v_sB = v_sB; f.v_sN = v_sN; f.o_sB = o_sB; f.o_sN = o_sN;
  f.B = B; f.H = H; f.Nq = Nq; f.N = Nkv;
  f.scale = scale; f.scale_log2e = scale * 1.4426950408889634f;
  // waves per block: 4 (two resident blocks fill the CU's eight wave slots; measured 285 us vs 319 us for the
  // less-padded 3-wave blocks at N = 1297), fewer only when the whole query range is shorter than that
  const int nw = Nq > 128 ? 4 : 2;
  const unsigned grid = (unsigned)((Nq + 64 * nw - 1) / (64 * nw) * B * H);
  if (g_flash_fwd >= 4 && Nq > 64) {      // round 5: S of the next key tile issued before the softmax of the current one
    const dim3 g4((unsigned)((Nq + 127) / 128 * B * H));
    if (g_flash_fwd == 5) hipLaunchKernelGGL((flash_fwd4_kernel<4, true>), g4, dim3(256), 0, (hipStream_t)stream, f, g_flash_defer);
    else hipLaunchKernelGGL((flash_fwd4_kernel<4, false>), g4, dim3(256), 0, (hipStream_t)stream, f, g_flash_defer);
  } else if (g_flash_fwd >= 3) {      // 32-query waves, four per SIMD
    if (Nq > 64) hipLaunchKernelGGL(flash_fwd3_kernel<4>, dim3((unsigned)((Nq + 127) / 128 * B * H)), dim3(256), 0, (hipStream_t)stream, f, g_flash_defer);
    else hipLaunchKernelGGL(flash_fwd3_kernel<2>, dim3((unsigned)((Nq + 63) / 64 * B * H)), dim3(128), 0, (hipStream_t)stream, f, g_flash_defer);
  } else if (nw == 4) hipLaunchKernelGGL(flash_fwd2_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, f);
  else hipLaunchKernelGGL(flash_fwd2_kernel<2>, dim3(grid), dim3(128), 0, (hipStream_t)stream, f);
  GDL_CHECK_LAUNCH("gdl_flash_attn_fwd2");
  return GDL_OK;
}

extern "C" int64_t gdl_flash_attn_bwd_workspace(int B, int H, int Nq, int Nkv) {
  if (B <= 0 || H <= 0 || Nq <= 0 || Nkv <= 0) return 0;
  const int s = dkv_qsplit(B, H, Nq, Nkv);
  return s > 1 ? (int64_t)s * B * H * Nkv * 128 * (int64_t)sizeof(float) : 0;
}

extern "C" int gdl_flash_attn_bwd(const void* q, int64_t q_sB, int64_t q_sN, const void* k, int64_t k_sB, int64_t k_sN,
                                  const void* v, int64_t v_sB, int64_t v_sN, const void* o, int64_t o_sB, int64_t o_sN,
                                  const void* dout, int64_t do_sB, int64_t do_sN, const float* lse, float* dvec,
                                  void* dq, int64_t dq_sB, int64_t dq_sN, void* dk, int64_t dk_sB, int64_t dk_sN,
                                  void* dv, int64_t dv_sB, int64_t dv_sN, int B, int H, int Nq, int Nkv, float scale,
                                  float* ws, int64_t ws_bytes, gdl_stream_t stream) {
  GDL_CHECK_ARG(q && k && v && o && dout && lse && dvec && dq && dk && dv, "gdl_flash_attn_bwd: null pointer");
  GDL_CHECK_ARG(B > 0 && H > 0 && Nq > 0 && Nkv > 0, "gdl_flash_attn_bwd: bad dims");
  GDL_CHECK_ARG(((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) && ((uintptr_t)dout % 16 == 0) &&
                    ((uintptr_t)dq % 8 == 0) && ((uintptr_t)dk % 8 == 0) && ((uintptr_t)dv % 8 == 0) &&
                    strides_ok(q_sB, q_sN) && strides_ok(k_sB, k_sN) && strides_ok(v_sB, v_sN) && strides_ok(do_sB, do_sN) &&
                    dq_sB % 4 == 0 && dq_sN % 4 == 0 && dk_sB % 4 == 0 && dk_sN % 4 == 0 && dv_sB % 4 == 0 && dv_sN % 4 == 0,
                "gdl_flash_attn_bwd: pointers / strides must keep 16-byte (inputs) / 8-byte (gradients) alignment");
  const int64_t lim = 0x7fffffffll;
  GDL_CHECK_ARG(((int64_t)Nkv - 1) * k_sN * 2 + 128 < lim && ((int64_t)Nkv - 1) * v_sN * 2 + 128 < lim &&
                    ((int64_t)Nq - 1) * q_sN * 2 + 128 < lim && ((int64_t)Nq - 1) * do_sN * 2 + 128 < lim,
                "gdl_flash_attn_bwd: one (batch, head) slab spans more than 2 GiB");
  Attn2Args f = {};
  f.q = (const uint16_t*)q; f.k = (const uint16_t*)k; f.v = (const uint16_t*)v; f.o = (const uint16_t*)o;
  f.dout = (const uint16_t*)dout; f.lse = const_cast<float*>(lse); f.dvec = dvec;
  f.dq = (uint16_t*)dq; f.dk = (uint16_t*)dk; f.dv = (uint16_t*)dv;
  f.q_sB = q_sB; f.q_sN = q_sN; f.k_sB = k_sB; f.k_sN = k_sN; f.v_sB = v_sB; f.v_sN = v_sN; f.o_sB = o_sB; f.o_sN = o_sN;
  f.do_sB = do_sB; f.do_sN = do_sN; f.dq_sB = dq_sB; f.dq_sN = dq_sN; f.dk_sB = dk_sB; f.dk_sN = dk_sN; f.dv_sB = dv_sB; f.dv_sN = dv_sN;
  f.B = B; f.H = H; f.Nq = Nq; f.N = Nkv;
  f.scale = scale; f.scale_log2e = scale * 1.4426950408889634f;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(flash_bwd_dot_kernel, dim3((unsigned)(((int64_t)B * Nq + 3) / 4)), dim3(256), 0, s, f);
  hipLaunchKernelGGL(flash_bwd_dq_kernel, dim3((unsigned)((Nq + 127) / 128 * B * H)), dim3(256), 0, s, f);
  f.qsplit = dkv_qsplit(B, H, Nq, Nkv);
  f.dkv_part = ws;
  GDL_CHECK_ARG(f.qsplit == 1 || (ws && ((uintptr_t)ws % 16 == 0) && ws_bytes >= gdl_flash_attn_bwd_workspace(B, H, Nq, Nkv)),
                "gdl_flash_attn_bwd: workspace of gdl_flash_attn_bwd_workspace() bytes needed (16-byte aligned)");
  hipLaunchKernelGGL(flash_bwd_dkv_kernel, dim3((unsigned)((Nkv + 127) / 128 * B * H), (unsigned)f.qsplit), dim3(256), 0, s, f);
  if (f.qsplit > 1)
    hipLaunchKernelGGL(flash_bwd_dkv_reduce_kernel, dim3((unsigned)(((int64_t)B * H * Nkv * 32 + 255) / 256)), dim3(256), 0, s, f);
  GDL_CHECK_LAUNCH("gdl_flash_attn_bwd");
  return GDL_OK;
}
