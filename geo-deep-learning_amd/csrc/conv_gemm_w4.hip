// Implicit-GEMM 256 x 256 tile with ONE wave per SIMD (round 4): four waves, wave tile 128 x 128 (256 accumulator registers
// of the 512 a lone wave may use), and a K loop in which EVERY non-MFMA instruction sits in the shadow of an MFMA.
//
// Why.  The 8-wave ping-pong kernel (conv_gemm.hip) runs a K-step in 2800-2900 cycles against 2048 of MFMA issue; its
// stand-alone skeleton (tools/probes/kloop_probe.hip) in 2533.  What it cannot avoid: two waves per SIMD that meet at a barrier
// every K-step, a loader whose burst of 16 DMA pieces (~100 cycles each in that context) is one serial chain with its own
// MFMAs, and 6 fragment reads per 8 MFMAs.  The skeleton of THIS structure (tools/probes/kloop4_probe.hip,
// profiles/r04d_kloop_one_wave_per_simd_probe.txt) measures, per K-step: MFMAs alone 2062; fragment reads and DMA issued as
// bursts 3289 (a lone in-order wave issues nothing else while it issues those); the same instructions placed one group behind
// each MFMA -- `M r M r M d M d` -- 2158 = 95 % of the matrix pipe.  A wave issues, per K-step, 64 MFMAs, 32 ds_read_b128 (8 per
// k16 group: a third fewer LDS reads per MFMA than the 128 x 64 wave tile) and 16 of the 64 DMA pieces.
//
// Schedule of one K-step (tile kt in LDS stage kt & 1; k16 groups kk = 0..3; micro-group q = the four MFMAs of activation
// fragment q against the four weight fragments):
//   behind MFMA 0 and 1 of every micro-group: one fragment read each of the NEXT k16 group (weights in q = 0, 1; activations in
//     q = 2, 3; group 3 reads group 0 of tile kt+1, after the barrier), double-buffered registers;
//   behind MFMA 2 and 3: DMA pieces of tile kt+1 -- 6 in group 0, 6 in group 1 (group 2 is landing time), and in group 3, after
//     the barrier that freed the stage, the first 4 pieces of tile kt+2;
//   after group 2: wait for the own pieces of tile kt+1, ONE barrier.
// Everything else (LDS-DMA through buffer descriptors with hardware zero fill, source-side swizzle, swapped MFMA operands,
// K order chunk-outer / tap-inner, the coalesced LDS-transposed epilogue, XCD-aware tile order with L2-sized N groups) is
// conv_gemm.hip's.  The epilogue runs per 64-channel half of the wave tile through the same conv_epilogue code.
#include "conv_gemm_common.h"

using namespace gdlconv;

namespace {

constexpr int W4_STAGE = 65536;          // [A: 256 rows | B: 256 rows] x 128 B
constexpr int W4_LDS = 2 * W4_STAGE;

template <bool DENSE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_gemm_w4_kernel(const KArgs k) {
  constexpr int ES = 2, BKE = 64;
  constexpr unsigned kOob = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const gdl_conv_args& a = k.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int lid = xcd_remap(blockIdx.x, k.tiles_m * k.tiles_n);
  int tile_m, tile_n;
  tile_order(k, lid, tile_m, tile_n);
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int z = blockIdx.y;
  const int z0 = z / a.nz_inner, z1 = z % a.nz_inner;
  const int64_t out_zoff = z0 * a.out_sZ0 + z1 * a.out_sZ1;
  const srd_t srd_a = make_srd((const unsigned char*)a.in + (z0 * a.in_sZ0 + z1 * a.in_sZ1) * ES, k.in_span);
  const srd_t srd_b = make_srd((const unsigned char*)a.w + (z0 * a.w_sZ0 + z1 * a.w_sZ1) * ES, k.w_span);
  const unsigned lds_base =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);

  // ---- DMA geometry: wave w, piece i (0..7 per operand) covers tile rows (i*4 + w)*8 .. +7; lane l writes LDS slot (l & 7) of
  //      row +(l >> 3) and therefore fetches source chunk slot ^ swz(row)
  const int lrow = lane >> 3, lslot = lane & 7;
  const int HoWo = a.Ho * a.Wo;
  int a_voff[8];
  unsigned a_mask[DENSE ? 1 : 8];
  unsigned b_voff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = (i * 4 + wave) * 8 + lrow;
    const int chunk = lslot ^ ((r >> 1) & 7);
    const int m = m0 + r;
    const bool ok = m < k.M;
    const int mm = ok ? m : 0;
    if constexpr (DENSE) {
      a_voff[i] = ok ? (int)((mm * a.in_sW + chunk * 8) * ES) : (int)kOob;
    } else {
      const int b = mm / HoWo, rem = mm - b * HoWo, oy = rem / a.Wo, ox = rem - oy * a.Wo;
      const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
      a_voff[i] = (int)((b * a.in_sB + iy0 * a.in_sH + ix0 * a.in_sW + chunk * 8) * ES);
      unsigned mask = 0;
      if (ok) {
        if (a.pad == 0) {
          mask = 0xffffffffu;
        } else {
          for (int tr = 0; tr < a.R; ++tr)
            for (int ts = 0; ts < a.S; ++ts)
              if ((unsigned)(iy0 + tr) < (unsigned)a.H && (unsigned)(ix0 + ts) < (unsigned)a.W)
                mask |= 1u << (tr * a.S + ts);
        }
      }
      a_mask[i] = mask;
    }
    const int n = n0 + r;
    b_voff[i] = n < a.N ? (unsigned)((n * a.w_sN + chunk * 8) * ES) : kOob;
  }

  // ---- the tile whose pieces are being issued (a tile's 16 pieces per wave are spread over two K-steps)
  int tap_r = 0, tap_s = 0, cc = 0;       // position of the NEXT tile to begin
  int cur_off = 0;                        // activation byte offset of the current tile's tap / chunk
  unsigned cur_bit = 1u, cur_wk = 0;      // its tap-validity bit and its byte offset along the weight rows
  auto begin_tile = [&]() {
    // (readfirstlane: the values ARE wave-uniform; said so, they live in scalar registers instead of being re-broadcast
    // from a vector register in front of every DMA piece)
    if constexpr (DENSE) {
      cur_wk = __builtin_amdgcn_readfirstlane((unsigned)(cc * BKE * ES));
      ++cc;
    } else {
      cur_off = __builtin_amdgcn_readfirstlane((int)((tap_r * a.in_sH + tap_s * a.in_sW + cc * BKE) * ES));
      cur_bit = __builtin_amdgcn_readfirstlane(a.pad == 0 ? 1u : 1u << (tap_r * a.S + tap_s));
      cur_wk = __builtin_amdgcn_readfirstlane((unsigned)(((tap_r * a.S + tap_s) * a.C + cc * BKE) * ES));
      if (++tap_s == a.S) { tap_s = 0; if (++tap_r == a.R) { tap_r = 0; ++cc; } }
    }
  };
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds_base + wave * 1024);
  auto piece = [&](int stage, int p) {    // p is a compile-time constant at every call site: 0..7 activations, 8..15 weights
    const unsigned lds = lds_wave + stage * W4_STAGE + ((p < 8 ? 0 : 32768) + (p & 7) * 4096);
    if (p < 8) {
      if constexpr (DENSE) {
        dma16_buf((unsigned)a_voff[p], srd_a, cur_wk, lds);
      } else {
        const unsigned v = (a_mask[p] & cur_bit) ? (unsigned)(a_voff[p] + cur_off) : kOob;
        dma16_buf(v, srd_a, 0u, lds);
      }
    } else {
      dma16_buf(b_voff[p - 8], srd_b, cur_wk, lds);
    }
  };

  f32x16_t acc[2][4][2];                  // [64-channel half of the wave's 128 columns][32-row tile][32-channel tile]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5;
  const int fswz = (frow >> 1) & 7;
  const int a_lds0 = (wm * 128 + frow) * 128;
  const int b_lds0 = 32768 + (wn * 128 + frow) * 128;
  uint4 fa[2][4], fb[2][4];
  auto frag_off = [&](int kk) { return ((2 * kk + fhalf) ^ fswz) << 4; };

  const unsigned long long t0c = k.probe ? __builtin_readcyclecounter() : 0;
  const unsigned long long t0r = k.probe ? __builtin_amdgcn_s_memrealtime() : 0;
  // ---- prologue: tile 0 complete, the first four pieces of tile 1, the fragments of tile 0's first k16 group
  begin_tile();
#pragma unroll
  for (int p = 0; p < 16; ++p) piece(0, p);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (k.KT > 1) {
    begin_tile();
#pragma unroll
    for (int p = 0; p < 4; ++p) piece(1, p);
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the scalar argument loads are complete (the waitcnt inserter then counts LDS reads only)
  {
    const int co = frag_off(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[0][i] = *(const uint4*)(smem + a_lds0 + i * 4096 + co);
      fb[0][i] = *(const uint4*)(smem + b_lds0 + i * 4096 + co);
    }
  }

  // One K-step.  T1: tile kt+1 exists (its pieces 4..15 go out in groups 0 and 1, its first fragments are read in group 3, the
  // step closes with wait + barrier); T2: tile kt+2 exists (begun in group 3).  COMPILE-TIME flags: the steady-state loop body
  // has no branch at all, the last two steps are separate instantiations.  Measured alternatives, all slower: run-time flags
  // (hipcc routes them through v_cndmask / v_cmp / s_cbranch in front of every DMA statement: 3296 instead of 2499 cycles per
  // K-step at K = 768); one instantiation with the pieces of tiles past the end issued as out-of-range zero-fill loads (3165)
  // or executed under an all-zero EXEC mask (3314).
  auto step = [&](auto t1c, auto t2c, int kt) {
    constexpr bool T1 = decltype(t1c)::value, T2 = decltype(t2c)::value;
    const unsigned char* st = smem + (kt & 1) * W4_STAGE;
    const unsigned char* nx = smem + ((kt + 1) & 1) * W4_STAGE;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int cb = kk & 1, nb = (kk + 1) & 1;
      const unsigned char* rs = kk < 3 ? st : nx;                 // where the next group's fragments live
      const int rco = frag_off(kk < 3 ? kk + 1 : 0);
      if (kk == 3 && T2) begin_tile();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        auto mf = [&](int j) {
          acc[j >> 1][q][j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[cb][j]),
                                                                          __builtin_bit_cast(bf16x8_t, fa[cb][q]), acc[j >> 1][q][j & 1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        };
        auto rd1 = [&](int e) {                                   // fragment read e (0 / 1) of this micro-group
          if (kk == 3 && !T1) return;
          if (q < 2) fb[nb][2 * q + e] = *(const uint4*)(rs + b_lds0 + (2 * q + e) * 4096 + rco);
          else fa[nb][2 * (q - 2) + e] = *(const uint4*)(rs + a_lds0 + (2 * (q - 2) + e) * 4096 + rco);
        };
        mf(0);
        rd1(0);
        __builtin_amdgcn_sched_barrier(0);
        mf(1);
        rd1(1);
        __builtin_amdgcn_sched_barrier(0);
        mf(2);
        if (kk == 3) { if (T2) piece(kt & 1, q); }
        else if (kk < 2) { if (T1) piece((kt + 1) & 1, 4 + kk * 6 + (q >> 1) * 3 + (q & 1) * 2); }
        __builtin_amdgcn_sched_barrier(0);
        mf(3);
        if (kk < 2 && (q & 1) == 0) { if (T1) piece((kt + 1) & 1, 4 + kk * 6 + (q >> 1) * 3 + 1); }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kk == 2 && T1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile kt+1 have landed
        __builtin_amdgcn_s_barrier();                      // ... and everyone's; nobody reads tile kt's stage any more (group 3's
                                                           // fragments are in registers)
      }
    }
  };
  using yes = std::integral_constant<bool, true>;
  using no = std::integral_constant<bool, false>;
  int kt = 0;
  for (; kt + 2 < k.KT; ++kt) step(yes{}, yes{}, kt);
  if (kt + 1 < k.KT) { step(yes{}, no{}, kt); ++kt; }
  step(no{}, no{}, kt);
  if (k.probe && tid == 0 && blockIdx.x < 2048) {
    k.probe[2 * blockIdx.x] = __builtin_readcyclecounter() - t0c;
    k.probe[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
  }
  __builtin_amdgcn_s_barrier();   // every wave is past its last fragment read: the stages become the epilogue's transpose buffers
  // One 64-channel half of the wave tile after the other, through the 8-wave kernels' epilogue code (static indices: a run-time
  // index into the accumulator array sends all 256 accumulators through scratch memory).  With four waves instead of eight
  // this takes ~1.75 x the 8-wave epilogue (10.7 k vs 6.2 k cycles for a bf16 tile): a lone wave has no partner to cover the
  // LDS-round-trip -> VALU -> store chain of a pass.  Tried and measured, no better: both halves in one instruction stream (the
  // accumulators live in AGPRs and leave the epilogue 256 VGPRs: spills, 15 k / 144 k cycles), all four passes parked in the LDS
  // before the first row group is finished (12.3 k).  So this tile is only chosen where the K loop dominates (>= 40 K-steps).
  conv_epilogue<4, 2, false>(k, acc[0], m0, n0, wm, 2 * wn, lane, out_zoff, smem + (2 * wave) * 8192, smem + 8 * 8192 + (2 * wave) * 1024);
  conv_epilogue<4, 2, false>(k, acc[1], m0, n0, wm, 2 * wn + 1, lane, out_zoff, smem + (2 * wave + 1) * 8192,
                             smem + 8 * 8192 + (2 * wave + 1) * 1024);
  if (k.probe && tid == 0 && blockIdx.x < 2048) {
    k.probe[4096 + blockIdx.x] = __builtin_readcyclecounter() - t0c;
    k.probe[8192 + 2 * blockIdx.x] = t0r;                                   // block timeline (100 MHz ticks)
    k.probe[8192 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
}

}  // namespace

namespace gdlconv {

bool conv_gemm_w4_applicable(const gdl_conv_args& a) {
  return a.dtype == GDL_BF16 && a.C % 64 == 0 && !a.aux_out && a.act != GDL_ACT_MUL_GELU_GRAD &&
         (a.pad == 0 || a.R * a.S <= 32);
}

int conv_gemm_w4_launch(const KArgs& k, hipStream_t stream) {
  KArgs kk = k;
  kk.tiles_m = (k.M + 255) / 256;
  kk.tiles_n = (k.a.N + 255) / 256;
  kk.n_group = conv_n_group(k.a, 256, 256, 32);
  dim3 grid(kk.tiles_m * kk.tiles_n, k.a.nz), block(256);
  if (k.in_dense) {
    GDL_SET_MAX_LDS_ONCE(conv_gemm_w4_kernel<true>, W4_LDS);
    hipLaunchKernelGGL(conv_gemm_w4_kernel<true>, grid, block, W4_LDS, stream, kk);
  } else {
    GDL_SET_MAX_LDS_ONCE(conv_gemm_w4_kernel<false>, W4_LDS);
    hipLaunchKernelGGL(conv_gemm_w4_kernel<false>, grid, block, W4_LDS, stream, kk);
  }
  GDL_CHECK_LAUNCH("gdl_conv_gemm(256x256, one wave per SIMD)");
  return GDL_OK;
}

}  // namespace gdlconv
