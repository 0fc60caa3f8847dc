// Persistent 256 x 256 tile with ONE wave per SIMD whose output stores are OFF the critical path (round 6): dense 1x1 layers /
// linears in bf16 with bf16 outputs and at least 12 K-steps (ViT qkv, the neck's tap products and 1x1 convolutions, laterals).
//
// Why.  Every 256^2 kernel so far ends a tile with a phase in which the CU issues no MFMA at all: 8.4 k cycles for a bf16 tile in the
// persistent 8-wave kernel (conv_gemm_persist.hip), 11.3 k with four waves (conv_gemm_w4.hip) -- against 30 k cycles for the 12 K-steps
// of a K = 768 layer.  The persistent 8-wave kernel hides the NEXT tile's prologue under the epilogue, not the epilogue itself: two
// waves per SIMD leave 256 registers per wave, 128 of them accumulators, and nowhere to keep a finished tile.
//
// Here a wave is alone on its SIMD (512 registers: 256 accumulators in the AGPR half, conv_gemm_w4.hip's K loop at 2.3 k cycles per
// K-step) and the tile leaves in two phases:
//   phase 1 "park" (exposed, 6.4 k cycles): raw accumulators -> wave-private LDS transpose -> store layout (a lane owns 16 consecutive
//     bytes of an output row) -> bias (+ folded BatchNorm + ReLU, + BatchNorm partial statistics) -> packed bf16 in 128 HOLD
//     registers.  No global memory access.  (It is bound by the LDS: 256 KiB of f32 accumulators per tile through ds_write_b128 at
//     ~79 B/clk/CU are 3.3 k cycles by themselves.)
//   phase 2 "drain" (hidden): the hold registers leave as full-line `buffer_store_dwordx4` in the MFMA shadows of the NEXT tile's
//     K loop, three per K-step behind the MFMAs that follow the step's `vmcnt(0)` + barrier -- stores share vmcnt with the DMA, and
//     issued in front of the wait they made every wave sit out their round trip; issued behind it they have a whole K-step.
// Measured (tools/probe_gemm_timeline.py parked, tools/bench_w4p.py; batch 32): qkv 143.7 -> 134.5 us (1022 -> 1092 TF/s), the
// neck's tap products 414.9 -> 385.6 us, its 1x1 convolutions 51.9 -> 50.0 us, the K-loop is not slowed by the stores (2292 vs
// 2276 cycles per K-step with the stores dropped).
//
// STATUS: opt-in (GDL_CONV_W4P=1 / gdl_debug_set_conv_w4p).  Faster per layer, but the whole training / inference step does not
// move: three same-box A/B repetitions each at per-GPU batch 32 (879.3-880.0 vs 879.5-880.3 train tiles/s, +0.3 % inference) and 64
// (951.6 vs 949.6).  These GEMMs run at the chip's 1400 W power cap at ~1.83 GHz (tools/probe_power_clock.sh: the same
// instruction stream on zero operands draws 1080 W at 2.39 GHz and runs 26 % faster): an instruction stream with fewer stall cycles
// does the same bit flips per tile, and the cycles it saves come back as a lower clock.  The kernel stays as a tested variant and
// as the record of what "take the epilogue off the critical path" buys on this chip.
//
// What was built beside it, measured and REMOVED (profiles/r06*):
//   * the same for 4 / 8 K-steps (K = 256 decoder layers: eight stores per K-step): those layers are HBM-bound at the 3-3.4 TB/s this
//     read/write mix reaches, the 8-wave kernel's store burst serves them better (dgrad lateral 144: 383 vs 406 us);
//   * the f32 residual stream of ViT proj / fc2 (hold = bf16(acc + bias), phase 2 = unpack, LayerScale, DropPath, residual rows by
//     LDS-DMA into spare LDS a K-step ahead, f32 store): correct and bit-identical to the other tiles under the same rounding, per
//     tile faster (proj 35.4 -> 25.8 us, fc2 93 -> 86 us) -- but with 1.9 tiles per CU the last tile's drain has no K loop to hide in
//     and is a chain of memory round trips: proj 85 -> 88-99 us, fc2 196 -> 208-215 us.  proj moves 318 MB in 85 us in EVERY variant:
//     it sits on the ~3.7 TB/s the f32 stream's read + write mix gets, not on its epilogue;
//   * a "fence load" (one dummy LDS-DMA load as the youngest operation + `vmcnt(1)`: loads complete in order, so older loads are
//     guaranteed complete while stores may stay in flight): no effect once the stores sat behind the wait;
//   * an L2 prefetch of the residual rows two K-steps ahead: slower.
//
// Everything outside the two phases is conv_gemm_w4.hip's / conv_gemm_persist.hip's: LDS-DMA through buffer descriptors (rows past M
// are out of range of num_records and arrive as zeros), source-side swizzle, swapped MFMA operands, `M r M r M d M d` micro-groups,
// one barrier per K-step, XCD-aware tile order with L2-sized N groups, next tile's first stage in flight under phase 1.  Stores of
// rows past M are dropped by the output descriptor's range check (no tail epilogue), and the hold registers of the first tile --
// which hold nothing yet -- drain through a descriptor with num_records = 0.
//
// Hold registers are indexed statically, so every K-step that drains is its own instantiation of the step body: eleven, three row
// groups each.  Three compiler facts shaped the code (each was a spill of dozens to hundreds of registers before):
//   * uniform terms derived per use from scalars that an empty asm makes opaque per K-step (`Fresh`): left visible, hipcc hoists the
//     32 piece addresses / offsets and the 32 row-group offsets in front of the tile loop and parks them in scratch memory;
//   * per-lane terms that are needed once per tile or once per K-step re-derived from an opaque thread id;
//   * "exactly 12 K-steps" vs "more" is a template parameter: a run-time branch between the two K-loop tails is a control-flow
//     diamond across which hipcc moved 192 accumulators through scratch memory.
#include "conv_gemm_common.h"

using namespace gdlconv;

namespace {

constexpr int P_STAGE = 65536;                 // [A: 256 rows | B: 256 rows] x 128 B
constexpr int P_LDS = 2 * P_STAGE;

typedef __attribute__((ext_vector_type(4))) unsigned pu4;

template <int V> using ic = std::integral_constant<int, V>;
using yes = std::integral_constant<bool, true>;
using no = std::integral_constant<bool, false>;

// SR: folded BatchNorm scale / shift + ReLU in phase 1; ST: BatchNorm partial statistics of the rounded outputs (layout of
// EPI_STATS); K12: the call has exactly 12 K-steps (else more).
template <bool SR, bool ST, bool K12>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_gemm_w4p_kernel(const KArgs k) {
#pragma clang fp contract(off)
  constexpr int ES = 2;
  constexpr int CH = 8;                        // channels per lane in the store layout (16 bytes of output)
  constexpr int OES = 2;
  constexpr int LPR = 64 / CH;                 // lanes per 64-channel row segment
  constexpr int RPG = 64 / LPR;                // rows per row group (one store instruction): 8
  constexpr int NT = 32 / RPG;                 // row groups per 32-row pass: 4
  constexpr int HW = 4;                        // hold registers per row group
  constexpr int UNITS = 8 * NT;                // row groups per wave tile: 32
  constexpr int UPS = 3;                       // row groups drained per K-step: the first eleven K-steps of a tile drain
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const gdl_conv_args& a = k.a;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = k.tiles_m * k.tiles_n;
  const srd_t srd_a = make_srd(a.in, k.in_span);
  const srd_t srd_b = make_srd(a.w, k.w_span);
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds_base + wave * 1024);

  // ---- DMA geometry (conv_gemm_persist.hip): piece i (0..7 per operand) of wave w covers tile rows (i*4 + w)*8 .. +7; lane l writes
  //      LDS slot (l & 7) of row +(l >> 3) and fetches source chunk slot ^ swz(row) -- the same chunk for every piece; the piece's
  //      row offset is wave-uniform and rides in soffset (range-checked together with voffset)
  const unsigned a_row = (unsigned)a.in_sW * ES, b_row = (unsigned)a.w_sN * ES;
  // (per-lane terms that are only needed now and then are re-derived from an opaque thread id where they are used: 128 hold + 64
  // fragment registers leave the K loop no room for passengers)
  auto lane_ab = [&](unsigned& al, unsigned& bl) {
    int t2 = threadIdx.x;
    asm volatile("" : "+v"(t2));
    const int l2 = t2 & 63, lr = l2 >> 3, ls = l2 & 7;
    const int ch = ls ^ ((((wave & 1) << 2) + (lr >> 1)) & 7);
    al = (unsigned)(wave * 8 + lr) * a_row + ch * 16;
    bl = (unsigned)(wave * 8 + lr) * b_row + ch * 16;
  };
  unsigned a_v = 0, b_v = 0;                   // lane offset + the tile's first row
  unsigned cur_wk = 0;                         // byte offset along K of the K-step whose pieces are being issued
  int cc = 0;
  auto begin_kstep = [&]() { cur_wk = __builtin_amdgcn_readfirstlane((unsigned)(cc * 128)); ++cc; };
  struct Fresh { unsigned lw, ar, br, orow; };   // wave's LDS base, operand row strides, output row stride (bytes)
  auto fresh = [&]() {
    Fresh f{lds_wave, a_row, b_row, (unsigned)a.out_sW * (unsigned)OES};
    asm volatile("" : "+s"(f.lw), "+s"(f.ar), "+s"(f.br), "+s"(f.orow));
    return f;
  };
  auto piece_at = [&](const Fresh& f, int stage, int p, unsigned av, unsigned bv, unsigned wk) {   // p compile-time: 0..7 activations, 8..15 weights
    const unsigned lds = f.lw + stage * P_STAGE + ((p < 8 ? 0 : 32768) + (p & 7) * 4096);
    if (p < 8) dma16_buf(av, srd_a, wk + (unsigned)(p * 32) * f.ar, lds);
    else dma16_buf(bv, srd_b, wk + (unsigned)((p - 8) * 32) * f.br, lds);
  };

  f32x16_t acc[2][4][2];                       // [64-channel half][32-row block][32-channel block]
  uint32_t hold[8][NT][HW];                    // the parked tile: pass (half * 4 + row block), row group, packed bf16
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < HW; ++e) hold[p][t][e] = 0u;

  uint4 fa[2][4], fb[2][4];
  // fragment addresses: row base + 16-byte slot ((2 kk + fhalf) ^ fswz) = (base | (fhalf ^ fswz) << 4) ^ (kk << 5) -- ONE register per
  // operand and a v_xor at the read: left to itself hipcc keeps all 2 stages x 4 groups x 2 operands addresses in registers across
  // the K loop
  auto frag_bases = [&](unsigned& fab, unsigned& fbb) {
    int t2 = threadIdx.x;
    asm volatile("" : "+v"(t2));
    const int l2 = t2 & 63, fr = l2 & 31, fh = l2 >> 5, sw = (fr >> 1) & 7;
    fab = (unsigned)((wm * 128 + fr) * 128) + (unsigned)((fh ^ sw) << 4);
    fbb = (unsigned)(32768 + (wn * 128 + fr) * 128) + (unsigned)((fh ^ sw) << 4);
  };

  // ---- drain state of the PARKED tile (set by park()): per-lane byte offset of (first row of the wave tile + lane / 8, first
  //      channel + 8 (lane % 8)) and the output descriptor's num_records -- 0 while nothing is parked
  unsigned d_out_lane = 0, d_span = 0;
  // row group u = pass * NT + t; pass = half * 4 + row block q
  auto unit_out_off = [&](const Fresh& f, int u) -> unsigned {   // uniform
    const int p = u / NT, t = u % NT, h = p >> 2, q = p & 3;
    return (unsigned)(q * 32 + RPG * t) * f.orow + (unsigned)(h * 64 * OES);
  };
  auto drain_store = [&](const Fresh& f, auto uc) {
    constexpr int u = decltype(uc)::value;
    if constexpr (u < UNITS) {
      constexpr int p = u / NT, t = u % NT;
      const pu4 o = {hold[p][t][0], hold[p][t][1], hold[p][t][2], hold[p][t][3]};
      // (readfirstlane: d_span IS uniform; said so, no waterfall loop is built around the store)
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)__builtin_amdgcn_readfirstlane(d_span), 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(o, rs, d_out_lane + unit_out_off(f, u), 0, 0);
    }
  };

  // ---- one K-step (conv_gemm_w4.hip).  T1: tile kt+1 exists; T2: tile kt+2 exists; DS >= 0: this step stores row groups
  //      3 DS .. 3 DS + 2 of the parked tile behind MFMAs of its LAST k16 group, i.e. behind the step's `vmcnt(0)` + barrier
  auto step = [&](auto t1c, auto t2c, auto dsc, int kt) {
    constexpr bool T1 = decltype(t1c)::value, T2 = decltype(t2c)::value;
    constexpr int DS = decltype(dsc)::value;
    const unsigned so_cur = (unsigned)(kt & 1) * P_STAGE, so_nxt = (unsigned)((kt + 1) & 1) * P_STAGE;
    unsigned fab, fbb;
    frag_bases(fab, fbb);
    const Fresh f = fresh();
    auto piece = [&](int stage, int p) { piece_at(f, stage, p, a_v, b_v, cur_wk); };
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int cb = kk & 1, nb = (kk + 1) & 1;
      const unsigned rso = kk < 3 ? so_cur : so_nxt;              // where the next group's fragments live
      const unsigned rx = (unsigned)((kk < 3 ? kk + 1 : 0) << 5);
      if (kk == 3 && T2) begin_kstep();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        auto fill = [&](int m) {
          if constexpr (DS >= 0) {
            if (kk == 3) {
              const int g = q * 4 + m;         // MFMA 0..15 of the group
              if (g == 1) drain_store(f, ic<DS * UPS + 0>{});
              if (g == 7) drain_store(f, ic<DS * UPS + 1>{});
              if (g == 13) drain_store(f, ic<DS * UPS + 2>{});
            }
          }
        };
        auto mf = [&](int j) {
          acc[j >> 1][q][j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[cb][j]),
                                                                          __builtin_bit_cast(bf16x8_t, fa[cb][q]), acc[j >> 1][q][j & 1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        };
        auto rd1 = [&](int e) {
          if (kk == 3 && !T1) return;
          if (q < 2) fb[nb][2 * q + e] = *(const uint4*)(smem + ((fbb ^ rx) + rso) + (2 * q + e) * 4096);
          else fa[nb][2 * (q - 2) + e] = *(const uint4*)(smem + ((fab ^ rx) + rso) + (2 * (q - 2) + e) * 4096);
        };
        mf(0);
        rd1(0);
        fill(0);
        __builtin_amdgcn_sched_barrier(0);
        mf(1);
        rd1(1);
        fill(1);
        __builtin_amdgcn_sched_barrier(0);
        mf(2);
        if (kk == 3) { if (T2) piece(kt & 1, q); }
        else if (kk < 2) { if (T1) piece((kt + 1) & 1, 4 + kk * 6 + (q >> 1) * 3 + (q & 1) * 2); }
        fill(2);
        __builtin_amdgcn_sched_barrier(0);
        mf(3);
        if (kk < 2 && (q & 1) == 0) { if (T1) piece((kt + 1) & 1, 4 + kk * 6 + (q >> 1) * 3 + 1); }
        fill(3);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kk == 2 && T1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own pieces of tile kt+1 have landed (and the previous step's stores are acknowledged)
        __builtin_amdgcn_s_barrier();
      }
    }
  };

  // ---- phase 1: accumulators -> wave-private LDS transpose (two 8 KiB buffers in stage 1, which the next tile's first stage
  //      does not touch) -> store layout -> element-wise terms -> hold registers.  Iteration p issues the writes of pass p + 1
  //      (straight from the accumulator registers) and the reads of pass p and finishes pass p - 1 from registers (the 64
  //      fragment registers are idle here), with no wait between a wave's own ds_write and ds_read: one wave's LDS operations
  //      execute in issue order.
  auto park = [&](int m0, int n0) {
    int tid_p = threadIdx.x;
    asm volatile("" : "+v"(tid_p));
    const int lane = tid_p & 63, frow = lane & 31, fhalf = lane >> 5, srow = lane / LPR, slot = lane % LPR;
    unsigned char* tb = smem + P_STAGE + wave * 16384;
    const int n_w = n0 + wn * 128;
    float cb[2][CH], cs[2][CH], chh[2][CH];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int qd = 0; qd < CH / 4; ++qd) {
        const int n = n_w + h * 64 + CH * slot + 4 * qd;
        float4 tbv = make_float4(0.f, 0.f, 0.f, 0.f), tsv = make_float4(1.f, 1.f, 1.f, 1.f), thv = tbv;
        if (a.bias) tbv = *(const float4*)(a.bias + n);
        if (SR && a.scale) tsv = *(const float4*)(a.scale + n);
        if (SR && a.shift) thv = *(const float4*)(a.shift + n);
        cb[h][4 * qd] = tbv.x; cb[h][4 * qd + 1] = tbv.y; cb[h][4 * qd + 2] = tbv.z; cb[h][4 * qd + 3] = tbv.w;
        cs[h][4 * qd] = tsv.x; cs[h][4 * qd + 1] = tsv.y; cs[h][4 * qd + 2] = tsv.z; cs[h][4 * qd + 3] = tsv.w;
        chh[h][4 * qd] = thv.x; chh[h][4 * qd + 1] = thv.y; chh[h][4 * qd + 2] = thv.z; chh[h][4 * qd + 3] = thv.w;
      }
    float st1[2][8], st2[2][8];
    if constexpr (ST) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) { st1[h][e] = 0.f; st2[h][e] = 0.f; }
    }
    float4 sv[2][NT][CH / 4];                  // the transposed values of two passes in flight
    auto wr = [&](auto pc) {
      constexpr int p = decltype(pc)::value, h = p >> 2, q = p & 3;
      unsigned char* b = tb + (p & 1) * 8192;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int sl = 8 * j + 2 * g + fhalf;
          *(float4*)(b + frow * 256 + ((sl ^ (frow & 7)) << 4)) =
              make_float4(acc[h][q][j][4 * g], acc[h][q][j][4 * g + 1], acc[h][q][j][4 * g + 2], acc[h][q][j][4 * g + 3]);
        }
    };
    auto rd = [&](auto pc) {
      constexpr int p = decltype(pc)::value;
      const unsigned char* b = tb + (p & 1) * 8192;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int rr = srow + RPG * t;
#pragma unroll
        for (int qd = 0; qd < CH / 4; ++qd) {
          const int sl = (CH / 4) * slot + qd;
          sv[p & 1][t][qd] = *(const float4*)(b + rr * 256 + ((sl ^ (rr & 7)) << 4));
        }
      }
    };
    auto fin = [&](auto pc) {
      constexpr int p = decltype(pc)::value, h = p >> 2;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float v[CH];
#pragma unroll
        for (int qd = 0; qd < CH / 4; ++qd) {
          const float4 x = sv[p & 1][t][qd];
          v[4 * qd] = x.x; v[4 * qd + 1] = x.y; v[4 * qd + 2] = x.z; v[4 * qd + 3] = x.w;
        }
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          float x = __builtin_fmaf(v[e], a.alpha, cb[h][e]);
          if constexpr (SR) { x = __builtin_fmaf(x, cs[h][e], chh[h][e]); x = fmaxf(x, 0.f); }
          v[e] = x;
        }
#pragma unroll
        for (int e = 0; e < HW; ++e) hold[p][t][e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
        if constexpr (ST) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(hold[p][t][e] << 16), hi = __uint_as_float(hold[p][t][e] & 0xffff0000u);
            st1[h][2 * e] += lo; st2[h][2 * e] = __builtin_fmaf(lo, lo, st2[h][2 * e]);
            st1[h][2 * e + 1] += hi; st2[h][2 * e + 1] = __builtin_fmaf(hi, hi, st2[h][2 * e + 1]);
          }
        }
      }
    };
    auto iter = [&](auto pc) {                 // p = 0..8
      constexpr int p = decltype(pc)::value;
      if constexpr (p + 1 < 8) wr(ic<p + 1>{});
      __builtin_amdgcn_wave_barrier();
      if constexpr (p < 8) rd(pc);
      __builtin_amdgcn_wave_barrier();
      if constexpr (p >= 1) fin(ic<p - 1>{});
    };
    wr(ic<0>{});
    iter(ic<0>{}); iter(ic<1>{}); iter(ic<2>{}); iter(ic<3>{}); iter(ic<4>{}); iter(ic<5>{}); iter(ic<6>{}); iter(ic<7>{}); iter(ic<8>{});
    if constexpr (ST) {
      // lanes with equal (lane & 7) hold the same 8 channels: fold lane bits 3, 4, 5 in a fixed order (conv_epilogue_rows2's order)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
          for (int o2 = 8; o2 < 64; o2 <<= 1) { st1[h][e] += __shfl_xor(st1[h][e], o2, 64); st2[h][e] += __shfl_xor(st2[h][e], o2, 64); }
        }
        if (lane < 8) {
          const int64_t row = (int64_t)(m0 + wm * 128) / 128;
          float* dst = a.stats_partial + (row * 2) * a.N + n_w + h * 64 + lane * 8;
          *(float4*)dst = make_float4(st1[h][0], st1[h][1], st1[h][2], st1[h][3]);
          *(float4*)(dst + 4) = make_float4(st1[h][4], st1[h][5], st1[h][6], st1[h][7]);
          *(float4*)(dst + a.N) = make_float4(st2[h][0], st2[h][1], st2[h][2], st2[h][3]);
          *(float4*)(dst + a.N + 4) = make_float4(st2[h][4], st2[h][5], st2[h][6], st2[h][7]);
        }
      }
    }
    d_out_lane = (unsigned)(m0 + wm * 128 + srow) * ((unsigned)a.out_sW * OES) + (unsigned)((n_w + CH * slot) * OES);
    d_span = k.out_span;
  };

  // ---- the tile loop: tile j of workgroup b is virtual block b + j * gridDim.x (conv_gemm_persist.hip)
  int v = blockIdx.x;
  {
    int tm, tn;
    tile_order(k, xcd_remap(v, ntiles), tm, tn);
    unsigned a_lane, b_lane;
    lane_ab(a_lane, b_lane);
    a_v = a_lane + (unsigned)(tm * 256) * a_row;
    b_v = b_lane + (unsigned)(tn * 256) * b_row;
    cc = 0;
    begin_kstep();
    const Fresh f = fresh();
#pragma unroll
    for (int p = 0; p < 16; ++p) piece_at(f, 0, p, a_v, b_v, cur_wk);
  }
  for (;;) {
    int tile_m, tile_n;
    tile_order(k, xcd_remap(v, ntiles), tile_m, tile_n);
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    cc = 1;                                    // K-step 0's pieces are in flight (issued in front of the previous tile's phase 1)
    const unsigned long long t0c = k.probe ? __builtin_readcyclecounter() : 0;
    const unsigned long long t0r = k.probe ? __builtin_amdgcn_s_memrealtime() : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();              // stage 0 complete; every wave is out of phase 1 (stage 1 is free)
    {
      begin_kstep();
      const Fresh f = fresh();
#pragma unroll
      for (int p = 0; p < 4; ++p) piece_at(f, 1, p, a_v, b_v, cur_wk);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): scalar loads complete, the waitcnt inserter then counts LDS reads only
    {
      unsigned fab0, fbb0;
      frag_bases(fab0, fbb0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[0][i] = *(const uint4*)(smem + fab0 + i * 4096);
        fb[0][i] = *(const uint4*)(smem + fbb0 + i * 4096);
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.f;

    // K loop: KT >= 12 (host).  The first eleven steps drain the parked tile (33 slots for 32 row groups).
    step(yes{}, yes{}, ic<0>{}, 0); step(yes{}, yes{}, ic<1>{}, 1); step(yes{}, yes{}, ic<2>{}, 2); step(yes{}, yes{}, ic<3>{}, 3);
    step(yes{}, yes{}, ic<4>{}, 4); step(yes{}, yes{}, ic<5>{}, 5); step(yes{}, yes{}, ic<6>{}, 6); step(yes{}, yes{}, ic<7>{}, 7);
    step(yes{}, yes{}, ic<8>{}, 8); step(yes{}, yes{}, ic<9>{}, 9);
    if constexpr (K12) {
      step(yes{}, no{}, ic<10>{}, 10);
      step(no{}, no{}, ic<-1>{}, 11);
    } else {
      step(yes{}, yes{}, ic<10>{}, 10);
      int kt = 11;
      for (; kt + 2 < k.KT; ++kt) step(yes{}, yes{}, ic<-1>{}, kt);
      step(yes{}, no{}, ic<-1>{}, kt); ++kt;
      step(no{}, no{}, ic<-1>{}, kt);
    }
    if (k.probe && tid == 0 && v < 2048) {
      k.probe[2 * v] = __builtin_readcyclecounter() - t0c;
      k.probe[2 * v + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
    }
    __builtin_amdgcn_s_barrier();              // every wave is past its last fragment read: both stages are free
    const int vn = v + (int)gridDim.x;
    const bool more = vn < ntiles;
    if (more) {                                // the next tile's first stage goes out now and lands under phase 1
      int tm2, tn2;
      tile_order(k, xcd_remap(vn, ntiles), tm2, tn2);
      unsigned a_lane, b_lane;
      lane_ab(a_lane, b_lane);
      a_v = a_lane + (unsigned)(tm2 * 256) * a_row;       // (the finished tile needs them no more)
      b_v = b_lane + (unsigned)(tn2 * 256) * b_row;
      const Fresh f = fresh();
#pragma unroll
      for (int p = 0; p < 16; ++p) piece_at(f, 0, p, a_v, b_v, 0u);
    }
    if (k.dbg != 21) park(m0, n0);             // (tuning: 21 = no phase 1, 20 = phase 2 stores dropped by a zero-length descriptor)
    if (k.dbg == 20) d_span = 0;
    if (k.probe && tid == 0 && v < 2048) {
      k.probe[4096 + v] = __builtin_readcyclecounter() - t0c;
      k.probe[8192 + 2 * v] = t0r;
      k.probe[8192 + 2 * v + 1] = __builtin_amdgcn_s_memrealtime();
    }
    if (!more) break;
    v = vn;
  }
  // ---- the last tile has no K loop to hide behind
  const Fresh f = fresh();
  auto all = [&](auto self, auto uc) {
    constexpr int u = decltype(uc)::value;
    if constexpr (u < UNITS) { drain_store(f, uc); self(self, ic<u + 1>{}); }
  };
  all(all, ic<0>{});
}

int num_cus() {
  static int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    return cus;
  }();
  return n;
}

int64_t span_bytes(const gdl_conv_args& a) {
  return (((int64_t)a.B - 1) * a.out_sB + ((int64_t)a.Ho - 1) * a.out_sH + ((int64_t)a.Wo - 1) * a.out_sW + a.N) * 2;
}

// 0 = not applicable, else 1 + (SR ? 1 : 0) + (ST ? 2 : 0)
int w4p_kind(const gdl_conv_args& a) {
  const bool dense_in = a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0 && a.H == a.Ho && a.W == a.Wo &&
                        a.in_sH == (int64_t)a.W * a.in_sW && a.in_sB == (int64_t)a.H * a.in_sH;
  if (!(a.dtype == GDL_BF16 && a.out_dtype == GDL_BF16 && a.C % 256 == 0 && a.C >= 768 && a.N % 256 == 0 && a.nz == 1 && !a.aux_out &&
        !a.resid && !a.batch_scale && dense_in))
    return 0;
  if (!(a.out_sH == (int64_t)a.Wo * a.out_sW && a.out_sB == (int64_t)a.Ho * a.out_sH)) return 0;          // pixel-dense output rows
  if ((uintptr_t)a.bias % 16 != 0 || (uintptr_t)a.scale % 16 != 0 || (uintptr_t)a.shift % 16 != 0) return 0;
  if ((uintptr_t)a.out % 16 != 0 || a.out_sW % 8 != 0 || span_bytes(a) > 0x7ffffff0ll) return 0;
  if (a.stats_partial) {
    if (a.scale || a.shift || a.act != GDL_ACT_NONE || ((int64_t)a.B * a.Ho * a.Wo) % 256 != 0) return 0;
    return 1 + 2;
  }
  if (!a.scale && !a.shift && a.act == GDL_ACT_NONE) return 1;
  if (a.scale && a.act == GDL_ACT_RELU) return 1 + 1;
  return 0;
}

}  // namespace

namespace gdlconv {

bool conv_gemm_w4p_applicable(const gdl_conv_args& a) { return w4p_kind(a) != 0; }

int conv_gemm_w4p_launch(const KArgs& k, hipStream_t stream) {
  const gdl_conv_args& a = k.a;
  const int kind = w4p_kind(a);
  GDL_CHECK_ARG(kind != 0 && k.KT >= 12, "gdl_conv_gemm(persistent, parked tile): call does not qualify");
  KArgs kk = k;
  kk.tiles_m = (k.M + 255) / 256;
  kk.tiles_n = (a.N + 255) / 256;
  kk.n_group = conv_n_group(a, 256, 256, 32);
  kk.out_span = (unsigned)span_bytes(a);
  const int ntiles = kk.tiles_m * kk.tiles_n;
  const int cus = num_cus();
  dim3 grid(ntiles < cus ? ntiles : cus), block(256);
#define GDL_W4P_LAUNCH(SR, ST)                                                                      \
  do {                                                                                              \
    if (k.KT == 12) {                                                                               \
      GDL_SET_MAX_LDS_ONCE((conv_gemm_w4p_kernel<SR, ST, true>), P_LDS);                            \
      hipLaunchKernelGGL((conv_gemm_w4p_kernel<SR, ST, true>), grid, block, P_LDS, stream, kk);     \
    } else {                                                                                        \
      GDL_SET_MAX_LDS_ONCE((conv_gemm_w4p_kernel<SR, ST, false>), P_LDS);                           \
      hipLaunchKernelGGL((conv_gemm_w4p_kernel<SR, ST, false>), grid, block, P_LDS, stream, kk);    \
    }                                                                                               \
  } while (0)
  switch (kind - 1) {
    case 0: GDL_W4P_LAUNCH(false, false); break;
    case 1: GDL_W4P_LAUNCH(true, false); break;
    default: GDL_W4P_LAUNCH(false, true); break;
  }
#undef GDL_W4P_LAUNCH
  GDL_CHECK_LAUNCH("gdl_conv_gemm(256x256 persistent, parked tile)");
  return GDL_OK;
}

}  // namespace gdlconv
