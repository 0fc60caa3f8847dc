// Implicit-GEMM convolution / linear for gfx950 (CDNA4) on MFMA.
//
//   out[m, n] = epi( alpha * sum_{tap, c} in[pixel(m) + tap, c] * w[n, tap*C + c] )
//
// * NHWC activations, [N][R*S*C] weights: both operands are K-contiguous, so an operand
//   tile is ROWS x 128 bytes of K (64 bf16 or 32 f32) -- im2col never exists in memory.
// * Staging is direct global -> LDS DMA (global_load_lds_dwordx4, 1 KiB = 8 tile rows per
//   wave-instruction, no VGPR round trip, no ds_write).  The 3x3 halo / zero padding / M and N
//   tails need no predication: a lane whose 16-byte chunk is out of range simply points its
//   (per-lane) SOURCE address at a zero page.
// * The DMA writes LDS lane-linearly, so the bank swizzle lives on the source side: LDS slot s
//   of row r receives source chunk s ^ ((r>>1)&7), and fragment reads apply the same involution
//   (ds_read_b128 of chunk c at slot c ^ ((r>>1)&7)): the 4 x 16-lane groups of a fragment fetch
//   hit 16 distinct 16-byte slots -> conflict-free.
// * Two LDS stages; per K-step: wait own DMA (vmcnt 0) -> barrier -> issue next tile's DMA ->
//   MFMA phase of the current tile (DMA in flight underneath).  One barrier per K-step.
// * MFMA: v_mfma_f32_32x32x16_bf16 (bf16) or v_mfma_f32_32x32x2_f32 (exact f32 -- the parity
//   path); a wave owns TM x TN tiles of 32x32, f32 accumulators stay in registers.  K order
//   inside a 32-byte chunk pair is permuted identically for both operands (legal for a
//   reduction), so both dtypes fetch fragments with the same ds_read_b128.
// * blockIdx -> tile mapping is XCD-aware: the blocks of one XCD walk neighbouring N tiles of
//   the same M rows so the activation rows are re-read from that XCD's L2.
#include "conv_gemm_common.h"

using namespace gdlconv;

namespace {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// EXTRA = the training-only epilogue features (aux_out store of the pre-activation, GDL_ACT_MUL_GELU_GRAD);
// they live in separate instantiations so the inference / frozen-encoder kernels keep their register budget.
// PP ("ping-pong", 8-wave tile only): the two waves that share a SIMD (w and w+4) take turns being the LOADER of a
// K-step: the loader half issues the whole next tile's DMA while its partners go straight to their MFMAs, so the
// matrix pipe never idles behind a workgroup-wide DMA-issue phase; the halves swap roles every K-step.
// CT ("channel tail", 64^2 / 128^2 / 256x64 tiles only): C is not a multiple of the K chunk (MiT-B0: 32 / 160 channels in bf16, 32-wide
// attention heads); the 16-byte pieces past C in the last chunk of every tap are fetched as zeros.
// TP ("tap packed", 4-wave tiles, C < one K chunk and a divisor of it): a K chunk holds 64 / C whole filter taps
// instead of one zero-padded tap, so a 3x3 conv on 16 channels takes 3 K-steps instead of 9.  The weight rows
// [N][(tap, c)] are already contiguous in that order; on the activation side a lane's 16-byte piece belongs to tap
// (chunk's first tap + piece / pieces-per-tap), which only shifts its pixel offset and its halo bit.
// TL ("timeline", tuning only, dbg mode 7): per-wave shader-cycle totals of the K loop's phases -> k.probe[10240 + wave*8 + i]
// of block 0: i = 0 MFMA groups 0-2 (with their fragment reads), 1 wait for own DMA, 2 barrier, 3 DMA issue, 4 last group.
// S (LDS stages, 2 or 4): with four stages the DMA of tile t+4 is issued at the end of K-step t, so a tile has three K-steps to land
// instead of one.  For layers whose tiles do not fill the chip twice (the reference's per-GPU batch 4: M = 5188 gives 246 tiles of
// 128^2 for the N = 768 layers, one workgroup per CU) the K loop is a chain of DMA latencies -- a K-step's 16 MFMAs per wave take
// 0.25 us, the fetch behind them 0.8-1 us -- and nothing else on the CU hides it.
template <typename T, int WARPS_M, int WARPS_N, int TM, int TN, bool EXTRA, bool PP = false, bool CT = false,
          bool TP = false, bool TL = false, int S = 2>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N) void conv_gemm_kernel(const KArgs k) {
  constexpr int ES = TileTraits<T>::ES;
  constexpr int BKE = TileTraits<T>::BKE;
  constexpr int BM = WARPS_M * TM * 32, BN = WARPS_N * TN * 32;
  constexpr int NWALL = WARPS_M * WARPS_N;
  constexpr int NW = PP ? NWALL / 2 : NWALL;             // waves that share one tile's DMA
  constexpr int CA = BM / (8 * NW), CB = BN / (8 * NW);  // DMA instructions per loading wave per tile
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile/waves mismatch");
  static_assert(!PP || NWALL == 8, "ping-pong needs two waves per SIMD");
  static_assert(!TP || (NW == 4 && !PP && !CT), "tap packing relies on the 4-wave DMA geometry");
  static_assert(S == 2 || (S == 4 && !PP && !TL), "two or four LDS stages; ping-pong loaders and the timeline probe use two");
  static_assert(S == 2 || 3 * (CA + CB) < 64, "vmcnt is a 6-bit counter");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE_BYTES = (BM + BN) * 128;  // [A: BM rows | B: BN rows] x 128 B

  const gdl_conv_args& a = k.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;

  const int lid = xcd_remap(blockIdx.x, k.tiles_m * k.tiles_n);
  int tile_m, tile_n;
  tile_order(k, lid, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.y;
  const int z0 = z / a.nz_inner, z1 = z % a.nz_inner;
  const int64_t out_zoff = z0 * a.out_sZ0 + z1 * a.out_sZ1;
  // Operands are fetched through buffer descriptors (raw, num_records = the operand's span in bytes, < 2 GiB):
  // a lane whose 16-byte chunk is out of range (halo / zero padding / M and N tails) gets voffset = kOob and the
  // hardware writes ZEROS to its LDS slot (probed: tools/probes/buffer_lds_probe.hip) -- no predication, no zero
  // page, and all address arithmetic is 32-bit.
  constexpr unsigned kOob = 0x80000000u;
  const srd_t srd_a = make_srd((const unsigned char*)a.in + (z0 * a.in_sZ0 + z1 * a.in_sZ1) * ES, k.in_span);
  const srd_t srd_b = make_srd((const unsigned char*)a.w + (z0 * a.w_sZ0 + z1 * a.w_sZ1) * ES, k.w_span);
  const unsigned lds_base =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);

  // ---- DMA geometry: wave w, instruction i covers tile rows (i*NW + w)*8 .. +7; lane l writes
  //      LDS slot (l & 7) of row +(l >> 3) and therefore fetches source chunk slot ^ swz(row).
  const int lrow = lane >> 3, lslot = lane & 7;
  const int lwave = PP ? (wave & 3) : wave;   // index among the waves that load a tile together
  const int half = PP ? (wave >> 2) : 0;      // waves w and w+4 share a SIMD
  int a_voff[CA];        // byte offset of (b, iy0, ix0, chunk) for the row at tap (0,0), cc = 0 (may be negative)
  unsigned a_mask[CA];   // bit t set <=> filter tap t of this row reads inside the image
  unsigned a_tail = 0, b_tail = 0;   // CT: bit i set <=> instruction i's 16-byte piece lies inside the last chunk's C tail
  // TP: with four loading waves a lane's source chunk is the same for every instruction i
  const int tp_chunk = lslot ^ ((((lwave & 1) << 2) + (lrow >> 1)) & 7);
  const int tp_ppt = TP ? a.C / (16 / ES) : 1;                 // 16-byte pieces per filter tap
  const int tp_tap = tp_chunk / tp_ppt, tp_tpk = 8 / tp_ppt;   // the lane's tap inside a chunk, taps per chunk
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int r = (i * NW + lwave) * 8 + lrow;
    const int chunk = TP ? tp_chunk % tp_ppt : lslot ^ ((r >> 1) & 7);
    if (CT && chunk * (16 / ES) < k.c_tail) a_tail |= 1u << i;
    const int m = m0 + r;
    const bool ok = m < k.M;
    const int mm = ok ? m : 0;
    if (k.in_dense) {  // 1x1, stride 1, dense rows: offset is linear in m, nothing to clip
      a_voff[i] = ok ? (int)((mm * a.in_sW + chunk * (16 / ES)) * ES) : (int)kOob;
      a_mask[i] = ok ? 1u : 0u;
    } else {
      const int b = mm / HoWo, rem = mm - b * HoWo, oy = rem / a.Wo, ox = rem - oy * a.Wo;
      const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
      a_voff[i] = (int)((b * a.in_sB + iy0 * a.in_sH + ix0 * a.in_sW + chunk * (16 / ES)) * ES);
      unsigned mask = 0;
      if (ok) {
        if (a.pad == 0) {
          mask = 0xffffffffu;                       // no halo: every tap of every valid row is inside
        } else {
          for (int tr = 0; tr < a.R; ++tr)
            for (int ts = 0; ts < a.S; ++ts)
              if ((unsigned)(iy0 + tr) < (unsigned)a.H && (unsigned)(ix0 + ts) < (unsigned)a.W)
                mask |= 1u << (tr * a.S + ts);
        }
      }
      a_mask[i] = mask;
    }
  }
  unsigned b_voff[CB];   // weight row + chunk (k = 0), or kOob for the N tail
#pragma unroll
  for (int i = 0; i < CB; ++i) {
    const int r = (i * NW + lwave) * 8 + lrow;
    const int chunk = lslot ^ ((r >> 1) & 7);
    const int n = n0 + r;
    b_voff[i] = n < a.N ? (unsigned)((n * a.w_sN + chunk * (16 / ES)) * ES) : kOob;
    if (CT && chunk * (16 / ES) < k.c_tail) b_tail |= 1u << i;
  }

  int tap_r = 0, tap_s = 0, cc = 0;  // position of the NEXT tile to fetch
  unsigned wk = 0;                   // its byte offset along the weight rows

  auto issue = [&](int stage, bool load) {
    if (load) {
      const unsigned lds_a = lds_base + stage * STAGE_BYTES + lwave * 1024;
      const unsigned lds_b = lds_a + BM * 128;
      // CT: in the last channel chunk only the pieces below C are real
      const unsigned ta = (CT && cc == k.kc - 1) ? a_tail : 0xffffffffu;
      const unsigned tb = (CT && cc == k.kc - 1) ? b_tail : 0xffffffffu;
      if constexpr (TP) {
        const int t = cc * tp_tpk + tp_tap;               // this lane's filter tap in K chunk cc
        const int tr = t / a.S, ts = t - tr * a.S;
        const int toff = (int)((tr * a.in_sH + ts * a.in_sW) * ES);
        const unsigned bit = (t < a.R * a.S) ? (a.pad == 0 ? 1u : 1u << (t & 31)) : 0u;
#pragma unroll
        for (int i = 0; i < CA; ++i) {
          const unsigned v = (a_mask[i] & bit) ? (unsigned)(a_voff[i] + toff) : kOob;
          dma16_buf(v, srd_a, 0u, lds_a + i * NW * 1024);
        }
        const bool bv = cc * 128 + tp_chunk * 16 < a.R * a.S * a.C * ES;   // past the end of the weight row
#pragma unroll
        for (int i = 0; i < CB; ++i) dma16_buf(bv ? b_voff[i] : kOob, srd_b, wk, lds_b + i * NW * 1024);
      } else
      if (k.in_dense) {
#pragma unroll
        for (int i = 0; i < CA; ++i) {
          const unsigned v = (!CT || ((ta >> i) & 1u)) ? (unsigned)a_voff[i] : kOob;
          dma16_buf(v, srd_a, wk, lds_a + i * NW * 1024);
        }
      } else {
        const int tap_off = (int)((tap_r * a.in_sH + tap_s * a.in_sW + cc * BKE) * ES);
        const unsigned bit = a.pad == 0 ? 1u : 1u << (tap_r * a.S + tap_s);
#pragma unroll
        for (int i = 0; i < CA; ++i) {
          const unsigned v = ((a_mask[i] & bit) && (!CT || ((ta >> i) & 1u))) ? (unsigned)(a_voff[i] + tap_off) : kOob;
          dma16_buf(v, srd_a, 0u, lds_a + i * NW * 1024);
        }
      }
      if constexpr (!TP) {
#pragma unroll
        for (int i = 0; i < CB; ++i) {
          const unsigned v = (!CT || ((tb >> i) & 1u)) ? b_voff[i] : kOob;
          dma16_buf(v, srd_b, wk, lds_b + i * NW * 1024);
        }
      }
    }
    // every wave tracks the tile position, loader or not.  K order = channel chunk OUTER, filter tap INNER: the
    // R*S taps of one chunk re-read (shifted) the same activation bytes back to back, while they are hot in L2
    if constexpr (TP) {
      ++cc;
      wk = (unsigned)(cc * 128);
    } else {
      if (k.tap_inner) {
        if (++tap_s == a.S) { tap_s = 0; if (++tap_r == a.R) { tap_r = 0; ++cc; } }
      } else {
        if (++cc == k.kc) { cc = 0; if (++tap_s == a.S) { tap_s = 0; ++tap_r; } }
      }
      wk = (unsigned)(((tap_r * a.S + tap_s) * a.C + cc * BKE) * ES);
    }
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5;
  const int fswz = (frow >> 1) & 7;
  const int a_lds0 = (wm * TM * 32 + frow) * 128;
  const int b_lds0 = BM * 128 + (wn * TN * 32 + frow) * 128;

  // ---- main loop.  Per K-step (tile kt in LDS stage kt&1, four k16 groups kk):
  //   kk = 0..2 : fetch fragments of kk+1 (ds_read_b128, double-buffered registers), MFMAs of kk
  //   then      : wait own DMA of tile kt+1 (vmcnt 0) -> barrier -> issue DMA of tile kt+2 into the stage tile kt
  //               just vacated (every wave holds its kk=3 fragments in registers) -> fetch kk=0 of tile kt+1
  //   kk = 3    : MFMAs, covering the barrier skew, the DMA issue and the first-fragment LDS latency.
  // One barrier per K-step, a tile's DMA has a full K-step to land, and no MFMA ever waits on a just-issued read.
  uint4 fa[2][TM], fb[2][TN];
  auto fetch = [&](const unsigned char* st, int kk, int buf) {
    const int coff = (((2 * kk + fhalf) ^ fswz) << 4);
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[buf][i] = *(const uint4*)(st + a_lds0 + i * 32 * 128 + coff);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[buf][j] = *(const uint4*)(st + b_lds0 + j * 32 * 128 + coff);
  };
  auto mfmas = [&](int buf) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (ES == 2) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(bf16x8_t, fb[buf][j]), __builtin_bit_cast(bf16x8_t, fa[buf][i]), acc[i][j], 0, 0, 0);
        } else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fb[buf][j].x), __uint_as_float(fa[buf][i].x), acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fb[buf][j].y), __uint_as_float(fa[buf][i].y), acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fb[buf][j].z), __uint_as_float(fa[buf][i].z), acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fb[buf][j].w), __uint_as_float(fa[buf][i].w), acc[i][j], 0, 0, 0);
        }
      }
  };
  const unsigned long long t0c = k.probe ? __builtin_readcyclecounter() : 0;
  const unsigned long long t0r = k.probe ? __builtin_amdgcn_s_memrealtime() : 0;
  // tile t is loaded by half (t & 1) in the ping-pong variant, by everyone otherwise
  // wait_tile(ahead): this wave's share of the oldest outstanding tile has landed when at most `ahead` younger tiles (CA + CB DMA
  // instructions each; loads return in order) are still in flight
  auto wait_tile = [&](int ahead) {
    if (S == 4 && ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (CA + CB)) : "memory");
    else if (S == 4 && ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (CA + CB)) : "memory");
    else if (S == 4 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CA + CB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  if constexpr (S == 2) {
    issue(0, !PP || half == 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (k.KT > 1) issue(1, !PP || half == 1);
  } else {
    // tiles 0 .. S-1 before the loop; K-step kt then issues tile kt + S into the stage tile kt just vacated
#pragma unroll
    for (int t = 0; t < S; ++t)
      if (t < k.KT) issue(t, true);
    wait_tile(k.KT - 1 < S - 1 ? k.KT - 1 : S - 1);
    __syncthreads();
  }
  // all scalar (kernel-argument) loads are complete here: tell the waitcnt inserter, so that inside the loop it can
  // wait for the OLDER fragment reads only (lgkmcnt(6)) instead of draining every LDS read before the first MFMAs
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
  fetch(smem, 0, 0);
  unsigned long long tl[5] = {0, 0, 0, 0, 0}, tl_t = 0;
  auto stamp = [&](int i) {
    if constexpr (TL) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long now = __builtin_readcyclecounter();
      tl[i] += now - tl_t;
      tl_t = now;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if constexpr (TL) { __builtin_amdgcn_sched_barrier(0); tl_t = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
  for (int kt = 0; kt < k.KT; ++kt) {
    const unsigned char* st = smem + (kt & (S - 1)) * STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
      fetch(st, kk + 1, (kk + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(kk & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    stamp(0);
    if (kt + 1 < k.KT) {
      if constexpr (S == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of tile kt+1 has landed
        stamp(1);
        __syncthreads();                                   // ... and everyone's; nobody reads tile kt's stage any more
        stamp(2);
        if (kt + 2 < k.KT) issue(kt & 1, (!PP || half == (kt & 1)) && k.dbg != 1);
        stamp(3);
      } else {
        // issued so far: tiles <= kt + S - 1; tile kt + 1 is needed next
        const int last = kt + S - 1 < k.KT - 1 ? kt + S - 1 : k.KT - 1;
        wait_tile(last - (kt + 1));
        __syncthreads();   // everyone's share of tile kt+1 is there; every wave holds its last fragments of tile kt in registers
        if (kt + S < k.KT) issue(kt & (S - 1), k.dbg != 1);
      }
      fetch(smem + ((kt + 1) & (S - 1)) * STAGE_BYTES, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfmas(1);
    __builtin_amdgcn_sched_barrier(0);
    stamp(4);
  }
  if constexpr (TL) {
    if (k.probe && lane == 0 && blockIdx.x == 0 && blockIdx.y == 0) {
#pragma unroll
      for (int i = 0; i < 5; ++i) k.probe[10240 + wave * 8 + i] = tl[i];
    }
  }

  if (k.probe && tid == 0 && blockIdx.x < 2048) {
    k.probe[2 * blockIdx.x] = __builtin_readcyclecounter() - t0c;
    k.probe[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
  }
  __syncthreads();   // every wave is past its last fragment read: the stages become the epilogue's transpose buffers
  conv_epilogue<TM, TN, EXTRA>(k, acc, m0, n0, wm, wn, lane, out_zoff, smem + wave * 8192, smem + NWALL * 8192 + wave * 1024);
  if (k.probe && tid == 0 && blockIdx.x < 2048) {
    k.probe[4096 + blockIdx.x] = __builtin_readcyclecounter() - t0c;
    k.probe[8192 + 2 * blockIdx.x] = t0r;                                   // block timeline (100 MHz ticks)
    k.probe[8192 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
}

template <typename T, int WARPS_M, int WARPS_N, int TM, int TN, bool EXTRA, bool PP = false, bool CT = false,
          bool TP = false, bool TL = false, int S = 2>
int launch_x(const KArgs& k, hipStream_t stream) {
  constexpr int BM = WARPS_M * TM * 32, BN = WARPS_N * TN * 32;
  KArgs kk = k;
  kk.tiles_m = (k.M + BM - 1) / BM;
  kk.tiles_n = (k.a.N + BN - 1) / BN;
  kk.n_group = conv_n_group(k.a, BM, BN, 32 * (BM * BN >= 65536 ? 1 : BM * BN >= 16384 ? 2 : 4));
  // (the coalesced epilogue of 64-channel wave tiles needs 8 KiB + 1 KiB of LDS per wave: more than the two stages of a 128^2 tile
  // shared by eight waves)
  const size_t lds_k = (size_t)S * (BM + BN) * 128, lds_e = TN == 2 ? (size_t)WARPS_M * WARPS_N * 9216 : 0;
  static_assert(TN != 2 || S * (BM + BN) * 128 >= WARPS_M * WARPS_N * 9216, "the stages must cover the epilogue's transpose buffers");
  const size_t lds = lds_k > lds_e ? lds_k : lds_e;
  auto kern = conv_gemm_kernel<T, WARPS_M, WARPS_N, TM, TN, EXTRA, PP, CT, TP, TL, S>;
  GDL_SET_MAX_LDS_ONCE(kern, lds);   // one flag per template instantiation of launch_x
  dim3 grid(kk.tiles_m * kk.tiles_n, k.a.nz), block(64 * WARPS_M * WARPS_N);
  hipLaunchKernelGGL(kern, grid, block, lds, stream, kk);
  GDL_CHECK_LAUNCH("gdl_conv_gemm");
  return GDL_OK;
}

template <typename T, int WARPS_M, int WARPS_N, int TM, int TN>
int launch(const KArgs& k, hipStream_t stream) {
  const bool extra = k.a.aux_out || k.a.act == GDL_ACT_MUL_GELU_GRAD;
  if (k.c_tail) {
    if (extra) return launch_x<T, WARPS_M, WARPS_N, TM, TN, true, false, true>(k, stream);
    return launch_x<T, WARPS_M, WARPS_N, TM, TN, false, false, true>(k, stream);
  }
  if (extra) return launch_x<T, WARPS_M, WARPS_N, TM, TN, true>(k, stream);
  return launch_x<T, WARPS_M, WARPS_N, TM, TN, false>(k, stream);
}

}  // namespace


// tuning hooks (gdl_debug_*): process-global words, written only by the tools/ scripts, read with relaxed atomics
#include <atomic>
static std::atomic<int> g_tap_inner{1}, g_dbg{0}, g_tap_packing{1}, g_epi_v2{1};
extern "C" void gdl_debug_set_conv_epilogue(int v2) { g_epi_v2 = v2; }  // A/B hook: 0 = the round-2 epilogue
extern "C" void gdl_debug_set_conv_tap_packing(int on) { g_tap_packing = on; }  // A/B hook
extern "C" void gdl_debug_set_conv_dbg(int mode) { g_dbg = mode; }
static std::atomic<unsigned long long*> g_probe{nullptr};
extern "C" void gdl_debug_set_conv_probe(void* dev_buf_2048x2_u64) { g_probe = (unsigned long long*)dev_buf_2048x2_u64; }
extern "C" void gdl_debug_set_conv_korder(int tap_inner) { g_tap_inner = tap_inner; }  // A/B hook

extern "C" int gdl_conv_gemm(const gdl_conv_args* ap, gdl_stream_t stream) {
  GDL_CHECK_ARG(ap != nullptr, "gdl_conv_gemm: null args");
  const gdl_conv_args& a = *ap;
  GDL_CHECK_ARG(a.dtype == GDL_F32 || a.dtype == GDL_BF16, "gdl_conv_gemm: bad dtype %d", a.dtype);
  GDL_CHECK_ARG(a.out_dtype == GDL_F32 || a.out_dtype == GDL_BF16, "gdl_conv_gemm: bad out_dtype");
  const int es = (int)gdl_elem_size(a.dtype);
  const int bke = 128 / es, al = 16 / es;
  GDL_CHECK_ARG(a.in && a.w && a.out, "gdl_conv_gemm: null tensor");
  GDL_CHECK_ARG(a.B > 0 && a.H > 0 && a.W > 0 && a.C > 0 && a.N > 0 && a.Ho > 0 && a.Wo > 0,
                "gdl_conv_gemm: non-positive dims");
  GDL_CHECK_ARG(a.R > 0 && a.S > 0 && a.stride > 0 && a.pad >= 0, "gdl_conv_gemm: bad filter");
  GDL_CHECK_ARG(a.C % al == 0, "gdl_conv_gemm: C=%d must be a multiple of %d", a.C, al);
  GDL_CHECK_ARG(a.in_sB % al == 0 && a.in_sH % al == 0 && a.in_sW % al == 0 && a.w_sN % al == 0 &&
                    a.in_sZ0 % al == 0 && a.in_sZ1 % al == 0 && a.w_sZ0 % al == 0 && a.w_sZ1 % al == 0,
                "gdl_conv_gemm: strides must keep 16-byte alignment");
  GDL_CHECK_ARG(((uintptr_t)a.in % 16 == 0) && ((uintptr_t)a.w % 16 == 0),
                "gdl_conv_gemm: operand pointers must be 16-byte aligned");
  GDL_CHECK_ARG(a.nz >= 1 && a.nz_inner >= 1, "gdl_conv_gemm: nz/nz_inner must be >= 1");
  GDL_CHECK_ARG(a.act >= GDL_ACT_NONE && a.act <= GDL_ACT_RESID_RELU, "gdl_conv_gemm: bad act %d", a.act);
  GDL_CHECK_ARG(a.act != GDL_ACT_RESID_RELU || a.resid, "gdl_conv_gemm: GDL_ACT_RESID_RELU needs `resid`");
  GDL_CHECK_ARG(a.act != GDL_ACT_MUL_GELU_GRAD || (a.resid && !a.aux_out),
                "gdl_conv_gemm: GDL_ACT_MUL_GELU_GRAD takes the pre-activation tensor in `resid`");
  GDL_CHECK_ARG((int64_t)a.B * a.Ho * a.Wo < (1ll << 31), "gdl_conv_gemm: M too large");
  // `stats_partial` selects an epilogue that ignores resid / scale / shift / batch_scale / act and writes bf16 whole wave tiles
  // only: a call that does not qualify (or whose tile variant was switched by a debug hook between the caller's query and this
  // launch) must fail here, not return wrong outputs and uninitialised statistics rows
  GDL_CHECK_ARG(!a.stats_partial || gdl_conv_gemm_stats_rows(ap) > 0,
                "gdl_conv_gemm: stats_partial set on a call that cannot emit BatchNorm statistics (gdl_conv_gemm_stats_rows() == 0: "
                "needs bf16 in/out, bias-only epilogue, pixel-dense 16-byte aligned output, whole 128/256-row tiles)");
  GDL_CHECK_ARG(a.pad == 0 || a.R * a.S <= 32, "gdl_conv_gemm: padded filters are limited to 32 taps (got %dx%d)", a.R, a.S);
  // operands are addressed through 32-bit buffer offsets: each must span < 2 GiB (split the batch otherwise)
  const int64_t kSpanMax = 0x7ffffff0ll;
  const int64_t in_span = (((int64_t)a.B - 1) * a.in_sB + ((int64_t)a.H - 1) * a.in_sH + ((int64_t)a.W - 1) * a.in_sW + a.C) * es;
  const int64_t w_span = (((int64_t)a.N - 1) * a.w_sN + (int64_t)a.R * a.S * a.C) * es;
  GDL_CHECK_ARG(w_span <= kSpanMax, "gdl_conv_gemm: weight matrix spans more than 2 GiB");
  if (in_span > kSpanMax) {
    GDL_CHECK_ARG(a.B > 1 && a.nz == 1, "gdl_conv_gemm: one image spans more than 2 GiB");
    const int b1 = a.B / 2;
    const int64_t oes = (int64_t)gdl_elem_size(a.out_dtype);
    gdl_conv_args lo = a, hi = a;
    lo.B = b1;
    hi.B = a.B - b1;
    hi.in = (const unsigned char*)a.in + (int64_t)b1 * a.in_sB * es;
    hi.out = (unsigned char*)a.out + (int64_t)b1 * a.out_sB * oes;
    if (a.aux_out) hi.aux_out = (unsigned char*)a.aux_out + (int64_t)b1 * a.out_sB * oes;
    if (a.resid) hi.resid = (const unsigned char*)a.resid + (int64_t)b1 * a.res_sB * (int64_t)gdl_elem_size(a.resid_dtype);
    if (a.batch_scale) hi.batch_scale = a.batch_scale + b1;
    const int st = gdl_conv_gemm(&lo, stream);
    return st != GDL_OK ? st : gdl_conv_gemm(&hi, stream);
  }
  KArgs k;
  k.a = a;
  k.M = a.B * a.Ho * a.Wo;
  k.kc = (a.C + bke - 1) / bke;
  k.c_tail = a.C % bke;
  k.KT = a.R * a.S * k.kc;
  // "dense" = the (b,oy,ox) -> offset map is linear in m, so no divisions are needed
  k.tap_inner = g_tap_inner;
  k.n_group = 0;
  k.tiles_m = k.tiles_n = 0;
  k.dbg = g_dbg;
  k.epi_v2 = g_epi_v2;
  k.probe = g_probe;
  k.in_span = (unsigned)in_span;
  k.w_span = (unsigned)w_span;
  k.in_dense = (a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0 && a.H == a.Ho && a.W == a.Wo &&
                a.in_sH == (int64_t)a.W * a.in_sW && a.in_sB == (int64_t)a.H * a.in_sH);
  k.out_dense = (a.out_sH == (int64_t)a.Wo * a.out_sW && a.out_sB == (int64_t)a.Ho * a.out_sH);
  k.res_dense = !a.resid || (a.res_sH == (int64_t)a.Wo * a.res_sW && a.res_sB == (int64_t)a.Ho * a.res_sH);
  hipStream_t s = (hipStream_t)stream;
  const int variant = gdl_conv_gemm_plan(ap, nullptr);
  // tap packing: several whole filter taps per K chunk when C divides the chunk (UNet++'s 16 / 32-channel stages)
  const bool tap_packed = g_tap_packing && variant == 5 && a.R * a.S > 1 && a.R * a.S <= 32 && !k.in_dense &&
                          a.C < bke && bke % a.C == 0 && a.C >= al;
  if (tap_packed) {
    k.kc = (a.R * a.S * a.C + bke - 1) / bke;
    k.KT = k.kc;
    k.c_tail = 0;
  }
  if (variant == 7) return conv3x3_narrow_launch(k, s);
  if (variant == 6) return conv_gemm_dual_launch(k, s);
  if (variant == 8) return conv_gemm_w4_launch(k, s);
  if (variant == 9) return conv_gemm_persist_launch(k, s);
  if (variant == 10) return conv_gemm_w4p_launch(k, s);
  if (variant == 12) {   // 64^2 tile with four LDS stages (bf16, whole K chunks): few tiles, long K
    const bool extra = a.aux_out || a.act == GDL_ACT_MUL_GELU_GRAD;
    return extra ? launch_x<bf16_tag, 2, 2, 1, 1, true, false, false, false, false, 4>(k, s)
                 : launch_x<bf16_tag, 2, 2, 1, 1, false, false, false, false, false, 4>(k, s);
  }
  if (a.dtype == GDL_BF16) {
    if (variant == 4) return conv3x3_sf_launch(k, s);
    if (variant == 3 && k.dbg == 7) return launch_x<bf16_tag, 2, 4, 4, 2, false, true, false, false, true>(k, s);
    if (variant == 3) return launch_x<bf16_tag, 2, 4, 4, 2, false, true>(k, s);
    if (variant == 2) return launch_x<bf16_tag, 2, 4, 4, 2, false>(k, s);
    if (variant == 5 && tap_packed) return launch_x<bf16_tag, 4, 1, 2, 2, false, false, false, true>(k, s);
    if (variant == 5) return k.c_tail ? launch_x<bf16_tag, 4, 1, 2, 2, false, false, true>(k, s) : launch_x<bf16_tag, 4, 1, 2, 2, false>(k, s);
    if (variant == 1) return launch<bf16_tag, 2, 2, 2, 2>(k, s);
    return launch<bf16_tag, 2, 2, 1, 1>(k, s);
  }
  if (variant == 4) return conv3x3_sf_launch(k, s);
  if (variant == 3) return launch_x<float, 2, 4, 4, 2, false, true>(k, s);
  if (variant == 2) return launch_x<float, 2, 4, 4, 2, false>(k, s);
  if (variant == 5 && tap_packed) return launch_x<float, 4, 1, 2, 2, false, false, false, true>(k, s);
  if (variant == 5) return k.c_tail ? launch_x<float, 4, 1, 2, 2, false, false, true>(k, s) : launch_x<float, 4, 1, 2, 2, false>(k, s);
  if (variant == 1) return launch<float, 2, 2, 2, 2>(k, s);
  return launch<float, 2, 2, 1, 1>(k, s);
}

// Tile selection by available parallelism (256 CUs) and output width: 256x256 tiles / 8 waves (variants 2-4) when
// they still give >= 512 blocks, 128x128 / 4 waves (variant 1) when that gives >= 256 blocks,
// else 64x64 (variant 0).  Also reports the ALGORITHMIC flops of the call (2*M*N*K, no padding).
static std::atomic<int> g_forced_variant{-1};
static std::atomic<int> g_w4_enabled{1};
extern "C" void gdl_debug_set_conv_w4(int on) { g_w4_enabled = on; }  // A/B hook: 256^2 one-wave-per-SIMD tile
static std::atomic<int> g_persist_enabled{1};
extern "C" void gdl_debug_set_conv_persist(int on) { g_persist_enabled = on; }  // A/B hook: persistent 256^2 tile for dense 1x1 layers
static std::atomic<int> g_w4p_enabled{0};   // see the planner: measured faster per layer, not end to end -- opt-in (GDL_CONV_W4P=1)
static std::atomic<int> g_w4p_min_n{768};
extern "C" void gdl_debug_set_conv_w4p(int on) { g_w4p_enabled = on != 0; if (on > 1) g_w4p_min_n = on; }  // A/B hook: persistent 256^2 tile with deferred stores (conv_gemm_w4p.hip)
static std::atomic<int> g_stage4_enabled{1};
extern "C" void gdl_debug_set_conv_stage4(int on) { g_stage4_enabled = on; }  // A/B hook: four-stage 64^2 tile for few-tile, long-K layers
static std::atomic<int> g_dual_enabled{1};
extern "C" void gdl_debug_set_conv_dual(int on) { g_dual_enabled = on; }  // A/B hook: dual-resident 256 x 128 tile
static std::atomic<int> g_ngroup_kb{2560};
extern "C" void gdl_debug_set_conv_ngroup_kb(int kb) { g_ngroup_kb = kb; }  // A/B hook: weight bytes (KiB) one N group may hold; 0 = no grouping
namespace gdlconv {
// The 32 CUs of an XCD run `conc` tiles at a time, arranged as (conc / w) M rows x w N tiles when N is walked in groups of w.
// They run in step (same k at the same time), so per round the XCD's L2 pulls (conc / w) activation panels plus -- unless the
// group's w weight panels stay resident (budget: g_ngroup_kb of the 4 MiB L2) -- w weight panels.  Compare "no grouping" with
// "the widest group that stays resident" and take the cheaper one.
int conv_n_group(const gdl_conv_args& a, int bm, int bn, int conc) {
  const int64_t budget = (int64_t)g_ngroup_kb * 1024;
  if (budget <= 0) return 0;
  const int64_t kbytes = (int64_t)a.R * a.S * a.C * (int64_t)gdl_elem_size(a.dtype);
  const int64_t a_panel = bm * kbytes, w_panel = bn * kbytes;
  const int64_t tiles_n = (a.N + bn - 1) / bn;
  const int64_t w_fit = budget / w_panel;
  if (w_fit < 1 || w_fit >= tiles_n) return 0;
  auto cost = [&](int64_t w) {
    const int64_t wc = w < conc ? w : conc;
    return (conc / wc) * a_panel + (w * w_panel <= budget ? 0 : wc * w_panel);
  };
  if (cost(w_fit) >= cost(tiles_n)) return 0;
  const int64_t groups = (tiles_n + w_fit - 1) / w_fit;          // equal-width groups
  return (int)((tiles_n + groups - 1) / groups);
}
}  // namespace gdlconv
static std::atomic<int> g_sf_enabled{1};
static std::atomic<int> g_narrow_enabled{1};
extern "C" void gdl_debug_set_conv_narrow(int on) { g_narrow_enabled = on; }  // A/B hook: direct narrow 3x3 kernel
extern "C" void gdl_debug_set_conv_sf(int on) { g_sf_enabled = on; }  // A/B hook: 3x3 shared-staging kernel
extern "C" void gdl_debug_force_conv_variant(int v) { g_forced_variant = v; }  // tuning hook (-1 = auto)

extern "C" int gdl_conv_gemm_plan(const gdl_conv_args* ap, int64_t* flops) {
  if (!ap) return -1;
  const gdl_conv_args& a = *ap;
  const int64_t M = (int64_t)a.B * a.Ho * a.Wo;
  if (flops) *flops = 2 * M * a.N * ((int64_t)a.R * a.S * a.C) * a.nz;
  if (g_forced_variant >= 0 && !(g_forced_variant >= 2 && (a.aux_out || a.act == GDL_ACT_MUL_GELU_GRAD)) &&
      !(g_forced_variant >= 2 && g_forced_variant != 5 && a.C % (a.dtype == GDL_BF16 ? 64 : 32) != 0) &&
      !(g_forced_variant == 5 && a.N > 64) &&
      !(g_forced_variant == 4 && !conv3x3_sf_applicable(a)) &&
      !(g_forced_variant == 6 && !conv_gemm_dual_applicable(a)) &&
      !(g_forced_variant == 8 && !conv_gemm_w4_applicable(a)) &&
      !(g_forced_variant == 9 && !conv_gemm_persist_applicable(a)) &&
      !(g_forced_variant == 10 && !conv_gemm_w4p_applicable(a)) &&
      !(g_forced_variant >= 11 && (g_forced_variant != 12 || a.dtype != GDL_BF16 || a.C % 64 != 0)) &&
      !(g_forced_variant == 7 && !conv3x3_narrow_applicable(a)))
    return g_forced_variant;
  // narrow 3x3 layers on large maps: direct kernel, one staged window per 4 x 64 pixels (HBM-bound layers)
  if (g_narrow_enabled && conv3x3_narrow_applicable(a)) return 7;
  const int64_t t256 = ((M + 255) / 256) * ((a.N + 255) / 256) * a.nz;
  const int64_t t128 = ((M + 127) / 128) * ((a.N + 127) / 128) * a.nz;
  const int64_t ksteps = (int64_t)a.R * a.S * a.C * (int64_t)gdl_elem_size(a.dtype) / 128;
  // measured (tools/bench_conv.py): 256^2 tiles win whenever K is deep, even at ~1 block per CU;
  // for shallow K (ViT linears, 12 K-steps) the 128^2 tile's shorter prologue/epilogue wins
  // the training-only epilogue (aux_out / GELU-grad) does not fit the 256^2 tile's register budget
  const bool extra = a.aux_out != nullptr || a.act == GDL_ACT_MUL_GELU_GRAD;
  const bool ctail = a.C % (a.dtype == GDL_BF16 ? 64 : 32) != 0;   // only the small tiles zero-fill a channel tail
  // (tools/bench_conv_variants.py, batch 32: ViT proj / neck 1x1 768 -> 768 with 486-489 tiles: 256^2 92 / 59 us vs 128^2 101 / 67 us)
  if (!extra && !ctail && a.N % 256 == 0 && (t256 >= 480 || (t256 >= 256 && ksteps >= 32))) {
    // short-K layers with a GELU epilogue (ViT fc1: 12 K-steps, then ~13 k cycles of VALU work per 256^2 tile): two resident
    // 256 x 128 workgroups per CU put one's epilogue under the other's K loop (+5..9 %, profiles/r04a_bench_short_k_*); for
    // every other epilogue the 256^2 tile is as fast or faster
    if (g_dual_enabled && a.act == GDL_ACT_GELU && ksteps <= 16 && t256 >= 1024 && conv_gemm_dual_applicable(a)) return 6;
    // dense 1x1 layers with bf16 outputs, N >= 768, 12 .. 39 K-steps and at least two rounds of tiles: one persistent workgroup per
    // CU, one wave per SIMD, the finished tile parked in registers and stored from the MFMA shadows of the next tile's K loop
    // (conv_gemm_w4p.hip).  OPT-IN (gdl_debug_set_conv_w4p / GDL_CONV_W4P=1): per layer it is faster -- ViT qkv 1022 -> 1092 TF/s, the
    // neck's tap products 1061 -> 1142 (tools/bench_w4p.py) -- but three same-box A/B runs of the whole step say +-0: 879.3-880.0 vs
    // 879.5-880.3 train tiles/s at batch 32 (+0.3 % inference), 951.6 vs 949.6 at batch 64 (profiles/r06g_*, r06n_*).  The GEMM
    // phases run at the 1400 W power cap (tools/probe_power_clock.sh): cycles saved there come back as clock, not as time.
    if (g_w4p_enabled && t256 >= 512 && a.N >= g_w4p_min_n && ksteps >= 12 && ksteps < 40 && conv_gemm_w4p_applicable(a)) return 10;
    // deep-K layers: one wave per SIMD, every load in an MFMA shadow (conv_gemm_w4.hip): its K-step is ~12 % shorter, its
    // epilogue (four waves instead of eight) ~1.8 x longer -- it wins from about 40 K-steps on (ViT fc2 +8 %, tap data
    // gradients +9 %, the 768-channel 3x3 convolutions +2.5 %; tools/bench_w4.py, profiles/r04d_*)
    if (g_w4_enabled && conv_gemm_w4_applicable(a) && ksteps >= (a.R * a.S > 1 ? 100 : 40)) return 8;
    // dense 1x1 layers with more tiles than CUs: one persistent workgroup per CU, the next tile's first stage lands under the
    // epilogue (conv_gemm_persist.hip)
    if (g_persist_enabled && t256 > 256 && conv_gemm_persist_applicable(a)) return 9;
    return (g_sf_enabled && conv3x3_sf_applicable(a)) ? 4 : 3;   // ping-pong 256^2 (4: 3x3 with shared staging)
  }
  // (224: the reference's own per-GPU batch 4 gives M = 5188 / 5184 -> 246 tiles for the N = 768 layers; one round of 128^2 tiles on 246 of
  // 256 CUs beats four rounds of 64^2 tiles: ViT proj 20.0 -> 16.3 us, fc2 48.0 -> 43.0, the neck's 3x3 at 36^2 86.1 -> 80.0; tools/bench_small_m.py)
  if (t128 >= 224 && a.N >= 128) return 1;
  // narrow outputs (N <= 64: UNet++ decoder, ResNet layer1, MiT stage 1): a 256 (m) x 64 (n) tile, four waves of
  // 64 x 64 -- one LDS fragment read per MFMA instead of the two of the 64^2 tile's 32 x 32 waves
  if (!extra && a.N <= 64 && ((M + 255) / 256) * a.nz >= 256) return 5;
  // at most one round of 64^2 tiles and a long K (batch 4: UperNet's pyramid-pooling bottleneck, 3x3 on 1792 channels at 18^2: 84
  // tiles, 252 K-steps; the neck's 3x3 at 18^2: 252 tiles, 108 K-steps): the K loop of the few resident workgroups is a chain of
  // DMA latencies -- four LDS stages (tile t+4 requested at K-step t): 125 -> 103 us and 57 -> 48 us; bit-identical results
  // (tools/bench_small_m.py, profiles/r06q_*).  With more tiles or a shorter K the two-stage tile is faster (more workgroups per CU).
  const int64_t t64 = ((M + 63) / 64) * ((a.N + 63) / 64) * a.nz;
  if (g_stage4_enabled && a.dtype == GDL_BF16 && !ctail && t64 <= 256 && ksteps >= 96) return 12;
  return 0;
}

// Can this call emit BatchNorm partial statistics from its epilogue, and how many partial rows?  Mirrors the conditions under
// which EVERY wave of the launch takes the coalesced epilogue (conv_epilogue_rows2) with whole wave tiles.
extern "C" int64_t gdl_conv_gemm_stats_rows(const gdl_conv_args* ap) {
  if (!ap || !g_epi_v2 || g_forced_variant >= 0) return 0;
  const gdl_conv_args& a = *ap;
  if (a.dtype != GDL_BF16 || a.out_dtype != GDL_BF16 || a.resid || a.scale || a.shift || a.batch_scale || a.aux_out ||
      a.act != GDL_ACT_NONE || a.nz != 1 || a.N % 16 != 0)
    return 0;
  if (!(a.out_sH == (int64_t)a.Wo * a.out_sW && a.out_sB == (int64_t)a.Ho * a.out_sH)) return 0;             // pixel-dense output
  if ((uintptr_t)a.out % 16 != 0 || a.out_sW % 8 != 0 || (uintptr_t)a.bias % 16 != 0) return 0;
  const int64_t M = (int64_t)a.B * a.Ho * a.Wo;
  const int v = gdl_conv_gemm_plan(ap, nullptr);
  int bm, bn, tm;
  if (v == 3 || v == 4 || v == 8 || v == 9 || v == 10) { bm = 256; bn = 256; tm = 4; }
  else if (v == 1) { bm = 128; bn = 128; tm = 2; }
  else return 0;
  if (M % bm != 0 || a.N % bn != 0) return 0;
  // (the 2 GiB operand split of gdl_conv_gemm would number the rows of the second half from zero)
  const int64_t es = 2;
  if ((((int64_t)a.B - 1) * a.in_sB + ((int64_t)a.H - 1) * a.in_sH + ((int64_t)a.W - 1) * a.in_sW + a.C) * es > 0x7ffffff0ll) return 0;
  return M / (32 * tm);
}
