// "Dual-resident" implicit-GEMM tile for short-K layers (ViT linears, 1x1 laterals, their data gradients): bf16,
// 256 (m) x 128 (n) outputs per workgroup, FOUR waves (one per SIMD, wave tile 128 x 64 = 128 accumulator registers),
// 80 KiB of LDS -- so that TWO workgroups are resident on a CU and run unsynchronised: while one of them is in its prologue
// (first DMA latency) or in its epilogue (10-30 k cycles of stores during which the single resident 256^2 / 8-wave block of
// conv_gemm.hip issues no MFMA at all), the other one owns the matrix pipe.  With K = 768 the 256^2 kernel spends 12 K-steps
// (~35 k cycles) between a 3 k-cycle prologue and a 10-30 k-cycle epilogue; here those phases of one workgroup sit under the
// K loop of its neighbour.
//
// LDS budget: the activation tile (256 rows x 128 B = 32 KiB per K-step) is double-buffered, the weight tile (128 rows x
// 128 B = 16 KiB) is NOT: a wave reads ALL its weight fragments of a K-step (4 k16 groups x 2 tiles = 32 registers) right
// after the step's first barrier, a second barrier says "everyone holds its weights", and only then is the next step's DMA
// issued (weights into the single buffer, activations into the other stage).  2 x 32 + 16 = 80 KiB, two workgroups = the
// CU's 160 KiB.  The bubble between the two barriers (eight ds_read_b128 with no MFMA in flight from this wave) is what the
// co-resident workgroup's waves fill.
//
// Everything else follows conv_gemm.hip: LDS-DMA through buffer descriptors (out-of-range lanes get hardware zeros), the
// bank swizzle on the source side (slot s of row r holds chunk s ^ ((r >> 1) & 7)), swapped MFMA operands so that a lane owns
// an output row, the LDS-transposed coalesced epilogue (conv_epilogue), XCD-aware tile order.
#include "conv_gemm_common.h"

using namespace gdlconv;

namespace {

constexpr int DBM = 256, DBN = 128;
constexpr int A_STAGE = DBM * 128;            // 32 KiB
constexpr int B_OFF = 2 * A_STAGE;            // weights behind the two activation stages
constexpr int DUAL_LDS = B_OFF + DBN * 128;   // 80 KiB

// DENSE: 1x1 / stride 1 / pixel-dense input rows (Linear layers, laterals): the source offset is linear in m.
template <bool DENSE>
__global__ __launch_bounds__(256, 2) void conv_gemm_dual_kernel(const KArgs k) {
  constexpr int ES = 2, BKE = 64;
  constexpr int TM = 4, TN = 2, WARPS_N = 2;
  constexpr int CA = 8, CB = 4;                // DMA pieces per wave and K-step: 32 + 16 pieces over four waves
  constexpr unsigned kOob = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const gdl_conv_args& a = k.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;

  const int lid = xcd_remap(blockIdx.x, k.tiles_m * k.tiles_n);
  int tile_m, tile_n;
  tile_order(k, lid, tile_m, tile_n);
  const int m0 = tile_m * DBM, n0 = tile_n * DBN;
  const int z = blockIdx.y;
  const int z0 = z / a.nz_inner, z1 = z % a.nz_inner;
  const int64_t out_zoff = z0 * a.out_sZ0 + z1 * a.out_sZ1;
  const srd_t srd_a = make_srd((const unsigned char*)a.in + (z0 * a.in_sZ0 + z1 * a.in_sZ1) * ES, k.in_span);
  const srd_t srd_b = make_srd((const unsigned char*)a.w + (z0 * a.w_sZ0 + z1 * a.w_sZ1) * ES, k.w_span);
  const unsigned lds_base =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
  // A/B experiment (gdl_debug_set_conv_dbg(8)): static priority for every other tile, so that the two workgroups of a CU
  // fall out of step instead of reaching their epilogues together
  if (k.dbg == 8 && (lid & 1)) __builtin_amdgcn_s_setprio(1);
  // dbg 9 / 10: start-up stagger.  The first 512 workgroups start together (two per CU, observed: block b on XCD b % 8, the
  // CUs of an XCD filled breadth first), run in lockstep and reach their epilogues together -- then nothing overlaps.  9: the
  // second workgroup of every CU waits about half a tile time; 10: additionally the CUs are spread over the first half
  if ((k.dbg == 9 || k.dbg == 10) && blockIdx.x < 512 && blockIdx.y == 0) {
    const unsigned slot = (blockIdx.x >> 3) & 63;                    // 0..31: first workgroup of a CU, 32..63: second
    const unsigned long long period = (unsigned long long)k.KT * 2600 + 12000;
    const unsigned long long wait = k.dbg == 9 ? (slot >= 32 ? period / 2 : 0) : period * slot / 64;
    const unsigned long long t_end = __builtin_readcyclecounter() + wait;
    while (__builtin_readcyclecounter() < t_end) __builtin_amdgcn_s_sleep(16);
  }

  // ---- DMA geometry: wave w, instruction i covers tile rows (i*4 + w)*8 .. +7; lane l writes LDS slot (l & 7) of row
  //      +(l >> 3) and therefore fetches source chunk slot ^ swz(row)
  const int lrow = lane >> 3, lslot = lane & 7;
  const int HoWo = a.Ho * a.Wo;
  int a_voff[CA];
  unsigned a_mask[DENSE ? 1 : CA];
#pragma unroll
  for (int i = 0; i < CA; ++i) {
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
  }
  unsigned b_voff[CB];
#pragma unroll
  for (int i = 0; i < CB; ++i) {
    const int r = (i * 4 + wave) * 8 + lrow;
    const int chunk = lslot ^ ((r >> 1) & 7);
    const int n = n0 + r;
    b_voff[i] = n < a.N ? (unsigned)((n * a.w_sN + chunk * 8) * ES) : kOob;
  }

  int tap_r = 0, tap_s = 0, cc = 0;  // position of the NEXT tile to fetch (K order: channel chunk outer, filter tap inner)
  unsigned wk = 0;                   // its byte offset along the weight rows
  auto issue = [&](int stage) {
    const unsigned lds_a = lds_base + stage * A_STAGE + wave * 1024;
    const unsigned lds_b = lds_base + B_OFF + wave * 1024;
    if constexpr (DENSE) {
#pragma unroll
      for (int i = 0; i < CA; ++i) dma16_buf((unsigned)a_voff[i], srd_a, wk, lds_a + i * 4096);
    } else {
      const int tap_off = (int)((tap_r * a.in_sH + tap_s * a.in_sW + cc * BKE) * ES);
      const unsigned bit = a.pad == 0 ? 1u : 1u << (tap_r * a.S + tap_s);
#pragma unroll
      for (int i = 0; i < CA; ++i) {
        const unsigned v = (a_mask[i] & bit) ? (unsigned)(a_voff[i] + tap_off) : kOob;
        dma16_buf(v, srd_a, 0u, lds_a + i * 4096);
      }
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) dma16_buf(b_voff[i], srd_b, wk, lds_b + i * 4096);
    if constexpr (DENSE) {
      ++cc;
      wk = (unsigned)(cc * BKE * ES);
    } else {
      if (++tap_s == a.S) { tap_s = 0; if (++tap_r == a.R) { tap_r = 0; ++cc; } }
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
  const int b_lds0 = B_OFF + (wn * TN * 32 + frow) * 128;

  uint4 fa[2][TM], fb[4][TN];
  auto fetch_a = [&](const unsigned char* st, int kk, int buf) {
    const int coff = (((2 * kk + fhalf) ^ fswz) << 4);
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[buf][i] = *(const uint4*)(st + a_lds0 + i * 32 * 128 + coff);
  };
  auto fetch_b = [&]() {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int coff = (((2 * kk + fhalf) ^ fswz) << 4);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[kk][j] = *(const uint4*)(smem + b_lds0 + j * 32 * 128 + coff);
    }
  };
  auto mfmas = [&](int kk, int buf) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[kk][j]),
                                                            __builtin_bit_cast(bf16x8_t, fa[buf][i]), acc[i][j], 0, 0, 0);
  };

  const unsigned long long t0c = k.probe ? __builtin_readcyclecounter() : 0;
  const unsigned long long t0r = k.probe ? __builtin_amdgcn_s_memrealtime() : 0;
  issue(0);
  // all scalar (kernel-argument) loads are complete here: the waitcnt inserter may then count LDS reads only
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
  for (int kt = 0; kt < k.KT; ++kt) {
    const unsigned char* st = smem + (kt & 1) * A_STAGE;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of tile kt has landed
    __builtin_amdgcn_s_barrier();                        // ... and everyone's; nobody reads the other activation stage any more
    fetch_b();
    fetch_a(st, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the weight fragments of this K-step are in registers
    __builtin_amdgcn_s_barrier();                        // ... in everyone's: the weight buffer is free
    if (kt + 1 < k.KT && k.dbg != 1) issue((kt + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk < 3) fetch_a(st, kk + 1, (kk + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(kk, kk & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (k.probe && tid == 0 && blockIdx.x < 2048) {
    k.probe[2 * blockIdx.x] = __builtin_readcyclecounter() - t0c;
    k.probe[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
  }
  __builtin_amdgcn_s_barrier();   // every wave is past its last fragment read: the stages become the epilogue's transpose buffers
  conv_epilogue<TM, TN, false>(k, acc, m0, n0, wm, wn, lane, out_zoff, smem + wave * 8192, smem + 4 * 8192 + wave * 1024);
  if (k.probe && tid == 0 && blockIdx.x < 2048) {
    k.probe[4096 + blockIdx.x] = __builtin_readcyclecounter() - t0c;
    k.probe[8192 + 2 * blockIdx.x] = t0r;                                   // block timeline (100 MHz ticks)
    k.probe[8192 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
}

}  // namespace

namespace gdlconv {

bool conv_gemm_dual_applicable(const gdl_conv_args& a) {
  return a.dtype == GDL_BF16 && a.C % 64 == 0 && !a.aux_out && a.act != GDL_ACT_MUL_GELU_GRAD &&
         (a.pad == 0 || a.R * a.S <= 32);
}

int conv_gemm_dual_launch(const KArgs& k, hipStream_t stream) {
  KArgs kk = k;
  kk.tiles_m = (k.M + DBM - 1) / DBM;
  kk.tiles_n = (k.a.N + DBN - 1) / DBN;
  kk.n_group = conv_n_group(k.a, DBM, DBN, 64);
  dim3 grid(kk.tiles_m * kk.tiles_n, k.a.nz), block(256);
  if (k.in_dense) {
    GDL_SET_MAX_LDS_ONCE(conv_gemm_dual_kernel<true>, DUAL_LDS);
    hipLaunchKernelGGL(conv_gemm_dual_kernel<true>, grid, block, DUAL_LDS, stream, kk);
  } else {
    GDL_SET_MAX_LDS_ONCE(conv_gemm_dual_kernel<false>, DUAL_LDS);
    hipLaunchKernelGGL(conv_gemm_dual_kernel<false>, grid, block, DUAL_LDS, stream, kk);
  }
  GDL_CHECK_LAUNCH("gdl_conv_gemm(dual-resident 256x128)");
  return GDL_OK;
}

}  // namespace gdlconv
