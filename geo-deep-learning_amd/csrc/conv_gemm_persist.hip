// Persistent form of the 256 x 256 ping-pong tile for dense 1x1 layers / linears in bf16 (round 4).
//
// Why.  One 256^2 workgroup owns a CU (128 KiB of LDS), so between two tiles the CU's matrix pipe idles through the next
// workgroup's launch, its address set-up and the first stage's trip to memory: ~5 k cycles per tile in the block timeline
// (tools/probe_gemm_timeline.py: 12 K-steps take 34-35 k cycles instead of 12 x 2850).  That is 10 % of a ViT qkv tile and
// 23 % of the 4-K-step tiles of the decoder's K = 256 layers, which the round-3 notes listed as "output-bound at 3 TB/s".
// Here ONE workgroup per CU walks its share of the tiles (same XCD-aware order as the one-tile kernel: tile j of workgroup b
// is virtual block b + j * gridDim.x) and issues the FIRST K-stage of its next tile before it starts the epilogue of the
// current one: the loads land while the epilogue runs (it stages through the OTHER stage's LDS), and the next K loop starts
// with its data resident.
//
// What makes it cheap: dense 1x1 only.  A lane's source offset is then linear in the tile row, so the per-tile DMA state is two
// registers (activation / weight offset of the lane's first row; the 8 pieces per operand add a wave-uniform stride), and rows
// past M need no mask: their offsets lie past the buffer descriptor's num_records and the hardware writes zeros
// (tools/probes/buffer_lds_probe.hip).  The K loop, the ping-pong loader roles, the swizzle, the MFMA operand order and the
// epilogue are conv_gemm.hip's, bit for bit (tests compare this variant with the 128^2 tile for equality).
//
// Counter note: the loader half's `s_waitcnt vmcnt(0)` at the top of a tile also waits for that wave's own epilogue stores
// (loads and stores share vmcnt on gfx9 and may retire out of order with respect to each other, so no counted wait can
// separate them); the other half skips the wait and only meets the barrier.
#include "conv_gemm_common.h"

using namespace gdlconv;

namespace {

constexpr int PS_STAGE = 65536;                 // [A: 256 rows | B: 256 rows] x 128 B
constexpr int PS_LDS = 2 * PS_STAGE + 8192;     // + the epilogue's per-wave constant rows

__global__ __launch_bounds__(512) void conv_gemm_persist_kernel(const KArgs) {
  constexpr int ES = 2;
  constexpr int TM = 4, TN = 2, WARPS_N = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef const __attribute__((address_space(4))) KArgs* kargs_ptr;
  using yes = std::integral_constant<bool, true>;
  using no = std::integral_constant<bool, false>;

  f32x16_t acc[TM][TN];
  int v = blockIdx.x;                       // virtual block index of the current tile
  bool first_tile = true;
  for (;;) {
    // Everything but the accumulators and v is RE-DERIVED per tile from the kernel-argument segment and the thread id, behind
    // an empty asm the optimiser cannot see through.  Written the usual way (geometry computed once in front of the tile
    // loop) the values stay live across the epilogue and the epilogue's tile-independent terms are hoisted in front of the
    // loop: 116 scalar + 142 vector spills around the K loop, against 17 / 0 in the one-tile kernel.
    kargs_ptr kp = (kargs_ptr)__builtin_amdgcn_kernarg_segment_ptr();   // KArgs is the only argument
    int tid = threadIdx.x;
    asm volatile("" : "+s"(kp), "+v"(tid));
    const KArgs& k = *(const KArgs*)kp;
    const gdl_conv_args& a = k.a;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WARPS_N, wn = wave % WARPS_N;
    const int ntiles = k.tiles_m * k.tiles_n;
    const srd_t srd_a = make_srd(a.in, k.in_span);
    const srd_t srd_b = make_srd(a.w, k.w_span);
    const unsigned lds_base =
        __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);

    // ---- DMA geometry (conv_gemm.hip): loading wave w of a half, piece i covers tile rows (i*4 + w)*8 .. +7; lane l writes LDS
    //      slot (l & 7) of row +(l >> 3) and fetches source chunk slot ^ swz(row) -- the same chunk for every piece
    const int lrow = lane >> 3, lslot = lane & 7;
    const int lwave = wave & 3, half = wave >> 2;      // waves w and w+4 share a SIMD and alternate as loaders
    const int chunk = lslot ^ ((((lwave & 1) << 2) + (lrow >> 1)) & 7);
    const unsigned a_row = (unsigned)a.in_sW * ES, b_row = (unsigned)a.w_sN * ES;     // bytes per activation / weight row
    const unsigned a_lane = (unsigned)(lwave * 8 + lrow) * a_row + chunk * 16;
    const unsigned b_lane = (unsigned)(lwave * 8 + lrow) * b_row + chunk * 16;
    auto issue = [&](int stage, unsigned a_v, unsigned b_v, unsigned wk) {
      const unsigned lds_a = lds_base + stage * PS_STAGE + lwave * 1024;
      const unsigned lds_b = lds_a + 256 * 128;
#pragma unroll
      for (int i = 0; i < 8; ++i) dma16_buf(a_v + i * 32 * a_row, srd_a, wk, lds_a + i * 4096);
#pragma unroll
      for (int i = 0; i < 8; ++i) dma16_buf(b_v + i * 32 * b_row, srd_b, wk, lds_b + i * 4096);
    };

    const int frow = lane & 31, fhalf = lane >> 5;
    const int fswz = (frow >> 1) & 7;
    const int a_lds0 = (wm * TM * 32 + frow) * 128;
    const int b_lds0 = 256 * 128 + (wn * TN * 32 + frow) * 128;
    uint4 fa[2][TM], fb[2][TN];
    auto fetch = [&](const unsigned char* st, int kk, int buf) {
      const int coff = (((2 * kk + fhalf) ^ fswz) << 4);
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[buf][i] = *(const uint4*)(st + a_lds0 + i * 32 * 128 + coff);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[buf][j] = *(const uint4*)(st + b_lds0 + j * 32 * 128 + coff);
    };
    // FIRST: the tile's first k16 group starts the sums from the constant 0 -- the accumulators are then dead between a tile's
    // epilogue and the next tile's first MFMAs instead of being 128 zero-filled registers carried around the tile loop
    auto mfmas = [&](int buf, auto first) {
      constexpr bool FIRST = decltype(first)::value;
      const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[buf][j]),
                                                              __builtin_bit_cast(bf16x8_t, fa[buf][i]), FIRST ? zero : acc[i][j], 0, 0, 0);
    };

    int tile_m, tile_n;
    tile_order(k, xcd_remap(v, ntiles), tile_m, tile_n);
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const unsigned a_v = a_lane + (unsigned)m0 * a_row, b_v = b_lane + (unsigned)n0 * b_row;
    if (first_tile && half == 0) issue(0, a_v, b_v, 0u);     // later tiles: issued in front of the previous tile's epilogue
    first_tile = false;

    const unsigned long long t0c = k.probe ? __builtin_readcyclecounter() : 0;
    const unsigned long long t0r = k.probe ? __builtin_amdgcn_s_memrealtime() : 0;
    if (half == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the loaders' share of stage 0 has landed
    __syncthreads();                        // ... everyone's; and every wave is out of the previous tile's epilogue
    if (k.KT > 1 && half == 1) issue(1, a_v, b_v, 128u);
    __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0): scalar loads complete, the waitcnt inserter then counts LDS reads only
    fetch(smem, 0, 0);
    auto kstep = [&](auto first, int kt) {
      const unsigned char* st = smem + (kt & 1) * PS_STAGE;
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        fetch(st, kk + 1, (kk + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        if (kk == 0) mfmas(0, first); else mfmas(kk & 1, no{});
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kt + 1 < k.KT) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of tile kt+1 has landed
        __syncthreads();                                   // ... and everyone's; nobody reads tile kt's stage any more
        if (kt + 2 < k.KT && half == (kt & 1)) issue(kt & 1, a_v, b_v, (unsigned)(kt + 2) * 128u);
        fetch(smem + ((kt + 1) & 1) * PS_STAGE, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfmas(1, no{});
      __builtin_amdgcn_sched_barrier(0);
    };
    kstep(yes{}, 0);
    for (int kt = 1; kt < k.KT; ++kt) kstep(no{}, kt);
    if (k.probe && tid == 0 && v < 2048) {
      k.probe[2 * v] = __builtin_readcyclecounter() - t0c;
      k.probe[2 * v + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
    }
    __syncthreads();   // every wave is past its last fragment read: both stages are free
    // the next tile's first stage goes out now and lands under the epilogue, which stages through stage 1's memory
    const int vn = v + (int)gridDim.x;
    const bool more = vn < ntiles;
    if (more && half == 0 && k.dbg != 11) {
      int tm2, tn2;
      tile_order(k, xcd_remap(vn, ntiles), tm2, tn2);
      issue(0, a_lane + (unsigned)(tm2 * 256) * a_row, b_lane + (unsigned)(tn2 * 256) * b_row, 0u);
    }
    {
      kargs_ptr kq = (kargs_ptr)__builtin_amdgcn_kernarg_segment_ptr();
      int tid_e = threadIdx.x, m0_e = m0, n0_e = n0;
      asm volatile("" : "+s"(kq), "+v"(tid_e), "+s"(m0_e), "+s"(n0_e));
      const KArgs& ke = *(const KArgs*)kq;
      const int wave_e = __builtin_amdgcn_readfirstlane(tid_e >> 6);
      conv_epilogue<TM, TN, false>(ke, acc, m0_e, n0_e, wave_e / WARPS_N, wave_e % WARPS_N, tid_e & 63, 0,
                                   smem + PS_STAGE + wave_e * 8192, smem + 2 * PS_STAGE + wave_e * 1024);
      if (ke.probe && tid_e == 0 && v < 2048) {
        ke.probe[4096 + v] = __builtin_readcyclecounter() - t0c;
        ke.probe[8192 + 2 * v] = t0r;                                   // tile timeline (100 MHz ticks)
        ke.probe[8192 + 2 * v + 1] = __builtin_amdgcn_s_memrealtime();
      }
      if (!more) break;
      if (ke.dbg == 11 && wave_e < 4) first_tile = true;               // tuning: no prefetch under the epilogue
    }
    v = vn;
  }
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

}  // namespace

namespace gdlconv {

bool conv_gemm_persist_applicable(const gdl_conv_args& a) {
  return a.dtype == GDL_BF16 && a.C % 64 == 0 && a.N % 256 == 0 && a.nz == 1 && !a.aux_out && a.act != GDL_ACT_MUL_GELU_GRAD &&
         a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0 && a.H == a.Ho && a.W == a.Wo &&
         a.in_sH == (int64_t)a.W * a.in_sW && a.in_sB == (int64_t)a.H * a.in_sH;
}

int conv_gemm_persist_launch(const KArgs& k, hipStream_t stream) {
  KArgs kk = k;
  kk.tiles_m = (k.M + 255) / 256;
  kk.tiles_n = (k.a.N + 255) / 256;
  kk.n_group = conv_n_group(k.a, 256, 256, 32);
  const int ntiles = kk.tiles_m * kk.tiles_n;
  const int cus = num_cus();
  dim3 grid(ntiles < cus ? ntiles : cus), block(512);
  GDL_SET_MAX_LDS_ONCE(conv_gemm_persist_kernel, PS_LDS);
  hipLaunchKernelGGL(conv_gemm_persist_kernel, grid, block, PS_LDS, stream, kk);
  GDL_CHECK_LAUNCH("gdl_conv_gemm(256x256 persistent)");
  return GDL_OK;
}

}  // namespace gdlconv
