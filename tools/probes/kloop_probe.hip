// Probe: a skeleton of the two-stage 256x256 ping-pong K loop (conv_gemm.hip, variant 3) that can be stripped down
// feature by feature, to find what makes a K-step cost ~2900 cycles instead of the 2048 of its 64 MFMAs per SIMD.
//   FR  fragments are read from LDS (ds_read_b128, double-buffered) and feed the MFMAs; else the MFMAs use constants
//   AV  per-piece address VALU of the implicit GEMM (tap mask test + offset add + OOB select) ; else a fixed voffset
//   PP  ping-pong loader roles (waves 0-3 / 4-7 alternate, 16 pieces each); else all 8 waves issue 8 pieces
//   LA  the wave waits for its PREVIOUS burst only (needs a third stage in a real kernel) instead of the one just issued
// One workgroup per CU, 128 KiB LDS, L2-hot source (the real kernel's K-step does not change with hot operands either).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/kloop_probe.hip -o /tmp/kl && /tmp/kl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned srd_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ void dma16(unsigned voff, srd_t srd, unsigned soff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(srd), "s"(soff), "s"(lds_addr) : "memory", "m0");
}

template <bool FR, bool AV, bool PP, bool LA, bool NODMA = false, int PRIO = 0>
__global__ __launch_bounds__(512) void k(const char* src, int iters, unsigned tapbits, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = 65536, NW = PP ? 4 : 8, CA = 256 / (8 * NW), CB = CA;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / 4, wn = wave % 4, half = PP ? wave >> 2 : 0, lwave = PP ? wave & 3 : wave;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
  const unsigned region = 1u << 20;
  unsigned long long a = (unsigned long long)(src + (size_t)(blockIdx.x & 7) * region);   // 8 regions: L2-hot
  srd_t s;
  s.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  s.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffff);
  s.z = region;
  s.w = 0x00020000;
  const int lrow = lane >> 3, lslot = lane & 7;
  int a_voff[CA];
  unsigned a_mask[CA], b_voff[CB];
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int r = (i * NW + lwave) * 8 + lrow;
    a_voff[i] = r * 1536 + ((lslot ^ ((r >> 1) & 7)) << 4);
    a_mask[i] = tapbits | (unsigned)r;          // runtime value: all taps valid, but the compiler cannot know
    b_voff[i] = 524288u + r * 1536 + ((lslot ^ ((r >> 1) & 7)) << 4);
  }
  int tap = 0;
  auto issue = [&](int stage) {
    const unsigned lds_a = lds_base + stage * STAGE + lwave * 1024, lds_b = lds_a + 32768;
    const int tap_off = tap * 128;
    const unsigned bit = 1u << tap;
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      unsigned v = (unsigned)a_voff[i];
      if (AV) v = (a_mask[i] & bit) ? (unsigned)(a_voff[i] + tap_off) : 0x80000000u;
      dma16(v, s, AV ? 0u : (unsigned)tap_off, lds_a + i * NW * 1024);
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) dma16(b_voff[i], s, (unsigned)tap_off, lds_b + i * NW * 1024);
    tap = (tap + 1) % 9;
  };
  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frow = lane & 31, fhalf = lane >> 5, fswz = (frow >> 1) & 7;
  const int a_lds0 = (wm * 128 + frow) * 128, b_lds0 = 32768 + (wn * 64 + frow) * 128;
  uint4 fa[2][4], fb[2][2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[b][i] = make_uint4(lane + i, lane * 3, b, i);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[b][j] = make_uint4(lane ^ j, lane * 5, b, j);
  }
  auto fetch = [&](const unsigned char* st, int kk, int buf) {
    if (!FR) return;
    const int coff = (((2 * kk + fhalf) ^ fswz) << 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[buf][i] = *(const uint4*)(st + a_lds0 + i * 4096 + coff);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[buf][j] = *(const uint4*)(st + b_lds0 + j * 4096 + coff);
  };
  auto mfmas = [&](int buf) {
    if (PRIO == 1) __builtin_amdgcn_s_setprio(3);      // PRIO 1: the wave in its MFMA group outranks its SIMD partner
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[buf][j]),
                                                            __builtin_bit_cast(bf16x8_t, fa[buf][i]), acc[i][j], 0, 0, 0);
    if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
  };
  const int KT = iters;
  unsigned long long t_issue = 0;
  if (!PP || half == 0) issue(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (!PP || half == 1) issue(1);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  const unsigned long long t0 = clock64();
  fetch(smem, 0, 0);
  for (int kt = 0; kt < KT; ++kt) {
    const unsigned char* st = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
      fetch(st, kk + 1, (kk + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(kk & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (LA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CA + CB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!NODMA && (!PP || half == (kt & 1))) {
      const unsigned long long b0 = clock64();
      if (PRIO == 2) __builtin_amdgcn_s_setprio(3);    // PRIO 2: the loader outranks its partner while it issues DMA
      issue(kt & 1);
      if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
      t_issue += clock64() - b0;
    }
    fetch(smem + ((kt + 1) & 1) * STAGE, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(1);
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) t += acc[i][j][0] + acc[i][j][15];
  if (lane == 0 && blockIdx.x == 0) { out[2 * wave] = t1 - t0; out[2 * wave + 1] = t_issue; }
  if (t == 1.2345f) out[31] = 1;
}

template <bool FR, bool AV, bool PP, bool LA, bool NODMA = false, int PRIO = 0>
static void run(const char* d, unsigned long long* dout) {
  const int iters = 1000;
  auto kern = k<FR, AV, PP, LA, NODMA, PRIO>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 131072, 0, d, iters, 0x1ffu, dout);
    hipDeviceSynchronize();
  }
  unsigned long long h[16];
  hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  const int pieces = PP ? 16 : 16;   // A + B pieces per issuing wave per issue() call
  const double calls = PP ? iters / 2.0 : iters;
  printf("%s%s%s %s %s %s: %7.1f cyc / K-step (MFMA floor 2048), DMA issue %5.1f cyc/piece (wave 0), %5.1f (wave 4)\n",
         PRIO == 1 ? "[setprio: MFMA] " : PRIO == 2 ? "[setprio: DMA issue] " : "", NODMA ? "[no DMA] " : "", FR ? "LDS fragments " : "const operands", AV ? "addr VALU" : "fixed addr", PP ? "ping-pong loaders" : "all waves load   ",
         LA ? "lookahead" : "no lookahead", (double)h[0] / iters, (double)h[1] / calls / pieces, (double)h[9] / calls / pieces);
}


// Balanced, interleaved schedule: EVERY wave loads 8 of the 64 pieces of a tile (rows wave*32 .. +31 of both operands),
// N3 of them around the MFMA group that follows the barrier (kk = 3 of K-step kt), N0 around group 0 and N1 around group 1
// of K-step kt+1 (N3 + N0 + N1 = 8); waves 0-3 issue their pieces BEFORE the group's MFMAs, waves 4-7 (their SIMD partners)
// AFTER them, so that one wave of a SIMD is in its MFMAs while the other is in DMA issue.  Two stages, one barrier per K-step.
template <int N3, int N0, int N1>
__global__ __launch_bounds__(512) void kb(const char* src, int iters, unsigned tapbits, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = 65536;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / 4, wn = wave % 4, late = wave >> 2;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
  const unsigned region = 1u << 20;
  unsigned long long a = (unsigned long long)(src + (size_t)(blockIdx.x & 7) * region);
  srd_t s;
  s.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  s.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffff);
  s.z = region;
  s.w = 0x00020000;
  const int lrow = lane >> 3, lslot = lane & 7;
  int voff[8];             // pieces 0-3: activation rows wave*32 + i*8 + lrow, pieces 4-7: weight rows
  unsigned a_mask[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave * 32 + i * 8 + lrow;
    voff[i] = r * 1536 + ((lslot ^ ((r >> 1) & 7)) << 4);
    a_mask[i] = tapbits | (unsigned)r;
    voff[4 + i] = 524288 + r * 1536 + ((lslot ^ ((r >> 1) & 7)) << 4);
  }
  int tap = 0;
  auto piece = [&](int stage, int i) {   // i is a compile-time constant at every call site
    const unsigned lds_a = lds_base + stage * STAGE + wave * 4096;
    const int tap_off = tap * 128;
    if (i < 4) {
      const unsigned v = (a_mask[i] & (1u << tap)) ? (unsigned)(voff[i] + tap_off) : 0x80000000u;
      dma16(v, s, 0u, lds_a + i * 1024);
    } else {
      dma16((unsigned)voff[i], s, (unsigned)tap_off, lds_a + 32768 + (i - 4) * 1024);
    }
  };
  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frow = lane & 31, fhalf = lane >> 5, fswz = (frow >> 1) & 7;
  const int a_lds0 = (wm * 128 + frow) * 128, b_lds0 = 32768 + (wn * 64 + frow) * 128;
  uint4 fa[2][4], fb[2][2];
  auto fetch = [&](const unsigned char* st, int kk, int buf) {
    const int coff = (((2 * kk + fhalf) ^ fswz) << 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[buf][i] = *(const uint4*)(st + a_lds0 + i * 4096 + coff);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[buf][j] = *(const uint4*)(st + b_lds0 + j * 4096 + coff);
  };
  auto mfmas = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[buf][j]),
                                                            __builtin_bit_cast(bf16x8_t, fa[buf][i]), acc[i][j], 0, 0, 0);
  };
  // one MFMA group with `n` pieces (first .. first+n-1) of the tile that goes into `stage`, before or after the MFMAs
  auto group = [&](int buf, int stage, auto first, auto n) {
    __builtin_amdgcn_sched_barrier(0);
    if (!late) {
#pragma unroll
      for (int i = 0; i < n(); ++i) piece(stage, first() + i);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfmas(buf);
    __builtin_amdgcn_sched_barrier(0);
    if (late) {
#pragma unroll
      for (int i = 0; i < n(); ++i) piece(stage, first() + i);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  const int KT = iters;
#pragma unroll
  for (int i = 0; i < 8; ++i) piece(0, i);
  tap = 1;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) piece(1, i);
  tap = 2;
  __builtin_amdgcn_s_waitcnt(0xC07F);
  const unsigned long long t0 = clock64();
  fetch(smem, 0, 0);
  for (int kt = 0; kt < KT; ++kt) {
    const unsigned char* st = smem + (kt & 1) * STAGE;
    // groups 0 and 1 carry the rest of tile kt+1's pieces (into the other stage, vacated at the previous barrier)
    fetch(st, 1, 1);
    group(0, (kt + 1) & 1, [] { return N3; }, [] { return N0; });
    fetch(st, 2, 0);
    group(1, (kt + 1) & 1, [] { return N3 + N0; }, [] { return N1; });
    if (kt > 0) tap = (tap + 1) % 9;            // tile kt+1 is complete (issued), next pieces belong to tile kt+2
    fetch(st, 3, 1);
    group(0, 0, [] { return 0; }, [] { return 0; });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    fetch(smem + ((kt + 1) & 1) * STAGE, 0, 0);
    group(1, kt & 1, [] { return 0; }, [] { return N3; });   // first pieces of tile kt+2 into the stage just vacated
  }
  const unsigned long long t1 = clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) t += acc[i][j][0] + acc[i][j][15];
  if (lane == 0 && blockIdx.x == 0) out[2 * wave] = t1 - t0;
  if (t == 1.2345f) out[31] = 1;
}

template <int N3, int N0, int N1>
static void run_b(const char* d, unsigned long long* dout) {
  const int iters = 1000;
  auto kern = kb<N3, N0, N1>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 131072, 0, d, iters, 0x1ffu, dout);
    hipDeviceSynchronize();
  }
  unsigned long long h[16];
  hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  printf("balanced interleaved schedule, pieces per wave around group 3 / 0 / 1 = %d / %d / %d: %7.1f cyc / K-step (wave 0), %7.1f (wave 4)\n",
         N3, N0, N1, (double)h[0] / iters, (double)h[8] / iters);
}


// "B direct": the weight fragments never touch the LDS.  Weights pre-packed in MFMA fragment order (one 1 KiB block =
// the 64 lanes x 16 B of one 32-row x 16-k fragment) are loaded straight into VGPRs, one K-step ahead (32 VGPRs), two
// loads per MFMA group; only the activation tile goes through LDS-DMA (8 pieces per loader wave) and ds_read (4 per group).
// LDS traffic per K-step: 32 KiB written + 128 KiB read instead of 64 + 192; TA traffic 32 + 64 KiB instead of 64.
typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
__device__ __forceinline__ u4_t ldg16(unsigned voff, srd_t srd, unsigned soff) {
  u4_t v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(srd), "s"(soff) : "memory");
  return v;
}
template <bool PP>
__global__ __launch_bounds__(512) void kg(const char* src, int iters, unsigned tapbits, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = 32768, NW = PP ? 4 : 8, CA = 256 / (8 * NW);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / 4, wn = wave % 4, half = PP ? wave >> 2 : 0, lwave = PP ? wave & 3 : wave;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
  const unsigned region = 1u << 20;
  unsigned long long a = (unsigned long long)(src + (size_t)(blockIdx.x & 7) * region);
  srd_t s;
  s.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  s.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffff);
  s.z = region;
  s.w = 0x00020000;
  const int lrow = lane >> 3, lslot = lane & 7;
  int a_voff[CA];
  unsigned a_mask[CA];
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int r = (i * NW + lwave) * 8 + lrow;
    a_voff[i] = r * 1536 + ((lslot ^ ((r >> 1) & 7)) << 4);
    a_mask[i] = tapbits | (unsigned)r;
  }
  int tap = 0;
  auto issue = [&](int stage) {
    const unsigned lds_a = lds_base + stage * STAGE + lwave * 1024;
    const int tap_off = tap * 128;
    const unsigned bit = 1u << tap;
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const unsigned v = (a_mask[i] & bit) ? (unsigned)(a_voff[i] + tap_off) : 0x80000000u;
      dma16(v, s, 0u, lds_a + i * NW * 1024);
    }
    tap = (tap + 1) % 9;
  };
  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frow = lane & 31, fhalf = lane >> 5, fswz = (frow >> 1) & 7;
  const int a_lds0 = (wm * 128 + frow) * 128;
  uint4 fa[2][4];
  u4_t fb[4][2];
  const unsigned bvoff = lane * 16;
  unsigned bpos = 524288u + wn * 65536u;          // this wave's packed weight stream (advances 8 KiB per K-step)
  auto fetch = [&](const unsigned char* st, int kk, int buf) {
    const int coff = (((2 * kk + fhalf) ^ fswz) << 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[buf][i] = *(const uint4*)(st + a_lds0 + i * 4096 + coff);
  };
  auto loadb = [&](int g) {
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[g][j] = ldg16(bvoff, s, (bpos + (g * 2 + j) * 1024) & 0xfffffu);
  };
  auto mfmas = [&](int buf, int g) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[g][j]),
                                                            __builtin_bit_cast(bf16x8_t, fa[buf][i]), acc[i][j], 0, 0, 0);
  };
  const int KT = iters;
  if (!PP || half == 0) issue(0);
#pragma unroll
  for (int g = 0; g < 4; ++g) loadb(g);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (!PP || half == 1) issue(1);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  const unsigned long long t0 = clock64();
  fetch(smem, 0, 0);
  // vm ops of a wave, in issue order, per K-step kt: B(kt+1,0) B(kt+1,1) B(kt+1,2) [8 or 4 A pieces if it loads] B(kt+1,3),
  // two loads per B group.  Before the MFMAs of group g the loads of B(kt, g) must have landed.
  bool loaded_prev = PP ? half == 1 : true;         // did this wave issue A pieces in the previous K-step?
  for (int kt = 0; kt < KT; ++kt) {
    const unsigned char* st = smem + (kt & 1) * STAGE;
    const bool loads_now = !PP || half == (kt & 1);
    bpos += 8192;
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
      fetch(st, kk + 1, (kk + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      // outstanding younger ops than B(kt,kk): the B groups kk+1..3 of this K-step's stream issued last K-step (2 each),
      // the A pieces of last K-step if any (they sit between group 2 and group 3), and this K-step's groups 0..kk-1
      if (kk == 0) { if (loaded_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + CA) : "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
      if (kk == 1) { if (loaded_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + CA) : "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
      if (kk == 2) { if (loaded_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + CA) : "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
      asm volatile("" : "+v"(fb[kk][0]), "+v"(fb[kk][1]));
      mfmas(kk & 1, kk);
      __builtin_amdgcn_sched_barrier(0);
      loadb(kk);                                   // B(kt+1, kk) into the registers just consumed
      __builtin_amdgcn_sched_barrier(0);
    }
    // A pieces of tile kt+1 (issued last K-step, followed by B(kt,3) and this K-step's 6 B loads) have landed
    if (loaded_prev) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();
    if (loads_now) issue(kt & 1);
    fetch(smem + ((kt + 1) & 1) * STAGE, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (loads_now) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + CA) : "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    asm volatile("" : "+v"(fb[3][0]), "+v"(fb[3][1]));
    mfmas(1, 3);
    __builtin_amdgcn_sched_barrier(0);
    loadb(3);
    __builtin_amdgcn_sched_barrier(0);
    loaded_prev = loads_now;
  }
  const unsigned long long t1 = clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) t += acc[i][j][0] + acc[i][j][15];
  if (lane == 0 && blockIdx.x == 0) out[2 * wave] = t1 - t0;
  if (t == 1.2345f) out[31] = 1;
}
template <bool PP>
static void run_g(const char* d, unsigned long long* dout) {
  const int iters = 1000;
  auto kern = kg<PP>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 65536, 0, d, iters, 0x1ffu, dout);
    hipDeviceSynchronize();
  }
  unsigned long long h[16];
  hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  printf("B direct to VGPRs (packed fragments), A through LDS, %s: %7.1f cyc / K-step (wave 0), %7.1f (wave 4)\n",
         PP ? "ping-pong loaders" : "all waves load   ", (double)h[0] / iters, (double)h[8] / iters);
}

int main() {
  char* d; unsigned long long* o;
  hipMalloc(&d, (size_t)256 << 20); hipMemset(d, 1, (size_t)256 << 20);
  hipMalloc(&o, 256);
  run<true, true, true, false, false, 0>(d, o);
  run<true, true, true, false, false, 1>(d, o);
  run<true, true, true, false, false, 2>(d, o);
  run<true, true, false, false, false, 1>(d, o);
  return 0;
}
