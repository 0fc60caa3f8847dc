// Probe: skeleton of a 256x256 K loop with ONE wave per SIMD (4 waves, wave tile 128 x 128 = 256 accumulator registers,
// <= 512 VGPRs), next to the 8-wave ping-pong skeleton of kloop_probe.hip (2533 cycles per K-step with LDS fragments + DMA,
// MFMA floor 2048).  Per K-step a wave reads 8 fragments per k16 group (4 activation + 4 weight: 32 ds_read_b128 for 64
// MFMAs -- a third fewer LDS reads per MFMA than the 128 x 64 wave tile) and issues 16 of the 64 DMA pieces.
//   FR    fragments come from the LDS (double-buffered per k16 group); else constants
//   DMA   0 none, 1 burst: all 16 pieces right after the barrier (tile kt+2, a full step to land), 2 spread: two pieces behind
//         every four MFMAs of the first two k16 groups (tile kt+1, the third group's 512 MFMA cycles to land)
//   PF    fragment prefetch distance in k16 groups (1 or 2)
// Two 64 KiB stages, one barrier per K-step, DMA of tile t+2 issued during step t+1 into the stage tile t vacated.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/kloop4_probe.hip -o /tmp/kl4 && /tmp/kl4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned srd_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ void dma16(unsigned voff, srd_t srd, unsigned soff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(srd), "s"(soff), "s"(lds_addr) : "memory", "m0");
}

template <bool FR, int DMA, int PF, bool RS = false, bool FINE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k4(const char* src, int iters, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = 65536;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
  const unsigned region = 1u << 20;
  unsigned long long a = (unsigned long long)(src + (size_t)(blockIdx.x & 7) * region);   // 8 regions: L2-hot
  srd_t s;
  s.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  s.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffff);
  s.z = region;
  s.w = 0x00020000;
  const int lrow = lane >> 3, lslot = lane & 7;
  unsigned voff[16];      // pieces 0-7: activation rows (i*4 + wave)*8 + lrow, pieces 8-15: weight rows
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = (i * 4 + wave) * 8 + lrow;
    voff[i] = r * 1536 + ((lslot ^ ((r >> 1) & 7)) << 4);
    voff[8 + i] = 524288u + r * 1536 + ((lslot ^ ((r >> 1) & 7)) << 4);
  }
  int tap = 0;
  auto piece = [&](int stage, int i) {   // i compile-time
    const unsigned lds = lds_base + stage * STAGE + (i < 8 ? 0 : 32768) + ((i & 7) * 4 + wave) * 1024;
    dma16(voff[i], s, (unsigned)(tap * 128), lds);
  };
  f32x16_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frow = lane & 31, fhalf = lane >> 5, fswz = (frow >> 1) & 7;
  const int a_lds0 = (wm * 128 + frow) * 128, b_lds0 = 32768 + (wn * 128 + frow) * 128;
  uint4 fa[3][4], fb[3][4];
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[b][i] = make_uint4(lane + i, lane * 3, b, i);
      fb[b][i] = make_uint4(lane ^ i, lane * 5, b, i);
    }
  auto fetch = [&](const unsigned char* st, int kk, int buf) {
    if (!FR) return;
    const int coff = (((2 * kk + fhalf) ^ fswz) << 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[buf][i] = *(const uint4*)(st + a_lds0 + i * 4096 + coff);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[buf][j] = *(const uint4*)(st + b_lds0 + j * 4096 + coff);
  };
  // 16 MFMAs of one k16 group in four micro-groups of four; `hook(q)` runs between them (DMA pieces go there)
  auto mfma4 = [&](int buf, int q) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      acc[q][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[buf][j]),
                                                          __builtin_bit_cast(bf16x8_t, fa[buf][q]), acc[q][j], 0, 0, 0);
  };
  const int KT = iters;
  // prologue: tile 0 -> stage 0, tile 1 -> stage 1
  if (DMA) {
#pragma unroll
    for (int i = 0; i < 16; ++i) piece(0, i);
    tap = 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (DMA) {
#pragma unroll
    for (int i = 0; i < 16; ++i) piece(1, i);
    tap = 2;
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  const unsigned long long t0 = clock64();
  fetch(smem, 0, 0);
  if (PF == 2) fetch(smem, 1, 1);
  if constexpr (RS) {
    // fully interleaved variant: behind every micro-group of four MFMAs two fragment reads of the NEXT k16 group (weights in
    // the first two micro-groups, activations in the last two) and 0-2 DMA pieces: tile kt+1 is issued from group 3 of step
    // kt-1 (4 pieces, right after the barrier that freed its stage) through groups 0 and 1 of step kt (6 + 6), group 2 is
    // landing time, then wait + barrier
    auto read2 = [&](const unsigned char* stg, int kk, int buf, int q) {
      if (!FR) return;
      const int coff = (((2 * kk + fhalf) ^ fswz) << 4);
      if (q < 2) {
        fb[buf][2 * q] = *(const uint4*)(stg + b_lds0 + (2 * q) * 4096 + coff);
        fb[buf][2 * q + 1] = *(const uint4*)(stg + b_lds0 + (2 * q + 1) * 4096 + coff);
      } else {
        fa[buf][2 * (q - 2)] = *(const uint4*)(stg + a_lds0 + (2 * (q - 2)) * 4096 + coff);
        fa[buf][2 * (q - 2) + 1] = *(const uint4*)(stg + a_lds0 + (2 * (q - 2) + 1) * 4096 + coff);
      }
    };
    for (int kt = 0; kt < KT; ++kt) {
      const unsigned char* st = smem + (kt & 1) * STAGE;
      const unsigned char* nx = smem + ((kt + 1) & 1) * STAGE;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if constexpr (FINE) {
            // one non-MFMA instruction group behind EACH MFMA: M R M R M D M D
            auto one = [&](int j) {
              acc[q][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[kk & 1][j]),
                                                                  __builtin_bit_cast(bf16x8_t, fa[kk & 1][q]), acc[q][j], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
            };
            const int nb = (kk + 1) & 1;
            const unsigned char* stg = kk < 3 ? st : nx;
            const int coff = ((((kk < 3 ? 2 * (kk + 1) : 0) + fhalf) ^ fswz) << 4);
            one(0);
            if (FR) { if (q < 2) fb[kk < 3 ? nb : 0][2 * q] = *(const uint4*)(stg + b_lds0 + (2 * q) * 4096 + coff);
                      else fa[kk < 3 ? nb : 0][2 * (q - 2)] = *(const uint4*)(stg + a_lds0 + (2 * (q - 2)) * 4096 + coff); }
            __builtin_amdgcn_sched_barrier(0);
            one(1);
            if (FR) { if (q < 2) fb[kk < 3 ? nb : 0][2 * q + 1] = *(const uint4*)(stg + b_lds0 + (2 * q + 1) * 4096 + coff);
                      else fa[kk < 3 ? nb : 0][2 * (q - 2) + 1] = *(const uint4*)(stg + a_lds0 + (2 * (q - 2) + 1) * 4096 + coff); }
            __builtin_amdgcn_sched_barrier(0);
            one(2);
            if (DMA == 2) {
              if (kk == 3) piece(kt & 1, q);
              else if (kk < 2) piece((kt + 1) & 1, 4 + kk * 6 + (q >> 1) * 3 + (q & 1) * 2);
            }
            __builtin_amdgcn_sched_barrier(0);
            one(3);
            if (DMA == 2 && kk < 2 && (q & 1) == 0) piece((kt + 1) & 1, 4 + kk * 6 + (q >> 1) * 3 + 1);
            __builtin_amdgcn_sched_barrier(0);
            continue;
          }
          mfma4(kk & 1, q);
          __builtin_amdgcn_sched_barrier(0);
          if (kk < 3) read2(st, kk + 1, (kk + 1) & 1, q);
          else read2(nx, 0, 0, q);
          if (DMA == 2) {
            if (kk == 3) piece(kt & 1, q);                                   // tile kt+2, pieces 0..3
            else if (kk < 2) {                                               // tile kt+1, pieces 4..15
              const int base = 4 + kk * 6 + (q >> 1) * 3;
              piece((kt + 1) & 1, base + (q & 1) * 2);
              if ((q & 1) == 0) piece((kt + 1) & 1, base + 1);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (kk == 2) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (DMA) tap = (tap + 1) % 9;
        }
      }
    }
  } else
  for (int kt = 0; kt < KT; ++kt) {
    const unsigned char* st = smem + (kt & 1) * STAGE;
    const unsigned char* nx = smem + ((kt + 1) & 1) * STAGE;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      // fragments of group kk + PF (possibly of the next tile: those are fetched after the barrier below)
      const int fk = kk + PF;
      if (fk < 4) fetch(st, fk, fk % (PF + 1));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        mfma4(kk % (PF + 1), q);
        if (DMA == 2 && kk < 2) {   // spread: tile kt+1 into stage (kt+1)&1 (free since the last barrier), two pieces behind each
          __builtin_amdgcn_sched_barrier(0);      // of the first eight micro-groups; group 2 (512 MFMA cycles) is landing time
          const int slot = kk * 4 + q;            // 0..7
          piece((kt + 1) & 1, slot);
          piece((kt + 1) & 1, 8 + slot);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kk == 2) {
        // end of the step's reads of this stage are the kk = 3 fragments, already in registers when PF >= 1: close the step
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own pieces of tile kt+1 (issued during step kt-1 .. kt) landed
        __syncthreads();
        if (DMA == 1) {
#pragma unroll
          for (int i = 0; i < 16; ++i) piece(kt & 1, i);   // tile kt+2 into the stage just vacated
        }
        if (DMA) tap = (tap + 1) % 9;
        // first fragments of the next tile
        if (PF == 1) fetch(nx, 0, 0);
        else { fetch(nx, 0, 1); }
      }
    }
    if (PF == 2) fetch(nx, 1, 2 % 3);
  }
  const unsigned long long t1 = clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][15];
  if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
  if (t == 1.2345f) out[31] = 1;
}

template <bool FR, int DMA, int PF, bool RS = false, bool FINE = false>
static void run(const char* d, unsigned long long* dout, const char* what) {
  const int iters = 1000;
  auto kern = k4<FR, DMA, PF, RS, FINE>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 131072, 0, d, iters, dout);
    hipDeviceSynchronize();
  }
  unsigned long long h[4];
  hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  printf("4 waves x (128 x 128): %-58s %7.1f cyc / K-step (wave 0), %7.1f (wave 3)   [MFMA floor 2048]\n", what, (double)h[0] / iters, (double)h[3] / iters);
}

int main() {
  char* d; unsigned long long* o;
  hipMalloc(&d, (size_t)256 << 20); hipMemset(d, 1, (size_t)256 << 20);
  hipMalloc(&o, 256);
  run<false, 0, 1>(d, o, "const operands, no DMA");
  run<true, 0, 1>(d, o, "LDS fragments (prefetch 1 group), no DMA");
  run<false, 1, 1>(d, o, "const operands, DMA burst after the barrier");
  run<true, 1, 1>(d, o, "LDS fragments (prefetch 1 group), DMA burst after the barrier");
  run<true, 2, 1>(d, o, "LDS fragments (prefetch 1 group), DMA spread (2 pieces / 4 MFMAs)");
  run<false, 2, 1>(d, o, "const operands, DMA spread (2 pieces / 4 MFMAs)");
  run<true, 0, 1, true>(d, o, "interleaved: LDS reads 2 / 4 MFMAs, no DMA");
  run<true, 2, 1, true>(d, o, "interleaved: LDS reads 2 / 4 MFMAs + DMA 0-2 pieces / 4 MFMAs");
  run<false, 2, 1, true>(d, o, "interleaved: const operands + DMA 0-2 pieces / 4 MFMAs");
  run<true, 0, 1, true, true>(d, o, "fine (one slot per MFMA): LDS reads, no DMA");
  run<true, 2, 1, true, true>(d, o, "fine (one slot per MFMA): LDS reads + DMA");
  run<false, 2, 1, true, true>(d, o, "fine (one slot per MFMA): const operands + DMA");
  return 0;
}
