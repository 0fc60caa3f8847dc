// Probe: how fast can ONE CU pull operand tiles into LDS on gfx950, by path?
//   mode 0  buffer_load_dwordx4 ... lds      (LDS-DMA, what the GEMM kernels use)
//   mode 1  buffer_load_dwordx4 -> VGPR -> ds_write_b128
//   mode 2  buffer_load_dwordx4 -> VGPR only (no LDS write): the vector-memory return path alone
//   mode 3  half the pieces by DMA, half through VGPRs (do the two paths add up or share one limit?)
// One "iteration" = 64 KiB per block (8 waves x 8 pieces of 1 KiB) = the staging of one 256x256x64 bf16 K-step; one
// barrier per iteration; the next iteration's loads are in flight while the previous one is awaited.
// Source: per-block region of `region` bytes walked cyclically (64 KiB = hot in L1/L2, 4 MiB = streamed).
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/ingest_probe.hip -o /tmp/ingest && /tmp/ingest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned srd_t;
typedef __attribute__((ext_vector_type(4))) unsigned u4;

__device__ __forceinline__ void dma16(unsigned voff, srd_t srd, unsigned soff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(srd), "s"(soff), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ u4 ld16(unsigned voff, __amdgpu_buffer_rsrc_t srd, unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(srd, voff, soff, 0);
}

template <int MODE>
__global__ __launch_bounds__(512) void k(const char* src, unsigned region, int iters, unsigned long long* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds);
  unsigned long long a = (unsigned long long)(src + (size_t)blockIdx.x * region);
  srd_t s;
  s.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  s.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffff);
  s.z = region;
  s.w = 0x00020000;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, region, 0x00020000);
  const unsigned voff = lane * 16;
  u4 cur[8], nxt[8];
  unsigned acc = 0;
  const unsigned mask = region - 1;
  __syncthreads();
  const unsigned long long t0 = clock64(), r0 = wall_clock64();
  auto issue = [&](int it, u4* dst) {
    const unsigned so = __builtin_amdgcn_readfirstlane(((unsigned)it * 65536u + wave * 8192u) & mask);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const bool by_dma = MODE == 0 || (MODE == 3 && p < 4);
      if (by_dma) dma16(voff, s, so + p * 1024, base + wave * 8192 + p * 1024);
      else dst[p] = ld16(voff, rs, so + p * 1024);
    }
  };
  auto consume = [&](u4* r) {
    if (MODE == 0 || MODE == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const bool by_dma = MODE == 0 || (MODE == 3 && p < 4);
      if (by_dma) continue;
      if (MODE == 2) acc ^= r[p].x ^ r[p].w;
      else *(u4*)(lds + wave * 8192 + p * 1024 + lane * 16) = r[p];
    }
    __syncthreads();
  };
  issue(0, cur);
  for (int it = 1; it <= iters; it += 2) {       // two stages in flight, no register copies between them
    issue(it, nxt);
    consume(cur);
    issue(it + 1, cur);
    consume(nxt);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = clock64(), r1 = wall_clock64();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
  if (acc == 0x12345678u || lds[threadIdx.x] == 0x5a) out[0] += 1;   // keep results alive
}

template <int MODE>
static void run(const char* name, const char* d, unsigned region, int grid, int iters, unsigned long long* dout) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, d, region, iters, dout);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, d, region, iters, dout);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * grid);
  hipMemcpy(h.data(), dout, 16 * grid, hipMemcpyDeviceToHost);
  double cyc = 0, real = 0;
  for (int b = 0; b < grid; ++b) { cyc += h[2 * b]; real += h[2 * b + 1]; }
  cyc /= grid; real /= grid;
  const double bytes = 65536.0 * iters;
  printf("%-34s region %7u KiB grid %4d: %7.1f cyc / 64 KiB iteration = %5.1f B/clk/CU, %6.1f ns/iter (block), clock %4.0f MHz; "
         "chip %6.2f TB/s (event %.3f ms)\n", name, region >> 10, grid, cyc / iters, bytes / cyc, real * 10.0 / iters, cyc / (real * 10.0) * 1000.0,
         bytes * grid / (ms * 1e-3) / 1e12, ms);
}

int main() {
  const size_t total = (size_t)256 * (4u << 20);
  char* d; unsigned long long* o;
  hipMalloc(&d, total); hipMemset(d, 1, total);
  hipMalloc(&o, 16 * 512);
  const int iters = 2000;
  for (unsigned region : {65536u, 4u << 20}) {
    for (int grid : {1, 256}) {
      run<0>("LDS-DMA", d, region, grid, iters, o);
      run<1>("VGPR + ds_write_b128", d, region, grid, iters, o);
      run<2>("VGPR only", d, region, grid, iters, o);
      run<3>("half DMA, half VGPR + ds_write", d, region, grid, iters, o);
    }
  }
  return 0;
}
