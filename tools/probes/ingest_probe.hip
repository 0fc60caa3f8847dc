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


// Same loop, LDS-DMA only, but the source of a piece is NOT 1 KiB of contiguous memory: it is 1024/ROWB row segments of
// ROWB bytes, `stride` bytes apart (an NHWC activation tile: rows = pixels, 128 B of K per pixel, stride = C * 2 bytes).
template <int ROWB, bool TO_LDS>
__global__ __launch_bounds__(512) void ks(const char* src, unsigned stride, unsigned nrows, int iters, unsigned long long* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds);
  const unsigned region = stride * nrows;
  unsigned long long a = (unsigned long long)(src + (size_t)blockIdx.x * region);
  srd_t s;
  s.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  s.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffff);
  s.z = region;
  s.w = 0x00020000;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, region, 0x00020000);
  constexpr unsigned LPR = ROWB / 16, RPP = 1024 / ROWB;    // lanes per row segment, row segments per piece
  unsigned voff[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const unsigned row = ((wave * 8 + p) * RPP + lane / LPR) % nrows;
    voff[p] = row * stride + (lane % LPR) * 16;
  }
  u4 cur[8], nxt[8];
  unsigned acc = 0;
  __syncthreads();
  const unsigned long long t0 = clock64(), r0 = wall_clock64();
  auto issue = [&](int it, u4* dst) {
    const unsigned so = __builtin_amdgcn_readfirstlane(((unsigned)it * ROWB) % stride);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      if (TO_LDS) dma16(voff[p], s, so, base + wave * 8192 + p * 1024);
      else dst[p] = ld16(voff[p], rs, so);
    }
  };
  auto consume = [&](u4* r) {
    if (TO_LDS) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else {
#pragma unroll
      for (int p = 0; p < 8; ++p) acc ^= r[p].x ^ r[p].w;
    }
    __syncthreads();
  };
  issue(0, cur);
  for (int it = 1; it <= iters; it += 2) {
    issue(it, nxt);
    consume(cur);
    issue(it + 1, cur);
    consume(nxt);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = clock64(), r1 = wall_clock64();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
  if (acc == 0x12345678u || lds[threadIdx.x] == 0x5a) out[0] += 1;
}

template <int ROWB, bool TO_LDS>
static void run_s(const char* d, unsigned stride, unsigned nrows, int grid, int iters, unsigned long long* dout) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((ks<ROWB, TO_LDS>), dim3(grid), dim3(512), 0, 0, d, stride, nrows, iters, dout);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((ks<ROWB, TO_LDS>), dim3(grid), dim3(512), 0, 0, d, stride, nrows, iters, dout);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * grid);
  hipMemcpy(h.data(), dout, 16 * grid, hipMemcpyDeviceToHost);
  double cyc = 0, real = 0;
  for (int b = 0; b < grid; ++b) { cyc += h[2 * b]; real += h[2 * b + 1]; }
  cyc /= grid; real /= grid;
  const double bytes = 65536.0 * iters;
  printf("%s row segments of %4d B, stride %5u B, %4u rows/block, grid %4d: %7.1f cyc / 64 KiB = %5.1f B/clk/CU (%4.1f cyc per segment), "
         "clock %4.0f MHz; chip %6.2f TB/s\n", TO_LDS ? "LDS-DMA  " : "VGPR only", ROWB, stride, nrows, grid, cyc / iters, bytes / cyc,
         cyc / iters / (65536.0 / ROWB), cyc / (real * 10.0) * 1000.0, bytes * grid / (ms * 1e-3) / 1e12);
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
  // strided row segments, L2-resident footprint (256 blocks x 64 rows x stride <= 24 MiB)
  for (int grid : {1, 256}) {
    for (unsigned stride : {1536u, 2048u, 512u}) {
      run_s<64, true>(d, stride, 64, grid, iters, o);
      run_s<128, true>(d, stride, 64, grid, iters, o);
      run_s<256, true>(d, stride, 64, grid, iters, o);
      run_s<512, true>(d, stride, 64, grid, iters, o);
      run_s<128, false>(d, stride, 64, grid, iters, o);
    }
    run_s<128, true>(d, 1536u, 2048, grid, iters, o);    // 3 MiB per block: streamed
  }
  return 0;
}
