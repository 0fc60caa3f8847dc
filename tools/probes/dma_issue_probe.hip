// Probe: what does it cost a WAVE to issue LDS-DMA pieces (buffer_load_dwordx4 ... lds, 1 KiB each), as a function of
// how many waves of the CU issue at the same time and of how many pieces each issues back to back?
// One workgroup of 8 waves on a CU (waves w and w+4 share a SIMD).  `wmask` selects the issuing waves; every iteration
// each issuing wave issues P pieces back to back (L2-hot source), then waits for them (vmcnt(0)), then the workgroup
// meets at a barrier -- the structure of a two-stage GEMM K loop without the MFMAs.
// Reported: shader cycles of the issue burst alone (first instruction -> the wave is past the last one), per piece,
// and cycles per iteration.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/dma_issue_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned srd_t;

__device__ __forceinline__ void dma16(unsigned voff, srd_t srd, unsigned soff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(srd), "s"(soff), "s"(lds_addr) : "memory", "m0");
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// MF: MFMAs (32x32x16 bf16, 8 independent accumulators) every wave issues per iteration after its DMA burst / reads;
// MF_FIRST: waves 4-7 (the SIMD partners of the issuing waves 0-3) do their MFMAs at the START of the iteration instead
template <int P, bool LOOKAHEAD, int READS = 0, int MF = 0, int RDPAT = 0>
__global__ __launch_bounds__(512) void k(const char* src, unsigned wmask, int iters, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds);
  const unsigned region = 131072;
  unsigned long long a = (unsigned long long)(src + (size_t)blockIdx.x * region);
  srd_t s;
  s.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  s.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffff);
  s.z = region;
  s.w = 0x00020000;
  const bool issuer = (wmask >> wave) & 1u;
  const unsigned voff = (lane >> 3) * 1536 + (lane & 7) * 16;     // 8 rows of 128 B, 1536 B apart (an NHWC tile row)
  unsigned long long burst = 0;
  // fragment reads of a GEMM wave: row = lane & 31 (+32 per read), 16-byte slot swizzled like conv_gemm.hip (conflict-free)
  const unsigned frow = lane & 31, fhalf = lane >> 5;
  const unsigned rd0 = frow * 128 + ((fhalf ^ ((frow >> 1) & 7)) << 4) + (wave & 1) * 32768;
  uint4 sink = make_uint4(0, 0, 0, 0);
  f32x16_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8_t fa, fb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(float)(lane + i); fb[i] = (__bf16)(float)(lane ^ i); }
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (issuer) {
      const unsigned so = __builtin_amdgcn_readfirstlane(((unsigned)it * 128u) % 1536u + wave * 12288u);
      const unsigned long long b0 = clock64();
#pragma unroll
      for (int p = 0; p < P; ++p) dma16(voff, s, so, base + ((it & 1) * 65536 + wave * 8192 + (p & 7) * 1024));
      burst += clock64() - b0;
    }
    if (READS) {
      const unsigned char* st = lds + ((it & 1) ^ 1) * 65536;     // the stage that is NOT being written
      uint4 v[READS ? READS : 1];
#pragma unroll
      for (int r = 0; r < READS; ++r)      // all reads in flight together (throughput, not latency)
        v[r] = RDPAT == 1 ? *(const uint4*)(st + lane * 16 + r * 1024 + (wave & 1) * 32768)     // linear: 1 KiB contiguous per read
                          : RDPAT == 2 ? *(const uint4*)(st + (lane & 31) * 128 + (lane >> 5) * 16 + (r & 7) * 4096 + ((r >> 3) & 3) * 32)   // rows of 128 B, no swizzle (8-way conflicts)
                                       : *(const uint4*)(st + ((rd0 + (r & 7) * 4096) ^ (((r >> 3) & 3) << 5)));
#pragma unroll
      for (int r = 0; r < READS; ++r) { sink.x ^= v[r].x; sink.y ^= v[r].y; sink.z ^= v[r].z; sink.w ^= v[r].w; }
    }
    if (MF) {
#pragma unroll
      for (int m = 0; m < MF; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[m & 7], 0, 0, 0);
    }
    if (issuer) {
      if (LOOKAHEAD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");   // the previous burst has landed
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = clock64();
  if (lane == 0) { out[2 * wave] = t1 - t0; out[2 * wave + 1] = burst; }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][15];
  if ((sink.x ^ sink.y ^ sink.z ^ sink.w) == 0x12345u || t == 1.2345f) out[15] = 1;
}

template <int P, bool LOOKAHEAD, int READS = 0, int MF = 0, int RDPAT = 0>
static void run(const char* d, unsigned wmask, unsigned long long* dout, int grid) {
  const int iters = 2000;
  auto kern = k<P, LOOKAHEAD, READS, MF, RDPAT>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 131072, 0, d, wmask, iters, dout);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 131072, 0, d, wmask, iters, dout);
  hipDeviceSynchronize();
  unsigned long long h[16];
  hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  int nw = 0, first = -1;
  for (int w = 0; w < 8; ++w) if ((wmask >> w) & 1u) { ++nw; if (first < 0) first = w; }
  const double per_iter = (double)h[0] / iters, per_piece = first < 0 ? 0.0 : (double)h[2 * first + 1] / iters / P;
  printf("[read pattern %d] %2d MFMA + %2d ds_read_b128 per wave, waves 0x%02x (%d issuing) x %2d pieces, %s, grid %3d: burst %6.1f cyc/piece (wave %d), iteration %7.1f cyc = %5.1f B/clk/CU\n",
         RDPAT, MF, READS, wmask, nw, P, LOOKAHEAD ? "one burst of lookahead" : "wait for own burst    ", grid, per_piece, first, per_iter, nw * P * 1024.0 / per_iter);
}

int main() {
  char* d; unsigned long long* o;
  hipMalloc(&d, (size_t)256 * 131072); hipMemset(d, 1, (size_t)256 * 131072);
  hipMalloc(&o, 256);
  // ds_read_b128 throughput by address pattern (0 = the GEMM's swizzled rows, 1 = linear, 2 = unswizzled rows)
  run<16, false, 24, 0, 0>(d, 0x00u, o, 256);
  run<16, false, 24, 0, 1>(d, 0x00u, o, 256);
  run<16, false, 24, 0, 2>(d, 0x00u, o, 256);
  run<16, false, 48, 0, 0>(d, 0x00u, o, 256);
  run<16, false, 48, 0, 1>(d, 0x00u, o, 256);
  run<16, false, 24, 0, 1>(d, 0x0fu, o, 256);
  return 0;
}
