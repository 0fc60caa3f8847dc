// Probe: semantics of `buffer_load_dwordx4 ... offen lds` on gfx950 that the conv kernels rely on.
//   (1) an out-of-range lane (voffset >= num_records) WRITES ZEROS to its LDS slot (does not skip it);
//   (2) whether soffset takes part in the range check;
//   (3) LDS destination = M0 + lane*16 (lane-linear), independent of voffset.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/buffer_lds_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned srd_t;
__device__ __forceinline__ void dma16_buf(unsigned voff, srd_t srd, unsigned soff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(srd), "s"(soff), "s"(lds_addr) : "memory", "m0");
}
__global__ void k(const unsigned* p, unsigned* o, unsigned nbytes) {
  __shared__ unsigned lds[3 * 256];
  for (int i = threadIdx.x; i < 3 * 256; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds);
  srd_t s;
  unsigned long long a = (unsigned long long)p;
  s.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  s.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffff);
  s.z = nbytes;
  s.w = 0x00020000;
  const unsigned lane = threadIdx.x;
  // test A: even lanes in range (reversed order), odd lanes voffset = 0x80000000
  dma16_buf((lane & 1) ? 0x80000000u : (63 - lane) * 16, s, 0, base);
  // test B: voffset in range, soffset pushes the last lanes past num_records (nbytes = 1024): lane*16 + 512
  dma16_buf(lane * 16, s, 512, base + 1024);
  // test C: "negative" voffset, positive soffset: (lane*16 - 256) + 256
  dma16_buf(lane * 16 - 256, s, 256, base + 2048);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * 256; i += 64) o[i] = lds[i];
}
int main() {
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;
  unsigned *d, *o;
  hipMalloc(&d, 4096); hipMalloc(&o, 3 * 1024);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 1024u);
  std::vector<unsigned> r(3 * 256);
  hipMemcpy(r.data(), o, 3 * 1024, hipMemcpyDeviceToHost);
  const char* names[3] = {"A oob-lanes", "B soffset-past-end", "C negative-voffset"};
  for (int t = 0; t < 3; ++t) {
    printf("%s (first dword of each lane's 16-byte slot):\n", names[t]);
    for (int l = 0; l < 64; ++l) printf("%08x%s", r[t * 256 + l * 4], (l % 8 == 7) ? "\n" : " ");
  }
  return 0;
}
