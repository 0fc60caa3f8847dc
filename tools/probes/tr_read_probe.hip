// Probe: exact lane/element semantics of ds_read_b64_tr_b16 on gfx950.
// LDS holds halfword h at byte 2h with value h.  Each lane supplies its own byte address.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(const int* addr_in, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const unsigned addr = (unsigned)(uintptr_t)lds + addr_in[threadIdx.x];
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[threadIdx.x * 4 + 0] = v.x & 0xffff; out[threadIdx.x * 4 + 1] = v.x >> 16;
  out[threadIdx.x * 4 + 2] = v.y & 0xffff; out[threadIdx.x * 4 + 3] = v.y >> 16;
}
int main() {
  int h_addr[64]; uint16_t h_out[256];
  int *d_addr; uint16_t* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int test = 0; test < 2; ++test) {
    // test 0: lane l -> byte address 8*l (contiguous 8-byte pieces)
    // test 1: lane l -> row (l & 15) of a [64 rows][128 B] tile, column block (l >> 4) * 8 bytes
    for (int l = 0; l < 64; ++l) h_addr[l] = test == 0 ? 8 * l : (l & 15) * 128 + (l >> 4) * 8;
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("test %d\n", test);
    for (int l = 0; l < 64; ++l)
      printf("lane %2d addr_hw %4d -> %4d %4d %4d %4d\n", l, h_addr[l] / 2, h_out[4 * l], h_out[4 * l + 1], h_out[4 * l + 2], h_out[4 * l + 3]);
  }
  return 0;
}
