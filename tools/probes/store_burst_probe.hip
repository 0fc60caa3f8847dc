// Probe: is the epilogue of the K = 768 GEMM tiles (256 x 256 outputs, 128 KiB of bf16 / 256 KiB of f32 per tile, measured at
// 6-7 B/clk/CU) bound by what ONE CU can push into the memory system, or by the chip-wide write bandwidth when all 256 CUs
// store at the same time?  If the latter, de-synchronising the CUs (so that at any moment only a fraction of them is in its
// store phase) would shorten every tile.
// Model of one GEMM block: 8 waves; `tiles` iterations of [ compute phase: NMF dependent-free MFMAs per wave ] + [ store phase:
// the wave's 128 rows x 128 B (bf16) as 16-byte pieces per lane, 8 rows per instruction, rows `row_stride` bytes apart ].
// grid = nblk blocks (one per CU at 128 KiB of LDS); `stagger` = block b sleeps (b % 8) * stagger cycles before it starts.
// Reported: average cycles of a store phase and of a whole tile, wall time of the launch.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/store_burst_probe.hip -o tools/probes/bin/store_burst_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__global__ __launch_bounds__(512) void k(unsigned char* out, long row_stride, int tiles, int nmf, int stagger, int bytes_per_row,
                                         unsigned long long* stats) {
  extern __shared__ unsigned char lds[];                       // 128 KiB: one block per CU, like the GEMM
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16_t acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8_t fa, fb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(float)(lane + i); fb[i] = (__bf16)(float)(lane ^ i); }
  if (stagger > 0) {
    const long long until = clock64() + (long long)(blockIdx.x % 8) * stagger;
    while (clock64() < until) __builtin_amdgcn_s_sleep(8);
  }
  __syncthreads();
  unsigned long long t_store = 0;
  const unsigned long long t0 = clock64();
  for (int t = 0; t < tiles; ++t) {
    for (int i = 0; i < nmf; i += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j], 0, 0, 0);
    }
    __syncthreads();
    const unsigned long long s0 = clock64();
    // this block's tile = 256 rows x 256 outputs; wave (wm, wn) owns rows wm * 128 .. +127, bytes wn * bytes_per_row .. of
    // every row; 4 passes of 32 rows, 16 B per lane: 128-B row pieces (bf16) -> 8 rows per instruction, 256-B (f32) -> 4
    const int lanes_per_row = bytes_per_row / 16, rows_per_instr = 64 / lanes_per_row;
    const int wm = wave >> 2, wn = wave & 3;
    const long tile_row0 = ((long)blockIdx.x * tiles + t) * 256 + wm * 128;
    const uint4 v = make_uint4(__float_as_uint(acc[0][0]), __float_as_uint(acc[1][1]), __float_as_uint(acc[2][2]), __float_as_uint(acc[3][3]));
    for (int pass = 0; pass < 4; ++pass) {
      for (int r = 0; r < 32; r += rows_per_instr) {
        const long row = tile_row0 + pass * 32 + r + lane / lanes_per_row;
        *(uint4*)(out + row * row_stride + wn * bytes_per_row + (lane % lanes_per_row) * 16) = v;
      }
    }
    t_store += clock64() - s0;
  }
  const unsigned long long t1 = clock64();
  if (threadIdx.x == 0) {
    stats[2 * blockIdx.x] = t_store;
    stats[2 * blockIdx.x + 1] = t1 - t0;
  }
  if (acc[0][0] == 123.456f) out[0] = 1;
}

int main(int argc, char** argv) {
  const int tiles = 6;
  unsigned char* out;
  const size_t out_bytes = (size_t)4 << 30;
  hipMalloc(&out, out_bytes);
  unsigned long long* stats;
  hipMalloc(&stats, 2 * 1024 * sizeof(unsigned long long));
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("%6s %5s %8s %9s | %12s %12s %10s\n", "blocks", "nmf", "stagger", "row bytes", "store cyc/tile", "tile cyc", "launch us");
  for (int bytes_per_row : {128, 256}) {                      // per wave and pass-column: 64 bf16 / 64 f32 channels
    const long row_stride = bytes_per_row == 128 ? 4608 : 3072;   // qkv [M][2304] bf16; proj [M][768] f32
    for (int nmf : {0, 192, 384}) {                           // 384 MFMAs per wave = a 12-K-step tile (32 per K-step)
      for (int nblk : {1, 32, 256}) {
        for (int stagger : {0, 2000, 6000}) {
          if (stagger && nblk < 256) continue;
          if ((size_t)nblk * tiles * 256 * row_stride > out_bytes) continue;
          for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(nblk), dim3(512), 131072, 0, out, row_stride, tiles, nmf, stagger, bytes_per_row, stats);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
          }
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          std::vector<unsigned long long> h(2 * nblk);
          hipMemcpy(h.data(), stats, h.size() * 8, hipMemcpyDeviceToHost);
          double s = 0, tt = 0;
          for (int b = 0; b < nblk; ++b) { s += h[2 * b]; tt += h[2 * b + 1]; }
          printf("%6d %5d %8d %9d | %12.0f %12.0f %10.1f\n", nblk, nmf, stagger, bytes_per_row, s / nblk / tiles, tt / nblk / tiles, ms * 1e3);
        }
      }
    }
  }
  return 0;
}
