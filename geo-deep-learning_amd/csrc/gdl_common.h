// Shared device/host helpers for libgdlhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gdlhip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

void gdl_set_error(const char* fmt, ...);

#define GDL_CHECK_ARG(cond, ...)              \
  do {                                        \
    if (!(cond)) {                            \
      gdl_set_error(__VA_ARGS__);             \
      return GDL_ERR_INVALID;                 \
    }                                         \
  } while (0)

#define GDL_CHECK_LAUNCH(what)                                                   \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      gdl_set_error("%s: launch failed: %s", what, hipGetErrorString(e__));      \
      return GDL_ERR_LAUNCH;                                                     \
    }                                                                            \
  } while (0)

static inline size_t gdl_elem_size(int dtype) { return dtype == GDL_BF16 ? 2 : 4; }

// ---- bf16 <-> f32 (round-to-nearest-even, like torch) ----
__device__ __forceinline__ float bf16_to_f32(uint16_t h) {
  return __uint_as_float(((uint32_t)h) << 16);
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

template <typename T> struct ElemIO;
template <> struct ElemIO<float> {
  static __device__ __forceinline__ float load(const void* p, int64_t i) { return ((const float*)p)[i]; }
  static __device__ __forceinline__ void store(void* p, int64_t i, float v) { ((float*)p)[i] = v; }
};
struct bf16_tag {};
template <> struct ElemIO<bf16_tag> {
  static __device__ __forceinline__ float load(const void* p, int64_t i) {
    return bf16_to_f32(((const uint16_t*)p)[i]);
  }
  static __device__ __forceinline__ void store(void* p, int64_t i, float v) {
    ((uint16_t*)p)[i] = f32_to_bf16(v);
  }
};

__device__ __forceinline__ float load_as_f32(const void* p, int64_t i, int dtype) {
  return dtype == GDL_BF16 ? bf16_to_f32(((const uint16_t*)p)[i]) : ((const float*)p)[i];
}
__device__ __forceinline__ void store_from_f32(void* p, int64_t i, float v, int dtype) {
  if (dtype == GDL_BF16) ((uint16_t*)p)[i] = f32_to_bf16(v);
  else ((float*)p)[i] = v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// 16-byte global -> LDS DMA (global_load_lds_dwordx4) issued through inline asm.
// Why asm and not __builtin_amdgcn_global_load_lds: hipcc (ROCm 7.2) treats the builtin as a
// pending LDS write that may alias ANY later ds_read and emits `s_waitcnt vmcnt(0)` in front of
// the first fragment read of the MFMA phase -- i.e. it waits for the NEXT tile's DMA before
// computing the current one and the copy/compute overlap is gone (seen in the .s: vmcnt(0) ->
// ds_read_b128).  The asm form is invisible to that bookkeeping; the kernels wait by hand
// (`s_waitcnt vmcnt(0)` + barrier) before a stage is read.  M0 (LDS base of the wave's 1 KiB
// piece, wave-uniform) is saved/restored inside the same statement (compiler-reserved register);
// `s_nop 0` covers the s_mov m0 -> LDS-DMA hazard.  Lane l lands at lds_base + 16*l.
__device__ __forceinline__ void dma16_to_lds(const void* gsrc, void* lds_wave_base) {
  const unsigned lds_addr = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_wave_base);
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_addr)
      : "memory");
}

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
// d/dx of gelu_erf: Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}
