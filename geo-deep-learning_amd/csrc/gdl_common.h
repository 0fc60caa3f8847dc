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

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
