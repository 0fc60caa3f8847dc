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

// Raise a kernel's dynamic-LDS limit exactly once per process, thread-safely: forward calls come from the Python main
// thread, backward calls from autograd's device thread (INTEGRATION.md "threading").  One flag per call site.
#include <mutex>
#define GDL_SET_MAX_LDS_ONCE(kernel, bytes)                                                                  \
  do {                                                                                                       \
    static std::once_flag once__;                                                                            \
    std::call_once(once__, [&] {                                                                             \
      (void)hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
    });                                                                                                      \
  } while (0)

// ---- bf16 <-> f32 (round-to-nearest-even, like torch) ----
__device__ __forceinline__ float bf16_to_f32(uint16_t h) {
  return __uint_as_float(((uint32_t)h) << 16);
}
typedef __attribute__((ext_vector_type(2))) float gdl_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 gdl_bf16x2_t;
// hardware conversion (v_cvt_pk_bf16_f32 on gfx950): round-to-nearest-even, quiet NaN
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const gdl_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, gdl_bf16x2_t));
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return (uint16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

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

// ---- buffer-descriptor DMA (buffer_load_dwordx4 ... offen lds): 32-bit per-lane offsets, hardware range check.
// Probed on gfx950 (tools/probes/buffer_lds_probe.hip): the LDS destination is M0 + lane*16; a lane is out of
// range when voffset >= num_records OR voffset + soffset >= num_records, and then ZEROS are written to its slot.
typedef __attribute__((ext_vector_type(4))) unsigned srd_t;

__device__ __forceinline__ srd_t make_srd(const void* base, unsigned num_bytes) {
  const unsigned long long p = (unsigned long long)base;
  srd_t r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)p);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32) & 0xffffu);   // stride 0: raw buffer
  r.z = __builtin_amdgcn_readfirstlane(num_bytes);
  r.w = 0x00020000u;                                                     // gfx9-family raw-buffer word 3
  return r;
}

// 64 lanes x 16 bytes: lane l fetches base + voffset[l] + soffset into LDS[lds_addr + 16*l]; lds_addr wave-uniform.
// M0 is only ever used by these DMA helpers in this library (the compiler itself needs it for nothing on gfx9+).
// POL selects the cache policy bits of the load: 0 default, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0 sc1 nt
template <int POL = 0>
__device__ __forceinline__ void dma16_buf(unsigned voffset, srd_t srd, unsigned soffset, unsigned lds_addr) {
#define GDL_DMA_ASM(MODS)                                                                          \
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen" MODS " lds"   \
               :                                                                                   \
               : "v"(voffset), "s"(srd), "s"(__builtin_amdgcn_readfirstlane(soffset)),             \
                 "s"(__builtin_amdgcn_readfirstlane(lds_addr))                                     \
               : "memory", "m0")
  if constexpr (POL == 1) GDL_DMA_ASM(" nt");
  else if constexpr (POL == 2) GDL_DMA_ASM(" sc1");
  else if constexpr (POL == 3) GDL_DMA_ASM(" sc0 sc1");
  else if constexpr (POL == 4) GDL_DMA_ASM(" sc0 sc1 nt");
  else GDL_DMA_ASM("");
#undef GDL_DMA_ASM
}

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
// d/dx of gelu_erf: Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

// Sum of f(i) for i = begin, begin + step, ... < end, added in exactly that order (deterministic, same result as the plain
// loop) but with EIGHT loads in flight: the final-reduction kernels walk a few dozen partials per thread and were bound by one
// memory latency per partial.
template <typename ACC, typename F>
__device__ __forceinline__ ACC ordered_sum8(int begin, int end, int step, F f) {
  ACC s = 0;
  int i = begin;
  for (; i + 7 * step < end; i += 8 * step) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = f(i + u * step);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; i < end; i += step) s += f(i);
  return s;
}
