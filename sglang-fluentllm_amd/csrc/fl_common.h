// Shared host/device helpers for libfluent_mi355 (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include "../../include/fluent_mi355.h"

void fl_set_error(const char* fmt, ...);
bool fl_mla_use_x();   // 128-row MLA decode workgroups for s_q*H > 64 + matching part count (FLUENT_MLA_X=0 disables)

#define FL_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      fl_set_error(__VA_ARGS__);           \
      return FL_ERR_INVALID;               \
    }                                      \
  } while (0)

#define FL_CHECK_LAUNCH(what)                                                        \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      fl_set_error("%s: launch failed: %s", what, hipGetErrorString(e__));           \
      return FL_ERR_LAUNCH;                                                          \
    }                                                                                \
  } while (0)

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

#define FL_FP8_MAX 448.0f

__device__ __forceinline__ float fl_bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even f32 -> bf16 (matches torch .to(bfloat16); NaN preserved)
__device__ __forceinline__ uint16_t fl_f32_to_bf16(float f) {
  const uint32_t u = __float_as_uint(f);
  const uint32_t r = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  const uint32_t n = (u >> 16) | 0x40u;
  return (uint16_t)(((u & 0x7fffffffu) > 0x7f800000u) ? n : r);   // branch-free select
}
// f32 -> OCP e4m3fn byte, RNE, |x| must be <= 448 (v_cvt_pk_fp8_f32 does not saturate: >464 -> NaN)
__device__ __forceinline__ uint32_t fl_cvt_pk_fp8(float a, float b) {
  return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
}
__device__ __forceinline__ float fl_fp8_to_f32(uint8_t v) {
  return __builtin_amdgcn_cvt_f32_fp8((int)v, 0);
}
__device__ __forceinline__ float fl_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
