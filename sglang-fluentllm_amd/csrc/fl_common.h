// Shared host/device helpers for libfluent_mi355 (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include "../../include/fluent_mi355.h"

void fl_set_error(const char* fmt, ...);

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
// max of three in one instruction (hipcc emits two v_max_f32 for nested fmaxf)
__device__ __forceinline__ float fl_max3(const float a, const float b, const float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// two f32 -> packed bf16 pair (lo = a), hardware RNE (v_cvt_pk_bf16_f32; same rounding as fl_f32_to_bf16 / torch)
typedef __bf16 fl_bf16x2 __attribute__((ext_vector_type(2)));
typedef float fl_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t fl_pack_bf16(float a, float b) {
  union { fl_bf16x2 h; uint32_t u; } x;
  x.h = __builtin_convertvector(fl_f32x2{a, b}, fl_bf16x2);
  return x.u;
}
// f32 -> OCP e4m3fn byte, RNE, |x| must be <= 448 (v_cvt_pk_fp8_f32 does not saturate: >464 -> NaN)
__device__ __forceinline__ uint32_t fl_cvt_pk_fp8(float a, float b) {
  return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
}
__device__ __forceinline__ float fl_fp8_to_f32(uint8_t v) {
  return __builtin_amdgcn_cvt_f32_fp8((int)v, 0);
}
// Eight values divided by ONE divisor (a row's / group's quantisation scale) and rounded to fp8 e4m3.  hipcc expands every
// `x / d` into the full correctly-rounded sequence (2 v_div_scale, v_rcp, 5 fma, v_div_fmas, v_div_fixup = 11 VALU ops, the
// reciprocal included each time): the quantise kernels were VALU-bound on it (K4: 110 of 205 VALU ops per row).  The fast
// branch is the SAME arithmetic (Newton-refined reciprocal, quotient, two residual corrections) with the divisor-only part
// done once and without the exponent rescue of v_div_scale / v_div_fixup, taken only where that rescue is the identity:
// d in [2^-60, 2^60] (uniform over the lanes that share d) and |x| <= 2^12 d.  Quotients so small that their fp32 value
// would go denormal are far below half the smallest fp8 subnormal (2^-10) on both paths and round to the same signed
// zero.  Bytes are bit-identical to cvt(x / d) (tests/: quantised bytes bit-exact vs the torch statement).
// lo/hi: optional clamp of the quotient before the conversion (the 1x128 statement clamps to +-448).
template <bool kClamp>
__device__ __forceinline__ uint2 fl_div8_to_fp8(const float (&x)[8], const float d) {
  float q[8];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(x[i]));
  if (d >= 0x1p-60f && d <= 0x1p60f && amax <= d * 0x1p12f) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float r = fmaf(fmaf(-d, r0, 1.f), r0, r0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float t = x[i] * r;
      t = fmaf(fmaf(-d, t, x[i]), r, t);
      t = fmaf(fmaf(-d, t, x[i]), r, t);
      q[i] = x[i] == 0.f ? x[i] : t;   // (-0 / d = -0: the corrections would turn it into +0)
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = x[i] / d;
  }
  if (kClamp) {
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = fminf(fmaxf(q[i], -FL_FP8_MAX), FL_FP8_MAX);
  }
  return make_uint2(fl_cvt_pk_fp8(q[0], q[1]) | (fl_cvt_pk_fp8(q[2], q[3]) << 16),
                    fl_cvt_pk_fp8(q[4], q[5]) | (fl_cvt_pk_fp8(q[6], q[7]) << 16));
}

// max over the 64 lanes, every lane gets it.  In registers: v_permlane32_swap / v_permlane16_swap across the four rows of 16, DPP row rotations and
// quad permutations inside a row (max is order-free: the same bits as any other reduction order; __shfl_xor is six dependent ds_bpermute
// round trips through the LDS crossbar — the quantise kernels' whole latency chain at decode sizes)
__device__ __forceinline__ float fl_wave_max(float v) {
  {
    const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1]));
  }
  {
    const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1]));
  }
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false)));   // row_ror:8
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false)));   // row_ror:4
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4e, 0xf, 0xf, false)));    // quad_perm:[2,3,0,1]
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xb1, 0xf, 0xf, false)));    // quad_perm:[1,0,3,2]
  return v;
}

// HBM -> LDS DMA (global_load_lds) as inline asm: lane i of the wave lands at lds_dst + 16 i (4 i); lds_dst must be
// wave-uniform.  NOT through __builtin_amdgcn_global_load_lds: hipcc's waitcnt pass books the builtin as a FLAT access
// that touches LDS ("pending flat") and, until a vmcnt(0) OF ITS OWN retires it, forces every later
// s_waitcnt lgkmcnt(N) to N = 0 — in a pipelined loop whose vmcnt waits are explicit, every LDS operand read then also
// waits for the reads issued after it.  The asm hides the DMA from the pass; completion is certified by the kernels'
// explicit s_waitcnt vmcnt + barrier.
__device__ __forceinline__ int fl_lds_addr(const void* lds_dst) {
  return __builtin_amdgcn_readfirstlane((int)(uintptr_t)lds_dst);   // low half of a generic LDS pointer = LDS offset
}
__device__ __forceinline__ void fl_dma16(const void* gsrc, const void* lds_dst) {   // per-lane 64-bit source
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(fl_lds_addr(lds_dst)), "v"(gsrc)
               : "memory", "m0");
}
// per-lane 64-bit source, cache policy chosen at run time (wave-uniform `nt`): non-temporal for a stream ONE workgroup reads once
__device__ __forceinline__ void fl_dma16_pol(const void* gsrc, const void* lds_dst, const bool nt) {
  if (nt)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(fl_lds_addr(lds_dst)), "v"(gsrc) : "memory", "m0");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(fl_lds_addr(lds_dst)), "v"(gsrc) : "memory", "m0");
}
__device__ __forceinline__ void fl_dma4(const void* gsrc, const void* lds_dst) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(fl_lds_addr(lds_dst)), "v"(gsrc)
               : "memory", "m0");
}
// same, with the builtin's argument list (drop-in for __builtin_amdgcn_global_load_lds(g, l, size, 0, 0))
__device__ __forceinline__ void fl_dma_lds(const __attribute__((address_space(1))) void* gsrc,
                                           __attribute__((address_space(3))) void* lds_dst, const int size, int, int) {
  const int la = __builtin_amdgcn_readfirstlane((int)(uintptr_t)lds_dst);
  if (size == 16)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(la), "v"(gsrc) : "memory", "m0");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(la), "v"(gsrc) : "memory", "m0");
}
// saddr forms: wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset
// NT: non-temporal cache policy for a stream that exactly ONE workgroup reads once (MI355X guide, "nt-weights": issued -> landed -18 %;
// a second reader on the same XCD would lose its L2 hit, so only where there is none)
template <bool NT>
__device__ __forceinline__ void fl_dma16_s_nt(const void* sbase, const unsigned voff, const void* lds_dst) {
  if constexpr (NT)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(fl_lds_addr(lds_dst)), "v"(voff), "s"(sbase)
                 : "memory", "m0");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(fl_lds_addr(lds_dst)), "v"(voff), "s"(sbase)
                 : "memory", "m0");
}
__device__ __forceinline__ void fl_dma16_s(const void* sbase, const unsigned voff, const void* lds_dst) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(fl_lds_addr(lds_dst)), "v"(voff), "s"(sbase)
               : "memory", "m0");
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: set once per (kernel, device) — a process that drives several
// GPUs must not rely on a process-wide static.  `done` is the caller's per-kernel flag array (zero-initialised static).
#include <atomic>
inline hipError_t fl_set_max_dynamic_lds(const void* fn, int bytes, std::atomic<unsigned char> (&done)[64]) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (done[dev].load(std::memory_order_acquire)) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done[dev].store(1, std::memory_order_release);
  return e;
}
