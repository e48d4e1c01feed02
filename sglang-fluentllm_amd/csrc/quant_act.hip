// Q1/Q2 — 1x128 per-token-group FP8 quantisation; A1 — SiLU·mul (optionally fused with the quantisation).
//
//   fl_quant_1x128  == flashinfer.quantization.quant_1x128 (moe/executors/fp8_eps_executor.py:53-55,75-77) and
//                      flashinfer.sgl_per_token_group_quant_fp8 (dense/gemms/fp8/fp8_kernel.py:430-462); values per
//                      python/sglang/test/test_block_fp8.py:15-40: s = max(amax, eps)/448 (fp32), q = clamp(x/s) -> e4m3
//   fl_silu_and_mul == eps.executor.silu (fp8_eps_executor.py:62) / flashinfer.silu_and_mul; math per
//                      python/sglang/srt/layers/activation.py:58-60 (silu in the bf16 domain: F.silu rounds to bf16,
//                      then a bf16 multiply), fused variant == flashinfer.activation.silu_and_mul_fuse_block_quant
//                      (activation.py:73, deep_ep_executor.py:676)
// HBM-bound elementwise work: 16 lanes per 128-element group (8 bf16 = 16 B per lane), 4 groups per wave, amax by
// 4 DPP/shuffle steps inside the 16-lane group.  No LDS.
#include "fl_common.h"

namespace {

__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

__device__ __forceinline__ void unpack8(const uint4 raw, float (&v)[8]) {
  v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xffff0000u);
  v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xffff0000u);
  v[4] = __uint_as_float(raw.z << 16); v[5] = __uint_as_float(raw.z & 0xffff0000u);
  v[6] = __uint_as_float(raw.w << 16); v[7] = __uint_as_float(raw.w & 0xffff0000u);
}

__device__ __forceinline__ uint2 quant8(const float (&v)[8], float s) {
  return fl_div8_to_fp8<true>(v, s);   // v / s (IEEE), clamp to +-448, e4m3 (fl_common.h)
}

// one 16-lane group per (row, k-group); grid-stride over groups
__global__ __launch_bounds__(256) void quant_1x128_kernel(const uint16_t* __restrict__ x, long long M, int K, float eps,
                                                          uint8_t* __restrict__ xq, float* __restrict__ xs,
                                                          long long s_stride_m, long long s_stride_k) {
  const int kg = K / 128;
  const long long total = M * kg;
  const int sub = threadIdx.x & 15;
  // The in-tree statement clamps the bf16 amax tensor (test_block_fp8.py:33: x_.abs().max().clamp(min=eps) BEFORE
  // .to(float32)), i.e. with eps rounded to the input dtype; only rows with amax < 1e-10 can tell the difference.
  eps = fl_bf16_to_f32(fl_f32_to_bf16(eps));
  for (long long g = (long long)blockIdx.x * 16 + (threadIdx.x >> 4); g < total; g += (long long)gridDim.x * 16) {
    // (32-bit division when it fits: the 64-bit one is ~80 VALU ops in a kernel that is one load -> reduce -> store chain)
    const long long m = total < (1ll << 31) ? (long long)((unsigned)g / (unsigned)kg) : g / kg;
    const int kb = (int)(g - m * kg);
    const uint16_t* p = x + m * K + kb * 128 + sub * 8;
    float v[8];
    unpack8(*reinterpret_cast<const uint4*>(p), v);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(v[i]));
    amax = group16_max(amax);
    const float s = fmaxf(amax, eps) / FL_FP8_MAX;
    *reinterpret_cast<uint2*>(xq + m * K + kb * 128 + sub * 8) = quant8(v, s);
    if (sub == 0) xs[m * s_stride_m + kb * s_stride_k] = s;
  }
}

__device__ __forceinline__ float silu_mul_bf16(float g, float u) {
  // activation.py:58-60 on bf16 tensors: F.silu(g) (computed in fp32 inside torch, rounded to bf16), then a bf16 multiply
  const float sg = fl_bf16_to_f32(fl_f32_to_bf16(g / (1.f + __expf(-g))));
  return fl_bf16_to_f32(fl_f32_to_bf16(sg * u));
}

// masked_m != nullptr: rows are [G groups][rows_per_group]; only the first masked_m[group] rows of a group are touched
// (silu_and_mul_masked_post_quant_fwd, deep_ep_executor.py:106-170), scales at qs + group*s_stride_g + row*s_stride_m + kb*s_stride_k
template <bool kQuant>
__global__ __launch_bounds__(256) void silu_mul_kernel(const uint16_t* __restrict__ x, long long M, int I,
                                                       uint16_t* __restrict__ out, uint8_t* __restrict__ q,
                                                       float* __restrict__ qs, long long s_stride_m, long long s_stride_k,
                                                       const int32_t* __restrict__ masked_m = nullptr,
                                                       long long rows_per_group = 0, long long s_stride_g = 0) {
  const int kg = I / 128;
  const long long total = M * kg;
  const int sub = threadIdx.x & 15;
  for (long long g = (long long)blockIdx.x * 16 + (threadIdx.x >> 4); g < total; g += (long long)gridDim.x * 16) {
    const long long m = total < (1ll << 31) ? (long long)((unsigned)g / (unsigned)kg) : g / kg;
    const int kb = (int)(g - m * kg);
    long long s_off = m * s_stride_m;
    if (masked_m != nullptr) {
      const long long grp = total < (1ll << 31) ? (long long)((unsigned)m / (unsigned)rows_per_group) : m / rows_per_group;
      const long long r = m - grp * rows_per_group;
      if (r >= masked_m[grp]) continue;   // (uniform over the 16 lanes of a group)
      s_off = grp * s_stride_g + r * s_stride_m;
    }
    const uint16_t* pg = x + m * (2ll * I) + kb * 128 + sub * 8;
    float a[8], b[8], r[8];
    unpack8(*reinterpret_cast<const uint4*>(pg), a);
    unpack8(*reinterpret_cast<const uint4*>(pg + I), b);
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = silu_mul_bf16(a[i], b[i]);
    if (out != nullptr) {
      uint32_t o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (uint32_t)fl_f32_to_bf16(r[2 * i]) | ((uint32_t)fl_f32_to_bf16(r[2 * i + 1]) << 16);
      *reinterpret_cast<uint4*>(out + m * I + kb * 128 + sub * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    if (kQuant) {
      float amax = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(r[i]));
      amax = group16_max(amax);
      const float s = fmaxf(amax, fl_bf16_to_f32(fl_f32_to_bf16(1e-10f))) / FL_FP8_MAX;
      *reinterpret_cast<uint2*>(q + m * I + kb * 128 + sub * 8) = quant8(r, s);
      if (sub == 0) qs[s_off + kb * s_stride_k] = s;
    }
  }
}

unsigned grid_for(long long groups) {
  long long blocks = (groups + 15) / 16;
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

}  // namespace

extern "C" int fl_quant_1x128(const void* x, int64_t M, int K, float eps, void* x_q, float* x_s, int64_t s_stride_m,
                              int64_t s_stride_k, fl_stream_t stream) {
  FL_CHECK_ARG(x && x_q && x_s, "fl_quant_1x128: null pointer");
  FL_CHECK_ARG(M >= 0 && K > 0 && K % 128 == 0, "fl_quant_1x128: K=%d must be a multiple of 128", K);
  if (M == 0) return FL_OK;
  quant_1x128_kernel<<<dim3(grid_for(M * (K / 128))), dim3(256), 0, (hipStream_t)stream>>>(
      (const uint16_t*)x, M, K, eps, (uint8_t*)x_q, x_s, s_stride_m, s_stride_k);
  FL_CHECK_LAUNCH("fl_quant_1x128");
  return FL_OK;
}

extern "C" int fl_silu_and_mul(const void* x, int64_t M, int I, void* out_bf16, void* q_out, float* s_out,
                               int64_t s_stride_m, int64_t s_stride_k, fl_stream_t stream) {
  FL_CHECK_ARG(x && (out_bf16 || q_out), "fl_silu_and_mul: null pointer");
  FL_CHECK_ARG(q_out == nullptr || s_out != nullptr, "fl_silu_and_mul: quantised output needs a scale buffer");
  FL_CHECK_ARG(M >= 0 && I > 0 && I % 128 == 0, "fl_silu_and_mul: I=%d must be a multiple of 128", I);
  if (M == 0) return FL_OK;
  const dim3 grid(grid_for(M * (I / 128))), block(256);
  if (q_out)
    silu_mul_kernel<true><<<grid, block, 0, (hipStream_t)stream>>>((const uint16_t*)x, M, I, (uint16_t*)out_bf16,
                                                                   (uint8_t*)q_out, s_out, s_stride_m, s_stride_k);
  else
    silu_mul_kernel<false><<<grid, block, 0, (hipStream_t)stream>>>((const uint16_t*)x, M, I, (uint16_t*)out_bf16, nullptr,
                                                                    nullptr, 0, 0);
  FL_CHECK_LAUNCH("fl_silu_and_mul");
  return FL_OK;
}

extern "C" int fl_silu_and_mul_masked(const void* x, int num_groups, int64_t rows_per_group, int I, const int32_t* masked_m,
                                      void* q_out, float* s_out, int64_t s_stride_g, int64_t s_stride_m, int64_t s_stride_k,
                                      fl_stream_t stream) {
  FL_CHECK_ARG(x && masked_m && q_out && s_out, "fl_silu_and_mul_masked: null pointer");
  FL_CHECK_ARG(num_groups >= 0 && rows_per_group >= 0 && I > 0 && I % 128 == 0,
               "fl_silu_and_mul_masked: I=%d must be a multiple of 128", I);
  const long long M = (long long)num_groups * rows_per_group;
  if (M == 0) return FL_OK;
  silu_mul_kernel<true><<<dim3(grid_for(M * (I / 128))), dim3(256), 0, (hipStream_t)stream>>>(
      (const uint16_t*)x, M, I, nullptr, (uint8_t*)q_out, s_out, s_stride_m, s_stride_k, masked_m, rows_per_group, s_stride_g);
  FL_CHECK_LAUNCH("fl_silu_and_mul_masked");
  return FL_OK;
}
