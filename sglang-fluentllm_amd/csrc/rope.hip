// R2 — rotary embedding of q_pe / k_pe in place: flashinfer.apply_rope_with_cos_sin_cache_inplace as RotaryEmbedding
// .forward_cuda calls it (python/sglang/srt/layers/rotary_embedding.py:203-218) on the MLA path
// (models/deepseek_v2.py:646-647,695-696: q_pe is a strided view of q, k_pe of the latent row).  Arithmetic of the
// reference's torch statement DeepseekScalingRotaryEmbedding.forward_native (:804-846) with the fp32 cache of the CUDA
// path (:113-115): fp32, each product and the sum rounded separately (no fma contraction), one rounding to bf16 —
// bit-exact against golden.  One thread per 4 rotation pairs (GPT-J: 8 neighbouring elements, one 16-B access; NeoX:
// 4 + 4 elements half a rotary_dim apart); q heads and k heads of a token are rows of one launch.  HBM/latency-bound.
#include "fl_common.h"
#pragma clang fp contract(off)   // a*c + b*s as three roundings, like the torch statement (hipcc contracts to fma by default)

namespace {
__device__ __forceinline__ float lo_f(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_f(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float rot_a(float a, float b, float c, float s) { return a * c + (-b) * s; }
__device__ __forceinline__ float rot_b(float a, float b, float c, float s) { return b * c + a * s; }

template <bool kNeox>
__global__ __launch_bounds__(256) void rope_kernel(const long long* __restrict__ positions, long long T, uint16_t* q,
                                                   long long q_stride_t, long long q_stride_h, int Hq, uint16_t* k,
                                                   long long k_stride_t, long long k_stride_h, int Hk,
                                                   const float* __restrict__ cache, long long max_pos, int R) {
  const int per_row = R / 8;   // threads per (token, head) row
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long row = gid / per_row;
  const int c = (int)(gid % per_row);
  const int H = Hq + Hk;
  if (row >= T * H) return;
  const long long t = row / H;
  const int h = (int)(row % H);
  long long pos = positions[t];
  pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);   // never read outside the cache (torch would raise)
  uint16_t* x = h < Hq ? q + t * q_stride_t + h * q_stride_h : k + t * k_stride_t + (h - Hq) * k_stride_h;
  const float4 cs = *reinterpret_cast<const float4*>(cache + pos * R + 4 * c);
  const float4 sn = *reinterpret_cast<const float4*>(cache + pos * R + R / 2 + 4 * c);
  if (kNeox) {
    uint2* p1 = reinterpret_cast<uint2*>(x + 4 * c);
    uint2* p2 = reinterpret_cast<uint2*>(x + R / 2 + 4 * c);
    const uint2 a = *p1, b = *p2;
    const float a0 = lo_f(a.x), a1 = hi_f(a.x), a2 = lo_f(a.y), a3 = hi_f(a.y);
    const float b0 = lo_f(b.x), b1 = hi_f(b.x), b2 = lo_f(b.y), b3 = hi_f(b.y);
    *p1 = make_uint2(fl_pack_bf16(rot_a(a0, b0, cs.x, sn.x), rot_a(a1, b1, cs.y, sn.y)),
                     fl_pack_bf16(rot_a(a2, b2, cs.z, sn.z), rot_a(a3, b3, cs.w, sn.w)));
    *p2 = make_uint2(fl_pack_bf16(rot_b(a0, b0, cs.x, sn.x), rot_b(a1, b1, cs.y, sn.y)),
                     fl_pack_bf16(rot_b(a2, b2, cs.z, sn.z), rot_b(a3, b3, cs.w, sn.w)));
  } else {
    uint4* p = reinterpret_cast<uint4*>(x + 8 * c);
    const uint4 v = *p;
    *p = make_uint4(fl_pack_bf16(rot_a(lo_f(v.x), hi_f(v.x), cs.x, sn.x), rot_b(lo_f(v.x), hi_f(v.x), cs.x, sn.x)),
                    fl_pack_bf16(rot_a(lo_f(v.y), hi_f(v.y), cs.y, sn.y), rot_b(lo_f(v.y), hi_f(v.y), cs.y, sn.y)),
                    fl_pack_bf16(rot_a(lo_f(v.z), hi_f(v.z), cs.z, sn.z), rot_b(lo_f(v.z), hi_f(v.z), cs.z, sn.z)),
                    fl_pack_bf16(rot_a(lo_f(v.w), hi_f(v.w), cs.w, sn.w), rot_b(lo_f(v.w), hi_f(v.w), cs.w, sn.w)));
  }
}
}  // namespace

extern "C" int fl_rope_inplace(const int64_t* positions, int64_t num_tokens, void* q, int64_t q_stride_token,
                               int64_t q_stride_head, int num_q_heads, void* k, int64_t k_stride_token,
                               int64_t k_stride_head, int num_k_heads, const float* cos_sin_cache, int64_t max_position,
                               int rotary_dim, int is_neox, fl_stream_t stream) {
  FL_CHECK_ARG(positions && cos_sin_cache && (q || num_q_heads == 0) && (k || num_k_heads == 0),
               "fl_rope_inplace: null pointer");
  FL_CHECK_ARG(num_tokens >= 0 && num_q_heads >= 0 && num_k_heads >= 0 && max_position > 0, "fl_rope_inplace: bad sizes");
  FL_CHECK_ARG(rotary_dim >= 8 && rotary_dim % 8 == 0, "fl_rope_inplace: rotary_dim=%d must be a multiple of 8", rotary_dim);
  FL_CHECK_ARG(q_stride_token % 8 == 0 && q_stride_head % 8 == 0 && k_stride_token % 8 == 0 && k_stride_head % 8 == 0 &&
                   ((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0 && ((uintptr_t)cos_sin_cache & 15) == 0,
               "fl_rope_inplace: rows must be 16-byte aligned (strides in multiples of 8 elements)");
  const long long threads = num_tokens * (long long)(num_q_heads + num_k_heads) * (rotary_dim / 8);
  if (threads == 0) return FL_OK;
  const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
  if (is_neox)
    rope_kernel<true><<<grid, block, 0, (hipStream_t)stream>>>((const long long*)positions, num_tokens, (uint16_t*)q,
                                                               q_stride_token, q_stride_head, num_q_heads, (uint16_t*)k,
                                                               k_stride_token, k_stride_head, num_k_heads, cos_sin_cache,
                                                               max_position, rotary_dim);
  else
    rope_kernel<false><<<grid, block, 0, (hipStream_t)stream>>>((const long long*)positions, num_tokens, (uint16_t*)q,
                                                                q_stride_token, q_stride_head, num_q_heads, (uint16_t*)k,
                                                                k_stride_token, k_stride_head, num_k_heads, cos_sin_cache,
                                                                max_position, rotary_dim);
  FL_CHECK_LAUNCH("fl_rope_inplace");
  return FL_OK;
}
