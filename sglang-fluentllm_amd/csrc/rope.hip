// R2 — rotary embedding of q_pe / k_pe in place: flashinfer.apply_rope_with_cos_sin_cache_inplace as RotaryEmbedding
// .forward_cuda calls it (python/sglang/srt/layers/rotary_embedding.py:203-218) on the MLA path
// (models/deepseek_v2.py:646-647,695-696: q_pe is a strided view of q, k_pe of the latent row).  Arithmetic of the
// reference's torch statement DeepseekScalingRotaryEmbedding.forward_native (:804-846) with the fp32 cache of the CUDA
// path (:113-115): fp32, each product and the sum rounded separately (no fma contraction), one rounding to bf16 —
// bit-exact against golden.  One thread per 4 rotation pairs (GPT-J: 8 neighbouring elements, one 16-B access; NeoX:
// 4 + 4 elements half a rotary_dim apart); q heads and k heads of a token are rows of one launch.  HBM/latency-bound.
// fl_rope adds what forward_absorb_prepare asks of the same call (models/deepseek_v2.py:843-861): the rotated query into a
// SEPARATE tensor (output_q_rope = the rope columns of the absorbed Q) and, for a bf16 KV cache, the fused set-KV
// (FusedSetKVBufferArg, models/utils.py:52-81): value -> v_buffer[cache_loc], rotated key -> k_buffer[cache_loc], same launch.
#include "fl_common.h"
#pragma clang fp contract(off)   // a*c + b*s as three roundings, like the torch statement (hipcc contracts to fma by default)

namespace {
__device__ __forceinline__ float lo_f(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_f(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float rot_a(float a, float b, float c, float s) { return a * c + (-b) * s; }
__device__ __forceinline__ float rot_b(float a, float b, float c, float s) { return b * c + a * s; }

template <bool kNeox>
__global__ __launch_bounds__(256) void rope_kernel(const FlRopeArgs A) {
  const int R = A.rotary_dim;
  const int per_row = R / 8;   // threads per (token, head) row
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const int H = A.num_q_heads + A.num_k_heads;
  const long long T = A.num_tokens;
  const long long rope_threads = T * H * per_row;
  if (gid >= rope_threads) {
    // fused set-KV, value half: v_buffer[cache_loc[t]] = value[t] (16 B per thread)
    const long long g2 = gid - rope_threads;
    const int vper = A.value_dim / 8;
    if (A.v_buffer == nullptr || g2 >= T * vper) return;
    const long long t = g2 / vper;
    const int c = (int)(g2 % vper);
    const long long loc = A.cache_loc_is_i64 ? reinterpret_cast<const long long*>(A.cache_loc)[t]
                                             : (long long)reinterpret_cast<const int*>(A.cache_loc)[t];
    if (loc < 0) return;
    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(A.v_buffer) + loc * A.v_buffer_stride + 8 * c) =
        *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(A.value) + t * A.value_stride + 8 * c);
    return;
  }
  const long long row = gid / per_row;
  const int c = (int)(gid % per_row);
  const long long t = row / H;
  const int h = (int)(row % H);
  long long pos = A.positions[t];
  pos = pos < 0 ? 0 : (pos >= A.max_position ? A.max_position - 1 : pos);   // never read outside the cache (torch would raise)
  const bool is_q = h < A.num_q_heads;
  const uint16_t* x = is_q ? reinterpret_cast<const uint16_t*>(A.q) + t * A.q_stride_token + h * A.q_stride_head
                           : reinterpret_cast<const uint16_t*>(A.k) + t * A.k_stride_token + (h - A.num_q_heads) * A.k_stride_head;
  // destination: in place unless a separate output is given; the rotated key additionally lands in the cache row
  uint16_t* y = is_q ? (A.q_out ? reinterpret_cast<uint16_t*>(A.q_out) + t * A.qo_stride_token + h * A.qo_stride_head
                                : const_cast<uint16_t*>(x))
                     : (A.k_out ? reinterpret_cast<uint16_t*>(A.k_out) + t * A.ko_stride_token + (h - A.num_q_heads) * A.ko_stride_head
                                : const_cast<uint16_t*>(x));
  uint16_t* y2 = nullptr;
  if (!is_q && A.k_buffer != nullptr) {
    const long long loc = A.cache_loc_is_i64 ? reinterpret_cast<const long long*>(A.cache_loc)[t]
                                             : (long long)reinterpret_cast<const int*>(A.cache_loc)[t];
    if (loc >= 0) y2 = reinterpret_cast<uint16_t*>(A.k_buffer) + loc * A.k_buffer_stride + (long long)(h - A.num_q_heads) * R;
  }
  const float* cache = A.cos_sin_cache;
  const float4 cs = *reinterpret_cast<const float4*>(cache + pos * R + 4 * c);
  const float4 sn = *reinterpret_cast<const float4*>(cache + pos * R + R / 2 + 4 * c);
  if (kNeox) {
    const uint2 a = *reinterpret_cast<const uint2*>(x + 4 * c), b = *reinterpret_cast<const uint2*>(x + R / 2 + 4 * c);
    const float a0 = lo_f(a.x), a1 = hi_f(a.x), a2 = lo_f(a.y), a3 = hi_f(a.y);
    const float b0 = lo_f(b.x), b1 = hi_f(b.x), b2 = lo_f(b.y), b3 = hi_f(b.y);
    const uint2 o1 = make_uint2(fl_pack_bf16(rot_a(a0, b0, cs.x, sn.x), rot_a(a1, b1, cs.y, sn.y)),
                                fl_pack_bf16(rot_a(a2, b2, cs.z, sn.z), rot_a(a3, b3, cs.w, sn.w)));
    const uint2 o2 = make_uint2(fl_pack_bf16(rot_b(a0, b0, cs.x, sn.x), rot_b(a1, b1, cs.y, sn.y)),
                                fl_pack_bf16(rot_b(a2, b2, cs.z, sn.z), rot_b(a3, b3, cs.w, sn.w)));
    *reinterpret_cast<uint2*>(y + 4 * c) = o1;
    *reinterpret_cast<uint2*>(y + R / 2 + 4 * c) = o2;
    if (y2 != nullptr) {
      *reinterpret_cast<uint2*>(y2 + 4 * c) = o1;
      *reinterpret_cast<uint2*>(y2 + R / 2 + 4 * c) = o2;
    }
  } else {
    const uint4 v = *reinterpret_cast<const uint4*>(x + 8 * c);
    const uint4 o =
        make_uint4(fl_pack_bf16(rot_a(lo_f(v.x), hi_f(v.x), cs.x, sn.x), rot_b(lo_f(v.x), hi_f(v.x), cs.x, sn.x)),
                   fl_pack_bf16(rot_a(lo_f(v.y), hi_f(v.y), cs.y, sn.y), rot_b(lo_f(v.y), hi_f(v.y), cs.y, sn.y)),
                   fl_pack_bf16(rot_a(lo_f(v.z), hi_f(v.z), cs.z, sn.z), rot_b(lo_f(v.z), hi_f(v.z), cs.z, sn.z)),
                   fl_pack_bf16(rot_a(lo_f(v.w), hi_f(v.w), cs.w, sn.w), rot_b(lo_f(v.w), hi_f(v.w), cs.w, sn.w)));
    *reinterpret_cast<uint4*>(y + 8 * c) = o;
    if (y2 != nullptr) *reinterpret_cast<uint4*>(y2 + 8 * c) = o;
  }
}
}  // namespace

extern "C" int fl_rope(const FlRopeArgs* args, fl_stream_t stream) {
  FL_CHECK_ARG(args != nullptr, "fl_rope: null arguments");
  const FlRopeArgs A = *args;
  FL_CHECK_ARG(A.positions && A.cos_sin_cache && (A.q || A.num_q_heads == 0) && (A.k || A.num_k_heads == 0), "fl_rope: null pointer");
  FL_CHECK_ARG(A.num_tokens >= 0 && A.num_q_heads >= 0 && A.num_k_heads >= 0 && A.max_position > 0, "fl_rope: bad sizes");
  FL_CHECK_ARG(A.rotary_dim >= 8 && A.rotary_dim % 8 == 0, "fl_rope: rotary_dim=%d must be a multiple of 8", A.rotary_dim);
  auto al = [](const void* p, int64_t a, int64_t b) { return (((uintptr_t)p) & 15) == 0 && a % 8 == 0 && b % 8 == 0; };
  FL_CHECK_ARG(al(A.q, A.q_stride_token, A.q_stride_head) && al(A.k, A.k_stride_token, A.k_stride_head) &&
                   al(A.q_out, A.qo_stride_token, A.qo_stride_head) && al(A.k_out, A.ko_stride_token, A.ko_stride_head) &&
                   ((uintptr_t)A.cos_sin_cache & 15) == 0,
               "fl_rope: rows must be 16-byte aligned (strides in multiples of 8 elements)");
  const bool set_kv = A.k_buffer != nullptr || A.v_buffer != nullptr;
  if (set_kv) {
    FL_CHECK_ARG(A.k_buffer && A.v_buffer && A.value && A.cache_loc, "fl_rope: fused set-KV needs k_buffer, v_buffer, value and cache_loc");
    FL_CHECK_ARG(A.value_dim > 0 && A.value_dim % 8 == 0 && al(A.k_buffer, A.k_buffer_stride, 0) &&
                     al(A.v_buffer, A.v_buffer_stride, 0) && al(A.value, A.value_stride, 0),
                 "fl_rope: fused set-KV rows must be 16-byte aligned, value_dim a multiple of 8");
  }
  const long long rope_threads = A.num_tokens * (long long)(A.num_q_heads + A.num_k_heads) * (A.rotary_dim / 8);
  const long long threads = rope_threads + (set_kv ? A.num_tokens * (long long)(A.value_dim / 8) : 0);
  if (threads == 0) return FL_OK;
  const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
  if (A.is_neox) rope_kernel<true><<<grid, block, 0, (hipStream_t)stream>>>(A);
  else rope_kernel<false><<<grid, block, 0, (hipStream_t)stream>>>(A);
  FL_CHECK_LAUNCH("fl_rope");
  return FL_OK;
}

extern "C" int fl_rope_inplace(const int64_t* positions, int64_t num_tokens, void* q, int64_t q_stride_token,
                               int64_t q_stride_head, int num_q_heads, void* k, int64_t k_stride_token,
                               int64_t k_stride_head, int num_k_heads, const float* cos_sin_cache, int64_t max_position,
                               int rotary_dim, int is_neox, fl_stream_t stream) {
  FlRopeArgs A{};
  A.positions = positions; A.num_tokens = num_tokens;
  A.q = q; A.q_stride_token = q_stride_token; A.q_stride_head = q_stride_head; A.num_q_heads = num_q_heads;
  A.k = k; A.k_stride_token = k_stride_token; A.k_stride_head = k_stride_head; A.num_k_heads = num_k_heads;
  A.cos_sin_cache = cos_sin_cache; A.max_position = max_position; A.rotary_dim = rotary_dim; A.is_neox = is_neox;
  return fl_rope(&A, stream);
}
