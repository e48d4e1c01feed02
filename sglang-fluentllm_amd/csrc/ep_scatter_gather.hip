// C8/C9 — ep_scatter / ep_gather of the DeepExecutor (python/sglang/srt/layers/moe/executors/deep_ep_executor.py:173-430,
// Triton in the reference): the step between the DeepEP dispatch and the contiguous grouped GEMM (G2), and back.
//   ep_scatter: expert_start_loc = exclusive cumsum of the (128-aligned) per-expert row counts, m_indices[row] = expert of
//     the row's group, then every received token row (fp8 [H] + 1x128 scales [H/128]) is copied to one row of each of its
//     local experts' groups; output_index[t, k] = that row.  Positions inside a group come from an atomic cursor per
//     expert, as in the reference (:247): the order inside a group is unspecified in both; expert_start_loc ends as
//     start + count like the reference's.
//   ep_gather: out[t] = sum_k (id >= 0) w[t,k] * y[index[t,k]], fp32, k ascending, one rounding to bf16.
// Integer / byte work bit-exact; rows are copied by a workgroup per token with unconditional 16-B loads.
#include "fl_common.h"

namespace {
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// one workgroup: cumsum (E <= 1024), m_indices fill
__global__ __launch_bounds__(256) void ep_scatter_plan_kernel(const int32_t* __restrict__ counts, int E,
                                                              int32_t* __restrict__ start_loc, int32_t* __restrict__ m_indices,
                                                              long long M) {
  __shared__ int s_start[1025];
  if (threadIdx.x == 0) {
    int run = 0;
    for (int e = 0; e < E; ++e) { s_start[e] = run; run += counts[e]; }
    s_start[E] = run;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += 256) start_loc[e] = s_start[e];
  for (int e = 0; e < E; ++e) {
    const long long b = s_start[e];
    long long n = ((long long)counts[e] + 127) / 128 * 128;   // (:200-204 stores whole 128-row blocks)
    if (b + n > M) n = M - b;
    for (long long i = threadIdx.x; i < n; i += 256) m_indices[b + i] = e;
  }
}

template <typename IdT>
__global__ __launch_bounds__(256) void ep_scatter_rows_kernel(const uint8_t* __restrict__ x, long long x_stride,
                                                              const float* __restrict__ xs, long long xs_stride,
                                                              const IdT* __restrict__ topk, long long topk_stride, int K, int E,
                                                              int32_t* __restrict__ cursor, uint8_t* __restrict__ out,
                                                              long long out_stride, float* __restrict__ outs,
                                                              long long outs_stride, int32_t* __restrict__ out_index,
                                                              long long oi_stride, int H, long long M) {
  __shared__ int s_dest[64];
  const long long t = blockIdx.x;
  if (threadIdx.x < K) {
    const long long e = (long long)topk[t * topk_stride + threadIdx.x];
    int d = -1;
    if (e >= 0 && e < E) {
      d = atomicAdd(&cursor[e], 1);
      out_index[t * oi_stride + threadIdx.x] = d;   // (entries of experts of other ranks are left as they are, :246)
      if (d < 0 || d >= M) d = -1;                  // never write outside the buffers
    }
    s_dest[threadIdx.x] = d;
  }
  __syncthreads();
  const int nc = H / 16, ns = H / 128;
  // the token's row and scales stay in registers (H <= 16 KiB): 4 x 16 B + 1 float per thread
  u32x4 v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = threadIdx.x + 256 * u;
    v[u] = *reinterpret_cast<const u32x4*>(x + t * x_stride + 16ll * (c < nc ? c : 0));
  }
  const float sc = xs[t * xs_stride + (threadIdx.x < ns ? threadIdx.x : 0)];
  asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
  for (int k = 0; k < K; ++k) {
    const int d = s_dest[k];
    if (d < 0) continue;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = threadIdx.x + 256 * u;
      if (c < nc) *reinterpret_cast<u32x4*>(out + d * out_stride + 16ll * c) = v[u];
    }
    if (threadIdx.x < ns) outs[d * outs_stride + threadIdx.x] = sc;
  }
}

template <typename IdT>
__global__ __launch_bounds__(256) void ep_gather_kernel(const uint16_t* __restrict__ y, long long y_stride,
                                                        const IdT* __restrict__ ids, long long ids_stride,
                                                        const float* __restrict__ w, long long w_stride,
                                                        const int32_t* __restrict__ index, long long idx_stride, int K, int H,
                                                        long long M, uint16_t* __restrict__ out, long long out_stride) {
  __shared__ long long s_row[64];
  __shared__ float s_w[64];
  const long long t = blockIdx.x;
  if (threadIdx.x < K) {
    const long long e = (long long)ids[t * ids_stride + threadIdx.x];
    const long long r = index[t * idx_stride + threadIdx.x];
    const bool ok = e >= 0 && r >= 0 && r < M;
    s_row[threadIdx.x] = ok ? r : -1;
    s_w[threadIdx.x] = ok ? w[t * w_stride + threadIdx.x] : 0.f;
  }
  __syncthreads();
  for (int c = blockIdx.y * 256 + threadIdx.x; c < H / 8; c += 256 * gridDim.y) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int kb = 0; kb < K; kb += 8) {
      u32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {   // unconditional loads (an empty slot reads row 0 and is skipped below)
        const long long r = kb + j < K ? s_row[kb + j] : -1;
        v[j] = *reinterpret_cast<const u32x4*>(y + (r >= 0 ? r : 0) * y_stride + 8ll * c);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (kb + j < K && s_row[kb + j] >= 0) {
          const float wk = s_w[kb + j];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[2 * q] += wk * __uint_as_float(v[j][q] << 16);
            acc[2 * q + 1] += wk * __uint_as_float(v[j][q] & 0xffff0000u);
          }
        }
      }
    }
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = fl_pack_bf16(acc[2 * q], acc[2 * q + 1]);
    *reinterpret_cast<u32x4*>(out + t * out_stride + 8ll * c) = o;
  }
}
}  // namespace

extern "C" int fl_ep_scatter(const void* recv_x, int64_t x_stride, const float* recv_x_scale, int64_t xs_stride,
                             const void* recv_topk, int topk_is_int64, int64_t topk_stride, int64_t num_tokens, int top_k,
                             int hidden, const int32_t* num_recv_tokens_per_expert, int num_experts,
                             int32_t* expert_start_loc, void* output_tensor, int64_t out_stride, float* output_tensor_scale,
                             int64_t outs_stride, int32_t* m_indices, int64_t num_rows, int32_t* output_index,
                             int64_t oi_stride, fl_stream_t stream) {
  FL_CHECK_ARG(num_recv_tokens_per_expert && expert_start_loc && m_indices, "fl_ep_scatter: null pointer");
  FL_CHECK_ARG(num_experts >= 1 && num_experts <= 1024 && top_k >= 1 && top_k <= 64 && num_tokens >= 0 && num_rows >= 0,
               "fl_ep_scatter: bad sizes");
  FL_CHECK_ARG(hidden % 128 == 0 && hidden >= 128 && hidden <= 16384 && x_stride % 16 == 0 && out_stride % 16 == 0,
               "fl_ep_scatter: hidden=%d (fp8 rows of 128 .. 16384 bytes, 16-byte aligned strides)", hidden);
  ep_scatter_plan_kernel<<<1, 256, 0, (hipStream_t)stream>>>(num_recv_tokens_per_expert, num_experts, expert_start_loc,
                                                             m_indices, num_rows);
  FL_CHECK_LAUNCH("fl_ep_scatter(plan)");
  if (num_tokens == 0) return FL_OK;
  FL_CHECK_ARG(recv_x && recv_x_scale && recv_topk && output_tensor && output_tensor_scale && output_index,
               "fl_ep_scatter: null pointer");
  if (topk_is_int64)
    ep_scatter_rows_kernel<long long><<<dim3((unsigned)num_tokens), 256, 0, (hipStream_t)stream>>>(
        (const uint8_t*)recv_x, x_stride, recv_x_scale, xs_stride, (const long long*)recv_topk, topk_stride, top_k, num_experts,
        expert_start_loc, (uint8_t*)output_tensor, out_stride, output_tensor_scale, outs_stride, output_index, oi_stride,
        hidden, num_rows);
  else
    ep_scatter_rows_kernel<int><<<dim3((unsigned)num_tokens), 256, 0, (hipStream_t)stream>>>(
        (const uint8_t*)recv_x, x_stride, recv_x_scale, xs_stride, (const int*)recv_topk, topk_stride, top_k, num_experts,
        expert_start_loc, (uint8_t*)output_tensor, out_stride, output_tensor_scale, outs_stride, output_index, oi_stride,
        hidden, num_rows);
  FL_CHECK_LAUNCH("fl_ep_scatter(rows)");
  return FL_OK;
}

extern "C" int fl_ep_gather(const void* input_tensor, int64_t in_stride, int64_t num_rows, const void* recv_topk_ids,
                            int ids_is_int64, int64_t ids_stride, const float* recv_topk_weight, int64_t w_stride,
                            const int32_t* input_index, int64_t idx_stride, int64_t num_tokens, int top_k, int hidden,
                            void* output_tensor, int64_t out_stride, fl_stream_t stream) {
  FL_CHECK_ARG(top_k >= 1 && top_k <= 64 && hidden % 8 == 0 && hidden >= 8 && num_tokens >= 0 && in_stride % 8 == 0 &&
                   out_stride % 8 == 0, "fl_ep_gather: bad sizes (bf16 rows, strides in multiples of 8 elements)");
  if (num_tokens == 0) return FL_OK;
  FL_CHECK_ARG(input_tensor && recv_topk_ids && recv_topk_weight && input_index && output_tensor, "fl_ep_gather: null pointer");
  const dim3 grid((unsigned)num_tokens, (unsigned)((hidden / 8 + 255) / 256));
  if (ids_is_int64)
    ep_gather_kernel<long long><<<grid, 256, 0, (hipStream_t)stream>>>(
        (const uint16_t*)input_tensor, in_stride, (const long long*)recv_topk_ids, ids_stride, recv_topk_weight, w_stride,
        input_index, idx_stride, top_k, hidden, num_rows, (uint16_t*)output_tensor, out_stride);
  else
    ep_gather_kernel<int><<<grid, 256, 0, (hipStream_t)stream>>>(
        (const uint16_t*)input_tensor, in_stride, (const int*)recv_topk_ids, ids_stride, recv_topk_weight, w_stride,
        input_index, idx_stride, top_k, hidden, num_rows, (uint16_t*)output_tensor, out_stride);
  FL_CHECK_LAUNCH("fl_ep_gather");
  return FL_OK;
}
