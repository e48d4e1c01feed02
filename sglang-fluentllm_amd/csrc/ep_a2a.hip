// C1/C2 — device side of the expert-parallel all-to-all (eps.fast_ep.AllToAll.dispatch / combine, call sites
// /root/reference/python/sglang/srt/layers/moe/dispatcher/fast_ep.py:45-51,73-78).
//
// The exchange itself is ONE equal-split all-to-all per direction over RCCL/xGMI (issued by the Python host mirror with
// torch.distributed); everything around it is integer/byte work done here, sync-free and with static shapes so that the
// whole MoE step stays hipGraph-capturable:
//   route   : (token, k) pair p -> slot dest_rank*cap + position inside the peer slab (+ local expert id per slot)
//   sort    : received slots -> rows grouped by local expert (exclusive_sum for the grouped GEMM) + their source slot
//   gather / scatter : 16-B vector row copies by index
//   combine : out[t] = sum_k w[t,k] * returned_row(slot of pair (t,k)), fp32 accumulate, bf16 out
// Order inside one expert's group is arbitrary (atomic cursors): every row's result is independent of its position.
#include "fl_common.h"

namespace {

__global__ __launch_bounds__(256) void ep_route_kernel(const int32_t* __restrict__ indices, int P, int experts_per_rank,
                                                       int world, int cap, int32_t* __restrict__ send_slot,
                                                       int32_t* __restrict__ send_eid) {
  __shared__ int cursor[64];
  for (int i = threadIdx.x; i < world * cap; i += 256) send_eid[i] = -1;
  if (threadIdx.x < 64) cursor[threadIdx.x] = 0;
  __syncthreads();
  __threadfence_block();
  for (int p = threadIdx.x; p < P; p += 256) {
    const int e = indices[p];
    int slot = -1;
    if (e >= 0 && e < experts_per_rank * world) {
      const int dst = e / experts_per_rank;
      const int pos = atomicAdd(&cursor[dst], 1);
      if (pos < cap) {
        slot = dst * cap + pos;
        send_eid[slot] = e - dst * experts_per_rank;
      }
    }
    send_slot[p] = slot;
  }
}

__global__ __launch_bounds__(256) void ep_sort_kernel(const int32_t* __restrict__ recv_eid, int S, int E,
                                                      int32_t* __restrict__ order, int32_t* __restrict__ exclusive_sum) {
  extern __shared__ int bins[];   // E + 1 counters, then E + 1 cursors
  int* cur = bins + E + 1;
  for (int i = threadIdx.x; i <= E; i += 256) bins[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < S; i += 256) {
    const int e = recv_eid[i];
    atomicAdd(&bins[(e >= 0 && e < E) ? e : E], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int e = 0; e <= E; ++e) {
      const int c = bins[e];
      cur[e] = run;
      if (e < E) exclusive_sum[e] = run;
      run += c;
      if (e == E - 1) exclusive_sum[E] = run;
    }
    if (E == 0) exclusive_sum[0] = 0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < S; i += 256) {
    const int e = recv_eid[i];
    const int pos = atomicAdd(&cur[(e >= 0 && e < E) ? e : E], 1);
    order[pos] = i;   // invalid slots end up after every valid row
  }
}

// one wave per row; hidden*2 bytes per row, 16 B per lane per step
template <bool kScatter>
__global__ __launch_bounds__(256) void ep_rows_kernel(const uint16_t* __restrict__ src, const int32_t* __restrict__ idx,
                                                      long long n, int hidden, long long src_rows, long long dst_rows,
                                                      uint16_t* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const long long j = idx[i];
  const long long s = kScatter ? i : j, d = kScatter ? j : i;
  if (s < 0 || s >= src_rows || d < 0 || d >= dst_rows) return;
  const uint4* sp = reinterpret_cast<const uint4*>(src + s * hidden);
  uint4* dp = reinterpret_cast<uint4*>(dst + d * hidden);
  for (int c = lane; c < hidden / 8; c += 64) dp[c] = sp[c];
}

__global__ __launch_bounds__(256) void ep_combine_kernel(const uint16_t* __restrict__ ret, const int32_t* __restrict__ slot,
                                                         const float* __restrict__ w, long long t_count, int top_k,
                                                         int hidden, long long ret_rows, uint16_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= t_count) return;
  for (int c = lane; c < hidden / 8; c += 64) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < top_k; ++k) {
      const long long s = slot[t * top_k + k];
      if (s < 0 || s >= ret_rows) continue;
      const float wk = w[t * top_k + k];
      const uint4 v = reinterpret_cast<const uint4*>(ret + s * hidden)[c];
      const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[2 * q] += wk * __uint_as_float(u[q] << 16);
        acc[2 * q + 1] += wk * __uint_as_float(u[q] & 0xffff0000u);
      }
    }
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = (uint32_t)fl_f32_to_bf16(acc[2 * q]) | ((uint32_t)fl_f32_to_bf16(acc[2 * q + 1]) << 16);
    reinterpret_cast<uint4*>(out + t * hidden)[c] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

}  // namespace

extern "C" int fl_ep_route(const int32_t* indices, int64_t num_pairs, int experts_per_rank, int world, int cap,
                           int32_t* send_slot, int32_t* send_eid, fl_stream_t stream) {
  FL_CHECK_ARG(indices && send_slot && send_eid, "fl_ep_route: null pointer");
  FL_CHECK_ARG(world >= 1 && world <= 64 && cap >= 1 && experts_per_rank >= 1 && num_pairs >= 0, "fl_ep_route: bad sizes");
  ep_route_kernel<<<1, 256, 0, (hipStream_t)stream>>>(indices, (int)num_pairs, experts_per_rank, world, cap, send_slot, send_eid);
  FL_CHECK_LAUNCH("fl_ep_route");
  return FL_OK;
}

extern "C" int fl_ep_sort(const int32_t* recv_eid, int64_t num_slots, int num_local_experts, int32_t* order,
                          int32_t* exclusive_sum, fl_stream_t stream) {
  FL_CHECK_ARG(recv_eid && order && exclusive_sum, "fl_ep_sort: null pointer");
  FL_CHECK_ARG(num_local_experts >= 1 && num_local_experts <= 4096 && num_slots >= 0, "fl_ep_sort: bad sizes");
  ep_sort_kernel<<<1, 256, 2 * (num_local_experts + 1) * sizeof(int), (hipStream_t)stream>>>(
      recv_eid, (int)num_slots, num_local_experts, order, exclusive_sum);
  FL_CHECK_LAUNCH("fl_ep_sort");
  return FL_OK;
}

extern "C" int fl_ep_gather_rows(const void* src, int64_t src_rows, const int32_t* idx, int64_t n, int hidden, void* dst,
                                 int64_t dst_rows, fl_stream_t stream) {
  FL_CHECK_ARG(src && idx && dst && hidden % 8 == 0, "fl_ep_gather_rows: bad args");
  if (n == 0) return FL_OK;
  ep_rows_kernel<false><<<dim3((unsigned)((n + 3) / 4)), 256, 0, (hipStream_t)stream>>>(
      (const uint16_t*)src, idx, n, hidden, src_rows, dst_rows, (uint16_t*)dst);
  FL_CHECK_LAUNCH("fl_ep_gather_rows");
  return FL_OK;
}

extern "C" int fl_ep_scatter_rows(const void* src, int64_t src_rows, const int32_t* idx, int64_t n, int hidden, void* dst,
                                  int64_t dst_rows, fl_stream_t stream) {
  FL_CHECK_ARG(src && idx && dst && hidden % 8 == 0, "fl_ep_scatter_rows: bad args");
  if (n == 0) return FL_OK;
  ep_rows_kernel<true><<<dim3((unsigned)((n + 3) / 4)), 256, 0, (hipStream_t)stream>>>(
      (const uint16_t*)src, idx, n, hidden, src_rows, dst_rows, (uint16_t*)dst);
  FL_CHECK_LAUNCH("fl_ep_scatter_rows");
  return FL_OK;
}

extern "C" int fl_ep_combine(const void* ret_rows, int64_t num_ret_rows, const int32_t* send_slot, const float* weights,
                             int64_t num_tokens, int top_k, int hidden, void* out, fl_stream_t stream) {
  FL_CHECK_ARG(ret_rows && send_slot && weights && out && hidden % 8 == 0 && top_k >= 1, "fl_ep_combine: bad args");
  if (num_tokens == 0) return FL_OK;
  ep_combine_kernel<<<dim3((unsigned)((num_tokens + 3) / 4)), 256, 0, (hipStream_t)stream>>>(
      (const uint16_t*)ret_rows, send_slot, weights, num_tokens, top_k, hidden, num_ret_rows, (uint16_t*)out);
  FL_CHECK_LAUNCH("fl_ep_combine");
  return FL_OK;
}
