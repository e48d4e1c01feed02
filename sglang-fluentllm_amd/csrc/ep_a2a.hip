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

// Token-once-per-peer routing: a token with several experts on one rank travels there ONCE (the receiving rank replicates
// the row to its experts locally), so a peer slab needs max_tokens rows, not max_tokens * top_k — 8x fewer xGMI bytes than
// one row per (token, expert) pair at top-8, and a token can never overflow a slab (it occupies at most one row of it).
//   tok_slot[t, d]      slab row (d*cap + position) of token t in rank d's slab, -1 if none of its experts lives there;
//                       positions follow the token order (deterministic)
//   send_eid[row, j]    local expert ids of the row's token on that rank (j-th of them in k order), -1 padded
//   pair_src[row, j]    t*top_k + k of the (token, expert) pair that occupies (row, j), -1 for padding: the combine
//                       weights travel in this layout (one gather, no zero-fill pass)
__global__ __launch_bounds__(256) void ep_route_dedup_kernel(const int32_t* __restrict__ indices, int T, int top_k,
                                                             int experts_per_rank, int world, int cap,
                                                             int32_t* __restrict__ tok_slot, int32_t* __restrict__ send_eid,
                                                             int32_t* __restrict__ pair_src, long long eid_stride,
                                                             const float* __restrict__ weights, float* __restrict__ send_w,
                                                             long long w_stride) {
  // (weights != nullptr: the routing weight of every pair is placed beside its expert id — send_w[row, j] = weights[t, k], 0 for
  //  padding — in the same pass: the dispatch message's tail is complete without a gather launch)
  // (eid_stride: ints between the id rows — top_k for a separate [rows, top_k] tensor, the message row for ids that travel in the
  //  TAIL of their slab row: one all-to-all instead of two)
  extern __shared__ unsigned long long s_mask[];   // [T] peers of every token (world <= 64)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long long i = threadIdx.x; i < (long long)world * cap * top_k; i += 256) {
    send_eid[(i / top_k) * eid_stride + i % top_k] = -1;
    pair_src[i] = -1;
    if (weights != nullptr) send_w[(i / top_k) * w_stride + i % top_k] = 0.f;
  }
  for (int t = threadIdx.x; t < T; t += 256) {
    unsigned long long m = 0;
    for (int k = 0; k < top_k; ++k) {
      const int e = indices[t * top_k + k];
      if (e >= 0 && e < experts_per_rank * world) m |= 1ull << (e / experts_per_rank);
    }
    s_mask[t] = m;
  }
  __syncthreads();
  // slab positions in token order: peer d is scanned by wave d % 4, 64 tokens per step (ballot + prefix popcount)
  for (int d = wave; d < world; d += 4) {
    int run = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
      const int t = t0 + lane;
      const bool on = t < T && ((s_mask[t < T ? t : 0] >> d) & 1ull);
      const unsigned long long b = __ballot(on);
      const int pos = run + __popcll(b & ((1ull << lane) - 1ull));
      if (t < T) tok_slot[t * world + d] = (on && pos < cap) ? d * cap + pos : -1;
      run += __popcll(b);
    }
  }
  __syncthreads();   // (tok_slot is re-read below by other threads of this workgroup: global writes of the workgroup are visible behind the barrier)
  for (int t = threadIdx.x; t < T; t += 256) {
    for (int k = 0; k < top_k; ++k) {
      const int e = indices[t * top_k + k];
      if (e >= 0 && e < experts_per_rank * world) {
        const int d = e / experts_per_rank;
        const int row = tok_slot[t * world + d];
        if (row >= 0) {
          int j = 0;
          for (int k2 = 0; k2 < k; ++k2) {
            const int e2 = indices[t * top_k + k2];
            j += (e2 >= 0 && e2 < experts_per_rank * world && e2 / experts_per_rank == d) ? 1 : 0;
          }
          send_eid[(long long)row * eid_stride + j] = e - d * experts_per_rank;
          pair_src[(long long)row * top_k + j] = t * top_k + k;
          if (weights != nullptr) send_w[(long long)row * w_stride + j] = weights[t * top_k + k];
        }
      }
    }
  }
}

// out[j] = vals[src[j]] where src[j] names a value (0 <= src[j] < n), else 0: the combine weights in the slab-pair layout
__global__ __launch_bounds__(256) void ep_gather_f32_kernel(const float* __restrict__ vals, long long n, const int32_t* __restrict__ src,
                                                            long long out_n, float* __restrict__ out, int per_row, long long out_stride) {
  const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
  if (j < out_n) {
    const long long i = src[j];
    out[(j / per_row) * out_stride + j % per_row] = (i >= 0 && i < n) ? vals[i] : 0.f;
  }
}

__global__ __launch_bounds__(256) void ep_sort_kernel(const int32_t* __restrict__ recv_eid_base, int S, int E,
                                                      int32_t* __restrict__ order, int32_t* __restrict__ exclusive_sum,
                                                      int32_t* __restrict__ inv /*optional: inv[order[i]] = i*/, int per_row,
                                                      long long eid_stride) {
  auto eid_at = [&](int i) { return recv_eid_base[(long long)(i / per_row) * eid_stride + i % per_row]; };
  extern __shared__ int bins[];   // E + 1 counters, then E + 1 cursors
  int* cur = bins + E + 1;
  for (int i = threadIdx.x; i <= E; i += 256) bins[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < S; i += 256) {
    const int e = eid_at(i);
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
    const int e = eid_at(i);
    const int pos = atomicAdd(&cur[(e >= 0 && e < E) ? e : E], 1);
    order[pos] = i;   // invalid slots end up after every valid row
    if (inv != nullptr) inv[i] = pos;
  }
}

// one workgroup per row (hidden*2 bytes), every thread's 16-B loads issued before its stores: a decode step moves a few
// hundred rows, so the depth of one row's copy, not the row count, sets the time
template <bool kScatter>
__global__ __launch_bounds__(256) void ep_rows_kernel(const uint16_t* __restrict__ src, const int32_t* __restrict__ idx,
                                                      long long n, int hidden, long long src_rows, long long dst_rows,
                                                      uint16_t* __restrict__ dst, int src_div,
                                                      const int32_t* __restrict__ n_valid = nullptr, long long src_stride = 0,
                                                      long long dst_stride = 0) {
  src_stride = src_stride > 0 ? src_stride : hidden;   // (elements between rows: a slab row may carry a tail behind its `hidden` elements)
  dst_stride = dst_stride > 0 ? dst_stride : hidden;
  const long long i = blockIdx.x;
  if (n_valid != nullptr && i >= *n_valid) return;   // (device-side row count: the static grid covers the worst case)
  const long long j = idx[i];
  // scatter: pair i carries token i / src_div (src_div = top_k); gather: index j names pair j of row j / src_div
  const long long s = kScatter ? (long long)((unsigned)i / (unsigned)src_div) : (j < 0 ? j : j / src_div), d = kScatter ? j : i;
  if (s < 0 || s >= src_rows || d < 0 || d >= dst_rows) return;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4* sp = reinterpret_cast<const u32x4*>(src + s * src_stride);
  u32x4* dp = reinterpret_cast<u32x4*>(dst + d * dst_stride);
  const int nc = hidden / 8;
  for (int c0 = threadIdx.x; c0 < nc; c0 += 4 * 256) {
    // loads are unconditional on a clamped column (a predicated load into an array makes hipcc stage it through LDS
    // with a full wait per load); only the stores are predicated
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = sp[c0 + u * 256 < nc ? c0 + u * 256 : c0];
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));   // all four loads in flight before the first store
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (c0 + u * 256 < nc) dp[c0 + u * 256] = v[u];
  }
}

// One WORKGROUP per token (a decode step has a few dozen tokens per rank: one wave per token looping over the hidden
// dimension with its slot / weight / row loads inside the loop was 112 dependent round trips, 62 us for 32 tokens).
// Slots and weights are staged once, the top_k row loads of a 16-B column group are issued together; the sum runs in
// expert order k = 0 .. top_k-1 in fp32 as before.
__global__ __launch_bounds__(256) void ep_combine_kernel(const uint16_t* __restrict__ ret, const int32_t* __restrict__ slot,
                                                         const float* __restrict__ w, long long t_count, int top_k,
                                                         int hidden, long long ret_rows, uint16_t* __restrict__ out,
                                                         long long w_stride) {
  __shared__ long long s_slot[64];
  __shared__ float s_w[64];
  const int tid = threadIdx.x;
  const long long t = blockIdx.x;
  for (int k0 = 0; k0 < top_k; k0 += 64) {   // (top_k > 64: in rounds; the partial sums stay in registers per column group)
    __syncthreads();
    if (tid < 64 && k0 + tid < top_k) {
      const long long sl = slot[t * top_k + k0 + tid];
      const float wv = w[t * w_stride + k0 + tid];
      // a zero weight skips the row altogether (0 x an uncomputed row must not turn into NaN: the padding pairs of a slab
      // row sort behind the last expert group, where the expert output buffer was never written)
      const bool ok = sl >= 0 && sl < ret_rows && wv != 0.f;
      s_slot[tid] = ok ? sl : -1;
      s_w[tid] = ok ? wv : 0.f;
    }
    __syncthreads();
    const int kn = top_k - k0 < 64 ? top_k - k0 : 64;
    // grid.y covers the row in chunks of 256 column groups: a decode step has few tokens, so the columns fill the chip
    for (int c = blockIdx.y * 256 + tid; c < hidden / 8; c += 256 * gridDim.y) {
      float acc[8];
      if (k0 == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
      } else {   // continue a previous round: read back what it stored (bf16-rounded only in the last round otherwise)
        const uint4 v = reinterpret_cast<const uint4*>(out + t * hidden)[c];
        const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc[2 * q] = __uint_as_float(u[q] << 16); acc[2 * q + 1] = __uint_as_float(u[q] & 0xffff0000u); }
      }
      for (int kb = 0; kb < kn; kb += 8) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // unconditional load (empty slot: row 0, skipped below) - see ep_rows_kernel
          const long long sl = kb + j < kn ? s_slot[kb + j] : -1;
          v[j] = reinterpret_cast<const uint4*>(ret + (sl >= 0 ? sl : 0) * hidden)[c];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (kb + j < kn && s_slot[kb + j] >= 0) {
            const float wk = s_w[kb + j];
            const uint32_t u[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              acc[2 * q] += wk * __uint_as_float(u[q] << 16);
              acc[2 * q + 1] += wk * __uint_as_float(u[q] & 0xffff0000u);
            }
          }
        }
      }
      reinterpret_cast<uint4*>(out + t * hidden)[c] = make_uint4(fl_pack_bf16(acc[0], acc[1]), fl_pack_bf16(acc[2], acc[3]),
                                                                fl_pack_bf16(acc[4], acc[5]), fl_pack_bf16(acc[6], acc[7]));
    }
  }
}

}  // namespace

extern "C" int fl_ep_route(const int32_t* indices, int64_t num_pairs, int experts_per_rank, int world, int cap,
                           int32_t* send_slot, int32_t* send_eid, fl_stream_t stream) {
  FL_CHECK_ARG((indices || num_pairs == 0) && (send_slot || num_pairs == 0) && send_eid, "fl_ep_route: null pointer");   // (an idle rank routes 0 pairs: empty tensors have null data pointers)
  FL_CHECK_ARG(world >= 1 && world <= 64 && cap >= 1 && experts_per_rank >= 1 && num_pairs >= 0, "fl_ep_route: bad sizes");
  ep_route_kernel<<<1, 256, 0, (hipStream_t)stream>>>(indices, (int)num_pairs, experts_per_rank, world, cap, send_slot, send_eid);
  FL_CHECK_LAUNCH("fl_ep_route");
  return FL_OK;
}

extern "C" int fl_ep_route_dedup(const int32_t* indices, int64_t num_tokens, int top_k, int experts_per_rank, int world, int cap,
                                 int32_t* tok_slot, int32_t* send_eid, int32_t* pair_src, int64_t eid_row_stride, const float* weights,
                                 float* send_w, int64_t w_row_stride, fl_stream_t stream) {
  FL_CHECK_ARG(weights == nullptr || (send_w != nullptr && (w_row_stride == 0 || w_row_stride >= top_k)), "fl_ep_route_dedup: bad weight outputs");
  FL_CHECK_ARG(eid_row_stride == 0 || eid_row_stride >= top_k, "fl_ep_route_dedup: eid_row_stride=%lld < top_k", (long long)eid_row_stride);
  FL_CHECK_ARG(send_eid && pair_src && (num_tokens == 0 || (indices && tok_slot)), "fl_ep_route_dedup: null pointer");
  FL_CHECK_ARG(world >= 1 && world <= 64 && cap >= 1 && experts_per_rank >= 1 && top_k >= 1 && num_tokens >= 0 &&
                   num_tokens <= 8192, "fl_ep_route_dedup: bad sizes (tokens per rank <= 8192, world <= 64)");
  ep_route_dedup_kernel<<<1, 256, (size_t)(num_tokens > 0 ? num_tokens : 1) * 8, (hipStream_t)stream>>>(
      indices, (int)num_tokens, top_k, experts_per_rank, world, cap, tok_slot, send_eid, pair_src,
      eid_row_stride > 0 ? eid_row_stride : top_k, weights, send_w, w_row_stride > 0 ? w_row_stride : top_k);
  FL_CHECK_LAUNCH("fl_ep_route_dedup");
  return FL_OK;
}

extern "C" int fl_ep_gather_f32(const float* vals, int64_t n, const int32_t* src, float* out, int64_t out_n, int per_row,
                                int64_t out_row_stride, fl_stream_t stream) {
  FL_CHECK_ARG(out_n >= 0 && n >= 0 && (out_n == 0 || (out && src)) && (n == 0 || vals), "fl_ep_gather_f32: bad args");
  FL_CHECK_ARG(per_row >= 1 && (out_row_stride == 0 || out_row_stride >= per_row), "fl_ep_gather_f32: bad row layout");
  if (out_n == 0) return FL_OK;
  ep_gather_f32_kernel<<<dim3((unsigned)((out_n + 255) / 256)), 256, 0, (hipStream_t)stream>>>(
      vals, n, src, out_n, out, per_row, out_row_stride > 0 ? out_row_stride : per_row);
  FL_CHECK_LAUNCH("fl_ep_gather_f32");
  return FL_OK;
}

extern "C" int fl_ep_gather_rows_div(const void* src, int64_t src_rows, const int32_t* idx, int64_t n, int div, int hidden,
                                     void* dst, int64_t dst_rows, const int32_t* n_valid, int64_t src_row_stride, fl_stream_t stream) {
  if (n == 0) return FL_OK;
  FL_CHECK_ARG(src && idx && dst && hidden % 8 == 0 && div >= 1, "fl_ep_gather_rows_div: bad args");
  FL_CHECK_ARG(src_row_stride == 0 || (src_row_stride >= hidden && src_row_stride % 8 == 0), "fl_ep_gather_rows_div: bad source row stride");
  ep_rows_kernel<false><<<dim3((unsigned)n), 256, 0, (hipStream_t)stream>>>(
      (const uint16_t*)src, idx, n, hidden, src_rows, dst_rows, (uint16_t*)dst, div, n_valid, src_row_stride, 0);
  FL_CHECK_LAUNCH("fl_ep_gather_rows_div");
  return FL_OK;
}

extern "C" int fl_ep_sort(const int32_t* recv_eid, int64_t num_slots, int num_local_experts, int32_t* order,
                          int32_t* exclusive_sum, int32_t* inverse, int per_row, int64_t eid_row_stride, fl_stream_t stream) {
  FL_CHECK_ARG(recv_eid && order && exclusive_sum, "fl_ep_sort: null pointer");
  FL_CHECK_ARG(num_local_experts >= 1 && num_local_experts <= 4096 && num_slots >= 0, "fl_ep_sort: bad sizes");
  FL_CHECK_ARG(per_row >= 1 && (eid_row_stride == 0 || eid_row_stride >= per_row), "fl_ep_sort: bad row layout");
  ep_sort_kernel<<<1, 256, 2 * (num_local_experts + 1) * sizeof(int), (hipStream_t)stream>>>(
      recv_eid, (int)num_slots, num_local_experts, order, exclusive_sum, inverse, per_row, eid_row_stride > 0 ? eid_row_stride : per_row);
  FL_CHECK_LAUNCH("fl_ep_sort");
  return FL_OK;
}

extern "C" int fl_ep_gather_rows(const void* src, int64_t src_rows, const int32_t* idx, int64_t n, int hidden, void* dst,
                                 int64_t dst_rows, fl_stream_t stream) {
  if (n == 0) return FL_OK;   // (before the pointer checks: empty tensors have null data pointers)
  FL_CHECK_ARG(src && idx && dst && hidden % 8 == 0, "fl_ep_gather_rows: bad args");
  ep_rows_kernel<false><<<dim3((unsigned)n), 256, 0, (hipStream_t)stream>>>(
      (const uint16_t*)src, idx, n, hidden, src_rows, dst_rows, (uint16_t*)dst, 1);
  FL_CHECK_LAUNCH("fl_ep_gather_rows");
  return FL_OK;
}

extern "C" int fl_ep_scatter_rows(const void* src, int64_t src_rows, const int32_t* idx, int64_t n, int hidden, void* dst,
                                  int64_t dst_rows, fl_stream_t stream) {
  if (n == 0) return FL_OK;
  FL_CHECK_ARG(src && idx && dst && hidden % 8 == 0, "fl_ep_scatter_rows: bad args");
  ep_rows_kernel<true><<<dim3((unsigned)n), 256, 0, (hipStream_t)stream>>>(
      (const uint16_t*)src, idx, n, hidden, src_rows, dst_rows, (uint16_t*)dst, 1);
  FL_CHECK_LAUNCH("fl_ep_scatter_rows");
  return FL_OK;
}

extern "C" int fl_ep_send_rows(const void* x, int64_t num_tokens, const int32_t* send_slot, int64_t num_pairs, int top_k,
                               int hidden, void* send_buf, int64_t send_rows, int64_t send_row_stride, fl_stream_t stream) {
  if (num_pairs == 0) return FL_OK;
  FL_CHECK_ARG(x && send_slot && send_buf && hidden % 8 == 0 && top_k >= 1, "fl_ep_send_rows: bad args");
  FL_CHECK_ARG(send_row_stride == 0 || (send_row_stride >= hidden && send_row_stride % 8 == 0), "fl_ep_send_rows: bad row stride");
  ep_rows_kernel<true><<<dim3((unsigned)num_pairs), 256, 0, (hipStream_t)stream>>>(
      (const uint16_t*)x, send_slot, num_pairs, hidden, num_tokens, send_rows, (uint16_t*)send_buf, top_k, nullptr, 0, send_row_stride);
  FL_CHECK_LAUNCH("fl_ep_send_rows");
  return FL_OK;
}

extern "C" int fl_ep_combine(const void* ret_rows, int64_t num_ret_rows, const int32_t* send_slot, const float* weights,
                             int64_t num_tokens, int top_k, int hidden, void* out, int64_t weight_row_stride, fl_stream_t stream) {
  if (num_tokens == 0) return FL_OK;
  FL_CHECK_ARG(ret_rows && send_slot && weights && out && hidden % 8 == 0 && top_k >= 1, "fl_ep_combine: bad args");
  FL_CHECK_ARG(weight_row_stride == 0 || weight_row_stride >= top_k, "fl_ep_combine: bad weight row stride");
  ep_combine_kernel<<<dim3((unsigned)num_tokens, (unsigned)((hidden / 8 + 255) / 256)), 256, 0, (hipStream_t)stream>>>(
      (const uint16_t*)ret_rows, send_slot, weights, num_tokens, top_k, hidden, num_ret_rows, (uint16_t*)out,
      weight_row_stride > 0 ? weight_row_stride : top_k);
  FL_CHECK_LAUNCH("fl_ep_combine");
  return FL_OK;
}
