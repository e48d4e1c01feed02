// R1 — DeepSeek-V3 router selection: flashinfer.moe_fused_gate as called by biased_grouped_topk_gpu
// (python/sglang/srt/layers/moe/topk.py:709-733); semantics = its torch statement biased_grouped_topk_impl
// (topk.py:596-663) with renormalize=True, num_fused_shared_experts=0:
//   s = sigmoid(logits); c = s + bias; group score = sum of the two largest c of a group; keep the topk_group best groups;
//   topk experts by c among the kept groups; weight = s (unbiased) / sum of the chosen s (* routed_scaling_factor).
// One wave per token: lane l holds experts [l*V, l*V + V) (V = E/64), a group is 64/num_expert_group neighbouring lanes;
// every step is a wave shuffle, no LDS.  Ties resolve to the lower index (torch.topk leaves them unspecified).
// Output rows are ordered by descending choice score.  Latency-bound (decode: a few hundred tokens).
#include "fl_common.h"

namespace {
constexpr int kMaxV = 16;   // experts per lane: E <= 1024

template <int V>
__global__ __launch_bounds__(256) void moe_gate_kernel(const float* __restrict__ logits, const float* __restrict__ bias,
                                                       long long T, int n_group, int topk_group, int topk, float out_scale,
                                                       const int* __restrict__ num_token_non_padded,
                                                       float* __restrict__ w_out, int* __restrict__ id_out) {
  const int lane = threadIdx.x & 63;
  const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const int E = V * 64;
  float s[V], c[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const float x = logits[t * E + lane * V + i];
    s[i] = 1.f / (1.f + expf(-x));
    c[i] = s[i] + bias[lane * V + i];
  }
  // group score: top-2 sum over the group's lanes
  float m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const float v = c[i];
    m2 = v > m1 ? m1 : (v > m2 ? v : m2);
    m1 = v > m1 ? v : m1;
  }
  const int lpg = 64 / n_group;   // lanes per group
  for (int o = 1; o < lpg; o <<= 1) {
    const float b1 = __shfl_xor(m1, o), b2 = __shfl_xor(m2, o);
    const float hi = fmaxf(m1, b1), lo = fminf(m1, b1);
    m2 = fmaxf(lo, fmaxf(m2, b2));
    m1 = hi;
  }
  const float gscore = (V * lpg >= 2) ? m1 + m2 : m1;
  const int my_group = lane / lpg;
  int rank = 0;
  for (int g = 0; g < n_group; ++g) {
    const float o = __shfl(gscore, g * lpg);
    rank += (o > gscore || (o == gscore && g < my_group)) ? 1 : 0;
  }
  const bool keep = rank < topk_group;
#pragma unroll
  for (int i = 0; i < V; ++i) c[i] = keep ? c[i] : -INFINITY;
  // top-k: k rounds of a wave arg-max (value desc, index asc); lane j keeps the j-th pick
  float my_w = 0.f, sum = 0.f;
  int my_id = -1;
  for (int j = 0; j < topk; ++j) {
    float bv = -INFINITY, bs = 0.f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const bool better = c[i] > bv;   // ascending i: the first of equal values stays
      bv = better ? c[i] : bv;
      bi = better ? lane * V + i : bi;
      bs = better ? s[i] : bs;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o), os = __shfl_xor(bs, o);
      const int oi = __shfl_xor(bi, o);
      const bool take = ov > bv || (ov == bv && oi < bi);
      bv = take ? ov : bv; bi = take ? oi : bi; bs = take ? os : bs;
    }
    if (bi == 0x7fffffff) bi = -1;   // fewer candidates than topk
#pragma unroll
    for (int i = 0; i < V; ++i) c[i] = (lane * V + i == bi) ? -INFINITY : c[i];
    if (lane == j) { my_w = bs; my_id = bi; }
    sum += bi >= 0 ? bs : 0.f;
  }
  if (lane < topk) {
    const bool padded = num_token_non_padded != nullptr && t >= (long long)*num_token_non_padded;
    w_out[t * topk + lane] = my_w / sum * out_scale;
    // (fewer finite candidates than topk — NaN logits, -inf biases: the slot gets weight 0 and a VALID id (its own slot number), never an
    //  index a consumer could read out of bounds with; padded rows keep the reference's -1)
    id_out[t * topk + lane] = padded ? -1 : (my_id >= 0 ? my_id : lane);
  }
}

// R1b — plain (ungrouped) top-k routers of the same file: flashinfer.topk_softmax (fused_topk, topk.py:505-520; torch statement
// fused_topk_torch_native :463-495), flashinfer.routing_flash (LongCat-Flash, topk.py:836-845; torch statement fused_topk_bias
// :51-70) and eps' topk_sigmoid (resolved at topk.py:44-47).  score = softmax / sigmoid of the row (fp32), choice = score (+ bias),
// top-k by choice (value descending, expert id ascending on ties), weight = the UNBIASED score, optionally divided by the sum of
// the chosen scores, times `scale`.  Same one-wave-per-token shuffle structure as the grouped kernel; any E <= 1024.
template <int V>
__global__ __launch_bounds__(256) void topk_gate_kernel(const float* __restrict__ logits, const float* __restrict__ bias, long long T,
                                                        int E, int topk, int sigmoid, int renorm, float scale,
                                                        float* __restrict__ w_out, int* __restrict__ id_out) {
  const int lane = threadIdx.x & 63;
  const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  float s[V], c[V];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int e = lane * V + i;
    s[i] = e < E ? logits[t * E + e] : -INFINITY;
    mx = fmaxf(mx, s[i]);
  }
  if (sigmoid) {
#pragma unroll
    for (int i = 0; i < V; ++i) s[i] = lane * V + i < E ? 1.f / (1.f + expf(-s[i])) : 0.f;
  } else {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float den = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      s[i] = lane * V + i < E ? expf(s[i] - mx) : 0.f;
      den += s[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) den += __shfl_xor(den, o);
#pragma unroll
    for (int i = 0; i < V; ++i) s[i] = s[i] / den;
  }
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int e = lane * V + i;
    c[i] = e < E ? s[i] + (bias != nullptr ? bias[e] : 0.f) : -INFINITY;
  }
  float my_w = 0.f, sum = 0.f;
  int my_id = -1;
  for (int j = 0; j < topk; ++j) {
    float bv = -INFINITY, bs = 0.f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const bool better = c[i] > bv;
      bv = better ? c[i] : bv;
      bi = better ? lane * V + i : bi;
      bs = better ? s[i] : bs;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o), os = __shfl_xor(bs, o);
      const int oi = __shfl_xor(bi, o);
      const bool take = ov > bv || (ov == bv && oi < bi);
      bv = take ? ov : bv; bi = take ? oi : bi; bs = take ? os : bs;
    }
    if (bi == 0x7fffffff) bi = -1;
#pragma unroll
    for (int i = 0; i < V; ++i) c[i] = (lane * V + i == bi) ? -INFINITY : c[i];
    if (lane == j) { my_w = bs; my_id = bi; }
    sum += bi >= 0 ? bs : 0.f;
  }
  if (lane < topk) {
    w_out[t * topk + lane] = (renorm ? my_w / sum : my_w) * scale;
    id_out[t * topk + lane] = my_id >= 0 ? my_id : lane;   // (no finite candidate left: weight 0, a valid id)
  }
}
}  // namespace

extern "C" int fl_moe_fused_gate(const float* logits, const float* bias, int64_t num_tokens, int num_experts,
                                 int num_expert_group, int topk_group, int topk, float routed_scaling_factor,
                                 int apply_routed_scaling_factor_on_output, const int32_t* num_token_non_padded,
                                 float* topk_weights, int32_t* topk_ids, fl_stream_t stream) {
  FL_CHECK_ARG(logits && bias && topk_weights && topk_ids, "fl_moe_fused_gate: null pointer");
  FL_CHECK_ARG(num_tokens >= 0 && num_experts >= 64 && num_experts % 64 == 0 && num_experts / 64 <= kMaxV &&
                   ((num_experts / 64) & (num_experts / 64 - 1)) == 0,
               "fl_moe_fused_gate: num_experts=%d must be 64 * 2^k <= %d", num_experts, 64 * kMaxV);
  FL_CHECK_ARG(num_expert_group >= 1 && num_expert_group <= 64 && (num_expert_group & (num_expert_group - 1)) == 0,
               "fl_moe_fused_gate: num_expert_group=%d must be a power of two <= 64", num_expert_group);
  FL_CHECK_ARG(topk_group >= 1 && topk_group <= num_expert_group && topk >= 1 && topk <= 64 &&
                   topk <= topk_group * (num_experts / num_expert_group),
               "fl_moe_fused_gate: topk=%d topk_group=%d out of range", topk, topk_group);
  if (num_tokens == 0) return FL_OK;
  const float out_scale = apply_routed_scaling_factor_on_output ? routed_scaling_factor : 1.f;
  const dim3 grid((unsigned)((num_tokens + 3) / 4)), block(256);
#define FL_GATE(V_)                                                                                                     \
  moe_gate_kernel<V_><<<grid, block, 0, (hipStream_t)stream>>>(logits, bias, num_tokens, num_expert_group, topk_group, \
                                                                topk, out_scale, num_token_non_padded, topk_weights,    \
                                                                topk_ids)
  switch (num_experts / 64) {
    case 1: FL_GATE(1); break;
    case 2: FL_GATE(2); break;
    case 4: FL_GATE(4); break;
    case 8: FL_GATE(8); break;
    default: FL_GATE(16); break;
  }
#undef FL_GATE
  FL_CHECK_LAUNCH("fl_moe_fused_gate");
  return FL_OK;
}

extern "C" int fl_topk_gate(const float* logits, const float* bias, int64_t num_tokens, int num_experts, int topk, int score_fn,
                            int renormalize, float scale, float* topk_weights, int32_t* topk_ids, fl_stream_t stream) {
  FL_CHECK_ARG(logits && topk_weights && topk_ids, "fl_topk_gate: null pointer");
  FL_CHECK_ARG(num_tokens >= 0 && num_experts >= 1 && num_experts <= 64 * kMaxV, "fl_topk_gate: num_experts=%d must be in 1..%d",
               num_experts, 64 * kMaxV);
  FL_CHECK_ARG(topk >= 1 && topk <= 64 && topk <= num_experts, "fl_topk_gate: topk=%d out of range", topk);
  FL_CHECK_ARG(score_fn == 0 || score_fn == 1, "fl_topk_gate: score_fn must be 0 (softmax) or 1 (sigmoid)");
  if (num_tokens == 0) return FL_OK;
  const dim3 grid((unsigned)((num_tokens + 3) / 4)), block(256);
  const int v = (num_experts + 63) / 64;
#define FL_TOPK(V_)                                                                                                       \
  topk_gate_kernel<V_><<<grid, block, 0, (hipStream_t)stream>>>(logits, bias, num_tokens, num_experts, topk, score_fn,    \
                                                                 renormalize ? 1 : 0, scale, topk_weights, topk_ids)
  if (v <= 1) FL_TOPK(1);
  else if (v <= 2) FL_TOPK(2);
  else if (v <= 4) FL_TOPK(4);
  else if (v <= 8) FL_TOPK(8);
  else FL_TOPK(16);
#undef FL_TOPK
  FL_CHECK_LAUNCH("fl_topk_gate");
  return FL_OK;
}
