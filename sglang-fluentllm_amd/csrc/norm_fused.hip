// C5/C6/C7 — the compute half of flashinfer.comm.trtllm_{allreduce,reducescatter,allgather}_fusion
// (call sites /root/reference/python/sglang/srt/layers/flashinfer_comm_fusion.py:286-401, 404-513, 516-640 <-
// layernorm.py:114-189, 305-359).  MI355X mapping: xGMI is a point-to-point mesh, so the exchange is ONE one-shot
// collective over RCCL (every rank receives its peers' pieces; host side: fluent_mi355/comm.py) and everything after it —
// the reduction over the received pieces, add_in, residual add, RMSNorm and the optional 1x128 FP8 block quantisation —
// is ONE HBM-bound kernel here (one workgroup per token row, 16-B vector accesses, fp32 math).
//
// RMSNorm math = RMSNorm.forward_native (layernorm.py:88-112): x32 = x + residual (fp32); residual_out = bf16(x32);
// y = bf16(x32 * rsqrt(mean(x32^2) + eps) * weight).  Quantisation = fl_quant_1x128 of the bf16 y (what the unfused
// norm -> sgl_per_token_group_quant_fp8 pipeline produces).
#include "fl_common.h"

namespace {

__device__ __forceinline__ void unpack8(const uint4 raw, float (&v)[8]) {
  v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xffff0000u);
  v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xffff0000u);
  v[4] = __uint_as_float(raw.z << 16); v[5] = __uint_as_float(raw.z & 0xffff0000u);
  v[6] = __uint_as_float(raw.w << 16); v[7] = __uint_as_float(raw.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  uint4 o;
  o.x = (uint32_t)fl_f32_to_bf16(v[0]) | ((uint32_t)fl_f32_to_bf16(v[1]) << 16);
  o.y = (uint32_t)fl_f32_to_bf16(v[2]) | ((uint32_t)fl_f32_to_bf16(v[3]) << 16);
  o.z = (uint32_t)fl_f32_to_bf16(v[4]) | ((uint32_t)fl_f32_to_bf16(v[5]) << 16);
  o.w = (uint32_t)fl_f32_to_bf16(v[6]) | ((uint32_t)fl_f32_to_bf16(v[7]) << 16);
  return o;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// y (bf16-rounded, as floats) -> e4m3 with the 1x128 scale of its 16-lane group
__device__ __forceinline__ uint2 quant_group(const float (&y)[8], float& s_out) {
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(y[i]));
  amax = group16_max(amax);
  const float eps = fl_bf16_to_f32(fl_f32_to_bf16(1e-10f));
  const float s = fmaxf(amax, eps) / FL_FP8_MAX;
  s_out = s;
  return fl_div8_to_fp8<true>(y, s);   // y / s (IEEE), clamp to +-448, e4m3 (fl_common.h)
}

constexpr int kMaxChunks = 16;   // H <= 8192
constexpr int kRowChunks = 4;    // chunks of 256 threads x 8 elements per thread

// One WORKGROUP (4 waves) per row; H % 8 == 0, H <= 8192.  x = sum of num_pieces pieces (piece w at x + w*piece_stride).
// A decode step has a few hundred rows: one wave per row leaves all but one wave slot per CU empty and serialises 14
// dependent 16-B accesses per lane (measured 32 us for 256 x 7168).  Here a thread owns <= 4 groups of 8 elements, every
// load of the row is issued before the first use, and the four waves meet once for the sum of squares.
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(const uint16_t* __restrict__ x, int num_pieces,
                                                          long long piece_stride, const uint16_t* __restrict__ add_in,
                                                          const uint16_t* __restrict__ residual_in,
                                                          const uint16_t* __restrict__ gamma, float eps, long long T, int H,
                                                          uint16_t* __restrict__ residual_out, uint16_t* __restrict__ norm_out,
                                                          uint8_t* __restrict__ quant_out, float* __restrict__ scale_out,
                                                          long long ss_t, long long ss_g) {
  __shared__ float wsum[4];
  const int tid = threadIdx.x;
  const long long row = blockIdx.x;
  float v[kRowChunks][8];
  uint4 gr[kRowChunks];
  float ssq = 0.f;
  {
    uint4 r0[kRowChunks], ra[kRowChunks], rr[kRowChunks];
#pragma unroll
    for (int c = 0; c < kRowChunks; ++c) {
      const int col = (c * 256 + tid) * 8;
      r0[c] = ra[c] = rr[c] = gr[c] = make_uint4(0, 0, 0, 0);
      if (col < H) {
        r0[c] = *reinterpret_cast<const uint4*>(x + row * H + col);
        if (add_in != nullptr) ra[c] = *reinterpret_cast<const uint4*>(add_in + row * H + col);
        if (residual_in != nullptr) rr[c] = *reinterpret_cast<const uint4*>(residual_in + row * H + col);
        if (gamma != nullptr) gr[c] = *reinterpret_cast<const uint4*>(gamma + col);
      }
    }
#pragma unroll
    for (int c = 0; c < kRowChunks; ++c) {
      const int col = (c * 256 + tid) * 8;
      float acc[8], t[8];
      unpack8(r0[c], acc);
      for (int w = 1; w < num_pieces; ++w) {   // (same summation order as before: pieces, add_in, residual)
        if (col < H) {
          unpack8(*reinterpret_cast<const uint4*>(x + w * piece_stride + row * H + col), t);
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] += t[i];
        }
      }
      unpack8(ra[c], t);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += t[i];
      unpack8(rr[c], t);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += t[i];
      if (col < H && residual_out != nullptr)
        *reinterpret_cast<uint4*>(residual_out + row * H + col) =
            make_uint4(fl_pack_bf16(acc[0], acc[1]), fl_pack_bf16(acc[2], acc[3]), fl_pack_bf16(acc[4], acc[5]),
                       fl_pack_bf16(acc[6], acc[7]));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v[c][i] = acc[i];
        ssq += acc[i] * acc[i];   // (columns beyond H hold zeros)
      }
    }
  }
  if (norm_out == nullptr && quant_out == nullptr) return;   // sum-only use (one-shot reduce-scatter, C3)
  ssq = wave_sum(ssq);
  if ((tid & 63) == 0) wsum[tid >> 6] = ssq;
  __syncthreads();
  ssq = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
  const float rinv = rsqrtf(ssq / (float)H + eps);
#pragma unroll
  for (int c = 0; c < kRowChunks; ++c) {
    const int col = (c * 256 + tid) * 8;
    if (col < H) {
      float g[8], y[8];
      unpack8(gr[c], g);
      const uint32_t p0 = fl_pack_bf16(v[c][0] * rinv * g[0], v[c][1] * rinv * g[1]);
      const uint32_t p1 = fl_pack_bf16(v[c][2] * rinv * g[2], v[c][3] * rinv * g[3]);
      const uint32_t p2 = fl_pack_bf16(v[c][4] * rinv * g[4], v[c][5] * rinv * g[5]);
      const uint32_t p3 = fl_pack_bf16(v[c][6] * rinv * g[6], v[c][7] * rinv * g[7]);
      if (norm_out != nullptr) *reinterpret_cast<uint4*>(norm_out + row * H + col) = make_uint4(p0, p1, p2, p3);
      if (quant_out != nullptr) {   // H % 128 == 0 checked by the host; a 128-column group = 16 consecutive threads
        unpack8(make_uint4(p0, p1, p2, p3), y);
        float s;
        const uint2 q = quant_group(y, s);
        *reinterpret_cast<uint2*>(quant_out + row * H + col) = q;
        if ((tid & 15) == 0) scale_out[row * ss_t + (col >> 7) * ss_g] = s;
      }
    }
  }
}

// C7: dual RMSNorm over the gathered [T, D] rows: cols [0, q_rank) -> x_norm_out (separate tensor, optional fp8 quant),
// cols [q_rank, q_rank + kv_rank) normalised IN PLACE (the reference aliases y_norm_out onto allgather_out).
__global__ __launch_bounds__(256) void dual_rmsnorm_kernel(uint16_t* __restrict__ ag, long long T, int D, int q_rank,
                                                           int kv_rank, const uint16_t* __restrict__ gamma_q,
                                                           const uint16_t* __restrict__ gamma_kv, float eps_q, float eps_kv,
                                                           uint16_t* __restrict__ x_norm_out, uint8_t* __restrict__ quant_out,
                                                           float* __restrict__ scale_out, long long ss_t, long long ss_g) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  uint16_t* r = ag + row * D;
  // ---- q part (q_rank <= 2048) ----
  {
    float v[4][8];
    float ssq = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = c * 512 + lane * 8;
      if (col < q_rank) {
        unpack8(*reinterpret_cast<const uint4*>(r + col), v[c]);
#pragma unroll
        for (int i = 0; i < 8; ++i) ssq += v[c][i] * v[c][i];
      }
    }
    ssq = wave_sum(ssq);
    const float rinv = rsqrtf(ssq / (float)q_rank + eps_q);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = c * 512 + lane * 8;
      if (col < q_rank) {
        float g[8], y[8];
        unpack8(*reinterpret_cast<const uint4*>(gamma_q + col), g);
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] = fl_bf16_to_f32(fl_f32_to_bf16(v[c][i] * rinv * g[i]));
        if (x_norm_out != nullptr) *reinterpret_cast<uint4*>(x_norm_out + row * q_rank + col) = pack8(y);
        if (quant_out != nullptr) {
          float s;
          const uint2 q = quant_group(y, s);
          *reinterpret_cast<uint2*>(quant_out + row * q_rank + col) = q;
          if ((lane & 15) == 0) scale_out[row * ss_t + (col >> 7) * ss_g] = s;
        }
      }
    }
  }
  // ---- kv part (kv_rank <= 1024), in place ----
  {
    float v[2][8];
    float ssq = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int col = c * 512 + lane * 8;
      if (col < kv_rank) {
        unpack8(*reinterpret_cast<const uint4*>(r + q_rank + col), v[c]);
#pragma unroll
        for (int i = 0; i < 8; ++i) ssq += v[c][i] * v[c][i];
      }
    }
    ssq = wave_sum(ssq);
    const float rinv = rsqrtf(ssq / (float)kv_rank + eps_kv);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int col = c * 512 + lane * 8;
      if (col < kv_rank) {
        float g[8], y[8];
        unpack8(*reinterpret_cast<const uint4*>(gamma_kv + col), g);
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] = v[c][i] * rinv * g[i];
        *reinterpret_cast<uint4*>(r + q_rank + col) = pack8(y);
      }
    }
  }
}

}  // namespace

extern "C" int fl_fused_add_rmsnorm(const void* x, int num_pieces, int64_t piece_stride, const void* add_in,
                                    const void* residual_in, const void* gamma, float eps, int64_t T, int H,
                                    void* residual_out, void* norm_out, void* quant_out, float* scale_out,
                                    int64_t s_stride_t, int64_t s_stride_g, fl_stream_t stream) {
  if (T == 0) return FL_OK;   // (a rank without token rows — T < world — passes empty tensors: null data pointers)
  FL_CHECK_ARG(x && num_pieces >= 1 && T >= 0, "fl_fused_add_rmsnorm: bad arguments");
  FL_CHECK_ARG(gamma != nullptr || (norm_out == nullptr && quant_out == nullptr), "fl_fused_add_rmsnorm: norm needs gamma");
  FL_CHECK_ARG(H > 0 && H % 8 == 0 && H <= kMaxChunks * 512, "fl_fused_add_rmsnorm: H=%d (need H %% 8 == 0, H <= 8192)", H);
  FL_CHECK_ARG(quant_out == nullptr || (scale_out != nullptr && H % 128 == 0), "fl_fused_add_rmsnorm: quant needs scales, H %% 128 == 0");
  if (T == 0) return FL_OK;
  add_rmsnorm_kernel<<<dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream>>>(
      (const uint16_t*)x, num_pieces, piece_stride, (const uint16_t*)add_in, (const uint16_t*)residual_in,
      (const uint16_t*)gamma, eps, T, H, (uint16_t*)residual_out, (uint16_t*)norm_out, (uint8_t*)quant_out, scale_out,
      s_stride_t, s_stride_g);
  FL_CHECK_LAUNCH("fl_fused_add_rmsnorm");
  return FL_OK;
}

extern "C" int fl_dual_rmsnorm(void* ag, int64_t T, int D, int q_rank, int kv_rank, const void* gamma_q,
                               const void* gamma_kv, float eps_q, float eps_kv, void* x_norm_out, void* quant_out,
                               float* scale_out, int64_t s_stride_t, int64_t s_stride_g, fl_stream_t stream) {
  if (T == 0) return FL_OK;
  FL_CHECK_ARG(ag && gamma_q && gamma_kv && T >= 0, "fl_dual_rmsnorm: bad arguments");
  FL_CHECK_ARG(q_rank > 0 && q_rank % 8 == 0 && q_rank <= 2048 && kv_rank > 0 && kv_rank % 8 == 0 && kv_rank <= 1024 &&
                   q_rank + kv_rank <= D && D % 8 == 0,
               "fl_dual_rmsnorm: D=%d q_rank=%d kv_rank=%d", D, q_rank, kv_rank);
  FL_CHECK_ARG(quant_out == nullptr || (scale_out != nullptr && q_rank % 128 == 0), "fl_dual_rmsnorm: quant needs scales");
  if (T == 0) return FL_OK;
  dual_rmsnorm_kernel<<<dim3((unsigned)((T + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
      (uint16_t*)ag, T, D, q_rank, kv_rank, (const uint16_t*)gamma_q, (const uint16_t*)gamma_kv, eps_q, eps_kv,
      (uint16_t*)x_norm_out, (uint8_t*)quant_out, scale_out, s_stride_t, s_stride_g);
  FL_CHECK_LAUNCH("fl_dual_rmsnorm");
  return FL_OK;
}
