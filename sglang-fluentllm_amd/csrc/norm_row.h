// The per-row work shared by the fused-norm kernels (norm_fused.hip: after an RCCL exchange; comm_oneshot.hip: after the
// peer-mapped one-shot exchange): x = sum of pieces (+ add_in) (+ residual) in fp32 -> residual_out, RMSNorm -> norm_out,
// optional 1x128 e4m3 quantisation.  Math = RMSNorm.forward_native (layernorm.py:88-112), see norm_fused.hip.
#pragma once
#include "fl_common.h"

namespace fl_norm {

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

// One WORKGROUP (256 threads) per row; H % 8 == 0, H <= 8192.  The row of piece w is at xrow + w*piece_stride; add_in /
// residual_in / the outputs are indexed by `row`.  `wsum`: 4 floats of LDS.  A thread owns <= 4 groups of 8 elements,
// every load of the row is issued before the first use, and the four waves meet once for the sum of squares.
// (xrow / residual_in may ALIAS norm_out / residual_out — flashinfer's in-place fused_add_rmsnorm: no __restrict__ on those four; every thread
//  reads its own elements of a row before it writes them.  gamma_offset: 1.0 for the Gemma form x * (1 + w), added in fp32 as the reference does.)
__device__ __forceinline__ void add_rmsnorm_row(const uint16_t* xrow, const int num_pieces, const long long piece_stride,
                                                const uint16_t* __restrict__ add_in, const uint16_t* residual_in,
                                                const uint16_t* __restrict__ gamma, const float eps, const long long row, const int H,
                                                uint16_t* residual_out, uint16_t* norm_out,
                                                uint8_t* __restrict__ quant_out, float* __restrict__ scale_out,
                                                const long long ss_t, const long long ss_g, float* __restrict__ wsum,
                                                const float gamma_offset = 0.f) {
  const int tid = threadIdx.x;
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
        r0[c] = *reinterpret_cast<const uint4*>(xrow + col);
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
          unpack8(*reinterpret_cast<const uint4*>(xrow + w * piece_stride + col), t);
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
      if (gamma_offset != 0.f) {
#pragma unroll
        for (int i = 0; i < 8; ++i) g[i] += gamma_offset;
      }
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

// C7's row work (trtllm_allgather_fusion's dual RMSNorm, flashinfer_comm_fusion.py:613-638 <- layernorm.py:305-359) by ONE
// WAVE: `r` = the gathered row [D]; cols [0, q_rank) -> x_norm_out row (separate tensor, optional 1x128 fp8 quant), cols
// [q_rank, q_rank + kv_rank) normalised IN PLACE (the reference aliases y_norm_out onto allgather_out).  q_rank <= 2048,
// kv_rank <= 1024.  Shared by norm_fused.hip (after an RCCL gather) and comm_oneshot.hip (after the peer-mapped gather).
__device__ __forceinline__ void dual_rmsnorm_row(uint16_t* __restrict__ r, const long long row, const int q_rank, const int kv_rank,
                                                 const uint16_t* __restrict__ gamma_q, const uint16_t* __restrict__ gamma_kv,
                                                 const float eps_q, const float eps_kv, uint16_t* __restrict__ x_norm_out,
                                                 uint8_t* __restrict__ quant_out, float* __restrict__ scale_out, const long long ss_t,
                                                 const long long ss_g, const int lane) {
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

}  // namespace fl_norm
