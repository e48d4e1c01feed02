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
#include "norm_row.h"

namespace {

using namespace fl_norm;

// One WORKGROUP (4 waves) per row; x = sum of num_pieces pieces (piece w at x + w*piece_stride).  A decode step has a few
// hundred rows: one wave per row leaves all but one wave slot per CU empty and serialises 14 dependent 16-B accesses per
// lane (measured 32 us for 256 x 7168); the row work itself is fl_norm::add_rmsnorm_row (norm_row.h).
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(const uint16_t* x, int num_pieces, long long piece_stride,
                                                          const uint16_t* __restrict__ add_in, const uint16_t* residual_in,
                                                          const uint16_t* __restrict__ gamma, float eps, long long T, int H,
                                                          uint16_t* residual_out, uint16_t* norm_out,
                                                          uint8_t* __restrict__ quant_out, float* __restrict__ scale_out,
                                                          long long ss_t, long long ss_g, float gamma_offset) {
  __shared__ float wsum[4];
  const long long row = blockIdx.x;
  add_rmsnorm_row(x + row * H, num_pieces, piece_stride, add_in, residual_in, gamma, eps, row, H, residual_out, norm_out,
                  quant_out, scale_out, ss_t, ss_g, wsum, gamma_offset);
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
  dual_rmsnorm_row(ag + row * D, row, q_rank, kv_rank, gamma_q, gamma_kv, eps_q, eps_kv, x_norm_out, quant_out, scale_out, ss_t,
                   ss_g, lane);
}

}  // namespace

static int launch_add_rmsnorm(const void* x, int num_pieces, int64_t piece_stride, const void* add_in, const void* residual_in,
                              const void* gamma, float eps, int64_t T, int H, void* residual_out, void* norm_out, void* quant_out,
                              float* scale_out, int64_t s_stride_t, int64_t s_stride_g, float gamma_offset, fl_stream_t stream) {
  if (T == 0) return FL_OK;   // (a rank without token rows — T < world — passes empty tensors: null data pointers)
  FL_CHECK_ARG(x && num_pieces >= 1 && T >= 0, "fl_fused_add_rmsnorm: bad arguments");
  FL_CHECK_ARG(gamma != nullptr || (norm_out == nullptr && quant_out == nullptr), "fl_fused_add_rmsnorm: norm needs gamma");
  FL_CHECK_ARG(H > 0 && H % 8 == 0 && H <= kMaxChunks * 512, "fl_fused_add_rmsnorm: H=%d (need H %% 8 == 0, H <= 8192)", H);
  FL_CHECK_ARG(quant_out == nullptr || (scale_out != nullptr && H % 128 == 0), "fl_fused_add_rmsnorm: quant needs scales, H %% 128 == 0");
  add_rmsnorm_kernel<<<dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream>>>(
      (const uint16_t*)x, num_pieces, piece_stride, (const uint16_t*)add_in, (const uint16_t*)residual_in,
      (const uint16_t*)gamma, eps, T, H, (uint16_t*)residual_out, (uint16_t*)norm_out, (uint8_t*)quant_out, scale_out,
      s_stride_t, s_stride_g, gamma_offset);
  FL_CHECK_LAUNCH("fl_fused_add_rmsnorm");
  return FL_OK;
}

extern "C" int fl_fused_add_rmsnorm(const void* x, int num_pieces, int64_t piece_stride, const void* add_in,
                                    const void* residual_in, const void* gamma, float eps, int64_t T, int H,
                                    void* residual_out, void* norm_out, void* quant_out, float* scale_out,
                                    int64_t s_stride_t, int64_t s_stride_g, fl_stream_t stream) {
  return launch_add_rmsnorm(x, num_pieces, piece_stride, add_in, residual_in, gamma, eps, T, H, residual_out, norm_out, quant_out,
                            scale_out, s_stride_t, s_stride_g, 0.f, stream);
}

// the Gemma form (layernorm.py:209-233: x * rsqrt(..) * (1 + w), the 1 added in fp32): same kernel, gamma_offset = 1
extern "C" int fl_fused_add_rmsnorm_offset(const void* x, int num_pieces, int64_t piece_stride, const void* add_in,
                                           const void* residual_in, const void* gamma, float gamma_offset, float eps, int64_t T, int H,
                                           void* residual_out, void* norm_out, fl_stream_t stream) {
  return launch_add_rmsnorm(x, num_pieces, piece_stride, add_in, residual_in, gamma, eps, T, H, residual_out, norm_out, nullptr, nullptr,
                            0, 0, gamma_offset, stream);
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
