// K1 / K2 dispatch — paged MLA decode over the FP8 latent KV cache, gfx950 (MI355X) only.
//
// Replaces flash_mla_fp8.flash_mla_ckv_fp8_per_token (call sites /root/reference/python/sglang/srt/layers/attention/flashmla_backend.py:208-222
// decode, :127-142 verify / draft-extend) and flash_mla_fp8.flash_mla_with_kvcache over a plain fp8 [.,576] cache (:227-239).
//
// Math (per request b, query row r = j*h_q + h, latent token t):
//   s[r,t]  = (q8[r,:]·k8[t,:] + qrope'[r,:]·krope'[t,:]) * q_scale[r] * k_scale[t] * softmax_scale
//             (rope is stored pre-divided by the scale on both sides: memory_pool.py:877)
//   o[r,:]  = sum_t softmax_t(s[r,:]) * k_scale[t] * k8[t,:512]          (V = dequantised latent)
//
// Both fp8 formats run on ONE kernel family, mla_decode_fp8_y.hip (role-specialised QK / PV waves, "SwapAB": tokens on the MFMA M side,
// query rows on the N side, one query row per lane): 64-row workgroups of 8 waves for more than 32 rows per request, 32-row workgroups
// of 4 waves (one per SIMD) for at most 32 — the TP8 shard's H = 16.  (Rounds 1-4 ran a separate kernel for <= 32 rows — two compute waves
// that each did QK, softmax and PV of a page in series, + two loader waves — probes/superseded/mla_decode_fp8_small_kernel.hip.txt; round 5
// measured the 4-wave instantiation of the role-specialised kernel faster at every shape: profiles/r05_k1_small_rows_nrt1.txt.)
// This file keeps the C-ABI entry point, the argument checks, the workspace sizing and the split-KV merge kernels (used when a request
// is cut into many pieces: small batches; a few pieces are merged inside the decode kernel).
//
// Algorithmic bytes per (request, layer call): seq*644 (KV) + s_q*h_q*(644 + 1024) (Q in, O out) + 4*ceil(seq/64).
#include "mla_decode_shared.h"
#include <cstdlib>

using namespace fl_mla;

namespace {

// ---- split-KV combine: out[req,row,:] = sum_s w_s * o_accum[slot_s,row,:], w_s = softmax_s(lse_s) ----
__global__ __launch_bounds__(256) void mla_combine_kernel(const Params p, const int32_t* __restrict__ g_num_splits) {
  // nothing is split (every request has exactly one part): one scalar load per workgroup and out
  if (g_num_splits[p.bs] == p.bs) return;
  const int lane = threadIdx.x & 63;
  // the wave index is uniform: say so, and the split counts / LSEs of the row come through the scalar cache
  const long long total = (long long)p.bs * p.rows;
  for (long long gw = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); gw < total;
       gw += (long long)gridDim.x * 4) {
    const int req = (int)(gw / p.rows), row = (int)(gw % p.rows);
    const int s0 = g_num_splits[req], ns = g_num_splits[req + 1] - s0;
    if (ns <= 1) continue;
    if (ns == 2) combine_row<2>(p, req, row, s0, ns, lane);   // the uniform full batch: two parts per request
    else if (ns == 3) combine_row<3>(p, req, row, s0, ns, lane);
    else if (ns == 4) combine_row<4>(p, req, row, s0, ns, lane);
    else combine_row<0>(p, req, row, s0, ns, lane);
  }
}

// few rows, many splits per row (decode at bs = 1..8): ONE row per workgroup, its splits over the four waves (combine_row QUAD)
__global__ __launch_bounds__(256) void mla_combine_quad_kernel(const Params p, const int32_t* __restrict__ g_num_splits) {
  __shared__ float red[3 * 64 * 10];
  if (g_num_splits[p.bs] == p.bs) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int req = (int)(blockIdx.x / p.rows), row = (int)(blockIdx.x % p.rows);
  const int s0 = g_num_splits[req], ns = g_num_splits[req + 1] - s0;
  if (ns <= 1) return;   // (uniform over the workgroup)
  combine_row<0, true>(p, req, row, s0, ns, lane, red, wave);
}

}  // namespace

int fl_mla_launch_combine(const Params& p, const int32_t* num_splits, hipStream_t stream) {
  const long long waves = (long long)p.bs * p.rows;
  // (requests x rows) below one wave per SIMD of the chip AND more parts than requests x 8 (every request is cut into many
  // pieces — the split counts themselves live on the device): a workgroup per row
  if (waves <= 1024 && p.num_parts >= 8 * p.bs) {
    mla_combine_quad_kernel<<<dim3((unsigned)waves), dim3(256), 0, stream>>>(p, num_splits);
    FL_CHECK_LAUNCH("mla_combine_quad_kernel");
    return FL_OK;
  }
  const long long blocks = (waves + 3) / 4;   // one wave per (request, row); grid-stride beyond 2048 workgroups
  mla_combine_kernel<<<dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, stream>>>(p, num_splits);
  FL_CHECK_LAUNCH("mla_combine_kernel");
  return FL_OK;
}

int fl_mla_decode_fp8_y_impl(const FlMlaDecodeArgs* a, const Params& p, hipStream_t stream);   // mla_decode_fp8_y.hip

// bytes of the split-partials workspaces fl_mla_decode needs for a launch of this shape: the fp8 formats' partials are bf16 rows
// (mla_decode_fp8_y.hip), the bf16 cache's f32 rows (mla_decode_bf16.hip)
extern "C" int fl_mla_workspace_bytes(int kv_format, int bs, int s_q, int h_q, int num_parts, int64_t* o_accum_bytes,
                                      int64_t* lse_accum_bytes) {
  FL_CHECK_ARG(o_accum_bytes && lse_accum_bytes && bs >= 0 && s_q >= 1 && h_q >= 1 && num_parts >= 1, "fl_mla_workspace_bytes: bad arguments");
  const long long rows = (long long)s_q * h_q;
  const bool bf16_partials = kv_format != FL_KV_BF16_576;
  *o_accum_bytes = (long long)(bs + num_parts) * rows * 512 * (bf16_partials ? 2 : 4);
  *lse_accum_bytes = (long long)(bs + num_parts) * rows * 2 * 4;
  return FL_OK;
}

int fl_mla_decode_fp8_impl(const FlMlaDecodeArgs* a, hipStream_t stream) {
  const bool per_token = a->kv_format == FL_KV_FP8_PER_TOKEN;
  FL_CHECK_ARG(a->d_nope == kDN && a->d_rope == kDR, "fl_mla_decode: only d_nope=512,d_rope=64 (got %d,%d)",
               a->d_nope, a->d_rope);
  const bool q_bf16 = a->q_bf16 != nullptr;
  FL_CHECK_ARG((q_bf16 || a->q_nope) && a->k_nope, "fl_mla_decode: null q/k pointer");
  FL_CHECK_ARG(!per_token || ((q_bf16 || (a->q_rope && a->q_scale)) && a->k_rope && a->k_scale),
               "fl_mla_decode(per-token fp8): null rope/scale pointer");
  FL_CHECK_ARG(!q_bf16 || (per_token && ((uintptr_t)a->q_bf16 % 16) == 0),
               "fl_mla_decode: q_bf16 (K4 inside the decode kernel) is served for the per-token format, 16-byte aligned");
  FL_CHECK_ARG(a->block_table && a->cache_seqlens && a->tile_scheduler_metadata && a->num_splits && a->out && a->lse &&
                   a->o_accum && a->lse_accum,
               "fl_mla_decode: null metadata/output pointer");
  FL_CHECK_ARG(a->bs >= 0 && a->s_q >= 1 && a->h_q >= 1 && a->num_parts >= 1, "fl_mla_decode: bad sizes");
  if (a->bs == 0) return FL_OK;
  Params p;
  p.bs = a->bs; p.s_q = a->s_q; p.h_q = a->h_q; p.rows = a->s_q * a->h_q; p.causal = a->causal;
  p.num_parts = a->num_parts;
  p.scale_log2e = a->softmax_scale * kLog2e;
  p.descale_q = a->descale_q; p.descale_k = a->descale_k;
  p.num_pages = a->num_pages; p.bt_stride = a->block_table_stride;
  p.bt_cols = a->block_table_cols > 0 ? a->block_table_cols : (a->block_table_stride > 0 ? a->block_table_stride : 1);
  p.out = (uint16_t*)a->out; p.lse = a->lse; p.o_accum = a->o_accum; p.lse_accum = a->lse_accum;
  p.partial_bf16 = 0;
  p.merge_in_kernel = 0;
  p.q_bf16 = (const uint16_t*)a->q_bf16;
  // (fl_mla_num_parts sizes the scheduler's part count with the kernel's workgroup rule: CUs / ceil(rows / 64) parts)
  return fl_mla_decode_fp8_y_impl(a, p, stream);
}

// ---- fl_mla_decode — C-ABI dispatch over the KV-cache formats of MLATokenToKVPool (memory_pool.py:635-658) ----
int fl_mla_decode_bf16_impl(const FlMlaDecodeArgs* a, hipStream_t stream);   // mla_decode_bf16.hip

extern "C" int fl_mla_decode(const FlMlaDecodeArgs* args, fl_stream_t stream) {
  FL_CHECK_ARG(args != nullptr, "fl_mla_decode: null args");
  FL_CHECK_ARG(args->struct_bytes == (int32_t)sizeof(FlMlaDecodeArgs),
               "fl_mla_decode: FlMlaDecodeArgs.struct_bytes = %d, this library (ABI %d) expects %d — caller and library were built against "
               "different include/fluent_mi355.h", args->struct_bytes, FL_ABI_VERSION, (int)sizeof(FlMlaDecodeArgs));
  switch (args->kv_format) {
    case FL_KV_FP8_PER_TOKEN:
    case FL_KV_FP8_576:
      return fl_mla_decode_fp8_impl(args, (hipStream_t)stream);
    case FL_KV_BF16_576:
      return fl_mla_decode_bf16_impl(args, (hipStream_t)stream);
    default:
      fl_set_error("fl_mla_decode: kv_format %d not implemented", args->kv_format);
      return FL_ERR_UNSUPPORTED;
  }
}
