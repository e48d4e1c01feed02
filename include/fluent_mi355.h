/* fluent_mi355.h — C-ABI of libfluent_mi355.so (MI355X / gfx950 native hot path for
 * SGLang-FluentLLM: FP8 MLA decode + FP8 block-scaled grouped GEMM).
 *
 * Every entry point below is what the reference's Python operator API for this path binds to
 * (citations are /root/reference/python/sglang/... file:line of the call site replaced).
 * Conventions: plain pointers to DEVICE memory unless the name says host; sizes in elements;
 * `stream` is a hipStream_t passed as void*; return 0 = ok, non-zero = error (message via
 * fl_last_error()).  Compute entry points neither synchronise the host nor allocate device memory: they
 * are hipGraph-capturable (srt/model_executor/cuda_graph_runner.py:433).  Exceptions, all named: the
 * one-shot communicator's setup / teardown / fl_comm_check (fl_comm_create .. fl_comm_destroy allocate,
 * map and synchronise — called outside the captured region), and the ONE piece of process-global
 * state, fl_gemm_set_num_cus (deep_gemm.set_num_sms is process-global in the reference as well);
 * besides that only a thread-local error string and a cached CU count are kept.
 */
#ifndef FLUENT_MI355_H
#define FLUENT_MI355_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* fl_stream_t;

#define FL_OK 0
#define FL_ERR_INVALID 1
#define FL_ERR_LAUNCH 2
#define FL_ERR_UNSUPPORTED 3

#define FL_MLA_PAGE 64        /* srt/layers/attention/flashmla_backend.py:25 PAGE_SIZE */
#define FL_MLA_META_W 8       /* ints per tile-scheduler row */
#define FL_MLA_ROWS_PER_WG 64 /* query rows (s_q*H) one workgroup owns */

/* KV-cache formats (srt/mem_cache/memory_pool.py:635-658) */
#define FL_KV_FP8_PER_TOKEN 0 /* (u8[.,512], f32[.,1], bf16[.,64]) tuple, quant_method="per_token_head" */
#define FL_KV_FP8_576 1       /* one fp8 [.,576] tensor, descale scalars */
#define FL_KV_BF16_576 2      /* one bf16 [.,576] tensor */

/* ABI version of THIS header.  A binding checks fl_version() == FL_ABI_VERSION once after loading the library: that check covers EVERY
 * argument struct below.  FlMlaDecodeArgs — the one struct that has grown so far — additionally carries its own size (`struct_bytes`, first
 * field): a caller compiled against another layout of it is refused with FL_ERR_INVALID instead of being read past its end.  The other
 * argument structs (FlGroupedGemmArgs, FlRopeArgs, FlMlaAbsorbArgs ...) have no size field and are read as this header lays them out.
 * 101: FlMlaDecodeArgs gained struct_bytes (FIRST field: every later field moved — a C caller built against 100 must be recompiled) and
 *      block_table_cols (last); fl_topk_gate added.
 * 102: additive — fl_fused_add_rmsnorm_offset (GemmaRMSNorm: the 1 added in fp32), fl_mla_set_merge_timeout (+ the in-kernel split merge
 *      reports a timeout through the next fl_mla_decode call); no struct changed. */
#define FL_ABI_VERSION 102
const char* fl_last_error(void);
int fl_version(void);
/* number of compute units of HIP device `device` (host query, cached) */
int fl_device_cu_count(int device, int* cu_count);

/* ---- K3: get_mla_metadata (flashmla_backend.py:261-265,276-280,307-321,335-339,380-384) ----
 * num_parts rows of FL_MLA_META_W int32 + num_splits[bs+1]; shapes static in bs. */
int fl_mla_num_parts(int cu_count, int rows_per_kv_head);
int fl_mla_get_metadata(const int32_t* cache_seqlens, int bs, int num_parts,
                        int32_t* tile_scheduler_metadata, int32_t* num_splits, fl_stream_t stream);

/* ---- K4: quantize_ckv_per_token_head (flashmla_backend.py:125,206) ----
 * q bf16 [rows, d_nope+d_rope] -> q_nope fp8 [rows,d_nope], q_scale f32 [rows], q_rope bf16 [rows,d_rope] */
int fl_mla_quant_q(const void* q, int64_t rows, int d_nope, int d_rope, void* q_nope, float* q_scale,
                   void* q_rope, fl_stream_t stream);

/* ---- K5: quantize_and_cache_k (srt/mem_cache/memory_pool.py:864-871,902-909; fallback :873-880) ----
 * key bf16 [n, d_nope+d_rope]; scatter into the three caches at slot indices[i]. */
int fl_mla_quant_store_k(const void* key, int64_t n, int d_nope, int d_rope, const int32_t* indices,
                         void* k_lora_cache, float* k_scale_cache, void* k_rope_cache, int64_t num_slots,
                         fl_stream_t stream);

/* K5 + K4 in one launch: fl_mla_quant_store_k (rows of `key`, scattered at `indices`) and fl_mla_quant_q (rows of `q`, dense
 * outputs) with the bytes of the two separate calls.  The reference issues them back to back in
 * FlashMLABackend.forward_decode (flashmla_backend.py:188-206: set_kv_buffer -> memory_pool.py:864-871, then
 * quantize_ckv_per_token_head); an integrator who wants one launch fewer per layer calls
 * flash_mla_fp8.quantize_q_and_cache_k.  Either row count may be 0. */
int fl_mla_quant_q_store_k(const void* key, int64_t n_k, const int32_t* indices, void* k_lora_cache, float* k_scale_cache,
                           void* k_rope_cache, int64_t num_slots, const void* q, int64_t q_rows, int d_nope, int d_rope,
                           void* q_nope, float* q_scale, void* q_rope, fl_stream_t stream);

/* ---- A2: the query side of the decode layer before the attention kernel, in ONE launch (an extension, like the entry point
 * above: INTEGRATION.md section 4) — torch.bmm(q_nope.transpose(0,1), w_kc) (srt/models/deepseek_v2.py:840), the rotary embedding of
 * q_pe and k_pe (:842-858), set_kv_buffer / K5 (flashmla_backend.py:188-196) and quantize_ckv_per_token_head / K4 (:198-206).
 * Outputs are K4's and K5's, bit-identical to the four-launch chain; the bf16 absorbed query is never written.
 * q [T, H, 192] bf16 (nope 128 | rope 64; element strides), w_kc [H, 512, 128] bf16 k-contiguous (head stride in elements),
 * cos_sin_cache [max_position, 64] f32 (cos 32 | sin 32), latent [T, 576] bf16 (k_nope | k_pe: k_pe is rotated IN PLACE; NULL = no
 * K rows), cache_loc i32 [T] (rows outside [0, num_slots) are skipped), outputs contiguous [T, H, 512] e4m3 / [T, H] f32 /
 * [T, H, 64] bf16. ---- */
typedef struct FlMlaAbsorbArgs {
  const void* q;
  int64_t q_stride_token, q_stride_head;
  int64_t num_tokens;
  int32_t num_heads;
  int32_t d_nope, d_rope, d_lora;   /* 128, 64, 512 */
  const void* w_kc;
  int64_t w_stride_head;
  const int64_t* positions;
  const float* cos_sin_cache;
  int64_t max_position;
  int32_t is_neox;
  void* latent;
  int64_t latent_stride;
  const int32_t* cache_loc;
  void* k_lora_cache;
  float* k_scale_cache;
  void* k_rope_cache;
  int64_t num_slots;
  void* q_nope_out;
  float* q_scale_out;
  void* q_rope_out;
} FlMlaAbsorbArgs;
int fl_mla_absorb_rope_quant(const FlMlaAbsorbArgs* args, fl_stream_t stream);

/* ---- K6: dequantize_ckv_fused_indexed (memory_pool.py:821-824; fallback :826-831) ---- */
int fl_mla_dequant_gather(const void* k_lora_cache, const void* k_rope_cache, const float* k_scale_cache,
                          const int32_t* indices, int64_t n, int d_nope, int d_rope, int64_t num_slots,
                          void* k_lora_out, void* k_rope_out, fl_stream_t stream);

/* ---- K1/K2: flash_mla_ckv_fp8_per_token / flash_mla_with_kvcache
 * (flashmla_backend.py:208-222,127-142 / :227-254,145-175) ---- */
typedef struct FlMlaDecodeArgs {
  int32_t struct_bytes;     /* = sizeof(FlMlaDecodeArgs) as the CALLER compiled it; anything else is refused */
  int32_t kv_format;        /* FL_KV_* */
  int32_t bs, s_q, h_q;     /* q rows per request = s_q*h_q (one latent KV head) */
  int32_t d_nope, d_rope;   /* 512, 64 */
  int32_t causal;           /* query j sees keys [0, seqlen-(s_q-1-j)) when causal */
  int32_t num_parts;        /* rows of tile_scheduler_metadata */
  float softmax_scale;
  const float* descale_q;   /* FL_KV_FP8_576 only: DEVICE scalars (flashmla_backend.py:237-238), NULL = 1.0 */
  const float* descale_k;
  /* query */
  const void* q_nope;       /* fp8 [bs,s_q,h_q,d_nope] (per-token) | fp8/bf16 [bs,s_q,h_q,576] */
  const void* q_rope;       /* bf16 [bs,s_q,h_q,d_rope] (per-token only) */
  const float* q_scale;     /* f32 [bs,s_q,h_q] (per-token only) */
  /* paged cache */
  const void* k_nope;       /* u8 [pages,64,d_nope] | [pages,64,576] */
  const void* k_rope;       /* bf16 [pages,64,d_rope] (per-token only) */
  const float* k_scale;     /* f32 [pages,64] (per-token only) */
  int64_t num_pages;
  const int32_t* block_table; /* i32 [bs, block_table_stride] page ids (bit-exact, allocator.py:60-102) */
  int64_t block_table_stride;
  const int32_t* cache_seqlens; /* i32 [bs] */
  int32_t* tile_scheduler_metadata; /* from fl_mla_get_metadata.  NOT const: columns 5..7 of every row are scratch counters of
                                      * the in-kernel split merge — zero on entry, WRITTEN by fl_mla_decode, zero again when
                                      * the launch has finished.  Ordering rule: launches that share one metadata tensor must
                                      * be ordered on ONE stream (or by events); two concurrent launches on the same tensor
                                      * would mix their arrival counts.  Give concurrent streams their own copy. */
  const int32_t* num_splits;
  /* outputs */
  void* out;                /* bf16 [bs,s_q,h_q,d_nope] */
  float* lse;               /* f32 [bs,h_q,s_q] natural-log LSE */
  float* o_accum;           /* split-KV workspace of (bs+num_parts)*s_q*h_q*d_nope*4 bytes (caller-owned, opaque:
                               f32 rows, or bf16 rows for the 128-row mapping; only fl_mla_decode reads it) */
  float* lse_accum;         /* f32 [bs+num_parts, s_q*h_q, 2] {weight LSE, exact LSE} workspace */
  const void* q_bf16;       /* optional (FL_KV_FP8_PER_TOKEN, s_q*h_q > 32): the UNQUANTISED query bf16 [bs,s_q,h_q,576]; the decode
                             * kernel then does quantize_ckv_per_token_head (K4, flashmla_backend.py:198-206) in its own prologue — same
                             * bytes, same result, one launch and a write + re-read of Q less; q_nope / q_rope / q_scale are ignored */
  int64_t block_table_cols; /* columns of a block_table row that may be read (block_table.shape[1]); 0 = block_table_stride.  The row
                             * stride is NOT the row width for an expanded (stride 0) or sliced table: the kernels clamp their
                             * unconditional page-id loads against THIS */
} FlMlaDecodeArgs;

int fl_mla_decode(const FlMlaDecodeArgs* args, fl_stream_t stream);
/* Budget of the in-kernel split merge's wait for a request's other pieces (default 2 s; env FLUENT_MLA_MERGE_TIMEOUT_S).  A merger that gives up
 * writes NaN into the request's rows and LSEs AND reports through a host-mapped word: the NEXT fl_mla_decode call returns FL_ERR_LAUNCH
 * (fl_last_error names the workgroup).  Synchronises the device (hipMemcpyToSymbol): call it outside captures. */
int fl_mla_set_merge_timeout(double seconds);
/* Sizes of the two workspaces above for a launch of this shape (the partial rows are bf16 or f32 depending on the
 * mapping the shape dispatches to; allocate exactly this much, e.g. from the graph's memory pool). */
int fl_mla_workspace_bytes(int kv_format, int bs, int s_q, int h_q, int num_parts, int64_t* o_accum_bytes,
                           int64_t* lse_accum_bytes);

/* ---- G1-G4: deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_{offset,contiguous,masked} / gemm_fp8_fp8_bf16_nt
 * (srt/layers/moe/gemms/fp8/fire.py:18 -> moe/executors/fp8_eps_executor.py:56,78;
 *  moe/executors/deep_ep_executor.py:583-586,607-613,655-662,681-688; dense/gemms/fp8/deep_geem.py:55) ---- */
#define FL_GEMM_OFFSET 0      /* rows [meta[e], meta[e+1]) use group e; meta = exclusive_sum i32 [E+1] */
#define FL_GEMM_CONTIGUOUS 1  /* row i uses group meta[i] (<0 = skip), groups 128-row aligned; meta = m_indices i32 [M] */
#define FL_GEMM_MASKED 2      /* A [G, rows_per_group, K], first meta[g] rows of group g valid; meta = masked_m i32 [G] */
#define FL_GEMM_DENSE 3       /* one group, meta unused */
typedef struct FlGemmArgs {
  int32_t mode;
  int32_t num_groups;         /* E (1 for dense) */
  int64_t M;                  /* total rows of A/out (masked: G*rows_per_group) */
  int32_t N, K;               /* W is [E, N, K] fp8, K % 128 == 0 */
  int64_t rows_per_group;     /* masked only */
  int64_t expected_m;         /* hint for the token-tile height (0 = derive) */
  const void* A;              /* fp8 [M, K] row-major */
  const float* As;            /* f32 scales of A, element (m, kb) at m*as_stride_m + kb*as_stride_k */
  int64_t as_stride_m, as_stride_k;
  int64_t as_stride_g;        /* masked only: stride between groups of As */
  const void* W;              /* fp8 [E, N, K] */
  const float* Ws;            /* f32 [E, ceil(N/128), K/128] */
  void* out;                  /* bf16 [M, N] */
  const int32_t* group_meta;
  void* workspace;            /* optional scratch (f32 partials) for split-K of problems with few tiles — the dense decode GEMMs of
                               * the MLA projections: [T<=256, K] x [N, K] has N/128 tiles for 256 CUs; NULL = never split */
  int64_t workspace_bytes;
} FlGemmArgs;
int fl_grouped_gemm_fp8(const FlGemmArgs* args, fl_stream_t stream);
int fl_gemm_set_num_cus(int n);   /* deep_gemm.set_num_sms (srt/tbo/tbo_executor.py:129-134) */
int fl_gemm_get_num_cus(void);

/* ---- Q1/Q2: 1x128 per-token-group FP8 quantisation (flashinfer.quantization.quant_1x128,
 * moe/executors/fp8_eps_executor.py:53-55,75-77; flashinfer.sgl_per_token_group_quant_fp8,
 * dense/gemms/fp8/fp8_kernel.py:457-460).  s = max(amax, eps)/448 (fp32), q = x/s -> e4m3 (RNE).
 * x bf16 [M, K]; scale (m, kb) written at m*s_stride_m + kb*s_stride_k. ---- */
int fl_quant_1x128(const void* x, int64_t M, int K, float eps, void* x_q, float* x_s, int64_t s_stride_m,
                   int64_t s_stride_k, fl_stream_t stream);

/* ---- A1: silu(x[:, :I]) * x[:, I:] (eps.executor.silu, fp8_eps_executor.py:62; flashinfer.silu_and_mul), bf16;
 * with q_out != NULL also emits the 1x128 quantisation of the result
 * (flashinfer.activation.silu_and_mul_fuse_block_quant, deep_ep_executor.py:676, activation.py:73). ---- */
int fl_silu_and_mul(const void* x, int64_t M, int I, void* out_bf16, void* q_out, float* s_out, int64_t s_stride_m,
                    int64_t s_stride_k, fl_stream_t stream);

/* ---- C5/C6/C7: compute half of flashinfer.comm.trtllm_{allreduce,reducescatter,allgather}_fusion
 * (srt/layers/flashinfer_comm_fusion.py:286-401, 404-513, 516-640 <- layernorm.py:114-189, 305-359).  The exchange is one
 * one-shot collective over RCCL (host: torch.distributed); this is everything after it in ONE kernel:
 * v = sum_{w<num_pieces} x[w*piece_stride + (t,h)] (+ add_in) (+ residual_in)  [fp32];  residual_out = bf16(v);
 * norm_out = bf16(v * rsqrt(mean_h v^2 + eps) * gamma)  (RMSNorm.forward_native, layernorm.py:88-112);
 * quant_out/scale_out = 1x128 e4m3 quantisation of norm_out (scale (t,g) at t*s_stride_t + g*s_stride_g).
 * All tensors bf16 [T, H] (gamma [H]); any output may be NULL; with norm_out == quant_out == NULL it is the plain
 * one-shot reduction (gamma may be NULL): eps.communication.TPDPConvertor.reduce_scatter (C3,
 * srt/distributed/decoder_comm_manager.py:42-85, layers/dp_attention.py:62-74). ---- */
int fl_fused_add_rmsnorm(const void* x, int num_pieces, int64_t piece_stride /*elements*/, const void* add_in,
                         const void* residual_in, const void* gamma, float eps, int64_t T, int H, void* residual_out,
                         void* norm_out, void* quant_out, float* scale_out, int64_t s_stride_t, int64_t s_stride_g,
                         fl_stream_t stream);
/* The same with norm_out = bf16(v * rsqrt(..) * (gamma + gamma_offset)), the offset added in fp32: GemmaRMSNorm (layernorm.py:209-233,
 * gamma_offset = 1).  x / residual_in may alias norm_out / residual_out (flashinfer's in-place fused_add_rmsnorm). */
int fl_fused_add_rmsnorm_offset(const void* x, int num_pieces, int64_t piece_stride /*elements*/, const void* add_in,
                                const void* residual_in, const void* gamma, float gamma_offset, float eps, int64_t T, int H,
                                void* residual_out, void* norm_out, fl_stream_t stream);
/* C7 (layernorm.py:305-359 <- models/deepseek_v2.py:799-807): dual RMSNorm over gathered rows ag bf16 [T, D]:
 * cols [0, q_rank) -> x_norm_out [T, q_rank] (+ optional 1x128 fp8 quant), cols [q_rank, q_rank+kv_rank) in place. */
int fl_dual_rmsnorm(void* ag, int64_t T, int D, int q_rank, int kv_rank, const void* gamma_q, const void* gamma_kv,
                    float eps_q, float eps_kv, void* x_norm_out, void* quant_out, float* scale_out, int64_t s_stride_t,
                    int64_t s_stride_g, fl_stream_t stream);

/* ---- C5/C6 one-shot: the exchange AND the fused epilogue in one kernel over peer-mapped (hipIpc) workspaces — replaces
 * flashinfer.comm.trtllm_create_ipc_workspace_for_all_reduce_fusion + trtllm_allreduce_fusion / trtllm_reducescatter_fusion
 * (srt/layers/flashinfer_comm_fusion.py:64-109 workspace, :286-401 C5, :404-513 C6) for token counts up to the workspace's
 * max_tokens.  Every rank writes its rows into each destination's inbox over xGMI, raises per-row flags, and the
 * workgroup owning a row reduces it in rank order (bit-identical on all ranks) + residual + RMSNorm (+ 1x128 fp8 quant).
 * Protocol: csrc/comm_protocol.h.  Setup: create on every rank -> exchange the 64-byte handles (any transport) ->
 * connect.  The epoch lives in device memory: launches may be captured in a hipGraph and replayed. ---- */
int fl_comm_workspace_size(int world, int64_t max_tokens, int hidden, int64_t* bytes_out);
int fl_comm_create(int rank, int world, int64_t max_tokens /*<= 1024*/, int hidden /*<= 8192*/, void** comm_out);
int fl_comm_local_handle(void* comm, void* handle_out /*64 bytes: hipIpcMemHandle_t of this rank's workspace*/);
int fl_comm_connect(void* comm, const void* handles /*world x 64 bytes in rank order; NULL allowed at world 1*/);
int fl_comm_set_timeout(void* comm, double seconds /*budget of one flag wait; default 10 s*/);
/* C5: all ranks pass in bf16 [T, H]; every rank gets sum (+ residual_in [T, H]) -> residual_out, RMSNorm -> norm_out,
 * quant_out/scale_out optional (any may be NULL; gamma NULL = sum only into residual_out). */
int fl_allreduce_fused(void* comm, const void* in, int64_t T, int H, const void* residual_in, const void* gamma, float eps,
                       void* residual_out, void* norm_out, void* quant_out, float* scale_out, int64_t s_stride_t,
                       int64_t s_stride_g, fl_stream_t stream);
/* C6: in bf16 [T, H] on every rank; rank r gets its token slice (get_num_tokens_per_rank: the first T % world ranks own
 * one more) of the sum; add_in / residual_in / outputs are [slice rows, H]. */
int fl_reducescatter_fused(void* comm, const void* in, int64_t T, int H, const void* add_in, const void* residual_in,
                           const void* gamma, float eps, void* residual_out, void* norm_out, void* quant_out, float* scale_out,
                           int64_t s_stride_t, int64_t s_stride_g, fl_stream_t stream);
/* C7 / the all-gather half of C3 as one kernel (flashinfer.comm.trtllm_allgather_fusion, flashinfer_comm_fusion.py:613-638;
 * eps.communication.TPDPConvertor.all_gather, layers/dp_attention.py:62-74): every rank holds `t_cur` =
 * get_num_tokens_per_rank(world, T)[rank] rows `in [t_cur, D]`; out [T, D] receives all rows in rank order.  q_rank > 0: dual
 * RMSNorm of every gathered row — cols [0, q_rank) -> x_norm_out [T, q_rank] (+ optional 1x128 fp8 quant_out / scale_out),
 * cols [q_rank, q_rank + kv_rank) in place in `out`.  T <= 1024, ceil(T / world) <= max_tokens, D <= hidden. */
int fl_allgather_fused(void* comm, const void* in, int64_t t_cur, int64_t T, int D, void* out, int q_rank, int kv_rank,
                       const void* gamma_q, const void* gamma_kv, float eps_q, float eps_kv, void* x_norm_out, void* quant_out,
                       float* scale_out, int64_t s_stride_t, int64_t s_stride_g, fl_stream_t stream);
/* C1 / C2 exchange on the same transport (eps.fast_ep.AllToAll over MSCCL++: srt/layers/moe/dispatcher/fast_ep.py:45-78,
 * distributed/parallel_state.py:965-977): equal-split all-to-all of `world` slabs of `cap` rows x D 2-byte elements — slab p of `send`
 * lands as slab `rank` of rank p's `recv`.  ids_col >= 0: a row's top_k int32 expert ids start at element ids_col of the row; rows whose
 * ids are all negative (nobody routed there) travel and are copied out as their 16-byte-aligned tail only.  cap * world <= 1024,
 * cap <= max_tokens, D <= hidden of the communicator. */
int fl_alltoall_oneshot(void* comm, const void* send, void* recv, int cap, int D, int ids_col, int top_k, fl_stream_t stream);
int fl_comm_check(void* comm /*synchronises; FL_ERR_LAUNCH if a flag wait ever timed out*/);
int fl_comm_destroy(void* comm);
/* The same protocol on plain host memory (shared-memory workspaces of several processes): drives the CPU-side protocol
 * test of the flag / epoch / parity logic.  Test infrastructure — nothing on the product path calls these. */
int fl_comm_host_init(void* ws, int world, int64_t max_tokens, int hidden);
int fl_comm_host_exchange(void* const* ws /*[world] workspaces as mapped in this process*/, int rank, int world, int64_t max_tokens,
                          int hidden, int reduce_scatter, const uint16_t* in /*bf16 [T, H]*/, int64_t T, int H,
                          float* out /*f32 [rows, H]: the sums*/, double timeout_s);

/* ---- C1/C2: device side of eps.fast_ep.AllToAll.dispatch / combine (srt/layers/moe/dispatcher/fast_ep.py:45-51,
 * 73-78).  The exchange is one equal-split all-to-all per direction over RCCL (host: torch.distributed); these do the
 * integer / row work around it, sync-free with static shapes.  Rows are bf16 [*, hidden], hidden % 8 == 0. ---- */
int fl_ep_route(const int32_t* indices /*[num_pairs] global expert ids*/, int64_t num_pairs, int experts_per_rank, int world,
                int cap /*rows per peer slab*/, int32_t* send_slot /*[num_pairs] -> dest*cap+pos or -1*/,
                int32_t* send_eid /*[world*cap] local expert id at the destination, -1 = empty*/, fl_stream_t stream);
/* Token-once-per-peer routing of the same dispatch (fluent_mi355/ep.py): tok_slot [tokens, world] = the token's row in
 * each peer slab (-1: none of its experts lives there), send_eid [world*cap, top_k] = local expert ids per slab row (-1
 * padded), pair_src [world*cap, top_k] = t*top_k + k of the (token, expert) pair behind every slab pair (-1 padded): the
 * layout the combine weights travel in (fl_ep_gather_f32).
 * ROW STRIDES (round 3): the ids (and, when the caller has them at dispatch time, the routing weights) can travel in the TAIL of
 * their slab row — message row = [hidden bf16 | top_k int32 ids | top_k f32 weights] — so that a direction is ONE all-to-all
 * instead of two.  The `*_row_stride` arguments are the distance between consecutive rows of that tensor in ITS OWN element
 * type (int32 / f32 / bf16); 0 = dense ([rows, top_k] resp. [rows, hidden]). */
int fl_ep_route_dedup(const int32_t* indices /*[tokens, top_k] global expert ids*/, int64_t num_tokens, int top_k,
                      int experts_per_rank, int world, int cap, int32_t* tok_slot, int32_t* send_eid, int32_t* pair_src,
                      int64_t eid_row_stride,
                      const float* weights /*optional [tokens, top_k] f32: placed beside the ids (send_w[row, j], 0 = padding)*/,
                      float* send_w, int64_t w_row_stride, fl_stream_t stream);
int fl_ep_gather_f32(const float* vals, int64_t n, const int32_t* src /*[out_n]*/, float* out /*out[j] = vals[src[j]], 0 where src[j] is not in [0, n)*/,
                     int64_t out_n, int per_row /*entries per output row*/, int64_t out_row_stride, fl_stream_t stream);
int fl_ep_gather_rows_div(const void* src, int64_t src_rows, const int32_t* idx, int64_t n, int div /*dst[i] = src[idx[i] / div]*/,
                          int hidden, void* dst, int64_t dst_rows,
                          const int32_t* n_valid /*optional DEVICE scalar: only rows i < *n_valid are copied (the launch is sized for n)*/,
                          int64_t src_row_stride, fl_stream_t stream);
int fl_ep_sort(const int32_t* recv_eid /*[num_slots] = [rows, per_row]*/, int64_t num_slots, int num_local_experts,
               int32_t* order /*[num_slots] slots grouped by expert, invalid last*/, int32_t* exclusive_sum /*[E_l+1]*/,
               int32_t* inverse /*optional [num_slots]: inverse[order[i]] = i*/, int per_row, int64_t eid_row_stride, fl_stream_t stream);
int fl_ep_gather_rows(const void* src, int64_t src_rows, const int32_t* idx, int64_t n, int hidden, void* dst,
                      int64_t dst_rows, fl_stream_t stream);   /* dst[i] = src[idx[i]] */
int fl_ep_scatter_rows(const void* src, int64_t src_rows, const int32_t* idx, int64_t n, int hidden, void* dst,
                       int64_t dst_rows, fl_stream_t stream);  /* dst[idx[i]] = src[i] */
int fl_ep_send_rows(const void* x, int64_t num_tokens, const int32_t* send_slot /*[num_pairs]*/, int64_t num_pairs, int top_k,
                    int hidden, void* send_buf, int64_t send_rows, int64_t send_row_stride, fl_stream_t stream);  /* send_buf[send_slot[p]] = x[p / top_k] */
int fl_ep_combine(const void* ret_rows, int64_t num_ret_rows, const int32_t* send_slot, const float* weights,
                  int64_t num_tokens, int top_k, int hidden, void* out, int64_t weight_row_stride, fl_stream_t stream);

/* A1-masked: silu_and_mul_masked_post_quant_fwd (srt/layers/moe/executors/deep_ep_executor.py:106-170, Triton kernel :31-104)
 * between the two masked grouped GEMMs of the low-latency DeepEP path: x bf16 [G, rows_per_group, 2I] contiguous; only rows
 * < masked_m[g] of group g are read / written: q_out fp8 [G, rows_per_group, I] contiguous, s_out f32 at
 * g*s_stride_g + row*s_stride_m + kb*s_stride_k.  Arithmetic of A1 + Q1 (bit-exact vs their golden vectors). */
int fl_silu_and_mul_masked(const void* x, int num_groups, int64_t rows_per_group, int I, const int32_t* masked_m, void* q_out,
                           float* s_out, int64_t s_stride_g, int64_t s_stride_m, int64_t s_stride_k, fl_stream_t stream);

/* ---- R1: flashinfer.moe_fused_gate as called by biased_grouped_topk_gpu (srt/layers/moe/topk.py:709-733);
 * semantics of its torch statement biased_grouped_topk_impl (topk.py:596-663), renormalize=True, no fused shared
 * experts.  logits f32 [T, E] (E = 64*2^k <= 1024), bias f32 [E]; num_expert_group a power of two <= 64;
 * num_token_non_padded: optional DEVICE int32 scalar (rows at or beyond it get ids -1, topk.py:673-680).
 * Rows of the outputs are ordered by descending (sigmoid + bias); ties -> lower expert id. ---- */
int fl_moe_fused_gate(const float* logits, const float* bias, int64_t num_tokens, int num_experts, int num_expert_group,
                      int topk_group, int topk, float routed_scaling_factor, int apply_routed_scaling_factor_on_output,
                      const int32_t* num_token_non_padded, float* topk_weights /*[T, topk]*/,
                      int32_t* topk_ids /*[T, topk]*/, fl_stream_t stream);

/* ---- R1b: the plain (ungrouped) top-k routers imported beside it — flashinfer.topk_softmax (srt/layers/moe/topk.py:33, called by
 * fused_topk :505-520; torch statement fused_topk_torch_native :463-495), flashinfer.routing_flash (LongCat-Flash router, :836-845;
 * torch statement fused_topk_bias :51-70) and eps.utils.ops._ops.topk_sigmoid (:44-47).  logits f32 [T, E] (1 <= E <= 1024),
 * bias f32 [E] or NULL; score_fn 0 = softmax over the row, 1 = sigmoid; choice = score + bias; top-k by choice (descending, ties ->
 * lower expert id); weight = UNBIASED score, divided by the sum of the chosen scores if `renormalize`, times `scale`. ---- */
int fl_topk_gate(const float* logits, const float* bias, int64_t num_tokens, int num_experts, int topk, int score_fn,
                 int renormalize, float scale, float* topk_weights /*[T, topk]*/, int32_t* topk_ids /*[T, topk]*/,
                 fl_stream_t stream);

/* ---- R2: flashinfer.apply_rope_with_cos_sin_cache_inplace as RotaryEmbedding.forward_cuda calls it
 * (srt/layers/rotary_embedding.py:203-218) on q_pe / k_pe of the MLA path (models/deepseek_v2.py:646-647,695-696).
 * q bf16 [T, Hq, >=rotary_dim] and k bf16 [T, Hk, >=rotary_dim] rotated IN PLACE over their first rotary_dim elements
 * (element strides given, last dim contiguous, rows 16-B aligned); positions int64 [T]; cos_sin_cache f32
 * [max_position, rotary_dim] = cos | sin halves (:141-149,791-802).  fp32 arithmetic of forward_native (:804-846),
 * bit-exact.  is_neox: rotate halves (:50-53), else neighbouring pairs (:56-60, DeepSeek). ---- */
int fl_rope_inplace(const int64_t* positions, int64_t num_tokens, void* q, int64_t q_stride_token, int64_t q_stride_head,
                    int num_q_heads, void* k, int64_t k_stride_token, int64_t k_stride_head, int num_k_heads,
                    const float* cos_sin_cache, int64_t max_position, int rotary_dim, int is_neox, fl_stream_t stream);

/* The same call with what models/deepseek_v2.py:843-861 (forward_absorb_prepare) passes on the decode path:
 * output_q_rope / output_k_rope (rotated rows into a SEPARATE tensor, the source is left untouched; NULL = in place) and
 * FusedSetKVBufferArg (models/utils.py:52-81, bf16 KV cache only — enable_fused_set_kv_buffer() is false for fp8 KV):
 * v_buffer[cache_loc[t]] = value[t], k_buffer[cache_loc[t]] = rotated key[t] (k heads side by side), in the same launch.
 * All strides in bf16 elements; cache_loc int64 or int32 [T], negative = skip the token. */
typedef struct {
  const int64_t* positions; int64_t num_tokens;
  void* q; int64_t q_stride_token, q_stride_head; int num_q_heads;
  void* k; int64_t k_stride_token, k_stride_head; int num_k_heads;
  const float* cos_sin_cache; int64_t max_position; int rotary_dim; int is_neox;
  void* q_out; int64_t qo_stride_token, qo_stride_head;
  void* k_out; int64_t ko_stride_token, ko_stride_head;
  void* k_buffer; int64_t k_buffer_stride;
  void* v_buffer; int64_t v_buffer_stride;
  const void* value; int64_t value_stride; int value_dim;
  const void* cache_loc; int cache_loc_is_i64;
} FlRopeArgs;
int fl_rope(const FlRopeArgs* args, fl_stream_t stream);

/* ---- K7: the work of copy_all_layer_kv_cache_tiled (srt/mem_cache/memory_pool.py:2055-2090) as
 * MLATokenToKVPool.move_kv_cache launches it (:746-777): for every buffer b of the table, rows tgt_loc[i] <- src_loc[i]
 * with the semantics of `buf[tgt] = buf[src]` (:756-763): all sources are read before any target is written (overlapping
 * sets are safe).  data_ptrs / row_bytes: DEVICE tables [num_buffers] (base address, bytes per row; rows multiples of 4 B)
 * like the reference's data_ptrs / data_strides (:330-343); max_row_bytes = host-known maximum of row_bytes; tgt/src int64
 * DEVICE [num_locs], num_locs <= 8192 (more: fl_kv_move_staged); num_slots > 0: rows outside [0, num_slots) are skipped.  Bit-exact byte work. ---- */
int fl_kv_move(const uint64_t* data_ptrs, const int64_t* row_bytes, int num_buffers, int64_t max_row_bytes,
               const int64_t* tgt_loc, const int64_t* src_loc, int64_t num_locs, int64_t num_slots, fl_stream_t stream);
/* The same move for MORE rows than fl_kv_move holds in registers (num_locs up to 4,194,240): two launches through a caller-owned staging
 * area of num_locs x sum_row_bytes bytes (sum over the table's buffers) — every source row is copied out, then every target written: the
 * same "all reads before any write" semantics.  row_prefix: DEVICE [num_buffers], exclusive prefix sums of row_bytes (buffer b's staging
 * area starts at row_prefix[b] x num_locs).  A pair with either index outside [0, num_slots) is skipped. */
int fl_kv_move_staged(const uint64_t* data_ptrs, const int64_t* row_bytes, const int64_t* row_prefix, int num_buffers,
                      const int64_t* tgt_loc, const int64_t* src_loc, int64_t num_locs, int64_t num_slots, void* workspace,
                      int64_t workspace_bytes, int64_t sum_row_bytes, fl_stream_t stream);

/* ---- C8/C9: ep_scatter / ep_gather of the DeepExecutor (srt/layers/moe/executors/deep_ep_executor.py:271-332,
 * 396-430; Triton kernels :173-268, :335-393): between the DeepEP dispatch and the contiguous grouped GEMM, and back.
 * Strides in ELEMENTS of the tensor they belong to; topk / ids int32 or int64 (flag).  num_recv_tokens_per_expert: int32
 * [E], each a multiple of 128; num_rows = rows of output_tensor / m_indices.  Order inside an expert's group: atomic
 * cursors, unspecified as in the reference (:247).  expert_start_loc ends as start + count, like the reference's. ---- */
int fl_ep_scatter(const void* recv_x /*fp8 [T, H]*/, int64_t x_stride, const float* recv_x_scale /*[T, H/128]*/,
                  int64_t xs_stride, const void* recv_topk /*[T, K] local expert id or < 0*/, int topk_is_int64,
                  int64_t topk_stride, int64_t num_tokens, int top_k, int hidden,
                  const int32_t* num_recv_tokens_per_expert, int num_experts, int32_t* expert_start_loc /*[E] out*/,
                  void* output_tensor /*fp8 [M, H]*/, int64_t out_stride, float* output_tensor_scale /*[M, H/128]*/,
                  int64_t outs_stride, int32_t* m_indices /*[M]*/, int64_t num_rows, int32_t* output_index /*[T, K]*/,
                  int64_t oi_stride, fl_stream_t stream);
int fl_ep_gather(const void* input_tensor /*bf16 [M, H]*/, int64_t in_stride, int64_t num_rows,
                 const void* recv_topk_ids /*[T, K]*/, int ids_is_int64, int64_t ids_stride,
                 const float* recv_topk_weight /*[T, K]*/, int64_t w_stride, const int32_t* input_index /*[T, K]*/,
                 int64_t idx_stride, int64_t num_tokens, int top_k, int hidden, void* output_tensor /*bf16 [T, H]*/,
                 int64_t out_stride, fl_stream_t stream);

/* ---- B1: the small dense bf16 GEMMs between the big kernels — weight absorption
 * torch.bmm(q_nope.transpose(0,1), w_kc, out=Q[..., :512].transpose(0,1)) / torch.bmm(attn_output.transpose(0,1), w_vc)
 * (srt/models/deepseek_v2.py:840,886; both weights are stored k-contiguous, :1632-1633) and the router GEMM
 * flashinfer.dsv3_router_gemm(hidden, weight, out_dtype) (:177-179).  C[b,m,n] = sum_k A[b,m,k] * B[b,n,k]: bf16 operands
 * contiguous in k, strides in elements, fp32 accumulation, C bf16 (RNE) or f32; N % 32 == 0, K % 16 == 0. ---- */
int fl_bmm_bf16_nt(const void* A, const void* B, void* C, int batch, int64_t M, int N, int K, int64_t a_stride_b,
                   int64_t a_stride_m, int64_t b_stride_b, int64_t b_stride_n, int64_t c_stride_b, int64_t c_stride_m,
                   int out_is_f32, fl_stream_t stream);

/* The same product (one batch) split over K across workgroups, for few output tiles and a long K — the router GEMM
 * flashinfer.dsv3_router_gemm (srt/models/deepseek_v2.py:177-179: hidden [T, 7168] x gate weight [256, 7168]^T -> f32 logits):
 * f32 partials in `workspace` (at least fl_gemm_bf16_nt_splitk_workspace_bytes(M, N, K) bytes, 16-byte aligned; launches that
 * share it must be ordered), summed in split order by a second launch (deterministic).  N % 64 == 0, K % 64 == 0. */
int64_t fl_gemm_bf16_nt_splitk_workspace_bytes(int64_t M, int N, int K);
int fl_gemm_bf16_nt_splitk(const void* A, const void* B, void* C, int64_t M, int N, int K, int64_t a_stride_m,
                           int64_t b_stride_n, int64_t c_stride_m, int out_is_f32, void* workspace, int64_t workspace_bytes,
                           fl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
