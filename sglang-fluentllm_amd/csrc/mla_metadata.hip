// K3 — split-KV tile scheduler: flash_mla_*.get_mla_metadata (flashmla_backend.py:261-265,307-321,335-341).
//
// The reference copies the result into persistent graph buffers whose shapes are static in max_bs, so this runs
// on-device, on the caller's stream, with no host sync.  The FORMAT is ours (only fl_mla_decode consumes it):
//   tile_scheduler_metadata[p] = {begin_req, begin_tile, end_req, end_tile, begin_split_idx, 0, 0, 0}
//   num_splits[b+1]-num_splits[b] = number of parts that touch request b (cumulative; split slots in o_accum)
// Greedy equal-payload partition of the row-major (request, 64-token tile) list; every request costs
// ntiles + FIXED_OVERHEAD tiles.  Python statement of the same algorithm: oracle/mla_ref.py:get_mla_metadata.
#include "fl_common.h"

namespace {
constexpr int kFixedOverhead = 2;

__global__ void mla_metadata_kernel(const int32_t* __restrict__ seqlens, int bs, int num_parts,
                                    int32_t* __restrict__ meta, int32_t* __restrict__ num_splits) {
  // bs is at most a few hundred: one wave computes the prefix sum, lane 0 walks the parts.
  const int lane = threadIdx.x;
  int total = 0;
  for (int b = lane; b < bs; b += 64) {
    const int L = seqlens[b];
    total += (L > 0 ? (L + FL_MLA_PAGE - 1) / FL_MLA_PAGE : 0) + kFixedOverhead;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
  if (lane != 0) return;
  int payload = (total + num_parts - 1) / num_parts + kFixedOverhead;
  if (payload < 1 + kFixedOverhead) payload = 1 + kFixedOverhead;
  int req = 0, tile = 0, split = 0, cum = 0;
  num_splits[0] = 0;
  for (int p = 0; p < num_parts; ++p) {
    int32_t* m = meta + p * FL_MLA_META_W;
    m[0] = req; m[1] = tile; m[4] = split; m[5] = 0; m[6] = 0; m[7] = 0;
    int remain = payload;
    while (req < bs) {
      const int L = seqlens[req];
      const int nt = L > 0 ? (L + FL_MLA_PAGE - 1) / FL_MLA_PAGE : 0;
      const int left = nt - tile;
      if (remain >= left + kFixedOverhead || p == num_parts - 1) {
        remain -= left + kFixedOverhead;
        cum += split + 1;
        num_splits[req + 1] = cum;
        ++req; tile = 0; split = 0;
      } else {
        const int take = remain - kFixedOverhead;
        if (take > 0) { tile += take; ++split; }
        break;
      }
    }
    m[2] = req; m[3] = tile;
  }
}
}  // namespace

extern "C" int fl_mla_num_parts(int cu_count, int rows_per_kv_head) {
  if (cu_count <= 0 || rows_per_kv_head <= 0) return 1;
  // 128-row workgroups (mla_decode_fp8_x.hip, opt-in) ingest every KV byte once per 128 rows; else 32/64-row workgroups
  const int per_wg = (fl_mla_use_x() && rows_per_kv_head > FL_MLA_ROWS_PER_WG) ? 2 * FL_MLA_ROWS_PER_WG : FL_MLA_ROWS_PER_WG;
  const int row_groups = (rows_per_kv_head + per_wg - 1) / per_wg;
  const int parts = cu_count / row_groups;
  return parts > 0 ? parts : 1;
}

extern "C" int fl_mla_get_metadata(const int32_t* cache_seqlens, int bs, int num_parts,
                                   int32_t* tile_scheduler_metadata, int32_t* num_splits, fl_stream_t stream) {
  FL_CHECK_ARG(bs >= 0 && num_parts > 0, "fl_mla_get_metadata: bs=%d num_parts=%d", bs, num_parts);
  FL_CHECK_ARG(cache_seqlens && tile_scheduler_metadata && num_splits, "fl_mla_get_metadata: null pointer");
  mla_metadata_kernel<<<1, 64, 0, (hipStream_t)stream>>>(cache_seqlens, bs, num_parts, tile_scheduler_metadata,
                                                          num_splits);
  FL_CHECK_LAUNCH("fl_mla_get_metadata");
  return FL_OK;
}
