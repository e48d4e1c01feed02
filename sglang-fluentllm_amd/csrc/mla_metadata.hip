// K3 — split-KV tile scheduler: flash_mla_*.get_mla_metadata (flashmla_backend.py:261-265,307-321,335-341).
//
// The reference copies the result into persistent graph buffers whose shapes are static in max_bs, so this runs
// on-device, on the caller's stream, with no host sync.  The FORMAT is ours (only fl_mla_decode consumes it):
//   tile_scheduler_metadata[p] = {begin_req, begin_tile, end_req, end_tile, begin_split_idx, 0, 0, 0}
//   num_splits[b+1]-num_splits[b] = number of parts that touch request b (cumulative; split slots in o_accum)
// Greedy partition of the row-major (request, 64-token tile) list into num_parts contiguous parts of capacity P; a
// piece of a request costs (its tiles + FIXED_OVERHEAD).  P is the SMALLEST capacity for which the greedy walk places
// everything (searched over [ceil(total/num_parts), +63], one candidate per lane), so that e.g. 128 equal requests on
// 256 parts split 32/32 pages instead of 33/31 (the fixed formula ceil(total/parts) + FIXED_OVERHEAD lets the first
// part of every request run ahead: the kernel ends with its slowest workgroup).  m[5] of every part is zeroed: it is
// the "partial ready" flag of the in-kernel split merge (mla_decode_fp8_x.hip), set and reset by the decode kernel.
// Python statement of the same algorithm: oracle/mla_ref.py:get_mla_metadata.
#include "fl_common.h"

namespace {
constexpr int kFixedOverhead = 2;
constexpr int kMaxBs = 8192;   // tile counts are staged in LDS

// parts the greedy walk needs with capacity P (stops counting beyond limit)
__device__ int parts_needed(const int* nt_of, const int bs, const int P, const int limit) {
  int req = 0, tile = 0, parts = 0;
  while (req < bs && parts <= limit) {
    int remain = P;
    ++parts;
    while (req < bs) {
      const int left = nt_of[req] - tile;
      if (remain >= left + kFixedOverhead) {
        remain -= left + kFixedOverhead;
        ++req; tile = 0;
      } else {
        const int take = remain - kFixedOverhead;
        if (take > 0) tile += take;
        break;
      }
    }
  }
  return parts;
}

__global__ void mla_metadata_kernel(const int32_t* __restrict__ seqlens, int bs, int num_parts,
                                    int32_t* __restrict__ meta, int32_t* __restrict__ num_splits) {
  __shared__ int nt_of[kMaxBs];
  const int lane = threadIdx.x;
  int total = 0;
  for (int b = lane; b < bs; b += 64) {
    const int L = seqlens[b];
    const int nt = L > 0 ? (L + FL_MLA_PAGE - 1) / FL_MLA_PAGE : 0;
    nt_of[b] = nt;
    total += nt + kFixedOverhead;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
  __syncthreads();
  int p_min = (total + num_parts - 1) / num_parts;
  if (p_min < 1 + kFixedOverhead) p_min = 1 + kFixedOverhead;
  // lane l tests capacity p_min + l; the fixed formula's capacity (p_min + FIXED_OVERHEAD) is the fallback
  const bool ok = parts_needed(nt_of, bs, p_min + lane, num_parts) <= num_parts;
  const unsigned long long okmask = __ballot(ok);
  const int payload = okmask ? p_min + __builtin_ctzll(okmask) : p_min + kFixedOverhead;
  if (lane != 0) return;
  int req = 0, tile = 0, split = 0, cum = 0;
  num_splits[0] = 0;
  for (int p = 0; p < num_parts; ++p) {
    int32_t* m = meta + p * FL_MLA_META_W;
    m[0] = req; m[1] = tile; m[4] = split; m[5] = 0; m[6] = 0; m[7] = 0;
    int remain = payload;
    while (req < bs) {
      const int left = nt_of[req] - tile;
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
  const int per_wg = (fl_mla_use_x() && rows_per_kv_head > FL_MLA_ROWS_PER_WG) ? fl_mla_x_rows_per_wg() : FL_MLA_ROWS_PER_WG;
  const int row_groups = (rows_per_kv_head + per_wg - 1) / per_wg;
  const int parts = cu_count / row_groups;
  return parts > 0 ? parts : 1;
}

extern "C" int fl_mla_get_metadata(const int32_t* cache_seqlens, int bs, int num_parts,
                                   int32_t* tile_scheduler_metadata, int32_t* num_splits, fl_stream_t stream) {
  FL_CHECK_ARG(bs >= 0 && bs <= kMaxBs && num_parts > 0, "fl_mla_get_metadata: bs=%d (max %d) num_parts=%d", bs, kMaxBs,
               num_parts);
  FL_CHECK_ARG(cache_seqlens && tile_scheduler_metadata && num_splits, "fl_mla_get_metadata: null pointer");
  mla_metadata_kernel<<<1, 64, 0, (hipStream_t)stream>>>(cache_seqlens, bs, num_parts, tile_scheduler_metadata,
                                                          num_splits);
  FL_CHECK_LAUNCH("fl_mla_get_metadata");
  return FL_OK;
}
