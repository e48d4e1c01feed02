// K7 — move KV rows between slots in every layer's buffers at once: the work of copy_all_layer_kv_cache_tiled
// (python/sglang/srt/mem_cache/memory_pool.py:2055-2090) as MLATokenToKVPool.move_kv_cache launches it (:746-777; used by
// speculative decoding to compact accepted tokens), semantics of the torch statement `buf[tgt] = buf[src]` per buffer
// (:756-763, move_kv_cache_native :2039-2052): EVERY source row is read before ANY target row is written, so source and
// target sets may overlap.  Like the reference's kernel, one workgroup owns a byte column tile of one buffer for all
// moved rows: all of its loads, a barrier, then its stores.  Byte work, bit-exact.  Buffers are described by a device
// table (base pointer, bytes per row) like the reference's data_ptrs / data_strides (:330-343).
#include "fl_common.h"

namespace {
constexpr int kItems = 32;                 // (row, chunk) items per thread, held in registers across the barrier
constexpr int kCap = 256 * kItems;         // items per workgroup

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename Chunk>   // u32x4: rows in 16-B chunks; uint32_t: 4-B chunks (the per-token scale rows)
__device__ __forceinline__ void move_tile(uint8_t* base, const long long row_bytes, const int tile, const int ct,
                                          const long long* __restrict__ tgt, const long long* __restrict__ src, const int n,
                                          const long long num_slots) {
  const int row_chunks = (int)(row_bytes / sizeof(Chunk));
  const int c0 = tile * ct;
  if (c0 >= row_chunks) return;
  const int width = row_chunks - c0 < ct ? row_chunks - c0 : ct;
  const int total = n * width;
  Chunk v[kItems];
  // loads are unconditional on a clamped item (a predicated load into a register array is staged through LDS by hipcc)
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    int idx = threadIdx.x + 256 * i;
    idx = idx < total ? idx : total - 1;
    long long s = src[idx / width];
    s = (s < 0 || (num_slots > 0 && s >= num_slots)) ? 0 : s;
    v[i] = *reinterpret_cast<const Chunk*>(base + s * row_bytes + (long long)(c0 + idx % width) * sizeof(Chunk));
  }
  __syncthreads();   // every read of this column tile is done before its first write
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int idx = threadIdx.x + 256 * i;
    if (idx < total) {
      const long long s = src[idx / width], t = tgt[idx / width];
      const bool ok = s >= 0 && t >= 0 && (num_slots <= 0 || (s < num_slots && t < num_slots));   // never write out of the pool
      if (ok) *reinterpret_cast<Chunk*>(base + t * row_bytes + (long long)(c0 + idx % width) * sizeof(Chunk)) = v[i];
    }
  }
}

__global__ __launch_bounds__(256) void kv_move_kernel(const unsigned long long* __restrict__ data_ptrs,
                                                      const long long* __restrict__ row_bytes_of, const long long* __restrict__ tgt,
                                                      const long long* __restrict__ src, int n, long long num_slots, int ct16,
                                                      int ct4) {
  const int buf = blockIdx.x;
  uint8_t* base = reinterpret_cast<uint8_t*>(data_ptrs[buf]);
  const long long rb = row_bytes_of[buf];
  if ((rb & 15) == 0 && ((unsigned long long)base & 15) == 0) move_tile<u32x4>(base, rb, blockIdx.y, ct16, tgt, src, n, num_slots);
  else move_tile<uint32_t>(base, rb, blockIdx.y, ct4, tgt, src, n, num_slots);
}

// More rows than a workgroup holds in registers (> kCap): the same semantics through a staging area — pass 1 copies every source row of
// every buffer to workspace[row_prefix[b] * n + i * row_bytes[b]], pass 2 (a second launch: all reads done) writes them to their targets.
// One workgroup per (buffer, 64 moved rows); a wave walks a row in 16-B (or 4-B) chunks.
template <bool kScatterPass>
__global__ __launch_bounds__(256) void kv_move_staged_kernel(const unsigned long long* __restrict__ data_ptrs,
                                                             const long long* __restrict__ row_bytes_of,
                                                             const long long* __restrict__ row_prefix, const long long* __restrict__ tgt,
                                                             const long long* __restrict__ src, long long n, long long num_slots,
                                                             uint8_t* __restrict__ ws) {
  const int buf = blockIdx.x;
  uint8_t* base = reinterpret_cast<uint8_t*>(data_ptrs[buf]);
  const long long rb = row_bytes_of[buf];
  uint8_t* stage = ws + row_prefix[buf] * n;
  const bool wide = (rb & 15) == 0 && ((unsigned long long)base & 15) == 0 && ((unsigned long long)stage & 15) == 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long long i = (long long)blockIdx.y * 64 + wave; i < n && i < (long long)(blockIdx.y + 1) * 64; i += 4) {
    const long long sr = src[i], tr = tgt[i];
    const bool ok = sr >= 0 && tr >= 0 && (num_slots <= 0 || (sr < num_slots && tr < num_slots));   // the pair moves or it does not
    if (!ok) continue;
    const uint8_t* from = kScatterPass ? stage + i * rb : base + sr * rb;
    uint8_t* to = kScatterPass ? base + tr * rb : stage + i * rb;
    if (wide) {
      for (long long c = lane; c < rb / 16; c += 64) reinterpret_cast<u32x4*>(to)[c] = reinterpret_cast<const u32x4*>(from)[c];
    } else {
      for (long long c = lane; c < rb / 4; c += 64) reinterpret_cast<uint32_t*>(to)[c] = reinterpret_cast<const uint32_t*>(from)[c];
    }
  }
}
}  // namespace

extern "C" int fl_kv_move(const uint64_t* data_ptrs, const int64_t* row_bytes, int num_buffers, int64_t max_row_bytes,
                          const int64_t* tgt_loc, const int64_t* src_loc, int64_t num_locs, int64_t num_slots,
                          fl_stream_t stream) {
  FL_CHECK_ARG(data_ptrs && row_bytes && (num_locs == 0 || (tgt_loc && src_loc)), "fl_kv_move: null pointer");
  FL_CHECK_ARG(num_buffers >= 0 && num_buffers <= 65535 && max_row_bytes > 0 && max_row_bytes % 4 == 0,
               "fl_kv_move: num_buffers=%d max_row_bytes=%lld (rows are multiples of 4 bytes)", num_buffers, (long long)max_row_bytes);
  FL_CHECK_ARG(num_locs >= 0 && num_locs <= kCap, "fl_kv_move: %lld rows per call (max %d: a workgroup holds one column of all "
               "moved rows in registers between its reads and its writes)", (long long)num_locs, kCap);
  if (num_locs == 0 || num_buffers == 0) return FL_OK;
  // column tile width: up to 1024 moved rows ~8 (row, chunk) items per thread — more, narrower tiles fill the chip when few
  // rows move (256 rows, 183 buffers: 64 -> 20 us); beyond that the register budget (kItems per thread: fewer, fuller
  // workgroups were faster at 4096 rows, 180 vs 323 us)
  int ct = (int)((num_locs <= 1024 ? 2048 : kCap) / num_locs);
  ct = ct < 1 ? 1 : ct;
  const int tiles16 = (int)((max_row_bytes / 16 + ct - 1) / ct), tiles4 = (int)((max_row_bytes / 4 + ct - 1) / ct);
  const int tiles = tiles16 > tiles4 ? tiles16 : tiles4;   // (4-B rows are short: the grid is sized for the worse case)
  kv_move_kernel<<<dim3((unsigned)num_buffers, (unsigned)(tiles > 0 ? tiles : 1)), 256, 0, (hipStream_t)stream>>>(
      (const unsigned long long*)data_ptrs, (const long long*)row_bytes, (const long long*)tgt_loc, (const long long*)src_loc,
      (int)num_locs, num_slots, ct, ct);
  FL_CHECK_LAUNCH("fl_kv_move");
  return FL_OK;
}

extern "C" int fl_kv_move_staged(const uint64_t* data_ptrs, const int64_t* row_bytes, const int64_t* row_prefix, int num_buffers,
                                 const int64_t* tgt_loc, const int64_t* src_loc, int64_t num_locs, int64_t num_slots, void* workspace,
                                 int64_t workspace_bytes, int64_t sum_row_bytes, fl_stream_t stream) {
  FL_CHECK_ARG(data_ptrs && row_bytes && row_prefix && (num_locs == 0 || (tgt_loc && src_loc)), "fl_kv_move_staged: null pointer");
  FL_CHECK_ARG(num_buffers >= 0 && num_buffers <= 65535 && num_locs >= 0 && sum_row_bytes > 0 && sum_row_bytes % 4 == 0,
               "fl_kv_move_staged: num_buffers=%d num_locs=%lld sum_row_bytes=%lld", num_buffers, (long long)num_locs, (long long)sum_row_bytes);
  if (num_locs == 0 || num_buffers == 0) return FL_OK;
  FL_CHECK_ARG(workspace && workspace_bytes >= num_locs * sum_row_bytes, "fl_kv_move_staged: workspace of %lld bytes, %lld needed (rows x sum of row bytes)",
               (long long)workspace_bytes, (long long)(num_locs * sum_row_bytes));
  const long long tiles = (num_locs + 63) / 64;
  FL_CHECK_ARG(tiles <= 65535, "fl_kv_move_staged: %lld rows per call (max %d)", (long long)num_locs, 65535 * 64);
  const dim3 grid((unsigned)num_buffers, (unsigned)tiles);
  kv_move_staged_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>((const unsigned long long*)data_ptrs, (const long long*)row_bytes,
                                                                       (const long long*)row_prefix, (const long long*)tgt_loc,
                                                                       (const long long*)src_loc, num_locs, num_slots, (uint8_t*)workspace);
  FL_CHECK_LAUNCH("fl_kv_move_staged (gather)");
  kv_move_staged_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>((const unsigned long long*)data_ptrs, (const long long*)row_bytes,
                                                                      (const long long*)row_prefix, (const long long*)tgt_loc,
                                                                      (const long long*)src_loc, num_locs, num_slots, (uint8_t*)workspace);
  FL_CHECK_LAUNCH("fl_kv_move_staged (scatter)");
  return FL_OK;
}
