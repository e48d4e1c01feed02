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
// the arrival counters of the in-kernel split merge (mla_decode_fp8_y.hip), bumped and put back by the decode kernel.
// Python statement of the same algorithm: oracle/mla_ref.py:get_mla_metadata.
//
// Round 6: ONE walk per candidate capacity instead of two (64 candidate walks + a recording walk of the winner: 51 us at bs = 128 on 128
// parts, on the host-critical path of every replay in the reference, flashmla_backend.py:380-387).  On the cost axis C[r] = sum_{k<r}
// (tiles_k + FIXED_OVERHEAD) a part that starts at (req, tile) sits at x = C[req] + tile and ends at y = x + P; the next part starts at
// x' = max(y - FIXED_OVERHEAD, C[r']), r' = the first request the part does not finish.  Each lane walks its capacity once — one event
// (request finished / part closed) per iteration, C from LDS two events ahead — and records the start x of every part in LDS;
// the 8-word rows and the split counts follow from the winning lane's records in parallel over the parts (256 threads), no second walk.
// Python statement of this formulation: oracle/mla_ref.py:get_mla_metadata_positions (checked equal to get_mla_metadata on random batches).
// More than 256 parts (no shape of this chip asks for them: fl_mla_num_parts <= CU count) take the two-walk kernel below.
#include "fl_common.h"

namespace {
constexpr int kFixedOverhead = 2;   // (round 4: 4 / 6 / 8 measured on the ragged cfg2 workload and bs = 16..96: within 1 % of 2)
constexpr int kMinSplitCap = 32, kPagesPerSplit = 8;   // parts per request <= max(32, pages / 8) (oracle/mla_ref.py)
constexpr int kMaxBs = 8192;   // tile counts are staged in LDS

// Both walks are FLAT loops — one event (a request consumed, or a part closed) per iteration, selects instead of nested
// loops, the next tile count fetched from LDS an iteration ahead: 64 lanes walking 64 capacities stay convergent, and an
// iteration is ~100 cycles of dependent VALU ops instead of ~290 (LDS latency + divergent branches), measured 139 -> 73 us
// for 128 requests on 256 parts.

// parts the greedy walk needs with capacity P (stops counting beyond limit)
__device__ int parts_needed(const int* nt_of, const int bs, const int P, const int limit) {
  if (bs <= 0) return 0;
  int req = 0, tile = 0, parts = 1, remain = P;
  int nt_cur = nt_of[0];
  while (req < bs && parts <= limit) {
    const int nt_next = nt_of[req + 1 < bs ? req + 1 : req];
    const int need = nt_cur - tile + kFixedOverhead;
    const bool fit = remain >= need;
    const int take = remain - kFixedOverhead;
    remain = fit ? remain - need : P;
    tile = fit ? 0 : (take > 0 ? tile + take : tile);
    parts += fit ? 0 : 1;
    req += fit ? 1 : 0;
    nt_cur = fit ? nt_next : nt_cur;
  }
  return parts;
}

constexpr int kMaxParts = 1024;   // part starts are staged in LDS

__global__ __launch_bounds__(64) void mla_metadata_kernel(const int32_t* __restrict__ seqlens, int bs, int num_parts,
                                                          int32_t* __restrict__ meta, int32_t* __restrict__ num_splits) {
  __shared__ int nt_of[kMaxBs];
  __shared__ unsigned short touched[kMaxBs + 1];   // parts that touch a request (+ a dummy slot for the select-free store)
  __shared__ int ps_req[kMaxParts + 2], ps_tile[kMaxParts + 2], ps_split[kMaxParts + 2];   // (+ end, + dummy)
  __shared__ int s_last;
  const int lane = threadIdx.x;
  int total = 0, nt_max = 0;
  for (int b = lane; b < bs; b += 64) {
    const int L = seqlens[b];
    const int nt = L > 0 ? (L + FL_MLA_PAGE - 1) / FL_MLA_PAGE : 0;
    nt_of[b] = nt;
    total += nt + kFixedOverhead;
    nt_max = nt > nt_max ? nt : nt_max;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    total += __shfl_xor(total, o);
    const int other = __shfl_xor(nt_max, o);
    nt_max = other > nt_max ? other : nt_max;
  }
  __syncthreads();
  int p_min = (total + num_parts - 1) / num_parts;
  if (p_min < 1 + kFixedOverhead) p_min = 1 + kFixedOverhead;
  // no request in more than max(32, pages/8) parts: a part of one or two pages is all prologue / epilogue and the merge of
  // a row grows with its split count (tools/bench_one_batch.py at bs = 1, 2: 32 parts are best up to 16 K tokens, 128 at 64 K)
  const int split_cap = nt_max / kPagesPerSplit > kMinSplitCap ? nt_max / kPagesPerSplit : kMinSplitCap;
  const int p_cap = (nt_max + split_cap - 1) / split_cap + kFixedOverhead;
  if (p_min < p_cap) p_min = p_cap;
  // lane l tests capacity p_min + l; the fixed formula's capacity (p_min + FIXED_OVERHEAD) is the fallback
  const bool ok = parts_needed(nt_of, bs, p_min + lane, num_parts) <= num_parts;
  const unsigned long long okmask = __ballot(ok);
  const int payload = okmask ? p_min + __builtin_ctzll(okmask) : p_min + kFixedOverhead;
  // the walk that records: sequential (one lane, LDS only); parts it never opens start at (bs, 0, 0); the 8-word rows and
  // the cumulative split counts leave the workgroup from all 64 lanes afterwards
  if (lane == 0) {
    int req = 0, tile = 0, split = 0, part = 0, remain = payload;
    int nt_cur = bs > 0 ? nt_of[0] : 0;
    ps_req[0] = 0; ps_tile[0] = 0; ps_split[0] = 0;
    while (req < bs) {
      const int nt_next = nt_of[req + 1 < bs ? req + 1 : req];
      const int need = nt_cur - tile + kFixedOverhead;
      const bool fit = remain >= need || part == num_parts - 1;   // the last part takes whatever is left
      const int take = remain - kFixedOverhead;
      touched[fit ? req : kMaxBs] = (unsigned short)(split + 1);
      remain = fit ? remain - need : payload;
      tile = fit ? 0 : (take > 0 ? tile + take : tile);
      split = fit ? 0 : (take > 0 ? split + 1 : split);
      part += fit ? 0 : 1;
      req += fit ? 1 : 0;
      nt_cur = fit ? nt_next : nt_cur;
      const int slot = fit ? kMaxParts + 1 : part;
      ps_req[slot] = req; ps_tile[slot] = tile; ps_split[slot] = split;
    }
    s_last = part;
  }
  __syncthreads();
  const int last = s_last;
  for (int p = lane; p < num_parts; p += 64) {
    const bool open = p <= last, open_n = p + 1 <= last;
    int4* m = reinterpret_cast<int4*>(meta + p * FL_MLA_META_W);
    m[0] = make_int4(open ? ps_req[p] : bs, open ? ps_tile[p] : 0, open_n ? ps_req[p + 1] : bs, open_n ? ps_tile[p + 1] : 0);
    m[1] = make_int4(open ? ps_split[p] : 0, 0, 0, 0);
  }
  int carry = 0;
  if (lane == 0) num_splits[0] = 0;
  for (int b0 = 0; b0 < bs; b0 += 64) {
    int v = b0 + lane < bs ? (int)touched[b0 + lane] : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(v, o);
      if (lane >= o) v += u;
    }
    if (b0 + lane < bs) num_splits[b0 + lane + 1] = carry + v;
    carry += __shfl(v, 63);
  }
}

// ---- one walk per capacity (num_parts <= kFastParts) ----
constexpr int kFastParts = 256;
__global__ __launch_bounds__(256) void mla_metadata_fast_kernel(const int32_t* __restrict__ seqlens, int bs, int num_parts,
                                                               int32_t* __restrict__ meta, int32_t* __restrict__ num_splits) {
  __shared__ int cost[kMaxBs + 4];                               // C[0..bs], then three "never reached" entries
  __shared__ unsigned short touched[kMaxBs];
  __shared__ int xs[(kFastParts + 2) * 64];                      // [part][lane]: position of the part's start on the cost axis
  __shared__ int s_red[8], s_win[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // tiles per request -> inclusive prefix of (tiles + overhead) over the batch, 256 requests per round
  int carry = 0, nt_max = 0;
  if (tid == 0) cost[0] = 0;
  for (int b0 = 0; b0 < bs; b0 += 256) {
    const int b = b0 + tid;
    const int L = b < bs ? seqlens[b] : 0;
    const int nt = L > 0 ? (L + FL_MLA_PAGE - 1) / FL_MLA_PAGE : 0;
    nt_max = nt > nt_max ? nt : nt_max;
    int v = b < bs ? nt + kFixedOverhead : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(v, o);
      if (lane >= o) v += u;
    }
    if (lane == 63) s_red[wave] = v;
    __syncthreads();
    int base = carry;
    for (int w = 0; w < wave; ++w) base += s_red[w];
    if (b < bs) { cost[b + 1] = base + v; touched[b] = 1; }
    carry += s_red[0] + s_red[1] + s_red[2] + s_red[3];
    __syncthreads();
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int other = __shfl_xor(nt_max, o);
    nt_max = other > nt_max ? other : nt_max;
  }
  if (lane == 0) s_red[4 + wave] = nt_max;
  if (tid < 3) cost[bs + 1 + tid] = 0x7fffffff;
  __syncthreads();
  const int total = carry;
  nt_max = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
  int p_min = (total + num_parts - 1) / num_parts;
  if (p_min < 1 + kFixedOverhead) p_min = 1 + kFixedOverhead;
  const int split_cap = nt_max / kPagesPerSplit > kMinSplitCap ? nt_max / kPagesPerSplit : kMinSplitCap;
  const int p_cap = (nt_max + split_cap - 1) / split_cap + kFixedOverhead;
  if (p_min < p_cap) p_min = p_cap;
  if (wave == 0) {
    // lane l walks capacity p_min + l.  State: r = first unfinished request, y = end of the open part, w0..w2 = C[r..r+2], ld = C[r + 3] in
    // flight (issued at the end of the previous iteration, consumed at the end of this one: the LDS latency sits beside the iteration's VALU
    // work).  SELECTS ONLY (a divergent branch per event costs more than the whole body: hipcc's exec-mask blocks wait for their LDS read), and
    // ONE unconditional store per event: the would-be start of part p + 1 goes to row p + 1 every iteration — the iteration that closes part
    // p writes it last.  A walk that has finished its
    // requests keeps closing empty parts at positions >= total (C[bs + 1] = "never reached" stops the advance): a row holds a real part start
    // iff its position is < total, and the capacity fits iff row num_parts does not.  ~18 instructions per event.
    const int P = p_min + lane;
    int r = 0, p = 0, y = P;
    int w0 = 0, w1 = cost[1], w2 = cost[2], ld = cost[3];
    xs[lane] = 0;
    // (the exit test — every lane has finished its requests or closed num_parts parts — costs a ballot: once per eight events)
    bool busy = bs > 0;
    while (__any(busy)) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool adv = w1 <= y;                                // the open part finishes request r
        const int yo = y - kFixedOverhead;
        const int x = yo > w0 ? yo : w0;
        const int row = p + 1 < num_parts + 1 ? p + 1 : num_parts + 1;
        xs[row * 64 + lane] = x;
        r = adv ? r + 1 : r;
        w0 = adv ? w1 : w0;
        w1 = adv ? w2 : w1;
        w2 = adv ? ld : w2;
        p = adv ? p : p + 1;
        y = adv ? y : x + P;
        ld = cost[r + 3];                                        // (r <= bs: inside the padding)
      }
      busy = r < bs && p < num_parts;
    }
    // rows 0 .. min(p, num_parts) of a lane are exact; it fits iff it finished and row num_parts (if it got that far) is not a real part start
    const bool ok = bs <= 0 || (r >= bs && (p < num_parts || xs[num_parts * 64 + lane] >= total));
    const unsigned long long okmask = __ballot(ok);
    const int win = okmask ? __builtin_ctzll(okmask) : kFixedOverhead;   // (fallback: capacity p_min + FIXED_OVERHEAD, the last part takes the rest)
    const int p_win = __shfl(p, win);
    if (lane == 0) { s_win[0] = win; s_win[1] = p_win < num_parts - 1 ? p_win : num_parts - 1; }
  }
  __syncthreads();
  // part p of the winning capacity starts at x_p: request = the last r with C[r] <= x_p, tile = x_p - C[r]; pieces of that request in earlier
  // parts = p - (the last part q with x_q <= C[r]).  Two binary searches per part, all parts at once.
  const int win = s_win[0];
  int real = 0;                               // part starts of the winning walk that lie inside the batch (positions < total)
  const int p_hi = s_win[1];                  // rows beyond the winner's last close were never written
  for (int p0 = 0; p0 < num_parts; p0 += 256) real += __syncthreads_count(p0 + tid <= p_hi && bs > 0 && xs[(p0 + tid) * 64 + win] < total);
  const int last = real - 1;
  auto decode = [&](const int p, int& req, int& tile, int& split) {
    const int x = xs[p * 64 + win];
    int lo = 0, hi = bs - 1;                 // C[0] = 0 <= x; x < C[bs] for every opened part
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (cost[mid] <= x) lo = mid; else hi = mid - 1;
    }
    req = lo;
    tile = x - cost[lo];
    split = 0;
    if (tile > 0) {
      const int c = cost[lo];
      int a = 0, b = p;                      // x_0 = 0 <= c
      while (a < b) {
        const int mid = (a + b + 1) >> 1;
        if (xs[mid * 64 + win] <= c) a = mid; else b = mid - 1;
      }
      split = p - a;
    }
  };
  for (int p = tid; p < num_parts; p += 256) {
    int req_b = bs, tile_b = 0, split_b = 0, req_e = bs, tile_e = 0, split_e = 0;
    if (p <= last) decode(p, req_b, tile_b, split_b);
    if (p + 1 <= last) decode(p + 1, req_e, tile_e, split_e);
    int4* m = reinterpret_cast<int4*>(meta + p * FL_MLA_META_W);
    m[0] = make_int4(req_b, tile_b, req_e, tile_e);
    m[1] = make_int4(split_b, 0, 0, 0);
    if (p <= last && req_e > req_b) touched[req_b] = (unsigned short)(split_b + 1);   // the part that finishes a request knows its piece count
  }
  __syncthreads();
  int run = 0;
  if (tid == 0) num_splits[0] = 0;
  for (int b0 = 0; b0 < bs; b0 += 256) {
    const int b = b0 + tid;
    int v = b < bs ? (int)touched[b] : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(v, o);
      if (lane >= o) v += u;
    }
    if (lane == 63) s_red[wave] = v;
    __syncthreads();
    int base = run;
    for (int w = 0; w < wave; ++w) base += s_red[w];
    if (b < bs) num_splits[b + 1] = base + v;
    run += s_red[0] + s_red[1] + s_red[2] + s_red[3];
    __syncthreads();
  }
}
}  // namespace

extern "C" int fl_mla_num_parts(int cu_count, int rows_per_kv_head) {
  if (cu_count <= 0 || rows_per_kv_head <= 0) return 1;
  // 64 query rows per workgroup for every shape (mla_decode_fp8_y.hip; at most 32 rows: ONE 32-row workgroup per part)
  const int per_wg = FL_MLA_ROWS_PER_WG;
  const int row_groups = (rows_per_kv_head + per_wg - 1) / per_wg;
  const int parts = cu_count / row_groups;
  return parts > 0 ? parts : 1;
}

extern "C" int fl_mla_get_metadata(const int32_t* cache_seqlens, int bs, int num_parts,
                                   int32_t* tile_scheduler_metadata, int32_t* num_splits, fl_stream_t stream) {
  FL_CHECK_ARG(bs >= 0 && bs <= kMaxBs && num_parts > 0 && num_parts <= kMaxParts,
               "fl_mla_get_metadata: bs=%d (max %d) num_parts=%d (max %d)", bs, kMaxBs, num_parts, kMaxParts);
  FL_CHECK_ARG(cache_seqlens && tile_scheduler_metadata && num_splits, "fl_mla_get_metadata: null pointer");
  // FLUENT_MLA_METADATA_TWO_WALKS=1: the round-1..5 kernel for every shape (A/B runs)
  static const bool two_walks = [] { const char* e = getenv("FLUENT_MLA_METADATA_TWO_WALKS"); return e != nullptr && e[0] == '1'; }();
  if (num_parts <= kFastParts && !two_walks)
    mla_metadata_fast_kernel<<<1, 256, 0, (hipStream_t)stream>>>(cache_seqlens, bs, num_parts, tile_scheduler_metadata, num_splits);
  else
    mla_metadata_kernel<<<1, 64, 0, (hipStream_t)stream>>>(cache_seqlens, bs, num_parts, tile_scheduler_metadata, num_splits);
  FL_CHECK_LAUNCH("fl_mla_get_metadata");
  return FL_OK;
}
