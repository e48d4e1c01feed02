// C5 / C6 as ONE kernel over peer-mapped buffers — the MI355X form of the reference's one-shot fused collectives
// (flashinfer.comm.trtllm_allreduce_fusion / trtllm_reducescatter_fusion over the IPC workspace of
// /root/reference/python/sglang/srt/layers/flashinfer_comm_fusion.py:64-109; call sites :286-401, :404-513; the same
// transport idea as vllm's custom all-reduce, srt/distributed/device_communicators/custom_all_reduce.py:43-44,284-297).
//
// xGMI is a point-to-point mesh: for a decode step's few dozen token rows the exchange is latency-, not bandwidth-bound,
// so every rank WRITES its rows straight into each destination's inbox over its own link (posted stores on all 7 links
// at once, no ring, no second launch), raises a per-row flag, and the workgroup that owns a row spins on that row's
// flags only — no grid-wide barrier, one workgroup per token row end to end:
//     push row -> fence -> flags -> wait for the peers' flags of MY row -> sum in rank order (+ add_in, + residual) ->
//     RMSNorm -> optional 1x128 e4m3 quantisation                      (row work: norm_row.h, same code as norm_fused.hip)
// Protocol, layout, and why two buffer parities suffice: comm_protocol.h.  The workspace is allocated uncached
// (fine-grained): a peer's stores must become visible to a kernel that is already running here.
// A wait that exceeds its time budget (default 120 s: far beyond any plausible rank skew — GC pause, module load, weight
// reload; fl_comm_set_timeout changes it) is FATAL for the communicator, never silent: the kernel writes NaN into every
// output row it owns, sets the sticky error word in the workspace AND in a host-mapped word, and does not advance the
// epoch.  Later launches — also replays of a captured graph — see the error word first, poison their outputs at once and
// touch no peer; the next host-side launch call (and fl_comm_check) returns an error, which the Python route raises.
// A lost peer therefore cannot hang the GPU, and a late one cannot make a rank consume rows of the wrong epoch.
#include <string.h>

#include <chrono>
#include <thread>

#include "comm_protocol.h"
#include "norm_row.h"

namespace {

constexpr int kMaxWorld = 16;
constexpr long long kMaxOneShotTokens = 1024;   // every workgroup of the launch must be resident: they wait on each other's peers

struct Peers {
  uint8_t* ws[kMaxWorld];
};

struct FlComm {
  int rank, world;
  FlCommLayout L;
  uint8_t* local;
  uint8_t* peer[kMaxWorld];
  bool opened[kMaxWorld];
  bool connected;
  double timeout_s;
  unsigned* host_err;       // host-mapped word (hipHostMalloc): != 0 once any kernel of this communicator has timed out
  unsigned* host_err_dev;   // its device address
};

__device__ __forceinline__ void store_flag(uint8_t* ws, const long long idx, const unsigned e) {
  unsigned* f = reinterpret_cast<unsigned*>(ws + fl_comm_flags_offset()) + idx;
  __hip_atomic_store(f, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// true = the flag arrived
__device__ __forceinline__ bool wait_flag(uint8_t* ws, const long long idx, const unsigned e, const unsigned long long budget) {
  unsigned* f = reinterpret_cast<unsigned*>(ws + fl_comm_flags_offset()) + idx;
  const unsigned long long t0 = wall_clock64();   // constant 100 MHz
  while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != e) {
    if (wall_clock64() - t0 > budget) return false;
    __builtin_amdgcn_s_sleep(4);
  }
  return true;
}

// the outputs of row `row` as NaN (bf16 0x7FC0, e4m3fn 0x7F, f32 NaN scales): what a failed exchange leaves behind
__device__ __forceinline__ void poison_row(const long long row, const int H, uint16_t* residual_out, uint16_t* norm_out,
                                           uint8_t* quant_out, float* scale_out, const long long ss_t, const long long ss_g) {
  for (int col = threadIdx.x * 8; col < H; col += 256 * 8) {
    const uint4 nan16 = make_uint4(0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u);
    if (residual_out != nullptr) *reinterpret_cast<uint4*>(residual_out + row * H + col) = nan16;
    if (norm_out != nullptr) *reinterpret_cast<uint4*>(norm_out + row * H + col) = nan16;
    if (quant_out != nullptr) {
      *reinterpret_cast<uint2*>(quant_out + row * H + col) = make_uint2(0x7F7F7F7Fu, 0x7F7F7F7Fu);
      if ((col & 127) == 0) scale_out[row * ss_t + (col >> 7) * ss_g] = __uint_as_float(0x7FC00000u);
    }
  }
}

// grid = T + 1 workgroups: block t < T owns token row t, block T keeps the ranks in step (the sync row)
template <bool kRS>
__global__ __launch_bounds__(256) void oneshot_kernel(const Peers peers, const int rank, const FlCommLayout L,
                                                      const uint16_t* __restrict__ in, const long long T, const int H,
                                                      const uint16_t* __restrict__ add_in, const uint16_t* __restrict__ residual_in,
                                                      const uint16_t* __restrict__ gamma, const float eps,
                                                      uint16_t* __restrict__ residual_out, uint16_t* __restrict__ norm_out,
                                                      uint8_t* __restrict__ quant_out, float* __restrict__ scale_out,
                                                      const long long ss_t, const long long ss_g,
                                                      const unsigned long long budget, unsigned* __restrict__ host_err) {
  __shared__ float wsum[4];
  __shared__ unsigned s_epoch;
  __shared__ int s_fail;
  __shared__ unsigned s_dead;
  uint8_t* me = peers.ws[rank];
  FlCommState* st = reinterpret_cast<FlCommState*>(me);
  const int tid = threadIdx.x;
  const int W = L.world;
  if (tid == 0) {
    s_epoch = __hip_atomic_load(&st->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_dead = __hip_atomic_load(&st->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    s_fail = 0;
  }
  __syncthreads();
  const unsigned e = s_epoch;
  const int par = (int)(e & 1u);
  const long long t = blockIdx.x;
  bool consume = false;
  long long row_l = 0;
  if (s_dead != 0u) {   // an earlier launch of this communicator timed out: no push, no wait, NaN outputs, epoch untouched
    if (t < T) {
      const int owner = kRS ? fl_comm_owner(T, W, t) : 0;
      if (!kRS || owner == rank)
        poison_row(kRS ? t - fl_comm_slice_lo(T, W, owner) : t, H, residual_out, norm_out, quant_out, scale_out, ss_t, ss_g);
    }
    return;
  }
  if (t < T) {
    // ---- push: this rank's row t into the inbox of its destination(s) ----
    uint4 r[fl_norm::kRowChunks];
#pragma unroll
    for (int c = 0; c < fl_norm::kRowChunks; ++c) {
      const int col = (c * 256 + tid) * 8;
      r[c] = col < H ? *reinterpret_cast<const uint4*>(in + t * H + col) : make_uint4(0, 0, 0, 0);
    }
    const int owner = kRS ? fl_comm_owner(T, W, t) : 0;
    const long long row_d = kRS ? t - fl_comm_slice_lo(T, W, owner) : t;
    for (int p = kRS ? owner : 0; p < (kRS ? owner + 1 : W); ++p) {
      uint16_t* dst = reinterpret_cast<uint16_t*>(peers.ws[p] + fl_comm_inbox_offset(L)) + fl_comm_inbox_row(L, par, rank, row_d);
#pragma unroll
      for (int c = 0; c < fl_norm::kRowChunks; ++c) {
        const int col = (c * 256 + tid) * 8;
        if (col < H) *reinterpret_cast<uint4*>(dst + col) = r[c];
      }
    }
    __threadfence_system();
    __syncthreads();
    if (kRS) {
      if (tid == 0) store_flag(peers.ws[owner], fl_comm_flag_index(L, par, rank, row_d), e);
    } else if (tid < W) {
      store_flag(peers.ws[tid], fl_comm_flag_index(L, par, rank, row_d), e);
    }
    // ---- wait: the owner of the row collects every source's copy ----
    consume = !kRS || owner == rank;
    row_l = row_d;
    if (consume && tid < W && !wait_flag(me, fl_comm_flag_index(L, par, tid, row_l), e, budget)) s_fail = 1;
  } else if (tid < W) {
    store_flag(peers.ws[tid], fl_comm_flag_index(L, par, W, rank), e);
    if (!wait_flag(me, fl_comm_flag_index(L, par, W, tid), e, budget)) s_fail = 1;
  }
  __syncthreads();
  if (s_fail) {
    if (tid == 0) {
      __hip_atomic_store(&st->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (consume) poison_row(row_l, H, residual_out, norm_out, quant_out, scale_out, ss_t, ss_g);
  } else if (consume) {
    const uint16_t* xrow = reinterpret_cast<const uint16_t*>(me + fl_comm_inbox_offset(L)) + fl_comm_inbox_row(L, par, 0, row_l);
    fl_norm::add_rmsnorm_row(xrow, W, L.max_tokens * (long long)L.hidden, add_in, residual_in, gamma, eps, row_l, H, residual_out,
                             norm_out, quant_out, scale_out, ss_t, ss_g, wsum);
  }
  // ---- the last workgroup out advances the epoch (the next launch on this stream starts after this one has ended) —
  //      unless any workgroup of this launch timed out: the ranks are out of step then, and the epoch stays where it is ----
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    const unsigned old = atomicAdd(&st->arrive, 1u);
    if (old == gridDim.x - 1) {
      st->arrive = 0;
      __threadfence();
      if (__hip_atomic_load(&st->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u)
        __hip_atomic_store(&st->epoch, e + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// C7 / C3 all-gather as ONE kernel (flashinfer.comm.trtllm_allgather_fusion, flashinfer_comm_fusion.py:516-640; the all-gather
// half of eps.communication.TPDPConvertor, layers/dp_attention.py:62-74): token rows are split over the ranks as
// get_num_tokens_per_rank does; workgroup t owns global row t — the rank that HOLDS the row pushes it into every rank's
// inbox[parity][owner][row - lo(owner)] and raises that row's flag, every rank waits for the flag in its own inbox, copies
// the row to out[t] and (C7) runs the dual RMSNorm on it (one wave; in place for the kv columns, q columns -> x_norm_out +
// optional 1x128 fp8).  Same epochs / parities / sync row / failure handling as oneshot_kernel above.
__global__ __launch_bounds__(256) void oneshot_ag_kernel(const Peers peers, const int rank, const FlCommLayout L,
                                                         const uint16_t* __restrict__ in, const long long T, const int D,
                                                         uint16_t* __restrict__ out, const int q_rank, const int kv_rank,
                                                         const uint16_t* __restrict__ gamma_q, const uint16_t* __restrict__ gamma_kv,
                                                         const float eps_q, const float eps_kv, uint16_t* __restrict__ x_norm_out,
                                                         uint8_t* __restrict__ quant_out, float* __restrict__ scale_out,
                                                         const long long ss_t, const long long ss_g, const unsigned long long budget,
                                                         unsigned* __restrict__ host_err) {
  __shared__ unsigned s_epoch;
  __shared__ int s_fail;
  __shared__ unsigned s_dead;
  uint8_t* me = peers.ws[rank];
  FlCommState* st = reinterpret_cast<FlCommState*>(me);
  const int tid = threadIdx.x;
  const int W = L.world;
  if (tid == 0) {
    s_epoch = __hip_atomic_load(&st->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_dead = __hip_atomic_load(&st->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    s_fail = 0;
  }
  __syncthreads();
  const unsigned e = s_epoch;
  const int par = (int)(e & 1u);
  const long long t = blockIdx.x;
  auto poison = [&]() {   // NaN into everything this workgroup would have written
    const uint4 nan16 = make_uint4(0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u);
    for (int col = tid * 8; col < D; col += 256 * 8) *reinterpret_cast<uint4*>(out + t * D + col) = nan16;
    if (q_rank > 0) poison_row(t, q_rank, nullptr, x_norm_out, quant_out, scale_out, ss_t, ss_g);
  };
  if (s_dead != 0u) {
    if (t < T) poison();
    return;
  }
  if (t < T) {
    const int owner = fl_comm_owner(T, W, t);
    const long long row_d = t - fl_comm_slice_lo(T, W, owner);
    if (owner == rank) {   // this rank holds the row: push it to everyone (itself included)
      for (int col = tid * 8; col < D; col += 256 * 8) {
        const uint4 v = *reinterpret_cast<const uint4*>(in + row_d * D + col);
        for (int p = 0; p < W; ++p)
          *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(peers.ws[p] + fl_comm_inbox_offset(L)) +
                                    fl_comm_inbox_row(L, par, rank, row_d) + col) = v;
      }
      __threadfence_system();
      __syncthreads();
      if (tid < W) store_flag(peers.ws[tid], fl_comm_flag_index(L, par, rank, row_d), e);
    }
    if (tid == 0 && !wait_flag(me, fl_comm_flag_index(L, par, owner, row_d), e, budget)) s_fail = 1;
    __syncthreads();
    if (!s_fail) {
      const uint16_t* src = reinterpret_cast<const uint16_t*>(me + fl_comm_inbox_offset(L)) + fl_comm_inbox_row(L, par, owner, row_d);
      for (int col = tid * 8; col < D; col += 256 * 8)
        *reinterpret_cast<uint4*>(out + t * D + col) = *reinterpret_cast<const uint4*>(src + col);
      if (q_rank > 0) {
        __syncthreads();   // the row is complete in `out` (this workgroup's own stores: visible to its wave 0)
        if (tid < 64)
          fl_norm::dual_rmsnorm_row(out + t * D, t, q_rank, kv_rank, gamma_q, gamma_kv, eps_q, eps_kv, x_norm_out, quant_out,
                                    scale_out, ss_t, ss_g, tid);
      }
    }
  } else if (tid < W) {
    store_flag(peers.ws[tid], fl_comm_flag_index(L, par, W, rank), e);
    if (!wait_flag(me, fl_comm_flag_index(L, par, W, tid), e, budget)) s_fail = 1;
  }
  __syncthreads();
  if (s_fail) {
    if (tid == 0) {
      __hip_atomic_store(&st->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (t < T) poison();
  }
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    const unsigned old = atomicAdd(&st->arrive, 1u);
    if (old == gridDim.x - 1) {
      st->arrive = 0;
      __threadfence();
      if (__hip_atomic_load(&st->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u)
        __hip_atomic_store(&st->epoch, e + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}


// C1 / C2 exchange as ONE kernel: the equal-split all-to-all of eps.fast_ep.AllToAll's peer slabs (fast_ep.py:45-78; the reference's
// AllToAll sits on an MSCCL++ communicator, distributed/parallel_state.py:965-977) on the peer-mapped transport instead of an RCCL
// all_to_all_single.  `send` holds `world` slabs of `cap` rows of D 2-byte elements (slab p goes to rank p); `recv` gets slab s from
// rank s.  Workgroup (p, r) pushes row r of slab p into rank p's inbox[parity][rank][r], raises that row's flag there, then waits for
// the flag of the row rank p sent HERE as its row r and copies it out — every workgroup is sender and receiver of one row, nobody waits
// for a grid.  A decode step's slabs are mostly empty rows (a token occupies one row per peer it routes to): with ids_col >= 0 the K
// int32 expert ids in a row's tail say so (all < 0), and only the tail of such a row travels and is copied out.
__global__ __launch_bounds__(256) void oneshot_a2a_kernel(const Peers peers, const int rank, const FlCommLayout L,
                                                          const uint16_t* __restrict__ send, uint16_t* __restrict__ recv, const int cap,
                                                          const int D, const int ids_col /*in 2-byte elements, -1: none*/, const int K,
                                                          const unsigned long long budget, unsigned* __restrict__ host_err) {
  __shared__ unsigned s_epoch;
  __shared__ int s_fail;
  __shared__ unsigned s_dead;
  __shared__ int s_live;
  uint8_t* me = peers.ws[rank];
  FlCommState* st = reinterpret_cast<FlCommState*>(me);
  const int tid = threadIdx.x;
  const int W = L.world;
  if (tid == 0) {
    s_epoch = __hip_atomic_load(&st->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_dead = __hip_atomic_load(&st->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    s_fail = 0;
  }
  __syncthreads();
  const unsigned e = s_epoch;
  const int par = (int)(e & 1u);
  const long long b = blockIdx.x;
  const long long rows = (long long)W * cap;
  const int p = (int)(b / cap), r = (int)(b % cap);
  // columns a row moves: everything, or (a row nobody routed to) the 16-byte-aligned span that holds its ids
  const int tail_lo = ids_col >= 0 ? (ids_col / 8) * 8 : 0;
  auto row_is_live = [&](const uint16_t* row) -> bool {   // (uniform per workgroup: thread 0 decides)
    if (ids_col < 0) return true;
    if (tid == 0) {
      const int32_t* ids = reinterpret_cast<const int32_t*>(row + ids_col);
      int live = 0;
      for (int k = 0; k < K; ++k) live |= ids[k] >= 0 ? 1 : 0;
      s_live = live;
    }
    __syncthreads();
    const bool v = s_live != 0;
    __syncthreads();
    return v;
  };
  auto poison = [&]() {
    const uint4 nan16 = make_uint4(0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u);
    for (int col = tid * 8; col < D; col += 256 * 8) *reinterpret_cast<uint4*>(recv + b * D + col) = nan16;
  };
  if (s_dead != 0u) {
    if (b < rows) poison();
    return;
  }
  if (b < rows) {
    const uint16_t* src = send + b * D;
    const bool live = row_is_live(src);
    uint16_t* dst = reinterpret_cast<uint16_t*>(peers.ws[p] + fl_comm_inbox_offset(L)) + fl_comm_inbox_row(L, par, rank, r);
    for (int col = (live ? 0 : tail_lo) + tid * 8; col < D; col += 256 * 8)
      *reinterpret_cast<uint4*>(dst + col) = *reinterpret_cast<const uint4*>(src + col);
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
      store_flag(peers.ws[p], fl_comm_flag_index(L, par, rank, r), e);
      if (!wait_flag(me, fl_comm_flag_index(L, par, p, r), e, budget)) s_fail = 1;
    }
    __syncthreads();
    if (!s_fail) {
      const uint16_t* in = reinterpret_cast<const uint16_t*>(me + fl_comm_inbox_offset(L)) + fl_comm_inbox_row(L, par, p, r);
      const bool got = row_is_live(in);
      for (int col = (got ? 0 : tail_lo) + tid * 8; col < D; col += 256 * 8)
        *reinterpret_cast<uint4*>(recv + b * D + col) = *reinterpret_cast<const uint4*>(in + col);
    }
  } else if (tid < W) {   // the sync row: nobody starts operation e + 2 before every peer has finished e (comm_protocol.h)
    store_flag(peers.ws[tid], fl_comm_flag_index(L, par, W, rank), e);
    if (!wait_flag(me, fl_comm_flag_index(L, par, W, tid), e, budget)) s_fail = 1;
  }
  __syncthreads();
  if (s_fail) {
    if (tid == 0) {
      __hip_atomic_store(&st->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (b < rows) poison();
  }
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    const unsigned old = atomicAdd(&st->arrive, 1u);
    if (old == gridDim.x - 1) {
      st->arrive = 0;
      __threadfence();
      if (__hip_atomic_load(&st->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u)
        __hip_atomic_store(&st->epoch, e + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int launch(FlComm* c, bool rs, const void* in, int64_t T, int H, const void* add_in, const void* residual_in, const void* gamma,
           float eps, void* residual_out, void* norm_out, void* quant_out, float* scale_out, int64_t ss_t, int64_t ss_g,
           fl_stream_t stream) {
  FL_CHECK_ARG(c != nullptr && c->connected, "one-shot comm: not connected (fl_comm_connect)");
  if (__atomic_load_n(c->host_err, __ATOMIC_RELAXED) != 0u) {   // (a plain host read: no device synchronisation)
    fl_set_error("one-shot comm: an earlier launch timed out waiting for a peer (rank %d of %d); the communicator is dead, "
                 "its outputs since then are NaN — re-create it", c->rank, c->world);
    return FL_ERR_LAUNCH;
  }
  FL_CHECK_ARG(T >= 0 && T <= c->L.max_tokens * (rs ? c->world : 1) && T <= kMaxOneShotTokens,
               "one-shot comm: T=%lld exceeds the workspace (max_tokens %lld)", (long long)T, c->L.max_tokens);
  FL_CHECK_ARG(H > 0 && H % 8 == 0 && H <= c->L.hidden && H <= fl_norm::kMaxChunks * 512, "one-shot comm: H=%d (workspace hidden %d)", H,
               c->L.hidden);
  FL_CHECK_ARG(in != nullptr || T == 0, "one-shot comm: null input");
  FL_CHECK_ARG(gamma != nullptr || (norm_out == nullptr && quant_out == nullptr), "one-shot comm: norm needs gamma");
  FL_CHECK_ARG(quant_out == nullptr || (scale_out != nullptr && H % 128 == 0), "one-shot comm: quant needs scales, H %% 128 == 0");
  if (rs) {   // a slice must fit the per-source inbox rows
    FL_CHECK_ARG((T + c->world - 1) / c->world <= c->L.max_tokens, "one-shot comm: token slice exceeds max_tokens");
  }
  Peers peers;
  for (int p = 0; p < kMaxWorld; ++p) peers.ws[p] = p < c->world ? c->peer[p] : nullptr;
  const unsigned long long budget = (unsigned long long)(c->timeout_s * 1e8);
  const dim3 grid((unsigned)T + 1);
  if (rs)
    oneshot_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(
        peers, c->rank, c->L, (const uint16_t*)in, T, H, (const uint16_t*)add_in, (const uint16_t*)residual_in, (const uint16_t*)gamma,
        eps, (uint16_t*)residual_out, (uint16_t*)norm_out, (uint8_t*)quant_out, scale_out, ss_t, ss_g, budget, c->host_err_dev);
  else
    oneshot_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(
        peers, c->rank, c->L, (const uint16_t*)in, T, H, (const uint16_t*)add_in, (const uint16_t*)residual_in, (const uint16_t*)gamma,
        eps, (uint16_t*)residual_out, (uint16_t*)norm_out, (uint8_t*)quant_out, scale_out, ss_t, ss_g, budget, c->host_err_dev);
  FL_CHECK_LAUNCH("oneshot_kernel");
  return FL_OK;
}

}  // namespace

extern "C" int fl_comm_workspace_size(int world, int64_t max_tokens, int hidden, int64_t* bytes_out) {
  FL_CHECK_ARG(world >= 1 && world <= kMaxWorld && max_tokens >= 1 && hidden >= 8 && hidden % 8 == 0 && bytes_out,
               "fl_comm_workspace_size: bad arguments");
  *bytes_out = fl_comm_workspace_bytes(FlCommLayout{world, max_tokens, hidden});
  return FL_OK;
}

extern "C" int fl_comm_create(int rank, int world, int64_t max_tokens, int hidden, void** comm_out) {
  FL_CHECK_ARG(comm_out && world >= 1 && world <= kMaxWorld && rank >= 0 && rank < world, "fl_comm_create: rank %d of %d", rank, world);
  FL_CHECK_ARG(max_tokens >= 1 && max_tokens <= kMaxOneShotTokens && hidden >= 8 && hidden % 8 == 0 && hidden <= 8192,
               "fl_comm_create: max_tokens %lld (<= %lld), hidden %d", (long long)max_tokens, kMaxOneShotTokens, hidden);
  FlComm* c = new FlComm();
  c->rank = rank;
  c->world = world;
  c->L = FlCommLayout{world, max_tokens, hidden};
  c->connected = false;
  c->timeout_s = 10.0;   // budget of ONE flag wait (a healthy exchange takes microseconds); FLUENT_ONESHOT_TIMEOUT_S / fl_comm_set_timeout change it
  c->host_err = c->host_err_dev = nullptr;
  for (int p = 0; p < kMaxWorld; ++p) { c->peer[p] = nullptr; c->opened[p] = false; }
  {
    void* hp = nullptr;
    void* dp = nullptr;
    if (hipHostMalloc(&hp, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
      (void)hipGetLastError();
      if (hp) (void)hipHostFree(hp);
      delete c;
      fl_set_error("fl_comm_create: cannot allocate the host-mapped error word");
      return FL_ERR_LAUNCH;
    }
    memset(hp, 0, 64);
    c->host_err = (unsigned*)hp;
    c->host_err_dev = (unsigned*)dp;
  }
  const size_t bytes = (size_t)fl_comm_workspace_bytes(c->L);
  void* ptr = nullptr;
  hipError_t err = hipExtMallocWithFlags(&ptr, bytes, hipDeviceMallocUncached);
  if (err != hipSuccess) {
    (void)hipGetLastError();
    err = hipExtMallocWithFlags(&ptr, bytes, hipDeviceMallocFinegrained);
  }
  if (err != hipSuccess) {
    (void)hipHostFree(c->host_err);
    delete c;
    fl_set_error("fl_comm_create: cannot allocate %zu bytes of fine-grained device memory: %s", bytes, hipGetErrorString(err));
    return FL_ERR_LAUNCH;
  }
  c->local = (uint8_t*)ptr;
  FlCommState st{};
  st.epoch = 1;
  if (hipMemset(ptr, 0, bytes) != hipSuccess || hipMemcpy(ptr, &st, sizeof(st), hipMemcpyHostToDevice) != hipSuccess ||
      hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(ptr);
    (void)hipHostFree(c->host_err);
    delete c;
    fl_set_error("fl_comm_create: cannot initialise the workspace");
    return FL_ERR_LAUNCH;
  }
  c->peer[rank] = c->local;
  if (world == 1) c->connected = true;
  *comm_out = c;
  return FL_OK;
}

extern "C" int fl_comm_local_handle(void* comm, void* handle_out /* 64 bytes */) {
  FlComm* c = (FlComm*)comm;
  FL_CHECK_ARG(c && handle_out, "fl_comm_local_handle: null argument");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  hipIpcMemHandle_t h;
  const hipError_t err = hipIpcGetMemHandle(&h, c->local);
  if (err != hipSuccess) {
    fl_set_error("fl_comm_local_handle: hipIpcGetMemHandle: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 needed on this driver)", hipGetErrorString(err));
    return FL_ERR_LAUNCH;
  }
  memcpy(handle_out, &h, 64);
  return FL_OK;
}

extern "C" int fl_comm_connect(void* comm, const void* handles /* world x 64 bytes, rank order; may be null at world 1 */) {
  FlComm* c = (FlComm*)comm;
  FL_CHECK_ARG(c && (handles || c->world == 1), "fl_comm_connect: null argument");
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank || c->opened[p]) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, (const uint8_t*)handles + 64 * p, 64);
    void* ptr = nullptr;
    const hipError_t err = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (err != hipSuccess) {
      fl_set_error("fl_comm_connect: hipIpcOpenMemHandle(rank %d): %s", p, hipGetErrorString(err));
      return FL_ERR_LAUNCH;
    }
    c->peer[p] = (uint8_t*)ptr;
    c->opened[p] = true;
  }
  c->connected = true;
  return FL_OK;
}

extern "C" int fl_comm_set_timeout(void* comm, double seconds) {
  FlComm* c = (FlComm*)comm;
  FL_CHECK_ARG(c && seconds > 0 && seconds <= 3600, "fl_comm_set_timeout: bad arguments");
  c->timeout_s = seconds;
  return FL_OK;
}

extern "C" int fl_allreduce_fused(void* comm, const void* in, int64_t T, int H, const void* residual_in, const void* gamma, float eps,
                                  void* residual_out, void* norm_out, void* quant_out, float* scale_out, int64_t s_stride_t,
                                  int64_t s_stride_g, fl_stream_t stream) {
  return launch((FlComm*)comm, false, in, T, H, nullptr, residual_in, gamma, eps, residual_out, norm_out, quant_out, scale_out,
                s_stride_t, s_stride_g, stream);
}

extern "C" int fl_reducescatter_fused(void* comm, const void* in, int64_t T, int H, const void* add_in, const void* residual_in,
                                      const void* gamma, float eps, void* residual_out, void* norm_out, void* quant_out,
                                      float* scale_out, int64_t s_stride_t, int64_t s_stride_g, fl_stream_t stream) {
  return launch((FlComm*)comm, true, in, T, H, add_in, residual_in, gamma, eps, residual_out, norm_out, quant_out, scale_out,
                s_stride_t, s_stride_g, stream);
}

extern "C" int fl_allgather_fused(void* comm, const void* in, int64_t t_cur, int64_t T, int D, void* out, int q_rank, int kv_rank,
                                  const void* gamma_q, const void* gamma_kv, float eps_q, float eps_kv, void* x_norm_out,
                                  void* quant_out, float* scale_out, int64_t s_stride_t, int64_t s_stride_g, fl_stream_t stream) {
  FlComm* c = (FlComm*)comm;
  FL_CHECK_ARG(c != nullptr && c->connected, "one-shot comm: not connected (fl_comm_connect)");
  if (__atomic_load_n(c->host_err, __ATOMIC_RELAXED) != 0u) {
    fl_set_error("one-shot comm: an earlier launch timed out waiting for a peer (rank %d of %d); the communicator is dead, "
                 "its outputs since then are NaN — re-create it", c->rank, c->world);
    return FL_ERR_LAUNCH;
  }
  FL_CHECK_ARG(T >= 0 && T <= kMaxOneShotTokens && (T + c->world - 1) / c->world <= c->L.max_tokens,
               "one-shot all-gather: T=%lld exceeds the workspace (max_tokens %lld per rank)", (long long)T, c->L.max_tokens);
  const long long mine = fl_comm_slice_lo(T, c->world, c->rank + 1) - fl_comm_slice_lo(T, c->world, c->rank);
  FL_CHECK_ARG(t_cur == mine, "one-shot all-gather: this rank holds %lld rows, get_num_tokens_per_rank gives it %lld of %lld",
               (long long)t_cur, mine, (long long)T);
  FL_CHECK_ARG(D > 0 && D % 8 == 0 && D <= c->L.hidden, "one-shot all-gather: D=%d (workspace hidden %d)", D, c->L.hidden);
  FL_CHECK_ARG(out != nullptr && (in != nullptr || t_cur == 0), "one-shot all-gather: null tensor");
  if (q_rank > 0) {
    FL_CHECK_ARG(gamma_q && gamma_kv && q_rank % 8 == 0 && q_rank <= 2048 && kv_rank > 0 && kv_rank % 8 == 0 && kv_rank <= 1024 &&
                     q_rank + kv_rank <= D, "one-shot all-gather: D=%d q_rank=%d kv_rank=%d", D, q_rank, kv_rank);
    FL_CHECK_ARG(quant_out == nullptr || (scale_out != nullptr && q_rank % 128 == 0), "one-shot all-gather: quant needs scales");
  }
  Peers peers;
  for (int p = 0; p < kMaxWorld; ++p) peers.ws[p] = p < c->world ? c->peer[p] : nullptr;
  const unsigned long long budget = (unsigned long long)(c->timeout_s * 1e8);
  oneshot_ag_kernel<<<dim3((unsigned)T + 1), 256, 0, (hipStream_t)stream>>>(
      peers, c->rank, c->L, (const uint16_t*)in, T, D, (uint16_t*)out, q_rank, kv_rank, (const uint16_t*)gamma_q,
      (const uint16_t*)gamma_kv, eps_q, eps_kv, (uint16_t*)x_norm_out, (uint8_t*)quant_out, scale_out, s_stride_t, s_stride_g, budget,
      c->host_err_dev);
  FL_CHECK_LAUNCH("oneshot_ag_kernel");
  return FL_OK;
}

extern "C" int fl_alltoall_oneshot(void* comm, const void* send, void* recv, int cap, int D, int ids_col, int top_k, fl_stream_t stream) {
  FlComm* c = (FlComm*)comm;
  FL_CHECK_ARG(c != nullptr && c->connected, "one-shot comm: not connected (fl_comm_connect)");
  if (__atomic_load_n(c->host_err, __ATOMIC_RELAXED) != 0u) {
    fl_set_error("one-shot comm: an earlier launch timed out waiting for a peer (rank %d of %d); the communicator is dead, "
                 "its outputs since then are NaN — re-create it", c->rank, c->world);
    return FL_ERR_LAUNCH;
  }
  FL_CHECK_ARG(cap >= 1 && cap <= c->L.max_tokens && (long long)cap * c->world <= kMaxOneShotTokens,
               "one-shot all-to-all: %d rows per peer x %d ranks exceed the workspace (max_tokens %lld, %lld rows per launch)", cap, c->world,
               c->L.max_tokens, kMaxOneShotTokens);
  FL_CHECK_ARG(D > 0 && D % 8 == 0 && D <= c->L.hidden, "one-shot all-to-all: D=%d (workspace hidden %d)", D, c->L.hidden);
  FL_CHECK_ARG(send != nullptr && recv != nullptr && send != recv, "one-shot all-to-all: null or aliased buffers");
  FL_CHECK_ARG(ids_col < 0 || (top_k >= 1 && ids_col % 2 == 0 && ids_col + 2 * top_k <= D),
               "one-shot all-to-all: the %d expert ids at element %d do not fit a row of %d", top_k, ids_col, D);
  Peers peers;
  for (int p = 0; p < kMaxWorld; ++p) peers.ws[p] = p < c->world ? c->peer[p] : nullptr;
  const unsigned long long budget = (unsigned long long)(c->timeout_s * 1e8);
  oneshot_a2a_kernel<<<dim3((unsigned)(cap * c->world) + 1), 256, 0, (hipStream_t)stream>>>(
      peers, c->rank, c->L, (const uint16_t*)send, (uint16_t*)recv, cap, D, ids_col, top_k, budget, c->host_err_dev);
  FL_CHECK_LAUNCH("oneshot_a2a_kernel");
  return FL_OK;
}

extern "C" int fl_comm_check(void* comm) {   // synchronises the device: not for the hot path
  FlComm* c = (FlComm*)comm;
  FL_CHECK_ARG(c, "fl_comm_check: null argument");
  FlCommState st;
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&st, c->local, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess) {
    fl_set_error("fl_comm_check: device error: %s", hipGetErrorString(hipGetLastError()));
    return FL_ERR_LAUNCH;
  }
  if (st.error != 0) {
    fl_set_error("one-shot comm: a wait for a peer's flag timed out (rank %d of %d, epoch %u)", c->rank, c->world, st.epoch);
    return FL_ERR_LAUNCH;
  }
  return FL_OK;
}

namespace {
// Every XCD writes its L2 back (a system-scope release fence per workgroup; far more workgroups than XCDs).  fl_comm_destroy runs it in front of
// hipFree of the UNCACHED workspace: round 6's suite soak saw a contiguous, line-aligned range of ANOTHER tensor written by the communicator's
// last kernel read back stale right after the workspace was freed (5 of ~26 full-suite runs, never with the communicator alive) — dirty lines
// must not depend on what the driver does to the caches when it unmaps uncached memory.
__global__ void l2_writeback_kernel() { __threadfence_system(); }
}  // namespace

extern "C" int fl_comm_destroy(void* comm) {
  FlComm* c = (FlComm*)comm;
  if (!c) return FL_OK;
  (void)hipDeviceSynchronize();
  l2_writeback_kernel<<<256, 64>>>();
  (void)hipDeviceSynchronize();
  for (int p = 0; p < c->world; ++p)
    if (c->opened[p]) (void)hipIpcCloseMemHandle(c->peer[p]);
  (void)hipFree(c->local);
  if (c->host_err) (void)hipHostFree(c->host_err);
  delete c;
  return FL_OK;
}

// ---- the same protocol on plain host memory: what the CPU-side multi-process protocol test drives (tests/ only; nothing
//      on the product path calls these).  `ws[p]` = rank p's workspace as mapped in THIS process (shared memory). ----
extern "C" int fl_comm_host_init(void* ws, int world, int64_t max_tokens, int hidden) {
  FL_CHECK_ARG(ws && world >= 1 && world <= kMaxWorld, "fl_comm_host_init: bad arguments");
  const FlCommLayout L{world, max_tokens, hidden};
  memset(ws, 0, (size_t)fl_comm_workspace_bytes(L));
  reinterpret_cast<FlCommState*>(ws)->epoch = 1;
  return FL_OK;
}

extern "C" int fl_comm_host_exchange(void* const* ws, int rank, int world, int64_t max_tokens, int hidden, int reduce_scatter,
                                     const uint16_t* in /*bf16 [T, H]*/, int64_t T, int H, float* out /*f32 [rows, H]*/,
                                     double timeout_s) {
  FL_CHECK_ARG(ws && rank >= 0 && rank < world && world <= kMaxWorld && H <= hidden && out, "fl_comm_host_exchange: bad arguments");
  const FlCommLayout L{world, max_tokens, hidden};
  uint8_t* me = (uint8_t*)ws[rank];
  FlCommState* st = reinterpret_cast<FlCommState*>(me);
  const unsigned e = __atomic_load_n(&st->epoch, __ATOMIC_ACQUIRE);
  const int par = (int)(e & 1u);
  auto flag = [&](int p, long long idx) { return reinterpret_cast<unsigned*>((uint8_t*)ws[p] + fl_comm_flags_offset()) + idx; };
  auto wait = [&](long long idx) {
    const auto t0 = std::chrono::steady_clock::now();
    while (__atomic_load_n(flag(rank, idx), __ATOMIC_ACQUIRE) != e) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
      std::this_thread::yield();
    }
    return true;
  };
  for (long long t = 0; t < T; ++t) {   // push + flags, row by row (a device workgroup per row)
    const int owner = reduce_scatter ? fl_comm_owner(T, world, t) : 0;
    const long long row_d = reduce_scatter ? t - fl_comm_slice_lo(T, world, owner) : t;
    for (int p = reduce_scatter ? owner : 0; p < (reduce_scatter ? owner + 1 : world); ++p) {
      uint16_t* dst = reinterpret_cast<uint16_t*>((uint8_t*)ws[p] + fl_comm_inbox_offset(L)) + fl_comm_inbox_row(L, par, rank, row_d);
      memcpy(dst, in + t * H, (size_t)H * 2);
      __atomic_store_n(flag(p, fl_comm_flag_index(L, par, rank, row_d)), e, __ATOMIC_RELEASE);
    }
  }
  for (int p = 0; p < world; ++p) __atomic_store_n(flag(p, fl_comm_flag_index(L, par, world, rank)), e, __ATOMIC_RELEASE);
  bool ok = true;
  for (int p = 0; p < world && ok; ++p) ok = wait(fl_comm_flag_index(L, par, world, p));
  const long long lo = reduce_scatter ? fl_comm_slice_lo(T, world, rank) : 0;
  const long long hi = reduce_scatter ? fl_comm_slice_lo(T, world, rank + 1) : T;
  for (long long r = 0; r < hi - lo && ok; ++r) {
    for (int s = 0; s < world && ok; ++s) ok = wait(fl_comm_flag_index(L, par, s, r));
    if (!ok) break;
    for (int col = 0; col < H; ++col) {
      float acc = 0.f;
      for (int s = 0; s < world; ++s) {
        const uint16_t v = (reinterpret_cast<const uint16_t*>(me + fl_comm_inbox_offset(L)) + fl_comm_inbox_row(L, par, s, r))[col];
        uint32_t bits = (uint32_t)v << 16;
        float f;
        memcpy(&f, &bits, 4);
        acc += f;
      }
      out[r * H + col] = acc;
    }
  }
  if (!ok) {
    st->error = 1;
    fl_set_error("fl_comm_host_exchange: rank %d timed out waiting for a flag at epoch %u", rank, e);
    return FL_ERR_LAUNCH;
  }
  __atomic_store_n(&st->epoch, e + 1u, __ATOMIC_RELEASE);
  return FL_OK;
}
