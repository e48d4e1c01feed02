// G1-G4 — FP8 (e4m3) block-scaled (1x128 activations, 128x128 weights) grouped GEMM, bf16 out, gfx950 only.
//
// Replaces deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_{offset,contiguous,masked} and gemm_fp8_fp8_bf16_nt
// (call sites /root/reference/python/sglang/srt/layers/moe/gemms/fp8/fire.py:18 via
//  moe/executors/fp8_eps_executor.py:56,78; moe/executors/deep_ep_executor.py:583-586,607-613,655-662,681-688;
//  dense/gemms/fp8/deep_geem.py:55).  Math = python/sglang/test/test_block_fp8.py:89-141:
//     out[m, n] = sum_kb ( sum_{k in kb} A[m,k] * W[e,n,k] ) * As[m,kb] * Ws[e, n/128, kb]     (fp32 accumulate)
//
// MI355X mapping ("SwapAB"): the WEIGHT rows sit on the MFMA M side and the token rows on the N side
// (D^T[n, m] = W_tile · A_tile^T on v_mfma_scale_f32_32x32x64_f8f6f4, unit E8M0 scales, 2 per 128-wide k block), so
//   * a lane owns ONE token row (lane&31): its per-(token, k-block) scale As[m,kb] * Ws[e,nb,kb] is a single
//     per-lane factor applied to the whole 32x32 partial tile with 16 FMAs — no per-row scale shuffles;
//   * the decode regime (a handful of rows per expert) wastes MFMA columns, not weight bandwidth: every workgroup
//     streams a [128 n x K] weight panel exactly once.
// One workgroup = 4 waves = 128 weight rows x (32*MT) token rows; wave w owns weight rows 32w..32w+31 and all MT
// token tiles.  HBM -> LDS by global_load_lds into a 4-stage ring of [W 16 KiB | A MT*4 KiB | As] k-block stages
// (XOR-swizzled on the source address for conflict-free ds_read_b128), counted vmcnt waits + one raw s_barrier per
// k block.  Tiles of experts with no rows are never scheduled; the grid is an upper bound, surplus blocks exit.
//
// Algorithmic bytes / flops per MoE layer: SURVEY.md §8(d) (weights 44.04 MB per expert hit, 88.08 MFLOP per row).
#include "grouped_gemm_shared.h"
#include <atomic>
#include <cstdlib>

using namespace fl_gemm;

namespace {

// Ring depth of the 128-token tile.  Measured at T=16384 (512 rows per expert), w13 / w2 TFLOP/s: 4 stages + 1 workgroup
// per CU 838 / 492; 3 stages + 1 workgroup 820 / 481; 2 stages + 2 workgroups per CU 1233 / 846 — two INDEPENDENT
// workgroups per CU drift apart, so one feeds the matrix pipe while the other sits in its barrier / LDS-DMA issue / LDS
// reads.  (A 256 x 128 tile with 8 waves in ONE workgroup has fewer bytes per flop but runs in lockstep behind its
// per-k-block barrier: 1171 / 755, software-pipelined with fenced slots 950 / 660.)
#ifndef FL_MT4_STAGES
#define FL_MT4_STAGES 2
#endif
template <int MT>
struct Smem {
  // ring depth: every tile shape runs 2 workgroups per CU (3 stages for the small token tiles, 2 for the 128-token tile)
  // so that one workgroup's barrier / DMA issue / prologue / epilogue overlaps the other's MFMAs
  static constexpr int kStages = MT == 4 ? FL_MT4_STAGES : 3;
  static constexpr int kABytes = MT * 32 * BK;
  static constexpr int kAsFloats = MT * 32 < 64 ? 64 : MT * 32;
  static constexpr int kAsPieces = kAsFloats / 64;
  static constexpr int kPiecesPerWave = 4 + MT + kAsPieces;          // LDS-DMA instructions per wave per stage
  static constexpr int kStageBytes = kWBytes + kABytes + kAsFloats * 4;
  static constexpr int kTotal = kStages * kStageBytes;
};

// One k block for one wave: wait for its stage, refill the stage freed one barrier ago, compute.  The LDS regions are
// distinct __restrict__ parameters of ONE inlined function so that hipcc's waitcnt pass does not assume every ds_read
// may alias the in-flight LDS-DMA and drain it with vmcnt(0) (see mla_decode_fp8.hip).
template <int MT>
__device__ __forceinline__ void kblock_body(v16f (&acc)[MT], const uint8_t* __restrict__ rd_w,
                                            const uint8_t* __restrict__ rd_a, const float* __restrict__ rd_as,
                                            uint8_t* __restrict__ dma_w, uint8_t* __restrict__ dma_a,
                                            float* __restrict__ dma_as, const uint8_t* const (&src_w)[4],
                                            const uint8_t* const (&src_a)[MT],
                                            const float* const (&src_as)[Smem<MT>::kAsFloats / 64], const long long k_off,
                                            const long long as_off, const bool issue, const bool more_in_flight,
                                            const float ws, const int (&rb)[4], const int wave, const int li, const bool w_nt) {
  constexpr int kAsPieces = Smem<MT>::kAsFloats / 64;
  // ---- stage kb landed for every wave; stages kb+1, kb+2 (issued later) may stay in flight ----
  if (more_in_flight) {   // leave the (kStages - 2) later stages in flight
    if constexpr (MT == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");         // 1 stage x (4 + 1 + 1) pieces
    else if constexpr (MT == 2) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");    // 1 x (4 + 2 + 1)
    else if constexpr (FL_MT4_STAGES == 4) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");   // 2 x (4 + 4 + 2)
    else if constexpr (FL_MT4_STAGES == 3) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  // ---- refill the stage every wave finished with before this barrier ----
  if (issue) {
#pragma unroll
    for (int k = 0; k < 4; ++k)   // W: 16 pieces of 8 rows x 128 B; this wave fills pieces 4*wave + k
      fl_dma16_pol(src_w[k] + k_off, dma_w + (wave * 4 + k) * 1024, w_nt);
#pragma unroll
    for (int k = 0; k < MT; ++k)  // A: 4*MT pieces; this wave fills pieces wave*MT + k
      fl_dma16(src_a[k] + k_off, dma_a + (wave * MT + k) * 1024);
#pragma unroll
    for (int k = 0; k < kAsPieces; ++k)
      fl_dma4(src_as[k] + as_off, dma_as + k * 64);
  }
  // ---- operands: 4 + 4*MT ds_read_b128 in flight, then the MFMAs ----
  const uint8_t* wp = rd_w + wave * (32 * BK);
  uint4 wa[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) wa[s] = *reinterpret_cast<const uint4*>(wp + rb[s]);
  uint4 ab[MT][4];
#pragma unroll
  for (int j = 0; j < MT; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s) ab[j][s] = *reinterpret_cast<const uint4*>(rd_a + j * (32 * BK) + rb[s]);
  float sc[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) sc[j] = rd_as[j * 32 + li] * ws;
  const v8i a0 = mk8(wa[0], wa[1]), a1 = mk8(wa[2], wa[3]);
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    v16f part;
#pragma unroll
    for (int r = 0; r < 16; ++r) part[r] = 0.f;
    part = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, mk8(ab[j][0], ab[j][1]), part, 0, 0, 0, kUnit, 0, kUnit);
    part = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, mk8(ab[j][2], ab[j][3]), part, 0, 0, 0, kUnit, 0, kUnit);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = fmaf(part[r], sc[j], acc[j][r]);
  }
}

// One output tile (or one k split of it) of one workgroup.  `bid` is the tile index: blockIdx.x for an ordinary launch, or
// the loop variable of a persistent launch capped by deep_gemm.set_num_sms (grouped_gemm_fp8_kernel below).
template <int MT>
__device__ __forceinline__ void gemm_tile(const GemmParams& p, const uint8_t* __restrict__ gA, const float* __restrict__ gAs,
                                          const uint8_t* __restrict__ gW, const float* __restrict__ gWs,
                                          const int32_t* __restrict__ gmeta, uint8_t* __restrict__ smem, const int bid) {
  constexpr int BM = 32 * MT;
  constexpr int kStages = Smem<MT>::kStages;
  static_assert(Smem<MT>::kPiecesPerWave * (kStages - 2) == (MT == 1 ? 6 : MT == 2 ? 7 : 10 * (FL_MT4_STAGES - 2)), "vmcnt immediates");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;

  // ---- tile -> (expert, token range, weight tile) ----
  const int split = p.ksplit > 1 ? bid % p.ksplit : 0;   // split-K: the k blocks of a tile over ksplit workgroups
  const int tile = p.ksplit > 1 ? bid / p.ksplit : bid;
  const int nt = tile % p.n_tiles;
  const int mt = tile / p.n_tiles;
  int e = 0;
  long long row0 = 0;     // first row of the tile in A / out
  long long row_end = 0;  // one past the last valid row
  if (!locate_tile<BM>(p, gmeta, mt, e, row0, row_end)) return;
  const int n0 = nt * BN;
  // The weight panel of this tile is streamed ONCE by ONE workgroup when its group fits one token tile; in the decode regime proper — at most 16 rows
  // per group — the non-temporal policy on it is worth +7...11 % of weight bandwidth (MI355X guide, "nt-weights"; measured 4 / 8 / 16 rows per expert:
  // 6.20 -> 6.86, 6.15 -> 6.83, 5.95 -> 6.49 TB/s), at 32 rows per expert and on the dense GEMMs at T >= 128 it costs 3-4 % (the token rows that every
  // weight tile of the group re-reads compete with the stream): profiles/r05_nt_policy.txt.  (Also true for a short LAST tile of a longer group.)
  const bool w_nt = row_end - row0 <= 16;
  const int KB_all = p.K / BK;
  const int kb_per = (KB_all + p.ksplit - 1) / p.ksplit;
  const int kb0 = split * kb_per;
  const int KB = (kb0 + kb_per < KB_all ? kb0 + kb_per : KB_all) - kb0;   // k blocks of this workgroup: [kb0, kb0 + KB)
  if (KB <= 0) return;

  // ---- per-lane DMA sources (k-invariant parts) ----
  // W piece (wave*4 + k): rows 8*(wave*4+k) + (lane>>3); LDS chunk position lane&7 holds source chunk (lane&7)^((r>>1)&7)
  const int wr = (wave * 4) * 8 + (lane >> 3);
  // pre-clamped row pointers for the 4 W pieces of this wave (rows beyond N are clamped: results discarded)
  const uint8_t* wsrc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = wr + 8 * k;
    int n = n0 + r;
    n = n < p.N ? n : p.N - 1;
    wsrc[k] = gW + ((long long)e * p.N + n) * p.K + (((lane & 7) ^ ((r >> 1) & 7)) << 4) + (long long)kb0 * BK;
  }
  const uint8_t* asrc[MT];
#pragma unroll
  for (int k = 0; k < MT; ++k) {
    const int r = (wave * MT + k) * 8 + (lane >> 3);   // row inside the token tile
    long long m = row0 + r;
    m = m < row_end ? m : row_end - 1;
    asrc[k] = gA + m * p.K + (((lane & 7) ^ ((r >> 1) & 7)) << 4) + (long long)kb0 * BK;
  }
  constexpr int kAsPieces = Smem<MT>::kAsFloats / 64;
  const float* assrc[kAsPieces];
#pragma unroll
  for (int k = 0; k < kAsPieces; ++k) {
    long long m = row0 + k * 64 + lane;
    m = m < row_end ? m : row_end - 1;
    // masked mode: scales are indexed [group, row in group, kb]
    assrc[k] = (p.mode == kMasked ? gAs + (long long)e * p.as_stride_g + (m - (long long)e * p.rows_per_group) * p.as_stride_m
                                  : gAs + m * p.as_stride_m) + (long long)kb0 * p.as_stride_k;
  }
  // operand read offsets inside a 32-row x 128 B sub-tile: row li, 16-B chunk c = 4*s2 + 2*lh + e2 (s = 2*s2 + e2)
  int rb[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int c = 4 * (s >> 1) + 2 * lh + (s & 1);
    rb[s] = li * BK + ((c ^ ((li >> 1) & 7)) << 4);
  }

  v16f acc[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  auto stage_w = [&](int st) { return smem + st * Smem<MT>::kStageBytes; };
  auto stage_a = [&](int st) { return smem + st * Smem<MT>::kStageBytes + kWBytes; };
  auto stage_as = [&](int st) {
    return reinterpret_cast<float*>(smem + st * Smem<MT>::kStageBytes + kWBytes + Smem<MT>::kABytes);
  };
  auto issue_stage = [&](int kb) {
    const int st = kb % kStages;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      fl_dma16_pol(wsrc[k] + (long long)kb * BK, stage_w(st) + (wave * 4 + k) * 1024, w_nt);
#pragma unroll
    for (int k = 0; k < MT; ++k)
      fl_dma16(asrc[k] + (long long)kb * BK, stage_a(st) + (wave * MT + k) * 1024);
#pragma unroll
    for (int k = 0; k < kAsPieces; ++k)
      fl_dma4(assrc[k] + (long long)kb * p.as_stride_k, stage_as(st) + k * 64);
  };

  // ---- prologue: stages 0 .. kStages-2 ----
#pragma unroll
  for (int s = 0; s < kStages - 1; ++s)
    if (s < KB) issue_stage(s);

  const float* wsrow = gWs + ((long long)e * ((p.N + BN - 1) / BN) + nt) * KB_all + kb0;
  // (a register-double-buffered variant of the 128-token tile was measured slower: 678 vs 833 TFLOP/s at T=16384)
  for (int kb = 0; kb < KB; ++kb) {
    const int st = kb % kStages;
    const int nst = (kb + kStages - 1) % kStages;
    kblock_body<MT>(acc, stage_w(st), stage_a(st), stage_as(st), stage_w(nst), stage_a(nst), stage_as(nst), wsrc, asrc,
                    assrc, (long long)(kb + kStages - 1) * BK, (long long)(kb + kStages - 1) * p.as_stride_k,
                    kb + kStages - 1 < KB, kb + kStages - 2 < KB, wsrow[kb], rb, wave, li, w_nt);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: D^T[n, m] -> out[m, n] bf16; lane (m = li, half lh) holds n = 32*wave + 8g + 4lh + (0..3) ----
  if (p.ksplit > 1) {   // split-K: f32 partials [split, M, N]; reduced + rounded by splitk_reduce_kernel
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const long long m = row0 + j * 32 + li;
      if (m < row_end) {
        float* prow = p.ws + ((long long)split * p.M + m) * p.N;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wave * 32 + 8 * g + 4 * lh;
          if (n + 3 < p.N) {
            *reinterpret_cast<float4*>(prow + n) = make_float4(acc[j][4 * g + 0], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
          } else {
#pragma unroll
            for (int x = 0; x < 4; ++x)
              if (n + x < p.N) prow[n + x] = acc[j][4 * g + x];
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const long long m = row0 + j * 32 + li;
    if (m < row_end) {
      uint16_t* orow = p.out + m * p.N;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wave * 32 + 8 * g + 4 * lh;
        if (n + 3 < p.N) {
          const uint32_t lo = (uint32_t)fl_f32_to_bf16(acc[j][4 * g + 0]) | ((uint32_t)fl_f32_to_bf16(acc[j][4 * g + 1]) << 16);
          const uint32_t hi = (uint32_t)fl_f32_to_bf16(acc[j][4 * g + 2]) | ((uint32_t)fl_f32_to_bf16(acc[j][4 * g + 3]) << 16);
          *reinterpret_cast<uint2*>(orow + n) = make_uint2(lo, hi);
        } else {
#pragma unroll
          for (int x = 0; x < 4; ++x)
            if (n + x < p.N) orow[n + x] = fl_f32_to_bf16(acc[j][4 * g + x]);
        }
      }
    }
  }
}

// Ordinary launch: one workgroup per tile (gridDim.x = total_blocks, one trip).  With deep_gemm.set_num_sms(n) in force
// (tbo_executor.py:129-134: the two-batch-overlap executor leaves SMs to the kernels of the other micro-batch) the grid is
// capped at n workgroups that walk the tile list with stride gridDim.x: at most n workgroup slots of the chip are taken.
template <int MT>
__global__ __launch_bounds__(256, (MT == 4 && FL_MT4_STAGES > 2) ? 1 : 2) void grouped_gemm_fp8_kernel(
    const GemmParams p, const uint8_t* __restrict__ gA, const float* __restrict__ gAs, const uint8_t* __restrict__ gW,
    const float* __restrict__ gWs, const int32_t* __restrict__ gmeta) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[Smem<MT>::kTotal];
  for (int bid = blockIdx.x; bid < p.total_blocks; bid += gridDim.x) {
    gemm_tile<MT>(p, gA, gAs, gW, gWs, gmeta, smem, bid);
    if (gridDim.x < (unsigned)p.total_blocks) {   // persistent launch: every wave is done with the LDS ring before the next tile
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
}

// out[m, n] = bf16(sum_s ws[s, m, n]) in split order; 4 consecutive columns per thread (N % 4 == 0).  The loads of up to 8 splits
// go out together (a loop of dependent load -> add round trips is ~1 us per split).
// Round 3, measured and NOT adopted (profiles/r03_dense_gemm_sweep.txt): reducing a tile inside the GEMM launch by its last-arriving
// k split (arrival counters; hand-off by write-through stores + agent-scope loads, or by agent release / acquire fences) is SLOWER
// than this second launch at every decode shape (18.9 / 37.8 vs 12.1 us at T = 32, [2176, 7168]): one workgroup per tile reads the
// partials past its L2, against the whole chip here; a deeper LDS ring with one workgroup per CU for these launches: also slower.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int ksplit, long long MN,
                                                            uint16_t* __restrict__ out) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= MN) return;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s0 = 0; s0 < ksplit; s0 += 8) {
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int sj = s0 + j < ksplit ? s0 + j : ksplit - 1;   // (clamped: unconditional loads)
      v[j] = *reinterpret_cast<const float4*>(ws + (long long)sj * MN + i);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (s0 + j < ksplit) {
        if (s0 + j == 0) a = v[0];
        else { a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w; }
      }
  }
  *reinterpret_cast<uint2*>(out + i) = make_uint2(fl_pack_bf16(a.x, a.y), fl_pack_bf16(a.z, a.w));
}

// deep_gemm.set_num_sms / get_num_sms (tbo/tbo_executor.py:129-134 saves, sets and restores it around a stage).  The one
// piece of process-wide state behind the C-ABI, by the reference API's own shape (a module-level setter); 0 = not set
// (get reports the device's CU count).  A launch reads it once, on the host.
std::atomic<int> g_num_cus_limit{0};

}  // namespace

extern "C" int fl_gemm_set_num_cus(int n) {
  FL_CHECK_ARG(n >= 0, "fl_gemm_set_num_cus: n=%d", n);
  g_num_cus_limit.store(n);
  return FL_OK;
}
extern "C" int fl_gemm_get_num_cus(void) {
  const int n = g_num_cus_limit.load();
  if (n > 0) return n;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || fl_device_cu_count(dev, &cus) != FL_OK) return 0;
  return cus;
}

extern "C" int fl_grouped_gemm_fp8(const FlGemmArgs* a, fl_stream_t stream) {
  FL_CHECK_ARG(a != nullptr, "fl_grouped_gemm_fp8: null args");
  FL_CHECK_ARG(a->mode >= 0 && a->mode <= 3, "fl_grouped_gemm_fp8: bad mode %d", a->mode);
  if (a->M == 0) return FL_OK;   // nothing routed to this rank (empty tensors have null pointers)
  FL_CHECK_ARG(a->A && a->As && a->W && a->Ws && a->out, "fl_grouped_gemm_fp8: null tensor pointer");
  FL_CHECK_ARG(a->mode == kDense || a->group_meta, "fl_grouped_gemm_fp8: null group metadata");
  FL_CHECK_ARG(a->K > 0 && a->K % BK == 0, "fl_grouped_gemm_fp8: K=%d must be a multiple of 128", a->K);
  FL_CHECK_ARG(a->N > 0 && a->N % 4 == 0, "fl_grouped_gemm_fp8: N=%d must be a multiple of 4", a->N);
  FL_CHECK_ARG(a->num_groups >= 1 && a->M >= 0, "fl_grouped_gemm_fp8: bad sizes");
  if (a->M == 0) return FL_OK;
  GemmParams p;
  p.mode = a->mode; p.E = a->num_groups; p.M = (int)a->M; p.N = a->N; p.K = a->K;
  p.as_stride_m = a->as_stride_m; p.as_stride_k = a->as_stride_k; p.as_stride_g = a->as_stride_g;
  p.rows_per_group = a->rows_per_group;
  p.out = (uint16_t*)a->out;
  p.n_tiles = (a->N + BN - 1) / BN;
  // token-tile height from the expected rows per group (host-side hint only: correctness does not depend on it)
  long long avg = a->expected_m > 0 ? a->expected_m : (a->mode == kMasked ? a->rows_per_group : a->M / a->num_groups);
  if (a->mode == kDense) avg = a->M;
  int mt = avg <= 32 ? 1 : (avg <= 64 ? 2 : 4);
  if (a->mode == kContiguous) mt = 4;   // groups are 128-row aligned by contract (deep_ep_executor.py:282,290)
  // many rows per group (prefill regime): 256 x 256 tiles, one 8-wave workgroup per CU (grouped_gemm_fp8_big2.hip).
  // FLUENT_GEMM_BIG=0 keeps the 128-row tiles for every shape (A/B runs); unset or empty: on.
  static const int big_sel = [] {   // unset / empty: the round-6 192 x 256 one-wave-per-SIMD kernel where its shape rules allow, else big2
    const char* e = getenv("FLUENT_GEMM_BIG");
    return (e != nullptr && e[0] >= '0' && e[0] <= '9') ? e[0] - '0' : 3;
  }();
  const bool big_on = big_sel != 0;
  // (contiguous groups are only 128-row aligned; the big tile addresses a weight panel with 32-bit offsets)
  // (measured, E = 256 top-8: 128 rows per expert: w13 872 vs 871, w2 816 vs 873 TFLOP/s with the 256 x 256 kernel; 64 rows per expert:
  //  559 vs 459 — the 128-row kernel stays below 128)
  constexpr long long big_min = 128;
  p.ksplit = 1;
  p.ws = nullptr;
  // Dense GEMMs with few rows (decode projections; a workspace is present): the token-tile height follows the weight
  // bytes — measured (tools/bench_dense.py): <= 8 MB 32-token tiles, <= 32 MB 64-token tiles (3-stage rings, 2
  // workgroups per CU), else 128-token tiles (fewer re-reads of W through L2); split-K fills the chip
  constexpr int dense_mt = 0;
  if (a->mode == kDense && a->workspace != nullptr) {
    const long long wb = (long long)a->N * a->K;
    const int want = dense_mt > 0 ? dense_mt : (wb <= (8ll << 20) ? 1 : (wb <= (32ll << 20) ? 2 : 4));
    if (mt > want) mt = want;
  }
  const bool few_tiles = a->mode == kDense && a->workspace != nullptr &&
                         ((a->M + 32 * mt - 1) / (32 * mt)) * (long long)p.n_tiles < 384;
  // (under a deep_gemm.set_num_sms cap the 128-row kernel's persistent tile walk is used: the 256 x 256 kernel is one
  //  workgroup per tile by construction)
  if (!few_tiles && big_on && avg >= big_min && a->K >= 2 * BK && a->mode != kContiguous &&
      (long long)a->N * a->K < (1ll << 32) && g_num_cus_limit.load() == 0) {
    if (big_sel >= 3 && a->N % 256 == 0 && a->K <= 64 * BK)
      return fl_gemm_launch_big3(p, a->A, a->As, a->W, a->Ws, a->group_meta, (hipStream_t)stream);
    return fl_gemm_launch_big2(p, a->A, a->As, a->W, a->Ws, a->group_meta, (hipStream_t)stream);
  }
  const int bm = 32 * mt;
  long long m_tiles;
  if (a->mode == kOffset) m_tiles = (a->M + bm - 1) / bm + a->num_groups;
  else if (a->mode == kMasked) m_tiles = (long long)a->num_groups * ((a->rows_per_group + bm - 1) / bm);
  else m_tiles = (a->M + bm - 1) / bm;
  p.m_tiles_upper = (int)m_tiles;
  long long blocks = m_tiles * p.n_tiles;
  // few tiles for the chip (dense decode GEMMs: [T <= 256, K] x [N, K] has N/128 .. 2N/128 tiles): split the k blocks so
  // that ~2 workgroups per CU stream the weights; f32 partials in the caller's workspace, one reduce kernel
  p.ws = (float*)a->workspace;
  if (few_tiles) {
    // bytes-equivalent cost model: the weight stream runs at full rate only with >= one workgroup per CU, and every split
    // writes + reads an f32 partial of the whole output: cost(ks) = W / min(1, tiles*ks/256) + 2*ks*M*N*4
    const int KBt = a->K / BK;
    const double wbytes = (double)a->N * a->K, pbytes = 8.0 * (double)a->M * a->N;
    int best = 1;
    double best_cost = wbytes / (blocks >= 256 ? 1.0 : (double)blocks / 256.0);
    for (int ks = 2; ks <= KBt / 2; ++ks) {
      if ((long long)ks * a->M * a->N * 4 > a->workspace_bytes) break;
      const double par = (double)blocks * ks / 256.0;
      const double cost = wbytes / (par >= 1.0 ? 1.0 : par) + pbytes * ks;
      if (cost < best_cost) { best_cost = cost; best = ks; }
    }
    if (best > 1) {
      const int per = (KBt + best - 1) / best;
      p.ksplit = (KBt + per - 1) / per;                               // no empty splits
      blocks *= p.ksplit;
    }
  }
  FL_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "fl_grouped_gemm_fp8: grid too large");
  p.total_blocks = (int)blocks;
  const int cu_cap = g_num_cus_limit.load();
  long long launch_blocks = blocks;
  if (cu_cap > 0 && launch_blocks > cu_cap) launch_blocks = cu_cap;   // persistent walk over the tiles (set_num_sms)
  const dim3 grid((unsigned)launch_blocks), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (mt == 1)
    grouped_gemm_fp8_kernel<1><<<grid, block, 0, s>>>(p, (const uint8_t*)a->A, a->As, (const uint8_t*)a->W, a->Ws, a->group_meta);
  else if (mt == 2)
    grouped_gemm_fp8_kernel<2><<<grid, block, 0, s>>>(p, (const uint8_t*)a->A, a->As, (const uint8_t*)a->W, a->Ws, a->group_meta);
  else
    grouped_gemm_fp8_kernel<4><<<grid, block, 0, s>>>(p, (const uint8_t*)a->A, a->As, (const uint8_t*)a->W, a->Ws, a->group_meta);
  FL_CHECK_LAUNCH("grouped_gemm_fp8_kernel");
  if (p.ksplit > 1) {
    const long long MN = a->M * (long long)a->N;
    splitk_reduce_kernel<<<dim3((unsigned)((MN / 4 + 255) / 256)), dim3(256), 0, s>>>(p.ws, p.ksplit, MN, p.out);
    FL_CHECK_LAUNCH("splitk_reduce_kernel");
  }
  return FL_OK;
}
