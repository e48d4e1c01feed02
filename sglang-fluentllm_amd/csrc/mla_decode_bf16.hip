// K2-bf16 — paged MLA decode over ONE bf16 [pages, 64, 1, 576] cache tensor, gfx950 (MI355X) only.
//
// Replaces flash_mla_swap / flash_mla_fp8 .flash_mla_with_kvcache for cache_dtype = bf16 (call sites
// /root/reference/python/sglang/srt/layers/attention/flashmla_backend.py:163-175 verify/draft-extend, :240-254 decode):
//   s[r,t] = (q[r,:576] · k[t,:576]) * softmax_scale,  o[r,:] = sum_t softmax_t(s[r,:]) * k[t,:512]
// with q, k, P in bf16 and fp32 accumulation.  Same C-ABI, scheduler metadata and split-KV combine as the FP8 kernels.
//
// Round 5: ROLE-SPECIALISED waves (the structure of mla_decode_fp8_y.hip; round 2's kernel — every wave a quarter of the k
// steps, an all-to-all exchange of the partial S^T, the whole softmax four times over, then a quarter of PV: 5,200 cycles per
// 32-token tile — is probes/superseded/mla_decode_bf16_v1.hip.txt).  Workgroup = 32 NRT query rows of one request part (NRT = 2 row
// tiles once a request has more than 32 rows: 8 waves, two per SIMD, ONE tile ring), tile = 32 tokens (half a page, 36 KiB of bf16):
//   * QK waves a in [0, 2 NRT): query rows [16a, 16a + 16) x ALL 32 tokens of tile i on v_mfma_f32_16x16x32_bf16 ("SwapAB":
//     S^T[16 tok x 16 rows] = K . Q^T, 2 token tiles x 18 k-steps = 36 MFMAs in two independent chains; the wave's Q fragment is 72
//     VGPRs; the K fragments run 5 k-steps ahead of their MFMAs).  A lane holds 8 scores of ONE query row (lane & 15): tokens
//     16 tt + 4 (lane >> 4) + r; the row maximum crosses the four lane groups by v_permlane16_swap / v_permlane32_swap.  The ROWS are split over the
//     waves, not the k steps: no partial sums travel and the softmax halves run side by side.  The row's reference is FIXED 24 log2
//     units above its first valid tile's maximum (mla_decode_fp8_y.hip's O reference: weights above 1 are fine in bf16 / fp32) and
//     moves only on a jump of more than 2^64; P = 2^(y - m) as bf16 -> LDS in the PV MFMA's B-operand order, the row's (almost
//     always 1) rescale factor beside it.
//   * PV waves (row tile rt, d half) one step later: O^T[256 dims x 32 rows] += V^T . P^T on v_mfma_f32_32x32x16_bf16 (8 dim tiles x
//     2 k-steps), V^T by ds_read_b64_tr_b16 (3 MFMAs ahead) from the same LDS bytes the QK waves read as K; the four tokens of a
//     transpose-read group are (T, T+1, T+8, T+9): conflict-free (consecutive tokens: 38 % of the LDS cycles were conflicts).
//   * 4-slot ring of 36 KiB tiles (tile i-1 read as V^T, tile i as K, tiles i+1, i+2 landing: two steps of flight), LDS-DMA
//     (global_load_lds, 1 KiB per wave instruction): the PV waves issue all 36 pieces of tile i+2 at the head of step i, in front of
//     their MFMAs (18 / NRT each; a piece blocks its wave's issue for 50-100 cycles wherever it sits: between MFMAs it stretched the
//     MFMA phase 2.5 x), the block-table entry read one step ahead; counted vmcnt, ONE s_barrier per tile step.  16-B chunks
//     XOR-swizzled on the source address (chunk c of token T at (c & ~7) | ((c & 7) ^ ((T >> 1) & 7))): the K reads of a 16-lane
//     group hit 16 distinct slots.  One row group per request: the pages stream non-temporal (nobody re-reads them from L2).
// Where it stands (profiles/r05_k2_bf16_rewrite.txt): H <= 32 streams 5.7-6.4 TB/s (the chip's LDS-DMA stream rate is 6.2-6.3); H = 128 is
// at the bf16 ridge (242 FLOP/B) with 45 % matrix-pipe duty — flight depth, priorities and prefetch distances measured flat.
#include "mla_decode_shared.h"

using namespace fl_mla;

typedef short v4s __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kTile = 32;                       // tokens per tile
constexpr int kRowB = (kDN + kDR) * 2;          // 1152 B per token
constexpr int kTileBytes = kTile * kRowB;       // 36 KiB
constexpr int kSlots = 4;
// NRT = row tiles of 32 query rows per workgroup (1: 4 waves, 2: 8 waves sharing one tile ring)
constexpr int kOffP = kSlots * kTileBytes;      // [parity 2][32 NRT rows][32 tokens] bf16 in PV order
template <int NRT> struct Lay {
  static constexpr int kRows = 32 * NRT;
  static constexpr int kPieces = 18 / NRT;   // LDS-DMA pieces of 1 KiB per PV wave and tile (36 in all)
  static constexpr int kThreads = 256 * NRT;
  // a row's 32 weights (64 B) sit in an 80-B line: at a 64-B pitch the 16 rows of a b128 read group share 4 bank sets (4-way) and the b32
  // stores of a 32-lane group 4 banks (8-way) — 320 conflict cycles per tile step (SQ_LDS_BANK_CONFLICT with the K / V^T reads removed)
  static constexpr int kPPitch = 80;
  static constexpr int kPBytes = kRows * kPPitch;
  static constexpr int kOffF = kOffP + 2 * kPBytes;       // [parity 2][rows] f32 rescale factor of O before adding the tile
  static constexpr int kOffLm = kOffF + 2 * kRows * 4;    // [rows] l, [rows] m (epilogue)
  static constexpr int kLds = kOffLm + 2 * kRows * 4;
  static_assert(kLds <= 160 * 1024, "LDS budget");
};
constexpr float kNegInit = -16384.0f;           // "no reference yet" (finite, integer)
constexpr float kRefLift = 24.f;                // the reference sits this far above the first tile's maximum ...
constexpr float kMaxUp = 64.f;                  // ... and moves only when a later tile exceeds it by more than this (|O| < 2^64 x |V| x tokens)

#ifdef FL_MLA_TIMING   // debug builds only (tools/time_phases_bf16.py): per-wave cycle accumulators of the tile loop's phases
__device__ int* g_dbg_b = nullptr;
#define FL_T(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); tacc[i] += t__ - tlast; tlast = t__; } while (0)
#else
#define FL_T(i) do { } while (0)
#endif

__device__ __forceinline__ int swz(const int c, const int T) { return (c & ~7) | ((c & 7) ^ ((T >> 1) & 7)); }

// byte offset inside a tile of this lane's 16 B of LDS-DMA piece `piece` (destination is lane-linear)
__device__ __forceinline__ unsigned dma_src_off(const int piece, const int lane) {
  const int off = piece * 1024 + lane * 16;
  const int T = off / kRowB;
  const int cp = (off - T * kRowB) >> 4;
  return (unsigned)(T * kRowB + 16 * swz(cp, T));
}

// (launch bounds 256 x 2: at most 256 registers per wave and none of them AGPRs — with the whole 512-register file hipcc parks O in AGPRs and
//  moves it through VGPRs around every VALU touch: copies of all 128 O registers in the tile loop)
template <int NRT, bool NT>
__global__ __launch_bounds__(Lay<NRT>::kThreads, 2) void mla_decode_bf16_kernel(
    const Params p, const int32_t* __restrict__ g_block_table, const int32_t* __restrict__ g_seqlens,
    const int32_t* __restrict__ g_meta, const int32_t* __restrict__ g_num_splits, const uint8_t* __restrict__ g_k,
    const uint8_t* __restrict__ g_q) {
  using L_ = Lay<NRT>;
  constexpr int kRows = L_::kRows, kPieces = L_::kPieces, kPBytes = L_::kPBytes;
  constexpr int kQKWaves = 2 * NRT;
  __shared__ __attribute__((aligned(16))) uint8_t smem[L_::kLds];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_pv = wave >= kQKWaves;
  const int pw = wave - kQKWaves;     // PV wave: row tile pw >> 1, d half pw & 1
  const int rt = NRT == 2 ? pw >> 1 : 0, dh = pw & 1;
  const int lane = tid & 63;
  // the row groups of one part stream the same KV pages: same XCD (workgroup ids go round-robin over the 8 XCDs, each with
  // its own L2), so the pages come from HBM once per part instead of once per row group
  int rgrp = blockIdx.x % p.row_groups;
  int part = blockIdx.x / p.row_groups;
  if (p.row_groups > 1 && p.num_parts % 8 == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    part = (slot / p.row_groups) * 8 + xcd;
    rgrp = slot % p.row_groups;
  }
  const int32_t* meta = g_meta + part * FL_MLA_META_W;
  int req = meta[0];
  int tile_b = meta[1];
  const int end_req = meta[2];
  const int end_tile = meta[3];
  int split_idx = meta[4];

  // The PV waves fill the tile ring: wave w owns LDS-DMA pieces [kPieces w, + kPieces) of every tile and issues them at the head of its step,
  // in front of its MFMAs.  A 1-KiB piece blocks its wave's issue for 50 - 100 cycles wherever it sits (18 pieces: 1,720 cycles bare, 12
  // pieces ~1,000 cycles spread between 16 MFMAs; one loader wave of its own: 36 pieces 2,140 cycles and only 64 in flight, vmcnt's range), so
  // the QK waves — the long pole of a step — issue none.  dsrc = the source offsets of this lane's 16 B of each piece.
  unsigned dsrc[kPieces];
  const int piece0 = (is_pv ? pw : 0) * kPieces;
#pragma unroll
  for (int k = 0; k < kPieces; ++k) dsrc[k] = dma_src_off(piece0 + k, lane);

#ifdef FL_MLA_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
  const unsigned long long tstart = tlast, wstart = wall_clock64();
#endif
  float* fbuf = reinterpret_cast<float*>(smem + L_::kOffF);
  float* lmbuf = reinterpret_cast<float*>(smem + L_::kOffLm);

  for (; req < p.bs; ++req, tile_b = 0, split_idx = 0) {
    if (req > end_req || (req == end_req && end_tile == 0)) break;
    const int L = g_seqlens[req];
    const int nt = L > 0 ? (L + kPage - 1) / kPage : 0;
    int tile_e = req < end_req ? nt : (end_tile < nt ? end_tile : nt);
    if (tile_e < tile_b) tile_e = tile_b;
    // 32-token tiles of the pages [tile_b, tile_e)
    const int t0 = 2 * tile_b;
    int t1 = 2 * tile_e;
    const int tmaxL = (L + kTile - 1) / kTile;
    if (t1 > tmaxL) t1 = tmaxL;
    const int n = t1 > t0 ? t1 - t0 : 0;
    const int split_base = g_num_splits[req];
    const bool is_split = (g_num_splits[req + 1] - split_base) > 1;

    // page of 32-token tile t0 + t (one block-table read: issued a step before its use, see the PV loop)
    auto tile_page = [&](const int t) {
      long long col = (t0 + t) >> 1;
      col = col < p.bt_cols ? col : p.bt_cols - 1;
      const int pg = g_block_table[(long long)req * p.bt_stride + col];
      return (pg < 0 || pg >= p.num_pages) ? 0 : pg;
    };
    auto slot = [&](int t) { return smem + (t & (kSlots - 1)) * kTileBytes; };
    auto issue_tile = [&](const int t, const int pg) {   // PV waves only
      const uint8_t* src = g_k + ((long long)pg * kPage + ((t0 + t) & 1) * kTile) * kRowB;
      uint8_t* dst = slot(t) + piece0 * 1024;
#pragma unroll
      for (int k = 0; k < kPieces; ++k) fl_dma16_pol(src + dsrc[k], dst + k * 1024, NT);
    };

    if (!is_pv) {
      // =========================== QK waves: rows [16 wave, 16 wave + 16) of the workgroup's 32 NRT ===========================
      const int r16 = lane & 15, g = lane >> 4;     // query row inside the half, lane group (k quarter of an MFMA step / token quad)
      const int row = rgrp * kRows + wave * 16 + r16;
      const bool row_ok = row < p.rows;
      const long long qrow = (long long)req * p.rows + row;
      // Q fragment: k-step ks covers elements [32 ks, 32 ks + 32): this lane's 8 of them at 32 ks + 8 g
      u32x4 qf[18];
#pragma unroll
      for (int ks = 0; ks < 18; ++ks) qf[ks] = u32x4{0, 0, 0, 0};
      if (row_ok) {
        const uint8_t* qp = g_q + qrow * kRowB + g * 16;
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) qf[ks] = *reinterpret_cast<const u32x4*>(qp + ks * 64);
      }
      int L_row = L;
      if (p.causal) L_row = L - (p.s_q - 1 - row / p.h_q);
      if (!row_ok) L_row = 0;
      const int L_min = p.causal ? L - (p.s_q - 1) : L;   // the shortest row of the request (rows past p.rows: never stored)
      float m_run = kNegInit, l_run = 0.f;
      // the QK chain + softmax is the long pole of a tile step and shares its SIMD's matrix pipe with a PV wave (NRT = 2): it goes first
      __builtin_amdgcn_s_setprio(2);

      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      // (the Q fragment is COMPLETE here as far as hipcc is concerned: without these uses its own counted waits for the 18 loads sit in
      //  front of the MFMAs inside the tile loop — vmcnt(17) ... vmcnt(0), every step — and drain the LDS-DMA of the tiles in flight)
#pragma unroll
      for (int ks = 0; ks < 18; ++ks) asm volatile("" : "+v"(qf[ks]));
      __builtin_amdgcn_s_barrier();   // R0: previous request's LDS reads are done
      FL_T(5);
      for (int i = 0; i < n; ++i) {
        FL_T(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // B_i: tile i landed (the PV waves waited for their pieces); P / f of parity i & 1 are free (PV read them in step i - 1)
        FL_T(1);
        const uint8_t* rd = slot(i);
        // ---- S^T[32 tok x 16 rows]: token tile tt, lane = (token 16 tt + r16, k quarter g) on the A side ----
        v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const uint8_t* ka0 = rd + r16 * kRowB;
        const uint8_t* ka1 = rd + (16 + r16) * kRowB;
        // (swizzle term of the two token rows: T and T + 16 share (T >> 1) & 7)
        const int sx = (r16 >> 1) & 7;
        // K fragments run kAhead k-steps in front of their MFMAs (hipcc on its own keeps ONE read in flight: 36 x (LDS latency) per tile)
        constexpr int kAhead = 5;
        uint4 fa0[18], fa1[18];
        auto load_k = [&](const int ks) {
          const int c = 4 * ks + g;
          const int cs = ((c & ~7) | ((c & 7) ^ sx)) << 4;
          fa0[ks] = *reinterpret_cast<const uint4*>(ka0 + cs);
          fa1[ks] = *reinterpret_cast<const uint4*>(ka1 + cs);
        };
#pragma unroll
        for (int ks = 0; ks < kAhead; ++ks) load_k(ks);
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) {
          if (ks + kAhead < 18) load_k(ks + kAhead);
          const uint4 qk = make_uint4(qf[ks][0], qf[ks][1], qf[ks][2], qf[ks][3]);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf8(fa0[ks]), as_bf8(qk), acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf8(fa1[ks]), as_bf8(qk), acc1, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        FL_T(2);
        // ---- online softmax in the log2 domain; lane holds tokens 16 tt + 4 g + r of row r16 ----
        const int tok0 = (t0 + i) * kTile;
        float y[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) y[r] = (r < 4 ? acc0[r & 3] : acc1[r & 3]) * p.scale_log2e;
        if (tok0 + kTile > L_min) {   // wave-uniform: only a request's last tile(s) hold tokens past some row's end
#pragma unroll
          for (int r = 0; r < 8; ++r)
            if (tok0 + 16 * (r >> 2) + 4 * g + (r & 3) >= L_row) y[r] = -INFINITY;
        }
        float tmax = fl_max3(fl_max3(y[0], y[1], y[2]), fl_max3(y[3], y[4], y[5]), fmaxf(y[6], y[7]));
        {   // max over the four lane groups in registers (v_permlane16_swap / v_permlane32_swap: no LDS round trip)
          const auto s16_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
          tmax = fmaxf(__uint_as_float(s16_[0]), __uint_as_float(s16_[1]));
          const auto s32_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
          tmax = fmaxf(__uint_as_float(s32_[0]), __uint_as_float(s32_[1]));
        }
        // The row's reference is FIXED kRefLift above its first valid tile's (integer) maximum: later tiles enter with weights 2^(y - m) that may
        // exceed 1 (bf16 / fp32 have the range), so nothing rescales O or l in the tile loop.  Only a tile whose maximum outruns the reference
        // by more than kMaxUp moves it (f < 1: the PV waves multiply O once, out of line) — the O-reference scheme of mla_decode_fp8_y.hip.
        float m_new = m_run;
        if (m_run == kNegInit) m_new = tmax > -INFINITY ? ceilf(tmax) + kRefLift : kNegInit;
        else if (tmax - m_run > kMaxUp) m_new = ceilf(tmax) + kRefLift;
        const float f = m_new == m_run ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);   // (first tile: 2^-inf-ish = 0 on an empty O)
        l_run *= f;
        m_run = m_new;
        float pv[8];
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          pv[r] = __builtin_amdgcn_exp2f(y[r] - m_new);
          psum += pv[r];
        }
        l_run += psum;   // (per-lane partial sums: reduced over the four lane groups once, in the epilogue)
        // P -> LDS in the PV MFMA's B-operand order: the row's 64-B line, token T at position q(T) = 16 (T >> 4) + 8 ((T >> 1) & 1) + (T & 1) +
        // 2 ((T >> 3) & 1) + 4 ((T >> 2) & 1) — the inverse of the PV waves' V^T read order (there: why).  This lane's tokens 16 tt + 4 g + r:
        // pairs r = (0, 1) at position 16 tt + 2 (g >> 1) + 4 (g & 1), r = (2, 3) eight further.
        uint8_t* prow = smem + kOffP + (i & 1) * kPBytes + (wave * 16 + r16) * L_::kPPitch + 4 * (g >> 1) + 8 * (g & 1);
        *reinterpret_cast<uint32_t*>(prow) = fl_pack_bf16(pv[0], pv[1]);
        *reinterpret_cast<uint32_t*>(prow + 16) = fl_pack_bf16(pv[2], pv[3]);
        *reinterpret_cast<uint32_t*>(prow + 32) = fl_pack_bf16(pv[4], pv[5]);
        *reinterpret_cast<uint32_t*>(prow + 48) = fl_pack_bf16(pv[6], pv[7]);
        if (g == 0) fbuf[(i & 1) * kRows + wave * 16 + r16] = f;
        FL_T(4);
      }
      // B_n: the PV waves run PV(n - 1); then the normalisers for their epilogue
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      {
        float l_tot = l_run + __shfl_xor(l_run, 16);
        l_tot += __shfl_xor(l_tot, 32);
        if (g == 0) {
          lmbuf[wave * 16 + r16] = l_tot;
          lmbuf[kRows + wave * 16 + r16] = m_run;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // E0
      FL_T(6);
      continue;
    }

    // =========================== PV waves: dims [256 dh, 256 dh + 256) of the 32 rows of row tile rt ===========================
    const int li = lane & 31, lh = lane >> 5;
    const int row = rgrp * kRows + rt * 32 + li;
    const bool row_ok = row < p.rows;
    const long long qrow = (long long)req * p.rows + row;
    v16f o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[j][r] = 0.f;

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // R0
    if (n > 0) issue_tile(0, tile_page(0));
    if (n > 1) issue_tile(1, tile_page(1));
    int pg_next = n > 2 ? tile_page(2) : 0;   // the page of the tile the NEXT step issues: its block-table read is a step old by then
    const int s16 = lane & 15;
    unsigned vbase[2][2];   // lane part of the V^T read addresses (see the tile loop)
    {
      const int j = s16 >> 2, q = s16 & 3, b16 = (lane >> 4) & 1;
      const int T_l = (j & 1) + 8 * (j >> 1) + 2 * lh;
      const int x_l = (2 * b16 + (q >> 1)) ^ (4 * (j >> 1) + lh);   // chunk-in-8 of the lane's dims ^ the token's swizzle term ((T >> 1) & 7 less 2 u)
#pragma unroll
      for (int mp = 0; mp < 2; ++mp)
#pragma unroll
        for (int u = 0; u < 2; ++u) vbase[mp][u] = (unsigned)(T_l * kRowB + 16 * (x_l ^ (4 * mp + 2 * u)) + 8 * (q & 1) + 512 * dh);
    }
    // B_i of the PV waves: tile i has landed; zero its rows past the end of the sequence (P is exactly 0 there, but 0 * NaN would poison
    // the PV MFMA of the next step; the QK waves mask those tokens by index)
    auto step_head = [&](const int i) {
      // tile i landed: this wave's pieces of tile i + 1 (issued a step later) may stay in flight
      if (i + 1 < n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPieces) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      FL_T(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // B_i (i = n: the QK waves' B_n)
      FL_T(1);
      if (i < n && (t0 + i) * kTile + kTile > L) {
        const int nvalid = L - (t0 + i) * kTile > 0 ? L - (t0 + i) * kTile : 0;
        uint8_t* wr_ = slot(i);
        for (int T = nvalid + pw; T < kTile; T += kQKWaves)
          for (int c = lane; c < kRowB / 16; c += 64) *reinterpret_cast<uint4*>(wr_ + T * kRowB + c * 16) = make_uint4(0, 0, 0, 0);
      }
    };
    FL_T(5);
    step_head(0);   // step 0 has no P yet: the loop below is one uniform body (a `continue` here made hipcc copy O at the loop edge)
    // refill: tile i + 2 into the slot tile i - 2 left (its last reader, PV of step i - 1, is behind B_i), then the block-table entry of the
    // step after (a step old at its use: read in front of the pieces, its latency was 650 cycles of every step)
    auto refill = [&](const int i) {
      if (i + 2 < n) issue_tile(i + 2, pg_next);
      if (i + 3 < n) pg_next = tile_page(i + 3);
    };
    refill(0);
    for (int i = 1; i <= n; ++i) {
      step_head(i);
      // ---- O^T += V^T(tile i - 1) . P^T(i - 1) ----
      const uint8_t* rd = slot(i - 1);
      const uint8_t* pb = smem + kOffP + ((i - 1) & 1) * kPBytes + (rt * 32 + li) * L_::kPPitch + lh * 16;
      const uint4 p0 = *reinterpret_cast<const uint4*>(pb);        // k-step 0: positions 8 lh .. + 7 of the row's line (q(T) above)
      const uint4 p1 = *reinterpret_cast<const uint4*>(pb + 32);   // k-step 1
      const float f = fbuf[((i - 1) & 1) * kRows + rt * 32 + li];
      if (__builtin_expect(__any(f != 1.f), 0)) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[j][r] *= f;
      }
      FL_T(2);
      refill(i);
      FL_T(4);
      // V^T fragments run kAheadV MFMAs in front of their MFMAs; nothing else shares the sequence
      constexpr int kAheadV = 3;
      union Frag { v8bf v; v4s h[2]; };
      Frag a[16];
      // tr16 read u fetches A elements e = 4u .. 4u + 3 of k-step s: source lane s16 -> token T(j = s16 >> 2), dims
      // 16 ((lane >> 4) & 1) + 4 (s16 & 3) .. + 3 of the 32-dim tile.  Which tokens a read's four j are is free (P is stored to match):
      // T = (j & 1) + 8 (j >> 1) + 2 lh + 4 u + 16 s puts the four tokens of one 32-lane group in the four 64-B quarters of the 256-B
      // bank period (a token's parity picks the half — 1,152 B = 4.5 periods —, bit 3 the quarter through the swizzle); j = 0 .. 3
      // consecutive tokens put T and T + 2 on the same 16 banks: 38 % of the kernel's LDS cycles were conflicts (SQ_LDS_BANK_CONFLICT).
      // The address of read (mt, s, u) = one of FOUR lane bases (vbase[mt & 1][u]: the swizzle XORs 4 (mt & 1) + 2 u into the chunk) + an
      // immediate (left to hipcc, every read got a base register of its own: 25 spilled in the tile loop).
      const uint8_t* rb[2][2] = {{rd + vbase[0][0], rd + vbase[0][1]}, {rd + vbase[1][0], rd + vbase[1][1]}};
      auto load_v = [&](const int k) {
        const int mt = k >> 1, s = k & 1;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const uint8_t* ap = rb[mt & 1][u] + (4 * u + 16 * s) * kRowB + 128 * (mt >> 1);
          a[k].h[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)ap);
        }
      };
#pragma unroll
      for (int k = 0; k < kAheadV; ++k) load_v(k);
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (k + kAheadV < 16) load_v(k + kAheadV);
        o[k >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k].v, as_bf8((k & 1) == 0 ? p0 : p1), o[k >> 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      FL_T(3);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // E0: the normalisers of the QK waves are in LDS
    FL_T(6);

    // ---- epilogue: normalise, store this wave's 256 dims of the 32 rows ----
    const float l_run = lmbuf[rt * 32 + li], m_run = lmbuf[kRows + rt * 32 + li];
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    const float lse_nat = l_run > 0.f ? (__builtin_amdgcn_logf(l_run) + m_run) * 0.6931471805599453f : -INFINITY;
    if (row_ok) {
      const int slot_idx = split_base + split_idx;
      if (lh == 0 && dh == 0) {
        if (is_split) {
          p.lse_accum[((long long)slot_idx * p.rows + row) * 2 + 0] = lse_nat;
          p.lse_accum[((long long)slot_idx * p.rows + row) * 2 + 1] = lse_nat;
        } else {
          const int j = row / p.h_q, h = row - j * p.h_q;
          p.lse[((long long)req * p.h_q + h) * p.s_q + j] = lse_nat;
        }
      }
      // C row i = (r&3) + 8(r>>2) + 4lh of tile mt -> d = 256 dh + 32 mt + i
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d0 = 256 * dh + 32 * mt + 8 * g + 4 * lh;
          if (is_split) {
            *reinterpret_cast<float4*>(p.o_accum + ((long long)slot_idx * p.rows + row) * kDN + d0) =
                make_float4(o[mt][4 * g] * inv, o[mt][4 * g + 1] * inv, o[mt][4 * g + 2] * inv, o[mt][4 * g + 3] * inv);
          } else {
            const uint32_t lo = (uint32_t)fl_f32_to_bf16(o[mt][4 * g] * inv) | ((uint32_t)fl_f32_to_bf16(o[mt][4 * g + 1] * inv) << 16);
            const uint32_t hi = (uint32_t)fl_f32_to_bf16(o[mt][4 * g + 2] * inv) | ((uint32_t)fl_f32_to_bf16(o[mt][4 * g + 3] * inv) << 16);
            *reinterpret_cast<uint2*>(p.out + qrow * kDN + d0) = make_uint2(lo, hi);
          }
        }
    }
    FL_T(7);
  }
#ifdef FL_MLA_TIMING
  if (g_dbg_b != nullptr && lane == 0) {
    unsigned long long* d = reinterpret_cast<unsigned long long*>(g_dbg_b) + ((long long)blockIdx.x * 8 + wave) * 10;
    for (int i = 0; i < 8; ++i) d[i] = tacc[i];
    d[8] = __builtin_readcyclecounter() - tstart;
    d[9] = wall_clock64() - wstart;
  }
#endif
}

}  // namespace

#ifdef FL_MLA_TIMING
extern "C" int fl_mla_debug_set_buffer_b(int* dev_ptr) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_b), &dev_ptr, sizeof(dev_ptr)); }
#endif

int fl_mla_decode_bf16_impl(const FlMlaDecodeArgs* a, hipStream_t stream) {
  FL_CHECK_ARG(a->d_nope == kDN && a->d_rope == kDR, "fl_mla_decode: only d_nope=512,d_rope=64 (got %d,%d)", a->d_nope,
               a->d_rope);
  FL_CHECK_ARG(a->q_nope && a->k_nope, "fl_mla_decode: null q/k pointer");
  FL_CHECK_ARG(a->block_table && a->cache_seqlens && a->tile_scheduler_metadata && a->num_splits && a->out && a->lse &&
                   a->o_accum && a->lse_accum,
               "fl_mla_decode: null metadata/output pointer");
  FL_CHECK_ARG(a->bs >= 0 && a->s_q >= 1 && a->h_q >= 1 && a->num_parts >= 1, "fl_mla_decode: bad sizes");
  if (a->bs == 0) return FL_OK;
  Params p;
  p.bs = a->bs; p.s_q = a->s_q; p.h_q = a->h_q; p.rows = a->s_q * a->h_q; p.causal = a->causal;
  p.num_parts = a->num_parts;
  p.scale_log2e = a->softmax_scale * kLog2e;
  p.descale_q = nullptr; p.descale_k = nullptr;
  p.num_pages = a->num_pages; p.bt_stride = a->block_table_stride;
  p.bt_cols = a->block_table_cols > 0 ? a->block_table_cols : (a->block_table_stride > 0 ? a->block_table_stride : 1);
  p.out = (uint16_t*)a->out; p.lse = a->lse; p.o_accum = a->o_accum; p.lse_accum = a->lse_accum;
  p.partial_bf16 = 0;
  p.merge_in_kernel = 0;
  p.q_bf16 = nullptr;
  // more than 32 query rows per request: 8-wave workgroups of 64 rows (two row tiles share the tile ring: half the LDS-DMA ingest per row and
  // two waves per SIMD, so one wave's softmax runs under the other's MFMAs)
  const int nrt = p.rows > 32 ? 2 : 1;
  p.row_groups = (p.rows + 32 * nrt - 1) / (32 * nrt);
  const dim3 grid((unsigned)(p.num_parts * p.row_groups)), block(nrt == 2 ? Lay<2>::kThreads : Lay<1>::kThreads);
  // one row group per request: nobody else reads this workgroup's pages, they stream past L2 non-temporal
  const bool nt = p.row_groups == 1;
#define FL_BF16_LAUNCH(NRT_, NT_)                                                                                                    \
  mla_decode_bf16_kernel<NRT_, NT_><<<grid, block, 0, stream>>>(p, a->block_table, a->cache_seqlens, a->tile_scheduler_metadata,    \
                                                                  a->num_splits, (const uint8_t*)a->k_nope, (const uint8_t*)a->q_nope)
  if (nrt == 2) { if (nt) FL_BF16_LAUNCH(2, true); else FL_BF16_LAUNCH(2, false); }
  else { if (nt) FL_BF16_LAUNCH(1, true); else FL_BF16_LAUNCH(1, false); }
#undef FL_BF16_LAUNCH
  FL_CHECK_LAUNCH("mla_decode_bf16_kernel");
  return fl_mla_launch_combine(p, a->num_splits, stream);
}
