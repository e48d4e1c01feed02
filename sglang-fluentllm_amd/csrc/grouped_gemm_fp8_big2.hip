// G1-G3 for MANY rows per group, round 3: the 256 x 256 tile of grouped_gemm_fp8_big.hip rebuilt around what round 2
// measured: the old k loop was a 2-stage ring of whole 128-byte k blocks with
// `vmcnt(0)` + barrier at the top of every k block — a prefetch distance of ONE k block, all eight waves in the same
// wait -> barrier -> compute phase, the matrix pipe idle through every wait (4,244 cycles per k block against 2,048 of
// MFMA per SIMD).  Same math, call sites and data formats (deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_*).
//
//   * RING: 4 slots of HALF k blocks (64 bytes of k = one MX MFMA deep): [W 256 rows x 64 B | A 256 rows x 64 B] =
//     32 KiB per slot, refilled THREE half steps ahead with counted `vmcnt` (never 0 in the loop): a piece has 2.5-3
//     half steps (~3,000 cycles) to land instead of one k block minus the issue time.
//   * NO PARTIAL TILES: a half step cannot keep 8 transient 32 x 32 partials (128 registers), so the per-128-k scale
//     s = As[m,kb] * Ws[e,nb,kb] is applied INSIDE the accumulate chain: s = 2^e * f with f in [1,2); the power of two
//     goes into the MX block scale of the token-side operand (E8M0 = the fp32 exponent field, exact), and the accumulator
//     is kept in units of the current mantissa: acc' <- acc' * (f_prev / f) once per k block (the same 16 VALU ops per
//     tile the old `acc += part * s` promotion cost), out = acc' * f_last.  fp32 rounding per step instead of none:
//     1e-7 relative per k block against the 1e-3 tolerance of the reference's block-fp8 tests.
//   * TWO WAVE GROUPS IN ANTI-PHASE: waves 0-3 (token half 0) and 4-7 (token half 1) share the four SIMDs pairwise and
//     run half a step apart, separated by workgroup barriers: while one wave of a SIMD issues its 8 MFMAs (segment M),
//     its partner issues LDS-DMA, reads operands and rescales (segment L).  Intervals:
//         X:  L0 | M0 | L1 | M1 | ...            (a barrier between all segments; Y executes one barrier more
//         Y:     | L0 | M0 | L1 | M1 | ...        up front, X one more at the end)
//   * XCD-aware tile order: the 32 workgroups of an XCD in one round take 32 CONSECUTIVE tiles (n fastest): the tiles of
//     one 512-row expert (2 x 16 tiles of w13) share their W and A panels through ONE L2.
#include "grouped_gemm_shared.h"

using namespace fl_gemm;

namespace {

#ifdef FL_GEMM2_TIMING
__device__ unsigned long long* g_g2dbg = nullptr;
#define GT(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); gt[i] += t__ - gl; gl = t__; } while (0)
#else
#define GT(i) do { } while (0)
#endif

constexpr int BMB = 256;                 // token rows per workgroup
constexpr int BNB = 256;                 // weight rows per workgroup
constexpr int BKH = 64;                  // half k block (bytes per row per ring slot)
constexpr int kWHalf = BNB * BKH;        // 16 KiB
constexpr int kAHalf = BMB * BKH;        // 16 KiB
constexpr int kSlot = kWHalf + kAHalf;   // 32 KiB
constexpr int kSlots = 4;
constexpr int kAsSlot = 8 * 64 * 4;      // 2 KiB: wave w stores the scales of tokens 32w + (lane & 31) at floats [64w, 64w + 64)
constexpr int kSmem2 = kSlots * kSlot + 2 * kAsSlot;   // 135,168 B

// MFMAs as inline asm (volatile: they keep their order relative to each other and to the LDS-DMA asm).  Scale operands: byte 0
// of a VGPR, E8M0; src0 (weights) always 2^0, src1 (tokens: one column per lane) the power-of-two part of the lane's
// block scale.  `s_nop 1`: the operands may have been written by the VALU just before (hipcc pads nothing inside asm).
__device__ __forceinline__ void mfma_zero(v16f& acc, const v8i a, const v8i b, const int sb) {
  asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %4 op_sel_hi:[0,0,0]"
               : "=&v"(acc)
               : "v"(a), "v"(b), "v"(kUnit), "v"(sb));
}
__device__ __forceinline__ void mfma_acc(v16f& acc, const v8i a, const v8i b, const int sb) {
  asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]"
               : "+v"(acc)
               : "v"(a), "v"(b), "v"(kUnit), "v"(sb));
}

// (The experiment switches of rounds 3-4 — packed rescale multiplies, refill pieces in the odd L segments, static priority for the younger
//  wave group, one barrier per half step, priority flips around the M segment, tile-local refill — live in
//  probes/r05_gemm_big2_lab_switches.patch.txt with their measured results: profiles/r03_gemm_big2_*.txt, profiles/r04_gemm_big2_bounding_ladder.txt.
//  The refill crosses tile boundaries: the stages in flight past the end of a tile belong to the next one.)
#define G2_BARRIER()                          \
  do {                                        \
    __builtin_amdgcn_sched_barrier(0);        \
    __builtin_amdgcn_s_barrier();             \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);        \
  } while (0)

__global__ __launch_bounds__(512, 1) void grouped_gemm_fp8_big2_kernel(const GemmParams p, const uint8_t* __restrict__ gA,
                                                                      const float* __restrict__ gAs,
                                                                      const uint8_t* __restrict__ gW,
                                                                      const float* __restrict__ gWs,
                                                                      const int32_t* __restrict__ gmeta) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[kSmem2];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wn = wave & 3, wm = wave >> 2;

  // ---- PERSISTENT tile walk: the launch has one workgroup per CU (135 KiB of LDS: one fits); workgroup b takes the slots
  //      b, b + grid, b + 2 grid, ... of the tile list.  Between two tiles of a workgroup nothing is torn down: no
  //      dispatch gap, the LDS stays allocated, and group X's epilogue + next prologue run beside group Y's last M segment.
  //      XCD-aware order (speed only): slot s runs on XCD s % 8 (grid % 8 == 0); within a round of 256 slots XCD x takes the
  //      logical tiles [32x, 32x + 32) of the round (n fastest: the 2 x 16 tiles of a 512-row expert share their W and A
  //      panels through ONE L2).  The last, partial round keeps the identity order. ----
  const int n_tiles = p.n_tiles;
  const int KB = p.K / BK;
  const int NH = 2 * KB;
  int gb = 0;   // stages issued by the tiles before this one (a multiple of 2): ring slot of stage hs = (gb + hs) & 3

  // What the refill and the epilogue of a tile need.  LDS-DMA sources: a half-step stage = 32 pieces of 1 KiB (16 rows x
  // 64 B): wave w issues W pieces 2w, 2w+1 and A pieces 2w, 2w+1.  Lane i of a piece lands at +16 i: row i >> 2, chunk
  // POSITION i & 3, which holds source chunk (i & 3) ^ ((row >> 2) & 3) — the swizzle is on the source address, the LDS
  // image is lane-linear (conflict-free ds_read_b128 below: a 16-lane read group covers all 16 bank slots of 16 B).
  struct TileSrc {
    const uint8_t* w_base;
    const uint8_t* a_base;
    const float* as_src;   // the block scales of token 32 wave + (lane & 31)
    unsigned vw0, vw1, va0, va1;
    int e, n0;
    long long row0, row_end;
  };
  auto setup_tile = [&](const int slot, TileSrc& t) -> bool {
    int lid = slot;
    const int round = lid >> 8;
    if ((round + 1) * 256 <= p.total_blocks) lid = (round << 8) + ((lid & 7) << 5) + ((lid & 255) >> 3);
    const int nt = lid % n_tiles, mt = lid / n_tiles;
    int e = 0;
    long long row0 = 0, row_end = 0;
    if (!locate_tile<BMB>(p, gmeta, mt, e, row0, row_end)) return false;   // (surplus slot of the upper-bound tile count: workgroup-uniform)
    t.e = e; t.row0 = row0; t.row_end = row_end; t.n0 = nt * BNB;
    t.w_base = gW + (long long)e * p.N * p.K;
    t.a_base = gA + row0 * p.K;
    const unsigned swz = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    const unsigned n_last = (unsigned)p.N - 1u;                    // rows beyond N: clamped, results discarded
    const unsigned m_last = (unsigned)(row_end - row0 - 1);        // rows beyond the group: clamped, never stored
    const unsigned r = (unsigned)(32 * wave + (lane >> 2));
    unsigned n = (unsigned)t.n0 + r;
    t.vw0 = __umul24(n < n_last ? n : n_last, (unsigned)p.K) + swz;
    n += 16u;
    t.vw1 = __umul24(n < n_last ? n : n_last, (unsigned)p.K) + swz;
    t.va0 = __umul24(r < m_last ? r : m_last, (unsigned)p.K) + swz;
    t.va1 = __umul24(r + 16u < m_last ? r + 16u : m_last, (unsigned)p.K) + swz;
    long long m = row0 + 32 * wave + li;
    m = m < row_end ? m : row_end - 1;
    t.as_src = p.mode == kMasked ? gAs + (long long)e * p.as_stride_g + (m - (long long)e * p.rows_per_group) * p.as_stride_m
                                 : gAs + m * p.as_stride_m;
    return true;
  };

  // ---- PERSISTENT tile walk: the launch has one workgroup per CU (135 KiB of LDS: one fits); workgroup b takes the slots
  //      b, b + grid, b + 2 grid, ... of the tile list.  Between two tiles of a workgroup nothing is torn down, and
  //      the refill runs ACROSS tiles: the three stages that are in flight past the end of a tile are the
  //      next tile's stages 0..2, the barrier that ends the tile is the next tile's first, and the next tile's lookup
  //      is done while its first stages fly.  XCD-aware order (speed only): slot s runs on XCD s % 8 (grid % 8 == 0); within
  //      a round of 256 slots XCD x takes the logical tiles [32x, 32x + 32) of the round (n fastest: the 2 x 16 tiles of a
  //      512-row expert share their W and A panels through ONE L2).  The last, partial round keeps the identity order. ----
  int slot = blockIdx.x;
  TileSrc cur;
  {
    bool found = false;
    for (; slot < p.total_blocks; slot += gridDim.x)
      if (setup_tile(slot, cur)) { found = true; break; }
    if (!found) return;
  }
  bool carried = false;   // stages 0..2 of `cur` are already in flight and its first barrier has been passed
#pragma unroll 1
  for (;;) {
#ifdef FL_GEMM2_TIMING
  const unsigned long long t_entry = __builtin_readcyclecounter();
  const unsigned long long w_entry = wall_clock64();
#endif
  TileSrc nxt = cur;
  bool has_next = false;
  int nslot = slot + gridDim.x;
  for (; nslot < p.total_blocks; nslot += gridDim.x)
    if (setup_tile(nslot, nxt)) { has_next = true; break; }
  const bool xt = has_next;   // the stages past the end of this tile belong to `nxt`
  const int e = cur.e, n0 = cur.n0;
  const long long row0 = cur.row0, row_end = cur.row_end;

  auto uniform = [](const uint8_t* ptr) {   // (keeps the 64-bit base in an SGPR pair: the asm operand is "s")
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const uint8_t*>(((unsigned long long)hi << 32) | lo);
  };
  // Piece k of this wave's share of stage hs: 0, 1 = W pieces, 2, 3 = A pieces, 4 = the token scales of k block hs / 2
  // (even stages only).  Stages NH .. NH+2 are the next tile's stages 0 .. 2 (or, without a next tile, re-fetches of the last
  // stage into idle slots): ONE loop body, constant vmcnt counts.
  auto issue_piece = [&](const int hs, const int k) {
    const bool nx = xt && hs >= NH;
    const int hc = nx ? hs - NH : (hs < NH ? hs : NH - 1);
    uint8_t* s = smem + ((gb + hs) & (kSlots - 1)) * kSlot + (2 * wave) * 1024;
    const uint8_t* wb = nx ? nxt.w_base : cur.w_base;
    const uint8_t* ab = nx ? nxt.a_base : cur.a_base;
    if (k == 0) fl_dma16_s(uniform(wb + (long long)hc * BKH), nx ? nxt.vw0 : cur.vw0, s);
    else if (k == 1) fl_dma16_s(uniform(wb + (long long)hc * BKH), nx ? nxt.vw1 : cur.vw1, s + 1024);
    else if (k == 2) fl_dma16_s(uniform(ab + (long long)hc * BKH), nx ? nxt.va0 : cur.va0, s + kWHalf);
    else if (k == 3) fl_dma16_s(uniform(ab + (long long)hc * BKH), nx ? nxt.va1 : cur.va1, s + kWHalf + 1024);
    else {
      const int kb = hs >> 1, kc = nx ? hc >> 1 : (kb < KB ? kb : KB - 1);
      fl_dma4((nx ? nxt.as_src : cur.as_src) + (long long)kc * p.as_stride_k,
              smem + kSlots * kSlot + (((gb >> 1) + kb) & 1) * kAsSlot + wave * 256);
    }
  };
  auto issue = [&](const int hs, const bool even) {
#pragma unroll
    for (int k = 0; k < 4; ++k) issue_piece(hs, k);
    if (even) issue_piece(hs, 4);
  };

  // operand read offsets: row li of a 32-row block (64 B per row), the lane half's 32 k bytes = chunks 2 lh, 2 lh + 1
  const int rb0 = li * BKH + ((((2 * lh) ^ ((li >> 2) & 3))) << 4);
  const int rb1 = li * BKH + ((((2 * lh + 1) ^ ((li >> 2) & 3))) << 4);
  auto ld8 = [&](const uint8_t* base) {
    const v4i lo = *reinterpret_cast<const v4i*>(base + rb0);
    const v4i hi = *reinterpret_cast<const v4i*>(base + rb1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  v16f acc[2][4];   // [weight-row block i][token block j]: D^T[32 weight rows, 32 tokens], one token per lane
  float cmant[4];   // per token block j: the mantissa the accumulators of column j are currently in units of
  int e8[4];        // E8M0 scale of the current k block (= the fp32 exponent field of As * Ws)
  float ratio[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { cmant[j] = 1.f; e8[j] = kUnit; ratio[j] = 1.f; }

  // (weight rows beyond N are clamped and their results discarded: their scale row is clamped too — no read past the tensor)
  const int nb_last = (p.N + BN - 1) / BN - 1;
  const int nb_w = (n0 + 64 * wn) / BN;
  const float* wsrow = gWs + ((long long)e * (nb_last + 1) + (nb_w < nb_last ? nb_w : nb_last)) * KB;
  v8i wa[2], tb[4];

  // ---- segment L: operand reads, (first half of a k block:) scales + rescale of token blocks 2, 3.  No LDS-DMA here: the
  //      four L waves of an interval would queue 18 pieces behind each other at the CU's vector-memory path (measured:
  //      L segments of ~810 cycles, the slowest wave of a group sets the interval) — the refill rides behind the MFMAs ----
  auto seg_load = [&](const int h, const bool even, const bool first_kb) {
    const uint8_t* sw = smem + ((gb + h) & (kSlots - 1)) * kSlot + (64 * wn) * BKH;
    const uint8_t* sa = smem + ((gb + h) & (kSlots - 1)) * kSlot + kWHalf + (128 * wm) * BKH;
    wa[0] = ld8(sw);
    wa[1] = ld8(sw + 32 * BKH);
#pragma unroll
    for (int j = 0; j < 4; ++j) tb[j] = ld8(sa + j * (32 * BKH));
    if (even) {
      const int kb = h >> 1;
      const float ws = wsrow[kb];
      const float* sas = reinterpret_cast<const float*>(smem + kSlots * kSlot + (((gb >> 1) + kb) & 1) * kAsSlot) + 64 * (4 * wm) + li;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float s = sas[64 * j] * ws;
        const unsigned bits = __float_as_uint(s);
        const unsigned eb = (bits >> 23) & 0xffu;
        // zero / denormal scale: 2^-127 x 1.0 (the term is below anything fp32 can add to the sum anyway)
        const float f = eb != 0u ? __uint_as_float((bits & 0x807fffffu) | 0x3f800000u) : 1.f;
        ratio[j] = cmant[j] * __builtin_amdgcn_rcpf(f);
        cmant[j] = f;
        e8[j] = (int)eb;
      }
      if (!first_kb) {
        // the MFMAs of these tiles were the FIRST four of the previous M segment: complete long before this point (four
        // more MFMAs, a barrier, the operand reads and the scale arithmetic lie in between) — the XDL-write -> VALU-read rule
        // that hipcc cannot see through the asm is met by the schedule
        asm volatile("" : "+v"(acc[0][2]), "+v"(acc[1][2]), "+v"(acc[0][3]), "+v"(acc[1][3]));
#pragma unroll
        for (int j = 2; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] *= ratio[j];
            asm volatile("" : "+v"(acc[i][j]));
          }
      }
    }
  };
  // ---- segment M of half step h: 8 MFMAs, token blocks 2, 3 first, and this wave's pieces of stage h + 3, one behind an
  //      MFMA each (the piece's issue stall hides under the 64 cycles of the MFMA in the pipe).  First half of a k block:
  //      the rescale of blocks 0, 1 rides behind the first four MFMAs, the pieces behind the last four ----
  auto seg_mma = [&](const int h, const bool even, const bool first_kb) {
    if (even && first_kb) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int j = (2 + (t >> 1)) & 3, i = t & 1;
        mfma_zero(acc[i][j], wa[i], tb[j], e8[j]);
        if (t >= 4) issue_piece(h + 3, t - 4);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = 2 + (t >> 1), i = t & 1;
        mfma_acc(acc[i][j], wa[i], tb[j], e8[j]);
        if (even) {   // rescale tile t of the blocks 0, 1 (its last MFMA was issued a whole M + L segment ago)
          const int jr = t >> 1, ir = t & 1;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[ir][jr][r] *= ratio[jr];
          asm volatile("" : "+v"(acc[ir][jr]));
        } else {
          issue_piece(h + 3, t);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = t >> 1, i = t & 1;
        mfma_acc(acc[i][j], wa[i], tb[j], e8[j]);
        if (even) issue_piece(h + 3, t);
        else if (t == 0) issue_piece(h + 3, 4);   // (h odd: stage h + 3 is even — it carries the scales)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

#ifdef FL_GEMM2_TIMING
  unsigned long long gt[4] = {0, 0, 0, 0};
  unsigned long long gl = __builtin_readcyclecounter();
  const unsigned long long g0 = gl;
#endif
  // ---- prologue: three half-step stages in flight (a carried tile's are: the previous tile issued them) ----
  if (!carried) {
    issue(0, true);
    issue(1, false);
    issue(2, true);
  }

  // One k block of each group's stream.  `first` is a literal at both call sites (k block 0 is peeled off the loop: ONE body
  // variant inside the loop — a run-time branch around the MFMAs makes hipcc copy the accumulators at the join).
  auto kblock_x = [&](const int kb, const bool first, const bool skip_head) __attribute__((always_inline)) {
    const int h = 2 * kb;
    GT(1);
    if (!skip_head) {   // (carried tile: the wait + barrier that ended the previous tile were these)
      asm volatile("s_waitcnt vmcnt(9)" ::: "memory");   // this wave's pieces of stage h (h + 1, h + 2 stay in flight)
      GT(2);
      G2_BARRIER();                                      // everyone's; slot h - 1 is free
    }
    GT(3);
    seg_load(h, true, first);
    GT(0);
    // (k block 0 keeps this barrier: its M segment issues stage 3 into the slot that was the PREVIOUS tile's epilogue staging area,
    //  and group Y passes this barrier only after its own epilogue)
    G2_BARRIER();
    GT(3);
    seg_mma(h, true, first);
    GT(1);
    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    GT(2);
    G2_BARRIER();
    GT(3);
    seg_load(h + 1, false, false);
    GT(0);
    G2_BARRIER();
    GT(3);
    seg_mma(h + 1, false, false);
  };
  auto kblock_y = [&](const int kb, const bool first) __attribute__((always_inline)) {
    const int h = 2 * kb;
    GT(1);
    G2_BARRIER();
    GT(3);
    seg_load(h, true, first);
    // reads of slot h complete (group X refills it right after the next barrier) + this wave's pieces of stage h + 1:
    // only stage h + 2 (even: 5 pieces, issued in M of step h - 1) may stay in flight — stage h + 3 goes out in the M below
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GT(0);
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    GT(2);
    G2_BARRIER();
    GT(3);
    seg_mma(h, true, first);
    GT(1);
    G2_BARRIER();
    GT(3);
    seg_load(h + 1, false, false);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GT(0);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // stage h + 3 (odd: 4 pieces) stays in flight
    GT(2);
    G2_BARRIER();
    GT(3);
    seg_mma(h + 1, false, false);
  };
  if (wm == 0) {   // ---------------- group X ----------------
    kblock_x(0, true, carried);
#pragma unroll 1
    for (int kb = 1; kb < KB; ++kb) kblock_x(kb, false, false);
    GT(1);
    // the next tile's stage 0 (= stage NH: 9 younger pieces stay in flight) has landed before the barrier that ends this
    // tile AND opens the next one.  The epilogue's stores come BEHIND this wait: stores may retire out of order with
    // the loads, so a count taken after them could be met with a stage-0 piece still in flight; every later wait has
    // at least one whole younger stage behind the one it waits for, which in-order load return makes sufficient
    if (xt) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    G2_BARRIER();   // (group Y's last M segment starts here)
  } else {         // ---------------- group Y: the same stream, one segment later ----------------
    if (!carried) {   // (carried: group Y's wait at the end of the previous tile's last L segment was for this stage)
      asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      G2_BARRIER();
    }
    kblock_y(0, true);
#pragma unroll 1
    for (int kb = 1; kb < KB; ++kb) kblock_y(kb, false);
    GT(1);
  }
#ifdef FL_GEMM2_TIMING
  const unsigned long long t_loop_end = __builtin_readcyclecounter();
#endif

  // ---- epilogue: D^T[n, m] -> out[m, n] bf16; lane (m = token li of block j, half lh) holds n = 8g + 4lh + (0..3) ----
  // (the last MFMAs are still in the pipe: a 16-pass XDL write needs 18 wait states before a VALU read)
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(acc[0][1]), "+v"(acc[1][1]));
  // Through LDS: a lane holds ONE token row, 4 weight columns at a time — stored directly, every store instruction touches
  // 32 rows x 16 B (measured: 18-20 k cycles per tile, as long as five k blocks; the epilogue of a K = 2048 tile was 27 % of
  // its time).  The ring is idle here (every wave's operand reads are behind the last barrier this wave passed, and the next
  // tile's stages 0..2 go to slots 0..2): each wave transposes one 32-token block at a time through 4 KiB of slot 3
  // ([32 rows][128 B], 16-B chunk c of row r at position c ^ (r & 7): conflict-free both ways) and stores 128-B row
  // segments, 8 rows per instruction.
  {
    // (slot of the LAST stage: the three stages in flight past the end of the tile — re-fetches of the last stage today, the
    //  next tile's first stages once the prefetch crosses tiles — go to the other three; the next tile's stage 3 is issued
    //  only behind two more workgroup barriers, which the other group passes after ITS epilogue)
    // (every wave has its OWN 4 KiB: the two groups' epilogues overlap in time — group Y's starts one M segment, ~700 cycles,
    //  after group X's)
    uint8_t* stg = smem + ((gb + NH - 1) & (kSlots - 1)) * kSlot + wave * 4096;
    const int rr = lane >> 3, rc = lane & 7;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float cm = cmant[j];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<uint2*>(stg + li * 128 + (((4 * i + g) ^ (li & 7)) << 4) + 8 * lh) =
              make_uint2(fl_pack_bf16(acc[i][j][4 * g + 0] * cm, acc[i][j][4 * g + 1] * cm),
                         fl_pack_bf16(acc[i][j][4 * g + 2] * cm, acc[i][j][4 * g + 3] * cm));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (wave-private buffer: the wave's own writes, in order)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = 8 * k + rr;
        const uint4 v = *reinterpret_cast<const uint4*>(stg + r * 128 + ((rc ^ (r & 7)) << 4));
        const long long m = row0 + 128 * wm + 32 * j + r;
        const int n = n0 + 64 * wn + 8 * rc;
        if (m < row_end) {
          uint16_t* orow = p.out + m * p.N;
          if (n + 8 <= p.N) {
            *reinterpret_cast<uint4*>(orow + n) = v;
          } else {
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int x = 0; x < 8; ++x)
              if (n + x < p.N) orow[n + x] = (uint16_t)(w[x >> 1] >> (16 * (x & 1)));
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next block overwrites the buffer
    }
  }
#ifdef FL_GEMM2_TIMING
  const unsigned long long t_epi_issued = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (g_g2dbg != nullptr && lane == 0 && slot < 8192) {
    unsigned long long* d = g_g2dbg + ((long long)slot * 8 + wave) * 8;
    d[0] = gt[0]; d[1] = gt[1]; d[2] = gt[2]; d[3] = gt[3]; d[4] = t_loop_end - g0;
    d[5] = t_epi_issued - t_loop_end;                       // epilogue: instructions issued (stores not yet acknowledged)
    d[6] = __builtin_readcyclecounter() - t_loop_end;      // epilogue incl. the wait for its stores
    d[7] = w_entry;                                        // wall clock (100 MHz) at entry: workgroup start times
  }
#endif
  gb += NH;
  if (!has_next) break;
  cur = nxt;
  slot = nslot;
  carried = xt;
  }   // persistent tile walk
}

}  // namespace

#ifdef FL_GEMM2_TIMING
extern "C" int fl_gemm2_debug_set_buffer(unsigned long long* dev_ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_g2dbg), &dev_ptr, sizeof(dev_ptr));
}
#endif

int fl_gemm_launch_big2(const GemmParams& p_in, const void* A, const float* As, const void* W, const float* Ws,
                        const int32_t* group_meta, hipStream_t stream) {
  GemmParams p = p_in;
  long long m_tiles;
  if (p.mode == kOffset) m_tiles = (p.M + BMB - 1) / BMB + p.E;
  else if (p.mode == kMasked) m_tiles = (long long)p.E * ((p.rows_per_group + BMB - 1) / BMB);
  else m_tiles = (p.M + BMB - 1) / BMB;
  p.n_tiles = (p.N + BNB - 1) / BNB;
  p.m_tiles_upper = (int)m_tiles;
  const long long blocks = m_tiles * p.n_tiles;
  FL_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "fl_grouped_gemm_fp8: grid too large");
  FL_CHECK_ARG(p.N < (1 << 24) && p.K < (1 << 24) && (long long)p.N * p.K < (1ll << 32),
               "fl_grouped_gemm_fp8: N*K too large for the 256x256 tile");
  p.total_blocks = (int)blocks;
  // one workgroup per CU walks the tile list
  long long grid = blocks;
  {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && fl_device_cu_count(dev, &cus) == FL_OK && cus > 0 && grid > cus) grid = cus;
  }
  grouped_gemm_fp8_big2_kernel<<<dim3((unsigned)grid), dim3(512), 0, stream>>>(p, (const uint8_t*)A, As, (const uint8_t*)W, Ws,
                                                                                group_meta);
  FL_CHECK_LAUNCH("grouped_gemm_fp8_big2_kernel");
  return FL_OK;
}
