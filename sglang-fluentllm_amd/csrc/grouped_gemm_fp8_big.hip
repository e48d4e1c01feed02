// G1-G3 for MANY rows per group (prefill / large-batch regime): 256 weight rows x 256 token rows per workgroup.
// Same math, call sites and data formats as grouped_gemm_fp8.hip (deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_*).
//
// Why a second tile shape: with the 128 x 128 tile a CU has to pull 32 KiB through its vector-memory path (64 B/clk
// peak) and 20 KiB per wave through LDS for every 8 MFMAs per wave — both as long as the MFMAs themselves
// (rocprof: matrix pipe 29 % busy at T=16384, HBM traffic 1.3x the unique bytes, i.e. not HBM).  Here
//   * ONE workgroup of 8 waves (two per SIMD, 256 registers each) owns a 256 x 256 output tile: 64 KiB of operands per
//     k block for 16 MFMAs per wave — half the ingest and 0.6x the LDS bytes per flop;
//   * wave (wn, wm) owns weight rows [64 wn, +64) x tokens [128 wm, +128): 2 x 4 tiles of 32 x 32, accumulators in
//     128 VGPRs; the per-128-k promotion acc += part * As[m,kb] * Ws[e,nb,kb] needs only ONE transient partial tile per
//     MFMA pair (a lane owns one token row: a single per-lane factor per tile);
//   * HBM -> LDS by LDS-DMA (inline asm, see fl_common.h) into a 2-stage ring of [W 32 KiB | A 32 KiB | As 1 KiB]
//     k-block stages; the refill of the other stage is issued piece by piece between the MFMAs of the current one.
#include "grouped_gemm_shared.h"

using namespace fl_gemm;

namespace {

#ifdef FL_GEMM_TIMING
__device__ unsigned long long* g_gdbg = nullptr;
#define GT(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); gt[i] += t__ - gl; gl = t__; } while (0)
#else
#define GT(i) do { } while (0)
#endif

constexpr int BMB = 256;                 // token rows per workgroup
constexpr int BNB = 256;                 // weight rows per workgroup
constexpr int kWTile = BNB * BK;         // 32 KiB
constexpr int kATile = BMB * BK;         // 32 KiB
constexpr int kStageB = kWTile + kATile + BMB * 4;
constexpr int kStagesB = 2;

// MFMAs as inline asm with VGPR accumulators: volatile asm statements keep their program order relative to each other
// and to the LDS-DMA asm, i.e. the software pipeline below is fixed in the source (as builtins, the pure MFMA calls are
// sunk to their first use: all 16 in one cluster after the last DMA piece).  The XDL-write -> VALU-read hazard that
// hipcc cannot see through asm is covered by the schedule: tile t-1 is promoted only after BOTH MFMAs of tile t have
// issued — the matrix pipe is in order, so tile t-1's results are then complete (+ s_nop for margin).
__device__ __forceinline__ void mfma_first(v16f& acc, const v8i a, const v8i b) {
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %3 op_sel_hi:[0,0,0]"
               : "=&v"(acc)
               : "v"(a), "v"(b), "v"(kUnit));
}
__device__ __forceinline__ void mfma_acc(v16f& acc, const v8i a, const v8i b) {
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
               : "+v"(acc)
               : "v"(a), "v"(b), "v"(kUnit));
}

__global__ __launch_bounds__(512, 1) void grouped_gemm_fp8_big_kernel(const GemmParams p, const uint8_t* __restrict__ gA,
                                                                     const float* __restrict__ gAs,
                                                                     const uint8_t* __restrict__ gW,
                                                                     const float* __restrict__ gWs,
                                                                     const int32_t* __restrict__ gmeta) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[kStagesB * kStageB];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wn = wave & 3, wm = wave >> 2;

  const int n_tiles = (p.N + BNB - 1) / BNB;
  const int nt = blockIdx.x % n_tiles;
  const int mt = blockIdx.x / n_tiles;
  int e = 0;
  long long row0 = 0, row_end = 0;
  if (!locate_tile<BMB>(p, gmeta, mt, e, row0, row_end)) return;
  const int n0 = nt * BNB;
  const int KB = p.K / BK;

  // ---- LDS-DMA sources: piece q = 4*wave + k of a tile = rows 8q + (lane>>3); LDS chunk position lane&7 holds source
  //      chunk (lane&7) ^ ((row>>1)&7) (conflict-free ds_read_b128 below) ----
  const uint8_t* w_base = gW + (long long)e * p.N * p.K;      // + per-lane row offset + kb*128
  const uint8_t* a_base = gA + row0 * p.K;
  // per-lane offsets are rebuilt per piece (2 VALU ops) from 4 registers: the row of piece k is r0 + 8k, and the
  // swizzle term (row>>1)&7 = (4k + (lane>>4))&7 takes two values (k even / odd)
#ifndef FL_GEMM_ASYM
#define FL_GEMM_ASYM 0   // experiment (measured: 1570 vs 1584 TFLOP/s on w13 at T=16384, w2 equal — no gain): the refill of a
                         // stage issued by ONE wave per SIMD (waves 0..3: 8 W + 8 A pieces + the scales beside the first tile of
                         // the k block) so that the partner waves run their MFMAs during the loaders' issue stalls.  That the
                         // per-wave LDS-DMA issue stall (~180 cycles per piece) is NOT what holds the matrix pipe at 48 % is the
                         // finding: the k block takes the same time whoever issues the pieces
#endif
  int r0 = (FL_GEMM_ASYM ? 64 * (wave & 3) : 32 * wave) + (lane >> 3);
  unsigned swz0 = (((lane & 7) ^ ((lane >> 4) & 7)) << 4), swz1 = (((lane & 7) ^ ((4 + (lane >> 4)) & 7)) << 4);
  const unsigned n_last = (unsigned)p.N - 1u;                    // rows beyond N: clamped, results discarded
  const unsigned m_last = (unsigned)(row_end - row0 - 1);        // rows beyond the group: clamped, never stored
  auto w_off = [&](const int k) {
    unsigned n = (unsigned)(n0 + r0 + 8 * k);
    n = n < n_last ? n : n_last;
    return __umul24(n, (unsigned)p.K) + ((k & 1) ? swz1 : swz0);
  };
  auto a_off = [&](const int k) {
    unsigned m = (unsigned)(r0 + 8 * k);
    m = m < m_last ? m : m_last;
    return __umul24(m, (unsigned)p.K) + ((k & 1) ? swz1 : swz0);
  };
  const float* as_src = nullptr;   // waves 0..3: the 64 tokens 64*wave + lane
  if (wave < 4) {
    long long m = row0 + 64 * wave + lane;
    m = m < row_end ? m : row_end - 1;
    as_src = p.mode == kMasked ? gAs + (long long)e * p.as_stride_g + (m - (long long)e * p.rows_per_group) * p.as_stride_m
                               : gAs + m * p.as_stride_m;
  }
  auto stage = [&](int st) { return smem + st * kStageB; };
  auto uniform = [](const uint8_t* ptr) {   // (keeps the 64-bit base in an SGPR pair: the asm operand is "s")
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const uint8_t*>(((unsigned long long)hi << 32) | lo);
  };
  auto dma_piece = [&](const int kb, const int k) {   // k = 0..3: W, 4..7: A, 8: As
    uint8_t* s = stage(kb & 1);
#if FL_GEMM_ASYM   // loader waves 0..3: k = 0..7: W pieces 8*wave + k, 8..15: A pieces, 16: As
    if (k < 8) fl_dma16_s(uniform(w_base + (long long)kb * BK), w_off(k), s + (8 * wave + k) * 1024);
    else if (k < 16) fl_dma16_s(uniform(a_base + (long long)kb * BK), a_off(k - 8), s + kWTile + (8 * wave + k - 8) * 1024);
    else fl_dma4(as_src + (long long)kb * p.as_stride_k, s + kWTile + kATile + wave * 256);
#else
    if (k < 4) fl_dma16_s(uniform(w_base + (long long)kb * BK), w_off(k), s + (4 * wave + k) * 1024);
    else if (k < 8) fl_dma16_s(uniform(a_base + (long long)kb * BK), a_off(k - 4), s + kWTile + (4 * wave + k - 4) * 1024);
    else if (wave < 4) fl_dma4(as_src + (long long)kb * p.as_stride_k, s + kWTile + kATile + wave * 256);
#endif
  };

  // operand read offsets inside a 32-row x 128 B sub-tile: row li, 16-B chunk c = 4*s2 + 2*lh + e2 (s = 2*s2 + e2)
  int rb[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int c = 4 * (s >> 1) + 2 * lh + (s & 1);
    rb[s] = li * BK + ((c ^ ((li >> 1) & 7)) << 4);
  }

  // acc += part * sc: 8 packed fp32 FMAs (v_pk_fma_f32) per tile instead of 16 scalar ones with FL_GEMM_PKFMA
  auto promote = [](v16f& a, const v16f& pt, const float sc) {
#ifdef FL_GEMM_PKFMA
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f s2 = {sc, sc};
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const v2f x = {pt[2 * r], pt[2 * r + 1]};
      v2f y = {a[2 * r], a[2 * r + 1]};
      y = __builtin_elementwise_fma(x, s2, y);
      a[2 * r] = y[0];
      a[2 * r + 1] = y[1];
    }
#else
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = fmaf(pt[r], sc, a[r]);
#endif
  };
  v16f acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float* wsrow = gWs + ((long long)e * ((p.N + BN - 1) / BN) + (n0 + 64 * wn) / BN) * KB;

  // ---- prologue: stage 0 ----
#if FL_GEMM_ASYM
  if (wave < 4) {
#pragma unroll
    for (int k = 0; k < 17; ++k) dma_piece(0, k);
  }
#else
#pragma unroll
  for (int k = 0; k < 9; ++k) dma_piece(0, k);
#endif

#ifdef FL_GEMM_TIMING
  unsigned long long gt[4] = {0, 0, 0, 0};
  unsigned long long gl = __builtin_readcyclecounter();
  const unsigned long long g0 = gl;
#endif
  for (int kb = 0; kb < KB; ++kb) {
    const float ws = wsrow[kb];
    asm volatile("" : "+v"(r0), "+v"(swz0), "+v"(swz1));   // (not loop-invariant to LICM: offsets are rebuilt, not hoisted + spilled)
    GT(2);   // compute (previous k block)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of stage kb have landed
    GT(0);   // wait for own pieces
    __builtin_amdgcn_s_barrier();                       // ... everyone's; and everyone is done with the other stage
    GT(1);   // barrier
    const uint8_t* sw = stage(kb & 1) + (64 * wn) * BK;
    const uint8_t* sa = stage(kb & 1) + kWTile + (128 * wm) * BK;
    const float* sas = reinterpret_cast<const float*>(stage(kb & 1) + kWTile + kATile) + 128 * wm;
    // the refill of the other stage is unconditional (the last k block re-fetches itself into the idle stage: 1/KB
    // wasted bytes) so that the body is ONE basic block
    const int kn = kb + 1 < KB ? kb + 1 : kb;
    // operands as 8-register tuples assembled from two 16-B LDS reads each (concat: no copies)
    auto ld8 = [&](const uint8_t* base, const int s) {
      const v4i lo = *reinterpret_cast<const v4i*>(base + rb[2 * s]);
      const v4i hi = *reinterpret_cast<const v4i*>(base + rb[2 * s + 1]);
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    v8i wa[2][2];   // [weight-row block][k half]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) wa[i][s] = ld8(sw + i * (32 * BK), s);
    v8i b0 = ld8(sa, 0), b1 = ld8(sa, 1);
    float sc = sas[li] * ws;
    v16f part[2];
    float sc_prev = 0.f;
    __builtin_amdgcn_sched_barrier(0);
    // 8 tiles (j = token tile, i = weight-row block), two transient partial tiles: the MFMA pair of tile t is issued,
    // then — while it runs — one refill piece, the next token tile's operands (after the tile's 4th MFMA has issued,
    // i.e. read its operands) and the promotion of tile t-1.  XDL write -> VALU read has no hardware interlock and hipcc
    // cannot see through the asm: tile t-1 is read only after BOTH MFMAs of tile t have issued; the matrix pipe is in
    // order, so its results are complete by then (+ s_nop margin).
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int j = t >> 1, i = t & 1;
#ifndef FL_GEMM_NOCOMPUTE   // experiment switch: operand streaming only (results are garbage)
      __builtin_amdgcn_s_setprio(1);   // the two waves of a SIMD are at different phases: the one with MFMAs to issue goes
      mfma_first(part[t & 1], wa[i][0], b0);   // first (+2.7 % / +1.5 % on w13 / w2 at T=16384)
      mfma_acc(part[t & 1], wa[i][1], b1);
      __builtin_amdgcn_s_setprio(0);
#else
#pragma unroll
      for (int r = 0; r < 16; ++r) part[t & 1][r] = (float)(wa[i][0][0] + b0[1] + wa[i][1][2] + b1[3]);
#endif
      const float sc_now = sc;
      if (i == 1 && j < 3) {   // operands + scale of the next token tile
        b0 = ld8(sa + (j + 1) * (32 * BK), 0);
        b1 = ld8(sa + (j + 1) * (32 * BK), 1);
        sc = sas[32 * (j + 1) + li] * ws;
      }
#ifndef FL_GEMM_STAGGER
#define FL_GEMM_STAGGER 1
#endif
#if FL_GEMM_ASYM
      if (t == 0 && wave < 4) {
#pragma unroll
        for (int k = 0; k < 17; ++k) dma_piece(kn, k);
      }
#elif FL_GEMM_STAGGER >= 0
      // wave pairs take turns at the CU's vector-memory path: waves 2s, 2s+1 issue their whole refill (9 pieces) beside
      // tile s = 0..3 instead of all eight waves issuing one piece beside every tile (+2.5 % on w13 at T=16384, w2 equal;
      // one wave per tile over all 8 tiles: -1.5 % — the late refills land after the next barrier)
      if (t == (wave >> FL_GEMM_STAGGER)) {
#pragma unroll
        for (int k = 0; k < 9; ++k) dma_piece(kn, k);
      }
#else
      dma_piece(kn, t);
      if (t == 7) dma_piece(kn, 8);
#endif
      if (t > 0) {
        const int tp = t - 1, jp = tp >> 1, ip = tp & 1;
        asm volatile("s_nop 3" : "+v"(part[tp & 1]));
        promote(acc[ip][jp], part[tp & 1], sc_prev);
        asm volatile("" : "+v"(acc[ip][jp]));   // pin: the promotion happens HERE (else it is sunk to the end of the k block
                                              // and every partial tile stays live: spills)
      }
      sc_prev = sc_now;
      __builtin_amdgcn_sched_barrier(0);
    }
    // the last tile: a 16-pass XDL write needs 18 wait states (of 4 clocks: probes/probe_snop.hip measures s_nop 7 = 36
    // clocks) before a VALU read; its pair issued before the last refill piece and the promotion of tile 6 (> 250 clocks)
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(part[1]));
    promote(acc[1][3], part[1], sc_prev);
    asm volatile("" : "+v"(acc[1][3]));
  }

#ifdef FL_GEMM_TIMING
  GT(2);
  if (g_gdbg != nullptr && lane == 0 && blockIdx.x < 4096) {
    unsigned long long* d = g_gdbg + ((long long)blockIdx.x * 8 + wave) * 4;
    d[0] = gt[0]; d[1] = gt[1]; d[2] = gt[2]; d[3] = __builtin_readcyclecounter() - g0;
  }
#endif
  // ---- epilogue: D^T[n, m] -> out[m, n] bf16; lane (m = token li of tile j, half lh) holds n = 8g + 4lh + (0..3) ----
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long m = row0 + 128 * wm + 32 * j + li;
    if (m < row_end) {
      uint16_t* orow = p.out + m * p.N;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + 64 * wn + 32 * i + 8 * g + 4 * lh;
          if (n + 3 < p.N) {
            *reinterpret_cast<uint2*>(orow + n) = make_uint2(fl_pack_bf16(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1]),
                                                            fl_pack_bf16(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]));
          } else {
#pragma unroll
            for (int x = 0; x < 4; ++x)
              if (n + x < p.N) orow[n + x] = fl_f32_to_bf16(acc[i][j][4 * g + x]);
          }
        }
      }
    }
  }
}

}  // namespace

#ifdef FL_GEMM_TIMING
extern "C" int fl_gemm_debug_set_buffer(unsigned long long* dev_ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_gdbg), &dev_ptr, sizeof(dev_ptr));
}
#endif

int fl_gemm_launch_big(const GemmParams& p_in, const void* A, const float* As, const void* W, const float* Ws,
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
  grouped_gemm_fp8_big_kernel<<<dim3((unsigned)blocks), dim3(512), 0, stream>>>(p, (const uint8_t*)A, As, (const uint8_t*)W, Ws,
                                                                               group_meta);
  FL_CHECK_LAUNCH("grouped_gemm_fp8_big_kernel");
  return FL_OK;
}
