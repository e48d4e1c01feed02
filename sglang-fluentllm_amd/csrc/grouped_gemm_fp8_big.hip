// G1-G4, compute regime (>= 128 rows per expert: prefill / large-batch EP) — FP8 block-scaled grouped GEMM with a
// 256 (weight rows) x 128 (tokens) workgroup tile, gfx950 only.  Same math, operands, grouping modes and call sites as
// grouped_gemm_fp8.hip (deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_{offset,contiguous,masked}, gemm_fp8_fp8_bf16_nt).
//
// Why a second tile: the 128x128 tile moves 32 KiB per 4.2 MFLOP — at 840 TFLOP/s it already pulls 6.5 TB/s through
// L2 -> LDS — and runs ONE wave per SIMD, whose LDS-DMA pieces, operand reads, MFMAs and scale FMAs issue in order.  Here:
//   * workgroup = 8 waves (TWO per SIMD, 256 registers each) as 4 x 2: wave (wn, wt) owns weight rows
//     [64 wn, 64 wn + 64) x tokens [64 wt, 64 wt + 64): 4 accumulator tiles.  48.5 KiB per 8.4 MFLOP k block
//     (5.8 KiB/MFLOP).  While one wave of a SIMD waits for LDS or folds partials, the other one feeds the matrix pipe.
//   * "SwapAB" as before: weights on the MFMA M side, tokens on the N side, so the block scale As[m,kb] * Ws[e,nb,kb]
//     is ONE per-lane factor.  Each tile's two k-steps accumulate into a fresh partial (v_mfma_scale 32x32x64, unit E8M0
//     scales), which 16 VALU FMAs fold into the fp32 accumulator.
//   * HBM/L2 -> LDS by global_load_lds into a 3-stage ring (7 pieces per wave per stage), counted vmcnt, one s_barrier
//     per k block; -fno-slp-vectorize (packed-f32 VALU stalls behind MFMAs on gfx950).
#include "fl_common.h"

namespace {

constexpr int BNB = 256;            // weight rows per workgroup
constexpr int BMB = 128;            // token rows per workgroup
constexpr int BK = 128;             // k block (bytes per row per stage)
constexpr int kStages = 3;
constexpr int kWB = BNB * BK;       // 32 KiB
constexpr int kAB = BMB * BK;       // 16 KiB
constexpr int kStageBytes = kWB + kAB + BMB * 4;
constexpr int kPieces = 4 + 2 + 1;  // LDS-DMA instructions per wave per stage (8 waves)
constexpr int kUnit = 0x7F;

enum Mode { kOffset = 0, kContiguous = 1, kMasked = 2, kDense = 3 };

struct BigParams {
  int mode, E, M, N, K;
  int n_tiles;
  long long as_stride_m, as_stride_k, as_stride_g;
  long long rows_per_group;
  uint16_t* out;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

#define FL_SLOT_END() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ v8i mk8(uint4 a, uint4 b) {
  v8i r;
  r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
  return r;
}

// One k block of one wave.  LDS regions are distinct __restrict__ parameters of ONE inlined function (otherwise hipcc
// drains the LDS-DMA queue before every ds_read, see mla_decode_fp8.hip).
__device__ __forceinline__ void kblock(v16f (&acc)[2][2], const float ws, const uint8_t* __restrict__ rd_w,
                                       const uint8_t* __restrict__ rd_a, const float* __restrict__ rd_as,
                                       uint8_t* __restrict__ dma_w, uint8_t* __restrict__ dma_a,
                                       float* __restrict__ dma_as, const uint8_t* __restrict__ src_w,
                                       const uint8_t* __restrict__ src_a, const float* __restrict__ src_as,
                                       const unsigned (&woff)[4], const unsigned (&aoff)[2], const unsigned asoff,
                                       const bool issue, const bool next_in_flight, const int (&rb)[4], const int wave,
                                       const int wn, const int wt, const int li) {
  // ---- stage kb landed for every wave; every wave is done with stage kb-1 (its slot is refilled below) ----
  if (next_in_flight) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");   // leave stage kb+1 in flight
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (issue) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src_w + woff[k]), (lds_ptr_t)(dma_w + (wave * 4 + k) * 1024), 16, 0, 0);
#pragma unroll
    for (int k = 0; k < 2; ++k)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src_a + aoff[k]), (lds_ptr_t)(dma_a + (wave * 2 + k) * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src_as + asoff), (lds_ptr_t)(dma_as + (wave & 1) * 64), 4, 0, 0);
  }
  uint4 w[2][4], a[2][4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) w[nt][s] = *reinterpret_cast<const uint4*>(rd_w + (64 * wn + 32 * nt) * BK + rb[s]);
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) a[tt][s] = *reinterpret_cast<const uint4*>(rd_a + (64 * wt + 32 * tt) * BK + rb[s]);
  }
  float sc[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) sc[tt] = rd_as[64 * wt + 32 * tt + li] * ws;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      v16f part;
#pragma unroll
      for (int r = 0; r < 16; ++r) part[r] = 0.f;
      part = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(mk8(w[nt][0], w[nt][1]), mk8(a[tt][0], a[tt][1]), part, 0, 0, 0,
                                                            kUnit, 0, kUnit);
      part = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(mk8(w[nt][2], w[nt][3]), mk8(a[tt][2], a[tt][3]), part, 0, 0, 0,
                                                            kUnit, 0, kUnit);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][tt][r] = fmaf(part[r], sc[tt], acc[nt][tt][r]);
    }
}

__global__ __launch_bounds__(512, 1) void grouped_gemm_fp8_big_kernel(const BigParams p, const uint8_t* __restrict__ gA,
                                                                      const float* __restrict__ gAs,
                                                                      const uint8_t* __restrict__ gW,
                                                                      const float* __restrict__ gWs,
                                                                      const int32_t* __restrict__ gmeta) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[kStages * kStageBytes];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wt = wave >> 2;
  const int li = lane & 31, lh = lane >> 5;

  // ---- tile -> (expert, token range, weight tile): as in grouped_gemm_fp8.hip with BM = 128 ----
  const int nt_id = blockIdx.x % p.n_tiles;
  const int mt_id = blockIdx.x / p.n_tiles;
  int e = 0;
  long long row0 = 0, row_end = 0;
  if (p.mode == kOffset) {
    int t = mt_id, found = 0;
    for (int g = 0; g < p.E; ++g) {
      const int lo = gmeta[g], hi = gmeta[g + 1];
      const int tiles = (hi - lo + BMB - 1) / BMB;
      if (t < tiles) { e = g; row0 = lo + (long long)t * BMB; row_end = hi; found = 1; break; }
      t -= tiles;
    }
    if (!found) return;
  } else if (p.mode == kContiguous) {
    row0 = (long long)mt_id * BMB;
    if (row0 >= p.M) return;
    e = gmeta[row0];
    if (e < 0 || e >= p.E) return;
    row_end = row0 + BMB < p.M ? row0 + BMB : p.M;
  } else if (p.mode == kMasked) {
    const int tpg = (int)((p.rows_per_group + BMB - 1) / BMB);
    e = mt_id / tpg;
    if (e >= p.E) return;
    const int mm = gmeta[e];
    const long long r = (long long)(mt_id % tpg) * BMB;
    if (r >= mm) return;
    row0 = (long long)e * p.rows_per_group + r;
    row_end = (long long)e * p.rows_per_group + mm;
  } else {
    row0 = (long long)mt_id * BMB;
    if (row0 >= p.M) return;
    row_end = p.M;
  }
  const int n0 = nt_id * BNB;
  const int KB = p.K / BK;

  // ---- LDS-DMA sources: uniform bases + 32-bit per-lane offsets (k-invariant part) ----
  const uint8_t* baseW = gW + ((long long)e * p.N + n0) * p.K;
  const uint8_t* baseA = gA + row0 * p.K;
  unsigned woff[4], aoff[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) {   // W piece (wave*4 + k): rows 8*(wave*4+k) + (lane>>3); LDS chunk lane&7 <- source chunk ^ swizzle
    const int r = (wave * 4 + k) * 8 + (lane >> 3);
    const int rr = n0 + r < p.N ? r : p.N - 1 - n0;   // rows beyond N: clamped (results discarded)
    woff[k] = (unsigned)rr * (unsigned)p.K + ((((lane & 7) ^ ((r >> 1) & 7))) << 4);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int r = (wave * 2 + k) * 8 + (lane >> 3);
    long long m = row0 + r;
    m = m < row_end ? m : row_end - 1;
    aoff[k] = (unsigned)(m - row0) * (unsigned)p.K + ((((lane & 7) ^ ((r >> 1) & 7))) << 4);
  }
  const float* baseAs;
  unsigned asoff;
  {
    long long m = row0 + (wave & 1) * 64 + lane;
    m = m < row_end ? m : row_end - 1;
    if (p.mode == kMasked) {
      baseAs = gAs + (long long)e * p.as_stride_g;
      asoff = (unsigned)((m - (long long)e * p.rows_per_group) * p.as_stride_m);
    } else {
      baseAs = gAs + row0 * p.as_stride_m;
      asoff = (unsigned)((m - row0) * p.as_stride_m);
    }
  }
  // operand read offsets inside a 32-row x 128 B sub-tile: row li, 16-B chunk c = 4*(s>>1) + 2*lh + (s&1)
  int rb[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int c = 4 * (s >> 1) + 2 * lh + (s & 1);
    rb[s] = li * BK + ((c ^ ((li >> 1) & 7)) << 4);
  }

  v16f acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto stage_w = [&](int st) { return smem + (st % kStages) * kStageBytes; };
  auto stage_a = [&](int st) { return smem + (st % kStages) * kStageBytes + kWB; };
  auto stage_as = [&](int st) { return reinterpret_cast<float*>(smem + (st % kStages) * kStageBytes + kWB + kAB); };
  auto issue_stage = [&](int kb) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(baseW + (long long)kb * BK + woff[k]),
                                       (lds_ptr_t)(stage_w(kb) + (wave * 4 + k) * 1024), 16, 0, 0);
#pragma unroll
    for (int k = 0; k < 2; ++k)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(baseA + (long long)kb * BK + aoff[k]),
                                       (lds_ptr_t)(stage_a(kb) + (wave * 2 + k) * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(baseAs + (long long)kb * p.as_stride_k + asoff),
                                     (lds_ptr_t)(stage_as(kb) + (wave & 1) * 64), 4, 0, 0);
  };

  // weight scales of this wave's 128-row block (uniform): Ws[e][nb][kb]
  int nb = (n0 + 64 * wn) / 128;
  const int nbs = (p.N + 127) / 128;
  nb = nb < nbs ? nb : nbs - 1;
  const float* wsrow = gWs + ((long long)e * nbs + nb) * KB;

  // ---- prologue: stages 0 and 1; iteration kb issues stage kb+2 into the slot of stage kb-1 ----
  issue_stage(0);
  if (KB > 1) issue_stage(1);
  for (int kb = 0; kb < KB; ++kb) {
    kblock(acc, wsrow[kb], stage_w(kb), stage_a(kb), stage_as(kb), stage_w(kb + 2), stage_a(kb + 2), stage_as(kb + 2),
           baseW + (long long)(kb + 2) * BK, baseA + (long long)(kb + 2) * BK, baseAs + (long long)(kb + 2) * p.as_stride_k,
           woff, aoff, asoff, kb + 2 < KB, kb + 1 < KB, rb, wave, wn, wt, li);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: D^T[n, m] -> out[m, n] bf16; lane (token li, half lh) holds n = 8g + 4lh + (0..3) of each tile ----
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const long long m = row0 + 64 * wt + 32 * tt + li;
    if (m < row_end) {
      uint16_t* orow = p.out + m * p.N;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + 64 * wn + 32 * nt + 8 * g + 4 * lh;
          if (n + 3 < p.N) {
            const uint32_t lo = (uint32_t)fl_f32_to_bf16(acc[nt][tt][4 * g + 0]) | ((uint32_t)fl_f32_to_bf16(acc[nt][tt][4 * g + 1]) << 16);
            const uint32_t hi = (uint32_t)fl_f32_to_bf16(acc[nt][tt][4 * g + 2]) | ((uint32_t)fl_f32_to_bf16(acc[nt][tt][4 * g + 3]) << 16);
            *reinterpret_cast<uint2*>(orow + n) = make_uint2(lo, hi);
          } else {
#pragma unroll
            for (int x = 0; x < 4; ++x)
              if (n + x < p.N) orow[n + x] = fl_f32_to_bf16(acc[nt][tt][4 * g + x]);
          }
        }
    }
  }
}

}  // namespace

// called from fl_grouped_gemm_fp8 (grouped_gemm_fp8.hip) when the expected rows per group reach the 128-token tile
int fl_grouped_gemm_fp8_big(const FlGemmArgs* a, hipStream_t stream) {
  BigParams p;
  p.mode = a->mode; p.E = a->num_groups; p.M = (int)a->M; p.N = a->N; p.K = a->K;
  p.as_stride_m = a->as_stride_m; p.as_stride_k = a->as_stride_k; p.as_stride_g = a->as_stride_g;
  p.rows_per_group = a->rows_per_group;
  p.out = (uint16_t*)a->out;
  p.n_tiles = (a->N + BNB - 1) / BNB;
  long long m_tiles;
  if (a->mode == kOffset) m_tiles = (a->M + BMB - 1) / BMB + a->num_groups;
  else if (a->mode == kMasked) m_tiles = (long long)a->num_groups * ((a->rows_per_group + BMB - 1) / BMB);
  else m_tiles = (a->M + BMB - 1) / BMB;
  const long long blocks = m_tiles * p.n_tiles;
  FL_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "fl_grouped_gemm_fp8: grid too large");
  FL_CHECK_ARG((long long)BNB * a->K < (1ll << 32) && (long long)BMB * a->K < (1ll << 32), "fl_grouped_gemm_fp8: K too large");
  grouped_gemm_fp8_big_kernel<<<dim3((unsigned)blocks), dim3(512), 0, stream>>>(p, (const uint8_t*)a->A, a->As,
                                                                               (const uint8_t*)a->W, a->Ws, a->group_meta);
  FL_CHECK_LAUNCH("grouped_gemm_fp8_big_kernel");
  return FL_OK;
}
