// K2-bf16 — paged MLA decode over ONE bf16 [pages, 64, 1, 576] cache tensor, gfx950 (MI355X) only.
//
// Replaces flash_mla_swap / flash_mla_fp8 .flash_mla_with_kvcache for cache_dtype = bf16 (call sites
// /root/reference/python/sglang/srt/layers/attention/flashmla_backend.py:163-175 verify/draft-extend, :240-254 decode):
//   s[r,t] = (q[r,:576] · k[t,:576]) * softmax_scale,  o[r,:] = sum_t softmax_t(s[r,:]) * k[t,:512]
// with q, k, P in bf16 and fp32 accumulation (v_mfma_f32_32x32x16_bf16).  This is the coverage kernel of the non-FP8
// KV dtype — same C-ABI, scheduler metadata and split-KV combine as the FP8 kernels; tuned far less (north_star is FP8).
// Measured limits (round 2): 669 us at bs=128, seq=4096, H=128 (12 % of 8 TB/s; 37 % at H=16).  A tile costs ~5,200 cycles for
// ~600 cycles of MFMA: every wave repeats the whole softmax of the tile (4x redundant) behind an LDS exchange of the partial
// S^T and two barriers.  Two 32-row tiles per workgroup sharing the operand reads (tried: 744 us) do not help — the per-tile
// cost doubles with the rows; the role-specialised structure of mla_decode_fp8_y.hip is what this kernel would need.
//
// Mapping (layouts probed in probes/probe_bf16.hip):
//   * workgroup = 4 waves = one 32-row group of one request part; tile = 32 tokens (half a page, 36 KiB of bf16).
//   * "SwapAB" as in the FP8 kernels: S^T[32 tok x 32 rows] = K · Q^T, one query row per lane.  The 36 k-steps of the
//     576-wide contraction are SPLIT over the 4 waves (9 each, Q fragment = 36 VGPRs per wave); the four partial S^T
//     are summed through LDS (16 KiB), after which every wave holds the full S^T, runs the same online softmax and
//     owns P in the register layout the PV MFMA wants as its B operand (contraction order over tokens is free).
//   * wave w accumulates O^T for the dims [128w, 128w+128): 4 tiles x 2 k-steps per 32 tokens.  V^T operands come from
//     the same LDS bytes as K through ds_read_b64_tr_b16 (hardware 16-bit transpose).
//   * HBM -> LDS by global_load_lds (1 KiB per wave instruction, 9 per wave per tile), 3-slot ring, 16-B chunks
//     XOR-swizzled on the source address (chunk c of token T at (c & ~7) | ((c & 7) ^ ((T >> 1) & 7))).
#include "mla_decode_shared.h"

using namespace fl_mla;

typedef short v4s __attribute__((ext_vector_type(4)));

namespace {

constexpr int kTile = 32;                       // tokens per tile
constexpr int kRowB = (kDN + kDR) * 2;          // 1152 B per token
constexpr int kTileBytes = kTile * kRowB;       // 36 KiB
constexpr int kSlots = 3;
constexpr int kPiecesPerWave = kTileBytes / 1024 / 4;   // 9
constexpr int kOffPart = kSlots * kTileBytes;   // [wave 4][64 lanes][16 f32] partial S^T
constexpr int kLds = kOffPart + 4 * 64 * 16 * 4;
static_assert(kLds <= 160 * 1024, "LDS budget");
constexpr float kNegInit = -1.0e30f;

__device__ __forceinline__ int swz(const int c, const int T) { return (c & ~7) | ((c & 7) ^ ((T >> 1) & 7)); }

template <int DUMMY>
__device__ __forceinline__ void tile_body(
    v16f (&o)[4], float& m_run, float& l_run, const uint4 (&qf)[9], const unsigned (&dsrc)[kPiecesPerWave],
    const uint8_t* __restrict__ rd, float* __restrict__ part, uint8_t* __restrict__ dma_dst,
    const uint8_t* __restrict__ dma_src, const int wave, const int lane, const int tok0, const int L, const int L_row,
    const float scale_log2e, const bool next_in_flight) {
  const int li = lane & 31, lh = lane >> 5;
  // ---- tile landed for every wave; every wave is done with the previous tile (its slot is refilled below) ----
  if (next_in_flight)
    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (dma_src != nullptr) {
#pragma unroll
    for (int k = 0; k < kPiecesPerWave; ++k)
      fl_dma_lds((gbl_ptr_t)(dma_src + dsrc[k]),
                                       (lds_ptr_t)(dma_dst + (wave * kPiecesPerWave + k) * 1024), 16, 0, 0);
  }
  // tail of the sequence: zero the rows past the end (P is exactly 0 there, but 0*NaN would poison the PV MFMA)
  if (tok0 + kTile > L) {
    const int nvalid = L - tok0 > 0 ? L - tok0 : 0;
    uint8_t* wr = const_cast<uint8_t*>(rd);
    for (int T = nvalid; T < kTile; ++T)
      for (int c = lane; c < kRowB / 16; c += 64) *reinterpret_cast<uint4*>(wr + T * kRowB + c * 16) = make_uint4(0, 0, 0, 0);
  }
  // ---- partial S^T over this wave's 9 k-steps: A = K (token li, 8 dims at 16 ks + 8 lh), B = Q fragment ----
  v16f acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int s = 0; s < 9; ++s) {
    const int c = 2 * (9 * wave + s) + lh;
    const uint4 ka = *reinterpret_cast<const uint4*>(rd + li * kRowB + 16 * swz(c, li));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(ka), as_bf8(qf[s]), acc, 0, 0, 0);
  }
  // ---- sum the four partials through LDS: afterwards every wave holds the full S^T ----
  float* mine = part + (wave * 64 + lane) * 16;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(mine + 4 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // (fp32 addition is not associative: ONE summation order, waves 0..3, so that all four waves hold identical S^T)
  {
    v16f s4;
#pragma unroll
    for (int r = 0; r < 16; ++r) s4[r] = 0.f;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const float* src = part + (x * 64 + lane) * 16;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(src + 4 * g);
        s4[4 * g] += v.x; s4[4 * g + 1] += v.y; s4[4 * g + 2] += v.z; s4[4 * g + 3] += v.w;
      }
    }
    acc = s4;
  }
  // ---- online softmax in the log2 domain; lane (row li, half lh) holds tokens T(r) = (r&3) + 8(r>>2) + 4lh ----
  float tmax = kNegInit;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int T = (r & 3) + 8 * (r >> 2) + 4 * lh;
    float y = acc[r] * scale_log2e;
    if (tok0 + T >= L_row || !(y == y)) y = -INFINITY;
    acc[r] = y;
    tmax = fmaxf(tmax, y);
  }
  tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
  const float m_new = fmaxf(m_run, tmax);
  const float f = __builtin_amdgcn_exp2f(m_run - m_new);   // 1 when unchanged
  if (__any(m_new > m_run)) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[j][r] *= f;
  }
  l_run *= f;
  m_run = m_new;
  float psum = 0.f;
  uint32_t pk[8];
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const float p0 = __builtin_amdgcn_exp2f(acc[r] - m_new), p1 = __builtin_amdgcn_exp2f(acc[r + 1] - m_new);
    psum += p0 + p1;
    pk[r >> 1] = (uint32_t)fl_f32_to_bf16(p0) | ((uint32_t)fl_f32_to_bf16(p1) << 16);
  }
  psum += __shfl_xor(psum, 32);
  l_run += psum;
  // ---- O^T[128w + 32mt + .., 32 rows] += V^T · P^T: k-step s covers B elements r = 8s..8s+7 of every lane half,
  //      i.e. tokens (j&3) + 8(2s + (j>>2)) + 4h; the tr16 read u fetches j = 4u..4u+3 (source lane s16 -> token
  //      (s16>>2) + 8(2s+u) + 4h, dims 16((lane>>4)&1) + 4(s16&3) .. +3) ----
  const int s16 = lane & 15;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      union { v8bf v; v4s h[2]; } a;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int Tt = (s16 >> 2) + 8 * (2 * s + u) + 4 * lh;
        const int d = 128 * wave + 32 * mt + 16 * ((lane >> 4) & 1) + 4 * (s16 & 3);
        const uint8_t* ap = rd + Tt * kRowB + 16 * swz(d >> 3, Tt) + (d & 7) * 2;
        a.h[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)ap);
      }
      const uint4 pb = make_uint4(pk[4 * s], pk[4 * s + 1], pk[4 * s + 2], pk[4 * s + 3]);
      o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, as_bf8(pb), o[mt], 0, 0, 0);
    }
  }
}

__global__ __launch_bounds__(256, 1) void mla_decode_bf16_kernel(
    const Params p, const int32_t* __restrict__ g_block_table, const int32_t* __restrict__ g_seqlens,
    const int32_t* __restrict__ g_meta, const int32_t* __restrict__ g_num_splits, const uint8_t* __restrict__ g_k,
    const uint8_t* __restrict__ g_q) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[kLds];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
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
  const int row = rgrp * 32 + li;
  const bool row_ok = row < p.rows;

  // LDS-DMA source offsets of this wave's 9 pieces inside a 32-token tile (destination is lane-linear)
  unsigned dsrc[kPiecesPerWave];
#pragma unroll
  for (int k = 0; k < kPiecesPerWave; ++k) {
    const int off = (wave * kPiecesPerWave + k) * 1024 + lane * 16;
    const int T = off / kRowB;
    const int cp = (off - T * kRowB) >> 4;
    dsrc[k] = (unsigned)(T * kRowB + 16 * swz(cp, T));
  }
  float* part_buf = reinterpret_cast<float*>(smem + kOffPart);

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

    // Q fragment of this wave: dims [144 wave, 144 wave + 144)
    const long long qrow = (long long)req * p.rows + row;
    uint4 qf[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) qf[s] = make_uint4(0, 0, 0, 0);
    if (row_ok) {
      const uint8_t* qp = g_q + qrow * kRowB + (9 * wave) * 32 + lh * 16;
#pragma unroll
      for (int s = 0; s < 9; ++s) qf[s] = *reinterpret_cast<const uint4*>(qp + s * 32);
    }
    int L_row = L;
    if (p.causal) L_row = L - (p.s_q - 1 - row / p.h_q);
    if (!row_ok) L_row = 0;

    v16f o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
    float m_run = kNegInit, l_run = 0.f;

    auto tile_src = [&](int t) {   // global address of 32-token tile t0 + t
      const int tt = t0 + t;
      int pg = g_block_table[(long long)req * p.bt_stride + (tt >> 1)];
      pg = (pg < 0 || pg >= p.num_pages) ? 0 : pg;
      return g_k + ((long long)pg * kPage + (tt & 1) * kTile) * kRowB;
    };
    auto slot = [&](int t) { return smem + (t % kSlots) * kTileBytes; };

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // previous request's LDS reads are done; Q loads left the vmcnt queue
    for (int t = 0; t < 2 && t < n; ++t) {
      const uint8_t* src = tile_src(t);
#pragma unroll
      for (int k = 0; k < kPiecesPerWave; ++k)
        fl_dma_lds((gbl_ptr_t)(src + dsrc[k]),
                                         (lds_ptr_t)(slot(t) + (wave * kPiecesPerWave + k) * 1024), 16, 0, 0);
    }
    for (int i = 0; i < n; ++i) {
      const uint8_t* src = i + 2 < n ? tile_src(i + 2) : nullptr;
      tile_body<0>(o, m_run, l_run, qf, dsrc, slot(i), part_buf, slot(i + 2), src, wave, lane, (t0 + i) * kTile, L, L_row,
                   p.scale_log2e, i + 1 < n);
    }

    // ---- epilogue: every wave writes its 128 dims of the 32 rows ----
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    const float lse_nat = l_run > 0.f ? (__builtin_amdgcn_logf(l_run) + m_run) * 0.6931471805599453f : -INFINITY;
    if (row_ok) {
      const int slot_idx = split_base + split_idx;
      if (lh == 0 && wave == 0) {
        if (is_split) {
          p.lse_accum[((long long)slot_idx * p.rows + row) * 2 + 0] = lse_nat;
          p.lse_accum[((long long)slot_idx * p.rows + row) * 2 + 1] = lse_nat;
        } else {
          const int j = row / p.h_q, h = row - j * p.h_q;
          p.lse[((long long)req * p.h_q + h) * p.s_q + j] = lse_nat;
        }
      }
      // C row i = (r&3) + 8(r>>2) + 4lh of tile mt -> d = 128 wave + 32 mt + i
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d0 = 128 * wave + 32 * mt + 8 * g + 4 * lh;
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
  }
}

}  // namespace

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
  p.row_groups = (p.rows + 31) / 32;
  const dim3 grid((unsigned)(p.num_parts * p.row_groups)), block(256);
  mla_decode_bf16_kernel<<<grid, block, 0, stream>>>(p, a->block_table, a->cache_seqlens, a->tile_scheduler_metadata,
                                                      a->num_splits, (const uint8_t*)a->k_nope, (const uint8_t*)a->q_nope);
  FL_CHECK_LAUNCH("mla_decode_bf16_kernel");
  return fl_mla_launch_combine(p, a->num_splits, stream);
}
