// K1 — paged MLA decode over the per-token-FP8 latent KV cache, gfx950 (MI355X) only.
//
// Replaces flash_mla_fp8.flash_mla_ckv_fp8_per_token (call sites
// /root/reference/python/sglang/srt/layers/attention/flashmla_backend.py:208-222 decode, :127-142 verify/draft-extend).
//
// Math (per request b, query row r = j*h_q + h, latent token t):
//   s[r,t]  = (q8[r,:]·k8[t,:] + qrope'[r,:]·krope'[t,:]) * q_scale[r] * k_scale[t] * softmax_scale
//             (rope is stored pre-divided by the scale on both sides: memory_pool.py:877)
//   o[r,:]  = sum_t softmax_t(s[r,:]) * k_scale[t] * k8[t,:512]          (V = dequantised latent)
//
// MI355X mapping ("SwapAB": tokens on the MFMA M side, query rows on the N side):
//   * one workgroup = 2*NRG waves (one per SIMD), NRG row groups of 32 query rows; every wave of the workgroup works
//     on the SAME 64-token page.  Wave (rg, W) computes S^T[32 tok x 32 rows] = K[32W..32W+31] · Q_rg^T on
//     v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3, the 2x-rate MX path; + 32x32x16 bf16 for the 64 rope dims): each
//     lane then holds ONE query row (lane&31), so the online-softmax reductions are in-register plus one cross-half
//     exchange.  The same wave owns the d-half [256W, 256W+256) of O^T = V^T · P^T (8 accumulator tiles, 128 VGPRs).
//   * the two waves of a row group keep INDEPENDENT integer softmax references m_W (ceil of the running max of
//     y = s*log2e + log2 k_scale[t], which also folds the per-token V scale into P).  P' = 2^(y - m_W + 8) is
//     re-quantised to fp8 and exchanged through 1 KiB of LDS; the PV MFMA reconciles the two references for free with
//     its E8M0 block scales 2^(m_W - M) (M = reference of the O accumulator), so no cross-wave max exchange and
//     exactly ONE s_barrier per page.  Two fp32 normalisers per wave, merged once per request: the ROUNDED P'/k_scale
//     (divides O: quantised weights sum to exactly 1) and the unrounded one (exact LSE).
//     (Measured on gfx950: MX block b of the 32x32x64 B operand = bytes [16b,16b+16) of BOTH lane halves, its scale
//     comes from lane n+32b — so every lane keeps its own 16 P bytes in registers as block W and only fetches the
//     partner lane's 16 bytes.)
//   * V^T is read from the SAME LDS bytes as K through ds_read_b64_tr_b8 (hardware byte transpose), in the token
//     order in which P sits in the B operand (the contraction order over tokens is free).
//   * HBM -> LDS exclusively by global_load_lds (LDS-DMA, 16 B/lane, 1 KiB per wave instruction): a 4-slot ring of
//     32 KiB latent pages + 2-slot rings for rope (8 KiB) and raw scales, XOR-swizzled on the SOURCE address so that
//     ds_read_b128 (K operand, rope) and ds_read_b64_tr_b8 (V^T) are bank-conflict free.  Waits are counted
//     (s_waitcnt vmcnt(8) + raw s_barrier): two pages stay in flight across every barrier.  The per-page body is one
//     inlined function whose LDS regions are distinct __restrict__ parameters — otherwise hipcc's waitcnt pass
//     assumes every ds_read may alias the in-flight LDS-DMA and drains it with vmcnt(0).
//   * rows past the sequence end are zero-filled in LDS by the consumers (P' is exactly 0 there, but 0*NaN from stale
//     fp8 NaN patterns would poison the PV MFMA).
//
// Algorithmic bytes per (request, layer call): seq*644 (KV) + s_q*h_q*(644 + 1024) (Q in, O out) + 4*ceil(seq/64).
#include "mla_decode_shared.h"
#include <cstdlib>

using namespace fl_mla;

namespace {

// ---- LDS map (one __shared__ array) ----
constexpr int kOffRing = 0;                                   // 4 x 32 KiB
constexpr int kOffRope = kOffRing + kRingSlots * kSlotBytes;  // 2 x 8 KiB
constexpr int kOffScale = kOffRope + 2 * kRopeBytes;          // 2 x 64 f32 raw k_scale
constexpr int kOffScratch = kOffScale + 2 * kPage * 4;        // 4 waves x {ks, log2 ks, 1/ks} x 32 tokens
constexpr int kScratchPerWave = 3 * 32 * 4;
constexpr int kOffPbuf = kOffScratch + 4 * kScratchPerWave;   // [parity 2][rg 2][W 2][64 lanes][16 B]
constexpr int kPbufPerParity = 2 * 2 * 32 * 32;
constexpr int kOffRef = kOffPbuf + 2 * kPbufPerParity;        // [parity 2][rg 2][W 2][32 rows] f32
constexpr int kRefPerParity = 2 * 2 * 32 * 4;
constexpr int kLdsBytes = kOffRef + 2 * kRefPerParity;
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");

#if defined(FL_MLA_DEBUG) || defined(FL_MLA_TIMING)
__device__ int* g_dbg = nullptr;   // debug builds only: set by fl_mla_debug_set_buffer
#endif
#ifdef FL_MLA_TIMING
#define FL_T(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); tacc[i] += t__ - tlast; tlast = t__; } while (0)
#else
#define FL_T(i) do { } while (0)
#endif

// Per-lane constants of the LDS access patterns (computed once per kernel).
struct LaneConst {
  int lane, li, lh;
  int kb[2][4];    // K operand: byte offset inside a slot, [e][s&3]; + (s>>2)*256 immediate; + W*16384
  int rb[4];       // rope operand: byte offset inside a rope slot, [s]
  int vb[8];       // V^T tr8 source: byte offset inside a slot, [(jb&3) | (jb>>2)<<2]; + u immediates
  unsigned dn[16]; // latent DMA: byte offset of this lane's 16 B inside the page, per piece of this wave
  unsigned dr[4];  // rope DMA: element offset inside the page's rope block, per piece of this wave
};

struct ReqState {
  v16f o[8];
  float l_run, lq_run, m_w, m_o;   // own exact / rounded-P normalisers, own integer reference, reference of O
};

// Operands of a page's QK that are fetched one page early (in the shadow of the previous page's PV MFMAs).
struct QkPrefetch {
  uint4 ra[4];       // rope A operand (FMT 0: 4 bf16 k-steps; FMT 1: 2 x 16 B of the fp8 k-step)
  uint4 ka[4][2];    // latent k-steps 0..3
};

// Per-token scale triples {k_scale, log2 k_scale, 1/k_scale} of this wave's 32 tokens -> wave-private scratch.
template <int FMT>
__device__ __forceinline__ void scale_prep(const float* __restrict__ rd_scale, float* __restrict__ scratch,
                                           const float ks_const, const int W, const int li, const int lh, const int tok0,
                                           const int L) {
  float ks = FMT == 0 ? rd_scale[32 * W + li] : ks_const;
  if (tok0 + 32 * W + li >= L || !(ks > 0.f) || !(ks < 3.0e38f)) ks = 1.f;
  if (lh == 0) {
    scratch[li] = ks;
    scratch[32 + li] = __builtin_amdgcn_logf(ks);
    scratch[64 + li] = __builtin_amdgcn_rcpf(ks);
  }
}

template <int FMT>
__device__ __forceinline__ void qk_prefetch(QkPrefetch& pre, const LaneConst& lc, const uint8_t* __restrict__ rd_nope,
                                            const uint8_t* __restrict__ rd_rope, const int W) {
  const uint8_t* rp = rd_rope + W * (32 * (FMT == 0 ? kDR * 2 : kDR));
#pragma unroll
  for (int s = 0; s < (FMT == 0 ? 4 : 2); ++s) pre.ra[s] = *reinterpret_cast<const uint4*>(rp + lc.rb[s]);
  const uint8_t* kp = rd_nope + W * (32 * kDN);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    pre.ka[s][0] = *reinterpret_cast<const uint4*>(kp + lc.kb[0][s & 3]);
    pre.ka[s][1] = *reinterpret_cast<const uint4*>(kp + lc.kb[1][s & 3]);
  }
}

// One page of one request for one wave.  All LDS regions are distinct __restrict__ parameters (see file header).
// FMT 0: per-token FP8 (u8 [.,512] + f32 scale + bf16 rope [.,64]); FMT 1: one fp8 [.,576] tensor, scalar descales
// (flashmla_backend.py:227-239): same pipeline, the 64 rope dims are a 9th fp8 k-step and the scales are constants.
template <int NRG, int FMT>
__device__ __forceinline__ void tile_body(
    ReqState& st, QkPrefetch& pre, const LaneConst& lc, const v8i (&qn)[8], const v8bf (&qr)[4], const v8i qr8,
    const float qs, const float ks_const, const int W, const int rg, const int wave,
    // ---- LDS regions of the NEXT page (landed: certified by this call's barrier), read in the PV shadow
    const uint8_t* __restrict__ nx_nope, const uint8_t* __restrict__ nx_rope, const float* __restrict__ nx_scale,
    const bool has_next,
    // ---- LDS regions consumed now
    const uint8_t* __restrict__ rd_nope, const uint8_t* __restrict__ rd_rope, const float* __restrict__ rd_scale,
    float* __restrict__ scratch, uint8_t* __restrict__ pbuf, float* __restrict__ refbuf,
    // ---- LDS regions filled by the DMA issued in this call (never read in this call)
    uint8_t* __restrict__ dma_nope, uint8_t* __restrict__ dma_rope, float* __restrict__ dma_scale,
    // ---- DMA sources (global; null = nothing to issue)
    const uint8_t* __restrict__ src_nope, const uint16_t* __restrict__ src_rope, const float* __restrict__ src_scale,
    // ---- page geometry
    const int tok0, const int L, const int L_row, const int L_min, const bool more_in_flight
#ifdef FL_MLA_TIMING
    , unsigned long long (&tacc)[8], unsigned long long& tlast
#endif
    ) {
  constexpr int NW = 2 * NRG;
  constexpr int kNopePerWave = kDmaNopePerTile / NW;
  constexpr int kRopePerWave = (FMT == 0 ? 8 : 4) / NW;
  const int lane = lc.lane, li = lc.li, lh = lc.lh;
#ifdef FL_EXP_NOCOMPUTE   // experiment: DMA pipeline only (streaming ceiling of this structure)
  if (more_in_flight)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (src_rope != nullptr) {
    for (int k = 0; k < kRopePerWave; ++k)
      fl_dma_lds((gbl_ptr_t)(reinterpret_cast<const uint8_t*>(src_rope) + lc.dr[k]),
                                       (lds_ptr_t)(dma_rope + (wave * kRopePerWave + k) * 1024), 16, 0, 0);
    fl_dma_lds((gbl_ptr_t)(src_scale + lane), (lds_ptr_t)dma_scale, 4, 0, 0);
  }
  if (src_nope != nullptr) {
    for (int k = 0; k < kNopePerWave; ++k)
      fl_dma_lds((gbl_ptr_t)(src_nope + lc.dn[k]),
                                       (lds_ptr_t)(dma_nope + (wave * kNopePerWave + k) * 1024), 16, 0, 0);
  }
  return;
#endif

  // ---- tail of the sequence: zero the rows past the end (every wave, all rows it may read) ----
  if (tok0 + kPage > L) {
    const int nvalid = L - tok0;
    uint8_t* wr = const_cast<uint8_t*>(rd_nope);
#pragma clang loop vectorize(disable) unroll(disable)
    for (int T = nvalid + lh; T < kPage; T += 2)
      *reinterpret_cast<uint4*>(wr + T * kDN + li * 16) = make_uint4(0, 0, 0, 0);
  }

  FL_T(0);   // prep (tail fill, scale scratch)
  // ---- A. S^T[32 tok x 32 rows] = K[32W + ..] · Q^T : all 20 operand reads in flight, then 12 back-to-back MFMAs ----
  v16f acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // V^T operands of the first PV tiles: issued inside the QK MFMA chain below
  const uint8_t* vp = rd_nope + W * 256;   // d half -> 16 chunks of 16 B further along every token row
  v8i va[8];
  auto load_vt = [&](int jb) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint8_t* ap = vp + lc.vb[(jb & 3) | ((jb >> 2) << 2)] + (u & 1) * (16 * kDN) + (u >> 1) * (32 * kDN);
      const v2i t2 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)ap);
      va[jb][2 * u] = t2[0];
      va[jb][2 * u + 1] = t2[1];
    }
  };
  __builtin_amdgcn_sched_barrier(0);
  float4 ks4[4], lk4[4], ik4[4];
  {
    // rope + k-steps 0..3 were prefetched during the previous page's PV; fetch k-steps 4..7 now
    const uint8_t* kp = rd_nope + W * (32 * kDN);
    uint4 ka[8][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) { ka[s][0] = pre.ka[s][0]; ka[s][1] = pre.ka[s][1]; }
#pragma unroll
    for (int s = 4; s < 8; ++s) {
      ka[s][0] = *reinterpret_cast<const uint4*>(kp + lc.kb[0][s & 3] + 256);
      ka[s][1] = *reinterpret_cast<const uint4*>(kp + lc.kb[1][s & 3] + 256);
    }
    uint4 ra[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) ra[s] = pre.ra[s];
    // per-token scale triples of this lane's 16 tokens (written above by this wave): queued behind the operand
    // reads, they land while the MFMAs run
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int tb = g * 8 + lh * 4;
      ks4[g] = *reinterpret_cast<const float4*>(scratch + tb);
      lk4[g] = *reinterpret_cast<const float4*>(scratch + 32 + tb);
      ik4[g] = *reinterpret_cast<const float4*>(scratch + 64 + tb);
    }
    if constexpr (FMT == 0) {
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(ra[s]), qr[s], acc, 0, 0, 0);
    } else {
      acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(make_v8i(ra[0], ra[1]), qr8, acc, 0, 0, 0, kUnitScale, 0,
                                                            kUnitScale);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s)
      acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(make_v8i(ka[s][0], ka[s][1]), qn[s], acc, 0, 0, 0,
                                                            kUnitScale, 0, kUnitScale);
    // V^T operands of the first three PV tiles ride in the shadow of the MFMA chain (otherwise empty)
    load_vt(0);
    load_vt(1);
    load_vt(2);
    __builtin_amdgcn_sched_group_barrier(0x100, 20, 0);                   // DS reads: 8 operand + 12 scale
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                    // MFMA
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                    // tr8 reads of PV tile 0
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                    // PV tile 1
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                    // PV tile 2
    __builtin_amdgcn_sched_group_barrier(0x008, FMT == 0 ? 5 : 2, 0);     // rest of the chain
  }
  __builtin_amdgcn_sched_barrier(0);

  FL_T(1);   // QK issue + V^T prefetch issue
  // ---- B. local online softmax on y = s*log2e + log2(k_scale[t]); tokens of lane: 32W + 8g + 4lh + e ----
  const bool need_mask = (tok0 + kPage > L_min);
  float tmax = -INFINITY;
  if (!need_mask) {
    // every token of the page is valid for every row of the wave: no selects
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      acc[g * 4 + 0] = fmaf(acc[g * 4 + 0] * qs, ks4[g].x, lk4[g].x);
      acc[g * 4 + 1] = fmaf(acc[g * 4 + 1] * qs, ks4[g].y, lk4[g].y);
      acc[g * 4 + 2] = fmaf(acc[g * 4 + 2] * qs, ks4[g].z, lk4[g].z);
      acc[g * 4 + 3] = fmaf(acc[g * 4 + 3] * qs, ks4[g].w, lk4[g].w);
      tmax = fmaxf(fmaxf(tmax, fmaxf(acc[g * 4 + 0], acc[g * 4 + 1])), fmaxf(acc[g * 4 + 2], acc[g * 4 + 3]));
    }
  } else {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int tb = g * 8 + lh * 4;
      const float ksv[4] = {ks4[g].x, ks4[g].y, ks4[g].z, ks4[g].w};
      const float lkv[4] = {lk4[g].x, lk4[g].y, lk4[g].z, lk4[g].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y = fmaf(acc[g * 4 + e] * qs, ksv[e], lkv[e]);
        if (tok0 + 32 * W + tb + e >= L_row) y = -INFINITY;
        if (!(y == y)) y = -INFINITY;   // NaN can only come from garbage beyond the row's limit
        acc[g * 4 + e] = y;
        tmax = fmaxf(tmax, y);
      }
    }
  }
  tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
  {
    const float m_new = tmax > st.m_w ? ceilf(tmax) + kRefHeadroom : st.m_w;
    const float f = __builtin_amdgcn_exp2f(st.m_w - m_new);   // exactly 1 when the reference did not move
    st.l_run *= f;
    st.lq_run *= f;
    st.m_w = m_new;
  }
  uint4 own_p;
  {
    const float moff = kPShift - st.m_w;
    int pk[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float e0 = __builtin_amdgcn_exp2f(acc[g * 4 + 0] + moff);
      const float e1 = __builtin_amdgcn_exp2f(acc[g * 4 + 1] + moff);
      const float e2 = __builtin_amdgcn_exp2f(acc[g * 4 + 2] + moff);
      const float e3 = __builtin_amdgcn_exp2f(acc[g * 4 + 3] + moff);
      st.l_run = fmaf(e0, ik4[g].x, st.l_run);   // unrounded sum: exact LSE (a rounded-sum LSE is off by up to 6 % on
      st.l_run = fmaf(e1, ik4[g].y, st.l_run);   // peaked rows)
      st.l_run = fmaf(e2, ik4[g].z, st.l_run);
      st.l_run = fmaf(e3, ik4[g].w, st.l_run);
      int v = __builtin_amdgcn_cvt_pk_fp8_f32(e0, e1, 0, false);
      pk[g] = __builtin_amdgcn_cvt_pk_fp8_f32(e2, e3, v, true);
      // the ROUNDED weights normalise O (numerator and denominator use the same weights: they sum to exactly 1)
      const float2v d01 = __builtin_amdgcn_cvt_pk_f32_fp8(pk[g], false);
      const float2v d23 = __builtin_amdgcn_cvt_pk_f32_fp8(pk[g], true);
      st.lq_run = fmaf(d01[0], ik4[g].x, st.lq_run);
      st.lq_run = fmaf(d01[1], ik4[g].y, st.lq_run);
      st.lq_run = fmaf(d23[0], ik4[g].z, st.lq_run);
      st.lq_run = fmaf(d23[1], ik4[g].w, st.lq_run);
    }
    // publish P (16 B) and the reference for the partner wave
    own_p = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    *reinterpret_cast<uint4*>(pbuf + (rg * 2 + W) * (64 * 16) + lane * 16) = own_p;
    if (lh == 0) refbuf[(rg * 2 + W) * 32 + li] = st.m_w;
  }

  FL_T(2);   // softmax + P publish
  // ---- C. page i+1 landed for every wave; P/refs visible ----
  if (more_in_flight) {
    if constexpr (NRG == 2)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // leave the latent pieces of page i+2 in flight
    else
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  FL_T(3);   // vmcnt + lgkmcnt + barrier
  // ---- E. B operand of the PV MFMA: bytes 0..15 = wave 0's P of this lane, 16..31 = wave 1's; both references ----
  const uint4 other_p = *reinterpret_cast<const uint4*>(pbuf + (rg * 2 + (1 - W)) * (64 * 16) + lane * 16);
  const float m0 = refbuf[(rg * 2 + 0) * 32 + li];
  const float m1 = refbuf[(rg * 2 + 1) * 32 + li];
  // next page's raw scale (landed) -> registers now, prepared in the PV shadow
  float ks_next = 1.f;
  if (has_next) ks_next = FMT == 0 ? nx_scale[32 * W + li] : ks_const;
  FL_T(4);
  const float mo_new = fmaxf(st.m_o, fmaxf(m0, m1));
  if (__any(mo_new > st.m_o)) {
    const float f = __builtin_amdgcn_exp2f(st.m_o - mo_new);   // exactly 1 where unchanged, 0 on the first page
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) st.o[j][r] *= f;
    st.m_o = mo_new;
  }
  int sb = 127 + (int)((lh ? m1 : m0) - st.m_o);
  sb = sb < 0 ? 0 : sb;
  const v8i pb = W == 0 ? make_v8i(own_p, other_p) : make_v8i(other_p, own_p);
  FL_T(5);   // O-reference update
  // ---- F. O^T[256W + .., 32 rows] += V^T · P^T.  In the shadow of the 8 MFMAs (64 cycles each): the V^T reads three
  //         tiles ahead, D. the LDS-DMA refill (rope/scale of page i+2 into the slots of page i, latent of page i+3 into
  //         the slot of page i-1), the next page's scale triples and the first half of its QK operands. ----
#ifdef FL_EXP_NODMA
  const bool do_rs = false, do_n = false;
#else
  const bool do_rs = src_rope != nullptr, do_n = src_nope != nullptr;
#endif
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    if (jb + 3 < 8) load_vt(jb + 3);
    st.o[jb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va[jb], pb, st.o[jb], 0, 0, 0, kUnitScale, 0, sb);
    if (jb == 0 && do_rs) {
#pragma unroll
      for (int k = 0; k < kRopePerWave; ++k)
        fl_dma_lds((gbl_ptr_t)(reinterpret_cast<const uint8_t*>(src_rope) + lc.dr[k]),
                                         (lds_ptr_t)(dma_rope + (wave * kRopePerWave + k) * 1024), 16, 0, 0);
      if constexpr (FMT == 0)
        fl_dma_lds((gbl_ptr_t)(src_scale + lane), (lds_ptr_t)dma_scale, 4, 0, 0);
    }
    if (do_n) {
      constexpr int kPer = (kNopePerWave + 7) / 8;   // latent pieces issued behind each PV MFMA
#pragma unroll
      for (int k = jb * kPer; k < (jb + 1) * kPer && k < kNopePerWave; ++k)
        fl_dma_lds((gbl_ptr_t)(src_nope + lc.dn[k]),
                                         (lds_ptr_t)(dma_nope + (wave * kNopePerWave + k) * 1024), 16, 0, 0);
    }
    if (jb == 4 && has_next) {
      // scale triples of the next page (this wave's scratch is free: its reads for this page completed before softmax)
      float ks = ks_next;
      if (tok0 + kPage + 32 * W + li >= L || !(ks > 0.f) || !(ks < 3.0e38f)) ks = 1.f;
      if (lh == 0) {
        scratch[li] = ks;
        scratch[32 + li] = __builtin_amdgcn_logf(ks);
        scratch[64 + li] = __builtin_amdgcn_rcpf(ks);
      }
    }
    if (jb == 5 && has_next) qk_prefetch<FMT>(pre, lc, nx_nope, nx_rope, W);
  }
  FL_T(6);   // PV issue
}

// Read-only inputs are separate `const __restrict__` kernel arguments so that hipcc proves them invariant: wave-
// uniform reads (page ids, lengths, scheduler rows) become s_load (lgkmcnt), never vector loads on the vmcnt queue.
// One page for a LOADER wave (NRG = 1): the barrier protocol of tile_body's section C, then the refill it certifies.
template <int NRG, int FMT>
__device__ __forceinline__ void loader_page(const LaneConst& lc, const int wave, uint8_t* __restrict__ dma_nope,
                                            uint8_t* __restrict__ dma_rope, float* __restrict__ dma_scale,
                                            const uint8_t* __restrict__ src_nope, const uint16_t* __restrict__ src_rope,
                                            const float* __restrict__ src_scale, const bool more_in_flight) {
  constexpr int NW = 2 * NRG;
  constexpr int kNopePerWave = kDmaNopePerTile / NW;
  constexpr int kRopePerWave = (FMT == 0 ? 8 : 4) / NW;
  if (more_in_flight) {
    if constexpr (NRG == 2)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (src_rope != nullptr) {
#pragma unroll
    for (int k = 0; k < kRopePerWave; ++k)
      fl_dma_lds((gbl_ptr_t)(reinterpret_cast<const uint8_t*>(src_rope) + lc.dr[k]),
                 (lds_ptr_t)(dma_rope + (wave * kRopePerWave + k) * 1024), 16, 0, 0);
    if constexpr (FMT == 0) fl_dma_lds((gbl_ptr_t)(src_scale + lc.lane), (lds_ptr_t)dma_scale, 4, 0, 0);
  }
  if (src_nope != nullptr) {
#pragma unroll
    for (int k = 0; k < kNopePerWave; ++k)
      fl_dma_lds((gbl_ptr_t)(src_nope + lc.dn[k]), (lds_ptr_t)(dma_nope + (wave * kNopePerWave + k) * 1024), 16, 0, 0);
  }
}

// NRG = 1 (at most 32 query rows, e.g. the TP8 shard H = 16): two LOADER waves join the two compute waves (the CU's
// other two SIMDs are idle otherwise).  They only issue the LDS-DMA refills (21 pieces per page and wave, 60-185 issue
// cycles each — in the compute waves that was most of the PV stage) and keep the barrier protocol.
constexpr int loader_waves(const int nrg) { return nrg == 1 ? 2 : 0; }

template <int NRG, int FMT>
__global__ __launch_bounds__(64 * (2 * NRG + loader_waves(NRG)), 1) void mla_decode_fp8_kernel(
    const Params p, const int32_t* __restrict__ g_block_table, const int32_t* __restrict__ g_seqlens,
    const int32_t* __restrict__ g_meta, const int32_t* __restrict__ g_num_splits,
    const uint8_t* __restrict__ g_k_nope, const uint16_t* __restrict__ g_k_rope, const float* __restrict__ g_k_scale,
    const uint8_t* __restrict__ g_q_nope, const uint16_t* __restrict__ g_q_rope, const float* __restrict__ g_q_scale) {
  constexpr int NW = 2 * NRG;
  constexpr int LD = loader_waves(NRG);
  static_assert(LD == 0 || LD == NW, "the loaders take over the compute waves' piece mapping one to one");
  constexpr int kTokBytes = FMT == 0 ? kDN : kDN + kDR;    // bytes per token row of the latent tensor in HBM
  constexpr int kRopeTok = FMT == 0 ? kDR * 2 : kDR;        // bytes per token of rope (bf16 / fp8)
  __shared__ __attribute__((aligned(16))) uint8_t smem[kLdsBytes];

  const int tid = threadIdx.x;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = LD > 0 && wave_id >= NW;
  const bool issues_dma = LD == 0 || is_loader;
  const int wave = is_loader ? wave_id - NW : wave_id;   // compute wave index, or DMA piece owner index
  const int rg = wave % NRG;   // row group inside the workgroup
  const int W = wave / NRG;    // token half (QK) / d half (PV)
  LaneConst lc;
  lc.lane = tid & 63;
  lc.li = lc.lane & 31;
  lc.lh = lc.lane >> 5;
  {
    const int li = lc.li, lh = lc.lh, lane = lc.lane;
    // K operand: token T = 32W + li, 32 B at d = 64s + 32lh -> chunks c = 4s + 2lh + e, stored at chunk c ^ (T&15).
    // bits of c: e->0, lh->1, s&3->2..3, s>>2->4 (not swizzled: +256 immediate).
    const int kx = li & 15;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2)
        lc.kb[e][k2] = li * kDN + (((((k2 ^ (kx >> 2)) << 2) | ((2 * lh + e) ^ (kx & 3))) << 4));
    // rope operand: token T = 32W + li (W*32 rows is a multiple of 16 -> same swizzle), 16-B chunk 2s + lh stored
    // at chunk (2s+lh) ^ ((T>>1)&7)
    if constexpr (FMT == 0) {
#pragma unroll
      for (int s = 0; s < 4; ++s) lc.rb[s] = li * (kDR * 2) + ((((2 * s + lh) ^ ((li >> 1) & 7))) << 4);
    } else {
      // fp8 rope: 64 B per token = 4 chunks; this lane reads chunks 2lh, 2lh+1, stored at chunk ^ ((T>>2)&3)
#pragma unroll
      for (int s = 0; s < 4; ++s) lc.rb[s] = li * kDR + ((((2 * lh + (s & 1)) ^ ((li >> 2) & 3))) << 4);
    }
    // V^T: B-operand byte q of lane (row n, half lh) is P of token 32(q>>4) + 4lh + (q&3) + 8((q&15)>>2) (bytes 0..15
    // = wave 0's 16 values of that lane, 16..31 = wave 1's).  tr8 read u covers q = 8u..8u+7: source lane
    // s16 = lane&15 reads 8 B of token T = 4lh + tok_in8 + 16(u&1) + 32(u>>1) (immediates), chunk
    // cj = 16W + (jb&3) + 8(jb>>2) + 4gi, half (s16&1)*8.  T&15 = 4lh + tok_in8 for every u.
    const int s16 = lane & 15;
    const int gi = (lane >> 4) & 1;
    const int tj = s16 >> 1;
    const int tok_in8 = (tj & 3) + ((tj >> 2) << 3);
    const int vrow = 4 * lh + tok_in8;
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) {
      const int c4 = (k3 & 3) | (gi << 2) | (((k3 >> 2) & 1) << 3);
      lc.vb[k3] = vrow * kDN + ((c4 ^ (vrow & 15)) << 4) + (s16 & 1) * 8;
    }
  }
  {
    constexpr int kNopePerWave = kDmaNopePerTile / NW;
    constexpr int kRopePerWave = (FMT == 0 ? 8 : 4) / NW;
#pragma unroll
    for (int k = 0; k < kNopePerWave; ++k) {
      const int T = ((wave * kNopePerWave + k) * 2 + lc.lh);   // token row this lane fills
      lc.dn[k] = (unsigned)(T * kTokBytes + ((lc.li ^ (T & 15)) << 4));
    }
#pragma unroll
    for (int k = 0; k < kRopePerWave; ++k) {   // BYTE offsets inside the page's rope block
      if constexpr (FMT == 0) {
        const int T = (wave * kRopePerWave + k) * 8 + (lc.lane >> 3);   // 8 token rows of 128 B per piece
        lc.dr[k] = (unsigned)(T * 128 + (((lc.lane & 7) ^ ((T >> 1) & 7)) << 4));
      } else {
        const int T = (wave * kRopePerWave + k) * 16 + (lc.lane >> 2);  // 16 token rows of 64 B per piece
        lc.dr[k] = (unsigned)(T * kTokBytes + kDN + (((lc.lane & 3) ^ ((T >> 2) & 3)) << 4));
      }
    }
  }
  const int lane = lc.lane, li = lc.li, lh = lc.lh;

  // ---- workgroup -> (part, row group); keep a request's row groups on one XCD (block b runs on XCD b%8) ----
  int part, rgrp;
  {
    const int id = blockIdx.x;
    if ((p.num_parts & 7) == 0) {
      const int xcd = id & 7, k = id >> 3;
      rgrp = k % p.row_groups;
      part = (k / p.row_groups) * 8 + xcd;
    } else {
      rgrp = id % p.row_groups;
      part = id / p.row_groups;
    }
  }
#ifdef FL_MLA_STAGGER
  // experiment: row group 1 starts late so that it finds row group 0's lines already in the XCD's L2
  if (rgrp & 1)
    for (int i = 0; i < FL_MLA_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
  const int32_t* meta = g_meta + part * FL_MLA_META_W;
  int req = meta[0];
  int tile_b = meta[1];
  const int end_req = meta[2];
  const int end_tile = meta[3];
  int split_idx = meta[4];

  const int row = rgrp * (32 * NRG) + rg * 32 + li;   // query row of this lane
  const bool row_ok = row < p.rows;

  float* scratch = reinterpret_cast<float*>(smem + kOffScratch + wave * kScratchPerWave);
#ifdef FL_MLA_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
  const unsigned long long tstart = tlast;
#endif

  if (is_loader) {
    // ---- loader waves: the request loop of the compute waves reduced to its barriers and its LDS-DMA ----
    for (; req < p.bs; ++req, tile_b = 0) {
      if (req > end_req || (req == end_req && end_tile == 0)) break;
      const int L = g_seqlens[req];
      const int nt = L > 0 ? (L + kPage - 1) / kPage : 0;
      int tile_e = req < end_req ? nt : (end_tile < nt ? end_tile : nt);
      if (tile_e < tile_b) tile_e = tile_b;
      const int n = tile_e - tile_b;
      int win_base = 0;
      int pg_vec = 0;
      auto load_window = [&](int base) {
        win_base = base;
        const int t = base + lane;
        int pg = 0;
        if (t < n) pg = g_block_table[(long long)req * p.bt_stride + tile_b + t];
        pg_vec = (pg < 0 || pg >= p.num_pages) ? 0 : pg;
      };
      load_window(0);
      auto page_of = [&](int t) { return (long long)__builtin_amdgcn_readlane(pg_vec, t - win_base); };
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      auto rope_src = [&](int t) {
        const long long pg = page_of(t);
        return FMT == 0 ? g_k_rope + pg * (kPage * kDR) : reinterpret_cast<const uint16_t*>(g_k_nope + pg * (kPage * kTokBytes));
      };
      auto scale_src = [&](int t) { return FMT == 0 ? g_k_scale + page_of(t) * kPage : g_k_scale; };
      auto nope_src = [&](int t) { return g_k_nope + page_of(t) * (kPage * kTokBytes); };
      auto ring = [&](int t) { return smem + kOffRing + (t & 3) * kSlotBytes; };
      auto rope_slot = [&](int t) { return smem + kOffRope + (t & 1) * kRopeBytes; };
      auto scale_slot = [&](int t) { return reinterpret_cast<float*>(smem + kOffScale + (t & 1) * (kPage * 4)); };
      // prologue [r0 s0 n0] [r1 s1 n1] [n2]: loader_page without its wait + barrier is exactly one "issue" call
      auto issue = [&](const uint8_t* sn, uint8_t* dn, const uint16_t* sr, const float* ss, uint8_t* dr, float* ds) {
        constexpr int kNopePerWave = kDmaNopePerTile / NW;
        constexpr int kRopePerWave = (FMT == 0 ? 8 : 4) / NW;
        if (sr != nullptr) {
#pragma unroll
          for (int k = 0; k < kRopePerWave; ++k)
            fl_dma_lds((gbl_ptr_t)(reinterpret_cast<const uint8_t*>(sr) + lc.dr[k]),
                       (lds_ptr_t)(dr + (wave * kRopePerWave + k) * 1024), 16, 0, 0);
          if constexpr (FMT == 0) fl_dma_lds((gbl_ptr_t)(ss + lane), (lds_ptr_t)ds, 4, 0, 0);
        }
        if (sn != nullptr) {
#pragma unroll
          for (int k = 0; k < kNopePerWave; ++k)
            fl_dma_lds((gbl_ptr_t)(sn + lc.dn[k]), (lds_ptr_t)(dn + (wave * kNopePerWave + k) * 1024), 16, 0, 0);
        }
      };
      if (n > 0) issue(nope_src(0), ring(0), rope_src(0), scale_src(0), rope_slot(0), scale_slot(0));
      if (n > 1) issue(nope_src(1), ring(1), rope_src(1), scale_src(1), rope_slot(1), scale_slot(1));
      if (n > 2) issue(nope_src(2), ring(2), nullptr, nullptr, nullptr, nullptr);
      if (n > 2) {   // leave [r1 s1 n1] [n2] in flight (same counts as the compute waves of NRG = 1 had)
        if constexpr (FMT == 0) asm volatile("s_waitcnt vmcnt(37)" ::: "memory");   // (4+1+16) + 16
        else asm volatile("s_waitcnt vmcnt(34)" ::: "memory");                      // (2+0+16) + 16
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      for (int i = 0; i < n; ++i) {
        if (i + 3 >= win_base + 64 && i + 3 < n) load_window(i);   // pages i .. i+63
        const uint8_t* sn = i + 3 < n ? nope_src(i + 3) : nullptr;
        const uint16_t* sr = i + 2 < n ? rope_src(i + 2) : nullptr;
        const float* ss = i + 2 < n ? scale_src(i + 2) : nullptr;
        loader_page<NRG, FMT>(lc, wave, ring(i + 3), rope_slot(i), scale_slot(i), sn, sr, ss, i + 2 < n);
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  for (; req < p.bs; ++req, tile_b = 0, split_idx = 0) {
    if (req > end_req || (req == end_req && end_tile == 0)) break;
    const int L = g_seqlens[req];
    const int nt = L > 0 ? (L + kPage - 1) / kPage : 0;
    int tile_e = req < end_req ? nt : (end_tile < nt ? end_tile : nt);
    if (tile_e < tile_b) tile_e = tile_b;
    const int n = tile_e - tile_b;
    const int split_base = g_num_splits[req];
    const bool is_split = (g_num_splits[req + 1] - split_base) > 1;

    // ---- Q fragments (B operands), once per request ----
    const long long qrow = (long long)req * p.rows + row;
    v8i qn[8];
    v8bf qr[4];
    v8i qr8 = v8i{0, 0, 0, 0, 0, 0, 0, 0};
    float qs = 0.f;
    float ks_const = 1.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) qr[s] = as_bf8(make_uint4(0, 0, 0, 0));
    if constexpr (FMT == 1) ks_const = p.descale_k ? *p.descale_k : 1.f;
    if (row_ok && !is_loader) {
      const uint8_t* qp = g_q_nope + qrow * kTokBytes + lh * 32;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const uint4 a = *reinterpret_cast<const uint4*>(qp + s * 64);
        const uint4 b = *reinterpret_cast<const uint4*>(qp + s * 64 + 16);
        qn[s] = make_v8i(a, b);
      }
      if constexpr (FMT == 0) {
        const uint16_t* rp = g_q_rope + qrow * kDR + lh * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) qr[s] = as_bf8(*reinterpret_cast<const uint4*>(rp + s * 16));
        qs = g_q_scale[qrow] * p.scale_log2e;
      } else {
        qr8 = make_v8i(*reinterpret_cast<const uint4*>(qp + 512), *reinterpret_cast<const uint4*>(qp + 528));
        qs = (p.descale_q ? *p.descale_q : 1.f) * p.scale_log2e;
      }
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) qn[s] = v8i{0, 0, 0, 0, 0, 0, 0, 0};
    }
    int L_row = L;
    if (p.causal) L_row = L - (p.s_q - 1 - row / p.h_q);   // query j sees keys [0, L - (s_q-1-j))
    if (!row_ok) L_row = 0;
    const int L_min = p.causal ? L - (p.s_q - 1) : L;

    ReqState st;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) st.o[j][r] = 0.f;
    st.l_run = 0.f;
    st.lq_run = 0.f;
    st.m_w = kNegRef;
    st.m_o = kNegRef;

    // page ids of a 64-page window live in ONE VGPR (lane j = page win_base + j); a lookup is a v_readlane, not an
    // s_load whose lgkmcnt(0) would stall every page on scalar-memory latency.  The window is reloaded (one vector load,
    // waited for inside the branch) only when the prefetch distance crosses its end.
    int win_base = 0;
    int pg_vec = 0;
    auto load_window = [&](int base) {
      win_base = base;
      const int t = base + lane;
      int pg = 0;
      if (t < n) pg = g_block_table[(long long)req * p.bt_stride + tile_b + t];
      pg_vec = (pg < 0 || pg >= p.num_pages) ? 0 : pg;   // (the use here keeps the load's wait inside this call)
    };
    load_window(0);
    auto page_of = [&](int t) { return (long long)__builtin_amdgcn_readlane(pg_vec, t - win_base); };

    // every wave finished with the LDS of the previous request (and the Q loads above are on the vmcnt queue: drain
    // them before counted waits start)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- prologue: [r0 s0 n0] [r1 s1 n1] [n2] ----
    {
      constexpr int kNopePerWave = kDmaNopePerTile / NW;
      constexpr int kRopePerWave = (FMT == 0 ? 8 : 4) / NW;
      auto dma_rs = [&](int t) {
        const long long pg = page_of(t);
        const uint8_t* sr = FMT == 0 ? reinterpret_cast<const uint8_t*>(g_k_rope) + pg * (kPage * kRopeTok)
                                     : g_k_nope + pg * (kPage * kTokBytes);
        uint8_t* dr = smem + kOffRope + (t & 1) * kRopeBytes;
#pragma unroll
        for (int k = 0; k < kRopePerWave; ++k)
          fl_dma_lds((gbl_ptr_t)(sr + lc.dr[k]), (lds_ptr_t)(dr + (wave * kRopePerWave + k) * 1024),
                                           16, 0, 0);
        if constexpr (FMT == 0) {
          const float* ss = g_k_scale + pg * kPage;
          float* ds = reinterpret_cast<float*>(smem + kOffScale + (t & 1) * (kPage * 4));
          fl_dma_lds((gbl_ptr_t)(ss + lane), (lds_ptr_t)ds, 4, 0, 0);
        }
      };
      auto dma_n = [&](int t) {
        const uint8_t* sn = g_k_nope + page_of(t) * (kPage * kTokBytes);
        uint8_t* dn = smem + kOffRing + (t & 3) * kSlotBytes;
#pragma unroll
        for (int k = 0; k < kNopePerWave; ++k)
          fl_dma_lds((gbl_ptr_t)(sn + lc.dn[k]), (lds_ptr_t)(dn + (wave * kNopePerWave + k) * 1024),
                                           16, 0, 0);
      };
      if (issues_dma) {
        if (n > 0) { dma_rs(0); dma_n(0); }
        if (n > 1) { dma_rs(1); dma_n(1); }
        if (n > 2) dma_n(2);
      }
      if (n > 2) {   // leave [r1 s1 n1] [n2] in flight
        if constexpr (NRG == 2 && FMT == 0) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");        // (2+1+8) + 8
        else if constexpr (NRG == 1 && FMT == 0) asm volatile("s_waitcnt vmcnt(37)" ::: "memory");   // (4+1+16) + 16
        else if constexpr (NRG == 2 && FMT == 1) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");   // (1+0+8) + 8
        else asm volatile("s_waitcnt vmcnt(34)" ::: "memory");                                       // (2+0+16) + 16
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
    // page 0: scale triples + first half of the QK operands (later pages get them in the previous page's PV shadow)
    QkPrefetch pre;
    if (n > 0 && !is_loader) {
      scale_prep<FMT>(reinterpret_cast<const float*>(smem + kOffScale), scratch, ks_const, W, li, lh, tile_b * kPage, L);
      qk_prefetch<FMT>(pre, lc, smem + kOffRing, smem + kOffRope, W);
    }

    for (int i = 0; i < n; ++i) {
      if (i + 3 >= win_base + 64 && i + 3 < n) load_window(i);   // pages i .. i+63
      const uint8_t* sn = nullptr;
      const uint16_t* sr = nullptr;
      const float* ss = nullptr;
      if (i + 3 < n) sn = g_k_nope + page_of(i + 3) * (kPage * kTokBytes);
      if (i + 2 < n) {
        const long long pg = page_of(i + 2);
        if constexpr (FMT == 0) {
          sr = g_k_rope + pg * (kPage * kDR);
          ss = g_k_scale + pg * kPage;
        } else {
          sr = reinterpret_cast<const uint16_t*>(g_k_nope + pg * (kPage * kTokBytes));
          ss = reinterpret_cast<const float*>(sr);   // unused (non-null)
        }
      }
      if (LD > 0) { sn = nullptr; sr = nullptr; ss = nullptr; }   // the loader waves issue the refill
      tile_body<NRG, FMT>(st, pre, lc, qn, qr, qr8, qs, ks_const, W, rg, wave,
                     smem + kOffRing + ((i + 1) & 3) * kSlotBytes, smem + kOffRope + ((i + 1) & 1) * kRopeBytes,
                     reinterpret_cast<const float*>(smem + kOffScale + ((i + 1) & 1) * (kPage * 4)), i + 1 < n,
                     smem + kOffRing + (i & 3) * kSlotBytes, smem + kOffRope + (i & 1) * kRopeBytes,
                     reinterpret_cast<const float*>(smem + kOffScale + (i & 1) * (kPage * 4)), scratch,
                     smem + kOffPbuf + (i & 1) * kPbufPerParity,
                     reinterpret_cast<float*>(smem + kOffRef + (i & 1) * kRefPerParity),
                     smem + kOffRing + ((i + 3) & 3) * kSlotBytes, smem + kOffRope + (i & 1) * kRopeBytes,
                     reinterpret_cast<float*>(smem + kOffScale + (i & 1) * (kPage * 4)), sn, sr, ss,
                     (tile_b + i) * kPage, L, L_row, L_min, i + 2 < n
#ifdef FL_MLA_TIMING
                     , tacc, tlast
#endif
                     );
    }

    // ---- per-request epilogue: merge the two normalisers of the row group, normalise, store this wave's d half ----
    const float l_tot = st.l_run + __shfl_xor(st.l_run, 32);
    const float lq_tot = st.lq_run + __shfl_xor(st.lq_run, 32);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // everyone is past the last page (P buffers free)
    float* lm = reinterpret_cast<float*>(smem + kOffPbuf);   // [rg][W][3][32]
    if (lh == 0 && !is_loader) {
      lm[((rg * 2 + W) * 3 + 0) * 32 + li] = l_tot;
      lm[((rg * 2 + W) * 3 + 1) * 32 + li] = lq_tot;
      lm[((rg * 2 + W) * 3 + 2) * 32 + li] = st.m_w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const float mA = lm[((rg * 2 + 0) * 3 + 2) * 32 + li], mB = lm[((rg * 2 + 1) * 3 + 2) * 32 + li];
    const float fA = __builtin_amdgcn_exp2f(mA - st.m_o), fB = __builtin_amdgcn_exp2f(mB - st.m_o);   // m_o >= mA, mB
    const float l = lm[((rg * 2 + 0) * 3 + 0) * 32 + li] * fA + lm[((rg * 2 + 1) * 3 + 0) * 32 + li] * fB;
    const float lq = lm[((rg * 2 + 0) * 3 + 1) * 32 + li] * fA + lm[((rg * 2 + 1) * 3 + 1) * 32 + li] * fB;
    const float inv = lq > 0.f ? 1.f / lq : 0.f;
    const float lse_nat = l > 0.f ? (__builtin_amdgcn_logf(l) + st.m_o - kPShift) * 0.6931471805599453f : -INFINITY;
    // split-KV partials are normalised by lq, so they must also be COMBINED with lq-based weights (then the combine is
    // exactly the unsplit sum O/lq); the exact LSE travels next to it for the reported lse.
    const float lseq_nat = lq > 0.f ? (__builtin_amdgcn_logf(lq) + st.m_o - kPShift) * 0.6931471805599453f : -INFINITY;
    if (row_ok && !is_loader) {
      const int slot_idx = split_base + split_idx;
      if (lh == 0 && W == 0) {
        if (is_split) {
          p.lse_accum[((long long)slot_idx * p.rows + row) * 2 + 0] = lseq_nat;
          p.lse_accum[((long long)slot_idx * p.rows + row) * 2 + 1] = lse_nat;
        } else {
          const int j = row / p.h_q, h = row - j * p.h_q;
          p.lse[((long long)req * p.h_q + h) * p.s_q + j] = lse_nat;
        }
      }
      // C row i = e + 8g + 4*lh of tile jb  ->  d = 256W + (jb>>2)*128 + (jb&3)*16 + (i&15) + 64*(i>>4)
      if (is_split) {
        float* dbase = p.o_accum + ((long long)slot_idx * p.rows + row) * kDN + 256 * W;
#pragma unroll
        for (int jb = 0; jb < 8; ++jb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int i0 = 8 * g + 4 * lh;
            const int d0 = (jb >> 2) * 128 + (jb & 3) * 16 + (i0 & 15) + 64 * (i0 >> 4);
            *reinterpret_cast<float4*>(dbase + d0) = make_float4(st.o[jb][g * 4 + 0] * inv, st.o[jb][g * 4 + 1] * inv,
                                                                  st.o[jb][g * 4 + 2] * inv, st.o[jb][g * 4 + 3] * inv);
          }
      } else {
        uint16_t* dbase = p.out + qrow * kDN + 256 * W;
#pragma unroll
        for (int jb = 0; jb < 8; ++jb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int i0 = 8 * g + 4 * lh;
            const int d0 = (jb >> 2) * 128 + (jb & 3) * 16 + (i0 & 15) + 64 * (i0 >> 4);
            const uint32_t lo = (uint32_t)fl_f32_to_bf16(st.o[jb][g * 4 + 0] * inv) |
                                ((uint32_t)fl_f32_to_bf16(st.o[jb][g * 4 + 1] * inv) << 16);
            const uint32_t hi = (uint32_t)fl_f32_to_bf16(st.o[jb][g * 4 + 2] * inv) |
                                ((uint32_t)fl_f32_to_bf16(st.o[jb][g * 4 + 3] * inv) << 16);
            *reinterpret_cast<uint2*>(dbase + d0) = make_uint2(lo, hi);
          }
      }
    }
  }
#ifdef FL_MLA_TIMING
  if (g_dbg != nullptr && lc.lane == 0) {
    unsigned long long* d = reinterpret_cast<unsigned long long*>(g_dbg) + ((long long)blockIdx.x * 4 + wave) * 10;
    for (int i = 0; i < 8; ++i) d[i] = tacc[i];
    d[8] = __builtin_readcyclecounter() - tstart;
    d[9] = tlast - tstart;
  }
#endif
}


// ---- split-KV combine: out[req,row,:] = sum_s w_s * o_accum[slot_s,row,:], w_s = softmax_s(lse_s) ----
__global__ __launch_bounds__(256) void mla_combine_kernel(const Params p, const int32_t* __restrict__ g_num_splits) {
  // nothing is split (every request has exactly one part): one scalar load per workgroup and out
  if (g_num_splits[p.bs] == p.bs) return;
  const int lane = threadIdx.x & 63;
  // the wave index is uniform: say so, and the split counts / LSEs of the row come through the scalar cache
  const long long total = (long long)p.bs * p.rows;
  for (long long gw = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); gw < total;
       gw += (long long)gridDim.x * 4) {
    const int req = (int)(gw / p.rows), row = (int)(gw % p.rows);
    const int s0 = g_num_splits[req], ns = g_num_splits[req + 1] - s0;
    if (ns <= 1) continue;
    if (ns == 2) combine_row<2>(p, req, row, s0, ns, lane);   // the uniform full batch: two parts per request
    else if (ns == 3) combine_row<3>(p, req, row, s0, ns, lane);
    else if (ns == 4) combine_row<4>(p, req, row, s0, ns, lane);
    else combine_row<0>(p, req, row, s0, ns, lane);
  }
}

// few rows, many splits per row (decode at bs = 1..8): ONE row per workgroup, its splits over the four waves (combine_row QUAD)
__global__ __launch_bounds__(256) void mla_combine_quad_kernel(const Params p, const int32_t* __restrict__ g_num_splits) {
  __shared__ float red[3 * 64 * 10];
  if (g_num_splits[p.bs] == p.bs) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int req = (int)(blockIdx.x / p.rows), row = (int)(blockIdx.x % p.rows);
  const int s0 = g_num_splits[req], ns = g_num_splits[req + 1] - s0;
  if (ns <= 1) return;   // (uniform over the workgroup)
  combine_row<0, true>(p, req, row, s0, ns, lane, red, wave);
}

}  // namespace

int fl_mla_launch_combine(const Params& p, const int32_t* num_splits, hipStream_t stream) {
  const long long waves = (long long)p.bs * p.rows;
  // (requests x rows) below one wave per SIMD of the chip AND more parts than requests x 8 (every request is cut into many
  // pieces — the split counts themselves live on the device): a workgroup per row
  if (waves <= 1024 && p.num_parts >= 8 * p.bs) {
    mla_combine_quad_kernel<<<dim3((unsigned)waves), dim3(256), 0, stream>>>(p, num_splits);
    FL_CHECK_LAUNCH("mla_combine_quad_kernel");
    return FL_OK;
  }
  const long long blocks = (waves + 3) / 4;   // one wave per (request, row); grid-stride beyond 2048 workgroups
  mla_combine_kernel<<<dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, stream>>>(p, num_splits);
  FL_CHECK_LAUNCH("mla_combine_kernel");
  return FL_OK;
}

int fl_mla_decode_fp8_y_impl(const FlMlaDecodeArgs* a, const Params& p, hipStream_t stream);   // mla_decode_fp8_y.hip

// bytes of the split-partials workspaces fl_mla_decode needs for a launch of this shape — the SAME dispatch rule as below
// decides whether partials are bf16 rows (role-specialised / slot-pipelined mappings) or f32 rows
extern "C" int fl_mla_workspace_bytes(int kv_format, int bs, int s_q, int h_q, int num_parts, int64_t* o_accum_bytes,
                                      int64_t* lse_accum_bytes) {
  FL_CHECK_ARG(o_accum_bytes && lse_accum_bytes && bs >= 0 && s_q >= 1 && h_q >= 1 && num_parts >= 1, "fl_mla_workspace_bytes: bad arguments");
  const long long rows = (long long)s_q * h_q;
  // more than 32 query rows per request (fp8 formats): the role-specialised kernel, whose split partials are bf16 rows
  const bool bf16_partials = kv_format != FL_KV_BF16_576 && rows > 32;
  *o_accum_bytes = (long long)(bs + num_parts) * rows * 512 * (bf16_partials ? 2 : 4);
  *lse_accum_bytes = (long long)(bs + num_parts) * rows * 2 * 4;
  return FL_OK;
}

int fl_mla_decode_fp8_impl(const FlMlaDecodeArgs* a, hipStream_t stream) {
  const bool per_token = a->kv_format == FL_KV_FP8_PER_TOKEN;
  FL_CHECK_ARG(a->d_nope == kDN && a->d_rope == kDR, "fl_mla_decode: only d_nope=512,d_rope=64 (got %d,%d)",
               a->d_nope, a->d_rope);
  const bool q_bf16 = a->q_bf16 != nullptr;
  FL_CHECK_ARG((q_bf16 || a->q_nope) && a->k_nope, "fl_mla_decode: null q/k pointer");
  FL_CHECK_ARG(!per_token || ((q_bf16 || (a->q_rope && a->q_scale)) && a->k_rope && a->k_scale),
               "fl_mla_decode(per-token fp8): null rope/scale pointer");
  FL_CHECK_ARG(!q_bf16 || (per_token && a->s_q * a->h_q > 32 && ((uintptr_t)a->q_bf16 % 16) == 0),
               "fl_mla_decode: q_bf16 (K4 inside the decode kernel) is served for the per-token format with more than 32 query rows "
               "per request (role-specialised kernel), 16-byte aligned");
  FL_CHECK_ARG(a->block_table && a->cache_seqlens && a->tile_scheduler_metadata && a->num_splits && a->out && a->lse &&
                   a->o_accum && a->lse_accum,
               "fl_mla_decode: null metadata/output pointer");
  FL_CHECK_ARG(a->bs >= 0 && a->s_q >= 1 && a->h_q >= 1 && a->num_parts >= 1, "fl_mla_decode: bad sizes");
  if (a->bs == 0) return FL_OK;
  Params p;
  p.bs = a->bs; p.s_q = a->s_q; p.h_q = a->h_q; p.rows = a->s_q * a->h_q; p.causal = a->causal;
  p.num_parts = a->num_parts;
  p.scale_log2e = a->softmax_scale * kLog2e;
  p.descale_q = a->descale_q; p.descale_k = a->descale_k;
  p.num_pages = a->num_pages; p.bt_stride = a->block_table_stride;
  p.bt_cols = a->block_table_cols > 0 ? a->block_table_cols : (a->block_table_stride > 0 ? a->block_table_stride : 1);
  p.out = (uint16_t*)a->out; p.lse = a->lse; p.o_accum = a->o_accum; p.lse_accum = a->lse_accum;
  p.partial_bf16 = 0;
  p.merge_in_kernel = 0;
  p.q_bf16 = (const uint16_t*)a->q_bf16;
  // More than 32 query rows per request: role-specialised 64-row workgroups (mla_decode_fp8_y.hip; both fp8 formats — the plain
  // [.,576] cache is the kernel's FMT = 1 instantiation).  fl_mla_num_parts sizes the scheduler's part count with the same rule.
  if (p.rows > 32) return fl_mla_decode_fp8_y_impl(a, p, stream);
  // rows <= 32 (e.g. the TP8 shard, H=16): one 32-row group per workgroup (2 compute waves + 2 loader waves)
  constexpr int nrg = 1;
  p.row_groups = (p.rows + 31) / 32;
  const dim3 grid((unsigned)(p.num_parts * p.row_groups)), block(64 * (2 * nrg + loader_waves(nrg)));
#define FL_LAUNCH(NRG_, FMT_)                                                                                          \
  mla_decode_fp8_kernel<NRG_, FMT_><<<grid, block, 0, stream>>>(                                                       \
      p, a->block_table, a->cache_seqlens, a->tile_scheduler_metadata, a->num_splits, (const uint8_t*)a->k_nope,        \
      (const uint16_t*)a->k_rope, a->k_scale, (const uint8_t*)a->q_nope, (const uint16_t*)a->q_rope, a->q_scale)
  if (per_token) FL_LAUNCH(1, 0);
  else FL_LAUNCH(1, 1);
#undef FL_LAUNCH
  FL_CHECK_LAUNCH("mla_decode_fp8_kernel");
  return fl_mla_launch_combine(p, a->num_splits, stream);
}

#if defined(FL_MLA_DEBUG) || defined(FL_MLA_TIMING)
extern "C" int fl_mla_debug_set_buffer(int* dev_ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &dev_ptr, sizeof(dev_ptr));
}
#endif

// ---- fl_mla_decode — C-ABI dispatch over the KV-cache formats of MLATokenToKVPool (memory_pool.py:635-658) ----
int fl_mla_decode_bf16_impl(const FlMlaDecodeArgs* a, hipStream_t stream);   // mla_decode_bf16.hip

extern "C" int fl_mla_decode(const FlMlaDecodeArgs* args, fl_stream_t stream) {
  FL_CHECK_ARG(args != nullptr, "fl_mla_decode: null args");
  FL_CHECK_ARG(args->struct_bytes == (int32_t)sizeof(FlMlaDecodeArgs),
               "fl_mla_decode: FlMlaDecodeArgs.struct_bytes = %d, this library (ABI %d) expects %d — caller and library were built against "
               "different include/fluent_mi355.h", args->struct_bytes, FL_ABI_VERSION, (int)sizeof(FlMlaDecodeArgs));
  switch (args->kv_format) {
    case FL_KV_FP8_PER_TOKEN:
    case FL_KV_FP8_576:
      return fl_mla_decode_fp8_impl(args, (hipStream_t)stream);
    case FL_KV_BF16_576:
      return fl_mla_decode_bf16_impl(args, (hipStream_t)stream);
    default:
      fl_set_error("fl_mla_decode: kv_format %d not implemented", args->kv_format);
      return FL_ERR_UNSUPPORTED;
  }
}
