// K1/K2 for more than 64 query rows per request (TP1: s_q*H = 128, 256, ...) — paged MLA decode over the FP8 latent
// KV cache, gfx950 (MI355X) only.  Same math, data formats and call sites as mla_decode_fp8.hip
// (flash_mla_fp8.flash_mla_ckv_fp8_per_token, flashmla_backend.py:208-222 / :127-142; flash_mla_with_kvcache :227-239).
//
// Why a second mapping: with 64-row workgroups a 128-head request needs two workgroups that each pull every KV page
// through their own LDS (measured: the per-CU LDS-DMA ingest, not HBM, then bounds the kernel).  Here ONE workgroup of
// 4 waves owns 128 query rows and every KV byte enters one LDS once; the scheduler splits the token axis instead
// (fl_mla_num_parts = CUs / ceil(rows/128)) and the split partials are merged by the combine kernel.
//
//   * wave w owns query rows [32w, 32w+32) for ALL 64 tokens of a page and ALL 512 latent dims:
//       S^T block b (tokens 32b..32b+31) = K_b · Q^T : 8 x v_mfma_scale_f32_32x32x64_f8f6f4 (+ 4 x 32x32x16 bf16 rope)
//       O^T[512 x 32 rows] += V^T · P^T              : 16 x v_mfma_scale_f32_32x32x64_f8f6f4, O = all 256 AGPRs
//     so P never leaves the lane that produced it: bytes 0..15 of the PV B operand are the lane's block-0 weights,
//     bytes 16..31 its block-1 weights (MX block b takes its E8M0 scale from lane n+32b: the two blocks keep
//     INDEPENDENT integer softmax references m_b, reconciled for free by the block scales 2^(m_b - M)).  No P exchange,
//     no cross-wave reduction; the only workgroup barrier per page certifies the LDS ring.
//   * software pipeline inside one wave (one wave per SIMD, in-order issue): iteration i runs
//       QK block 0 of page i      ||  softmax of block 1 of page i-1
//       PV of page i-1 (16 MFMA)  ||  softmax of block 0 of page i
//       QK block 1 of page i
//     so the VALU/transcendental work of a block always sits in the shadow of independent MFMAs.
//   * LDS: 4-slot ring of 32 KiB latent pages (page i-1 is still read for V^T while page i is read for K and pages
//     i+1, i+2 are landing), 3-slot rings for rope and raw scales, per-wave scale-triple scratch (2 parities).
//     HBM -> LDS only by global_load_lds; swizzles, operand layouts and tr8 V^T reads as in mla_decode_fp8.hip.
#include "mla_decode_shared.h"

using namespace fl_mla;

namespace {

constexpr int kNW = 4;                                          // waves per workgroup, 32 query rows each
constexpr int kRopeSlots = 3;
constexpr int kOffRing = 0;                                     // 4 x 32 KiB
constexpr int kOffRope = kOffRing + kRingSlots * kSlotBytes;    // 3 x 8 KiB
constexpr int kOffScale = kOffRope + kRopeSlots * kRopeBytes;   // 3 x 64 f32 raw k_scale
constexpr int kScratchPerWave = 3 * kPage * 4;                  // {ks, log2 ks, 1/ks} x 64 tokens
constexpr int kOffScratch = kOffScale + kRopeSlots * kPage * 4; // [wave 4]
constexpr int kOffFlag = kOffScratch + 2 * kNW * kScratchPerWave;  // [parity 2][wave 4] scratch; then 4 ints: redo votes
constexpr int kLdsBytes = kOffFlag + 16;
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");

constexpr int kNopePerWave = kDmaNopePerTile / kNW;             // 8 LDS-DMA pieces of 1 KiB per wave per page
constexpr float kMaxUp = 100.f;                                 // largest block-scale exponent above the O reference
#ifndef FL_X_INPLACE
#define FL_X_INPLACE 0   // 1: lane constants opaque in place (9 fewer v_mov per step, but hipcc then spills Q fragments)
#endif
#ifndef FL_X_KAHEAD
#define FL_X_KAHEAD 2
#endif
#ifndef FL_X_OVERLAP
#define FL_X_OVERLAP true    // softmax of block 0 inside the QK MFMA chain of block 1 (register pressure!)
#endif

#ifdef FL_MLA_TIMING
__device__ int* g_dbg_x = nullptr;   // debug builds only: set by fl_mla_debug_set_buffer_x
#define FL_T(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); tacc[i] += t__ - tlast; tlast = t__; } while (0)
#define FL_TV(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); if (i) tacc[i] += t__ - tvar; tvar = t__; } while (0)
#define FL_T_PARAMS , unsigned long long (&tacc)[16], unsigned long long& tlast
#define FL_T_ARGS , tacc, tlast
#else
#define FL_T(i) do { } while (0)
#define FL_TV(i) do { } while (0)
#define FL_T_PARAMS
#define FL_T_ARGS
#endif

struct LaneConst {
  int lane, li, lh;
  int kb0;         // K operand: byte offset inside a slot for k-step 0, first 16 B (second: ^16); k-step s: ^ ((s&3) << 6), + (s>>2)*256; + b*16384
  int rb0;         // rope operand, k-step 0 (k-step s: ^ (s << 5); FMT 1 second half: ^16)
  int vb0;         // V^T tr8 source of tile jb = 0 (tile jb: ^ ((jb&3) << 4) ^ ((jb>>2) << 7)); + u immediates; + dh*256
  unsigned dn_row; // latent DMA: byte offset of this lane's token row of piece 0 of this wave
  unsigned dn_x;   // ... and its swizzled 16-B chunk (piece k: ^ (k << 5))
  unsigned dr[2];
};

// Online softmax of one S^T block on y = s*log2e + log2(k_scale[t]) (folds the per-token V scale into P), in chunks.
// Lane (li, lh) holds query row li and tokens 32b + 8g + 4lh + e.  Result: 16 fp8 weights P' = 2^(y - m + 8).
struct Soft {
  v16f y;
  float4 a4[4], b4[3];   // a4[g]: k_scale of group g, later 1/k_scale; b4[g % 3]: log2 k_scale of group g
  float tmax, moff;
  int pk[4];
};
#ifndef FL_X_C0QK
#define FL_X_C0QK 12
#endif
#ifndef FL_X_C1PV
#define FL_X_C1PV 10
#endif
constexpr int kC0UnderQK = FL_X_C0QK;   // softmax pieces of block 0 under QK block 1 (slots -1..10); the rest under PV
constexpr int kC1UnderPV = FL_X_C1PV;   // softmax pieces of block 1 under PV; the rest under the next page's QK block 0

struct ReqState {
  v16f o[16];            // O^T: tile dh*8 + jb
  float l[2], lq[2];     // exact / rounded-weight normalisers per block, relative to mw[b]
  float mw[2];           // integer softmax references per block
  float mo;              // reference of O (fixed once set)
  Soft c1;               // softmax of block 1 of the newest page, kC1UnderPV pieces done (rest: next page's QK block 0)
  uint4 p0;              // fp8 weights of block 0 of the newest page (PV pending)
  int redo;              // a block reference outran mo by more than kMaxUp: repeat the request with mo preset
};

// ---- hand-scheduled building blocks.  hipcc's scheduler, left alone with a whole pipeline step, hoists every LDS
// operand read to the top (hundreds of live registers, spills).  A step is therefore written as a sequence of SLOTS —
// one MFMA + a bounded chunk of the overlapped softmax + the operand reads of two slots ahead — fenced by
// sched_barrier(0), i.e. the software pipeline is fixed in the source. ----
#define FL_SLOT_END() __builtin_amdgcn_sched_barrier(0)

// K operand (A side) of QK k-step s of block b: token 32b + li, 32 B at d = 64s + 32lh
struct KOp { uint4 lo, hi; };
__device__ __forceinline__ KOp k_load(const int (&kbk)[4], const uint8_t* __restrict__ kp, const int s) {
  KOp r;   // (k-steps s and s + 4 share their two addresses: + 256 is a ds_read immediate)
  r.lo = *reinterpret_cast<const uint4*>(kp + kbk[s & 3] + (s >> 2) * 256);
  r.hi = *reinterpret_cast<const uint4*>(kp + (kbk[s & 3] ^ 16) + (s >> 2) * 256);
  return r;
}

// V^T operand (A side) of PV tile t = dh*8 + jb: ds_read_b64_tr_b8 x 4
__device__ __forceinline__ v8i vt_load(const LaneConst& lc, const uint8_t* __restrict__ v_nope, const int t) {
  const int dh = t >> 3, jb = t & 7;
  const uint8_t* vp = v_nope + dh * 256 + (lc.vb0 ^ (((jb & 3) << 4) | ((jb >> 2) << 7)));
  v8i va;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint8_t* ap = vp + (u & 1) * (16 * kDN) + (u >> 1) * (32 * kDN);
    const v2i t2 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)ap);
    va[2 * u] = t2[0];
    va[2 * u + 1] = t2[1];
  }
  return va;
}

// QK MFMAs with the S accumulator pinned to the VGPR half.  This file is compiled in hipcc's default MFMA form for one
// wave per SIMD (accumulators in AGPRs): exactly right for the 16 O tiles, which fill all 256 AGPRs, and wrong for S,
// which the softmax reads with VALU instructions — there is no per-instruction switch (and -amdgpu-mfma-vgpr-form=1
// moves O through VGPRs around every PV MFMA).  Hence inline asm with "v" constraints; the XDL-write -> VALU-read
// hazard that the compiler cannot see through inline asm is covered by FL_MFMA_DRAIN() before S is first read.
__device__ __forceinline__ v4i as_v4i(const uint4 a) { return v4i{(int)a.x, (int)a.y, (int)a.z, (int)a.w}; }
__device__ __forceinline__ v4i as_v4i(const v8bf b) {
  union { v8bf b; v4i i; } x;
  x.b = b;
  return x.i;
}
__device__ __forceinline__ void mfma_rope_first(v16f& acc, const uint4 a, const v8bf b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(as_v4i(a)), "v"(as_v4i(b)));
}
__device__ __forceinline__ void mfma_rope(v16f& acc, const uint4 a, const v8bf b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(as_v4i(a)), "v"(as_v4i(b)));
}
__device__ __forceinline__ void mfma_fp8_first(v16f& acc, const v8i a, const v8i b) {
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %3 op_sel_hi:[0,0,0]"
               : "=&v"(acc)
               : "v"(a), "v"(b), "v"(kUnitScale));
}
__device__ __forceinline__ void mfma_fp8(v16f& acc, const v8i a, const v8i b) {
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
               : "+v"(acc)
               : "v"(a), "v"(b), "v"(kUnitScale));
}
// (a 16-pass XDL write needs 18 wait states of 4 clocks before a VALU read; s_nop 7 = 8 wait states, measured 36 clocks)
#define FL_MFMA_DRAIN() asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 1" ::: "memory")

// Every LDS read of the softmax is issued TWO chunks (MFMA slots) before its first use: the LDS round trip is ~140
// cycles unloaded, more than one 64-cycle slot.
__device__ __forceinline__ void sm_load_kl(Soft& c, const float* __restrict__ scr, const int g, const int b,
                                           const int lh) {
  const int tb = 32 * b + g * 8 + lh * 4;
  c.a4[g] = *reinterpret_cast<const float4*>(scr + tb);
  c.b4[g % 3] = *reinterpret_cast<const float4*>(scr + kPage + tb);
}
__device__ __forceinline__ void sm_load_ik(Soft& c, const float* __restrict__ scr, const int g, const int b,
                                           const int lh) {
  c.a4[g] = *reinterpret_cast<const float4*>(scr + 2 * kPage + 32 * b + g * 8 + lh * 4);
}
__device__ __forceinline__ void sm_scale(Soft& c, const int g, const float qs, const int b, const int lh, const int tok0,
                                         const int L_row, const bool need_mask) {
  const float ksv[4] = {c.a4[g].x, c.a4[g].y, c.a4[g].z, c.a4[g].w};
  const float lkv[4] = {c.b4[g % 3].x, c.b4[g % 3].y, c.b4[g % 3].z, c.b4[g % 3].w};
  float y[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) y[e] = fmaf(c.y[g * 4 + e] * qs, ksv[e], lkv[e]);
  if (need_mask) {   // wave-uniform: only the last page(s) of a sequence
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (tok0 + 32 * b + g * 8 + lh * 4 + e >= L_row) y[e] = -INFINITY;
      if (!(y[e] == y[e])) y[e] = -INFINITY;   // NaN can only come from garbage beyond the row's limit
    }
  }
  c.tmax = fmaxf(fmaxf(c.tmax, fmaxf(y[0], y[1])), fmaxf(y[2], y[3]));
  // pin: the chunk's results are "used" here, so that the optimiser cannot sink its VALU work out of the MFMA slot
  asm volatile("" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(c.tmax));
#pragma unroll
  for (int e = 0; e < 4; ++e) c.y[g * 4 + e] = y[e];
}
__device__ __forceinline__ void sm_ref(Soft& c, float& m_w, float& l_run, float& lq_run) {
  const float tmax = fmaxf(c.tmax, __shfl_xor(c.tmax, 32));
  const float m_new = tmax > m_w ? ceilf(tmax) + kRefHeadroom : m_w;
  const float f = __builtin_amdgcn_exp2f(m_w - m_new);   // exactly 1 when the reference did not move
  l_run *= f;
  lq_run *= f;
  m_w = m_new;
  c.moff = kPShift - m_new;
  asm volatile("" : "+v"(l_run), "+v"(lq_run), "+v"(m_w), "+v"(c.moff));   // pin (see sm_scale)
  // unspecified start values for the packed weights (v_cvt_pk_fp8_f32 writes half a register and keeps the other half:
  // a defined 0 costs a v_mov per group)
  asm volatile("" : "=v"(c.pk[0]), "=v"(c.pk[1]), "=v"(c.pk[2]), "=v"(c.pk[3]));
}
// half h (elements 2h, 2h+1) of group g: the unit that fits beside one 64-cycle MFMA
__device__ __forceinline__ void sm_exp_half(Soft& c, const int g, const int h, float& l_run, float& lq_run) {
  const float e0 = __builtin_amdgcn_exp2f(c.y[g * 4 + 2 * h + 0] + c.moff);
  const float e1 = __builtin_amdgcn_exp2f(c.y[g * 4 + 2 * h + 1] + c.moff);
  const float k0 = h ? c.a4[g].z : c.a4[g].x, k1 = h ? c.a4[g].w : c.a4[g].y;
  l_run = fmaf(e0, k0, l_run);   // unrounded sum: exact LSE
  l_run = fmaf(e1, k1, l_run);
  c.pk[g] = h ? __builtin_amdgcn_cvt_pk_fp8_f32(e0, e1, c.pk[g], true) : __builtin_amdgcn_cvt_pk_fp8_f32(e0, e1, c.pk[g], false);
  // the ROUNDED weights normalise O (numerator and denominator use the same weights: they sum to exactly 1)
  const float2v d = h ? __builtin_amdgcn_cvt_pk_f32_fp8(c.pk[g], true) : __builtin_amdgcn_cvt_pk_f32_fp8(c.pk[g], false);
  lq_run = fmaf(d[0], k0, lq_run);
  lq_run = fmaf(d[1], k1, lq_run);
  asm volatile("" : "+v"(l_run), "+v"(lq_run), "+v"(c.pk[g]));   // pin (see sm_scale)
}
// piece k (0..13) of a block's softmax; every LDS read is issued two or more pieces before its first use:
//   0: loads {ks, lk}(0), (1)     1..4: scale g = k-1 (+ loads (k+1) for k <= 2, ik(0) at k = 4)     5: reference, ik(1)
//   6 + 2g + h: exp of half h of group g (+ ik(g + 2) at h = 0, g < 2)
constexpr int kSoftPieces = 14;
template <int DUMMY = 0>
__device__ __forceinline__ void sm_piece(Soft& c, const int k, const float* __restrict__ scr, const int b, const int lh,
                                         const float qs, const int tok0, const int L_row, const bool need_mask,
                                         float& m_w, float& l_run, float& lq_run) {
  if (k == 0) {
    c.tmax = -INFINITY;
    sm_load_kl(c, scr, 0, b, lh);
    sm_load_kl(c, scr, 1, b, lh);
  } else if (k <= 4) {
    if (k <= 2) sm_load_kl(c, scr, k + 1, b, lh);
    sm_scale(c, k - 1, qs, b, lh, tok0, L_row, need_mask);
    if (k == 4) sm_load_ik(c, scr, 0, b, lh);
  } else if (k == 5) {
    sm_ref(c, m_w, l_run, lq_run);
    sm_load_ik(c, scr, 1, b, lh);
  } else if (k < kSoftPieces) {
    const int g = (k - 6) >> 1, h = (k - 6) & 1;
    if (h == 0 && g < 2) sm_load_ik(c, scr, g + 2, b, lh);
    sm_exp_half(c, g, h, l_run, lq_run);
  }
}

struct NoHook { __device__ __forceinline__ void operator()(int) const {} };

// One QK block: 4 rope slots (FMT 0; one fp8 slot for FMT 1) + 8 latent slots.  hook(-1) runs before the first MFMA,
// hook(s), s = 0..11, inside slot s AFTER its MFMA and operand reads — the overlapped softmax pieces / LDS-DMA issue.
template <int FMT, class Hook = NoHook>
__device__ __forceinline__ v16f qk_stage(const LaneConst& lc, const uint8_t* __restrict__ k_nope,
                                         const uint8_t* __restrict__ k_rope, const int b, const v8i (&qn)[8],
                                         const v8bf (&qr)[4], const v8i qr8, const Hook& hook = Hook()) {
  const uint8_t* kp = k_nope + b * (32 * kDN);
  const uint8_t* rp = k_rope + b * (32 * (FMT == 0 ? kDR * 2 : kDR));
  uint4 ra[4];
  ra[0] = *reinterpret_cast<const uint4*>(rp + lc.rb0);
  ra[1] = *reinterpret_cast<const uint4*>(rp + (lc.rb0 ^ (FMT == 0 ? 32 : 16)));
  constexpr int kAhead = FL_X_KAHEAD;   // latent operands are read kAhead slots before their MFMA (LDS round trip ~140+ cycles)
  KOp ka[kAhead + 1];
  int kb0 = lc.kb0;   // opaque per stage: the three derived k-step offsets are not kept live across stages (spills)
  asm volatile("" : "+v"(kb0));
  const int kbk[4] = {kb0, kb0 ^ 64, kb0 ^ 128, kb0 ^ 192};
  ka[0] = k_load(kbk, kp, 0);
  hook(-1);
  FL_SLOT_END();
  v16f acc;
  if constexpr (FMT == 0) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      // hipcc drains lgkmcnt to 0 before every inline-asm MFMA (it cannot count across asm): the operand reads of
      // later slots are therefore issued AFTER this slot's MFMA, so that a drain only ever waits for reads that are at
      // least one full slot old
      if (s == 0) mfma_rope_first(acc, ra[s], qr[s]);
      else mfma_rope(acc, ra[s], qr[s]);
      if (s < 2) ra[s + 2] = *reinterpret_cast<const uint4*>(rp + (lc.rb0 ^ ((s + 2) << 5)));
      if (s == 1) ka[1] = k_load(kbk, kp, 1);
      if (s == 3 && kAhead == 3) ka[2] = k_load(kbk, kp, 2);
      hook(s);
      FL_SLOT_END();
    }
  } else {
    ka[1] = k_load(kbk, kp, 1);
    if (kAhead == 3) ka[2] = k_load(kbk, kp, 2);
    mfma_fp8_first(acc, make_v8i(ra[0], ra[1]), qr8);
    hook(0); hook(1); hook(2); hook(3);
    FL_SLOT_END();
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    mfma_fp8(acc, make_v8i(ka[s % (kAhead + 1)].lo, ka[s % (kAhead + 1)].hi), qn[s]);
    if (s + kAhead < 8) ka[(s + kAhead) % (kAhead + 1)] = k_load(kbk, kp, s + kAhead);
    hook(4 + s);
    FL_SLOT_END();
  }
  FL_MFMA_DRAIN();   // acc is read by VALU instructions next
  return acc;
}

// HBM -> LDS DMA as inline asm.  Through the builtin, hipcc's waitcnt pass sees a FLAT instruction that touches LDS
// ("pending flat"): until a vmcnt(0) retires it, EVERY lgkmcnt wait is forced to 0 — and the page loop never drains
// vmcnt, so each ds_read wait also waited for the operand reads issued just before it (~140+ cycles, several times per
// stage).  The asm hides the DMA from the pass; its completion is certified by the explicit vmcnt waits + barrier of
// page_step.  M0 = LDS base of the wave's 1-KiB piece (lane i lands at base + 16 i / 4 i).
__device__ __forceinline__ void dma_x4(const uint8_t* sbase, const unsigned voff, const uint8_t* lds_dst) {
#ifdef FL_X_DMA_BUILTIN
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)(sbase + voff), (lds_ptr_t)(const_cast<uint8_t*>(lds_dst)), 16, 0, 0);
  return;
#endif
  const int la = __builtin_amdgcn_readfirstlane((int)(uintptr_t)lds_dst);   // low half of a generic LDS pointer = offset
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(la), "v"(voff), "s"(sbase)
               : "memory", "m0");
}
__device__ __forceinline__ void dma_x1(const float* sbase, const unsigned voff, const float* lds_dst) {
#ifdef FL_X_DMA_BUILTIN
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)(reinterpret_cast<const uint8_t*>(sbase) + voff), (lds_ptr_t)(const_cast<float*>(lds_dst)), 4, 0, 0);
  return;
#endif
  const int la = __builtin_amdgcn_readfirstlane((int)(uintptr_t)lds_dst);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(la), "v"(voff), "s"(sbase)
               : "memory", "m0");
}

// Copy of the lane constants that the optimiser cannot see through: everything derived from it stays where it is used
// instead of being hoisted (LICM) into dozens of long-lived registers.
__device__ __forceinline__ void opaque_inplace(LaneConst& lc) {
  asm volatile("" : "+v"(lc.lane), "+v"(lc.li), "+v"(lc.lh), "+v"(lc.kb0), "+v"(lc.rb0), "+v"(lc.vb0), "+v"(lc.dn_row),
               "+v"(lc.dn_x), "+v"(lc.dr[0]), "+v"(lc.dr[1]));
}
__device__ __forceinline__ LaneConst opaque(const LaneConst& in) {
  LaneConst lc = in;
  asm volatile("" : "+v"(lc.lane), "+v"(lc.li), "+v"(lc.lh), "+v"(lc.kb0), "+v"(lc.rb0), "+v"(lc.vb0), "+v"(lc.dn_row), "+v"(lc.dn_x), "+v"(lc.dr[0]),
               "+v"(lc.dr[1]));
  return lc;
}

// byte offset inside the page of this lane's 16 B of latent piece k of this wave (2 token rows of 512 B per piece,
// chunk c of token T stored at chunk c ^ (T & 15))
template <int FMT>
__device__ __forceinline__ unsigned dn_off(const LaneConst& lc, const int k) {
  constexpr unsigned kTokBytes = FMT == 0 ? kDN : kDN + kDR;
  return lc.dn_row + (unsigned)k * 2u * kTokBytes + (lc.dn_x ^ ((unsigned)k << 5));
}

// One pipeline step of one wave.  The first (no previous page) and last (no current page) steps are separate
// instantiations OUTSIDE the page loop; inside the loop there is exactly one variant — two variants joined inside the
// loop, or a run-time condition around the PV MFMAs, make hipcc copy all 256 O registers at the join:
//     QK block 0 of page i
//     QK block 1 of page i        ||  softmax of block 0 of page i
//     PV of page i-1 (16 MFMAs)   ||  softmax of block 1 of page i
// so the VALU / transcendental work of a block always sits in the shadow of independent MFMAs.  All LDS regions are
// distinct __restrict__ parameters of ONE inlined function (see mla_decode_fp8.hip: otherwise hipcc drains the LDS-DMA
// queue with vmcnt(0) before every ds_read).
template <int FMT, bool has_cur, bool has_prev, bool FAST, bool SELF_DMA>
__device__ __forceinline__ void page_step(
    ReqState& st, LaneConst& lc_io, const v8i (&qn)[8], const v8bf (&qr)[4], const v8i qr8, const float qs,
    const float ks_const, const int wave,
    // ---- page i: latent slot, rope slot, raw scales; wave-private triple scratch
    const uint8_t* __restrict__ k_nope, const uint8_t* __restrict__ k_rope, const float* __restrict__ k_scale_raw,
    float* __restrict__ scr,
    // ---- page i-1: latent slot (V^T), triple scratch (written by the previous step)
    const uint8_t* __restrict__ v_nope, const float* __restrict__ scr_prev,
    // ---- LDS regions filled by the DMA issued in this call (page i+2; never read in this call)
    uint8_t* __restrict__ dma_nope, uint8_t* __restrict__ dma_rope, float* __restrict__ dma_scale,
    const uint8_t* __restrict__ src_nope, const uint8_t* __restrict__ src_rope, const float* __restrict__ src_scale,
    // ---- geometry
    const int tok0, const int L, const int L_row, const bool need_mask_rt, const bool need_mask_prev_rt,
    const bool next_in_flight_rt FL_T_PARAMS) {
  constexpr int kRopePerWave = (FMT == 0 ? 8 : 4) / kNW;
  // FAST = steady-state step: pages i-1 and i lie fully inside every row's limit (no masks, no tail fill) and page i+2
  // exists (the refill is unconditional, page i+1 is in flight).  With these three conditions compiled out a stage is
  // ONE basic block, and hipcc's s_waitcnt lgkmcnt(N) are counted; with the wave-uniform branches in place it falls
  // back to lgkmcnt(0) at every block entry and the ~140-cycle LDS round trip is exposed half a dozen times per stage.
  const bool need_mask = FAST ? false : need_mask_rt;
  const bool need_mask_prev = FAST ? false : need_mask_prev_rt;
  const bool next_in_flight = FAST ? true : next_in_flight_rt;
  // the lane constants are made opaque IN PLACE: to the optimiser they are loop-carried values that change every step
  // (nothing derived from them is hoisted, and no per-step copies of the originals are kept alive beside them)
#if FL_X_INPLACE
  opaque_inplace(lc_io);
  const LaneConst& lc = lc_io;
#else
  const LaneConst lc = opaque(lc_io);
#endif
  const int lane = lc.lane, li = lc.li, lh = lc.lh;

  // ---- page i landed for every wave (issue order per step: rope, scale, latent of ONE page); every wave is done with
  //      page i-2, whose slots are refilled below ----
  // (SELF_DMA = false: loader waves issue and wait for the LDS-DMA, mla_decode_x_kernel; the barrier carries their wait)
#ifndef FL_X_NOWAIT   // experiment switch: timing without the page-landed wait (results are garbage)
  if (!SELF_DMA) {
  } else if (next_in_flight) {
    if constexpr (FMT == 0) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");   // page i+1: 2 rope + 1 scale + 8 latent
    else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
#endif
  FL_T(6);   // wait for page i
  __builtin_amdgcn_s_barrier();
  FL_T(0);   // barrier
  // The LDS-DMA refill (page i+2: 2 rope + 1 scale + 8 latent pieces per wave, in this order — the counted waits rely
  // on it) and the scale triples of page i are spread over the 12 MFMA slots of QK block 0: issued in one burst at the
  // top of the step they cost 100-185 cycles per piece (measured: 820 + most of a 2700-cycle QK stage).
  const bool do_dma = SELF_DMA && (FAST ? true : (src_nope != nullptr));
  auto dma_piece = [&](const int k) {   // k = 0 .. kRopePerWave + (FMT == 0) + kNopePerWave - 1
    if (k < kRopePerWave) {
      dma_x4(src_rope, lc.dr[k < kRopePerWave ? k : 0], dma_rope + (wave * kRopePerWave + k) * 1024);
    } else if (FMT == 0 && k == kRopePerWave) {
      dma_x1(src_scale, (unsigned)lane * 4u, dma_scale);
    } else {
      const int kk = k - kRopePerWave - (FMT == 0 ? 1 : 0);
      dma_x4(src_nope, dn_off<FMT>(lc, kk), dma_nope + (wave * kNopePerWave + kk) * 1024);
    }
  };
  constexpr int kPieces = kRopePerWave + (FMT == 0 ? 1 : 0) + kNopePerWave;   // 11 / 9: fits the 12 (9) slots
  if constexpr (!has_cur) {
    if (do_dma) {
#pragma unroll
      for (int k = 0; k < kPieces; ++k) dma_piece(k);
    }
  }

  // c1: block 1 of page i-1 — its first kC1UnderPV softmax pieces ran under the PV MFMAs of the previous step (state in
  // st.c1), the rest runs here under QK block 0.  c0: block 0 of page i.  c1n: block 1 of page i.
  Soft c0, c1 = st.c1, c1n;
  auto c1_piece = [&](const int k) {
    sm_piece(c1, k, scr_prev, 1, lh, qs, tok0 - kPage, L_row, need_mask_prev, st.mw[1], st.l[1], st.lq[1]);
  };
  auto c0_piece = [&](const int k) {
    sm_piece(c0, k, scr, 0, lh, qs, tok0, L_row, need_mask, st.mw[0], st.l[0], st.lq[0]);
  };
  auto c1n_piece = [&](const int k) {
    sm_piece(c1n, k, scr, 1, lh, qs, tok0, L_row, need_mask, st.mw[1], st.l[1], st.lq[1]);
  };
  if constexpr (has_cur) {
    float ks_raw = ks_const;
    if constexpr (FMT == 0) ks_raw = k_scale_raw[lane];   // consumed two slots later
    // tail of the sequence: zero the rows past the end (P' is exactly 0 there, but 0*NaN would poison the PV MFMA)
    if (!FAST && tok0 + kPage > L) {
      const int nvalid = L - tok0;
      uint8_t* wr = const_cast<uint8_t*>(k_nope);
#pragma clang loop vectorize(disable) unroll(disable)
      for (int T = nvalid + lh; T < kPage; T += 2)
        *reinterpret_cast<uint4*>(wr + T * kDN + li * 16) = make_uint4(0, 0, 0, 0);
    }
    FL_SLOT_END();
    FL_T(1);   // tail fill
    auto hook = [&](const int slot) {
      if (slot == 2) {
        // scale triples of the page (lane = token), wave-private scratch; first read by the softmax in QK block 1
        float ks = ks_raw;
        if (tok0 + lane >= L || !(ks > 0.f) || !(ks < 3.0e38f)) ks = 1.f;
        scr[lane] = ks;
        scr[kPage + lane] = __builtin_amdgcn_logf(ks);
        scr[2 * kPage + lane] = __builtin_amdgcn_rcpf(ks);
      }
      constexpr int kSlots = FMT == 0 ? 12 : 9;
      if (slot >= 0 && do_dma && slot < kSlots && slot < kPieces) dma_piece(slot);
      // remaining pieces of block 1 of page i-1 in the 64-cycle latent slots
      if constexpr (has_prev) {
        constexpr int kRest = kSoftPieces - kC1UnderPV;
        static_assert(kRest <= 8, "QK block 0 has 8 latent slots for the tail of block 1");
        constexpr int kStride = kRest <= 4 ? 2 : 1;
        if (slot >= 4 && (slot - 4) % kStride == 0 && (slot - 4) / kStride < kRest)
          c1_piece(kC1UnderPV + (slot - 4) / kStride);
      }
    };
    // ---- A. QK block 0 of page i || tail of the softmax of block 1 of page i-1 ----
    c0.y = qk_stage<FMT>(lc, k_nope, k_rope, 0, qn, qr, qr8, hook);
    FL_SLOT_END();
    FL_T(2);   // QK block 0 || softmax 1 of the previous page
  } else {
#pragma unroll
    for (int k = kC1UnderPV; k < kSoftPieces; ++k) c1_piece(k);
    FL_SLOT_END();
  }

  // ---- page i-1 is complete: fix / check the O reference and build the PV operands BEFORE the softmax of page i moves
  //      mw[0].  The reference of O is fixed when a row sees its first valid token and NEVER moves in this pass: later
  //      blocks with a larger reference m_b enter with an E8M0 block scale 2^(m_b - mo) > 1 (exact; fp32 O has the
  //      range).  Only a reference more than kMaxUp above mo (a logit that beats the row's first-page maximum by > 69
  //      nats) cannot be represented: it raises st.redo and the kernel repeats the request with mo preset. ----
  v8i pb = v8i{0, 0, 0, 0, 0, 0, 0, 0};
  int sb = 0;
  if constexpr (has_prev) {
    const float mw_max = fmaxf(st.mw[0], st.mw[1]);
    st.mo = st.mo > kNegRef ? st.mo : mw_max;
    st.redo |= (mw_max - st.mo > kMaxUp) ? 1 : 0;
    sb = 127 + (int)fminf((lh ? st.mw[1] : st.mw[0]) - st.mo, kMaxUp);
    sb = sb < 0 ? 0 : sb;
    pb = make_v8i(st.p0, make_uint4(c1.pk[0], c1.pk[1], c1.pk[2], c1.pk[3]));
  }
  // ---- B. QK block 1 of page i || first kC0UnderQK pieces of the softmax of block 0 ----
  if constexpr (has_cur) {
    c1n.y = qk_stage<FMT>(lc, k_nope, k_rope, 1, qn, qr, qr8, [&](const int slot) {
      if (slot + 1 < kC0UnderQK) c0_piece(slot + 1);
    });
    FL_SLOT_END();
    FL_T(3);   // QK block 1 || softmax 0
  }

  // ---- C. O^T += V^T(i-1) · P^T(i-1) || rest of block 0 and first kC1UnderPV pieces of block 1 of page i.  Nothing but
  //      the MFMA touches O in the page loop. ----
  auto pv_piece = [&](const int t) {   // t = 0 .. 15
    constexpr int kC0Rest = kSoftPieces - kC0UnderQK;
    if (t < kC0Rest) c0_piece(kC0UnderQK + t);
    else if (t - kC0Rest < kC1UnderPV) c1n_piece(t - kC0Rest);
  };
  static_assert(kSoftPieces - kC0UnderQK + kC1UnderPV <= 16, "PV has 16 slots");
  if constexpr (has_prev) {
    v8i va[4];
    va[0] = vt_load(lc, v_nope, 0);
    va[1] = vt_load(lc, v_nope, 1);
    va[2] = vt_load(lc, v_nope, 2);
    FL_SLOT_END();
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if (t + 3 < 16) va[(t + 3) % 4] = vt_load(lc, v_nope, t + 3);
      st.o[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va[t % 4], pb, st.o[t], 0, 0, 0, kUnitScale, 0, sb);
      if constexpr (has_cur) pv_piece(t);
      FL_SLOT_END();
    }
  } else if constexpr (has_cur) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      pv_piece(t);
      FL_SLOT_END();
    }
  }
  if constexpr (has_cur) {
    st.p0 = make_uint4(c0.pk[0], c0.pk[1], c0.pk[2], c0.pk[3]);
    st.c1 = c1n;
  }
  FL_T(4);   // PV
}

// NWC compute waves of 32 query rows each.  NWC = 4: every wave also issues its share of the LDS-DMA inside its MFMA
// slots.  NWC = 2 (33..64 query rows per request, e.g. MTP draft steps at H = 64; a single compute wave — NWC = 1 —
// is slower than the two half-waves of mla_decode_fp8.hip):
// the CU's remaining SIMDs run two LOADER waves that issue every refill and keep the barrier protocol — an LDS-DMA piece
// costs 60-185 issue cycles, ~1100 of a compute wave's ~4400 cycles per page.
template <int FMT, int NWC>
__global__ __launch_bounds__(64 * (NWC + (NWC == 4 ? 0 : 2)), 1) void mla_decode_x_kernel(
    const Params p, const int32_t* __restrict__ g_block_table, const int32_t* __restrict__ g_seqlens,
    const int32_t* __restrict__ g_meta, const int32_t* __restrict__ g_num_splits,
    const uint8_t* __restrict__ g_k_nope, const uint16_t* __restrict__ g_k_rope, const float* __restrict__ g_k_scale,
    const uint8_t* __restrict__ g_q_nope, const uint16_t* __restrict__ g_q_rope, const float* __restrict__ g_q_scale) {
  constexpr int kTokBytes = FMT == 0 ? kDN : kDN + kDR;    // bytes per token row of the latent tensor in HBM
  constexpr int kRopeTok = FMT == 0 ? kDR * 2 : kDR;        // bytes per token of rope (bf16 / fp8)
  constexpr bool SELF_DMA = NWC == 4;
  constexpr int kDW = SELF_DMA ? 4 : 2;                       // waves that issue LDS-DMA
  constexpr int kNopePW = kDmaNopePerTile / kDW;              // latent pieces per DMA wave per page (8 / 16)
  constexpr int kRopePerWave = (FMT == 0 ? 8 : 4) / kDW;      // rope pieces per DMA wave per page
  static_assert(SELF_DMA || kRopePerWave <= 4, "loader rope pieces");
  __shared__ __attribute__((aligned(16))) uint8_t smem[kLdsBytes];

  const int tid = threadIdx.x;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = !SELF_DMA && wave_id >= NWC;
  const int wave = is_loader ? wave_id - NWC : wave_id;       // compute wave index, or DMA wave index of a loader
  LaneConst lc0;
  lc0.lane = tid & 63;
  lc0.li = lc0.lane & 31;
  lc0.lh = lc0.lane >> 5;
  lc0.dr[0] = lc0.dr[1] = 0;
  {
    // operand layouts and swizzles: identical to mla_decode_fp8.hip (derivations there)
    LaneConst& lc = lc0;
    const int li = lc.li, lh = lc.lh, lane = lc.lane;
    const int kx = li & 15;
    lc.kb0 = li * kDN + (((((kx >> 2)) << 2) | ((2 * lh) ^ (kx & 3))) << 4);
    if constexpr (FMT == 0) lc.rb0 = li * (kDR * 2) + (((lh ^ ((li >> 1) & 7))) << 4);
    else lc.rb0 = li * kDR + ((((2 * lh) ^ ((li >> 2) & 3))) << 4);
    const int s16 = lane & 15;
    const int gi = (lane >> 4) & 1;
    const int tj = s16 >> 1;
    const int tok_in8 = (tj & 3) + ((tj >> 2) << 3);
    const int vrow = 4 * lh + tok_in8;
    lc.vb0 = vrow * kDN + ((((gi << 2)) ^ (vrow & 15)) << 4) + (s16 & 1) * 8;
    // latent DMA piece k of this wave: token row T = (wave*8 + k)*2 + lh, chunk li stored from source chunk li ^ (T&15)
    lc.dn_row = (unsigned)((wave * kNopePerWave * 2 + lh) * kTokBytes);
    lc.dn_x = (unsigned)((li ^ lh) << 4);
#pragma unroll
    for (int k = 0; k < (SELF_DMA ? kRopePerWave : 0); ++k) {   // BYTE offsets inside the page's rope block
      if constexpr (FMT == 0) {
        const int T = (wave * kRopePerWave + k) * 8 + (lane >> 3);   // 8 token rows of 128 B per piece
        lc.dr[k] = (unsigned)(T * 128 + (((lane & 7) ^ ((T >> 1) & 7)) << 4));
      } else {
        const int T = (wave * kRopePerWave + k) * 16 + (lane >> 2);  // 16 token rows of 64 B per piece
        lc.dr[k] = (unsigned)(T * kTokBytes + kDN + (((lane & 3) ^ ((T >> 2) & 3)) << 4));
      }
    }
  }
  const int lane = lc0.lane, li = lc0.li, lh = lc0.lh;

  // ---- workgroup -> (part, row group) ----
  // The row groups of one part read the same KV pages: they are placed on the SAME XCD (consecutive workgroup ids go
  // round-robin over the 8 XCDs, each with its own L2), so that the pages come from HBM once per part, not once per
  // row group (s_q = 4 verify at H = 128: 4 row groups).
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

#ifdef FL_MLA_TIMING
  unsigned long long tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
  unsigned long long tvar = tlast;
  const unsigned long long tstart = tlast;
#endif
  if (is_loader) {
    // ---- loader waves (NWC < 4): the request loop of the compute waves reduced to its barriers and its LDS-DMA.  Piece
    //      mapping as in the compute waves of NWC = 4, with 16 latent + 4 (2) rope + 1 scale pieces per loader and page;
    //      issue order per page: rope, scale, latent (the counted waits rely on it). ----
    const unsigned ldn_row = (unsigned)((wave * kNopePW * 2 + lh) * kTokBytes);
    const unsigned ldn_x = (unsigned)((li ^ lh) << 4);
    unsigned ldr[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < kRopePerWave; ++k) {
      if constexpr (FMT == 0) {
        const int T = (wave * kRopePerWave + k) * 8 + (lane >> 3);
        ldr[k] = (unsigned)(T * 128 + (((lane & 7) ^ ((T >> 1) & 7)) << 4));
      } else {
        const int T = (wave * kRopePerWave + k) * 16 + (lane >> 2);
        ldr[k] = (unsigned)(T * kTokBytes + kDN + (((lane & 3) ^ ((T >> 2) & 3)) << 4));
      }
    }
    constexpr int kPiecesL = kRopePerWave + (FMT == 0 ? 1 : 0) + kNopePW;   // 21 / 18
    static_assert(SELF_DMA || kPiecesL == (FMT == 0 ? 21 : 18), "vmcnt immediates below");
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
      auto page_of = [&](int t) { return (long long)__builtin_amdgcn_readlane(pg_vec, t - win_base); };
      auto issue_page = [&](const int t) {
        const long long pg = page_of(t);
        const uint8_t* sn = g_k_nope + pg * (kPage * kTokBytes);
        const uint8_t* sr = FMT == 0 ? reinterpret_cast<const uint8_t*>(g_k_rope) + pg * (kPage * kRopeTok) : sn;
        uint8_t* dr = smem + kOffRope + (t % kRopeSlots) * kRopeBytes;
#pragma unroll
        for (int k = 0; k < kRopePerWave; ++k) dma_x4(sr, ldr[k], dr + (wave * kRopePerWave + k) * 1024);
        if constexpr (FMT == 0)
          dma_x1(g_k_scale + pg * kPage, (unsigned)lane * 4u,
                 reinterpret_cast<float*>(smem + kOffScale + (t % kRopeSlots) * (kPage * 4)));
        uint8_t* dn = smem + kOffRing + (t & 3) * kSlotBytes;
#pragma unroll
        for (int k = 0; k < kNopePW; ++k)
          dma_x4(sn, ldn_row + (unsigned)k * 2u * kTokBytes + (ldn_x ^ (((unsigned)k & 7u) << 5)), dn + (wave * kNopePW + k) * 1024);
      };
      load_window(0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // request start
      for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
          load_window(0);
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
        if (n > 0) issue_page(0);
        if (n > 1) issue_page(1);
        if (n > 0) {
          for (int i = 0; i <= n; ++i) {   // the n + 1 pipeline steps of the compute waves
            if (i + 2 >= win_base + 64 && i + 2 < n) load_window(i + 2);
            if (i > 0 && i + 1 < n) {      // page i has landed, page i + 1 may stay in flight
              if constexpr (FMT == 0) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
              else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (i + 2 < n) issue_page(i + 2);
          }
        }
        if (pass == 1) break;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // redo vote
        const int* flag = reinterpret_cast<const int*>(smem + kOffFlag);
        int any_redo = 0;
#pragma unroll
        for (int w = 0; w < NWC; ++w) any_redo |= flag[w];
        if (!any_redo) break;
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();     // epilogue
    }
    return;
  }
  const int row = rgrp * (32 * NWC) + wave * 32 + li;   // query row of this lane
  const bool row_ok = row < p.rows;

  for (; req < p.bs; ++req, tile_b = 0, split_idx = 0) {
    if (req > end_req || (req == end_req && end_tile == 0)) break;
    const int L = g_seqlens[req];
    const int nt = L > 0 ? (L + kPage - 1) / kPage : 0;
    int tile_e = req < end_req ? nt : (end_tile < nt ? end_tile : nt);
    if (tile_e < tile_b) tile_e = tile_b;
    const int n = tile_e - tile_b;
    const int split_base = g_num_splits[req];
    const bool is_split = (g_num_splits[req + 1] - split_base) > 1;

    // page ids of a 64-page window live in ONE VGPR (lane j = page win_base + j); a lookup is a v_readlane
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

    // every wave finished with the LDS of the previous request (and its stores left the vmcnt queue)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    FL_T(8);   // prologue: scheduler row, lengths, page window

    auto src_nope_of = [&](int t) { return g_k_nope + page_of(t) * (kPage * kTokBytes); };
    auto src_rope_of = [&](int t) {
      return FMT == 0 ? reinterpret_cast<const uint8_t*>(g_k_rope) + page_of(t) * (kPage * kRopeTok)
                      : g_k_nope + page_of(t) * (kPage * kTokBytes);
    };
    auto src_scale_of = [&](int t) { return FMT == 0 ? g_k_scale + page_of(t) * kPage : g_k_scale; };
    auto ring = [&](int t) { return smem + kOffRing + (t & 3) * kSlotBytes; };
    auto rope_slot = [&](int t) { return smem + kOffRope + (t % kRopeSlots) * kRopeBytes; };
    auto scale_slot = [&](int t) { return reinterpret_cast<float*>(smem + kOffScale + (t % kRopeSlots) * (kPage * 4)); };
    auto scr = [&](int t) {
      return reinterpret_cast<float*>(smem + kOffScratch + ((t & 1) * NWC + wave) * kScratchPerWave);
    };
    // ---- prologue: pages 0 and 1 (issue order per page: rope, scale, latent — the counted waits rely on it), issued
    //      BEFORE the Q loads and the O initialisation so that their HBM latency overlaps; step 0 waits with vmcnt(0) ----
    auto dma_page = [&](int t) {
      const LaneConst lc = opaque(lc0);
      const uint8_t* sr = src_rope_of(t);
#pragma unroll
      for (int k = 0; k < kRopePerWave; ++k)
        dma_x4(sr, lc.dr[k], rope_slot(t) + (wave * kRopePerWave + k) * 1024);
      if constexpr (FMT == 0)
        dma_x1(src_scale_of(t), (unsigned)lane * 4u, scale_slot(t));
      const uint8_t* sn = src_nope_of(t);
#pragma unroll
      for (int k = 0; k < kNopePerWave; ++k)
        dma_x4(sn, dn_off<FMT>(lc, k), ring(t) + (wave * kNopePerWave + k) * 1024);
    };
    if constexpr (SELF_DMA) {
      if (n > 0) dma_page(0);
      if (n > 1) dma_page(1);
    }
    FL_T(9);   // prologue: DMA issue of pages 0 and 1

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
    if (row_ok) {
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

    FL_T(10);  // prologue: Q loads
    ReqState st;
    float mo_preset = kNegRef;
    // pass 0 fixes the O reference at each row's first valid page; pass 1 runs only if some block reference outran it
    // by more than kMaxUp (page_step), with the reference preset to the final one
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st.o[j][r] = 0.f;
      FL_SLOT_END();
    }
    st.l[0] = st.l[1] = st.lq[0] = st.lq[1] = 0.f;
    st.mw[0] = st.mw[1] = kNegRef;
    st.mo = mo_preset;
    st.redo = 0;
    st.p0 = make_uint4(0, 0, 0, 0);
    st.c1 = Soft{};

    if (pass == 1) {   // (pass 0 issued its prologue before the Q loads)
      load_window(0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if constexpr (SELF_DMA) {
        if (n > 0) dma_page(0);
        if (n > 1) dma_page(1);
      }
    }

    // ---- n + 1 pipeline steps: step i = QK(i), softmax(i) || PV(i-1) ----
#define FL_STEP(HC, HP, FA)                                                                                               \
  {                                                                                                                    \
    if (i + 2 >= win_base + 64 && i + 2 < n) load_window(i + 2); /* pages i+2 .. i+65 */                               \
    const uint8_t* sn = nullptr;                                                                                       \
    const uint8_t* sr = nullptr;                                                                                       \
    const float* ss = nullptr;                                                                                         \
    if (i + 2 < n) {                                                                                                   \
      sn = src_nope_of(i + 2);                                                                                         \
      sr = src_rope_of(i + 2);                                                                                         \
      ss = src_scale_of(i + 2);                                                                                        \
    }                                                                                                                  \
    const int tok0 = (tile_b + i) * kPage;                                                                             \
    page_step<FMT, HC, HP, FA, SELF_DMA>(st, lc0, qn, qr, qr8, qs, ks_const, wave, ring(i), rope_slot(i), scale_slot(i), scr(i),     \
                           ring(i + 3), scr(i + 1), ring(i + 2), rope_slot(i + 2), scale_slot(i + 2), sn, sr, ss, tok0, \
                           L, L_row, tok0 + kPage > L_min, tok0 > L_min, i > 0 && i + 1 < n FL_T_ARGS);                \
  }
    FL_T(7);   // request prologue (window, DMA of pages 0/1, Q loads, O init)
    if (n > 0) {
      int i = 0;
      FL_TV(0);
      FL_STEP(true, false, false);
      FL_TV(12);   // whole first step
      // steady state: pages i-1, i unmasked for every row and page i+2 present
      int n_fast = L_min / kPage - tile_b;
      n_fast = n_fast < n - 2 ? n_fast : n - 2;
      for (i = 1; i < n_fast; ++i) FL_STEP(true, true, true);
      FL_TV(13);   // all FAST steps
      for (; i < n; ++i) FL_STEP(true, true, false);
      FL_TV(14);   // generic steps
      FL_STEP(false, true, false);
      FL_TV(15);   // last step
    }
#undef FL_STEP
    if (pass == 1) break;
    {
      // workgroup-uniform decision (the page loop has workgroup barriers)
      int* flag = reinterpret_cast<int*>(smem + kOffFlag);
      const int vote = __any(st.redo != 0) ? 1 : 0;
      if (lane == 0) flag[wave] = vote;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      int any_redo = 0;
#pragma unroll
      for (int w = 0; w < NWC; ++w) any_redo |= flag[w];
      if (!any_redo) break;
      mo_preset = fmaxf(st.mw[0], st.mw[1]);
    }
    }   // pass

    // ---- per-request epilogue: merge the normalisers of the two blocks and lane halves, normalise, store ----
    FL_MFMA_DRAIN();
    const float f0 = __builtin_amdgcn_exp2f(st.mw[0] - st.mo), f1 = __builtin_amdgcn_exp2f(st.mw[1] - st.mo);
    float l = st.l[0] * f0 + st.l[1] * f1;
    float lq = st.lq[0] * f0 + st.lq[1] * f1;
    l += __shfl_xor(l, 32);
    lq += __shfl_xor(lq, 32);
    const float inv = lq > 0.f ? 1.f / lq : 0.f;
    const float lse_nat = l > 0.f ? (__builtin_amdgcn_logf(l) + st.mo - kPShift) * 0.6931471805599453f : -INFINITY;
    // split-KV partials are normalised by lq, so they are COMBINED with lq-based weights; the exact LSE travels along
    const float lseq_nat = lq > 0.f ? (__builtin_amdgcn_logf(lq) + st.mo - kPShift) * 0.6931471805599453f : -INFINITY;
    if (row_ok) {
      const int slot_idx = split_base + split_idx;
      if (lh == 0) {
        if (is_split) {
          p.lse_accum[((long long)slot_idx * p.rows + row) * 2 + 0] = lseq_nat;
          p.lse_accum[((long long)slot_idx * p.rows + row) * 2 + 1] = lse_nat;
        } else {
          const int j = row / p.h_q, h = row - j * p.h_q;
          p.lse[((long long)req * p.h_q + h) * p.s_q + j] = lse_nat;
        }
      }
    }
    // O -> memory through a wave-private LDS transpose (the ring is free now).  A lane holds ONE row, 4 dims at a time:
    // stored directly, every store instruction would touch 64 rows x 16 B (store-issue-bound, measured 26k cycles per
    // workgroup).  Tiles 4c..4c+3 cover the contiguous dims [128c, 128c+128) (C row i = e + 8g + 4lh of tile 4c + jq is
    // d = 128c + 16jq + (i&15) + 64(i>>4)), so chunk c is staged as [32 rows][128 f32] (+4 pad) and leaves as full
    // 256-B bf16 row segments (final output, or the split partial of this part).
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave is past its last K / V^T read
    FL_T(11);  // epilogue: normalisers, LSE, barrier
    {
      constexpr int kStgStride = 128 + 4;   // floats per staged row
      // (opaque lane id: keeps the store addresses from being hoisted out of the request loop and spilled)
      int lane = lc0.lane;
      asm volatile("" : "+v"(lane));
      const int li = lane & 31, lh = lane >> 5;
      float* stg = reinterpret_cast<float*>(smem + kOffRing) + wave * (32 * kStgStride);
      const int row0 = rgrp * (32 * NWC) + wave * 32;
      const int slot_idx = split_base + split_idx;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int d_off = 16 * jq + 8 * (g & 1) + 4 * lh + 64 * (g >> 1);
            *reinterpret_cast<float4*>(stg + li * kStgStride + d_off) =
                make_float4(st.o[4 * c + jq][g * 4 + 0] * inv, st.o[4 * c + jq][g * 4 + 1] * inv,
                            st.o[4 * c + jq][g * 4 + 2] * inv, st.o[4 * c + jq][g * 4 + 3] * inv);
          }
          FL_SLOT_END();
        }
        {
          // split parts write the same bf16 row segments into their o_accum slot (read back by the combine kernel)
          uint16_t* dst = is_split ? reinterpret_cast<uint16_t*>(p.o_accum) + ((long long)slot_idx * p.rows + row0) * kDN
                                   : p.out + ((long long)req * p.rows + row0) * kDN;
          uint16_t* dbase = dst + 128 * c + (lane & 15) * 8;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int r = (lane >> 4) + 4 * k;
            const float4 v0 = *reinterpret_cast<const float4*>(stg + r * kStgStride + (lane & 15) * 8);
            const float4 v1 = *reinterpret_cast<const float4*>(stg + r * kStgStride + (lane & 15) * 8 + 4);
            uint4 o;   // v_cvt_pk_bf16_f32: RNE, two values per instruction
            o.x = fl_pack_bf16(v0.x, v0.y);
            o.y = fl_pack_bf16(v0.z, v0.w);
            o.z = fl_pack_bf16(v1.x, v1.y);
            o.w = fl_pack_bf16(v1.z, v1.w);
            if (row0 + r < p.rows) *reinterpret_cast<uint4*>(dbase + (long long)r * kDN) = o;
          }
        }
        FL_SLOT_END();
      }
    }
#ifdef FL_MLA_TIMING
    FL_T(5);   // epilogue
#endif
  }
#ifdef FL_MLA_TIMING
  if (g_dbg_x != nullptr && lane == 0) {
    unsigned long long* d = reinterpret_cast<unsigned long long*>(g_dbg_x) + ((long long)blockIdx.x * 4 + wave) * 18;
    for (int i = 0; i < 16; ++i) d[i] = tacc[i];
    d[16] = __builtin_readcyclecounter() - tstart;
    d[17] = tlast - tstart;
  }
#endif
}

}  // namespace

int fl_mla_decode_fp8_x_impl(const FlMlaDecodeArgs* a, const Params& p_in, hipStream_t stream) {
  Params p = p_in;
  // compute waves per workgroup from the query rows per request (fl_mla_num_parts sizes the part count with the same rule)
  const int nwc = (p.rows <= 64 || fl_mla_x_rows_per_wg() == 64) ? 2 : 4;   // (at most 32 rows: mla_decode_fp8.hip, see its dispatcher)
  p.row_groups = (p.rows + 32 * nwc - 1) / (32 * nwc);
  const dim3 grid((unsigned)(p.num_parts * p.row_groups)), block(64 * (nwc + (nwc == 4 ? 0 : 2)));
  p.partial_bf16 = 1;   // split partials travel as bf16 rows (half the bytes of the f32 layout of mla_decode_fp8.hip)
#define FL_X_LAUNCH(FMT_, NWC_)                                                                                        \
  mla_decode_x_kernel<FMT_, NWC_><<<grid, block, 0, stream>>>(                                                         \
      p, a->block_table, a->cache_seqlens, a->tile_scheduler_metadata, a->num_splits, (const uint8_t*)a->k_nope,       \
      (const uint16_t*)a->k_rope, a->k_scale, (const uint8_t*)a->q_nope, (const uint16_t*)a->q_rope, a->q_scale)
  if (a->kv_format == FL_KV_FP8_PER_TOKEN) {
    if (nwc == 4) FL_X_LAUNCH(0, 4); else FL_X_LAUNCH(0, 2);
  } else {
    if (nwc == 4) FL_X_LAUNCH(1, 4); else FL_X_LAUNCH(1, 2);
  }
#undef FL_X_LAUNCH
  FL_CHECK_LAUNCH("mla_decode_x_kernel");
  // (an in-kernel merge by the last-arriving part was measured: +77 us — one workgroup per request merging row by row is
  //  latency-bound, and the bytes are the same; the combine kernel spreads them over the whole chip)
  return fl_mla_launch_combine(p, a->num_splits, stream);
}

#ifdef FL_MLA_TIMING
extern "C" int fl_mla_debug_set_buffer_x(int* dev_ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_x), &dev_ptr, sizeof(dev_ptr));
}
#endif
