// K1 for more than 32 query rows per request — paged MLA decode over the per-token-FP8 latent KV cache with
// ROLE-SPECIALISED waves, gfx950 (MI355X) only.  Same math, data formats and call sites as mla_decode_fp8.hip
// (flash_mla_fp8.flash_mla_ckv_fp8_per_token, flashmla_backend.py:208-222 decode / :127-142 verify).
//
// Why a third mapping.  One wave per SIMD issues in order: in mla_decode_fp8_x.hip (128 rows, 4 waves x 512 registers) a
// page costs ~4400 issue cycles against 2304 cycles of matrix pipe, and the whole batch had to be split along KV (bf16
// partials + a merge kernel: 31 of 118 us at bs=128, seq=4096, H=128).  Here a workgroup owns 64 query rows and runs
// EIGHT waves, two per SIMD, at <= 256 registers each:
//   * QK waves q(rt, W) (wave ids 0..3): S^T[32 tok x 32 rows] = K[32W..] . Q_rt^T (4 bf16 rope + 8 MX-fp8 MFMAs), the
//     online softmax of that block on y = s*log2e + log2 k_scale[t] with an integer reference per (row, block), the
//     e4m3 weights P' = 2^(y - m + 8) and both normalisers; P' (16 B per lane) and the reference go to LDS.
//   * PV waves v(rt, Wd) (wave ids 4..7, on the SAME SIMD as q(rt', W') with the same low id bits): one step later,
//     O^T[256 dims of half Wd x 32 rows] += V^T . P^T (8 MX-fp8 MFMAs, K = 64 tokens; the two blocks' references are
//     reconciled by the E8M0 block scales), V^T by ds_read_b64_tr_b8 from the same LDS bytes the QK waves read as K;
//     they also issue every LDS-DMA refill (8 pieces of 1 KiB per wave and page) and zero the tail of the last page.
//   so the softmax VALU work of one wave runs beside the other wave's MFMAs on each SIMD, nobody holds both Q (80
//   registers) and O (128), and a 128-head request is two neighbouring workgroups of one XCD that walk ALL its pages
//   (the second reader of a page hits the XCD's L2): no KV split, no partials and no merge kernel for a full batch.
//   * LDS: 4-slot ring of 32 KiB latent pages (page i-1 read as V^T, page i as K, pages i+1, i+2 landing) + double-
//     buffered P' / references + wave-private scale scratch.  Rope (bf16, 4 KiB per QK wave and page) and the raw
//     k_scale go global -> registers in the QK waves, two pages ahead (a rope ring does not fit beside the P buffers).
//   * ONE s_barrier per page step: step i = QK(i) + softmax(i) || PV(i-1) + refill of page i+2.
// Split requests (fewer requests than parts) write normalised bf16 partial rows + {weight-LSE, exact-LSE} exactly like
// mla_decode_fp8_x.hip and are merged by mla_combine_kernel.
#include "mla_decode_shared.h"

using namespace fl_mla;

namespace {

constexpr int kOffRing = 0;                                        // 4 x 32 KiB latent pages
constexpr int kRopeSlots = 3;
constexpr int kOffRope = kOffRing + kRingSlots * kSlotBytes;       // 3 x 8 KiB rope pages (bf16)
constexpr int kOffPbuf = kOffRope + kRopeSlots * kRopeBytes;       // [rt 2][W 2][64 lanes][16 B]: P' of ONE page
constexpr int kPbufBytes = 2 * 2 * 64 * 16;
constexpr int kOffRef = kOffPbuf + kPbufBytes;                     // [rt 2][W 2][32 rows] f32: block references
constexpr int kRefBytes = 2 * 2 * 32 * 4;
constexpr int kScratchPerWave = 3 * 32 * 4;                        // {ks, log2 ks, 1/ks} x 32 tokens
constexpr int kOffScratch = kOffRef + kRefBytes;                   // [QK wave 4]
constexpr int kOffLm = kOffScratch + 4 * kScratchPerWave;          // [rt 2][W 2][3][32] f32: l, lq, m per row
constexpr int kOffFlag = kOffLm + 2 * 2 * 3 * 32 * 4;              // 4 ints: redo votes of the PV waves
constexpr int kLdsBytes = kOffFlag + 16;
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
constexpr int kStgStride = 128 + 4;                                // epilogue staging: floats per row (in the ring)
static_assert(4 * 32 * kStgStride * 4 <= kRingSlots * kSlotBytes, "epilogue staging fits the ring");

constexpr int kPiecesPerWave = kDmaNopePerTile / 4;                // 8 latent LDS-DMA pieces of 1 KiB per PV wave and page
constexpr int kRopePiecesPerWave = 2;                              // + 2 rope pieces (8 token rows of 128 B each)
constexpr int kDmaPerWavePage = kPiecesPerWave + kRopePiecesPerWave;   // 10: the counted vmcnt waits rely on it

#ifndef FL_Y_NOWAIT
#define FL_Y_NOWAIT 0
#endif
#ifdef FL_MLA_TIMING
__device__ int* g_dbg_y = nullptr;   // debug builds only: set by fl_mla_debug_set_buffer_y
#define FL_T(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); tacc[i] += t__ - tlast; tlast = t__; } while (0)
#define FL_T_PARAMS , unsigned long long (&tacc)[12], unsigned long long& tlast
#define FL_T_ARGS , tacc, tlast
#else
#define FL_T(i) do { } while (0)
#define FL_T_PARAMS
#define FL_T_ARGS
#endif


struct QkLane {
  int rb0;   // rope operand: byte offset inside a rope slot for k-step 0 (k-step s: ^ (s << 5)); + W*4096
  int kb0;   // K operand: byte offset inside a slot for k-step 0, first 16 B (second: ^16); k-step s: ^ ((s&3) << 6), + (s>>2)*256
};
struct PvLane {
  int vb0;          // V^T tr8 source of tile jb = 0 (tile jb: ^ ((jb&3) << 4) ^ ((jb>>2) << 7)); + u immediates; + Wd*256
  unsigned dn_row;  // latent DMA: byte offset of this lane's token row of piece 0 of this wave (piece k: + k * 1024)
  unsigned dn_x;    // ... and its swizzled 16-B chunk (piece k: ^ (k << 5))
  unsigned dr[kRopePiecesPerWave];   // rope DMA: byte offset of this lane's 16 B inside the page's rope block
};
// Four consecutive 1-KiB LDS-DMA pieces behind ONE M0 write: the instruction's immediate offset is added to the global
// address AND to the LDS address (M0 + offset + 16 * lane), so pieces whose source and destination both advance by 1 KiB
// only differ in that immediate (13-bit signed: 0..3 KiB) and in the swizzle of the per-lane offset.
template <int K0, int N>
__device__ __forceinline__ void dma_group(const uint8_t* sbase, const unsigned voff0, const unsigned (&xorv)[4],
                                          const uint8_t* lds_dst) {
  const int la = __builtin_amdgcn_readfirstlane((int)(uintptr_t)lds_dst);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(la) : "memory", "m0");
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const unsigned v = voff0 ^ xorv[k];
    if (k == 0) asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(v), "s"(sbase) : "memory", "m0");
    if (k == 1) asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" ::"v"(v), "s"(sbase) : "memory", "m0");
    if (k == 2) asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" ::"v"(v), "s"(sbase) : "memory", "m0");
    if (k == 3) asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" ::"v"(v), "s"(sbase) : "memory", "m0");
  }
}
// all DMA pieces of one page for one PV wave: 2 rope pieces + 8 latent pieces (kDmaPerWavePage = 10 instructions)
struct PvLane;
__device__ __forceinline__ void dma_page_pieces(const PvLane& lc, const uint8_t* src_rope, uint8_t* dst_rope,
                                                const uint8_t* src_nope, uint8_t* dst_nope);
// byte offset inside the page of this lane's 16 B of latent piece k of this wave (2 token rows of 512 B per piece, chunk c
// of token T stored at chunk c ^ (T & 15))
__device__ __forceinline__ unsigned dn_off(const PvLane& lc, const int k) {
  return lc.dn_row + (unsigned)k * 1024u + (lc.dn_x ^ ((unsigned)k << 5));
}
__device__ __forceinline__ void dma_page_pieces(const PvLane& lc, const uint8_t* src_rope, uint8_t* dst_rope,
                                                const uint8_t* src_nope, uint8_t* dst_nope) {
  // rope piece k: source offset dr[0] ^ (k << 6), + k KiB (the row swizzle (T >> 1) & 7 flips bit 2 with k)
  const unsigned xr[4] = {0u, 64u, 0u, 0u};
  dma_group<0, kRopePiecesPerWave>(src_rope, lc.dr[0], xr, dst_rope);
  // latent piece k: source offset dn_row + (dn_x ^ (k << 5)), + k KiB (dn_row is a multiple of 512: the XOR of the
  // 16-B chunk bits 5..7 commutes with the sum)
  const unsigned voff = lc.dn_row + lc.dn_x;
  const unsigned x0[4] = {0u, 32u, 64u, 96u}, x1[4] = {128u, 160u, 192u, 224u};
  dma_group<0, 4>(src_nope, voff, x0, dst_nope);
  dma_group<4, 4>(src_nope + 4096, voff, x1, dst_nope + 4096);
}

// scale triples {ks, log2 ks, 1/ks} of a QK wave's 32 tokens (lane li = token 32W + li) -> wave-private scratch
__device__ __forceinline__ void scale_prep(float* __restrict__ scratch, float ks, const int tok0w, const int li, const int L) {
  if (tok0w + li >= L || !(ks > 0.f) || !(ks < 3.0e38f)) ks = 1.f;
  // (both lane halves store the same values to the same addresses: no exec-mask branch)
  scratch[li] = ks;
  scratch[32 + li] = __builtin_amdgcn_logf(ks);
  scratch[64 + li] = __builtin_amdgcn_rcpf(ks);
}

// ---- QK wave: one page step.  Beside a running MFMA every other instruction of the SIMD is (almost) free, and a wave
//      alone issues in order: the 12-deep dependent MFMA chain of a block blocks its wave for ~800 cycles, and the ~130
//      VALU instructions of a block's softmax cost ~5 cycles each behind it (measured: 2700 cycles per step, the matrix
//      pipe 35 % busy).  So a step is software-pipelined across TWO pages and split by a second barrier:
//        B_i   page i landed
//              QK chain of page i (4 bf16 rope + 8 MX-fp8 MFMAs into acc_cur), one SLOT per MFMA; in the slots: the softmax of
//              page i-1 (its S^T waits in y_prev since the previous step) up to the P' / reference store, the operand
//              reads four k-steps ahead, the rope / scale loads of page i+2
//        M_i   P'(i-1) is in LDS -> the PV waves run PV(i-1) now, beside the rest of this wave's step:
//              the last chain MFMAs, the normaliser sums of page i-1, the scale triples of page i
//      (P' is written before M_i and read after it: ONE buffer).  sched_barrier(0) fences the slots: the software
//      pipeline is fixed in the source. ----
#define FL_SLOT_END() __builtin_amdgcn_sched_barrier(0)
constexpr int kMidSlot = 9;

// The chain's MFMAs are inline asm: as builtins they are register-only instructions without side effects, and hipcc
// sinks them towards their only consumer — the NEXT step's softmax — across both barriers.  (The S accumulator is read
// by VALU instructions a whole step later: no XDL-write -> VALU-read wait states are needed here.)
__device__ __forceinline__ v4i y_as_v4i(const uint4 a) { return v4i{(int)a.x, (int)a.y, (int)a.z, (int)a.w}; }
__device__ __forceinline__ v4i y_as_v4i(const v8bf b) {
  union { v8bf b; v4i i; } x;
  x.b = b;
  return x.i;
}
__device__ __forceinline__ void mfma_rope_first(v16f& acc, const uint4 a, const v8bf b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(y_as_v4i(a)), "v"(y_as_v4i(b)));
}
__device__ __forceinline__ void mfma_rope(v16f& acc, const uint4 a, const v8bf b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(y_as_v4i(a)), "v"(y_as_v4i(b)));
}
__device__ __forceinline__ void mfma_fp8(v16f& acc, const v8i a, const v8i b) {
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
               : "+v"(acc)
               : "v"(a), "v"(b), "v"(kUnitScale));
}   // M_i sits behind MFMA slot kMidSlot (0..3 rope, 4..11 latent)

template <bool HAS_CUR, bool HAS_PREV, bool MASK>
__device__ __forceinline__ void qk_step(v16f& acc_cur, v16f& y_prev, float& l_run, float& lq_run, float& m_w,
                                        const QkLane& lc, const int lane, const v8i (&qn)[8], const v8bf (&qr)[4],
                                        const float qs, float& ks_reg, const float* __restrict__ scale_next,
                                        const uint8_t* __restrict__ kp, const uint8_t* __restrict__ rp,
                                        float* __restrict__ scratch, uint8_t* __restrict__ pbuf_w,
                                        float* __restrict__ ref_w, const int tok0w_cur, const int tok0w_prev, const int L,
                                        const int L_row FL_T_PARAMS) {
  // MASK = false: every token of page i-1 is valid for every row of the wave (no selects); MASK = true: the pages around
  // the end of a sequence (rows past the end, causal limits of s_q > 1)
  constexpr bool need_mask_prev = MASK;
  const int li = lane & 31, lh = lane >> 5;
#if defined(FL_Y_NOCOMPUTE) || defined(FL_Y_NOQK)   // experiment: the DMA stream alone / PV waves alone (results are garbage)
  __builtin_amdgcn_s_barrier();   // M_i
  return;
#endif
  const float ks_cur = ks_reg;   // raw scale of page i (the register is refilled with page i+2 below)
  // ---- softmax of page i-1 in pieces ----
  float4 ks4[2], lk4[2], ik4[4];   // {ks, log2 ks} of two groups at a time; 1/ks of all four for the tail
  float tmax = -INFINITY, m_new = m_w, moff = 0.f;
  int pk[4];   // (the exponentials overwrite y_prev in place)
  auto load_kl = [&](const int g) {
    ks4[g & 1] = *reinterpret_cast<const float4*>(scratch + g * 8 + lh * 4);
    lk4[g & 1] = *reinterpret_cast<const float4*>(scratch + 32 + g * 8 + lh * 4);
  };
  auto scale_g = [&](const int g) {   // y = s * qs * ks + log2 ks; running max
    const float4 k4 = ks4[g & 1], l4 = lk4[g & 1];
    if (!need_mask_prev) {
      const float2v qs2 = {qs, qs};
      const float2v y01 = __builtin_elementwise_fma(float2v{y_prev[g * 4 + 0], y_prev[g * 4 + 1]} * qs2, float2v{k4.x, k4.y},
                                                    float2v{l4.x, l4.y});
      const float2v y23 = __builtin_elementwise_fma(float2v{y_prev[g * 4 + 2], y_prev[g * 4 + 3]} * qs2, float2v{k4.z, k4.w},
                                                    float2v{l4.z, l4.w});
      y_prev[g * 4 + 0] = y01[0];
      y_prev[g * 4 + 1] = y01[1];
      y_prev[g * 4 + 2] = y23[0];
      y_prev[g * 4 + 3] = y23[1];
      tmax = fmaxf(fmaxf(tmax, fmaxf(y01[0], y01[1])), fmaxf(y23[0], y23[1]));
    } else {
      const float ksv[4] = {k4.x, k4.y, k4.z, k4.w};
      const float lkv[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y = fmaf(y_prev[g * 4 + e] * qs, ksv[e], lkv[e]);
        if (tok0w_prev + g * 8 + lh * 4 + e >= L_row) y = -INFINITY;
        if (!(y == y)) y = -INFINITY;   // NaN can only come from garbage beyond the row's limit
        y_prev[g * 4 + e] = y;
        tmax = fmaxf(tmax, y);
      }
    }
  };
  auto ref_piece = [&]() {
    // max over the two lane halves without an LDS round trip: v_permlane32_swap exchanges lanes 32..63 of its first
    // operand with lanes 0..31 of its second
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
    tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    m_new = tmax > m_w ? ceilf(tmax) + kRefHeadroom : m_w;
    moff = kPShift - m_new;
  };
  auto exp_g = [&](const int g) {
    const float2v a01 = float2v{y_prev[g * 4 + 0], y_prev[g * 4 + 1]} + float2v{moff, moff};
    const float2v a23 = float2v{y_prev[g * 4 + 2], y_prev[g * 4 + 3]} + float2v{moff, moff};
    y_prev[g * 4 + 0] = __builtin_amdgcn_exp2f(a01[0]);
    y_prev[g * 4 + 1] = __builtin_amdgcn_exp2f(a01[1]);
    y_prev[g * 4 + 2] = __builtin_amdgcn_exp2f(a23[0]);
    y_prev[g * 4 + 3] = __builtin_amdgcn_exp2f(a23[1]);
    const int v = __builtin_amdgcn_cvt_pk_fp8_f32(y_prev[g * 4 + 0], y_prev[g * 4 + 1], 0, false);
    pk[g] = __builtin_amdgcn_cvt_pk_fp8_f32(y_prev[g * 4 + 2], y_prev[g * 4 + 3], v, true);
  };
  auto publish = [&]() {
    *reinterpret_cast<uint4*>(pbuf_w + lane * 16) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    ref_w[li] = m_new;   // (identical in both lane halves)
  };
  // piece k of the previous page's softmax (runs in MFMA slot k)
  auto sm_piece = [&](const int k) {
    if (!HAS_PREV) return;
#ifdef FL_Y_NOSOFTMAX   // experiment: QK waves without the softmax arithmetic (results are garbage)
    if (k == 9) { pk[0] = pk[1] = pk[2] = pk[3] = 0x38383838; m_new = 0.f; publish(); }
    return;
#endif
    if (k == 0) { load_kl(0); load_kl(1); }
    else if (k == 1) scale_g(0);
    else if (k == 2) { scale_g(1); load_kl(2); load_kl(3); }
    else if (k == 3) scale_g(2);
    else if (k == 4) scale_g(3);
    else if (k == 5) ref_piece();
    else if (k == 6) exp_g(0);
    else if (k == 7) exp_g(1);
    else if (k == 8) exp_g(2);
    else if (k == 9) { exp_g(3); publish(); }
  };
  static_assert(kMidSlot == 9, "the P' store is piece 9");
static_assert(kDmaPerWavePage == 10, "vmcnt(10) in the PV step");

  v16f acc;
  if (HAS_CUR) {
    uint4 ka[8][2];
    int kb0 = lc.kb0;   // opaque per step: the derived k-step offsets are not kept live across steps
    asm volatile("" : "+v"(kb0));
    auto k_load = [&](const int s) {
      ka[s][0] = *reinterpret_cast<const uint4*>(kp + (kb0 ^ ((s & 3) << 6)) + (s >> 2) * 256);
      ka[s][1] = *reinterpret_cast<const uint4*>(kp + (kb0 ^ ((s & 3) << 6) ^ 16) + (s >> 2) * 256);
    };
    // rope operand of this lane: token 32W + li, 16-B chunk 2s + lh of its 128-B row (stored at chunk ^ ((T>>1)&7))
    uint4 ra[4];
    int rb0 = lc.rb0;
    asm volatile("" : "+v"(rb0));
#pragma unroll
    for (int s = 0; s < 4; ++s) ra[s] = *reinterpret_cast<const uint4*>(rp + (rb0 ^ (s << 5)));
#pragma unroll
    for (int s = 0; s < 3; ++s) k_load(s);
    FL_SLOT_END();
#pragma unroll
    for (int slot = 0; slot < 12; ++slot) {
      // (operand reads three k-steps ahead: behind the MFMA of k-step s goes the read of k-step s + 3)
#ifdef FL_Y_NOQKMFMA   // experiment: QK waves without their MFMA chain (results are garbage)
      if (slot == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = __uint_as_float(ra[r & 3].x ^ ka[r & 1][0].y);
      }
      if (slot >= 4 && slot - 4 + 3 < 8) { k_load(slot - 4 + 3); acc[slot] += __uint_as_float(ka[slot - 4][1].z); }
#else
      if (slot == 0) {
        mfma_rope_first(acc, ra[0], qr[0]);
      } else if (slot < 4) {
        mfma_rope(acc, ra[slot], qr[slot]);
      } else {
        const int s = slot - 4;
        mfma_fp8(acc, make_v8i(ka[s][0], ka[s][1]), qn[s]);
        if (s + 3 < 8) k_load(s + 3);
      }
#endif
      // raw scale of page i+2 (UNCONDITIONAL load: the caller clamps the page into the part — behind a conditional load
      // hipcc can only wait with vmcnt(0))
      if (slot == 4) ks_reg = *scale_next;
      sm_piece(slot);
      FL_SLOT_END();
      if (slot == kMidSlot) {
        FL_T(6);   // chain up to the mid-step barrier || softmax of the previous page
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // P' / reference are in LDS
        __builtin_amdgcn_s_barrier();   // M_i
        FL_T(7);   // mid-step barrier
      }
    }
    acc_cur = acc;
  } else {
#pragma unroll
    for (int k = 0; k <= kMidSlot; ++k) {
      sm_piece(k);
      FL_SLOT_END();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // M_i
  }
  // ---- behind M_i: the two normalisers of page i-1, then the scale triples of page i ----
#ifdef FL_Y_NOSOFTMAX
  if (false) {
#else
  if (HAS_PREV) {
#endif
#pragma unroll
    for (int g = 0; g < 4; ++g) ik4[g] = *reinterpret_cast<const float4*>(scratch + 64 + g * 8 + lh * 4);
    const float f = __builtin_amdgcn_exp2f(m_w - m_new);   // exactly 1 when the reference did not move
    // two partial sums each, as packed f32 math
    float2v l2 = {l_run * f, 0.f}, q2 = {lq_run * f, 0.f};
    m_w = m_new;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float2v ik01 = {ik4[g].x, ik4[g].y}, ik23 = {ik4[g].z, ik4[g].w};
      // unrounded sum: exact LSE
      l2 = __builtin_elementwise_fma(float2v{y_prev[g * 4 + 0], y_prev[g * 4 + 1]}, ik01, l2);
      l2 = __builtin_elementwise_fma(float2v{y_prev[g * 4 + 2], y_prev[g * 4 + 3]}, ik23, l2);
      // the ROUNDED weights normalise O (numerator and denominator use the same weights: they sum to exactly 1)
      q2 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8(pk[g], false), ik01, q2);
      q2 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8(pk[g], true), ik23, q2);
    }
    l_run = l2[0] + l2[1];
    lq_run = q2[0] + q2[1];
  }
  if (HAS_CUR) scale_prep(scratch, ks_cur, tok0w_cur, li, L);   // (behind this wave's last read of page i-1's triples)
}

// ---- PV wave: O^T[256 dims x 32 rows] += V^T(page) . P^T, with the LDS-DMA refill of a later page in the MFMA shadow ----
// The reference of O is fixed when a row sees its first valid token and NEVER moves in a pass: later blocks with a larger
// reference m_b enter with an E8M0 block scale 2^(m_b - m_o) > 1 (exact; fp32 O has the range), so nothing but the MFMA
// touches O in the page loop (a conditional rescale makes hipcc copy all 128 O registers at the join, every page).  Only
// a reference more than kMaxUp above m_o cannot be represented: it raises `redo` and the workgroup repeats the request
// with m_o preset to the final reference.
constexpr float kMaxUp = 64.f;
template <bool DMA>
__device__ __forceinline__ void pv_step(v16f (&o)[8], float& m_o, int& redo, const PvLane& lc_in, const int lane,
                                        const uint8_t* __restrict__ vp, const uint8_t* __restrict__ pbuf_rt,
                                        const float* __restrict__ ref_rt, const uint8_t* __restrict__ src_nope,
                                        uint8_t* __restrict__ dma_dst, const uint8_t* __restrict__ src_rope,
                                        uint8_t* __restrict__ dma_dst_rope FL_T_PARAMS) {
  const int li = lane & 31, lh = lane >> 5;
  // opaque copy of the lane constants: nothing derived from them is hoisted out of the page loop (and spilled)
  PvLane lc = lc_in;
  asm volatile("" : "+v"(lc.vb0), "+v"(lc.dn_row), "+v"(lc.dn_x), "+v"(lc.dr[0]), "+v"(lc.dr[1]));
  // ---- B_i .. M_i (the QK waves run their MFMA chains): the refill of page i+2 and the first V^T operands ----
#if !defined(FL_Y_NODMA)
  if (DMA) dma_page_pieces(lc, src_rope, dma_dst_rope, src_nope, dma_dst);
#endif
#ifdef FL_Y_NOCOMPUTE
  __builtin_amdgcn_s_barrier();   // M_i
  return;
#endif
  v8i va[8];
  auto load_vt = [&](int jb) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint8_t* ap = vp + (lc.vb0 ^ (((jb & 3) << 4) | ((jb >> 2) << 7))) + (u & 1) * (16 * kDN) + (u >> 1) * (32 * kDN);
      const v2i t2 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)ap);
      va[jb][2 * u] = t2[0];
      va[jb][2 * u + 1] = t2[1];
    }
  };
  load_vt(0);
  load_vt(1);
  load_vt(2);
  __builtin_amdgcn_sched_barrier(0);
  FL_T(2);   // refill issue + first V^T reads
  __builtin_amdgcn_s_barrier();   // M_i: P'(i-1) and its references are in LDS
  FL_T(6);   // mid-step barrier
  const uint4 p0 = *reinterpret_cast<const uint4*>(pbuf_rt + lane * 16);
  const uint4 p1 = *reinterpret_cast<const uint4*>(pbuf_rt + 64 * 16 + lane * 16);
  const float m0 = ref_rt[li];
  const float m1 = ref_rt[32 + li];
  const float mw_max = fmaxf(m0, m1);
  m_o = m_o > kNegRef ? m_o : mw_max;
  redo |= (mw_max - m_o > kMaxUp) ? 1 : 0;
  int sb = 127 + (int)fminf((lh ? m1 : m0) - m_o, kMaxUp);
  sb = sb < 0 ? 0 : sb;
  const v8i pb = make_v8i(p0, p1);
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    if (jb + 3 < 8) load_vt(jb + 3);
    o[jb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va[jb], pb, o[jb], 0, 0, 0, kUnitScale, 0, sb);
  }
  // V^T operand reads three tiles ahead of their MFMA
  __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // P', references
#pragma unroll
  for (int jb = 0; jb < 5; ++jb) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
  }
  __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
  __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(512) void mla_decode_y_kernel(
    const Params p, const int32_t* __restrict__ g_block_table, const int32_t* __restrict__ g_seqlens,
    const int32_t* __restrict__ g_meta, const int32_t* __restrict__ g_num_splits,
    const uint8_t* __restrict__ g_k_nope, const uint16_t* __restrict__ g_k_rope, const float* __restrict__ g_k_scale,
    const uint8_t* __restrict__ g_q_nope, const uint16_t* __restrict__ g_q_rope, const float* __restrict__ g_q_scale) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[kLdsBytes];

  const int tid = threadIdx.x;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_pv = wave_id >= 4;
  const int w4 = wave_id & 3;
  const int rt = w4 & 1;    // row tile (32 query rows) inside the workgroup
  const int W = w4 >> 1;    // QK wave: token half; PV wave: d half
  const int lane = tid & 63, li = lane & 31, lh = lane >> 5;

  // ---- workgroup -> (part, row group); the row groups of a part read the same pages: same XCD (block b -> XCD b%8) ----
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
  const int32_t* meta = g_meta + part * FL_MLA_META_W;
  int req = meta[0];
  int tile_b = meta[1];
  const int end_req = meta[2];
  const int end_tile = meta[3];
  int split_idx = meta[4];

  const int row = rgrp * 64 + rt * 32 + li;   // query row of this lane (both roles: lane&31 = row inside the tile)
  const bool row_ok = row < p.rows;

#ifdef FL_MLA_TIMING
  unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
  const unsigned long long tstart = tlast;
  const unsigned long long wstart = wall_clock64();   // 100 MHz
#endif

  // per-request geometry shared by both roles
#define FL_Y_REQUEST_HEAD()                                                                                            \
  if (req > end_req || (req == end_req && end_tile == 0)) break;                                                       \
  const int L = g_seqlens[req];                                                                                        \
  const int nt = L > 0 ? (L + kPage - 1) / kPage : 0;                                                                  \
  int tile_e = req < end_req ? nt : (end_tile < nt ? end_tile : nt);                                                   \
  if (tile_e < tile_b) tile_e = tile_b;                                                                                \
  const int n = tile_e - tile_b;                                                                                       \
  /* page ids of a 64-page window live in ONE VGPR (lane j = page win_base + j); a lookup is a v_readlane */           \
  int win_base = 0;                                                                                                    \
  int pg_vec = 0;                                                                                                      \
  auto load_window = [&](int base) {                                                                                   \
    win_base = base;                                                                                                   \
    const int t = base + lane;                                                                                         \
    int pg = 0;                                                                                                        \
    if (t < n) pg = g_block_table[(long long)req * p.bt_stride + tile_b + t];                                          \
    pg_vec = (pg < 0 || pg >= p.num_pages) ? 0 : pg;                                                                   \
  };                                                                                                                   \
  load_window(0);                                                                                                      \
  auto page_of = [&](int t) { return (long long)__builtin_amdgcn_readlane(pg_vec, t - win_base); }

  const int* redo_flag = reinterpret_cast<const int*>(smem + kOffFlag);

  if (!is_pv) {
    // =========================== QK waves ===========================
    QkLane lc;
    {
      // K operand: token T = 32W + li, 32 B at d = 64s + 32lh -> 16-B chunks c = 4s + 2lh + e, stored at chunk c ^ (T&15)
      const int kx = li & 15;
      lc.rb0 = li * (kDR * 2) + (((lh ^ ((li >> 1) & 7))) << 4);
      lc.kb0 = li * kDN + (((((kx >> 2)) << 2) | ((2 * lh) ^ (kx & 3))) << 4);
    }
    float* scratch = reinterpret_cast<float*>(smem + kOffScratch + w4 * kScratchPerWave);
    for (; req < p.bs; ++req, tile_b = 0) {
      FL_Y_REQUEST_HEAD();
      // Q fragments (B operands), once per request
      const long long qrow = (long long)req * p.rows + row;
      v8i qn[8];
      v8bf qr[4];
      float qs = 0.f;
      if (row_ok) {
        const uint8_t* qp = g_q_nope + qrow * kDN + lh * 32;
#pragma unroll
        for (int s = 0; s < 8; ++s)
          qn[s] = make_v8i(*reinterpret_cast<const uint4*>(qp + s * 64), *reinterpret_cast<const uint4*>(qp + s * 64 + 16));
        const uint16_t* rp = g_q_rope + qrow * kDR + lh * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) qr[s] = as_bf8(*reinterpret_cast<const uint4*>(rp + s * 16));
        qs = g_q_scale[qrow] * p.scale_log2e;
      } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) qn[s] = v8i{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 4; ++s) qr[s] = as_bf8(make_uint4(0, 0, 0, 0));
      }
      int L_row = L;
      if (p.causal) L_row = L - (p.s_q - 1 - row / p.h_q);   // query j sees keys [0, L - (s_q-1-j))
      if (!row_ok) L_row = 0;
      const int L_min = p.causal ? L - (p.s_q - 1) : L;

      // raw k_scale of token 32W + li of a page, global -> register two pages ahead.  Every load is UNCONDITIONAL (page
      // index clamped into the part; an empty part reads the padding page 0): behind a conditional load hipcc can only
      // wait with vmcnt(0), which would expose the whole latency of the load issued one step earlier, every step.
      auto scale_src = [&](const int t) { return g_k_scale + page_of(t) * kPage + 32 * W + li; };
      for (int pass = 0; pass < 2; ++pass) {
        float l_run = 0.f, lq_run = 0.f, m_w = kNegRef;
        float ksA, ksB;
        v16f accA, accB;
        if (pass == 1) load_window(0);
        ksA = *scale_src(0);
        ksB = *scale_src(n > 1 ? 1 : 0);

        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // R0: every wave is done with the LDS of the previous request / pass
        FL_T(5);   // request prologue

        // step i: chain of page i into `cur` (rope / scale registers `rr`, refilled with page i+2) || softmax of page
        // i-1 from `prev`
#define FL_Y_QK_STEP(HC, HP, MK, RR, CUR, PREV)                                                                         \
  {                                                                                                                    \
    if (HC && i + 2 >= win_base + 64 && i + 2 < n) load_window(i + 2); /* pages i+2 .. i+65 */                          \
    const int t2 = i + 2 < n ? i + 2 : (n > 0 ? n - 1 : 0);                                                            \
    const float* scale_next = scale_src(t2);                                                                           \
    FL_T(2); /* loop control */                                                                                        \
    __builtin_amdgcn_s_barrier(); /* B_i: page i landed */                                                             \
    FL_T(0); /* barrier */                                                                                             \
    qk_step<HC, HP, MK>(CUR, PREV, l_run, lq_run, m_w, lc, lane, qn, qr, qs, RR, scale_next,                           \
                        smem + kOffRing + (i & 3) * kSlotBytes + W * (32 * kDN),                                       \
                        smem + kOffRope + (i % kRopeSlots) * kRopeBytes + W * (32 * kDR * 2), scratch,                 \
                        smem + kOffPbuf + (rt * 2 + W) * (64 * 16),                                                    \
                        reinterpret_cast<float*>(smem + kOffRef) + (rt * 2 + W) * 32,                                  \
                        (tile_b + i) * kPage + 32 * W, (tile_b + i - 1) * kPage + 32 * W, L, L_row FL_T_ARGS);          \
    FL_T(1); /* rest of the chain, normalisers, next triples */                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                 \
    FL_T(3); /* LDS drain */                                                                                           \
  }
        {
          int i = 0;
          if (n > 0) {
            FL_Y_QK_STEP(true, false, false, ksA, accA, accB)
            // steady state: the softmax page i-1 lies inside every row's limit.  Pairs of steps without a condition in
            // between (a skipped second step would leave a path with nothing issued behind a rope buffer's loads:
            // vmcnt(0) again); then the pages around the end of the sequence, one step at a time
            int n_fast = L_min / kPage - tile_b + 1;   // steps 1 .. n_fast - 1 are unmasked
            n_fast = n_fast < n ? n_fast : n;
            for (i = 1; i + 1 < n_fast; i += 2) {
              FL_Y_QK_STEP(true, true, false, ksB, accB, accA)
              ++i;
              FL_Y_QK_STEP(true, true, false, ksA, accA, accB)
              --i;
            }
            for (; i < n; ++i) {
              if (i & 1) FL_Y_QK_STEP(true, true, true, ksB, accB, accA)
              else FL_Y_QK_STEP(true, true, true, ksA, accA, accB)
            }
            if (n & 1) FL_Y_QK_STEP(false, true, true, ksB, accB, accA)   // i = n: softmax of page n-1 (in accA)
            else FL_Y_QK_STEP(false, true, true, ksA, accA, accB)         // (in accB)
          } else {
            FL_Y_QK_STEP(false, false, false, ksA, accA, accB)
          }
        }
#undef FL_Y_QK_STEP
        // normalisers of this wave's blocks -> LDS for the PV waves' epilogue
        {
          const float l_tot = l_run + __shfl_xor(l_run, 32);
          const float lq_tot = lq_run + __shfl_xor(lq_run, 32);
          float* lm = reinterpret_cast<float*>(smem + kOffLm) + (rt * 2 + W) * 96;
          if (lh == 0) {
            lm[li] = l_tot;
            lm[32 + li] = lq_tot;
            lm[64 + li] = m_w;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // E0
        FL_T(4);   // B_n, normalisers, E0
        if (pass == 1) break;
        if ((redo_flag[0] | redo_flag[1] | redo_flag[2] | redo_flag[3]) == 0) break;   // workgroup-uniform
      }
    }
#ifdef FL_MLA_TIMING
    if (g_dbg_y != nullptr && lane == 0) {
      unsigned long long* d = reinterpret_cast<unsigned long long*>(g_dbg_y) + ((long long)blockIdx.x * 8 + wave_id) * 14;
      for (int i = 0; i < 12; ++i) d[i] = tacc[i];
      d[12] = __builtin_readcyclecounter() - tstart;
      d[13] = wall_clock64() - wstart;
    }
#endif
    return;
  }

  // =========================== PV waves ===========================
  PvLane lc;
  {
    const int s16 = lane & 15;
    const int gi = (lane >> 4) & 1;
    const int tj = s16 >> 1;
    const int tok_in8 = (tj & 3) + ((tj >> 2) << 3);
    const int vrow = 4 * lh + tok_in8;
    lc.vb0 = vrow * kDN + ((((gi << 2)) ^ (vrow & 15)) << 4) + (s16 & 1) * 8;
    // latent DMA piece k of this wave: token row T = (w4*8 + k)*2 + lh, chunk li stored from source chunk li ^ (T&15)
    lc.dn_row = (unsigned)((w4 * kPiecesPerWave * 2 + lh) * kDN);
    lc.dn_x = (unsigned)((li ^ lh) << 4);
#pragma unroll
    for (int k = 0; k < kRopePiecesPerWave; ++k) {   // rope piece: 8 token rows of 128 B, chunk c of row T from source chunk c ^ ((T>>1)&7)
      const int T = (w4 * kRopePiecesPerWave + k) * 8 + (lane >> 3);
      lc.dr[k] = (unsigned)(T * 128 + (((lane & 7) ^ ((T >> 1) & 7)) << 4));
    }
  }
  for (; req < p.bs; ++req, tile_b = 0, split_idx = 0) {
    FL_Y_REQUEST_HEAD();
    const int split_base = g_num_splits[req];
    const bool is_split = (g_num_splits[req + 1] - split_base) > 1;
    auto ring = [&](int t) { return smem + kOffRing + (t & 3) * kSlotBytes; };
    auto src_of = [&](int t) { return g_k_nope + page_of(t) * (long long)(kPage * kDN); };
    auto rope_src_of = [&](int t) { return reinterpret_cast<const uint8_t*>(g_k_rope) + page_of(t) * (long long)(kPage * kDR * 2); };
    auto rope_ring = [&](int t) { return smem + kOffRope + (t % kRopeSlots) * kRopeBytes; };
    // (issue order per page: rope, latent — kDmaPerWavePage pieces per wave: the counted waits rely on it)
    auto issue_page = [&](int t) {
      dma_page_pieces(lc, rope_src_of(t), rope_ring(t) + w4 * (kRopePiecesPerWave * 1024), src_of(t),
                      ring(t) + w4 * (kPiecesPerWave * 1024));
    };
    const float* lm = reinterpret_cast<const float*>(smem + kOffLm) + rt * 192;

    v16f o[8];
    float m_o = kNegRef;
    float mo_preset = kNegRef;
    // pass 0 fixes the O reference at each row's first valid page; pass 1 runs only if some block reference outran it
    // by more than kMaxUp (pv_step), with the reference preset to the final one
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
      m_o = mo_preset;
      int redo = 0;
      if (pass == 1) load_window(0);

      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // R0
      if (n > 0) issue_page(0);
      if (n > 1) issue_page(1);
      FL_T(5);   // request prologue

      // step i: wait for page i, barrier B_i, PV(i-1) with the refill of page i+2 in its MFMA shadow, tail fill of page i.
      // The steps with a refill and the last two without are separate loops, each with ONE step variant (two variants
      // joined inside a loop make hipcc copy the O registers at the join).
#define FL_Y_PV_STEP(HAS_PREV, HAS_DMA)                                                                                \
  {                                                                                                                    \
    if (FL_Y_NOWAIT) { /* experiment: timing without the page-landed wait (results are garbage) */                     \
    } else if (i + 1 < n) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); /* page i landed, page i+1 may stay in flight */ \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                 \
    FL_T(0); /* page-landed wait */                                                                                    \
    __builtin_amdgcn_s_barrier(); /* B_i */                                                                            \
    FL_T(1); /* barrier */                                                                                             \
    if (HAS_DMA && i + 2 >= win_base + 64) load_window(i + 2);                                                         \
    const uint8_t* sn = HAS_DMA ? src_of(i + 2) : nullptr;                                                             \
    const uint8_t* sr = HAS_DMA ? rope_src_of(i + 2) : nullptr;                                                        \
    uint8_t* dst = ring(i + 2) + w4 * (kPiecesPerWave * 1024);                                                         \
    uint8_t* dstr = rope_ring(i + 2) + w4 * (kRopePiecesPerWave * 1024);                                               \
    if (HAS_PREV) {                                                                                                    \
      pv_step<HAS_DMA>(o, m_o, redo, lc, lane, ring(i - 1) + W * 256,                                                  \
                       smem + kOffPbuf + rt * (2 * 64 * 16), reinterpret_cast<const float*>(smem + kOffRef) + rt * 64, \
                       sn, dst, sr, dstr FL_T_ARGS);                                                                   \
    } else {                                                                                                           \
      if (HAS_DMA) dma_page_pieces(lc, sr, dstr, sn, dst);                                                             \
      __builtin_amdgcn_s_barrier(); /* M_i */                                                                          \
    }                                                                                                                  \
    /* tail of the sequence: zero the rows of page i past the end (P' is exactly 0 there, but 0 * NaN from stale fp8   \
       NaN patterns would poison the PV MFMA of the next step); the QK waves mask those tokens by index */             \
    if (i < n && (tile_b + i) * kPage + kPage > L) {                                                                   \
      const int nvalid = L - (tile_b + i) * kPage;                                                                     \
      uint8_t* wr = ring(i);                                                                                           \
      _Pragma("clang loop vectorize(disable) unroll(disable)")                                                         \
      for (int T = nvalid + 2 * w4 + lh; T < kPage; T += 8)                                                            \
        *reinterpret_cast<uint4*>(wr + T * kDN + li * 16) = make_uint4(0, 0, 0, 0);                                    \
    }                                                                                                                  \
    FL_T(7); /* PV MFMAs + tail fill */                                                                                \
  }
      {
        int i = 0;
        if (n > 2) FL_Y_PV_STEP(false, true) else FL_Y_PV_STEP(false, false)
        for (i = 1; i + 2 < n; ++i) FL_Y_PV_STEP(true, true)
        for (; i <= n; ++i) FL_Y_PV_STEP(true, false)
      }
#undef FL_Y_PV_STEP
      // workgroup-uniform redo decision (the page loop has workgroup barriers)
      if (lane == 0) reinterpret_cast<int*>(smem + kOffFlag)[w4] = (pass == 0 && __any(redo != 0)) ? 1 : 0;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // E0: normalisers of the QK waves and the redo votes are in LDS, the ring is free
      FL_T(3);   // E0
      if (pass == 1) break;
      if ((redo_flag[0] | redo_flag[1] | redo_flag[2] | redo_flag[3]) == 0) break;
      mo_preset = fmaxf(lm[64 + li], lm[96 + 64 + li]);   // final references of the two blocks
    }

    // ---- per-request epilogue: merge the normalisers of the two blocks, normalise, store this wave's d half ----
    const float mA = lm[64 + li], mB = lm[96 + 64 + li];
    const float fA = __builtin_amdgcn_exp2f(mA - m_o), fB = __builtin_amdgcn_exp2f(mB - m_o);   // <= 2^kMaxUp
    const float l = lm[li] * fA + lm[96 + li] * fB;
    const float lq = lm[32 + li] * fA + lm[96 + 32 + li] * fB;
    const float inv = lq > 0.f ? 1.f / lq : 0.f;
    const float lse_nat = l > 0.f ? (__builtin_amdgcn_logf(l) + m_o - kPShift) * 0.6931471805599453f : -INFINITY;
    // split-KV partials are normalised by lq, so they are COMBINED with lq-based weights; the exact LSE travels along
    const float lseq_nat = lq > 0.f ? (__builtin_amdgcn_logf(lq) + m_o - kPShift) * 0.6931471805599453f : -INFINITY;
    const int slot_idx = split_base + split_idx;
    if (row_ok && lh == 0 && W == 0) {
      if (is_split) {
        p.lse_accum[((long long)slot_idx * p.rows + row) * 2 + 0] = lseq_nat;
        p.lse_accum[((long long)slot_idx * p.rows + row) * 2 + 1] = lse_nat;
      } else {
        const int j = row / p.h_q, h = row - j * p.h_q;
        p.lse[((long long)req * p.h_q + h) * p.s_q + j] = lse_nat;
      }
    }
    // O -> memory through a wave-private LDS transpose (the ring is free now).  A lane holds ONE row, 4 dims at a
    // time: stored directly, every store instruction would touch 64 rows x 8 B.  Tiles 4c..4c+3 cover the contiguous
    // dims [256W + 128c, +128) (C row i = e + 8g + 4lh of tile 4c + jq is d = 128c + 16jq + (i&15) + 64(i>>4)), so
    // chunk c is staged as [32 rows][128 f32] (+4 pad) and leaves as 256-B bf16 row segments.
    {
      float* stg = reinterpret_cast<float*>(smem + kOffRing) + w4 * (32 * kStgStride);
      const int row0 = rgrp * 64 + rt * 32;
      uint16_t* dst = is_split ? reinterpret_cast<uint16_t*>(p.o_accum) + ((long long)slot_idx * p.rows + row0) * kDN
                               : p.out + ((long long)req * p.rows + row0) * kDN;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int jq = 0; jq < 4; ++jq)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int d_off = 16 * jq + 8 * (g & 1) + 4 * lh + 64 * (g >> 1);
            *reinterpret_cast<float4*>(stg + li * kStgStride + d_off) =
                make_float4(o[4 * c + jq][g * 4 + 0] * inv, o[4 * c + jq][g * 4 + 1] * inv,
                            o[4 * c + jq][g * 4 + 2] * inv, o[4 * c + jq][g * 4 + 3] * inv);
          }
        uint16_t* dbase = dst + 256 * W + 128 * c + (lane & 15) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int r = (lane >> 4) + 4 * k;
          const float4 v0 = *reinterpret_cast<const float4*>(stg + r * kStgStride + (lane & 15) * 8);
          const float4 v1 = *reinterpret_cast<const float4*>(stg + r * kStgStride + (lane & 15) * 8 + 4);
          uint4 ov;   // v_cvt_pk_bf16_f32: RNE, two values per instruction
          ov.x = fl_pack_bf16(v0.x, v0.y);
          ov.y = fl_pack_bf16(v0.z, v0.w);
          ov.z = fl_pack_bf16(v1.x, v1.y);
          ov.w = fl_pack_bf16(v1.z, v1.w);
          if (row0 + r < p.rows) *reinterpret_cast<uint4*>(dbase + (long long)r * kDN) = ov;
        }
      }
    }
    FL_T(4);   // epilogue
  }
#undef FL_Y_REQUEST_HEAD
#ifdef FL_MLA_TIMING
  if (g_dbg_y != nullptr && lane == 0) {
    unsigned long long* d = reinterpret_cast<unsigned long long*>(g_dbg_y) + ((long long)blockIdx.x * 8 + wave_id) * 14;
    for (int i = 0; i < 12; ++i) d[i] = tacc[i];
    d[12] = __builtin_readcyclecounter() - tstart;
    d[13] = wall_clock64() - wstart;
  }
#endif
}

}  // namespace

int fl_mla_decode_fp8_y_impl(const FlMlaDecodeArgs* a, const Params& p_in, hipStream_t stream) {
  Params p = p_in;
  p.row_groups = (p.rows + 63) / 64;
  p.partial_bf16 = 1;   // split partials travel as bf16 rows
  const dim3 grid((unsigned)(p.num_parts * p.row_groups)), block(512);
  mla_decode_y_kernel<<<grid, block, 0, stream>>>(
      p, a->block_table, a->cache_seqlens, a->tile_scheduler_metadata, a->num_splits, (const uint8_t*)a->k_nope,
      (const uint16_t*)a->k_rope, a->k_scale, (const uint8_t*)a->q_nope, (const uint16_t*)a->q_rope, a->q_scale);
  FL_CHECK_LAUNCH("mla_decode_y_kernel");
  return fl_mla_launch_combine(p, a->num_splits, stream);
}

#ifdef FL_MLA_TIMING
extern "C" int fl_mla_debug_set_buffer_y(int* dev_ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_y), &dev_ptr, sizeof(dev_ptr));
}
#endif
