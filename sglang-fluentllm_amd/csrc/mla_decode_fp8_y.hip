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
// Split requests write normalised bf16 partial rows + {weight-LSE, exact-LSE}; with at least half as many requests as parts
// the request's FIRST piece merges them inside this kernel (see the epilogue), otherwise mla_combine_kernel does.
#include "mla_decode_shared.h"

using namespace fl_mla;

namespace {

constexpr int kOffRing = 0;                                        // 4 x 32 KiB
constexpr int kPbufPerParity = 2 * 2 * 64 * 16;                    // [rt 2][W 2][64 lanes][16 B]
constexpr int kOffPbuf = kOffRing + kRingSlots * kSlotBytes;       // [parity 2]
constexpr int kRefPerParity = 2 * 2 * 32 * 4;                      // [rt 2][W 2][32 rows] f32
constexpr int kOffRef = kOffPbuf + 2 * kPbufPerParity;             // [parity 2]
constexpr int kScratchPerWave = 2 * 3 * 32 * 4;                    // [page parity 2] {ks, log2 ks, 1/ks} x 32 tokens
constexpr int kOffScratch = kOffRef + 2 * kRefPerParity;           // [QK wave 4]
constexpr int kOffLm = kOffScratch + 4 * kScratchPerWave;          // [rt 2][W 2][3][32] f32: l, lq, m per row
constexpr int kOffQr = kOffLm + 2 * 2 * 3 * 32 * 4;                   // [QK wave 4][4 k-steps][64 lanes] 16 B: Q rope fragments
constexpr int kOffMerge = kOffQr + 4 * 4096;                            // 1 int: PV waves of this workgroup past a merge's poll (in-kernel split merge)
constexpr int kLdsBytes = kOffMerge + 16;
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
constexpr int kStgBytes = 512 + 16;                                // epilogue staging: bytes per row (256 bf16 dims of a PV wave + pad; in the ring)
static_assert(4 * 32 * kStgBytes <= kRingSlots * kSlotBytes, "epilogue staging fits the ring");

// NRT = row tiles of 32 query rows per workgroup: 2 (64 rows, 8 waves: two per SIMD) for more than 32 rows per request; 1 (32 rows, 4
// waves: q(W), v(Wd), one per SIMD) for at most 32 rows — the TP8 shard's H = 16.  The 32 LDS-DMA pieces of a page are shared by the
// 2 NRT PV waves: 8 or 16 pieces of 1 KiB per PV wave and page.
template <int NRT> constexpr int pieces_per_wave() { return kDmaNopePerTile / (2 * NRT); }
// Budget of the in-kernel split merge's poll, in ticks of the 100 MHz wall clock (default 2 s; fl_mla_set_merge_timeout / FLUENT_MLA_MERGE_TIMEOUT_S),
// and the host-mapped word a merger that gives up reports to (0 = none yet): the outputs of that launch are NaN AND the next fl_mla_decode call
// on the host returns FL_ERR_LAUNCH with the part's index (a timeout must not stay a silent wrong answer).  Device globals, read on the
// cold path only: nothing is added to the kernel's arguments.
__device__ unsigned long long g_merge_timeout_ticks = 200000000ull;
__device__ unsigned* g_merge_err = nullptr;

// (The experiment switches of rounds 2-4 — mid-step barrier, dual accumulate chain, lagged reference, tail loads inside the chain,
//  L2 prefetch touches, DMA placement, the garbage-result bounding builds — live in probes/r05_k1_lab_switches.patch.txt with
//  their measured results; the shipped file keeps the phase timer only.)
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

// Rope A operand (4 bf16 k-steps) and raw k_scale of one page for one QK wave, held in registers two pages ahead.
struct RopeRegs {
  uint4 ra[4];
  float ks;
};

struct QkLane {
  int kb0;   // K operand: byte offset inside a slot for k-step 0, first 16 B (second: ^16); k-step s: ^ ((s&3) << 6), + (s>>2)*256
};
struct PvLane {
  int vb0;          // V^T tr8 source of tile jb = 0 (tile jb: ^ ((jb&3) << 4) ^ ((jb>>2) << 7)); + u immediates; + Wd*256
  unsigned dn_row;  // latent DMA: byte offset of this lane's token row of piece 0 of this wave (piece k: + k * 1024)
  unsigned dn_x;    // ... and its swizzled 16-B chunk (piece k: ^ (k << 5))
};
// byte offset inside the page of this lane's 16 B of latent piece k of this wave (2 token rows of 512 B per piece, chunk c
// of token T stored at chunk c ^ (T & 15))
// (FMT 1: the plain-fp8 cache has 576-B token rows — two rows of a piece are 1152 source bytes apart; the destination in LDS
//  is the same 512-B-row slot: only the latent part of a row is staged, the row's rope bytes go to registers)
template <int FMT>
__device__ __forceinline__ unsigned dn_off(const PvLane& lc, const int k) {
  return lc.dn_row + (unsigned)k * (FMT == 1 ? 1152u : 1024u) + (lc.dn_x ^ ((unsigned)(k & 7) << 5));
}

// the eight K operand chunk addresses of a lane (ring slot base included): chunk pair of k-step s (and s + 4 at + 256) at
// kaddr[2 (s&3)], kaddr[2 (s&3) + 1].  base is a multiple of 16 KiB and kb0 < 16 KiB: (base + kb0) ^ c == base + (kb0 ^ c).
__device__ __forceinline__ void qk_addr(int (&kaddr)[8], const int kb0_in, const int base) {
  int b = kb0_in;
  asm volatile("" : "+v"(b));   // opaque per pair: nothing derived from it is hoisted out of the page loop (and spilled)
  b += base;
#pragma unroll
  for (int j = 0; j < 8; ++j) kaddr[j] = b ^ (((j >> 1) << 6) | ((j & 1) << 4));
}

// scale triples {ks, log2 ks, 1/ks} of a QK wave's 32 tokens (lane li = token 32W + li) -> wave-private scratch
__device__ __forceinline__ void scale_prep(float* __restrict__ scratch, float ks, const int tok0w, const int li, const int L) {
  // (a usable scale is a positive normal / subnormal float: one v_cmp_class_f32)
  ks = ((tok0w + li >= L) | !__builtin_amdgcn_classf(ks, 0x180)) ? 1.f : ks;
  // (both lane halves store the same values to the same addresses: no exec-mask branch)
  scratch[li] = ks;
  scratch[32 + li] = __builtin_amdgcn_logf(ks);
  scratch[64 + li] = __builtin_amdgcn_rcpf(ks);
}

// ---- QK wave: one page step (block W of page i for row tile rt).  The critical path of a step is this wave's chain
//      barrier -> K reads -> 12 dependent MFMAs -> softmax -> P' in LDS -> barrier; everything else is moved off it:
//      the scale triples of the page were written at the end of the previous step, the rope / scale loads of page i+2
//      go out behind the MFMA issue (into the registers the rope MFMAs just read), the normaliser sums and the next
//      page's triples follow the P' store.  s_setprio 1 around the MFMA chain: the PV wave of this SIMD has its 8
//      MFMAs ready at the same time, and they belong beside this wave's softmax, not inside its chain. ----
__device__ __forceinline__ void qk_step(float& l_run, float& lq_run, float& m_w, const int (&kaddr)[8], const int koff, const int lane,
                                        const v8i (&qn)[8], const uint8_t* __restrict__ qr_lds, const float qs, RopeRegs& rr,
                                        const float ks_next, const uint8_t* __restrict__ rope_next,
                                        const float* __restrict__ scale_next, const uint8_t* __restrict__ kp,
                                        float* __restrict__ scratch, float* __restrict__ scratch_next,
                                        uint8_t* __restrict__ pbuf_w,
                                        float* __restrict__ ref_w, const int tok0w, const int L, const int L_row,
                                        const bool need_mask FL_T_PARAMS) {
  const int li = lane & 31, lh = lane >> 5;
  __builtin_amdgcn_s_setprio(1);   // (a scheduling barrier for hipcc: it stays OUTSIDE the read / MFMA interleave below)
  // ---- S^T[32 tok x 32 rows] = K . Q^T ----
  v16f acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // the Q rope fragments (16 registers) live in LDS between steps: with them and the second accumulator resident hipcc
  // spilled a Q fragment (a scratch reload waits with vmcnt(0): the whole latency of the rope prefetch, every step)
  v8bf qr[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qr[s] = as_bf8(*reinterpret_cast<const uint4*>(qr_lds + s * 1024 + lane * 16));
  uint4 ka[8][2];
  // K operand addresses: the eight swizzled 16-B chunk addresses of this lane's token row are formed ONCE PER PAIR of steps
  // (qk_addr: 1 add + 7 xor) — the second step of a pair reads the next ring slot through the instruction's offset field
  // (koff = 32 KiB), k-steps 4..7 through + 256.  Round 4 formed them per step (16 VALU of the QK wave's 133 per page).
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    ka[s][0] = *reinterpret_cast<const uint4*>(kp + kaddr[(s & 3) * 2 + 0] + koff + (s >> 2) * 256);
    ka[s][1] = *reinterpret_cast<const uint4*>(kp + kaddr[(s & 3) * 2 + 1] + koff + (s >> 2) * 256);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(rr.ra[s]), qr[s], acc, 0, 0, 0);
#pragma unroll
  for (int s = 0; s < 8; ++s)
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(make_v8i(ka[s][0], ka[s][1]), qn[s], acc, 0, 0, 0, kUnitScale,
                                                          0, kUnitScale);
  // operand reads: k-steps 0..3 before the rope MFMAs, k-step 4 + s behind the MFMA of k-step s (four k-steps = 32
  // registers in flight: with all eight hipcc runs out of registers beside Q and the two rope buffers)
  __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);   // DS reads: Q rope fragments, k-steps 0..3
  __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);    // rope MFMAs
  // the scale triples of the lane's 16 tokens take over the operand registers of k-steps 0..3 behind the MFMAs of
  // k-steps 4..7, so that they have landed when the chain drains ({ks, log2 ks} now, 1/ks behind the scaling: 48
  // registers at once do not fit beside Q and the rope buffers)
  float4 ks4[4], lk4[4], ik4[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int tb = g * 8 + lh * 4;
    ks4[g] = *reinterpret_cast<const float4*>(scratch + tb);
    lk4[g] = *reinterpret_cast<const float4*>(scratch + 32 + tb);
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_setprio(0);
  FL_T(6);   // operand reads + MFMA issue
  // rope A operand / raw scale of page i+2 into the registers the rope MFMAs have read: token 32W + li, 16-B chunks
  // 2s + lh of its 128-B row.  UNCONDITIONAL (the caller clamps the page into the part): behind a conditional load hipcc
  // can only wait with vmcnt(0), which would expose the whole latency of the loads issued one step earlier, every step
#pragma unroll
  for (int s = 0; s < 4; ++s) rr.ra[s] = *reinterpret_cast<const uint4*>(rope_next + s * 32);
  rr.ks = *scale_next;
  __builtin_amdgcn_sched_barrier(0);
  FL_T(8);   // rope / scale load issue

  // ---- online softmax of the block; tokens of lane: 32W + 8g + 4lh + e ----
  float tmax = -INFINITY;
  if (!need_mask) {
    // PACKED f32 math (v_pk_mul_f32 / v_pk_fma_f32: two elements per instruction).  Beside a running MFMA a VALU
    // instruction of either wave of the SIMD gets an issue slot only every ~8 cycles, packed or not: what counts here is
    // the instruction COUNT of the softmax, not its flops
    const float2v qs2 = {qs, qs};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float2v y01 = __builtin_elementwise_fma(float2v{acc[g * 4 + 0], acc[g * 4 + 1]} * qs2,
                                                    float2v{ks4[g].x, ks4[g].y}, float2v{lk4[g].x, lk4[g].y});
      const float2v y23 = __builtin_elementwise_fma(float2v{acc[g * 4 + 2], acc[g * 4 + 3]} * qs2,
                                                    float2v{ks4[g].z, ks4[g].w}, float2v{lk4[g].z, lk4[g].w});
      acc[g * 4 + 0] = y01[0];
      acc[g * 4 + 1] = y01[1];
      acc[g * 4 + 2] = y23[0];
      acc[g * 4 + 3] = y23[1];
      tmax = fl_max3(fl_max3(tmax, y01[0], y01[1]), y23[0], y23[1]);   // (two v_max3_f32 per four scores)
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
        if (tok0w + tb + e >= L_row) y = -INFINITY;
        if (!(y == y)) y = -INFINITY;   // NaN can only come from garbage beyond the row's limit
        acc[g * 4 + e] = y;
        tmax = fmaxf(tmax, y);
      }
    }
  }
  FL_T(9);   // MFMA drain + scaling + max (lanes)
#pragma unroll
  for (int g = 0; g < 4; ++g) ik4[g] = *reinterpret_cast<const float4*>(scratch + 64 + g * 8 + lh * 4);
  {
    // max over the two lane halves without an LDS round trip: v_permlane32_swap exchanges lanes 32..63 of its first
    // operand with lanes 0..31 of its second
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
    tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  }
  FL_T(7);   // MFMA drain + scaling + max
  float m_new = tmax > m_w ? ceilf(tmax) + kRefHeadroom : m_w;
  float ev[16];
  int pk[4];
  auto expo = [&](const float moff) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float2v a01 = float2v{acc[g * 4 + 0], acc[g * 4 + 1]} + float2v{moff, moff};
      const float2v a23 = float2v{acc[g * 4 + 2], acc[g * 4 + 3]} + float2v{moff, moff};
      ev[g * 4 + 0] = __builtin_amdgcn_exp2f(a01[0]);
      ev[g * 4 + 1] = __builtin_amdgcn_exp2f(a01[1]);
      ev[g * 4 + 2] = __builtin_amdgcn_exp2f(a23[0]);
      ev[g * 4 + 3] = __builtin_amdgcn_exp2f(a23[1]);
      // (low half written first, high half second: the "old" operand is a score that dies here — its register becomes the
      //  destination, no v_mov 0 for the half the first conversion leaves untouched)
      const int v = __builtin_amdgcn_cvt_pk_fp8_f32(ev[g * 4 + 0], ev[g * 4 + 1], __float_as_int(acc[g * 4 + 0]), false);
      pk[g] = __builtin_amdgcn_cvt_pk_fp8_f32(ev[g * 4 + 2], ev[g * 4 + 3], v, true);
    }
  };
  expo(kPShift - m_new);
  // publish P' (16 B per lane) and the block reference for the PV waves of this row tile
  *reinterpret_cast<uint4*>(pbuf_w + lane * 16) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
  ref_w[li] = m_new;   // (identical in both lane halves)
  __builtin_amdgcn_sched_barrier(0);
  FL_T(10);  // exp + quantise + publish
  // ---- off the critical path: the two normalisers, then the next page's scale triples ----
  {
    const float f = __builtin_amdgcn_exp2f(m_w - m_new);   // exactly 1 when the reference did not move
    // two partial sums each, as PACKED f32 math (v_pk_fma_f32: nothing of this wave runs on the matrix pipe here)
    float2v l2 = {l_run * f, 0.f}, q2 = {lq_run * f, 0.f};
    m_w = m_new;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float2v ik01 = {ik4[g].x, ik4[g].y}, ik23 = {ik4[g].z, ik4[g].w};
      // unrounded sum: exact LSE
      l2 = __builtin_elementwise_fma(float2v{ev[g * 4 + 0], ev[g * 4 + 1]}, ik01, l2);
      l2 = __builtin_elementwise_fma(float2v{ev[g * 4 + 2], ev[g * 4 + 3]}, ik23, l2);
      // the ROUNDED weights normalise O (numerator and denominator use the same weights: they sum to exactly 1)
      q2 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8(pk[g], false), ik01, q2);
      q2 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8(pk[g], true), ik23, q2);
    }
    l_run = l2[0] + l2[1];
    lq_run = q2[0] + q2[1];
  }
  scale_prep(scratch_next, ks_next, tok0w + kPage, li, L);
}

// ---- PV wave: O^T[256 dims x 32 rows] += V^T(page) . P^T, with the LDS-DMA refill of a later page in the MFMA shadow ----
// The reference of O is fixed when a row sees its first valid token, kRefLift ABOVE that block's reference (a power-of-two offset: exact,
// fp32 has the range both ways): later blocks with a larger reference m_b enter with an E8M0 block scale 2^(m_b - m_o) > 1 (exact), so nothing
// but the MFMA touches O in the page loop.  A later block may exceed the first by kRefLift + kMaxUp = 128 log2 units (89 nats) that way;
// beyond that (never seen on model data; synthetic projection weights do it) the row's reference moves up IN PLACE, in an out-of-line
// block of the step: O *= 2^(m_o - m_new).  (Rounds 2-4 repeated the whole request with the final reference instead — a `redo` pass that
// doubled the request's time whenever a row's logits spread over more than 2^64: 0.87 instead of 0.47 ms per launch at BASELINE config 4
// with random projection weights.)  Overflow bound: |O| <= 2^6 (P') x 2^9 (fp8 V) x 2^17 tokens x 2^kMaxUp = 2^120.
constexpr float kMaxUp = 88.f;
constexpr float kRefLift = 40.f;
template <bool DMA, int FMT, int NRT, bool NT>
__device__ __forceinline__ void pv_step(v16f (&o)[8], float& m_o, const PvLane& lc_in, const int lane,
                                        const uint8_t* __restrict__ vp, const uint8_t* __restrict__ pbuf_rt,
                                        const float* __restrict__ ref_rt, const uint8_t* __restrict__ src_nope,
                                        uint8_t* __restrict__ dma_dst FL_T_PARAMS) {
  const int li = lane & 31, lh = lane >> 5;
  // opaque copy of the lane constants: nothing derived from them is hoisted out of the page loop (and spilled)
  PvLane lc = lc_in;
  asm volatile("" : "+v"(lc.vb0), "+v"(lc.dn_row), "+v"(lc.dn_x));
  const uint4 p0 = *reinterpret_cast<const uint4*>(pbuf_rt + lane * 16);
  const uint4 p1 = *reinterpret_cast<const uint4*>(pbuf_rt + 64 * 16 + lane * 16);
  const float m0 = ref_rt[li];
  const float m1 = ref_rt[32 + li];
  // (the refill goes out one piece behind each PV MFMA)
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
  const float mw_max = fmaxf(m0, m1);
  m_o = m_o > kNegRef ? m_o : (mw_max > kNegRef ? mw_max + kRefLift : kNegRef);
  if (__builtin_expect(__any(mw_max - m_o > kMaxUp), 0)) {
    // a block reference outran the O reference by more than kMaxUp: move this row's reference up IN PLACE — O *= 2^(m_o - m_new), exact in
    // fp32 down to the subnormals, which lie 2^-100 below the row's new maximum (hipcc keeps the 128 multiplies out of line: no copy of O on
    // the common path)
    const float m_up = mw_max - m_o > kMaxUp ? mw_max + kRefLift : m_o;
    const float f_up = __builtin_amdgcn_exp2f(m_o - m_up);   // exactly 1 for the lanes whose row did not jump
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[j][r] *= f_up;
    m_o = m_up;
  }
  int sb = 127 + (int)fminf((lh ? m1 : m0) - m_o, kMaxUp);
  sb = sb < 0 ? 0 : sb;
  const v8i pb = make_v8i(p0, p1);
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    if (jb + 3 < 8) load_vt(jb + 3);
    o[jb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va[jb], pb, o[jb], 0, 0, 0, kUnitScale, 0, sb);
    if (DMA) {
      constexpr int kPer = pieces_per_wave<NRT>() / 8;   // refill pieces behind each PV MFMA: 1 (NRT = 2) or 2
#pragma unroll
      for (int k = jb * kPer; k < (jb + 1) * kPer; ++k) fl_dma16_s_nt<NT>(src_nope, dn_off<FMT>(lc, k), dma_dst + k * 1024);
    }
  }
  // V^T operand reads three tiles ahead of their MFMA
  __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
  for (int jb = 0; jb < 5; ++jb) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
  }
  __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
  __builtin_amdgcn_sched_barrier(0);
}

// ---- QK wave, plain-fp8 [.,576] format (FL_KV_FP8_576, flash_mla_with_kvcache over an fp8 cache): the 64 rope dims are a
//      NINTH fp8 k-step whose A operand comes straight from bytes 512..575 of the token's row in global memory (two pages
//      ahead, like the bf16 rope of the per-token format); there are no per-token scales — the device-scalar descales are
//      folded into qs (scores) and into the epilogue (V).  Same block-reference / P' arithmetic as qk_step. ----
__device__ __forceinline__ void qk_step1(float& l_run, float& lq_run, float& m_w, const QkLane& lc, const int lane,
                                         const v8i (&qn)[8], const v8i& qr8, const float qs, RopeRegs& rr,
                                         const uint8_t* __restrict__ rope_next, const uint8_t* __restrict__ kp,
                                         uint8_t* __restrict__ pbuf_w, float* __restrict__ ref_w, const int tok0w,
                                         const int L_row, const bool need_mask) {
  const int li = lane & 31, lh = lane >> 5;
  __builtin_amdgcn_s_setprio(1);
  v16f acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  uint4 ka[8][2];
  int kb0 = lc.kb0;
  asm volatile("" : "+v"(kb0));
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    ka[s][0] = *reinterpret_cast<const uint4*>(kp + (kb0 ^ ((s & 3) << 6)) + (s >> 2) * 256);
    ka[s][1] = *reinterpret_cast<const uint4*>(kp + (kb0 ^ ((s & 3) << 6) ^ 16) + (s >> 2) * 256);
  }
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(make_v8i(rr.ra[0], rr.ra[1]), qr8, acc, 0, 0, 0, kUnitScale, 0, kUnitScale);
#pragma unroll
  for (int s = 0; s < 8; ++s)
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(make_v8i(ka[s][0], ka[s][1]), qn[s], acc, 0, 0, 0, kUnitScale, 0,
                                                          kUnitScale);
  __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);    // DS reads: k-steps 0..3
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // rope k-step
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
  }
  __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_setprio(0);
  // rope bytes of page i+2 into the registers the rope MFMA has read (UNCONDITIONAL, the caller clamps the page)
  rr.ra[0] = *reinterpret_cast<const uint4*>(rope_next);
  rr.ra[1] = *reinterpret_cast<const uint4*>(rope_next + 16);
  __builtin_amdgcn_sched_barrier(0);
  // ---- online softmax of the block; tokens of lane: 32W + 8g + 4lh + e ----
  float tmax = -INFINITY;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int tb = g * 8 + lh * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float y = acc[g * 4 + e] * qs;
      if (need_mask) {
        if (tok0w + tb + e >= L_row) y = -INFINITY;
        if (!(y == y)) y = -INFINITY;   // NaN can only come from garbage beyond the row's limit
      }
      acc[g * 4 + e] = y;
      tmax = fmaxf(tmax, y);
    }
  }
  {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
    tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  }
  const float m_new = tmax > m_w ? ceilf(tmax) + kRefHeadroom : m_w;
  const float moff = kPShift - m_new;
  float ev[16];
  int pk[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int e = 0; e < 4; ++e) ev[g * 4 + e] = __builtin_amdgcn_exp2f(acc[g * 4 + e] + moff);
    const int v = __builtin_amdgcn_cvt_pk_fp8_f32(ev[g * 4 + 0], ev[g * 4 + 1], 0, false);
    pk[g] = __builtin_amdgcn_cvt_pk_fp8_f32(ev[g * 4 + 2], ev[g * 4 + 3], v, true);
  }
  *reinterpret_cast<uint4*>(pbuf_w + lane * 16) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
  ref_w[li] = m_new;
  __builtin_amdgcn_sched_barrier(0);
  {
    const float f = __builtin_amdgcn_exp2f(m_w - m_new);
    float2v l2 = {l_run * f, 0.f}, q2 = {lq_run * f, 0.f};
    m_w = m_new;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      l2 = l2 + float2v{ev[g * 4 + 0], ev[g * 4 + 1]} + float2v{ev[g * 4 + 2], ev[g * 4 + 3]};
      q2 = q2 + __builtin_amdgcn_cvt_pk_f32_fp8(pk[g], false) + __builtin_amdgcn_cvt_pk_f32_fp8(pk[g], true);
    }
    l_run = l2[0] + l2[1];
    lq_run = q2[0] + q2[1];
  }
}

// QBF: the query arrives unquantised (bf16) and K4 runs in the QK waves' request prologue.  NT: the latent pages go through the LDS-DMA with the
// non-temporal policy — for launches in which ONE workgroup reads a page (one row group per request: H <= 64 at s_q = 1): measured -6.4 % at H = 64,
// -8 % at H = 16; with two row groups per request (H = 128) the second reader loses part of its L2 hits (HBM traffic 1.09 x instead of 1.006 x
// algorithmic, +-1 % in time): default policy there (profiles/r05_nt_policy.txt)
template <int FMT, bool QBF = false, int NRT = 2, bool NT = false>
__global__ __launch_bounds__(256 * NRT) void mla_decode_y_kernel(
    const Params p, const int32_t* __restrict__ g_block_table, const int32_t* __restrict__ g_seqlens,
    const int32_t* __restrict__ g_meta, int32_t* g_merge_ctr, const int32_t* __restrict__ g_num_splits,
    const uint8_t* __restrict__ g_k_nope, const uint16_t* __restrict__ g_k_rope, const float* __restrict__ g_k_scale,
    const uint8_t* __restrict__ g_q_nope, const uint16_t* __restrict__ g_q_rope, const float* __restrict__ g_q_scale) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[kLdsBytes];

  const int tid = threadIdx.x;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int kRoleWaves = 2 * NRT;                      // QK waves = PV waves = 2 per row tile
  constexpr int kPiecesPerWave = pieces_per_wave<NRT>();
  const bool is_pv = wave_id >= kRoleWaves;
  const int w4 = is_pv ? wave_id - kRoleWaves : wave_id;   // index of the wave inside its role
  const int rt = NRT == 2 ? (w4 & 1) : 0;                  // row tile (32 query rows) inside the workgroup
  const int W = NRT == 2 ? (w4 >> 1) : w4;                 // QK wave: token half; PV wave: d half
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

  const int row = rgrp * (32 * NRT) + rt * 32 + li;   // query row of this lane (both roles: lane&31 = row inside the tile)
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
    /* UNCONDITIONAL load on an index clamped into the request's table row: the load does not wait for the sequence  */ \
    /* length (n) — at a request's start both go out together, one memory round trip less in the prologue chain       */ \
    const long long col = tile_b + t < p.bt_cols ? tile_b + t : p.bt_cols - 1;                                         \
    const int pg = g_block_table[(long long)req * p.bt_stride + (col < 0 ? 0 : col)];                                  \
    pg_vec = (t >= n || pg < 0 || pg >= p.num_pages) ? 0 : pg;                                                         \
  };                                                                                                                   \
  load_window(0);                                                                                                      \
  auto page_of = [&](int t) { return (long long)__builtin_amdgcn_readlane(pg_vec, t - win_base); }

  if (!is_pv) {
    // =========================== QK waves ===========================
    QkLane lc;
    {
      // K operand: token T = 32W + li, 32 B at d = 64s + 32lh -> 16-B chunks c = 4s + 2lh + e, stored at chunk c ^ (T&15)
      const int kx = li & 15;
      lc.kb0 = li * kDN + (((((kx >> 2)) << 2) | ((2 * lh) ^ (kx & 3))) << 4);
    }
    float* scratch = reinterpret_cast<float*>(smem + kOffScratch + w4 * kScratchPerWave);
    for (; req < p.bs; ++req, tile_b = 0) {
      FL_Y_REQUEST_HEAD();
      // Q fragments (B operands), once per request
      const long long qrow = (long long)req * p.rows + row;
      v8i qn[8];
      v8i qr8 = v8i{0, 0, 0, 0, 0, 0, 0, 0};   // FMT 1: the rope k-step of Q (fp8)
      uint8_t* qr_lds = smem + kOffQr + w4 * 4096;
      float qs = 0.f;
      constexpr int kQRow = FMT == 1 ? kDN + kDR : kDN;   // bytes per query row (FMT 1: one fp8 [.,576] tensor)
      if constexpr (QBF) {
        // K4 (quantize_ckv_per_token_head, flashmla_backend.py:198-206; arithmetic of mla_quant.hip) on this lane's half of its
        // row: k-step s needs latent elements [64 s + 32 lh, + 32) as fp8 = 64 B of bf16; the row maximum spans both lane halves.
        // The conversion (~1,000 VALU instructions) runs while the PV waves wait for the request's first page.
        const uint16_t* qb = p.q_bf16 + (row_ok ? qrow : 0) * (kDN + kDR);   // (clamped: the loads are unconditional)
        // pass 1: the row maximum — this lane's 256 elements streamed through 16 registers per k-step, then the other lane half's
        float amax = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          uint4 raw[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const uint4*>(qb + 64 * s + 32 * lh + 8 * u);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t w4[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
              amax = fmaxf(amax, fmaxf(fabsf(__uint_as_float(w4[q4] << 16)), fabsf(__uint_as_float(w4[q4] & 0xffff0000u))));
          }
        }
        {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(amax), __float_as_uint(amax), false, false);
          amax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const float qscale = fmaxf(amax, 1e-26f) / FL_FP8_MAX;
        // pass 2: the same 64 B per k-step again (L2 hits), divided by the scale -> the k-step's 32 fp8 bytes; one k-step at a time
        // (kept apart by scheduling barriers: with all eight in flight hipcc spilled 506 registers)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          uint4 raw[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const uint4*>(qb + 64 * s + 32 * lh + 8 * u);
          uint2 w[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t w4[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
            float v[8];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) { v[2 * q4] = __uint_as_float(w4[q4] << 16); v[2 * q4 + 1] = __uint_as_float(w4[q4] & 0xffff0000u); }
            w[u] = fl_div8_to_fp8<false>(v, qscale);
          }
          qn[s] = row_ok ? v8i{(int)w[0].x, (int)w[0].y, (int)w[1].x, (int)w[1].y, (int)w[2].x, (int)w[2].y, (int)w[3].x, (int)w[3].y}
                         : v8i{0, 0, 0, 0, 0, 0, 0, 0};
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const uint4 rr = *reinterpret_cast<const uint4*>(qb + kDN + lh * 8 + s * 16);
          const uint32_t w4[4] = {rr.x, rr.y, rr.z, rr.w};
          uint32_t o4[4];
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4)
            o4[q4] = (uint32_t)fl_f32_to_bf16(__uint_as_float(w4[q4] << 16) / qscale) |
                     ((uint32_t)fl_f32_to_bf16(__uint_as_float(w4[q4] & 0xffff0000u) / qscale) << 16);
          *reinterpret_cast<uint4*>(qr_lds + s * 1024 + lane * 16) = row_ok ? make_uint4(o4[0], o4[1], o4[2], o4[3]) : make_uint4(0, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        qs = row_ok ? qscale * p.scale_log2e : 0.f;
      } else if (row_ok) {
        const uint8_t* qp = g_q_nope + qrow * kQRow + lh * 32;
#pragma unroll
        for (int s = 0; s < 8; ++s)
          qn[s] = make_v8i(*reinterpret_cast<const uint4*>(qp + s * 64), *reinterpret_cast<const uint4*>(qp + s * 64 + 16));
        if constexpr (FMT == 1) {
          qr8 = make_v8i(*reinterpret_cast<const uint4*>(qp + 512), *reinterpret_cast<const uint4*>(qp + 512 + 16));
          qs = (p.descale_q ? *p.descale_q : 1.f) * (p.descale_k ? *p.descale_k : 1.f) * p.scale_log2e;
        } else {
          const uint16_t* rp = g_q_rope + qrow * kDR + lh * 8;
#pragma unroll
          for (int s = 0; s < 4; ++s)
            *reinterpret_cast<uint4*>(qr_lds + s * 1024 + lane * 16) = *reinterpret_cast<const uint4*>(rp + s * 16);
          qs = g_q_scale[qrow] * p.scale_log2e;
        }
      } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) qn[s] = v8i{0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (FMT == 0) {
#pragma unroll
          for (int s = 0; s < 4; ++s) *reinterpret_cast<uint4*>(qr_lds + s * 1024 + lane * 16) = make_uint4(0, 0, 0, 0);
        }
      }
      int L_row = L;
      if (p.causal) L_row = L - (p.s_q - 1 - row / p.h_q);   // query j sees keys [0, L - (s_q-1-j))
      if (!row_ok) L_row = 0;
      const int L_min = p.causal ? L - (p.s_q - 1) : L;

      // rope A operand of this lane: token 32W + li, 16-B chunks 2s + lh of its 128-B row; raw scale of token 32W + li.
      // Every load is UNCONDITIONAL (page index clamped into the part; an empty part reads the padding page 0).
      auto rope_src = [&](const int t) {
        if constexpr (FMT == 1)   // bytes 512 + 32 lh .. of the token's 576-B row: the rope k-step's A operand
          return g_k_nope + (page_of(t) * kPage + 32 * W + li) * (long long)(kDN + kDR) + kDN + lh * 32;
        return reinterpret_cast<const uint8_t*>(g_k_rope) + (page_of(t) * kPage + 32 * W + li) * (kDR * 2) + lh * 16;
      };
      auto scale_src = [&](const int t) { return g_k_scale + page_of(t) * kPage + 32 * W + li; };
      auto load_rope = [&](RopeRegs& r, const int t) {
        const uint8_t* rp = rope_src(t);
        if constexpr (FMT == 1) {
          r.ra[0] = *reinterpret_cast<const uint4*>(rp);
          r.ra[1] = *reinterpret_cast<const uint4*>(rp + 16);
          r.ks = 1.f;
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) r.ra[s] = *reinterpret_cast<const uint4*>(rp + s * 32);
          r.ks = *scale_src(t);
        }
      };
      {
        float l_run = 0.f, lq_run = 0.f, m_w = kNegRef;
        RopeRegs rA, rB;
        load_rope(rA, 0);
        load_rope(rB, n > 1 ? 1 : 0);
        if constexpr (FMT == 0)
          scale_prep(scratch, rA.ks, tile_b * kPage + 32 * W, li, L);   // page 0 -> parity 0 (this wave's reads of the previous request are done)

        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // R0: every wave is done with the LDS of the previous request
        FL_T(5);   // request prologue

        // step i uses `rr` (page i) and refills it with page i+2; `rn` holds page i+1
        int kaddr[8];
        auto step = [&](const int i, RopeRegs& rr, const RopeRegs& rn, const int koff) {
          if (i + 2 >= win_base + 64 && i + 2 < n) load_window(i + 2);   // pages i+2 .. i+65
          const int t2 = i + 2 < n ? i + 2 : n - 1;
          const uint8_t* rope_next = rope_src(t2);
          const float* scale_next = FMT == 0 ? scale_src(t2) : nullptr;
          FL_T(2);   // (loop control)
          __builtin_amdgcn_s_barrier();   // B_i: page i landed, P buffers of parity i free
          FL_T(0);   // barrier
          const int tok0w = (tile_b + i) * kPage + 32 * W;
          const bool need_mask = (tile_b + i) * kPage + kPage > L_min;
          if constexpr (FMT == 1)
            qk_step1(l_run, lq_run, m_w, lc, lane, qn, qr8, qs, rr, rope_next,
                     smem + kOffRing + (i & 3) * kSlotBytes + W * (32 * kDN),
                     smem + kOffPbuf + (i & 1) * kPbufPerParity + (rt * 2 + W) * (64 * 16),
                     reinterpret_cast<float*>(smem + kOffRef + (i & 1) * kRefPerParity) + (rt * 2 + W) * 32, tok0w, L_row, need_mask);
          else
          qk_step(l_run, lq_run, m_w, kaddr, koff, lane, qn, qr_lds, qs, rr, rn.ks, rope_next, scale_next,
                  smem, scratch + (i & 1) * 96, scratch + ((i + 1) & 1) * 96,
                  smem + kOffPbuf + (i & 1) * kPbufPerParity + (rt * 2 + W) * (64 * 16),
                  reinterpret_cast<float*>(smem + kOffRef + (i & 1) * kRefPerParity) + (rt * 2 + W) * 32, tok0w, L, L_row,
                  need_mask FL_T_ARGS);
          FL_T(1);   // softmax tail + publish (issue)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // P', reference published before the next barrier
          FL_T(3);   // LDS drain
        };
        {
          // pairs of steps without a condition in between (a skipped second step would leave a path with nothing issued
          // behind rA's loads: vmcnt(0) again), then the odd tail
          int i = 0;
          // (i even: ring slot i & 3 is 0 or 2; the odd step of the pair reads slot + 1 through the offset field)
          for (; i + 1 < n; i += 2) {
            qk_addr(kaddr, lc.kb0, kOffRing + (i & 2) * kSlotBytes + W * (32 * kDN));
            step(i, rA, rB, 0);
            step(i + 1, rB, rA, kSlotBytes);
          }
          if (i < n) {
            qk_addr(kaddr, lc.kb0, kOffRing + (i & 2) * kSlotBytes + W * (32 * kDN));
            step(i, rA, rB, 0);
          }
        }
        __builtin_amdgcn_s_barrier();   // B_n: the PV waves run PV(n-1)
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
    lc.dn_row = (unsigned)((w4 * kPiecesPerWave * 2 + lh) * (FMT == 1 ? kDN + kDR : kDN));
    lc.dn_x = (unsigned)((li ^ lh) << 4);
  }
  int* msync = reinterpret_cast<int*>(smem + kOffMerge);
  if (w4 == 0 && lane == 0) msync[0] = 0;   // (first PV wave; first use behind the first request's barriers)
  // "my partial rows of a split request are in memory" signal of this wave, owed to the request's merging piece: sent behind the
  // NEXT s_waitcnt vmcnt(0) the wave executes anyway (the next request's R0, or the end of the workgroup)
  int* pend_ctr = nullptr;
  int pend_add = 0;
  bool first_item = true;
  for (; req < p.bs; ++req, tile_b = 0, split_idx = 0, first_item = false) {
    FL_Y_REQUEST_HEAD();
    const int split_base = g_num_splits[req];
    const int nsplit = g_num_splits[req + 1] - split_base;
    const bool is_split = nsplit > 1;
    auto ring = [&](int t) { return smem + kOffRing + (t & 3) * kSlotBytes; };
    auto src_of = [&](int t) { return g_k_nope + page_of(t) * (long long)(kPage * (FMT == 1 ? kDN + kDR : kDN)); };
    auto issue_page = [&](int t) {
      const uint8_t* sn = src_of(t);
      uint8_t* dst = ring(t) + w4 * (kPiecesPerWave * 1024);
#pragma unroll
      for (int k = 0; k < kPiecesPerWave; ++k) fl_dma16_s_nt<NT>(sn, dn_off<FMT>(lc, k), dst + k * 1024);
    };
    const float* lm = reinterpret_cast<const float*>(smem + kOffLm) + rt * 192;

    v16f o[8];
    float m_o = kNegRef;
    {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
      m_o = kNegRef;

      if (first_item) {
        // the workgroup's first request: nobody has touched the ring yet — the first two pages go out as soon as their ids are
        // here, not behind the QK waves' rope / scale loads (R0): one memory round trip less in the start-up chain
        if (n > 0) issue_page(0);
        if (n > 1) issue_page(1);
        __builtin_amdgcn_s_barrier();   // R0
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (pend_ctr != nullptr) {   // (the partial rows of the previous request have been acknowledged)
          if (lane == 0) __hip_atomic_fetch_add(pend_ctr, pend_add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          pend_ctr = nullptr;
        }
        __builtin_amdgcn_s_barrier();   // R0
        if (n > 0) issue_page(0);
        if (n > 1) issue_page(1);
      }
      FL_T(5);   // request prologue

      // step i: wait for page i, barrier B_i, PV(i-1) with the refill of page i+2 in its MFMA shadow, tail fill of page i.
      // The steps with a refill and the last two without are separate loops, each with ONE step variant (two variants
      // joined inside a loop make hipcc copy the O registers at the join).
#define FL_Y_PV_STEP(HAS_PREV, HAS_DMA)                                                                                \
  {                                                                                                                    \
    if (i + 1 < n) { /* page i landed, page i+1 (this wave's kPiecesPerWave pieces) may stay in flight */               \
      if constexpr (NRT == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                        \
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                                          \
    }                                                                                                                  \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                 \
    FL_T(0); /* page-landed wait */                                                                                    \
    __builtin_amdgcn_s_barrier(); /* B_i */                                                                            \
    FL_T(1); /* barrier */                                                                                             \
    if (HAS_DMA && i + 2 >= win_base + 64) load_window(i + 2);                                                         \
    const uint8_t* sn = HAS_DMA ? src_of(i + 2) : nullptr;                                                             \
    uint8_t* dst = ring(i + 2) + w4 * (kPiecesPerWave * 1024);                                                         \
    if (HAS_PREV) {                                                                                                    \
      pv_step<HAS_DMA, FMT, NRT, NT>(o, m_o, lc, lane, ring(i - 1) + W * 256,                                                  \
                       smem + kOffPbuf + ((i - 1) & 1) * kPbufPerParity + rt * (2 * 64 * 16),                          \
                       reinterpret_cast<const float*>(smem + kOffRef + ((i - 1) & 1) * kRefPerParity) + rt * 64, sn,   \
                       dst FL_T_ARGS);                                                                                 \
    } else {                                                                                                           \
      if (HAS_DMA) {                                                                                                   \
        _Pragma("unroll") for (int k = 0; k < kPiecesPerWave; ++k) fl_dma16_s_nt<NT>(sn, dn_off<FMT>(lc, k), dst + k * 1024);      \
      }                                                                                                                \
    }                                                                                                                  \
    /* tail of the sequence: zero the rows of page i past the end (P' is exactly 0 there, but 0 * NaN from stale fp8   \
       NaN patterns would poison the PV MFMA of the next step); the QK waves mask those tokens by index */             \
    if (i < n && (tile_b + i) * kPage + kPage > L) {                                                                   \
      const int nvalid = L - (tile_b + i) * kPage;                                                                     \
      uint8_t* wr = ring(i);                                                                                           \
      _Pragma("clang loop vectorize(disable) unroll(disable)")                                                         \
      for (int T = nvalid + 2 * w4 + lh; T < kPage; T += 2 * kRoleWaves)                                               \
        *reinterpret_cast<uint4*>(wr + T * kDN + li * 16) = make_uint4(0, 0, 0, 0);                                    \
    }                                                                                                                  \
    FL_T(2); /* PV + refill issue */                                                                                   \
  }
      {
        int i = 0;
        if (n > 2) FL_Y_PV_STEP(false, true) else FL_Y_PV_STEP(false, false)
        for (i = 1; i + 2 < n; ++i) FL_Y_PV_STEP(true, true)
        for (; i <= n; ++i) FL_Y_PV_STEP(true, false)
      }
#undef FL_Y_PV_STEP
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // E0: the normalisers of the QK waves are in LDS, the ring is free
      FL_T(3);   // E0
    }

    // ---- per-request epilogue: merge the normalisers of the two blocks, normalise, store this wave's d half ----
    const float mA = lm[64 + li], mB = lm[96 + 64 + li];
    const float fA = __builtin_amdgcn_exp2f(mA - m_o), fB = __builtin_amdgcn_exp2f(mB - m_o);   // <= 2^kMaxUp
    const float l = lm[li] * fA + lm[96 + li] * fB;
    const float lq = lm[32 + li] * fA + lm[96 + 32 + li] * fB;
    // (FMT 1: V = fp8 x descale_k, one device scalar for the whole cache)
    const float inv = (lq > 0.f ? 1.f / lq : 0.f) * (FMT == 1 ? (p.descale_k ? *p.descale_k : 1.f) : 1.f);
    const float lse_nat = l > 0.f ? (__builtin_amdgcn_logf(l) + m_o - kPShift) * 0.6931471805599453f : -INFINITY;
    // split-KV partials are normalised by lq, so they are COMBINED with lq-based weights; the exact LSE travels along
    const float lseq_nat = lq > 0.f ? (__builtin_amdgcn_logf(lq) + m_o - kPShift) * 0.6931471805599453f : -INFINITY;
    const int slot_idx = split_base + split_idx;
    // In-kernel split merge (no second kernel): the request's FIRST piece merges.  The pieces of a request lie in consecutive
    // parts, the first piece is the LAST item of its part and every other piece is the first (or only) item of a later part, so
    // the merging piece normally finishes last; it keeps its own rows in registers / LDS (they never travel through memory),
    // waits until the other pieces' PV waves have signalled (4 per piece on the (request, row group) counter in the spare
    // metadata columns of the merging part: 16 bits per row group, K3 zeroes them, the merging workgroup puts them back), reads
    // their bf16 rows and writes the final rows.  A waiting piece never waits for a piece that itself waits (only LAST items
    // wait, only on FIRST / only items of later parts), so any dispatch order makes progress.
    // Hand-off form (MI355X guide, "handoff-flag": write-through payload -> drained vmcnt -> counter -> sc1 loads): partial
    // rows / LSE pairs are stored with the agent-scope policy (sc1 = write-through past this XCD's non-coherent L2) and read
    // back with sc1 loads, served by the memory side — no cache-wide write-back / invalidate on either side (validated in
    // rounds 2-3: tools/determinism_ragged.py).  The arithmetic is combine_rows16's / the merge kernel's, piece by piece in order.
    const bool merger = is_split && p.merge_in_kernel && split_idx == 0;
    if (row_ok && lh == 0 && W == 0) {
      // (the row index is made opaque HERE: derived 64-bit offsets — row / h_q, the LSE slots — are otherwise formed before the page
      //  loop and spilled across it)
      unsigned row_e = (unsigned)row;
      asm volatile("" : "+v"(row_e));
      if (is_split) {
        if (!merger) {   // (agent-scope stores: read by the merging piece / the merge kernel)
          float* la = p.lse_accum + (long long)slot_idx * p.rows * 2;
          st_agent_f32(la + row_e * 2u + 0, lseq_nat);
          st_agent_f32(la + row_e * 2u + 1, lse_nat);
        }
      } else {
        const unsigned j = row_e / (unsigned)p.h_q, h = row_e - j * (unsigned)p.h_q;
        p.lse[((long long)req * p.h_q + h) * p.s_q + j] = lse_nat;
      }
    }
    // O -> memory through a wave-private LDS transpose (the ring is free now).  A lane holds ONE row, 4 dims at a time:
    // stored directly, every store instruction would touch 64 rows x 8 B.  C row i = e + 8g + 4lh of tile 4c + jq is
    // d = 128c + 16jq + (i&15) + 64(i>>4) of the wave's 256 dims: the normalised values are rounded to bf16 (v_cvt_pk_bf16_f32,
    // RNE) and staged as [32 rows][256 bf16] (+16 B pad), then leave as 256-B row segments, 4 rows per store instruction.
    {
      uint8_t* stg = smem + kOffRing + w4 * (32 * kStgBytes);
      const int row0 = rgrp * (32 * NRT) + rt * 32;
      int* ctr = g_merge_ctr + (long long)part * FL_MLA_META_W + 5 + (rgrp >> 1);   // (merging piece: split_idx == 0)
      // first look at the counter: issued here, used behind the staging (its round trip runs under the conversions)
      int arrived = merger ? __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      uint16_t* dst = (is_split && !merger) ? reinterpret_cast<uint16_t*>(p.o_accum) + ((long long)slot_idx * p.rows + row0) * kDN
                                            : p.out + ((long long)req * p.rows + row0) * kDN;
      // the row addresses as ONE uniform base (SGPR pair, pinned behind the page loop) + 32-bit lane offsets: hipcc otherwise
      // forms 64-bit row pointers BEFORE the page loop (they depend on the request only) and, with the PV wave at 256
      // registers, spills them
      {
        unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)dst);
        unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)dst >> 32));
        asm volatile("" : "+s"(lo), "+s"(hi));   // (computed HERE, after the loop)
        dst = reinterpret_cast<uint16_t*>(((unsigned long long)hi << 32) | lo);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int jq = 0; jq < 4; ++jq)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int d_off = 128 * c + 16 * jq + 8 * (g & 1) + 4 * lh + 64 * (g >> 1);
            *reinterpret_cast<uint2*>(stg + li * kStgBytes + 2 * d_off) =
                make_uint2(fl_pack_bf16(o[4 * c + jq][g * 4 + 0] * inv, o[4 * c + jq][g * 4 + 1] * inv),
                           fl_pack_bf16(o[4 * c + jq][g * 4 + 2] * inv, o[4 * c + jq][g * 4 + 3] * inv));
          }
      // (the lane index is made opaque HERE: everything derived from it below — row numbers, row predicates, the pieces' offsets — would
      //  otherwise be formed before the page loop, they depend on the request only, and spilled across it)
      int lane_q = lane;
      asm volatile("" : "+v"(lane_q));
      unsigned voff = (unsigned)(256 * W + (lane_q & 15) * 8 + (lane_q >> 4) * kDN);   // elements; row r = (lane >> 4) + 4 k: + 4 k kDN; half c: + 128
      asm volatile("" : "+v"(voff));
      const uint8_t* rd = stg + (lane_q >> 4) * kStgBytes + (lane_q & 15) * 16;          // + 4 k kStgBytes + 256 c
      if (!merger) {
        // all sixteen row segments come back from LDS before the first store: left to itself hipcc sinks each read under its
        // store's row predicate (read -> wait -> store, sixteen times in series)
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        u32x4 ov[2][8];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int k = 0; k < 8; ++k) ov[c][k] = *reinterpret_cast<const u32x4*>(rd + 4 * k * kStgBytes + 256 * c);
#pragma unroll
        for (int c = 0; c < 2; ++c)
          asm volatile("" : "+v"(ov[c][0]), "+v"(ov[c][1]), "+v"(ov[c][2]), "+v"(ov[c][3]), "+v"(ov[c][4]), "+v"(ov[c][5]), "+v"(ov[c][6]), "+v"(ov[c][7]));
        if (is_split) {
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < 8; ++k)
              if (row0 + (lane_q >> 4) + 4 * k < p.rows)
                st_agent_16B(dst + (voff + (unsigned)(4 * k * kDN + 128 * c)), make_uint4(ov[c][k][0], ov[c][k][1], ov[c][k][2], ov[c][k][3]));
        } else {
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < 8; ++k)
              if (row0 + (lane_q >> 4) + 4 * k < p.rows)
                *reinterpret_cast<u32x4*>(dst + (voff + (unsigned)(4 * k * kDN + 128 * c))) = ov[c][k];
        }
        if (is_split && p.merge_in_kernel) {   // owed: one count per PV wave, behind the wave's next s_waitcnt vmcnt(0)
          pend_ctr = g_merge_ctr + (long long)(part - split_idx) * FL_MLA_META_W + 5 + (rgrp >> 1);
          pend_add = 1 << (16 * (rgrp & 1));
        }
      } else {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const int sh = 16 * (rgrp & 1);
        const int tgt = kRoleWaves * (nsplit - 1);   // one count per PV wave of every other piece
        FL_T(4);   // epilogue
        // BOUNDED poll: the counter must be exactly 0 at entry (K3 zeroes it, the merging workgroup puts it back) and launches that
        // share one metadata tensor must be ordered on one stream (include/fluent_mi355.h) — a stale or double count (an aborted launch, two
        // concurrent launches on one tensor) would otherwise spin forever.  After kMergeTimeoutTicks of the 100 MHz wall clock the merge gives
        // up and POISONS its rows and LSEs with NaN: a visible wrong answer, not a hung GPU.
        bool merge_timed_out = false;
        {
          const unsigned long long t_poll = wall_clock64();
          while (((arrived >> sh) & 0xffff) != tgt) {
            __builtin_amdgcn_s_sleep(2);
            arrived = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (wall_clock64() - t_poll > g_merge_timeout_ticks) { merge_timed_out = true; break; }
          }
        }
        if (merge_timed_out && lane == 0) {
          unsigned* e = g_merge_err;
          if (e != nullptr) __hip_atomic_store(e, 1u + (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        FL_T(7);   // merging piece: wait for the other pieces
        if (lane == 0) {   // the last of the four PV waves past the poll puts the counter back (the same metadata serves every layer's launch)
          const int old = __hip_atomic_fetch_add(msync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (old % kRoleWaves == kRoleWaves - 1) __hip_atomic_fetch_sub(ctr, tgt << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const uint16_t* parts = reinterpret_cast<const uint16_t*>(p.o_accum);
        auto ld16 = [&](const uint16_t* src) {
          const unsigned long long* q8 = reinterpret_cast<const unsigned long long*>(src);
          const unsigned long long lo = __hip_atomic_load(q8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long hi = __hip_atomic_load(q8 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          return u32x4{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
        };
        // every load of the common case (2 pieces; up to 5 for the weights) goes out before the first use: ONE round trip
        const long long rowg = row_ok ? row : p.rows - 1;   // this lane's own row (li), clamped: loads are unconditional
        float a1[4], b1[4];   // {weight LSE, exact LSE} of pieces 1..4 for this lane's row (clamped duplicates beyond the last)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int sj = 1 + j < nsplit ? 1 + j : nsplit - 1;
          const float* la = p.lse_accum + ((long long)(split_base + sj) * p.rows + rowg) * 2;
          a1[j] = ld_agent_f32(la);
          b1[j] = ld_agent_f32(la + 1);
        }
        // row r = (lane >> 4) + 4 k of the tile, 8 dims per lane: element offset of this lane's chunk inside a piece's rows
        unsigned poff[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int rr = row0 + (lane_q >> 4) + 4 * k;
          poff[k] = (unsigned)((rr < p.rows ? rr : p.rows - 1) * kDN + 256 * W + (lane_q & 15) * 8);
        }
        u32x4 ch[2][8];
        {
          const uint16_t* p1 = parts + (long long)(split_base + 1) * p.rows * kDN;
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < 8; ++k) ch[c][k] = ld16(p1 + poff[k] + 128 * c);
        }
        float mx = lseq_nat, mxx = lse_nat;
#pragma unroll
        for (int j = 0; j < 4; ++j) { mx = fmaxf(mx, a1[j]); mxx = fmaxf(mxx, b1[j]); }
        for (int s5 = 5; s5 < nsplit; ++s5) {
          const float* la = p.lse_accum + ((long long)(split_base + s5) * p.rows + rowg) * 2;
          mx = fmaxf(mx, ld_agent_f32(la));
          mxx = fmaxf(mxx, ld_agent_f32(la + 1));
        }
        const float w0 = mx == -INFINITY ? 0.f : __expf(lseq_nat - mx);
        float den = 0.f, denx = 0.f;
        den += w0;
        denx += mxx == -INFINITY ? 0.f : __expf(lse_nat - mxx);
        float wj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool live = 1 + j < nsplit;
          wj[j] = (!live || mx == -INFINITY) ? 0.f : __expf(a1[j] - mx);
          den += wj[j];
          denx += (!live || mxx == -INFINITY) ? 0.f : __expf(b1[j] - mxx);
        }
        for (int s5 = 5; s5 < nsplit; ++s5) {
          const float* la = p.lse_accum + ((long long)(split_base + s5) * p.rows + rowg) * 2;
          den += mx == -INFINITY ? 0.f : __expf(ld_agent_f32(la) - mx);
          denx += mxx == -INFINITY ? 0.f : __expf(ld_agent_f32(la + 1) - mxx);
        }
        const float invd = merge_timed_out ? __builtin_nanf("") : (den > 0.f ? 1.f / den : 0.f);
        if (row_ok && lh == 0 && W == 0) {
          const unsigned j = (unsigned)row / (unsigned)p.h_q, h = (unsigned)row - j * (unsigned)p.h_q;
          p.lse[((long long)req * p.h_q + h) * p.s_q + j] = merge_timed_out ? __builtin_nanf("") : (denx > 0.f ? mxx + __logf(denx) : -INFINITY);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          float acc[8][8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int r = (lane_q >> 4) + 4 * k;
            const uint4 own = *reinterpret_cast<const uint4*>(rd + 4 * k * kStgBytes + 256 * c);
            const uint32_t ow[4] = {own.x, own.y, own.z, own.w};
            const float w0r = __shfl(w0, r), w1r = __shfl(wj[0], r);
#pragma unroll
            for (int h = 0; h < 4; ++h) {
              acc[k][2 * h] = 0.f;
              acc[k][2 * h + 1] = 0.f;
              acc[k][2 * h] += w0r * __uint_as_float(ow[h] << 16);
              acc[k][2 * h + 1] += w0r * __uint_as_float(ow[h] & 0xffff0000u);
              acc[k][2 * h] += w1r * __uint_as_float(ch[c][k][h] << 16);
              acc[k][2 * h + 1] += w1r * __uint_as_float(ch[c][k][h] & 0xffff0000u);
            }
          }
          for (int s2 = 2; s2 < nsplit; ++s2) {   // third and later pieces: one round trip per (piece, half)
            float ws = s2 == 2 ? wj[1] : s2 == 3 ? wj[2] : wj[3];
            if (s2 > 4) {
              const float v = ld_agent_f32(p.lse_accum + ((long long)(split_base + s2) * p.rows + rowg) * 2);
              ws = mx == -INFINITY ? 0.f : __expf(v - mx);
            }
            const uint16_t* ps = parts + (long long)(split_base + s2) * p.rows * kDN;
            u32x4 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = ld16(ps + poff[k] + 128 * c);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const float wr = __shfl(ws, (lane_q >> 4) + 4 * k);
#pragma unroll
              for (int h = 0; h < 4; ++h) {
                acc[k][2 * h] += wr * __uint_as_float(t[k][h] << 16);
                acc[k][2 * h + 1] += wr * __uint_as_float(t[k][h] & 0xffff0000u);
              }
            }
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int r = (lane_q >> 4) + 4 * k;
            const float ir = __shfl(invd, r);
            if (row0 + r < p.rows)
              *reinterpret_cast<uint4*>(dst + (voff + (unsigned)(4 * k * kDN + 128 * c))) =
                  make_uint4(fl_pack_bf16(acc[k][0] * ir, acc[k][1] * ir), fl_pack_bf16(acc[k][2] * ir, acc[k][3] * ir),
                             fl_pack_bf16(acc[k][4] * ir, acc[k][5] * ir), fl_pack_bf16(acc[k][6] * ir, acc[k][7] * ir));
          }
        }
        FL_T(8);   // merging piece: read the other pieces, combine, store
      }
    }
    FL_T(4);   // epilogue
  }
#undef FL_Y_REQUEST_HEAD
  if (pend_ctr != nullptr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(pend_ctr, pend_add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
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

namespace {
unsigned* g_merge_err_host = nullptr;   // host side of the mapped error word
void merge_err_init(hipStream_t stream) {
  static bool tried = false;
  if (tried) return;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return;   // (not under capture: first eager call)
  tried = true;
  void *hp = nullptr, *dp = nullptr;
  if (hipHostMalloc(&hp, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) { (void)hipGetLastError(); return; }
  *(volatile unsigned*)hp = 0u;
  unsigned* dpu = (unsigned*)dp;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_merge_err), &dpu, sizeof(dpu)) != hipSuccess) { (void)hipGetLastError(); return; }
  g_merge_err_host = (unsigned*)hp;
  if (const char* e = getenv("FLUENT_MLA_MERGE_TIMEOUT_S")) {
    const double sec = atof(e);
    if (sec > 0 && sec <= 3600) {
      const unsigned long long ticks = (unsigned long long)(sec * 1e8);
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_merge_timeout_ticks), &ticks, sizeof(ticks));
    }
  }
}
}  // namespace

extern "C" int fl_mla_set_merge_timeout(double seconds) {
  FL_CHECK_ARG(seconds > 0 && seconds <= 3600, "fl_mla_set_merge_timeout: %g s (0 < s <= 3600)", seconds);
  const unsigned long long ticks = (unsigned long long)(seconds * 1e8);
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_merge_timeout_ticks), &ticks, sizeof(ticks)) != hipSuccess) {
    fl_set_error("fl_mla_set_merge_timeout: hipMemcpyToSymbol failed");
    return FL_ERR_LAUNCH;
  }
  return FL_OK;
}

int fl_mla_decode_fp8_y_impl(const FlMlaDecodeArgs* a, const Params& p_in, hipStream_t stream) {
  merge_err_init(stream);
  if (g_merge_err_host != nullptr) {
    const unsigned w = *(volatile unsigned*)g_merge_err_host;
    if (w != 0u) {
      *(volatile unsigned*)g_merge_err_host = 0u;
      fl_set_error("fl_mla_decode: the in-kernel split merge of an EARLIER launch gave up waiting for a request's other pieces (workgroup %u): that "
                   "launch wrote NaN into the request's rows and LSEs; its scheduler metadata must be rebuilt (fl_mla_get_metadata) before reuse", w - 1u);
      return FL_ERR_LAUNCH;
    }
  }
  Params p = p_in;
  const int nrt = p.rows > 32 ? 2 : 1;   // row tiles per workgroup: 64-row workgroups of 8 waves, or 32-row workgroups of 4 (TP8 shard: H = 16)
  p.row_groups = (p.rows + 32 * nrt - 1) / (32 * nrt);
  p.partial_bf16 = 1;   // split partials travel as bf16 rows
  static const int merge_env = [] {   // FLUENT_MLA_MERGE_KERNEL=1: always the merge kernel; =0: always in-kernel (tests); unset: the rule below
    const char* e = getenv("FLUENT_MLA_MERGE_KERNEL");
    return (e == nullptr || e[0] == '\0') ? -1 : (e[0] == '1' ? 1 : 0);
  }();
  // in-kernel merge when a request is cut into few pieces (at least half as many requests as parts: 2-3 pieces each); with
  // fewer requests EVERY request is cut into many pieces and the row-parallel merge kernel is the better tool.  Measured
  // (tools/bench_one_batch.py, 61-layer step, seq 4096, H = 128): bs = 64 (2 pieces) 4.38-4.41 ms in-kernel vs 4.46 with the merge
  // kernel; bs = 32 (4 pieces) 3.23 vs 2.99; bs = 16 3.00 vs 2.32; bs = 1 4.55 vs 1.79
  p.merge_in_kernel = (p.row_groups <= 6 && (merge_env == 0 || (merge_env < 0 && 2 * p.bs >= p.num_parts))) ? 1 : 0;
  const dim3 grid((unsigned)(p.num_parts * p.row_groups)), block(256 * nrt);
#define FL_Y_LAUNCH1(FMT_, QBF_, NRT_, NT_, QN_, QR_, QS_)                                                              \
  mla_decode_y_kernel<FMT_, QBF_, NRT_, NT_><<<grid, block, 0, stream>>>(                                               \
      p, a->block_table, a->cache_seqlens, a->tile_scheduler_metadata, const_cast<int32_t*>(a->tile_scheduler_metadata), \
      a->num_splits, (const uint8_t*)a->k_nope, (const uint16_t*)a->k_rope, a->k_scale, QN_, QR_, QS_)
  // (one row group per request = one reader per page: non-temporal page stream)
#define FL_Y_LAUNCH(FMT_, QBF_, NRT_, QN_, QR_, QS_)                                                                    \
  do {                                                                                                                  \
    if (NRT_ == 1 || p.row_groups == 1) FL_Y_LAUNCH1(FMT_, QBF_, NRT_, true, QN_, QR_, QS_);                            \
    else FL_Y_LAUNCH1(FMT_, QBF_, NRT_, false, QN_, QR_, QS_);                                                          \
  } while (0)
  const uint8_t* qn = (const uint8_t*)a->q_nope;
  const uint16_t* qr = (const uint16_t*)a->q_rope;
  if (a->kv_format == FL_KV_FP8_576) {
    if (nrt == 2) FL_Y_LAUNCH(1, false, 2, qn, qr, a->q_scale); else FL_Y_LAUNCH(1, false, 1, qn, qr, a->q_scale);
  } else if (p.q_bf16 != nullptr) {
    if (nrt == 2) FL_Y_LAUNCH(0, true, 2, nullptr, nullptr, nullptr); else FL_Y_LAUNCH(0, true, 1, nullptr, nullptr, nullptr);
  } else {
    if (nrt == 2) FL_Y_LAUNCH(0, false, 2, qn, qr, a->q_scale); else FL_Y_LAUNCH(0, false, 1, qn, qr, a->q_scale);
  }
#undef FL_Y_LAUNCH
#undef FL_Y_LAUNCH1
  FL_CHECK_LAUNCH("mla_decode_y_kernel");
  // split requests are merged inside the kernel by their first piece (six 16-bit arrival counters per part: up to 6 row
  // groups); beyond that, or with FLUENT_MLA_MERGE_KERNEL=1, the separate merge kernel runs
  return p.merge_in_kernel ? FL_OK : fl_mla_launch_combine(p, a->num_splits, stream);
}

#ifdef FL_MLA_TIMING
extern "C" int fl_mla_debug_set_buffer_y(int* dev_ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_y), &dev_ptr, sizeof(dev_ptr));
}
#endif
