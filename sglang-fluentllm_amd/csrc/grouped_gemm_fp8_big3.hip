// G1-G3 for MANY rows per group, round 6: ONE WAVE PER SIMD.  grouped_gemm_fp8_big2.hip (256 x 256 tile, 8 waves, two wave groups in
// anti-phase) sat at 0.30 of the fp8 sheet for three rounds: a k block cost 3,900 cycles against 2,048 of matrix pipe — M_x + M_y + ~750
// cycles of hand-off between the groups, with every LDS-DMA piece stalling the wave that was also the only MFMA issuer of its SIMD
// (profiles/r04_gemm_big2_bounding_ladder.txt).  This kernel changes the mapping:
//
//   * 4 waves x 512 registers, one per SIMD, nobody to hand over to: each wave issues ONE continuous MFMA stream (24 per k block) and every
//     other instruction of the k block — operand reads, LDS-DMA refill, the per-k-block rescale — sits in the 64-cycle shadow of an MFMA.
//     One workgroup barrier per half step (k = 64) instead of four per k block.
//   * tile 192 tokens x 256 weight rows; wave (wm, wn) = 96 tokens x 128 weight rows = 3 x 4 accumulator tiles of 32 x 32 = 192 ARCHITECTURAL
//     VGPRs (the in-accumulator 1x128 / 128x128 rescale is VALU work and the VALU cannot address AGPRs); the operand fragments of TWO half
//     steps (2 x 7 x 8 registers) live in AGPRs — `ds_read_b128 a[..]` targets and MFMA srcA/srcB straight from there.  7 fragment reads per
//     12 MFMAs (0.58 per MFMA; the 8-wave layout needed 0.75), one 128-row scale block per wave (one Ws scalar per k block).
//   * 192-token tiles also quantise a ~512-row expert better than 256-token tiles (3 x 192 = 576 against 2 or 3 x 256).
//   * tile order: m fastest inside an expert, then n — the three token tiles of one weight panel run back to back on ONE XCD (weights are read
//     once from HBM; the much smaller token panels are shared through the Infinity Cache).
//   * the scale s = As[m,kb] * Ws[e,nb,kb] = 2^e * f, f in [1,2): 2^e goes into the MX block scale of the token operand (E8M0, exact), the
//     accumulator is kept in units of the current mantissa (acc' <- acc' * f_prev / f once per k block: 16 multiplies per tile, the same VALU
//     work as a promotion `acc += part * s`), out = acc' * f_last.  Token scales come straight from global memory into registers, two k
//     blocks ahead (no LDS staging); the weight scales of a tile row sit one per lane in ONE register (`v_readlane` per k block).
//   * ring: 4 slots of half k blocks ([W 256 rows | A 192 rows] x 64 B = 28 KiB), refilled three half steps ahead with counted `vmcnt`;
//     the refill runs across tile boundaries (persistent walk, one workgroup per CU); the epilogue has its own 32 KiB of LDS.
// Same math, call sites and data formats as big2 (deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset / _masked, layers/moe/gemms/fp8/fire.py:18,
// layers/moe/executors/fp8_eps_executor.py:56,78).
#include <type_traits>

#include "grouped_gemm_shared.h"

using namespace fl_gemm;

namespace {

#if defined(FL_GEMM3_TIMING) || defined(FL_GEMM3_SLOTS)
__device__ unsigned long long* g_g3dbg = nullptr;
#endif
#ifdef FL_GEMM3_TIMING
#define G3T(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); gt[i] += t__ - gl; gl = t__; } while (0)
#else
#define G3T(i) do { } while (0)
#endif

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int BM3 = 192;                  // token rows per workgroup
constexpr int BN3 = 256;                  // weight rows per workgroup
constexpr int BKH3 = 64;                  // half k block (bytes per row per ring slot)
constexpr int kW3 = BN3 * BKH3;           // 16 KiB
constexpr int kA3 = BM3 * BKH3;           // 12 KiB
constexpr int kSlot3 = kW3 + kA3;         // 28 KiB
constexpr int kSlots3 = 4;
constexpr int kStg3 = 4 * 8192;           // epilogue staging: 32 tokens x 128 weight columns bf16 per wave
constexpr int kSmem3 = kSlots3 * kSlot3 + kStg3;   // 147,456 B

// ---- the filler schedule of a k block (24 MFMA slots: 0..11 = first half step, 12..23 = second; tile t = 4 j + i sits in slots t and
//      12 + t).  Tunables (FL_G3_* macros let tools/build_gemm3_var.sh build variants):
//      kRd[f]   slot (within a half step) behind whose MFMA fragment f of the NEXT half step is read (f 0..3: weight blocks, 4..6: token blocks)
//      kDma[k]  slot behind whose MFMA LDS-DMA piece k of stage h + 3 is issued (k 0..3: W pieces, 4..6: A pieces)
//      kCap[.]  rescale multiplies behind the MFMA of timeline position tau: tau 0..9 = slots 2..11 of the second half step, tau 10..19 =
//               slots 0..9 of the NEXT k block's first half step.  Tile t may be rescaled in tau t .. t + 8 only (two slots behind its last
//               MFMA: an XDL write needs 19 issue cycles before a VALU read; one slot ahead of its next MFMA).
#ifndef FL_G3_RD
#define FL_G3_RD {2, 3, 4, 5, 6, 7, 8}
#endif
#ifndef FL_G3_DMA_E
#define FL_G3_DMA_E {2, 3, 5, 7, 9, 10, 11}
#endif
#ifndef FL_G3_DMA_O
#define FL_G3_DMA_O {2, 3, 4, 5, 7, 9, 11}
#endif
#ifndef FL_G3_CAP
#define FL_G3_CAP {12, 8, 12, 8, 12, 8, 12, 8, 12, 8,  12, 8, 12, 8, 12, 8, 12, 8, 8, 4}
#endif
constexpr int kRd[7] = FL_G3_RD;
constexpr int kDmaE[7] = FL_G3_DMA_E;     // even half step (first of a k block)
constexpr int kDmaO[7] = FL_G3_DMA_O;     // odd half step
constexpr int kCap[20] = FL_G3_CAP;
constexpr bool slots_behind_barrier() {   // the half step's wait + barrier sit behind its SECOND MFMA: reads and refill pieces come after them
  for (int i = 0; i < 7; ++i)
    if (kRd[i] < 2 || kDmaE[i] < 2 || kDmaO[i] < 2) return false;
  return true;
}
static_assert(slots_behind_barrier(), "fragment reads and LDS-DMA pieces must sit in slots 2..11");
constexpr int cap_cum(const int tau) {    // multiplies scheduled before timeline position tau
  int s = 0;
  for (int i = 0; i < tau && i < 20; ++i) s += kCap[i];
  return s;
}
static_assert(cap_cum(20) == 192, "the rescale schedule must cover 12 tiles x 16 registers");
constexpr bool cap_ok() {
  for (int t = 0; t < 12; ++t) {
    if (cap_cum(t) > 16 * t) return false;             // tile t not before tau = t
    if (cap_cum(t + 9) < 16 * (t + 1)) return false;   // ... and done by tau = t + 8
  }
  return true;
}
static_assert(cap_ok(), "rescale schedule violates a tile's window");

// ---- operand fragments live in FIXED accumulation registers a[144:255] (set s, fragment f at a[144 + 56 s + 8 f ..+7]; f 0..3 weight blocks,
//      4..6 token blocks).  hipcc's own allocation of a loop-carried v8i in AGPRs put 8 v_accvgpr_mov in front of every MFMA (one per
//      register: the two ds_read_b128 halves of a fragment are separate values to it); with literal registers nothing is copied.  Every asm
//      that writes them names them as clobbers, so the compiler never keeps a value of its own there (it allocates AGPRs from a0 upwards;
//      tools/check_gemm3_isa.py asserts that no compiler-generated instruction touches a144 or above).
#ifdef FL_G3_MFMA_NOP
#define G3_MFMA_PAD "s_nop 1\n\t"
#else
#define G3_MFMA_PAD
#endif
__device__ __forceinline__ void mfma3(v16f& acc, const int set, const int i, const int j, const int sb, const bool zero) {
  switch ((zero ? 24 : 0) + set * 12 + i * 3 + j) {
    case 0: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[144:151], a[176:183], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 1: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[144:151], a[184:191], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 2: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[144:151], a[192:199], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 3: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[152:159], a[176:183], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 4: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[152:159], a[184:191], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 5: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[152:159], a[192:199], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 6: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[160:167], a[176:183], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 7: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[160:167], a[184:191], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 8: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[160:167], a[192:199], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 9: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[168:175], a[176:183], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 10: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[168:175], a[184:191], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 11: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[168:175], a[192:199], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 12: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[200:207], a[232:239], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 13: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[200:207], a[240:247], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 14: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[200:207], a[248:255], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 15: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[208:215], a[232:239], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 16: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[208:215], a[240:247], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 17: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[208:215], a[248:255], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 18: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[216:223], a[232:239], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 19: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[216:223], a[240:247], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 20: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[216:223], a[248:255], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 21: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[224:231], a[232:239], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 22: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[224:231], a[240:247], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 23: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[224:231], a[248:255], %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 24: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[144:151], a[176:183], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 25: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[144:151], a[184:191], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 26: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[144:151], a[192:199], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 27: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[152:159], a[176:183], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 28: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[152:159], a[184:191], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 29: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[152:159], a[192:199], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 30: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[160:167], a[176:183], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 31: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[160:167], a[184:191], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 32: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[160:167], a[192:199], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 33: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[168:175], a[176:183], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 34: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[168:175], a[184:191], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 35: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[168:175], a[192:199], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 36: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[200:207], a[232:239], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 37: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[200:207], a[240:247], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 38: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[200:207], a[248:255], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 39: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[208:215], a[232:239], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 40: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[208:215], a[240:247], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 41: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[208:215], a[248:255], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 42: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[216:223], a[232:239], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 43: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[216:223], a[240:247], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 44: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[216:223], a[248:255], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 45: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[224:231], a[232:239], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 46: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[224:231], a[240:247], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    case 47: asm volatile(G3_MFMA_PAD "v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[224:231], a[248:255], 0, %1, %2 op_sel_hi:[0,0,0]" : "=&v"(acc) : "v"(kUnit), "v"(sb)); break;
    default: break;
  }
}
// fragment f of register set `set`: the lane's two 16-byte chunks (addresses a0, a1 + 2 KiB per 32-row block)
__device__ __forceinline__ void rd_frag3(const int set, const int f, const int a0, const int a1) {
  switch (set * 7 + f) {
    case 0: asm volatile("ds_read_b128 a[144:147], %0 offset:0\n\tds_read_b128 a[148:151], %1 offset:0" :: "v"(a0), "v"(a1) : "memory", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151"); break;
    case 1: asm volatile("ds_read_b128 a[152:155], %0 offset:2048\n\tds_read_b128 a[156:159], %1 offset:2048" :: "v"(a0), "v"(a1) : "memory", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159"); break;
    case 2: asm volatile("ds_read_b128 a[160:163], %0 offset:4096\n\tds_read_b128 a[164:167], %1 offset:4096" :: "v"(a0), "v"(a1) : "memory", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167"); break;
    case 3: asm volatile("ds_read_b128 a[168:171], %0 offset:6144\n\tds_read_b128 a[172:175], %1 offset:6144" :: "v"(a0), "v"(a1) : "memory", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175"); break;
    case 4: asm volatile("ds_read_b128 a[176:179], %0 offset:0\n\tds_read_b128 a[180:183], %1 offset:0" :: "v"(a0), "v"(a1) : "memory", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183"); break;
    case 5: asm volatile("ds_read_b128 a[184:187], %0 offset:4096\n\tds_read_b128 a[188:191], %1 offset:4096" :: "v"(a0), "v"(a1) : "memory", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191"); break;
    case 6: asm volatile("ds_read_b128 a[192:195], %0 offset:8192\n\tds_read_b128 a[196:199], %1 offset:8192" :: "v"(a0), "v"(a1) : "memory", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199"); break;
    case 7: asm volatile("ds_read_b128 a[200:203], %0 offset:0\n\tds_read_b128 a[204:207], %1 offset:0" :: "v"(a0), "v"(a1) : "memory", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207"); break;
    case 8: asm volatile("ds_read_b128 a[208:211], %0 offset:2048\n\tds_read_b128 a[212:215], %1 offset:2048" :: "v"(a0), "v"(a1) : "memory", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215"); break;
    case 9: asm volatile("ds_read_b128 a[216:219], %0 offset:4096\n\tds_read_b128 a[220:223], %1 offset:4096" :: "v"(a0), "v"(a1) : "memory", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223"); break;
    case 10: asm volatile("ds_read_b128 a[224:227], %0 offset:6144\n\tds_read_b128 a[228:231], %1 offset:6144" :: "v"(a0), "v"(a1) : "memory", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231"); break;
    case 11: asm volatile("ds_read_b128 a[232:235], %0 offset:0\n\tds_read_b128 a[236:239], %1 offset:0" :: "v"(a0), "v"(a1) : "memory", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239"); break;
    case 12: asm volatile("ds_read_b128 a[240:243], %0 offset:4096\n\tds_read_b128 a[244:247], %1 offset:4096" :: "v"(a0), "v"(a1) : "memory", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247"); break;
    case 13: asm volatile("ds_read_b128 a[248:251], %0 offset:8192\n\tds_read_b128 a[252:255], %1 offset:8192" :: "v"(a0), "v"(a1) : "memory", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"); break;
    default: break;
  }
}
// Token scales (and the next tile's weight-scale row) straight from global memory into FIXED accumulation registers a140 (weight scales of the
// next tile) and a141..a143 (token scales of block j), read back with v_accvgpr_read behind a counted `s_waitcnt vmcnt` (loads return in
// order).  Not into a C++ variable: an asm output is "defined" for hipcc the moment the statement is issued, and it is free to copy such a
// value (a phi copy at a branch join, a spill) BEFORE the data has arrived — measured: wrong scales for token blocks 0 and 1 as soon as the k
// loop existed in three instantiations.  (`s_nop 4`: the base may come out of a v_readlane / v_readfirstlane — a spilled SGPR — right in front
// of the statement, and a VALU write of an SGPR needs 5 wait states before a VMEM instruction reads it; hipcc pads nothing for an asm.)
__device__ __forceinline__ void ldg_agpr(const int idx, const unsigned voff, const void* sbase) {
  switch (idx) {
    case 0: asm volatile("s_nop 4\n\tglobal_load_dword a140, %0, %1" :: "v"(voff), "s"(sbase) : "memory", "a140"); break;
    case 1: asm volatile("s_nop 4\n\tglobal_load_dword a141, %0, %1" :: "v"(voff), "s"(sbase) : "memory", "a141"); break;
    case 2: asm volatile("s_nop 4\n\tglobal_load_dword a142, %0, %1" :: "v"(voff), "s"(sbase) : "memory", "a142"); break;
    default: asm volatile("s_nop 4\n\tglobal_load_dword a143, %0, %1" :: "v"(voff), "s"(sbase) : "memory", "a143"); break;
  }
}
__device__ __forceinline__ float rd_agpr(const int idx) {
  float v;
  switch (idx) {
    case 0: asm volatile("v_accvgpr_read_b32 %0, a140" : "=v"(v)); break;
    case 1: asm volatile("v_accvgpr_read_b32 %0, a141" : "=v"(v)); break;
    case 2: asm volatile("v_accvgpr_read_b32 %0, a142" : "=v"(v)); break;
    default: asm volatile("v_accvgpr_read_b32 %0, a143" : "=v"(v)); break;
  }
  return v;
}
// (a plain register destination: only where the wait follows in the same statement sequence, before anything else can happen to the value)
__device__ __forceinline__ void ldg_f32_async(float& dst, const unsigned voff, const void* sbase) {
  asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

template <int OFF>
__device__ __forceinline__ void fl_dma16_lds(const void* sbase, const unsigned voff, const int lds_slot) {
  asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_slot), "v"(voff), "s"(sbase), "n"(OFF)
               : "memory", "m0", "scc");
}

// outside the k loop (prologue): same, padded against a VALU-written SGPR operand (see ldg_f32_async); the k loop's pieces take their
// operands from SALU results only — tools/check_gemm3_isa.py checks the built code for that
__device__ __forceinline__ void fl_dma16_lds_padded(const void* sbase, const unsigned voff, const int lds_addr) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory", "m0");
}

#define G3_BARRIER()                          \
  do {                                        \
    __builtin_amdgcn_sched_barrier(0);        \
    __builtin_amdgcn_s_barrier();             \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);        \
  } while (0)

struct Tile3 {
  const uint8_t* w_base;    // row n0 + 64 wave of the expert's weight panel
  const uint8_t* a_base;    // row0 of the token panel
  const float* as_base;     // gAs (+ the group's base in masked mode)
  const float* ws_row;      // Ws[e, nb of this wave, 0]
  unsigned va[3];           // A pieces 3 wave + (0..2): clamped row * K + swizzled chunk
  unsigned vs[3];           // token scales of token block j: clamped row * as_stride_m * 4
  int n0;
  int nj;               // token blocks of this wave that hold rows of the group (0..3): the others' MFMAs are skipped
  long long row0, row_end;
};

__global__ __launch_bounds__(256, 1) void grouped_gemm_fp8_big3_kernel(const GemmParams p, const uint8_t* __restrict__ gA,
                                                                      const float* __restrict__ gAs,
                                                                      const uint8_t* __restrict__ gW,
                                                                      const float* __restrict__ gWs,
                                                                      const int32_t* __restrict__ gmeta) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[kSmem3];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wn = wave & 1, wm = wave >> 1;
  const int n_tiles = p.n_tiles;
  const int KB = p.K / BK;
  const int NH = 2 * KB;

  auto uniform = [](const void* ptr) {   // (keeps a 64-bit base in an SGPR pair: the asm operand is "s")
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const uint8_t*>(((unsigned long long)hi << 32) | lo);
  };

  // W pieces: wave w issues pieces 4 w .. 4 w + 3 of a stage (16 rows x 64 B each); the per-lane part of the source address is the same for
  // every piece and every tile (the tile's row goes into the scalar base): row (lane >> 2) of the piece, chunk (lane & 3) ^ ((lane >> 4) & 3)
  // — the swizzle is on the SOURCE, the LDS image is lane-linear (conflict-free ds_read_b128: big2's layout).  N % 256 == 0 (launcher).
  const unsigned swz = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
  unsigned vw4[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) vw4[k] = __umul24((unsigned)(16 * k + (lane >> 2)), (unsigned)p.K) + swz;

  // ---- tile list, order (expert, n, m): slot -> XCD-aware logical index -> (expert, n tile, m tile of the expert) ----
  auto setup_tile = [&](const int slot, Tile3& t) -> bool {
    int lid = slot;
    const int round = lid >> 8;
    if ((round + 1) * 256 <= p.total_blocks) lid = (round << 8) + ((lid & 7) << 5) + ((lid & 255) >> 3);
    // expert whose m-tile range [base, base + cnt) holds lid / n_tiles (ranges scaled by n_tiles are contiguous in lid)
    int e = 0, base_mt = 0, cnt = 0;
    long long lo_row = 0, hi_row = 0;
    const int mt_virtual = lid / n_tiles;
    if (p.mode == kOffset) {
      bool found = false;
      int base = 0;
      for (int g0 = 0; g0 < p.E && !found; g0 += 64) {
        const int g = g0 + lane;
        const int lo = g < p.E ? gmeta[g] : 0;
        const int hi = g < p.E ? gmeta[g + 1] : 0;
        const int tiles = g < p.E ? (hi - lo + BM3 - 1) / BM3 : 0;
        int incl = tiles;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int v = __shfl_up(incl, o);
          if (lane >= o) incl += v;
        }
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (mt_virtual < base + total) {
          const int excl = incl - tiles;
          const unsigned long long hit = __ballot(tiles > 0 && mt_virtual >= base + excl && mt_virtual < base + incl);
          const int src = __builtin_ctzll(hit);
          e = g0 + src;
          base_mt = base + __builtin_amdgcn_readlane(excl, src);
          cnt = __builtin_amdgcn_readlane(tiles, src);
          lo_row = __builtin_amdgcn_readlane(lo, src);
          hi_row = __builtin_amdgcn_readlane(hi, src);
          found = true;
        }
        base += total;
      }
      if (!found) return false;
    } else if (p.mode == kMasked) {
      const int tpg = (int)((p.rows_per_group + BM3 - 1) / BM3);
      e = mt_virtual / tpg;
      if (e >= p.E) return false;
      const int mm = gmeta[e];
      cnt = (mm + BM3 - 1) / BM3;
      // masked groups keep tpg tile slots each: the first cnt * n_tiles of the group's tpg * n_tiles indices are real tiles
      const int local_m = lid - e * tpg * n_tiles;
      if (local_m >= cnt * n_tiles) return false;
      base_mt = e * tpg;
      lo_row = (long long)e * p.rows_per_group;
      hi_row = lo_row + mm;
    } else {   // dense
      cnt = (p.M + BM3 - 1) / BM3;
      if (mt_virtual >= cnt) return false;
      base_mt = 0;
      lo_row = 0;
      hi_row = p.M;
    }
    const int local = lid - base_mt * n_tiles;
    const int mt = local % cnt, nt = local / cnt;
    t.n0 = nt * BN3;
    t.row0 = lo_row + (long long)mt * BM3;
    t.row_end = hi_row;
    t.w_base = gW + ((long long)e * p.N + t.n0 + 64 * wave) * p.K;
    t.a_base = gA + t.row0 * p.K;
    const unsigned m_last = (unsigned)(t.row_end - t.row0 - 1);   // rows beyond the group: clamped, never stored
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const unsigned r = (unsigned)(48 * wave + 16 * k + (lane >> 2));
      t.va[k] = __umul24(r < m_last ? r : m_last, (unsigned)p.K) + swz;
    }
    t.as_base = p.mode == kMasked ? gAs + (long long)e * p.as_stride_g - (long long)e * p.rows_per_group * p.as_stride_m : gAs;
    // token block j of wave half wm = tile rows [32 (2 j + wm), + 32): the 32-row blocks alternate between the two wave halves, so a tile
    // with nblk blocks of real rows costs ceil(nblk / 2) MFMA groups per half step instead of 3 (a 512-row expert = 2 2/3 tiles)
    {
      const int rows = (int)(t.row_end - t.row0 < BM3 ? t.row_end - t.row0 : BM3);
      const int nblk = (rows + 31) >> 5;
      t.nj = (nblk - wm + 1) >> 1;
#ifdef FL_G3_NJ3
      t.nj = 3;
#endif
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      long long m = t.row0 + 32 * (2 * j + wm) + li;
      m = m < t.row_end ? m : t.row_end - 1;
      t.vs[j] = (unsigned)(m * p.as_stride_m * 4);
    }
    const int nbs = (p.N + BN - 1) / BN;
    t.ws_row = gWs + ((long long)e * nbs + (t.n0 + 128 * wn) / BN) * KB;
    return true;
  };

  int slot = blockIdx.x;
  Tile3 cur;
  {
    bool found = false;
    for (; slot < p.total_blocks; slot += gridDim.x)
      if (setup_tile(slot, cur)) { found = true; break; }
    if (!found) return;
  }
  // the tile row's weight scales, k block `lane` (clamped), by the same asynchronous load as the token scales: a compiler-issued load would be
  // waited for with `vmcnt(0)` at its first use INSIDE the k loop (it cannot see the LDS-DMA queue), draining the refill every k block
  const unsigned ws_voff = (unsigned)((lane < KB ? lane : KB - 1) * 4);
  float ws_cur = 0.f;
  ldg_f32_async(ws_cur, ws_voff, uniform(cur.ws_row));   // (waits inside: nothing else is in flight yet)

  // operand read offsets inside a ring slot: row li of a 32-row block (64 B per row), the lane half's 32 k bytes = chunks 2 lh, 2 lh + 1
  const int rb0 = li * BKH3 + ((((2 * lh) ^ ((li >> 2) & 3))) << 4);
  const int rb1 = li * BKH3 + ((((2 * lh + 1) ^ ((li >> 2) & 3))) << 4);
  const int lds0 = (int)(uintptr_t)smem;   // LDS byte address of the ring
  const int rw0 = (128 * wn) * BKH3 + rb0, rw1 = (128 * wn) * BKH3 + rb1;
  const int ra0 = kW3 + (32 * wm) * BKH3 + rb0, ra1 = kW3 + (32 * wm) * BKH3 + rb1;   // token block j: + j * 64 rows

  v16f acc[4][3];     // [weight-row block i][token block j]: D^T[32 weight rows, 32 tokens], one token per lane
  float cmant[3], ratio[3], cm_fin[3];
  int e8[3], e8n[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) { cmant[j] = 1.f; ratio[j] = 1.f; cm_fin[j] = 1.f; e8[j] = kUnit; e8n[j] = kUnit; }

  const int lds_a_off = kW3 + (3 * wave) * 1024 - (4 * wave) * 1024;   // this wave's first A piece relative to its first W piece
  const int lds_w_end = lds0 + kSlots3 * kSlot3 + (4 * wave) * 1024;
  int lds_iss_w = __builtin_amdgcn_readfirstlane(lds0 + (4 * wave) * 1024);   // this wave's first W piece in the slot of the next stage to issue
  int lds_iss_a = lds_iss_w + lds_a_off;
  int lds_rd = __builtin_amdgcn_readfirstlane(lds0);                          // slot of the next stage to read
  int iss = 0;
  const uint8_t* wrun = cur.w_base;
  const uint8_t* arun = cur.a_base;
  unsigned va_run[3] = {cur.va[0], cur.va[1], cur.va[2]};
  const uint8_t* as_run = nullptr;
  const long long as_step = p.as_stride_k * 4;
  bool carried = false;   // the stages 0..2 of `cur`, its first fragments, its block-0 scales and As[1] are already there
#pragma unroll 1
  for (;;) {
    Tile3 nxt = cur;
    bool has_next = false;
    int nslot = slot + gridDim.x;
    for (; nslot < p.total_blocks; nslot += gridDim.x)
      if (setup_tile(nslot, nxt)) { has_next = true; break; }
    const bool xt = has_next;
    const int nj_cur = __builtin_amdgcn_readfirstlane(cur.nj);
    ldg_agpr(0, ws_voff, uniform(nxt.ws_row));   // -> a140 (landed long before its first use: every half step has a counted wait)

    // ---- the refill's running state (wave-uniform, SALU only): the stage to issue next is `iss` (numbered in the current tile: NH .. NH + 2
    //      are the next tile's stages 0 .. 2; without a next tile the last stage is fetched again into idle slots — one loop body, constant
    //      vmcnt counts), its sources wrun / arun / va_run, its ring slot lds_iss.  Piece k of this wave's share: 0..3 = W, 4..6 = A. ----
    auto issue_piece = [&](const int k) {
      switch (k) {
        case 0: fl_dma16_lds<0>(wrun, vw4[0], lds_iss_w); break;
        case 1: fl_dma16_lds<1024>(wrun, vw4[1], lds_iss_w); break;
        case 2: fl_dma16_lds<2048>(wrun, vw4[2], lds_iss_w); break;
        case 3: fl_dma16_lds<3072>(wrun, vw4[3], lds_iss_w); break;
        case 4: fl_dma16_lds<0>(arun, va_run[0], lds_iss_a); break;
        case 5: fl_dma16_lds<1024>(arun, va_run[1], lds_iss_a); break;
        default: fl_dma16_lds<2048>(arun, va_run[2], lds_iss_a); break;
      }
    };
    auto issue_piece_padded = [&](const int k) {
      const int la = k < 4 ? lds_iss_w + k * 1024 : lds_iss_a + (k - 4) * 1024;
      fl_dma16_lds_padded(k < 4 ? wrun : arun, k < 4 ? vw4[k] : va_run[k - 4], la);
    };
    auto advance_lds = [&]() {
      const int nl = lds_iss_w + kSlot3;
      lds_iss_w = nl >= lds_w_end ? nl - kSlots3 * kSlot3 : nl;
      lds_iss_a = lds_iss_w + lds_a_off;
    };
    auto advance_issue_simple = [&]() {   // inside the tile: the next stage is 64 bytes further along k
      ++iss;
      advance_lds();
      wrun += BKH3;
      arun += BKH3;
    };
    auto advance_issue = [&]() {          // anywhere: may cross into the next tile (or stay on the last stage without one)
      ++iss;
      advance_lds();
      const bool sw = xt && iss == NH;             // the first stage of the next tile
      const bool step = iss < NH || (xt && iss > NH);
      wrun = sw ? nxt.w_base : (step ? wrun + BKH3 : wrun);
      arun = sw ? nxt.a_base : (step ? arun + BKH3 : arun);
#pragma unroll
      for (int k = 0; k < 3; ++k) va_run[k] = sw ? nxt.va[k] : va_run[k];
    };
    auto advance_read = [&]() {
      const int nl = lds_rd + kSlot3;
      lds_rd = nl >= lds0 + kSlots3 * kSlot3 ? nl - kSlots3 * kSlot3 : nl;
    };
    // token scales of k block kbx (kbx >= KB: the next tile's k block kbx - KB) -> as_raw
    auto issue_as = [&](const int kbx) {
      const bool nx = xt && kbx >= KB;
      const int kc = nx ? kbx - KB : (kbx < KB ? kbx : KB - 1);
      const float* b = (nx ? nxt.as_base : cur.as_base) + (long long)kc * p.as_stride_k;
#pragma unroll
      for (int j = 0; j < 3; ++j) ldg_agpr(1 + j, nx ? nxt.vs[j] : cur.vs[j], uniform(b));
    };
    auto issue_as_simple = [&]() {   // k block kb + 2 of the current tile through the running pointer
#pragma unroll
      for (int j = 0; j < 3; ++j) ldg_agpr(1 + j, cur.vs[j], as_run);
      as_run += as_step;
    };
    // scales of k block kbx from as_raw: E8M0 part -> e8n, mantissa -> cmant, ratio = old mantissa / new.  Crossing into the next tile
    // (kbx == KB): no rescale (ratio 1), the finished tile's mantissa is kept for its epilogue.
    auto scale_math = [&](const int kbx) {
      const bool last = kbx >= KB;
      const float ws_n = rd_agpr(0);
      const float ws = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(last ? ws_n : ws_cur), last ? 0 : kbx));
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float s = rd_agpr(1 + j) * ws;
        const unsigned bits = __float_as_uint(s);
        const unsigned eb = (bits >> 23) & 0xffu;
        // zero / denormal scale: 2^-127 x 1.0 (the term is below anything fp32 can add to the sum anyway)
        const float f = eb != 0u ? __uint_as_float((bits & 0x807fffffu) | 0x3f800000u) : 1.f;
        const float r = cmant[j] * __builtin_amdgcn_rcpf(f);
        cm_fin[j] = cmant[j];
        ratio[j] = last ? 1.f : r;
        cmant[j] = f;
        e8n[j] = (int)eb;
      }
    };
    auto scale_math_simple = [&](const int kbx) {   // kbx < KB
      const float ws = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ws_cur), kbx));
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float s = rd_agpr(1 + j) * ws;
        const unsigned bits = __float_as_uint(s);
        const unsigned eb = (bits >> 23) & 0xffu;
        const float f = eb != 0u ? __uint_as_float((bits & 0x807fffffu) | 0x3f800000u) : 1.f;
        ratio[j] = cmant[j] * __builtin_amdgcn_rcpf(f);
        cmant[j] = f;
        e8n[j] = (int)eb;
      }
    };
    auto read_frag = [&](const int set, const int f) {   // fragment f of the half step in slot lds_rd into register set `set`
      if (f < 4) rd_frag3(set, f, lds_rd + rw0, lds_rd + rw1);
      else rd_frag3(set, f, lds_rd + ra0, lds_rd + ra1);
    };

    // ---- prologue of a tile that was not prefetched by its predecessor ----
    if (!carried) {
      iss = 0;
      wrun = cur.w_base;
      arun = cur.a_base;
#pragma unroll
      for (int k = 0; k < 3; ++k) va_run[k] = cur.va[k];
      issue_as(0);
      for (int k = 0; k < 7; ++k) issue_piece_padded(k);
      advance_issue();
      asm volatile("s_waitcnt vmcnt(7)" ::: "memory");   // As[0]
      __builtin_amdgcn_sched_barrier(0);
      scale_math(0);
      __builtin_amdgcn_sched_barrier(0);
      issue_as(1);
      for (int k = 0; k < 7; ++k) issue_piece_padded(k);
      advance_issue();
      for (int k = 0; k < 7; ++k) issue_piece_padded(k);
      advance_issue();
      asm volatile("s_waitcnt vmcnt(17)" ::: "memory");  // stage 0 (As[1] + stages 1, 2 = 3 + 14 stay in flight)
      G3_BARRIER();
#pragma unroll
      for (int f = 0; f < 7; ++f) read_frag(0, f);
      advance_read();   // (As[1] and stage 1: the wait + barrier of the first half step)
    }

#ifdef FL_GEMM3_SLOTS
    unsigned long long ts[26];
#pragma unroll
    for (int i = 0; i < 26; ++i) ts[i] = 0;
#endif
#ifdef FL_GEMM3_TIMING
    unsigned long long gt[4] = {0, 0, 0, 0};
    unsigned long long gl = __builtin_readcyclecounter();
    const unsigned long long g0 = gl;
#endif
    // ---- one half step: 12 MFMAs, everything else behind them ----
    auto half_step = [&](auto first_tag, auto odd_tag, auto general_tag, auto stamp_tag, auto nj_tag, const int kb) __attribute__((always_inline)) {
      constexpr int NJT = decltype(nj_tag)::value;   // token blocks of this wave with rows of the group: MFMAs and rescales of the others are left out
      constexpr bool STAMP = decltype(stamp_tag)::value;   // (FL_GEMM3_SLOTS builds: cycle stamps in front of every MFMA of ONE k block per tile)
      constexpr bool FIRST = decltype(first_tag)::value;
      constexpr bool GEN = decltype(general_tag)::value;   // may issue / scale across the tile boundary
      constexpr bool ODD = decltype(odd_tag)::value;
      constexpr int cs = ODD ? 1 : 0;        // fragment set of this half step
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the fragments of this half step (read behind the previous step's MFMAs)
      __builtin_amdgcn_sched_barrier(0);
      G3T(3);
#pragma unroll
      for (int s = 0; s < 12; ++s) {
        const int j = s >> 2, i = s & 3;
#ifdef FL_GEMM3_SLOTS
        if (STAMP) ts[(ODD ? 13 : 0) + s] = __builtin_readcyclecounter();
#endif
        if (j < NJT) mfma3(acc[i][j], cs, i, j, e8[j], FIRST && !ODD);
        // fragments of the next half step
#pragma unroll
        for (int f = 0; f < 7; ++f)
#ifndef FL_G3_NOREAD
          if (kRd[f] == s) read_frag(cs ^ 1, f);
#else
          if (kRd[f] == s) asm volatile("" ::: "a255");   // (keeps the kernel's AGPR allocation at 256)
#endif
        // token scales two k blocks ahead (odd half step, in front of this step's pieces: the wait at the end of the step covers them)
        if (ODD && s == 0) { if (GEN) issue_as(kb + 2); else issue_as_simple(); }
        // LDS-DMA refill, stage h + 3
#pragma unroll
        for (int k = 0; k < 7; ++k)
#ifndef FL_G3_NODMA   // (bounding builds, garbage results: tools/build_gemm3_var.sh)
          if ((ODD ? kDmaO[k] : kDmaE[k]) == s) issue_piece(k);
#endif
        // scales of the next k block (its token scales landed before the barrier that opened this k block)
        if (!ODD && s == 10) { if (GEN) scale_math(kb + 1); else scale_math_simple(kb + 1); }
        // rescale: timeline position tau
        const int tau = ODD ? s - 2 : s + 10;
        if (tau >= 0 && tau < 20 && !(FIRST && !ODD)) {
#pragma unroll
          for (int q = cap_cum(tau); q < cap_cum(tau + 1); ++q) {
            const int t = q >> 4, r = q & 15, ti = t & 3, tj = t >> 2;
#ifndef FL_G3_NORESCALE
            if (tj < NJT) acc[ti][tj][r] *= ratio[tj];
#endif
          }
        }
        if (s == 11) { if (GEN) advance_issue(); else advance_issue_simple(); advance_read(); }
        __builtin_amdgcn_sched_barrier(0);
        if (s == 1) {
          // The half step's synchronisation, BEHIND its first two MFMAs (their operands were read during the previous half step: the pipe
          // works through them while the waves meet): this wave's pieces of stage h + 1 have landed (stage h + 2 stays in flight), then
          // everyone's — the reads of stage h + 1 and the refill of the slot of stage h - 1 follow.
          G3T(0);
#ifndef FL_G3_NOBAR
          asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
          G3T(1);
          G3_BARRIER();
#endif
          G3T(2);
        }
      }
      G3T(0);
#ifdef FL_GEMM3_SLOTS
      if (STAMP) ts[(ODD ? 13 : 0) + 12] = __builtin_readcyclecounter();
#endif
    };
    auto kblock = [&](auto first_tag, auto general_tag, auto nj_tag, const int kb) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 3; ++j) e8[j] = e8n[j];
      half_step(first_tag, std::false_type{}, general_tag, std::false_type{}, nj_tag, kb);
      half_step(first_tag, std::true_type{}, general_tag, std::false_type{}, nj_tag, kb);
    };
#ifdef FL_GEMM3_SLOTS
    auto kblock_stamped = [&](auto nj_tag, const int kb) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 3; ++j) e8[j] = e8n[j];
      half_step(std::false_type{}, std::false_type{}, std::false_type{}, std::true_type{}, nj_tag, kb);
      half_step(std::false_type{}, std::true_type{}, std::false_type{}, std::true_type{}, nj_tag, kb);
    };
#endif
    // The k loop of a tile, one instantiation per number of live token blocks.  k block 0 (zero-initialising MFMAs) and the last two k blocks
    // (their refill / token scales / scales reach into the next tile) run the general bookkeeping; the blocks in between only step 64 bytes along k.
    auto run_tile = [&](auto nj_tag) __attribute__((always_inline)) {
      kblock(std::true_type{}, std::true_type{}, nj_tag, 0);
      const int kb_tail = KB - 2 > 1 ? KB - 2 : 1;
      as_run = uniform(cur.as_base + 3ll * p.as_stride_k);   // k block kb + 2 of kb = 1
#ifdef FL_GEMM3_SLOTS
      {   // (needs KB >= 8) k blocks 1..3 plain, k block 4 stamped, then on
        int kb = 1;
#pragma unroll 1
        for (; kb < 4; ++kb) kblock(std::false_type{}, std::false_type{}, nj_tag, kb);
        kblock_stamped(nj_tag, kb);
        ++kb;
#pragma unroll 1
        for (; kb < kb_tail; ++kb) kblock(std::false_type{}, std::false_type{}, nj_tag, kb);
      }
#else
#pragma unroll 1
      for (int kb = 1; kb < kb_tail; ++kb) kblock(std::false_type{}, std::false_type{}, nj_tag, kb);
#endif
#pragma unroll 1
      for (int kb = kb_tail; kb < KB; ++kb) kblock(std::false_type{}, std::true_type{}, nj_tag, kb);
    };
    // (a tile is the last of its expert when it has fewer than 6 blocks of 32 rows; the blocks alternate between the wave halves, so such
    //  a tile costs ceil(blocks / 2) MFMA groups per half step: a 512-row expert is 2 2/3 tiles, not 3)
    if (nj_cur >= 3) run_tile(std::integral_constant<int, 3>{});
    else if (nj_cur == 2) run_tile(std::integral_constant<int, 2>{});
    else run_tile(std::integral_constant<int, 1>{});

#ifdef FL_GEMM3_TIMING
    const unsigned long long t_loop_end = __builtin_readcyclecounter();
#endif
    // ---- epilogue: D^T[n, m] -> out[m, n] bf16; lane (token li of block j, half lh) holds weight rows 8 g + 4 lh + (0..3) of block i ----
    // (the last MFMAs are still in the pipe: a 16-pass XDL write needs 19 issue cycles before a VALU read)
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    {
      uint8_t* stg = smem + kSlots3 * kSlot3 + wave * 8192;   // [32 tokens][256 B]: 16-B chunk c of row r at position c ^ (r & 15)
      const int rr = lane >> 4, rc = lane & 15;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
#ifndef FL_G3_NOBREAK
        if (j >= nj_cur) break;   // (wave-uniform: no rows of the group in this block)
#endif
        const float cm = cm_fin[j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<uint2*>(stg + li * 256 + (((4 * i + g) ^ (li & 15)) << 4) + 8 * lh) =
                make_uint2(fl_pack_bf16(acc[i][j][4 * g + 0] * cm, acc[i][j][4 * g + 1] * cm),
                           fl_pack_bf16(acc[i][j][4 * g + 2] * cm, acc[i][j][4 * g + 3] * cm));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (wave-private buffer: the wave's own writes, in order)
        // All eight reads of the block go out together and the stores take a 32-bit lane offset against a scalar row base (round 6: hipcc's form of
        // "read one row, wait, multiply a 64-bit address, store, next row" behind a per-row exec branch was 7,400 cycles per tile — a quarter of a
        // K = 2048 tile).  Row r = 4 k + rr of the block, 16-byte chunk rc of the staged row = weight columns 8 (rc ^ (r & 15)).
        u32x4 ev[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) ev[k] = *reinterpret_cast<const u32x4*>(stg + (4 * k + rr) * 256 + (rc << 4));
        const int blk_row = 32 * (2 * j + wm);
        const long long rows_left = cur.row_end - cur.row0 - blk_row;                  // rows of the group in this block (wave-uniform)
        const uint8_t* ob = uniform(reinterpret_cast<const uint8_t*>(p.out + (cur.row0 + blk_row) * p.N + cur.n0 + 128 * wn));
        const unsigned row_bytes = (unsigned)p.N * 2u;
        const unsigned lane_off = (unsigned)rr * row_bytes;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (rows_left >= 32) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const unsigned voff = lane_off + (unsigned)((rc ^ ((4 * k + rr) & 15)) << 4);
            asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2" ::"v"(voff), "v"(ev[k]), "s"(ob + (long long)(4 * k) * row_bytes) : "memory");
          }
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const unsigned voff = lane_off + (unsigned)((rc ^ ((4 * k + rr) & 15)) << 4);
            if (4 * k + rr < rows_left)
              asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2" ::"v"(voff), "v"(ev[k]), "s"(ob + (long long)(4 * k) * row_bytes) : "memory");
          }
        }
      }
    }
#ifdef FL_GEMM3_TIMING
    if (g_g3dbg != nullptr && lane == 0 && slot < 8192) {
      unsigned long long* d = g_g3dbg + ((long long)slot * 4 + wave) * 8;
      d[0] = gt[0]; d[1] = gt[1]; d[2] = gt[2]; d[3] = gt[3]; d[4] = t_loop_end - g0; d[5] = __builtin_readcyclecounter() - t_loop_end;
      d[6] = wall_clock64();
    }
#endif
#ifdef FL_GEMM3_SLOTS
    if (g_g3dbg != nullptr && lane == 0 && slot < 2048) {
      unsigned long long* d = g_g3dbg + 8192 * 4 * 8 + ((long long)slot * 4 + wave) * 26;
#pragma unroll
      for (int i = 0; i < 26; ++i) d[i] = ts[i];
    }
#endif
    iss -= NH;
    if (!has_next) break;
    cur = nxt;
    ws_cur = rd_agpr(0);
    slot = nslot;
    carried = true;
  }   // persistent tile walk
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-fetches past the last tile's end
}

}  // namespace

#if defined(FL_GEMM3_TIMING) || defined(FL_GEMM3_SLOTS)
extern "C" int fl_gemm3_debug_set_buffer(unsigned long long* dev_ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_g3dbg), &dev_ptr, sizeof(dev_ptr));
}
#endif

int fl_gemm_launch_big3(const GemmParams& p_in, const void* A, const float* As, const void* W, const float* Ws,
                        const int32_t* group_meta, hipStream_t stream) {
  GemmParams p = p_in;
  long long m_tiles;
  if (p.mode == kOffset) m_tiles = (p.M + BM3 - 1) / BM3 + p.E;
  else if (p.mode == kMasked) m_tiles = (long long)p.E * ((p.rows_per_group + BM3 - 1) / BM3);
  else m_tiles = (p.M + BM3 - 1) / BM3;
  p.n_tiles = p.N / BN3;
  p.m_tiles_upper = (int)m_tiles;
  const long long blocks = m_tiles * p.n_tiles;
  FL_CHECK_ARG(p.N % BN3 == 0 && p.K / BK >= 2 && p.K / BK <= 64, "fl_grouped_gemm_fp8: shape not for the 192x256 tile");
  FL_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "fl_grouped_gemm_fp8: grid too large");
  FL_CHECK_ARG(p.N < (1 << 24) && p.K < (1 << 24) && (long long)256 * p.K < (1ll << 31), "fl_grouped_gemm_fp8: N, K too large for the 192x256 tile");
  p.total_blocks = (int)blocks;
  long long grid = blocks;   // one workgroup per CU walks the tile list
  {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && fl_device_cu_count(dev, &cus) == FL_OK && cus > 0 && grid > cus) grid = cus;
  }
  grouped_gemm_fp8_big3_kernel<<<dim3((unsigned)grid), dim3(256), 0, stream>>>(p, (const uint8_t*)A, As, (const uint8_t*)W, Ws, group_meta);
  FL_CHECK_LAUNCH("grouped_gemm_fp8_big3_kernel");
  return FL_OK;
}
