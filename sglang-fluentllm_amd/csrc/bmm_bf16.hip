// B1 — batched bf16 "NT" GEMM on MFMA for the small dense ops BETWEEN the big kernels of a DeepSeek decode layer
// (VERDICT r2 "what's missing" 1 / "next round" 7):
//   * weight absorption  q_nope @ w_kc   torch.bmm(q_nope.transpose(0,1) [H,T,128], w_kc [H,128,512], out=Q[..., :512])
//                                        (srt/models/deepseek_v2.py:840; w_kc is stored k-contiguous: :1632)
//   * value absorption   attn_out @ w_vc torch.bmm(attn_output.transpose(0,1) [H,T,512], w_vc [H,512,128])   (:886; :1633)
//   * router logits      flashinfer.dsv3_router_gemm(hidden [T,7168], weight [256,7168], out_dtype=f32)     (:177-179)
// One kernel:  C[b, m, n] = sum_k A[b, m, k] * B[b, n, k]   — both operands k-contiguous (what the reference's weight
// layouts are), fp32 accumulation on v_mfma_f32_32x32x16_bf16, output bf16 (RNE, like torch.bmm's bf16 result) or f32.
//
// Mapping (latency-, not throughput-bound: T <= a few hundred rows, K = 128 .. 7168): a wave owns a 32 x 32 output tile,
// "SwapAB" like the big kernels — the weight rows (n) go on the MFMA M side, the token rows (m) on the N side, so a lane
// holds ONE token row and 4 consecutive n per accumulator group: its stores are 8 / 16 contiguous bytes.  Operand
// fragments are 16-byte global loads straight into registers (lane (i, kq) reads k = 8 kq .. 8 kq + 7 of row i): with a
// few dozen k steps and operands that sit in L2 an LDS round trip buys nothing.  A workgroup = 4 waves = 4 neighbouring
// n tiles of one (batch, m tile): the token fragment is the same for the four (L1 hits).  Long-k problems (the router:
// K = 7168, only 8 x ceil(T/32) tiles) split k over the 4 waves instead and reduce through LDS in a fixed order.
#include "fl_common.h"
#include <cstdlib>

namespace {

struct BmmParams {
  const uint16_t* A;
  const uint16_t* B;
  void* C;
  int batch, M, N, K;
  long long sAb, sAm, sBb, sBn, sCb, sCm;   // element strides (k and n are contiguous)
  int out_f32;
  int ksplit;                               // 1: four n tiles per workgroup; 4 / 16: one tile, a k range per wave
};

__device__ __forceinline__ v8bf as_v8bf(const uint4 u) {
  union { uint4 u; v8bf v; } x;
  x.u = u;
  return x.v;
}
__device__ __forceinline__ v8bf ld_frag(const uint16_t* p) { return as_v8bf(*reinterpret_cast<const uint4*>(p)); }

template <int WAVES>   // 4: four n tiles per workgroup, or k quarters of one tile; 16 (k split only): k sixteenths
__global__ __launch_bounds__(64 * WAVES) void bmm_bf16_nt_kernel(const BmmParams p) {
  __shared__ float4 red[WAVES - 1][4][64];   // k split: partial tiles of waves 1 .. WAVES-1 ([group of 4 registers][lane]: 16 B per
                                             // lane, conflict-free — [lane][16] floats put all lanes on two banks: 32-way conflicts)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, kq = lane >> 5;
  const int n_groups = p.ksplit > 1 ? p.N / 32 : (p.N / 32 + 3) / 4;
  const int m_tiles = (p.M + 31) / 32;
  int id = blockIdx.x;
  const int ng = id % n_groups;
  id /= n_groups;
  const int mt = id % m_tiles;
  const int b = id / m_tiles;
  const int nt = p.ksplit > 1 ? ng : ng * 4 + wave;
  const bool tile_ok = nt * 32 < p.N;
  const int n0 = tile_ok ? nt * 32 : 0;
  const int m0 = mt * 32;
  const int mrow = m0 + li < p.M ? m0 + li : p.M - 1;   // (clamped loads; the tail rows are not stored)
  const uint16_t* pa = p.A + (long long)b * p.sAb + (long long)mrow * p.sAm + 8 * kq;
  const uint16_t* pb = p.B + (long long)b * p.sBb + (long long)(n0 + li) * p.sBn + 8 * kq;
  int k_lo = 0, k_hi = p.K;
  if (p.ksplit > 1) {   // one k range per wave (multiples of 16)
    const int per = ((p.K / 16 + WAVES - 1) / WAVES) * 16;
    k_lo = wave * per < p.K ? wave * per : p.K;
    k_hi = k_lo + per < p.K ? k_lo + per : p.K;
  }
  v16f acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int k = k_lo;
  for (; k + 128 <= k_hi; k += 128) {   // eight k steps per trip: 16 loads in flight
    v8bf fa[8], fb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      fa[u] = ld_frag(pa + k + 16 * u);
      fb[u] = ld_frag(pb + k + 16 * u);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[u], fa[u], acc, 0, 0, 0);
  }
  for (; k + 64 <= k_hi; k += 64) {   // four k steps per trip: 8 loads in flight
    v8bf fa[4], fb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      fa[u] = ld_frag(pa + k + 16 * u);
      fb[u] = ld_frag(pb + k + 16 * u);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[u], fa[u], acc, 0, 0, 0);
  }
  for (; k < k_hi; k += 16) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(pb + k), ld_frag(pa + k), acc, 0, 0, 0);

  if (p.ksplit > 1) {   // fixed-order reduction: wave 0 adds the partials of waves 1, 2, ...
    if (wave > 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) red[wave - 1][g][lane] = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < WAVES - 1; ++w)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = red[w][g][lane];
        acc[4 * g] += v.x; acc[4 * g + 1] += v.y; acc[4 * g + 2] += v.z; acc[4 * g + 3] += v.w;
      }
  }
  if (!tile_ok || m0 + li >= p.M) return;
  // D^T[n, m]: lane = token row m0 + li, registers 4g .. 4g+3 = n0 + 8g + 4kq + (0..3)
  if (p.out_f32) {
    float* c = reinterpret_cast<float*>(p.C) + (long long)b * p.sCb + (long long)(m0 + li) * p.sCm + n0 + 4 * kq;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(c + 8 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
  } else {
    uint16_t* c = reinterpret_cast<uint16_t*>(p.C) + (long long)b * p.sCb + (long long)(m0 + li) * p.sCm + n0 + 4 * kq;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<uint2*>(c + 8 * g) = make_uint2(fl_pack_bf16(acc[4 * g], acc[4 * g + 1]), fl_pack_bf16(acc[4 * g + 2], acc[4 * g + 3]));
  }
}

// ---- B2 (round 3): the weight-absorption shapes with the batch's WEIGHT MATRIX RESIDENT IN LDS ----
// bmm_bf16_nt_kernel above re-reads both operands for every 32 x 32 output tile (16 flop per byte from L2: 268 MB of L1 fills for
// the 4.3 GFLOP of q_nope @ w_kc at T = 256 — measured 43 us).  Both absorption shapes have a per-head weight matrix of exactly
// 128 KiB ([512 x 128] w_kc, [128 x 512] w_vc): one workgroup stages it ONCE in LDS by LDS-DMA (every weight byte leaves L2 once
// per workgroup), and each of its 4 waves walks its own token rows against the whole matrix: token fragments (the MFMA B operand,
// K / 16 k-steps of 16 B per lane) stay in registers, weight fragments come from LDS by conflict-free ds_read_b128 (16-B chunk c
// of row n stored at chunk c ^ (n & 15); the XOR is applied on the DMA's source address).  One accumulator tile is live at a time
// (n-outer loop), so K = 512 (128 registers of token fragments) and K = 128 (32, two token tiles per wave when there are enough
// rows) are the same code.  Output: the two lane halves exchange accumulator groups (v_permlane32_swap) so that a lane stores
// 2 x 16 B = 16 consecutive n of its token row — a wave writes whole rows of the (possibly strided) destination.
constexpr int kB2LdsBytes = 128 * 1024;
template <int KS>
struct B2 {
  static constexpr int kRowBytes = KS * 32;               // K * 2
  static constexpr int kChunks = kRowBytes / 16;
};

// Work split.  Grid = batch x m_splits x n_splits; a workgroup stages the n_per_wg weight rows of its slice.  BY_M: its 4 waves own
// 32 * MT2 token rows each (m_per_wg = 128 * MT2) and walk all n tiles of the slice; !BY_M (at most 32 rows per batch): one
// 32-row token tile per workgroup, shared by the 4 waves, wave w takes n tiles w, w + 4, ... of the slice.
template <int KS, int MT2, bool BY_M>
__global__ __launch_bounds__(256) void bmm_bf16_wlds_kernel(const BmmParams p, const int m_per_wg, const int n_per_wg) {
  extern __shared__ __attribute__((aligned(16))) uint8_t b2_smem[];   // n_per_wg * RB bytes
  constexpr int RB = B2<KS>::kRowBytes;
  constexpr int CH = KS / 8;   // chunks of 8 k-steps per n tile
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, kq = lane >> 5;
  const int m_splits = (p.M + m_per_wg - 1) / m_per_wg;
  const int n_splits = p.N / n_per_wg;
  int id = blockIdx.x;
  const int ns = id % n_splits;
  id /= n_splits;
  const int b = id / m_splits;
  const int m_base = (id % m_splits) * m_per_wg;
  const int n_base = ns * n_per_wg;
  // ---- stage B[b][n_base .. n_base + n_per_wg) ([., K] k-contiguous) in LDS: 1-KiB pieces round-robin over the waves ----
  {
    const int pieces = n_per_wg * RB / 1024;
    const uint8_t* gb = reinterpret_cast<const uint8_t*>(p.B + (long long)b * p.sBb + (long long)n_base * p.sBn);
    for (int P = wave; P < pieces; P += 4) {
      const int off = P * 1024 + lane * 16;        // this lane's 16 B of LDS
      const int n = off / RB;                      // row inside the slice (n_base is a multiple of 32: the XOR key is the same)
      const int pc = (off % RB) >> 4;              // stored chunk position
      fl_dma16(gb + (long long)n * p.sBn * 2 + ((pc ^ (n & 15)) << 4), b2_smem + P * 1024);
    }
  }
  // ---- token fragments (B operand): rows m0 + 32 mm + li.  A dot product does not care in which ORDER k is walked as long as both
  //      operands agree: MFMA step s = 4 a + t takes from lane (row, kq) the 8 elements at k = 64 a + 32 kq + 8 t — so a lane
  //      reads 64 CONTIGUOUS bytes per a (a lane pair one whole 128-B line, used up by four back-to-back loads) instead of 16 B
  //      out of every 32; the weight fragment of the step is LDS chunk 8 a + 4 kq + t (any chunk is as cheap as any other). ----
  const int m0 = BY_M ? m_base + wave * (32 * MT2) : m_base;
  v8bf fb[MT2][KS];
  bool any = false;
#pragma unroll
  for (int mm = 0; mm < MT2; ++mm) {
    const int m = m0 + 32 * mm + li;
    any |= m0 + 32 * mm < p.M && m0 + 32 * mm < m_base + m_per_wg;
    const int mc = m < p.M ? m : p.M - 1;   // (clamped loads; the tail rows are not stored)
    const uint16_t* pa = p.A + (long long)b * p.sAb + (long long)mc * p.sAm + 32 * kq;
#pragma unroll
    for (int s = 0; s < KS; ++s) fb[mm][s] = ld_frag(pa + 64 * (s >> 2) + 8 * (s & 3));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA pieces (and the fragments) have landed
  __syncthreads();
  const int n_tiles = n_per_wg / 32;
  const int nt0 = BY_M ? 0 : wave, nt_step = BY_M ? 1 : 4;
  if (!any || nt0 >= n_tiles) return;

  // weight fragments of 8 k-steps (chunk cc of n tile nt) — double-buffered against the MFMAs of the previous chunk
  auto load8 = [&](v8bf (&fa)[8], int nt, const int cc) {
    nt = nt < n_tiles ? nt : n_tiles - 1;   // (the prefetch past the last tile re-reads it)
    const int n = nt * 32 + li;
    const uint8_t* wrow = b2_smem + n * RB;
#pragma unroll
    for (int s = 0; s < 8; ++s)   // step 8 cc + s = 4 a + t  ->  chunk 8 a + 4 kq + t
      fa[s] = as_v8bf(*reinterpret_cast<const uint4*>(wrow + (((8 * (2 * cc + (s >> 2)) + 4 * kq + (s & 3)) ^ (n & 15)) << 4)));
  };
  auto mfma8 = [&](v16f (&acc)[MT2], const v8bf (&fa)[8], const int cc) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int mm = 0; mm < MT2; ++mm) acc[mm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s], fb[mm][8 * cc + s], acc[mm], 0, 0, 0);
  };
  // D^T[n, m]: lane (li = token row, kq) holds n = 8 g + 4 kq + (0..3) in registers 4 g .. 4 g + 3.  After the swap of
  // (acc[r], acc[r + 8]) between the lane halves, half 0 holds n 0..15 and half 1 holds n 16..31 of its row: 2 x 16-B stores.
  auto store_tile = [&](const v16f (&acc)[MT2], const int nt) {
#pragma unroll
    for (int mm = 0; mm < MT2; ++mm) {
      uint32_t own[8], oth[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mm][r]), __float_as_uint(acc[mm][r + 8]), false, false);
        own[r] = kq == 0 ? sw[0] : sw[1];   // this lane's own value of the group pair it keeps
        oth[r] = kq == 0 ? sw[1] : sw[0];   // the partner's value of the same registers
      }
      const int m = m0 + 32 * mm + li;
      if (m < p.M && m < m_base + m_per_wg) {
        uint16_t* c = reinterpret_cast<uint16_t*>(p.C) + (long long)b * p.sCb + (long long)m * p.sCm + n_base + nt * 32 + 16 * kq;
        // n (relative to 16 kq + 8 g): 0..3 from the kq = 0 lane of the pair, 4..7 from the kq = 1 lane
        const uint32_t* lo = kq == 0 ? own : oth;
        const uint32_t* hi = kq == 0 ? oth : own;
#pragma unroll
        for (int g = 0; g < 2; ++g)
          *reinterpret_cast<uint4*>(c + 8 * g) =
              make_uint4(fl_pack_bf16(__uint_as_float(lo[4 * g]), __uint_as_float(lo[4 * g + 1])),
                         fl_pack_bf16(__uint_as_float(lo[4 * g + 2]), __uint_as_float(lo[4 * g + 3])),
                         fl_pack_bf16(__uint_as_float(hi[4 * g]), __uint_as_float(hi[4 * g + 1])),
                         fl_pack_bf16(__uint_as_float(hi[4 * g + 2]), __uint_as_float(hi[4 * g + 3])));
      }
    }
  };
  auto zero = [&](v16f (&acc)[MT2]) {
#pragma unroll
    for (int mm = 0; mm < MT2; ++mm)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mm][r] = 0.f;
  };

  v8bf fa0[8], fa1[8];
  load8(fa0, nt0, 0);
  if constexpr (CH == 1) {
    for (int nt = nt0; nt < n_tiles; nt += 2 * nt_step) {   // two n tiles per trip: the buffer parity is static
      v16f acc[MT2], acc2[MT2];
      zero(acc);
      load8(fa1, nt + nt_step, 0);
      mfma8(acc, fa0, 0);
      zero(acc2);
      load8(fa0, nt + 2 * nt_step, 0);
      if (nt + nt_step < n_tiles) mfma8(acc2, fa1, 0);
      store_tile(acc, nt);
      if (nt + nt_step < n_tiles) store_tile(acc2, nt + nt_step);
    }
  } else {
    static_assert(CH == 1 || CH == 4, "K = 128 or 512");
    for (int nt = nt0; nt < n_tiles; nt += nt_step) {
      v16f acc[MT2];
      zero(acc);
      load8(fa1, nt, 1);
      mfma8(acc, fa0, 0);
      load8(fa0, nt, 2);
      mfma8(acc, fa1, 1);
      load8(fa1, nt, 3);
      mfma8(acc, fa0, 2);
      load8(fa0, nt + nt_step, 0);
      mfma8(acc, fa1, 3);
      store_tile(acc, nt);
    }
  }
}

// ---- B3 (round 3): the router GEMM (and any [T <= a few hundred, K long] x [N small, K]^T product) split over K ACROSS workgroups ----
// flashinfer.dsv3_router_gemm (srt/models/deepseek_v2.py:177-179): hidden [T, 7168] x gate weight [256, 7168]^T -> f32 logits.  The
// output has 8 x 8 tiles of 32 x 32 at T = 256: bmm_bf16_nt_kernel ran it on 64 CUs and — measured — at the rate its CU's vector-memory
// path retires row-strided 16-byte loads (one 128-B line per lane row: 32 cycles per wave load, the same whether 4 or 16 waves split
// the k loop: 27.5 us either way).  Here a workgroup owns a 64 x 64 output tile and ONE k range of K / ksplit elements: both operand
// slabs come in by LDS-DMA once (row-contiguous 1-KiB pieces), four waves compute its 32 x 32 quarters from LDS (conflict-free
// ds_read_b128: 16-B chunk c of row r at c ^ (r & 7) inside its 128-B segment), f32 partials [ksplit, T, N] go to the caller's
// workspace and a second launch adds them in split order (deterministic: no float atomics) and rounds.
struct RtParams {
  const uint16_t* A;   // [M, K] tokens
  const uint16_t* B;   // [N, K] weight rows
  float* ws;           // [ksplit, M, N]
  void* C;
  int M, N, K, kc, ksplit, out_f32;
  long long sAm, sBn, sCm;
};

__global__ __launch_bounds__(256) void router_partial_kernel(const RtParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t rt_smem[];   // A slab [64][kc] bf16, then B slab [64][kc]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, kq = lane >> 5;
  const int n_tiles = p.N / 64, m_tiles = (p.M + 63) / 64;
  int id = blockIdx.x;
  const int split = id % p.ksplit;
  id /= p.ksplit;
  const int nt = id % n_tiles, mt = id / n_tiles;
  (void)m_tiles;
  const int RB = p.kc * 2;                 // bytes per slab row: a multiple of 128
  const int slab = 64 * RB;
  const int pieces = slab / 1024;          // per operand
  const long long k0 = (long long)split * p.kc;
  for (int P = wave; P < 2 * pieces; P += 4) {
    const bool isB = P >= pieces;
    const int off = (isB ? P - pieces : P) * 1024 + lane * 16;
    const int r = off / RB;
    const int cb = off % RB;                                     // stored byte position inside the row
    const int src_b = (cb & ~127) | ((((cb >> 4) & 7) ^ (r & 7)) << 4);
    long long row = isB ? nt * 64 + r : mt * 64 + r;
    if (!isB && row >= p.M) row = p.M - 1;                       // (clamped; tail rows are not stored)
    const uint16_t* src = (isB ? p.B + row * p.sBn : p.A + row * p.sAm) + k0;
    fl_dma16(reinterpret_cast<const uint8_t*>(src) + src_b, rt_smem + P * 1024);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int mw = wave & 1, nw = wave >> 1;
  const uint8_t* a_row = rt_smem + (mw * 32 + li) * RB;          // token row  -> MFMA B operand
  const uint8_t* b_row = rt_smem + slab + (nw * 32 + li) * RB;   // weight row -> MFMA A operand
  const int key = li & 7;                                        // ((mw or nw) * 32 + li) & 7
  v16f acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int steps = p.kc / 16;
#pragma unroll 4
  for (int s = 0; s < steps; ++s) {
    const int c = 2 * s + kq;                                     // logical 16-B chunk of the row
    const int o = ((c & ~7) | ((c & 7) ^ key)) << 4;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_v8bf(*reinterpret_cast<const uint4*>(b_row + o)),
                                                  as_v8bf(*reinterpret_cast<const uint4*>(a_row + o)), acc, 0, 0, 0);
  }
  const int m = mt * 64 + mw * 32 + li;
  if (m >= p.M) return;
  // D^T[n, m]: lane = token row, registers 4 g .. 4 g + 3 = n 8 g + 4 kq + (0..3)
  float* w = p.ws + ((long long)split * p.M + m) * p.N + nt * 64 + nw * 32 + 4 * kq;
#pragma unroll
  for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(w + 8 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
}

// C[m, n] = sum over the splits: 8 lanes per 4 consecutive n — lane j adds splits j, j + 8, ... (all its loads in flight at once),
// then a fixed xor tree over the 8 lanes: the same association every run (deterministic; no float atomics).  One thread per
// element with a loop over the splits was a chain of up to 14 memory round trips on 16 workgroups at T = 64.
__global__ __launch_bounds__(256) void router_reduce_kernel(const RtParams p) {
  const long long MN = (long long)p.M * p.N;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const int j = (int)(gid & 7);
  long long i = (gid >> 3) * 4;
  const bool ok = i < MN;
  if (!ok) i = 0;   // (whole 8-lane groups are in or out; out-of-range groups still take part in the shuffles)
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s0 = j; s0 < p.ksplit; s0 += 64) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int su = s0 + 8 * u < p.ksplit ? s0 + 8 * u : p.ksplit - 1;   // (clamped: unconditional loads)
      v[u] = *reinterpret_cast<const float4*>(p.ws + (long long)su * MN + i);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (s0 + 8 * u < p.ksplit) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    a.x += __shfl_xor(a.x, o); a.y += __shfl_xor(a.y, o); a.z += __shfl_xor(a.z, o); a.w += __shfl_xor(a.w, o);
  }
  if (!ok || j != 0) return;
  const long long m = i / p.N;
  const int n = (int)(i % p.N);
  if (p.out_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + m * p.sCm + n) = a;
  else *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.C) + m * p.sCm + n) = make_uint2(fl_pack_bf16(a.x, a.y), fl_pack_bf16(a.z, a.w));
}

}  // namespace

extern "C" int fl_bmm_bf16_nt(const void* A, const void* B, void* C, int batch, int64_t M, int N, int K, int64_t a_stride_b,
                              int64_t a_stride_m, int64_t b_stride_b, int64_t b_stride_n, int64_t c_stride_b, int64_t c_stride_m,
                              int out_is_f32, fl_stream_t stream) {
  if (batch == 0 || M == 0) return FL_OK;
  FL_CHECK_ARG(A && B && C && batch > 0 && M > 0 && M < (1ll << 30), "fl_bmm_bf16_nt: bad arguments");
  FL_CHECK_ARG(N > 0 && N % 32 == 0 && K > 0 && K % 16 == 0, "fl_bmm_bf16_nt: N=%d must be a multiple of 32, K=%d of 16", N, K);
  FL_CHECK_ARG(a_stride_m % 8 == 0 && b_stride_n % 8 == 0 && a_stride_b % 8 == 0 && b_stride_b % 8 == 0 &&
                   ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0,
               "fl_bmm_bf16_nt: operand rows must be 16-byte aligned (strides multiples of 8 elements)");
  FL_CHECK_ARG(c_stride_m % 4 == 0 && c_stride_b % 4 == 0 && ((uintptr_t)C % (out_is_f32 ? 16 : 8)) == 0,
               "fl_bmm_bf16_nt: output rows must be 8-byte (bf16) / 16-byte (f32) aligned");
  BmmParams p;
  p.A = (const uint16_t*)A; p.B = (const uint16_t*)B; p.C = C;
  p.batch = batch; p.M = (int)M; p.N = N; p.K = K;
  p.sAb = a_stride_b; p.sAm = a_stride_m; p.sBb = b_stride_b; p.sBn = b_stride_n; p.sCb = c_stride_b; p.sCm = c_stride_m;
  p.out_f32 = out_is_f32;
  p.ksplit = 1;
  // the absorption shapes (bf16 out, a 128-KiB weight matrix per batch): weights resident in LDS (bmm_bf16_wlds_kernel)
  if (!out_is_f32 && (K == 128 || K == 512) && (long long)N * K * 2 <= kB2LdsBytes && b_stride_n == K &&
      c_stride_m % 8 == 0 && c_stride_b % 8 == 0 && ((uintptr_t)C % 16) == 0 && b_stride_b % 8 == 0) {
    // at most 32 rows per batch: one 32-row token tile per workgroup, its n tiles over the 4 waves; more: a wave per token tile
    // (measured at T = 64, H = 128: 7.7 / 10.2 us by token tile vs 10.4 / 12.0 us by n tile).
    // The n slices per batch bring the launch to >= one workgroup per CU (a slice has >= 4 n tiles by-n, >= 1 by-m).
    const bool by_m = M > 32;
    const long long wgs128 = (long long)batch * ((M + 127) / 128);
    const bool two = by_m && K == 128 && wgs128 > 512;   // plenty of workgroups: two token tiles per wave (half the LDS reads per flop)
    const int m_per_wg = by_m ? (two ? 256 : 128) : 32;
    const long long wg_mb = (long long)batch * ((M + m_per_wg - 1) / m_per_wg);
    // n slices per batch: towards two workgroups per CU (slices of at most 64 KiB: one stages while the other computes)
    int n_splits = 1;
    const int min_tiles = by_m ? 1 : 4;
    // (every slice's workgroup re-reads the token rows: no more slices than keeps that below the weight bytes — N / m_per_wg;
    //  the value absorption, whose token rows are as large as its weight matrix, is never sliced: measured 16.4 vs 11.3 us at T = 128)
    while (wg_mb * n_splits < 512 && (N / 32) % (2 * n_splits) == 0 && (N / 32) / (2 * n_splits) >= min_tiles &&
           2 * n_splits * m_per_wg <= N)
      n_splits *= 2;
    const int n_per_wg = N / n_splits;
    const long long blocks = wg_mb * n_splits;
    FL_CHECK_ARG(blocks < (1ll << 31), "fl_bmm_bf16_nt: grid too large");
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)n_per_wg * K * 2;
#define FL_B2_LAUNCH(KS_, MT2_, BYM_)                                                                                          \
  do {                                                                                                                         \
    static std::atomic<unsigned char> done_[64];                                                                               \
    const hipError_t attr_ = fl_set_max_dynamic_lds(reinterpret_cast<const void*>(&bmm_bf16_wlds_kernel<KS_, MT2_, BYM_>),      \
                                                    kB2LdsBytes, done_);                                                        \
    FL_CHECK_ARG(attr_ == hipSuccess, "fl_bmm_bf16_nt: hipFuncSetAttribute(%d)", (int)attr_);                                  \
    bmm_bf16_wlds_kernel<KS_, MT2_, BYM_><<<grid, block, lds, s>>>(p, m_per_wg, n_per_wg);                                      \
  } while (0)
    if (K == 128) {
      if (!by_m) FL_B2_LAUNCH(8, 1, false);
      else if (two) FL_B2_LAUNCH(8, 2, true);
      else FL_B2_LAUNCH(8, 1, true);
    } else {
      if (!by_m) FL_B2_LAUNCH(32, 1, false);
      else FL_B2_LAUNCH(32, 1, true);
    }
#undef FL_B2_LAUNCH
    FL_CHECK_LAUNCH("bmm_bf16_wlds_kernel");
    return FL_OK;
  }
  const long long m_tiles = (M + 31) / 32, n_tiles = N / 32;
  // few tiles and a long k (the router GEMM: 64 tiles x 448 k steps at T = 256): split k over the waves of a workgroup — 16 waves
  // (28 k steps each, two trips of loads) when k is long enough: the tile's loop is a chain of memory round trips, 29 us with 4 waves
  p.ksplit = (batch * m_tiles * n_tiles < 512 && K >= 1024) ? (K >= 4096 ? 16 : 4) : 1;
  const long long groups = p.ksplit > 1 ? n_tiles : (n_tiles + 3) / 4;
  const long long blocks = (long long)batch * m_tiles * groups;
  FL_CHECK_ARG(blocks < (1ll << 31), "fl_bmm_bf16_nt: grid too large");
  if (p.ksplit == 16) bmm_bf16_nt_kernel<16><<<dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream>>>(p);
  else bmm_bf16_nt_kernel<4><<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(p);
  FL_CHECK_LAUNCH("bmm_bf16_nt_kernel");
  return FL_OK;
}

// Split-K form for few output tiles and a long K (the router GEMM).  `workspace` holds the f32 partials; returns
// FL_ERR_INVALID_ARG when the shape does not fit this form (the caller then uses fl_bmm_bf16_nt).
// k range per workgroup: a multiple of 64 elements that divides K, both slabs (128 rows x 2 kc bytes) within 128 KiB, and as many
// splits as bring the launch towards one workgroup per CU; 0: no usable split
static int rt_pick_ksplit(int64_t M, int N, int K) {
  const long long tiles = ((M + 63) / 64) * (long long)(N / 64);
  int ksplit = 0;
  for (int ks = 1; ks <= K / 64; ++ks) {
    if (K % ks != 0 || (K / ks) % 64 != 0) continue;
    if ((long long)(K / ks) * 2 * 128 > kB2LdsBytes) continue;
    ksplit = ks;
    if (tiles * ks >= 200) break;
  }
  return ksplit;
}
extern "C" int64_t fl_gemm_bf16_nt_splitk_workspace_bytes(int64_t M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || N % 64 != 0 || K % 64 != 0) return 0;
  return (int64_t)rt_pick_ksplit(M, N, K) * M * N * 4;
}
extern "C" int fl_gemm_bf16_nt_splitk(const void* A, const void* B, void* C, int64_t M, int N, int K, int64_t a_stride_m,
                                      int64_t b_stride_n, int64_t c_stride_m, int out_is_f32, void* workspace,
                                      int64_t workspace_bytes, fl_stream_t stream) {
  if (M == 0) return FL_OK;
  FL_CHECK_ARG(A && B && C && workspace && M > 0 && M < (1ll << 24), "fl_gemm_bf16_nt_splitk: bad arguments");
  FL_CHECK_ARG(N > 0 && N % 64 == 0 && K > 0 && K % 64 == 0, "fl_gemm_bf16_nt_splitk: N=%d, K=%d must be multiples of 64", N, K);
  FL_CHECK_ARG(a_stride_m % 8 == 0 && b_stride_n % 8 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0,
               "fl_gemm_bf16_nt_splitk: operand rows must be 16-byte aligned");
  FL_CHECK_ARG(c_stride_m % 4 == 0 && ((uintptr_t)C % 16) == 0 && ((uintptr_t)workspace % 16) == 0, "fl_gemm_bf16_nt_splitk: output alignment");
  const long long tiles = ((M + 63) / 64) * (long long)(N / 64);
  const int ksplit = rt_pick_ksplit(M, N, K);
  FL_CHECK_ARG(ksplit > 0, "fl_gemm_bf16_nt_splitk: K=%d has no usable split", K);
  FL_CHECK_ARG((long long)ksplit * M * N * 4 <= workspace_bytes, "fl_gemm_bf16_nt_splitk: workspace too small (%lld bytes)",
               (long long)workspace_bytes);
  RtParams p;
  p.A = (const uint16_t*)A; p.B = (const uint16_t*)B; p.ws = (float*)workspace; p.C = C;
  p.M = (int)M; p.N = N; p.K = K; p.kc = K / ksplit; p.ksplit = ksplit; p.out_f32 = out_is_f32;
  p.sAm = a_stride_m; p.sBn = b_stride_n; p.sCm = c_stride_m;
  static std::atomic<unsigned char> done_[64];
  const hipError_t attr_ = fl_set_max_dynamic_lds(reinterpret_cast<const void*>(&router_partial_kernel), kB2LdsBytes, done_);
  FL_CHECK_ARG(attr_ == hipSuccess, "fl_gemm_bf16_nt_splitk: hipFuncSetAttribute(%d)", (int)attr_);
  hipStream_t s = (hipStream_t)stream;
  router_partial_kernel<<<dim3((unsigned)(tiles * ksplit)), dim3(256), (size_t)p.kc * 2 * 128, s>>>(p);
  FL_CHECK_LAUNCH("router_partial_kernel");
  router_reduce_kernel<<<dim3((unsigned)((M * N / 4 * 8 + 255) / 256)), dim3(256), 0, s>>>(p);
  FL_CHECK_LAUNCH("router_reduce_kernel");
  return FL_OK;
}
