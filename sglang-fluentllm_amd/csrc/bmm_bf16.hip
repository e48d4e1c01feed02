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
  int ksplit;                               // 1: four n tiles per workgroup; 4: one tile, k quarters per wave
};

__device__ __forceinline__ v8bf as_v8bf(const uint4 u) {
  union { uint4 u; v8bf v; } x;
  x.u = u;
  return x.v;
}
__device__ __forceinline__ v8bf ld_frag(const uint16_t* p) { return as_v8bf(*reinterpret_cast<const uint4*>(p)); }

__global__ __launch_bounds__(256) void bmm_bf16_nt_kernel(const BmmParams p) {
  __shared__ float red[3][64][16];   // ksplit = 4: partial tiles of waves 1..3
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
  if (p.ksplit > 1) {   // k quarters (multiples of 16)
    const int per = ((p.K / 16 + 3) / 4) * 16;
    k_lo = wave * per < p.K ? wave * per : p.K;
    k_hi = k_lo + per < p.K ? k_lo + per : p.K;
  }
  v16f acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int k = k_lo;
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

  if (p.ksplit > 1) {   // fixed-order reduction: wave 0 adds the partials of waves 1, 2, 3
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave - 1][lane][r] = acc[r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += red[w][lane][r];
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
// 32 * MT2 token rows each (m_per_wg = 128 * MT2) and walk all n tiles of the slice; !BY_M (at most 64 rows per batch): one
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
  static const bool wlds_on = [] {
    const char* e = getenv("FLUENT_BMM_WLDS");
    return e == nullptr || e[0] == '\0' || atoi(e) != 0;
  }();
  if (wlds_on && !out_is_f32 && (K == 128 || K == 512) && (long long)N * K * 2 <= kB2LdsBytes && b_stride_n == K &&
      c_stride_m % 8 == 0 && c_stride_b % 8 == 0 && ((uintptr_t)C % 16) == 0 && b_stride_b % 8 == 0) {
    // at most 64 rows per batch: one 32-row token tile per workgroup, its n tiles over the 4 waves; more: a wave per token tile.
    // The n slices per batch bring the launch to >= one workgroup per CU (a slice has >= 4 n tiles by-n, >= 1 by-m).
    static const int bym_min = [] {   // experiment knob: rows per batch above which a wave owns a token tile
      const char* e = getenv("FLUENT_BMM_BYM_MIN");
      return e != nullptr && e[0] != '\0' ? atoi(e) : 64;
    }();
    const bool by_m = M > bym_min;
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
    static const hipError_t attr_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&bmm_bf16_wlds_kernel<KS_, MT2_, BYM_>),  \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, kB2LdsBytes);              \
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
  // few tiles and a long k (the router GEMM): split k over the four waves of a workgroup
  p.ksplit = (batch * m_tiles * n_tiles < 512 && K >= 1024) ? 4 : 1;
  const long long groups = p.ksplit > 1 ? n_tiles : (n_tiles + 3) / 4;
  const long long blocks = (long long)batch * m_tiles * groups;
  FL_CHECK_ARG(blocks < (1ll << 31), "fl_bmm_bf16_nt: grid too large");
  bmm_bf16_nt_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(p);
  FL_CHECK_LAUNCH("bmm_bf16_nt_kernel");
  return FL_OK;
}
