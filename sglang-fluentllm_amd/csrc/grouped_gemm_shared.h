// Shared between the grouped-GEMM kernels (grouped_gemm_fp8.hip: 128 x 32..128 tiles; grouped_gemm_fp8_big.hip: 256 x 256).
#pragma once
#include "fl_common.h"

namespace fl_gemm {

constexpr int BN = 128;            // weight rows per workgroup (= one 128-row scale block)
constexpr int BK = 128;            // k block (bytes per row per stage)
constexpr int kWBytes = BN * BK;   // 16 KiB
constexpr int kUnit = 0x7F;

enum Mode { kOffset = 0, kContiguous = 1, kMasked = 2, kDense = 3 };

struct GemmParams {
  int mode, E, M, N, K;
  int n_tiles, m_tiles_upper;
  int total_blocks;                     // tiles (x k splits) of the launch; the grid is smaller under deep_gemm.set_num_sms
  long long as_stride_m, as_stride_k, as_stride_g;   // element strides of As (g: masked mode only)
  long long rows_per_group;             // masked mode: padded rows per group
  uint16_t* out;
  int ksplit;                           // > 1: the k blocks are split over ksplit workgroups per tile, f32 partials in ws
  float* ws;                            // [ksplit, M, N] f32
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ v8i mk8(uint4 a, uint4 b) {
  v8i r;
  r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
  return r;
}


// tile index -> (expert, first row, one-past-last valid row) for a token tile of BM rows; false: nothing to do
template <int BM>
__device__ __forceinline__ bool locate_tile(const GemmParams& p, const int32_t* __restrict__ gmeta, const int mt, int& e,
                                            long long& row0, long long& row_end) {
  if (p.mode == kOffset) {
    // wave-parallel search (every wave of the workgroup computes the same answer): lane l looks at group g0 + l, a wave
    // prefix sum over the groups' tile counts locates tile mt — a few dozen instructions per 64 groups instead of a chain
    // of E dependent scalar loads (256 experts: ~20k cycles per workgroup, as long as a whole short-K tile)
    const int lane = threadIdx.x & 63;
    // up to 256 groups: the four rounds' loads go out TOGETHER (one memory round trip instead of four dependent ones:
    // ~2.2 k -> ~1 k cycles per tile; round 3) — more groups: the remaining rounds load as they go
    int lo4[4], hi4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int g = 64 * r + lane;
      lo4[r] = g < p.E ? gmeta[g] : 0;
      hi4[r] = g < p.E ? gmeta[g + 1] : 0;
    }
    int base = 0;
    for (int g0 = 0; g0 < p.E; g0 += 64) {
      const int g = g0 + lane;
      int lo, hi;
      if (g0 < 256) {
        const int r = g0 >> 6;
        lo = r == 0 ? lo4[0] : r == 1 ? lo4[1] : r == 2 ? lo4[2] : lo4[3];
        hi = r == 0 ? hi4[0] : r == 1 ? hi4[1] : r == 2 ? hi4[2] : hi4[3];
      } else {
        lo = g < p.E ? gmeta[g] : 0;
        hi = g < p.E ? gmeta[g + 1] : 0;
      }
      const int tiles = g < p.E ? (hi - lo + BM - 1) / BM : 0;
      int incl = tiles;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
      }
      const int total = __builtin_amdgcn_readlane(incl, 63);
      if (mt < base + total) {
        const int excl = incl - tiles;
        const unsigned long long hit = __ballot(tiles > 0 && mt >= base + excl && mt < base + incl);
        const int src = __builtin_ctzll(hit);
        e = g0 + src;
        row0 = __builtin_amdgcn_readlane(lo, src) + (long long)(mt - base - __builtin_amdgcn_readlane(excl, src)) * BM;
        row_end = __builtin_amdgcn_readlane(hi, src);
        return true;
      }
      base += total;
    }
    return false;
  } else if (p.mode == kContiguous) {
    row0 = (long long)mt * BM;
    if (row0 >= p.M) return false;
    e = gmeta[row0];
    if (e < 0 || e >= p.E) return false;
    row_end = row0 + BM < p.M ? row0 + BM : p.M;
    return true;
  } else if (p.mode == kMasked) {
    const int tpg = (int)((p.rows_per_group + BM - 1) / BM);
    e = mt / tpg;
    if (e >= p.E) return false;
    const int mm = gmeta[e];
    const long long r = (long long)(mt % tpg) * BM;
    if (r >= mm) return false;
    row0 = (long long)e * p.rows_per_group + r;
    row_end = (long long)e * p.rows_per_group + mm;
    return true;
  }
  e = 0;
  row0 = (long long)mt * BM;
  if (row0 >= p.M) return false;
  row_end = p.M;
  return true;
}

}  // namespace fl_gemm

// 256 x 256 tile kernel: many rows per group (grouped_gemm_fp8_big2.hip, round 3)
int fl_gemm_launch_big2(const fl_gemm::GemmParams& p, const void* A, const float* As, const void* W, const float* Ws,
                        const int32_t* group_meta, hipStream_t stream);
// 192 x 256 tile kernel, one wave per SIMD (grouped_gemm_fp8_big3.hip, round 6); needs N % 256 == 0 and K <= 8192
int fl_gemm_launch_big3(const fl_gemm::GemmParams& p, const void* A, const float* As, const void* W, const float* Ws,
                        const int32_t* group_meta, hipStream_t stream);
