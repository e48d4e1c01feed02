// Shared between the MLA decode kernels (mla_decode_fp8.hip: 32/64-row workgroups; mla_decode_fp8_x.hip: 128-row
// workgroups).  gfx950 only.
#pragma once
#include "fl_common.h"

namespace fl_mla {

constexpr int kPage = FL_MLA_PAGE;            // 64 tokens per page / tile
constexpr int kDN = 512;                      // latent (nope) dims, fp8
constexpr int kDR = 64;                       // rope dims, bf16
constexpr int kSlotBytes = kPage * kDN;       // 32 KiB
constexpr int kRopeBytes = kPage * kDR * 2;   // 8 KiB
constexpr int kRingSlots = 4;

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kPShift = 8.0f;               // P' = 2^(y - m_W + 8) <= 256 < 448
constexpr float kRefHeadroom = 2.0f;          // new reference = ceil(max) + 2: fewer reference moves
constexpr float kNegRef = -16384.0f;          // "no reference yet" (finite, integer)
constexpr int kUnitScale = 0x7F;              // E8M0 127 = 2^0
constexpr int kDmaNopePerTile = 32;           // 1-KiB pieces per page

struct Params {
  int bs, s_q, h_q, rows, causal, num_parts, row_groups;
  float scale_log2e;
  const float* descale_q;   // FL_KV_FP8_576 only (device scalars, may be null = 1)
  const float* descale_k;
  long long num_pages;
  long long bt_stride;
  long long bt_cols;   // readable columns of a block-table row (>= 1)
  uint16_t* out;
  float* lse;
  float* o_accum;
  float* lse_accum;
  int partial_bf16;   // o_accum holds bf16 rows (128-row mapping) instead of f32 rows
  int merge_in_kernel;   // mla_decode_fp8_y.hip: split requests are merged by their first piece (no merge kernel)
  const uint16_t* q_bf16;   // mla_decode_fp8_y.hip <0, true>: unquantised query rows [bs * rows, 576] (K4 in the prologue)
};

typedef float float2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ v8bf as_bf8(uint4 v) {
  union { uint4 u; v8bf b; } x;
  x.u = v;
  return x.b;
}
__device__ __forceinline__ v8i make_v8i(uint4 a, uint4 b) {
  v8i r;
  r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
  return r;
}

// split-KV merge of one (request, row): out[req,row,:] = sum_s w_s * o_accum[slot_s,row,:], w_s = softmax_s(weight LSE_s);
// reported lse from the exact LSEs.  One wave per row, lane = 8 consecutive dims.
// NS > 0: compile-time split count (all loads of a row are independent and issued together); NS = 0: run-time count.
// QUAD (NS = 0 only): the four waves of a workgroup share ONE row — wave w accumulates splits [w q, (w + 1) q), q = ceil(ns / 4), against
// the common maxima, wave 0 adds the partial sums of waves 1..3 (through `red`, [3][64][10] floats) in wave order and writes: with a
// handful of rows and dozens of splits per row (decode at bs = 1..8) a wave per row is a chain of ns / 8 memory round trips on a
// mostly empty chip (8.7 us per launch at bs = 1, 64 splits: round 3)
template <int NS, bool QUAD = false>
__device__ __forceinline__ void combine_row(const Params& p, const int req, const int row, const int s0, const int ns_rt,
                                            const int lane, float* red = nullptr, const int wave = 0) {
  const int ns = NS > 0 ? NS : ns_rt;
  float mx = -INFINITY, mxx = -INFINITY;
  if (NS > 0) {
#pragma unroll
    for (int s = 0; s < ns; ++s) {
      mx = fmaxf(mx, p.lse_accum[((long long)(s0 + s) * p.rows + row) * 2 + 0]);
      mxx = fmaxf(mxx, p.lse_accum[((long long)(s0 + s) * p.rows + row) * 2 + 1]);
    }
  } else {
    // run-time split count (up to 64 and more at small batch): lane l looks at splits l, l + 64, ..., then a wave maximum — one
    // memory round trip instead of `ns` of them (round 3: the merge kernel at bs = 1, 64 splits per row)
    for (int s = lane; s < ns; s += 64) {
      const float2 v = *reinterpret_cast<const float2*>(p.lse_accum + ((long long)(s0 + s) * p.rows + row) * 2);
      mx = fmaxf(mx, v.x);
      mxx = fmaxf(mxx, v.y);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mx = fmaxf(mx, __shfl_xor(mx, o));
      mxx = fmaxf(mxx, __shfl_xor(mxx, o));
    }
  }
  float den = 0.f, denx = 0.f;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // splits in chunks of kChunk: the partial rows of a chunk are loaded before the first is used (a run-time split count
  // up to 64 would otherwise be that many dependent round trips per row)
  constexpr int kChunk = NS > 0 ? NS : 8;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  int c_lo = 0, c_hi = ns;
  if (QUAD) {
    const int per = (ns + 3) / 4;
    c_lo = wave * per < ns ? wave * per : ns;
    c_hi = c_lo + per < ns ? c_lo + per : ns;
  }
  for (int c0 = c_lo; c0 < c_hi; c0 += kChunk) {
    u32x4 ua[kChunk], ub[kChunk];
    float ls[kChunk], lx[kChunk];
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {
      const int s = c0 + j < c_hi ? c0 + j : c_hi - 1;   // (unconditional loads; the tail is weighted 0 below)
      const long long base = (long long)(s0 + s) * p.rows + row;
      ls[j] = p.lse_accum[base * 2 + 0];
      lx[j] = p.lse_accum[base * 2 + 1];
      if (p.partial_bf16) {
        ua[j] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(p.o_accum) + base * kDN + lane * 8);
      } else {
        ua[j] = *reinterpret_cast<const u32x4*>(p.o_accum + base * kDN + lane * 8);
        ub[j] = *reinterpret_cast<const u32x4*>(p.o_accum + base * kDN + lane * 8 + 4);
      }
    }
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {
      const bool live = c0 + j < c_hi;
      const float wgt = (!live || mx == -INFINITY) ? 0.f : __expf(ls[j] - mx);
      den += wgt;
      denx += (!live || mxx == -INFINITY) ? 0.f : __expf(lx[j] - mxx);
      float f[8];
      if (p.partial_bf16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { f[2 * q] = __uint_as_float(ua[j][q] << 16); f[2 * q + 1] = __uint_as_float(ua[j][q] & 0xffff0000u); }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) { f[q] = __uint_as_float(ua[j][q]); f[4 + q] = __uint_as_float(ub[j][q]); }
      }
      if (live) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += wgt * f[i];
      }
    }
  }
  if (QUAD) {
    if (wave > 0) {
      float* r = red + ((wave - 1) * 64 + lane) * 10;
#pragma unroll
      for (int i = 0; i < 8; ++i) r[i] = acc[i];
      r[8] = den;
      r[9] = denx;
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      const float* r = red + (w * 64 + lane) * 10;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += r[i];
      den += r[8];
      denx += r[9];
    }
  }
  const float inv = den > 0.f ? 1.f / den : 0.f;
  uint32_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = fl_pack_bf16(acc[2 * i] * inv, acc[2 * i + 1] * inv);
  *reinterpret_cast<uint4*>(p.out + ((long long)req * p.rows + row) * kDN + lane * 8) = make_uint4(o[0], o[1], o[2], o[3]);
  if (lane == 0) {
    const int j = row / p.h_q, h = row - j * p.h_q;
    p.lse[((long long)req * p.h_q + h) * p.s_q + j] = denx > 0.f ? mxx + __logf(denx) : -INFINITY;
  }
}

// The same merge for a BLOCK of 16 rows by one wave (in-kernel merge of mla_decode_fp8_y.hip: the merging workgroup has
// only four waves for 64 rows, so a wave-per-row loop would be 16 dependent round trips to memory behind the fence).
// 8 lanes per row (lane = 8*jr + q: row jr of the pass, 16-B chunks q + 8k of its 512 dims), two passes of 8 rows; per pass
// every LSE and every partial chunk is loaded before the first use: ~3 round trips per 16 rows.  bf16 partials only.
// NS = compile-time split count (2..4: all splits in registers); NS = 0: any count, splits streamed in pairs.
// Every read of a partial is an AGENT-scope relaxed atomic load (sc1: served by the memory side, not by this XCD's
// non-coherent L2) — the writers store the same way, so no cache-wide invalidate / write-back fence is needed: a
// `__threadfence()` per split fragment invalidated the XCD's L2 under the other workgroups' page streams (measured:
// 234 us instead of 161 on the ragged cfg2 workload).
__device__ __forceinline__ float ld_agent_f32(const float* ptr) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent_f32(float* ptr, const float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(ptr), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_16B(void* ptr, const uint4 v) {
  // one 16-B store with the agent-scope cache policy (sc1: what the compiler emits for an agent-scope atomic store; the
  // atomic builtins stop at 8 bytes).  The caller's s_waitcnt vmcnt(0) covers it.
  typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
  const u32x4_ d = {v.x, v.y, v.z, v.w};
  // (s_nop: a VMEM store of more than 8 bytes reads its data registers for two more cycles — the hazard recogniser does
  //  not look inside inline asm, and the next loop iteration overwrites them)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(ptr), "v"(d) : "memory");
}
template <int NS>
__device__ __forceinline__ void combine_rows16(const Params& p, const int req, const int row_base, const int s0, const int ns_rt,
                                               const int lane) {
  const int ns = NS > 0 ? NS : ns_rt;
  const int jr = lane >> 3, q = lane & 7;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const uint16_t* part = reinterpret_cast<const uint16_t*>(p.o_accum);
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    const int row = row_base + pass * 8 + jr;
    const bool ok = row < p.rows;
    const int rr = ok ? row : p.rows - 1;   // (unconditional loads on a clamped row; stores are predicated)
    float mx = -INFINITY, mxx = -INFINITY;
    for (int s = 0; s < ns; ++s) {
      mx = fmaxf(mx, ld_agent_f32(p.lse_accum + ((long long)(s0 + s) * p.rows + rr) * 2 + 0));
      mxx = fmaxf(mxx, ld_agent_f32(p.lse_accum + ((long long)(s0 + s) * p.rows + rr) * 2 + 1));
    }
    float acc[8][8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[k][i] = 0.f;
    float den = 0.f, denx = 0.f;
    constexpr int kChunk = NS > 0 ? NS : 2;
    for (int c0 = 0; c0 < ns; c0 += kChunk) {
      u32x4 ua[kChunk][8];
      float ls[kChunk], lx[kChunk];
#pragma unroll
      for (int j = 0; j < kChunk; ++j) {
        const int s = c0 + j < ns ? c0 + j : ns - 1;
        const long long base = (long long)(s0 + s) * p.rows + rr;
        ls[j] = ld_agent_f32(p.lse_accum + base * 2 + 0);
        lx[j] = ld_agent_f32(p.lse_accum + base * 2 + 1);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const unsigned long long* src = reinterpret_cast<const unsigned long long*>(part + base * kDN + (q + 8 * k) * 8);
          const unsigned long long lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ua[j][k] = u32x4{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
        }
      }
#pragma unroll
      for (int j = 0; j < kChunk; ++j) {
        const bool live = c0 + j < ns;
        const float wgt = (!live || mx == -INFINITY) ? 0.f : __expf(ls[j] - mx);
        den += wgt;
        denx += (!live || mxx == -INFINITY) ? 0.f : __expf(lx[j] - mxx);
        if (live) {
#pragma unroll
          for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int h = 0; h < 4; ++h) {
              acc[k][2 * h] += wgt * __uint_as_float(ua[j][k][h] << 16);
              acc[k][2 * h + 1] += wgt * __uint_as_float(ua[j][k][h] & 0xffff0000u);
            }
        }
      }
    }
    const float inv = den > 0.f ? 1.f / den : 0.f;
    if (ok) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        *reinterpret_cast<uint4*>(p.out + ((long long)req * p.rows + row) * kDN + (q + 8 * k) * 8) =
            make_uint4(fl_pack_bf16(acc[k][0] * inv, acc[k][1] * inv), fl_pack_bf16(acc[k][2] * inv, acc[k][3] * inv),
                       fl_pack_bf16(acc[k][4] * inv, acc[k][5] * inv), fl_pack_bf16(acc[k][6] * inv, acc[k][7] * inv));
      if (q == 0) {
        const int j = row / p.h_q, h = row - j * p.h_q;
        p.lse[((long long)req * p.h_q + h) * p.s_q + j] = denx > 0.f ? mxx + __logf(denx) : -INFINITY;
      }
    }
  }
}

}  // namespace fl_mla

// split-KV combine (defined in mla_decode_fp8.hip)
int fl_mla_launch_combine(const fl_mla::Params& p, const int32_t* num_splits, hipStream_t stream);
