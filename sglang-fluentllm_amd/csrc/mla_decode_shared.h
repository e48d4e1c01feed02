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
  uint16_t* out;
  float* lse;
  float* o_accum;
  float* lse_accum;
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

}  // namespace fl_mla

// split-KV combine (defined in mla_decode_fp8.hip)
int fl_mla_launch_combine(const fl_mla::Params& p, const int32_t* num_splits, hipStream_t stream);
