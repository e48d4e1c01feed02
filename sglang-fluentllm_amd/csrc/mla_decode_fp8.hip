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
//   * one workgroup = 2*WH waves, one wave per SIMD, 512 registers per lane.  Wave (wh, wt) owns 32 query rows
//     (wh) and every second 64-token page of the workgroup's page list (wt); the two token-waves of a row group are
//     merged once per request through LDS, so a full request needs no split-KV round trip through HBM.
//   * S^T[64 tok x 32 rows] = K_tile · Q^T on v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3, unit E8M0 scales, the
//     2x-rate MX path) + 32x32x16 bf16 for the 64 rope dims.  Each lane then holds ONE query row (lane&31) and 32
//     tokens: the online-softmax row reductions are in-register plus a single cross-half exchange.
//   * P is re-quantised to fp8 in registers, already in the B-operand layout of the PV MFMA (the contraction order
//     over tokens is free, so V^T is gathered in the same token order): O^T[512 x 32 rows] += V^T · P^T, V^T read
//     from the SAME LDS bytes as K through ds_read_b64_tr_b8 (hardware byte transpose).
//   * k_scale[t] is folded into the exponent: y = s*log2e + log2(k_scale[t]), running max over y, P' = 2^(y-m+8)
//     (fp8 range [2^-9, 256]), l accumulates P'/k_scale[t] in fp32.  No per-tile rescale of V is needed.
//   * HBM -> LDS by global_load_lds (16 B/lane, 1 KiB per wave instruction) into a 4-slot ring of 32 KiB pages,
//     XOR-swizzled on the SOURCE address (16-B chunk c of token T lands at chunk c ^ (T&15)) so that both the
//     K-operand ds_read_b128 and the V^T ds_read_b64_tr_b8 are bank-conflict free.  Tokens beyond the sequence end
//     are sourced from a zero line (never NaN * 0 in the PV MFMA).  Rope (8 KiB/page) and the raw scales go
//     straight to VGPRs one page ahead (they are waited for by the ring barrier's vmcnt(0), never earlier).
//
// Algorithmic bytes per (request, layer call): seq*644 (KV) + s_q*h_q*(644 + 1024) (Q in, O out) + 4*ceil(seq/64).
#include "fl_common.h"

namespace {

constexpr int kPage = FL_MLA_PAGE;            // 64 tokens per page / tile
constexpr int kDN = 512;                      // latent (nope) dims, fp8
constexpr int kDR = 64;                       // rope dims, bf16
constexpr int kSlotBytes = kPage * kDN;       // 32 KiB
constexpr int kRingSlots = 4;
constexpr int kRingBytes = kRingSlots * kSlotBytes;  // 128 KiB
constexpr int kScaleScratchPerWave = 3 * kPage * 4;  // ks, log2 ks, 1/ks
constexpr int kMaxWaves = 4;
constexpr int kLdsBytes = kRingBytes + kMaxWaves * kScaleScratchPerWave;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kPShift = 4.0f;               // P' = 2^(y - m + 4): 16 at the running reference ...
constexpr float kRescaleThr = 4.0f;           // ... which may lag the true max by <= 4 (P' <= 256 < 448, T13 defer-max)
constexpr int kUnitScale = 0x7F7F7F7F;        // E8M0 127 = 2^0

__device__ __attribute__((aligned(16))) const uint32_t g_zero_line[4] = {0, 0, 0, 0};

struct Params {
  int bs, s_q, h_q, rows, causal, num_parts, row_groups;
  float scale_log2e;
  const uint8_t* q_nope;
  const uint16_t* q_rope;
  const float* q_scale;
  const uint8_t* k_nope;
  const uint16_t* k_rope;
  const float* k_scale;
  long long num_pages;
  const int32_t* block_table;
  long long bt_stride;
  const int32_t* seqlens;
  const int32_t* meta;
  const int32_t* num_splits;
  uint16_t* out;
  float* lse;
  float* o_accum;
  float* lse_accum;
};

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

// Read-only inputs are separate `const __restrict__` kernel arguments (not struct members) so that hipcc proves them
// invariant: wave-uniform reads (page ids, lengths, scheduler rows) become s_load (lgkmcnt) instead of vector loads
// whose vmcnt(0) would drain the LDS-DMA queue between every piece.
template <int WH>
__global__ __launch_bounds__(128 * WH, 1) void mla_decode_fp8_kernel(
    const Params p, const int32_t* __restrict__ g_block_table, const int32_t* __restrict__ g_seqlens,
    const int32_t* __restrict__ g_meta, const int32_t* __restrict__ g_num_splits,
    const uint8_t* __restrict__ g_k_nope, const uint16_t* __restrict__ g_k_rope, const float* __restrict__ g_k_scale,
    const uint8_t* __restrict__ g_q_nope, const uint16_t* __restrict__ g_q_rope, const float* __restrict__ g_q_scale) {
  constexpr int NW = 2 * WH;                  // waves per workgroup
  constexpr int kDmaPerWave = 64 / NW;        // 1-KiB pieces per wave per page pair
  __shared__ __attribute__((aligned(16))) uint8_t smem[kLdsBytes];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wh = w % WH;
  const int wt = w / WH;
  const int li = lane & 31;   // MFMA row/col index held by this lane
  const int lh = lane >> 5;   // k-half

  // ---- workgroup -> (part, row group); keep a request's row groups on one XCD (block b runs on XCD b%8) ----
  int part, rg;
  {
    const int id = blockIdx.x;
    if ((p.num_parts & 7) == 0) {
      const int xcd = id & 7, k = id >> 3;
      rg = k % p.row_groups;
      part = (k / p.row_groups) * 8 + xcd;
    } else {
      rg = id % p.row_groups;
      part = id / p.row_groups;
    }
  }
  const int32_t* meta = g_meta + part * FL_MLA_META_W;
  int req = meta[0];
  int tile_b = meta[1];
  const int end_req = meta[2];
  const int end_tile = meta[3];
  int split_idx = meta[4];

  const int row = rg * (32 * WH) + wh * 32 + li;   // query row of this lane
  const bool row_ok = row < p.rows;

  uint8_t* sc_base = smem + kRingBytes + w * kScaleScratchPerWave;
  float* sc_ks = reinterpret_cast<float*>(sc_base);
  float* sc_lks = sc_ks + kPage;
  float* sc_iks = sc_lks + kPage;

  // ---- per-lane LDS offsets (within a slot): every lane-dependent XOR bit lives in 16 base registers, all
  //      per-instruction variation is a compile-time immediate (ds offset field) ----
  // K operand (QK): token T = 32*mb + li, 32 B at d = 64*s + 32*lh -> chunks c = 4s + 2lh + e (e=0,1), stored at
  // chunk c ^ (T&15).  bits of c: e->0, lh->1, s&3->2..3, s>>2->4 (not swizzled: +256 immediate).
  const int kx = li & 15;
  int kb[2][4];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2)
      kb[e][k2] = li * kDN + (((((k2 ^ (kx >> 2)) << 2) | ((2 * lh + e) ^ (kx & 3))) << 4));
  // V^T operand (PV): ds_read_b64_tr_b8 source lane s16 = lane&15 reads 8 B of token T_u = 8192-B-immediate(u) +
  // (4*lh + tok_in8) rows, d chunk cj = (jb>>2)*8 + (jb&3) + 4*gi (rows 16..31 of the M block <-> d + 64), half
  // (s16&1)*8.  T_u & 15 = (4*lh + tok_in8) & 15 for every u.  bits of cj: jb&3->0..1, gi->2, (jb>>2)&1->3,
  // jb>>3->4 (+256 immediate).
  const int s16 = lane & 15;
  const int gi = (lane >> 4) & 1;
  const int tj = (s16 >> 1);
  const int tok_in8 = (tj & 3) + ((tj >> 2) << 3);
  const int vrow = 4 * lh + tok_in8;
  const int vx = vrow & 15;
  int vb[8];
#pragma unroll
  for (int k3 = 0; k3 < 8; ++k3) {
    const int low4 = ((k3 & 3) ^ (vx & 3)) | ((gi ^ ((vx >> 2) & 1)) << 2) | ((((k3 >> 2) & 1) ^ (vx >> 3)) << 3);
    vb[k3] = vrow * kDN + (low4 << 4) + (s16 & 1) * 8;
  }

  for (; req < p.bs; ++req, tile_b = 0, split_idx = 0) {
    if (req > end_req || (req == end_req && end_tile == 0)) break;
    const int L = g_seqlens[req];
    const int nt = L > 0 ? (L + kPage - 1) / kPage : 0;
    int tile_e = req < end_req ? nt : (end_tile < nt ? end_tile : nt);
    if (tile_e < tile_b) tile_e = tile_b;
    const int ntile = tile_e - tile_b;
    const int npairs = (ntile + 1) >> 1;
    const int split_base = g_num_splits[req];
    const int nsp = g_num_splits[req + 1] - split_base;
    const bool is_split = nsp > 1;

    // ---- Q fragments (B operands), once per request ----
    const long long qrow = (long long)req * p.rows + row;
    v8i qn[8];
    v8bf qr[4];
    float qs = 0.f;
    if (row_ok) {
      const uint8_t* qp = g_q_nope + qrow * kDN + lh * 32;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const uint4 a = *reinterpret_cast<const uint4*>(qp + s * 64);
        const uint4 b = *reinterpret_cast<const uint4*>(qp + s * 64 + 16);
        qn[s] = make_v8i(a, b);
      }
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
    // causal limit of this row: query j sees keys [0, L - (s_q-1-j))
    int L_row = L;
    if (p.causal) L_row = L - (p.s_q - 1 - row / p.h_q);
    if (!row_ok) L_row = 0;
    const int L_min = p.causal ? L - (p.s_q - 1) : L;   // smallest limit of any row (wave-uniform)

    v16f o[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
    float m_run = -1e30f;
    float l_run = 0.f;
    bool first_tile = true;

    // ---- DMA of one page pair into ring slots (pair&1)*2 + {0,1}: wave w moves kDmaPerWave 1-KiB pieces of ONE
    //      page (all 64 token rows are read, the page is always fully allocated; rows past the sequence end are
    //      zero-filled in LDS by the consumer) ----
    const int dma_tp = (w * kDmaPerWave) >> 5;              // which page of the pair this wave fills
    const int dma_row0 = ((w * kDmaPerWave) & 31) * 2 + lh;  // first token row this lane fills
    const int dma_x = (lane & 31) << 4;
    auto issue_pair = [&](int pair) {
      const int tt = tile_b + 2 * pair + dma_tp;
      if (tt < tile_e) {
        int page = g_block_table[(long long)req * p.bt_stride + tt];
        if (page < 0 || page >= p.num_pages) page = 0;
        const uint8_t* pbase = g_k_nope + (long long)page * (kPage * kDN);
        uint8_t* dst = smem + (((pair & 1) * 2 + dma_tp) * kSlotBytes) + ((w * kDmaPerWave) & 31) * 1024;
#pragma unroll
        for (int k = 0; k < kDmaPerWave; ++k) {
          // token row T = dma_row0 + 2k; LDS chunk position lane&31 holds source chunk (lane&31) ^ (T&15)
          const int T = dma_row0 + 2 * k;
          const int off = T * kDN + (dma_x ^ ((T & 15) << 4));
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(pbase + off), (lds_ptr_t)(dst + k * 1024), 16, 0, 0);
        }
      }
    };

    // rope + raw scale of one page straight to registers
    uint4 rope[2][4];
    float ks_raw = 1.f;
    auto load_rope_scale = [&](int tt) {
      int page = g_block_table[(long long)req * p.bt_stride + tt];
      if (page < 0 || page >= p.num_pages) page = 0;
      const uint16_t* rp = g_k_rope + ((long long)page * kPage + li) * kDR + lh * 8;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int s = 0; s < 4; ++s) rope[mb][s] = *reinterpret_cast<const uint4*>(rp + mb * 32 * kDR + s * 16);
      ks_raw = g_k_scale[(long long)page * kPage + lane];
    };

    __syncthreads();   // previous request finished with the ring / merge area
    if (npairs > 0) {
      issue_pair(0);
      if (tile_b + wt < tile_e) load_rope_scale(tile_b + wt);
    }

    for (int it = 0; it < npairs; ++it) {
      __syncthreads();   // vmcnt(0): pair `it` landed (and rope/scale regs); slots of pair it+1 are free
      if (it + 1 < npairs) issue_pair(it + 1);
      const int tt = tile_b + 2 * it + wt;
      if (tt < tile_e) {
        uint8_t* slot = smem + (((it & 1) * 2 + wt) * kSlotBytes);
        const int tok0 = tt * kPage;
        if (tok0 + kPage > L) {
          // last page of the sequence: rows past the end hold whatever the page held (possibly fp8 NaN patterns);
          // P' is exactly 0 there but 0*NaN would poison the PV MFMA, so every consumer wave zeroes them itself.
          const int nvalid = L - tok0;
          for (int T = nvalid + (lane >> 5); T < kPage; T += 2)
            *reinterpret_cast<uint4*>(slot + T * kDN + (lane & 31) * 16) = make_uint4(0, 0, 0, 0);
        }
        // ---- per-token scale preprocessing (lane t handles token t), wave-private scratch ----
        {
          float ks = ks_raw;
          if (tok0 + lane >= L || !(ks > 0.f) || !(ks < 3.0e38f)) ks = 1.f;
          sc_ks[lane] = ks;
          sc_lks[lane] = __builtin_amdgcn_logf(ks);
          sc_iks[lane] = __builtin_amdgcn_rcpf(ks);
        }
        // ---- S^T = K · Q^T ----
        v16f acc[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(rope[mb][s]), qr[s], acc[mb], 0, 0, 0);
        // next page's rope/scale (same registers; waited for only at the next ring barrier)
        if (tt + 2 < tile_e) load_rope_scale(tt + 2);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) {
            const int imm = mb * (32 * kDN) + (s >> 2) * 256;
            const uint4 a0 = *reinterpret_cast<const uint4*>(slot + kb[0][s & 3] + imm);
            const uint4 a1 = *reinterpret_cast<const uint4*>(slot + kb[1][s & 3] + imm);
            acc[mb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(make_v8i(a0, a1), qn[s], acc[mb], 0, 0, 0,
                                                                      kUnitScale, 0, kUnitScale);
          }
        }
        // ---- online softmax on y = s*log2e + log2(k_scale[t]) ----
        const bool need_mask = (tok0 + kPage > L_min);
        float tmax = -INFINITY;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int tb = mb * 32 + g * 8 + lh * 4;   // first of 4 consecutive tokens
            const float4 ks4 = *reinterpret_cast<const float4*>(sc_ks + tb);
            const float4 lk4 = *reinterpret_cast<const float4*>(sc_lks + tb);
            const float ksv[4] = {ks4.x, ks4.y, ks4.z, ks4.w};
            const float lkv[4] = {lk4.x, lk4.y, lk4.z, lk4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float y = fmaf(acc[mb][g * 4 + e] * qs, ksv[e], lkv[e]);
              if (need_mask && (tok0 + tb + e >= L_row)) y = -INFINITY;
              if (!(y == y)) y = -INFINITY;   // NaN from garbage beyond the row's limit can only be masked data
              acc[mb][g * 4 + e] = y;
              tmax = fmaxf(tmax, y);
            }
          }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        // defer-max (T13): keep the old reference unless some row's max grew by more than kRescaleThr; the PV of the
        // previous page is complete, so O, l and the reference move together exactly once.
        if (__any(tmax > m_run + kRescaleThr)) {
          const float m_new = fmaxf(m_run, tmax);
          if (!first_tile) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) o[j][r] *= alpha;
          }
          m_run = m_new;
        }
        first_tile = false;
        const float moff = kPShift - m_run;
        v8i pb;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int tb = mb * 32 + g * 8 + lh * 4;
            const float4 ik4 = *reinterpret_cast<const float4*>(sc_iks + tb);
            const float e0 = __builtin_amdgcn_exp2f(acc[mb][g * 4 + 0] + moff);
            const float e1 = __builtin_amdgcn_exp2f(acc[mb][g * 4 + 1] + moff);
            const float e2 = __builtin_amdgcn_exp2f(acc[mb][g * 4 + 2] + moff);
            const float e3 = __builtin_amdgcn_exp2f(acc[mb][g * 4 + 3] + moff);
            l_run = fmaf(e0, ik4.x, l_run);
            l_run = fmaf(e1, ik4.y, l_run);
            l_run = fmaf(e2, ik4.z, l_run);
            l_run = fmaf(e3, ik4.w, l_run);
            int pk = __builtin_amdgcn_cvt_pk_fp8_f32(e0, e1, 0, false);
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(e2, e3, pk, true);
            pb[mb * 4 + g] = pk;
          }
        }
        // ---- O^T += V^T · P^T ----
#pragma unroll
        for (int jb = 0; jb < 16; ++jb) {
          v8i a;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint8_t* ap = slot + vb[(jb & 3) | (((jb >> 2) & 1) << 2)] + u * (16 * kDN) + (jb >> 3) * 256;
            const v2i t2 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)ap);
            a[2 * u] = t2[0];
            a[2 * u + 1] = t2[1];
          }
          o[jb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, pb, o[jb], 0, 0, 0, kUnitScale, 0, kUnitScale);
        }
      }
    }

    // ---- merge the two token-waves of each row group through LDS (aliases the ring) ----
    l_run += __shfl_xor(l_run, 32);
    __syncthreads();   // every wave is done reading ring slots
    float* mg = reinterpret_cast<float*>(smem) + wh * (16 * 16 * 64);   // [jb][reg][lane] fp32, 64 KiB per row group
    float* mg_ml = reinterpret_cast<float*>(smem + kRingBytes);        // scale scratch reused: [wh][2][64]
    if (wt == 1) {
#pragma unroll
      for (int jb = 0; jb < 16; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mg[(jb * 16 + r) * 64 + lane] = o[jb][r];
      mg_ml[(wh * 2 + 0) * 64 + lane] = m_run;
      mg_ml[(wh * 2 + 1) * 64 + lane] = l_run;
    }
    __syncthreads();
    if (wt == 0) {
      const float m1 = mg_ml[(wh * 2 + 0) * 64 + lane];
      const float l1 = mg_ml[(wh * 2 + 1) * 64 + lane];
      const float m = fmaxf(m_run, m1);
      const float a0 = __builtin_amdgcn_exp2f(m_run - m);
      const float a1 = __builtin_amdgcn_exp2f(m1 - m);
      const float l = l_run * a0 + l1 * a1;
      const float inv = l > 0.f ? 1.f / l : 0.f;
      const float w0 = a0 * inv, w1 = a1 * inv;
      // natural-log LSE of the rows: log2(sum 2^x) = log2(l) + m - kPShift
      const float lse_nat = l > 0.f ? (__builtin_amdgcn_logf(l) + m - kPShift) * 0.6931471805599453f : -INFINITY;
      if (row_ok) {
        const int slot_idx = split_base + split_idx;
        if (lh == 0) {
          if (is_split) {
            p.lse_accum[(long long)slot_idx * p.rows + row] = lse_nat;
          } else {
            const int j = row / p.h_q, h = row - j * p.h_q;
            p.lse[((long long)req * p.h_q + h) * p.s_q + j] = lse_nat;
          }
        }
        if (is_split) {
          float* dbase = p.o_accum + ((long long)slot_idx * p.rows + row) * kDN;
#pragma unroll
          for (int jb = 0; jb < 16; ++jb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int r = g * 4 + e;
                v[e] = o[jb][r] * w0 + mg[(jb * 16 + r) * 64 + lane] * w1;
              }
              // C row i = e + 8g + 4*lh  ->  d = (jb>>2)*128 + (jb&3)*16 + (i&15) + 64*(i>>4)
              const int i0 = 8 * g + 4 * lh;
              const int d0 = (jb >> 2) * 128 + (jb & 3) * 16 + (i0 & 15) + 64 * (i0 >> 4);
              *reinterpret_cast<float4*>(dbase + d0) = make_float4(v[0], v[1], v[2], v[3]);
            }
          }
        } else {
          uint16_t* dbase = p.out + qrow * kDN;
#pragma unroll
          for (int jb = 0; jb < 16; ++jb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int r = g * 4 + e;
                v[e] = o[jb][r] * w0 + mg[(jb * 16 + r) * 64 + lane] * w1;
              }
              const int i0 = 8 * g + 4 * lh;
              const int d0 = (jb >> 2) * 128 + (jb & 3) * 16 + (i0 & 15) + 64 * (i0 >> 4);
              const uint32_t lo = (uint32_t)fl_f32_to_bf16(v[0]) | ((uint32_t)fl_f32_to_bf16(v[1]) << 16);
              const uint32_t hi = (uint32_t)fl_f32_to_bf16(v[2]) | ((uint32_t)fl_f32_to_bf16(v[3]) << 16);
              *reinterpret_cast<uint2*>(dbase + d0) = make_uint2(lo, hi);
            }
          }
        }
      }
    }
  }
}

// ---- split-KV combine: out[req,row,:] = sum_s w_s * o_accum[slot_s,row,:], w_s = softmax_s(lse_s) ----
__global__ __launch_bounds__(256) void mla_combine_kernel(const Params p) {
  const int lane = threadIdx.x & 63;
  const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gw >= (long long)p.bs * p.rows) return;
  const int req = (int)(gw / p.rows), row = (int)(gw % p.rows);
  const int s0 = p.num_splits[req], ns = p.num_splits[req + 1] - s0;
  if (ns <= 1) return;
  float mx = -INFINITY;
  for (int s = 0; s < ns; ++s) mx = fmaxf(mx, p.lse_accum[(long long)(s0 + s) * p.rows + row]);
  float den = 0.f;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int s = 0; s < ns; ++s) {
    const float ls = p.lse_accum[(long long)(s0 + s) * p.rows + row];
    const float wgt = (mx == -INFINITY) ? 0.f : __expf(ls - mx);
    den += wgt;
    const float* src = p.o_accum + ((long long)(s0 + s) * p.rows + row) * kDN + lane * 8;
    const float4 a = *reinterpret_cast<const float4*>(src);
    const float4 b = *reinterpret_cast<const float4*>(src + 4);
    acc[0] += wgt * a.x; acc[1] += wgt * a.y; acc[2] += wgt * a.z; acc[3] += wgt * a.w;
    acc[4] += wgt * b.x; acc[5] += wgt * b.y; acc[6] += wgt * b.z; acc[7] += wgt * b.w;
  }
  const float inv = den > 0.f ? 1.f / den : 0.f;
  uint32_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    o[i] = (uint32_t)fl_f32_to_bf16(acc[2 * i] * inv) | ((uint32_t)fl_f32_to_bf16(acc[2 * i + 1] * inv) << 16);
  *reinterpret_cast<uint4*>(p.out + ((long long)req * p.rows + row) * kDN + lane * 8) = make_uint4(o[0], o[1], o[2], o[3]);
  if (lane == 0) {
    const int j = row / p.h_q, h = row - j * p.h_q;
    p.lse[((long long)req * p.h_q + h) * p.s_q + j] = den > 0.f ? mx + __logf(den) : -INFINITY;
  }
}

}  // namespace

int fl_mla_decode_fp8_per_token_impl(const FlMlaDecodeArgs* a, hipStream_t stream) {
  FL_CHECK_ARG(a->d_nope == kDN && a->d_rope == kDR, "fl_mla_decode: only d_nope=512,d_rope=64 (got %d,%d)",
               a->d_nope, a->d_rope);
  FL_CHECK_ARG(a->q_nope && a->q_rope && a->q_scale && a->k_nope && a->k_rope && a->k_scale,
               "fl_mla_decode(per-token fp8): null q/k pointer");
  FL_CHECK_ARG(a->block_table && a->cache_seqlens && a->tile_scheduler_metadata && a->num_splits && a->out && a->lse &&
                   a->o_accum && a->lse_accum,
               "fl_mla_decode: null metadata/output pointer");
  FL_CHECK_ARG(a->bs >= 0 && a->s_q >= 1 && a->h_q >= 1 && a->num_parts >= 1, "fl_mla_decode: bad sizes");
  if (a->bs == 0) return FL_OK;
  Params p;
  p.bs = a->bs; p.s_q = a->s_q; p.h_q = a->h_q; p.rows = a->s_q * a->h_q; p.causal = a->causal;
  p.num_parts = a->num_parts;
  p.scale_log2e = a->softmax_scale * kLog2e;
  p.q_nope = (const uint8_t*)a->q_nope; p.q_rope = (const uint16_t*)a->q_rope; p.q_scale = a->q_scale;
  p.k_nope = (const uint8_t*)a->k_nope; p.k_rope = (const uint16_t*)a->k_rope; p.k_scale = a->k_scale;
  p.num_pages = a->num_pages; p.block_table = a->block_table; p.bt_stride = a->block_table_stride;
  p.seqlens = a->cache_seqlens; p.meta = a->tile_scheduler_metadata; p.num_splits = a->num_splits;
  p.out = (uint16_t*)a->out; p.lse = a->lse; p.o_accum = a->o_accum; p.lse_accum = a->lse_accum;
  // One shape for now: 2 row waves x 2 token waves (rows <= 32 leave the second row wave idle; the small-M
  // "swap" kernel that splits tokens/d across waves instead is the next step).
  const int wh = 2;
  p.row_groups = (p.rows + 32 * wh - 1) / (32 * wh);
  const unsigned grid = (unsigned)(p.num_parts * p.row_groups);
  mla_decode_fp8_kernel<2><<<dim3(grid), dim3(256), 0, stream>>>(p, p.block_table, p.seqlens, p.meta, p.num_splits,
                                                                  p.k_nope, p.k_rope, p.k_scale, p.q_nope, p.q_rope,
                                                                  p.q_scale);
  FL_CHECK_LAUNCH("mla_decode_fp8_kernel");
  const long long waves = (long long)p.bs * p.rows;
  mla_combine_kernel<<<dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream>>>(p);
  FL_CHECK_LAUNCH("mla_combine_kernel");
  return FL_OK;
}
