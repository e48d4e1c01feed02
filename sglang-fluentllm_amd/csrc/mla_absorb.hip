// A2 — the query side of the MLA decode layer BEFORE the attention kernel, in ONE launch (VERDICT r2 "what's missing" 1 and 5):
//   q_nope @ w_kc           torch.bmm(q_nope.transpose(0,1), w_kc, out=Q[..., :512].transpose(0,1))      (srt/models/deepseek_v2.py:840)
//   RoPE of q_pe / k_pe     self.rotary_emb(positions, q_pe, K[..., 512:], output_q_rope=Q[..., 512:])   (:842-858; rotary_embedding.py:203-218)
//   K5 set_kv_buffer        quantize_and_cache_k(K)                  (flashmla_backend.py:188-196 -> mem_cache/memory_pool.py:864-871)
//   K4 quantize q           quantize_ckv_per_token_head(Q, 512)      (flashmla_backend.py:198-206)
// i.e. the bf16 absorbed query Q [T, H, 576] never exists in memory: the kernel's outputs are K4's (fp8 latent part, per-(token, head)
// scale, rope part divided by the scale) and K5's (the cache rows of the new tokens).  The unfused chain is four launches (B2, R2,
// K5 + K4) that write and re-read Q: T = 128, H = 128: 16.8 MB written + 18.9 MB read back.
// Arithmetic = the chain's, operation for operation (bit-identical outputs, tests/test_absorb_gpu.py):
//   * the product is the B2 kernel's (bmm_bf16.hip: the head's [512 x 128] weight matrix staged once in LDS, token fragments in
//     registers, the same k order), rounded to bf16 like torch.bmm's result;
//   * the rotation is R2's (rope.hip: fp32, each product and the sum rounded separately, one rounding to bf16);
//   * the quantisation is K4 / K5's (mla_quant.hip: scale = max(amax, 1e-26) / 448 over the 512 latent values, fl_div8_to_fp8).
// The first heads x ceil(T / M_WG) workgroups do the query rows of one head (M_WG = 128, 64 or 32 token rows: 1, 2 or 4 waves share a
// 32-row tile's 16 n tiles, see the template parameter); the rest quantise the new K rows, one wave per token (k_pe is rotated in
// place in the latent row, as the reference's call leaves it).
// Measured (MI355X, tools/time_absorb.py, hipGraph of 8 calls, H = 128; four launches vs this one): T = 256 33.0 vs 26.8 us, T = 128
// 23.1 vs 20.0, T = 64 18.6 vs 16.5, T = 16 10.7 vs 11.2, T = 1 8.9 vs 13.0; H = 16 (a TP8 shard): T = 256 13.2 vs 11.8.  The launch
// moves 2.5 x fewer bytes than the chain (T = 128: 17 vs 71 MB of HBM traffic) but each workgroup runs its phases one behind the
// other — stage 128 KiB of weights, 16 MFMA tiles per row tile, then ~10 VALU instructions per output element for K4's exact
// division — with one workgroup per CU (the weight matrix fills the LDS): it pays from about 64 tokens up; below that the
// separate launches (which slice the weight matrix over four workgroups per head) are faster.
#include "fl_common.h"
#pragma clang fp contract(off)   // the rotation's a*c + b*s as three roundings, like rope.hip / the torch statement

namespace {

constexpr int kDNope = 128, kDRope = 64, kDLora = 512;
constexpr int kLdsBytes = kDLora * kDNope * 2;   // 128 KiB: w_kc of one head
constexpr int RB = kDNope * 2;                   // bytes per weight row

struct AbsorbParams {
  const uint16_t* q;          // [T, H, 192] bf16: nope 128 | rope 64
  long long q_st, q_sh;       // element strides
  const uint16_t* w_kc;       // [H, 512, 128] bf16, k-contiguous
  long long w_sh;
  const long long* positions;
  const float* cache;         // [max_position, 64] f32: cos 32 | sin 32
  long long max_position;
  int is_neox;
  int T, H;
  uint16_t* latent;           // [T, 576] bf16 (k_nope | k_pe): k_pe rotated in place; may be null (no K rows)
  long long latent_stride;
  const int32_t* cache_loc;
  uint8_t* k_nope_cache;
  float* k_scale_cache;
  uint16_t* k_rope_cache;
  long long num_slots;
  uint8_t* q_nope_out;        // [T, H, 512] e4m3
  float* q_scale_out;         // [T, H]
  uint16_t* q_rope_out;       // [T, H, 64] bf16
  int q_blocks;
};

__device__ __forceinline__ v8bf as_v8bf(const uint4 u) {
  union { uint4 u; v8bf v; } x;
  x.u = u;
  return x.v;
}
__device__ __forceinline__ float lo_f(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_f(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float rot_a(float a, float b, float c, float s) { return a * c + (-b) * s; }
__device__ __forceinline__ float rot_b(float a, float b, float c, float s) { return b * c + a * s; }

// WN = waves that share one 32-row token tile (each takes 16 / WN of the 16 n tiles; the row maximum goes through LDS): a workgroup
// covers 128 / WN token rows — fewer rows per workgroup = more workgroups and less serial work per wave when there are few tokens
template <int WN>
__global__ __launch_bounds__(256) void mla_absorb_kernel(const AbsorbParams p) {
  constexpr int NT = 16 / WN;          // n tiles per wave
  constexpr int M_WG = 128 / WN;       // token rows per workgroup
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  if ((int)blockIdx.x >= p.q_blocks) {
    // ======================= new K rows: rotate k_pe, K5 =======================
    const long long t = ((long long)blockIdx.x - p.q_blocks) * 4 + wave;
    if (t >= p.T) return;
    uint16_t* row = p.latent + t * p.latent_stride;
    const uint4 raw = *reinterpret_cast<const uint4*>(row + lane * 8);
    const float x = fl_bf16_to_f32(row[kDLora + lane]);
    const int32_t loc = p.cache_loc[t];
    long long pos = p.positions[t];
    pos = pos < 0 ? 0 : (pos >= p.max_position ? p.max_position - 1 : pos);
    // rotation pair of element `lane`: GPT-J (lane ^ 1, angle lane / 2), NeoX (lane ^ 32, angle lane & 31)
    const float y = __shfl_xor(x, p.is_neox ? 32 : 1);
    const int ang = p.is_neox ? (lane & 31) : (lane >> 1);
    const bool first = p.is_neox ? lane < 32 : (lane & 1) == 0;
    const float c = p.cache[pos * kDRope + ang], s = p.cache[pos * kDRope + kDRope / 2 + ang];
    const uint16_t rb = fl_f32_to_bf16(first ? rot_a(x, y, c, s) : rot_b(y, x, c, s));
    row[kDLora + lane] = rb;   // in place, like the reference's call
    float v[8];
    v[0] = lo_f(raw.x); v[1] = hi_f(raw.x); v[2] = lo_f(raw.y); v[3] = hi_f(raw.y);
    v[4] = lo_f(raw.z); v[5] = hi_f(raw.z); v[6] = lo_f(raw.w); v[7] = hi_f(raw.w);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(v[i]));
    amax = fl_wave_max(amax);
    const float scale = fmaxf(amax, 1e-26f) / FL_FP8_MAX;
    if (loc < 0 || loc >= p.num_slots) return;   // never write out of the pool
    *reinterpret_cast<uint2*>(p.k_nope_cache + (long long)loc * kDLora + lane * 8) = fl_div8_to_fp8<false>(v, scale);
    p.k_rope_cache[(long long)loc * kDRope + lane] = fl_f32_to_bf16(fl_bf16_to_f32(rb) / scale);
    if (lane == 0) p.k_scale_cache[loc] = scale;
    return;
  }

  // ======================= query rows: absorb, rotate, K4 =======================
  const int li = lane & 31, kq = lane >> 5;
  const int m_splits = (p.T + M_WG - 1) / M_WG;
  const int h = blockIdx.x / m_splits;
  const int m_base = (blockIdx.x % m_splits) * M_WG;
  const int wm = wave / WN, wn = wave % WN;
  float* xch = reinterpret_cast<float*>(smem + kLdsBytes);   // [4 waves][32 rows]: per-wave partial row maxima
  {   // stage w_kc[h] ([512, 128] k-contiguous): 128 pieces of 1 KiB; 16-B chunk c of row n stored at chunk c ^ (n & 15)
    const uint8_t* gb = reinterpret_cast<const uint8_t*>(p.w_kc + (long long)h * p.w_sh);
    for (int P = wave; P < kLdsBytes / 1024; P += 4) {
      const int off = P * 1024 + lane * 16;
      const int n = off / RB;
      const int pc = (off % RB) >> 4;
      fl_dma16(gb + (long long)n * RB + ((pc ^ (n & 15)) << 4), smem + P * 1024);
    }
  }
  const int m0 = m_base + wm * 32;
  const int m = m0 + li;
  const int mc = m < p.T ? m : p.T - 1;   // (clamped loads; the tail rows are not stored)
  const uint16_t* qrow = p.q + (long long)mc * p.q_st + (long long)h * p.q_sh;
  // token fragments in B2's k order: MFMA step 4 a + t takes k = 64 a + 32 kq + 8 t
  v8bf fb[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) fb[s] = as_v8bf(*reinterpret_cast<const uint4*>(qrow + 32 * kq + 64 * (s >> 2) + 8 * (s & 3)));
  // q_pe: this lane's 16 rotation pairs (GPT-J: elements 32 kq .. 32 kq + 31; NeoX: 16 kq .. + 15 and 32 + 16 kq .. + 15)
  uint4 pe[4];
  {
    const uint16_t* pr = qrow + kDNope;
    if (p.is_neox) {
      pe[0] = *reinterpret_cast<const uint4*>(pr + 16 * kq);
      pe[1] = *reinterpret_cast<const uint4*>(pr + 16 * kq + 8);
      pe[2] = *reinterpret_cast<const uint4*>(pr + 32 + 16 * kq);
      pe[3] = *reinterpret_cast<const uint4*>(pr + 32 + 16 * kq + 8);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) pe[i] = *reinterpret_cast<const uint4*>(pr + 32 * kq + 8 * i);
    }
  }
  long long pos = p.positions[mc];
  pos = pos < 0 ? 0 : (pos >= p.max_position ? p.max_position - 1 : pos);
  float4 cs[4], sn[4];   // angles 16 kq .. 16 kq + 15
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    cs[i] = *reinterpret_cast<const float4*>(p.cache + pos * kDRope + 16 * kq + 4 * i);
    sn[i] = *reinterpret_cast<const float4*>(p.cache + pos * kDRope + kDRope / 2 + 16 * kq + 4 * i);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA pieces (and the row's loads) have landed
  __syncthreads();
  const bool active = m0 < p.T;

  // ---- Y^T[n, m] = w_kc[h] . q_nope^T, 16 n tiles of 32; kept as bf16 (torch.bmm's rounding): after the swap of accumulator
  //      groups between the lane halves a lane holds n = 32 nt + 16 kq + (0..15) of its token row: yb[nt][0..7] ----
  auto load8 = [&](v8bf (&fa)[8], int nt) {   // nt: global n tile (this wave's are wn * NT .. wn * NT + NT - 1)
    nt = nt < 16 ? nt : 15;
    const int n = nt * 32 + li;
    const uint8_t* wrow = smem + n * RB;
#pragma unroll
    for (int s = 0; s < 8; ++s) fa[s] = as_v8bf(*reinterpret_cast<const uint4*>(wrow + (((8 * (s >> 2) + 4 * kq + (s & 3)) ^ (n & 15)) << 4)));
  };
  uint32_t yb[NT][8];
  float amax = 0.f;
  auto finish = [&](const v16f& acc, const int nt) {   // nt: index among this wave's tiles
    uint32_t own[8], oth[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[r]), __float_as_uint(acc[r + 8]), false, false);
      own[r] = kq == 0 ? sw[0] : sw[1];
      oth[r] = kq == 0 ? sw[1] : sw[0];
    }
    const uint32_t* lo = kq == 0 ? own : oth;
    const uint32_t* hi = kq == 0 ? oth : own;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      yb[nt][4 * g + 0] = fl_pack_bf16(__uint_as_float(lo[4 * g]), __uint_as_float(lo[4 * g + 1]));
      yb[nt][4 * g + 1] = fl_pack_bf16(__uint_as_float(lo[4 * g + 2]), __uint_as_float(lo[4 * g + 3]));
      yb[nt][4 * g + 2] = fl_pack_bf16(__uint_as_float(hi[4 * g]), __uint_as_float(hi[4 * g + 1]));
      yb[nt][4 * g + 3] = fl_pack_bf16(__uint_as_float(hi[4 * g + 2]), __uint_as_float(hi[4 * g + 3]));
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) amax = fmaxf(amax, fmaxf(fabsf(lo_f(yb[nt][r])), fabsf(hi_f(yb[nt][r]))));
  };
  if (active) {
    v8bf fa0[8], fa1[8];
    load8(fa0, wn * NT);
#pragma unroll
    for (int nt = 0; nt < NT; nt += 2) {
      v16f acc, acc2;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
      load8(fa1, wn * NT + nt + 1);
#pragma unroll
      for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[s], fb[s], acc, 0, 0, 0);
      load8(fa0, wn * NT + nt + 2);
#pragma unroll
      for (int s = 0; s < 8; ++s) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[s], fb[s], acc2, 0, 0, 0);
      finish(acc, nt);
      finish(acc2, nt + 1);
    }
  }
  {   // the row's maximum: this lane's values and the other half's, then the other waves of the token tile (through LDS)
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(amax), __float_as_uint(amax), false, false);
    amax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  }
  if constexpr (WN > 1) {
    if (kq == 0) xch[wave * 32 + li] = amax;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < WN; ++j) amax = fmaxf(amax, xch[(wm * WN + j) * 32 + li]);
  }
  if (!active) return;
  const float scale = fmaxf(amax, 1e-26f) / FL_FP8_MAX;
  if (m >= p.T) return;
  const long long orow = (long long)m * p.H + h;
  if (kq == 0 && wn == 0) p.q_scale_out[orow] = scale;
  // ---- K4, latent part: 16 fp8 bytes per n tile ----
  uint8_t* qn = p.q_nope_out + orow * kDLora + 32 * (wn * NT) + 16 * kq;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    float v0[8], v1[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v0[2 * r] = lo_f(yb[nt][r]);
      v0[2 * r + 1] = hi_f(yb[nt][r]);
      v1[2 * r] = lo_f(yb[nt][4 + r]);
      v1[2 * r + 1] = hi_f(yb[nt][4 + r]);
    }
    const uint2 w0 = fl_div8_to_fp8<false>(v0, scale), w1 = fl_div8_to_fp8<false>(v1, scale);
    *reinterpret_cast<uint4*>(qn + 32 * nt) = make_uint4(w0.x, w0.y, w1.x, w1.y);
  }
  // ---- rope part (the token tile's last wave): rotate (R2's arithmetic, one rounding to bf16 = the value Q[..., 512:] would hold),
  //      divide by the scale ----
  if (wn != WN - 1) return;
  const float csf[16] = {cs[0].x, cs[0].y, cs[0].z, cs[0].w, cs[1].x, cs[1].y, cs[1].z, cs[1].w,
                         cs[2].x, cs[2].y, cs[2].z, cs[2].w, cs[3].x, cs[3].y, cs[3].z, cs[3].w};
  const float snf[16] = {sn[0].x, sn[0].y, sn[0].z, sn[0].w, sn[1].x, sn[1].y, sn[1].z, sn[1].w,
                         sn[2].x, sn[2].y, sn[2].z, sn[2].w, sn[3].x, sn[3].y, sn[3].z, sn[3].w};
  const uint32_t pw[16] = {pe[0].x, pe[0].y, pe[0].z, pe[0].w, pe[1].x, pe[1].y, pe[1].z, pe[1].w,
                           pe[2].x, pe[2].y, pe[2].z, pe[2].w, pe[3].x, pe[3].y, pe[3].z, pe[3].w};
  uint16_t* qr = p.q_rope_out + orow * kDRope;
  auto quant = [&](const float rotated) {   // bf16 round (the rotation's output), then K4's division, bf16 again
    return fl_f32_to_bf16(fl_bf16_to_f32(fl_f32_to_bf16(rotated)) / scale);
  };
  if (p.is_neox) {
    // pair i = 16 kq + j: a = element i (words pw[0..7]), b = element 32 + i (words pw[8..15])
    uint32_t oa[8], ob[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const float a0 = lo_f(pw[w]), a1 = hi_f(pw[w]), b0 = lo_f(pw[8 + w]), b1 = hi_f(pw[8 + w]);
      oa[w] = (uint32_t)quant(rot_a(a0, b0, csf[2 * w], snf[2 * w])) | ((uint32_t)quant(rot_a(a1, b1, csf[2 * w + 1], snf[2 * w + 1])) << 16);
      ob[w] = (uint32_t)quant(rot_b(a0, b0, csf[2 * w], snf[2 * w])) | ((uint32_t)quant(rot_b(a1, b1, csf[2 * w + 1], snf[2 * w + 1])) << 16);
    }
    *reinterpret_cast<uint4*>(qr + 16 * kq) = make_uint4(oa[0], oa[1], oa[2], oa[3]);
    *reinterpret_cast<uint4*>(qr + 16 * kq + 8) = make_uint4(oa[4], oa[5], oa[6], oa[7]);
    *reinterpret_cast<uint4*>(qr + 32 + 16 * kq) = make_uint4(ob[0], ob[1], ob[2], ob[3]);
    *reinterpret_cast<uint4*>(qr + 32 + 16 * kq + 8) = make_uint4(ob[4], ob[5], ob[6], ob[7]);
  } else {
    // pair i = 16 kq + j = elements (2 i, 2 i + 1) = word j of this lane's 16
    uint32_t o[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float a = lo_f(pw[j]), b = hi_f(pw[j]);
      o[j] = (uint32_t)quant(rot_a(a, b, csf[j], snf[j])) | ((uint32_t)quant(rot_b(a, b, csf[j], snf[j])) << 16);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(qr + 32 * kq + 8 * i) = make_uint4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
  }
}

}  // namespace

extern "C" int fl_mla_absorb_rope_quant(const FlMlaAbsorbArgs* a, fl_stream_t stream) {
  FL_CHECK_ARG(a != nullptr, "fl_mla_absorb_rope_quant: null arguments");
  FL_CHECK_ARG(a->num_tokens >= 0 && a->num_tokens < (1ll << 24) && a->num_heads >= 1, "fl_mla_absorb_rope_quant: bad sizes");
  if (a->num_tokens == 0) return FL_OK;
  FL_CHECK_ARG(a->q && a->w_kc && a->positions && a->cos_sin_cache && a->q_nope_out && a->q_scale_out && a->q_rope_out,
               "fl_mla_absorb_rope_quant: null pointer");
  FL_CHECK_ARG(a->d_nope == kDNope && a->d_rope == kDRope && a->d_lora == kDLora,
               "fl_mla_absorb_rope_quant: only qk_nope 128 / rope 64 / kv_lora 512 (got %d / %d / %d)", a->d_nope, a->d_rope, a->d_lora);
  FL_CHECK_ARG(a->q_stride_token % 8 == 0 && a->q_stride_head % 8 == 0 && ((uintptr_t)a->q % 16) == 0 && ((uintptr_t)a->w_kc % 16) == 0 &&
                   a->w_stride_head % 8 == 0 && ((uintptr_t)a->cos_sin_cache % 16) == 0 && ((uintptr_t)a->q_nope_out % 16) == 0 &&
                   ((uintptr_t)a->q_rope_out % 16) == 0,
               "fl_mla_absorb_rope_quant: rows must be 16-byte aligned");
  FL_CHECK_ARG(a->max_position > 0, "fl_mla_absorb_rope_quant: empty cos / sin cache");
  const bool has_k = a->latent != nullptr;
  if (has_k)
    FL_CHECK_ARG(a->cache_loc && a->k_lora_cache && a->k_scale_cache && a->k_rope_cache && a->num_slots > 0 &&
                     a->latent_stride % 8 == 0 && ((uintptr_t)a->latent % 16) == 0,
                 "fl_mla_absorb_rope_quant: the K rows need cache_loc and the three cache tensors (16-byte aligned rows)");
  AbsorbParams p;
  p.q = (const uint16_t*)a->q; p.q_st = a->q_stride_token; p.q_sh = a->q_stride_head;
  p.w_kc = (const uint16_t*)a->w_kc; p.w_sh = a->w_stride_head;
  p.positions = (const long long*)a->positions; p.cache = a->cos_sin_cache; p.max_position = a->max_position; p.is_neox = a->is_neox;
  p.T = (int)a->num_tokens; p.H = a->num_heads;
  p.latent = (uint16_t*)a->latent; p.latent_stride = a->latent_stride; p.cache_loc = a->cache_loc;
  p.k_nope_cache = (uint8_t*)a->k_lora_cache; p.k_scale_cache = a->k_scale_cache; p.k_rope_cache = (uint16_t*)a->k_rope_cache;
  p.num_slots = a->num_slots;
  p.q_nope_out = (uint8_t*)a->q_nope_out; p.q_scale_out = a->q_scale_out; p.q_rope_out = (uint16_t*)a->q_rope_out;
  // token rows per workgroup: 128 (a wave per 32-row tile), 64 or 32 (2 / 4 waves share a tile's n range) — the smallest that still
  // gives every CU a workgroup, and never more rows than there are
  int wn = 1;
  while (wn < 4 && ((long long)a->num_heads * ((a->num_tokens + 128 / wn - 1) / (128 / wn)) < 256 || a->num_tokens <= 128 / (2 * wn))) wn *= 2;
  const int m_wg = 128 / wn;
  const long long qb = (long long)a->num_heads * ((a->num_tokens + m_wg - 1) / m_wg);
  const long long kb = has_k ? (a->num_tokens + 3) / 4 : 0;
  FL_CHECK_ARG(qb + kb < (1ll << 31), "fl_mla_absorb_rope_quant: grid too large");
  p.q_blocks = (int)qb;
  const dim3 grid((unsigned)(qb + kb)), block(256);
  const size_t lds = kLdsBytes + 512;
#define FL_ABSORB_LAUNCH(WN_)                                                                                                  \
  do {                                                                                                                         \
    static std::atomic<unsigned char> done_[64];                                                                               \
    const hipError_t attr_ = fl_set_max_dynamic_lds(reinterpret_cast<const void*>(&mla_absorb_kernel<WN_>),                     \
                                                    (int)(kLdsBytes + 512), done_);                                             \
    FL_CHECK_ARG(attr_ == hipSuccess, "fl_mla_absorb_rope_quant: hipFuncSetAttribute(%d)", (int)attr_);                        \
    mla_absorb_kernel<WN_><<<grid, block, lds, (hipStream_t)stream>>>(p);                                                       \
  } while (0)
  if (wn == 1) FL_ABSORB_LAUNCH(1);
  else if (wn == 2) FL_ABSORB_LAUNCH(2);
  else FL_ABSORB_LAUNCH(4);
#undef FL_ABSORB_LAUNCH
  FL_CHECK_LAUNCH("mla_absorb_kernel");
  return FL_OK;
}
