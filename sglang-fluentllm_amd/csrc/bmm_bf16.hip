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

__device__ __forceinline__ v8bf ld_frag(const uint16_t* p) {
  union { uint4 u; v8bf v; } x;
  x.u = *reinterpret_cast<const uint4*>(p);
  return x.v;
}

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
