// Hardware probe (not product code), gfx950: what does ONE filler instruction cost a wave that is alone on its SIMD and issues a continuous
// stream of v_mfma_scale_f32_32x32x64_f8f6f4 (64 cycles of matrix pipe each)?  4 waves per workgroup (one per SIMD), one workgroup per CU,
// every CU busy.  After every MFMA the wave issues N fillers of one kind; printed: cycles per MFMA slot against N.  Kinds: v_mul_f32 on an
// accumulator written two slots earlier (the block-scale rescale), ds_read_b128 -> AGPR, global_load_lds_dwordx4 (bare / with its m0 update),
// global_load_dwordx4 -> AGPR, ds_write_b128 from AGPR, and the last two together.  All four waves issue the same kind at the same time (as
// the grouped GEMM's waves do between two barriers); SKEW > 0 delays wave w by 16 w cycles once per 12 slots.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));

template <int KIND, int N, int SKEW>
__global__ __launch_bounds__(256, 1) void fillers(const char* __restrict__ src, int iters, unsigned long long* out, float* sink) {
  __shared__ __attribute__((aligned(16))) char smem[144 * 1024];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v16f acc[12];
#pragma unroll
  for (int t = 0; t < 12; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = (float)(lane + r + t);
  v8i opa, opb;
#pragma unroll
  for (int r = 0; r < 8; ++r) { opa[r] = 0x38383838 + lane * 0x01010101 * (r & 1); opb[r] = 0x3c343c34 ^ (lane << (r & 7)); }
  const int unit = 127;
  const float ratio = 1.0001f;
  const unsigned voff = (unsigned)(lane * 16);
  const char* sbase = src + (size_t)blockIdx.x * 65536 + wave * 16384;
  const int lds_rd = (int)(uintptr_t)smem + wave * 8192 + lane * 16;
  int lds_wr = (int)(uintptr_t)smem + 65536 + wave * 16384;
  const int lds_wr_lane = lds_wr + lane * 16;
  asm volatile("s_mov_b32 m0, %0" ::"s"(lds_wr) : "m0");
  unsigned long long t0 = 0, t1 = 0;
  auto loop = [&](auto wtag) __attribute__((always_inline)) {
  constexpr int W = decltype(wtag)::value;
  for (int it = -8; it < iters; ++it) {
    if (it == 0) t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc[s]) : "v"(opa), "v"(opb), "v"(unit));
      if (SKEW > 0 && s == 0) {
        if (wave & 1) for (int q = 0; q < SKEW; ++q) asm volatile("s_nop 15");
        if (wave & 2) for (int q = 0; q < 2 * SKEW; ++q) asm volatile("s_nop 15");
      }
      const int t = (s + 10) % 12;
      if (KIND == 8 || KIND == 9) {
        // the grouped GEMM's half step: 96 rescale multiplies, 7 fragment reads (2 ds_read_b128 each), 7 LDS-DMA pieces per wave.
        // KIND 8: every wave issues one piece behind the MFMAs of slots 2,3,5,7,9,10,11 (all four waves meet at the texture addresser), 8 multiplies per slot.
        // KIND 9: wave w issues its 4 weight pieces behind slot 2 + w and its 3 token pieces behind slot 6 + w and no multiplies there (N = multiplies in
        //         the token-burst slot); the other ten slots carry 12 (or 12 - N/... ) multiplies.
        const bool wburst = KIND == 9 && s == 2 + W, aburst = KIND == 9 && s == 6 + W;
        const bool dma8 = KIND == 8 && (s == 2 || s == 3 || s == 5 || s == 7 || s == 9 || s == 10 || s == 11);
        const int pieces = dma8 ? 1 : (wburst ? 4 : (aburst ? 3 : 0));
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < pieces) asm volatile("s_add_u32 m0, %2, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_wr) : "memory", "m0", "scc");
        const int muls = KIND == 8 ? 8 : (wburst ? 0 : (aburst ? N : (s < 2 ? 12 - (N + 1) / 2 : 12)));
#pragma unroll
        for (int n = 0; n < 16; ++n)
          if (n < muls) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(acc[t][n & 15]) : "v"(ratio));
        const bool rd = KIND == 8 ? (s >= 2 && s <= 8) : (s >= 2 && s <= 10 && !wburst && !aburst);
        if (rd) {
          asm volatile("ds_read_b128 a[0:3], %0" ::"v"(lds_rd) : "memory", "a0", "a1", "a2", "a3");
          asm volatile("ds_read_b128 a[4:7], %0 offset:2048" ::"v"(lds_rd) : "memory", "a4", "a5", "a6", "a7");
        }
        if (s == 1) { asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        if (s == 11) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
#pragma unroll
      for (int n = 0; n < (KIND >= 8 ? 0 : N); ++n) {
        if (KIND == 1) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(acc[t][n & 15]) : "v"(ratio));
        if (KIND == 2) {
          if (n & 1) asm volatile("ds_read_b128 a[4:7], %0 offset:2048" ::"v"(lds_rd) : "memory", "a4", "a5", "a6", "a7");
          else asm volatile("ds_read_b128 a[0:3], %0" ::"v"(lds_rd) : "memory", "a0", "a1", "a2", "a3");
        }
        if (KIND == 3) asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase) : "memory");
        if (KIND == 4) asm volatile("s_add_u32 m0, %2, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_wr) : "memory", "m0", "scc");
        if (KIND == 5 || KIND == 7) {
          if (n & 1) asm volatile("global_load_dwordx4 a[12:15], %0, %1 offset:1024" ::"v"(voff), "s"(sbase) : "memory", "a12", "a13", "a14", "a15");
          else asm volatile("global_load_dwordx4 a[8:11], %0, %1" ::"v"(voff), "s"(sbase) : "memory", "a8", "a9", "a10", "a11");
        }
        if (KIND == 6 || KIND == 7) {
          if (n & 1) asm volatile("ds_write_b128 %0, a[20:23] offset:1024" ::"v"(lds_wr_lane) : "memory");
          else asm volatile("ds_write_b128 %0, a[16:19]" ::"v"(lds_wr_lane) : "memory");
        }
      }
    }
    // (bound the queues: at most one group's requests stay in flight across the group boundary)
    if (KIND < 8) if (KIND == 3 || KIND == 4 || KIND == 5 || KIND == 7) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N * 12 > 63 ? 63 : N * 12) : "memory");
    if (KIND == 2 || KIND == 6 || KIND == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  };
  if (KIND == 9) {
    switch (wave) {
      case 0: loop(std::integral_constant<int, 0>{}); break;
      case 1: loop(std::integral_constant<int, 1>{}); break;
      case 2: loop(std::integral_constant<int, 2>{}); break;
      default: loop(std::integral_constant<int, 3>{}); break;
    }
  } else loop(std::integral_constant<int, 0>{});
  t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < 12; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += acc[t][r];
  if (sum == 12345.678f) sink[0] = sum;
  if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int KIND, int N, int SKEW>
static void run(const char* name, const char* src, unsigned long long* out, float* sink, int cus) {
  const int iters = 400;
  fillers<KIND, N, SKEW><<<cus, 256>>>(src, iters, out, sink);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  fillers<KIND, N, SKEW><<<cus, 256>>>(src, iters, out, sink);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long* h = (unsigned long long*)malloc(cus * 4 * 8);
  hipMemcpy(h, out, cus * 4 * 8, hipMemcpyDeviceToHost);
  double tot = 0;
  for (int i = 0; i < cus * 4; ++i) tot += (double)h[i];
  free(h);
  const double per = tot / (cus * 4) / iters / 12.0;
  printf("%-46s N=%d skew=%d: %6.1f cycles per MFMA slot (+%5.1f over 64; %5.1f per filler)   [%.3f ms, %.0f MHz]\n", name, N, SKEW, per, per - 64.0,
         N ? (per - 64.0) / N : 0.0, ms, tot / (cus * 4) / (ms * 1e3));
}

int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  char* src; hipMalloc(&src, (size_t)cus * 65536 + 4096); hipMemset(src, 0x3a, (size_t)cus * 65536 + 4096);
  unsigned long long* out; hipMalloc(&out, cus * 4 * 8);
  float* sink; hipMalloc(&sink, 64);
  run<0, 0, 0>("bare MFMA stream", src, out, sink, cus);
#define ROW(K, name) run<K, 1, 0>(name, src, out, sink, cus); run<K, 2, 0>(name, src, out, sink, cus); run<K, 4, 0>(name, src, out, sink, cus);
  run<1, 4, 0>("v_mul_f32 (acc of two slots back)", src, out, sink, cus);
  run<1, 8, 0>("v_mul_f32 (acc of two slots back)", src, out, sink, cus);
  run<1, 12, 0>("v_mul_f32 (acc of two slots back)", src, out, sink, cus);
  run<1, 16, 0>("v_mul_f32 (acc of two slots back)", src, out, sink, cus);
  ROW(2, "ds_read_b128 -> AGPR")
  ROW(3, "global_load_lds_dwordx4, m0 fixed")
  ROW(4, "s_add m0 + s_nop + global_load_lds_dwordx4")
  ROW(5, "global_load_dwordx4 -> AGPR")
  ROW(6, "ds_write_b128 from AGPR")
  ROW(7, "global_load_dwordx4 -> AGPR + ds_write_b128")
  run<8, 0, 0>("GEMM half step, shipped pattern (1 piece x 4 waves in 7 slots)", src, out, sink, cus);
  run<9, 0, 0>("GEMM half step, one wave's burst per slot", src, out, sink, cus);
  run<9, 4, 0>("GEMM half step, one wave's burst per slot", src, out, sink, cus);
  run<4, 1, 1>("s_add m0 + s_nop + global_load_lds_dwordx4", src, out, sink, cus);
  run<4, 1, 2>("s_add m0 + s_nop + global_load_lds_dwordx4", src, out, sink, cus);
  run<4, 2, 1>("s_add m0 + s_nop + global_load_lds_dwordx4", src, out, sink, cus);
  run<7, 1, 1>("global_load_dwordx4 -> AGPR + ds_write_b128", src, out, sink, cus);
  return 0;
}
