// Probe (gfx950): what ONE dependent accumulate chain of MFMAs costs per instruction (the QK block of the MLA decode
// kernels is such a chain: 4 x 32x32x16 bf16 + 8 x MX-fp8 32x32x64 into one accumulator), against independent
// accumulators; plus the unit of __builtin_readcyclecounter (s_memtime) against wall_clock64 (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NACC, int KIND>   // KIND 0: MX fp8 32x32x64, 1: bf16 32x32x16
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (tid * 2654435761u + i * 0x9E3779B9u) & 0x37373737; b[i] = (tid * 40503u + i * 0x85EBCA6Bu) & 0x37373737; }
  v4i a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
  v16f acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  const int unit = 0x7f;
  const unsigned long long w0 = wall_clock64();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8 / NACC; ++r)
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        if (KIND == 0) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc[j]) : "v"(a), "v"(b), "v"(unit));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a4), "v"(b4));
      }
  }
  asm volatile("s_nop 7\ns_nop 7\ns_nop 7" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long w1 = wall_clock64();
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  if (threadIdx.x == 0) { out[blockIdx.x * 4 + 0] = t1 - t0; out[blockIdx.x * 4 + 1] = w1 - w0; out[blockIdx.x * 4 + 2] = (unsigned long long)s; }
}

template <int NACC, int KIND>
void run(unsigned long long* out, int blocks, const char* tag) {
  const int iters = 2000;
  k<NACC, KIND><<<blocks, 256>>>(out, 100);
  hipDeviceSynchronize();
  k<NACC, KIND><<<blocks, 256>>>(out, iters);
  hipDeviceSynchronize();
  unsigned long long h[4];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-34s blocks=%3d: %.1f s_memtime ticks per MFMA, %.2f ticks per wall-ns (wall_clock64 = 100 MHz)\n", tag, blocks,
         (double)h[0] / (iters * 8.0), (double)h[0] / (h[1] * 10.0));
}

int main() {
  unsigned long long* out;
  hipMalloc(&out, 256 * 4 * 8);
  for (int blocks : {1, 256}) {
    run<1, 0>(out, blocks, "fp8 MX 32x32x64, 1 dependent chain");
    run<2, 0>(out, blocks, "fp8 MX 32x32x64, 2 accumulators");
    run<4, 0>(out, blocks, "fp8 MX 32x32x64, 4 accumulators");
    run<1, 1>(out, blocks, "bf16 32x32x16, 1 dependent chain");
    run<4, 1>(out, blocks, "bf16 32x32x16, 4 accumulators");
  }
  return 0;
}
