// Probe (gfx950): sustained v_mfma_scale_f32_32x32x64_f8f6f4 rate of the WHOLE chip with register-resident operands —
// the power-capped MFMA ceiling for random vs zero operand bits (DVFS: the chip clocks to its power budget).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 1) void k(float* out, int iters, int random_bits) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  v8i a, b;
  for (int i = 0; i < 8; ++i) {
    unsigned h = (tid * 2654435761u) ^ (i * 0x9E3779B9u);
    h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13;
    unsigned ha = h & 0x77777777u, hb = (h * 0xC2B2AE35u) & 0x77777777u;   // finite e4m3 patterns
    a[i] = random_bits ? (int)ha : 0;
    b[i] = random_bits ? (int)hb : 0;
  }
  v16f acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  const int unit = 0x7f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc[j]) : "v"(a), "v"(b), "v"(unit));
  }
  asm volatile("s_nop 7\ns_nop 7\ns_nop 7" ::: "memory");
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  out[tid] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 8 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd)
    for (int rnd = 0; rnd < 2; ++rnd) {
      const int iters = 20000, threads = 256 * waves_per_simd, blocks = 256;
      k<<<blocks, threads>>>(out, 2000, rnd);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 5; ++r) k<<<blocks, threads>>>(out, iters, rnd);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = 5.0 * blocks * (threads / 64) * (double)iters * 4 * (32.0 * 32 * 64 * 2);
      printf("%d wave(s)/SIMD, %s operands: %.0f TFLOP/s sustained over %.1f ms (5 PF nominal)\n", waves_per_simd,
             rnd ? "random" : "zero", flops / ms / 1e9, ms);
    }
  return 0;
}
