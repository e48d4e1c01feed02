// Probe (gfx950): does VALU / transcendental / LDS-read work of a wave hide under its own in-flight MFMAs?
// One wave per SIMD.  Each iteration: one v_mfma_scale_f32_32x32x64_f8f6f4 of a DEPENDENT accumulate chain followed by
// N independent VALU ops (v_fma_f32 / v_exp_f32) or N ds_read_b128.  Prints ticks (s_memtime) per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));

template <int N, int KIND, int MF>
__global__ __launch_bounds__(256, 1) void probe(float* out, unsigned long long* ticks, int iters) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[65536];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<int*>(lds)[i] = i;
  __syncthreads();
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + lane; b[i] = 0x30303030 + i; }
  v16f acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = 1.0f + lane * 1e-3f + i;
  v4i ld[4] = {};
  const int unit = 0x7f;
  const uint8_t* lp = lds + lane * 16;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MF) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(a), "v"(b), "v"(unit));
#pragma unroll
    for (int k = 0; k < N; ++k) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[k & 7]) : "v"(x[(k + 1) & 7]));
      else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[k & 7]));
      else if (KIND == 2) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(ld[k & 3]) : "v"((unsigned)(uintptr_t)(lp + ((it * N + k) & 31) * 1024) & 0xffffu));
      else asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double*>(&x[(k & 3) * 2])) : "v"(*reinterpret_cast<double*>(&x[((k + 1) & 3) * 2])));
    }
    if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_nop 7\ns_nop 7\ns_nop 7" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  for (int i = 0; i < 4; ++i) s += (float)ld[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int N, int KIND, int MF>
void run(const char* name, float* out, unsigned long long* ticks) {
  const int iters = 2000;
  probe<N, KIND, MF><<<1, 256>>>(out, ticks, iters);
  probe<N, KIND, MF><<<1, 256>>>(out, ticks, iters);
  hipDeviceSynchronize();
  unsigned long long t;
  hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  printf("%-12s mfma=%d N=%2d : %7.1f ticks/iter\n", name, MF, N, (double)t / iters);
}

// 4 INDEPENDENT accumulators round-robin (the PV pattern), each MFMA followed by N v_fma / M v_exp
template <int N, int M, int KM = 0>
__global__ __launch_bounds__(256, 1) void probe_ind(float* out, unsigned long long* ticks, int iters) {
  const int lane = threadIdx.x & 63;
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + lane; b[i] = 0x30303030 + i; }
  v16f acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = 1.0f + lane * 1e-3f + i;
  const int unit = 0x7f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (KM == 0) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+a"(acc[j]) : "v"(a), "v"(b), "v"(unit));
      else if (KM == 1) asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(*reinterpret_cast<v4i*>(&a)), "v"(*reinterpret_cast<v4i*>(&b)));
#pragma unroll
      for (int k = 0; k < N; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[k & 7]) : "v"(x[(k + 1) & 7]));
#pragma unroll
      for (int k = 0; k < M; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(x[k & 7]));
    }
  }
  asm volatile("s_nop 7\ns_nop 7\ns_nop 7" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int N, int M, int KM = 0>
void run_ind(float* out, unsigned long long* ticks) {
  const int iters = 2000;
  probe_ind<N, M, KM><<<1, 256>>>(out, ticks, iters);
  probe_ind<N, M, KM><<<1, 256>>>(out, ticks, iters);
  hipDeviceSynchronize();
  unsigned long long t;
  hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  printf("independent acc (AGPR) mfma kind %d (0 scale, 1 plain fp8, 2 bf16 x16): per MFMA  v_fma N=%2d v_exp M=%d : %7.1f ticks\n", KM, N, M, (double)t / iters / 4);
}

int main() {
  float* out; unsigned long long* ticks;
  hipMalloc(&out, 256 * 4 * 4); hipMalloc(&ticks, 64);
  run<0, 0, 1>("none", out, ticks);
  run<4, 0, 1>("v_fma", out, ticks);  run<8, 0, 1>("v_fma", out, ticks);  run<12, 0, 1>("v_fma", out, ticks);
  run<16, 0, 1>("v_fma", out, ticks); run<24, 0, 1>("v_fma", out, ticks); run<32, 0, 1>("v_fma", out, ticks);
  run<16, 0, 0>("v_fma", out, ticks); run<32, 0, 0>("v_fma", out, ticks);
  run<8, 3, 1>("v_pk_fma", out, ticks); run<16, 3, 1>("v_pk_fma", out, ticks); run<16, 3, 0>("v_pk_fma", out, ticks);
  run<2, 1, 1>("v_exp", out, ticks);  run<4, 1, 1>("v_exp", out, ticks);  run<8, 1, 1>("v_exp", out, ticks);
  run<8, 1, 0>("v_exp", out, ticks);
  run<2, 2, 1>("ds_read128", out, ticks); run<4, 2, 1>("ds_read128", out, ticks); run<8, 2, 1>("ds_read128", out, ticks);
  run<16, 2, 1>("ds_read128", out, ticks);
  run<4, 2, 0>("ds_read128", out, ticks); run<8, 2, 0>("ds_read128", out, ticks); run<16, 2, 0>("ds_read128", out, ticks);
  run_ind<0, 0>(out, ticks); run_ind<4, 0>(out, ticks); run_ind<8, 0>(out, ticks); run_ind<12, 0>(out, ticks);
  run_ind<16, 0>(out, ticks); run_ind<24, 0>(out, ticks); run_ind<8, 2>(out, ticks); run_ind<6, 4>(out, ticks);
  run_ind<0, 0, 1>(out, ticks); run_ind<8, 0, 1>(out, ticks); run_ind<12, 0, 1>(out, ticks); run_ind<16, 0, 1>(out, ticks);
  run_ind<0, 0, 2>(out, ticks); run_ind<4, 0, 2>(out, ticks); run_ind<8, 0, 2>(out, ticks); run_ind<12, 0, 2>(out, ticks);
  return 0;
}
