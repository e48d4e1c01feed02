// Probe (gfx950): cost of s_nop 7 in cycles (is a "wait state" one clock or four?)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* t, int iters) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) asm volatile("s_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\ns_nop 7" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) asm volatile("s_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0" ::: "memory");
  const unsigned long long t2 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { t[0] = t1 - t0; t[1] = t2 - t1; }
}
int main() {
  unsigned long long* d; hipMalloc(&d, 16);
  k<<<1, 64>>>(d, 1000); k<<<1, 64>>>(d, 1000); hipDeviceSynchronize();
  unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("s_nop 7: %.1f cycles each; s_nop 0: %.1f cycles each\n", h[0] / 8000.0, h[1] / 8000.0);
  return 0;
}
