// Hardware probe (not product code): how long does a WAVE stay in the issue of one vector-memory instruction?  The grouped GEMM and the MLA
// decode kernel feed their LDS rings with global_load_lds_dwordx4 (LDS-DMA); their phase timers show the issuing wave stalled 130-180 cycles
// per instruction.  Here: bursts of B instructions per wave with an idle gap between bursts (so that no queue stays full), s_memtime around
// each burst BEFORE any s_waitcnt; 1, 4 or 8 waves of one workgroup per CU, every CU busy.  Forms: LDS-DMA x4 (16 B per lane), LDS-DMA x1
// (4 B per lane), global_load_dwordx4 into VGPRs, and the VGPR form followed by its ds_write_b128.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int FORM, int B>
__global__ __launch_bounds__(512, 1) void issue_cost(const char* __restrict__ src, size_t region, int iters, unsigned long long* out) {
  __shared__ __attribute__((aligned(16))) char smem[128 * 1024];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const char* base = src + (size_t)blockIdx.x * region;
  const size_t pieces = region / 1024;
  size_t p = wave;
  unsigned long long tot = 0;
  uint4 keep = make_uint4(0, 0, 0, 0);
  for (int i = 0; i < iters; ++i) {
    u32x4 v[B];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_sleep(16);              // idle gap
    for (int w = 0; w < wave; ++w) __builtin_amdgcn_s_sleep(3);   // de-phased between the waves
    unsigned long long t0 = __builtin_readcyclecounter();
    if (FORM < 2) {
#pragma unroll
      for (int k = 0; k < B; ++k) {
        const char* g = base + p * 1024 + lane * 16;
        if (FORM == 0) __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(smem + wave * 16384 + k * 1024), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + p * 1024 + lane * 4), (lds_ptr_t)(smem + wave * 16384 + k * 1024), 4, 0, 0);
        p += nw;
        if (p >= pieces) p -= pieces;
      }
    } else {
      // (the loads are invisible to hipcc's wait-count bookkeeping: the destination registers are pinned by the "+v" operands of the
      //  s_waitcnt statement below and nothing but the timer read sits between)
      const char* g[B];
#pragma unroll
      for (int k = 0; k < B; ++k) {
        g[k] = base + p * 1024 + lane * 16;
        p += nw;
        if (p >= pieces) p -= pieces;
      }
      t0 = __builtin_readcyclecounter();
#pragma unroll
      for (int k = 0; k < B; ++k) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v[k]) : "v"(g[k]) : "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    tot += t1 - t0;
    if (FORM >= 2) {
#pragma unroll
      for (int k = 0; k < B; ++k) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[k])::"memory");
      if (FORM == 3) {
        const unsigned long long t2 = __builtin_readcyclecounter();
#pragma unroll
        for (int k = 0; k < B; ++k) *reinterpret_cast<u32x4*>(smem + wave * 16384 + k * 1024 + lane * 16) = v[k];
        tot += __builtin_readcyclecounter() - t2;
      } else {
#pragma unroll
        for (int k = 0; k < B; ++k) { keep.x ^= v[k][0]; keep.y ^= v[k][3]; }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (lane == 0) out[blockIdx.x * 8 + wave] = tot + (keep.x == 0x12345 ? smem[17] : 0);
}

template <int FORM, int B>
void run(const char* d, int waves, size_t region, const char* tag) {
  unsigned long long* out; hipMalloc(&out, 256 * 8 * 8); hipMemset(out, 0, 256 * 8 * 8);
  const int iters = 400;
  issue_cost<FORM, B><<<256, waves * 64>>>(d, region, 20, out);
  hipDeviceSynchronize();
  issue_cost<FORM, B><<<256, waves * 64>>>(d, region, iters, out);
  hipDeviceSynchronize();
  static unsigned long long h[256 * 8];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; int n = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) { s += (double)h[b * 8 + w]; ++n; }
  printf("%-44s waves/CU=%d burst=%d region=%5zu KB: %7.1f cycles of wave time per instruction\n", tag, waves, B, region / 1024, s / n / iters / B);
  hipFree(out);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  char* d; hipMalloc(&d, (size_t)1 << 30); hipMemset(d, 1, (size_t)1 << 30);
  const size_t L2 = 64 << 10, HBM = 4 << 20;
  for (int pass = 0; pass < 2; ++pass) {
    const size_t r = pass ? HBM : L2;
    const char* where = pass ? "HBM stream" : "L2-resident";
    printf("---- %s ----\n", where);
    run<0, 1>(d, 1, r, "LDS-DMA x4"); run<0, 4>(d, 1, r, "LDS-DMA x4"); run<0, 4>(d, 4, r, "LDS-DMA x4"); run<0, 4>(d, 8, r, "LDS-DMA x4"); run<0, 8>(d, 8, r, "LDS-DMA x4");
    run<1, 4>(d, 8, r, "LDS-DMA x1 (4 B per lane)");
    run<2, 1>(d, 1, r, "global_load_dwordx4 -> VGPR"); run<2, 4>(d, 1, r, "global_load_dwordx4 -> VGPR"); run<2, 4>(d, 4, r, "global_load_dwordx4 -> VGPR");
    run<2, 4>(d, 8, r, "global_load_dwordx4 -> VGPR"); run<2, 8>(d, 8, r, "global_load_dwordx4 -> VGPR");
    run<3, 4>(d, 8, r, "global_load_dwordx4 -> VGPR + ds_write_b128"); run<3, 8>(d, 8, r, "global_load_dwordx4 -> VGPR + ds_write_b128");
  }
  return 0;
}
