// Hardware probe (not product code): per-CU LDS-DMA (global_load_lds) ingest rate from L2-resident / HBM data
// as a function of pieces in flight per wave.  Each workgroup (4 waves) streams its own region round-robin.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int DEPTH, int SWZ>
__global__ __launch_bounds__(256, 1) void dma_rate(const char* __restrict__ src, size_t region_bytes, int iters, int* sink) {
  __shared__ __attribute__((aligned(16))) char smem[128 * 1024];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src + (size_t)blockIdx.x * region_bytes;
  const size_t pieces = region_bytes / 1024;
  size_t p = wave;
  for (int i = 0; i < iters; ++i) {
    // each wave: DEPTH pieces per step into its own 32 KiB LDS quarter
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
      // SWZ: the MLA kernel's source pattern: piece = 2 rows of 512 B, lane (li, lh) reads chunk li ^ (T & 15) of row T
      const int T = (int)(p * 2) + (lane >> 5);
      const size_t off = SWZ ? (size_t)p * 1024 + (lane >> 5) * 512 + (((lane & 31) ^ (T & 15)) << 4) : (size_t)p * 1024 + lane * 16;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + off), (lds_ptr_t)(smem + wave * 32768 + k * 1024), 16, 0, 0);
      p += 4;
      if (p >= pieces) p -= pieces;
    }
    if (DEPTH >= 32) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (DEPTH >= 16) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (DEPTH >= 8) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = smem[123];
}

// the same stream through REGISTERS: DEPTH global_load_dwordx4 in flight per wave, then DEPTH ds_write_b128 (what an
// LDS-staged kernel would do without LDS-DMA): is the CU's vector-memory path faster when the data returns to VGPRs?
template <int DEPTH>
__global__ __launch_bounds__(256, 1) void ld_rate(const char* __restrict__ src, size_t region_bytes, int iters, int* sink) {
  __shared__ __attribute__((aligned(16))) char smem[128 * 1024];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src + (size_t)blockIdx.x * region_bytes;
  const size_t pieces = region_bytes / 1024;
  size_t p = wave;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int i = 0; i < iters; ++i) {
    uint4 v[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
      v[k] = *reinterpret_cast<const uint4*>(base + (size_t)p * 1024 + lane * 16);
      p += 4;
      if (p >= pieces) p -= pieces;
    }
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) *reinterpret_cast<uint4*>(smem + wave * 32768 + k * 1024 + lane * 16) = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = smem[123] + acc.x;
}

template <int DEPTH>
void run_ld(const char* d, size_t region, int blocks, const char* tag) {
  int* sink; hipMalloc(&sink, 4096 * 4);
  const int iters = 2000 / DEPTH * 8;
  ld_rate<DEPTH><<<blocks, 256>>>(d, region, 50, sink);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  ld_rate<DEPTH><<<blocks, 256>>>(d, region, iters, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)blocks * 4 * iters * DEPTH * 1024;
  printf("%-28s blocks=%4d depth=%2d region=%6zu KB: %7.1f GB/s total, %6.1f GB/s per CU   (global_load -> VGPR -> ds_write)\n", tag,
         blocks, DEPTH, region / 1024, bytes / ms / 1e6, bytes / ms / 1e6 / blocks);
  hipFree(sink);
}

template <int DEPTH, int SWZ = 0>
void run(const char* d, size_t region, int blocks, const char* tag) {
  int* sink; hipMalloc(&sink, 4096 * 4);
  const int iters = 2000 / DEPTH * 8;
  dma_rate<DEPTH, SWZ><<<blocks, 256>>>(d, region, 50, sink);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  dma_rate<DEPTH, SWZ><<<blocks, 256>>>(d, region, iters, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)blocks * 4 * iters * DEPTH * 1024;
  printf("%-28s blocks=%4d depth=%2d region=%6zu KB: %7.1f GB/s total, %6.1f GB/s per CU\n", tag, blocks, DEPTH, region / 1024,
         bytes / ms / 1e6, bytes / ms / 1e6 / blocks);
  hipFree(sink);
}

int main() {
  const size_t total = (size_t)8 << 30;
  char* d; hipMalloc(&d, total); hipMemset(d, 1, total);
  // L2-resident: 64 KB per WG (256 WGs -> 16 MB total, 2 MB per XCD)
  run<4>(d, 64 << 10, 256, "L2-resident 64KB/WG");
  run<8>(d, 64 << 10, 256, "L2-resident 64KB/WG");
  run<16>(d, 64 << 10, 256, "L2-resident 64KB/WG");
  run<16>(d, 64 << 10, 128, "L2-resident 64KB/WG");
  // HBM stream: 16 MB per WG (4 GB total)
  run<4>(d, 16 << 20, 256, "HBM stream 16MB/WG");
  run<8>(d, 16 << 20, 256, "HBM stream 16MB/WG");
  run<16>(d, 16 << 20, 256, "HBM stream 16MB/WG");
  run<16>(d, 16 << 20, 128, "HBM stream 16MB/WG");
  run<32>(d, 16 << 20, 128, "HBM stream 16MB/WG");
  run<32>(d, 16 << 20, 256, "HBM stream 16MB/WG");
  run<8>(d, 16 << 20, 128, "HBM stream 16MB/WG");
  run<4>(d, 16 << 20, 128, "HBM stream 16MB/WG");
  run<16>(d, 16 << 20, 64, "HBM stream 16MB/WG");
  run<32>(d, 16 << 20, 64, "HBM stream 16MB/WG");
  run<16, 1>(d, 16 << 20, 128, "HBM stream SWIZZLED");
  run<16, 1>(d, 16 << 20, 256, "HBM stream SWIZZLED");
  run<16, 1>(d, 64 << 10, 256, "L2-resident SWIZZLED");
  run_ld<4>(d, 64 << 10, 256, "L2-resident 64KB/WG");
  run_ld<8>(d, 64 << 10, 256, "L2-resident 64KB/WG");
  run_ld<16>(d, 64 << 10, 256, "L2-resident 64KB/WG");
  run_ld<8>(d, 16 << 20, 256, "HBM stream 16MB/WG");
  run_ld<16>(d, 16 << 20, 256, "HBM stream 16MB/WG");
  // MALL-resident: 512 KB per WG (128 MB total)
  run<16>(d, 512 << 10, 256, "MALL-resident 512KB/WG");
  return 0;
}
