// Hardware probe (not product code), gfx950: operand layouts for the bf16 MLA kernel.
//  1. v_mfma_f32_32x32x16_bf16: (lane, element) -> (row, k) of A, (k, col) of B.
//  2. ds_read_b64_tr_b16: which (source lane, element) each destination element comes from.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef short v4s __attribute__((ext_vector_type(4)));

__device__ v8bf onehot(int lane, int la, int ja) {
  union { v8bf v; uint16_t u[8]; } x;
  for (int i = 0; i < 8; ++i) x.u[i] = 0;
  if (lane == la) x.u[ja] = 0x3f80;   // bf16 1.0
  return x.v;
}
__device__ v8bf ones() {
  union { v8bf v; uint16_t u[8]; } x;
  for (int i = 0; i < 8; ++i) x.u[i] = 0x3f80;
  return x.v;
}

__global__ void probe_mfma(int* rowA, int* colB, int* kOfA, int* kOfB) {
  const int lane = threadIdx.x;
  // A one-hot, B ones -> nonzero C row (C layout: col = lane&31, row = (r&3) + 8(r>>2) + 4(lane>>5))
  for (int t = 0; t < 64 * 8; ++t) {
    v16f c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(onehot(lane, t / 8, t % 8), ones(), c, 0, 0, 0);
    for (int r = 0; r < 16; ++r)
      if (c[r] != 0.f && (lane & 31) == 0) rowA[t] = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
  }
  for (int t = 0; t < 64 * 8; ++t) {
    v16f c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones(), onehot(lane, t / 8, t % 8), c, 0, 0, 0);
    bool nz = false;
    for (int r = 0; r < 16; ++r) nz |= (c[r] != 0.f);
    const unsigned long long m = __ballot(nz);
    if (lane == 0) colB[t] = __ffsll((long long)m) - 1;
  }
  // k index: A one-hot (la, ja) against B one-hot (lb, jb) for lb in {0, 32} (column 0): nonzero <=> same k
  for (int t = 0; t < 64 * 8; ++t) {
    int found = -1;
    for (int q = 0; q < 16; ++q) {   // q = (lb half)*8 + jb
      v16f c = {0};
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(onehot(lane, t / 8, t % 8), onehot(lane, (q >> 3) * 32, q & 7), c, 0, 0, 0);
      bool nz = false;
      for (int r = 0; r < 16; ++r) nz |= (c[r] != 0.f);
      if (__ballot(nz)) found = q;
    }
    if (lane == 0) kOfA[t] = found;
  }
  if (lane == 0) for (int q = 0; q < 16; ++q) kOfB[q] = q;
}

__global__ void probe_tr16(int* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 4];
  const int lane = threadIdx.x;
  for (int j = 0; j < 4; ++j) lds[lane * 4 + j] = (uint16_t)(lane * 4 + j);   // value = source lane*4 + element
  __syncthreads();
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + lane * 4));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (int)(uint16_t)r[j];
}

int main() {
  int *d; hipMalloc(&d, 4 * 64 * 8 * 4 * 4);
  int* rowA = d; int* colB = d + 512; int* kA = d + 1024; int* kB = d + 1536;
  hipMemset(d, 0xff, 4 * 64 * 8 * 4 * 4);
  probe_mfma<<<1, 64>>>(rowA, colB, kA, kB);
  int h[2048]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("A: lane -> row (elem 0) :"); for (int l = 0; l < 64; ++l) printf(" %d", h[l * 8]); printf("\n");
  printf("A: (lane,elem) -> k, lanes 0 and 32:");
  for (int l = 0; l < 64; l += 32) { printf(" | lane %d:", l); for (int j = 0; j < 8; ++j) printf(" %d", h[1024 + l * 8 + j]); }
  printf("\nA row varies with elem? lane 5:"); for (int j = 0; j < 8; ++j) printf(" %d", h[5 * 8 + j]);
  printf("\nB: lane -> col (elem 0) :"); for (int l = 0; l < 64; ++l) printf(" %d", h[512 + l * 8]); printf("\n");
  int* o; hipMalloc(&o, 256 * 4);
  probe_tr16<<<1, 64>>>(o);
  int t[256]; hipMemcpy(t, o, sizeof(t), hipMemcpyDeviceToHost);
  printf("tr16: dst lane i elems j <- (src lane, src elem):\n");
  for (int l = 0; l < 64; ++l) {
    printf("  %2d:", l);
    for (int j = 0; j < 4; ++j) printf(" (%2d,%d)", t[l * 4 + j] / 4, t[l * 4 + j] % 4);
    if (l % 4 == 3) printf("\n");
  }
  return 0;
}
