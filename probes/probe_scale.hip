// Hardware probe (not product code): per-lane E8M0 block-scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4.
// Hypothesis: lane l supplies the scale of its own 32-element K block: A scale for (row l&31, kblock l>>5),
// B scale for (col l&31, kblock l>>5); value = 2^(byte-127); opsel picks the byte of the 32-bit operand.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k(float* out, int mode) {
  int lane = threadIdx.x;
  const uint8_t ONE = 0x38;
  uint8_t ab[32], bb[32];
  for (int i = 0; i < 32; ++i) { ab[i] = ONE; bb[i] = ONE; }
  v8i a, b; memcpy(&a, ab, 32); memcpy(&b, bb, 32);
  v16f c = {0};
  // B scale: lane-dependent exponent: col n = lane&31, block H = lane>>5: byte = 127 - (n%4) - 8*H
  int sb = 127 - (lane & 3) - 8 * (lane >> 5);
  int sa = 127;
  if (mode == 1) { sa = 127 - (lane & 1); sb = 127; }          // A scale per (row, block)
  if (mode == 2) { sb = (sb << 8) | 0x7F; }                      // scale in byte 1, opsel_b = 1
  if (mode == 2)
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 1, sb);
  else
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}
int main() {
  float* d; hipMalloc(&d, 64 * 16 * 4);
  float h[1024];
  for (int mode = 0; mode < 3; ++mode) {
    k<<<1, 64>>>(d, mode); hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 16; ++r) {
      int col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      double expect;
      if (mode == 1) expect = 32.0 * (1.0 / (1 << (row & 1))) * 2;   // both k-blocks of row use lane row and row+32: (row&1) same
      else expect = 32.0 * std::pow(2.0, -(col & 3)) + 32.0 * std::pow(2.0, -(col & 3) - 8);
      if (std::fabs(h[lane * 16 + r] - expect) > 1e-3 * expect) { if (bad < 5) printf("mode %d mismatch lane %d r %d (row %d col %d): got %g expect %g\n", mode, lane, r, row, col, h[lane * 16 + r], expect); bad++; }
    }
    printf("mode %d: bad=%d  sample c[lane0]=%g c[lane1]=%g c[lane2]=%g c[lane3]=%g\n", mode, bad, h[0], h[16], h[32], h[48]);
  }
  return 0;
}
