// Calibration microbenchmark (not product): VALU issue rate per SIMD on gfx950 for integer ops,
// to convert SQ_INSTS_VALU into a utilisation figure.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32;
template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256) void k(u32* out, int iters) {
  u32 a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      a0 = a0 * 3 + a1; a1 = a1 ^ (a2 >> 1); a2 = a2 + a3; a3 = a3 | (a4 << 1); a4 = a4 + a5; a5 = a5 ^ a6; a6 = a6 + a7; a7 = a7 + a0;
    }
  }
  if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345) out[0] = a0;
}
int main() {
  u32* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks_per_cu = 1; blocks_per_cu <= 8; blocks_per_cu *= 2) {
    int iters = 2000;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<1>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // ops per thread per iter: 16 * ~11 VALU (mul, add, xor, shr, add, or, shl, add, xor, add, add)
      double wave_instr = 256.0 * blocks_per_cu * 4 /*waves*/ * (double) iters * 16 * 11;
      if (rep) printf("waves/SIMD %d: %.3f ms, %.2f wave-instr per ns chip-wide, %.3f per SIMD-cycle at 2.4 GHz\n", blocks_per_cu, ms,
                      wave_instr / ms / 1e6, wave_instr / ms / 1e6 / 1024 / 2.4);
    }
  }
  return 0;
}
