// Calibration microbenchmark (not product): HBM read bandwidth of MI355X as a function of how many
// 16-byte loads each wavefront keeps in flight, to size the merge kernel's prefetch depth.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(512) void k_read(const u32x4* __restrict__ p, size_t n_vec, u32* out) {
  // each block walks contiguous 12 KB "tiles" (768 vec), DEPTH loads per thread in flight
  const size_t tiles = n_vec / (512 * DEPTH);
  u32 acc = 0;
  for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const u32x4* base = p + t * 512 * DEPTH;
    u32x4 v[DEPTH];
#pragma unroll
    for (int j = 0; j < DEPTH; j++) v[j] = base[j * 512 + threadIdx.x];
#pragma unroll
    for (int j = 0; j < DEPTH; j++) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
  }
  if (acc == 0x12345678) out[0] = acc;
}

template <int DEPTH>
void run(const u32x4* d, size_t n_vec, u32* out, int blocks_per_cu) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int grid = 256 * blocks_per_cu;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_read<DEPTH>, dim3(grid), dim3(512), 0, 0, d, n_vec, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep == 1) printf("depth %2d blocks/CU %d : %.2f ms  %.0f GB/s\n", DEPTH, blocks_per_cu, ms, n_vec * 16.0 / ms / 1e6);
  }
}

int main() {
  size_t bytes = 24ull << 30;
  u32x4* d; u32* out;
  if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, 4);
  hipMemset(d, 1, bytes);
  size_t n_vec = bytes / 16;
  for (int b = 1; b <= 4; b++) { run<1>(d, n_vec, out, b); run<3>(d, n_vec, out, b); run<6>(d, n_vec, out, b); run<12>(d, n_vec, out, b); }
  return 0;
}
