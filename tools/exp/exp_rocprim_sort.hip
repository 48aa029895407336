// Experiment (not product): the vendor library's radix sort on the sort workload's input, as a yardstick
// for gt4hip_sort.hip.  hipcc --offload-arch=gfx950 -O3 exp_rocprim_sort.hip -o exp_rocprim_sort
#include <cstring>
#include <string.h>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstring>
#include <cstdlib>
__global__ void fill (unsigned long long *w, unsigned long long n, unsigned long long mask)
{
  for (unsigned long long i = blockIdx.x * (unsigned long long) blockDim.x + threadIdx.x; i < n; i += (unsigned long long) gridDim.x * blockDim.x) {
    unsigned long long x = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29; x *= 0x94D049BB133111EBull; x ^= x >> 32;
    w[i] = x & mask;
  }
}
int main (int argc, char **argv)
{
  unsigned long long n = argc > 1 ? strtoull (argv[1], 0, 10) : 1000000000ull;
  unsigned bits = argc > 2 ? atoi (argv[2]) : 50;
  unsigned long long *a, *b; void *tmp = 0; size_t tb = 0;
  hipMalloc (&a, n * 8); hipMalloc (&b, n * 8);
  rocprim::radix_sort_keys (tmp, tb, a, b, n, 0, bits, 0);
  hipMalloc (&tmp, tb);
  hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
  for (int rep = 0; rep < 3; rep++) {
    fill<<<4096, 256>>> (a, n, bits >= 64 ? ~0ull : (1ull << bits) - 1);
    hipEventRecord (e0, 0);
    rocprim::radix_sort_keys (tmp, tb, a, b, n, 0, bits, 0);
    hipEventRecord (e1, 0); hipEventSynchronize (e1);
    float ms; hipEventElapsedTime (&ms, e0, e1);
    printf ("rocprim radix_sort_keys n=%llu bits=%u temp=%zu MB: %.2f ms\n", n, bits, tb >> 20, ms);
  }
  return 0;
}
