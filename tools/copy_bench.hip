// copy_bench.hip — what read+write HBM rate does this MI355X sustain? (ceiling for the FFT kernel)
// build: hipcc -O3 --offload-arch=gfx950 -o tools/copy_bench tools/copy_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int MODE, int UNROLL>
__global__ __launch_bounds__(512) void copy_k(const u4* __restrict__ src, u4* __restrict__ dst, size_t n) {
  // each workgroup walks contiguous UNROLL KiB*8 chunks, grid-stride
  const size_t per_iter = (size_t)blockDim.x * UNROLL;
  for (size_t base = (size_t)blockIdx.x * per_iter; base < n; base += (size_t)gridDim.x * per_iter) {
    u4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
      if (MODE & 1) v[u] = __builtin_nontemporal_load(src + i); else v[u] = src[i];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
      if (MODE & 2) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u];
    }
  }
}

template <int MODE, int UNROLL>
float run(const u4* s, u4* d, size_t n, int grid, int block) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((copy_k<MODE, UNROLL>), dim3(grid), dim3(block), 0, 0, s, d, n);
  hipEventRecord(e0);
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((copy_k<MODE, UNROLL>), dim3(grid), dim3(block), 0, 0, s, d, n);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  const size_t bytes = 1ull << 30, n = bytes / 16;
  u4 *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes);
  hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
  const int grids[] = {256, 512, 1024, 2048, 4096, 16384};
  for (int g : grids) {
    float a = run<0, 4>(s, d, n, g, 512), b = run<3, 4>(s, d, n, g, 512), c = run<1, 4>(s, d, n, g, 512), e = run<2, 4>(s, d, n, g, 512);
    float a8 = run<0, 8>(s, d, n, g, 512), b8 = run<3, 8>(s, d, n, g, 512);
    float a256 = run<0, 4>(s, d, n, g, 256), b256 = run<3, 4>(s, d, n, g, 256);
    printf("grid %5d: plain %6.0f  nt-both %6.0f  nt-load %6.0f  nt-store %6.0f | unroll8 plain %6.0f nt %6.0f | block256 plain %6.0f nt %6.0f  GB/s\n", g,
           2e-6 * bytes / a, 2e-6 * bytes / b, 2e-6 * bytes / c, 2e-6 * bytes / e, 2e-6 * bytes / a8, 2e-6 * bytes / b8, 2e-6 * bytes / a256, 2e-6 * bytes / b256);
  }
  // read-only and write-only rates
  return 0;
}
