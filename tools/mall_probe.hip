// mall_probe.hip — can the 256-MiB Infinity Cache carry the intermediate of a two-pass transform? A producer copy writes a
// 64-MiB buffer T (the size of one 4096 x 4096 fp16 complex image), a consumer copy reads it back out, image after image:
//   per-image, T distinct   in_i -> T_i, T_i -> out_i            (T never re-used: every byte goes through HBM)
//   per-image, T re-used    in_i -> T,   T   -> out_i            (T stays within 64 MiB: may live in the Infinity Cache)
//   whole batch per pass    in -> T_all, T_all -> out            (what the library does: two launches over 8 GiB each)
// build: hipcc -O3 --offload-arch=gfx950 -o tools/mall_probe tools/mall_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void copy_k(const u4* __restrict__ src, u4* __restrict__ dst, uint64_t n16) {
  for (uint64_t i = blockIdx.x * 512ull + threadIdx.x; i < n16; i += gridDim.x * 512ull) dst[i] = src[i];
}

int main() {
  const uint64_t image = 64ull << 20, images = 128;          // bytes
  uint8_t *in, *out, *t;
  hipMalloc(&in, image * images);
  hipMalloc(&out, image * images);
  hipMalloc(&t, image * images);
  hipMemset(in, 1, image * images);
  hipMemset(t, 0, image * images);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto cp = [&](const uint8_t* s, uint8_t* d, uint64_t bytes, int grid) {
    hipLaunchKernelGGL(copy_k, dim3(grid), dim3(512), 0, 0, reinterpret_cast<const u4*>(s), reinterpret_cast<u4*>(d), bytes / 16);
  };
  for (int grid : {256, 1024}) {
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (mode == 2) {
          cp(in, t, image * images, grid);
          cp(t, out, image * images, grid);
        } else {
          const uint64_t per = mode == 3 ? 4 : 1;              // mode 3: four images at a time (256 MiB of T re-used)
          for (uint64_t i = 0; i < images; i += per) {
            uint8_t* ti = (mode == 0) ? t + i * image : t;
            cp(in + i * image, ti, image * per, grid);
            cp(ti, out + i * image, image * per, grid);
          }
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      const char* names[4] = {"per image, T distinct      ", "per image, T re-used (64 MiB)", "whole batch per pass       ", "4 images, T re-used (256 MiB)"};
      printf("grid %4d  %s: %8.3f ms  %6.0f GB/s (2 passes x 16 GiB)\n", grid, names[mode], best, 4.0 * image * images / best * 1e-6);
    }
  }
  return 0;
}
