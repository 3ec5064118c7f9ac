// rows2d_copy.hip — copy ceiling of the access pattern of the fused 2D row pass (k4096r.hpp, ROWS): a workgroup iteration
// moves the 8 rows r0 + 512 i of a 4096 x 4096 image (8 KiB per row and plane, rows 4 MiB apart, planes 2 GiB apart), wave i
// row i, in the kernel's rotated iteration order; `adj` = 1 takes 8 ADJACENT rows instead (what a re-laid-out intermediate
// image would give the write side), `nt` = non-temporal accesses.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/rows2d_copy tools/rows2d_copy.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(512) void k(const uint16_t* in_re, const uint16_t* in_im, uint16_t* out_re, uint16_t* out_im,
                                        uint32_t iterations, int adj_in, int adj_out, int ilv_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t rnd = 0, pos = blockIdx.x;
  for (uint32_t it = blockIdx.x; it < iterations; ++rnd, pos = (pos + 1 == gridDim.x) ? 0 : pos + 1, it = rnd * gridDim.x + pos) {
    const uint64_t img = static_cast<uint64_t>(it >> 9) * 4096 * 4096;
    const uint32_t r0 = it & 511;
    const uint64_t row_in = adj_in ? (8u * r0 + wave) : (r0 + 512u * wave);
    const uint64_t row_out = adj_out ? (8u * r0 + wave) : (r0 + 512u * wave);
    const uint16_t* sr = in_re + img + row_in * 4096;
    const uint16_t* si = in_im + img + row_in * 4096;
    // ilv_out: the output image keeps [RE row | IM row] pairs (row pitch 2 x 4096 halves): what the library's own intermediate
    // image set could look like
    uint16_t* dr = ilv_out ? out_re + 2 * img + row_out * 8192 : out_re + img + row_out * 4096;
    uint16_t* di = ilv_out ? dr + 4096 : out_im + img + row_out * 4096;
    u4 vr[8], vi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (NT) {
        vr[i] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(sr + 512 * i + 8 * lane));
        vi[i] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(si + 512 * i + 8 * lane));
      } else {
        vr[i] = *reinterpret_cast<const u4*>(sr + 512 * i + 8 * lane);
        vi[i] = *reinterpret_cast<const u4*>(si + 512 * i + 8 * lane);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (NT) {
        __builtin_nontemporal_store(vr[i], reinterpret_cast<u4*>(dr + 512 * i + 8 * lane));
        __builtin_nontemporal_store(vi[i], reinterpret_cast<u4*>(di + 512 * i + 8 * lane));
      } else {
        *reinterpret_cast<u4*>(dr + 512 * i + 8 * lane) = vr[i];
        *reinterpret_cast<u4*>(di + 512 * i + 8 * lane) = vi[i];
      }
    }
  }
}

int main() {
  const uint64_t images = 64, plane = images * 4096 * 4096;      // halves per plane
  uint16_t *in, *out;
  hipMalloc(&in, 4 * plane);
  hipMalloc(&out, 4 * plane);
  hipMemset(in, 1, 4 * plane);
  const uint32_t iterations = images * 512;
  for (int nt = 0; nt < 2; ++nt)
    for (int adj_in = 0; adj_in < 2; ++adj_in)
      for (int mode = 0; mode < 3; ++mode) {
        const int adj_out = mode == 1, ilv_out = mode == 2;
        if (adj_in && ilv_out) continue;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        auto launch = [&] {
          if (nt) hipLaunchKernelGGL(k<true>, dim3(256), dim3(512), 0, 0, in, in + plane, out, out + plane, iterations, adj_in, adj_out, ilv_out);
          else hipLaunchKernelGGL(k<false>, dim3(256), dim3(512), 0, 0, in, in + plane, out, out + plane, iterations, adj_in, adj_out, ilv_out);
        };
        for (int w = 0; w < 20; ++w) launch();
        hipEventRecord(e0);
        const int reps = 10;
        for (int w = 0; w < reps; ++w) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= reps;
        printf("%s accesses, input rows %s, output rows %s: %8.1f us  %6.0f GB/s\n", nt ? "non-temporal" : "plain       ",
               adj_in ? "adjacent    " : "4 MiB apart ", ilv_out ? "[RE|IM] rows, 8 MiB apart" : (adj_out ? "adjacent    " : "4 MiB apart "), ms * 1e3, 4.0 * plane * 2 / ms * 1e-6);
      }
  return 0;
}
