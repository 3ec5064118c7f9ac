// rows2d_sched.hip — which placement of the 32 vector-memory instructions per wave and iteration suits the fused 2D row pass
// (k4096r.hpp, ROWS)? A copy kernel with that pass's access pattern (8 rows r0 + 512 i of a 4096 x 4096 image per workgroup
// iteration, 8 KiB per row and plane, one 8-wave workgroup per CU, rotated iteration order) and its loop structure (front end /
// barrier B / stages / read-back / barrier D), the arithmetic replaced by s_sleep of the same length, so that schedules can
// be compared without fighting the compiler over the real kernel:
//   mode 0  burst:   [FE sleep] B [16 loads] [stage sleep] [16 stores] D                      (the round-3 kernel)
//   mode 1  loads spread over the 16 pieces of the stage sleep, stores in one burst
//   mode 2  loads spread over the stage pieces, stores (of the previous iteration) spread over the FE pieces of the next
//   mode 3  every piece of both phases issues one load or one store, alternating (8 + 8 in each phase)
//   mode 4  as 3 but the two halves of the workgroup (waves 0-3 / 4-7) alternate in opposite order
// build: hipcc -O3 --offload-arch=gfx950 -o tools/rows2d_sched tools/rows2d_sched.hip ; run: tools/rows2d_sched [fe_cycles stage_cycles]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void nap(int n) {          // ~64 n cycles without touching any execution unit
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
}

template <int MODE, bool BAR>
__global__ __launch_bounds__(512) void k(const uint16_t* in_re, const uint16_t* in_im, uint16_t* out_re, uint16_t* out_im,
                                        uint32_t iterations, int fe_piece, int st_piece) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t rnd = 0, pos = blockIdx.x;
  u4 a[16], b[16];
  auto src = [&](uint32_t it, int i) {
    const uint64_t img = static_cast<uint64_t>(it >> 9) * 4096 * 4096;
    const uint64_t row = (it & 511) + 512u * wave;
    return ((i & 1) ? in_im : in_re) + img + row * 4096 + 512 * (i >> 1) + 8 * lane;
  };
  auto dst = [&](uint32_t it, int i) {
    const uint64_t img = static_cast<uint64_t>(it >> 9) * 4096 * 4096;
    const uint64_t row = (it & 511) + 512u * wave;
    return ((i & 1) ? out_im : out_re) + img + row * 4096 + 512 * (i >> 1) + 8 * lane;
  };
  uint32_t it = blockIdx.x;
  if (it >= iterations) return;
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = *reinterpret_cast<const u4*>(src(it, i));
  uint32_t prev = it;
  bool have_prev = false;
  for (;;) {
    const uint32_t npos = (pos + 1 == gridDim.x) ? 0 : pos + 1;
    const uint32_t nxt_raw = (rnd + 1) * gridDim.x + npos;
    const uint32_t nxt = nxt_raw < iterations ? nxt_raw : it;
    // ---- front end
    u4 c[16];
    const bool swapped = (MODE == 4) && (wave >= 4);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      nap(fe_piece);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 2 && have_prev) *reinterpret_cast<u4*>(dst(prev, i)) = b[i];
      if (MODE == 3 || MODE == 4) {
        if (((i & 1) == 0) != swapped) { if (have_prev) *reinterpret_cast<u4*>(dst(prev, i / 2)) = b[i / 2]; }
        else c[i / 2] = *reinterpret_cast<const u4*>(src(nxt, i / 2));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (BAR) __builtin_amdgcn_s_barrier();      // B
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) b[i] = *reinterpret_cast<const u4*>(src(nxt, i));
    }
    // ---- stages
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      nap(st_piece);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 1 || MODE == 2) c[i] = *reinterpret_cast<const u4*>(src(nxt, i));
      if (MODE == 3 || MODE == 4) {
        if (((i & 1) == 0) != swapped) c[8 + i / 2] = *reinterpret_cast<const u4*>(src(nxt, 8 + i / 2));
        else if (have_prev) *reinterpret_cast<u4*>(dst(prev, 8 + i / 2)) = b[8 + i / 2];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- read-back / stores
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) *reinterpret_cast<u4*>(dst(it, i)) = a[i];
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = b[i];
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) *reinterpret_cast<u4*>(dst(it, i)) = a[i];
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = c[i];
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) { b[i] = a[i]; a[i] = c[i]; }
      prev = it;
      have_prev = true;
    }
    if (BAR) __builtin_amdgcn_s_barrier();      // D
    if (nxt_raw >= iterations) break;
    it = nxt_raw;
    ++rnd;
    pos = npos;
  }
  if (MODE >= 2 && have_prev) {
#pragma unroll
    for (int i = 0; i < 16; ++i) *reinterpret_cast<u4*>(dst(prev, i)) = b[i];
  }
}

int main(int argc, char** argv) {
  const uint64_t images = 64, plane = images * 4096 * 4096;      // halves per plane
  uint16_t *in, *out;
  hipMalloc(&in, 4 * plane);
  hipMalloc(&out, 4 * plane);
  hipMemset(in, 1, 4 * plane);
  const uint32_t iterations = images * 512;
  const int fe = argc > 1 ? atoi(argv[1]) : 4000, st = argc > 2 ? atoi(argv[2]) : 5800;
  const int fe_piece = fe / 16 / 64, st_piece = st / 16 / 64;
  printf("front end %d cycles, stages %d cycles (s_sleep pieces of %d and %d x 64 cycles)\n", fe, st, fe_piece, st_piece);
  for (int mode = 0; mode < 10; ++mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
#define L(M, B) hipLaunchKernelGGL((k<M, B>), dim3(256), dim3(512), 0, 0, in, in + plane, out, out + plane, iterations, fe_piece, st_piece)
    auto launch = [&] {
      switch (mode) {
        case 0: L(0, true); break;
        case 1: L(1, true); break;
        case 2: L(2, true); break;
        case 3: L(3, true); break;
        case 4: L(4, true); break;
        case 5: L(0, false); break;
        case 6: L(1, false); break;
        case 7: L(2, false); break;
        case 8: L(3, false); break;
        default: L(4, false); break;
      }
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
    printf("mode %d%s: %8.1f us  %6.0f GB/s\n", mode % 5, mode >= 5 ? " without barriers" : "", ms * 1e3, 4.0 * plane * 2 / ms * 1e-6);
  }
  return 0;
}
