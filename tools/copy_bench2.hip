// copy_bench2.hip — which part of the N=4096 kernel's data-movement structure costs bandwidth?
// Copies 65536 "transforms" (8 KiB RE + 8 KiB IM each) between planar buffers with the kernel's exact
// work distribution (persistent grid, one wave per transform) while varying how the bytes are moved.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/copy_bench2 tools/copy_bench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

enum { LD_VGPR = 0, LD_DMA = 1 };

template <int LD, bool NTL, bool NTS, int WAVES, int DYN = 0>
__global__ __launch_bounds__(64 * WAVES) void k(const uint8_t* in_re, const uint8_t* in_im, uint8_t* out_re,
                                               uint8_t* out_im, uint32_t batch, int table_bytes, uint32_t* counter) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // optional constant-table fill (what a non-persistent FFT kernel would pay per workgroup)
  for (int i = threadIdx.x; i < table_bytes / 16; i += 64 * WAVES)
    reinterpret_cast<u4*>(lds)[i] = reinterpret_cast<const u4*>(in_re)[i];
  if (table_bytes) __syncthreads();
  uint8_t* wl = lds + table_bytes + wave * 16384;
  const uint32_t wl_off = __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t*)wl)));
  const uint32_t stride_b = gridDim.x * WAVES;
  uint32_t b = blockIdx.x * WAVES + wave;
  if (DYN) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(counter, 1u);
    b = __builtin_amdgcn_readfirstlane(t);
  }
  for (; b < batch;) {
    const uint8_t* sr = in_re + (size_t)b * 8192;
    const uint8_t* si = in_im + (size_t)b * 8192;
    uint8_t* dr = out_re + (size_t)b * 8192;
    uint8_t* di = out_im + (size_t)b * 8192;
    u4 vr[8], vi[8];
    if (LD == LD_DMA) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint8_t* gr = sr + 1024 * i + 16 * lane;
        const uint8_t* gi = si + 1024 * i + 16 * lane;
        const uint32_t d0 = wl_off + 1024 * i, d1 = wl_off + 8192 + 1024 * i;
        uint32_t keep;
        if (NTL)
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\t"
                       "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off nt\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(gr), "v"(gi), "s"(d0), "s"(d1) : "memory");
        else
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                       "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(gr), "v"(gi), "s"(d0), "s"(d1) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        vr[i] = *reinterpret_cast<const u4*>(wl + 1024 * i + 16 * lane);
        vi[i] = *reinterpret_cast<const u4*>(wl + 8192 + 1024 * i + 16 * lane);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const u4* pr = reinterpret_cast<const u4*>(sr + 1024 * i + 16 * lane);
        const u4* pi = reinterpret_cast<const u4*>(si + 1024 * i + 16 * lane);
        vr[i] = NTL ? __builtin_nontemporal_load(pr) : *pr;
        vi[i] = NTL ? __builtin_nontemporal_load(pi) : *pi;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      u4* pr = reinterpret_cast<u4*>(dr + 1024 * i + 16 * lane);
      u4* pi = reinterpret_cast<u4*>(di + 1024 * i + 16 * lane);
      if (NTS) { __builtin_nontemporal_store(vr[i], pr); __builtin_nontemporal_store(vi[i], pi); }
      else { *pr = vr[i]; *pi = vi[i]; }
    }
    if (DYN) {
      uint32_t t = 0;
      if (lane == 0) t = atomicAdd(counter, 1u);
      b = __builtin_amdgcn_readfirstlane(t);
    } else {
      b += stride_b;
    }
  }
}

template <int LD, bool NTL, bool NTS, int WAVES, int DYN = 0>
void run(const char* name, uint8_t* buf, uint32_t batch, int grid, int lds_bytes, int table_bytes = 0) {
  static uint32_t* counter = nullptr;
  if (!counter) hipMalloc(&counter, 4);
  const size_t plane = (size_t)batch * 8192;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<LD, NTL, NTS, WAVES, DYN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&] {
    if (DYN) hipMemsetAsync(counter, 0, 4, 0);
    hipLaunchKernelGGL((k<LD, NTL, NTS, WAVES, DYN>), dim3(grid), dim3(64 * WAVES), lds_bytes, 0, buf, buf + plane, buf + 2 * plane,
                       buf + 3 * plane, batch, table_bytes, counter);
  };
  for (int w = 0; w < 3; ++w) launch();
  hipEventRecord(e0);
  const int reps = 20;
  for (int w = 0; w < reps; ++w) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  printf("%-44s grid %5d lds %6d: %7.1f us  %6.0f GB/s\n", name, grid, lds_bytes, ms * 1e3, 4.0 * plane / ms * 1e-6);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
  const uint32_t batch = 65536;
  uint8_t* buf; hipMalloc(&buf, (size_t)batch * 8192 * 4);
  hipMemset(buf, 1, (size_t)batch * 8192 * 4);
  for (int rep = 0; rep < 2; ++rep) {
    run<LD_DMA, true, true, 8>("dma nt/nt 8w persistent static (as kernel)", buf, batch, 256, 163840, 32768);
    run<LD_DMA, true, true, 8, 1>("dma nt/nt 8w persistent dynamic (atomic)", buf, batch, 256, 163840, 32768);
    run<LD_DMA, true, true, 8>("dma nt/nt 8w one-shot, tables refilled", buf, batch, 8192, 163840, 32768);
    run<LD_DMA, true, true, 8>("dma nt/nt 8w one-shot, no tables", buf, batch, 8192, 131072, 0);
    run<LD_DMA, true, true, 8>("dma nt/nt 8w grid 2048 (4 iters), tables", buf, batch, 2048, 163840, 32768);
    run<LD_DMA, true, true, 8>("dma nt/nt 8w grid 1024 (8 iters), tables", buf, batch, 1024, 163840, 32768);
    run<LD_DMA, true, true, 4>("dma nt/nt 4w x2 blocks, tables 16K, static", buf, batch, 512, 81920, 16384);
    run<LD_DMA, true, true, 4, 1>("dma nt/nt 4w x2 blocks, tables 16K, dynamic", buf, batch, 512, 81920, 16384);
    run<LD_DMA, true, true, 4>("dma nt/nt 4w one-shot, tables 16K", buf, batch, 16384, 81920, 16384);
    run<LD_VGPR, true, true, 8, 1>("vgpr nt/nt 8w persistent dynamic", buf, batch, 256, 163840, 32768);
    printf("\n");
  }
  return 0;
}
