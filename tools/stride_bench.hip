// stride_bench.hip — what does the memory system deliver for the column-pass access pattern? Each workgroup tile is
// 256 rows x 256 bytes per plane, rows `in_pitch` bytes apart on the way in and `out_pitch` bytes apart on the way
// out; adjacent tiles (adjacent 256-byte column blocks) run on different CUs at the same time, as in colfft.hpp.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/stride_bench tools/stride_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(const uint8_t* in, uint8_t* out, uint64_t in_pitch, uint64_t out_pitch,
                                        uint32_t blocks_per_entry, uint32_t total, uint64_t in_entry, uint64_t out_entry,
                                        uint64_t plane_in, uint64_t plane_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t blk = blockIdx.x; blk < total; blk += gridDim.x) {
    const uint32_t e = blk / blocks_per_entry, cb = blk % blocks_per_entry;
    const uint8_t* src = in + e * in_entry + cb * 256ull;
    uint8_t* dst = out + e * out_entry + cb * 256ull;
    u4 vr[8], vi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint64_t r = 32 * wave + 4 * i + (lane >> 4);
      vr[i] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(src + r * in_pitch + 16 * (lane & 15)));
      vi[i] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(src + plane_in + r * in_pitch + 16 * (lane & 15)));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint64_t r = 32 * wave + 4 * i + (lane >> 4);
      __builtin_nontemporal_store(vr[i], reinterpret_cast<u4*>(dst + r * out_pitch + 16 * (lane & 15)));
      __builtin_nontemporal_store(vi[i], reinterpret_cast<u4*>(dst + plane_out + r * out_pitch + 16 * (lane & 15)));
    }
  }
}

int main() {
  const uint64_t plane = 1ull << 30;   // 1 GiB per plane: the region the tiles cover (plus padding room)
  uint8_t *in, *out;
  hipMalloc(&in, 2 * plane + (256ull << 20)); hipMalloc(&out, 2 * plane + (256ull << 20));
  hipMemset(in, 1, 2 * plane + (256ull << 20));
  struct Case { const char* name; uint64_t in_pitch, out_pitch; };
  const Case cases[] = {
      {"contiguous tiles (pitch 256 B both sides)", 256, 256},
      {"2^20 pass 2: in 8 KiB, out 512 B x256.. (in 8 KiB, out 8 KiB)", 8192, 8192},
      {"in 8 KiB + 256 B pad, out 8 KiB + 256 B pad", 8192 + 256, 8192 + 256},
      {"2D column pass 1: in 128 KiB, out 8 KiB", 131072, 8192},
      {"in 128 KiB + 256 B pad, out 8 KiB", 131072 + 256, 8192},
      {"in 128 KiB + 4 KiB pad, out 8 KiB", 131072 + 4096, 8192},
      {"in 128 KiB + 256 B, out 8 KiB + 256 B", 131072 + 256, 8192 + 256},
      {"in 128 KiB, out 128 KiB (2^24 pass 2/3)", 131072, 131072},
      {"in 128 KiB + 256 B, out 128 KiB + 256 B", 131072 + 256, 131072 + 256},
      {"in 2 MiB, out 2 MiB", 2097152, 2097152},
  };
  for (const Case& c : cases) {
    // an "entry" is 256 rows of max(pitch) bytes; tiles per entry = pitch / 256 (the row is pitch bytes wide)
    const uint64_t row_in = c.in_pitch & ~255ull ? (c.in_pitch / 256) : 1;
    const uint32_t bpe = static_cast<uint32_t>(c.in_pitch >= 512 ? (c.in_pitch & ~(c.in_pitch - 1) ? (1ull << (63 - __builtin_clzll(c.in_pitch))) / 256 : 1) : 1);
    (void)row_in;
    const uint64_t in_entry = 256 * c.in_pitch, out_entry = 256 * c.out_pitch;
    const uint64_t entry_span = in_entry > out_entry ? in_entry : out_entry;
    uint32_t entries = static_cast<uint32_t>(plane / entry_span);
    if (entries == 0) entries = 1;
    // tiles per entry limited so that output rows (out_pitch wide) can hold them too
    uint32_t bpe_out = static_cast<uint32_t>(c.out_pitch >= 512 ? (1ull << (63 - __builtin_clzll(c.out_pitch))) / 256 : 1);
    const uint32_t b = bpe < bpe_out ? bpe : bpe_out;
    const uint32_t total = entries * b;
    const double bytes = 4.0 * total * 65536;   // 2 planes x 64 KiB in + out
    printf("%-66s tiles %7u (%4u/entry):", c.name, total, b);
    for (uint32_t iters : {1000000u, 8u, 4u, 2u, 1u}) {
      uint32_t grid = (total + iters - 1) / iters;
      if (grid < 256) grid = 256 < total ? 256 : total;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, in, out, c.in_pitch, c.out_pitch, b, total, in_entry, out_entry, plane + (128ull << 20), plane + (128ull << 20));
      hipEventRecord(e0);
      const int reps = 5;
      for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, in, out, c.in_pitch, c.out_pitch, b, total, in_entry, out_entry, plane + (128ull << 20), plane + (128ull << 20));
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
      printf("  it%-7u %5.0f", iters, bytes / ms * 1e-6);
    }
    printf("  GB/s\n");
  }
  return 0;
}
