// conc_bench.hip — how many waves per CU and how many CUs should move data at once? Round 4 found the 4-wave cooperative
// radix-256 column kernel FASTER with one workgroup per CU than with two (tools/exp_one_wave_per_simd.py: 5.4-5.7 vs 5.1-5.3 TB/s),
// and a copy with the 2D row pass's loop structure faster WITH s_sleep in it than without (tools/rows2d_sched.hip). This probe
// copies column-pass tiles (ROWS rows x SEG bytes per plane, rows `pitch` apart; the rotated work distribution of the library) with
// W waves per workgroup, `wg_per_cu` workgroups per CU (enforced through the dynamic LDS size) and `grid` workgroups in all.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/conc_bench tools/conc_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// one tile = TILE bytes per plane; every wave moves TILE / W bytes of each plane per tile, 1 KiB per instruction
template <int SEG, int W, int TILE>
__global__ __launch_bounds__(64 * W) void k(const uint8_t* in, uint8_t* out, uint64_t pitch, uint32_t blocks_per_entry, uint32_t total,
                                           uint64_t entry_bytes, uint64_t plane, uint32_t order) {
  extern __shared__ uint8_t lds[];
  constexpr int RPI = 1024 / SEG, LPR = SEG / 16, NI = TILE / W / 1024;      // instructions per wave, plane and tile
  constexpr int ROWS = TILE / SEG;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t rnd = 0, pos = blockIdx.x;
  for (uint32_t t = blockIdx.x; t < total;) {
    // order 0: consecutive tiles = consecutive column blocks of one entry (what the library launches); 1: consecutive tiles =
    // the same column block of consecutive entries; 2: column blocks in a stride-8 interleave (tile t -> block (t % 8) * bpe / 8 + t / 8)
    const uint32_t entries = total / blocks_per_entry;
    uint32_t e = t / blocks_per_entry, cb = t % blocks_per_entry;
    if (order == 1) {
      e = t % entries;
      cb = t / entries;
    } else if (order == 2) {
      cb = (cb & 7) * (blocks_per_entry / 8) + (cb >> 3);
    }
    const uint8_t* src = in + e * entry_bytes + static_cast<uint64_t>(cb) * SEG;
    uint8_t* dst = out + e * entry_bytes + static_cast<uint64_t>(cb) * SEG;
    u4 vr[NI], vi[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const uint64_t r = (ROWS / W) * wave + RPI * i + lane / LPR;
      vr[i] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(src + r * pitch + 16 * (lane % LPR)));
      vi[i] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(src + plane + r * pitch + 16 * (lane % LPR)));
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const uint64_t r = (ROWS / W) * wave + RPI * i + lane / LPR;
      __builtin_nontemporal_store(vr[i], reinterpret_cast<u4*>(dst + r * pitch + 16 * (lane % LPR)));
      __builtin_nontemporal_store(vi[i], reinterpret_cast<u4*>(dst + plane + r * pitch + 16 * (lane % LPR)));
    }
    ++rnd;
    pos = (pos + 1 == gridDim.x) ? 0 : pos + 1;
    t = rnd * gridDim.x + pos;
  }
  if (threadIdx.x == 9999) lds[0] = 1;
}

template <int SEG, int W, int TILE>
void run(const uint8_t* in, uint8_t* out, uint64_t pitch, uint64_t plane_bytes, int wg_per_cu, int grid, uint32_t order = 0) {
  constexpr int ROWS = TILE / SEG;
  const uint64_t entry = ROWS * pitch;
  const uint32_t bpe = static_cast<uint32_t>(pitch / SEG);
  const uint32_t entries = static_cast<uint32_t>((1ull << 30) / entry);
  const uint32_t total = entries * bpe;
  const int lds = wg_per_cu == 1 ? 96 * 1024 : (wg_per_cu == 2 ? 64 * 1024 : 36 * 1024);     // 160 KiB per CU
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<SEG, W, TILE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((k<SEG, W, TILE>), dim3(grid), dim3(64 * W), lds, 0, in, out, pitch, bpe, total, entry, plane_bytes, order);
  hipEventRecord(e0);
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<SEG, W, TILE>), dim3(grid), dim3(64 * W), lds, 0, in, out, pitch, bpe, total, entry, plane_bytes, order);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  printf("  seg %3d B, tile %3d KiB/plane, %d waves/WG, %d WG/CU, grid %4d (%2d waves per CU on %3d CUs), order %u: %8.1f us  %6.0f GB/s\n", SEG, TILE / 1024, W,
         wg_per_cu, grid, W * wg_per_cu, grid / wg_per_cu > 256 ? 256 : grid / wg_per_cu, order, ms * 1e3, 4.0 * total * TILE / ms * 1e-6);
}

int main(int argc, char** argv) {
  const uint64_t plane = 1ull << 30;
  uint8_t *in, *out;
  hipMalloc(&in, 2 * plane);
  hipMalloc(&out, 2 * plane);
  hipMemset(in, 1, 2 * plane);
  if (argc > 1) {
    // round 5: conc_bench PITCH_BYTES = the copy ceiling of a radix-512 / radix-1024 column pass with that row pitch (64-column
    // tiles of 512 or 1024 rows, one 8-wave workgroup per CU, static partition and two generations: what the library launches)
    const uint64_t pitch_arg = std::strtoull(argv[1], nullptr, 0);
    printf("row pitch %llu B, planes 1 GiB apart, 2 GiB moved per launch\n", (unsigned long long)pitch_arg);
    for (int grid : {256, 512}) {
      run<128, 8, 65536>(in, out, pitch_arg, plane, 1, grid);      // 512 rows x 128 B
      run<128, 8, 131072>(in, out, pitch_arg, plane, 1, grid);     // 1024 rows x 128 B
      run<256, 8, 65536>(in, out, pitch_arg, plane, 1, grid);      // 256 rows x 256 B
    }
    // (tile orders: does it matter WHICH tiles are in flight together?)
    for (uint32_t order : {1u, 2u}) {
      run<128, 8, 65536>(in, out, pitch_arg, plane, 1, 256, order);
      run<256, 8, 65536>(in, out, pitch_arg, plane, 1, 256, order);
    }
    return 0;
  }
  const uint64_t pitch = 8192;
  printf("row pitch %llu B, planes 1 GiB apart, 2 GiB moved per launch\n", (unsigned long long)pitch);
  for (int grid : {128, 192, 256}) {
    run<128, 4, 32768>(in, out, pitch, plane, 1, grid);
    run<128, 8, 65536>(in, out, pitch, plane, 1, grid);
    run<256, 4, 32768>(in, out, pitch, plane, 1, grid);
    run<256, 8, 65536>(in, out, pitch, plane, 1, grid);
  }
  for (int grid : {256, 384, 512}) {
    run<128, 4, 32768>(in, out, pitch, plane, 2, grid);
    run<256, 4, 32768>(in, out, pitch, plane, 2, grid);
    run<256, 8, 65536>(in, out, pitch, plane, 2, grid);
  }
  for (int grid : {512, 1024}) {
    run<128, 4, 32768>(in, out, pitch, plane, 4, grid);
    run<256, 4, 32768>(in, out, pitch, plane, 4, grid);
    run<256, 2, 32768>(in, out, pitch, plane, 4, grid);
  }
  run<256, 2, 32768>(in, out, pitch, plane, 1, 256);
  run<256, 2, 32768>(in, out, pitch, plane, 2, 512);
  run<128, 2, 32768>(in, out, pitch, plane, 1, 256);
  return 0;
}
