// stride_bench2.hip — bandwidth of column-pass tiles as a function of the row-segment width: a workgroup tile is
// ROWS rows x SEG bytes per plane (64 KiB per plane in all cases), rows `pitch` bytes apart, in and out.
// SEG = 256: the radix-256 column kernel today; SEG = 128 / 64: what a 512- / 1024-row column kernel would move.
// `pair` = 1: adjacent column blocks go to workgroups blockIdx and blockIdx + 8 (same XCD) instead of blockIdx + 1.
// `pair` = 2: the rotated work distribution of the library (k4096::Rotor: round t, workgroup g -> tile t G + (g + t) mod G), so
// that no workgroup is pinned to one residue class of the tile index modulo 8.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/stride_bench2 tools/stride_bench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int SEG>
__global__ __launch_bounds__(512) void k(const uint8_t* in, uint8_t* out, uint64_t pitch, uint32_t blocks_per_entry,
                                        uint32_t total, uint64_t entry_bytes, uint64_t plane, int pair) {
  constexpr int ROWS = 65536 / SEG;            // rows per tile
  constexpr int RPI = 1024 / SEG;              // rows per wave instruction
  constexpr int LPR = SEG / 16;                // lanes per row
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t rnd = 0, pos = blockIdx.x;
  for (uint32_t t = blockIdx.x; t < total; ) {
    uint32_t blk = t;
    if (pair == 1) {   // t = 16 a + 8 b + c  ->  column block 16 a + 2 c + b: blocks 2c and 2c+1 run as t and t + 8
      const uint32_t a = t >> 4, b = (t >> 3) & 1, c = t & 7;
      blk = 16 * a + 2 * c + b;
    }
    const uint32_t e = blk / blocks_per_entry, cb = blk % blocks_per_entry;
    const uint8_t* src = in + e * entry_bytes + static_cast<uint64_t>(cb) * SEG;
    uint8_t* dst = out + e * entry_bytes + static_cast<uint64_t>(cb) * SEG;
    u4 vr[8], vi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint64_t r = (ROWS / 8) * wave + RPI * i + lane / LPR;
      vr[i] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(src + r * pitch + 16 * (lane % LPR)));
      vi[i] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(src + plane + r * pitch + 16 * (lane % LPR)));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint64_t r = (ROWS / 8) * wave + RPI * i + lane / LPR;
      __builtin_nontemporal_store(vr[i], reinterpret_cast<u4*>(dst + r * pitch + 16 * (lane % LPR)));
      __builtin_nontemporal_store(vi[i], reinterpret_cast<u4*>(dst + plane + r * pitch + 16 * (lane % LPR)));
    }
    ++rnd;
    if (pair == 2) {
      pos = (pos + 1 == gridDim.x) ? 0 : pos + 1;
      t = rnd * gridDim.x + pos;
    } else {
      t += gridDim.x;
    }
  }
}

template <int SEG>
void run(const uint8_t* in, uint8_t* out, uint64_t pitch, uint64_t plane_bytes, int pair, bool nt_off, uint64_t row_bytes = 0) {
  constexpr int ROWS = 65536 / SEG;
  const uint64_t entry = ROWS * pitch;                   // one transform: ROWS rows of pitch bytes
  const uint32_t bpe = static_cast<uint32_t>((row_bytes ? row_bytes : pitch) / SEG);
  const uint32_t entries = static_cast<uint32_t>((1ull << 30) / (row_bytes ? ROWS * row_bytes : entry));
  const uint32_t total = entries * bpe;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<SEG>, dim3(256), dim3(512), 0, 0, in, out, pitch, bpe, total, entry, plane_bytes, pair);
  hipEventRecord(e0);
  const int reps = 5;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<SEG>, dim3(256), dim3(512), 0, 0, in, out, pitch, bpe, total, entry, plane_bytes, pair);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("  seg %3d B x %4d rows, pitch %7llu B, %s: %8.1f us  %6.0f GB/s\n", SEG, ROWS, (unsigned long long)pitch,
         pair == 2 ? "rotated assignment" : (pair ? "paired on one XCD" : "neighbours on different XCDs"), ms * 1e3, 4.0 * total * 65536 / ms * 1e-6);
  (void)nt_off;
}

int main(int argc, char** argv) {
  const uint64_t plane = 1ull << 30;
  uint8_t *in, *out;
  hipMalloc(&in, 2 * plane + (64 << 20)); hipMalloc(&out, 2 * plane + (64 << 20));
  hipMemset(in, 1, 2 * plane);
  if (argc > 1) {
    // round 3: does a row pitch or a plane distance that is NOT a power of two change the ceiling? (2D column pass: rows 8 KiB
    // apart, planes 2^31 B apart in the planar interface; the intermediate image set is the library's own and could be padded)
    for (uint64_t pitch : {8192ull, 8192ull + 128, 8192ull + 256, 8192ull + 512, 8192ull + 2048}) {
      for (uint64_t pad : {0ull, 4096ull + 256, 1ull << 20}) {
        printf("plane distance 2^30 + %llu:", (unsigned long long)pad);
        run<128>(in, out, pitch, plane + pad, 2, false);
        printf("plane distance 2^30 + %llu:", (unsigned long long)pad);
        run<256>(in, out, pitch, plane + pad, 2, false);
      }
    }
    // the same tiles read from / written to an image set whose rows are [RE row | IM row] pairs: planes 8 KiB apart, row pitch 16 KiB
    printf("[RE|IM] row pairs (plane distance 8192 B, row pitch 16384 B):");
    run<128>(in, out, 16384, 8192, 2, false, 8192);
    printf("[RE|IM] row pairs (plane distance 8192 B, row pitch 16384 B):");
    run<256>(in, out, 16384, 8192, 2, false, 8192);
    return 0;
  }
  for (uint64_t pitch : {2048ull, 8192ull, 131072ull}) {
    for (int pair = 0; pair < 3; ++pair) {
      run<256>(in, out, pitch, plane, pair, false);
      run<128>(in, out, pitch, plane, pair, false);
      run<64>(in, out, pitch, plane, pair, false);
    }
  }
  return 0;
}
