// lat_probe.hip — what ONE dependent pass over a few MiB costs on MI355X, piece by piece (round 5: single-transform latency).
//   1. chains of empty kernels in a HIP graph (grid, block, dynamic LDS): the floor of a dependent launch
//   2. chains of tile-copy kernels with the column passes' access pattern (ROWS rows of SEG bytes, row pitch = plane / ROWS, out
//      contiguous per tile), every workgroup ONE tile, with wall-clock stamps per workgroup: kernel span, gap to the next kernel,
//      time from entry to "loads landed", to exit
//   3. the same passes inside ONE launch, separated by a grid barrier (every workgroup resident, one per CU): write-through (sc1)
//      stores + sc1 loads + one atomic counter, or plain accesses with release / acquire fences
// build: hipcc -O3 --offload-arch=gfx950 -o tools/lat_probe tools/lat_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

struct Stamp {
  unsigned long long t0, t1, t2;
};

__global__ void k_empty(int* p) {
  extern __shared__ uint8_t lds[];
  if (p && threadIdx.x == 99999) *p = lds[0];
}

// one tile per workgroup; chunk c of the tile (16 bytes): row c / (SEG / 16), column chunk c % (SEG / 16)
template <int THREADS, int ROWS, int SEG>
__device__ __forceinline__ void tile_addr(uint32_t tile, uint32_t c, uint32_t plane_bytes, uint32_t& in_off, uint32_t& out_off) {
  constexpr uint32_t kCpr = SEG / 16;
  const uint32_t pitch = plane_bytes / ROWS;
  const uint32_t tiles_per_row = pitch / SEG;
  // (for a plane larger than one ROWS x pitch matrix there are several matrices one after the other: batch)
  const uint32_t mat = tile / tiles_per_row, tcol = tile % tiles_per_row;
  in_off = mat * plane_bytes + (c / kCpr) * pitch + tcol * SEG + (c % kCpr) * 16;
  out_off = tile * (ROWS * SEG) + c * 16;
}

template <int THREADS, int ROWS, int SEG>
__global__ __launch_bounds__(THREADS) void tile_copy(const uint8_t* in, uint8_t* out, uint32_t plane_bytes, uint32_t planes_dist,
                                                     Stamp* st) {
  constexpr int kN = ROWS * SEG / 16 / THREADS;      // loads per plane and thread
  const unsigned long long t0 = (unsigned long long)wall_clock64();
  u4 v[2][kN];
  uint32_t oo[kN];
#pragma unroll
  for (int j = 0; j < kN; ++j) {
    uint32_t io;
    tile_addr<THREADS, ROWS, SEG>(blockIdx.x, j * THREADS + threadIdx.x, plane_bytes, io, oo[j]);
    v[0][j] = *reinterpret_cast<const u4*>(in + io);
    v[1][j] = *reinterpret_cast<const u4*>(in + planes_dist + io);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = (unsigned long long)wall_clock64();
#pragma unroll
  for (int j = 0; j < kN; ++j) {
    *reinterpret_cast<u4*>(out + oo[j]) = v[0][j];
    *reinterpret_cast<u4*>(out + planes_dist + oo[j]) = v[1][j];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (st && threadIdx.x == 0) st[blockIdx.x] = Stamp{t0, t1, (unsigned long long)wall_clock64()};
}

// ---- the same passes in one launch
__device__ __forceinline__ u4 ld_sc1(const uint8_t* p) {
  u4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// (the s_nop: hipcc pads nothing behind inline asm, and a VALU write of the store's data registers right behind a store of more than
// 8 bytes is a hazard: the first version of this probe stored garbage without it)
__device__ __forceinline__ void st_sc1(uint8_t* p, u4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_sc01(uint8_t* p, u4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }

// cache policies of a pass: LD 0 plain, 1 nt, 2 sc1; ST 0 plain, 1 nt, 2 sc1 (write-through), 3 sc0 sc1
template <int THREADS, int ROWS, int SEG, int LD, int ST>
__global__ __launch_bounds__(THREADS) void tile_copy_pol(const uint8_t* in, uint8_t* out, uint32_t plane_bytes, uint32_t planes_dist, Stamp* st) {
  constexpr int kN = ROWS * SEG / 16 / THREADS;
  const unsigned long long t0 = (unsigned long long)wall_clock64();
  u4 v[2][kN];
  uint32_t oo[kN];
#pragma unroll
  for (int j = 0; j < kN; ++j) {
    uint32_t io;
    tile_addr<THREADS, ROWS, SEG>(blockIdx.x, j * THREADS + threadIdx.x, plane_bytes, io, oo[j]);
    if (LD == 0) {
      v[0][j] = *reinterpret_cast<const u4*>(in + io);
      v[1][j] = *reinterpret_cast<const u4*>(in + planes_dist + io);
    } else if (LD == 1) {
      v[0][j] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(in + io));
      v[1][j] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(in + planes_dist + io));
    } else {
      v[0][j] = ld_sc1(in + io);
      v[1][j] = ld_sc1(in + planes_dist + io);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = (unsigned long long)wall_clock64();
#pragma unroll
  for (int j = 0; j < kN; ++j) {
    if (ST == 0) {
      *reinterpret_cast<u4*>(out + oo[j]) = v[0][j];
      *reinterpret_cast<u4*>(out + planes_dist + oo[j]) = v[1][j];
    } else if (ST == 1) {
      __builtin_nontemporal_store(v[0][j], reinterpret_cast<u4*>(out + oo[j]));
      __builtin_nontemporal_store(v[1][j], reinterpret_cast<u4*>(out + planes_dist + oo[j]));
    } else if (ST == 2) {
      st_sc1(out + oo[j], v[0][j]);
      st_sc1(out + planes_dist + oo[j], v[1][j]);
    } else {
      st_sc01(out + oo[j], v[0][j]);
      st_sc01(out + planes_dist + oo[j], v[1][j]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (st && threadIdx.x == 0) st[blockIdx.x] = Stamp{t0, t1, (unsigned long long)wall_clock64()};
}

// instruction fetch: a straight-line body of KB KiB (4-byte VALU instructions), run TWICE inside one launch by one wave: cycles of
// the first (cold?) and of the second (warm) run, launch after launch: is the instruction cache kept across dispatches?
template <int KB>
__global__ __launch_bounds__(64) void k_icache(unsigned long long* out, unsigned* sink) {
  unsigned x = threadIdx.x;
  unsigned long long t[3];
  t[0] = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 2; ++it) {
    asm volatile(".rept %1\n\tv_add_u32 %0, 1, %0\n\t.endr" : "+v"(x) : "n"(KB * 256));
    asm volatile("s_nop 0" ::: "memory");
    t[it + 1] = __builtin_amdgcn_s_memtime();
  }
  if (threadIdx.x == 0) {
    out[0] = t[1] - t[0];
    out[1] = t[2] - t[1];
  }
  if (x == 0xffffffffu) *sink = x;
}


// MODE 0: sc1 stores, sc1 loads, counter.  MODE 1: plain stores + release fence, counter, acquire fence + plain loads.
template <int THREADS, int ROWS, int SEG, int MODE>
__global__ __launch_bounds__(THREADS) void fused_phases(uint8_t* a, uint8_t* b, uint32_t plane_bytes, uint32_t planes_dist, int phases,
                                                        unsigned* counter, unsigned* timeout, Stamp* st) {
  constexpr int kN = ROWS * SEG / 16 / THREADS;
  const unsigned long long t0 = (unsigned long long)wall_clock64();
  unsigned long long t1 = 0;
  uint8_t *src = a, *dst = b;
  for (int ph = 0; ph < phases; ++ph) {
    u4 v[2][kN];
    uint32_t oo[kN];
#pragma unroll
    for (int j = 0; j < kN; ++j) {
      uint32_t io;
      tile_addr<THREADS, ROWS, SEG>(blockIdx.x, j * THREADS + threadIdx.x, plane_bytes, io, oo[j]);
      if (MODE == 0 && ph > 0) {
        v[0][j] = ld_sc1(src + io);
        v[1][j] = ld_sc1(src + planes_dist + io);
      } else {
        v[0][j] = *reinterpret_cast<const u4*>(src + io);
        v[1][j] = *reinterpret_cast<const u4*>(src + planes_dist + io);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < kN; ++j) {
      if (MODE == 0) {
        st_sc1(dst + oo[j], v[0][j]);
        st_sc1(dst + planes_dist + oo[j], v[1][j]);
      } else {
        *reinterpret_cast<u4*>(dst + oo[j]) = v[0][j];
        *reinterpret_cast<u4*>(dst + planes_dist + oo[j]) = v[1][j];
      }
    }
    if (ph + 1 == phases) break;
    // ---- grid barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      if (MODE == 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = gridDim.x * (ph + 1);
      unsigned spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 4000000u) {
          *timeout = 1;
          break;
        }
      }
      if (MODE == 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    if (ph == 0) t1 = (unsigned long long)wall_clock64();
    uint8_t* t = src;
    src = dst;
    dst = t;
    if (ph == 0) src = b, dst = a;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) {
    if (st) st[blockIdx.x] = Stamp{t0, t1, (unsigned long long)wall_clock64()};
    // the last workgroup to leave re-arms the barrier for the next launch (every workgroup has passed every barrier by then)
    if (__hip_atomic_fetch_add(counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// apply one pass to chunk ids on the host
template <int THREADS, int ROWS, int SEG>
static void host_pass(const std::vector<uint32_t>& in, std::vector<uint32_t>& out, uint32_t plane_bytes, uint32_t tiles) {
  const uint32_t chunks_per_tile = ROWS * SEG / 16;
  const uint32_t kCpr = SEG / 16, pitch = plane_bytes / ROWS, tpr = pitch / SEG;
  for (uint32_t t = 0; t < tiles; ++t)
    for (uint32_t c = 0; c < chunks_per_tile; ++c) {
      const uint32_t mat = t / tpr, tcol = t % tpr;
      const uint32_t io = mat * plane_bytes + (c / kCpr) * pitch + tcol * SEG + (c % kCpr) * 16;
      const uint32_t oo = t * (ROWS * SEG) + c * 16;
      out[oo / 16] = in[io / 16];
    }
}

struct Ctx {
  uint8_t *a, *b;
  Stamp* st;
  unsigned *counter, *timeout;
  hipStream_t s;
  hipEvent_t e0, e1;
};

template <typename F>
static float time_graph(Ctx& c, int reps, F&& body) {   // body enqueues on c.s; returns us per graph launch
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(c.s, hipStreamCaptureModeGlobal);
  body();
  hipStreamEndCapture(c.s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, c.s);
  hipStreamSynchronize(c.s);
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(c.e0, c.s);
    for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, c.s);
    hipEventRecord(c.e1, c.s);
    hipEventSynchronize(c.e1);
    float ms;
    hipEventElapsedTime(&ms, c.e0, c.e1);
    best = std::min(best, ms * 1000.f / reps);
  }
  hipGraphExecDestroy(ge);
  hipGraphDestroy(g);
  return best;
}

template <int THREADS, int ROWS, int SEG>
static int run_tiles(Ctx& c, uint32_t plane_bytes, uint32_t mats, const char* what) {
  const uint32_t total = plane_bytes * mats;                 // bytes per plane over the batch
  const uint32_t tiles = total / (ROWS * SEG);
  const uint32_t planes_dist = total;
  constexpr int K = 16;
  // ---- separate launches: K kernels, a -> b -> a ...; stamps of the last two
  std::vector<Stamp> s0(tiles), s1(tiles);
  const float us = time_graph(c, 8, [&] {
    for (int k = 0; k < K; ++k)
      hipLaunchKernelGGL((tile_copy<THREADS, ROWS, SEG>), dim3(tiles), dim3(THREADS), 0, c.s, (k & 1) ? c.b : c.a, (k & 1) ? c.a : c.b,
                         plane_bytes, planes_dist, k >= K - 2 ? c.st + (k - (K - 2)) * 4096 : nullptr);
  });
  CK(hipMemcpy(s0.data(), c.st, tiles * sizeof(Stamp), hipMemcpyDeviceToHost));
  CK(hipMemcpy(s1.data(), c.st + 4096, tiles * sizeof(Stamp), hipMemcpyDeviceToHost));
  auto mn = [&](const std::vector<Stamp>& s, int f) {
    unsigned long long m = ~0ull;
    for (auto& x : s) m = std::min(m, f == 0 ? x.t0 : (f == 1 ? x.t1 : x.t2));
    return m;
  };
  auto mx = [&](const std::vector<Stamp>& s, int f) {
    unsigned long long m = 0;
    for (auto& x : s) m = std::max(m, f == 0 ? x.t0 : (f == 1 ? x.t1 : x.t2));
    return m;
  };
  double wg = 0, ld = 0;
  for (auto& x : s1) wg += (x.t2 - x.t0) * 0.01, ld += (x.t1 - x.t0) * 0.01;
  printf("%-44s %4u WGs x %3d thr: %6.2f us/kernel in a chain of %d | span %5.2f us (entry spread %4.2f) gap to next %5.2f us | per WG: "
         "%5.2f us, loads landed after %5.2f us\n",
         what, tiles, THREADS, us / K, K, (mx(s1, 2) - mn(s1, 0)) * 0.01, (mx(s1, 0) - mn(s1, 0)) * 0.01,
         (double)((long long)mn(s1, 0) - (long long)mx(s0, 2)) * 0.01, wg / tiles, ld / tiles);
  // ---- one launch, grid barrier between the passes (only when every workgroup is resident: one per CU)
  if (tiles <= 256) {
    for (int mode = 0; mode < 2; ++mode) {
      for (int phases : {1, 2, 4}) {
        // correctness first (ids), then time
        const uint32_t nchunk = total / 16;
        std::vector<uint32_t> h(nchunk * 4 * 2), ref(nchunk), tmp(nchunk);
        for (uint32_t i = 0; i < nchunk; ++i) ref[i] = i;
        for (uint32_t pl = 0; pl < 2; ++pl)
          for (uint32_t i = 0; i < nchunk; ++i)
            for (int w = 0; w < 4; ++w) h[(pl * nchunk + i) * 4 + w] = i * 8 + pl * 4 + w;
        CK(hipMemcpy(c.a, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(c.b, 0xff, h.size() * 4));
        CK(hipMemset(c.counter, 0, 64));
        auto launch = [&] {
          if (mode == 0)
            hipLaunchKernelGGL((fused_phases<THREADS, ROWS, SEG, 0>), dim3(tiles), dim3(THREADS), 0, c.s, c.a, c.b, plane_bytes, planes_dist,
                               phases, c.counter, c.timeout, c.st);
          else
            hipLaunchKernelGGL((fused_phases<THREADS, ROWS, SEG, 1>), dim3(tiles), dim3(THREADS), 0, c.s, c.a, c.b, plane_bytes, planes_dist,
                               phases, c.counter, c.timeout, c.st);
        };
        launch();
        CK(hipStreamSynchronize(c.s));
        unsigned tmo = 0;
        CK(hipMemcpy(&tmo, c.timeout, 4, hipMemcpyDeviceToHost));
        if (tmo) {
          printf("   fused mode %d phases %d: BARRIER TIMEOUT\n", mode, phases);
          CK(hipMemset(c.timeout, 0, 4));
          continue;
        }
        for (int p = 0; p < phases; ++p) {
          host_pass<THREADS, ROWS, SEG>(ref, tmp, plane_bytes, tiles);
          ref.swap(tmp);
        }
        // result buffer: phase p writes b for even p, a for odd p
        CK(hipMemcpy(h.data(), (phases & 1) ? c.b : c.a, h.size() * 4, hipMemcpyDeviceToHost));
        uint64_t bad = 0;
        for (uint32_t pl = 0; pl < 2; ++pl)
          for (uint32_t i = 0; i < nchunk; ++i)
            for (int w = 0; w < 4; ++w) bad += h[(pl * nchunk + i) * 4 + w] != ref[i] * 8 + pl * 4 + w;
        const float usf = time_graph(c, 8, [&] {
          for (int k = 0; k < 4; ++k) launch();
        });
        printf("   one launch, %s, %d pass%s: %6.2f us per launch = %5.2f us per pass%s\n",
               mode == 0 ? "sc1 stores + sc1 loads + counter " : "plain + release / acquire fences ", phases, phases > 1 ? "es" : "  ", usf / 4,
               usf / 4 / phases, bad ? "   ** WRONG DATA **" : "");
        if (bad) printf("      (%llu wrong words)\n", (unsigned long long)bad);
      }
    }
  }
  return 0;
}

template <int THREADS, int ROWS, int SEG, int LD, int ST>
static void run_pol(Ctx& c, uint32_t plane_bytes, const char* what) {
  const uint32_t tiles = plane_bytes / (ROWS * SEG);
  constexpr int K = 16;
  std::vector<Stamp> s0(tiles), s1(tiles);
  const float us = time_graph(c, 8, [&] {
    for (int k = 0; k < K; ++k)
      hipLaunchKernelGGL((tile_copy_pol<THREADS, ROWS, SEG, LD, ST>), dim3(tiles), dim3(THREADS), 0, c.s, (k & 1) ? c.b : c.a, (k & 1) ? c.a : c.b,
                         plane_bytes, plane_bytes, k >= K - 2 ? c.st + (k - (K - 2)) * 4096 : nullptr);
  });
  hipMemcpy(s0.data(), c.st, tiles * sizeof(Stamp), hipMemcpyDeviceToHost);
  hipMemcpy(s1.data(), c.st + 4096, tiles * sizeof(Stamp), hipMemcpyDeviceToHost);
  unsigned long long e0 = 0, b1 = ~0ull, e1 = 0;
  double ld = 0, wg = 0;
  for (auto& x : s0) e0 = std::max(e0, x.t2);
  for (auto& x : s1) b1 = std::min(b1, x.t0), e1 = std::max(e1, x.t2), ld += (x.t1 - x.t0) * 0.01, wg += (x.t2 - x.t0) * 0.01;
  printf("   %-34s: %6.2f us/kernel | span %5.2f gap %5.2f | per WG %5.2f us, loads landed after %5.2f us\n", what, us / K, (e1 - b1) * 0.01,
         (double)((long long)b1 - (long long)e0) * 0.01, wg / tiles, ld / tiles);
}

template <int THREADS, int ROWS, int SEG>
static void run_policies(Ctx& c, uint32_t plane_bytes, const char* shape) {
  printf("-- cache policies, %s, plane %u KiB (%u workgroups)\n", shape, plane_bytes >> 10, plane_bytes / (ROWS * SEG));
  run_pol<THREADS, ROWS, SEG, 0, 0>(c, plane_bytes, "plain loads, plain stores");
  run_pol<THREADS, ROWS, SEG, 1, 1>(c, plane_bytes, "nt loads, nt stores");
  run_pol<THREADS, ROWS, SEG, 0, 1>(c, plane_bytes, "plain loads, nt stores");
  run_pol<THREADS, ROWS, SEG, 0, 2>(c, plane_bytes, "plain loads, sc1 stores");
  run_pol<THREADS, ROWS, SEG, 0, 3>(c, plane_bytes, "plain loads, sc0 sc1 stores");
  run_pol<THREADS, ROWS, SEG, 2, 2>(c, plane_bytes, "sc1 loads, sc1 stores");
  run_pol<THREADS, ROWS, SEG, 1, 2>(c, plane_bytes, "nt loads, sc1 stores");
}

template <int KB>
static void run_icache(Ctx& c) {
  unsigned long long* d;
  hipMalloc(&d, 4 * 16);
  hipStreamBeginCapture(c.s, hipStreamCaptureModeGlobal);
  for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_icache<KB>, dim3(1), dim3(64), 0, c.s, d + 2 * k, c.counter + 8);
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamEndCapture(c.s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  unsigned long long h[8];
  for (int r = 0; r < 3; ++r) {
    hipGraphLaunch(ge, c.s);
    hipStreamSynchronize(c.s);
  }
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("straight-line body of %3d KiB (%5d instructions), cycles first / second run in the launch:", KB, KB * 256);
  for (int k = 0; k < 4; ++k) printf("  %llu / %llu", h[2 * k], h[2 * k + 1]);
  printf("\n");
  hipGraphExecDestroy(ge);
  hipGraphDestroy(g);
  hipFree(d);
}

int main(int argc, char** argv) {
  Ctx c;
  const size_t kBuf = 64u << 20;
  CK(hipMalloc(&c.a, kBuf));
  CK(hipMalloc(&c.b, kBuf));
  CK(hipMalloc(&c.st, 2 * 4096 * sizeof(Stamp)));
  CK(hipMalloc(&c.counter, 256));
  CK(hipMalloc(&c.timeout, 256));
  CK(hipMemset(c.a, 1, kBuf));
  CK(hipMemset(c.b, 2, kBuf));
  CK(hipMemset(c.timeout, 0, 256));
  CK(hipStreamCreate(&c.s));
  CK(hipEventCreate(&c.e0));
  CK(hipEventCreate(&c.e1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_empty), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

  const bool part2 = argc > 1 && !strcmp(argv[1], "policies");
  if (part2) {
    printf("== 4. instruction fetch (four launches of the same kernel back to back in a graph, third replay)\n");
    run_icache<4>(c);
    run_icache<16>(c);
    run_icache<48>(c);
    printf("== 5. cache policies of a tile-copy pass (chains of 16 dependent launches)\n");
    for (uint32_t lg : {16u, 18u, 20u}) {
      run_policies<256, 256, 128>(c, 2u << lg, "256 rows x 128 B, 4 waves");
      run_policies<64, 256, 32>(c, 2u << lg, "256 rows x 32 B, one wave");
    }
    run_policies<512, 256, 256>(c, 2u << 22, "256 rows x 256 B, 8 waves");
    return 0;
  }
  printf("== 1. chains of 16 empty kernels in a graph (us per kernel)\n");
  for (int grid : {1, 64, 256, 1024})
    for (int block : {64, 256, 512})
      for (int lds : {0, 80 * 1024, 160 * 1024}) {
        const float us = time_graph(c, 8, [&] {
          for (int k = 0; k < 16; ++k) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(block), lds, c.s, nullptr);
        });
        printf("grid %4d block %3d LDS %6d: %5.2f us\n", grid, block, lds, us / 16);
      }

  printf("== 2./3. tile-copy passes: one tile per workgroup, planar RE | IM, 4 B per complex sample\n");
  // N = 2^16 (128 KiB per plane), 2^18, 2^20 (2 MiB per plane), 2^22
  for (uint32_t lg : {16u, 18u, 20u, 22u}) {
    const uint32_t plane = 2u << lg;
    printf("-- one transform of 2^%u (plane %u KiB)\n", lg, plane >> 10);
    if (lg <= 20) {
      if (run_tiles<64, 256, 32>(c, plane, 1, "256 rows x 32 B (16 columns, one wave)")) return 1;
      if (run_tiles<256, 256, 128>(c, plane, 1, "256 rows x 128 B (64 columns, 4 waves)")) return 1;
    }
    if (lg >= 18 && run_tiles<512, 256, 256>(c, plane, 1, "256 rows x 256 B (128 columns, 8 waves)")) return 1;
    if (run_tiles<256, 64, 128>(c, plane, 1, "64 rows x 128 B (4 waves)")) return 1;
    if (run_tiles<64, 16, 512>(c, plane, 1, "16 rows x 512 B (one wave)")) return 1;
    if (run_tiles<256, 16, 2048>(c, plane, 1, "16 rows x 2 KiB (4 waves)")) return 1;
  }
  return 0;
}
