// probe_pk_hazard.hip — how many wait states does a consumer of a v_mfma_f32_16x16x32_f16 result need on gfx950 when the
// consumer is (a) a scalar fp32 VALU op, (b) a packed fp32 op (v_pk_mul_f32)?
//
// Background: with clang's SLP vectoriser on, the column kernels carried v_pk_*_f32 ops between MFMAs
// and one twiddled output was intermittently wrong; every MFMA -> consumer distance in that build is >= 8 wait
// states, which is what LLVM's gfx950 table asks for a 4-pass MFMA, for packed and scalar consumers alike
// (tools/isa_lint.py). This probe measures the distance the hardware really needs for each kind of consumer:
// MFMA, N wait states (s_nop), consumer, everything inside ONE asm statement so the compiler adds nothing.
//   hipcc -O2 --offload-arch=gfx950 -o tools/probe_pk_hazard tools/probe_pk_hazard.hip && tools/probe_pk_hazard
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// out[0], out[1]: 2 x acc[0], 2 x acc[1] as seen N wait states after the MFMA; out[2], out[3]: the same after it has
// certainly retired. The accumulator registers hold a sentinel before the MFMA, so an early read shows as the sentinel.
template <int N, bool PK>
__global__ void probe(float* out, float a_val, float b_val) {
  const h8 av = {(_Float16)a_val, (_Float16)a_val, (_Float16)a_val, (_Float16)a_val,
                 (_Float16)a_val, (_Float16)a_val, (_Float16)a_val, (_Float16)a_val};
  const _Float16 bb = (_Float16)(b_val + (threadIdx.x & 3));
  const h8 bv = {bb, bb, bb, bb, bb, bb, bb, bb};
  const float sentinel = 12345.0f;
  const float two = 2.0f;
  float e0, e1, l0, l1;
  if (PK) {
    asm volatile(
        "v_mov_b32 v100, %[s]\n\t"
        "v_mov_b32 v101, %[s]\n\t"
        "v_mov_b32 v102, %[s]\n\t"
        "v_mov_b32 v103, %[s]\n\t"
        "v_mov_b32 v106, %[two]\n\t"
        "v_mov_b32 v107, %[two]\n\t"
        "s_nop 15\n\t"
        "v_mfma_f32_16x16x32_f16 v[100:103], %[a], %[b], 0\n\t"
        "s_nop %c[n]\n\t"
        "v_pk_mul_f32 v[104:105], v[100:101], v[106:107]\n\t"
        "s_nop 15\n\t"
        "s_nop 15\n\t"
        "v_mul_f32 %[l0], v100, %[two]\n\t"
        "v_mul_f32 %[l1], v101, %[two]\n\t"
        "v_mov_b32 %[e0], v104\n\t"
        "v_mov_b32 %[e1], v105\n\t"
        : [e0] "=&v"(e0), [e1] "=&v"(e1), [l0] "=&v"(l0), [l1] "=&v"(l1)
        : [a] "v"(av), [b] "v"(bv), [s] "v"(sentinel), [two] "v"(two), [n] "i"(N - 1)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");
  } else {
    asm volatile(
        "v_mov_b32 v100, %[s]\n\t"
        "v_mov_b32 v101, %[s]\n\t"
        "v_mov_b32 v102, %[s]\n\t"
        "v_mov_b32 v103, %[s]\n\t"
        "s_nop 15\n\t"
        "v_mfma_f32_16x16x32_f16 v[100:103], %[a], %[b], 0\n\t"
        "s_nop %c[n]\n\t"
        "v_mul_f32 v104, v100, %[two]\n\t"
        "v_mul_f32 v105, v101, %[two]\n\t"
        "s_nop 15\n\t"
        "s_nop 15\n\t"
        "v_mul_f32 %[l0], v100, %[two]\n\t"
        "v_mul_f32 %[l1], v101, %[two]\n\t"
        "v_mov_b32 %[e0], v104\n\t"
        "v_mov_b32 %[e1], v105\n\t"
        : [e0] "=&v"(e0), [e1] "=&v"(e1), [l0] "=&v"(l0), [l1] "=&v"(l1)
        : [a] "v"(av), [b] "v"(bv), [s] "v"(sentinel), [two] "v"(two), [n] "i"(N - 1)
        : "v100", "v101", "v102", "v103", "v104", "v105");
  }
  const size_t t = (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 4;
  out[t] = e0;
  out[t + 1] = e1;
  out[t + 2] = l0;
  out[t + 3] = l1;
}

template <int N, bool PK>
void run(float* d, std::vector<float>& h, int blocks, int threads) {
  long bad0 = 0, bad1 = 0, total = 0;
  for (int rep = 0; rep < 20; ++rep) {
    hipLaunchKernelGGL((probe<N, PK>), dim3(blocks), dim3(threads), 0, 0, d, 0.5f, 1.0f + rep);
    (void)hipMemcpy(h.data(), d, h.size() * sizeof(float), hipMemcpyDeviceToHost);
    for (size_t i = 0; i < h.size(); i += 4) {
      bad0 += h[i] != h[i + 2];
      bad1 += h[i + 1] != h[i + 3];
      ++total;
    }
  }
  std::printf("%-6s consumer %2d wait states after the MFMA: first result register wrong in %8ld / %ld lanes, second in %8ld\n",
              PK ? "packed" : "scalar", N, bad0, total, bad1);
}

int main() {
  const int blocks = 2048, threads = 64;     // one wave per workgroup: waves land alone or paired on a SIMD
  std::vector<float> h(static_cast<size_t>(blocks) * threads * 4);
  float* d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(float)) != hipSuccess) return 2;
#define BOTH(N) run<N, false>(d, h, blocks, threads); run<N, true>(d, h, blocks, threads);
  BOTH(1) BOTH(2) BOTH(3) BOTH(4) BOTH(5) BOTH(6) BOTH(7) BOTH(8) BOTH(9) BOTH(10) BOTH(11) BOTH(12)
  (void)hipFree(d);
  return 0;
}
