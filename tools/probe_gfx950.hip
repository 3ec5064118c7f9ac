// probe_gfx950.hip — checks, on a real MI355X, the instruction semantics k4096.hpp relies on
// (operand/accumulator lane maps of v_mfma_f32_16x16x32_f16, ds_read_b64_tr_b16,
// v_permlane16_swap / v_permlane32_swap, global_load_lds_dwordx4). Diagnostic only.
// build: hipcc -O2 --offload-arch=gfx950 -o tools/probe_gfx950 tools/probe_gfx950.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));

__global__ void probe(const _Float16* A, const _Float16* B, float* D, const _Float16* src, _Float16* tr_out,
                      unsigned* perm_out, _Float16* dma_out) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[4096];
  const int l = threadIdx.x;
  // 1. MFMA: assumed A[row l&15][k = 8(l>>4)+j], B[k = 8(l>>4)+j][col l&15], D[row 4(l>>4)+r][col l&15]
  h8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j];
    b[j] = B[(8 * (l >> 4) + j) * 16 + (l & 15)];
  }
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
  // 2. LDS-DMA: 64 lanes x 16 B, lane l sources chunk (l ^ 6); LDS slot l must hold it
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 8 * (l ^ 6)),
                                   (__attribute__((address_space(3))) void*)(lds), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int j = 0; j < 8; ++j) dma_out[8 * l + j] = lds[8 * l + j];
  __syncthreads();
  // 3. tr read: LDS = 64 rows x 16 halfs, value = 16*row + col. lane = 16g+4q+p supplies row 4g+q, cols 4p..
  {
    volatile unsigned short* l16 = reinterpret_cast<volatile unsigned short*>(lds);
    for (int i = l; i < 1024; i += 64) l16[i] = __builtin_bit_cast(unsigned short, (_Float16)(float)i);
    __syncthreads();
    const int g = l >> 4, q = (l >> 2) & 3, p = l & 3;
    const s4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s4*)(lds + 16 * (4 * g + q) + 4 * p));
    typedef unsigned u2_t __attribute__((ext_vector_type(2)));
    const u2_t raw = __builtin_bit_cast(u2_t, t);
    reinterpret_cast<unsigned*>(tr_out)[2 * l] = raw.x;
    reinterpret_cast<unsigned*>(tr_out)[2 * l + 1] = raw.y;
  }
  // 4. permlane swaps
  unsigned x = l, y = 100 + l;
  auto r16 = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  auto r32 = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  perm_out[l] = r16[0];
  perm_out[64 + l] = r16[1];
  perm_out[128 + l] = r32[0];
  perm_out[192 + l] = r32[1];
}

int main() {
  std::vector<_Float16> A(16 * 32), B(32 * 16), src(512);
  for (int i = 0; i < 16; ++i)
    for (int k = 0; k < 32; ++k) A[i * 32 + k] = (_Float16)(float)((i * 7 + k * 3) % 5 - 2);
  for (int k = 0; k < 32; ++k)
    for (int j = 0; j < 16; ++j) B[k * 16 + j] = (_Float16)(float)((k * 5 + j * 11) % 7 - 3);
  for (int i = 0; i < 512; ++i) src[i] = (_Float16)(float)i;
  _Float16 *dA, *dB, *dsrc, *dtr, *ddma;
  float* dD;
  unsigned* dperm;
  hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dsrc, 1024); hipMalloc(&dtr, 512);
  hipMalloc(&ddma, 1024); hipMalloc(&dD, 1024); hipMalloc(&dperm, 1024);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dsrc, src.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dsrc, dtr, dperm, ddma);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
  std::vector<float> D(256);
  std::vector<_Float16> tr(256), dma(512);
  std::vector<unsigned> perm(256);
  hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  hipMemcpy(tr.data(), dtr, 512, hipMemcpyDeviceToHost);
  hipMemcpy(dma.data(), ddma, 1024, hipMemcpyDeviceToHost);
  hipMemcpy(perm.data(), dperm, 1024, hipMemcpyDeviceToHost);
  int bad = 0, fails = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      float s = 0;
      for (int k = 0; k < 32; ++k) s += (float)A[i * 32 + k] * (float)B[k * 16 + j];
      if (s != D[i * 16 + j]) ++bad;
    }
  printf("mfma_f32_16x16x32_f16 lane maps: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  fails += bad != 0;
  bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 8; ++j) bad += ((float)dma[8 * l + j] != (float)(8 * (l ^ 6) + j));
  printf("global_load_lds_dwordx4 lane-linear destination: %s (%d)\n", bad ? "FAIL" : "OK", bad);
  fails += bad != 0;
  bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) bad += ((float)tr[4 * l + j] != (float)(16 * (4 * (l >> 4) + j) + (l & 15)));
  printf("ds_read_b64_tr_b16 (lane i gets column i of rows 0..3 of its group's block): %s (%d)\n", bad ? "FAIL" : "OK", bad);
  if (bad) { for (int l = 0; l < 64; ++l) printf("  lane %2d: %g %g %g %g\n", l, (float)tr[4*l], (float)tr[4*l+1], (float)tr[4*l+2], (float)tr[4*l+3]); }
  fails += bad != 0;
  bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int row = l >> 4;
    const unsigned e0 = (row & 1) ? 100 + (l - 16) : l;          // vdst' = [v.r0, s.r0, v.r2, s.r2]
    const unsigned e1 = (row & 1) ? 100 + l : (l + 16);          // src'  = [v.r1, s.r1, v.r3, s.r3]
    bad += (perm[l] != e0) + (perm[64 + l] != e1);
  }
  printf("v_permlane16_swap (vdst odd rows <-> src even rows): %s (%d)\n", bad ? "FAIL" : "OK", bad);
  if (bad) for (int l = 0; l < 64; l += 16) printf("  row %d: vdst' starts %u, src' starts %u\n", l >> 4, perm[l], perm[64 + l]);
  fails += bad != 0;
  bad = 0;
  for (int l = 0; l < 64; ++l) {
    const unsigned e0 = l < 32 ? l : 100 + (l - 32);             // vdst' = [v.lo, s.lo]
    const unsigned e1 = l < 32 ? (l + 32) : 100 + l;             // src'  = [v.hi, s.hi]
    bad += (perm[128 + l] != e0) + (perm[192 + l] != e1);
  }
  printf("v_permlane32_swap (vdst hi <-> src lo): %s (%d)\n", bad ? "FAIL" : "OK", bad);
  if (bad) for (int l = 0; l < 64; l += 32) printf("  half %d: vdst' starts %u, src' starts %u\n", l >> 5, perm[128 + l], perm[192 + l]);
  fails += bad != 0;
  return fails ? 1 : 0;
}
