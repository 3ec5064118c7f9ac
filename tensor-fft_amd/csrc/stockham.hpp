// stockham.hpp — generic batched radix-{2,4,8,16} autosort passes through HBM.
//
// Covers every power-of-two N that has no dedicated LDS-resident kernel yet. It
// stands where the reference chains TensorFFT256 -> TensorRadix16* -> Radix2Kernel*
// (src/base/ComputeFFT.h:72-145; kernels TensorRadix16.cu:36-214, Radix2.cu:20-77)
// and keeps its contract: one full read + one full write of the array per pass,
// ping-pong between two buffers, 1/radix scaling per pass (TensorRadix16.cu:132-136,
// Radix2.cu:67-76). Differences: one launch per pass for the WHOLE batch (the
// reference launches per FFT and, for radix 2, per pair of sub-FFTs:
// ComputeFFT.h:123-138); autosort indexing instead of an up-front digit-reversal
// gather (TensorFFT256.cu:125-178), so reads are always contiguous over the
// thread index; butterflies in fp32 registers; twiddles from a two-level fp32
// table computed in fp64 on the host instead of per-element cosf/sinf
// (TensorRadix16.cu:117-125).
//
// Pass with sub-transform length Ns (Ns = product of the radices already done):
//   j in [0, N/R):  k = j mod Ns
//   v[i] = x[j + i N/R] * w_{Ns R}^(i k),  i = 0..R-1
//   y[(j - k) R + k + i Ns] = DFT_R(v)[i] / R
// With an inner batch C (transform along a strided axis) every index above is flattened with c
// (j -> j C + c, Ns -> Ns C); only the twiddle uses the unflattened k.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace stockham {

constexpr int kBlock = 256;
constexpr int kTwLoBits = 13;                 // w_N^e = lo[e & 8191] * hi[e >> 13]
constexpr uint32_t kTwLoSize = 1u << kTwLoBits;

struct cf {
  float re, im;
};
__device__ __forceinline__ cf cmul(cf a, cf b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cf cadd(cf a, cf b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cf mul_mi(cf a) { return {a.im, -a.re}; }     // a * (-i)

// forward DFT-4, natural order in and out
__device__ __forceinline__ void dft4(cf& a, cf& b, cf& c, cf& d) {
  const cf s0 = cadd(a, c), s1 = csub(a, c), s2 = cadd(b, d), s3 = mul_mi(csub(b, d));
  a = cadd(s0, s2);
  b = cadd(s1, s3);
  c = csub(s0, s2);
  d = csub(s1, s3);
}

template <int R>
__device__ __forceinline__ void dft(cf (&v)[R]);

template <>
__device__ __forceinline__ void dft<2>(cf (&v)[2]) {
  const cf a = v[0], b = v[1];
  v[0] = cadd(a, b);
  v[1] = csub(a, b);
}
template <>
__device__ __forceinline__ void dft<4>(cf (&v)[4]) {
  dft4(v[0], v[1], v[2], v[3]);
}
// n = n0 + 4 n1 (n0<4, n1<2), k = k0 + 2 k1 (k0<2, k1<4)
template <>
__device__ __forceinline__ void dft<8>(cf (&v)[8]) {
  const float h = 0.70710678118654752f;
  cf y[4][2];
#pragma unroll
  for (int n0 = 0; n0 < 4; ++n0) {
    y[n0][0] = cadd(v[n0], v[n0 + 4]);
    y[n0][1] = csub(v[n0], v[n0 + 4]);
  }
  // y[n0][1] *= w8^n0
  y[1][1] = cmul(y[1][1], cf{h, -h});
  y[2][1] = mul_mi(y[2][1]);
  y[3][1] = cmul(y[3][1], cf{-h, -h});
#pragma unroll
  for (int k0 = 0; k0 < 2; ++k0) {
    cf a = y[0][k0], b = y[1][k0], c = y[2][k0], d = y[3][k0];
    dft4(a, b, c, d);
    v[k0] = a;
    v[k0 + 2] = b;
    v[k0 + 4] = c;
    v[k0 + 6] = d;
  }
}
// n = n0 + 4 n1, k = k0 + 4 k1
template <>
__device__ __forceinline__ void dft<16>(cf (&v)[16]) {
  const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
  // w16^e for e = 0..9 (only products n0*k0 <= 9 occur)
  const cf w[10] = {{1.f, 0.f}, {c1, -s1}, {h, -h},   {s1, -c1},  {0.f, -1.f},
                    {-s1, -c1}, {-h, -h},  {-c1, -s1}, {-1.f, 0.f}, {-c1, s1}};
  cf y[4][4];
#pragma unroll
  for (int n0 = 0; n0 < 4; ++n0) {
    cf a = v[n0], b = v[n0 + 4], c = v[n0 + 8], d = v[n0 + 12];
    dft4(a, b, c, d);
    y[n0][0] = a;
    y[n0][1] = b;
    y[n0][2] = c;
    y[n0][3] = d;
  }
#pragma unroll
  for (int n0 = 1; n0 < 4; ++n0)
#pragma unroll
    for (int k0 = 1; k0 < 4; ++k0) y[n0][k0] = cmul(y[n0][k0], w[n0 * k0]);
#pragma unroll
  for (int k0 = 0; k0 < 4; ++k0) {
    cf a = y[0][k0], b = y[1][k0], c = y[2][k0], d = y[3][k0];
    dft4(a, b, c, d);
    v[k0] = a;
    v[k0 + 4] = b;
    v[k0 + 8] = c;
    v[k0 + 12] = d;
  }
}

// w64^e = exp(-2 pi i e / 64): constants of the composite butterflies below
__device__ constexpr cf kW64[64] = {
    {1.0f, -0.0f}, {0.995184727f, -0.0980171403f}, {0.98078528f, -0.195090322f}, {0.956940336f, -0.290284677f},
    {0.923879533f, -0.382683432f}, {0.881921264f, -0.471396737f}, {0.831469612f, -0.555570233f}, {0.773010453f, -0.634393284f},
    {0.707106781f, -0.707106781f}, {0.634393284f, -0.773010453f}, {0.555570233f, -0.831469612f}, {0.471396737f, -0.881921264f},
    {0.382683432f, -0.923879533f}, {0.290284677f, -0.956940336f}, {0.195090322f, -0.98078528f}, {0.0980171403f, -0.995184727f},
    {6.123234e-17f, -1.0f}, {-0.0980171403f, -0.995184727f}, {-0.195090322f, -0.98078528f}, {-0.290284677f, -0.956940336f},
    {-0.382683432f, -0.923879533f}, {-0.471396737f, -0.881921264f}, {-0.555570233f, -0.831469612f}, {-0.634393284f, -0.773010453f},
    {-0.707106781f, -0.707106781f}, {-0.773010453f, -0.634393284f}, {-0.831469612f, -0.555570233f}, {-0.881921264f, -0.471396737f},
    {-0.923879533f, -0.382683432f}, {-0.956940336f, -0.290284677f}, {-0.98078528f, -0.195090322f}, {-0.995184727f, -0.0980171403f},
    {-1.0f, -1.2246468e-16f}, {-0.995184727f, 0.0980171403f}, {-0.98078528f, 0.195090322f}, {-0.956940336f, 0.290284677f},
    {-0.923879533f, 0.382683432f}, {-0.881921264f, 0.471396737f}, {-0.831469612f, 0.555570233f}, {-0.773010453f, 0.634393284f},
    {-0.707106781f, 0.707106781f}, {-0.634393284f, 0.773010453f}, {-0.555570233f, 0.831469612f}, {-0.471396737f, 0.881921264f},
    {-0.382683432f, 0.923879533f}, {-0.290284677f, 0.956940336f}, {-0.195090322f, 0.98078528f}, {-0.0980171403f, 0.995184727f},
    {-1.8369702e-16f, 1.0f}, {0.0980171403f, 0.995184727f}, {0.195090322f, 0.98078528f}, {0.290284677f, 0.956940336f},
    {0.382683432f, 0.923879533f}, {0.471396737f, 0.881921264f}, {0.555570233f, 0.831469612f}, {0.634393284f, 0.773010453f},
    {0.707106781f, 0.707106781f}, {0.773010453f, 0.634393284f}, {0.831469612f, 0.555570233f}, {0.881921264f, 0.471396737f},
    {0.923879533f, 0.382683432f}, {0.956940336f, 0.290284677f}, {0.98078528f, 0.195090322f}, {0.995184727f, 0.0980171403f},
};

// Composite forward DFT of length RA * RB (natural order in and out): n = a + RA b, k = RB ka + kb:
//   X[RB ka + kb] = sum_a w_{RA RB}^(a kb) w_RA^(a ka) sum_b x[a + RA b] w_RB^(b kb)
template <int RA, int RB>
__device__ __forceinline__ void dft_composite(cf (&v)[RA * RB]) {
  constexpr int N = RA * RB;
  cf y[RA][RB];
#pragma unroll
  for (int a = 0; a < RA; ++a) {
    cf t[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) t[b] = v[a + RA * b];
    dft<RB>(t);
#pragma unroll
    for (int kb = 0; kb < RB; ++kb) y[a][kb] = (a * kb) ? cmul(t[kb], kW64[(a * kb * (64 / N)) & 63]) : t[kb];
  }
#pragma unroll
  for (int kb = 0; kb < RB; ++kb) {
    cf t[RA];
#pragma unroll
    for (int a = 0; a < RA; ++a) t[a] = y[a][kb];
    dft<RA>(t);
#pragma unroll
    for (int ka = 0; ka < RA; ++ka) v[RB * ka + kb] = t[ka];
  }
}
template <>
__device__ __forceinline__ void dft<32>(cf (&v)[32]) {
  dft_composite<2, 16>(v);
}
template <>
__device__ __forceinline__ void dft<64>(cf (&v)[64]) {
  dft_composite<4, 16>(v);
}

struct PassArgs {
  const _Float16* in_re;
  const _Float16* in_im;
  _Float16* out_re;
  _Float16* out_im;
  uint64_t in_stride;    // halves between FFTs
  uint64_t out_stride;
  uint64_t n;            // FFT length (twiddle table modulus)
  uint64_t m_f;          // flattened butterflies per transform = (n / R) * inner
  uint64_t ns;           // flattened sub-transform length before this pass = Ns * inner
  uint32_t inner_shift;  // log2(inner): flattened index = index * inner + c (FFT along a strided axis)
  uint32_t skip_tw;      // the previous pass already applied this pass's input twiddles
  uint64_t tw_mul;       // N / (Ns * R): exponent of w_N per unit of i*k
  uint64_t batch;
  uint32_t m_shift;      // log2(m_f)
  const float2* tw_lo;   // w_N^e, e < min(N, 8192)
  const float2* tw_hi;   // w_N^(e * 8192), e < N / 8192 (unused when N <= 8192)
  float scale;           // factor on the butterfly output: 1/R (sequential scaling), 1, or 1/N on the last pass (include/tfft.h)
};

template <int R>
__global__ __launch_bounds__(kBlock) void pass_kernel(PassArgs a) {
  // (transform, butterfly) flattened into one index so that short transforms still fill their blocks
  const uint64_t gid = blockIdx.x * static_cast<uint64_t>(kBlock) + threadIdx.x;
  const uint64_t m = a.m_f;
  const uint64_t fft = gid >> a.m_shift;
  if (fft >= a.batch) return;
  const uint64_t j = gid & (m - 1);
  const uint64_t k = j & (a.ns - 1);           // flattened k
  const _Float16* xr = a.in_re + fft * a.in_stride + j;
  const _Float16* xi = a.in_im + fft * a.in_stride + j;
  cf v[R];
#pragma unroll
  for (int i = 0; i < R; ++i) v[i] = cf{static_cast<float>(xr[i * m]), static_cast<float>(xi[i * m])};
  // (radix 32 / 64 passes exist only directly behind a column pass, which applies their input twiddles)
  if (R <= 16 && (a.ns >> a.inner_shift) > 1 && !a.skip_tw) {
    const uint64_t step = (k >> a.inner_shift) * a.tw_mul;      // < N / R
    // w^(i step), i = 1..R-1: one table look-up (two loads for N > 8192) and products w_i = w_(i/2) w_(i - i/2)
    // in fp32 (log2 R deep: the error stays ~1e-7, far below binary16) instead of R - 1 look-ups.
    cf w[R];
    {
      const uint64_t e = step & (a.n - 1);
      const float2 lo = a.tw_lo[e & (kTwLoSize - 1)];
      w[1] = cf{lo.x, lo.y};
      if (a.n > kTwLoSize) {
        const float2 hi = a.tw_hi[e >> kTwLoBits];
        w[1] = cmul(w[1], cf{hi.x, hi.y});
      }
    }
#pragma unroll
    for (int i = 2; i < R; ++i) w[i] = cmul(w[i >> 1], w[i - (i >> 1)]);
#pragma unroll
    for (int i = 1; i < R; ++i) v[i] = cmul(v[i], w[i]);
  }
  dft<R>(v);
  const float sc = a.scale;
  _Float16* yr = a.out_re + fft * a.out_stride + (j - k) * R + k;
  _Float16* yi = a.out_im + fft * a.out_stride + (j - k) * R + k;
  if (a.ns == 1) {
    // first pass of a plain transform: this thread's R outputs are contiguous (y[j R + i]): one or two wide stores
    typedef _Float16 hvR __attribute__((ext_vector_type(R)));
    hvR pr, pi;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      pr[i] = static_cast<_Float16>(v[i].re * sc);
      pi[i] = static_cast<_Float16>(v[i].im * sc);
    }
    *reinterpret_cast<hvR*>(yr) = pr;
    *reinterpret_cast<hvR*>(yi) = pi;
    return;
  }
  if (R == 16 && a.ns == 16) {
    // second radix-16 pass: the 16 threads of a lane group own the 16 x 16 block y[base + i 16 + k] column by
    // column (thread = k). Transpose it through LDS (the group lives in one wave, so LDS program order is
    // enough) and let thread t store row i = t: 32 contiguous bytes per plane instead of 16 two-byte stores.
    __shared__ uint32_t xchg[kBlock * 16];
    const uint32_t t = threadIdx.x & 15, grp = threadIdx.x >> 4;
    uint32_t* mine = xchg + grp * 256 + t * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      typedef _Float16 hv2 __attribute__((ext_vector_type(2)));
      const hv2 pair = {static_cast<_Float16>(v[i].re * sc), static_cast<_Float16>(v[i].im * sc)};
      mine[i] = __builtin_bit_cast(uint32_t, pair);
    }
    uint32_t row[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) row[kk] = xchg[grp * 256 + kk * 16 + t];
    typedef uint32_t uv4 __attribute__((ext_vector_type(4)));
    uv4 re0, re1, im0, im1;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      re0[w] = (row[2 * w] & 0xffffu) | (row[2 * w + 1] << 16);
      im0[w] = (row[2 * w] >> 16) | (row[2 * w + 1] & 0xffff0000u);
      re1[w] = (row[8 + 2 * w] & 0xffffu) | (row[8 + 2 * w + 1] << 16);
      im1[w] = (row[8 + 2 * w] >> 16) | (row[8 + 2 * w + 1] & 0xffff0000u);
    }
    _Float16* br = a.out_re + fft * a.out_stride + (j - k) * 16 + t * 16;   // (j - k) is the group's first j
    _Float16* bi = a.out_im + fft * a.out_stride + (j - k) * 16 + t * 16;
    reinterpret_cast<uv4*>(br)[0] = re0;
    reinterpret_cast<uv4*>(br)[1] = re1;
    reinterpret_cast<uv4*>(bi)[0] = im0;
    reinterpret_cast<uv4*>(bi)[1] = im1;
    return;
  }
#pragma unroll
  for (int i = 0; i < R; ++i) {
    yr[i * a.ns] = static_cast<_Float16>(v[i].re * sc);
    yi[i * a.ns] = static_cast<_Float16>(v[i].im * sc);
  }
}

// Two adjacent butterflies per thread (j and j + 1), for passes whose input twiddles were already applied by the
// preceding column pass (skip_tw) and whose sub-transform length is even: every global access is 4 bytes per lane
// (256 contiguous bytes per wave instruction) instead of 2.
template <int R>
__global__ __launch_bounds__(kBlock) void pass_pair_kernel(PassArgs a) {
  const uint64_t gid = blockIdx.x * static_cast<uint64_t>(kBlock) + threadIdx.x;
  const uint64_t m = a.m_f;
  const uint64_t fft = gid >> (a.m_shift - 1);
  if (fft >= a.batch) return;
  const uint64_t j = (gid & ((m >> 1) - 1)) << 1;
  const uint64_t k = j & (a.ns - 1);
  const _Float16* xr = a.in_re + fft * a.in_stride + j;
  const _Float16* xi = a.in_im + fft * a.in_stride + j;
  typedef _Float16 hv2 __attribute__((ext_vector_type(2)));
  cf v0[R], v1[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const hv2 pr = *reinterpret_cast<const hv2*>(xr + i * m);
    const hv2 pi = *reinterpret_cast<const hv2*>(xi + i * m);
    v0[i] = cf{static_cast<float>(pr[0]), static_cast<float>(pi[0])};
    v1[i] = cf{static_cast<float>(pr[1]), static_cast<float>(pi[1])};
  }
  dft<R>(v0);
  dft<R>(v1);
  const float sc = a.scale;
  _Float16* yr = a.out_re + fft * a.out_stride + (j - k) * R + k;
  _Float16* yi = a.out_im + fft * a.out_stride + (j - k) * R + k;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const hv2 pr = {static_cast<_Float16>(v0[i].re * sc), static_cast<_Float16>(v1[i].re * sc)};
    const hv2 pi = {static_cast<_Float16>(v0[i].im * sc), static_cast<_Float16>(v1[i].im * sc)};
    *reinterpret_cast<hv2*>(yr + i * a.ns) = pr;
    *reinterpret_cast<hv2*>(yi + i * a.ns) = pi;
  }
}

// Final radix-32 / 64 / 128 pass of ONE or a few transforms (2^13 = 256 x 32, 2^14 = 256 x 64, 2^15 = 256 x 128 behind the latency column kernel,
// collat.hpp), workgroup-cooperative: a radix-64 butterfly per THREAD (pass_kernel<64>) leaves a single 2^14 with one workgroup
// of 256 threads that each grind through a thousand fp32 instructions and 256 two-byte memory instructions (2^14 as 256 x 64:
// 13.5 us, profiles/r5_lat_shapes.txt). Here a workgroup of 2 R threads takes 8 columns of the [R][m] matrix (16-byte row
// segments, ONE vector load and ONE vector store per thread) and runs the R-point transforms of those columns as three autosort
// steps 4 x 8 (R = 32), 4 x 4 x 4 (R = 64) or 4 x 4 x 8 (R = 128) through LDS in fp32 (one rounding to binary16, at the end); the input twiddles
// were applied by the column pass in front (skip_tw), and because this is the plan's last pass (Ns = m) output row k of column j
// is element k m + j: the input's own layout. m % 8 == 0. (The same pass for R = 512 ... 2048 loses to the three-launch plans,
// profiles/r5_coop_tail_radices.txt.)
constexpr int kCoopCols = 8;
template <int R>
__global__ __launch_bounds__(2 * R) void tail_coop_kernel(PassArgs a) {
  static_assert(R == 32 || R == 64 || R == 128, "4 x 8, 4 x 4 x 4 or 4 x 4 x 8");
  __shared__ float s_re[2][R * kCoopCols];
  __shared__ float s_im[2][R * kCoopCols];
  __shared__ __attribute__((aligned(16))) _Float16 s_out[2][R * kCoopCols];
  const uint32_t t = threadIdx.x, c = t & 7, q = t >> 3;       // q < R / 4: one radix-4 butterfly per thread and step
  const uint64_t tiles = a.m_f / kCoopCols;
  const uint64_t fft = blockIdx.x / tiles;
  const uint64_t j0 = (blockIdx.x - fft * tiles) * kCoopCols;
  typedef _Float16 hv8 __attribute__((ext_vector_type(8)));
  {
    // thread t < R: row t of the RE plane, t >= R: row t - R of the IM plane
    const uint32_t row = t & (R - 1);
    const _Float16* src = (t < R ? a.in_re : a.in_im) + fft * a.in_stride + row * a.m_f + j0;
    const hv8 v = *reinterpret_cast<const hv8*>(src);
    float* dst = (t < R ? s_re[0] : s_im[0]) + row * kCoopCols;
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[e] = static_cast<float>(v[e]);
  }
  __syncthreads();
  auto at = [&](int buf, uint32_t row) { return cf{s_re[buf][row * kCoopCols + c], s_im[buf][row * kCoopCols + c]}; };
  auto put = [&](int buf, uint32_t row, cf v) {
    s_re[buf][row * kCoopCols + c] = v.re;
    s_im[buf][row * kCoopCols + c] = v.im;
  };
  auto w = [](uint32_t e, float inv) {       // exp(-2 pi i e inv): v_cos / v_sin take revolutions (|error| ~ 1e-6)
    const float r = static_cast<float>(e) * inv;
    return cf{__builtin_amdgcn_cosf(r), -__builtin_amdgcn_sinf(r)};
  };
  auto out = [&](uint32_t row, cf v) {
    s_out[0][row * kCoopCols + c] = static_cast<_Float16>(v.re * a.scale);
    s_out[1][row * kCoopCols + c] = static_cast<_Float16>(v.im * a.scale);
  };
  {  // step 1: radix 4, Ns = 1: j = q, y[4 j + i]
    cf v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = at(0, q + (R / 4) * i);
    dft<4>(v);
#pragma unroll
    for (int i = 0; i < 4; ++i) put(1, 4 * q + i, v[i]);
  }
  __syncthreads();
  if (R == 32) {
    if (q < 4) {  // step 2 of 4 x 8: radix 8, Ns = 4: j = k = q < 4, twiddle w_32^(i k), output row k + 4 i
      cf v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = at(1, q + 4 * i);
#pragma unroll
      for (int i = 1; i < 8; ++i) v[i] = cmul(v[i], w(i * q, 1.0f / 32));
      dft<8>(v);
#pragma unroll
      for (int i = 0; i < 8; ++i) out(q + 4 * i, v[i]);
    }
  } else {  // step 2: radix 4, Ns = 4: j = q, k = j & 3, twiddle w_16^(i k), y[(j - k) 4 + k + 4 i]
    const uint32_t k = q & 3;
    cf v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = at(1, q + (R / 4) * i);
#pragma unroll
    for (int i = 1; i < 4; ++i) v[i] = cmul(v[i], w(i * k, 1.0f / 16));
    dft<4>(v);
#pragma unroll
    for (int i = 0; i < 4; ++i) put(0, (q - k) * 4 + k + 4 * i, v[i]);
  }
  __syncthreads();
  if (R == 32) {
  } else if (R == 64) {  // step 3: radix 4, Ns = 16: j = k = q < 16, twiddle w_64^(i k), output row k + 16 i
    cf v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = at(0, q + 16 * i);
#pragma unroll
    for (int i = 1; i < 4; ++i) v[i] = cmul(v[i], w(i * q, 1.0f / 64));
    dft<4>(v);
#pragma unroll
    for (int i = 0; i < 4; ++i) out(q + 16 * i, v[i]);
  } else if (q < 16) {  // step 3: radix 8, Ns = 16: j = k = q, twiddle w_128^(i k), output row k + 16 i
    cf v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = at(0, q + 16 * i);
#pragma unroll
    for (int i = 1; i < 8; ++i) v[i] = cmul(v[i], w(i * q, 1.0f / 128));
    dft<8>(v);
#pragma unroll
    for (int i = 0; i < 8; ++i) out(q + 16 * i, v[i]);
  }
  __syncthreads();
  {
    const uint32_t row = t & (R - 1);
    _Float16* dst = (t < R ? a.out_re : a.out_im) + fft * a.out_stride + row * a.m_f + j0;
    *reinterpret_cast<hv8*>(dst) = *reinterpret_cast<const hv8*>(s_out[t < R ? 0 : 1] + row * kCoopCols);
  }
}

// plain planar copy (in-place requests whose pass chain cannot start from `in`)
__global__ __launch_bounds__(kBlock) void copy_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                       uint64_t n32) {
  for (uint64_t i = blockIdx.x * static_cast<uint64_t>(kBlock) + threadIdx.x; i < n32;
       i += static_cast<uint64_t>(gridDim.x) * kBlock)
    dst[i] = src[i];
}

}  // namespace stockham
