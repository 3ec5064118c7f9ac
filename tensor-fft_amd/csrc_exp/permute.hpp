// permute.hpp — out[b][a][c] = in[a][b][c] * w_N^((e0 + b) (a C + c)), planar fp16, c contiguous.
//
// The data-movement step either side of the one all-to-all of the distributed transform
// (SURVEY 8e): with N = 0 it is the pure block re-ordering that packs / unpacks the exchange
// buffers, with N > 0 it also applies the four-step twiddle between the column and the row
// transforms. The reference has no counterpart (it has no transform larger than one device).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace permute {

constexpr int kBlock = 256;

struct Args {
  const uint16_t* in_re;
  const uint16_t* in_im;
  uint16_t* out_re;
  uint16_t* out_im;
  uint64_t A, B, C8;       // C8 = C / 8 (16-byte vectors)
  uint64_t n_tw;           // 0: no twiddle
  uint64_t e0;
};

typedef _Float16 hv8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(kBlock) void permute_twiddle_kernel(Args p) {
  const uint64_t total = p.A * p.B * p.C8;
  for (uint64_t t = blockIdx.x * static_cast<uint64_t>(kBlock) + threadIdx.x; t < total;
       t += static_cast<uint64_t>(gridDim.x) * kBlock) {
    // t enumerates the OUTPUT in order: (b, a, c8)
    const uint64_t c8 = t % p.C8;
    const uint64_t ba = t / p.C8;
    const uint64_t a = ba % p.A, b = ba / p.A;
    const uint64_t src = ((a * p.B + b) * p.C8 + c8) * 8;
    hv8 re = *reinterpret_cast<const hv8*>(p.in_re + src);
    hv8 im = *reinterpret_cast<const hv8*>(p.in_im + src);
    if (p.n_tw) {
      const uint64_t row = (p.e0 + b) % p.n_tw;
      const uint64_t col0 = a * p.C8 * 8 + c8 * 8;
      // w^(row col0) and w^row from exact reduced exponents, then a 7-step recurrence in fp32
      const unsigned __int128 prod = static_cast<unsigned __int128>(row) * col0;
      const uint64_t e = static_cast<uint64_t>(prod % p.n_tw);
      double s0, c0, s1, c1;
      sincospi(-2.0 * static_cast<double>(e) / static_cast<double>(p.n_tw), &s0, &c0);
      sincospi(-2.0 * static_cast<double>(row) / static_cast<double>(p.n_tw), &s1, &c1);
      float wr = static_cast<float>(c0), wi = static_cast<float>(s0);
      const float sr = static_cast<float>(c1), si = static_cast<float>(s1);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xr = static_cast<float>(re[j]), xi = static_cast<float>(im[j]);
        re[j] = static_cast<_Float16>(xr * wr - xi * wi);
        im[j] = static_cast<_Float16>(xr * wi + xi * wr);
        const float nr = wr * sr - wi * si;
        wi = wr * si + wi * sr;
        wr = nr;
      }
    }
    *reinterpret_cast<hv8*>(p.out_re + t * 8) = re;
    *reinterpret_cast<hv8*>(p.out_im + t * 8) = im;
  }
}

// interleaved half2 [n] (re, im pairs: the layout cuFFT / hipFFT use, reference AccuracyCalculator.h:35-48,
// TestingDataCreation.h half2 generators) <-> planar re[n], im[n]. 8 complex samples per thread.
typedef uint32_t uv4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kBlock) void deinterleave_kernel(const uv4* __restrict__ in, uv4* __restrict__ re,
                                                               uv4* __restrict__ im, uint64_t n8) {
  for (uint64_t t = blockIdx.x * static_cast<uint64_t>(kBlock) + threadIdx.x; t < n8;
       t += static_cast<uint64_t>(gridDim.x) * kBlock) {
    const uv4 a = in[2 * t], b = in[2 * t + 1];       // 8 (re, im) pairs
    uv4 r, i;
    r.x = (a.x & 0xffffu) | (a.y << 16);  i.x = (a.x >> 16) | (a.y & 0xffff0000u);
    r.y = (a.z & 0xffffu) | (a.w << 16);  i.y = (a.z >> 16) | (a.w & 0xffff0000u);
    r.z = (b.x & 0xffffu) | (b.y << 16);  i.z = (b.x >> 16) | (b.y & 0xffff0000u);
    r.w = (b.z & 0xffffu) | (b.w << 16);  i.w = (b.z >> 16) | (b.w & 0xffff0000u);
    re[t] = r;
    im[t] = i;
  }
}

__global__ __launch_bounds__(kBlock) void interleave_kernel(const uv4* __restrict__ re, const uv4* __restrict__ im,
                                                             uv4* __restrict__ out, uint64_t n8) {
  for (uint64_t t = blockIdx.x * static_cast<uint64_t>(kBlock) + threadIdx.x; t < n8;
       t += static_cast<uint64_t>(gridDim.x) * kBlock) {
    const uv4 r = re[t], i = im[t];
    uv4 a, b;
    a.x = (r.x & 0xffffu) | (i.x << 16);  a.y = (r.x >> 16) | (i.x & 0xffff0000u);
    a.z = (r.y & 0xffffu) | (i.y << 16);  a.w = (r.y >> 16) | (i.y & 0xffff0000u);
    b.x = (r.z & 0xffffu) | (i.z << 16);  b.y = (r.z >> 16) | (i.z & 0xffff0000u);
    b.z = (r.w & 0xffffu) | (i.w << 16);  b.w = (r.w >> 16) | (i.w & 0xffff0000u);
    out[2 * t] = a;
    out[2 * t + 1] = b;
  }
}

}  // namespace permute
