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

}  // namespace permute
