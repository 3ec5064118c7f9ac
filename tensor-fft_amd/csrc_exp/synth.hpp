// synth.hpp — synthetic input on the device: uniform(-1, 1) binary16 samples that are a pure function of
// (seed, transform index, plane, sample index).
//
// The reference synthesises its test signals on the GPU as well (src/testing/TestingDataCreation.h:29-147, 152-193) and
// copies them through the host; at BASELINE sizes (1 GiB .. 32 GiB of input per GPU) the data has to be born in HBM.
// Because every element is a counter-based hash, any sub-batch can be regenerated anywhere: the test infrastructure
// restates the same function on the CPU, so a benchmark can check sampled transforms of a batch that never existed
// on the host.
//
//   h = mix(mix(seed + 0x9E3779B97F4A7C15 (fft + 1)) ^ (2 j + plane)),  mix = the splitmix64 finaliser
//   x = ((h >> 41) - 2^22 + 1/2) 2^-22        23 random bits -> 2^23 equidistant values in (-1, 1), exact in fp32
//   sample = binary16(x), round to nearest even
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace synth {

constexpr int kBlock = 256;

__host__ __device__ inline uint64_t mix64(uint64_t z) {
  z ^= z >> 30;
  z *= 0xbf58476d1ce4e5b9ull;
  z ^= z >> 27;
  z *= 0x94d049bb133111ebull;
  z ^= z >> 31;
  return z;
}
__host__ __device__ inline float uniform_pm1(uint64_t seed, uint64_t fft, uint32_t plane, uint64_t j) {
  const uint64_t h = mix64(mix64(seed + 0x9E3779B97F4A7C15ull * (fft + 1)) ^ (2 * j + plane));
  const int32_t k = static_cast<int32_t>(h >> 41) - (1 << 22);
  return (static_cast<float>(k) + 0.5f) * (1.0f / 4194304.0f);
}

typedef _Float16 hv8 __attribute__((ext_vector_type(8)));

// planes re / im: transform b at + b * stride halves, n halves each (n % 8 == 0)
__global__ __launch_bounds__(kBlock) void uniform_kernel(uint16_t* re, uint16_t* im, uint64_t n8, uint64_t batch,
                                                         uint64_t stride, uint64_t first_fft, uint64_t seed) {
  const uint64_t total = batch * 2 * n8;
  for (uint64_t t = blockIdx.x * static_cast<uint64_t>(kBlock) + threadIdx.x; t < total;
       t += static_cast<uint64_t>(gridDim.x) * kBlock) {
    const uint64_t v = t % n8, bp = t / n8;
    const uint32_t plane = static_cast<uint32_t>(bp & 1);
    const uint64_t b = bp >> 1;
    hv8 out;
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = static_cast<_Float16>(uniform_pm1(seed, first_fft + b, plane, 8 * v + e));
    *reinterpret_cast<hv8*>((plane ? im : re) + b * stride + 8 * v) = out;
  }
}

}  // namespace synth
