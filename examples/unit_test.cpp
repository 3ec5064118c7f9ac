// unit_test.cpp — the reference's functional test (src/testing/unitTesting/UnitTest.cu:7-56, FFTTest.cu:20-190) over
// include/tensor_fft.hpp on MI355X, with the same protocol and thresholds:
//   N = 2^8 .. 2^20 (x 2 per step), 10 signals per length: sine superposition with 20 harmonics, weights
//   GetRandomWeights(20, 42 i) / (20, 42 * 42 i); the transform under test is CreatePlan -> DataHandler -> ComputeFFT;
//   the comparison data is the vendor library's double-precision complex FFT of the same binary16 input, divided by N
//   (the reference: cuFFT Z2Z, CuFFTTest.h:218-261; here: hipFFT Z2Z); pass when the mean / sigma / max of |delta| over the 2N
//   reals stay below 1e-3 / 1e-2 / 0.5.
// A C++ host of the reference's own shape, written against the shim: nothing here knows about the C ABI or the kernels.
// (tests/ holds the same protocol in Python against the CPU oracle; this one has no dependency outside ROCm.)
//
// usage: unit_test [max_log2 = 20]        exit code 0 = "All tests passed!"
#include <hipfft/hipfft.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <random>
#include <vector>

#include "tensor_fft.hpp"

namespace {

// uniform(-1, 1) weights from the standard library's default engine seeded through a seed_seq: the reference's generator of
// test-signal weights (TestingDataCreation.h:15-27); the values depend on the standard library, as they do there
std::vector<float> GetRandomWeights(int max_frequencies, int seed) {
  std::seed_seq seq = {seed};
  std::default_random_engine generator(seq);
  std::uniform_real_distribution<float> distro(-1.0, 1.0);
  std::vector<float> w;
  for (int i = 0; i < max_frequencies; ++i) w.push_back(distro(generator));
  return w;
}

// x_re[t] = sum_f w_re[f] sin(2 pi f t / N), x_im likewise, f < cutoff; float sine of a double phase, accumulated in double,
// rounded to binary16 (the reference builds it on the GPU, TestingDataCreation.h:89-117; the host does the same arithmetic)
std::unique_ptr<__half[]> CreateSineSuperposition(int n, const std::vector<float>& w_re, const std::vector<float>& w_im, int cutoff) {
  auto data = std::make_unique<__half[]>(2 * static_cast<size_t>(n));
  for (int t = 0; t < n; ++t) {
    double re = 0, im = 0;
    for (int f = 0; f < cutoff; ++f) {
      const float s = sinf(static_cast<float>((2 * M_PI * f * static_cast<double>(t)) / static_cast<double>(n)));
      re += w_re[f] * s;
      im += w_im[f] * s;
    }
    data[t] = __float2half(static_cast<float>(re));
    data[t + n] = __float2half(static_cast<float>(im));
  }
  return data;
}

// the vendor library in double precision on the SAME binary16 input, divided by N, as split re | im doubles
std::unique_ptr<double[]> ComparisonData(int n, const __half* data) {
  std::vector<hipfftDoubleComplex> host(n);
  for (int i = 0; i < n; ++i) host[i] = hipfftDoubleComplex{static_cast<double>(__half2float(data[i])), static_cast<double>(__half2float(data[i + n]))};
  hipfftDoubleComplex* dev = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&dev), sizeof(hipfftDoubleComplex) * n) != hipSuccess) return nullptr;
  (void)hipMemcpy(dev, host.data(), sizeof(hipfftDoubleComplex) * n, hipMemcpyHostToDevice);
  hipfftHandle plan;
  if (hipfftPlan1d(&plan, n, HIPFFT_Z2Z, 1) != HIPFFT_SUCCESS) return nullptr;
  const bool ok = hipfftExecZ2Z(plan, dev, dev, HIPFFT_FORWARD) == HIPFFT_SUCCESS;
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(host.data(), dev, sizeof(hipfftDoubleComplex) * n, hipMemcpyDeviceToHost);
  (void)hipfftDestroy(plan);
  (void)hipFree(dev);
  if (!ok) return nullptr;
  auto out = std::make_unique<double[]>(2 * static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) {
    out[i] = host[i].x / n;
    out[i + n] = host[i].y / n;
  }
  return out;
}

// CreatePlan -> PlanWorksOnDevice -> DataHandler -> CopyDataHostToDevice -> ComputeFFT -> CopyResultsDeviceToHost
// (the reference's FullSingleFFTComputation, FFTTest.cu:22-86)
bool FullSingleFFTComputation(int n, __half* data) {
  auto possible_plan = CreatePlan(n);
  if (!possible_plan) { std::cout << "Plan creation failed" << std::endl; return false; }
  Plan<int> my_plan = possible_plan.value();
  int device_id = 0;
  (void)hipGetDevice(&device_id);
  if (!PlanWorksOnDevice(my_plan, device_id)) { std::cout << "Error Plan doesnt work on used device." << std::endl; return false; }
  DataHandler<int> my_handler(n);
  if (auto e = my_handler.PeakAtLastError()) { std::cout << e.value() << std::endl; return false; }
  if (auto e = my_handler.CopyDataHostToDevice(data)) { std::cout << e.value() << std::endl; return false; }
  if (auto e = ComputeFFT(my_plan, my_handler, GetMaxNoOptInSharedMem(device_id))) { std::cout << e.value() << std::endl; return false; }
  if (auto e = my_handler.CopyResultsDeviceToHost(data, my_plan.results_in_results_)) { std::cout << e.value() << std::endl; return false; }
  (void)hipDeviceSynchronize();
  return true;
}

bool TestFullFFT(int n, double avg_thr, double sigma_thr, double max_thr, const std::vector<float>& w_re, const std::vector<float>& w_im,
                 double* worst_max) {
  auto data = CreateSineSuperposition(n, w_re, w_im, static_cast<int>(w_re.size()));
  auto exact = ComparisonData(n, data.get());
  if (!exact) { std::cout << "Error! Failed to create comparision data." << std::endl; return false; }
  if (!FullSingleFFTComputation(n, data.get())) return false;
  double mx = 0, sum = 0;
  const size_t cnt = 2 * static_cast<size_t>(n);
  for (size_t i = 0; i < cnt; ++i) {
    const double d = std::fabs(static_cast<double>(__half2float(data[i])) - exact[i]);
    mx = std::fmax(mx, d);
    sum += d;
  }
  const double avg = sum / cnt;
  double var = 0;
  for (size_t i = 0; i < cnt; ++i) {
    const double d = std::fabs(static_cast<double>(__half2float(data[i])) - exact[i]) - avg;
    var += d * d;
  }
  const double sigma = std::sqrt(var / (cnt - 1));
  *worst_max = std::fmax(*worst_max, mx);
  if (avg > avg_thr || sigma > sigma_thr || mx > max_thr) {
    std::cout << "avg " << avg << " sigma " << sigma << " max " << mx << std::endl;
    return false;
  }
  return true;
}

}  // namespace

int main(int argc, char** argv) {
  const int start_fft_length = 16 * 16;
  const int end_fft_length = 1 << (argc > 1 ? std::atoi(argv[1]) : 20);
  constexpr int runs_per_fft_length = 10;
  constexpr int highest_harmonic = 20;
  constexpr double average_deviation_threshold = 0.001;
  constexpr double sigma_deviation_threshold = 0.01;
  constexpr double max_deviation_threshold = 0.5;

  std::vector<std::vector<float>> weights_RE, weights_IM;
  for (int i = 0; i < runs_per_fft_length; ++i) {
    weights_RE.push_back(GetRandomWeights(highest_harmonic, 42 * i));
    weights_IM.push_back(GetRandomWeights(highest_harmonic, 42 * 42 * i));
  }
  for (int fft_length = start_fft_length; fft_length <= end_fft_length; fft_length *= 2) {
    double worst = 0;
    for (int j = 0; j < runs_per_fft_length; ++j) {
      if (!TestFullFFT(fft_length, average_deviation_threshold, sigma_deviation_threshold, max_deviation_threshold, weights_RE[j],
                       weights_IM[j], &worst)) {
        std::cout << "Error! Test at fft_length: " << fft_length << " failed!" << std::endl;
        return 1;
      }
    }
    std::cout << "Testing fft_length: " << fft_length << "  ok (largest deviation over 10 signals " << worst << ")\n";
  }
  ReleaseComputeFFTPlans();
  std::cout << "All tests passed!" << std::endl;
  return 0;
}
