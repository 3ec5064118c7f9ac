// example_single_fft.cpp — the reference's five-step recipe (src/base/ComputeFFT.h:1-16,
// src/testing/ExampleSingleFFT.cu, ExampleBatchFFT.cu) against include/tensor_fft.hpp on MI355X:
// CreatePlan -> DataHandler -> CopyDataHostToDevice -> ComputeFFT -> CopyResultsDeviceToHost.
// Checks the spectrum of an integer-frequency tone mix against its closed form and exits 0 / 1.
//
// usage: example_single_fft [log2_N = 12] [batch = 4]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "tensor_fft.hpp"

namespace {
// re[n] = sum a_f sin(2 pi f n / N), im[n] = sum b_f sin(2 pi f n / N)  =>  X[f] = (b_f - i a_f) / 2,
// X[N - f] = -X[f]  (1 <= f < harmonics), the reference's test signal in closed form.
constexpr int kHarmonics = 6;
const double kA[kHarmonics] = {0.0, 0.8125, 0.25, -0.625, 0.5, -0.75};
const double kB[kHarmonics] = {0.0, 0.6875, -0.375, 0.5, -0.5625, 0.3125};

void make_signal(int n, __half* dst) {
  for (int t = 0; t < n; ++t) {
    double re = 0, im = 0;
    for (int f = 1; f < kHarmonics; ++f) {
      const double s = std::sin(2.0 * M_PI * f * t / n);
      re += kA[f] * s;
      im += kB[f] * s;
    }
    dst[t] = __float2half(static_cast<float>(re));
    dst[t + n] = __float2half(static_cast<float>(im));
  }
}

double max_error(int n, const __half* got) {
  double worst = 0;
  for (int k = 0; k < n; ++k) {
    double er = 0, ei = 0;
    if (k >= 1 && k < kHarmonics) { er = kB[k] / 2; ei = -kA[k] / 2; }
    if (n - k >= 1 && n - k < kHarmonics) { er = -kB[n - k] / 2; ei = kA[n - k] / 2; }
    worst = std::fmax(worst, std::fabs(__half2float(got[k]) - er));
    worst = std::fmax(worst, std::fabs(__half2float(got[k + n]) - ei));
  }
  return worst;
}
}  // namespace

int main(int argc, char** argv) {
  const int lg = argc > 1 ? std::atoi(argv[1]) : 12;
  const int batch = argc > 2 ? std::atoi(argv[2]) : 4;
  const int n = 1 << lg;
  const BaseFFTMode mode = n >= 4096 ? Mode_4096 : Mode_256;

  auto maybe_plan = CreatePlan(n, mode, mode == Mode_4096 ? 16 : 1, 1, 256);
  if (!maybe_plan) return 1;
  Plan<int> plan = maybe_plan.value();
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!PlanWorksOnDevice(plan, dev)) return 1;

  std::vector<__half> host(2 * static_cast<size_t>(n));
  make_signal(n, host.data());

  // single transform
  DataHandler<int> handler(n);
  if (auto e = handler.PeakAtLastError()) { std::printf("%s\n", e->c_str()); return 1; }
  if (auto e = handler.CopyDataHostToDevice(host.data())) { std::printf("%s\n", e->c_str()); return 1; }
  if (auto e = ComputeFFT(plan, handler, GetMaxNoOptInSharedMem(dev))) { std::printf("%s\n", e->c_str()); return 1; }
  std::vector<__half> out(2 * static_cast<size_t>(n));
  if (auto e = handler.CopyResultsDeviceToHost(out.data(), plan.results_in_results_)) { std::printf("%s\n", e->c_str()); return 1; }
  (void)hipDeviceSynchronize();
  const double e1 = max_error(n, out.data());

  // batch
  DataBatchHandler<int> bh(n, batch);
  std::vector<__half> hb(2 * static_cast<size_t>(n) * batch), ob(hb.size());
  for (int b = 0; b < batch; ++b) std::copy(host.begin(), host.end(), hb.begin() + 2 * static_cast<size_t>(n) * b);
  if (auto e = bh.CopyDataHostToDevice(hb.data())) { std::printf("%s\n", e->c_str()); return 1; }
  if (auto e = ComputeFFT(plan, bh, GetMaxNoOptInSharedMem(dev))) { std::printf("%s\n", e->c_str()); return 1; }
  if (auto e = bh.CopyResultsDeviceToHost(ob.data(), plan.results_in_results_)) { std::printf("%s\n", e->c_str()); return 1; }
  double e2 = 0;
  for (int b = 0; b < batch; ++b) e2 = std::fmax(e2, max_error(n, ob.data() + 2 * static_cast<size_t>(n) * b));

  std::printf("N=%d results_in_results=%d  max|err| single %.3e  batch(%d) %.3e\n", n, int(plan.results_in_results_), e1, batch, e2);
  const double tol = 2e-3;   // input rounding (fp16) + ~3 fp16 roundings on O(0.4) spectral lines
  return (e1 < tol && e2 < tol) ? 0 : 1;
}
