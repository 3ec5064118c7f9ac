// bench_batched.cpp — C++ host driver over include/tensor_fft.hpp, the counterpart of the reference's batch
// benchmark mains (src/testing/benchmarks/FFTBenchBatch.cu, Bench.h:91-142: create plan and handler, copy the signal
// in once, time sample_size executions of ComputeFFT with the H2D copy excluded, report the mean and sigma).
// It times exactly what bench.py times (one tfft_exec per step over a resident batch), without Python in the loop.
//
// usage: bench_batched [log2_N = 12] [batch = 65536] [samples = 200] [warmup = 100]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "tensor_fft.hpp"

int main(int argc, char** argv) {
  const int lg = argc > 1 ? std::atoi(argv[1]) : 12;
  const int batch = argc > 2 ? std::atoi(argv[2]) : 65536;
  const int samples = argc > 3 ? std::atoi(argv[3]) : 200;
  const int warmup = argc > 4 ? std::atoi(argv[4]) : 100;
  const int n = 1 << lg;
  const BaseFFTMode mode = n >= 4096 ? Mode_4096 : Mode_256;
  auto maybe_plan = CreatePlan(n, mode, mode == Mode_4096 ? 16 : 1, 1, 256);
  if (!maybe_plan) return 1;
  Plan<int> plan = maybe_plan.value();
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!PlanWorksOnDevice(plan, dev)) return 1;

  DataBatchHandler<int> data(n, batch);
  if (auto e = data.PeakAtLastError()) { std::printf("%s\n", e->c_str()); return 1; }
  {
    // 4096 distinct random transforms, repeated over the batch (the values do not affect the timing)
    const size_t unit = 2 * static_cast<size_t>(n), distinct = static_cast<size_t>(batch < 4096 ? batch : 4096);
    std::vector<__half> block(unit * distinct);
    std::mt19937 gen(42);
    std::uniform_real_distribution<float> dist(-1.f, 1.f);
    for (auto& v : block) v = __float2half(dist(gen));
    std::vector<__half> host(unit * batch);
    for (size_t b = 0; b < static_cast<size_t>(batch); ++b)
      std::copy(block.begin() + (b % distinct) * unit, block.begin() + (b % distinct + 1) * unit, host.begin() + b * unit);
    if (auto e = data.CopyDataHostToDevice(host.data())) { std::printf("%s\n", e->c_str()); return 1; }
    // (c) the step either side of the path: the handlers' host <-> device copies of PAGEABLE memory through the pinned staging
    // ring (tfft_copy_h2d / _d2h) next to one plain blocking hipMemcpy of the same bytes (what the reference's handlers do,
    // src/base/DataHandler.h:116-153). Bound: PCIe Gen5 x16, ~63 GB/s per direction.
    const double gb = static_cast<double>(host.size()) * sizeof(__half) * 1e-9;
    auto secs = [](auto&& fn) {
      const auto t0 = std::chrono::steady_clock::now();
      fn();
      (void)hipDeviceSynchronize();
      return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    std::vector<__half> back(host.size());
    double h2d_staged = 1e30, h2d_plain = 1e30, d2h_staged = 1e30, d2h_plain = 1e30;
    for (int r = 0; r < 3; ++r) {
      h2d_staged = std::min(h2d_staged, secs([&] { (void)data.CopyDataHostToDevice(host.data()); }));
      h2d_plain = std::min(h2d_plain, secs([&] { (void)hipMemcpy(data.dptr_input_RE_[0], host.data(), host.size() * sizeof(__half), hipMemcpyHostToDevice); }));
      d2h_staged = std::min(d2h_staged, secs([&] { (void)data.CopyResultsDeviceToHost(back.data(), false); }));
      d2h_plain = std::min(d2h_plain, secs([&] { (void)hipMemcpy(back.data(), data.dptr_input_RE_[0], host.size() * sizeof(__half), hipMemcpyDeviceToHost); }));
    }
    if (std::memcmp(back.data(), host.data(), host.size() * sizeof(__half)) != 0) { std::printf("staged copy round trip differs\n"); return 1; }
    std::printf("host <-> device, %.2f GB pageable: H2D staged %.1f GB/s (plain hipMemcpy %.1f), D2H staged %.1f GB/s (plain %.1f); "
                "PCIe Gen5 x16 bound ~63 GB/s\n", gb, gb / h2d_staged, gb / h2d_plain, gb / d2h_staged, gb / d2h_plain);
  }
  const int smem = GetMaxNoOptInSharedMem(dev);
  for (int i = 0; i < warmup; ++i)
    if (auto e = ComputeFFT(plan, data, smem)) { std::printf("%s\n", e->c_str()); return 1; }

  // (a) per-call wall time, as the reference measures it (ComputeFFT's batch overload ends with a device sync)
  std::vector<double> us(samples);
  for (int i = 0; i < samples; ++i) {
    const auto t0 = std::chrono::steady_clock::now();
    if (auto e = ComputeFFT(plan, data, smem)) { std::printf("%s\n", e->c_str()); return 1; }
    us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  }
  double mean = 0, var = 0;
  for (double v : us) mean += v;
  mean /= samples;
  for (double v : us) var += (v - mean) * (v - mean);
  const double sigma = samples > 1 ? std::sqrt(var / (samples - 1)) : 0.0;
  std::vector<double> sorted(us);
  std::sort(sorted.begin(), sorted.end());
  const double median = sorted[sorted.size() / 2];

  // (b) back-to-back launches through the C ABI on one stream, HIP events around the burst (what bench.py reports)
  tfft_plan* p = nullptr;
  if (tfft_plan_create(static_cast<uint64_t>(n), static_cast<uint64_t>(batch), dev, nullptr, &p) != TFFT_OK) {
    std::printf("%s\n", tfft_last_error());
    return 1;
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, nullptr);
  for (int i = 0; i < samples; ++i)
    if (tfft_exec(p, data.dptr_input_RE_[0], data.dptr_input_IM_[0], data.dptr_results_RE_[0], data.dptr_results_IM_[0],
                  nullptr) != TFFT_OK) {
      std::printf("%s\n", tfft_last_error());
      return 1;
    }
  (void)hipEventRecord(e1, nullptr);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double kernel_us = 1e3 * ms / samples;
  const double total = static_cast<double>(n) * batch;
  std::printf("N=%d batch=%d passes=%d: ComputeFFT+sync mean %.1f us (sigma %.1f, median %.1f) = %.1f Gsamples/s; "
              "back-to-back %.1f us = %.1f Gsamples/s, %.0f GB/s algorithmic per pass\n",
              n, batch, tfft_plan_num_launches(p), mean, sigma, median, total / mean * 1e-3, kernel_us, total / kernel_us * 1e-3,
              8.0 * total * tfft_plan_num_launches(p) / kernel_us * 1e-3);
  tfft_plan_destroy(p);
  return 0;
}
