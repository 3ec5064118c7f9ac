// example_multi_gpu_fft.cpp — ONE transform spread over the GPUs this process sees (BASELINE configs[4b]) and one batch
// sharded over them (configs[4a]), through the reference-shaped multi-GPU interface of include/tensor_fft.hpp
// (DataHandlerMultiGPU / ComputeFFTMultiGPU, DataBatchHandlerMultiGPU / ComputeFFTsMultiGPU; the reference's own versions
// are commented out, src/base/ComputeFFT.h:295-557). Host code is C++ over the C ABI only: RCCL is reached through libtfft.so.
// Checks the spectra of an integer-frequency tone mix against their closed form and exits 0 / 1.
//
// usage: example_multi_gpu_fft [log2_N = 20] [devices = all (largest power of two)] [via = 0 | 1]
//   via = 1: the own chunk goes through ncclSend / ncclRecv as well, so that a box with ONE GPU still runs
//            kernel -> RCCL collective -> kernel on one stream (real communicator with one rank).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "tensor_fft.hpp"

namespace {
constexpr int kHarmonics = 6;
const double kA[kHarmonics] = {0.0, 0.8125, 0.25, -0.625, 0.5, -0.75};
const double kB[kHarmonics] = {0.0, 0.6875, -0.375, 0.5, -0.5625, 0.3125};

// re[t] = sum a_f sin(2 pi f t / N), im[t] = sum b_f sin(2 pi f t / N)  =>  X[f] = (b_f - i a_f) / 2, X[N - f] = -X[f]
void make_signal(size_t n, __half* dst) {
  for (size_t t = 0; t < n; ++t) {
    double re = 0, im = 0;
    for (int f = 1; f < kHarmonics; ++f) {
      const double s = std::sin(2.0 * M_PI * static_cast<double>((f * t) % n) / static_cast<double>(n));
      re += kA[f] * s;
      im += kB[f] * s;
    }
    dst[t] = __float2half(static_cast<float>(re));
    dst[t + n] = __float2half(static_cast<float>(im));
  }
}

double max_error(size_t n, const __half* got) {
  double worst = 0;
  for (size_t k = 0; k < n; ++k) {
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
  const int lg = argc > 1 ? std::atoi(argv[1]) : 20;
  int want = argc > 2 ? std::atoi(argv[2]) : 0;
  const bool via = argc > 3 && std::atoi(argv[3]) != 0;
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || have < 1) { std::printf("no GPU\n"); return 1; }
  int nd = 1;
  while (2 * nd <= have && (want == 0 || 2 * nd <= want)) nd *= 2;
  std::vector<int> devices;
  for (int i = 0; i < nd; ++i) devices.push_back(i);
  const size_t n = size_t{1} << lg;

  auto maybe_plan = CreatePlan(static_cast<long long>(n), Mode_4096, 16, 1, 256);
  if (!maybe_plan) return 1;
  Plan<long long> plan = maybe_plan.value();
  for (int d : devices) if (!PlanWorksOnDevice(plan, d)) return 1;

  std::vector<__half> host(2 * n), out(2 * n);
  make_signal(n, host.data());

  // ---- one transform over all devices
  DataHandlerMultiGPU<long long> handler(static_cast<long long>(n), devices, via);
  if (auto e = handler.PeakAtLastError()) { std::printf("%s\n", e->c_str()); return 1; }
  if (auto e = handler.CopyDataHostToDevice(host.data())) { std::printf("%s\n", e->c_str()); return 1; }
  if (auto e = ComputeFFTMultiGPU(plan, handler)) { std::printf("%s\n", e->c_str()); return 1; }
  if (auto e = handler.CopyResultsDeviceToHost(out.data())) { std::printf("%s\n", e->c_str()); return 1; }
  const double e1 = max_error(n, out.data());
  // timing: 20 transforms back to back, all devices
  for (int d : devices) { (void)hipSetDevice(d); (void)hipDeviceSynchronize(); }
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < 20; ++r)
    if (auto e = ComputeFFTMultiGPU(plan, handler)) { std::printf("%s\n", e->c_str()); return 1; }
  for (int d : devices) { (void)hipSetDevice(d); (void)hipDeviceSynchronize(); }
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 20;
  const tfft_dist_geometry& g = handler.geometry_;
  std::printf("one transform, N = 2^%d over %d device(s)%s: N1 = %llu, N2 = %llu, %llu columns and %llu rows per device, re-order pass %d, "
              "%d local passes, %.3f ms per transform (%.1f Gsamples/s), max |error| = %.3e\n",
              lg, nd, via ? " (own chunk through RCCL)" : "", static_cast<unsigned long long>(g.n1), static_cast<unsigned long long>(g.n2),
              static_cast<unsigned long long>(g.cols), static_cast<unsigned long long>(g.rows), g.reorder, g.local_passes, ms,
              static_cast<double>(n) / ms / 1e6, e1);

  // ---- one batch of N = 4096 sharded over the devices (no collective)
  const int bn = 4096, batch = 64 * nd + 3;      // ragged on purpose
  auto bplan = CreatePlan(bn, Mode_4096, 16, 1, 256);
  if (!bplan) return 1;
  std::vector<__half> hb(2 * static_cast<size_t>(bn) * batch), ob(hb.size());
  std::vector<__half> one(2 * static_cast<size_t>(bn));
  make_signal(bn, one.data());
  for (int b = 0; b < batch; ++b) std::copy(one.begin(), one.end(), hb.begin() + 2 * static_cast<size_t>(bn) * b);
  DataBatchHandlerMultiGPU<int> bh(bn, batch, devices);
  if (auto e = bh.PeakAtLastError()) { std::printf("%s\n", e->c_str()); return 1; }
  if (auto e = bh.CopyDataHostToDevice(hb.data())) { std::printf("%s\n", e->c_str()); return 1; }
  if (auto e = ComputeFFTsMultiGPU(bplan.value(), bh)) { std::printf("%s\n", e->c_str()); return 1; }
  if (auto e = bh.CopyResultsDeviceToHost(ob.data(), bplan->results_in_results_)) { std::printf("%s\n", e->c_str()); return 1; }
  double e2 = 0;
  for (int b = 0; b < batch; ++b) e2 = std::fmax(e2, max_error(bn, ob.data() + 2 * static_cast<size_t>(bn) * b));
  std::printf("batch of %d x N = %d sharded over %d device(s): max |error| = %.3e\n", batch, bn, nd, e2);

  const bool ok = e1 < 2e-3 && e2 < 2e-3;
  std::printf(ok ? "OK\n" : "FAILED\n");
  return ok ? 0 : 1;
}
