// example_batch_fft.cpp — a batch of transforms through the reference's host interface, with the calls in the order the
// reference's batch example makes them (src/testing/ExampleBatchFFT.cu:20-85):
//   CreatePlan(N)  ->  PlanWorksOnDevice  ->  DataBatchHandler(N, batch)  ->  PeakAtLastError  ->  CopyDataHostToDevice
//   ->  ComputeFFT(plan, handler, GetMaxNoOptInSharedMem(device))  ->  CopyResultsDeviceToHost(data, results_in_results_)
// A program written against src/base/{Plan,DataHandler,ComputeFFT}.h needs only its include line changed to build
// against include/tensor_fft.hpp. Unlike the reference's example this one checks itself: batch entry b is the complex
// tone 0.5 exp(+2 pi i f_b n / N), whose scaled spectrum is 0.5 at bin f_b and zero elsewhere (exit code 0 = all right).
#include <cmath>
#include <cstdio>
#include <optional>
#include <string>
#include <vector>

#include "tensor_fft.hpp"

namespace {
constexpr int kN = 16 * 16 * 16;
constexpr int kBatch = 20;
int tone_of(int b) { return 3 + 7 * b; }

bool failed(const std::optional<std::string>& e, const char* what) {
  if (e) std::printf("%s: %s\n", what, e->c_str());
  return e.has_value();
}
}  // namespace

int main() {
  std::vector<__half> host(2 * static_cast<size_t>(kN) * kBatch);
  for (int b = 0; b < kBatch; ++b) {
    __half* plane_re = host.data() + 2 * static_cast<size_t>(kN) * b;
    __half* plane_im = plane_re + kN;
    for (int t = 0; t < kN; ++t) {
      const double phase = 2.0 * M_PI * static_cast<double>((static_cast<long>(tone_of(b)) * t) % kN) / kN;
      plane_re[t] = __float2half(static_cast<float>(0.5 * std::cos(phase)));
      plane_im[t] = __float2half(static_cast<float>(0.5 * std::sin(phase)));
    }
  }

  const std::optional<Plan<int>> maybe_plan = CreatePlan(kN);          // default mode and launch parameters
  if (!maybe_plan) {
    std::printf("Plan creation failed\n");
    return 1;
  }
  Plan<int> plan = *maybe_plan;

  int device = 0;
  (void)hipGetDevice(&device);
  if (!PlanWorksOnDevice(plan, device)) return 1;

  DataBatchHandler handler(kN, kBatch);                                // class template argument deduced, as in the reference
  if (failed(handler.PeakAtLastError(), "allocation")) return 1;
  if (failed(handler.CopyDataHostToDevice(host.data()), "host -> device")) return 1;
  if (failed(ComputeFFT(plan, handler, GetMaxNoOptInSharedMem(device)), "ComputeFFT")) return 1;
  if (failed(handler.CopyResultsDeviceToHost(host.data(), plan.results_in_results_), "device -> host")) return 1;
  (void)hipDeviceSynchronize();

  double worst = 0;
  for (int b = 0; b < kBatch; ++b) {
    const __half* re = host.data() + 2 * static_cast<size_t>(kN) * b;
    const __half* im = re + kN;
    for (int k = 0; k < kN; ++k) {
      worst = std::fmax(worst, std::fabs(__half2float(re[k]) - (k == tone_of(b) ? 0.5 : 0.0)));
      worst = std::fmax(worst, std::fabs(__half2float(im[k])));
    }
  }
  std::printf("batch of %d transforms of length %d: max |error| = %.3g\n", kBatch, kN, worst);
  return worst < 2e-3 ? 0 : 1;
}
