// example_batch_fft.cpp — a batch of transforms through the reference's host interface, in the order the reference's own
// batch example makes its calls (src/testing/ExampleBatchFFT.cu:20-85): CreatePlan(fft_length) with the default mode ->
// PlanWorksOnDevice -> DataBatchHandler(fft_length, batch_size) -> PeakAtLastError -> CopyDataHostToDevice ->
// ComputeFFT(plan, handler, GetMaxNoOptInSharedMem(device)) -> CopyResultsDeviceToHost(data, plan.results_in_results_)
// -> device synchronise. Only the include line and the signal differ from a program written against the reference's
// src/base headers: here each batch entry is one complex tone exp(+2 pi i f_b n / N) of amplitude 1/2, whose scaled
// spectrum is 1/2 at bin f_b and zero elsewhere, so the program can check itself (exit code 0 = all entries right).
#include <cassert>
#include <cmath>
#include <iostream>
#include <memory>
#include <optional>
#include <string>

#include "tensor_fft.hpp"

int main() {
  constexpr int fft_length = 16 * 16 * 16;
  constexpr int batch_size = 20;

  std::unique_ptr<__half[]> data(new __half[2 * static_cast<size_t>(fft_length) * batch_size]);
  for (int b = 0; b < batch_size; ++b) {
    const int f = 3 + 7 * b;
    __half* re = data.get() + 2 * static_cast<size_t>(fft_length) * b;
    __half* im = re + fft_length;
    for (int n = 0; n < fft_length; ++n) {
      const double ph = 2.0 * M_PI * static_cast<double>((static_cast<long>(f) * n) % fft_length) / fft_length;
      re[n] = __float2half(static_cast<float>(0.5 * std::cos(ph)));
      im[n] = __float2half(static_cast<float>(0.5 * std::sin(ph)));
    }
  }

  std::optional<std::string> error_mess;

  std::optional<Plan<int>> possible_plan = CreatePlan(fft_length);
  Plan<int> my_plan;
  if (possible_plan) {
    my_plan = possible_plan.value();
  } else {
    std::cout << "Plan creation failed" << std::endl;
    return 1;
  }

  int device_id;
  (void)hipGetDevice(&device_id);
  assert((PlanWorksOnDevice(my_plan, device_id)));

  DataBatchHandler my_handler(fft_length, batch_size);
  error_mess = my_handler.PeakAtLastError();
  if (error_mess) {
    std::cout << error_mess.value() << std::endl;
    return 1;
  }

  error_mess = my_handler.CopyDataHostToDevice(data.get());
  if (error_mess) {
    std::cout << error_mess.value() << std::endl;
    return 1;
  }

  error_mess = ComputeFFT(my_plan, my_handler, GetMaxNoOptInSharedMem(device_id));
  if (error_mess) {
    std::cout << error_mess.value() << std::endl;
    return 1;
  }

  error_mess = my_handler.CopyResultsDeviceToHost(data.get(), my_plan.results_in_results_);
  if (error_mess) {
    std::cout << error_mess.value() << std::endl;
    return 1;
  }

  (void)hipDeviceSynchronize();

  double worst = 0;
  for (int b = 0; b < batch_size; ++b) {
    const int f = 3 + 7 * b;
    const __half* re = data.get() + 2 * static_cast<size_t>(fft_length) * b;
    const __half* im = re + fft_length;
    for (int k = 0; k < fft_length; ++k) {
      worst = std::fmax(worst, std::fabs(__half2float(re[k]) - (k == f ? 0.5 : 0.0)));
      worst = std::fmax(worst, std::fabs(__half2float(im[k])));
    }
  }
  std::cout << "batch of " << batch_size << " transforms of length " << fft_length << ": max |error| = " << worst << std::endl;
  return worst < 2e-3 ? 0 : 1;
}
