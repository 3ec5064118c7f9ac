// copy_order_check.cpp — ComputeFFT is asynchronous (src/base/ComputeFFT.h:49-53); a CopyDataHostToDevice that follows it without
// a synchronisation must still not overwrite the input under the running kernels: the reference's blocking cudaMemcpy on the
// null stream orders itself behind them (src/base/DataHandler.h:45-53). Both copy paths of include/tensor_fft.hpp (one blocking
// hipMemcpy; the library's pinned ring, tfft_copy_h2d) are driven with a long queue of transforms in front of the copy:
// the LAST queued transform must still see signal A. exit 0 / 1.
//
// usage: copy_order_check [log2_N = 14] [queued transforms = 2000]   (a length whose transform leaves the input block intact: the
// result must land in the RESULTS half - Plan::results_in_results_, Plan.h:109-115: 2^12, 2^14, 2^17 ... - and the plan must not use
// the input as scratch: a single kernel (to 2^15) or an even number of passes. A single 2^17 .. 2^21 takes three passes since round
// 5, tfft_plan_default_variant, and uses the input block as scratch, the reference's contract, ComputeFFT.h:89-93.)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tensor_fft.hpp"

int main(int argc, char** argv) {
  const int lg = argc > 1 ? std::atoi(argv[1]) : 14;
  const int queued = argc > 2 ? std::atoi(argv[2]) : 2000;
  const int n = 1 << lg;
  auto maybe_plan = CreatePlan(n, Mode_4096, 16, 1, 256);
  if (!maybe_plan) return 1;
  Plan<int> plan = maybe_plan.value();
  std::vector<__half> a(2 * static_cast<size_t>(n)), b(a.size()), want_a(a.size()), want_b(a.size()), got(a.size());
  unsigned s = 12345u;
  for (size_t i = 0; i < a.size(); ++i) {
    s = s * 1664525u + 1013904223u;
    a[i] = __float2half(static_cast<float>(s >> 8) / 8388608.0f - 1.0f);
    s = s * 1664525u + 1013904223u;
    b[i] = __float2half(static_cast<float>(s >> 8) / 8388608.0f - 1.0f);
  }
  DataHandler<int> h(n);
  auto fail = [](const char* what, const std::string& e) { std::printf("%s: %s\n", what, e.c_str()); return 1; };
  // reference spectra, everything synchronised
  for (int which = 0; which < 2; ++which) {
    if (auto e = h.CopyDataHostToDevice(which ? b.data() : a.data())) return fail("copy", *e);
    if (auto e = ComputeFFT(plan, h)) return fail("ComputeFFT", *e);
    (void)hipDeviceSynchronize();
    if (auto e = h.CopyResultsDeviceToHost(which ? want_b.data() : want_a.data(), plan.results_in_results_)) return fail("copy back", *e);
  }
  if (std::memcmp(want_a.data(), want_b.data(), a.size() * sizeof(__half)) == 0) return fail("setup", "the two signals give the same spectrum");
  int bad = 0;
  for (int staged = 0; staged < 2; ++staged) {
    SetStagedCopies(staged != 0);
    if (auto e = h.CopyDataHostToDevice(a.data())) return fail("copy", *e);
    (void)hipDeviceSynchronize();
    // a queue of transforms of A (each re-reads the input block; an even pass count leaves it intact), then B on top WITHOUT a sync
    for (int q = 0; q < queued; ++q)
      if (auto e = ComputeFFT(plan, h)) return fail("ComputeFFT", *e);
    if (auto e = h.CopyDataHostToDevice(b.data())) return fail("copy", *e);
    // the result buffer now holds what the LAST queued transform computed; it has to be the spectrum of A
    if (auto e = h.CopyResultsDeviceToHost(got.data(), plan.results_in_results_)) return fail("copy back", *e);
    const bool ok_a = std::memcmp(got.data(), want_a.data(), a.size() * sizeof(__half)) == 0;
    // ... and B did arrive
    if (auto e = ComputeFFT(plan, h)) return fail("ComputeFFT", *e);
    if (auto e = h.CopyResultsDeviceToHost(got.data(), plan.results_in_results_)) return fail("copy back", *e);
    const bool ok_b = std::memcmp(got.data(), want_b.data(), a.size() * sizeof(__half)) == 0;
    std::printf("%s copies: last queued transform saw %s, the new input %s\n", staged ? "staged (pinned ring)" : "plain hipMemcpy    ",
                ok_a ? "signal A (ordered)" : "OVERWRITTEN INPUT", ok_b ? "arrived" : "DID NOT ARRIVE");
    bad += !(ok_a && ok_b);
  }
  SetStagedCopies(false);
  if (!bad) std::printf("OK\n");
  return bad ? 1 : 0;
}
