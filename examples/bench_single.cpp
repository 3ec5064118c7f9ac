// bench_single.cpp — the reference's single-transform benchmark (src/testing/benchmarks/FFTBenchSinlge.cu:11-15, Bench.h:91-142)
// over include/tensor_fft.hpp on MI355X: ONE transform per length, N = 2^12, 2^13, ... , 10 warm-up + 100 timed samples, each
// sample = CopyDataHostToDevice, synchronise, start the clock, ComputeFFT, synchronise, stop; average and sigma per length into
// BenchResults.dat (`N average_ns sigma_ns`, FileWriter.h:295-310; the TRUE mean, not the reference's sum / (n - 1),
// BenchUtil.h:41-48). The signal is the reference's: sine superposition, 10 harmonics, weights GetRandomWeights(10, 42) / (10, 4242).
//
// What the wall clock of that protocol contains on this machine is mostly the host: an eager launch plus a device
// synchronisation costs ~10 us whatever the transform. The third and fourth columns therefore give the DEVICE time of the same
// transform: 16 executions captured in one HIP graph, replayed, per transform (what bench.py reports as
// other_configs["reference_protocol_single"]).
//
// usage: bench_single [max_log2 = 26] [tuner_file]      with a tuner file the plans come from CreatePlan(N, file), as in
//        the reference (`"TunerResults.dat"`, FFTBenchSinlge.cu:30), and the file is also loaded as the library's wisdom
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <random>
#include <vector>

#include "tensor_fft.hpp"

namespace {

std::vector<float> GetRandomWeights(int max_frequencies, int seed) {      // TestingDataCreation.h:15-27
  std::seed_seq seq = {seed};
  std::default_random_engine generator(seq);
  std::uniform_real_distribution<float> distro(-1.0, 1.0);
  std::vector<float> w;
  for (int i = 0; i < max_frequencies; ++i) w.push_back(distro(generator));
  return w;
}

// x[t] = sum_f w[f] sin(2 pi f t / N) per plane (TestingDataCreation.h:89-117), by the angle-addition recurrence per harmonic
// so that 2^29 samples do not take minutes of host sinf (the values only feed a timing)
std::unique_ptr<__half[]> CreateSineSuperposition(size_t n, const std::vector<float>& w_re, const std::vector<float>& w_im, int cutoff) {
  auto data = std::make_unique<__half[]>(2 * n);
  std::vector<float> re(n, 0.f), im(n, 0.f);
  for (int f = 1; f < cutoff; ++f) {
    const double d = 2 * M_PI * f / static_cast<double>(n), cd = std::cos(d), sd = std::sin(d);
    double s = 0, c = 1;
    for (size_t t = 0; t < n; ++t) {
      if ((t & 4095) == 0) {                                  // re-anchor the recurrence
        s = std::sin(d * static_cast<double>(t));
        c = std::cos(d * static_cast<double>(t));
      }
      re[t] += w_re[f] * static_cast<float>(s);
      im[t] += w_im[f] * static_cast<float>(s);
      const double ns = s * cd + c * sd;
      c = c * cd - s * sd;
      s = ns;
    }
  }
  for (size_t t = 0; t < n; ++t) {
    data[t] = __float2half(re[t]);
    data[t + n] = __float2half(im[t]);
  }
  return data;
}

struct BenchResult {
  double average_time_, std_deviation_, device_average_, device_min_;
};

}  // namespace

int main(int argc, char** argv) {
  const int max_lg = argc > 1 ? std::atoi(argv[1]) : 26;
  const char* tuner_file = argc > 2 ? argv[2] : nullptr;
  constexpr int start_lg = 12;                 // FFTBenchSinlge.cu:11 start_fft_length = 16^3
  constexpr int sample_size = 100, warmup_samples = 10;
  if (tuner_file) {
    int taken = 0;
    if (tfft_tuning_load(tuner_file, &taken) != TFFT_OK) {
      std::cout << tfft_last_error() << std::endl;
      return 1;
    }
    std::printf("# wisdom: %d tuner lines from %s\n", taken, tuner_file);
  }
  const std::vector<float> w_re = GetRandomWeights(10, 42), w_im = GetRandomWeights(10, 4242);
  std::vector<long long> fft_length;
  std::vector<BenchResult> bench_data;
  std::printf("# N  wall_average_ns  wall_sigma_ns  device_us_per_transform(graph of 16: mean, best)\n");
  for (int lg = start_lg; lg <= max_lg; ++lg) {
    const long long n = 1ll << lg;
    auto data = CreateSineSuperposition(static_cast<size_t>(n), w_re, w_im, 10);
    std::optional<Plan<long long>> possible_plan =
        tuner_file ? CreatePlan(n, std::string(tuner_file)) : CreatePlan(n, n >= 4096 ? Mode_4096 : Mode_256, n >= 4096 ? 16 : 1, 1, 256);
    if (!possible_plan) {
      std::cout << "Plan creation failed" << std::endl;
      return 1;
    }
    Plan<long long> my_plan = possible_plan.value();
    int device_id = 0;
    (void)hipGetDevice(&device_id);
    if (!PlanWorksOnDevice(my_plan, device_id)) return 1;
    const int max_no_optin_shared_mem = GetMaxNoOptInSharedMem(device_id);
    DataHandler<long long> my_handler(n);
    if (auto e = my_handler.PeakAtLastError()) {
      std::cout << e.value() << std::endl;
      return 1;
    }
    std::vector<double> runtime;
    for (int k = 0; k < sample_size + warmup_samples; ++k) {
      if (auto e = my_handler.CopyDataHostToDevice(data.get())) {
        std::cout << e.value() << std::endl;
        return 1;
      }
      (void)hipDeviceSynchronize();
      const auto t0 = std::chrono::steady_clock::now();
      if (auto e = ComputeFFT(my_plan, my_handler, max_no_optin_shared_mem)) {
        std::cout << e.value() << std::endl;
        return 1;
      }
      (void)hipDeviceSynchronize();
      const auto t1 = std::chrono::steady_clock::now();
      if (k >= warmup_samples) runtime.push_back(std::chrono::duration<double, std::nano>(t1 - t0).count());
    }
    BenchResult r{0, 0, 0, 0};
    for (double v : runtime) r.average_time_ += v / runtime.size();
    for (double v : runtime) r.std_deviation_ += (v - r.average_time_) * (v - r.average_time_);
    r.std_deviation_ = std::sqrt(r.std_deviation_ / (runtime.size() - 1));
    // ---- device time: the same ComputeFFT, 16 per graph (the input half is scratch for multi-pass lengths, as in the reference:
    // the values degrade from replay to replay, the timing does not care)
    {
      hipStream_t s;
      (void)hipStreamCreate(&s);
      std::string err;
      const auto tuned = tfft_detail::tuned_for_batch(my_plan, 1);
      tfft_plan* p = tfft_detail::exec_plan(static_cast<uint64_t>(n), 1, tuned.first, &err, tuned.second);
      if (!p) {
        std::cout << err << std::endl;
        return 1;
      }
      __half* out_re = my_plan.results_in_results_ ? my_handler.dptr_results_RE_ : my_handler.dptr_input_RE_;
      __half* out_im = my_plan.results_in_results_ ? my_handler.dptr_results_IM_ : my_handler.dptr_input_IM_;
      hipGraph_t g;
      hipGraphExec_t ge;
      (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
      for (int i = 0; i < 16; ++i) (void)tfft_exec(p, my_handler.dptr_input_RE_, my_handler.dptr_input_IM_, out_re, out_im, s);
      if (hipStreamEndCapture(s, &g) != hipSuccess || hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) {
        std::cout << "graph capture failed" << std::endl;
        return 1;
      }
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(ge, s);
      (void)hipStreamSynchronize(s);
      double sum = 0, best = 1e30;
      const int rounds = 7, reps = lg <= 22 ? 8 : 2;
      for (int rr = 0; rr < rounds; ++rr) {
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < reps; ++i) (void)hipGraphLaunch(ge, s);
        (void)hipEventRecord(e1, s);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps / 16;
        sum += us / rounds;
        best = std::min(best, us);
      }
      r.device_average_ = sum;
      r.device_min_ = best;
      (void)hipGraphExecDestroy(ge);
      (void)hipGraphDestroy(g);
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
      (void)hipStreamDestroy(s);
    }
    std::printf("%lld %.0f %.0f %.2f %.2f\n", n, r.average_time_, r.std_deviation_, r.device_average_, r.device_min_);
    std::fflush(stdout);
    fft_length.push_back(n);
    bench_data.push_back(r);
    ReleaseComputeFFTPlans();                    // (a 2^26 plan holds 512 MiB of workspace)
  }
  std::ofstream myfile("BenchResults.dat");      // FileWriter.h:295-310
  if (!myfile.is_open()) {
    std::cout << "Error! Unable to open file." << std::endl;
    return 1;
  }
  for (size_t i = 0; i < bench_data.size(); ++i)
    myfile << fft_length[i] << " " << bench_data[i].average_time_ << " " << bench_data[i].std_deviation_ << "\n";
  return 0;
}
