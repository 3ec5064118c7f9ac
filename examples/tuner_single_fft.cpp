// tuner_single_fft.cpp — the reference's tuner main (src/testing/benchmarks/TunerSingleFFT.cu:10-56, BenchUtil.h:60-150) over
// include/tensor_fft.hpp on MI355X. Same protocol: for every length from 256 on, x 2 per step, build the search space, time every
// configuration of it (sample_size timed samples after warmup_samples untimed ones; a sample = ComputeFFT + synchronise on a
// resident signal, Bench.h:121-142), keep the fastest (GetFastestConfig: smallest average), write the winners to
// TunerResults.dat. What is searched is this library's knobs, not the reference's block sizes (which mean nothing on gfx950):
//   variant       (tfft_plan_opts.variant: which of several equivalent kernels / splits; every value is checked with
//                  tfft_variant_check, so a configuration that this library would refuse is never timed)
//   launch_iters  (rounds per workgroup: 0 = the library's default shape, 1, 2, 4, 8, TFFT_LAUNCH_PERSISTENT)
// and the file keeps the reference's five columns (`N mode base_wpb r16_wpb r2_blocksize`, FileWriter.h:250-269) in front of
// the three this library reads: `variant launch_iters batch`. Both CreatePlan(N, "TunerResults.dat") of the shim and
// tfft_tuning_load("TunerResults.dat") of the C ABI take it; a reference build would read its five columns and ignore the rest.
// A configuration only replaces the default (variant 0, launch_iters 0) when it wins by more than `margin` (default 5 %:
// twice the run-to-run spread of one box, profiles/r4_buffer_offsets.txt) in BOTH of two interleaved rounds.
//
// usage: tuner_single_fft [max_log2 = 24] [batch = 1] [out = TunerResults.dat] [margin_percent = 5]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <random>
#include <vector>

#include "tensor_fft.hpp"

namespace {

struct RunConfig {
  int variant_, launch_iters_;
};
struct RunResults {
  RunConfig config_;
  double average_time_, std_deviation_;      // ns
};

// the knobs that exist for this length (BenchUtil.h:77-104 GetSearchSpace: what the device allows)
std::vector<RunConfig> GetRunConfigs(long long fft_length) {
  static const int kVariants[] = {0, 32, 524288, 8388608, 33554432, 8388608 | 33554432, 134217728, 16777216, 2097152, 268435456,
                                  1048576, 262144, 536870912, 1073741824, 1073741824 | 8388608 | 33554432,
                                  8388608 | 33554432 | 16777216};      // (2^15 as 256 x 128 with the cooperative radix-128 pass)
  static const int kIters[] = {0, 1, 2, 4, 8, TFFT_LAUNCH_PERSISTENT};
  std::vector<RunConfig> configs;
  char base[256], desc[256];
  if (tfft_plan_describe(static_cast<uint64_t>(fft_length), 1, 0, base, sizeof(base)) != TFFT_OK) return configs;
  std::vector<std::string> seen;
  for (int v : kVariants) {
    if (tfft_variant_check(static_cast<uint64_t>(fft_length), 1, v) != TFFT_OK) continue;
    // a planner bit that does not change this length's decomposition and no kernel bit either: the same plan again
    const bool planner_bit = v & (32 | 8388608 | 33554432 | 134217728 | 16777216 | 2097152);
    if (tfft_plan_describe(static_cast<uint64_t>(fft_length), 1, v, desc, sizeof(desc)) != TFFT_OK) continue;
    if (v && planner_bit && !(v & ~(32 | 8388608 | 33554432 | 134217728 | 16777216 | 2097152)) && std::string(desc) == base) continue;
    for (int it : kIters) {
      if (it && v) continue;                  // launch shapes are searched on the default kernels only
      configs.push_back(RunConfig{v, it});
    }
  }
  return configs;
}

double Sample(tfft_plan* p, const DataBatchHandler<long long>& data, bool in_results) {
  __half* out_re = in_results ? data.dptr_results_RE_[0] : data.dptr_input_RE_[0];
  __half* out_im = in_results ? data.dptr_results_IM_[0] : data.dptr_input_IM_[0];
  (void)hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  (void)tfft_exec(p, data.dptr_input_RE_[0], data.dptr_input_IM_[0], out_re, out_im, nullptr);
  (void)hipDeviceSynchronize();
  return std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

int main(int argc, char** argv) {
  const int max_lg = argc > 1 ? std::atoi(argv[1]) : 24;
  const int batch = argc > 2 ? std::atoi(argv[2]) : 1;
  const char* out_path = argc > 3 ? argv[3] : "TunerResults.dat";
  const double margin = (argc > 4 ? std::atof(argv[4]) : 5.0) / 100.0;
  constexpr int sample_size = 100, warmup_samples = 5;        // TunerSingleFFT.cu:14-15
  constexpr int device_id = 0;
  (void)hipSetDevice(device_id);
  std::vector<long long> fft_length;
  std::vector<RunConfig> optimal_config;
  std::ofstream tuner_data("TunerData.dat");                   // every configuration's time (WriteTunerDataToFile)
  for (int lg = 8; lg <= max_lg; ++lg) {
    const long long n = 1ll << lg;
    std::cout << "Current fft_length: " << n << std::endl;
    auto plan = CreatePlan(n, n >= 4096 ? Mode_4096 : Mode_256, n >= 4096 ? 16 : 1, 1, 256);
    if (!plan || !PlanWorksOnDevice(plan.value(), device_id)) return 1;
    DataBatchHandler<long long> data(n, batch);
    if (auto e = data.PeakAtLastError()) {
      std::cout << e.value() << std::endl;
      return 1;
    }
    {
      std::vector<__half> host(static_cast<size_t>(batch) * 2 * n);
      std::mt19937 gen(42);
      std::uniform_real_distribution<float> dist(-1.f, 1.f);
      for (auto& v : host) v = __float2half(dist(gen));
      if (auto e = data.CopyDataHostToDevice(host.data())) {
        std::cout << e.value() << std::endl;
        return 1;
      }
    }
    const std::vector<RunConfig> configs = GetRunConfigs(n);
    std::vector<tfft_plan*> plans;
    for (const RunConfig& c : configs) {
      tfft_plan_opts o = TFFT_PLAN_OPTS_INIT;
      o.variant = c.variant_;
      o.launch_iters = static_cast<uint32_t>(c.launch_iters_);
      o.preserve_input = 1;                                    // every configuration sees the same signal
      tfft_plan* p = nullptr;
      if (tfft_plan_create(static_cast<uint64_t>(n), static_cast<uint64_t>(batch), device_id, &o, &p) != TFFT_OK || tfft_plan_prepare(p) != TFFT_OK) {
        if (p) tfft_plan_destroy(p);
        p = nullptr;
      }
      plans.push_back(p);
    }
    // RunBenchOverSearchSpace, twice, interleaved: round r of configuration c directly after round r of configuration c - 1
    std::vector<RunResults> bench_data[2];
    for (int round = 0; round < 2; ++round)
      for (size_t c = 0; c < configs.size(); ++c) {
        if (!plans[c]) continue;
        std::vector<double> runtime;
        for (int k = 0; k < sample_size + warmup_samples; ++k) {
          const double t = Sample(plans[c], data, true);
          if (k >= warmup_samples) runtime.push_back(t);
        }
        RunResults r{configs[c], 0, 0};
        for (double v : runtime) r.average_time_ += v / runtime.size();
        for (double v : runtime) r.std_deviation_ += (v - r.average_time_) * (v - r.average_time_);
        r.std_deviation_ = std::sqrt(r.std_deviation_ / (runtime.size() - 1));
        bench_data[round].push_back(r);
        tuner_data << n << " " << batch << " " << r.config_.variant_ << " " << r.config_.launch_iters_ << " " << r.average_time_ << " " << r.std_deviation_ << "\n";
      }
    for (tfft_plan* p : plans)
      if (p) tfft_plan_destroy(p);
    // GetFastestConfig, with the margin: the default stays unless something beats it by `margin` in both rounds
    RunConfig best{0, 0};
    double best_gain = margin;
    for (size_t i = 1; i < bench_data[0].size(); ++i) {
      const double g0 = 1.0 - bench_data[0][i].average_time_ / bench_data[0][0].average_time_;
      const double g1 = 1.0 - bench_data[1][i].average_time_ / bench_data[1][0].average_time_;
      const double g = std::min(g0, g1);
      if (g > best_gain) {
        best_gain = g;
        best = bench_data[0][i].config_;
      }
    }
    std::printf("  default %.0f / %.0f ns; kept variant %d launch_iters %d%s\n", bench_data[0][0].average_time_, bench_data[1][0].average_time_,
                best.variant_, best.launch_iters_, (best.variant_ || best.launch_iters_) ? "" : " (the default)");
    optimal_config.push_back(best);
    fft_length.push_back(n);
  }
  std::ofstream myfile(out_path);                              // WriteTunerResultsToFile, FileWriter.h:250-269, + three columns
  if (!myfile.is_open()) {
    std::cout << "Error! Unable to open file." << std::endl;
    return 1;
  }
  for (size_t i = 0; i < optimal_config.size(); ++i)
    myfile << fft_length[i] << " " << (fft_length[i] >= 4096 ? 4096 : 256) << " " << (fft_length[i] >= 4096 ? 16 : 1) << " 1 256 "
           << optimal_config[i].variant_ << " " << optimal_config[i].launch_iters_ << " " << batch << "\n";
  return 0;
}
