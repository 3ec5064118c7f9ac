// tensor_fft.hpp — header-only C++ face of libtfft.so with the reference's names.
//
// A translation unit written against CPestka/Tensor-FFT's src/base headers
// (Plan.h, DataHandler.h, ComputeFFT.h) compiles against this file instead, with
// hipcc, and runs on MI355X: same free functions, same class members, same
// std::optional error convention. Everything here is glue over the C ABI in
// tfft.h; no kernel is launched from this header.
//
//   reference                                   here
//   ------------------------------------------  ---------------------------------
//   enum BaseFFTMode, struct Plan<Integer>      same            (Plan.h:14-39)
//   CreatePlan(N, mode, wpb, wpb, r2bs)         tfft_ref_create_plan   (Plan.h:77-194)
//   CreatePlan(N, tuner_file)                   parsed here     (Plan.h:197-255)
//   PlanWorksOnDevice / GetMaxNoOptInSharedMem  tfft_device_check / tfft_max_no_optin_shared_mem
//   DataHandler / DataBatchHandler              hipMalloc'ed blocks, same layout (DataHandler.h:22-166)
//   ComputeFFT (2 overloads)                    tfft_exec       (ComputeFFT.h:54-151,162-293)
#pragma once

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <sstream>
#include <string>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

#include "tfft.h"

enum BaseFFTMode { Mode_256 = TFFT_MODE_256, Mode_4096 = TFFT_MODE_4096 };

template <typename Integer>
struct Plan {
  Integer fft_length_;
  int amount_of_r16_steps_;
  int amount_of_r2_steps_;
  BaseFFTMode base_fft_mode_;
  bool results_in_results_;   // true: spectrum in the results half, false: in the input half
  int base_fft_warps_per_block_;
  int base_fft_blocksize_;
  int base_fft_gridsize_;
  int base_fft_shared_mem_in_bytes_;
  int r16_warps_per_block_;
  int r16_blocksize_;
  int r16_gridsize_;
  int r16_shared_mem_in_bytes_;
  int r2_blocksize_;
  // MI355X extension (not in the reference's struct): tuned kernel variant = sixth column of a tuner file written by
  // tools/tuner.py (tfft_plan_opts.variant); 0 = library default. CreatePlan(N, mode, ...) leaves it 0.
  int tfft_variant_ = 0;
  // One entry per tuner-file line of this length (columns 6 - 8: variant, launch_iters, the batch the line was tuned at;
  // batch 0 = no batch column). ComputeFFT picks the entry whose batch is nearest on a log scale to the batch it runs.
  struct Tuned { long long batch; int variant; int launch_iters; };
  std::vector<Tuned> tfft_tuned_;
};

template <typename Integer>
bool IsPowerOf2(const Integer x) {
  return x > 0 && (x & (x - 1)) == 0;
}

template <typename Integer>
int ExactLog2(const Integer x) {
  int l = 0;
  for (Integer t = x; t > 1; t /= 2) ++l;
  return l;
}

template <typename Integer>
Integer ExactPowerOf2(const int exponent) {
  if (exponent < 0) std::cout << "Error! Negative exponent not allowed." << std::endl;
  Integer r = 1;
  for (int i = 0; i < exponent; ++i) r *= 2;
  return r;
}

template <typename Integer>
std::optional<Plan<Integer>> CreatePlan(const Integer fft_length, const BaseFFTMode mode = Mode_256,
                                        const int base_fft_warps_per_block = 8,
                                        const int r16_warps_per_block = 8, const int r2_blocksize = 256) {
  tfft_ref_plan c;
  const int rc = tfft_ref_create_plan(static_cast<uint64_t>(fft_length), static_cast<int>(mode),
                                      base_fft_warps_per_block, r16_warps_per_block, r2_blocksize, &c);
  const std::string msg = tfft_last_error();
  if (!msg.empty()) std::cout << msg << std::endl;
  if (rc != TFFT_OK) return std::nullopt;
  Plan<Integer> p;
  p.fft_length_ = static_cast<Integer>(c.fft_length);
  p.amount_of_r16_steps_ = c.amount_of_r16_steps;
  p.amount_of_r2_steps_ = c.amount_of_r2_steps;
  p.base_fft_mode_ = static_cast<BaseFFTMode>(c.base_fft_mode);
  p.results_in_results_ = c.results_in_results != 0;
  p.base_fft_warps_per_block_ = c.base_fft_warps_per_block;
  p.base_fft_blocksize_ = c.base_fft_blocksize;
  p.base_fft_gridsize_ = c.base_fft_gridsize;
  p.base_fft_shared_mem_in_bytes_ = c.base_fft_shared_mem_in_bytes;
  p.r16_warps_per_block_ = c.r16_warps_per_block;
  p.r16_blocksize_ = c.r16_blocksize;
  p.r16_gridsize_ = c.r16_gridsize;
  p.r16_shared_mem_in_bytes_ = c.r16_shared_mem_in_bytes;
  p.r2_blocksize_ = c.r2_blocksize;
  return p;
}

// Tuner-file overload: first line whose leading number equals fft_length, fields
// `N mode base_wpb r16_wpb r2_blocksize` with mode written as 256 or 4096, and optionally the sixth column
// tools/tuner.py appends (the tuned kernel variant; checked with tfft_variant_check, a line with an unusable
// value is refused like a missing line).
template <typename Integer>
std::optional<Plan<Integer>> CreatePlan(const Integer fft_length, const std::string tuner_results_file) {
  std::ifstream file(tuner_results_file);
  if (!file.is_open()) {
    std::cout << "Error! Failed to open tuner file." << std::endl;
    return std::nullopt;
  }
  std::string line;
  std::optional<Plan<Integer>> plan;
  while (std::getline(file, line)) {
    std::istringstream ss(line);
    double len;
    int mode_num, bw, rw, r2;
    if (!(ss >> len >> mode_num >> bw >> rw >> r2)) continue;
    if (static_cast<Integer>(len) != fft_length) continue;
    if (!plan) {
      plan = CreatePlan(fft_length, mode_num == 256 ? Mode_256 : Mode_4096, bw, rw, r2);
      if (!plan) return std::nullopt;
    }
    int variant = 0;
    if (!(ss >> variant)) break;          // a plain reference line: nothing more to read for this length
    long long iters = 0, batch = 0;
    if (ss >> iters) ss >> batch;
    if (tfft_variant_check(static_cast<uint64_t>(fft_length), 1, variant) != TFFT_OK || iters < 0 || iters > 65535 || batch < 0) {
      std::cout << "Error! Tuner file holds an unusable kernel variant for this fft length: " << tfft_last_error() << std::endl;
      return std::nullopt;
    }
    if (plan->tfft_tuned_.empty()) plan->tfft_variant_ = variant;
    plan->tfft_tuned_.push_back({batch, variant, static_cast<int>(iters)});
  }
  if (plan) return plan;
  std::cout << "Error! Tuner file didnt contain requested fft length." << std::endl;
  return std::nullopt;
}

template <typename Integer>
bool PlanWorksOnDevice(const Plan<Integer>, const int device_id) {
  if (tfft_device_check(device_id) == TFFT_OK) return true;
  std::cout << tfft_last_error() << std::endl;
  return false;
}

inline int GetMaxNoOptInSharedMem(const int device_id) { return tfft_max_no_optin_shared_mem(device_id); }

namespace tfft_detail {
inline std::optional<std::string> hip_status(hipError_t e) {
  if (e == hipSuccess) return std::nullopt;
  return std::string(hipGetErrorString(e));
}
inline std::optional<std::string> peek() { return hip_status(hipPeekAtLastError()); }
inline std::optional<std::string> staged(int rc) {
  if (rc == TFFT_OK) return std::nullopt;
  return std::string(tfft_last_error());
}
// Host <-> device copies of the handlers. Default: ONE blocking hipMemcpy, exactly the reference's cudaMemcpy
// (src/base/DataHandler.h:45-70,116-153): ROCm 7.2's pageable path already pipelines through pinned buffers and measured
// 55.6 / 55.8 GB/s on MI355X against 53.7 / 53.3 for the library's own pinned ring (tfft_copy_h2d / _d2h; 1.07 GB,
// profiles/r3_bench_batched_cxx.txt), both ~0.85 of PCIe Gen5 x16. SetStagedCopies(true) selects the ring (per-device
// locks: host threads that feed different devices overlap).
inline bool& staged_copies_flag() {
  static bool on = false;
  return on;
}
inline std::optional<std::string> copy_h2d(void* dst, const void* src, size_t bytes) {
  if (staged_copies_flag()) return staged(tfft_copy_h2d(dst, src, bytes));
  return hip_status(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
}
inline std::optional<std::string> copy_d2h(void* dst, const void* src, size_t bytes) {
  if (staged_copies_flag()) return staged(tfft_copy_d2h(dst, src, bytes));
  return hip_status(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
}
// fn(i) for every device slot i at once, one host thread per device: the transfers of all devices are in flight together
// (each device has its own PCIe link; one thread issuing blocking copies device after device would use one link at a time)
template <typename Fn>
inline std::optional<std::string> for_each_device_parallel(size_t n, Fn fn) {
  std::vector<std::optional<std::string>> err(n);
  std::vector<std::thread> th;
  for (size_t i = 1; i < n; ++i) th.emplace_back([&, i] { err[i] = fn(i); });
  if (n) err[0] = fn(0);
  for (auto& t : th) t.join();
  for (auto& e : err)
    if (e) return e;
  return std::nullopt;
}

// One execution plan per (N, batch, device, variant), kept for the life of the process so that
// ComputeFFT stays a pure launch, like the reference's. The cache is shared by all host threads (a tfft_plan is
// immutable and thread-safe, tfft.h), hence the lock.
struct PlanCache {
  std::mutex lock;
  std::map<std::tuple<uint64_t, uint64_t, int, int, int>, tfft_plan*> plans;
};
inline PlanCache& plan_cache() {
  static PlanCache c;
  return c;
}
inline tfft_plan* exec_plan(uint64_t n, uint64_t batch, int variant, std::string* err, int launch_iters = 0) {
  if (tfft_abi_version() != TFFT_ABI_VERSION) {      // the struct layouts this translation unit was compiled with (tfft.h)
    *err = "libtfft.so speaks ABI " + std::to_string(tfft_abi_version()) + ", this program was built against ABI " + std::to_string(TFFT_ABI_VERSION);
    return nullptr;
  }
  std::mutex& lock = plan_cache().lock;
  auto& cache = plan_cache().plans;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    *err = "hipGetDevice failed";
    return nullptr;
  }
  const auto key = std::make_tuple(n, batch, dev, variant, launch_iters);
  std::lock_guard<std::mutex> guard(lock);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  tfft_plan_opts opts = TFFT_PLAN_OPTS_INIT;
  opts.variant = variant;
  opts.launch_iters = static_cast<uint32_t>(launch_iters);
  tfft_plan* p = nullptr;
  if (tfft_plan_create(n, batch, dev, &opts, &p) != TFFT_OK) {
    *err = tfft_last_error();
    return nullptr;
  }
  // a multi-pass plan gets its workspace here, at creation: ComputeFFT itself never allocates
  if (tfft_plan_prepare(p) != TFFT_OK) {
    *err = tfft_last_error();
    tfft_plan_destroy(p);
    return nullptr;
  }
  cache[key] = p;
  return p;
}
// (variant, launch_iters) of the tuner-file entry nearest to `batch` on a log scale; the plan's default without entries
template <typename PlanT>
inline std::pair<int, int> tuned_for_batch(const PlanT& plan, uint64_t batch) {
  std::pair<int, int> best{plan.tfft_variant_, 0};
  double best_d = -1;
  for (const auto& t : plan.tfft_tuned_) {
    const double d = t.batch > 0 ? std::fabs(std::log2(static_cast<double>(batch ? batch : 1)) - std::log2(static_cast<double>(t.batch))) : 1e9;
    if (best_d < 0 || d < best_d) {
      best = {t.variant, t.launch_iters};
      best_d = d;
    }
  }
  return best;
}
}  // namespace tfft_detail

// Not in the reference: selects the library's pinned staging ring (tfft_copy_h2d / _d2h) for the handlers' host copies
// instead of one blocking hipMemcpy (see tfft_detail::copy_h2d for the measurement behind the default).
inline void SetStagedCopies(bool on) { tfft_detail::staged_copies_flag() = on; }

// Not in the reference (its plans own no device memory): destroys the execution plans ComputeFFT has cached, with their
// constant tables and workspaces (a 2^26 plan holds 512 MiB). Call with no ComputeFFT in flight; later calls re-create.
inline void ReleaseComputeFFTPlans() {
  auto& c = tfft_detail::plan_cache();
  std::lock_guard<std::mutex> guard(c.lock);
  for (auto& kv : c.plans) {
    (void)hipSetDevice(std::get<2>(kv.first));
    (void)hipDeviceSynchronize();
    tfft_plan_destroy(kv.second);
  }
  c.plans.clear();
}

// 4*N halves on the device: in_RE | in_IM | out_RE | out_IM.
template <typename Integer>
class DataHandler {
 public:
  explicit DataHandler(const Integer fft_length) : fft_length_(fft_length), dptr_data_(nullptr) {
    if (hipMalloc(reinterpret_cast<void**>(&dptr_data_), 4 * sizeof(__half) * fft_length_) != hipSuccess)
      std::cout << hipGetErrorString(hipPeekAtLastError()) << std::endl;
    dptr_input_RE_ = dptr_data_;
    dptr_input_IM_ = dptr_input_RE_ + fft_length_;
    dptr_results_RE_ = dptr_input_IM_ + fft_length_;
    dptr_results_IM_ = dptr_results_RE_ + fft_length_;
  }
  DataHandler(const DataHandler&) = delete;
  DataHandler& operator=(const DataHandler&) = delete;
  ~DataHandler() { (void)hipFree(dptr_data_); }

  std::optional<std::string> PeakAtLastError() { return tfft_detail::peek(); }

  // one blocking copy like the reference's (DataHandler.h:45-70): ordered behind whatever ComputeFFT has queued on the device
  std::optional<std::string> CopyDataHostToDevice(__half* data) {
    return tfft_detail::copy_h2d(dptr_input_RE_, data, 2 * static_cast<size_t>(fft_length_) * sizeof(__half));
  }

  std::optional<std::string> CopyResultsDeviceToHost(__half* data, bool results_in_results) {
    const __half* src = results_in_results ? dptr_results_RE_ : dptr_input_RE_;
    return tfft_detail::copy_d2h(data, src, 2 * static_cast<size_t>(fft_length_) * sizeof(__half));
  }

  Integer fft_length_;
  __half* dptr_data_;
  __half* dptr_input_RE_;
  __half* dptr_input_IM_;
  __half* dptr_results_RE_;
  __half* dptr_results_IM_;
};

// amount_of_ffts * 4 * N halves: all inputs [fft_i RE | fft_i IM]..., then all results likewise.
template <typename Integer>
class DataBatchHandler {
 public:
  DataBatchHandler(const Integer fft_length, const int amount_of_ffts)
      : fft_length_(fft_length), amount_of_ffts_(amount_of_ffts), dptr_data_(nullptr) {
    if (hipMalloc(reinterpret_cast<void**>(&dptr_data_),
                  static_cast<size_t>(amount_of_ffts_) * 4 * sizeof(__half) * fft_length_) != hipSuccess)
      std::cout << hipGetErrorString(hipPeekAtLastError()) << std::endl;
    __half* results = dptr_data_ + static_cast<size_t>(amount_of_ffts_) * 2 * fft_length_;
    for (int i = 0; i < amount_of_ffts_; ++i) {
      const size_t off = static_cast<size_t>(i) * 2 * fft_length_;
      dptr_input_RE_.push_back(dptr_data_ + off);
      dptr_input_IM_.push_back(dptr_data_ + off + fft_length_);
      dptr_results_RE_.push_back(results + off);
      dptr_results_IM_.push_back(results + off + fft_length_);
    }
  }
  DataBatchHandler(const DataBatchHandler&) = delete;
  DataBatchHandler& operator=(const DataBatchHandler&) = delete;
  ~DataBatchHandler() { (void)hipFree(dptr_data_); }

  std::optional<std::string> PeakAtLastError() { return tfft_detail::peek(); }

  // one blocking copy, the reference's: DataHandler.h:116-153 (tfft_detail::copy_h2d)
  std::optional<std::string> CopyDataHostToDevice(__half* data) {
    return tfft_detail::copy_h2d(dptr_input_RE_[0], data, static_cast<size_t>(amount_of_ffts_) * 2 * fft_length_ * sizeof(__half));
  }

  std::optional<std::string> CopyResultsDeviceToHost(__half* data, bool results_in_results) {
    const __half* src = results_in_results ? dptr_results_RE_[0] : dptr_input_RE_[0];
    return tfft_detail::copy_d2h(data, src, static_cast<size_t>(amount_of_ffts_) * 2 * fft_length_ * sizeof(__half));
  }

  Integer fft_length_;
  int amount_of_ffts_;
  __half* dptr_data_;
  std::vector<__half*> dptr_input_RE_;
  std::vector<__half*> dptr_input_IM_;
  std::vector<__half*> dptr_results_RE_;
  std::vector<__half*> dptr_results_IM_;
};

// Single transform on the default stream, asynchronous. The spectrum is left in the
// half of the handler that fft_plan.results_in_results_ names.
template <typename Integer>
std::optional<std::string> ComputeFFT(Plan<Integer>& fft_plan, const DataHandler<Integer>& data,
                                      const int /*max_no_optin_shared_mem*/ = 32768) {
  std::string err;
  const auto tuned = tfft_detail::tuned_for_batch(fft_plan, 1);
  tfft_plan* p = tfft_detail::exec_plan(static_cast<uint64_t>(fft_plan.fft_length_), 1, tuned.first, &err, tuned.second);
  if (!p) return err;
  __half* out_re = fft_plan.results_in_results_ ? data.dptr_results_RE_ : data.dptr_input_RE_;
  __half* out_im = fft_plan.results_in_results_ ? data.dptr_results_IM_ : data.dptr_input_IM_;
  if (tfft_exec(p, data.dptr_input_RE_, data.dptr_input_IM_, out_re, out_im, nullptr) != TFFT_OK)
    return std::string(tfft_last_error());
  return tfft_detail::peek();
}

// Whole batch in one launch sequence on the default stream, then a device synchronise
// (the reference's batch overload ends with one too).
template <typename Integer>
std::optional<std::string> ComputeFFT(const Plan<Integer>& fft_plan, const DataBatchHandler<Integer>& data,
                                      const int /*max_no_optin_shared_mem*/ = 32768) {
  std::string err;
  const auto tuned = tfft_detail::tuned_for_batch(fft_plan, static_cast<uint64_t>(data.amount_of_ffts_));
  tfft_plan* p = tfft_detail::exec_plan(static_cast<uint64_t>(fft_plan.fft_length_),
                                        static_cast<uint64_t>(data.amount_of_ffts_), tuned.first, &err, tuned.second);
  if (!p) return err;
  __half* out_re = fft_plan.results_in_results_ ? data.dptr_results_RE_[0] : data.dptr_input_RE_[0];
  __half* out_im = fft_plan.results_in_results_ ? data.dptr_results_IM_[0] : data.dptr_input_IM_[0];
  if (tfft_exec(p, data.dptr_input_RE_[0], data.dptr_input_IM_[0], out_re, out_im, nullptr) != TFFT_OK)
    return std::string(tfft_last_error());
  (void)hipDeviceSynchronize();
  return tfft_detail::peek();
}

// ---------------------------------------------------------------------------------------------------------------------
// Multi-GPU. The reference's counterpart is dead code (ComputeFFTMultiGPU / ComputeFFTsMultiGPU,
// src/base/ComputeFFT.h:295-557; DataHandlerMultiGPU / DataBatchHandlerMultiGPU, src/base/DataHandler.h:168-403) that ran
// one independent transform (or batch) per device. Same names and call sequence here, MI355X meaning:
//
//   DataBatchHandlerMultiGPU + ComputeFFTsMultiGPU   ONE batch sharded contiguously over the devices (BASELINE configs[4a]):
//                                                    independent transforms, no collective, one enqueue per device
//   DataHandlerMultiGPU + ComputeFFTMultiGPU         ONE transform of fft_length spread over the devices (configs[4b]):
//                                                    four-step FFT with a single RCCL exchange over xGMI (tfft_dist_*)
//
// Both drive all devices from the calling thread (the shape of the reference's loops over cudaSetDevice); a program with
// one process per GPU uses tfft_dist_* / tfft_exec directly (tensor-fft_amd/distributed.py, bench.py).
// ---------------------------------------------------------------------------------------------------------------------
template <typename Integer>
class DataBatchHandlerMultiGPU {
 public:
  // amount_of_ffts = the WHOLE batch; device i takes transforms [first_[i], first_[i] + count_[i])
  DataBatchHandlerMultiGPU(const Integer fft_length, const int amount_of_ffts, std::vector<int> device_ids)
      : fft_length_(fft_length), amount_of_ffts_(amount_of_ffts), device_ids_(std::move(device_ids)) {
    const int nd = static_cast<int>(device_ids_.size());
    dptr_data_.assign(nd, nullptr);
    for (int i = 0; i < nd; ++i) {
      const int per = amount_of_ffts_ / nd, extra = amount_of_ffts_ % nd;
      first_.push_back(i * per + std::min(i, extra));
      count_.push_back(per + (i < extra ? 1 : 0));
      (void)hipSetDevice(device_ids_[i]);
      if (count_[i] && hipMalloc(reinterpret_cast<void**>(&dptr_data_[i]),
                                 static_cast<size_t>(count_[i]) * 4 * sizeof(__half) * fft_length_) != hipSuccess)
        std::cout << hipGetErrorString(hipPeekAtLastError()) << std::endl;
    }
  }
  DataBatchHandlerMultiGPU(const DataBatchHandlerMultiGPU&) = delete;
  DataBatchHandlerMultiGPU& operator=(const DataBatchHandlerMultiGPU&) = delete;
  ~DataBatchHandlerMultiGPU() {
    for (size_t i = 0; i < device_ids_.size(); ++i) {
      (void)hipSetDevice(device_ids_[i]);
      (void)hipFree(dptr_data_[i]);
    }
  }

  __half* input(int i) const { return dptr_data_[i]; }
  __half* results(int i) const { return dptr_data_[i] + static_cast<size_t>(count_[i]) * 2 * fft_length_; }

  std::optional<std::string> PeakAtLastError() {
    for (int d : device_ids_) {
      (void)hipSetDevice(d);
      if (auto e = tfft_detail::peek()) return e;
    }
    return std::nullopt;
  }

  // data: the whole batch, [fft_i RE | fft_i IM] blocks in order (DataBatchHandler's layout)
  // (all devices' slices in flight at once, one host thread per device: every GPU has its own PCIe link)
  std::optional<std::string> CopyDataHostToDevice(__half* data) {
    return tfft_detail::for_each_device_parallel(device_ids_.size(), [&](size_t i) -> std::optional<std::string> {
      if (!count_[i]) return std::nullopt;
      if (hipSetDevice(device_ids_[i]) != hipSuccess) return std::string("hipSetDevice failed");
      return tfft_detail::copy_h2d(input(static_cast<int>(i)), data + static_cast<size_t>(first_[i]) * 2 * fft_length_,
                                   static_cast<size_t>(count_[i]) * 2 * fft_length_ * sizeof(__half));
    });
  }

  std::optional<std::string> CopyResultsDeviceToHost(__half* data, bool results_in_results) {
    return tfft_detail::for_each_device_parallel(device_ids_.size(), [&](size_t i) -> std::optional<std::string> {
      if (!count_[i]) return std::nullopt;
      if (hipSetDevice(device_ids_[i]) != hipSuccess) return std::string("hipSetDevice failed");
      const __half* src = results_in_results ? results(static_cast<int>(i)) : input(static_cast<int>(i));
      return tfft_detail::copy_d2h(data + static_cast<size_t>(first_[i]) * 2 * fft_length_, src,
                                   static_cast<size_t>(count_[i]) * 2 * fft_length_ * sizeof(__half));
    });
  }

  Integer fft_length_;
  int amount_of_ffts_;
  std::vector<int> device_ids_;
  std::vector<int> first_, count_;
  std::vector<__half*> dptr_data_;
};

// The whole batch: every device gets one enqueue for its slice, then all devices are synchronised (the reference's batch
// overload ends with a device synchronise too).
template <typename Integer>
std::optional<std::string> ComputeFFTsMultiGPU(const Plan<Integer>& fft_plan, const DataBatchHandlerMultiGPU<Integer>& data) {
  for (size_t i = 0; i < data.device_ids_.size(); ++i) {
    if (!data.count_[i]) continue;
    if (hipSetDevice(data.device_ids_[i]) != hipSuccess) return std::string("hipSetDevice failed");
    std::string err;
    const auto tuned = tfft_detail::tuned_for_batch(fft_plan, static_cast<uint64_t>(data.count_[i]));
    tfft_plan* p = tfft_detail::exec_plan(static_cast<uint64_t>(fft_plan.fft_length_), static_cast<uint64_t>(data.count_[i]),
                                          tuned.first, &err, tuned.second);
    if (!p) return err;
    const int ii = static_cast<int>(i);
    __half* out = fft_plan.results_in_results_ ? data.results(ii) : data.input(ii);
    if (tfft_exec(p, data.input(ii), data.input(ii) + fft_plan.fft_length_, out, out + fft_plan.fft_length_, nullptr) != TFFT_OK)
      return std::string(tfft_last_error());
  }
  for (int d : data.device_ids_) {
    (void)hipSetDevice(d);
    (void)hipDeviceSynchronize();
    if (auto e = tfft_detail::peek()) return e;
  }
  return std::nullopt;
}

// ONE transform of fft_length over the devices. Device p holds the "columns" slice x[n1 N2 + p C + c] as an [N1][C] matrix per
// plane and ends up with X[k1 + N1 k2], k1 in its block of K rows, as a [K][N2] matrix per plane (tfft.h, tfft_dist_*).
// The host copies below scatter a natural-order signal / gather a natural-order spectrum.
template <typename Integer>
class DataHandlerMultiGPU {
 public:
  // self_via_comm: route the own chunk through RCCL too (lets a box with one GPU run the collective path; tests)
  DataHandlerMultiGPU(const Integer fft_length, std::vector<int> device_ids, bool self_via_comm = false)
      : fft_length_(fft_length), device_ids_(std::move(device_ids)) {
    const int nd = static_cast<int>(device_ids_.size());
    plans_.assign(nd, nullptr);
    comms_.assign(nd, nullptr);
    dptr_data_.assign(nd, nullptr);
    if (nd > 1 || self_via_comm) {
      if (tfft_dist_comm_create_all(nd, device_ids_.data(), comms_.data()) != TFFT_OK) {
        error_ = tfft_last_error();
        std::cout << error_ << std::endl;
        return;
      }
    }
    for (int i = 0; i < nd; ++i) {
      if (tfft_dist_plan_create(static_cast<uint64_t>(fft_length_), nd, i, device_ids_[i], comms_[i],
                                self_via_comm ? TFFT_DIST_SELF_VIA_COMM : 0, &plans_[i]) != TFFT_OK) {
        error_ = tfft_last_error();
        std::cout << error_ << std::endl;
        return;
      }
      (void)hipSetDevice(device_ids_[i]);
      if (hipMalloc(reinterpret_cast<void**>(&dptr_data_[i]), 4 * sizeof(__half) * local()) != hipSuccess)
        std::cout << hipGetErrorString(hipPeekAtLastError()) << std::endl;
    }
    (void)tfft_dist_plan_geometry(plans_[0], &geometry_);
  }
  DataHandlerMultiGPU(const DataHandlerMultiGPU&) = delete;
  DataHandlerMultiGPU& operator=(const DataHandlerMultiGPU&) = delete;
  ~DataHandlerMultiGPU() {
    for (size_t i = 0; i < device_ids_.size(); ++i) {
      (void)hipSetDevice(device_ids_[i]);
      (void)hipDeviceSynchronize();
      tfft_dist_plan_destroy(plans_[i]);
      (void)hipFree(dptr_data_[i]);
    }
    for (void* c : comms_) (void)tfft_dist_comm_destroy(c);
  }

  size_t local() const { return static_cast<size_t>(fft_length_) / device_ids_.size(); }   // samples per device
  __half* input_RE(int i) const { return dptr_data_[i]; }
  __half* input_IM(int i) const { return dptr_data_[i] + local(); }
  __half* results_RE(int i) const { return dptr_data_[i] + 2 * local(); }
  __half* results_IM(int i) const { return dptr_data_[i] + 3 * local(); }

  std::optional<std::string> PeakAtLastError() {
    if (!error_.empty()) return error_;
    for (int d : device_ids_) {
      (void)hipSetDevice(d);
      if (auto e = tfft_detail::peek()) return e;
    }
    return std::nullopt;
  }

  // data = [RE: N halves | IM: N halves] in natural order; device p receives columns [p C, (p + 1) C) of the [N1][N2] view
  std::optional<std::string> CopyDataHostToDevice(__half* data) {
    if (!error_.empty()) return error_;
    const size_t n1 = geometry_.n1, n2 = geometry_.n2, c = geometry_.cols;
    for (size_t i = 0; i < device_ids_.size(); ++i) {
      (void)hipSetDevice(device_ids_[i]);
      for (int plane = 0; plane < 2; ++plane) {
        __half* dst = plane ? input_IM(static_cast<int>(i)) : input_RE(static_cast<int>(i));
        const __half* src = data + static_cast<size_t>(plane) * fft_length_ + i * c;
        if (auto e = tfft_detail::hip_status(hipMemcpy2D(dst, c * sizeof(__half), src, n2 * sizeof(__half), c * sizeof(__half), n1,
                                                         hipMemcpyHostToDevice)))
          return e;
      }
    }
    return std::nullopt;
  }

  // data = [RE | IM] of X in natural order (the devices' [K][N2] blocks hold X[k1 + N1 k2]; the transposition to
  // natural order is done here on the host: a convenience path, the device-resident result stays in its blocks)
  std::optional<std::string> CopyResultsDeviceToHost(__half* data) {
    if (!error_.empty()) return error_;
    const size_t n1 = geometry_.n1, n2 = geometry_.n2, k = geometry_.rows;
    std::vector<__half> stage(local());
    for (size_t i = 0; i < device_ids_.size(); ++i) {
      (void)hipSetDevice(device_ids_[i]);
      for (int plane = 0; plane < 2; ++plane) {
        const __half* src = plane ? results_IM(static_cast<int>(i)) : results_RE(static_cast<int>(i));
        if (auto e = tfft_detail::hip_status(hipMemcpy(stage.data(), src, local() * sizeof(__half), hipMemcpyDeviceToHost))) return e;
        __half* dst = data + static_cast<size_t>(plane) * fft_length_;
        for (size_t kk = 0; kk < k; ++kk) {
          const size_t k1 = i * k + kk;
          for (size_t k2 = 0; k2 < n2; ++k2) dst[k1 + n1 * k2] = stage[kk * n2 + k2];
        }
      }
    }
    return std::nullopt;
  }

  Integer fft_length_;
  std::vector<int> device_ids_;
  std::vector<tfft_dist_plan*> plans_;
  std::vector<void*> comms_;
  std::vector<__half*> dptr_data_;
  tfft_dist_geometry geometry_ = TFFT_DIST_GEOMETRY_INIT;
  std::string error_;
};

// Column pass on every device, ONE grouped exchange (all devices' sends and receives inside one RCCL group), row transforms
// on every device; asynchronous like the single-transform ComputeFFT (default stream of each device, no synchronise).
template <typename Integer>
std::optional<std::string> ComputeFFTMultiGPU(Plan<Integer>& fft_plan, DataHandlerMultiGPU<Integer>& data) {
  if (fft_plan.fft_length_ != data.fft_length_) return std::string("Error! Plan and data handler have different fft lengths.");
  if (!data.error_.empty()) return data.error_;
  const int nd = static_cast<int>(data.device_ids_.size());
  for (int i = 0; i < nd; ++i) {
    if (hipSetDevice(data.device_ids_[i]) != hipSuccess) return std::string("hipSetDevice failed");
    if (tfft_dist_exec_pre(data.plans_[i], data.input_RE(i), data.input_IM(i), nullptr) != TFFT_OK) return std::string(tfft_last_error());
  }
  if (data.comms_[0]) {
    if (tfft_dist_group_start() != TFFT_OK) return std::string(tfft_last_error());
    for (int i = 0; i < nd; ++i) {
      (void)hipSetDevice(data.device_ids_[i]);
      if (tfft_dist_exec_exchange(data.plans_[i], nullptr) != TFFT_OK) {
        const std::string keep = tfft_last_error();
        (void)tfft_dist_group_end();
        return keep;
      }
    }
    if (tfft_dist_group_end() != TFFT_OK) return std::string(tfft_last_error());
  }
  for (int i = 0; i < nd; ++i) {
    (void)hipSetDevice(data.device_ids_[i]);
    if (tfft_dist_exec_post(data.plans_[i], data.results_RE(i), data.results_IM(i), nullptr) != TFFT_OK)
      return std::string(tfft_last_error());
  }
  return std::nullopt;
}
