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

#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <sstream>
#include <string>
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
  while (std::getline(file, line)) {
    std::istringstream ss(line);
    double len;
    int mode_num, bw, rw, r2;
    if (!(ss >> len >> mode_num >> bw >> rw >> r2)) continue;
    if (static_cast<Integer>(len) != fft_length) continue;
    auto plan = CreatePlan(fft_length, mode_num == 256 ? Mode_256 : Mode_4096, bw, rw, r2);
    int variant = 0;
    if (plan && (ss >> variant)) {
      if (tfft_variant_check(static_cast<uint64_t>(fft_length), 1, variant) != TFFT_OK) {
        std::cout << "Error! Tuner file holds an unusable kernel variant for this fft length: " << tfft_last_error() << std::endl;
        return std::nullopt;
      }
      plan->tfft_variant_ = variant;
    }
    return plan;
  }
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

// One execution plan per (N, batch, device, variant), kept for the life of the process so that
// ComputeFFT stays a pure launch, like the reference's. The cache is shared by all host threads (a tfft_plan is
// immutable and thread-safe, tfft.h), hence the lock.
inline tfft_plan* exec_plan(uint64_t n, uint64_t batch, int variant, std::string* err) {
  static std::mutex lock;
  static std::map<std::tuple<uint64_t, uint64_t, int, int>, tfft_plan*> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    *err = "hipGetDevice failed";
    return nullptr;
  }
  const auto key = std::make_tuple(n, batch, dev, variant);
  std::lock_guard<std::mutex> guard(lock);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  tfft_plan_opts opts{};
  opts.variant = variant;
  tfft_plan* p = nullptr;
  if (tfft_plan_create(n, batch, dev, &opts, &p) != TFFT_OK) {
    *err = tfft_last_error();
    return nullptr;
  }
  cache[key] = p;
  return p;
}
}  // namespace tfft_detail

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

  std::optional<std::string> CopyDataHostToDevice(__half* data) {
    return tfft_detail::hip_status(
        hipMemcpy(dptr_input_RE_, data, 2 * fft_length_ * sizeof(__half), hipMemcpyHostToDevice));
  }

  std::optional<std::string> CopyResultsDeviceToHost(__half* data, bool results_in_results) {
    const __half* src = results_in_results ? dptr_results_RE_ : dptr_input_RE_;
    return tfft_detail::hip_status(hipMemcpy(data, src, 2 * fft_length_ * sizeof(__half), hipMemcpyDeviceToHost));
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

  std::optional<std::string> CopyDataHostToDevice(__half* data) {
    auto r = tfft_detail::hip_status(hipMemcpy(dptr_input_RE_[0], data,
                                               static_cast<size_t>(amount_of_ffts_) * 2 * fft_length_ * sizeof(__half),
                                               hipMemcpyHostToDevice));
    if (r) return r;
    (void)hipDeviceSynchronize();
    return std::nullopt;
  }

  std::optional<std::string> CopyResultsDeviceToHost(__half* data, bool results_in_results) {
    const __half* src = results_in_results ? dptr_results_RE_[0] : dptr_input_RE_[0];
    return tfft_detail::hip_status(hipMemcpy(data, src,
                                             static_cast<size_t>(amount_of_ffts_) * 2 * fft_length_ * sizeof(__half),
                                             hipMemcpyDeviceToHost));
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
  tfft_plan* p = tfft_detail::exec_plan(static_cast<uint64_t>(fft_plan.fft_length_), 1, fft_plan.tfft_variant_, &err);
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
  tfft_plan* p = tfft_detail::exec_plan(static_cast<uint64_t>(fft_plan.fft_length_),
                                        static_cast<uint64_t>(data.amount_of_ffts_), fft_plan.tfft_variant_, &err);
  if (!p) return err;
  __half* out_re = fft_plan.results_in_results_ ? data.dptr_results_RE_[0] : data.dptr_input_RE_[0];
  __half* out_im = fft_plan.results_in_results_ ? data.dptr_results_IM_[0] : data.dptr_input_IM_[0];
  if (tfft_exec(p, data.dptr_input_RE_[0], data.dptr_input_IM_[0], out_re, out_im, nullptr) != TFFT_OK)
    return std::string(tfft_last_error());
  (void)hipDeviceSynchronize();
  return tfft_detail::peek();
}
