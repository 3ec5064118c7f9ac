// staging.hpp — host <-> device copies through pinned staging buffers (include/tfft.h: tfft_copy_h2d / tfft_copy_d2h).
// Included by tfft.hip.
//
// The step either side of the path: the reference's handlers move pageable host arrays with one blocking
// cudaMemcpy (src/base/DataHandler.h:45-70,116-153). Here the transfer is cut into chunks that travel through a ring of pinned
// buffers: while the DMA engine moves chunk i over PCIe (hipMemcpyAsync on a private stream), a few host threads copy chunk
// i + 1 between the caller's pageable memory and the next pinned slot. Bounded by min(host memcpy rate, PCIe Gen5 x16 ~ 63 GB/s).
#pragma once

#include <algorithm>
#include <thread>

namespace staging {

constexpr size_t kChunk = size_t{16} << 20;     // 16 MiB per slot
constexpr int kSlots = 4;
constexpr size_t kSmall = size_t{1} << 20;      // below this a plain hipMemcpy is as good

struct Ring {
  std::mutex busy;      // one transfer at a time PER DEVICE: host threads that feed different devices do not serialise
  int device = -1;
  void* slot[kSlots] = {};
  hipEvent_t done[kSlots] = {};
  hipStream_t stream = nullptr;
  bool ok = false;
};

// One ring per device, created on first use, kept for the life of the process (64 MiB of pinned memory per device used). The
// map is guarded by a short global lock (std::map nodes do not move, so the returned pointer stays valid); a transfer then
// holds only ITS device's Ring::busy. Call with the ring's `busy` NOT held; creation happens under the global lock.
inline Ring* ring_for(int device, std::string* err) {
  static std::mutex map_mutex;
  static std::map<int, Ring> rings;
  std::lock_guard<std::mutex> lock(map_mutex);
  Ring& r = rings[device];
  if (r.ok) return &r;
  r.device = device;
  hipError_t e = hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking);
  for (int i = 0; i < kSlots && e == hipSuccess; ++i) {
    e = hipHostMalloc(&r.slot[i], kChunk, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r.done[i], hipEventDisableTiming);
  }
  if (e != hipSuccess) {
    *err = std::string("pinned staging ring: ") + hipGetErrorString(e);
    return nullptr;
  }
  r.ok = true;
  return &r;
}

// memcpy split over a few threads (one thread moves ~10-15 GB/s; PCIe Gen5 x16 wants ~60)
inline void host_copy(void* dst, const void* src, size_t bytes) {
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const unsigned nt = static_cast<unsigned>(std::min<size_t>(std::min(4u, hw), std::max<size_t>(1, bytes >> 21)));
  if (nt <= 1) {
    std::memcpy(dst, src, bytes);
    return;
  }
  std::thread th[4];
  const size_t per = ((bytes / nt) + 63) & ~size_t{63};
  for (unsigned t = 1; t < nt; ++t) {
    const size_t lo = std::min(bytes, t * per), hi = std::min(bytes, (t + 1) * per);
    th[t] = std::thread([=] { std::memcpy(static_cast<char*>(dst) + lo, static_cast<const char*>(src) + lo, hi - lo); });
  }
  std::memcpy(dst, src, std::min(bytes, per));
  for (unsigned t = 1; t < nt; ++t) th[t].join();
}

}  // namespace staging

extern "C" {

// Blocking (like the reference's cudaMemcpy, src/base/DataHandler.h:45-53): returns when the bytes are on the device, and, like
// a blocking hipMemcpy on the null stream, it first waits for the work already queued on the device: a kernel that still reads
// or writes dst (ComputeFFT is asynchronous; a following CopyDataHostToDevice must not overwrite its input under it).
int tfft_copy_h2d(void* dst_device, const void* src_host, size_t bytes) {
  g_err.clear();
  if (!dst_device || !src_host) return fail(TFFT_ERR_ARG, "null pointer");
  if (bytes == 0) return TFFT_OK;
  if (bytes < staging::kSmall) {
    TFFT_HIP(hipMemcpy(dst_device, src_host, bytes, hipMemcpyHostToDevice));
    return TFFT_OK;
  }
  int dev = 0;
  TFFT_HIP(hipGetDevice(&dev));
  TFFT_HIP(hipDeviceSynchronize());      // order against everything queued on this device (see above)
  std::string err;
  staging::Ring* r = staging::ring_for(dev, &err);
  if (!r) return fail(TFFT_ERR_HIP, err);
  std::lock_guard<std::mutex> lock(r->busy);
  size_t off = 0;
  for (int i = 0; off < bytes; ++i) {
    const int s = i % staging::kSlots;
    const size_t len = std::min(staging::kChunk, bytes - off);
    if (i >= staging::kSlots) TFFT_HIP(hipEventSynchronize(r->done[s]));      // the slot's previous DMA has drained
    staging::host_copy(r->slot[s], static_cast<const char*>(src_host) + off, len);
    TFFT_HIP(hipMemcpyAsync(static_cast<char*>(dst_device) + off, r->slot[s], len, hipMemcpyHostToDevice, r->stream));
    TFFT_HIP(hipEventRecord(r->done[s], r->stream));
    off += len;
  }
  TFFT_HIP(hipStreamSynchronize(r->stream));
  return TFFT_OK;
}

int tfft_copy_d2h(void* dst_host, const void* src_device, size_t bytes) {
  g_err.clear();
  if (!dst_host || !src_device) return fail(TFFT_ERR_ARG, "null pointer");
  if (bytes == 0) return TFFT_OK;
  if (bytes < staging::kSmall) {
    TFFT_HIP(hipMemcpy(dst_host, src_device, bytes, hipMemcpyDeviceToHost));
    return TFFT_OK;
  }
  int dev = 0;
  TFFT_HIP(hipGetDevice(&dev));
  TFFT_HIP(hipDeviceSynchronize());      // (as a blocking hipMemcpy would: everything that produces src has finished)
  std::string err;
  staging::Ring* r = staging::ring_for(dev, &err);
  if (!r) return fail(TFFT_ERR_HIP, err);
  std::lock_guard<std::mutex> lock(r->busy);
  const size_t chunks = (bytes + staging::kChunk - 1) / staging::kChunk;
  auto issue = [&](size_t c) -> hipError_t {
    const int s = static_cast<int>(c % staging::kSlots);
    const size_t off = c * staging::kChunk, len = std::min(staging::kChunk, bytes - off);
    hipError_t e = hipMemcpyAsync(r->slot[s], static_cast<const char*>(src_device) + off, len, hipMemcpyDeviceToHost, r->stream);
    if (e == hipSuccess) e = hipEventRecord(r->done[s], r->stream);
    return e;
  };
  for (size_t c = 0; c < std::min<size_t>(chunks, staging::kSlots - 1); ++c) TFFT_HIP(issue(c));      // DMA runs ahead
  for (size_t c = 0; c < chunks; ++c) {
    const int s = static_cast<int>(c % staging::kSlots);
    const size_t off = c * staging::kChunk, len = std::min(staging::kChunk, bytes - off);
    if (c + staging::kSlots - 1 < chunks) TFFT_HIP(issue(c + staging::kSlots - 1));   // (its slot was emptied one step ago)
    TFFT_HIP(hipEventSynchronize(r->done[s]));
    staging::host_copy(static_cast<char*>(dst_host) + off, r->slot[s], len);
  }
  return TFFT_OK;
}

}  // extern "C"
