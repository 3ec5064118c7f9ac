// Host-only entry points of the C ABI under AddressSanitizer (CPU build of the library; the device code is untouched): the planner,
// the reference CreatePlan arithmetic, the distributed geometry and the variant check over their whole argument ranges, including
// too-small output buffers and nonsense arguments. Built and run by tests/test_capi_host.py; touches no device.
#include <cstdio>
#include <cstring>
#include <cstdint>
#include "tfft.h"
int main() {
  char buf[64];
  int bad = 0;
  for (int lg = 1; lg <= 30; ++lg)
    for (uint64_t inner : {1ull, 8ull, 16ull, 64ull, 4096ull})
      for (int v : {0, 32, 8388608, 33554432, 134217728, 16777216, 2097152, 268435456}) {
        char big[256];
        const int rc = tfft_plan_describe(1ull << lg, inner, v, big, sizeof(big));
        if (rc != TFFT_OK && rc != TFFT_ERR_ARG) ++bad;
        (void)tfft_plan_describe(1ull << lg, inner, v, buf, 8);      // too small: must fail cleanly, not overflow
      }
  tfft_ref_plan rp;
  for (int lg = 0; lg < 40; ++lg)
    for (int mode = -1; mode <= 2; ++mode)
      for (int w : {0, 1, 3, 8, 16, 1000}) (void)tfft_ref_create_plan(1ull << lg, mode, w, w, 256, &rp);
  (void)tfft_ref_create_plan(3000, 0, 8, 8, 256, &rp);
  tfft_dist_geometry g = TFFT_DIST_GEOMETRY_INIT;
  int ok_queries = 0;
  for (int lg = 1; lg < 40; ++lg)
    for (int world : {0, 1, 2, 3, 4, 8, 16, 64, 1024})
      for (int rank : {-1, 0, 1, 7}) {
        const int rc = tfft_dist_geometry_query(1ull << lg, world, rank, &g);
        if (rc == TFFT_OK) {
          ++ok_queries;
          if (g.struct_size != sizeof(g) || g.n != (1ull << lg) || g.n1 * g.n2 != g.n || g.world != world || g.rank != rank) ++bad;
        }
      }
  if (ok_queries == 0) ++bad;                                        // the success path must have run under the sanitizer
  // struct_size is read before anything is written: sizes that are not a layout of this library are refused ...
  for (uint32_t sz : {0u, 72u, 76u, 4104u}) {
    tfft_dist_geometry q;
    std::memset(&q, 0x5a, sizeof(q));
    q.struct_size = sz;
    if (tfft_dist_geometry_query(1ull << 26, 8, 0, &q) != TFFT_ERR_ARG) ++bad;
    if (q.n != 0x5a5a5a5a5a5a5a5aull) ++bad;                          // ... and nothing is written
  }
  // ... and a caller whose struct is LONGER than the library's gets exactly sizeof(tfft_dist_geometry) bytes, none behind them
  {
    struct { tfft_dist_geometry q; unsigned char tail[16]; } big;
    std::memset(&big, 0x5a, sizeof(big));
    big.q.struct_size = static_cast<uint32_t>(sizeof(tfft_dist_geometry) + 8);
    if (tfft_dist_geometry_query(1ull << 26, 8, 3, &big.q) != TFFT_OK) ++bad;
    if (big.q.struct_size != sizeof(tfft_dist_geometry) + 8 || big.q.rank != 3) ++bad;
    for (unsigned char c : big.tail)
      if (c != 0x5a) ++bad;
  }
  for (int lg = 1; lg < 34; ++lg) (void)tfft_plan_transposed_n2(1ull << lg);
  for (int v = -2; v < 70000; v += 97) (void)tfft_variant_check(4096, 1, v);
  std::printf("%s last error: %.60s\n", bad ? "BAD" : "ok", tfft_last_error());
  return bad;
}
