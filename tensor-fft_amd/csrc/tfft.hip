// tfft.hip — C ABI (include/tfft.h) over the gfx950 kernels. Built into
// tensor-fft_amd/libtfft.so by __graft_entry__.build() (hipcc --offload-arch=gfx950).
//
// Host side of the hot path: what CreatePlan / PlanWorksOnDevice / ComputeFFT do in
// the reference (src/base/Plan.h:77-303, src/base/ComputeFFT.h:54-293), re-cut for
// one process per GPU and explicit streams.
#include "../../include/tfft.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "k4096.hpp"
#include "k256.hpp"
#include "k256r.hpp"
#include "k4096r.hpp"
#include "colfft.hpp"
#include "colfft1024.hpp"
#include "colfft512r.hpp"
#include "collat.hpp"
#include "permute.hpp"
#include "stockham.hpp"
#include "synth.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
int hip_fail(hipError_t e, const char* what) {
  return fail(TFFT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define TFFT_HIP(call)                                  \
  do {                                                  \
    hipError_t e_ = (call);                             \
    if (e_ != hipSuccess) return hip_fail(e_, #call);   \
  } while (0)

inline bool is_pow2(uint64_t x) { return x && !(x & (x - 1)); }
inline int ilog2(uint64_t x) {
  int l = 0;
  while ((x >> l) > 1) ++l;
  return l;
}

// ---- tfft_plan_opts.variant: which bits exist (include/tfft.h). The debugging aids give WRONG or partial results and
// are refused unless TFFT_DEBUG_VARIANTS=1 is set in the environment of the process that creates the plan.
constexpr int kVarK4096 = 1 | 2 | 8 | 16;
constexpr int kVarNoLat = 1073741824;      // column passes of small work by the throughput kernels, not collat.hpp
constexpr int kVarDebug = 4 | 64 | 128 | 65536 | (15 << 8);
constexpr int kVarTuner = kVarK4096 | 32 | 4096 | 8192 | 131072 | 262144 | 524288 | 1048576 | 2097152 | 4194304 |
                          8388608 | 16777216 | 33554432 | 67108864 | 134217728 | 268435456 | 536870912 | kVarNoLat;
// The shipped libtfft.so holds NO timing-only kernel, no environment knob and no measurement hook: all of that is compiled
// only with -DTFFT_DEBUG_KERNELS (tensor-fft_amd/libtfft_debug.so, built on demand for the drivers under tools/), and even
// there the debugging bits need TFFT_DEBUG_VARIANTS=1 in the environment of the process that creates the plan.
#ifdef TFFT_DEBUG_KERNELS
constexpr bool kDebugBuild = true;
inline bool debug_variants_enabled() {
  const char* e = std::getenv("TFFT_DEBUG_VARIANTS");
  return e && e[0] == '1' && e[1] == 0;
}
#else
constexpr bool kDebugBuild = false;
inline bool debug_variants_enabled() { return false; }
#endif

// Opt-in to more than 64 KiB of dynamic LDS, once per (kernel, device). The outcome is STICKY: a failure is returned on
// every later call too (a std::call_once would report it once and then launch without the attribute). Plans run this
// for every kernel they can launch at creation time (prepare mode below), so tfft_exec stays a pure launch, also under
// stream capture; the call here then only finds its map entry.
int lds_opt_in(const void* fn, int device, int bytes) {
  static std::mutex m;
  static std::map<std::pair<const void*, int>, hipError_t> done;
  hipError_t e;
  {
    std::lock_guard<std::mutex> lock(m);
    const auto key = std::make_pair(fn, device);
    auto it = done.find(key);
    if (it == done.end()) it = done.emplace(key, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)).first;
    e = it->second;
  }
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
  return TFFT_OK;
}
// prepare mode: walk the launch logic of a plan, run lds_opt_in for every kernel it selects, launch nothing
thread_local bool g_prepare = false;
// Column slab of a four-step radix-256 pass (dist.hpp: the exchange of a distributed transform overlapped slab by slab): set around
// a launch_chain call by tfft_dist_exec*, read by launch_col. Columns, not blocks: the block width is chosen at the launch site.
struct SlabCtx {
  bool on = false;
  uint64_t col_first = 0, col_count = 0;
  uint32_t out_pitch_shift = 0, out_seg_shift = 31;
  uint64_t out_seg_gap = 0, out_base = 0;
};
thread_local SlabCtx g_slab;
#define TFFT_LAUNCH(kernel, grid, block, lds, stream, ...)                                        \
  do {                                                                                            \
    const int rc_ = lds_opt_in(reinterpret_cast<const void*>(kernel), p->device, (lds));          \
    if (rc_) return rc_;                                                                          \
    if (!g_prepare) hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);             \
  } while (0)

enum class PassKind { K4096, K4096R, K256, K256R, Col256, Stockham };

struct Pass {
  PassKind kind;
  int radix;          // K4096: 4096; K256: 256; K256R: N / 256; Col256: 256, 512 or 1024; Stockham: 2 .. 64
  uint64_t ns;        // unflattened product of the radices before this pass
  bool tw_next;       // Col256: apply the next pass's input twiddles to the output
  bool skip_tw;       // Stockham: input twiddles were applied by the previous pass
  int next_radix;     // radix of the following pass (Col256 with tw_next)
  // tfft_plan_opts.scale bookkeeping (apply_scale_mode): factor on the butterfly output (Stockham), on the fp32 radix-2
  // combine (radix-512 column pass), and on the twiddles a column pass applies
  float scale = 1.0f;
  float tw_scale = 1.0f;
};

// what the public tfft_plan_opts cannot say: knobs of the plans that other plans are built from
struct InternalOpts {
  uint32_t group_shift = 0;              // grouped batch addressing of a single-kernel plan (k4096::Addr)
  uint64_t in_gstride = 0, out_gstride = 0;
  int once_log2 = -1;                    // TFFT_SCALE_ONCE: exponent of the single factor (default log2 n)
  bool rows2d = false;                   // tables for the fused 2D row pass (k4096r front end in front of the 4096 kernel)
  // segmented input (row transforms of a distributed transform, dist.hpp): a transform's n samples arrive as n / in_seg_len
  // segments of in_seg_len contiguous halves, in_seg_stride apart; in_batch_stride is then the distance between the first
  // segments of consecutive transforms (< n). Only plans whose first pass is a cooperative radix-256 / 512 column pass.
  uint64_t in_seg_len = 0, in_seg_stride = 0;
  // row pass of a transposed-INPUT plan (create_transposed_in): output q of row k1 = b & (2^group_shift - 1) times w_otw_n^(k1 q)
  uint64_t otw_n = 0;
};

}  // namespace

struct tfft_plan {
  uint64_t n = 0, batch = 0, inner = 1;
  int device = 0;
  uint64_t in_stride = 0, out_stride = 0;
  bool preserve_input = false;
  int num_cus = 0;
  int variant = 0;
  uint32_t launch_iters = 0;    // tfft_plan_opts.launch_iters
  // output re-mapping of a single radix-512 column pass (second pass of the fused 2D plan, see tfft_plan2d_create)
  uint32_t out_row_shift = 0, out_sub_shift = 0;
  uint64_t out_sub_stride = 0;
  std::vector<Pass> passes;
  int scale_mode = TFFT_SCALE_SEQUENTIAL;
  // four-step twiddle of a single column pass (tfft_plan_opts.fourstep_n): w_M^(k (col0 + c)); the w tables below are
  // then built for M instead of n
  uint64_t tw4_modulus = 0, tw4_col0 = 0;
  bool plain_acc = false;                // column passes with plain instead of non-temporal global accesses (cache_policy())
  k4096::Addr in_map{}, out_map{};       // single-kernel plans: where transform b starts (plain or grouped)
  k4096::OutTw otw{};                    // single-kernel plans: output twiddle of a transposed-input plan's row pass (n_mask = 0: off)
  // TFFT_ORDER_TRANSPOSED input: contiguous N2-point row pass (with that output twiddle) into the workspace, then ONE plain
  // radix-N1 column pass out of it; sub_row / sub_col below, run in this order
  uint32_t in_seg_shift = 31;            // segmented input rows of the first column pass (colfft::Args::in_seg_*; 31 / 0 = off)
  uint64_t in_seg_gap = 0;
  // TFFT_ORDER_TRANSPOSED: strided radix-N1 column pass (with the four-step twiddle) into the workspace, then N1 * batch
  // contiguous N2-point transforms out of it; this plan then only owns the two sub-plans and the workspace
  tfft_plan* sub_col = nullptr;
  tfft_plan* sub_row = nullptr;
  bool rows_first = false;
  // Round 4: both sub-plans are built for `chunk` transforms and the batch runs chunk by chunk, both passes of a chunk back to
  // back through ONE workspace of chunk transforms: the contiguous pass then finds (or leaves) the intermediate in the 256-MiB
  // Infinity Cache. 2^20 x 1024, transposed output: 341-348 -> 375 Gsamples/s (profiles/r4_chunked_transposed.txt). *_tail: the
  // sub-plans of the last, shorter chunk.
  uint64_t chunk = 0;
  tfft_plan* sub_col_tail = nullptr;
  tfft_plan* sub_row_tail = nullptr;
  void* d_tables = nullptr;     // k4096::build_tables blob (K4096 and Col256 passes)
  float2* d_tw_lo = nullptr;    // w_n tables (Col256 and Stockham passes)
  float2* d_tw_hi = nullptr;
  // scratch: one [batch][2 n inner] block of halves
  mutable std::mutex ws_mutex;
  mutable void* ws = nullptr;
  mutable size_t ws_bytes = 0;
  mutable bool ws_owned = false;
};

namespace {

// plans that are one LDS-resident kernel (no ping-pong chain, no workspace, no autosort twiddle tables)
inline bool single_kernel(const tfft_plan* p) {
  return p->passes.size() == 1 && (p->passes[0].kind == PassKind::K4096 || p->passes[0].kind == PassKind::K256 ||
                                   p->passes[0].kind == PassKind::K256R || p->passes[0].kind == PassKind::K4096R);
}

// Pass decomposition of a plan: pure host logic (no device needed; tfft_plan_describe exposes it to the CPU tests).
// Single LDS-resident kernels for N = 256 .. 32768 with a contiguous axis; otherwise radix-256 / radix-512 column passes
// first (they hand the next pass its twiddles), then radix-16 and one radix-2/4/8 autosort pass, a radix-16 + radix-2/4
// tail fused into radix-32/64. variant bit 32 forces the plain autosort chain.
void plan_passes(uint64_t n, uint64_t inner, int variant, std::vector<Pass>& passes) {
  passes.clear();
  const int lg = ilog2(n);
  const bool force_stockham = variant & 32;
  if (n == 4096 && inner == 1 && !force_stockham) {
    passes.push_back(Pass{PassKind::K4096, 4096, 1, false, false, 0});
  } else if (n == 256 && inner == 1 && !force_stockham) {
    passes.push_back(Pass{PassKind::K256, 256, 1, false, false, 0});
  } else if ((n == 8192 || n == 16384 || n == 32768) && inner == 1 && !force_stockham && !(variant & 16777216)) {
    // one pass: R waves share a transform, radix-R step in front of the 4096 kernel's stages (variant bit 16777216:
    // the multi-pass column plan instead)
    passes.push_back(Pass{PassKind::K4096R, static_cast<int>(n / 4096), 1, false, false, 0});
  } else if ((n == 512 || n == 1024 || n == 2048) && inner == 1 && !force_stockham) {
    passes.push_back(Pass{PassKind::K256R, static_cast<int>(n / 256), 1, false, false, 0});
  } else {
    int n256 = 0;
    // radix-256 column passes wherever the geometry allows them
    const bool col_ok = !force_stockham && (inner == 1 ? lg >= 13 : (lg >= 8 && inner >= 16));
    if (col_ok) n256 = lg / 8;
    int rem = lg - 8 * n256;
    std::vector<int> radices(n256, 256);
    // Radix-512 / radix-1024 column passes where they save a whole pass (2^15 = 512 x 64, 2^17 = 512 x 256, 2^19 = 512 x 1024,
    // 2^20 = 1024 x 1024, 2^26 = 512 x 512 x 256, 2^28 .. 2^30 in three passes). Contiguous axis, 2^16 .. 2^30, default
    // variant: the split measured fastest among all orders of all splits with the fewest passes (tools/plan_scan.py,
    // profiles/r2_plan_scan.txt, last run: with the rotated work distribution of k4096::Rotor the candidates lie within a few
    // per cent of each other).
    static const int kColSplit[15][3] = {
        {256, 256, 0},    {512, 256, 0},    {512, 512, 0},     {512, 1024, 0},     {1024, 1024, 0},      // 2^16 .. 2^20
        {512, 512, 0},    {512, 512, 0},    {512, 512, 0},     {512, 1024, 0},     {1024, 1024, 0},      // 2^21 .. 2^25
        {512, 512, 256},  {1024, 512, 256}, {256, 1024, 1024}, {1024, 512, 1024},  {1024, 1024, 1024}};  // 2^26 .. 2^30
    const bool use512 = col_ok && inner == 1 && lg >= 15 && !(variant & 8388608);
    const bool use1024 = col_ok && inner == 1 && lg >= 16 && !(variant & 33554432);
    if (use512 && use1024 && lg >= 16 && lg <= 30 && !(variant & 134217728)) {
      radices.clear();
      int bits = 0;
      for (int i = 0; i < 3 && kColSplit[lg - 16][i]; ++i) {
        radices.push_back(kColSplit[lg - 16][i]);
        bits += ilog2(static_cast<uint64_t>(kColSplit[lg - 16][i]));
      }
      n256 = static_cast<int>(radices.size());
      rem = lg - bits;
    } else if (use512 || use1024) {
      // restricted variants (no radix-512: 8388608, no radix-1024: 33554432) and 2^15: the split lg = 8 a + 9 b + 10 c + t,
      // t <= 7, with the fewest passes (a tail of t bits costs 0 / 1 / 2 passes for t = 0 / 1..6 / 7); ties go to the
      // fewest radix-1024, then the fewest radix-512 passes, or with variant bit 134217728 to the most; wide radices first.
      auto tail_cost = [](int t) { return t == 0 ? 0 : (t <= 6 ? 1 : 2); };
      int best_a = n256, best_b = 0, best_c = 0, best_cost = n256 + tail_cost(rem);
      for (int c = 0; c <= (use1024 ? 3 : 0); ++c)
        for (int b = 0; b <= (use512 ? 3 : 0); ++b)
          for (int a2 = 0; 8 * a2 + 9 * b + 10 * c <= lg; ++a2) {
            if (b + c == 0) continue;
            const int t = lg - 8 * a2 - 9 * b - 10 * c;
            if (t > 7) continue;
            const int cost = a2 + b + c + tail_cost(t);
            const bool wide = (variant & 134217728) && cost == best_cost &&
                              (c > best_c || (c == best_c && b > best_b));
            if (cost < best_cost || wide) {
              best_cost = cost;
              best_a = a2;
              best_b = b;
              best_c = c;
            }
          }
      n256 = best_a + best_b + best_c;      // column passes in total
      radices.assign(best_c, 1024);
      radices.insert(radices.end(), best_b, 512);
      radices.insert(radices.end(), best_a, 256);
      rem = lg - 8 * best_a - 9 * best_b - 10 * best_c;
    }
    // n = 512 along a strided axis as ONE radix-512 column pass (variant bit 67108864; the second pass of the fused 2D plan)
    if (n == 512 && inner >= 64 && (variant & 67108864) && !force_stockham) {
      radices.assign(1, 512);
      n256 = 1;
      rem = 0;
    }
    // experiment knob (honoured only with TFFT_DEBUG_VARIANTS=1): TFFT_PLAN_COLS="512,512,256" replaces the column passes of a
    // contiguous-axis plan by the given radices in the given order (their product must divide n; the tail follows as usual)
#ifdef TFFT_DEBUG_KERNELS
    if (col_ok && inner == 1 && lg >= 16 && debug_variants_enabled()) {
      if (const char* e = std::getenv("TFFT_PLAN_COLS")) {
        std::vector<int> cols;
        int bits = 0;
        for (const char* q = e; *q;) {
          const int r = std::atoi(q);
          if (r == 256 || r == 512 || r == 1024) {
            cols.push_back(r);
            bits += ilog2(static_cast<uint64_t>(r));
          }
          while (*q && *q != ',') ++q;
          if (*q == ',') ++q;
        }
        if (!cols.empty() && bits <= lg && (n >> bits) * 1 >= 1 && lg - bits <= 7) {
          radices = cols;
          n256 = static_cast<int>(cols.size());
          rem = lg - bits;
        }
      }
    }
#endif
    for (; rem >= 4; rem -= 4) radices.push_back(16);
    if (rem) radices.push_back(1 << rem);
    // a radix-16 pass followed by a radix-2 / radix-4 pass behind a column pass fuses into one radix-32 / radix-64
    // pass (the butterfly fits in registers; its input twiddles come from the column pass). variant bit 2097152 keeps them apart.
    const bool fuse_tail = !(variant & 2097152);
    if (fuse_tail && n256 >= 1 && radices.size() >= static_cast<size_t>(n256) + 2) {
      const size_t last = radices.size() - 1;
      // (... and 2^15 = 256 x 16 x 8 into 256 x 128 with the workgroup-cooperative radix-128 pass, stockham::tail_coop_kernel)
      const bool coop128 = lg == 15 && inner == 1 && n256 == 1 && radices[0] == 256 && radices[last] == 8;
      if (radices[last - 1] == 16 && (radices[last] == 2 || radices[last] == 4 || coop128) && last - 1 == static_cast<size_t>(n256)) {
        radices[last - 1] = 16 * radices[last];
        radices.pop_back();
      }
    }
    uint64_t ns = 1;
    for (size_t i = 0; i < radices.size(); ++i) {
      const int R = radices[i];
      const bool last = (i + 1 == radices.size());
      if (R >= 256) {
        const bool no_tw = kDebugBuild && (variant & 128);   // debugging aid: WRONG results, timing/determinism only
        passes.push_back(Pass{PassKind::Col256, R, ns, !last && !no_tw, false, last ? 0 : radices[i + 1]});
      } else {
        const bool prev_col = i > 0 && radices[i - 1] >= 256;
        passes.push_back(Pass{PassKind::Stockham, R, ns, false, prev_col, 0});
      }
      ns *= static_cast<uint64_t>(R);
    }
  }
}

// Grid of a grid-stride ("persistent") kernel whose workgroups each own `iters` work items per wave slot: at least one
// workgroup per CU's worth when there is that much work, otherwise blocks_needed / iters so that the hardware
// dispatcher hands out workgroups as CUs drain (keeps CUs out of lock-step; see launch_k4096_v).
// (launch-shape experiment knobs: environment variables in the debug build only; the shipped library takes its launch
// shapes from the plan, see tfft_plan_opts.launch_iters)
inline uint32_t env_iters(const char* name, uint32_t dflt) {
#ifdef TFFT_DEBUG_KERNELS
  const char* e = std::getenv(name);
  return e ? static_cast<uint32_t>(std::max(0, std::atoi(e))) : dflt;
#else
  (void)name;
  return dflt;
#endif
}
// tfft_plan_opts.launch_iters -> rounds per workgroup: 0 keeps the kernel's measured default, TFFT_LAUNCH_PERSISTENT one
// workgroup per CU for the whole batch
inline uint32_t plan_iters(uint32_t launch_iters, uint32_t dflt) {
  if (launch_iters == 0) return dflt;
  return launch_iters >= TFFT_LAUNCH_PERSISTENT ? 1000000u : launch_iters;
}
inline uint32_t pick_grid(uint64_t blocks_needed, int num_cus, uint32_t iters) {
  const uint64_t lo = std::min<uint64_t>(blocks_needed, static_cast<uint64_t>(num_cus));
  return static_cast<uint32_t>(std::max<uint64_t>(lo, (blocks_needed + iters - 1) / std::max(iters, 1u)));
}

// Grid of the kernels that used to run as persistent workgroups (column passes, k4096r, the fused 2D row pass). Round 4: a
// static partition (one workgroup per CU for the whole batch) ends when the SLOWEST CU ends, and on some boxes CUs differ by
// several per cent; `gens` generations of workgroups, handed out by the hardware dispatcher as CUs drain, balance
// that dynamically. What it buys depends on the box (three boxes, profiles/r4_iters_scan.txt, r4_gens_scan.txt): 2^16 x 16384
// (4-wave radix-256 workgroups, 8 generations) +2.6 %, +2.6 %, +6.5 %; the 8-wave radix-256 / 512 / 1024 kernels with 2 generations
// 0 ... +0.5 % on two boxes and +3.6 ... +5 % on the third (2^20 x 1024: 329 -> 341 Gsamples/s); more generations lose again (tables
// and pipeline fill are paid per workgroup: 2^20 at 8 generations -3 %). The fused 2D row pass and the 8192 ... 32768 kernels
// (one iteration = 13 us) keep the static partition (-1 ... -3 % with two generations). Whole multiples of the resident capacity
// only (a grid of 2.7 capacities leaves a third of the chip idle in its last round: 301 Gsamples/s), and only while every
// workgroup still gets kMinRounds rounds. tfft_plan_opts.launch_iters overrides (the tuner's knob; TFFT_LAUNCH_PERSISTENT = the
// static partition).
inline uint32_t gens_grid(uint64_t blocks, uint32_t capacity, uint32_t launch_iters, uint32_t gens_dflt) {
  if (launch_iters) return pick_grid(blocks, static_cast<int>(capacity), plan_iters(launch_iters, 1000000u));
  static const uint32_t gens_env = env_iters("TFFT_GENS", 0);            // experiment knob (debug build only)
  const uint32_t gens = gens_env ? gens_env : gens_dflt;
  static const uint64_t kMinRounds = env_iters("TFFT_GENS_MIN_ROUNDS", 8);     // (experiment knob in the debug build; 8 otherwise)
  if (gens > 1 && blocks >= static_cast<uint64_t>(capacity) * gens * kMinRounds) return capacity * gens;
  return static_cast<uint32_t>(std::min<uint64_t>(blocks, capacity));
}
constexpr uint32_t kGensStatic = 1, kGensCol8 = 2;    // per kernel family, see above (the radix-256 workgroup kernel: rounds_grid below)

// The radix-256 workgroup kernel since its tables are fetched behind its first block's copy-in (colfft.hpp, end of round 4): a
// workgroup's start-up is cheap enough for about FOUR rounds per workgroup to be the best shape, however many generations that
// makes (profiles/r4_gens_after_prologue.txt, one process: 2^16 x 4096 345 -> 369 Gsamples/s, 2^16 x 16384 357 -> 365, 256-point
// transforms along a strided axis +1 ... +1.5 %; one or two rounds per workgroup lose 1 ... 9 %). Whole multiples of the resident
// capacity, as above.
inline uint32_t rounds_grid(uint64_t blocks, uint32_t capacity, uint32_t launch_iters) {
  if (launch_iters) return pick_grid(blocks, static_cast<int>(capacity), plan_iters(launch_iters, 1000000u));
  constexpr uint64_t kRounds = 4, kMaxGens = 64;
  const uint64_t gens = std::min<uint64_t>(kMaxGens, blocks / (static_cast<uint64_t>(capacity) * kRounds));
  if (gens >= 2) return static_cast<uint32_t>(capacity * gens);
  return static_cast<uint32_t>(std::min<uint64_t>(blocks, capacity));
}

// Waves per workgroup that take work in the single-pass kernels (one transform, or one group of transforms, per wave): 8 when the
// batch fills the chip; for `units` wave-tasks that do not, the fewest (1, 2, 4) that still fit one workgroup per CU, so that the
// tasks spread over the CUs with one wave per SIMD instead of filling a few CUs with two (profiles/r5_small_scan.txt, last part).
// Variant bit 4194304 keeps the packed shape (A/B).
inline uint32_t live_waves(const tfft_plan* p, uint64_t units) {
  if (p->variant & 4194304) return 8;
  const uint64_t cus = static_cast<uint64_t>(p->num_cus);
  for (uint32_t live = 1; live <= 4; live *= 2)
    if (units <= cus * live) return live;
  return 8;
}

template <int V>
int launch_k4096_v(const tfft_plan* p, const void* in_re, const void* in_im, void* out_re, void* out_im,
                   k4096::Addr in_stride, k4096::Addr out_stride, hipStream_t s) {
  const uint32_t live = live_waves(p, p->batch);
  const uint32_t blocks_needed = static_cast<uint32_t>((p->batch + live - 1) / live);
  // Workgroups are sized so that each wave runs about two transforms: the second one's HBM->LDS copy flies under
  // the first one's stores, and the hardware dispatcher hands out the remaining workgroups as CUs drain, which keeps
  // the CUs out of lock-step (measured: 256 persistent workgroups 5.3 TB/s, two transforms per wave 6.1 TB/s, one
  // transform per wave 5.1 TB/s; profiles/r1_k4096_grid_scan.txt).
  static const uint32_t iters_env = env_iters("TFFT_K4096_ITERS", 0);   // experiment knob (debug build only)
  const uint32_t iters = iters_env ? iters_env : plan_iters(p->launch_iters, blocks_needed >= 4u * static_cast<uint32_t>(p->num_cus) ? 2u : 1u);
  const uint32_t grid = pick_grid(blocks_needed, p->num_cus, iters);
  if constexpr (V == (k4096::kStageOut | k4096::kNonTemporal)) {
    if (p->otw.n_mask) {       // row pass of a transposed-input plan (default variant only, create_transposed_in)
      TFFT_LAUNCH((k4096::fft4096_kernel<V, true>), dim3(grid), dim3(k4096::kThreads), k4096::kLdsBytes, s,
                         static_cast<const uint16_t*>(in_re), static_cast<const uint16_t*>(in_im),
                         static_cast<uint16_t*>(out_re), static_cast<uint16_t*>(out_im), in_stride, out_stride,
                         static_cast<uint32_t>(p->batch), live, static_cast<const uint8_t*>(p->d_tables), p->otw);
      return TFFT_OK;
    }
  }
  TFFT_LAUNCH((k4096::fft4096_kernel<V, false>), dim3(grid), dim3(k4096::kThreads), k4096::kLdsBytes, s,
                     static_cast<const uint16_t*>(in_re), static_cast<const uint16_t*>(in_im),
                     static_cast<uint16_t*>(out_re), static_cast<uint16_t*>(out_im), in_stride, out_stride,
                     static_cast<uint32_t>(p->batch), live, static_cast<const uint8_t*>(p->d_tables), p->otw);
  return TFFT_OK;
}

int launch_k256(const tfft_plan* p, const void* in_re, const void* in_im, void* out_re, void* out_im,
                k4096::Addr in_stride, k4096::Addr out_stride, hipStream_t s) {
  const uint64_t groups = (p->batch + k256::kFftsPerWave - 1) / k256::kFftsPerWave;
  const uint32_t live = live_waves(p, groups);
  const uint32_t blocks_needed = static_cast<uint32_t>((groups + live - 1) / live);
  static const uint32_t iters_dflt = env_iters("TFFT_K256_ITERS", 2);
  const uint32_t grid = pick_grid(blocks_needed, p->num_cus, plan_iters(p->launch_iters, iters_dflt));
  if (p->otw.n_mask)
    TFFT_LAUNCH(k256::fft256_kernel<true>, dim3(grid), dim3(k4096::kThreads), k256::kLdsBytes, s,
                       static_cast<const uint16_t*>(in_re), static_cast<const uint16_t*>(in_im),
                       static_cast<uint16_t*>(out_re), static_cast<uint16_t*>(out_im), in_stride, out_stride,
                       static_cast<uint32_t>(p->batch), live, static_cast<const uint8_t*>(p->d_tables), p->otw);
  else
    TFFT_LAUNCH(k256::fft256_kernel<false>, dim3(grid), dim3(k4096::kThreads), k256::kLdsBytes, s,
                       static_cast<const uint16_t*>(in_re), static_cast<const uint16_t*>(in_im),
                       static_cast<uint16_t*>(out_re), static_cast<uint16_t*>(out_im), in_stride, out_stride,
                       static_cast<uint32_t>(p->batch), live, static_cast<const uint8_t*>(p->d_tables), p->otw);
  return TFFT_OK;
}

template <int R, bool STG>
int launch_k256r_t(const tfft_plan* p, const void* in_re, const void* in_im, void* out_re, void* out_im,
                   k4096::Addr in_stride, k4096::Addr out_stride, hipStream_t s) {
  const uint64_t groups = (p->batch + (16 / R) - 1) / (16 / R);
  const uint32_t live = live_waves(p, groups);
  const uint32_t blocks_needed = static_cast<uint32_t>((groups + live - 1) / live);
  static const uint32_t iters_dflt = env_iters("TFFT_K256_ITERS", 2);
  const uint32_t grid = pick_grid(blocks_needed, p->num_cus, plan_iters(p->launch_iters, iters_dflt));
  if constexpr (STG) {
    if (p->otw.n_mask) {       // row pass of a transposed-input plan (staged stores only, create_transposed_in)
      TFFT_LAUNCH((k256r::fft256r_kernel<R, true, true>), dim3(grid), dim3(k4096::kThreads), k256r::lds_bytes<R>(), s,
                         static_cast<const uint16_t*>(in_re), static_cast<const uint16_t*>(in_im),
                         static_cast<uint16_t*>(out_re), static_cast<uint16_t*>(out_im), in_stride, out_stride,
                         static_cast<uint32_t>(p->batch), live, static_cast<const uint8_t*>(p->d_tables), p->otw);
      return TFFT_OK;
    }
  }
  TFFT_LAUNCH((k256r::fft256r_kernel<R, STG, false>), dim3(grid), dim3(k4096::kThreads), k256r::lds_bytes<R>(), s,
                     static_cast<const uint16_t*>(in_re), static_cast<const uint16_t*>(in_im),
                     static_cast<uint16_t*>(out_re), static_cast<uint16_t*>(out_im), in_stride, out_stride,
                     static_cast<uint32_t>(p->batch), live, static_cast<const uint8_t*>(p->d_tables), p->otw);
  return TFFT_OK;
}

int launch_k256r(const tfft_plan* p, int radix, const void* in_re, const void* in_im, void* out_re, void* out_im,
                 k4096::Addr in_stride, k4096::Addr out_stride, hipStream_t s) {
  const bool direct = p->variant & 1048576;   // 8-byte stores straight from registers instead of staged full rows
  switch (radix) {
    case 2:
      return direct ? launch_k256r_t<2, false>(p, in_re, in_im, out_re, out_im, in_stride, out_stride, s)
                    : launch_k256r_t<2, true>(p, in_re, in_im, out_re, out_im, in_stride, out_stride, s);
    case 4:
      return direct ? launch_k256r_t<4, false>(p, in_re, in_im, out_re, out_im, in_stride, out_stride, s)
                    : launch_k256r_t<4, true>(p, in_re, in_im, out_re, out_im, in_stride, out_stride, s);
    default:
      return direct ? launch_k256r_t<8, false>(p, in_re, in_im, out_re, out_im, in_stride, out_stride, s)
                    : launch_k256r_t<8, true>(p, in_re, in_im, out_re, out_im, in_stride, out_stride, s);
  }
}

template <int R>
int launch_k4096r_t(const tfft_plan* p, const void* in_re, const void* in_im, void* out_re, void* out_im,
                    k4096::Addr in_stride, k4096::Addr out_stride, hipStream_t s) {
  // transforms per workgroup iteration: 8 / R, or ONE while that still gives every transform a CU of its own (the R waves of a
  // transform then have the SIMDs to themselves: 2^13 x 4 11.7 -> 7.6 us, 2^14 x 2 12.2 -> 8.5 us)
  const uint64_t cus = static_cast<uint64_t>(p->num_cus);
  const uint32_t per_wg = (p->variant & 4194304) ? k4096::kWavesPerBlock / R
                          : (R < 8 && p->batch <= cus)   ? 1u
                          : (R == 2 && p->batch <= 2 * cus) ? 2u      // (four waves: still one per SIMD)
                                                            : k4096::kWavesPerBlock / R;
  const uint32_t blocks_needed = static_cast<uint32_t>((p->batch + per_wg - 1) / per_wg);
  // persistent workgroups: with four workgroup barriers per transform the short-lived launch shape of the 4096
  // kernel does not help here (measured at 2^13: 405 / 425 / 440 / 457 Gsamples/s for 1 / 2 / 4 / all iterations)
  const uint32_t grid = gens_grid(blocks_needed, static_cast<uint32_t>(p->num_cus), p->launch_iters, kGensStatic);
#ifdef TFFT_DEBUG_KERNELS
  unsigned long long* stamps1d = nullptr;      // measurement hook of tools/exp_k4096r_phases.py
  if (debug_variants_enabled())
    if (const char* e = std::getenv("TFFT_ROWS_STAMPS_PTR")) stamps1d = reinterpret_cast<unsigned long long*>(std::strtoull(e, nullptr, 0));
#define TFFT_NO_STAMPS , stamps1d
#else
#define TFFT_NO_STAMPS
#endif
  if (p->otw.n_mask)           // row pass of a transposed-input plan
    TFFT_LAUNCH((k4096r::fft4096r_kernel<R, false, true>), dim3(grid), dim3(k4096::kThreads), k4096::kLdsBytes, s,
                       static_cast<const uint16_t*>(in_re), static_cast<const uint16_t*>(in_im),
                       static_cast<uint16_t*>(out_re), static_cast<uint16_t*>(out_im), in_stride, out_stride,
                       static_cast<uint32_t>(p->batch), per_wg, static_cast<const uint8_t*>(p->d_tables), p->otw TFFT_NO_STAMPS);
  else
    TFFT_LAUNCH((k4096r::fft4096r_kernel<R, false, false>), dim3(grid), dim3(k4096::kThreads), k4096::kLdsBytes, s,
                       static_cast<const uint16_t*>(in_re), static_cast<const uint16_t*>(in_im),
                       static_cast<uint16_t*>(out_re), static_cast<uint16_t*>(out_im), in_stride, out_stride,
                       static_cast<uint32_t>(p->batch), per_wg, static_cast<const uint8_t*>(p->d_tables), p->otw TFFT_NO_STAMPS);
#undef TFFT_NO_STAMPS
  return TFFT_OK;
}

int launch_k4096r(const tfft_plan* p, int radix, const void* in_re, const void* in_im, void* out_re, void* out_im,
                  k4096::Addr in_stride, k4096::Addr out_stride, hipStream_t s) {
  switch (radix) {
    case 2: return launch_k4096r_t<2>(p, in_re, in_im, out_re, out_im, in_stride, out_stride, s);
    case 4: return launch_k4096r_t<4>(p, in_re, in_im, out_re, out_im, in_stride, out_stride, s);
    default: return launch_k4096r_t<8>(p, in_re, in_im, out_re, out_im, in_stride, out_stride, s);
  }
}

// first pass of the fused 2D plan: iterations = images * 512 (k4096r.hpp, ROWS); p only lends its device and tables
int launch_rows2d(const tfft_plan* p, const void* in_re, const void* in_im, void* out_re, void* out_im,
                  uint64_t image_stride, uint32_t iterations, hipStream_t s) {
  const uint32_t grid = gens_grid(iterations, static_cast<uint32_t>(p->num_cus), 0, kGensStatic);
#ifdef TFFT_DEBUG_KERNELS
  unsigned long long* stamps = nullptr;      // measurement hook of tools/exp_rows_phases.py
  if (debug_variants_enabled())
    if (const char* e = std::getenv("TFFT_ROWS_STAMPS_PTR")) stamps = reinterpret_cast<unsigned long long*>(std::strtoull(e, nullptr, 0));
#endif
  TFFT_LAUNCH((k4096r::fft4096r_kernel<8, true>), dim3(grid), dim3(k4096::kThreads), k4096::kLdsBytes, s,
                     static_cast<const uint16_t*>(in_re), static_cast<const uint16_t*>(in_im),
                     static_cast<uint16_t*>(out_re), static_cast<uint16_t*>(out_im), k4096::Addr{image_stride, image_stride, 0, 0},
                     k4096::Addr{image_stride, image_stride, 0, 0}, iterations, 1u, static_cast<const uint8_t*>(p->d_tables), k4096::OutTw{}
#ifdef TFFT_DEBUG_KERNELS
                     , stamps
#endif
                     );
  if (!g_prepare) TFFT_HIP(hipGetLastError());
  return TFFT_OK;
}

int launch_k4096(const tfft_plan* p, const void* in_re, const void* in_im, void* out_re, void* out_im,
                 k4096::Addr in_stride, k4096::Addr out_stride, hipStream_t s) {
  // opts.variant: 0 = default (staged, coalesced, non-temporal stores: the fastest measured on MI355X);
  // otherwise a mask of k4096::kPrefetch / kStageOut / kFakeStore / kNonTemporal, with 16 = "none of them".
  const int v = (p->variant & (15 | 16 | 64)) == 0 ? (k4096::kStageOut | k4096::kNonTemporal) : (p->variant & (15 | 64));
#define TFFT_V(N) case N: return launch_k4096_v<N>(p, in_re, in_im, out_re, out_im, in_stride, out_stride, s)
  switch (v) {
    TFFT_V(0); TFFT_V(1); TFFT_V(2); TFFT_V(8); TFFT_V(9); TFFT_V(10);
#ifdef TFFT_DEBUG_KERNELS      // timing-only instantiations (WRONG output): fake stores / no compute
    TFFT_V(4); TFFT_V(5); TFFT_V(13); TFFT_V(64); TFFT_V(72); TFFT_V(73);
#endif
    default: return fail(TFFT_ERR_ARG, "unknown kernel variant");
  }
#undef TFFT_V
}

struct Planes {
  _Float16* re;
  _Float16* im;
  uint64_t stride;
};

// ---------------------------------------------------------------------------------------------------------------------
// The column kernels this library ships: ONE list per kernel family. The dispatch table below, the names tfft_kernel_list()
// reports and (through tests/test_isa_lint.py, which compares that list with the symbols of the gfx950 code object) the set of
// instantiations in libtfft.so all come from these lists: a further variant is a further row, not a further branch of a ladder.
// Arguments are written the way the demangler prints them (bools as false / true), because the row's name is built from them.
// ---------------------------------------------------------------------------------------------------------------------
// colfft256_kernel<MODE, TW, STAGE, LUT>: per-wave radix-256 pass (16-column tiles)
#define TFFT_COL_WAVE(X)                                                                                     \
  X(0, 1, false, false) X(0, 1, false, true) X(0, 1, true, false) X(0, 1, true, true) X(0, 0, false, false)  \
  X(0, 0, true, false) X(1, 1, false, false) X(1, 1, false, true) X(1, 1, true, false) X(1, 1, true, true)   \
  X(1, 0, false, false) X(1, 0, true, false)
// colfft256_wg_kernel<MODE, TW, NT, W, STG>: workgroup-cooperative radix-256 pass, W = 4 / 8 waves; STG only with MODE 0
#define TFFT_COL_WG256_W(X, W)                                                                               \
  X(0, 0, false, W, false) X(0, 0, false, W, true) X(0, 0, true, W, false) X(0, 0, true, W, true)            \
  X(0, 1, false, W, false) X(0, 1, false, W, true) X(0, 1, true, W, false) X(0, 1, true, W, true)            \
  X(1, 0, false, W, false) X(1, 0, true, W, false) X(1, 1, false, W, false) X(1, 1, true, W, false)          \
  X(1, 2, false, W, false) X(1, 2, true, W, false)
#define TFFT_COL_WG256(X) TFFT_COL_WG256_W(X, 4) TFFT_COL_WG256_W(X, 8)
// colfft512_wg_kernel<MODE, TW, SC, PLAIN> and colfft1024_wg_kernel<MODE, TW, SC, PLAIN>; SC = "scale once" read-out of a final pass
#define TFFT_COL_512(X)                                                                                      \
  X(0, 0, false, false) X(0, 0, false, true) X(0, 1, false, false) X(0, 1, false, true) X(1, 0, false, false) \
  X(1, 0, false, true) X(1, 0, true, false) X(1, 1, false, false) X(1, 1, false, true) X(1, 2, false, false)  \
  X(1, 2, false, true)
#define TFFT_COL_1024(X)                                                                                     \
  X(0, 0, false, false) X(0, 0, false, true) X(0, 1, false, false) X(0, 1, false, true) X(1, 0, false, false) \
  X(1, 0, false, true) X(1, 0, true, false) X(1, 1, false, false) X(1, 1, false, true)
// colfft512r_wg_kernel<W, SC, PF, PLAIN>: two-round radix-512 pass (PF = next tile prefetched through registers: the 8-wave form)
#define TFFT_COL_512R(X)                                                                                     \
  X(8, false, true, false) X(8, false, true, true) X(8, true, true, false) X(4, false, false, false)         \
  X(4, false, false, true) X(4, true, false, false)

// collat256_kernel<MODE, TW, CG, HH, PP>: radix-256 pass for work that does not fill the chip (collat.hpp): 16 CG columns per
// workgroup, a column group's stage 2 split over HH waves and PP = 1 or 2 workgroups
#define TFFT_COL_LAT_S(X, CG, HH, PP) X(0, 0, CG, HH, PP) X(0, 1, CG, HH, PP) X(1, 0, CG, HH, PP) X(1, 1, CG, HH, PP)
#define TFFT_COL_LAT(X)                                                                                      \
  TFFT_COL_LAT_S(X, 4, 2, 1) TFFT_COL_LAT_S(X, 2, 2, 1) TFFT_COL_LAT_S(X, 1, 4, 1) TFFT_COL_LAT_S(X, 2, 2, 2) \
  TFFT_COL_LAT_S(X, 1, 4, 2)

enum : uint32_t { kFamWave = 1, kFamWg256 = 2, kFam512 = 3, kFam512R = 4, kFam1024 = 5, kFamLat = 6 };
using ColKernel = void (*)(colfft::Args);
struct ColRow {
  uint32_t key;
  ColKernel fn;
  uint32_t threads, lds;
  const char* name;
};
constexpr uint32_t col_key(uint32_t fam, int a, int b, int c, int d, int e = 0) {
  return (fam << 20) | (static_cast<uint32_t>(a) << 16) | (static_cast<uint32_t>(b) << 12) | (static_cast<uint32_t>(c) << 8) |
         (static_cast<uint32_t>(d) << 4) | static_cast<uint32_t>(e);
}
const ColRow kColTable[] = {
#define X(MODE, TW, STAGE, LUT)                                                                                      \
  {col_key(kFamWave, MODE, TW, STAGE, LUT), colfft::colfft256_kernel<MODE, TW, STAGE, LUT>, k4096::kThreads, colfft::kLdsBytes, \
   "colfft::colfft256_kernel<" #MODE ", " #TW ", " #STAGE ", " #LUT ">"},
    TFFT_COL_WAVE(X)
#undef X
#define X(MODE, TW, NT, W, STG)                                                                                      \
  {col_key(kFamWg256, MODE, TW, NT, W, STG), colfft::colfft256_wg_kernel<MODE, TW, NT, W, STG>, 64 * W, colfft::WgGeom<W>::kLds,   \
   "colfft::colfft256_wg_kernel<" #MODE ", " #TW ", " #NT ", " #W ", " #STG ">"},
    TFFT_COL_WG256(X)
#undef X
#define X(MODE, TW, SC, PLAIN)                                                                                       \
  {col_key(kFam512, MODE, TW, SC, PLAIN), colfft::colfft512_wg_kernel<MODE, TW, SC, PLAIN>, k4096::kThreads, colfft::kWg512LdsBytes, \
   "colfft::colfft512_wg_kernel<" #MODE ", " #TW ", " #SC ", " #PLAIN ">"},
    TFFT_COL_512(X)
#undef X
#define X(MODE, TW, SC, PLAIN)                                                                                       \
  {col_key(kFam1024, MODE, TW, SC, PLAIN), colfft::colfft1024_wg_kernel<MODE, TW, SC, PLAIN>, k4096::kThreads, colfft::kWg1024LdsBytes, \
   "colfft::colfft1024_wg_kernel<" #MODE ", " #TW ", " #SC ", " #PLAIN ">"},
    TFFT_COL_1024(X)
#undef X
#define X(W, SC, PF, PLAIN)                                                                                          \
  {col_key(kFam512R, W, SC, PF, PLAIN), colfft::colfft512r_wg_kernel<W, SC, PF, PLAIN>, 64 * W, colfft::wg512r_lds_bytes<W>(),  \
   "colfft::colfft512r_wg_kernel<" #W ", " #SC ", " #PF ", " #PLAIN ">"},
    TFFT_COL_512R(X)
#undef X
#define X(MODE, TW, CG, HH, PP)                                                                                      \
  {col_key(kFamLat, MODE, TW, CG, HH, PP), colfft::collat256_kernel<MODE, TW, CG, HH, PP>, colfft::LatGeom<CG, HH, PP>::kThreads, \
   colfft::LatGeom<CG, HH, PP>::kLds, "colfft::collat256_kernel<" #MODE ", " #TW ", " #CG ", " #HH ", " #PP ">"},
    TFFT_COL_LAT(X)
#undef X
};
constexpr size_t kColRows = sizeof(kColTable) / sizeof(kColTable[0]);

inline const ColRow* col_row(uint32_t key) {
  for (const ColRow& r : kColTable)
    if (r.key == key) return &r;
  return nullptr;
}

// the one launch site of every column kernel (lds_bytes = 0: the row's own LDS size)
int launch_col_row(const tfft_plan* p, uint32_t key, uint32_t grid, const colfft::Args& a, hipStream_t s, uint32_t lds_bytes = 0) {
  const ColRow* const r = col_row(key);
  if (!r) return fail(TFFT_ERR_ARG, "internal error: column kernel " + std::to_string(key) + " is not in the dispatch table");
  const uint32_t lds = lds_bytes ? lds_bytes : r->lds;
  const int rc = lds_opt_in(reinterpret_cast<const void*>(r->fn), p->device, static_cast<int>(lds));
  if (rc) return rc;
  if (!g_prepare) hipLaunchKernelGGL(r->fn, dim3(grid), dim3(r->threads), lds, s, a);
  return TFFT_OK;
}

// per-wave kernel: 8 waves per workgroup, one 16-column tile per wave and round
int launch_col_wave(const tfft_plan* p, int mode, int tw, bool stage, bool lut, const colfft::Args& a, hipStream_t s) {
  const uint32_t blocks_needed = (a.tasks + k4096::kWavesPerBlock - 1) / k4096::kWavesPerBlock;
  static const uint32_t iters_dflt = env_iters("TFFT_COL_ITERS", 1000000);
  const uint32_t grid = pick_grid(blocks_needed, p->num_cus, plan_iters(p->launch_iters, iters_dflt));
  return launch_col_row(p, col_key(kFamWave, mode, tw, stage, tw == colfft::kTwNone ? false : lut), grid, a, s);
}

// workgroup-cooperative radix-256 kernel, W = 4 or 8 waves
int launch_col_wg(const tfft_plan* p, int mode, int tw, int w, const colfft::Args& a_in, hipStream_t s) {
  const uint32_t cols = 16u * static_cast<uint32_t>(w);
  colfft::Args a = a_in;
  uint64_t blocks = (a.tasks / a.groups) * a.pitch / cols;
  if (g_slab.on) {                      // a slab of the pass's columns (whole blocks: the distributed plan checks the divisibility)
    if (g_slab.col_first % cols || g_slab.col_count % cols || tw != colfft::kTwFourStep)
      return fail(TFFT_ERR_ARG, "internal error: column slab not a whole number of blocks of a four-step pass");
    a.blk_first = static_cast<uint32_t>(g_slab.col_first / cols);
    a.blk_count = static_cast<uint32_t>(g_slab.col_count / cols);
    blocks = a.blk_count;
  }
  // non-temporal copy-in and row stores unless the plan's cache policy says plain (tfft_plan_cache_policy, variant bit 262144);
  // columns-on-lanes form: staged full-row stores (variant bit 1048576: direct 16-byte pieces)
  const bool nt = !p->plain_acc;
  const bool stg = mode == colfft::kColsOnLanes && !(p->variant & 1048576);
  const uint32_t key = col_key(kFamWg256, mode, tw, nt, w, stg);
  static const uint32_t iters_dflt = env_iters("TFFT_COLWG_ITERS", 1000000);
#ifdef TFFT_DEBUG_KERNELS
  // experiment knob: TFFT_WG4_ONE_PER_CU=1 launches the 4-wave workgroups with so much dynamic LDS (96 KiB) that only ONE fits a CU:
  // the same kernel at one wave per SIMD instead of two (what a radix-1024 pass with 128-column tiles would have to run at).
  // Round 4, against the static partition: +3 ... +10 % on one box; against today's default (8 generations of two per CU), as a
  // variant bit in one process: -3 ... -20 % (profiles/r4_one_wave_per_simd.txt): not a launch shape worth keeping.
  static const bool one_per_cu = env_iters("TFFT_WG4_ONE_PER_CU", 0) != 0;
  if (one_per_cu && w == 4)
    return launch_col_row(p, key, gens_grid(blocks, static_cast<uint32_t>(p->num_cus), p->launch_iters, env_iters("TFFT_GENS", 1)), a, s, 96 * 1024);
#endif
  const uint32_t capacity = static_cast<uint32_t>(p->num_cus * (8 / w));
  const uint32_t grid = iters_dflt != 1000000u ? pick_grid(blocks, static_cast<int>(capacity), plan_iters(p->launch_iters, iters_dflt))
                                               : rounds_grid(blocks, capacity, p->launch_iters);
  return launch_col_row(p, key, grid, a, s);
}

// which (MODE, TW) a radix-256 pass needs: the four-step form, columns on lanes for the first pass of a plain transform
// (Ns = 1), columns in registers otherwise
inline void col_mode_tw(const tfft_plan* p, const Pass& ps, const colfft::Args& a, int& mode, int& tw) {
  if (p->tw4_modulus) {
    mode = colfft::kColsInRegs;
    tw = colfft::kTwFourStep;
    return;
  }
  mode = a.ns_f == 1 ? colfft::kColsOnLanes : colfft::kColsInRegs;
  tw = ps.tw_next ? colfft::kTwNext : colfft::kTwNone;
}

int launch_col(const tfft_plan* p, const Pass& ps, Planes src, Planes dst, hipStream_t s) {
  colfft::Args a;
  a.in_re = reinterpret_cast<const uint16_t*>(src.re);
  a.in_im = reinterpret_cast<const uint16_t*>(src.im);
  a.out_re = reinterpret_cast<uint16_t*>(dst.re);
  a.out_im = reinterpret_cast<uint16_t*>(dst.im);
  a.in_stride = src.stride;
  a.out_stride = dst.stride;
  const uint64_t radix = static_cast<uint64_t>(ps.radix);   // 256, or 512 (columns-in-registers form only)
  a.pitch = (p->n / radix) * p->inner;
  a.ns_f = ps.ns * p->inner;
  a.ns_f_shift = static_cast<uint32_t>(ilog2(a.ns_f));
  a.groups = static_cast<uint32_t>(a.pitch / 16);
  a.tasks = static_cast<uint32_t>(a.groups * p->batch);
  a.inner_shift = static_cast<uint32_t>(ilog2(p->inner));
  a.ns = ps.ns;
  a.tw_lo = p->d_tw_lo;
  a.tw_hi = p->d_tw_hi;
  a.tables = static_cast<const uint8_t*>(p->d_tables);
  a.n_mask = (p->tw4_modulus ? p->tw4_modulus : p->n) - 1;
  a.tw_scale = ps.tw_scale;
  a.comb_scale = ps.scale;
  a.tw4_col0 = p->tw4_col0;
  const bool first_pass = &ps == &p->passes[0];
  a.in_seg_shift = first_pass ? p->in_seg_shift : 31;
  a.in_seg_gap = first_pass ? p->in_seg_gap : 0;
  a.blk_first = 0;
  a.blk_count = 0;
  a.out_pitch_shift = static_cast<uint32_t>(ilog2(ps.ns * p->inner));
  a.out_seg_shift = 31;
  a.out_seg_gap = 0;
  a.out_col0 = 0;
  a.out_base = 0;
  if (g_slab.on) {
    if (!(p->tw4_modulus && ps.radix == 256 && p->passes.size() == 1))
      return fail(TFFT_ERR_ARG, "internal error: a column slab needs a single four-step radix-256 pass");
    a.out_pitch_shift = g_slab.out_pitch_shift;
    a.out_seg_shift = g_slab.out_seg_shift;
    a.out_seg_gap = g_slab.out_seg_gap;
    a.out_col0 = g_slab.col_first;
    a.out_base = g_slab.out_base;
  }
#ifdef TFFT_DEBUG_KERNELS
  a.wg_times = nullptr;
  if (debug_variants_enabled())          // measurement hook of tools/exp_wg_end_times.py
    if (const char* e = std::getenv("TFFT_WG_TIMES_PTR"))     // (one block of 16 x 8192 words per pass of the plan)
      a.wg_times = reinterpret_cast<unsigned long long*>(std::strtoull(e, nullptr, 0)) + (&ps - &p->passes[0]) * 16 * 8192;
  a.copy_only = (p->variant & 65536) ? 1u : 0u;
#endif
  a.out_row_shift = p->out_row_shift;
  a.out_sub_shift = p->out_sub_shift;
  a.out_sub_stride = p->out_sub_stride;
  a.a_shift = 0;
  a.t_mask = 0;
  a.n_over_t = 1;
  a.inv_t = 1.0;
  if (ps.tw_next) {
    // next pass: radix R', Ns'' = ns * 256; it wants w_T^(i' k''), T = Ns'' R', on element
    // o = rest (ns_f 256) + k ns_f + kprev_f:  k'' = k ns + kprev,  i' = o / (n_f / R') = rest >> a_shift
    const uint64_t t = ps.ns * radix * static_cast<uint64_t>(ps.next_radix);
    a.t_mask = t - 1;
    a.n_over_t = p->n / t;
    a.inv_t = 1.0 / static_cast<double>(t);
    a.a_shift = static_cast<uint32_t>(ilog2(p->n / (static_cast<uint64_t>(ps.next_radix) * ps.ns * radix)));
  }
  const bool plain = p->plain_acc;
  const bool sc = ps.scale != 1.0f;              // TFFT_SCALE_ONCE, last pass: the single factor in fp32 at the read-out / combine
  if (radix == 1024 || radix == 512) {
    // (plan creation only emits these passes where the geometry fits: pitch, and ns_f unless it is 1, multiples of 64)
    const uint64_t blocks = (a.tasks / a.groups) * a.pitch / 64;
    const uint32_t grid = gens_grid(blocks, static_cast<uint32_t>(p->num_cus), p->launch_iters, kGensCol8);
    const uint32_t fam = radix == 1024 ? kFam1024 : kFam512;
    if (a.ns_f == 1) return launch_col_row(p, col_key(fam, colfft::kColsOnLanes, ps.tw_next ? colfft::kTwNext : colfft::kTwNone, false, plain), grid, a, s);
    if (radix == 512 && p->tw4_modulus) return launch_col_row(p, col_key(fam, colfft::kColsInRegs, colfft::kTwFourStep, false, plain), grid, a, s);
    if (ps.tw_next) return launch_col_row(p, col_key(fam, colfft::kColsInRegs, colfft::kTwNext, false, plain), grid, a, s);
    if (radix == 512 && (((a.pitch == 256 || a.pitch == 512) && a.ns_f % 128 == 0) != ((p->variant & 268435456) != 0))) {
      // last pass of a plan / 2D column pass by the two-round kernel (colfft512r.hpp). A/B in one process on MI355X, 8 GiB per
      // launch (profiles/r3_ab_colfft512r.txt): the 128-column two-round form is 2-4 % faster than the 8-wave single-round
      // kernel at row pitches of 256 and 512 columns (2^18 = 512 x 512: 335 -> 342 Gsamples/s) and at 2048, 2-4 % slower at
      // 128, 1024 and 4096 (the 2D column pass); the default follows that, variant bit 268435456 flips the choice
      // 128-column tiles (256-byte row segments, one 8-wave workgroup per CU) where the geometry allows and variant bit
      // 524288 does not ask for 4-wave workgroups; otherwise 64-column tiles, two 4-wave workgroups per CU
      const bool w8 = !(p->variant & 524288) && a.pitch % 128 == 0 && a.ns_f % 128 == 0;
      const uint32_t grid2 = gens_grid(w8 ? blocks / 2 : blocks, static_cast<uint32_t>((w8 ? 1 : 2) * p->num_cus), p->launch_iters, kGensCol8);
      return launch_col_row(p, col_key(kFam512R, w8 ? 8 : 4, sc, w8, !sc && plain), grid2, a, s);
    }
    return launch_col_row(p, col_key(fam, colfft::kColsInRegs, colfft::kTwNone, sc, !sc && plain), grid, a, s);
  }
  int mode, tw;
  col_mode_tw(p, ps, a, mode, tw);
  // default: stores straight from registers (8- / 16-byte pieces); variant bit 4096: stage the output through
  // LDS (16-byte coalesced stores). Measured in one process on MI355X: direct wins at 2^16 and 2^20, staging at 2^13.
  // workgroup-cooperative form (full 256-byte row segments) whenever the geometry allows; variant bit 131072
  // forces the per-wave kernel
  const bool wg_allowed = !(p->variant & (131072 | 4096 | 8192));
  // per-wave kernel: variant bit 4096 = LDS-staged stores, 8192 = twiddles from v_sin / v_cos instead of the two-level tables
  const bool stage = p->variant & 4096, lut = !(p->variant & 8192);
  const uint64_t entries = a.tasks / a.groups;
  // narrow pitch (N = 256 pitch contiguous, columns-on-lanes form): a workgroup spans 128 / pitch whole batch
  // entries; entries that do not fill a workgroup go to the per-wave kernel in a second launch.
  // (a few entries of 64 columns, the first pass of a small batch of 2^14 = 256 x 64: the latency kernel below)
  const bool lat_narrow = (a.pitch == 64 || a.pitch == 32) && entries <= 64 && tw != colfft::kTwFourStep && !(p->variant & kVarNoLat);
  if (wg_allowed && a.ns_f == 1 && a.pitch >= 16 && a.pitch < 128 && !lat_narrow) {
    const uint64_t per = 128 / a.pitch;
    const uint64_t main_entries = entries - entries % per;
    if (main_entries) {
      colfft::Args am = a;
      am.tasks = static_cast<uint32_t>(main_entries * a.groups);
      const int rc = launch_col_wg(p, mode, tw, 8, am, s);
      if (rc || main_entries == entries) return rc;
      a.in_re += main_entries * a.in_stride;
      a.in_im += main_entries * a.in_stride;
      a.out_re += main_entries * a.out_stride;
      a.out_im += main_entries * a.out_stride;
      a.tasks = static_cast<uint32_t>((entries - main_entries) * a.groups);
    }
    if (a.pitch % 64 == 0) return launch_col_wg(p, mode, tw, 4, a, s);   // 64 columns: one 4-wave workgroup per entry
    return launch_col_wave(p, mode, tw, false, true, a, s);
  }
  const bool wg8_ok = (a.pitch % 128 == 0) && (a.ns_f == 1 || a.ns_f % 128 == 0);
  const bool wg4_ok = (a.pitch % 64 == 0) && (a.ns_f == 1 || a.ns_f % 64 == 0);
  // Work that does not fill the chip (the reference's single-transform benchmark, FFTBenchSinlge.cu): up to 64 blocks of 64
  // columns (2^20 samples per pass) -> the latency kernel (collat.hpp: one memory round trip before the block is in LDS, a column
  // group's stage 2 split over waves on different SIMDs). Variant bit 1073741824 keeps the throughput kernels (A/B, tuner).
  // Device time per execution, latency / throughput kernels (profiles/r5_small_scan.txt): 2^16 x 1: 8.8 / 14.3 us, x 16: 13.1 /
  // 16.3; 2^18 x 1 as 256 x 256 x 4: 12.8 / 18.1; 2^20 x 1: 18.4 / 22.2. Beyond 64 blocks the sign depends on the pass (2^16 x 32:
  // 20.0 / 17.7, 2^21 x 1: 26.0 / 28.7, 2^17 x 32: 26.7 / 23.3): the throughput kernels keep everything from there on. 128 blocks
  // still win by 7-9 % wherever a row is at least 512 columns wide (2^21 x 1, 2^20 x 2: 26.5 / 28.6, 2^19 x 4: 26.1 / 28.0, 2^18 x 8
  // as 256 x 256 x 4: 25.9 / 27.6) and lose 13 % at a pitch of 256 (2^16 x 32, above).
  // (in units of 16 columns: a pitch of 32, the first pass of 2^13 = 256 x 32, is two 16-column blocks per entry)
  const uint64_t blocks16 = entries * a.pitch / 16, blocks64 = (blocks16 + 3) / 4;
  const bool lat_geom = wg4_ok || (a.pitch == 32 && a.ns_f == 1);
  // (tfft_plan_opts.launch_iters shapes the grids of the grid-stride kernels; this kernel's grid is one workgroup per block either way,
  // so a launch shape never changes WHICH kernel runs, and with it the bits: test_launch_shape_never_changes_results)
  if (wg_allowed && lat_geom && tw != colfft::kTwFourStep && !(p->variant & kVarNoLat) && blocks64 <= (a.pitch >= 512 ? 128u : 64u)) {
    // Workgroup shape (column groups of 16 per workgroup, waves per column group): stage 2 is bound by instruction issue, so the
    // finer the split the shorter the pass - until the row segments get too narrow for the memory system (32-byte segments over
    // 4 MiB: loads land after 2.3 us instead of 0.9, tools/lat_probe). One box, device time per transform, shapes 4 x 2 / 2 x 2 /
    // 1 x 4 (profiles/r5_lat_shapes.txt): 2^16: 12.4 / 9.5 / 8.6 us, 2^18: 16.3 / 13.1 / 12.5, 2^19: 17.2 / 14.5 / 14.8, 2^20:
    // 20.0 / 18.2 / 21.7.
    const int cgs = blocks64 <= 16 ? 1 : 2;
    const int hh = cgs == 1 ? 4 : 2;
    // ... and beyond the workgroup: PP = 2 workgroups per block, each with the whole block in its LDS and half of the stage-2 tiles
    // and of the rows to store, while that still leaves one workgroup per CU (one wave per SIMD is the point) and every wave keeps
    // two tiles. One box, PP = 1 / 2 / 4 (profiles/r5_lat_shapes.txt, second part): 2^16 8.54 / 8.24 us, 2^17 11.50 / 11.18, 2^18
    // 12.45 / 12.25, 2^19 14.39 / 13.80 / 14.17, 2^20 18.12 / 17.77 / 20.23, 2^21 (256 workgroups already) 26.0 / 27.5: 2-4 %, and
    // four-way loses what two-way gains (every partner repeats the loads and stage 1). The partners READ the same block and WRITE
    // disjoint bytes of dst: never for a pass in place.
    const uint32_t wgs = static_cast<uint32_t>(blocks16 / cgs);
    const int pp = (a.in_re != a.out_re && a.in_im != a.out_im && wgs * 2 <= static_cast<uint32_t>(p->num_cus) && 16 / (hh * 2) >= 2) ? 2 : 1;
    int cgs_used = cgs, hh_used = hh, pp_used = pp;
#ifdef TFFT_DEBUG_KERNELS
    if (const uint32_t shape = env_iters("TFFT_LAT_SHAPE", 0)) {      // experiment knob, digits CG HH [PP]: 42, 22, 14; 222, 142
      const uint32_t two = shape >= 100 ? shape / 10 : shape;
      cgs_used = static_cast<int>(two / 10);
      hh_used = static_cast<int>(two % 10);
      pp_used = shape >= 100 ? static_cast<int>(shape % 10) : 1;
    }
#endif
    return launch_col_row(p, col_key(kFamLat, mode, tw, cgs_used, hh_used, pp_used),
                          static_cast<uint32_t>(blocks16 / cgs_used) * static_cast<uint32_t>(pp_used), a, s);
  }
  // variant bit 524288: 4-wave workgroups (two per CU) instead of one 8-wave workgroup
  static const uint32_t wg4_max_pitch_lanes = env_iters("TFFT_WG4_MAX_PITCH", 1024);          // experiment knobs
  // (the columns-in-registers form, whose output is staged behind two more barriers, gains from two workgroups per
  // CU up to a pitch of 16384: 2^20 x 1024 221.6 -> 228.9 Gsamples/s, 2^22 215.5 -> 220.1; beyond that the 128-byte
  // segments lose more than the overlap gives: 2^24 194.6 -> 170.8)
  static const uint32_t wg4_max_pitch_regs = env_iters("TFFT_WG4_MAX_PITCH_INREGS", 16384);
  // Round 3, today's kernels (rotated work distribution, conflict-free staging), W = 4 against W = 8 on one box
  // (profiles/r3_ab_w4_w8.txt): the plain and next-pass-twiddle forms prefer 8-wave workgroups (256-byte segments) from a pitch
  // of 512 columns on: +4 % at 512 (last pass of 2^17: 345 -> 358 Gsamples/s), +6 % at 1024, +10 % at 4096, +4 % at 16384; at 256
  // the two 4-wave workgroups per CU still win by 2-3 %. The four-step form is within 1 % either way and keeps its threshold.
  const uint32_t wg4_regs = p->tw4_modulus ? wg4_max_pitch_regs : std::min<uint32_t>(wg4_max_pitch_regs, 256u);
  const uint32_t wg4_max_pitch = (a.ns_f == 1) ? wg4_max_pitch_lanes : wg4_regs;
  // ... also when 8-wave workgroups would leave CUs idle (single long transforms: 2^20 x 1 is 32 blocks of 128 columns)
  const bool few_blocks = entries * a.pitch / 128 < static_cast<uint64_t>(p->num_cus);
  if (wg_allowed && wg4_ok && ((p->variant & 524288) || !wg8_ok || a.pitch <= wg4_max_pitch || few_blocks))
    return launch_col_wg(p, mode, tw, 4, a, s);
  if (wg_allowed && wg8_ok) return launch_col_wg(p, mode, tw, 8, a, s);
  return launch_col_wave(p, mode, tw, stage, lut, a, s);
}

template <int R>
void launch_pass(const stockham::PassArgs& a, uint64_t batch, hipStream_t s) {
  const uint64_t grid = (a.m_f * batch + stockham::kBlock - 1) / stockham::kBlock;
  if (g_prepare) return;
  hipLaunchKernelGGL(stockham::pass_kernel<R>, dim3(static_cast<uint32_t>(grid)), dim3(stockham::kBlock), 0,
                     s, a);
}

template <int R>
void launch_pass_pair(const stockham::PassArgs& a, uint64_t batch, hipStream_t s) {
  const uint64_t grid = ((a.m_f / 2) * batch + stockham::kBlock - 1) / stockham::kBlock;
  if (g_prepare) return;
  hipLaunchKernelGGL(stockham::pass_pair_kernel<R>, dim3(static_cast<uint32_t>(grid)), dim3(stockham::kBlock), 0,
                     s, a);
}

void launch_stockham_pass(const tfft_plan* p, const Pass& ps, Planes src, Planes dst, hipStream_t s) {
  stockham::PassArgs a;
  a.in_re = src.re;
  a.in_im = src.im;
  a.out_re = dst.re;
  a.out_im = dst.im;
  a.in_stride = src.stride;
  a.out_stride = dst.stride;
  a.n = p->n;
  const int R = ps.radix;
  a.m_f = (p->n / R) * p->inner;
  a.ns = ps.ns * p->inner;
  a.inner_shift = static_cast<uint32_t>(ilog2(p->inner));
  a.skip_tw = ps.skip_tw ? 1u : 0u;
  a.tw_mul = p->n / (ps.ns * R);
  a.batch = p->batch;
  a.m_shift = static_cast<uint32_t>(ilog2(a.m_f));
  a.tw_lo = p->d_tw_lo;
  a.tw_hi = p->d_tw_hi;
  a.scale = ps.scale;
  // pre-twiddled radix-2/4/8 pass with an even sub-transform length: two butterflies per thread, 4-byte accesses
  // (measured +5 % on the whole 2^17 transform; for radix 16 it is neutral in 1D and -6 % on the 2D column pass, so
  // those keep one butterfly per thread). variant bit 4194304 keeps the one-butterfly kernel.
  if (a.skip_tw && a.ns >= 2 && a.m_f >= 2 && R <= 8 && !(p->variant & 4194304)) {
    switch (R) {
      case 2: launch_pass_pair<2>(a, p->batch, s); return;
      case 4: launch_pass_pair<4>(a, p->batch, s); return;
      default: launch_pass_pair<8>(a, p->batch, s); return;
    }
  }
  // Workgroup-cooperative final pass (stockham::tail_coop_kernel): radix 128 always (plan_passes emits it only as 2^15 = 256 x 128),
  // radix 64 / 32 for the last pass of 2^14 = 256 x 64 / 2^13 = 256 x 32 while the batch is small (16-byte row segments: a large batch keeps the
  // butterfly-per-thread kernel, whose accesses are whole lines)
  const bool coop_geom = a.skip_tw && a.ns == a.m_f && a.m_f % stockham::kCoopCols == 0 && p->inner == 1;
  if (R == 128 || ((R == 64 || R == 32) && coop_geom && p->n == 256ull * R && p->batch <= 16 && !(p->variant & 4194304))) {
    if (!coop_geom) {
      (void)fail(TFFT_ERR_ARG, "internal error: radix-128 pass outside its geometry");
      return;
    }
    const dim3 grid(static_cast<uint32_t>(a.m_f / stockham::kCoopCols * p->batch));
    if (!g_prepare) {
      if (R == 128) hipLaunchKernelGGL(stockham::tail_coop_kernel<128>, grid, dim3(256), 0, s, a);
      else if (R == 64) hipLaunchKernelGGL(stockham::tail_coop_kernel<64>, grid, dim3(128), 0, s, a);
      else hipLaunchKernelGGL(stockham::tail_coop_kernel<32>, grid, dim3(64), 0, s, a);
    }
    return;
  }
  switch (R) {
    case 2: launch_pass<2>(a, p->batch, s); break;
    case 4: launch_pass<4>(a, p->batch, s); break;
    case 8: launch_pass<8>(a, p->batch, s); break;
    case 32: launch_pass<32>(a, p->batch, s); break;
    case 64: launch_pass<64>(a, p->batch, s); break;
    default: launch_pass<16>(a, p->batch, s); break;
  }
}

// blocks = 2: a second block behind the first (in-place execution of a plan with an odd number of passes, launch_chain); only
// asked for when the workspace is the library's own or the caller's is large enough (workspace_blocks_available)
int ensure_workspace(const tfft_plan* p, size_t blocks = 1) {
  std::lock_guard<std::mutex> lock(p->ws_mutex);
  const size_t need = tfft_plan_workspace_bytes(p) * blocks;
  if (need == 0 || (p->ws && p->ws_bytes >= need)) return TFFT_OK;
  if (p->ws && !p->ws_owned) return fail(TFFT_ERR_WORKSPACE, "workspace handed to tfft_plan_set_workspace is too small");
  if (p->ws) (void)hipFree(p->ws);
  p->ws = nullptr;
  TFFT_HIP(hipMalloc(&p->ws, need));
  p->ws_bytes = need;
  p->ws_owned = true;
  return TFFT_OK;
}

int launch_chain(const tfft_plan* p, const void* in_re, const void* in_im, void* out_re, void* out_im,
                 hipStream_t s) {
  if (p->sub_col) {
    // TFFT_ORDER_TRANSPOSED: column pass in -> planar workspace, row pass workspace -> out (in place is fine: the input
    // has been read completely before the row pass writes)
    _Float16* w = nullptr;
    if (!g_prepare) {
      const int rc = ensure_workspace(p);
      if (rc) return rc;
      w = static_cast<_Float16*>(p->ws);
    }
    _Float16* const w_im = w + p->chunk * p->n;
    const _Float16 *i_re = static_cast<const _Float16*>(in_re), *i_im = static_cast<const _Float16*>(in_im);
    _Float16 *o_re = static_cast<_Float16*>(out_re), *o_im = static_cast<_Float16*>(out_im);
    for (uint64_t b0 = 0; b0 < p->batch; b0 += p->chunk) {
      const bool tail = p->batch - b0 < p->chunk;
      const tfft_plan* const col = tail ? p->sub_col_tail : p->sub_col;
      const tfft_plan* const row = tail ? p->sub_row_tail : p->sub_row;
      const uint64_t io = b0 * p->in_stride, oo = b0 * p->out_stride;
      int rc;
      if (p->rows_first) {
        // transposed-order INPUT: N1 contiguous N2-point transforms per [N1][N2] block (output twiddle w_N^(k1 q) in their
        // epilogue) into the workspace, then one radix-N1 column pass over k1 writes X[q + N2 p] in natural order
        rc = launch_chain(row, i_re + io, i_im + io, w, w_im, s);
        if (rc) return rc;
        rc = launch_chain(col, w, w_im, o_re + oo, o_im + oo, s);
      } else {
        rc = launch_chain(col, i_re + io, i_im + io, w, w_im, s);
        if (rc) return rc;
        rc = launch_chain(row, w, w_im, o_re + oo, o_im + oo, s);
      }
      if (rc) return rc;
      if (g_prepare && !p->sub_col_tail) break;      // (prepare mode: one walk per distinct sub-plan is enough)
    }
    return TFFT_OK;
  }
  int np = static_cast<int>(p->passes.size());
  if (kDebugBuild && ((p->variant >> 8) & 15)) np = std::min(np, (p->variant >> 8) & 15);   // debugging aid: run only the first passes
  if (single_kernel(p)) {
    const PassKind kind = p->passes[0].kind;
    const int rc = kind == PassKind::K4096
                       ? launch_k4096(p, in_re, in_im, out_re, out_im, p->in_map, p->out_map, s)
                   : kind == PassKind::K4096R
                       ? launch_k4096r(p, p->passes[0].radix, in_re, in_im, out_re, out_im, p->in_map, p->out_map, s)
                       : (kind == PassKind::K256
                              ? launch_k256(p, in_re, in_im, out_re, out_im, p->in_map, p->out_map, s)
                              : launch_k256r(p, p->passes[0].radix, in_re, in_im, out_re, out_im, p->in_map,
                                             p->out_map, s));
    if (rc) return rc;
    TFFT_HIP(hipGetLastError());
    return TFFT_OK;
  }
  const uint64_t nf = p->n * p->inner;
  Planes IN{const_cast<_Float16*>(static_cast<const _Float16*>(in_re)),
            const_cast<_Float16*>(static_cast<const _Float16*>(in_im)), p->in_stride};
  Planes OUT{static_cast<_Float16*>(out_re), static_cast<_Float16*>(out_im), p->out_stride};
  const bool in_place = (in_re == out_re) || (in_im == out_im);
  // Targets alternate OUT / SCR so that the last pass writes OUT. SCR is the input
  // block when the reference's "input is scratch" contract allows it and the chain
  // does not start by overwriting what it reads; otherwise the plan's workspace.
  const bool odd = (np % 2) == 1;
  const bool use_in_as_scratch = !p->preserve_input && !in_place && odd;
  Planes SCR = IN;
  Planes SRC = IN;
  // In place with an odd number (>= 3) of passes: the chain needs a third buffer, IN -> A -> B -> ... -> OUT (= IN). A second
  // workspace block behind the first when the workspace is the library's own (or a caller's of twice tfft_plan_workspace_bytes);
  // otherwise, and for a single pass, the chain starts from a copy of the input (one more launch: the reference's own
  // single-transform benchmark runs 2^18 and 2^21 in place, results_in_results_ = false, and paid 3 / 11 us for that copy).
  Planes SCR_B{};
  bool two_blocks = false;
  if (in_place && odd && np >= 3 && !g_prepare) {
    bool can;
    {
      std::lock_guard<std::mutex> lock(p->ws_mutex);
      can = !p->ws || p->ws_owned || p->ws_bytes >= 2 * tfft_plan_workspace_bytes(p);
    }
    two_blocks = can;
  }
  if (!use_in_as_scratch && (np > 1 || in_place) && !g_prepare) {
    const int rc = ensure_workspace(p, two_blocks ? 2 : 1);
    if (rc) return rc;
    _Float16* w = static_cast<_Float16*>(p->ws);
    SCR = Planes{w, w + nf, 2 * nf};
    if (two_blocks) {
      _Float16* w2 = w + p->batch * 2 * nf;
      SCR_B = Planes{w2, w2 + nf, 2 * nf};
    } else if (in_place && odd) {
      // chain IN -> OUT would read and write the same block: start from a copy.
      if (p->in_stride != 2 * nf || static_cast<const _Float16*>(in_im) != static_cast<const _Float16*>(in_re) + nf)
        return fail(TFFT_ERR_ARG, "in-place execution of this length needs the [RE|IM] block layout (batch stride 2N)");
      const uint64_t n32 = p->batch * nf;          // 4 bytes per complex sample
      hipLaunchKernelGGL(stockham::copy_kernel, dim3(static_cast<uint32_t>(std::min<uint64_t>((n32 + 255) / 256, 8192))),
                         dim3(stockham::kBlock), 0, s, static_cast<const uint32_t*>(in_re), static_cast<uint32_t*>(p->ws), n32);
      SRC = SCR;
    }
  }
  Planes cur = SRC;
  for (int i = 0; i < np; ++i) {
    const bool to_out = ((np - 1 - i) % 2) == 0;
    const Planes dst = two_blocks ? (i + 1 == np ? OUT : ((i % 2) ? SCR_B : SCR)) : (to_out ? OUT : SCR);
    const Pass& ps = p->passes[i];
    if (ps.kind == PassKind::Col256) {
      const int rc = launch_col(p, ps, cur, dst, s);
      if (rc) return rc;
    } else {
      launch_stockham_pass(p, ps, cur, dst, s);
    }
    cur = dst;
  }
  if (!g_prepare) TFFT_HIP(hipGetLastError());
  return TFFT_OK;
}

// Runs the launch logic of a plan without launching: every kernel it can select gets its LDS opt-in now, so that a
// failure surfaces from tfft_plan_create and tfft_exec makes no runtime call besides the launches.
int prepare_kernels(const tfft_plan* p) {
  uint8_t* const fake = reinterpret_cast<uint8_t*>(uintptr_t{1} << 20);     // never dereferenced
  const uint64_t span = 4 * (p->batch * std::max(p->in_stride, p->out_stride) + p->n * p->inner);
  g_prepare = true;
  const int rc = launch_chain(p, fake, fake + span, fake + 2 * span, fake + 3 * span, nullptr);
  g_prepare = false;
  return rc;
}

// Element-exact test whether two planes (batch blocks of nf halves, `stride` halves apart) share a half.
bool planes_overlap(const void* pa, uint64_t sa, const void* pb, uint64_t sb, uint64_t batch, uint64_t nf) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(pa), b = reinterpret_cast<uintptr_t>(pb);
  const uintptr_t a_end = a + 2 * ((batch - 1) * sa + nf), b_end = b + 2 * ((batch - 1) * sb + nf);
  if (a_end <= b || b_end <= a) return false;
  if (batch == 1 || sa != sb) return true;            // different strides: conservative
  const uint64_t d = static_cast<uint64_t>(a > b ? a - b : b - a) / 2 % sa;     // halves, modulo the common stride
  return d < nf || sa - d < nf;
}

int check_variant(uint64_t n, uint64_t inner, int variant) {
  if (variant < 0 || (variant & ~(kVarTuner | kVarDebug)))
    return fail(TFFT_ERR_ARG, "unknown bits in tfft_plan_opts.variant (" + std::to_string(variant) + ")");
  if ((variant & kVarDebug) && !debug_variants_enabled())
    return fail(TFFT_ERR_ARG, "tfft_plan_opts.variant " + std::to_string(variant) +
                                  " holds a timing / debugging bit that produces WRONG or partial results "
                                  "(4, 64, 128, 65536, p << 8)" +
                                  (kDebugBuild ? std::string("; set TFFT_DEBUG_VARIANTS=1 to allow it")
                                               : std::string("; this library holds no such kernels (they exist only in a "
                                                             "-DTFFT_DEBUG_KERNELS build, libtfft_debug.so, with TFFT_DEBUG_VARIANTS=1)")));
  if (n == 4096 && inner <= 1 && !(variant & 32)) {
    const int v = variant & 15;
    if ((variant & 16) && v) return fail(TFFT_ERR_ARG, "variant bit 16 (plain N = 4096 kernel) excludes bits 1, 2, 8");
    if ((v & 1) && (v & 2)) return fail(TFFT_ERR_ARG, "variant bits 1 (prefetch) and 2 (staged stores) of the N = 4096 kernel exclude each other");
  }
  return TFFT_OK;
}

}  // namespace

extern "C" {

const char* tfft_last_error(void) { return g_err.c_str(); }
const char* tfft_version(void) { return "tfft 0.5 (gfx950, ABI 2)"; }
int tfft_abi_version(void) { return TFFT_ABI_VERSION; }

int tfft_ref_create_plan(uint64_t n, int mode, int base_wpb, int r16_wpb, int r2_bs, tfft_ref_plan* out) {
  g_err.clear();
  if (!out) return fail(TFFT_ERR_ARG, "null plan pointer");
  if (!is_pow2(n)) return fail(TFFT_ERR_NOT_POW2, "Error! Input size has to be a power of 2!");
  const int lg = ilog2(n);
  if (lg < 8) return fail(TFFT_ERR_TOO_SMALL, "Error! Input size has to be larger than 256 i.e. 16^2");
  if (mode != TFFT_MODE_256 && mode != TFFT_MODE_4096) return fail(TFFT_ERR_MODE, "unknown base FFT mode");
  if (mode == TFFT_MODE_4096 && n < 4096)
    return fail(TFFT_ERR_MODE, "Error! Baselayer fft length cant be longer that fft_length.");
  if (base_wpb <= 0 || r16_wpb <= 0 || r2_bs <= 0) return fail(TFFT_ERR_ARG, "non-positive launch parameter");
  tfft_ref_plan pl;
  std::memset(&pl, 0, sizeof(pl));
  pl.fft_length = n;
  pl.amount_of_r16_steps = lg / 4 - 1;
  pl.amount_of_r2_steps = lg % 4;
  pl.base_fft_mode = mode;
  const int after_base = pl.amount_of_r16_steps + pl.amount_of_r2_steps - (mode == TFFT_MODE_256 ? 1 : 2);
  pl.results_in_results = (after_base % 2 == 0) ? 1 : 0;
  const uint64_t warps = n / 256;
  std::string warn;
  if (warps < static_cast<uint64_t>(base_wpb)) {
    pl.base_fft_warps_per_block = static_cast<int>(warps);
    warn += "Warning! base_fft_warps_per_block overwritten to total_amount_of_warps. ";
  } else {
    if (warps % base_wpb) return fail(TFFT_ERR_GEOMETRY, "Error! Total amount of warps (fft_length/256) has to be evenly devisable by base_fft_warps_per_block.");
    if (mode == TFFT_MODE_4096) {
      if (base_wpb != 16) warn += "Warning! base_fft_warps_per_block overwritten to 16 (mandatory for mode=4096). ";
      pl.base_fft_warps_per_block = 16;
    } else {
      pl.base_fft_warps_per_block = base_wpb;
    }
  }
  pl.base_fft_blocksize = pl.base_fft_warps_per_block * 32;
  pl.base_fft_gridsize = static_cast<int>(warps / pl.base_fft_warps_per_block);
  pl.base_fft_shared_mem_in_bytes = pl.base_fft_warps_per_block * 1024 * 2;
  if (warps < static_cast<uint64_t>(r16_wpb)) {
    pl.r16_warps_per_block = static_cast<int>(warps);
    warn += "Warning! r16_warps_per_block overwritten to total_amount_of_warps. ";
  } else {
    if (warps % r16_wpb) return fail(TFFT_ERR_GEOMETRY, "Error! Total amount of warps (fft_length/256) has to be evenly devisable by amount_of_r16_warps_per_block.");
    pl.r16_warps_per_block = r16_wpb;
  }
  pl.r16_blocksize = pl.r16_warps_per_block * 32;
  pl.r16_gridsize = static_cast<int>(warps / pl.r16_warps_per_block);
  pl.r16_shared_mem_in_bytes = pl.r16_warps_per_block * 768 * 2;
  const uint64_t smallest_r2 = n >> pl.amount_of_r2_steps;
  if (smallest_r2 % r2_bs) return fail(TFFT_ERR_GEOMETRY, "Error! smallest_r2_subfft_length has to be evenly devisable by r2_blocksize.");
  pl.r2_blocksize = r2_bs;
  *out = pl;
  g_err = warn;
  return TFFT_OK;
}

int tfft_device_check(int device_id) {
  g_err.clear();
  hipDeviceProp_t prop;
  TFFT_HIP(hipGetDeviceProperties(&prop, device_id));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(TFFT_ERR_DEVICE, std::string("Error! gfx950 (MI355X) is required, found ") + prop.gcnArchName);
  if (prop.warpSize != 64) return fail(TFFT_ERR_DEVICE, "Error! Wavefront size of 64 required.");
  if (prop.maxThreadsPerBlock < k4096::kThreads) return fail(TFFT_ERR_DEVICE, "Error! Kernel exceeds max threads per block.");
  int lds_optin = 0;
  TFFT_HIP(hipDeviceGetAttribute(&lds_optin, hipDeviceAttributeMaxSharedMemoryPerBlock, device_id));
  if (lds_optin < k4096::kLdsBytes) return fail(TFFT_ERR_DEVICE, "Error! Kernel exceeds max shared memory per block (160 KiB LDS needed).");
  return TFFT_OK;
}

int tfft_max_no_optin_shared_mem(int device_id) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return -1;
  return static_cast<int>(prop.sharedMemPerBlock);
}

}  // extern "C" (reopened below)

namespace {

// tfft_plan_opts.scale -> constant operands and per-pass factors (include/tfft.h, TFFT_SCALE_*). Sequential: 1/16 per
// MFMA stage in F / G / H, 1/R per autosort pass, 1/2 and 1/4 in the G_q of the radix-512 / radix-1024 passes. None: all
// of them 1. Once: as none, and
// the single factor 2^-once_log2 rides on the LAST fp32 multiply of the plan: the inter-stage twiddle block of a
// single-kernel plan, the butterfly output of a final autosort pass, the combine of a final radix-512 / radix-1024 pass, the
// twiddles the pass in front of a final radix-256 column pass applies (or that pass's own four-step twiddle), and for
// a lone radix-256 column pass, which has no fp32 multiply at all, the stage-2 matrix G (2^-8 keeps G's entries normal).
int apply_scale_mode(tfft_plan* p, const InternalOpts& io, k4096::TableScale& ts, double& r_fa, double& r_fb, double& r_s) {
  const bool seq = p->scale_mode == TFFT_SCALE_SEQUENTIAL;
  const double s16 = seq ? 1.0 / 16 : 1.0;
  const int once_log2 = io.once_log2 >= 0 ? io.once_log2 : ilog2(p->n);
  const double fin = p->scale_mode == TFFT_SCALE_ONCE ? std::ldexp(1.0, -once_log2) : 1.0;
  ts = k4096::TableScale{s16, s16, s16, 1.0, seq ? s16 / 4 : s16, seq ? s16 / 2 : s16};
  r_fa = r_fb = s16;
  r_s = 1.0;
  if (io.rows2d) ts.tw = 2.0;                  // fused 2D row pass: the front end's headroom factor (k4096r.hpp)
  if (single_kernel(p)) {
    const Pass& ps = p->passes[0];
    if (ps.kind == PassKind::K4096 || ps.kind == PassKind::K256) ts.tw *= fin;
    if (ps.kind == PassKind::K4096R) ts.tw = (seq ? 2.0 : 2.0 * ps.radix) * fin;   // the front end keeps its 1 / (2 R)
    if (ps.kind == PassKind::K256R) r_s = (seq ? 1.0 / ps.radix : 1.0) * fin;
    return TFFT_OK;
  }
  for (Pass& ps : p->passes) {
    ps.tw_scale = 1.0f;
    if (ps.kind == PassKind::Stockham) ps.scale = seq ? 1.0f / ps.radix : 1.0f;
    else ps.scale = 1.0f;                      // column passes: the radix-512 / radix-1024 combines have their 1/2, 1/4 in the
                                               // constant operands (TableScale::g512, g1024); read-out factor of "scale once" only
  }
  if (fin != 1.0) {
    Pass& last = p->passes.back();
    const float f = static_cast<float>(fin);
    // (a four-step pass first: its epilogue multiplies by tw_scale in every form, whereas the radix-512 four-step read-out is
    // not the SC instantiation and would drop a factor put on comb_scale)
    if (p->tw4_modulus) last.tw_scale = f;
    else if (last.kind == PassKind::Stockham || last.radix == 512 || last.radix == 1024) last.scale *= f;
    else if (p->passes.size() >= 2) p->passes[p->passes.size() - 2].tw_scale = f;
    else if (once_log2 <= 8) ts.g *= fin;
    else return fail(TFFT_ERR_ARG, "TFFT_SCALE_ONCE: this plan has no fp32 multiply to carry the scaling step");
  }
  return TFFT_OK;
}

int create_plan(uint64_t n, uint64_t batch, int device_id, const tfft_plan_opts* opts, const InternalOpts& io, tfft_plan** out);

// Cache policy of a multi-pass plan's column passes from its footprint F (input + output + workspace = 3 x 4 B per sample): what
// fits the Infinity Cache (256 MiB) or comes close should stay in it - plain accesses: a pass then finds part of what the pass
// before it (or the caller) wrote still in the cache - what is much larger must stream (non-temporal: a line that is evicted
// before it is used again only displaces one that would have been hit). Measured with tools/scan_cache_policy.py, both on one data
// set again and again and on a ring of buffers of > 1 GiB (profiles/r4_cache_policy.txt), plain against streaming accesses:
//   two passes of radix 512 / 1024 (2^18..2^20): -3..-24 % for every F <= 384 MiB, +3..+9 % at 768 MiB
//   three passes (2^21..2^25): -2..-19 % for 192 and 384 MiB, +12..+19 % at 48 MiB, +-3 % at 96 MiB, +3..+10 % from 768 MiB on
//   radix-256 passes (2^16, 2^17): 0..+34 % (never better on the ring of buffers): they keep streaming
//   column pass of a transposed-output plan: -1..-15 % on one data set, +-3 % on the ring, for 192 and 384 MiB; worse outside
// Strided-axis plans (inner > 1: the 2D plan's and the distributed transform's column passes, transposed-input plans) keep the
// streaming policy they were tuned with; the transposed-output plan decides for its column sub-plan with footprint_policy().
// ---- plan "wisdom": tuner results loaded through the C ABI (tfft_tuning_load / tfft_tuning_add), consulted by tfft_plan_create
// for a natural-order, contiguous-axis plan whose caller left variant AND launch_iters at 0. The file is the reference's tuner
// file (N mode base_wpb r16_wpb r2_blocksize, Plan.h:197-255; written by FileWriter.h:250-269) with the three columns this
// library's tuners append: variant, launch_iters, batch. Choices that one or two boxes' measurements put inside the run-to-run
// spread live THERE, not in this source. A line applies to the batch nearest to its own on a log scale, up
// to a factor of 8 away (the tuners write lines at batch 1, 64, 4096 and the largest that fits); batch 0 = any batch.
struct WisdomLine {
  uint64_t n, batch;
  int variant;
  uint32_t launch_iters;
};
std::mutex g_wisdom_mutex;
std::vector<WisdomLine> g_wisdom;
bool wisdom_lookup(uint64_t n, uint64_t batch, int* variant, uint32_t* launch_iters) {
  std::lock_guard<std::mutex> lock(g_wisdom_mutex);
  const WisdomLine* best = nullptr;
  double best_d = 0;
  for (const WisdomLine& w : g_wisdom) {
    if (w.n != n) continue;
    const double d = w.batch ? std::fabs(std::log2(static_cast<double>(batch)) - std::log2(static_cast<double>(w.batch))) : 3.0;
    if (d > 3.0) continue;
    if (!best || d < best_d) {
      best = &w;
      best_d = d;
    }
  }
  if (!best) return false;
  *variant = best->variant;
  *launch_iters = best->launch_iters;
  return true;
}

constexpr uint64_t kMiB = 1ull << 20;
inline bool footprint_policy(uint64_t samples, uint64_t lo_mib) {
  const uint64_t f = samples * 12;
  return f >= lo_mib * kMiB && f <= 512 * kMiB;
}
// The default splits (kColSplit) were measured at 2^30 samples per launch, where every pass has thousands of workgroups. A plan
// of a few transforms is another regime: the radix-1024 kernel and the two-round radix-512 kernel take 65536 samples per
// workgroup, so ONE 2^20-point transform is 16 workgroups on 256 CUs. Below the measured limits the planner bits that give more,
// smaller workgroups are added to a default (variant 0) plan (tools/scan_small_batch.py, profiles/r4_small_batch_scan.txt; times
// of executions back to back on one stream):
//   2^18: the single-round radix-512 kernel for the final pass (64-column tiles)    x 1: 26.4 -> 19.7 us, x 16: 32.5 -> 28.4 us
//         then, to 2^25 samples, 4-wave workgroups                                   x 32: 38.5 -> 33.3 us, x 64: 57.9 -> 52.1 us, x 128: 116.6 -> 111.7 us
//   2^17: 4-wave workgroups from 2^23 to 2^25 samples                                x 64: 30.2 -> 27.9 us, x 256: 114.8 -> 111.3 us
//   2^19: 256 x 256 x 8 instead of 512 x 1024, to 2^22 samples                       x 1: 27.4 -> 22.0 us, x 4: 31.6 -> 29.2 us, x 8: 35.2 -> 32.9 us
//   2^20: 256 x 256 x 16 instead of 1024 x 1024                                      x 1: 40.1 -> 24.1 us, x 4: 48.8 -> 38.1 us
//   2^21: 256 x 256 x 32 instead of 512 x 512 x 8, to 2^22 samples                   x 1: 35.1 -> 30.4 us, x 2: 35.8 -> 32.9 us
// Round 5: the two rules of round 4 that sat inside the recorded run-to-run spread (profiles/r4_buffer_offsets.txt: 2-5 % on fixed
// addresses, +-5 % box to box) - 2^24 x 2 (6 %, with x 1 and x 4 going the other way) and 2^25 x 1 (9 % on one box) - are no longer
// rules of this source: they are lines of profiles/r5_TunerResults.dat, which a caller loads with tfft_tuning_load.
// Only variant 0 is touched: a caller (or tuner file) that names any bit gets exactly what it names.
// Round 5: with the latency column kernel (collat.hpp) a small transform of 2^17 ... 2^21 points is fastest as 256 x 256 x R, two
// latency passes and a radix-R tail (profiles/r5_small_scan.txt, device time per execution):
//   2^17 x 1: 15.8 -> 11.7 us, x 8: 19.0 -> 18.0 us (x 16: 20.4 with 512 x 256 on the throughput kernels against 25.5)
//   2^18 x 1: 19.7 -> 12.8 us, x 4: 21.7 -> 18.2 us (x 8: 24.3 with 512 x 512 against 25.9)
//   2^19 x 1: 27.5 -> 14.7 us, 2^20 x 1: 39.9 -> 18.4 us, x 4: 47.3 -> 31.5 us, 2^21 x 1: 35.1 -> 26.6 us (as in round 4, now on the new kernel)
// and the rules of round 4 that this scan puts inside the spread are gone (4-wave workgroups for 2^17 / 2^18 at 2^23 ... 2^25
// samples: 35.6 / 36.1 us, 34.3 / 34.2 us, 52.8 / 52.3 us; 2^20 x 8: 51.7 / 50.8 us).
inline int small_work_variant(uint64_t n, uint64_t inner, uint64_t batch) {
  if (inner != 1 || !is_pow2(n) || batch == 0 || batch > (1ull << 30)) return 0;
  const int lg = ilog2(n);
  const uint64_t work = n * batch;
  constexpr int kSplit256 = 8388608 | 33554432;          // no radix-512 / radix-1024 passes: 256 x 256 x R
  // 2^15 up to 8 transforms: 256 x 128 on the latency column kernel + the cooperative radix-128 pass instead of the single-pass
  // kernel, whose eight 4096-point sub-transforms share ONE CU (profiles/r5_small_scan.txt, last part: x 1: 12.0 -> 7.9 us, x 4:
  // 12.2 -> 8.8, x 8: 12.3 -> 10.3, x 16: 12.3 against 14.6)
  if (lg == 15) return work <= (1ull << 18) ? (kSplit256 | 16777216) : 0;
  // 2^14 up to 4 transforms: 256 x 64, the same two launches with the cooperative radix-64 pass (x 1: 8.4 -> 7.2 us, x 4: 8.7 -> 7.5,
  // x 8: 8.6 / 8.3, x 16: 8.8 against 9.9). 2^13 up to 4 transforms: 256 x 32 likewise (x 1: 7.6 -> 7.1 us, x 4: 7.8 -> 7.2, x 8: 7.8 /
  // 7.5, x 16: 7.8 against 8.2).
  if (lg == 14) return work <= (1ull << 16) ? (kSplit256 | 16777216) : 0;
  if (lg == 13) return work <= (1ull << 15) ? (kSplit256 | 16777216) : 0;
  if (lg < 17 || lg > 21) return 0;
  if (work <= (lg <= 18 ? (1ull << 20) : (1ull << 22))) return kSplit256;
  if (lg == 18 && work <= (1ull << 22)) return 268435456;  // 512 x 512 with the single-round radix-512 kernel last (round 4: x 16: 32.5 -> 28.4 us)
  return 0;
}

inline bool cache_policy(uint64_t n, uint64_t inner, uint64_t batch) {
  if (inner != 1 || n < (1ull << 18) || batch > (1ull << 40) / n) return false;
  return footprint_policy(n * batch, n <= (1ull << 20) ? 0 : 128);
}

// Transforms per chunk of a transposed-order plan (measured, profiles/r4_chunked_transposed.txt: 2^20 x 1024 in chunks of 256 /
// 128 / 64 / 32 transforms 368 / 375 / 374 / 349 Gsamples/s against 348 for the whole batch), never chunks so short that a launch
// has less than 2^26 samples. Transposed OUTPUT: 256 MiB of intermediate, the size of the Infinity Cache: boxes differ (second
// part of that file): on some 512 MiB is 1-2 % faster at every length, on others it loses the whole gain (2^20 x 1024: 344 with
// 512 MiB, 373 with 256 MiB, same process), so the size that holds on both is taken. Transposed INPUT: 512 MiB (256 MiB:
// -3 ... -5 % at 2^16 ... 2^22 on the box where it was tried, +2 % on the other) except 2^24 (4 transforms: 317 against 277).
inline uint64_t transposed_chunk(uint64_t n, uint64_t batch, bool transposed_in) {
  const int lg_samples = (transposed_in && n < (uint64_t{1} << 24)) ? 27 : 26;          // x 4 B = 512 / 256 MiB
  const uint64_t per = std::max<uint64_t>(1, (uint64_t{1} << lg_samples) / n);
  return std::min<uint64_t>(batch, per);
}

int create_transposed(tfft_plan* p, const tfft_plan_opts* opts, int device_id) {
  // N = N1 N2: column pass (n = N1 along the strided axis, N2 columns, four-step twiddle w_N^(k1 n2)) into the planar
  // workspace [RE: batch x N | IM: batch x N], then batch * N1 contiguous N2-point transforms from it into `out`, where
  // the N1 rows of one transform sit N2 apart inside the caller's block (grouped addressing).
  const uint64_t n = p->n, n2 = tfft_plan_transposed_n2(n), n1 = n / n2;
  if (transposed_chunk(n, p->batch, false) * n1 > 0xffffffffull) return fail(TFFT_ERR_ARG, "chunk * N1 too large for one launch");
  const int mode = p->scale_mode;
  tfft_plan_opts co = TFFT_PLAN_OPTS_INIT;
  co.in_batch_stride = p->in_stride;
  co.out_batch_stride = n;
  co.inner = n2;
  co.preserve_input = 1;
  // tuner bits: the column-pass bits go to the column sub-plan, the single-kernel bits of the N2 kernel to the row sub-plan;
  // everything else has no meaning for this plan shape and is refused instead of being dropped silently
  constexpr int kColBits = 262144 | 524288 | 536870912, kRowBits = kVarK4096 | 1048576;
  if (p->variant & ~(kColBits | kRowBits))
    return fail(TFFT_ERR_ARG, "tfft_plan_opts.variant " + std::to_string(p->variant) + ": a TFFT_ORDER_TRANSPOSED plan honours only the "
                              "column-pass bits 262144 / 524288 / 536870912 and the single-kernel bits 1 / 2 / 8 / 16 / 1048576");
  co.variant = (p->variant & kColBits) | (n1 == 512 ? 67108864 : 0);
  if (!(p->variant & (262144 | 536870912)) && footprint_policy(n * transposed_chunk(n, p->batch, false), 128)) co.variant |= 262144;
  co.scale = mode == TFFT_SCALE_SEQUENTIAL ? TFFT_SCALE_SEQUENTIAL : TFFT_SCALE_NONE;
  co.fourstep_n = n;
  co.launch_iters = p->launch_iters;
  p->chunk = transposed_chunk(n, p->batch, false);
  const uint64_t tail = p->batch % p->chunk;
  int rc = create_plan(n1, p->chunk, device_id, &co, InternalOpts{}, &p->sub_col);
  if (rc == TFFT_OK && tail) rc = create_plan(n1, tail, device_id, &co, InternalOpts{}, &p->sub_col_tail);
  if (rc) return rc;
  tfft_plan_opts ro = TFFT_PLAN_OPTS_INIT;
  ro.in_batch_stride = n2;
  ro.out_batch_stride = n2;
  ro.preserve_input = 1;
  ro.scale = mode;
  ro.launch_iters = p->launch_iters;
  ro.variant = p->variant & ((n2 == 4096 ? kVarK4096 : 0) | ((n2 == 512 || n2 == 1024 || n2 == 2048) ? 1048576 : 0));
  InternalOpts ri;
  ri.group_shift = static_cast<uint32_t>(ilog2(n1));
  ri.in_gstride = n;                       // planar workspace: row b at b * N2 either way
  ri.out_gstride = p->out_stride;
  ri.once_log2 = ilog2(n);
  rc = create_plan(n2, p->chunk * n1, device_id, &ro, ri, &p->sub_row);
  if (rc == TFFT_OK && tail) rc = create_plan(n2, tail * n1, device_id, &ro, ri, &p->sub_row_tail);
  return rc;
}

// tfft_plan_opts as the caller holds it -> the library's own (current) layout: exactly struct_size bytes are read, every
// field beyond them is 0. Sizes: include/tfft.h (struct_size).
constexpr size_t kOptsSizes[] = {48, 64, 72};
static_assert(sizeof(tfft_plan_opts) == 72 && offsetof(tfft_plan_opts, fourstep_n) == 48 && offsetof(tfft_plan_opts, launch_iters) == 64,
              "tfft_plan_opts layout changed: add the new size to kOptsSizes and to include/tfft.h");
inline bool opts_size_known(size_t bytes) {
  for (size_t k : kOptsSizes)
    if (k == bytes) return true;
  return false;
}
int normalise_opts(const tfft_plan_opts* opts, tfft_plan_opts* o) {
  std::memset(o, 0, sizeof(*o));
  o->struct_size = static_cast<uint32_t>(sizeof(*o));
  if (!opts) return TFFT_OK;
  uint32_t sz = 0;
  std::memcpy(&sz, opts, sizeof(sz));           // (only the first four bytes are known to exist)
  if (!opts_size_known(sz))
    return fail(TFFT_ERR_ARG, "tfft_plan_opts.struct_size = " + std::to_string(sz) + " is not the size of a layout this library knows (48, 64, " +
                                  std::to_string(sizeof(*o)) + "): initialise the options with TFFT_PLAN_OPTS_INIT or tfft_plan_opts_init()");
  std::memcpy(o, opts, sz);
  o->struct_size = static_cast<uint32_t>(sizeof(*o));
  if (o->reserved_) return fail(TFFT_ERR_ARG, "tfft_plan_opts.reserved_ must be 0");
  return TFFT_OK;
}

int create_transposed_in(tfft_plan* p, int device_id) {
  // N = N1 N2 as for the transposed OUTPUT order. The caller's block holds in[k1 N2 + k2] = x[k1 + N1 k2]: N1 rows, row k1 the
  // decimated sequence x[k1 + N1 .]. Decimation in time:  X[q + N2 p] = sum_k1 w_N1^(k1 p) [ w_N^(k1 q) DFT_N2(row k1)[q] ].
  // Pass 1: batch * N1 contiguous N2-point transforms (single-pass kernels, grouped addressing, the bracket's twiddle applied to
  // their fp32 accumulators) into the planar workspace [RE: batch x N | IM: batch x N]; pass 2: one plain radix-N1 column pass
  // along k1 (N2 columns) from it into `out`, whose row p, column q is X[q + N2 p]: natural order.
  const uint64_t n = p->n, n2 = tfft_plan_transposed_n2(n), n1 = n / n2;
  if (transposed_chunk(n, p->batch, true) * n1 > 0xffffffffull) return fail(TFFT_ERR_ARG, "chunk * N1 too large for one launch");
  if (p->scale_mode == TFFT_SCALE_ONCE)
    return fail(TFFT_ERR_ARG, "TFFT_SCALE_ONCE is not available with transposed-order input: the plan's last fp32 multiply lies in "
                              "front of its last stage (use TFFT_SCALE_SEQUENTIAL or TFFT_SCALE_NONE)");
  constexpr int kColBits = 262144 | 524288 | 536870912;
  if (p->variant & ~kColBits)
    return fail(TFFT_ERR_ARG, "tfft_plan_opts.variant " + std::to_string(p->variant) + ": a plan with transposed-order input honours only the "
                              "column-pass bits 262144 / 524288 / 536870912");
  tfft_plan_opts ro = TFFT_PLAN_OPTS_INIT;
  ro.in_batch_stride = n2;
  ro.out_batch_stride = n2;
  ro.preserve_input = 1;
  ro.scale = p->scale_mode;
  ro.launch_iters = p->launch_iters;
  InternalOpts ri;
  ri.group_shift = static_cast<uint32_t>(ilog2(n1));
  ri.in_gstride = p->in_stride;            // the rows of one transform sit N2 apart inside the caller's [RE | IM] block
  ri.out_gstride = n;                      // planar workspace: row b at b * N2 either way
  ri.otw_n = n;
  p->chunk = transposed_chunk(n, p->batch, true);
  const uint64_t tail = p->batch % p->chunk;
  int rc = create_plan(n2, p->chunk * n1, device_id, &ro, ri, &p->sub_row);
  if (rc == TFFT_OK && tail) rc = create_plan(n2, tail * n1, device_id, &ro, ri, &p->sub_row_tail);
  if (rc) return rc;
  tfft_plan_opts co = TFFT_PLAN_OPTS_INIT;
  co.in_batch_stride = n;
  co.out_batch_stride = p->out_stride;
  co.inner = n2;
  co.preserve_input = 1;
  co.scale = p->scale_mode;
  co.launch_iters = p->launch_iters;
  co.variant = (p->variant & kColBits) | (n1 == 512 ? 67108864 : 0);
  rc = create_plan(n1, p->chunk, device_id, &co, InternalOpts{}, &p->sub_col);
  if (rc == TFFT_OK && tail) rc = create_plan(n1, tail, device_id, &co, InternalOpts{}, &p->sub_col_tail);
  if (rc) return rc;
  if (p->sub_col->passes.size() != 1 || p->sub_col->passes[0].kind != PassKind::Col256)
    return fail(TFFT_ERR_ARG, "transposed-order input: the column transform of this length does not plan as one pass");
  p->rows_first = true;
  return TFFT_OK;
}

int create_plan(uint64_t n, uint64_t batch, int device_id, const tfft_plan_opts* caller_opts, const InternalOpts& io, tfft_plan** out) {
  g_err.clear();
  if (!out) return fail(TFFT_ERR_ARG, "null plan pointer");
  *out = nullptr;
  tfft_plan_opts norm;
  {
    const int rc0 = normalise_opts(caller_opts, &norm);
    if (rc0) return rc0;
  }
  const tfft_plan_opts* const opts = &norm;
  if (!is_pow2(n)) return fail(TFFT_ERR_NOT_POW2, "Error! Input size has to be a power of 2!");
  if (n < 2) return fail(TFFT_ERR_TOO_SMALL, "Error! Input size has to be at least 2");
  if (batch == 0 || batch > 0xffffffffull) return fail(TFFT_ERR_ARG, "batch must be in [1, 2^32)");
  const uint64_t inner = (opts && opts->inner) ? opts->inner : 1;
  if (!is_pow2(inner) || (inner > 1 && inner < 8)) return fail(TFFT_ERR_ARG, "inner (strided-axis batch) must be 1 or a power of two >= 8");
  const uint64_t nf = n * inner;
  const uint64_t in_stride = (opts && opts->in_batch_stride) ? opts->in_batch_stride : 2 * nf;
  const uint64_t out_stride = (opts && opts->out_batch_stride) ? opts->out_batch_stride : 2 * nf;
  if (nf >= 8 && ((in_stride % 8) || (out_stride % 8))) return fail(TFFT_ERR_ARG, "batch strides must be multiples of 8 halves (16 bytes)");
  if ((in_stride < nf && !io.in_seg_len) || out_stride < nf) return fail(TFFT_ERR_ARG, "batch stride smaller than the FFT length");
  const int scale_mode = opts ? opts->scale : 0;
  if (scale_mode < TFFT_SCALE_SEQUENTIAL || scale_mode > TFFT_SCALE_ONCE) return fail(TFFT_ERR_ARG, "unknown tfft_plan_opts.scale");
  const int order = opts ? opts->output_order : 0;
  if (order != TFFT_ORDER_NATURAL && order != TFFT_ORDER_TRANSPOSED) return fail(TFFT_ERR_ARG, "unknown tfft_plan_opts.output_order");
  if (order == TFFT_ORDER_TRANSPOSED && inner > 1)
    return fail(TFFT_ERR_ARG, "TFFT_ORDER_TRANSPOSED exists for a contiguous axis only (inner <= 1)");
  const int in_order = opts->input_order;
  if (in_order != TFFT_ORDER_NATURAL && in_order != TFFT_ORDER_TRANSPOSED) return fail(TFFT_ERR_ARG, "unknown tfft_plan_opts.input_order");
  if (in_order == TFFT_ORDER_TRANSPOSED) {
    if (inner > 1) return fail(TFFT_ERR_ARG, "TFFT_ORDER_TRANSPOSED exists for a contiguous axis only (inner <= 1)");
    if (order == TFFT_ORDER_TRANSPOSED)
      return fail(TFFT_ERR_ARG, "input_order and output_order cannot both be TFFT_ORDER_TRANSPOSED (one side of a plan is in natural order)");
    if (!tfft_plan_transposed_n2(n))
      return fail(TFFT_ERR_ARG, "input_order = TFFT_ORDER_TRANSPOSED: this length has no [N1][N2] layout (2^16 <= N <= 2^24, tfft_plan_transposed_n2)");
    if (opts->fourstep_n) return fail(TFFT_ERR_ARG, "fourstep_n and TFFT_ORDER_TRANSPOSED exclude each other");
  }
  const uint64_t tw4 = opts ? opts->fourstep_n : 0;
  if (tw4) {
    if (!is_pow2(tw4) || tw4 < n || (n != 256 && n != 512) || inner < 64)
      return fail(TFFT_ERR_ARG, "fourstep_n: the four-step twiddle exists for n = 256 or 512 along a strided axis of inner >= 64 columns, "
                                "with fourstep_n a power of two >= n");
    if (order == TFFT_ORDER_TRANSPOSED) return fail(TFFT_ERR_ARG, "fourstep_n and TFFT_ORDER_TRANSPOSED exclude each other");
  }
  int pvariant = opts ? opts->variant : 0;
  if (tw4 && n == 512) pvariant |= 67108864;    // one radix-512 pass
  // the default plan of a caller-facing, natural-order transform that does not fill the chip: the split with more workgroups
  // (caller-facing: not a sub-plan with grouped / segmented addressing or a fused epilogue: those keep what they were tuned with)
#ifdef TFFT_SUBPLAN_POLICY   // A/B knob: the footprint cache policy for sub-plans too
  constexpr bool kSubplanPolicy = true;
#else
  constexpr bool kSubplanPolicy = false;
#endif
  const bool caller_facing = io.group_shift == 0 && !io.rows2d && io.in_seg_len == 0 && io.otw_n == 0;
  uint32_t launch_iters = opts->launch_iters;
  // (a TFFT_ORDER_TRANSPOSED request for a length without an [N1][N2] layout IS the natural-order plan: same defaults)
  const bool natural_out = order == TFFT_ORDER_NATURAL || !tfft_plan_transposed_n2(n);
  const bool plannable = !tw4 && natural_out && in_order == TFFT_ORDER_NATURAL && caller_facing;
  if (pvariant == 0 && launch_iters == 0 && plannable && inner == 1) {
    int wv = 0;
    uint32_t wi = 0;
    if (wisdom_lookup(n, batch, &wv, &wi)) {      // a loaded tuner line for this (N, batch); variant 0 in it = the library's default
      pvariant = wv;
      launch_iters = wi;
    }
  }
  if (pvariant == 0 && plannable) pvariant = small_work_variant(n, inner, batch);
  int rc = check_variant(n, inner, pvariant);
  if (rc) return rc;
  if (tw4 && (pvariant & (32 | 131072 | 4096 | 8192)))
    return fail(TFFT_ERR_ARG, "fourstep_n needs the workgroup-cooperative column kernels (variant bits 32, 4096, 8192, 131072 exclude it)");
  rc = tfft_device_check(device_id);
  if (rc) return rc;
  int prev = 0;
  TFFT_HIP(hipGetDevice(&prev));
  TFFT_HIP(hipSetDevice(device_id));
  tfft_plan* p = new tfft_plan;
  p->n = n;
  p->batch = batch;
  p->inner = inner;
  p->device = device_id;
  p->in_stride = in_stride;
  p->out_stride = out_stride;
  p->preserve_input = opts && opts->preserve_input;
  p->variant = pvariant;
  p->launch_iters = launch_iters;
  p->scale_mode = scale_mode;
  // cache policy of the column passes: variant bit 262144 = plain accesses, 536870912 = non-temporal (streaming) accesses,
  // neither = by the plan's footprint (cache_policy)
  // (also for the row transforms of a distributed transform, the segmented-input sub-plan: 2^26 over 8 ranks, 2^18 x 32 rows per
  // rank, local work 65.4 -> 59.6 us on one box, profiles/r4_dist_local_work.txt)
  p->plain_acc = (pvariant & 262144) ? true : ((pvariant & 536870912) ? false : ((caller_facing || io.in_seg_len != 0 || kSubplanPolicy) && cache_policy(n, inner, batch)));
  p->tw4_modulus = tw4;
  p->tw4_col0 = opts ? opts->fourstep_col0 : 0;
  p->in_map = k4096::Addr{in_stride, io.group_shift ? io.in_gstride : in_stride, io.group_shift, (1u << io.group_shift) - 1u};
  p->out_map = k4096::Addr{out_stride, io.group_shift ? io.out_gstride : out_stride, io.group_shift, (1u << io.group_shift) - 1u};
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device_id);
  p->num_cus = (e == hipSuccess) ? prop.multiProcessorCount : 256;
#ifdef TFFT_DEBUG_KERNELS
  // experiment knob: TFFT_NUM_CUS=192 sizes every persistent grid for that many CUs (fewer workgroups than the chip has CUs:
  // does the memory system serve fewer concurrent streams better?)
  if (const uint32_t forced = env_iters("TFFT_NUM_CUS", 0)) p->num_cus = static_cast<int>(forced);
#endif
  auto bail = [&](int code) {
    const std::string keep = g_err;
    tfft_plan_destroy(p);
    (void)hipSetDevice(prev);
    g_err = keep;
    return code;
  };
  if (order == TFFT_ORDER_TRANSPOSED && tfft_plan_transposed_n2(n)) {
    rc = create_transposed(p, opts, device_id);
    if (rc) return bail(rc);
    (void)hipSetDevice(prev);
    *out = p;
    return TFFT_OK;
  }
  if (in_order == TFFT_ORDER_TRANSPOSED) {
    rc = create_transposed_in(p, device_id);
    if (rc) return bail(rc);
    (void)hipSetDevice(prev);
    *out = p;
    return TFFT_OK;
  }
  // ---- pass list (plan_passes: pure host logic, also behind tfft_plan_describe)
  plan_passes(n, inner, pvariant, p->passes);
  if (io.group_shift && !single_kernel(p)) return bail(fail(TFFT_ERR_ARG, "grouped addressing needs a single-kernel plan"));
  if (io.otw_n) {
    if (!single_kernel(p) || (pvariant & ~0) != 0 || io.otw_n > (uint64_t{1} << 24))
      return bail(fail(TFFT_ERR_ARG, "output twiddle: needs a single-kernel row plan with the default variant and N <= 2^24"));
    p->otw.n_mask = static_cast<uint32_t>(io.otw_n - 1);
    p->otw.row_mask = (1u << io.group_shift) - 1u;
    p->otw.inv_n = 1.0f / static_cast<float>(io.otw_n);
  }
  if (tw4 && !(p->passes.size() == 1 && p->passes[0].kind == PassKind::Col256))
    return bail(fail(TFFT_ERR_ARG, "fourstep_n: this (n, inner) does not plan as one column pass"));
  if (io.in_seg_len) {
    // the first pass must read rows of `pitch` halves through the cooperative radix-256 / radix-512 kernels, whole rows per
    // segment; everything behind it works on the plan's own buffers
    const Pass& f = p->passes[0];
    const uint64_t pitch = f.kind == PassKind::Col256 ? n / static_cast<uint64_t>(f.radix) : 0;
    const bool ok = inner == 1 && p->passes.size() >= 2 && p->preserve_input && f.kind == PassKind::Col256 &&
                    (f.radix == 256 || f.radix == 512) && pitch % 128 == 0 && !(pvariant & (131072 | 4096 | 8192)) &&
                    is_pow2(io.in_seg_len) && io.in_seg_len >= pitch && io.in_seg_len < n && io.in_seg_stride >= io.in_seg_len;
    if (!ok) return bail(fail(TFFT_ERR_ARG, "segmented input: this length does not start with a cooperative radix-256 / radix-512 column pass "
                                            "whose rows fit the segments"));
    p->in_seg_shift = static_cast<uint32_t>(ilog2(io.in_seg_len / pitch));
    p->in_seg_gap = io.in_seg_stride - io.in_seg_len;
  }
  k4096::TableScale ts;
  double r_fa, r_fb, r_s;
  rc = apply_scale_mode(p, io, ts, r_fa, r_fb, r_s);
  if (rc) return bail(rc);
  bool need_tables = false;
  for (const Pass& ps : p->passes)
    need_tables = need_tables || ps.kind == PassKind::K4096 || ps.kind == PassKind::K256 || ps.kind == PassKind::K4096R ||
                  ps.kind == PassKind::Col256;
  if (p->passes.size() == 1 && p->passes[0].kind == PassKind::K256R) {
    std::vector<uint8_t> blob;
    k256r::build_tables(p->passes[0].radix, blob, r_fa, r_fb, r_s);
    e = hipMalloc(&p->d_tables, blob.size());
    if (e != hipSuccess) return bail(hip_fail(e, "hipMalloc(tables)"));
    e = hipMemcpy(p->d_tables, blob.data(), blob.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) return bail(hip_fail(e, "hipMemcpy(tables)"));
  }
  if (need_tables) {
    std::vector<uint8_t> blob;
    k4096::build_tables(blob, ts);
    e = hipMalloc(&p->d_tables, blob.size());
    if (e != hipSuccess) return bail(hip_fail(e, "hipMalloc(tables)"));
    e = hipMemcpy(p->d_tables, blob.data(), blob.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) return bail(hip_fail(e, "hipMemcpy(tables)"));
  }
  if (!single_kernel(p)) {
    const uint64_t tn = tw4 ? tw4 : n;          // modulus of the w tables
    const uint64_t lo_n = std::min<uint64_t>(tn, stockham::kTwLoSize);
    const uint64_t hi_n = tn > stockham::kTwLoSize ? tn / stockham::kTwLoSize : 1;   // entry 0 = 1 + 0 i always exists
    std::vector<float2> lo(lo_n), hi(hi_n);
    for (uint64_t t = 0; t < lo_n; ++t) {
      const double a = -2.0 * M_PI * static_cast<double>(t) / static_cast<double>(tn);
      lo[t] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
    }
    for (uint64_t t = 0; t < hi_n; ++t) {
      const double a = -2.0 * M_PI * static_cast<double>(t) * stockham::kTwLoSize / static_cast<double>(tn);
      hi[t] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
    }
    e = hipMalloc(reinterpret_cast<void**>(&p->d_tw_lo), lo_n * sizeof(float2));
    if (e != hipSuccess) return bail(hip_fail(e, "hipMalloc(tw_lo)"));
    e = hipMemcpy(p->d_tw_lo, lo.data(), lo_n * sizeof(float2), hipMemcpyHostToDevice);
    if (e != hipSuccess) return bail(hip_fail(e, "hipMemcpy(tw_lo)"));
    if (hi_n) {
      e = hipMalloc(reinterpret_cast<void**>(&p->d_tw_hi), hi_n * sizeof(float2));
      if (e != hipSuccess) return bail(hip_fail(e, "hipMalloc(tw_hi)"));
      e = hipMemcpy(p->d_tw_hi, hi.data(), hi_n * sizeof(float2), hipMemcpyHostToDevice);
      if (e != hipSuccess) return bail(hip_fail(e, "hipMemcpy(tw_hi)"));
    }
    const uint64_t max_blocks = ((nf / 2 + stockham::kBlock - 1) / stockham::kBlock) * batch;
    if (max_blocks > 0x7fffffffull) return bail(fail(TFFT_ERR_ARG, "batch * N too large for one launch"));
    if ((nf / 16) * batch > 0xffffffffull) return bail(fail(TFFT_ERR_ARG, "batch * N too large for one launch"));
  }
  rc = prepare_kernels(p);
  if (rc) return bail(rc);
  (void)hipSetDevice(prev);
  *out = p;
  return TFFT_OK;
}

}  // namespace

extern "C" {

int tfft_plan_create(uint64_t n, uint64_t batch, int device_id, const tfft_plan_opts* opts, tfft_plan** out) {
  return create_plan(n, batch, device_id, opts, InternalOpts{}, out);
}

int tfft_plan_opts_known_size(size_t bytes) { return opts_size_known(bytes) ? 1 : 0; }

int tfft_plan_opts_init(tfft_plan_opts* opts, size_t bytes) {
  g_err.clear();
  if (!opts) return fail(TFFT_ERR_ARG, "null options pointer");
  if (!opts_size_known(bytes)) return fail(TFFT_ERR_ARG, "tfft_plan_opts_init: " + std::to_string(bytes) + " bytes is not the size of a layout this library knows");
  std::memset(opts, 0, bytes);
  const uint32_t sz = static_cast<uint32_t>(bytes);
  std::memcpy(opts, &sz, sizeof(sz));
  return TFFT_OK;
}

uint64_t tfft_plan_transposed_n2(uint64_t n) {
  if (!is_pow2(n)) return 0;
  const int lg = ilog2(n);
  if (lg < 16 || lg > 24) return 0;
  // N1 = 256 (256-byte row segments in the column pass) unless N2 would then leave the single-kernel range or 4096 is
  // reachable with N1 = 512 (the N = 4096 kernel is the fastest second pass)
  const int lg2 = (lg == 21 || lg == 24) ? lg - 9 : lg - 8;
  return uint64_t{1} << lg2;
}

int tfft_plan_cache_policy(uint64_t n, uint64_t inner, uint64_t batch) {
  if (!is_pow2(n) || !inner || !batch) return 0;
  return cache_policy(n, inner, batch) ? 1 : 0;
}

int tfft_plan_default_variant(uint64_t n, uint64_t inner, uint64_t batch) {
  if (!is_pow2(n) || !inner || !batch) return 0;
  int wv = 0;
  uint32_t wi = 0;
  if (inner == 1 && wisdom_lookup(n, batch, &wv, &wi) && wv) return wv;
  return small_work_variant(n, inner, batch);
}

int tfft_tuning_add(uint64_t n, uint64_t batch, int variant, uint32_t launch_iters) {
  g_err.clear();
  if (!is_pow2(n) || n < 2) return fail(TFFT_ERR_NOT_POW2, "Error! Input size has to be a power of 2!");
  if (launch_iters > TFFT_LAUNCH_PERSISTENT) return fail(TFFT_ERR_ARG, "launch_iters must be 0 .. 65535");
  if (batch > 0xffffffffull) return fail(TFFT_ERR_ARG, "batch must be in [0, 2^32)");
  const int rc = check_variant(n, 1, variant);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(g_wisdom_mutex);
  for (WisdomLine& w : g_wisdom)
    if (w.n == n && w.batch == batch) {          // a later line for the same (N, batch) replaces the earlier one
      w.variant = variant;
      w.launch_iters = launch_iters;
      return TFFT_OK;
    }
  g_wisdom.push_back(WisdomLine{n, batch, variant, launch_iters});
  return TFFT_OK;
}

int tfft_tuning_load(const char* path, int* lines_taken) {
  g_err.clear();
  if (lines_taken) *lines_taken = 0;
  if (!path) return fail(TFFT_ERR_ARG, "null path");
  std::FILE* f = std::fopen(path, "r");
  if (!f) return fail(TFFT_ERR_ARG, std::string("Error! Failed to open tuner file: ") + path);
  // parse everything first, commit only a file without a bad line
  std::vector<WisdomLine> lines;
  char buf[512];
  int line_no = 0, rc = TFFT_OK;
  while (std::fgets(buf, sizeof(buf), f)) {
    ++line_no;
    double len = 0;
    long long mode = 0, bw = 0, rw = 0, r2 = 0, variant = 0, iters = 0, batch = 0;
    const int got = std::sscanf(buf, "%lf %lld %lld %lld %lld %lld %lld %lld", &len, &mode, &bw, &rw, &r2, &variant, &iters, &batch);
    if (got < 6) continue;                       // blank, comment, or a plain reference line (no variant column): nothing to learn
    const uint64_t n = static_cast<uint64_t>(len);
    if (static_cast<double>(n) != len || !is_pow2(n) || n < 2 || variant < 0 || variant > 0x7fffffffll || iters < 0 || iters > TFFT_LAUNCH_PERSISTENT ||
        batch < 0 || batch > 0xffffffffll || check_variant(n, 1, static_cast<int>(variant)) != TFFT_OK) {
      rc = fail(TFFT_ERR_ARG, std::string(path) + ":" + std::to_string(line_no) + ": unusable tuner line (length, variant, launch_iters or batch out of range): " +
                                  std::string(buf).substr(0, 80));
      break;
    }
    lines.push_back(WisdomLine{n, static_cast<uint64_t>(batch), static_cast<int>(variant), static_cast<uint32_t>(iters)});
  }
  std::fclose(f);
  if (rc) return rc;
  for (const WisdomLine& w : lines) {
    const int r = tfft_tuning_add(w.n, w.batch, w.variant, w.launch_iters);
    if (r) return r;
  }
  if (lines_taken) *lines_taken = static_cast<int>(lines.size());
  return TFFT_OK;
}

void tfft_tuning_clear(void) {
  std::lock_guard<std::mutex> lock(g_wisdom_mutex);
  g_wisdom.clear();
}

int tfft_tuning_query(uint64_t n, uint64_t batch, int* variant, uint32_t* launch_iters) {
  int v = 0;
  uint32_t it = 0;
  if (!batch || !wisdom_lookup(n, batch, &v, &it)) return 0;
  if (variant) *variant = v;
  if (launch_iters) *launch_iters = it;
  return 1;
}

int tfft_variant_check(uint64_t n, uint64_t inner, int variant) {
  g_err.clear();
  if (!is_pow2(n) || n < 2) return fail(TFFT_ERR_NOT_POW2, "Error! Input size has to be a power of 2!");
  return check_variant(n, inner ? inner : 1, variant);
}

void tfft_plan_destroy(tfft_plan* p) {
  if (!p) return;
  tfft_plan_destroy(p->sub_col);
  tfft_plan_destroy(p->sub_row);
  tfft_plan_destroy(p->sub_col_tail);
  tfft_plan_destroy(p->sub_row_tail);
  if (p->d_tables) (void)hipFree(p->d_tables);
  if (p->d_tw_lo) (void)hipFree(p->d_tw_lo);
  if (p->d_tw_hi) (void)hipFree(p->d_tw_hi);
  if (p->ws && p->ws_owned) (void)hipFree(p->ws);
  delete p;
}

int tfft_plan_describe(uint64_t n, uint64_t inner, int variant, char* buf, size_t bytes) {
  g_err.clear();
  if (!buf || bytes == 0) return fail(TFFT_ERR_ARG, "null buffer");
  if (!is_pow2(n) || n < 2) return fail(TFFT_ERR_NOT_POW2, "Error! Input size has to be a power of 2!");
  if (inner == 0) inner = 1;
  if (!is_pow2(inner) || (inner > 1 && inner < 8)) return fail(TFFT_ERR_ARG, "inner (strided-axis batch) must be 1 or a power of two >= 8");
  const int vrc = check_variant(n, inner, variant);
  if (vrc) return vrc;
  std::vector<Pass> passes;
  plan_passes(n, inner, variant, passes);
  std::string out;
  for (const Pass& ps : passes) {
    const char* k = ps.kind == PassKind::K4096 ? "k4096" : ps.kind == PassKind::K4096R ? "k4096r" : ps.kind == PassKind::K256 ? "k256"
                    : ps.kind == PassKind::K256R ? "k256r" : ps.kind == PassKind::Col256 ? "col" : "autosort";
    if (!out.empty()) out += " ";
    out += std::string(k) + ":" + std::to_string(ps.radix);
    if (ps.tw_next) out += "+tw";
    if (ps.skip_tw) out += "-tw";
  }
  if (out.size() + 1 > bytes) return fail(TFFT_ERR_ARG, "buffer too small");
  std::memcpy(buf, out.c_str(), out.size() + 1);
  return TFFT_OK;
}

// passes over the data (a narrow column pass with a ragged batch takes two launches for its one pass)
int tfft_plan_num_launches(const tfft_plan* p) {
  if (!p) return 0;
  if (p->sub_col) return tfft_plan_num_launches(p->sub_col) + tfft_plan_num_launches(p->sub_row);
  return static_cast<int>(p->passes.size());
}

size_t tfft_plan_workspace_bytes(const tfft_plan* p) {
  if (p && p->sub_col) return static_cast<size_t>(p->chunk) * p->n * 4;   // planar intermediate [RE | IM] of ONE chunk of the batch
  if (!p || p->passes.size() == 1) {
    // a single pass needs scratch only when asked to run in place
    if (!p || single_kernel(p)) return 0;
  }
  return static_cast<size_t>(p->batch) * p->n * p->inner * 4;   // [batch][RE | IM] halves
}

int tfft_plan_prepare(tfft_plan* p) {
  g_err.clear();
  if (!p) return fail(TFFT_ERR_ARG, "null plan");
  int prev = 0;
  TFFT_HIP(hipGetDevice(&prev));
  TFFT_HIP(hipSetDevice(p->device));
  const int rc = ensure_workspace(p);
  (void)hipSetDevice(prev);
  return rc;
}

int tfft_plan_set_workspace(tfft_plan* p, void* device_ptr, size_t bytes) {
  g_err.clear();
  if (!p) return fail(TFFT_ERR_ARG, "null plan");
  std::lock_guard<std::mutex> lock(p->ws_mutex);
  if (p->ws && p->ws_owned) (void)hipFree(p->ws);
  p->ws = device_ptr;
  p->ws_bytes = device_ptr ? bytes : 0;
  p->ws_owned = false;
  return TFFT_OK;
}

int tfft_exec(const tfft_plan* p, const void* in_re, const void* in_im, void* out_re, void* out_im, void* stream) {
  g_err.clear();
  if (!p) return fail(TFFT_ERR_ARG, "null plan");
  if (!in_re || !in_im || !out_re || !out_im) return fail(TFFT_ERR_ARG, "null data pointer");
  const uint64_t nf = p->n * p->inner;
  const uintptr_t align = (nf >= 8) ? 15 : (2 * nf - 1);
  if ((reinterpret_cast<uintptr_t>(in_re) | reinterpret_cast<uintptr_t>(in_im) | reinterpret_cast<uintptr_t>(out_re) |
       reinterpret_cast<uintptr_t>(out_im)) & align)
    return fail(TFFT_ERR_ARG, "data pointers must be 16-byte aligned");
  // Aliasing: exact in-place (out plane == in plane, same stride) or fully disjoint planes. Anything else (out_re on
  // in_im, shifted or partially overlapping blocks) would be read after it has been overwritten by another workgroup
  // or by an earlier pass of the chain.
  const bool same_re = in_re == out_re, same_im = in_im == out_im;
  // (a pass with re-mapped output rows, the second pass of the fused 2D plan, is only reachable through
  // tfft_plan2d_exec, which owns both of its buffers)
  const bool remapped = p->out_row_shift || p->out_sub_shift;
  if (remapped) {
    int cur0 = 0;
    TFFT_HIP(hipGetDevice(&cur0));
    if (cur0 != p->device) return fail(TFFT_ERR_ARG, "plan was created for another device than the current one");
    return launch_chain(p, in_re, in_im, out_re, out_im, static_cast<hipStream_t>(stream));
  }
  if ((same_re || same_im) && p->in_stride != p->out_stride)
    return fail(TFFT_ERR_ARG, "in-place execution needs equal input and output batch strides");
  if ((!same_re && planes_overlap(in_re, p->in_stride, out_re, p->out_stride, p->batch, nf)) ||
      (!same_im && planes_overlap(in_im, p->in_stride, out_im, p->out_stride, p->batch, nf)) ||
      planes_overlap(in_re, p->in_stride, out_im, p->out_stride, p->batch, nf) ||
      planes_overlap(in_im, p->in_stride, out_re, p->out_stride, p->batch, nf) ||
      planes_overlap(out_re, p->out_stride, out_im, p->out_stride, p->batch, nf))
    return fail(TFFT_ERR_ARG, "input and output planes overlap without being identical (only exact in-place or disjoint planes are supported)");
  int cur = 0;
  TFFT_HIP(hipGetDevice(&cur));
  if (cur != p->device) return fail(TFFT_ERR_ARG, "plan was created for another device than the current one");
  return launch_chain(p, in_re, in_im, out_re, out_im, static_cast<hipStream_t>(stream));
}

// ---------------------------------------------------------------------------
// 2D = row pass + column pass (see include/tfft.h)
// ---------------------------------------------------------------------------
struct tfft_plan2d {
  tfft_plan* row = nullptr;
  tfft_plan* col = nullptr;
  uint64_t rows = 0, cols = 0, batch = 0;
  int device = 0;
  bool fused = false;       // 4096 x 4096: radix-8 column butterfly fused into the row pass + one radix-512 column pass
  // Fused plan, round 4: the batch runs in chunks of `chunk` images, both passes of a chunk back to back, through ONE
  // intermediate image set of chunk images that every chunk re-uses. The 256-MiB Infinity Cache then still holds part of what the
  // row pass wrote when the column pass reads it: 4096^2 x 64 3.45-3.50 -> 3.33 ms (+3.5 ... +5 %, profiles/r4_2d_chunked.txt;
  // 4 images = 256 MiB of intermediate is the measured optimum: 2 images pay more for the ramp and tail of their 50-us launches
  // than the cache returns, 8 images no longer fit), and the workspace is 256 MiB instead of 4 GiB. col_tail: the column plan
  // of the last, shorter chunk.
  uint64_t chunk = 0;
  tfft_plan* col_tail = nullptr;
  mutable std::mutex ws_mutex;
  mutable void* ws = nullptr;
  mutable size_t ws_bytes = 0;
  mutable bool ws_owned = false;
};

int tfft_plan2d_create(uint64_t rows, uint64_t cols, uint64_t batch, int device_id, tfft_plan2d** out) {
  g_err.clear();
  if (!out) return fail(TFFT_ERR_ARG, "null plan pointer");
  *out = nullptr;
  if (!is_pow2(rows) || !is_pow2(cols) || cols < 8 || rows < 2) return fail(TFFT_ERR_ARG, "rows and cols must be powers of two (cols >= 8)");
  if (batch == 0) return fail(TFFT_ERR_ARG, "batch must be positive");
  tfft_plan2d* p = new tfft_plan2d;
  p->rows = rows;
  p->cols = cols;
  p->batch = batch;
  p->device = device_id;
#ifdef TFFT_DEBUG_KERNELS
  static const bool no_fuse = std::getenv("TFFT_2D_NO_FUSE") != nullptr;   // experiment knob
#else
  constexpr bool no_fuse = false;
#endif
  p->fused = (rows == 4096 && cols == 4096 && batch * 512 <= 0xffffffffull && !no_fuse);
  int rc;
  if (p->fused) {
    // pass 1 (k4096r.hpp, ROWS): radix-8 column butterfly over the rows r0 + 512 i, fused with the 4096-point row
    // transforms; it only needs the 4096 kernel's constant tables, which a (4096, 1) plan owns.
    InternalOpts rio;
    rio.rows2d = true;
    rc = create_plan(4096, 1, device_id, nullptr, rio, &p->row);
    if (rc == TFFT_OK) {
      // pass 2: one radix-512 column pass per block of 512 intermediate rows; block s of an image writes rows 8 k' + s
      tfft_plan_opts co = TFFT_PLAN_OPTS_INIT;
      co.in_batch_stride = 512 * cols;
      co.out_batch_stride = rows * cols;
      co.inner = cols;
      co.variant = 67108864;
#ifdef TFFT_2D_COL_VARIANT     // A/B knob: cache-policy bits (262144 / 536870912) for the column pass
      co.variant |= TFFT_2D_COL_VARIANT;
#endif
#ifdef TFFT_2D_COL_ITERS       // A/B knob: rounds per workgroup of the column pass (default: the static partition for a chunk)
      co.launch_iters = TFFT_2D_COL_ITERS;
#endif
#ifdef TFFT_2D_CHUNK           // A/B knob
      constexpr uint64_t kChunkImages = TFFT_2D_CHUNK;
#else
      constexpr uint64_t kChunkImages = 4;
#endif
      p->chunk = std::min<uint64_t>(batch, kChunkImages);
      rc = tfft_plan_create(512, p->chunk * 8, device_id, &co, &p->col);
      if (rc == TFFT_OK && batch % p->chunk) {
        rc = tfft_plan_create(512, (batch % p->chunk) * 8, device_id, &co, &p->col_tail);
        if (rc == TFFT_OK) {
          p->col_tail->out_row_shift = 3;
          p->col_tail->out_sub_shift = 3;
          p->col_tail->out_sub_stride = cols;
        }
      }
      if (rc == TFFT_OK) {
        p->col->out_row_shift = 3;
        p->col->out_sub_shift = 3;
        p->col->out_sub_stride = cols;
        uint8_t* const fake = reinterpret_cast<uint8_t*>(uintptr_t{1} << 20);     // LDS opt-in of the fused row kernel now
        int prev = 0;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(device_id);
        g_prepare = true;
        rc = launch_rows2d(p->row, fake, fake, fake, fake, rows * cols, 512, nullptr);
        g_prepare = false;
        (void)hipSetDevice(prev);
      }
    }
  } else {
    tfft_plan_opts ro = TFFT_PLAN_OPTS_INIT;
    ro.in_batch_stride = cols;        // fully planar lines
    ro.out_batch_stride = cols;
    ro.preserve_input = 1;
    rc = tfft_plan_create(cols, batch * rows, device_id, &ro, &p->row);
    if (rc == TFFT_OK) {
      tfft_plan_opts co = TFFT_PLAN_OPTS_INIT;
      co.in_batch_stride = rows * cols;
      co.out_batch_stride = rows * cols;
      co.inner = cols;
      rc = tfft_plan_create(rows, batch, device_id, &co, &p->col);
    }
  }
  if (rc != TFFT_OK) {
    const std::string keep = g_err;
    tfft_plan2d_destroy(p);
    g_err = keep;
    return rc;
  }
  *out = p;
  return TFFT_OK;
}

void tfft_plan2d_destroy(tfft_plan2d* p) {
  if (!p) return;
  tfft_plan_destroy(p->row);
  tfft_plan_destroy(p->col);
  tfft_plan_destroy(p->col_tail);
  if (p->ws && p->ws_owned) (void)hipFree(p->ws);
  delete p;
}

int tfft_plan2d_num_launches(const tfft_plan2d* p) {
  if (!p) return 0;
  // (passes over the data; the fused plan issues them chunk by chunk, 2 launches per chunk of images)
  return (p->fused ? 1 : tfft_plan_num_launches(p->row)) + tfft_plan_num_launches(p->col);
}

namespace {
inline size_t plan2d_tmp_bytes(const tfft_plan2d* p) { return static_cast<size_t>(p->fused ? p->chunk : p->batch) * p->rows * p->cols * 4; }
inline size_t plan2d_part(size_t bytes) { return (bytes + 255) & ~static_cast<size_t>(255); }
}  // namespace

size_t tfft_plan2d_workspace_bytes(const tfft_plan2d* p) {
  if (!p) return 0;
  if (p->fused) return plan2d_part(plan2d_tmp_bytes(p));          // the intermediate image set only
  return plan2d_part(plan2d_tmp_bytes(p)) + plan2d_part(tfft_plan_workspace_bytes(p->row)) +
         tfft_plan_workspace_bytes(p->col);
}

int tfft_plan2d_set_workspace(tfft_plan2d* p, void* device_ptr, size_t bytes) {
  g_err.clear();
  if (!p) return fail(TFFT_ERR_ARG, "null plan");
  std::lock_guard<std::mutex> lock(p->ws_mutex);
  if (p->ws && p->ws_owned) (void)hipFree(p->ws);
  p->ws = device_ptr;
  p->ws_bytes = device_ptr ? bytes : 0;
  p->ws_owned = false;
  return TFFT_OK;
}

int tfft_plan2d_exec(const tfft_plan2d* p, const void* in_re, const void* in_im, void* out_re, void* out_im,
                     void* stream) {
  g_err.clear();
  if (!p) return fail(TFFT_ERR_ARG, "null plan");
  const size_t need = tfft_plan2d_workspace_bytes(p);
  {
    std::lock_guard<std::mutex> lock(p->ws_mutex);
    if (!p->ws || p->ws_bytes < need) {
      if (p->ws && !p->ws_owned) return fail(TFFT_ERR_WORKSPACE, "workspace handed to tfft_plan2d_set_workspace is too small");
      int prev = 0;
      TFFT_HIP(hipGetDevice(&prev));
      TFFT_HIP(hipSetDevice(p->device));
      if (p->ws) (void)hipFree(p->ws);
      p->ws = nullptr;
      const hipError_t e = hipMalloc(&p->ws, need);
      (void)hipSetDevice(prev);
      if (e != hipSuccess) return hip_fail(e, "hipMalloc(2D workspace)");
      p->ws_bytes = need;
      p->ws_owned = true;
    }
    if (!p->fused) {
      uint8_t* base = static_cast<uint8_t*>(p->ws);
      const size_t tmp = plan2d_part(plan2d_tmp_bytes(p)), row_ws = plan2d_part(tfft_plan_workspace_bytes(p->row));
      int rc = tfft_plan_set_workspace(p->row, row_ws ? base + tmp : nullptr, row_ws);
      if (rc == TFFT_OK) {
        const size_t col_ws = tfft_plan_workspace_bytes(p->col);
        rc = tfft_plan_set_workspace(p->col, col_ws ? base + tmp + row_ws : nullptr, col_ws);
      }
      if (rc != TFFT_OK) return rc;
    }
  }
  _Float16* t_re = static_cast<_Float16*>(p->ws);
  _Float16* t_im = t_re + static_cast<size_t>(p->fused ? p->chunk : p->batch) * p->rows * p->cols;
  if (p->fused) {
    if (!in_re || !in_im || !out_re || !out_im) return fail(TFFT_ERR_ARG, "null data pointer");
    if ((reinterpret_cast<uintptr_t>(in_re) | reinterpret_cast<uintptr_t>(in_im) | reinterpret_cast<uintptr_t>(out_re) |
         reinterpret_cast<uintptr_t>(out_im)) & 15)
      return fail(TFFT_ERR_ARG, "data pointers must be 16-byte aligned");
    int cur = 0;
    TFFT_HIP(hipGetDevice(&cur));
    if (cur != p->device) return fail(TFFT_ERR_ARG, "plan was created for another device than the current one");
    const uint64_t image = p->rows * p->cols;
    for (uint64_t i = 0; i < p->batch; i += p->chunk) {
      const uint64_t c = std::min<uint64_t>(p->chunk, p->batch - i);
      int rc = launch_rows2d(p->row, static_cast<const _Float16*>(in_re) + i * image, static_cast<const _Float16*>(in_im) + i * image, t_re, t_im,
                             image, static_cast<uint32_t>(c * 512), static_cast<hipStream_t>(stream));
      if (rc != TFFT_OK) return rc;
      rc = tfft_exec(c == p->chunk ? p->col : p->col_tail, t_re, t_im, static_cast<_Float16*>(out_re) + i * image,
                     static_cast<_Float16*>(out_im) + i * image, stream);
      if (rc != TFFT_OK) return rc;
    }
    return TFFT_OK;
  }
  int rc = tfft_exec(p->row, in_re, in_im, t_re, t_im, stream);
  if (rc != TFFT_OK) return rc;
  return tfft_exec(p->col, t_re, t_im, out_re, out_im, stream);
}

int tfft_plan2d_exec_inverse(const tfft_plan2d* p, const void* in_re, const void* in_im, void* out_re, void* out_im,
                             void* stream) {
  return tfft_plan2d_exec(p, in_im, in_re, out_im, out_re, stream);
}

int tfft_permute_twiddle(const void* in_re, const void* in_im, void* out_re, void* out_im, uint64_t a, uint64_t b,
                         uint64_t c, uint64_t n_tw, uint64_t e0, void* stream) {
  g_err.clear();
  if (!in_re || !in_im || !out_re || !out_im) return fail(TFFT_ERR_ARG, "null data pointer");
  if (a == 0 || b == 0 || c == 0 || (c % 8)) return fail(TFFT_ERR_ARG, "extents must be positive and C a multiple of 8");
  if (n_tw && !is_pow2(n_tw)) return fail(TFFT_ERR_NOT_POW2, "twiddle modulus has to be a power of 2");
  if ((reinterpret_cast<uintptr_t>(in_re) | reinterpret_cast<uintptr_t>(in_im) | reinterpret_cast<uintptr_t>(out_re) |
       reinterpret_cast<uintptr_t>(out_im)) & 15)
    return fail(TFFT_ERR_ARG, "data pointers must be 16-byte aligned");
  if (in_re == out_re || in_im == out_im) return fail(TFFT_ERR_ARG, "permute cannot run in place");
  permute::Args pa{static_cast<const uint16_t*>(in_re), static_cast<const uint16_t*>(in_im),
                   static_cast<uint16_t*>(out_re), static_cast<uint16_t*>(out_im), a, b, c / 8, n_tw, e0};
  const uint64_t total = a * b * (c / 8);
  const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>((total + permute::kBlock - 1) / permute::kBlock, 16384));
  hipLaunchKernelGGL(permute::permute_twiddle_kernel, dim3(grid), dim3(permute::kBlock), 0,
                     static_cast<hipStream_t>(stream), pa);
  TFFT_HIP(hipGetLastError());
  return TFFT_OK;
}

int tfft_exec_inverse(const tfft_plan* p, const void* in_re, const void* in_im, void* out_re, void* out_im, void* stream) {
  // (1/N) sum_j x[j] exp(+2 pi i jk/N) = swap(F(swap(x))) with swap(a + ib) = b + ia and F the forward,
  // 1/N-scaled transform: exchanging the RE and IM planes on both sides is the whole inverse.
  return tfft_exec(p, in_im, in_re, out_im, out_re, stream);
}

int tfft_deinterleave(const void* in_half2, void* out_re, void* out_im, uint64_t count, void* stream) {
  g_err.clear();
  if (!in_half2 || !out_re || !out_im) return fail(TFFT_ERR_ARG, "null data pointer");
  if (count == 0 || (count % 8)) return fail(TFFT_ERR_ARG, "count must be a positive multiple of 8 complex samples");
  if ((reinterpret_cast<uintptr_t>(in_half2) | reinterpret_cast<uintptr_t>(out_re) | reinterpret_cast<uintptr_t>(out_im)) & 15)
    return fail(TFFT_ERR_ARG, "data pointers must be 16-byte aligned");
  const uint64_t n8 = count / 8;
  const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>((n8 + permute::kBlock - 1) / permute::kBlock, 16384));
  hipLaunchKernelGGL(permute::deinterleave_kernel, dim3(grid), dim3(permute::kBlock), 0, static_cast<hipStream_t>(stream),
                     static_cast<const permute::uv4*>(in_half2), static_cast<permute::uv4*>(out_re),
                     static_cast<permute::uv4*>(out_im), n8);
  TFFT_HIP(hipGetLastError());
  return TFFT_OK;
}

int tfft_interleave(const void* in_re, const void* in_im, void* out_half2, uint64_t count, void* stream) {
  g_err.clear();
  if (!in_re || !in_im || !out_half2) return fail(TFFT_ERR_ARG, "null data pointer");
  if (count == 0 || (count % 8)) return fail(TFFT_ERR_ARG, "count must be a positive multiple of 8 complex samples");
  if ((reinterpret_cast<uintptr_t>(in_re) | reinterpret_cast<uintptr_t>(in_im) | reinterpret_cast<uintptr_t>(out_half2)) & 15)
    return fail(TFFT_ERR_ARG, "data pointers must be 16-byte aligned");
  const uint64_t n8 = count / 8;
  const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>((n8 + permute::kBlock - 1) / permute::kBlock, 16384));
  hipLaunchKernelGGL(permute::interleave_kernel, dim3(grid), dim3(permute::kBlock), 0, static_cast<hipStream_t>(stream),
                     static_cast<const permute::uv4*>(in_re), static_cast<const permute::uv4*>(in_im),
                     static_cast<permute::uv4*>(out_half2), n8);
  TFFT_HIP(hipGetLastError());
  return TFFT_OK;
}

int tfft_synth_uniform(void* re, void* im, uint64_t n, uint64_t batch, uint64_t batch_stride, uint64_t first_fft,
                       uint64_t seed, void* stream) {
  g_err.clear();
  if (!re || !im) return fail(TFFT_ERR_ARG, "null data pointer");
  if (n == 0 || (n % 8) || batch == 0) return fail(TFFT_ERR_ARG, "n must be a positive multiple of 8 and batch positive");
  if (batch_stride == 0) batch_stride = 2 * n;
  if (batch_stride < n || (batch_stride % 8)) return fail(TFFT_ERR_ARG, "batch stride must be a multiple of 8 halves and >= n");
  if ((reinterpret_cast<uintptr_t>(re) | reinterpret_cast<uintptr_t>(im)) & 15) return fail(TFFT_ERR_ARG, "data pointers must be 16-byte aligned");
  const uint64_t total = batch * 2 * (n / 8);
  const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>((total + synth::kBlock - 1) / synth::kBlock, 1u << 20));
  hipLaunchKernelGGL(synth::uniform_kernel, dim3(grid), dim3(synth::kBlock), 0, static_cast<hipStream_t>(stream),
                     static_cast<uint16_t*>(re), static_cast<uint16_t*>(im), n / 8, batch, batch_stride, first_fft, seed);
  TFFT_HIP(hipGetLastError());
  return TFFT_OK;
}

int tfft_kernel_list(char* buf, size_t bytes) {
  g_err.clear();
  std::string out;
  for (const ColRow& r : kColTable) out += std::string(r.name) + "\n";
  if (!buf || out.size() + 1 > bytes) return fail(TFFT_ERR_ARG, "tfft_kernel_list: buffer too small (" + std::to_string(out.size() + 1) + " bytes needed)");
  std::memcpy(buf, out.c_str(), out.size() + 1);
  return static_cast<int>(kColRows);
}

const char* tfft_plan_kernel_name(const tfft_plan* p) {
  if (!p) return "";
  if (p->sub_col) return tfft_plan_kernel_name(p->sub_col);
  switch (p->passes[0].kind) {
    case PassKind::K4096: return "fft4096_kernel";
    case PassKind::K256: return "fft256_kernel";
    case PassKind::K256R: return "fft256r_kernel";
    case PassKind::K4096R: return "fft4096r_kernel";
    case PassKind::Col256: return "colfft256_kernel";
    default: return "pass_kernel";
  }
}

double tfft_plan_algorithmic_bytes(const tfft_plan* p) {
  if (!p) return 0.0;
  const double samples = static_cast<double>(p->n) * static_cast<double>(p->inner) * static_cast<double>(p->batch);
  return 8.0 * samples * tfft_plan_num_launches(p);   // 4 B read + 4 B written per sample per pass
}

double tfft_plan_mfma_flops(const tfft_plan* p) {
  if (!p) return 0.0;
  if (p->sub_col) {
    // the sub-plans are built for ONE chunk of the batch (launch_chain runs batch / chunk of them, plus the tail's own pair)
    const double full = static_cast<double>(p->batch / p->chunk);
    double f = full * (tfft_plan_mfma_flops(p->sub_col) + tfft_plan_mfma_flops(p->sub_row));
    if (p->sub_col_tail) f += tfft_plan_mfma_flops(p->sub_col_tail) + tfft_plan_mfma_flops(p->sub_row_tail);
    return f;
  }
  // one radix-16 MFMA stage = 16 tiles x 2 MFMA(16x16x32) x 16384 flop per 4096 samples = 128 flop/sample
  double stages = 0;
  for (const Pass& ps : p->passes)
    stages += (ps.kind == PassKind::K4096 || ps.kind == PassKind::K4096R) ? 3 : ((ps.kind == PassKind::Col256 || ps.kind == PassKind::K256 || ps.kind == PassKind::K256R) ? 2 : 0);
  return 128.0 * stages * static_cast<double>(p->n) * static_cast<double>(p->inner) * static_cast<double>(p->batch);
}

}  // extern "C"

#include "dist.hpp"
#include "staging.hpp"
