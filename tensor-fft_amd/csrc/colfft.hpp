// colfft.hpp — 256-point FFT along a strided axis, 16 adjacent columns per wave (gfx950).
//
// The building block for everything longer than 4096 and for FFTs along a non-contiguous axis
// (2D column pass, local passes of the distributed transform). It plays the role of the reference's
// TensorRadix16 pass (src/base/TensorRadix16.cu:36-214: one L -> 16 L combine per launch, 32-byte
// global segments) but covers a radix-256 step per launch and keeps the autosort ("Stockham")
// indexing of stockham.hpp so the two kinds of pass can be chained:
//
//   input   x[i * pitch + m]      i = 0..255 (the digit being transformed), m = flattened column
//   output  y[rest * 256 Ns + k * Ns + kprev]   with m = rest * Ns + kprev   (Ns = "ns_f")
//   Ns == 1 (first pass of a plain 1D transform): y[m * 256 + k], each column's spectrum contiguous.
//
// If another pass follows, the twiddle that pass needs on its input, w_T^(a (kprev' + Ns' k)), is
// applied here to the fp32 accumulators (per lane: five table look-ups and a 16-step recurrence), so
// the next pass reads plain data.
//
// Machine mapping = stages 1 and 2 of k4096.hpp with the column index in the place of n0:
//   i = i_lo + 16 i_hi; stage 1 contracts i_hi (transposed LDS reads, F as the A operand), the
//   4x4 permlane transposes move i_lo's high bits next to the lanes, stage 2 contracts i_lo against
//   the same G_ka tables (twiddle w256^(i_lo ka) folded in). k = ka + 16 kb.
//   Mode kColsOnLanes  (Ns == 1): stage 2 with the data as B operand: lane = column, each lane
//                      ends up with 64 consecutive k -> 128-byte runs per lane.
//   Mode kColsInRegs   (Ns >= 16): data as A operand: lane = kb, registers = 4 adjacent columns
//                      -> 8-byte pieces, 32-byte runs per lane group, rows k at stride Ns.
#pragma once

#include "k4096.hpp"

// Streaming hint of the column passes' global accesses (every byte is touched once per pass). -DTFFT_COL_NO_NT builds them as
// plain accesses: the A/B behind the chunked execution experiments of round 4 (does a non-temporal access keep an intermediate
// out of the Infinity Cache?).
// kPlainAcc: a constexpr every kernel that uses these macros defines at its top (template parameter PLAIN, or !NT in the radix-256
// kernel): plain instead of non-temporal global accesses, for plans whose whole footprint fits the Infinity Cache (tfft.hip
// cache_policy). A COMPILE-time choice on purpose. A run-time branch was tried first (round 4): identical loads / stores in an
// if / else are merged by the optimiser, which keeps only the metadata both sides carry and so silently DROPS the non-temporal
// hint (seen in the ISA: every access came out plain); keeping the sides apart with empty asm statements costs the radix-1024
// kernel, which sits at 256 VGPRs, 500-700 B of scratch.
#if defined(TFFT_COL_NO_NT) || defined(TFFT_COL_PLAIN_LOADS)
#define TFFT_NT_LOAD(p) (*(p))
#else
#define TFFT_NT_LOAD(p) (kPlainAcc ? *(p) : __builtin_nontemporal_load(p))
#endif
#if defined(TFFT_COL_NO_NT) || defined(TFFT_COL_PLAIN_STORES)
#define TFFT_NT_STORE(v, p) (*(p) = (v))
#else
#define TFFT_NT_STORE(v, p)                         \
  do {                                              \
    if (kPlainAcc) *(p) = (v);                      \
    else __builtin_nontemporal_store(v, p);         \
  } while (0)
#endif

namespace colfft {

using namespace k4096;

constexpr int kLdsTable = 16384;                                   // G only
constexpr int kWaveRegion = 18432;                                 // 16 KiB copy-in image, reused (padded) for the output
constexpr int kStagePitch = 528;                                   // 512 + 16: conflict-free / 2-way staging writes
constexpr int kStagePlane = 16 * kStagePitch;                      // 8448 B per plane
constexpr int kLdsBytes = kLdsTable + kWavesPerBlock * kWaveRegion;     // 160 KiB

enum : int { kColsOnLanes = 0, kColsInRegs = 1 };

struct Args {
  const uint16_t* in_re;
  const uint16_t* in_im;
  uint16_t* out_re;
  uint16_t* out_im;
  uint64_t in_stride, out_stride;   // halves between the (outer) batch entries
  uint64_t pitch;                   // halves between consecutive i (= number of flattened columns)
  uint64_t ns_f;                    // flattened Ns (a power of two: 1, or a multiple of 16)
  uint32_t ns_f_shift;              // log2(ns_f)
  uint32_t groups;                  // column groups of 16 per batch entry = pitch / 16
  uint32_t tasks;                   // groups * batch
  // twiddle for the next pass (unused when !TW): E = a * (kprev + ns * k) mod T, looked up in w_N tables
  uint32_t inner_shift;             // log2 C: flattened column = column * C + c
  uint32_t a_shift;                 // a = rest >> a_shift        (rest = m >> ns_f_shift)
  uint64_t ns;                      // unflattened Ns of THIS pass = ns_f / C
  uint64_t t_mask;                  // T - 1
  uint64_t n_over_t;                // N / T
  double inv_t;                     // 1 / T
  uint64_t n_mask;                  // N - 1
#ifdef TFFT_DEBUG_KERNELS
  uint32_t copy_only;               // timing experiment (WRONG output): move the image straight back out
#endif
  // radix-512 columns-in-registers pass as the second pass of a 2D transform (see k4096r.hpp, ROWS): output row k of
  // batch entry e goes to (e >> out_sub_shift) * out_stride + (e & mask) * out_sub_stride + (k << out_row_shift) rows
  uint32_t out_row_shift;
  uint32_t out_sub_shift;
  uint64_t out_sub_stride;
  // Segmented input rows (workgroup-cooperative radix-256 / radix-512 forms): row i of the [radix][pitch] input matrix starts
  // (i >> in_seg_shift) * in_seg_gap halves further on. The row transforms of a transform distributed over P GPUs read what
  // the all-to-all delivered, P chunks [p'][k][c], in place: row k of the local [K][N2] matrix is P segments of C contiguous
  // samples, K C apart (tfft_dist_*; no re-order pass). Off: shift 31, gap 0.
  uint32_t in_seg_shift;
  uint64_t in_seg_gap;
  // Column slab of a four-step radix-256 pass (kTwFourStep forms of colfft256_wg_kernel only; a transform distributed over several
  // GPUs with its exchange overlapped slab by slab, dist.hpp): the launch covers column blocks [blk_first, blk_first + blk_count)
  // (blk_count 0 = all), and output row k of flattened column m goes to
  //     (k << out_pitch_shift) + (k >> out_seg_shift) * out_seg_gap + (m - out_col0) + out_base        halves
  // The defaults (out_pitch_shift = ns_f_shift, out_seg_shift = 31, everything else 0) are the plain [k][m] matrix; a slab writes
  // the send layout [peer q][slab s][k][c_s], whose (q, s) pieces are contiguous for ncclSend.
  uint32_t blk_first, blk_count;
  uint32_t out_pitch_shift, out_seg_shift;
  uint64_t out_seg_gap, out_col0, out_base;
  const float2* tw_lo;
  const float2* tw_hi;
  const uint8_t* tables;            // k4096::build_tables blob
  // tfft_plan_opts.scale (include/tfft.h): tw_scale multiplies the twiddles a TW pass applies (1, or the plan's single
  // 1/N of "scale once" when this is the plan's last fp32 multiply); comb_scale: read-out factor of a final radix-512 /
  // radix-1024 pass under "scale once" (template parameter SC), otherwise unused
  float tw_scale;
  float comb_scale;
  // TW == kTwFourStep: output row k of flattened column m is multiplied by w_M^(k (tw4_col0 + m)), M = n_mask + 1 (the
  // twiddle tables are then built for M, the length of the whole four-step transform, not for this pass's radix):
  // the w_N^(k1 n2) step of a transform split as N = N1 N2 (transposed-order plans, local passes of a distributed one)
  uint64_t tw4_col0;
#ifdef TFFT_DEBUG_KERNELS
  // measurement hook (tools/exp_wg_end_times.py; null in normal use): wall_clock64 at entry [i] and exit [8192 + i] and the XCC
  // id [16384 + i] of workgroup i of the radix-1024 pass
  unsigned long long* wg_times;
#endif
};

// twiddle forms of a column pass: none, the next autosort pass's input twiddles, the four-step twiddle
enum : int { kTwNone = 0, kTwNext = 1, kTwFourStep = 2 };

// Measurement build only (tools/exp_lat_phases.py): wall clock (100 MHz) of workgroup i at phase ph into wg_times[ph * 8192 + i],
// the shader clock (s_memtime) into wg_times[(8 + ph) * 8192 + i]. Nothing in the shipped library.
#ifdef TFFT_DEBUG_KERNELS
#define TFFT_WG_STAMP(a, ph)                                                                   \
  do {                                                                                         \
    if ((a).wg_times && threadIdx.x == 0 && blockIdx.x < 8192) {                               \
      (a).wg_times[(ph) * 8192 + blockIdx.x] = wall_clock64();                                 \
      (a).wg_times[(8 + (ph)) * 8192 + blockIdx.x] = __builtin_amdgcn_s_memtime();             \
    }                                                                                          \
  } while (0)
#else
#define TFFT_WG_STAMP(a, ph) do { } while (0)
#endif

// Store policy of a column pass (round 4). A four-step pass (kTwFourStep) writes the INTERMEDIATE of a transposed-order plan, which
// the contiguous row pass of the same chunk of the batch reads right back (tfft.hip launch_chain runs those plans chunk by
// chunk): plain stores, so that the lines stay in the 256-MiB Infinity Cache: 2^20 x 1024 in transposed output order 341-348 ->
// 375 Gsamples/s (profiles/r4_chunked_transposed.txt). Everything else streams out non-temporally: a plan's last pass so that its
// output does not push intermediates out, and the kTwNext passes of natural-order plans, which run over the whole batch (chunks
// and plain stores were measured there too: 2^20 x 1024 341 -> 307-326, profiles/r4_chunked_1d.txt: the strided second pass does
// not profit from the cache the way a contiguous one does). All loads stay non-temporal: the caller's input must not displace
// the intermediates, and an intermediate is dead once read (plain loads of it: -16 % in the 2D plan, profiles/r4_ab_2d_chunk.txt).
// -DTFFT_COL_STREAM_ALL = round 3's policy (everything non-temporal), for A/B builds.
#ifdef TFFT_COL_STREAM_ALL
#define TFFT_ST_PASS(TW, v, p) TFFT_NT_STORE(v, p)
#else
#define TFFT_ST_PASS(TW, v, p)                      \
  do {                                              \
    if ((TW) == kTwFourStep) *(p) = (v);            \
    else TFFT_NT_STORE(v, p);                       \
  } while (0)
#endif

struct cpx {
  float re, im;
};
__device__ __forceinline__ cpx cmul(cpx a, cpx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }

// w_T^(e_t) = exp(-2 pi i e_t / T), e_t already reduced mod T.
// LUT: two-level fp32 tables of w_N (exact to fp32 rounding, but five dependent L2 round trips per task);
// otherwise v_cos_f32 / v_sin_f32, which take their argument in revolutions (absolute error ~1e-6, far below
// the fp16 resolution of the data they multiply; no memory access).
template <bool LUT>
__device__ __forceinline__ cpx lookup(const Args& a, uint64_t e_t) {
  if (LUT) {
    const uint64_t e = (e_t * a.n_over_t) & a.n_mask;
    // (tw_hi always has entry 0 = 1 + 0 i, also for N <= 8192: two independent loads and one product, no branch)
    const float2 lo = a.tw_lo[e & 8191];
    const float2 hi = a.tw_hi[e >> 13];
    return cmul(cpx{lo.x, lo.y}, cpx{hi.x, hi.y});
  }
  const float frac = static_cast<float>(static_cast<double>(e_t) * a.inv_t);
  return cpx{__builtin_amdgcn_cosf(frac), -__builtin_amdgcn_sinf(frac)};
}

__device__ __forceinline__ cpx lookup_n(const Args& a, uint64_t e_n) {   // w_N^(e_n), e_n already reduced mod N
  const float2 lo = a.tw_lo[e_n & 8191];
  const float2 hi = a.tw_hi[e_n >> 13];
  return cmul(cpx{lo.x, lo.y}, cpx{hi.x, hi.y});
}

template <int MODE, int TW, bool STAGE, bool LUT>
__global__ __launch_bounds__(kThreads, 2) void colfft256_kernel(Args a) {
  static_assert(TW == kTwNone || TW == kTwNext, "the per-wave kernel has no four-step twiddle form");
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < kLdsTable / 16; i += kThreads)
    reinterpret_cast<u4*>(lds)[i] = reinterpret_cast<const u4*>(a.tables + kOffG)[i];
  const h8 f_re = *reinterpret_cast<const h8*>(a.tables + kOffF1n + lane * 32);
  const h8 f_im = *reinterpret_cast<const h8*>(a.tables + kOffF1n + lane * 32 + 16);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  uint8_t* const wl = lds + kLdsTable + wave * kWaveRegion;
  const uint32_t wl_off = __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t*)wl)));
  const uint8_t* const g_tab = lds + lane * 16;
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, x = lane & 15;
  // LDS image of a plane: 16-byte slot (i_lo * 16 + i_hi) * 2 + h holds columns 8h..8h+7 of row i_lo + 16 i_hi.
  // transposed read of tile i_lo: lane 16g + 4q + p supplies row i_hi = 4g + q, columns 4p..4p+3.
  const uint8_t* const tr_base = wl + (4 * g + q) * 32 + 8 * p;
  // copy-in geometry: instruction i, lane l: h = l & 1, i_hi = (l >> 1) & 15, i_lo = 2 i + (l >> 5)
  const uint64_t in_lane_off = (static_cast<uint64_t>(16 * ((lane >> 1) & 15) + (lane >> 5)) * a.pitch + 8 * (lane & 1)) * 2;

  Rotor rot(blockIdx.x, gridDim.x);                                      // (workgroup order: k4096::Rotor)
  for (uint32_t task = rot.item() * kWavesPerBlock + wave; task < a.tasks; rot.advance(), task = rot.item() * kWavesPerBlock + wave) {
    const uint32_t bidx = task / a.groups;
    const uint64_t m0 = static_cast<uint64_t>(task - bidx * a.groups) * 16;
    const uint8_t* src_re = reinterpret_cast<const uint8_t*>(a.in_re + bidx * a.in_stride + m0) + in_lane_off;
    const uint8_t* src_im = reinterpret_cast<const uint8_t*>(a.in_im + bidx * a.in_stride + m0) + in_lane_off;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint8_t* gr = src_re + static_cast<uint64_t>(2 * i) * a.pitch * 2;
      const uint8_t* gi = src_im + static_cast<uint64_t>(2 * i) * a.pitch * 2;
      const uint32_t d0 = wl_off + i * 1024, d1 = wl_off + 8192 + i * 1024;
      uint32_t keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %3\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %1, off\n\t"
          "s_mov_b32 m0, %4\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %2, off\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(gr), "v"(gi), "s"(d0), "s"(d1)
          : "memory");
    }
    // (table look-ups issued while the copy is in flight)
    // ---- twiddle set-up: tw(col, k) = base_r * step^ka with k = ka + 16 kb,
    //      exponent E = a (kprev + ns k) mod T, a and kprev functions of the column
    const uint64_t rest = m0 >> a.ns_f_shift;
    const uint64_t kprev_f0 = m0 - (rest << a.ns_f_shift);
    cpx base[4], step = {1.f, 0.f};
    if (TW) {
      const uint64_t rest_l = (MODE == kColsOnLanes) ? ((m0 + x) >> a.ns_f_shift) : rest;   // this lane's column
      const uint64_t av = rest_l >> a.a_shift;
      step = lookup<LUT>(a, (av * (a.ns & a.t_mask)) & a.t_mask);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t kb = (MODE == kColsOnLanes) ? (4 * g + r) : x;
        const uint64_t kprev = (MODE == kColsOnLanes)
                                   ? (((m0 + x) - (rest_l << a.ns_f_shift)) >> a.inner_shift)
                                   : ((kprev_f0 + 4 * g + r) >> a.inner_shift);
        base[r] = lookup<LUT>(a, (av * ((kprev + a.ns * 16 * kb) & a.t_mask)) & a.t_mask);
        base[r].re *= a.tw_scale;
        base[r].im *= a.tw_scale;
      }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef TFFT_DEBUG_KERNELS
    if (a.copy_only) {
      // data-movement ceiling of this kernel's access pattern: same 32-byte-per-row pieces out as in
      uint16_t* const c_re = a.out_re + bidx * a.out_stride + m0;
      uint16_t* const c_im = a.out_im + bidx * a.out_stride + m0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const u4 vr = *reinterpret_cast<const u4*>(wl + i * 1024 + 16 * lane);
        const u4 vi = *reinterpret_cast<const u4*>(wl + 8192 + i * 1024 + 16 * lane);
        const uint64_t o = static_cast<uint64_t>(16 * ((lane >> 1) & 15) + (lane >> 5) + 2 * i) * a.pitch + 8 * (lane & 1);
        *reinterpret_cast<u4*>(c_re + o) = vr;
        *reinterpret_cast<u4*>(c_im + o) = vi;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      continue;
    }
#endif

    // ---- stage 1: D1_ilo[ka = 4g + r][column = lane & 15]
    uint32_t pr[8][4], pi[8][4];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      f4 dre[2], dim[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const uint8_t* ad = tr_base + (2 * t + e) * 512;
        const s4 xr = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(ad));
        const s4 xi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(ad + 8192));
        const u4 raw = {__builtin_bit_cast(u2, xr).x, __builtin_bit_cast(u2, xr).y,
                        __builtin_bit_cast(u2, xi).x, __builtin_bit_cast(u2, xi).y};
        const h8 xv = __builtin_bit_cast(h8, raw);
        dre[e] = mfma(f_re, xv);
        dim[e] = mfma(f_im, xv);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pr[t][r] = pk(dre[0][r], dre[1][r]);
        pi[t][r] = pk(dim[0][r], dim[1][r]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // image consumed: the next task may overwrite it
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        transpose4(pr[0 + pp][r], pr[2 + pp][r], pr[4 + pp][r], pr[6 + pp][r]);
        transpose4(pi[0 + pp][r], pi[2 + pp][r], pi[4 + pp][r], pi[6 + pp][r]);
      }

    uint16_t* const o_re = a.out_re + bidx * a.out_stride;
    uint16_t* const o_im = a.out_im + bidx * a.out_stride;
    cpx pw = {1.f, 0.f};                                   // step^ka, advanced tile by tile
    float hold_re[4], hold_im[4];                          // kColsOnLanes: even tile waiting for its odd partner
    uint32_t acc_re[4][4], acc_im[4][4];                   // kColsOnLanes: [r][pair within half]
#pragma unroll
    for (int ka = 0; ka < 16; ++ka) {
      const int aa = ka >> 2, r0 = ka & 3;
      const u4 draw = {pr[2 * aa][r0], pr[2 * aa + 1][r0], pi[2 * aa][r0], pi[2 * aa + 1][r0]};
      const h8 dop = __builtin_bit_cast(h8, draw);
      const u4 graw = *reinterpret_cast<const u4*>(g_tab + ka * 1024);
      f4 e_re, e_im;
      if (MODE == kColsOnLanes) {      // rows kb = 4g + r, column on the lane
        e_re = mfma(__builtin_bit_cast(h8, graw), dop);
        e_im = mfma(im_form(graw), dop);
      } else {                         // rows = columns 4g + r, kb on the lane
        e_re = mfma(dop, __builtin_bit_cast(h8, graw));
        e_im = mfma(dop, im_form(graw));
      }
      if (TW) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const cpx w = cmul(base[r], pw);
          const float vr = e_re[r] * w.re - e_im[r] * w.im;
          const float vi = e_re[r] * w.im + e_im[r] * w.re;
          e_re[r] = vr;
          e_im[r] = vi;
        }
        pw = cmul(pw, step);
      }
      if (MODE == kColsInRegs) {
        // 4 adjacent columns 4g..4g+3 of row k = ka + 16 kb (kb = lane & 15): one 8-byte piece per plane
        const u2 vr = {pk(e_re[0], e_re[1]), pk(e_re[2], e_re[3])};
        const u2 vi = {pk(e_im[0], e_im[1]), pk(e_im[2], e_im[3])};
        if (STAGE) {   // staged at 528 kb + 32 ka + 8 g (the 16-byte pad per kb block spreads the lanes over the banks)
          *reinterpret_cast<u2*>(wl + kStagePitch * x + 32 * ka + 8 * g) = vr;
          *reinterpret_cast<u2*>(wl + kStagePlane + kStagePitch * x + 32 * ka + 8 * g) = vi;
        } else {
          const uint64_t o = ((rest << 8) << a.ns_f_shift) + (static_cast<uint64_t>(ka + 16 * x) << a.ns_f_shift) + kprev_f0 + 4 * g;
          *reinterpret_cast<u2*>(o_re + o) = vr;
          *reinterpret_cast<u2*>(o_im + o) = vi;
        }
      } else {
        if ((ka & 1) == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            hold_re[r] = e_re[r];   // (no __builtin_bit_cast on a vector element: clang reads element 0)
            hold_im[r] = e_im[r];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc_re[r][(ka >> 1) & 3] = pk(hold_re[r], e_re[r]);
            acc_im[r][(ka >> 1) & 3] = pk(hold_im[r], e_im[r]);
          }
          if ((ka & 7) == 7) {
            // k = 16 (4g + r) + 8 half .. + 7 of column x: 16 bytes per plane
            const int half = ka >> 3;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const u4 vr = {acc_re[r][0], acc_re[r][1], acc_re[r][2], acc_re[r][3]};
              const u4 vi = {acc_im[r][0], acc_im[r][1], acc_im[r][2], acc_im[r][3]};
              if (STAGE) {   // row x of a [column][k] image with 528-byte rows
                *reinterpret_cast<u4*>(wl + kStagePitch * x + 32 * (4 * g + r) + 16 * half) = vr;
                *reinterpret_cast<u4*>(wl + kStagePlane + kStagePitch * x + 32 * (4 * g + r) + 16 * half) = vi;
              } else {
                const uint64_t o = (m0 + x) * 256 + 16 * (4 * g + r) + 8 * half;
                *reinterpret_cast<u4*>(o_re + o) = vr;
                *reinterpret_cast<u4*>(o_im + o) = vi;
              }
            }
          }
        }
      }
    }
    // ---- (STAGE) staged image -> HBM, 16 bytes per lane
#pragma unroll
    for (int j = 0; STAGE && j < 8; ++j) {
      const uint32_t off = kStagePitch * (2 * j + (lane >> 5)) + 16 * (lane & 31);
      const u4 vr = *reinterpret_cast<const u4*>(wl + off);
      const u4 vi = *reinterpret_cast<const u4*>(wl + kStagePlane + off);
      uint64_t o;
      if (MODE == kColsInRegs) {
        // image row = output row k = 32 j + (lane >> 1), columns 8 (lane & 1) .. + 7: a 32-byte run per row
        const uint64_t k = 32 * j + (lane >> 1);
        o = ((rest << 8) << a.ns_f_shift) + (k << a.ns_f_shift) + kprev_f0 + 8 * (lane & 1);
      } else {
        // image row = column f = 2 j + (lane >> 5): its 256 outputs are 512 contiguous bytes
        o = (m0 + 2 * j + (lane >> 5)) * 256 + 8 * (lane & 31);
      }
      *reinterpret_cast<u4*>(o_re + o) = vr;
      *reinterpret_cast<u4*>(o_im + o) = vi;
    }
    if (STAGE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // image read out before the next copy-in lands on it
  }
}


// ---------------------------------------------------------------------------
// Workgroup-cooperative form: the 8 waves of a workgroup take 8 ADJACENT column groups (128 columns) and move
// them through one shared LDS image [256 rows][256 bytes] per plane, so that every global access of the
// workgroup is a full 256-byte row segment (4 rows x 256 B per wave instruction) instead of 32-byte pieces:
// 4x fewer L2 requests (PMC: the per-wave kernel issues 3x the requests of an autosort pass for the same bytes).
// 16-byte chunk c of row r lives at slot c ^ (2 ((r >> 4) & 7)) of the row (swizzle applied on the SOURCE
// address of the LDS-DMA), which makes each wave's transposed reads of its own 32-byte slab conflict free.
// Needs pitch % 128 == 0; for the columns-in-registers form also ns_f % 128 == 0 (the 128 columns then share
// `rest` and an output row is 256 contiguous bytes, staged through the same image and stored as full rows).
// Four workgroup barriers per 128-column block (two in the columns-on-lanes form, whose stores are per wave).
// ---------------------------------------------------------------------------
// W = waves per workgroup (8 or 4): a workgroup owns 16 W adjacent columns; the image of a plane is 256 rows of
// 32 W bytes, addressed as "super-rows" of 256 bytes (= one row for W = 8, two rows for W = 4) so that the same
// 16-slot swizzle keeps the transposed reads conflict free. With W = 4 a workgroup needs 80 KiB and two of them
// share a CU: one loads or stores while the other computes (a single 8-wave workgroup runs its copy-in, compute
// and store phases strictly one after the other, with nothing in flight during the compute phase).
template <int W>
struct WgGeom {
  static constexpr int kThreadsW = 64 * W;
  static constexpr int kRowBytes = 32 * W;            // one image row of one plane
  static constexpr int kRps = 8 / W;                  // rows per 256-byte super-row
  static constexpr int kCpr = 2 * W;                  // 16-byte chunks per row
  static constexpr int kPlane = 256 * kRowBytes;      // 64 KiB (W = 8) / 32 KiB (W = 4)
  static constexpr int kLds = kLdsTable + 2 * kPlane; // 144 KiB / 80 KiB
  static constexpr int kCols = 16 * W;
};
constexpr int kWgLdsBytes = WgGeom<8>::kLds;

// STG (columns-on-lanes form only): stage a wave's 16 output rows (one per column, 512 contiguous bytes each)
// through its own 8-KiB slice of the image and store them as full rows, instead of 16-byte pieces from registers.
template <int MODE, int TW, bool NT, int W, bool STG = false>
__global__ __launch_bounds__(64 * W, 2) void colfft256_wg_kernel(Args a) {
  static_assert(TW != kTwFourStep || MODE == kColsInRegs, "the four-step twiddle exists for the columns-in-registers form");
  constexpr bool kPlainAcc = !NT;
  using G = WgGeom<W>;
  constexpr int kPlane = G::kPlane, kRps = G::kRps, kCpr = G::kCpr;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  TFFT_WG_STAMP(a, 0);
  uint8_t* const img = lds + kLdsTable;
  const uint32_t img_off = __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t*)img)));
  const uint8_t* const g_tab = lds + lane * 16;
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, x = lane & 15;
  // Image addressing: row r, 16-byte chunk c (of this workgroup's 2 W chunks) lives in super-row r / kRps at slot
  // ((r % kRps) kCpr + c) ^ 2 ((r >> 4) & 7). Transposed read of tile row i_lo by this wave: row r = i_lo + 16 ihi
  // (ihi = 4g + q), chunk 2 wave + (p >> 1), bytes 8 (p & 1) of it; tr_base[h] serves the rows with r % kRps == h.
  const int ihi = 4 * g + q;
  const uint8_t* tr_base[kRps];
#pragma unroll
  for (int h = 0; h < kRps; ++h)
    tr_base[h] = img + (16 / kRps) * ihi * 256 + 16 * ((h * kCpr + 2 * wave + (p >> 1)) ^ (2 * (ihi & 7))) + 8 * (p & 1);
  // cooperative copy-in / copy-out: wave instruction i of this wave covers the 1024 LDS bytes at
  // 8192 wave + 1024 i (4 super-rows); lane -> super-row + (lane >> 4), slot lane & 15
  // Column blocks are counted over (batch entry, column) flattened: for pitch >= 16 W a block is 16 W adjacent
  // columns of one entry; for a narrower pitch (columns-on-lanes form only) it spans 16 W / pitch whole entries,
  // whose rows are then contiguous in memory (N = 256 pitch), so the copy-in still moves full lines.
  const uint32_t pshift = static_cast<uint32_t>(__builtin_ctzll(a.pitch));
  // (four-step forms: a launch may cover a slab of the pass's column blocks only, Args::blk_first / blk_count)
  const uint32_t blk_first = (TW == kTwFourStep) ? a.blk_first : 0u;
  const uint32_t total = (TW == kTwFourStep && a.blk_count) ? a.blk_count
                                                            : static_cast<uint32_t>(((a.tasks / a.groups) << pshift) / G::kCols);

  // (adjacent column blocks run on different CUs at the same time: measured 1-2 % faster than giving each
  // workgroup a contiguous range of blocks)
  Rotor rot(blockIdx.x, gridDim.x);                                      // (block order: k4096::Rotor)
  // copy-in of block blk_in (LDS-DMA, asynchronous): issued for the first block AHEAD of the table fetch below, for every further
  // block at the end of the iteration before it (the loop's increment expression; the image is free by then, barrier D)
  auto copy_in = [&](uint32_t blk_in) {
    const uint64_t gc0 = static_cast<uint64_t>(blk_in + blk_first) * G::kCols;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t sr = 32 * wave + 4 * i + (lane >> 4);
      const uint32_t v = (lane & 15) ^ (2 * (((sr * kRps) >> 4) & 7));
      const uint32_t r = sr * kRps + v / kCpr;
      const uint32_t chunk = v % kCpr;
      const uint64_t gcol = gc0 + 8 * chunk;
      const uint64_t off = (r * a.pitch + static_cast<uint64_t>(r >> a.in_seg_shift) * a.in_seg_gap + (gcol & (a.pitch - 1))) * 2;
      const uint8_t* gr = reinterpret_cast<const uint8_t*>(a.in_re + (gcol >> pshift) * a.in_stride) + off;
      const uint8_t* gi = reinterpret_cast<const uint8_t*>(a.in_im + (gcol >> pshift) * a.in_stride) + off;
      const uint32_t d0 = img_off + 8192 * wave + 1024 * i, d1 = d0 + kPlane;
      uint32_t keep;
      if (NT)
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, off nt\n\t"
            "s_mov_b32 m0, %4\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %2, off nt\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(gr), "v"(gi), "s"(d0), "s"(d1)
            : "memory");
      else
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, off\n\t"
            "s_mov_b32 m0, %4\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %2, off\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(gr), "v"(gi), "s"(d0), "s"(d1)
            : "memory");
    }
  };
  // (Round 5, built behind a macro and A/B-ed in one process: the final 8-wave pass taking the NEXT block through registers,
  // issued behind barrier C under the read-out and the stores as the radix-512 kernel does, instead of this LDS-DMA behind barrier
  // D: 2^26 x 16 +1.6 %, 2^17 x 8192 -0.4 %, single 2^26 -0.3 %, 256-point transforms along a strided axis -1.0 %: not kept.)
  const uint32_t blk0 = rot.item();
  if (blk0 < total) copy_in(blk0);
  // constant tables (G operands to LDS, F operands to registers) BEHIND the first block's loads: a workgroup with one or two blocks
  // (a plan that does not fill the chip; the generations launch shape) would otherwise spend a whole memory latency on its tables
  // before its first data load is issued (2^16 x 1: 16.0 -> 13.6 us, x 64: 25.7 -> 20.7 us, profiles/r4_table_prologue.txt)
  for (int i = tid; i < kLdsTable / 16; i += G::kThreadsW)
    reinterpret_cast<u4*>(lds)[i] = reinterpret_cast<const u4*>(a.tables + kOffG)[i];
  const h8 f_re = *reinterpret_cast<const h8*>(a.tables + kOffF1n + lane * 32);
  const h8 f_im = *reinterpret_cast<const h8*>(a.tables + kOffF1n + lane * 32 + 16);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  auto next_block = [&]() {
    rot.advance();
    const uint32_t nb = rot.item();
    if (nb < total) copy_in(nb);
    return nb;
  };
  for (uint32_t blk = blk0; blk < total; blk = next_block()) {
    const uint64_t gc0 = static_cast<uint64_t>(blk + blk_first) * G::kCols;   // first flattened column of the block
    const uint64_t gcw = gc0 + 16 * wave;                              // ... of this wave
    const uint64_t bidx = gcw >> pshift;                               // this wave's batch entry
    const uint64_t mb = gc0 & (a.pitch - 1);                           // first column of the block (pitch >= 16 W)
    const uint64_t m0 = gcw & (a.pitch - 1);                           // first column of this wave
    // twiddle set-up (table look-ups fly with the copy-in), exactly as in the per-wave kernel
    const uint64_t rest = m0 >> a.ns_f_shift;
    const uint64_t kprev_f0 = m0 - (rest << a.ns_f_shift);
    cpx base[4], step = {1.f, 0.f};
    cpx step4[4];                            // four-step form: one step per column
    if (TW == kTwNext) {
      const uint64_t rest_l = (MODE == kColsOnLanes) ? ((m0 + x) >> a.ns_f_shift) : rest;
      const uint64_t av = rest_l >> a.a_shift;
      step = lookup<true>(a, (av * (a.ns & a.t_mask)) & a.t_mask);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t kb = (MODE == kColsOnLanes) ? (4 * g + r) : x;
        const uint64_t kprev = (MODE == kColsOnLanes)
                                   ? (((m0 + x) - (rest_l << a.ns_f_shift)) >> a.inner_shift)
                                   : ((kprev_f0 + 4 * g + r) >> a.inner_shift);
        base[r] = lookup<true>(a, (av * ((kprev + a.ns * 16 * kb) & a.t_mask)) & a.t_mask);
        base[r].re *= a.tw_scale;
        base[r].im *= a.tw_scale;
      }
    }
    if (TW == kTwFourStep) {
      // w_M^(k col), k = ka + 16 kb (kb = x), col = tw4_col0 + m0 + 4 g + r: base = w_M^(16 x col), step = w_M^col
      // (v_sin / v_cos on exactly reduced exponents: 16 transcendental ops per block instead of 16 scattered 8-byte
      // table loads)
      const float inv_m = 1.0f / static_cast<float>(a.n_mask + 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint64_t col = a.tw4_col0 + m0 + 4 * g + r;
        const float fs = static_cast<float>(col & a.n_mask) * inv_m;
        const float fb = static_cast<float>((col * (16 * x)) & a.n_mask) * inv_m;
        step4[r] = cpx{__builtin_amdgcn_cosf(fs), -__builtin_amdgcn_sinf(fs)};
        base[r] = cpx{__builtin_amdgcn_cosf(fb) * a.tw_scale, -__builtin_amdgcn_sinf(fb) * a.tw_scale};
      }
    }
    TFFT_WG_STAMP(a, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // A: the whole block is in LDS
    TFFT_WG_STAMP(a, 2);

#ifdef TFFT_DEBUG_KERNELS
    if (MODE == kColsInRegs && a.copy_only) {
      // timing experiment (WRONG output): the image goes straight back out through the row stores below
      uint16_t* const c_re = a.out_re + bidx * a.out_stride;
      uint16_t* const c_im = a.out_im + bidx * a.out_stride;
      const uint64_t restb = mb >> a.ns_f_shift;
      const uint64_t obase = ((restb << 8) << a.ns_f_shift) + (mb - (restb << a.ns_f_shift));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t sr = 32 * wave + 4 * i + (lane >> 4);
        const uint32_t v = (lane & 15) ^ (2 * (((sr * kRps) >> 4) & 7));
        const uint32_t k = sr * kRps + v / kCpr;
        const uint32_t chunk = v % kCpr;
        const u4 vr = *reinterpret_cast<const u4*>(img + 8192 * wave + 1024 * i + 16 * lane);
        const u4 vi = *reinterpret_cast<const u4*>(img + kPlane + 8192 * wave + 1024 * i + 16 * lane);
        const uint64_t o = obase + (static_cast<uint64_t>(k) << a.ns_f_shift) + 8 * chunk;
        TFFT_ST_PASS(TW, vr, reinterpret_cast<u4*>(c_re + o));
        TFFT_ST_PASS(TW, vi, reinterpret_cast<u4*>(c_im + o));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      continue;
    }
#endif

    // ---- stage 1
    uint32_t pr[8][4], pi[8][4];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      f4 dre[2], dim[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int i_lo = 2 * t + e;
        const uint8_t* ad = tr_base[i_lo % kRps] + (i_lo / kRps) * 256;
        const s4 xr = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(ad));
        const s4 xi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(ad + kPlane));
        const u4 raw = {__builtin_bit_cast(u2, xr).x, __builtin_bit_cast(u2, xr).y,
                        __builtin_bit_cast(u2, xi).x, __builtin_bit_cast(u2, xi).y};
        const h8 xv = __builtin_bit_cast(h8, raw);
        dre[e] = mfma(f_re, xv);
        dim[e] = mfma(f_im, xv);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pr[t][r] = pk(dre[0][r], dre[1][r]);
        pi[t][r] = pk(dim[0][r], dim[1][r]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // B: every wave has read its slab; the image may be overwritten
    TFFT_WG_STAMP(a, 3);
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        transpose4(pr[0 + pp][r], pr[2 + pp][r], pr[4 + pp][r], pr[6 + pp][r]);
        transpose4(pi[0 + pp][r], pi[2 + pp][r], pi[4 + pp][r], pi[6 + pp][r]);
      }

    uint16_t* const o_re = a.out_re + bidx * a.out_stride;
    uint16_t* const o_im = a.out_im + bidx * a.out_stride;
    cpx pw = {1.f, 0.f};
    float hold_re[4], hold_im[4];
    uint32_t acc_re[4][4], acc_im[4][4];
#pragma unroll
    for (int ka = 0; ka < 16; ++ka) {
      const int aa = ka >> 2, r0 = ka & 3;
      const u4 draw = {pr[2 * aa][r0], pr[2 * aa + 1][r0], pi[2 * aa][r0], pi[2 * aa + 1][r0]};
      const h8 dop = __builtin_bit_cast(h8, draw);
      const u4 graw = *reinterpret_cast<const u4*>(g_tab + ka * 1024);
      f4 e_re, e_im;
      if (MODE == kColsOnLanes) {
        e_re = mfma(__builtin_bit_cast(h8, graw), dop);
        e_im = mfma(im_form(graw), dop);
      } else {
        e_re = mfma(dop, __builtin_bit_cast(h8, graw));
        e_im = mfma(dop, im_form(graw));
      }
      if (TW == kTwNext) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const cpx w = cmul(base[r], pw);
          const float vr = e_re[r] * w.re - e_im[r] * w.im;
          const float vi = e_re[r] * w.im + e_im[r] * w.re;
          e_re[r] = vr;
          e_im[r] = vi;
        }
        pw = cmul(pw, step);
      }
      if (TW == kTwFourStep) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float vr = e_re[r] * base[r].re - e_im[r] * base[r].im;
          const float vi = e_re[r] * base[r].im + e_im[r] * base[r].re;
          e_re[r] = vr;
          e_im[r] = vi;
          base[r] = cmul(base[r], step4[r]);       // next tile: k + 1
        }
      }
      if (MODE == kColsInRegs) {
        // row k = ka + 16 kb (kb = x) of the shared output image, this wave's columns 16 wave + 4g .. + 3:
        // chunk 2 wave + (g >> 1), bytes 8 (g & 1) of it. The OUTPUT image has its own swizzle, slot ^ kb over all four
        // bits of kb: the 16 lanes of a group then hit 16 different slots (with the input image's 2 (kb & 7), lanes kb
        // and kb + 8 collided: PMC showed bank-conflict cycles = 54 % of the LDS-active cycles of this kernel). Within the
        // slot the 8-byte piece of an even / odd g takes the lower / upper half for kb < 8 and the other way round for kb >= 8:
        // a ds_write_b64 is banked modulo 32 dwords over 16-lane groups, so pieces 128 bytes apart (kb, kb + 8) in the SAME
        // half still collided 2-way on every store (all of the remaining SQ_LDS_BANK_CONFLICT of the column kernels in round 2,
        // 20-25 % of their LDS-active cycles); the read-out swaps the halves back for rows with kb >= 8.
        const u2 vr = {pk(e_re[0], e_re[1]), pk(e_re[2], e_re[3])};
        const u2 vi = {pk(e_im[0], e_im[1]), pk(e_im[2], e_im[3])};
        uint8_t* dst = img + ((ka / kRps) + (16 / kRps) * x) * 256 +
                       16 * (((ka % kRps) * kCpr + 2 * wave + (g >> 1)) ^ x) + 8 * ((g & 1) ^ (x >> 3));
        *reinterpret_cast<u2*>(dst) = vr;
        *reinterpret_cast<u2*>(dst + kPlane) = vi;
      } else {
        if ((ka & 1) == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            hold_re[r] = e_re[r];
            hold_im[r] = e_im[r];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc_re[r][(ka >> 1) & 3] = pk(hold_re[r], e_re[r]);
            acc_im[r][(ka >> 1) & 3] = pk(hold_im[r], e_im[r]);
          }
          if ((ka & 7) == 7) {
            const int half = ka >> 3;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const u4 vr = {acc_re[r][0], acc_re[r][1], acc_re[r][2], acc_re[r][3]};
              const u4 vi = {acc_im[r][0], acc_im[r][1], acc_im[r][2], acc_im[r][3]};
              if (STG) {
                // image row 16 wave + x (512 B), 16-byte chunk c = 2 (4g + r) + half at slot c ^ x
                uint8_t* dst = img + 8192 * wave + 512 * x + 16 * ((2 * (4 * g + r) + half) ^ x);
                *reinterpret_cast<u4*>(dst) = vr;
                *reinterpret_cast<u4*>(dst + kPlane) = vi;
              } else {
                const uint64_t o = (m0 + x) * 256 + 16 * (4 * g + r) + 8 * half;
                *reinterpret_cast<u4*>(o_re + o) = vr;
                *reinterpret_cast<u4*>(o_im + o) = vi;
              }
            }
          }
        }
      }
    }
    if (MODE == kColsOnLanes) TFFT_WG_STAMP(a, 4);
    if (MODE == kColsOnLanes && STG) {
      // the slice is private to this wave (it is also exactly the LDS range of this wave's next copy-in): no barrier
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t f = 2 * i + (lane >> 5);                 // column within the wave's 16
        const uint32_t chunk = (lane & 31) ^ f;
        const u4 vr = *reinterpret_cast<const u4*>(img + 8192 * wave + 1024 * i + 16 * lane);
        const u4 vi = *reinterpret_cast<const u4*>(img + kPlane + 8192 * wave + 1024 * i + 16 * lane);
        const uint64_t o = (m0 + f) * 256 + 8 * chunk;
        if (NT) {
          TFFT_ST_PASS(TW, vr, reinterpret_cast<u4*>(o_re + o));
          TFFT_ST_PASS(TW, vi, reinterpret_cast<u4*>(o_im + o));
        } else {
          *reinterpret_cast<u4*>(o_re + o) = vr;
          *reinterpret_cast<u4*>(o_im + o) = vi;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read out before this wave's next copy-in lands on the slice
    }
    if (MODE == kColsInRegs) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();          // C: the output image is complete
      TFFT_WG_STAMP(a, 4);
      const uint64_t restb = mb >> a.ns_f_shift;                 // the block's columns share it (ns_f % (16 W) == 0)
      const uint64_t obase = ((restb << 8) << a.ns_f_shift) + (mb - (restb << a.ns_f_shift));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t sr = 32 * wave + 4 * i + (lane >> 4);
        const uint32_t v = (lane & 15) ^ (((sr * kRps) >> 4) & 15);   // output image: slot ^ kb (see the stage-2 stores)
        const uint32_t k = sr * kRps + v / kCpr;
        const uint32_t chunk = v % kCpr;
        u4 vr = *reinterpret_cast<const u4*>(img + 8192 * wave + 1024 * i + 16 * lane);
        u4 vi = *reinterpret_cast<const u4*>(img + kPlane + 8192 * wave + 1024 * i + 16 * lane);
        if (k & 128) {                                                // kb >= 8: the two 8-byte halves were stored flipped
          vr = u4{vr.z, vr.w, vr.x, vr.y};
          vi = u4{vi.z, vi.w, vi.x, vi.y};
        }
        const uint64_t o = (TW == kTwFourStep)
                               ? (mb - a.out_col0) + a.out_base + (static_cast<uint64_t>(k) << a.out_pitch_shift) +
                                     static_cast<uint64_t>(k >> a.out_seg_shift) * a.out_seg_gap + 8 * chunk
                               : obase + (static_cast<uint64_t>(k) << a.ns_f_shift) + 8 * chunk;
        if (NT) {
          TFFT_ST_PASS(TW, vr, reinterpret_cast<u4*>(o_re + o));
          TFFT_ST_PASS(TW, vi, reinterpret_cast<u4*>(o_im + o));
        } else {
          *reinterpret_cast<u4*>(o_re + o) = vr;
          *reinterpret_cast<u4*>(o_im + o) = vi;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();          // D: read out; the next block's copy-in may overwrite the image
    }
    TFFT_WG_STAMP(a, 5);
  }
#ifdef TFFT_DEBUG_KERNELS
  if (a.wg_times) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (measurement build: exit stamp behind the stores)
#endif
  TFFT_WG_STAMP(a, 6);
}

// ---------------------------------------------------------------------------
// Radix-512 column pass (columns-in-registers form, Ns >= 64): rows i = 2 m + q. The two decimated sequences
// q = 0, 1 are two radix-256 problems, handled by waves 0-3 and 4-7 of the workgroup exactly as the 4-wave
// cooperative kernel above handles one (64 columns, 128-byte row segments, one 32-KiB half-image per plane and
// sequence). The q = 1 half's stage-2 operand is G with the combine twiddle folded in, G'[ka][n1][kb] = w_512^(ka + 16 kb)
// G_ka[n1][kb] (k4096::build_tables, kOffG512): one rounding of a constant instead of an fp32 multiply per output and a
// 16-step recurrence in the half that is the workgroup's critical path. Both halves leave A_q[k][column] in LDS; the read-out
// forms X[k] = A_0 + A_1, X[k + 256] = A_0 - A_1 in fp32, applies the next pass's input twiddles there (they depend
// on the full output index), rounds once and stores two 128-byte row segments per lane group.
// One pass over HBM for a radix the 160-KiB LDS could not hold as one 512-row image of 256-byte segments:
// 2^17 = 256 x 512 and 2^26 = 256 x 512 x 512 take one pass fewer.
// ---------------------------------------------------------------------------
constexpr int kTab512 = 2 * kLdsTable;                                  // G for the even sequence, G with w_512^k folded in for the odd one
constexpr int kWg512LdsBytes = kTab512 + 4 * WgGeom<4>::kPlane;       // 160 KiB


// Next-pass twiddles of the radix-512 read-out: v_sin / v_cos (absolute error ~1e-6, three orders below binary16's
// resolution) instead of the two-level fp32 tables: measured +3-6 % on 2^15 / 2^18 (six dependent loads per chunk less).
#ifndef TFFT_LUT512
#define TFFT_LUT512 0
#endif
constexpr bool kLut512 = TFFT_LUT512;

// SC: multiply the output by Args::comb_scale in fp32 at the read-out (TFFT_SCALE_ONCE with this pass as the plan's last;
// otherwise the combine's 1/2 is part of the constant operands and the combine is a plain sum).
template <int MODE, int TW, bool SC = false, bool PLAIN = false>
__global__ __launch_bounds__(kThreads, 2) void colfft512_wg_kernel(Args a) {
  constexpr bool kPlainAcc = PLAIN;
  static_assert(TW != kTwFourStep || MODE == kColsInRegs, "the four-step twiddle exists for the columns-in-registers form");
  static_assert(!SC || (MODE == kColsInRegs && TW == kTwNone), "the read-out factor exists for a final pass");
  using G = WgGeom<4>;
  constexpr int kHalf = G::kPlane;        // one sequence, one plane: 256 rows x 128 B
  constexpr int kPlaneAll = 2 * kHalf;    // RE -> IM distance
  constexpr int kRps = G::kRps, kCpr = G::kCpr;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seq = wave >> 2, w4 = wave & 3;   // decimated sequence q, wave within its half
  for (int i = tid; i < kLdsTable / 16; i += kThreads) {
    reinterpret_cast<u4*>(lds)[i] = reinterpret_cast<const u4*>(a.tables + kOffG512)[i];
    reinterpret_cast<u4*>(lds + kLdsTable)[i] = reinterpret_cast<const u4*>(a.tables + kOffG512 + kLdsTable)[i];
  }
  const h8 f_re = *reinterpret_cast<const h8*>(a.tables + kOffF1n + lane * 32);
  const h8 f_im = *reinterpret_cast<const h8*>(a.tables + kOffF1n + lane * 32 + 16);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  uint8_t* const img = lds + kTab512;
  uint8_t* const img_q = img + seq * kHalf;
  const uint8_t* const g_tab = lds + seq * kLdsTable + lane * 16;   // this sequence's stage-2 operand
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, x = lane & 15;
  const int ihi = 4 * g + q;
  const uint8_t* tr_base[kRps];
#pragma unroll
  for (int h = 0; h < kRps; ++h)
    tr_base[h] = img_q + (16 / kRps) * ihi * 256 + 16 * ((h * kCpr + 2 * w4 + (p >> 1)) ^ (2 * (ihi & 7))) + 8 * (p & 1);
  const uint32_t pshift = static_cast<uint32_t>(__builtin_ctzll(a.pitch));
  const uint32_t total = static_cast<uint32_t>(((a.tasks / a.groups) << pshift) / G::kCols);
  // Copy-in through registers: lane l of wave instruction i loads the 16 bytes that belong at LDS byte
  // 8192 w4 + 1024 i + 16 l of its sequence's (swizzled) half-image. The loads of block n + 1 are issued behind barrier C of
  // block n (stage 2 has consumed its operands, 64 registers are free) and fly under the fp32 read-out, the stores and
  // barrier D; they are written to LDS at the top of the next iteration. The workgroup fills the CU's LDS, so nothing
  // else can overlap its HBM reads with its arithmetic (the LDS-DMA version started the copy-in after barrier D and
  // waited for it: 42 % of the wave time parked, profiles/r2_c3_pmc_summary.json).
  u4 raw_re[8], raw_im[8];
  auto issue_loads = [&](uint32_t blk) {
    const uint64_t gc0 = static_cast<uint64_t>(blk) * G::kCols;
    const uint64_t bidx = gc0 >> pshift;
    const uint64_t mb = gc0 & (a.pitch - 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t sr = 32 * w4 + 4 * i + (lane >> 4);
      const uint32_t v = (lane & 15) ^ (2 * (((sr * kRps) >> 4) & 7));
      const uint32_t r = sr * kRps + v / kCpr;               // row of the sequence's 256-row image
      const uint32_t chunk = v % kCpr;
      const uint32_t row = 2 * r + seq;
      const uint64_t off = (row * a.pitch + static_cast<uint64_t>(row >> a.in_seg_shift) * a.in_seg_gap + mb + 8 * chunk) * 2;
      raw_re[i] = TFFT_NT_LOAD(reinterpret_cast<const u4*>(reinterpret_cast<const uint8_t*>(a.in_re + bidx * a.in_stride) + off));
      raw_im[i] = TFFT_NT_LOAD(reinterpret_cast<const u4*>(reinterpret_cast<const uint8_t*>(a.in_im + bidx * a.in_stride) + off));
    }
  };
  Rotor rot(blockIdx.x, gridDim.x);                          // (block order: k4096::Rotor)
  if (rot.item() < total) issue_loads(rot.item());

  for (uint32_t blk = rot.item(); blk < total; rot.advance(), blk = rot.item()) {
    const uint64_t gc0 = static_cast<uint64_t>(blk) * G::kCols;
    const uint64_t bidx = gc0 >> pshift;                     // pitch >= 64: one batch entry per block
    const uint64_t mb = gc0 & (a.pitch - 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      *reinterpret_cast<u4*>(img_q + 8192 * w4 + 1024 * i + 16 * lane) = raw_re[i];
      *reinterpret_cast<u4*>(img_q + kPlaneAll + 8192 * w4 + 1024 * i + 16 * lane) = raw_im[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // A: both half-images are in LDS

    // ---- stage 1
    uint32_t pr[8][4], pi[8][4];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      f4 dre[2], dim[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int i_lo = 2 * t + e;
        const uint8_t* ad = tr_base[i_lo % kRps] + (i_lo / kRps) * 256;
        const s4 xr = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(ad));
        const s4 xi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(ad + kPlaneAll));
        const u4 raw = {__builtin_bit_cast(u2, xr).x, __builtin_bit_cast(u2, xr).y,
                        __builtin_bit_cast(u2, xi).x, __builtin_bit_cast(u2, xi).y};
        const h8 xv = __builtin_bit_cast(h8, raw);
        dre[e] = mfma(f_re, xv);
        dim[e] = mfma(f_im, xv);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pr[t][r] = pk(dre[0][r], dre[1][r]);
        pi[t][r] = pk(dim[0][r], dim[1][r]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // B: every wave has read its slab; the images may be overwritten
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        transpose4(pr[0 + pp][r], pr[2 + pp][r], pr[4 + pp][r], pr[6 + pp][r]);
        transpose4(pi[0 + pp][r], pi[2 + pp][r], pi[4 + pp][r], pi[6 + pp][r]);
      }

    // ---- stage 2, combine twiddle of the odd sequence, A_q -> LDS
    //   columns in registers: data as the A operand, lane = kb, registers = 4 adjacent columns; image rows = k
    //   columns on lanes:     data as the B operand, lane = column, registers = kb; image rows = columns (512 B of k)
    float hold_re[4], hold_im[4];
    uint32_t acc_re[4][4], acc_im[4][4];
#pragma unroll
    for (int ka = 0; ka < 16; ++ka) {
      const int aa = ka >> 2, r0 = ka & 3;
      const u4 draw = {pr[2 * aa][r0], pr[2 * aa + 1][r0], pi[2 * aa][r0], pi[2 * aa + 1][r0]};
      const h8 dop = __builtin_bit_cast(h8, draw);
      const u4 graw = *reinterpret_cast<const u4*>(g_tab + ka * 1024);
      f4 e_re, e_im;
      if (MODE == kColsOnLanes) {
        e_re = mfma(__builtin_bit_cast(h8, graw), dop);
        e_im = mfma(im_form(graw), dop);
      } else {
        e_re = mfma(dop, __builtin_bit_cast(h8, graw));
        e_im = mfma(dop, im_form(graw));
      }
      if (MODE == kColsInRegs) {
        const u2 vr = {pk(e_re[0], e_re[1]), pk(e_re[2], e_re[3])};
        const u2 vi = {pk(e_im[0], e_im[1]), pk(e_im[2], e_im[3])};
        uint8_t* dst = img_q + ((ka / kRps) + (16 / kRps) * x) * 256 +
                       16 * (((ka % kRps) * kCpr + 2 * w4 + (g >> 1)) ^ x) + 8 * ((g & 1) ^ (x >> 3));
        *reinterpret_cast<u2*>(dst) = vr;
        *reinterpret_cast<u2*>(dst + kPlaneAll) = vi;
      } else if ((ka & 1) == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          hold_re[r] = e_re[r];
          hold_im[r] = e_im[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc_re[r][(ka >> 1) & 3] = pk(hold_re[r], e_re[r]);
          acc_im[r][(ka >> 1) & 3] = pk(hold_im[r], e_im[r]);
        }
        if ((ka & 7) == 7) {
          const int half = ka >> 3;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // image row 16 w4 + x (512 B: this column's A_q[k]), 16-byte chunk c = 2 (4g + r) + half at slot c ^ x
            uint8_t* dst = img_q + 8192 * w4 + 512 * x + 16 * ((2 * (4 * g + r) + half) ^ x);
            *reinterpret_cast<u4*>(dst) = u4{acc_re[r][0], acc_re[r][1], acc_re[r][2], acc_re[r][3]};
            *reinterpret_cast<u4*>(dst + kPlaneAll) = u4{acc_im[r][0], acc_im[r][1], acc_im[r][2], acc_im[r][3]};
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // C: A_0 and A_1 are complete
    // (issued here, not at the top of the tile: with the loads in flight under the whole tile the pass ran 6 % SLOWER, 1707
    // against 1613 us for 8 GiB, although it then needs fewer registers; the pass already moves data at the rate a plain copy
    // with its 128-byte row segments reaches, 5.3-5.4 TB/s, profiles/r3_stride_pad.txt, and more requests in flight only
    // lengthen the queues)
    if (rot.peek() < total) issue_loads(rot.peek());               // the next block's input starts flying now

    if (MODE == kColsOnLanes) {
      // ---- radix-2 combine at read-out: 16-byte chunks = 8 consecutive k of one column; a column's 512 outputs are
      // 1 KiB contiguous. The next pass's twiddle is w_T^(av k') with av from the column (Ns = 1: kprev = 0).
      uint16_t* const c_re = a.out_re + bidx * a.out_stride;
      uint16_t* const c_im = a.out_im + bidx * a.out_stride;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const uint32_t L = it * kThreads + tid;
        const uint32_t f = L >> 5;                               // column within the block's 64
        const uint32_t k0 = 8 * ((L & 31) ^ (f & 15));
        const h8 ar = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(img + 16 * L));
        const h8 br = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(img + kHalf + 16 * L));
        const h8 ai = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(img + kPlaneAll + 16 * L));
        const h8 bi = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(img + kPlaneAll + kHalf + 16 * L));
        float x0r[8], x0i[8], x1r[8], x1i[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float Ar = static_cast<float>(ar[e]), Br = static_cast<float>(br[e]);
          const float Ai = static_cast<float>(ai[e]), Bi = static_cast<float>(bi[e]);
          x0r[e] = Ar + Br;
          x0i[e] = Ai + Bi;
          x1r[e] = Ar - Br;
          x1i[e] = Ai - Bi;
        }
        if (TW) {
          const uint64_t av = (mb + f) >> a.a_shift;
          const cpx w1 = lookup<kLut512>(a, av & a.t_mask);
          cpx t0 = lookup<kLut512>(a, (av * k0) & a.t_mask);
          t0.re *= a.tw_scale;
          t0.im *= a.tw_scale;
          cpx t1 = cmul(t0, lookup<kLut512>(a, (av * 256) & a.t_mask));
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float r0 = x0r[e] * t0.re - x0i[e] * t0.im, i0 = x0r[e] * t0.im + x0i[e] * t0.re;
            const float r1 = x1r[e] * t1.re - x1i[e] * t1.im, i1 = x1r[e] * t1.im + x1i[e] * t1.re;
            x0r[e] = r0; x0i[e] = i0; x1r[e] = r1; x1i[e] = i1;
            t0 = cmul(t0, w1);
            t1 = cmul(t1, w1);
          }
        }
        const u4 s0r = {pk(x0r[0], x0r[1]), pk(x0r[2], x0r[3]), pk(x0r[4], x0r[5]), pk(x0r[6], x0r[7])};
        const u4 s0i = {pk(x0i[0], x0i[1]), pk(x0i[2], x0i[3]), pk(x0i[4], x0i[5]), pk(x0i[6], x0i[7])};
        const u4 s1r = {pk(x1r[0], x1r[1]), pk(x1r[2], x1r[3]), pk(x1r[4], x1r[5]), pk(x1r[6], x1r[7])};
        const u4 s1i = {pk(x1i[0], x1i[1]), pk(x1i[2], x1i[3]), pk(x1i[4], x1i[5]), pk(x1i[6], x1i[7])};
        const uint64_t o0 = (mb + f) * 512 + k0;
        TFFT_ST_PASS(TW, s0r, reinterpret_cast<u4*>(c_re + o0));
        TFFT_ST_PASS(TW, s0i, reinterpret_cast<u4*>(c_im + o0));
        TFFT_ST_PASS(TW, s1r, reinterpret_cast<u4*>(c_re + o0 + 256));
        TFFT_ST_PASS(TW, s1i, reinterpret_cast<u4*>(c_im + o0 + 256));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();          // D
      continue;
    }

    // ---- radix-2 combine at read-out: this thread takes 16-byte chunks (8 columns) of rows k and k + 256
    const uint64_t o_entry = (bidx >> a.out_sub_shift) * a.out_stride +
                             (bidx & ((1ull << a.out_sub_shift) - 1)) * a.out_sub_stride;
    uint16_t* const o_re = a.out_re + o_entry;
    uint16_t* const o_im = a.out_im + o_entry;
    const uint32_t row_shift = a.ns_f_shift + a.out_row_shift;
    const uint64_t restb = mb >> a.ns_f_shift;                 // shared by the block's 64 columns (ns_f % 64 == 0)
    const uint64_t obase = ((restb << 9) << a.ns_f_shift) + (mb - (restb << a.ns_f_shift));
    cpx w_av = {1.f, 0.f}, w_half = {1.f, 0.f};
    uint64_t av = 0;
    if (TW == kTwNext) {
      av = restb >> a.a_shift;
      w_av = lookup<kLut512>(a, av & a.t_mask);                                            // w_T^av (per unit of kprev)
      w_half = lookup<kLut512>(a, (av * ((a.ns * 256) & a.t_mask)) & a.t_mask);            // w_T^(av ns 256)
    }
    const float inv_m = 1.0f / static_cast<float>(a.n_mask + 1);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const uint32_t L = it * kThreads + tid;                  // 16-byte slot of the half-image
      const uint32_t sr = L >> 4;
      const uint32_t v = (L & 15) ^ (((sr * kRps) >> 4) & 15);   // output image: slot ^ kb
      const uint32_t k = sr * kRps + v / kCpr;
      const uint32_t chunk = v % kCpr;
      u4 a_re = *reinterpret_cast<const u4*>(img + 16 * L);
      u4 b_re = *reinterpret_cast<const u4*>(img + kHalf + 16 * L);
      u4 a_im = *reinterpret_cast<const u4*>(img + kPlaneAll + 16 * L);
      u4 b_im = *reinterpret_cast<const u4*>(img + kPlaneAll + kHalf + 16 * L);
      if (k & 128) {                                                  // kb >= 8: the two 8-byte halves were stored flipped
        a_re = u4{a_re.z, a_re.w, a_re.x, a_re.y};
        b_re = u4{b_re.z, b_re.w, b_re.x, b_re.y};
        a_im = u4{a_im.z, a_im.w, a_im.x, a_im.y};
        b_im = u4{b_im.z, b_im.w, b_im.x, b_im.y};
      }
      const h8 ar = __builtin_bit_cast(h8, a_re), br = __builtin_bit_cast(h8, b_re);
      const h8 ai = __builtin_bit_cast(h8, a_im), bi = __builtin_bit_cast(h8, b_im);
      const uint64_t o0 = obase + (static_cast<uint64_t>(k) << row_shift) + 8 * chunk;
      const uint64_t o1 = o0 + (static_cast<uint64_t>(256) << row_shift);
      if (TW == kTwNone && !SC) {
        // last pass: X = A_0 +- A_1 is the output itself: packed binary16 sums (one correct rounding each, exactly what the
        // fp32 path's sum-then-round gives, in 16 instructions instead of 80)
        TFFT_ST_PASS(TW, __builtin_bit_cast(u4, ar + br), reinterpret_cast<u4*>(o_re + o0));
        TFFT_ST_PASS(TW, __builtin_bit_cast(u4, ai + bi), reinterpret_cast<u4*>(o_im + o0));
        TFFT_ST_PASS(TW, __builtin_bit_cast(u4, ar - br), reinterpret_cast<u4*>(o_re + o1));
        TFFT_ST_PASS(TW, __builtin_bit_cast(u4, ai - bi), reinterpret_cast<u4*>(o_im + o1));
        continue;
      }
      float x0r[8], x0i[8], x1r[8], x1i[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float Ar = static_cast<float>(ar[e]), Br = static_cast<float>(br[e]);
        const float Ai = static_cast<float>(ai[e]), Bi = static_cast<float>(bi[e]);
        x0r[e] = Ar + Br;
        x0i[e] = Ai + Bi;
        x1r[e] = Ar - Br;
        x1i[e] = Ai - Bi;
        if (SC) {
          x0r[e] *= a.comb_scale;
          x0i[e] *= a.comb_scale;
          x1r[e] *= a.comb_scale;
          x1i[e] *= a.comb_scale;
        }
      }
      if (TW == kTwFourStep) {
        // rows k and k + 256 of column c_e = tw4_col0 + mb + 8 chunk + e: t0(e) = w_M^(k c_e), t1(e) = t0(e) w_M^(256 c_e);
        // both run along e as recurrences (steps w_M^k and w_M^256). v_sin / v_cos on exactly reduced exponents.
        const uint64_t c0 = a.tw4_col0 + mb + 8 * chunk;
        auto wm = [&](uint64_t e) {
          const float frac = static_cast<float>(e & a.n_mask) * inv_m;
          return cpx{__builtin_amdgcn_cosf(frac), -__builtin_amdgcn_sinf(frac)};
        };
        cpx t0 = wm(c0 * k), u = wm(c0 * 256);
        t0.re *= a.tw_scale;
        t0.im *= a.tw_scale;
        const cpx sk = wm(k), s256 = wm(256);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const cpx t1 = cmul(t0, u);
          const float r0 = x0r[e] * t0.re - x0i[e] * t0.im, i0 = x0r[e] * t0.im + x0i[e] * t0.re;
          const float r1 = x1r[e] * t1.re - x1i[e] * t1.im, i1 = x1r[e] * t1.im + x1i[e] * t1.re;
          x0r[e] = r0; x0i[e] = i0; x1r[e] = r1; x1i[e] = i1;
          t0 = cmul(t0, sk);
          u = cmul(u, s256);
        }
      }
      if (TW == kTwNext) {
        // E = av (kprev + ns k') mod T; kprev of column e of this chunk = (kprev_f0 + e) >> inner_shift. Both rows' twiddles
        // run along e as recurrences (inner = 1, the only geometry this pass is planned for with a following pass: step
        // w_T^av per column); inner > 1: the chunk's 8 columns share kprev.
        const uint64_t kprev_f0 = (mb + 8 * chunk) - (restb << a.ns_f_shift);
        const cpx row0 = lookup<kLut512>(a, (av * ((a.ns * k) & a.t_mask)) & a.t_mask);
        cpx col = lookup<kLut512>(a, (av * ((kprev_f0 >> a.inner_shift) & a.t_mask)) & a.t_mask);
        col.re *= a.tw_scale;
        col.im *= a.tw_scale;
        cpx t0 = cmul(col, row0), t1 = cmul(t0, w_half);
        const cpx stp = a.inner_shift == 0 ? w_av : cpx{1.f, 0.f};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float r0 = x0r[e] * t0.re - x0i[e] * t0.im, i0 = x0r[e] * t0.im + x0i[e] * t0.re;
          const float r1 = x1r[e] * t1.re - x1i[e] * t1.im, i1 = x1r[e] * t1.im + x1i[e] * t1.re;
          x0r[e] = r0; x0i[e] = i0; x1r[e] = r1; x1i[e] = i1;
          t0 = cmul(t0, stp);
          t1 = cmul(t1, stp);
        }
      }
      const u4 s0r = {pk(x0r[0], x0r[1]), pk(x0r[2], x0r[3]), pk(x0r[4], x0r[5]), pk(x0r[6], x0r[7])};
      const u4 s0i = {pk(x0i[0], x0i[1]), pk(x0i[2], x0i[3]), pk(x0i[4], x0i[5]), pk(x0i[6], x0i[7])};
      const u4 s1r = {pk(x1r[0], x1r[1]), pk(x1r[2], x1r[3]), pk(x1r[4], x1r[5]), pk(x1r[6], x1r[7])};
      const u4 s1i = {pk(x1i[0], x1i[1]), pk(x1i[2], x1i[3]), pk(x1i[4], x1i[5]), pk(x1i[6], x1i[7])};
      TFFT_ST_PASS(TW, s0r, reinterpret_cast<u4*>(o_re + o0));
      TFFT_ST_PASS(TW, s0i, reinterpret_cast<u4*>(o_im + o0));
      TFFT_ST_PASS(TW, s1r, reinterpret_cast<u4*>(o_re + o1));
      TFFT_ST_PASS(TW, s1i, reinterpret_cast<u4*>(o_im + o1));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // D: read out; the next block's copy-in may overwrite the images
  }
}

}  // namespace colfft
