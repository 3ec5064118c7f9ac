// collat.hpp — radix-256 column pass for work that does NOT fill the chip: one transform, or a few (gfx950).
//
// The reference benchmarks exactly this regime: ONE transform per length, 10 warm-up + 100 timed runs, wall clock around
// ComputeFFT (src/testing/benchmarks/FFTBenchSinlge.cu:11-15, Bench.h:121-142; SURVEY 8a: "launch latency at small N"). The
// throughput kernels of colfft.hpp are shaped for thousands of column blocks per launch; with ONE block per workgroup their
// start-up is the whole run. Measured on MI355X (round 5, tools/lat_probe.hip, tools/exp_lat_phases.py, profiles/r5_lat_*):
//   * a dependent launch costs 2.1 us before it does anything (chain of empty kernels in a graph); a plain tile copy of the
//     column pass's access pattern 2.9-3.3 us per pass up to 2^18 and 4.6 us at 2^20; the same passes inside ONE launch behind
//     a grid barrier (write-through stores + sc1 loads, or release / acquire fences) cost 3.7-6 us per pass, i.e. MORE than
//     the launch they replace: the passes stay separate launches;
//   * the 4-wave throughput kernel spends, of 5.8 us in the kernel at 2^16, 2.1-2.7 us before its block is in LDS (kernel
//     arguments -> tables -> wait -> barrier -> twiddle table look-ups -> wait: three dependent memory round trips, and
//     LDS-DMA lands 16 KiB per wave in ~0.65 us where plain loads to registers land a whole 64-KiB tile in 0.9 us), 0.4 us in
//     stage 1 and 2.2 us in stage 2 (one wave per SIMD: the fp32 twiddle products are VALU-issue bound), 0.65 us storing.
// This kernel is the same mathematics and the same LDS images as colfft256_wg_kernel<.., W = 4> (64 columns x 256 rows per
// workgroup, 128-byte row segments), re-cut for latency:
//   * a column group's stage 2 SPLIT over HH waves: wave (cg, h) owns column group cg (16 columns) and 16 / HH of its stage-2
//     tiles (ka in [h 16 / HH, (h + 1) 16 / HH)); all HH waves of a group run the cheap stage 1 (0.5 us, redundantly). Stage 2
//     is bound by instruction ISSUE (a wave64 fp32 instruction holds its SIMD for 4 cycles, ~50 of them per tile with the
//     twiddle products; a second wave on the SAME SIMD buys nothing: the first shape tried, 8 waves on 64 columns, ran stage 2
//     exactly as long as 4 waves did), so the split pays only when it reaches more SIMDs: the workgroup is 4 waves on 4 SIMDs
//     and owns 16 CG columns, CG x HH = 2 x 2 (32 columns, 64-byte row segments) or 1 x 4 (16 columns, 32-byte segments);
//   * ONE memory round trip before the block is in LDS: the block's 64 KiB arrive through registers (8 x 16 B per
//     thread), the constant operands (G table, F fragments) are requested behind them and consumed behind them, and the
//     next pass's twiddles come from v_sin / v_cos (computed while the loads fly) instead of dependent table look-ups;
//   * plain (cached) global accesses: a plan this small lives in L2 / Infinity Cache (tfft.hip cache_policy; nt loads cost
//     +0.5 us per pass at these sizes, write-through stores move the end-of-kernel write-back into the kernel and lose
//     0.2 us, tools/lat_probe policies).
// Selected by tfft.hip launch_col for plain / next-pass-twiddle passes of small work (variant bit
// 1073741824 keeps the throughput kernels). Results are within the library's stated tolerance of the other forms, not
// bit-identical to them (hardware sin / cos twiddles, |error| ~ 1e-6, instead of the two-level fp32 tables).
#pragma once

#include <type_traits>

#include "colfft.hpp"

namespace colfft {

// CG column groups of 16 columns per workgroup, HH waves per column group, PP = 1 or 2 workgroups per block (each takes 16 / PP of
// every column group's stage-2 tiles and writes the rows that come out of them; four-way was built and measured: it loses what
// two-way gains, profiles/r5_lat_shapes.txt)
template <int CG, int HH, int PP = 1>
struct LatGeom {
  static constexpr int kThreads = 64 * CG * HH;
  static constexpr int kCols = 16 * CG;
  static constexpr int kRowBytes = 32 * CG;           // one image row of one plane
  static constexpr int kRps = 256 / kRowBytes;        // rows per 256-byte super-row
  static constexpr int kCpr = 2 * CG;                 // 16-byte chunks per row
  static constexpr int kPlane = 256 * kRowBytes;
  static constexpr int kPieces = kPlane / 16 / kThreads;   // 16-byte pieces per thread and plane
  static constexpr int kLds = kLdsTable + 2 * kPlane; // G (16 KiB) + one image
  static constexpr int kSplit = HH * PP;              // ways a column group's 16 stage-2 tiles are split
  static constexpr int kTiles = 16 / kSplit;          // stage-2 tiles per wave
  static_assert(CG * HH <= 8 && kPieces >= 1 && (HH == 2 || HH == 4) && (PP == 1 || PP == 2) && kTiles >= 2, "shape");
};

template <int MODE, int TW, int CG, int HH, int PP = 1>
__global__ __launch_bounds__(64 * CG * HH, 2) void collat256_kernel(Args a) {
  static_assert(TW == kTwNone || TW == kTwNext, "the latency kernel has no four-step twiddle form");
  using G = LatGeom<CG, HH, PP>;
  constexpr int kPlane = G::kPlane, kRps = G::kRps, kCpr = G::kCpr, kT = G::kThreads, kPieces = G::kPieces;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave % CG, hh = wave / CG;      // column group of 16 columns; which of its waves
  // PP > 1: workgroups b, b + nblk, ... share block b (same XCD when nblk is a multiple of 8: the partner's loads hit its L2);
  // each loads the whole block, runs stage 1 and takes tiles sp = part HH + hh of the kSplit-way split of stage 2
  const uint32_t nblk = gridDim.x / PP;
  const int part = PP == 1 ? 0 : __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x / nblk));
  const int sp = part * HH + hh;

  TFFT_WG_STAMP(a, 0);
  uint8_t* const img = lds + kLdsTable;
  const uint8_t* const g_tab = lds + lane * 16;
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, x = lane & 15;
  // image addressing as in colfft256_wg_kernel: row r, 16-byte chunk c of the block's kCpr chunks lives in super-row r / kRps at
  // slot ((r % kRps) kCpr + c) ^ 2 ((r >> 4) & 7)
  const int ihi = 4 * g + q;
  const uint8_t* tr_base[kRps];
#pragma unroll
  for (int h = 0; h < kRps; ++h)
    tr_base[h] = img + (16 / kRps) * ihi * 256 + 16 * ((h * kCpr + 2 * cg + (p >> 1)) ^ (2 * (ihi & 7))) + 8 * (p & 1);
  const uint32_t pshift = static_cast<uint32_t>(__builtin_ctzll(a.pitch));
  const uint32_t total = static_cast<uint32_t>(((a.tasks / a.groups) << pshift) / G::kCols);

  // ---- a block through registers: piece i of thread t is the 16 bytes at byte 16 (kT i + t) of the plane's (swizzled) image;
  // a wave instruction = 4 super-rows = 1 KiB of whole row segments
  u4 raw_re[kPieces], raw_im[kPieces];
  auto issue_loads = [&](uint32_t blk_in) {
    const uint64_t gc0 = static_cast<uint64_t>(blk_in) * G::kCols;
    const uint64_t bidx = gc0 >> pshift;
    const uint64_t mb = gc0 & (a.pitch - 1);
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
      const uint32_t sr = (kT / 16) * i + (tid >> 4);
      const uint32_t v = (tid & 15) ^ (2 * (((sr * kRps) >> 4) & 7));
      const uint32_t r = sr * kRps + v / kCpr;
      const uint32_t chunk = v % kCpr;
      const uint64_t off = (r * a.pitch + static_cast<uint64_t>(r >> a.in_seg_shift) * a.in_seg_gap + mb + 8 * chunk) * 2;
      raw_re[i] = *reinterpret_cast<const u4*>(reinterpret_cast<const uint8_t*>(a.in_re + bidx * a.in_stride) + off);
      raw_im[i] = *reinterpret_cast<const u4*>(reinterpret_cast<const uint8_t*>(a.in_im + bidx * a.in_stride) + off);
    }
  };
  // ---- next pass's input twiddle on output k = ka + 16 kb of a column: w = base[r] step^(ka - kTiles sp), E = a (kprev + ns k) mod T
  // exactly as in colfft256_wg_kernel, with v_sin / v_cos (revolutions) in place of the table look-ups and the wave's first tile
  // (ka = kTiles sp) folded into the base
  cpx base[4], step = {1.f, 0.f};
  auto twiddle_setup = [&](uint32_t blk_in) {
    if (TW != kTwNext) return;
    const uint64_t m0 = (static_cast<uint64_t>(blk_in) * G::kCols + 16 * cg) & (a.pitch - 1);
    const uint64_t rest = m0 >> a.ns_f_shift;
    const uint64_t kprev_f0 = m0 - (rest << a.ns_f_shift);
    const uint64_t rest_l = (MODE == kColsOnLanes) ? ((m0 + x) >> a.ns_f_shift) : rest;
    const uint64_t av = rest_l >> a.a_shift;
    step = lookup<false>(a, (av * (a.ns & a.t_mask)) & a.t_mask);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t kb = (MODE == kColsOnLanes) ? (4 * g + r) : x;
      const uint64_t kprev = (MODE == kColsOnLanes) ? (((m0 + x) - (rest_l << a.ns_f_shift)) >> a.inner_shift)
                                                    : ((kprev_f0 + 4 * g + r) >> a.inner_shift);
      base[r] = lookup<false>(a, (av * ((kprev + a.ns * (16 * kb + G::kTiles * sp)) & a.t_mask)) & a.t_mask);
      base[r].re *= a.tw_scale;
      base[r].im *= a.tw_scale;
    }
  };

  Rotor rot(blockIdx.x - part * nblk, nblk);                             // (block order: k4096::Rotor)
  uint32_t blk = rot.item();
  if (blk >= total) return;
  issue_loads(blk);
  // constant operands behind the block's loads (same queue: they land behind them), consumed behind them
  constexpr int kTabPieces = kLdsTable / 16 / kT;
  u4 tab[kTabPieces];
#pragma unroll
  for (int i = 0; i < kTabPieces; ++i) tab[i] = *reinterpret_cast<const u4*>(a.tables + kOffG + 16 * (tid + kT * i));
  const h8 f_re = *reinterpret_cast<const h8*>(a.tables + kOffF1n + lane * 32);
  const h8 f_im = *reinterpret_cast<const h8*>(a.tables + kOffF1n + lane * 32 + 16);
  twiddle_setup(blk);
  TFFT_WG_STAMP(a, 1);
#pragma unroll
  for (int i = 0; i < kTabPieces; ++i) reinterpret_cast<u4*>(lds)[tid + kT * i] = tab[i];

  for (;;) {
    const uint64_t gc0 = static_cast<uint64_t>(blk) * G::kCols;        // first flattened column of the block
    const uint64_t bidx = gc0 >> pshift;                               // its batch entry (pitch >= 64: one per block)
    const uint64_t mb = gc0 & (a.pitch - 1);                           // first column of the block
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
      *reinterpret_cast<u4*>(img + 16 * (kT * i + tid)) = raw_re[i];
      *reinterpret_cast<u4*>(img + kPlane + 16 * (kT * i + tid)) = raw_im[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // A: the whole block (and, first time round, G) is in LDS
    TFFT_WG_STAMP(a, 2);

    // ---- stage 1 (all HH waves of a column group, redundantly)
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

    // ---- stage 2: this wave's tiles ka = kTiles sp + kk. The register arrays are indexed with compile-time constants only
    // (a runtime index would send them to scratch), hence one unrolled copy per sp behind a wave-uniform branch.
    cpx pw = {1.f, 0.f};
    float hold_re[4], hold_im[4];
    uint32_t acc_re[4][4], acc_im[4][4];
    auto tile2 = [&](const int ka) {
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
      if (MODE == kColsInRegs) {
        // row k = ka + 16 kb (kb = x) of the shared output image, columns 16 cg + 4 g .. + 3 (colfft256_wg_kernel's layout:
        // slot ^ kb, the 8-byte halves flipped for kb >= 8)
        const u2 vr = {pk(e_re[0], e_re[1]), pk(e_re[2], e_re[3])};
        const u2 vi = {pk(e_im[0], e_im[1]), pk(e_im[2], e_im[3])};
        uint8_t* dst = img + ((ka / kRps) + (16 / kRps) * x) * 256 +
                       16 * (((ka % kRps) * kCpr + 2 * cg + (g >> 1)) ^ x) + 8 * ((g & 1) ^ (x >> 3));
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
        }
      }
    };
    constexpr int kTl = G::kTiles, kSp = G::kSplit;
    // the tiles of split index S0, then (columns on lanes) their outputs into the image: column 16 cg + x is image row 16 cg + x
    // (512 B per column); its outputs k = 16 (4 g + r) + ka are 32 bytes per (g, r), this wave's share of them the 32 / kSplit
    // bytes at 2 kTl S0: 16-byte chunk c = 2 (4 g + r) + (that offset >> 4) at slot c ^ x
    auto run_split = [&](auto s0_tag) {
      constexpr int S0 = decltype(s0_tag)::value;
#pragma unroll
      for (int kk = 0; kk < kTl; ++kk) tile2(S0 * kTl + kk);
      if (MODE == kColsOnLanes) {
        constexpr int kByte = 2 * kTl * S0;                     // offset inside the 32 bytes of one (g, r)
        constexpr int j0 = (kByte & 15) / 4;                    // first of the kTl / 2 acc entries ((ka >> 1) & 3) this wave filled
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          uint8_t* dst = img + 8192 * cg + 512 * x + 16 * ((2 * (4 * g + r) + (kByte >> 4)) ^ x) + (kByte & 15);
          if (kTl == 8) {
            *reinterpret_cast<u4*>(dst) = u4{acc_re[r][0], acc_re[r][1], acc_re[r][2], acc_re[r][3]};
            *reinterpret_cast<u4*>(dst + kPlane) = u4{acc_im[r][0], acc_im[r][1], acc_im[r][2], acc_im[r][3]};
          } else if (kTl == 4) {
            *reinterpret_cast<u2*>(dst) = u2{acc_re[r][j0], acc_re[r][j0 + 1]};
            *reinterpret_cast<u2*>(dst + kPlane) = u2{acc_im[r][j0], acc_im[r][j0 + 1]};
          } else {
            *reinterpret_cast<uint32_t*>(dst) = acc_re[r][j0];
            *reinterpret_cast<uint32_t*>(dst + kPlane) = acc_im[r][j0];
          }
        }
      }
    };
#define TFFT_LAT_SPLIT(S0) \
  if (kSp > S0 && sp == S0) run_split(std::integral_constant<int, (S0 < kSp ? S0 : 0)>{});
    TFFT_LAT_SPLIT(0) TFFT_LAT_SPLIT(1) TFFT_LAT_SPLIT(2) TFFT_LAT_SPLIT(3)
    TFFT_LAT_SPLIT(4) TFFT_LAT_SPLIT(5) TFFT_LAT_SPLIT(6) TFFT_LAT_SPLIT(7)
#undef TFFT_LAT_SPLIT
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // C: the output image is complete
    TFFT_WG_STAMP(a, 4);

    // ---- read-out by all waves: full rows, 16 bytes per lane
    uint16_t* const o_re = a.out_re + bidx * a.out_stride;
    uint16_t* const o_im = a.out_im + bidx * a.out_stride;
    if (MODE == kColsOnLanes) {
      // the image is the block's output as it lies in memory: 16 CG columns x 512 contiguous bytes per plane, chunk c of column
      // f (within its group of 16) at slot c ^ f. PP = 2: this workgroup made the chunks of parity `part`
#pragma unroll
      for (int i = 0; i < kPieces; ++i) {
        const uint32_t col = (kT / 32) * i + (tid >> 5);          // column within the block
        const uint32_t chunk = (tid & 31) ^ (col & 15);
        const uint64_t o = (mb + col) * 256 + 8 * chunk;
        if (PP == 1) {
          const u4 vr = *reinterpret_cast<const u4*>(img + 16 * (kT * i + tid));
          const u4 vi = *reinterpret_cast<const u4*>(img + kPlane + 16 * (kT * i + tid));
          *reinterpret_cast<u4*>(o_re + o) = vr;
          *reinterpret_cast<u4*>(o_im + o) = vi;
        } else if ((chunk & 1) == static_cast<uint32_t>(part)) {
          const u4 vr = *reinterpret_cast<const u4*>(img + 16 * (kT * i + tid));
          const u4 vi = *reinterpret_cast<const u4*>(img + kPlane + 16 * (kT * i + tid));
          *reinterpret_cast<u4*>(o_re + o) = vr;
          *reinterpret_cast<u4*>(o_im + o) = vi;
        }
      }
    } else {
      const uint64_t restb = mb >> a.ns_f_shift;                 // the block's columns share it (ns_f % (16 CG) == 0)
      const uint64_t obase = ((restb << 8) << a.ns_f_shift) + (mb - (restb << a.ns_f_shift));
#pragma unroll
      for (int i = 0; i < kPieces; ++i) {
        const uint32_t sr = (kT / 16) * i + (tid >> 4);
        const uint32_t v = (tid & 15) ^ (((sr * kRps) >> 4) & 15);   // output image: slot ^ kb (see the stage-2 stores)
        const uint32_t k = sr * kRps + v / kCpr;
        const uint32_t chunk = v % kCpr;
        // (PP > 1: rows k = ka + 16 kb with ka in this workgroup's 16 / PP tiles; the other rows of the image were never written)
        if (PP == 1 || ((k & 15) * PP) >> 4 == static_cast<uint32_t>(part)) {
          u4 vr = *reinterpret_cast<const u4*>(img + 16 * (kT * i + tid));
          u4 vi = *reinterpret_cast<const u4*>(img + kPlane + 16 * (kT * i + tid));
          if (k & 128) {                                                // kb >= 8: the two 8-byte halves were stored flipped
            vr = u4{vr.z, vr.w, vr.x, vr.y};
            vi = u4{vi.z, vi.w, vi.x, vi.y};
          }
          const uint64_t o = obase + (static_cast<uint64_t>(k) << a.ns_f_shift) + 8 * chunk;
          *reinterpret_cast<u4*>(o_re + o) = vr;
          *reinterpret_cast<u4*>(o_im + o) = vi;
        }
      }
    }
    TFFT_WG_STAMP(a, 5);
    rot.advance();
    blk = rot.item();
    if (blk >= total) break;
    issue_loads(blk);
    twiddle_setup(blk);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // D: read out; the next block may overwrite the image
  }
#ifdef TFFT_DEBUG_KERNELS
  if (a.wg_times) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (measurement build: exit stamp behind the stores)
#endif
  TFFT_WG_STAMP(a, 6);
}

}  // namespace colfft
