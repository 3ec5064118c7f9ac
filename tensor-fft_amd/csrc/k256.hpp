// k256.hpp — batched N = 256 fp16 C2C FFT for gfx950, 16 transforms per wave.
//
// Replaces TensorFFT256 (reference src/base/TensorFFT256.cu:20-306: one warp per 256 points, digit-reversal
// gather, DFT-16, twiddle + transpose through shared memory, second DFT-16) for plans whose length is exactly
// 256 (CreatePlan's smallest size, Plan.h:92-96). Same two radix-16 stages, result DFT(x)/256:
//
//   n = n0 + 16 n1,  k = k0 + 16 k1:   X[k0 + 16 k1] = sum_n0 w256^(n0 k0) w16^(n0 k1) sum_n1 x[n0 + 16 n1] w16^(n1 k0)
//
// One wave owns 16 transforms (tiles); inside a tile the 16 x 16 matrix [n1][n0] sits in LDS exactly as in
// memory (rows of 32 bytes), so HBM -> LDS is one fully contiguous 1-KiB global_load_lds per transform and plane
// pair, and no swizzle is needed: the transposed read of a tile touches 8 consecutive rows per 32-lane half.
//   stage 1  data as the A operand (rows n0, slots n1), F = w16^(n1 k0)/16 as B  ->  D1[n0 = 4g + r][k0 = lane & 15]
//            n0 lands on (lane >> 4, register): exactly the contraction slots of stage 2, so there is no
//            exchange step at all; w256^(n0 k0) is 4 fp32 constants per lane (the same for every tile).
//   stage 2  data as A again (rows k0, slots n0), F as B  ->  D2[k0 = 4g + r][k1 = lane & 15]
//            a lane ends with 4 consecutive outputs k0 = 4g..4g+3 of k = k0 + 16 k1: 8-byte pieces that tile the
//            transform's 512-byte plane exactly; they are staged through the transform's LDS slot and leave as
//            one 1-KiB non-temporal row per transform.
// No constant tables in LDS (F lives in 8 VGPRs), no inter-wave synchronisation.
#pragma once

#include "k4096.hpp"

namespace k256 {

using namespace k4096;

constexpr int kLdsBytes = kWavesPerBlock * kLdsWaveBytes;   // 128 KiB: 8 waves x 16 transforms x 1 KiB
constexpr int kFftsPerWave = 16;

// in_*/out_*: planar binary16; transform b at +b*stride halves. tables: k4096::build_tables() blob
// (uses the natural-order F operand forms and the w256 twiddle block).
template <bool OTW = false>
__global__ __launch_bounds__(kThreads, 2) void fft256_kernel(const uint16_t* in_re, const uint16_t* in_im,
                                                             uint16_t* out_re, uint16_t* out_im, Addr in_map,
                                                             Addr out_map, uint32_t batch, uint32_t live,
                                                             const uint8_t* __restrict__ tables, OutTw otw) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // B operand forms of F: B[slot (g, j) = contraction index 4g + j][col = lane & 15]: F is symmetric, so the
  // A-operand images built for the other kernels serve unchanged.
  const h8 f_re = *reinterpret_cast<const h8*>(tables + kOffF1n + lane * 32);
  const h8 f_im = *reinterpret_cast<const h8*>(tables + kOffF1n + lane * 32 + 16);
  // w256^(n0 k0), n0 = 4g + r, k0 = lane & 15: the same numbers as the 4096 kernel's w256^(n0 k1) block
  const f4 tw_re = *reinterpret_cast<const f4*>(tables + kOffTw + lane * 32);
  const f4 tw_im = *reinterpret_cast<const f4*>(tables + kOffTw + lane * 32 + 16);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  uint8_t* const wl = lds + wave * kLdsWaveBytes;
  const uint32_t wl_off = __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t*)wl)));
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, x = lane & 15;
  // LDS image of tile t: [RE 512 B | IM 512 B] at t * 1024, row n1 at 32 n1. Transposed read: lane 16g + 4q + p
  // supplies row n1 = 4g + q, columns n0 = 4p..4p+3.
  const uint8_t* const tr_base = wl + 32 * (4 * g + q) + 8 * p;
  // copy-in: lanes 0-31 carry the RE plane of a transform (16 bytes each), lanes 32-63 its IM plane
  const uint16_t* const in_plane = (lane < 32) ? in_re : in_im;
  const uint32_t in_lane = 8 * (lane & 31);
  // output: lane (k1 = x, g) writes k = 4g..4g+3 + 16 k1: halves 16 x + 4 g
  const uint32_t out_lane = 16 * x + 4 * g;
  // Staging position of that 8-byte piece inside the plane's 512-byte image. Plain (byte 32 x + 8 g) the 16 lanes of a
  // ds_write_b64 group sit 32 bytes apart: x, x + 4, x + 8, x + 12 on one bank, 4-way (round 5 PMC: SQ_LDS_BANK_CONFLICT = 60 % of
  // the LDS-active cycles of this kernel, profiles/r5_n256_pmc_summary.json). The 16-byte half of the unit is flipped with bit 2 of
  // x and the 8-byte half inside it with bit 3: four different banks; the read-out below undoes both.
  // Measured in one process (profiles/r5_ab_k256_stage.txt): the conflicts go, the kernel gets 1.1 % SLOWER here (354.2 against
  // 350.3 us per 2^31 bytes: the extra read-out swap costs more than the conflicts did under the memory wait), while k256r.hpp
  // gains 0.3-0.6 % from the same layout and keeps it. So this kernel stays on the plain layout; the knob keeps the experiment.
#ifndef TFFT_K256_SWIZZLED_STAGE
  const uint32_t stage_off = 2 * out_lane;
#else
  const uint32_t stage_off = 32u * x + 16u * ((g >> 1) ^ ((x >> 2) & 1)) + 8u * ((g & 1) ^ ((x >> 3) & 1));
#endif

  const uint32_t groups = (batch + kFftsPerWave - 1) / kFftsPerWave;
  // (live: waves of a workgroup that take groups, 8 or fewer for a small batch: k4096.hpp, tfft.hip live_waves())
  for (uint32_t grp = static_cast<uint32_t>(wave) < live ? blockIdx.x * live + wave : groups; grp < groups; grp += gridDim.x * live) {
    const uint32_t b0 = grp * kFftsPerWave;
    const uint32_t nb = (batch - b0 < kFftsPerWave) ? (batch - b0) : kFftsPerWave;   // ragged last group
#pragma unroll
    for (int t = 0; t < kFftsPerWave; ++t) {
      // transforms past the end of the batch re-read the last valid one (their results are not stored)
      const uint32_t b = b0 + (static_cast<uint32_t>(t) < nb ? t : nb - 1);
      const uint16_t* src = in_plane + in_map.off(b) + in_lane;
      const uint32_t d = wl_off + t * 1024;
      uint32_t keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %2\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %1, off nt\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(src), "s"(d)
          : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#pragma unroll
    for (int t = 0; t < kFftsPerWave; ++t) {
      const uint8_t* ad = tr_base + t * 1024;
      const s4 xr = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(ad));
      const s4 xi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(ad + 512));
      const u4 raw = {__builtin_bit_cast(u2, xr).x, __builtin_bit_cast(u2, xr).y, __builtin_bit_cast(u2, xi).x,
                      __builtin_bit_cast(u2, xi).y};
      const h8 a1 = __builtin_bit_cast(h8, raw);
      // stage 1: rows n0 (this operand's lane & 15), slots n1  ->  D1[n0 = 4g + r][k0 = lane & 15]
      const f4 d_re = mfma(a1, f_re);
      const f4 d_im = mfma(a1, f_im);
      f4 t_re, t_im;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        t_re[r] = __builtin_fmaf(d_re[r], tw_re[r], -(d_im[r] * tw_im[r]));
        t_im[r] = __builtin_fmaf(d_re[r], tw_im[r], d_im[r] * tw_re[r]);
      }
      const u4 raw2 = {pk(t_re[0], t_re[1]), pk(t_re[2], t_re[3]), pk(t_im[0], t_im[1]), pk(t_im[2], t_im[3])};
      const h8 a2 = __builtin_bit_cast(h8, raw2);
      // stage 2: rows k0, slots n0  ->  D2[k0 = 4g + r][k1 = lane & 15]
      f4 o_re = mfma(a2, f_re);
      f4 o_im = mfma(a2, f_im);
      if (OTW) {                       // transposed-input plan: output k = 4g + r + 16 k1 of row (b0 + t) & row_mask times w_N^(row k)
        const uint32_t row = (b0 + static_cast<uint32_t>(t)) & otw.row_mask;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float vr = o_re[r], vi = o_im[r];
          otw_apply(otw, row, 4u * g + r + 16u * x, vr, vi);
          o_re[r] = vr;
          o_im[r] = vi;
        }
      }
      // stage the spectrum in the transform's own (consumed) 1-KiB slot [RE 512 B | IM 512 B] ...
      const u2 vr = {pk(o_re[0], o_re[1]), pk(o_re[2], o_re[3])};
      const u2 vi = {pk(o_im[0], o_im[1]), pk(o_im[2], o_im[3])};
      *reinterpret_cast<u2*>(wl + t * 1024 + stage_off) = vr;
      *reinterpret_cast<u2*>(wl + t * 1024 + 512 + stage_off) = vi;
    }
    // ... and store it as one 1-KiB row per transform: lanes 0-31 the RE plane, lanes 32-63 the IM plane, 16-byte
    // non-temporal stores (+2-4 % over 8-byte pieces straight from registers)
#pragma unroll
    for (int t = 0; t < kFftsPerWave; ++t) {
      u4 v = *reinterpret_cast<const u4*>(wl + t * 1024 + 16 * lane);
#ifndef TFFT_K256_SWIZZLED_STAGE
      const uint32_t chunk = lane & 31;
#else
      // lane l of a plane holds unit x = l >> 1: the 16-byte halves of the unit trade places where bit 2 of x is set, the 8-byte
      // halves inside where bit 3 is
      const uint32_t chunk = (lane & 31) ^ ((lane >> 3) & 1);
      if ((lane >> 4) & 1) v = u4{v.z, v.w, v.x, v.y};
#endif
      if (static_cast<uint32_t>(t) < nb) {
        uint16_t* dst = ((lane < 32) ? out_re : out_im) + out_map.off(b0 + t) + 8 * chunk;
        if (OTW) *reinterpret_cast<u4*>(dst) = v;      // intermediate of a transposed-input plan: stays in the Infinity Cache for the column pass
        else __builtin_nontemporal_store(v, reinterpret_cast<u4*>(dst));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // image consumed before the next copy-in lands on it
  }
}

}  // namespace k256
