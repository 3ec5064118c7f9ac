// colfft1024.hpp — radix-1024 pass along a strided axis (gfx950), same autosort indexing as colfft.hpp.
//
// Rows i = 4 m + q: four decimated sequences = four radix-256 problems. Waves 0-3 of the workgroup (seq 0) take q = 0
// and then q = 2, waves 4-7 (seq 1) take q = 1 and then q = 3, each round exactly as one half of the radix-512 kernel
// (64 columns, 128-byte row segments, one 32-KiB half-image per plane and wave group). The combine twiddle w_1024^(q k) is
// part of the stage-2 operand: G_q[ka][n1][kb] = w_1024^(q (ka + 16 kb)) G_ka[n1][kb] / 4 (k4096::build_tables, kOffG1024;
// the 1/4 is the radix-4 combine's share of sequential scaling), one 16-KiB table per wave group and round, re-loaded by
// LDS-DMA from L2 while stage 1 runs. B_q[k] = w_1024^(q k) A_q[k] of the first round stays in REGISTERS (64 VGPRs of packed binary16 per lane: stage 2 leaves element (k, column) in the
// same lane and register in both rounds), so the second round's epilogue forms lane-locally
//     seq 0:  E_0 = B_0 + B_2,  E_1 = B_0 - B_2          seq 1:  O_0 = B_1 + B_3,  D_1 = B_1 - B_3
// E_0 / O_0 go to the LDS image and are read out as X[k] = E_0 + O_0, X[k + 512] = E_0 - O_0; E_1 / D_1 (in the
// registers that held B) follow through the same image: X[k + 256] = E_1 - i D_1, X[k + 768] = E_1 + i D_1.
// A tile is 64 columns x 1024 rows = 256 KiB of input behind a 128-KiB image (a 1024-row image of 128-byte segments
// would not fit the CU's 160 KiB): 2^19 = 512 x 1024 and 2^20 = 1024 x 1024 take two passes over HBM in natural order
// instead of three, 2^28 .. 2^30 three instead of four.
// Stands where the reference runs one TensorRadix16 launch per radix-16 level (src/base/TensorRadix16.cu:36-214,
// src/base/ComputeFFT.h:101-121): 2.5 of its levels per launch.
#pragma once

#include "colfft.hpp"

namespace colfft {

constexpr int kTab1024 = 2 * 16384;                                  // one G_q per wave group
constexpr int kWg1024LdsBytes = kTab1024 + 4 * WgGeom<4>::kPlane;   // 160 KiB

// SC: multiply the output by Args::comb_scale in fp32 at the read-out (TFFT_SCALE_ONCE with this pass as the plan's last:
// the single 1/N cannot ride on the binary16 operand G_q).
template <int MODE, int TW, bool SC = false, bool PLAIN = false>
__global__ __launch_bounds__(kThreads, 2) void colfft1024_wg_kernel(Args a) {
  constexpr bool kPlainAcc = PLAIN;
  static_assert(TW != kTwFourStep, "the four-step twiddle exists for the radix-256 and radix-512 passes");
  static_assert(!SC || (MODE == kColsInRegs && TW == kTwNone), "the read-out factor exists for a final pass");
  using G = WgGeom<4>;
  constexpr int kHalf = G::kPlane;        // one wave group, one plane: 256 rows x 128 B
  constexpr int kPlaneAll = 2 * kHalf;    // RE -> IM distance
  constexpr int kRps = G::kRps, kCpr = G::kCpr;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seq = wave >> 2, w4 = wave & 3;   // wave group (parity of q), wave within it
  h8 f_re = *reinterpret_cast<const h8*>(a.tables + kOffF1n + lane * 32);
  h8 f_im = *reinterpret_cast<const h8*>(a.tables + kOffF1n + lane * 32 + 16);
  // G_q of this wave group's current sequence: 16 KiB at lds + 16384 seq, four LDS-DMA instructions per wave
  const uint32_t tab_off = __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t*)lds)) + 16384 * seq + 4096 * w4);
  auto dma_table = [&](int qn) {
    const uint8_t* src = a.tables + kOffG1024 + 16384 * qn + 4096 * w4 + 16 * lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint8_t* gp = src + 1024 * i;
      const uint32_t d0 = tab_off + 1024 * i;
      uint32_t keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %2\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %1, off\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(gp), "s"(d0)
          : "memory");
    }
  };
  dma_table(seq);
  // (operands: the constants are in registers here and their loads have landed, see colfft512_wg_kernel)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(f_re), "+v"(f_im) : : "memory");
  __syncthreads();

  uint8_t* const img = lds + kTab1024;
  uint8_t* const img_q = img + seq * kHalf;
  const uint8_t* const g_tab = lds + 16384 * seq + lane * 16;
  const int g = lane >> 4, q4 = (lane >> 2) & 3, p = lane & 3, x = lane & 15;
  const int ihi = 4 * g + q4;
  const uint8_t* tr_base[kRps];
#pragma unroll
  for (int h = 0; h < kRps; ++h)
    tr_base[h] = img_q + (16 / kRps) * ihi * 256 + 16 * ((h * kCpr + 2 * w4 + (p >> 1)) ^ (2 * (ihi & 7))) + 8 * (p & 1);
  const uint32_t pshift = static_cast<uint32_t>(__builtin_ctzll(a.pitch));
  const uint32_t total = static_cast<uint32_t>(((a.tasks / a.groups) << pshift) / G::kCols);
  constexpr float kInv1024 = 1.0f / 1024.0f;

  // Copy-in. Lane l of wave instruction i moves the 16 bytes that belong at LDS byte 8192 w4 + 1024 i + 16 l of its group's
  // (swizzled) half-image; round rd reads rows 4 r + 2 rd + seq.
  //   round 0 of block n + 1: through registers, issued behind barrier C of block n (the image is busy until both read-outs
  //   are done), so the loads fly under the read-outs; written to LDS at the top of the next block.
  //   round 1: LDS-DMA straight into the image, issued behind barrier B of round 0 (stage 2 of the first round writes
  //   registers only, so the image is free), flying under that stage 2; no registers (stage 2 holds B_q next to its own
  //   operands: a second register-staged copy does not fit 256 VGPRs).
  u4 ra_re[8], ra_im[8];
  const uint32_t img_off = __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t*)img_q)));
  auto row_offset = [&](uint32_t blk, int i, int rd, int lane) {
    const uint64_t mb = (static_cast<uint64_t>(blk) * G::kCols) & (a.pitch - 1);
    const uint32_t sr = 32 * w4 + 4 * i + (lane >> 4);
    const uint32_t v = (lane & 15) ^ (2 * (((sr * kRps) >> 4) & 7));
    const uint32_t r = sr * kRps + v / kCpr;               // row of the group's 256-row image
    const uint32_t chunk = v % kCpr;
    return ((4 * r + 2 * rd + seq) * a.pitch + mb + 8 * chunk) * 2;
  };
  // (the opaque copies of the lane / thread index below keep the compiler from hoisting every per-lane address of the
  // loop body out of the loop, where they would sit in ~100 registers across all phases)
  auto issue_loads = [&](uint32_t blk) {
    int ll = lane;
    asm volatile("" : "+v"(ll));
    const uint64_t bidx = (static_cast<uint64_t>(blk) * G::kCols) >> pshift;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint64_t off = row_offset(blk, i, 0, ll);
      ra_re[i] = TFFT_NT_LOAD(reinterpret_cast<const u4*>(reinterpret_cast<const uint8_t*>(a.in_re + bidx * a.in_stride) + off));
      ra_im[i] = TFFT_NT_LOAD(reinterpret_cast<const u4*>(reinterpret_cast<const uint8_t*>(a.in_im + bidx * a.in_stride) + off));
    }
  };
  auto dma_round1 = [&](uint32_t blk) {
    int ll = lane;
    asm volatile("" : "+v"(ll));
    const uint64_t bidx = (static_cast<uint64_t>(blk) * G::kCols) >> pshift;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint64_t off = row_offset(blk, i, 1, ll);
      const uint8_t* gr = reinterpret_cast<const uint8_t*>(a.in_re + bidx * a.in_stride) + off;
      const uint8_t* gi = reinterpret_cast<const uint8_t*>(a.in_im + bidx * a.in_stride) + off;
      const uint32_t d0 = img_off + 8192 * w4 + 1024 * i, d1 = d0 + kPlaneAll;
      uint32_t keep;
      if (kPlainAcc)
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
      else
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
      __builtin_amdgcn_sched_barrier(0);       // one address pair at a time (16 pairs up front would cost 64 registers)
    }
  };
  auto to_image = [&]() {
    int ll = lane;
    asm volatile("" : "+v"(ll));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      *reinterpret_cast<u4*>(img_q + 8192 * w4 + 1024 * i + 16 * ll) = ra_re[i];
      *reinterpret_cast<u4*>(img_q + kPlaneAll + 8192 * w4 + 1024 * i + 16 * ll) = ra_im[i];
    }
  };
  // (block order: k4096::Rotor. An XCD-aware order, 32 adjacent blocks per XCD at a time, changed nothing: +0.9 % at 2^20,
  // -0.3 % at 2^25)
  Rotor rot(blockIdx.x, gridDim.x);
  const uint32_t bid0 = rot.item();
#ifdef TFFT_DEBUG_KERNELS
  if (a.wg_times && tid == 0 && blockIdx.x < 8192) a.wg_times[blockIdx.x] = wall_clock64();
#endif
  if (bid0 < total) issue_loads(bid0);

  // B_q of the first round, then E_1 / D_1. Columns in registers: [ka >> 1][2 (ka & 1) + {0, 1}] = columns {0,1}, {2,3} of
  // tile ka; columns on lanes: [ka >> 1][r] = rows (ka - 1, ka) of register r.
  uint32_t sv_re[8][4], sv_im[8][4];

  for (uint32_t blk = bid0; blk < total; rot.advance(), blk = rot.item()) {
    const uint64_t gc0 = static_cast<uint64_t>(blk) * G::kCols;
    const uint64_t bidx = gc0 >> pshift;                     // pitch >= 64: one batch entry per block
    const uint64_t mb = gc0 & (a.pitch - 1);
    to_image();

#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      if (rd == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();            // A: both half-images of this round are in LDS
      if (rd == 1) dma_table(2 + seq);         // every wave is through stage 2 of the first round: G_q of the second flies under stage 1

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
      if (rd == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();            // B: every wave has read its slab; the images may be overwritten (and G_q is in place)
      if (rd == 0) dma_round1(blk);            // stage 2 of the first round writes registers only
#pragma unroll
      for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          transpose4(pr[0 + pp][r], pr[2 + pp][r], pr[4 + pp][r], pr[6 + pp][r]);
          transpose4(pi[0 + pp][r], pi[2 + pp][r], pi[4 + pp][r], pi[6 + pp][r]);
        }

      // ---- stage 2
      int xl = lane;
      asm volatile("" : "+v"(xl));
      const int gl = xl >> 4;
      xl &= 15;
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
          const int o = 2 * (ka & 1), kh = ka >> 1;
          if (rd == 0) {
            sv_re[kh][o] = pk(e_re[0], e_re[1]);
            sv_re[kh][o + 1] = pk(e_re[2], e_re[3]);
            sv_im[kh][o] = pk(e_im[0], e_im[1]);
            sv_im[kh][o + 1] = pk(e_im[2], e_im[3]);
          } else {
            float sr[4], si[4], dr[4], di[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const h2 br = __builtin_bit_cast(h2, sv_re[kh][o + (r >> 1)]);
              const h2 bi = __builtin_bit_cast(h2, sv_im[kh][o + (r >> 1)]);
              const float b_re = static_cast<float>(br[r & 1]), b_im = static_cast<float>(bi[r & 1]);
              sr[r] = b_re + e_re[r];
              si[r] = b_im + e_im[r];
              dr[r] = b_re - e_re[r];
              di[r] = b_im - e_im[r];
            }
            sv_re[kh][o] = pk(dr[0], dr[1]);
            sv_re[kh][o + 1] = pk(dr[2], dr[3]);
            sv_im[kh][o] = pk(di[0], di[1]);
            sv_im[kh][o + 1] = pk(di[2], di[3]);
            const u2 vr = {pk(sr[0], sr[1]), pk(sr[2], sr[3])};
            const u2 vi = {pk(si[0], si[1]), pk(si[2], si[3])};
            uint8_t* dst = img_q + ((ka / kRps) + (16 / kRps) * xl) * 256 +
                           16 * (((ka % kRps) * kCpr + 2 * w4 + (gl >> 1)) ^ xl) + 8 * ((gl & 1) ^ (xl >> 3));
            *reinterpret_cast<u2*>(dst) = vr;
            *reinterpret_cast<u2*>(dst + kPlaneAll) = vi;
          }
        } else if ((ka & 1) == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            hold_re[r] = e_re[r];
            hold_im[r] = e_im[r];
          }
        } else if (rd == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sv_re[ka >> 1][r] = pk(hold_re[r], e_re[r]);
            sv_im[ka >> 1][r] = pk(hold_im[r], e_im[r]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const h2 br = __builtin_bit_cast(h2, sv_re[ka >> 1][r]);
            const h2 bi = __builtin_bit_cast(h2, sv_im[ka >> 1][r]);
            const float b0r = static_cast<float>(br[0]), b1r = static_cast<float>(br[1]);
            const float b0i = static_cast<float>(bi[0]), b1i = static_cast<float>(bi[1]);
            acc_re[r][(ka >> 1) & 3] = pk(b0r + hold_re[r], b1r + e_re[r]);
            acc_im[r][(ka >> 1) & 3] = pk(b0i + hold_im[r], b1i + e_im[r]);
            sv_re[ka >> 1][r] = pk(b0r - hold_re[r], b1r - e_re[r]);
            sv_im[ka >> 1][r] = pk(b0i - hold_im[r], b1i - e_im[r]);
          }
          if ((ka & 7) == 7) {
            const int half = ka >> 3;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              // image row 16 w4 + x (512 B: this column's 256 values), 16-byte chunk c = 2 (4g + r) + half at slot c ^ x
              uint8_t* dst = img_q + 8192 * w4 + 512 * xl + 16 * ((2 * (4 * gl + r) + half) ^ xl);
              *reinterpret_cast<u4*>(dst) = u4{acc_re[r][0], acc_re[r][1], acc_re[r][2], acc_re[r][3]};
              *reinterpret_cast<u4*>(dst + kPlaneAll) = u4{acc_im[r][0], acc_im[r][1], acc_im[r][2], acc_im[r][3]};
            }
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // C: E_0 and O_0 are complete
    dma_table(seq);                          // G_q of the next block's first round (older than the register loads below: it has
                                             // landed when they have)
    // the next block's first round starts flying now. Unconditional (the last iteration re-reads its own block): a
    // conditional load keeps the OLD register contents alive through the whole loop body as far as the compiler can tell.
    issue_loads(rot.peek() < total ? rot.peek() : blk);

    // ---- read-out j: rows k + 256 j and k + 256 j + 512 from the pair (E_j, O_j / D_j) in the image
    //   j = 0:  X = E_0 +- O_0        j = 1:  X = E_1 -+ i D_1   (-i D = (D.im, -D.re))
    auto read_out = [&](const int j) {
      uint32_t tl = tid;
      asm volatile("" : "+v"(tl));
      if (MODE == kColsOnLanes) {
        // 16-byte chunks = 8 consecutive k of one column; a column's 1024 outputs are 2 KiB contiguous. The next pass's
        // twiddle is w_T^(av k_out) with av from the column (Ns = 1: kprev = 0).
        uint16_t* const c_re = a.out_re + bidx * a.out_stride;
        uint16_t* const c_im = a.out_im + bidx * a.out_stride;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const uint32_t L = it * kThreads + tl;
          const uint32_t f = L >> 5;                               // column within the block's 64
          const uint32_t k0 = 8 * ((L & 31) ^ (f & 15)) + 256 * j;
          const h8 ar = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(img + 16 * L));
          const h8 br = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(img + kHalf + 16 * L));
          const h8 ai = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(img + kPlaneAll + 16 * L));
          const h8 bi = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(img + kPlaneAll + kHalf + 16 * L));
          float x0r[8], x0i[8], x1r[8], x1i[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float Ar = static_cast<float>(ar[e]), Ai = static_cast<float>(ai[e]);
            const float Br = j ? static_cast<float>(bi[e]) : static_cast<float>(br[e]);
            const float Bi = j ? -static_cast<float>(br[e]) : static_cast<float>(bi[e]);
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
            cpx t1 = cmul(t0, lookup<kLut512>(a, (av * 512) & a.t_mask));
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
          const uint64_t o0 = (mb + f) * 1024 + k0;
          TFFT_ST_PASS(TW, s0r, reinterpret_cast<u4*>(c_re + o0));
          TFFT_ST_PASS(TW, s0i, reinterpret_cast<u4*>(c_im + o0));
          TFFT_ST_PASS(TW, s1r, reinterpret_cast<u4*>(c_re + o0 + 512));
          TFFT_ST_PASS(TW, s1i, reinterpret_cast<u4*>(c_im + o0 + 512));
          __builtin_amdgcn_sched_barrier(0);     // one chunk at a time (registers)
        }
        return;
      }
      // columns in registers: this thread takes 16-byte chunks (8 columns) of rows k + 256 j and k + 256 j + 512
      uint16_t* const o_re = a.out_re + bidx * a.out_stride;
      uint16_t* const o_im = a.out_im + bidx * a.out_stride;
      const uint64_t restb = mb >> a.ns_f_shift;                 // shared by the block's 64 columns (ns_f % 64 == 0)
      const uint64_t obase = ((restb << 10) << a.ns_f_shift) + (mb - (restb << a.ns_f_shift));
      cpx w_av = {1.f, 0.f}, w_half = {1.f, 0.f};
      uint64_t av = 0;
      if (TW == kTwNext) {
        av = restb >> a.a_shift;
        w_av = lookup<kLut512>(a, av & a.t_mask);                                            // w_T^av (per unit of kprev)
        w_half = lookup<kLut512>(a, (av * ((a.ns * 512) & a.t_mask)) & a.t_mask);            // w_T^(av ns 512)
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const uint32_t L = it * kThreads + tl;                  // 16-byte slot of the half-image
        const uint32_t sr = L >> 4;
        const uint32_t v = (L & 15) ^ (((sr * kRps) >> 4) & 15);   // output image: slot ^ kb
        const uint32_t k = sr * kRps + v / kCpr + 256 * j;
        const uint32_t chunk = v % kCpr;
        // (rows with kb >= 8 were staged with their two 8-byte halves flipped, see colfft256_wg_kernel's stage-2 stores)
        auto ld = [&](const uint8_t* ptr) {
          const u4 v4 = *reinterpret_cast<const u4*>(ptr);
          return __builtin_bit_cast(h8, (k & 128) ? u4{v4.z, v4.w, v4.x, v4.y} : v4);
        };
        const h8 ar = ld(img + 16 * L);
        const h8 br = ld(img + kHalf + 16 * L);
        const h8 ai = ld(img + kPlaneAll + 16 * L);
        const h8 bi = ld(img + kPlaneAll + kHalf + 16 * L);
        const uint64_t o0 = obase + (static_cast<uint64_t>(k) << a.ns_f_shift) + 8 * chunk;
        const uint64_t o1 = o0 + (static_cast<uint64_t>(512) << a.ns_f_shift);
        if (TW == kTwNone && !SC) {
          // last pass: the combine IS the output: packed binary16 sums (one correct rounding each, what the fp32 path's
          // sum-then-round gives, in 16 instructions instead of 80). -i D = (D.im, -D.re).
          TFFT_ST_PASS(TW, __builtin_bit_cast(u4, j ? ar + bi : ar + br), reinterpret_cast<u4*>(o_re + o0));
          TFFT_ST_PASS(TW, __builtin_bit_cast(u4, j ? ai - br : ai + bi), reinterpret_cast<u4*>(o_im + o0));
          TFFT_ST_PASS(TW, __builtin_bit_cast(u4, j ? ar - bi : ar - br), reinterpret_cast<u4*>(o_re + o1));
          TFFT_ST_PASS(TW, __builtin_bit_cast(u4, j ? ai + br : ai - bi), reinterpret_cast<u4*>(o_im + o1));
          __builtin_amdgcn_sched_barrier(0);
          continue;
        }
        float x0r[8], x0i[8], x1r[8], x1i[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float Ar = static_cast<float>(ar[e]), Ai = static_cast<float>(ai[e]);
          const float Br = j ? static_cast<float>(bi[e]) : static_cast<float>(br[e]);
          const float Bi = j ? -static_cast<float>(br[e]) : static_cast<float>(bi[e]);
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
        if (TW == kTwNext) {
          // E = av (kprev + ns k_out) mod T; kprev of column e of this chunk = kprev_f0 + e (inner = 1 wherever this pass is
          // planned): both rows' twiddles run along e as recurrences with step w_T^av
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
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    read_out(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // D: E_0 / O_0 are read out; E_1 / D_1 take their place
    int xl = lane;
    asm volatile("" : "+v"(xl));
    const int gl = xl >> 4;
    xl &= 15;
    if (MODE == kColsInRegs) {
#pragma unroll
      for (int ka = 0; ka < 16; ++ka) {
        const int o = 2 * (ka & 1);
        uint8_t* dst = img_q + ((ka / kRps) + (16 / kRps) * xl) * 256 +
                       16 * (((ka % kRps) * kCpr + 2 * w4 + (gl >> 1)) ^ xl) + 8 * ((gl & 1) ^ (xl >> 3));
        *reinterpret_cast<u2*>(dst) = u2{sv_re[ka >> 1][o], sv_re[ka >> 1][o + 1]};
        *reinterpret_cast<u2*>(dst + kPlaneAll) = u2{sv_im[ka >> 1][o], sv_im[ka >> 1][o + 1]};
      }
    } else {
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          uint8_t* dst = img_q + 8192 * w4 + 512 * xl + 16 * ((2 * (4 * gl + r) + half) ^ xl);
          *reinterpret_cast<u4*>(dst) = u4{sv_re[4 * half][r], sv_re[4 * half + 1][r], sv_re[4 * half + 2][r], sv_re[4 * half + 3][r]};
          *reinterpret_cast<u4*>(dst + kPlaneAll) = u4{sv_im[4 * half][r], sv_im[4 * half + 1][r], sv_im[4 * half + 2][r], sv_im[4 * half + 3][r]};
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // E
    read_out(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // F: read out; the next block's copy-in may overwrite the images
  }
  // the last iteration's look-ahead (table LDS-DMA, register loads of its own block again) must have landed before the
  // workgroup gives its LDS back
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef TFFT_DEBUG_KERNELS
  if (a.wg_times && tid == 0 && blockIdx.x < 8192) {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    a.wg_times[8192 + blockIdx.x] = wall_clock64();
    a.wg_times[16384 + blockIdx.x] = xcc;
  }
#endif
}

}  // namespace colfft
