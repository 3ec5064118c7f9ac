// colfft512r.hpp — radix-512 pass along a strided axis in TWO ROUNDS per 4-wave workgroup (gfx950): 80 KiB of LDS, so TWO
// workgroups share a CU and one moves data while the other computes.
//
// colfft512_wg_kernel (colfft.hpp) gives each of the two decimated sequences q = 0, 1 (rows i = 2 m + q) to one half of an
// 8-wave workgroup; its 128-KiB image plus 32 KiB of operands fill the CU's LDS, so copy-in, the two MFMA stages and the
// read-out of ONE tile run strictly one after the other (PMC, round 2: 66 % of the wave time parked in s_waitcnt / s_barrier,
// profiles/r2_c3b_pmc_summary.json). Here a workgroup of four waves takes the same 64-column tile through ONE 64-KiB image twice:
//
//   round 0   rows 2 m      LDS-DMA -> image -> stage 1 -> stage 2 with G_0:  B_0[k][column] stays in 64 VGPRs (packed binary16)
//   round 1   rows 2 m + 1  LDS-DMA (issued when stage 1 of round 0 has emptied the image, flying under its stage 2) -> stage 1 ->
//             stage 2 with G_1 = w_512^k G (combine twiddle folded into the operand): A_1 in the accumulators, combined lane by lane,
//             in fp32:  S = B_0 + A_1 = X[k] -> image,   D = B_0 - A_1 = X[k + 256] -> the registers that held B_0
//   read-out  S rows through the image as 128-byte segments, then D rows the same way
//
// Each stage-2 operand table (16 KiB) is fetched into the one table slot by LDS-DMA from L2 while the stage 1 in front of it runs
// (as colfft1024_wg_kernel does). Eight barriers per tile, but of a 256-thread workgroup whose partner on the CU is in
// another phase. The sums are formed from one binary16 (B_0) and one fp32 (A_1) term and rounded once (the 8-wave kernel rounds
// both terms and adds them as packed binary16).
//
// Only the form the 2D column pass and the last pass of a 1D plan need: columns in registers (Ns >= 64), no twiddles behind it;
// SC = the single factor of TFFT_SCALE_ONCE at the combine. Output-row re-mapping of the fused 2D plan as in colfft512_wg_kernel.
// Stands where the reference runs one TensorRadix16 launch per radix-16 level (src/base/TensorRadix16.cu:86-213).
#pragma once

#include "colfft.hpp"

namespace colfft {

// W = 4: 64-column tiles (128-byte row segments), 80 KiB of LDS, two workgroups per CU. W = 8: 128-column tiles (256-byte row
// segments, which this memory system moves 5 % faster: profiles/r3_stride_pad.txt), 144 KiB, one workgroup per CU.
template <int W>
constexpr int wg512r_lds_bytes() { return kLdsTable + 2 * WgGeom<W>::kPlane; }

template <int W, bool SC, bool PF = (W == 8), bool PLAIN = false>
__global__ __launch_bounds__(64 * W, 2) void colfft512r_wg_kernel(Args a) {
  constexpr bool kPlainAcc = PLAIN;
  using G = WgGeom<W>;
  constexpr int kPlane = G::kPlane, kRps = G::kRps, kCpr = G::kCpr;      // W = 4: 32 KiB, 2 rows per 256-byte super-row, 8 chunks per row
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  h8 f_re = *reinterpret_cast<const h8*>(a.tables + kOffF1n + lane * 32);
  h8 f_im = *reinterpret_cast<const h8*>(a.tables + kOffF1n + lane * 32 + 16);
  // the table slot: 16 KiB at the start of LDS, 16 / W KiB (16 / W LDS-DMA instructions) per wave
  constexpr int kTabPerWave = 16384 / W;
  const uint32_t tab_off = __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t*)lds)) + kTabPerWave * wave);
  auto dma_table = [&](int q) {
    const uint8_t* src = a.tables + kOffG512 + 16384 * q + kTabPerWave * wave + 16 * lane;
#pragma unroll
    for (int i = 0; i < kTabPerWave / 1024; ++i) {
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
  // (the constants are operands of this statement: their loads have landed and they sit in registers from here on)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(f_re), "+v"(f_im) : : "memory");

  uint8_t* const img = lds + kLdsTable;
  const uint32_t img_off = __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t*)img)));
  const uint8_t* const g_tab = lds + lane * 16;
  const int g = lane >> 4, q4 = (lane >> 2) & 3, p = lane & 3, x = lane & 15;
  const int ihi = 4 * g + q4;
  // image addressing exactly as colfft256_wg_kernel<.., W>: row r, 16-byte chunk c lives in super-row r / kRps at slot
  // ((r % kRps) kCpr + c) ^ 2 ((r >> 4) & 7)
  const uint8_t* tr_base[kRps];
#pragma unroll
  for (int h = 0; h < kRps; ++h)
    tr_base[h] = img + (16 / kRps) * ihi * 256 + 16 * ((h * kCpr + 2 * wave + (p >> 1)) ^ (2 * (ihi & 7))) + 8 * (p & 1);
  const uint32_t pshift = static_cast<uint32_t>(__builtin_ctzll(a.pitch));
  const uint32_t total = static_cast<uint32_t>(((a.tasks / a.groups) << pshift) / G::kCols);

  // copy-in of round rd: image row r = input row 2 r + rd; lane l of wave instruction i fills LDS byte 8192 wave + 1024 i + 16 l
  // (the opaque copies of the lane index in dma_in / out_slot / read_out keep the compiler from hoisting every per-lane address
  // of the unrolled loop body out of the loop, where they would sit in ~100 registers across all phases: colfft1024.hpp)
  auto dma_in = [&](uint32_t blk, int rd) {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const uint64_t gc0 = static_cast<uint64_t>(blk) * G::kCols;
    const uint64_t bidx = gc0 >> pshift;
    const uint64_t mb = gc0 & (a.pitch - 1);
    const uint16_t* const b_re = a.in_re + bidx * a.in_stride + mb;
    const uint16_t* const b_im = a.in_im + bidx * a.in_stride + mb;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t sr = 32 * wave + 4 * i + (lane >> 4);
      const uint32_t v = (lane & 15) ^ (2 * (((sr * kRps) >> 4) & 7));
      const uint32_t r = sr * kRps + v / kCpr;
      const uint32_t chunk = v % kCpr;
      const uint32_t row = 2 * r + rd;
      const uint64_t off = (row * a.pitch + static_cast<uint64_t>(row >> a.in_seg_shift) * a.in_seg_gap + 8 * chunk) * 2;
      const uint8_t* gr = reinterpret_cast<const uint8_t*>(b_re) + off;
      const uint8_t* gi = reinterpret_cast<const uint8_t*>(b_im) + off;
      const uint32_t d0 = img_off + 8192 * wave + 1024 * i, d1 = d0 + kPlane;
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
    }
  };

  // stage 1 of the image into pr / pi (then the 4 x 4 lane-group transposes), as in every column kernel
  auto stage1 = [&](uint32_t (&pr)[8][4], uint32_t (&pi)[8][4]) {
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
  };
  auto transposes = [&](uint32_t (&pr)[8][4], uint32_t (&pi)[8][4]) {
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        transpose4(pr[0 + pp][r], pr[2 + pp][r], pr[4 + pp][r], pr[6 + pp][r]);
        transpose4(pi[0 + pp][r], pi[2 + pp][r], pi[4 + pp][r], pi[6 + pp][r]);
      }
  };
  // Output image: row k = ka + 16 kb (kb = x) of this wave's columns 16 wave + 4 g .. + 3 is an 8-byte piece at super-row k / 2,
  // 16-byte slot ((k % 2) 8 + 2 wave + (g >> 1)) ^ kb. Within that slot the piece of an even / odd g sits in the lower /
  // upper half for kb < 8 and the other way round for kb >= 8: a 16-lane ds_write_b64 group (kb = 0 .. 15 at fixed g) then
  // spreads over all 32 banks (without the flip lanes kb and kb + 8 collide on every store: the 2-way conflict that made up
  // all of the 8-wave kernel's SQ_LDS_BANK_CONFLICT count). The read-out swaps the halves back for rows with kb >= 8.
  auto out_slot = [&](int ka, int xl, int gl) -> uint8_t* {
    return img + ((ka / kRps) + (16 / kRps) * xl) * 256 + 16 * (((ka % kRps) * kCpr + 2 * wave + (gl >> 1)) ^ xl) +
           8 * ((gl & 1) ^ (xl >> 3));
  };

  // PF: round 0 of the NEXT tile travels through registers instead: loaded behind barrier C (the image is busy until both
  // read-outs are done), in flight under the read-outs and their stores, written to the image at the top of the next tile
  // (what the 8-wave single-round kernel and colfft1024_wg_kernel do; 64 VGPRs that are free in that interval).
  u4 ra_re[PF ? 8 : 1], ra_im[PF ? 8 : 1];
  auto issue_loads = [&](uint32_t blk) {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const uint64_t gc0 = static_cast<uint64_t>(blk) * G::kCols;
    const uint64_t bidx = gc0 >> pshift;
    const uint64_t mb = gc0 & (a.pitch - 1);
    const uint16_t* const b_re = a.in_re + bidx * a.in_stride + mb;
    const uint16_t* const b_im = a.in_im + bidx * a.in_stride + mb;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t sr = 32 * wave + 4 * i + (lane >> 4);
      const uint32_t v = (lane & 15) ^ (2 * (((sr * kRps) >> 4) & 7));
      const uint32_t row = 2 * (sr * kRps + v / kCpr);
      const uint64_t off = (row * a.pitch + static_cast<uint64_t>(row >> a.in_seg_shift) * a.in_seg_gap + 8 * (v % kCpr)) * 2;
      ra_re[PF ? i : 0] = TFFT_NT_LOAD(reinterpret_cast<const u4*>(reinterpret_cast<const uint8_t*>(b_re) + off));
      ra_im[PF ? i : 0] = TFFT_NT_LOAD(reinterpret_cast<const u4*>(reinterpret_cast<const uint8_t*>(b_im) + off));
    }
  };
  auto to_image = [&]() {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      *reinterpret_cast<u4*>(img + 8192 * wave + 1024 * i + 16 * lane) = ra_re[PF ? i : 0];
      *reinterpret_cast<u4*>(img + kPlane + 8192 * wave + 1024 * i + 16 * lane) = ra_im[PF ? i : 0];
    }
  };

  Rotor rot(blockIdx.x, gridDim.x);                          // (block order: k4096::Rotor)
  dma_table(0);
  if (rot.item() < total) {
    if (PF) issue_loads(rot.item());
    else dma_in(rot.item(), 0);
  }

  uint32_t sv_re[8][4], sv_im[8][4];                         // B_0, then D: [ka >> 1][2 (ka & 1) + {0, 1}] = columns {0,1}, {2,3} of tile ka

  for (uint32_t blk = rot.item(); blk < total; rot.advance(), blk = rot.item()) {
    const uint64_t gc0 = static_cast<uint64_t>(blk) * G::kCols;
    const uint64_t bidx = gc0 >> pshift;                     // pitch >= 16 W: one batch entry per block
    const uint64_t mb = gc0 & (a.pitch - 1);

    // ---------------- round 0: rows 2 m
    if (PF) {
      to_image();                            // (the compiler waits for the register loads here)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();            // A0: image (round 0) and G_0 are in LDS
    {
      uint32_t pr[8][4], pi[8][4];
      stage1(pr, pi);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();          // B0: every wave has read its slab: the image may be overwritten
      dma_in(blk, 1);                        // round 1 flies under stage 2 of round 0 (which writes registers only)
      transposes(pr, pi);
#pragma unroll
      for (int ka = 0; ka < 16; ++ka) {
        const int aa = ka >> 2, r0 = ka & 3;
        const u4 draw = {pr[2 * aa][r0], pr[2 * aa + 1][r0], pi[2 * aa][r0], pi[2 * aa + 1][r0]};
        const h8 dop = __builtin_bit_cast(h8, draw);
        const u4 graw = *reinterpret_cast<const u4*>(g_tab + ka * 1024);
        const f4 e_re = mfma(dop, __builtin_bit_cast(h8, graw));
        const f4 e_im = mfma(dop, im_form(graw));
        const int o = 2 * (ka & 1), kh = ka >> 1;
        sv_re[kh][o] = pk(e_re[0], e_re[1]);
        sv_re[kh][o + 1] = pk(e_re[2], e_re[3]);
        sv_im[kh][o] = pk(e_im[0], e_im[1]);
        sv_im[kh][o + 1] = pk(e_im[2], e_im[3]);
      }
    }
    // ---------------- round 1: rows 2 m + 1
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // A1: image (round 1) is in LDS; every wave is through stage 2 of round 0
    dma_table(1);                            // G_1 flies under stage 1
    {
      uint32_t pr[8][4], pi[8][4];
      stage1(pr, pi);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();          // B1: image free again, G_1 in place
      transposes(pr, pi);
      int xl = threadIdx.x & 63;
      asm volatile("" : "+v"(xl));
      const int gl = xl >> 4;
      xl &= 15;
#pragma unroll
      for (int ka = 0; ka < 16; ++ka) {
        const int aa = ka >> 2, r0 = ka & 3;
        const u4 draw = {pr[2 * aa][r0], pr[2 * aa + 1][r0], pi[2 * aa][r0], pi[2 * aa + 1][r0]};
        const h8 dop = __builtin_bit_cast(h8, draw);
        const u4 graw = *reinterpret_cast<const u4*>(g_tab + ka * 1024);
        const f4 e_re = mfma(dop, __builtin_bit_cast(h8, graw));
        const f4 e_im = mfma(dop, im_form(graw));
        const int o = 2 * (ka & 1), kh = ka >> 1;
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
          if (SC) {
            sr[r] *= a.comb_scale;
            si[r] *= a.comb_scale;
            dr[r] *= a.comb_scale;
            di[r] *= a.comb_scale;
          }
        }
        sv_re[kh][o] = pk(dr[0], dr[1]);
        sv_re[kh][o + 1] = pk(dr[2], dr[3]);
        sv_im[kh][o] = pk(di[0], di[1]);
        sv_im[kh][o + 1] = pk(di[2], di[3]);
        uint8_t* dst = out_slot(ka, xl, gl);
        *reinterpret_cast<u2*>(dst) = u2{pk(sr[0], sr[1]), pk(sr[2], sr[3])};
        *reinterpret_cast<u2*>(dst + kPlane) = u2{pk(si[0], si[1]), pk(si[2], si[3])};
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // C: S is complete in the image; the table slot is free
    dma_table(0);                            // G_0 of the next tile (lands long before it is needed)
    // unconditional (the last tile re-reads itself): a conditional load keeps the old register contents alive through the loop
    if (PF) issue_loads(rot.peek() < total ? rot.peek() : blk);

    // ---------------- read-out: rows k (S) and, through the same image, rows k + 256 (D)
    const uint64_t o_entry = (bidx >> a.out_sub_shift) * a.out_stride +
                             (bidx & ((1ull << a.out_sub_shift) - 1)) * a.out_sub_stride;
    uint16_t* const o_re = a.out_re + o_entry;
    uint16_t* const o_im = a.out_im + o_entry;
    const uint32_t row_shift = a.ns_f_shift + a.out_row_shift;
    const uint64_t restb = mb >> a.ns_f_shift;                 // shared by the block's 16 W columns (ns_f % (16 W) == 0)
    const uint64_t obase = ((restb << 9) << a.ns_f_shift) + (mb - (restb << a.ns_f_shift));
    auto read_out = [&](const uint32_t krow0) {
      int lane = threadIdx.x & 63;
      asm volatile("" : "+v"(lane));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t sr = 32 * wave + 4 * i + (lane >> 4);
        const uint32_t v = (lane & 15) ^ (((sr * kRps) >> 4) & 15);   // output image: slot ^ kb
        const uint32_t k = sr * kRps + v / kCpr;
        const uint32_t chunk = v % kCpr;
        u4 vr = *reinterpret_cast<const u4*>(img + 8192 * wave + 1024 * i + 16 * lane);
        u4 vi = *reinterpret_cast<const u4*>(img + kPlane + 8192 * wave + 1024 * i + 16 * lane);
        if (k & 128) {                                                // kb >= 8: the two 8-byte halves were stored flipped
          vr = u4{vr.z, vr.w, vr.x, vr.y};
          vi = u4{vi.z, vi.w, vi.x, vi.y};
        }
        const uint64_t o = obase + (static_cast<uint64_t>(k + krow0) << row_shift) + 8 * chunk;
        TFFT_NT_STORE(vr, reinterpret_cast<u4*>(o_re + o));
        TFFT_NT_STORE(vi, reinterpret_cast<u4*>(o_im + o));
      }
    };
    read_out(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // D: S has been read out; D takes its place
    int xl = threadIdx.x & 63;
    asm volatile("" : "+v"(xl));
    const int gl = xl >> 4;
    xl &= 15;
#pragma unroll
    for (int ka = 0; ka < 16; ++ka) {
      const int o = 2 * (ka & 1);
      uint8_t* dst = out_slot(ka, xl, gl);
      *reinterpret_cast<u2*>(dst) = u2{sv_re[ka >> 1][o], sv_re[ka >> 1][o + 1]};
      *reinterpret_cast<u2*>(dst + kPlane) = u2{sv_im[ka >> 1][o], sv_im[ka >> 1][o + 1]};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // E
    read_out(256);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // F: read out; the next tile's copy-in may overwrite the image
    if (!PF && rot.peek() < total) dma_in(rot.peek(), 0);
  }
  // (a table LDS-DMA of the last iteration may still be in flight: the workgroup must not give its LDS back before it lands)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace colfft
