// k4096r.hpp — batched N = 4096 R (R = 2, 4, 8: N = 8192, 16384, 32768) fp16 C2C FFT for gfx950 in ONE pass over HBM.
//
// The reference runs these lengths as TensorFFT4096 per 4096-point block followed by log2(R) radix-2 launches per
// transform (src/base/ComputeFFT.h:72-145, Radix2.cu:20-77), i.e. 1 + log2(R) round trips through global memory;
// this library's own multi-pass plan needs two (radix-256/512 column pass + tail). Here R waves share one transform
// and the radix-R step comes FIRST, on the way from LDS into the stage-1 operands (decimation in frequency):
//
//   n = m + 4096 r  (r < R),   k = R kk + s  (s < R)
//   u_s[m] = w_N^(m s) / (2 R)  sum_r x[m + 4096 r] w_R^(r s)       (fp32, rounded once to binary16; the extra 1/2 is
//                                                                    headroom for the rotation, returned after stage 2)
//   X[R kk + s] = DFT_4096(u_s)[kk] / 4096                           (the three MFMA stages of k4096.hpp, unchanged)
//
// Every wave owns 1/R of the sample positions m of its group's transform: it loads the R samples x[m + 4096 r] of each
// straight from HBM into registers (16-byte coalesced loads, issued one iteration ahead), computes all R outputs (DFT_R
// in registers, twiddles w_N^(m s) as powers of w_N^m, one rounding to binary16) and writes u_s[m] into LDS region s
// (16 KiB per wave) in the 4096 kernel's swizzled image. After a workgroup barrier region s holds u_s exactly as the 4096
// kernel expects it, and stages 1-3 are that kernel's, wave s working on region s. (The first version copied the blocks
// into LDS by LDS-DMA and did the butterfly in place there: one more barrier, 256 KiB more LDS traffic per iteration and,
// decisively, nothing in flight from HBM while the workgroup computed.) The R spectra are staged in the R regions and read out interleaved (X[R kk + s]: 8 / R consecutive
// kk from each of the R images make one 16-byte store), so global traffic is full 1-KiB rows in both directions.
// Three workgroup barriers per iteration. (A first version let every wave read all R blocks per stage-1 tile and form only
// its own u_s: R x the LDS reads and 2.3-3 x the arithmetic per sample; 2^15 ran at 350 Gsamples/s with it.)
#pragma once

#include "k4096.hpp"
#include "stockham.hpp"

namespace k4096r {

using namespace k4096;


// (Tried: two independent 4-wave workgroups per CU with the G table read from global memory instead of LDS, so that one
// workgroup computes while the other waits for memory: 2^13 487 -> 456 Gsamples/s, 2^14 435 -> 445; not kept.)
//
// ROWS (R = 8 only): the same machinery as the FIRST pass of a 2D transform of 4096 x 4096 images (include/tfft.h,
// tfft_plan2d_*). The column transform of length 4096 = 8 x 512 is split decimation-in-frequency, r = r0 + 512 i:
//   Y_s[r0][c] = w_4096^(r0 s) / 16  sum_i x[r0 + 512 i][c] w_8^(i s),     X[8 k' + s][kc] = DFT_512 over r0 of rowDFT(Y_s[r0])[kc]
// One workgroup iteration takes the 8 rows r0 + 512 i of an image (wave i copies row i), does the radix-8 butterfly in
// place across the 8 regions (the twiddle is one scalar per s), runs the 4096-point ROW transform of Y_s in wave s and
// stores it as row 512 s + r0 of the intermediate image. What remains is a radix-512 column pass over each block of
// 512 rows (colfft512_wg_kernel) that writes rows 8 k' + s: two passes over HBM instead of three. `batch` then counts
// workgroup iterations (images x 512) and the strides are per image.
// Round 4, where this pass's time goes (phase clock of the measurement build, tools/exp_rows_phases.py, profiles/r4_rows_phases.txt;
// cycles per iteration and wave, 31.5 k in all): front end 4.2 k, stages 1-3 5.8 k, and 10.5 k stalled at the ISSUE of the 16 loads
// behind barrier B (waves 4-7: 16.6 k) plus 5.3 k at the 16 stores: the CU's memory pipeline accepts a 1-KiB wave instruction about
// every 115 cycles and an in-order wave waits at the instruction (that is the 54 % SQ_WAIT_INST_ANY of the round-3 PMC run); the
// input itself has long landed when the front end asks for it (160 cycles). Spreading the loads over the stage tiles and the
// stores, held in registers, over the next front end (both built and A/B-ed in one process, bit-identical output) gives 0 % and
// -3 %: profiles/r4_rows_same_box.txt. On ONE box: a plain copy with this pass's access pattern 1776 us, the same copy with this
// loop's barriers and the arithmetic replaced by s_sleep (tools/rows2d_sched.hip) 1783 us, this kernel 1806 us. The pass runs at
// the copy rate of its pattern; what round 3 read as 10 % of headroom was the box-to-box spread of that copy rate (1611-1776 us).
template <int R, bool ROWS = false, bool OTW = false>
__global__ __launch_bounds__(kThreads, 2) void fft4096r_kernel(const uint16_t* in_re, const uint16_t* in_im,
                                                               uint16_t* out_re, uint16_t* out_im, Addr in_map,
                                                               Addr out_map, uint32_t batch, uint32_t gstep,
                                                               const uint8_t* __restrict__ tables, OutTw otw
#ifdef TFFT_DEBUG_KERNELS
                                                               , unsigned long long* stamps   // tools/exp_rows_phases.py; null in normal use
#endif
                                                               ) {
#ifdef TFFT_DEBUG_KERNELS
  // phase clock of the measurement build: per wave, the s_memtime cycles spent in each phase, summed over the iterations
  // (stamps[(block * 8 + wave) * 8 + phase]); costs a few SALU instructions per phase and nothing when stamps == null
  unsigned long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long ph_t = 0;
  const unsigned long long ph_entry = stamps ? __builtin_amdgcn_s_memtime() : 0;      // [7]: kernel entry -> loop start
#define TFFT_PHASE(i)                                                  \
  if (stamps) {                                                        \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();      \
    ph_acc[i] += now_ - ph_t;                                          \
    ph_t = now_;                                                       \
  }
#else
#define TFFT_PHASE(i)
#endif
  constexpr int kGroups = kWavesPerBlock / R;      // transforms per workgroup iteration
  constexpr int kN = 4096 * R;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave / R, s = wave % R;
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));

  // (Round 5 tried the first iteration's loads AHEAD of this table fill, one memory round trip instead of two: 7.0 / 8.0 us per
  // transform at 2^13 / 2^14 against 7.2 / 7.9 in this order: nothing, so the order of rounds 1-4 stands.)
  for (int i = tid; i < kLdsTableBytes / 16; i += kThreads)
    reinterpret_cast<u4*>(lds)[i] = reinterpret_cast<const u4*>(tables + kOffG)[i];
  h8 f_re = *reinterpret_cast<const h8*>(tables + kOffF1 + lane * 32);
  h8 f_im = *reinterpret_cast<const h8*>(tables + kOffF1 + lane * 32 + 16);
  // (the plan builds this block with a factor 2, k4096::TableScale::tw: it gives back the headroom factor of the front end
  // after two averaging MFMA stages, exact in fp32)
  f4 tw_re = *reinterpret_cast<const f4*>(tables + kOffTw + lane * 32);
  f4 tw_im = *reinterpret_cast<const f4*>(tables + kOffTw + lane * 32 + 16);
  h4 w_op = *reinterpret_cast<const h4*>(tables + kOffWR + (R == 2 ? 0 : (R == 4 ? 512 : 1024)) + lane * 8);   // radix-R front end (A operand)
  // The constants are operands of this statement, so the compiler has to have them in registers HERE (it waits for their
  // loads now and knows they have landed). Left to itself it sinks these loads (restrict + const: movable across the
  // "memory" clobber) below the first prefetch and then guards their first use, inside the loop, with s_waitcnt vmcnt(0..3),
  // which in steady state waits for the NEXT iteration's input that was issued just before: no overlap left.
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(f_re), "+v"(f_im), "+v"(tw_re), "+v"(tw_im), "+v"(w_op) : : "memory");
  __syncthreads();

  uint8_t* const wl = lds + kLdsTableBytes + wave * kLdsWaveBytes;                 // this wave's region
  uint8_t* const gl = lds + kLdsTableBytes + (grp * R) * kLdsWaveBytes;            // region of the group's block 0
  const uint8_t* const g_tab = lds + lane * 16;
  const uint8_t* const h_tab = lds + 16384 + lane * 16;
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int mrow = q + 4 * (g & 1), bb = g >> 1;
  const uint8_t* const tr_base = wl + mrow * 1024 + bb * 512 + 8 * p;

  // ---- constant of the radix-R front end: w_N^1 (v_cos / v_sin take revolutions; their ~1e-6 absolute error is three
  // orders below binary16's resolution)

  const uint32_t out_chunk = 4096u * s;     // this wave stores halves [4096 s, 4096 (s + 1)) of each output plane

  static_assert(!ROWS || R == 8, "the 2D row form takes the 8 rows r0 + 512 i of an image");
  // gstep = transforms per workgroup iteration: kGroups, or 1 for a batch of at most one transform per CU (tfft.hip): then only
  // group 0 of every workgroup has work, its R waves alone on the CU's SIMDs (2^13 x 4: 11.7 us with four transforms on one CU,
  // 7.6 us spread over four, profiles/r5_small_scan.txt)
  const uint32_t groups_total = ROWS ? batch : (batch + gstep - 1) / gstep;
  constexpr int kPs = 8 / R;                 // 16-byte chunks per lane, block and plane that this wave owns

  // Raw samples of one iteration, straight from HBM into registers, already in the shape of MFMA B operands: the
  // radix-R butterfly is a 16 x 16 x 16 product  U[rho'][column] = W[rho'][rho] X[rho][column]  with rho = (column set h,
  // plane, block i) and rho' = (column set h, output s, plane) (k4096::build_tables, kOffWR). Lane (g, n) supplies k-slots
  // rho = 4 g + jj for column n of a tile, so it loads, for j < 4 and jj < 4, the 16-byte chunk 64 (s kPs + h) + n + 16 j of
  // block i, plane pl: 8 consecutive columns e of 4 rows. Tile (j, e) takes element e of those four registers (two
  // v_perm_b32). A wave instruction is four 256-byte segments. The loads of iteration i + 1 are issued as soon as the
  // front end of iteration i has consumed these registers and fly under the three MFMA stages, the read-out and the
  // stores of iteration i (the LDS-DMA version of round 1 started its copy-in after the last barrier and waited for it:
  // one 160-KiB workgroup per CU, 3.6-3.9 TB/s, 35 % of the wave time parked in s_waitcnt / s_barrier).
  const int fg = lane >> 4, fn = lane & 15;
  u4 raw[4][4];
  auto issue_loads = [&](uint32_t it) {
    const uint32_t b_raw = ROWS ? (it >> 9) : it * gstep + grp;
    // a group past the end of the batch loads nothing (round 5: it used to re-read the last transform, harmless in a full batch,
    // but for ONE transform of 2^13 that is 128 KiB through the CU's load path instead of 32 KiB)
    if (!ROWS && __builtin_amdgcn_readfirstlane(static_cast<int>(b_raw >= batch || static_cast<uint32_t>(grp) >= gstep))) return;
    const uint32_t b = b_raw;
    const uint32_t r0 = it & 511;
    const uint64_t base = in_map.off(b) + (ROWS ? static_cast<uint64_t>(r0) * 4096 : 0);
    constexpr uint64_t kBlockStep = ROWS ? 512ull * 4096 : 4096ull;     // block i of the group: rows r0 + 512 i, or samples 4096 i
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int rho = 4 * fg + jj;
      const int h = rho / (2 * R), pl = (rho % (2 * R)) / R, i = rho % R;
      const uint16_t* const src = (pl ? in_im : in_re) + base + kBlockStep * i + 8u * (64u * (s * kPs + h) + fn);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        // Non-temporal in both forms. The 2D row pass ran with plain loads through round 3 (whole batch per pass: 3.65-3.75 ->
        // 3.47 ms); since round 4 the fused 2D plan runs chunk by chunk so that the column pass finds the intermediate images in
        // the Infinity Cache, and then the input stream must not push them out: 4096^2 x 64 with chunks of 4 images 3.57 ms with
        // plain input loads, 3.06-3.10 ms with non-temporal ones (profiles/r4_ab_2d_chunk.txt)
        raw[j][jj] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(src + 128 * j));
    }
  };

  Rotor rot(blockIdx.x, gridDim.x);                       // (iteration order: k4096::Rotor)
  if (rot.item() < groups_total) issue_loads(rot.item());

  // output side of the front-end product: lane (g, n) holds rows rho' = 4 g + r: outputs s2a (r = 0, 1: re, im) and
  // s2a + 1 (r = 2, 3) of column set h'
  const int hq = (4 * fg) / (2 * R), s2a = ((4 * fg) % (2 * R)) >> 1;
  const int mm_out = s * kPs + hq;                                       // 1-KiB block of the plane this lane writes
  uint8_t* const reg_a = gl + s2a * kLdsWaveBytes + mm_out * 1024;
  uint8_t* const reg_b = reg_a + kLdsWaveBytes;
  // 1D: twiddle w_N^(m s2), m = 8 (64 mm + n + 16 j) + e: per-lane steps w_N^(s2) along e
  const float sa_re = __builtin_amdgcn_cosf(static_cast<float>(s2a) * (1.0f / kN)),
              sa_im = -__builtin_amdgcn_sinf(static_cast<float>(s2a) * (1.0f / kN));
  const float sb_re = __builtin_amdgcn_cosf(static_cast<float>(s2a + 1) * (1.0f / kN)),
              sb_im = -__builtin_amdgcn_sinf(static_cast<float>(s2a + 1) * (1.0f / kN));

#ifdef TFFT_DEBUG_KERNELS
  if (stamps) {
    ph_t = __builtin_amdgcn_s_memtime();
    ph_acc[7] = ph_t - ph_entry;
  }
#endif
  for (uint32_t it = rot.item(); it < groups_total; rot.advance(), it = rot.item()) {
#ifdef TFFT_DEBUG_KERNELS
    if (stamps) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // this iteration's input has landed (the 16 younger operations are the previous iteration's stores)
#endif
    TFFT_PHASE(0)
    // A group past the end of the batch only keeps the barriers uniform: it skips the front end, the stages and the read-out.
    // (Through round 4 it re-did the last transform without storing. For ONE transform of 2^13 that is six of eight waves doing
    // useless work on the SIMDs the two live waves need: the phase clock showed them waiting 1.2 + 1.7 us at barriers B and C and
    // running the stages in 2.6 us instead of 1.3: profiles/r5_k4096r_phases.txt; 2^13 x 1: 10.1 us per transform.)
    const uint32_t b_raw = ROWS ? (it >> 9) : it * gstep + grp;      // ROWS: image index; r0 = it & 511
    const bool live = ROWS || (b_raw < batch && static_cast<uint32_t>(grp) < gstep);
    const uint32_t b = live ? b_raw : batch - 1;
    const uint32_t r0 = it & 511;

    // ---- radix-R front end on the matrix pipe: 32 tiles of 16 columns per wave. u_s[m] = w_N^(m s) / (2 R) sum_i x_i[m] w_R^(i s)
    // (2D rows: the scalar w_4096^(r0 s) instead of w_N^(m s)), rounded once to binary16 and written to region s in the
    // 4096 kernel's swizzled image: a lane's 8 tiles e of one j are the 8 columns of one 16-byte chunk.
    if (live) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float ta_re, ta_im, tb_re, tb_im;                               // twiddles of outputs s2a, s2a + 1 at e = 0
      if (ROWS) {
        const float ra = static_cast<float>((r0 * s2a) & 4095) * (1.0f / 4096), rb = static_cast<float>((r0 * (s2a + 1)) & 4095) * (1.0f / 4096);
        ta_re = __builtin_amdgcn_cosf(ra); ta_im = -__builtin_amdgcn_sinf(ra);
        tb_re = __builtin_amdgcn_cosf(rb); tb_im = -__builtin_amdgcn_sinf(rb);
      } else {
        const uint32_t m0 = 8u * (64u * mm_out + fn + 16u * j);
        float ra = static_cast<float>((m0 * s2a) & (kN - 1)) * (1.0f / kN), rb = static_cast<float>((m0 * (s2a + 1)) & (kN - 1)) * (1.0f / kN);
        // (these twiddles do not depend on the iteration: left alone, the compiler hoists all 32 tiles' worth of them out
        // of the loop, 128 floats, and spills them; the empty statement makes the angles opaque here)
        asm volatile("" : "+v"(ra), "+v"(rb));
        ta_re = __builtin_amdgcn_cosf(ra); ta_im = -__builtin_amdgcn_sinf(ra);
        tb_re = __builtin_amdgcn_cosf(rb); tb_im = -__builtin_amdgcn_sinf(rb);
      }
      uint32_t oa_re[4], oa_im[4], ob_re[4], ob_im[4];
      float ka_re = 0.f, ka_im = 0.f, kb_re = 0.f, kb_im = 0.f;       // even column waiting for its odd partner
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t sel = (e & 1) ? 0x07060302u : 0x05040100u;
        uint32_t d0[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) d0[jj] = (e >> 1) == 0 ? raw[j][jj].x : ((e >> 1) == 1 ? raw[j][jj].y : ((e >> 1) == 2 ? raw[j][jj].z : raw[j][jj].w));
        const u2 bop = {__builtin_amdgcn_perm(d0[1], d0[0], sel), __builtin_amdgcn_perm(d0[3], d0[2], sel)};
        const f4 z = {0.f, 0.f, 0.f, 0.f};
        const f4 d = __builtin_amdgcn_mfma_f32_16x16x16f16(w_op, __builtin_bit_cast(h4, bop), z, 0, 0, 0);
        const float ua_re = __builtin_fmaf(d[0], ta_re, -(d[1] * ta_im)), ua_im = __builtin_fmaf(d[0], ta_im, d[1] * ta_re);
        const float ub_re = __builtin_fmaf(d[2], tb_re, -(d[3] * tb_im)), ub_im = __builtin_fmaf(d[2], tb_im, d[3] * tb_re);
        if ((e & 1) == 0) {
          ka_re = ua_re; ka_im = ua_im; kb_re = ub_re; kb_im = ub_im;
        } else {
          oa_re[e >> 1] = pk(ka_re, ua_re); oa_im[e >> 1] = pk(ka_im, ua_im);
          ob_re[e >> 1] = pk(kb_re, ub_re); ob_im[e >> 1] = pk(kb_im, ub_im);
        }
        if (!ROWS && e < 7) {                                        // next column: w_N^((m + 1) s2)
          const float na_re = __builtin_fmaf(ta_re, sa_re, -(ta_im * sa_im)), na_im = __builtin_fmaf(ta_re, sa_im, ta_im * sa_re);
          const float nb_re = __builtin_fmaf(tb_re, sb_re, -(tb_im * sb_im)), nb_im = __builtin_fmaf(tb_re, sb_im, tb_im * sb_re);
          ta_re = na_re; ta_im = na_im; tb_re = nb_re; tb_im = nb_im;
        }
      }
      const uint32_t slot = 16u * ((fn + 16u * j) ^ (2u * mm_out));   // chunk n + 16 j of block mm_out, swizzled
      *reinterpret_cast<u4*>(reg_a + slot) = u4{oa_re[0], oa_re[1], oa_re[2], oa_re[3]};
      *reinterpret_cast<u4*>(reg_a + 8192 + slot) = u4{oa_im[0], oa_im[1], oa_im[2], oa_im[3]};
      *reinterpret_cast<u4*>(reg_b + slot) = u4{ob_re[0], ob_re[1], ob_re[2], ob_re[3]};
      *reinterpret_cast<u4*>(reg_b + 8192 + slot) = u4{ob_im[0], ob_im[1], ob_im[2], ob_im[3]};
      // keep the scheduler from pulling the next chunk's 8 products (32 accumulator registers) and twiddle chains up here:
      // with all 32 tiles in one scheduling region it spills
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    TFFT_PHASE(1)
    __builtin_amdgcn_s_barrier();            // B: u_0 .. u_(R-1) are complete
    TFFT_PHASE(2)
    // the raw registers are free: the next iteration's input starts flying now
    if (rot.peek() < groups_total) issue_loads(rot.peek());
    TFFT_PHASE(3)

    if (live) {
    // ---- stage 1 on this wave's own region, exactly the 4096 kernel's: D1_n1[k0 = 4g + r][n0 = lane & 15]
    uint32_t pr[8][4], pi[8][4];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      f4 dre[2], dim[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int n1 = 2 * t + e;
        const uint8_t* ad = tr_base + 32 * (n1 ^ mrow);
        const s4 xr = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(ad));
        const s4 xi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(ad + 8192));
        const u4 raw = {__builtin_bit_cast(u2, xr).x, __builtin_bit_cast(u2, xr).y,
                        __builtin_bit_cast(u2, xi).x, __builtin_bit_cast(u2, xi).y};
        const h8 xop = __builtin_bit_cast(h8, raw);
        dre[e] = mfma(f_re, xop);
        dim[e] = mfma(f_im, xop);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pr[t][r] = pk(dre[0][r], dre[1][r]);
        pi[t][r] = pk(dim[0][r], dim[1][r]);
      }
    }
    // (the transposed reads above have returned before the staging stores below are issued: their data feeds the
    // MFMAs whose results those stores depend on; the region is private to this wave from here to barrier C)

    // ---- n1 high bits (register index) <-> k0 high bits (lane group), as in the 4096 kernel
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        transpose4(pr[0 + pp][r], pr[2 + pp][r], pr[4 + pp][r], pr[6 + pp][r]);
        transpose4(pi[0 + pp][r], pi[2 + pp][r], pi[4 + pp][r], pi[6 + pp][r]);
      }

    Cf otw_t[4], otw_s;
    if (OTW) {
      const uint32_t row = b & otw.row_mask;
      otw_s = otw_w(otw, row, R);
#pragma unroll
      for (int r2 = 0; r2 < 4; ++r2) otw_t[r2] = otw_w(otw, row, static_cast<uint32_t>(R) * (16u * (lane & 15) + 256u * (4 * g + r2)) + s);
    }
    // ---- stages 2 and 3 tile by tile; the spectrum of u_s is staged in this wave's region in natural order
    auto tile23 = [&](int k0, f4& o_re, f4& o_im) {
      const int a = k0 >> 2, r = k0 & 3;
      const u4 araw = {pr[2 * a][r], pr[2 * a + 1][r], pi[2 * a][r], pi[2 * a + 1][r]};
      const h8 aop = __builtin_bit_cast(h8, araw);
      const u4 graw = *reinterpret_cast<const u4*>(g_tab + k0 * 1024);
      const f4 e_re = mfma(aop, __builtin_bit_cast(h8, graw));
      const f4 e_im = mfma(aop, im_form(graw));
      f4 t_re, t_im;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        t_re[r4] = __builtin_fmaf(e_re[r4], tw_re[r4], -(e_im[r4] * tw_im[r4]));
        t_im[r4] = __builtin_fmaf(e_re[r4], tw_im[r4], e_im[r4] * tw_re[r4]);
      }
      const u4 braw = {pk(t_re[0], t_re[1]), pk(t_re[2], t_re[3]), pk(t_im[0], t_im[1]), pk(t_im[2], t_im[3])};
      const h8 bop = __builtin_bit_cast(h8, braw);
      const u4 hraw = *reinterpret_cast<const u4*>(h_tab + k0 * 1024);
      o_re = mfma(__builtin_bit_cast(h8, hraw), bop);   // o[r2] = U_s[k0 + 16 k1 + 256 (4g + r2)]
      o_im = mfma(im_form(hraw), bop);
      if (OTW) {
        // transposed-input plan: output R kk + s (kk = k0 + 16 k1 + 256 (4g + r2)) of row b & row_mask times w_N^(row (R kk + s));
        // otw_t[r2] is the twiddle of the current tile (k0 = 0 .. 15 in order) and steps by w_N^(row R)
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2) {
          float vr = o_re[r2], vi = o_im[r2];
          cmul_to(vr, vi, otw_t[r2]);
          o_re[r2] = vr;
          o_im[r2] = vi;
          cmul_to(otw_t[r2].re, otw_t[r2].im, otw_s);
        }
      }
    };
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      uint32_t ore[4][4], oim[4][4];
#pragma unroll
      for (int kp = 0; kp < 4; ++kp) {
        f4 e_re, e_im, o_re, o_im;
        tile23(8 * half + 2 * kp, e_re, e_im);
        tile23(8 * half + 2 * kp + 1, o_re, o_im);
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2) {
          ore[r2][kp] = pk(e_re[r2], o_re[r2]);
          oim[r2][kp] = pk(e_im[r2], o_im[r2]);
        }
      }
#pragma unroll
      for (int r2 = 0; r2 < 4; ++r2) {
        // 16-byte slot 2 k1 + half of row 4g + r2 (512 B per row of 256 kk); plain layout: the interleaving
        // read-out below touches 2-, 4- or 8-byte pieces, so the 4096 kernel's slot swizzle is not used here
        // (R = 8, 1D: the 32-byte unit k1 of a row sits at unit k1 ^ s, so that the transposed read-out below, which
        // takes the same 32 bytes from all 8 regions at once, finds them in 8 different bank groups)
        // (2D rows: each wave reads its own region back as plain rows, so the 4096 kernel's slot swizzle applies: slot index
        // XOR bit 3 of itself, which puts lanes k1 and k1 + 4 of one 8-lane ds_write_b128 group on different banks; the plain
        // layout cost a 2-way conflict on every staging store: 20 % of this kernel's LDS-active cycles in round 2)
        const uint32_t k1s = (R == 8 && !ROWS) ? ((lane & 15) ^ s) : (lane & 15);
        const uint32_t slot = 2u * k1s + half;
        const uint32_t off = 16u * (ROWS ? (slot ^ ((slot >> 3) & 1u)) : slot) + 512u * (4 * g + r2);
        *reinterpret_cast<u4*>(wl + off) = u4{ore[r2][0], ore[r2][1], ore[r2][2], ore[r2][3]};
        *reinterpret_cast<u4*>(wl + 8192 + off) = u4{oim[r2][0], oim[r2][1], oim[r2][2], oim[r2][3]};
      }
    }
    }
    if (ROWS) {
      // the row spectrum leaves from this wave's own region (no other wave needs it): row 512 s + r0 of the image
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      TFFT_PHASE(4)
      uint16_t* const row_re = out_re + out_map.off(b) + static_cast<uint64_t>(512 * s + r0) * 4096;
      uint16_t* const row_im = out_im + out_map.off(b) + static_cast<uint64_t>(512 * s + r0) * 4096;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t rd = 16u * (lane ^ ((lane >> 3) & 1));
        const u4 vr = *reinterpret_cast<const u4*>(wl + 1024 * i + rd);
        const u4 vi = *reinterpret_cast<const u4*>(wl + 8192 + 1024 * i + rd);
        *reinterpret_cast<u4*>(row_re + 512 * i + 8 * lane) = vr;
        *reinterpret_cast<u4*>(row_im + 512 * i + 8 * lane) = vi;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      TFFT_PHASE(5)
      __builtin_amdgcn_s_barrier();          // D: every region has been read out before the next front end writes into it
      TFFT_PHASE(6)
      continue;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    TFFT_PHASE(4)
    __builtin_amdgcn_s_barrier();            // C: the R spectra of every group are staged
    TFFT_PHASE(6)

    // ---- interleaved read-out: X[R kk + s'] ; this wave stores output halves [4096 s, 4096 (s + 1)) of both planes
    uint16_t* const f_out_re = out_re + out_map.off(b) + out_chunk;
    uint16_t* const f_out_im = out_im + out_map.off(b) + out_chunk;
    if (live) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t idx0 = out_chunk + 512u * i + 8u * lane;   // first output index of this lane's 16 bytes
      const uint32_t kk0 = idx0 / R;                            // 8 / R consecutive kk from each image
      u4 vr, vi;
      if (R == 2) {
        const u2 a0 = *reinterpret_cast<const u2*>(gl + 2 * kk0), a1 = *reinterpret_cast<const u2*>(gl + kLdsWaveBytes + 2 * kk0);
        const u2 b0 = *reinterpret_cast<const u2*>(gl + 8192 + 2 * kk0), b1 = *reinterpret_cast<const u2*>(gl + kLdsWaveBytes + 8192 + 2 * kk0);
        vr = u4{(a0.x & 0xffffu) | (a1.x << 16), (a0.x >> 16) | (a1.x & 0xffff0000u), (a0.y & 0xffffu) | (a1.y << 16), (a0.y >> 16) | (a1.y & 0xffff0000u)};
        vi = u4{(b0.x & 0xffffu) | (b1.x << 16), (b0.x >> 16) | (b1.x & 0xffff0000u), (b0.y & 0xffffu) | (b1.y << 16), (b0.y >> 16) | (b1.y & 0xffff0000u)};
      } else if (R == 4) {
        uint32_t pa[4], pb[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
          pa[s2] = *reinterpret_cast<const uint32_t*>(gl + s2 * kLdsWaveBytes + 2 * kk0);
          pb[s2] = *reinterpret_cast<const uint32_t*>(gl + s2 * kLdsWaveBytes + 8192 + 2 * kk0);
        }
        vr = u4{(pa[0] & 0xffffu) | (pa[1] << 16), (pa[2] & 0xffffu) | (pa[3] << 16), (pa[0] >> 16) | (pa[1] & 0xffff0000u), (pa[2] >> 16) | (pa[3] & 0xffff0000u)};
        vi = u4{(pb[0] & 0xffffu) | (pb[1] << 16), (pb[2] & 0xffffu) | (pb[3] << 16), (pb[0] >> 16) | (pb[1] & 0xffff0000u), (pb[2] >> 16) | (pb[3] & 0xffff0000u)};
      } else {
        // R = 8: X[8 kk + s'] for 64 consecutive kk. A transposed LDS read hands lane (group rg, x) the four halves
        // U_s'[kk], s' = 4 (rg & 1) .. + 3, of kk = block start + x (rows = regions, columns = 16 consecutive kk = one
        // 32-byte unit of each region): 8 of the 16 output bytes. Groups (0, 1) and (2, 3) read two kk blocks each
        // and one v_permlane16_swap per dword gives every lane both halves of ITS kk: 2 reads + 2 swaps per vector
        // instead of 16 two-byte reads and their packing.
        const uint32_t rg = lane >> 4, region = 4 * (rg & 1) + ((lane >> 2) & 3);
        const uint32_t kka = out_chunk / 8 + 64u * i + 32u * (rg >> 1);      // first kk of this group pair's block A
        const uint8_t* const rb = gl + region * kLdsWaveBytes + 8 * (lane & 3);
        auto unit = [&](uint32_t kk) { return 512u * (kk >> 8) + 32u * (((kk >> 4) & 15) ^ region); };
        auto two = [&](const uint8_t* plane_base, u4& v) {
          const s4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(plane_base + unit(kka)));
          const s4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(plane_base + unit(kka + 16)));
          const u2 a = __builtin_bit_cast(u2, t0), b2 = __builtin_bit_cast(u2, t1);
          auto sx = __builtin_amdgcn_permlane16_swap(a.x, b2.x, false, false);
          auto sy = __builtin_amdgcn_permlane16_swap(a.y, b2.y, false, false);
          v = u4{sx[0], sy[0], sx[1], sy[1]};
        };
        two(rb, vr);
        two(rb + 8192, vi);
      }
      if (OTW) {       // intermediate of a transposed-input plan: plain stores, it stays in the Infinity Cache for the column pass
        *reinterpret_cast<u4*>(f_out_re + 512 * i + 8 * lane) = vr;
        *reinterpret_cast<u4*>(f_out_im + 512 * i + 8 * lane) = vi;
      } else {
        __builtin_nontemporal_store(vr, reinterpret_cast<u4*>(f_out_re + 512 * i + 8 * lane));
        __builtin_nontemporal_store(vi, reinterpret_cast<u4*>(f_out_im + 512 * i + 8 * lane));
      }
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef TFFT_DEBUG_KERNELS
    if (stamps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (measurement build: the stores acknowledged)
#endif
    TFFT_PHASE(5)
    __builtin_amdgcn_s_barrier();            // D: the staged spectra have been read; regions may be refilled
  }
#ifdef TFFT_DEBUG_KERNELS
  if (stamps && lane == 0 && blockIdx.x < 256)
    for (int i = 0; i < 8; ++i) stamps[(blockIdx.x * 8 + wave) * 8 + i] = ph_acc[i];
#endif
#undef TFFT_PHASE
}

}  // namespace k4096r
