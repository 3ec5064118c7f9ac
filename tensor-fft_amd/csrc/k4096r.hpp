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
template <int R, bool ROWS = false>
__global__ __launch_bounds__(kThreads, 2) void fft4096r_kernel(const uint16_t* in_re, const uint16_t* in_im,
                                                               uint16_t* out_re, uint16_t* out_im, Addr in_map,
                                                               Addr out_map, uint32_t batch,
                                                               const uint8_t* __restrict__ tables) {
  constexpr int kGroups = kWavesPerBlock / R;      // transforms per workgroup iteration
  constexpr int kN = 4096 * R;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave / R, s = wave % R;

  for (int i = tid; i < kLdsTableBytes / 16; i += kThreads)
    reinterpret_cast<u4*>(lds)[i] = reinterpret_cast<const u4*>(tables + kOffG)[i];
  h8 f_re = *reinterpret_cast<const h8*>(tables + kOffF1 + lane * 32);
  h8 f_im = *reinterpret_cast<const h8*>(tables + kOffF1 + lane * 32 + 16);
  // (the plan builds this block with a factor 2, k4096::TableScale::tw: it gives back the headroom factor of the front end
  // after two averaging MFMA stages, exact in fp32)
  f4 tw_re = *reinterpret_cast<const f4*>(tables + kOffTw + lane * 32);
  f4 tw_im = *reinterpret_cast<const f4*>(tables + kOffTw + lane * 32 + 16);
  // The constants are operands of this statement, so the compiler has to have them in registers HERE (it waits for their
  // loads now and knows they have landed). Left to itself it sinks these loads (restrict + const: movable across the
  // "memory" clobber) below the first prefetch and then guards their first use, inside the loop, with s_waitcnt vmcnt(0..3),
  // which in steady state waits for the NEXT iteration's input that was issued just before: no overlap left.
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(f_re), "+v"(f_im), "+v"(tw_re), "+v"(tw_im) : : "memory");
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
  const float st_re = __builtin_amdgcn_cosf(1.0f / kN), st_im = -__builtin_amdgcn_sinf(1.0f / kN);

  const uint32_t out_chunk = 4096u * s;     // this wave stores halves [4096 s, 4096 (s + 1)) of each output plane

  static_assert(!ROWS || R == 8, "the 2D row form takes the 8 rows r0 + 512 i of an image");
  const uint32_t groups_total = ROWS ? batch : (batch + kGroups - 1) / kGroups;
  constexpr int kPs = 8 / R;                 // 16-byte chunks per lane, block and plane that this wave owns

  // Raw samples of one iteration: this wave's chunk positions of all R blocks, both planes, straight from HBM into
  // registers (16 x global_load_dwordx4, 1 KiB per wave instruction, non-temporal). The loads of iteration i + 1 are
  // issued as soon as the front end of iteration i has consumed these registers, so they fly under the three MFMA
  // stages, the read-out and the stores of iteration i: the kernel never sits waiting for its input with nothing
  // else to do, which is what the LDS-DMA version did between its barriers D and A (one 160-KiB workgroup per CU,
  // 3.6-3.9 TB/s; PMC: 35 % of the wave time parked in s_waitcnt / s_barrier).
  u4 raw_r[kPs][R], raw_i[kPs][R];
  auto issue_loads = [&](uint32_t it) {
    const uint32_t b_raw = ROWS ? (it >> 9) : it * kGroups + grp;
    const uint32_t b = (ROWS || b_raw < batch) ? b_raw : batch - 1;     // past the end: re-read the last transform
    const uint32_t r0 = it & 511;
    const uint16_t* const base_re = in_re + in_map.off(b) + (ROWS ? static_cast<uint64_t>(r0) * 4096 : 0);
    const uint16_t* const base_im = in_im + in_map.off(b) + (ROWS ? static_cast<uint64_t>(r0) * 4096 : 0);
    constexpr uint64_t kBlockStep = ROWS ? 512ull * 4096 : 4096ull;     // block r of the group: rows r0 + 512 r, or samples 4096 r
#pragma unroll
    for (int ps = 0; ps < kPs; ++ps) {
      const uint32_t chunk = 64u * (s * kPs + ps) + lane;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        raw_r[ps][r] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(base_re + kBlockStep * r + 8 * chunk));
        raw_i[ps][r] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(base_im + kBlockStep * r + 8 * chunk));
      }
    }
  };
  if (blockIdx.x < groups_total) issue_loads(blockIdx.x);

  for (uint32_t it = blockIdx.x; it < groups_total; it += gridDim.x) {
    // a group past the end of the batch re-does the last transform (it keeps the barriers uniform) without storing
    const uint32_t b_raw = ROWS ? (it >> 9) : it * kGroups + grp;      // ROWS: image index; r0 = it & 511
    const bool live = ROWS || b_raw < batch;
    const uint32_t b = live ? b_raw : batch - 1;
    const uint32_t r0 = it & 511;

    // ---- radix-R front end, registers -> the group's R LDS regions: this wave owns 8 / R chunks per lane (16 bytes =
    // 8 consecutive m) of every block, so each sample is read once and each u_s[m] written once. Region s ends up
    // holding u_s in the 4096 kernel's (swizzled) image.
#pragma unroll
    for (int ps = 0; ps < kPs; ++ps) {
      const int mm = s * kPs + ps;                                   // 1-KiB block of the plane: chunks 64 mm .. 64 mm + 63
      const uint32_t slot = mm * 1024 + 16 * (lane ^ (2 * mm));      // LDS slot of global chunk c = 64 mm + lane
      // w_N^m, m = m0 + e (1D) or the per-iteration scalar w_4096^r0 (2D rows)
      const float rev0 = ROWS ? static_cast<float>(r0) * (1.0f / 4096) : static_cast<float>(8 * (64 * mm + lane)) * (1.0f / kN);
      float w1_re = __builtin_amdgcn_cosf(rev0), w1_im = -__builtin_amdgcn_sinf(rev0);
      // (plain dword arrays: __builtin_bit_cast applied directly to an element of an ext-vector reads element 0)
      uint32_t in_r[R][4], in_i[R][4];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const u4 vr = raw_r[ps][r];
        const u4 vi = raw_i[ps][r];
        in_r[r][0] = vr.x; in_r[r][1] = vr.y; in_r[r][2] = vr.z; in_r[r][3] = vr.w;
        in_i[r][0] = vi.x; in_i[r][1] = vi.y; in_i[r][2] = vi.z; in_i[r][3] = vi.w;
      }
      uint32_t o_r[R][4], o_i[R][4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {        // two samples (one dword of each plane) at a time
        float ur[2][R], ui[2][R];
#pragma unroll
        for (int lo = 0; lo < 2; ++lo) {
          stockham::cf v[R];
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const h2 hr = __builtin_bit_cast(h2, in_r[r][e2]), hi = __builtin_bit_cast(h2, in_i[r][e2]);
            v[r] = stockham::cf{static_cast<float>(hr[lo]), static_cast<float>(hi[lo])};
          }
          stockham::dft<R>(v);
          // u_s = v[s] w_N^(s m) / (2 R): powers of w1 = w_N^m, the factor 1 / (2 R) riding on them
          float pw_re = w1_re * (0.5f / R), pw_im = w1_im * (0.5f / R);
          ur[lo][0] = v[0].re * (0.5f / R);
          ui[lo][0] = v[0].im * (0.5f / R);
#pragma unroll
          for (int s2 = 1; s2 < R; ++s2) {
            const float xr = v[s2].re, xi = v[s2].im;
            ur[lo][s2] = __builtin_fmaf(xr, pw_re, -(xi * pw_im));
            ui[lo][s2] = __builtin_fmaf(xr, pw_im, xi * pw_re);
            if (s2 + 1 < R) {
              const float nr = __builtin_fmaf(pw_re, w1_re, -(pw_im * w1_im));
              const float ni = __builtin_fmaf(pw_re, w1_im, pw_im * w1_re);
              pw_re = nr;
              pw_im = ni;
            }
          }
          if (!ROWS) {        // next sample: w_N^(m + 1)
            // The recurrence does not depend on the data, so the scheduler would run all 8 samples' twiddle powers
            // ahead (128 live floats for R = 8: 256 VGPRs and spills). Tie it to this sample's result (no instruction).
            asm volatile("" : "+v"(w1_re), "+v"(w1_im) : "v"(ur[lo][R - 1]));
            const float nr = __builtin_fmaf(w1_re, st_re, -(w1_im * st_im));
            const float ni = __builtin_fmaf(w1_re, st_im, w1_im * st_re);
            w1_re = nr;
            w1_im = ni;
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < R; ++s2) {
          o_r[s2][e2] = pk(ur[0][s2], ur[1][s2]);
          o_i[s2][e2] = pk(ui[0][s2], ui[1][s2]);
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < R; ++s2) {
        *reinterpret_cast<u4*>(gl + s2 * kLdsWaveBytes + slot) = u4{o_r[s2][0], o_r[s2][1], o_r[s2][2], o_r[s2][3]};
        *reinterpret_cast<u4*>(gl + s2 * kLdsWaveBytes + 8192 + slot) = u4{o_i[s2][0], o_i[s2][1], o_i[s2][2], o_i[s2][3]};
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // B: u_0 .. u_(R-1) are complete
    // the raw registers are free: the next iteration's input starts flying now
    if (it + gridDim.x < groups_total) issue_loads(it + gridDim.x);

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
        const uint32_t off = 16u * (2u * (lane & 15) + half) + 512u * (4 * g + r2);
        *reinterpret_cast<u4*>(wl + off) = u4{ore[r2][0], ore[r2][1], ore[r2][2], ore[r2][3]};
        *reinterpret_cast<u4*>(wl + 8192 + off) = u4{oim[r2][0], oim[r2][1], oim[r2][2], oim[r2][3]};
      }
    }
    if (ROWS) {
      // the row spectrum leaves from this wave's own region (no other wave needs it): row 512 s + r0 of the image
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      uint16_t* const row_re = out_re + out_map.off(b) + static_cast<uint64_t>(512 * s + r0) * 4096;
      uint16_t* const row_im = out_im + out_map.off(b) + static_cast<uint64_t>(512 * s + r0) * 4096;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const u4 vr = *reinterpret_cast<const u4*>(wl + 1024 * i + 16 * lane);
        const u4 vi = *reinterpret_cast<const u4*>(wl + 8192 + 1024 * i + 16 * lane);
        __builtin_nontemporal_store(vr, reinterpret_cast<u4*>(row_re + 512 * i + 8 * lane));
        __builtin_nontemporal_store(vi, reinterpret_cast<u4*>(row_im + 512 * i + 8 * lane));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();          // D: every region has been read out before the next front end writes into it
      continue;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // C: the R spectra of every group are staged

    // ---- interleaved read-out: X[R kk + s'] ; this wave stores output halves [4096 s, 4096 (s + 1)) of both planes
    uint16_t* const f_out_re = out_re + out_map.off(b) + out_chunk;
    uint16_t* const f_out_im = out_im + out_map.off(b) + out_chunk;
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
        uint32_t pa[8], pb[8];
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
          pa[s2] = *reinterpret_cast<const uint16_t*>(gl + s2 * kLdsWaveBytes + 2 * kk0);
          pb[s2] = *reinterpret_cast<const uint16_t*>(gl + s2 * kLdsWaveBytes + 8192 + 2 * kk0);
        }
        vr = u4{pa[0] | (pa[1] << 16), pa[2] | (pa[3] << 16), pa[4] | (pa[5] << 16), pa[6] | (pa[7] << 16)};
        vi = u4{pb[0] | (pb[1] << 16), pb[2] | (pb[3] << 16), pb[4] | (pb[5] << 16), pb[6] | (pb[7] << 16)};
      }
      if (live) {
        __builtin_nontemporal_store(vr, reinterpret_cast<u4*>(f_out_re + 512 * i + 8 * lane));
        __builtin_nontemporal_store(vi, reinterpret_cast<u4*>(f_out_im + 512 * i + 8 * lane));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // D: the staged spectra have been read; regions may be refilled
  }
}

}  // namespace k4096r
