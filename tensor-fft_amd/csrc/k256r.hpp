// k256r.hpp — batched N = 256 R (R = 2, 4, 8: N = 512, 1024, 2048) fp16 C2C FFT for gfx950 in ONE pass over HBM.
//
// The reference runs these lengths as TensorFFT256 followed by log2(R) launches of its radix-2 kernel per transform
// (src/base/ComputeFFT.h:72-145, Radix2.cu:20-77; Plan.h:99-118 counts the radix-2 steps), i.e. 1 + log2(R) round
// trips through global memory. Here the radix-R step is the LAST one and happens in registers:
//
//   n = R m + r  (r < R, m = n0 + 16 n1 < 256),   k = kk + 256 s  (kk = k0 + 16 k1 < 256, s < R)
//   X[kk + 256 s] = sum_r  w_R^(r s)  w_N^(r kk)  X_r[kk],        X_r = DFT_256 of the decimated sequence x[R m + r]
//
// One wave owns 16 / R transforms per iteration (4096 points, as in k256.hpp). A transform sits in LDS as in
// memory ([RE plane | IM plane], one contiguous LDS-DMA stream), except that the 32-byte blocks of a row n1
// (16 R elements) are XOR-swizzled on the SOURCE address so that the transposed reads are conflict free.
//   stage 1  tile u (u < R) of a transform = elements 16 u .. 16 u + 15 of every row n1: a transposed read hands lane
//            (g, x) the column x of the tile for the rows n1 = 4 g + j; column x is the sample (n0, r) =
//            ((16 u + x) / R, x mod R), so a tile mixes the R decimated sequences, which is fine: stage 1 contracts
//            n1 for every column independently (data as the A operand, F = w16^(n1 k0)/16 as B).
//            D1_u[row x = 4 g + reg][k0 = lane & 15], then  * w256^(n0 k0) / R  in fp32.
//   exchange none for R = 2, 4: the accumulator registers of the R tiles that belong to sequence r, taken in a fixed
//            order, ARE an A operand (rows k0, slots n0) once the constant operand of stage 2 lists n0 in that same
//            order. R = 8: a sequence lives in only two of the four 16-lane groups; 16 v_permlane16_swap pair them up.
//   stage 2  D2_r[k0 = 4 g + reg][k1 = lane & 15] = X_r[k0 + 16 k1], fp32.
//   radix R  per lane and register: multiply by w_N^(r kk) (fp32 table), DFT_R in registers, round once to binary16.
//            A lane holds 4 consecutive kk for every s: 8-byte pieces, 512 contiguous bytes per wave instruction.
// Result = DFT(x) / N (1/16 per MFMA stage folded into F, 1/R into the stage-1 twiddles).
#pragma once

#include "k4096.hpp"
#include "stockham.hpp"

namespace k256r {

using namespace k4096;

constexpr int kDataBytes = kWavesPerBlock * kLdsWaveBytes;   // 128 KiB: 8 waves x 4096 points
// constant blob (built per R on the host)
constexpr int kOffFa = 0;       // stage-1 B operand, slots n1 = 4 g + j: 64 lanes x {RE-form, IM-form} x 8 halfs
constexpr int kOffFb = 2048;    // stage-2 B operand, slots n0 = slot_n0<R>(g, j)
constexpr int kOffS = 4096;     // S[u][lane] = {float re[4], float im[4]}: w256^(n0 k0) / R for row 4 g + reg of tile u
template <int R> constexpr int off_t() { return kOffS + 2048 * R; }             // T[r - 1][lane]: w_N^(r kk)
template <int R> constexpr int table_bytes() { return off_t<R>() + 2048 * (R - 1); }
template <int R> constexpr int lds_bytes() { return kDataBytes + 2048 * R + 2048 * (R - 1); }   // data + S + T

// n0 held by contraction slot (lane group g, j) of the stage-2 A operand (see "exchange" above)
__host__ __device__ constexpr int slot_n0(int R, int g, int j) {
  return R == 2 ? 8 * (j >> 1) + 2 * g + (j & 1) : (R == 4 ? 4 * j + g : 2 * j + 8 * (g & 1) + (g >> 1));
}
// swizzle of the 32-byte blocks of row n1
__host__ __device__ constexpr int swz(int R, int n1) { return (n1 >> (R == 2 ? 2 : (R == 4 ? 1 : 0))) & (R - 1); }

// fa, fb: factors on the stage-1 / stage-2 matrices (1/16 each for sequential scaling); s_scale: factor of the fp32
// inter-stage twiddle block S (1/R for sequential scaling: the radix-R step's share)
inline void build_tables(int R, std::vector<uint8_t>& blob, double fa = 1.0 / 16, double fb = 1.0 / 16, double s_scale = -1.0) {
  if (s_scale < 0) s_scale = 1.0 / R;
  const int bytes = kOffS + 2048 * R + 2048 * (R - 1);
  blob.assign(bytes, 0);
  auto put_h = [&](int off, double v) {
    const _Float16 h = static_cast<_Float16>(v);
    std::memcpy(blob.data() + off, &h, 2);
  };
  auto put_f = [&](int off, double v) {
    const float f = static_cast<float>(v);
    std::memcpy(blob.data() + off, &f, 4);
  };
  auto cexp = [](long num, long den, double& c, double& s) {
    const long r = ((num % den) + den) % den;
    const double a = -2.0 * M_PI * static_cast<double>(r) / static_cast<double>(den);
    c = std::cos(a);
    s = std::sin(a);
  };
  const long n = 256L * R;
  for (int lane = 0; lane < 64; ++lane) {
    const int g = lane >> 4, x = lane & 15;
    for (int j = 0; j < 4; ++j) {
      double c, s;
      cexp(static_cast<long>(4 * g + j) * x, 16, c, s);
      int base = kOffFa + lane * 32;
      put_h(base + 2 * j, c * fa);
      put_h(base + 2 * (4 + j), -s * fa);
      put_h(base + 16 + 2 * j, s * fa);
      put_h(base + 16 + 2 * (4 + j), c * fa);
      cexp(static_cast<long>(slot_n0(R, g, j)) * x, 16, c, s);
      base = kOffFb + lane * 32;
      put_h(base + 2 * j, c * fb);
      put_h(base + 2 * (4 + j), -s * fb);
      put_h(base + 16 + 2 * j, s * fb);
      put_h(base + 16 + 2 * (4 + j), c * fb);
    }
    for (int u = 0; u < R; ++u)
      for (int reg = 0; reg < 4; ++reg) {
        const int n0 = (16 * u + 4 * g + reg) / R;
        double c, s;
        cexp(static_cast<long>(n0) * x, 256, c, s);
        put_f(kOffS + u * 2048 + lane * 32 + 4 * reg, c * s_scale);
        put_f(kOffS + u * 2048 + lane * 32 + 16 + 4 * reg, s * s_scale);
      }
    for (int r = 1; r < R; ++r)
      for (int reg = 0; reg < 4; ++reg) {
        const long kk = (4 * g + reg) + 16 * x;
        double c, s;
        cexp(r * kk, n, c, s);
        const int base = kOffS + 2048 * R + (r - 1) * 2048 + lane * 32;
        put_f(base + 4 * reg, c);
        put_f(base + 16 + 4 * reg, s);
      }
  }
}

// in_*/out_*: planar binary16; transform b at +b*stride halves. tables: build_tables(R) blob.
// STG: stage each transform's spectrum through its own (already consumed) LDS slot and store it as full 1-KiB rows with
// non-temporal 16-byte stores, instead of 8-byte pieces straight from registers.
template <int R, bool STG = true, bool OTW = false>
__global__ __launch_bounds__(kThreads, 2) void fft256r_kernel(const uint16_t* in_re, const uint16_t* in_im,
                                                              uint16_t* out_re, uint16_t* out_im, Addr in_map,
                                                              Addr out_map, uint32_t batch, uint32_t live,
                                                              const uint8_t* __restrict__ tables, OutTw otw) {
  constexpr int kPerWave = 16 / R;          // transforms per wave iteration
  constexpr int kPlane = 512 * R;           // bytes of one plane of one transform
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // S and T tables behind the data regions
  uint8_t* const tab = lds + kDataBytes;
  for (int i = tid; i < (2048 * R + 2048 * (R - 1)) / 16; i += kThreads)
    reinterpret_cast<u4*>(tab)[i] = reinterpret_cast<const u4*>(tables + kOffS)[i];
  const h8 fa_re = *reinterpret_cast<const h8*>(tables + kOffFa + lane * 32);
  const h8 fa_im = *reinterpret_cast<const h8*>(tables + kOffFa + lane * 32 + 16);
  const h8 fb_re = *reinterpret_cast<const h8*>(tables + kOffFb + lane * 32);
  const h8 fb_im = *reinterpret_cast<const h8*>(tables + kOffFb + lane * 32 + 16);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const uint8_t* const s_tab = tab + lane * 32;
  const uint8_t* const t_tab = tab + 2048 * R + lane * 32;

  uint8_t* const wl = lds + wave * kLdsWaveBytes;
  const uint32_t wl_off = __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t*)wl)));
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, x = lane & 15;
  // transposed read of tile u: row n1 = 4 g + q, block u ^ swz(n1), bytes 8 p of it
  const int n1r = 4 * g + q;
  const uint8_t* const tr_row = wl + 32 * R * n1r + 8 * p;
  const int sw = swz(R, n1r);
  // copy-in: LDS chunk c = 64 pq + lane of a plane <- row n1 = c / 2R, block (c % 2R) / 2 ^ swz(n1), half c & 1
  uint32_t src_off[R / 2];   // byte offset inside the plane, per 1-KiB piece pq of the plane
#pragma unroll
  for (int pq = 0; pq < R / 2; ++pq) {
    const int c = 64 * pq + lane;
    const int n1 = c / (2 * R), cc = c % (2 * R);
    src_off[pq] = 32 * R * n1 + 32 * ((cc >> 1) ^ swz(R, n1)) + 16 * (cc & 1);
  }
  // output: lane (k1 = x, g) writes kk = 4 g .. 4 g + 3 + 16 k1 of every segment s: halves 256 s + 16 x + 4 g
  const uint32_t out_lane = 16 * x + 4 * g;
  // staging position of that 8-byte piece inside a segment's 512 bytes, bank-conflict free (k256.hpp: plain, 32 x + 8 g, the 16
  // lanes of a ds_write_b64 group collide 4-way: 56 % of this kernel's LDS-active cycles, profiles/r5_n1024_pmc_summary.json)
#ifdef TFFT_K256_PLAIN_STAGE       // A/B knob: the layout of rounds 1-4
  const uint32_t stage_off = 2 * out_lane;
#else
  const uint32_t stage_off = 32u * x + 16u * ((g >> 1) ^ ((x >> 2) & 1)) + 8u * ((g & 1) ^ ((x >> 3) & 1));
#endif

  const uint32_t groups = (batch + kPerWave - 1) / kPerWave;
  // (live: waves of a workgroup that take groups, 8 or fewer for a small batch: k4096.hpp, tfft.hip live_waves())
  for (uint32_t grp = static_cast<uint32_t>(wave) < live ? blockIdx.x * live + wave : groups; grp < groups; grp += gridDim.x * live) {
    const uint32_t b0 = grp * kPerWave;
    const uint32_t nb = (batch - b0 < kPerWave) ? (batch - b0) : kPerWave;   // ragged last group
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = i / R, ii = i % R, plane = ii / (R / 2), pq = ii % (R / 2);
      // transforms past the end of the batch re-read the last valid one (their results are not stored)
      const uint32_t b = b0 + (static_cast<uint32_t>(t) < nb ? t : nb - 1);
      const uint8_t* src = reinterpret_cast<const uint8_t*>((plane ? in_im : in_re) + in_map.off(b)) +
                           src_off[pq];
      const uint32_t d = wl_off + i * 1024;
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
    for (int t = 0; t < kPerWave; ++t) {
      const uint8_t* const tb = tr_row + t * 2 * kPlane;
      // ---- stage 1 + twiddle, all R tiles of the transform
      f4 t_re[R], t_im[R];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const uint8_t* ad = tb + 32 * (u ^ sw);
        const s4 xr = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(ad));
        const s4 xi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(ad + kPlane));
        const u4 raw = {__builtin_bit_cast(u2, xr).x, __builtin_bit_cast(u2, xr).y, __builtin_bit_cast(u2, xi).x,
                        __builtin_bit_cast(u2, xi).y};
        const h8 a1 = __builtin_bit_cast(h8, raw);
        const f4 d_re = mfma(a1, fa_re);
        const f4 d_im = mfma(a1, fa_im);
        const f4 s_re = *reinterpret_cast<const f4*>(s_tab + u * 2048);
        const f4 s_im = *reinterpret_cast<const f4*>(s_tab + u * 2048 + 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          t_re[u][r] = __builtin_fmaf(d_re[r], s_re[r], -(d_im[r] * s_im[r]));
          t_im[u][r] = __builtin_fmaf(d_re[r], s_im[r], d_im[r] * s_re[r]);
        }
      }
      // ---- stage-2 operands per decimated sequence r
      u4 op[R];
      if constexpr (R == 2) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
          op[r] = u4{pk(t_re[0][r], t_re[0][r + 2]), pk(t_re[1][r], t_re[1][r + 2]), pk(t_im[0][r], t_im[0][r + 2]),
                     pk(t_im[1][r], t_im[1][r + 2])};
      } else if constexpr (R == 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          op[r] = u4{pk(t_re[0][r], t_re[1][r]), pk(t_re[2][r], t_re[3][r]), pk(t_im[0][r], t_im[1][r]),
                     pk(t_im[2][r], t_im[3][r])};
      } else {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          // lane groups 0, 2 hold sequence reg, groups 1, 3 sequence reg + 4; tiles (0,1),(2,3) <-> (4,5),(6,7)
          auto a_re = __builtin_amdgcn_permlane16_swap(pk(t_re[0][reg], t_re[1][reg]), pk(t_re[4][reg], t_re[5][reg]), false, false);
          auto b_re = __builtin_amdgcn_permlane16_swap(pk(t_re[2][reg], t_re[3][reg]), pk(t_re[6][reg], t_re[7][reg]), false, false);
          auto a_im = __builtin_amdgcn_permlane16_swap(pk(t_im[0][reg], t_im[1][reg]), pk(t_im[4][reg], t_im[5][reg]), false, false);
          auto b_im = __builtin_amdgcn_permlane16_swap(pk(t_im[2][reg], t_im[3][reg]), pk(t_im[6][reg], t_im[7][reg]), false, false);
          op[reg] = u4{a_re[0], b_re[0], a_im[0], b_im[0]};
          op[reg + 4] = u4{a_re[1], b_re[1], a_im[1], b_im[1]};
        }
      }
      // ---- stage 2, twiddle, radix-R butterfly, store
      f4 o_re[R], o_im[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const h8 a2 = __builtin_bit_cast(h8, op[r]);
        o_re[r] = mfma(a2, fb_re);
        o_im[r] = mfma(a2, fb_im);
      }
      uint32_t pk_re[R][2], pk_im[R][2];   // [s][register pair]
      float keep_re[R], keep_im[R];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        stockham::cf v[R];
        v[0] = {o_re[0][reg], o_im[0][reg]};
#pragma unroll
        for (int r = 1; r < R; ++r) {
          const f4 w_re = *reinterpret_cast<const f4*>(t_tab + (r - 1) * 2048);
          const f4 w_im = *reinterpret_cast<const f4*>(t_tab + (r - 1) * 2048 + 16);
          v[r].re = __builtin_fmaf(o_re[r][reg], w_re[reg], -(o_im[r][reg] * w_im[reg]));
          v[r].im = __builtin_fmaf(o_re[r][reg], w_im[reg], o_im[r][reg] * w_re[reg]);
        }
        stockham::dft<R>(v);
        if (OTW) {                     // transposed-input plan: output kk + 256 s of row (b0 + t) & row_mask times w_N^(row (kk + 256 s))
          const uint32_t row = (b0 + static_cast<uint32_t>(t)) & otw.row_mask;
          const Cf step = otw_w(otw, row, 256);
          Cf tw = otw_w(otw, row, 4u * g + reg + 16u * x);
#pragma unroll
          for (int s = 0; s < R; ++s) {
            cmul_to(v[s].re, v[s].im, tw);
            if (s + 1 < R) cmul_to(tw.re, tw.im, step);
          }
        }
#pragma unroll
        for (int s = 0; s < R; ++s) {
          if ((reg & 1) == 0) {
            keep_re[s] = v[s].re;
            keep_im[s] = v[s].im;
          } else {
            pk_re[s][reg >> 1] = pk(keep_re[s], v[s].re);
            pk_im[s][reg >> 1] = pk(keep_im[s], v[s].im);
          }
        }
      }
      if (STG) {
        // the transposed reads of transform t have all returned (their data went through the MFMAs above)
        uint8_t* const slot = wl + t * 2 * kPlane + stage_off;
#pragma unroll
        for (int s = 0; s < R; ++s) {
          *reinterpret_cast<u2*>(slot + 512 * s) = u2{pk_re[s][0], pk_re[s][1]};
          *reinterpret_cast<u2*>(slot + kPlane + 512 * s) = u2{pk_im[s][0], pk_im[s][1]};
        }
      } else if (static_cast<uint32_t>(t) < nb) {
        const uint64_t o = out_map.off(b0 + t) + out_lane;
#pragma unroll
        for (int s = 0; s < R; ++s) {
          const u2 vr = {pk_re[s][0], pk_re[s][1]};
          const u2 vi = {pk_im[s][0], pk_im[s][1]};
          *reinterpret_cast<u2*>(out_re + o + 256 * s) = vr;
          *reinterpret_cast<u2*>(out_im + o + 256 * s) = vi;
        }
      }
    }
    if (STG) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int t = i / R, ii = i % R, plane = ii / (R / 2), pq = ii % (R / 2);
        u4 v = *reinterpret_cast<const u4*>(wl + i * 1024 + 16 * lane);
#ifdef TFFT_K256_PLAIN_STAGE
        const uint32_t chunk = lane;
#else
        const uint32_t chunk = lane ^ ((lane >> 3) & 1);          // (two 512-byte segments per piece; the swizzle is per segment)
        if ((lane >> 4) & 1) v = u4{v.z, v.w, v.x, v.y};
#endif
        if (static_cast<uint32_t>(t) < nb) {
          uint16_t* dst = (plane ? out_im : out_re) + out_map.off(b0 + t) + 512 * pq + 8 * chunk;
          if (OTW) *reinterpret_cast<u4*>(dst) = v;      // intermediate of a transposed-input plan (see k256.hpp)
          else __builtin_nontemporal_store(v, reinterpret_cast<u4*>(dst));
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // image consumed before the next copy-in lands on it
  }
}

}  // namespace k256r
