// k4096.hpp — batched N = 4096 fp16 C2C FFT for gfx950 (MI355X), one wave per FFT.
//
// Replaces TensorFFT4096 (reference src/base/TensorFFT4096.cu:22-413) launched
// once per FFT by ComputeFFT's batch overload (src/base/ComputeFFT.h:189-208)
// with ONE persistent launch over the whole batch. Same mathematics
// (DFT-16 -> radix-16 combine -> radix-16 combine, result = DFT(x)/4096), a
// different machine mapping:
//
//   n = n0 + 16 n1 + 256 n2   (input index),   k = k0 + 16 k1 + 256 k2  (output)
//
//   stage 1  contract n2 -> k0   constant operand F  = w16^(n2 k0)            / 16
//   stage 2  contract n1 -> k1   constant operand G_k0 = w256^(n1 (k0+16 k1)) / 16
//            elementwise         w256^(n0 k1)     (fp32, 4 constants per lane)
//   stage 3  contract n0 -> k2   constant operand H_k0 = w4096^(n0 (k0+256 k2)) / 16
//
// Each contraction is a complex 16x16x16 product done as TWO
// v_mfma_f32_16x16x32_f16: the K = 32 slots hold [re(0..3) | im(0..3)] per lane
// group, against [C_re ; -C_im] for the real part and [C_im ; C_re] for the
// imaginary part. The twiddles of the reference's combine steps
// (TensorFFT4096.cu:239-255, 331-350) are folded into G and H (they only depend
// on the contraction index and on k0, which is the tile index here), except the
// w256^(n0 k1) factor, applied to the fp32 accumulators.
//
// Data movement per FFT (16 KiB in, 16 KiB out, nothing else touches HBM):
//   HBM -> LDS    16 x global_load_lds_dwordx4 (1 KiB each, full lines); the LDS
//                 image is the global image with a 16-byte-chunk XOR swizzle applied
//                 on the SOURCE address so the transposed reads are conflict free.
//   LDS -> VGPR   ds_read_b64_tr_b16: the digit reversal of the reference's
//                 uncoalesced gather (TensorFFT4096.cu:128-180) becomes LDS
//                 addressing.
//   stage 1 -> 2  the accumulator layout of stage 1 (row on lane>>4 and register,
//                 column on lane&15) is the operand layout of stage 2 once n1's
//                 high bits and k0's high bits trade places between registers and
//                 lane groups: 4x4 transposes with v_permlane16_swap /
//                 v_permlane32_swap, no LDS.
//   stage 2 -> 3  free: stage 2 is issued with the data as the A operand, which
//                 leaves n0 exactly where stage 3's contraction slots live.
//   VGPR -> HBM   each lane ends up with 16 consecutive outputs (k0 = 0..15) for
//                 4 values of k2: 32-byte runs, stored as 16-byte vectors.
//
// Scaling: 1/16 per stage folded into F, G, H (exact in binary16), so every
// intermediate is bounded by max|x| and nothing is pushed into the subnormal
// range the way the reference's up-front x/4096 is (TensorFFT4096.cu:169-173).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstring>
#include <vector>

namespace k4096 {

constexpr int kWavesPerBlock = 8;
constexpr int kThreads = 64 * kWavesPerBlock;
constexpr int kLdsTableBytes = 32768;          // G (16 KiB) + H (16 KiB)
constexpr int kLdsWaveBytes = 16384;           // one FFT: RE plane + IM plane
constexpr int kLdsBytes = kLdsTableBytes + kWavesPerBlock * kLdsWaveBytes;  // 160 KiB

// Layout of the constant blob (built on the host by build_tables()).
constexpr int kOffF1 = 0;        // 64 lanes x {RE-form 8 halfs, IM-form 8 halfs}
constexpr int kOffTw = 2048;     // 64 lanes x {float re[4], float im[4]}
constexpr int kOffG = 4096;      // 16 tiles x 64 lanes x 8 halfs (RE-form)
constexpr int kOffH = kOffG + 16384;
constexpr int kOffF1n = kOffH + 16384;   // as kOffF1 but slots in natural order (column-pass kernel)
constexpr int kOffWR = kOffF1n + 2048;   // radix-R front end of k4096r.hpp as an MFMA A operand: R = 2, 4, 8 -> 3 x 64 lanes x 8 B
// Column passes of radix 512 / 1024 (colfft.hpp, colfft1024.hpp): G with the combine twiddle of a decimated sequence folded
// in, G_q[ka][n1][kb] = w_R^(q (ka + 16 kb)) G_ka[n1][kb]: one rounding of the constant instead of an fp32 multiply per output.
constexpr int kOffG1024 = kOffWR + 3 * 512;     // R = 1024: q = 0 .. 3, 4 x 16 KiB (q = 0 differs from G by the scale only)
constexpr int kOffG512 = kOffG1024 + 4 * 16384; // R = 512: q = 0, 1, 2 x 16 KiB
constexpr int kTableBytes = kOffG512 + 2 * 16384;

// Contraction slot (lane group g, j) of stage 1 holds n2 = sigma(g, j): even rows
// for lanes 0-31, odd rows for lanes 32-63, so that each 32-lane half of a
// ds_read_b64_tr_b16 covers 8 distinct 32-byte bank ranges.
__host__ __device__ constexpr int sigma(int g, int j) { return 2 * j + 8 * (g & 1) + (g >> 1); }

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

// Where transform b of a batch starts inside a plane, in halves: (b >> gshift) * gstride + (b & gmask) * stride.
// Plain batches: gshift = 0, gmask = 0, gstride = stride. Grouped: the 2^gshift rows of one outer entry are `stride`
// apart and the outer entries `gstride` apart (row pass of a transposed-order plan: rows of a [N1][N2] matrix inside
// each [RE | IM] block).
struct Addr {
  uint64_t stride, gstride;
  uint32_t gshift, gmask;
  __host__ __device__ uint64_t off(uint32_t b) const {
    return static_cast<uint64_t>(b >> gshift) * gstride + static_cast<uint64_t>(b & gmask) * stride;
  }
};

// Output twiddle of the row transforms of a transposed-INPUT plan (tfft_plan_opts.input_order = TFFT_ORDER_TRANSPOSED, tfft.hip
// create_transposed_in): the caller's block is the [N1][N2] matrix in[k1 N2 + k2] = x[k1 + N1 k2]; transform b of the row pass is
// row k1 = b & row_mask of its matrix, and its output q must leave as  w_N^(k1 q) DFT_N2(row k1)[q]  so that one plain radix-N1
// column pass over k1 finishes X[q + N2 p] in natural order (decimation in time; the four-step twiddle sits between the two
// passes, and the only fp32 values between them are this kernel's final accumulators: no extra rounding). k1 q < N <= 2^24 is
// exact in fp32; v_sin / v_cos take revolutions.
struct OutTw {
  uint32_t n_mask = 0, row_mask = 0;
  float inv_n = 0.f;
};
struct Cf {
  float re, im;
};
// w_N^(row q)
__device__ __forceinline__ Cf otw_w(const OutTw& t, uint32_t row, uint32_t q) {
  const float a = static_cast<float>((row * q) & t.n_mask) * t.inv_n;
  return Cf{__builtin_amdgcn_cosf(a), -__builtin_amdgcn_sinf(a)};
}
__device__ __forceinline__ void cmul_to(float& re, float& im, const Cf w) {
  const float r = __builtin_fmaf(re, w.re, -(im * w.im));
  im = __builtin_fmaf(re, w.im, im * w.re);
  re = r;
}
__device__ __forceinline__ void otw_apply(const OutTw& t, uint32_t row, uint32_t q, float& re, float& im) {
  cmul_to(re, im, otw_w(t, row, q));
}

// ---------------------------------------------------------------------------
// host: constant operands
// ---------------------------------------------------------------------------
// Power-of-two factors folded into the constant operands (tfft_plan_opts.scale, include/tfft.h): f on the stage-1
// matrix F (both slot orders), g on G, h on H, tw on the fp32 inter-stage twiddle block. Sequential scaling = 1/16 per
// MFMA stage; unscaled = 1; the k4096r front end leaves a factor 1/2 of headroom that tw gives back (tw = 2).
struct TableScale {
  double f = 1.0 / 16, g = 1.0 / 16, h = 1.0 / 16, tw = 1.0;
  // G_q of the radix-1024 / radix-512 passes: g times the factor of their radix-4 / radix-2 combine (1/4, 1/2 sequential; 1
  // unscaled), so that the combines are plain sums
  double g1024 = 1.0 / 64, g512 = 1.0 / 32;
};

inline void build_tables(std::vector<uint8_t>& blob, const TableScale ts = TableScale()) {
  blob.assign(kTableBytes, 0);
  auto put_h = [&](int off, double v) {
    const _Float16 h = static_cast<_Float16>(v);
    std::memcpy(blob.data() + off, &h, 2);
  };
  auto put_f = [&](int off, double v) {
    const float f = static_cast<float>(v);
    std::memcpy(blob.data() + off, &f, 4);
  };
  auto cexp = [](long num, long den, double& c, double& s) {   // exp(-2 pi i num/den)
    const long r = ((num % den) + den) % den;
    const double a = -2.0 * M_PI * static_cast<double>(r) / static_cast<double>(den);
    c = std::cos(a);
    s = std::sin(a);
  };
  for (int lane = 0; lane < 64; ++lane) {
    const int g = lane >> 4, x = lane & 15;
    // stage 1: rows k0 = x, slots n2 = sigma(g, j)
    for (int j = 0; j < 4; ++j) {
      double c, s;
      cexp(static_cast<long>(sigma(g, j)) * x, 16, c, s);
      c *= ts.f;
      s *= ts.f;
      const int base = kOffF1 + lane * 32;
      put_h(base + 2 * j, c);              // RE-form  [ C_re | -C_im ]
      put_h(base + 2 * (4 + j), -s);
      put_h(base + 16 + 2 * j, s);         // IM-form  [ C_im |  C_re ]
      put_h(base + 16 + 2 * (4 + j), c);
    }
    for (int j = 0; j < 4; ++j) {          // natural slot order: contraction index 4g + j
      double c, s;
      cexp(static_cast<long>(4 * g + j) * x, 16, c, s);
      c *= ts.f;
      s *= ts.f;
      const int base = kOffF1n + lane * 32;
      put_h(base + 2 * j, c);
      put_h(base + 2 * (4 + j), -s);
      put_h(base + 16 + 2 * j, s);
      put_h(base + 16 + 2 * (4 + j), c);
    }
    // elementwise twiddle w256^(n0 k1), n0 = 4 g + r, k1 = x
    for (int r = 0; r < 4; ++r) {
      double c, s;
      cexp(static_cast<long>(4 * g + r) * x, 256, c, s);
      put_f(kOffTw + lane * 32 + 4 * r, c * ts.tw);
      put_f(kOffTw + lane * 32 + 16 + 4 * r, s * ts.tw);
    }
    // Radix-R butterfly of k4096r.hpp as a 16 x 16 real matrix (A operand of v_mfma_f32_16x16x16_f16: lane = row, 4 k-slots
    // per lane group). Rows rho' = 2R h' + 2 s2 + pl' (output s2, plane pl'), columns rho = 2R h + R pl + i (block i, plane
    // pl); h, h' < 8 / R number the independent column sets that share one product (block diagonal). Entry = the real
    // 2 x 2 form of w_R^(i s2) / (2 R): the 1 / (2 R) is the front end's scaling incl. its factor 1/2 of headroom.
    for (int lr = 0; lr < 3; ++lr) {
      const int R = 2 << lr;
      for (int j = 0; j < 4; ++j) {
        const int rp = x, rho = 4 * g + j;
        const int hp = rp / (2 * R), s2 = (rp % (2 * R)) >> 1, plp = rp & 1;
        const int h = rho / (2 * R), pl = (rho % (2 * R)) / R, i = rho % R;
        double c, sn, v = 0.0;
        cexp(static_cast<long>(i) * s2, R, c, sn);
        c /= 2.0 * R;
        sn /= 2.0 * R;
        if (h == hp) v = (plp == 0) ? (pl == 0 ? c : -sn) : (pl == 0 ? sn : c);
        put_h(kOffWR + lr * 512 + lane * 8 + 2 * j, v);
      }
    }
    for (int k0 = 0; k0 < 16; ++k0)
      for (int j = 0; j < 4; ++j) {
        const int idx = 4 * g + j;   // contraction index n1 (G) or n0 (H)
        double c, s;
        cexp(static_cast<long>(idx) * (k0 + 16 * x), 256, c, s);      // G_k0[n1][k1 = x]
        put_h(kOffG + k0 * 1024 + lane * 16 + 2 * j, c * ts.g);
        put_h(kOffG + k0 * 1024 + lane * 16 + 2 * (4 + j), -s * ts.g);
        cexp(static_cast<long>(idx) * (k0 + 256 * x), 4096, c, s);    // H_k0[n0][k2 = x]
        put_h(kOffH + k0 * 1024 + lane * 16 + 2 * j, c * ts.h);
        put_h(kOffH + k0 * 1024 + lane * 16 + 2 * (4 + j), -s * ts.h);
        for (int q = 0; q < 4; ++q) {                                 // (4 n1 + q)(k0 + 16 x) / 1024
          cexp(static_cast<long>(4 * idx + q) * (k0 + 16 * x), 1024, c, s);
          put_h(kOffG1024 + q * 16384 + k0 * 1024 + lane * 16 + 2 * j, c * ts.g1024);
          put_h(kOffG1024 + q * 16384 + k0 * 1024 + lane * 16 + 2 * (4 + j), -s * ts.g1024);
        }
        for (int q = 0; q < 2; ++q) {                                 // (2 n1 + q)(k0 + 16 x) / 512
          cexp(static_cast<long>(2 * idx + q) * (k0 + 16 * x), 512, c, s);
          put_h(kOffG512 + q * 16384 + k0 * 1024 + lane * 16 + 2 * j, c * ts.g512);
          put_h(kOffG512 + q * 16384 + k0 * 1024 + lane * 16 + 2 * (4 + j), -s * ts.g512);
        }
      }
  }
}

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
#define TFFT_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#define TFFT_GLB(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ uint32_t pk(float lo, float hi) {
  const h2 v = {static_cast<_Float16>(lo), static_cast<_Float16>(hi)};   // v_cvt_pk_f16_f32, RNE
  return __builtin_bit_cast(uint32_t, v);
}

// [C_re | -C_im]  ->  [C_im | C_re]
__device__ __forceinline__ h8 im_form(u4 re_form) {
  const u4 v = {re_form.z ^ 0x80008000u, re_form.w ^ 0x80008000u, re_form.x, re_form.y};
  return __builtin_bit_cast(h8, v);
}

// out[a][lane group b] = in[b][lane group a] over the four 16-lane groups.
__device__ __forceinline__ void transpose4(uint32_t& r0, uint32_t& r1, uint32_t& r2, uint32_t& r3) {
  auto s01 = __builtin_amdgcn_permlane16_swap(r0, r1, false, false);
  auto s23 = __builtin_amdgcn_permlane16_swap(r2, r3, false, false);
  auto t02 = __builtin_amdgcn_permlane32_swap(s01[0], s23[0], false, false);
  auto t13 = __builtin_amdgcn_permlane32_swap(s01[1], s23[1], false, false);
  r0 = t02[0];
  r2 = t02[1];
  r1 = t13[0];
  r3 = t13[1];
}

__device__ __forceinline__ f4 mfma(h8 a, h8 b) {
  const f4 z = {0.f, 0.f, 0.f, 0.f};
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, z, 0, 0, 0);
}

// Work distribution of the persistent (grid-stride) kernels: in round t workgroup g takes item t G + (g + t) mod G instead
// of t G + g. With the plain stride a workgroup only ever sees items of ONE residue class modulo 8 (G = 256 CUs): for a column
// pass that is one value of the address bits 7..9 of its 128-byte row segments, and the segments with bits 7..9 = 011 are
// served ~23 % more slowly by this memory system than the other seven classes (measured per workgroup with wall_clock64:
// the 32 workgroups of that class finish at 1.91 ms, all others at 1.53 ms; with the rotation everything ends at 1.50-1.53
// ms: 2^20 x 1024 295 -> 346 Gsamples/s). Rotating makes every workgroup take all eight classes in turn.
struct Rotor {
  uint32_t g, n, pos, rnd;
  __host__ __device__ Rotor(uint32_t block, uint32_t grid) : g(block), n(grid), pos(block), rnd(0) {}
  __host__ __device__ uint32_t item() const { return rnd * n + pos; }
  __host__ __device__ uint32_t peek() const { return (rnd + 1) * n + (pos + 1 == n ? 0 : pos + 1); }      // next round's item
  __host__ __device__ void advance() {
    ++rnd;
    pos = (pos + 1 == n) ? 0 : pos + 1;
  }
};

// Kernel variants (tuner knob, see tfft_plan_opts::variant):
//   kPrefetch   issue the next transform's HBM->LDS copy as soon as stage 1 has read the
//               current one out of LDS, so it flies under stages 2/3 and the stores.
//   kStageOut   stage the spectrum through the wave's LDS region and store it as
//               full 1-KiB rows (mutually exclusive with kPrefetch: same LDS bytes).
//   kNonTemporal  nt cache policy on the streamed loads and stores (every byte is touched once).
// Only with -DTFFT_DEBUG_KERNELS (libtfft_debug.so for the drivers under tools/; the shipped library has no such code):
//   kFakeStore  timing experiment only (WRONG output): coalesced stores of the raw registers.
//   kNoCompute  timing experiment only (WRONG output): copy the LDS image straight out (data-movement ceiling).
enum : int { kPrefetch = 1, kStageOut = 2, kNonTemporal = 8 };
#ifdef TFFT_DEBUG_KERNELS
enum : int { kFakeStore = 4, kNoCompute = 64 };
#else
enum : int { kFakeStore = 0, kNoCompute = 0 };     // (V & 0): those branches do not exist in the shipped library
#endif

// LDS-DMA of one transform: 16 x global_load_lds_dwordx4 hidden from the compiler's
// wait-count bookkeeping (inline asm), so that the only waits are the counted ones below.
template <bool NT>
__device__ __forceinline__ void dma_in(const uint8_t* src_re, const uint8_t* src_im, uint32_t lds_off, int lane) {
  // block mm of a plane = rows n2 = 2mm, 2mm+1 = 64 chunks of 16 B; LDS slot l of the block
  // receives global chunk l ^ 2mm (swizzle applied on the source address).
#pragma unroll
  for (int mm = 0; mm < 8; ++mm) {
    const uint32_t chunk = static_cast<uint32_t>(mm * 64 + (lane ^ (2 * mm))) * 16u;
    const uint8_t* gr = src_re + chunk;
    const uint8_t* gi = src_im + chunk;
    const uint32_t d0 = lds_off + mm * 1024, d1 = lds_off + 8192 + mm * 1024;
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
}

// (KEEP: the row pass of a transposed-input plan writes an intermediate that the column pass of the same chunk reads right
// back: plain stores, so that it stays in the Infinity Cache)
template <int V, bool KEEP = false>
__device__ __forceinline__ void st(uint16_t* p, u4 v) {
  if ((V & kNonTemporal) && !KEEP)
    __builtin_nontemporal_store(v, reinterpret_cast<u4*>(p));
  else
    *reinterpret_cast<u4*>(p) = v;
}

// in_*/out_*: planar binary16; FFT b at +b*stride halves. tables: build_tables() blob.
template <int V, bool OTW = false>
__global__ __launch_bounds__(kThreads, 2) void fft4096_kernel(
    const uint16_t* in_re, const uint16_t* in_im, uint16_t* out_re, uint16_t* out_im, Addr in_map,
    Addr out_map, uint32_t batch, uint32_t live, const uint8_t* __restrict__ tables, OutTw otw) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  uint8_t* const wl = lds + kLdsTableBytes + wave * kLdsWaveBytes;
  // LDS byte address of this wave's region (M0 base of its LDS-DMA)
  const uint32_t wl_off = __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t*)wl)));

  // live = waves of a workgroup that take transforms: 8, or fewer for a batch that does not fill the chip (tfft.hip live_waves():
  // one wave per SIMD, on as many CUs as there are; N = 4096 x 8 ... x 1024: 8.4-8.7 us with eight waves per CU, 5.7-6.3 with up to
  // four). The other waves help to fill the tables and leave.
  const uint32_t stride_b = gridDim.x * live;
  uint32_t b = static_cast<uint32_t>(wave) < live ? blockIdx.x * live + wave : batch;

  // G and H into LDS, once per workgroup.
  for (int i = tid; i < kLdsTableBytes / 16; i += kThreads)
    reinterpret_cast<u4*>(lds)[i] = reinterpret_cast<const u4*>(tables + kOffG)[i];

  const h8 f_re = *reinterpret_cast<const h8*>(tables + kOffF1 + lane * 32);
  const h8 f_im = *reinterpret_cast<const h8*>(tables + kOffF1 + lane * 32 + 16);
  const f4 tw_re = *reinterpret_cast<const f4*>(tables + kOffTw + lane * 32);
  const f4 tw_im = *reinterpret_cast<const f4*>(tables + kOffTw + lane * 32 + 16);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // table loads retired: vmcnt below counts only loop traffic
  __syncthreads();
  if (b >= batch) return;
  // (Issuing this first copy ahead of the table fill was measured 9-20 % SLOWER at batch 65536: workgroups then start their HBM
  // reads in lock-step, profiles/r1_k4096_grid_scan.txt. Round 5 tried it again for ONE transform, where it would save a memory
  // round trip on paper: 5.5 us per transform against 4.9 in this order, 6.2 when the idle waves copied as well. Not kept.)
  dma_in<(V & kNonTemporal) != 0>(reinterpret_cast<const uint8_t*>(in_re + in_map.off(b)),
           reinterpret_cast<const uint8_t*>(in_im + in_map.off(b)), wl_off, lane);

  const uint8_t* const g_tab = lds + lane * 16;
  const uint8_t* const h_tab = lds + 16384 + lane * 16;

  // transposed-read geometry: lane = 16 g + 4 q + p reads row n2 = sigma(g, q),
  // columns n0 = 4p..4p+3 of tile n1. Row n2 = 2m + bb sits in 1-KiB block m.
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int m = q + 4 * (g & 1), bb = g >> 1;
  uint8_t* const tr_base = wl + m * 1024 + bb * 512 + 8 * p;

  // output geometry: lane (k1 = lane & 15, g) owns k2 = 4 g + r, k0 = 0..15
  const uint32_t out_lane_off = 16u * (lane & 15) + 1024u * g;   // halves, + 256 r + k0

  bool first = true;

  // Loop shape: the only exits lie BEFORE an iteration's look-ahead copy is issued (`if (nb >= batch) break`), so no
  // control-flow path leads from an LDS-DMA to the end of the program without passing the s_waitcnt vmcnt(0) at the loop
  // top (tools/isa_lint.py checks exactly that on the disassembly; the prefetch variants, whose look-ahead sits in the
  // middle of the body, drain explicitly behind the loop).
  for (;;) {
    // The 16 copies of this transform are the oldest outstanding vector-memory operations; with
    // prefetch the previous iteration issued its 16 output stores after them, and vmcnt retires in order.
    if ((V & kPrefetch) && !first)
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    first = false;

    if (V & kNoCompute) {
      uint16_t* const f_re = out_re + out_map.off(b);
      uint16_t* const f_im = out_im + out_map.off(b);
      u4 vr[8], vi[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        vr[i] = *reinterpret_cast<const u4*>(wl + 1024 * i + 16 * lane);
        vi[i] = *reinterpret_cast<const u4*>(wl + 8192 + 1024 * i + 16 * lane);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const uint32_t nb0 = b + stride_b;
      if (nb0 < batch)
        dma_in<(V & kNonTemporal) != 0>(reinterpret_cast<const uint8_t*>(in_re + in_map.off(nb0)),
               reinterpret_cast<const uint8_t*>(in_im + in_map.off(nb0)), wl_off, lane);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        st<V, OTW>(f_re + 512 * i + 8 * lane, vr[i]);
        st<V, OTW>(f_im + 512 * i + 8 * lane, vi[i]);
      }
      if (nb0 >= batch) break;
      b = nb0;
      continue;
    }
    // ---- stage 1: D1_n1[k0 = 4g + r][n0 = lane & 15], packed over tile pairs
    uint32_t pr[8][4], pi[8][4];   // [t = n1 >> 1][r]: lo half n1 = 2t, hi half n1 = 2t + 1
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      f4 dre[2], dim[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int n1 = 2 * t + e;
        uint8_t* a = tr_base + 32 * (n1 ^ m);
        const s4 xr = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s4*)(a));
        const s4 xi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s4*)(a + 8192));
        const u4 raw = {__builtin_bit_cast(u2, xr).x, __builtin_bit_cast(u2, xr).y,
                        __builtin_bit_cast(u2, xi).x, __builtin_bit_cast(u2, xi).y};
        const h8 x = __builtin_bit_cast(h8, raw);
        dre[e] = mfma(f_re, x);
        dim[e] = mfma(f_im, x);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pr[t][r] = pk(dre[0][r], dre[1][r]);
        pi[t][r] = pk(dim[0][r], dim[1][r]);
      }
    }

    const uint32_t nb = b + stride_b;
    if (V & kPrefetch) {
      // every transposed read above has returned (its data fed an MFMA whose result is consumed
      // below, but make it explicit) before the region is overwritten by the next transform
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (nb < batch)
        dma_in<(V & kNonTemporal) != 0>(reinterpret_cast<const uint8_t*>(in_re + in_map.off(nb)),
               reinterpret_cast<const uint8_t*>(in_im + in_map.off(nb)), wl_off, lane);
    }

    // ---- n1 high bits (register index a = t >> 1) <-> k0 high bits (lane group)
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        transpose4(pr[0 + pp][r], pr[2 + pp][r], pr[4 + pp][r], pr[6 + pp][r]);
        transpose4(pi[0 + pp][r], pi[2 + pp][r], pi[4 + pp][r], pi[6 + pp][r]);
      }
    // now pr[2a + pp][r] holds, for tile k0 = 4a + r, slots n1 = 4g + 2pp + {0,1}.

    uint16_t* const fft_re = out_re + out_map.off(b);
    uint16_t* const fft_im = out_im + out_map.off(b);
    Cf otw_t[4], otw_s;
    if (OTW) {
      const uint32_t row = b & otw.row_mask;
      otw_s = otw_w(otw, row, 1);
#pragma unroll
      for (int r2 = 0; r2 < 4; ++r2) otw_t[r2] = otw_w(otw, row, 16u * (lane & 15) + 256u * (4 * g + r2));
    }

    // ---- stages 2 and 3, tile by tile; 8 tiles fill one 16-byte output vector
    auto tile23 = [&](int k0, f4& o_re, f4& o_im) {
      const int a = k0 >> 2, r = k0 & 3;
      const u4 araw = {pr[2 * a][r], pr[2 * a + 1][r], pi[2 * a][r], pi[2 * a + 1][r]};
      const h8 aop = __builtin_bit_cast(h8, araw);
      const u4 graw = *reinterpret_cast<const u4*>(g_tab + k0 * 1024);
      const f4 e_re = mfma(aop, __builtin_bit_cast(h8, graw));
      const f4 e_im = mfma(aop, im_form(graw));
      // (e_re + i e_im) * w256^(n0 k1). Scalar fp32 on purpose (and the library is built with
      // -fno-slp-vectorize): packed v_pk_*_f32 sequences next to MFMAs were seen to drop an addend
      // intermittently on gfx950 (colfft.hpp twiddles), and they buy nothing here.
      f4 t_re, t_im;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        t_re[r4] = __builtin_fmaf(e_re[r4], tw_re[r4], -(e_im[r4] * tw_im[r4]));
        t_im[r4] = __builtin_fmaf(e_re[r4], tw_im[r4], e_im[r4] * tw_re[r4]);
      }
      const u4 braw = {pk(t_re[0], t_re[1]), pk(t_re[2], t_re[3]), pk(t_im[0], t_im[1]),
                       pk(t_im[2], t_im[3])};
      const h8 bop = __builtin_bit_cast(h8, braw);
      const u4 hraw = *reinterpret_cast<const u4*>(h_tab + k0 * 1024);
      o_re = mfma(__builtin_bit_cast(h8, hraw), bop);   // o[r2] = X[k0 + 16 k1 + 256 (4g + r2)]
      o_im = mfma(im_form(hraw), bop);
      if (OTW) {
        // output k0 + 16 k1 + 256 (4g + r2) of row `b & row_mask` times w_N^(row k): otw_t[r2] holds the twiddle of the CURRENT
        // tile (the tiles run k0 = 0, 1, ..., 15 in order) and steps by w_N^row: 10 v_sin / v_cos per transform instead of 128
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
      uint32_t ore[4][4], oim[4][4];   // [r2][k0 pair within this half]
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
        const u4 vr = {ore[r2][0], ore[r2][1], ore[r2][2], ore[r2][3]};
        const u4 vi = {oim[r2][0], oim[r2][1], oim[r2][2], oim[r2][3]};
        if (V & kFakeStore) {
          st<V, OTW>(fft_re + 8 * lane + 512 * (2 * r2 + half), vr);
          st<V, OTW>(fft_im + 8 * lane + 512 * (2 * r2 + half), vi);
        } else if (V & kStageOut) {
          // byte offset of this piece in the [RE 8 KiB | IM 8 KiB] image: 32 k1 + 16 half + 512 (4g + r2);
          // the 16-byte slot index (2 k1 + half) is XORed with bit 3 of itself so that lanes k1 and
          // k1 + 4 of one store group hit different banks.
          const uint32_t slot = 2u * (lane & 15) + half;
          const uint32_t off = 16u * (slot ^ ((slot >> 3) & 1)) + 512u * (4 * g + r2);
          *reinterpret_cast<u4*>(wl + off) = vr;
          *reinterpret_cast<u4*>(wl + 8192 + off) = vi;
        } else {
          st<V, OTW>(fft_re + out_lane_off + 256 * r2 + 8 * half, vr);
          st<V, OTW>(fft_im + out_lane_off + 256 * r2 + 8 * half, vi);
        }
      }
    }
    if (V & kStageOut) {
      // read the image back row by row (1 KiB per wave instruction) and store it coalesced
      const uint32_t rd = 16u * (lane ^ ((lane >> 3) & 1));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const u4 vr = *reinterpret_cast<const u4*>(wl + 1024 * i + rd);
        const u4 vi = *reinterpret_cast<const u4*>(wl + 8192 + 1024 * i + rd);
        st<V, OTW>(fft_re + 512 * i + 8 * lane, vr);
        st<V, OTW>(fft_im + 512 * i + 8 * lane, vi);
      }
    }
    if (nb >= batch) break;
    if (!(V & kPrefetch)) {
      if (V & kStageOut) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // image read out before reuse
      dma_in<(V & kNonTemporal) != 0>(reinterpret_cast<const uint8_t*>(in_re + in_map.off(nb)),
             reinterpret_cast<const uint8_t*>(in_im + in_map.off(nb)), wl_off, lane);
    }
    b = nb;
  }
  // prefetch / timing-only variants: their look-ahead copy is issued under a condition in the middle of the body; make the
  // drain explicit (costs the default kernel nothing: it is not instantiated with these bits)
  if (V & (kPrefetch | kNoCompute)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace k4096
