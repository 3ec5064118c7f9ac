// tfft_oracle.cpp — CPU oracle for the Tensor-FFT hot path.
//
// TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is part of the product: only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library, and only as the checker / reported CPU baseline. The HIP library in
// tensor-fft_amd/csrc never links, loads or calls it.
//
// PARITY STATUS: "parity unpinned" at the bit level. The reference
// (CPestka/Tensor-FFT) stores no golden vectors (its .dat outputs are
// git-ignored) and is CUDA-only (nvcuda::wmma, warp 32), so it can neither be
// built nor run in this image. What pins this oracle instead:
//   * the reference's own acceptance thresholds on DFT(x)/N
//     (src/testing/unitTesting/UnitTest.cu:14-16) on the reference's own test
//     signal (src/testing/TestingDataCreation.h:89-117) -- checked in tests/;
//   * GetRandomWeights() is pure libstdc++ (TestingDataCreation.h:15-27) and is
//     called here through the very same std:: classes, so the weight vectors are
//     the reference's by construction (values recorded in tests/golden/);
//   * closed-form known answers (impulse, constant, integer-frequency tones).
//
// Contents
//   1. software IEEE binary16 (gcc 11 has no _Float16 on x86-64)
//   2. orc_dft64 / orc_fft64      fp64 DFT(x)/N: the stand-in for the reference's
//                                 cuFFT Z2Z / N oracle (CuFFTTest.h:218-261,
//                                 AccuracyCalculator.h:70-84)
//   3. orc_ref_*                  restatement of the reference kernels' fp16
//                                 arithmetic, pass by pass (citations inline)
//   4. orc_random_weights / orc_sine_superposition   the reference test signal
//   5. orc_deviation_stats        max / mean / sigma of |delta|
//                                 (AccuracyCalculator.h:86-148)
//
// Build: see oracle/Makefile (g++ -O2 -fopenmp -shared -fPIC).

#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>
#include <algorithm>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------------------
// 1. binary16 in software, round-to-nearest-even
// ---------------------------------------------------------------------------
inline double h2d(uint16_t h) {
  const int sign = h >> 15;
  const int e = (h >> 10) & 31;
  const int m = h & 1023;
  double v;
  if (e == 0) {
    v = std::ldexp(static_cast<double>(m), -24);
  } else if (e == 31) {
    v = m ? NAN : INFINITY;
  } else {
    v = std::ldexp(static_cast<double>(m + 1024), e - 25);
  }
  return sign ? -v : v;
}

inline uint16_t d2h(double x) {
  const uint16_t sign = std::signbit(x) ? 0x8000u : 0u;
  const double a = std::fabs(x);
  if (std::isnan(a)) return sign | 0x7e00u;
  if (a >= 65520.0) return sign | 0x7c00u;      // rounds to infinity
  if (a == 0.0) return sign;
  int e;
  std::frexp(a, &e);                            // a = m * 2^e, m in [0.5,1)
  int E = e - 1;                                // a in [2^E, 2^(E+1))
  if (E < -14) E = -14;                         // subnormal quantum 2^-24
  // nearbyint honours the default rounding mode: to nearest, ties to even.
  long q = std::lrint(std::nearbyint(std::ldexp(a, 10 - E)));
  if (q == 2048) { q = 1024; ++E; }
  if (q < 1024) return sign | static_cast<uint16_t>(q);      // subnormal / zero
  return sign | static_cast<uint16_t>(((E + 15) << 10) | (q - 1024));
}

// A value that *is* a binary16 number, carried in a double. Every arithmetic
// helper rounds its exact (double) result once to binary16, which is what the
// CUDA half intrinsics do. Sums/products of binary16 numbers are exact in
// double (<= 41 / 22 significant bits); fma goes through std::fma.
struct H {
  double v;
};
inline H hq(double x) { return H{h2d(d2h(x))}; }
inline H hadd(H a, H b) { return hq(a.v + b.v); }
inline H hsub(H a, H b) { return hq(a.v - b.v); }
inline H hmul(H a, H b) { return hq(a.v * b.v); }
inline H hfma(H a, H b, H c) { return hq(std::fma(a.v, b.v, c.v)); }
inline H hdiv(H a, double d) { return hq(a.v / d); }

// Complex scalar twiddle multiply, rounding exactly as the reference does
// (TensorFFT256.cu:246-253, TensorFFT4096.cu:344-350, TensorRadix16.cu:138-144,
//  Radix2.cu:45-48): re = hsub(hmul(a,c), hmul(b,d)); im = hfma(a, d, hmul(b,c)).
inline void twiddle_mul(H a, H b, H c, H d, H& re, H& im) {
  re = hsub(hmul(a, c), hmul(b, d));
  im = hfma(a, d, hmul(b, c));
}

// cosf/-sinf of a float phase, rounded to binary16: how every kernel of the
// reference builds its DFT matrix and twiddles (e.g. TensorFFT256.cu:56-69).
inline void trig_half(float phase, H& c, H& s) {
  c = hq(static_cast<double>(cosf(phase)));
  s = hq(static_cast<double>(-sinf(phase)));
}

struct Dft16 {
  H re[16][16], im[16][16];
  Dft16() {
    for (int j = 0; j < 16; ++j)
      for (int i = 0; i < 16; ++i) {
        // float phase = (float(j*i) * M_PI) / 8.0  -- double math, float store
        const float phase =
            static_cast<float>((static_cast<double>(static_cast<float>(j * i)) * M_PI) / 8.0);
        trig_half(phase, re[j][i], im[j][i]);
      }
  }
};

// One "tensor core" complex tile product C = A * F on 16x16 tiles with binary16
// accumulator fragments (TensorFFT256.cu:87-97,191-215):
//   RE1 = A_re*F_re ; RE2 = A_im*F_im ; IM = A_im*F_re ; IM += A_re*F_im ;
//   RE = hsub(RE1, RE2).
// Model of one mma_sync with half accumulators: exact products, one wide sum
// including the incoming accumulator, one rounding to binary16. The real HMMA
// datapath is undocumented; this is its ideal form.
inline H mma_dot(const H* a_row, const H (*f)[16], int col, H acc) {
  double s = acc.v;
  for (int k = 0; k < 16; ++k) s += a_row[k].v * f[k][col].v;
  return hq(s);
}

void tile_product(const Dft16& F, const H a_re[16][16], const H a_im[16][16],
                  H c_re[16][16], H c_im[16][16]) {
  const H zero{0.0};
  for (int r = 0; r < 16; ++r)
    for (int c = 0; c < 16; ++c) {
      const H re1 = mma_dot(a_re[r], F.re, c, zero);
      const H re2 = mma_dot(a_im[r], F.im, c, zero);
      H im = mma_dot(a_im[r], F.re, c, zero);
      im = mma_dot(a_re[r], F.im, c, im);
      c_re[r][c] = hsub(re1, re2);
      c_im[r][c] = im;
    }
}

// Step A of the base kernels: where output slot o of the gathered buffer reads
// from (TensorFFT256.cu:125-160 == TensorFFT4096.cu:128-163).
inline uint64_t gather_index(uint64_t o, int r16, int r2) {
  uint64_t t = o;
  uint64_t inp = 16 * (t % 16);
  t /= 16;
  inp += t % 16;
  for (int i = 1; i < r16; ++i) {
    t /= 16;
    inp = 16 * inp + (t % 16);
  }
  if (r2 > 0) {
    t /= 16;
    inp = 2 * inp + (t % 2);
    for (int i = 1; i < r2; ++i) {
      t /= 2;
      inp = 2 * inp + (t % 2);
    }
  }
  return inp;
}

// Steps A+B+C for one 256-point chunk (one warp of the reference):
// gather (scaled by 1/scale), DFT-16, twiddle+transpose, DFT-16.
// On return d_re/d_im hold D with X[i + 16k] = D[i][k] (TensorFFT256.cu:294-305).
void chunk256(const Dft16& F, const uint16_t* in_re, const uint16_t* in_im,
              uint64_t chunk, int r16, int r2, double scale,
              H d_re[16][16], H d_im[16][16]) {
  static thread_local H a_re[16][16], a_im[16][16], c_re[16][16], c_im[16][16],
      t_re[16][16], t_im[16][16];
  for (int r = 0; r < 16; ++r)
    for (int c = 0; c < 16; ++c) {
      const uint64_t src = gather_index(chunk * 256 + 16 * r + c, r16, r2);
      a_re[r][c] = hdiv(H{h2d(in_re[src])}, scale);
      a_im[r][c] = hdiv(H{h2d(in_im[src])}, scale);
    }
  tile_product(F, a_re, a_im, c_re, c_im);
  // T[i][j] = C[j][i] * w_256^(i*j)   (TensorFFT256.cu:225-254)
  for (int j = 0; j < 16; ++j)
    for (int i = 0; i < 16; ++i) {
      const float phase =
          static_cast<float>((static_cast<double>(static_cast<float>(i * j)) * M_PI) / 128.0);
      H wc, ws;
      trig_half(phase, wc, ws);
      twiddle_mul(c_re[j][i], c_im[j][i], wc, ws, t_re[i][j], t_im[i][j]);
    }
  tile_product(F, t_re, t_im, d_re, d_im);
}

// TensorFFT256 (TensorFFT256.cu:20-306) over the whole length-N buffer.
void base256(const Dft16& F, const uint16_t* in_re, const uint16_t* in_im,
             uint16_t* out_re, uint16_t* out_im, uint64_t n, int r16, int r2) {
  H d_re[16][16], d_im[16][16];
  for (uint64_t w = 0; w < n / 256; ++w) {
    chunk256(F, in_re, in_im, w, r16, r2, 256.0, d_re, d_im);
    for (int i = 0; i < 16; ++i)
      for (int k = 0; k < 16; ++k) {
        out_re[256 * w + i + 16 * k] = d2h(d_re[i][k].v);
        out_im[256 * w + i + 16 * k] = d2h(d_im[i][k].v);
      }
  }
}

// TensorFFT4096 (TensorFFT4096.cu:22-413): 16 chunks ("warps") per block, then
// the in-block 256 -> 4096 combine (:314-412).
void base4096(const Dft16& F, const uint16_t* in_re, const uint16_t* in_im,
              uint16_t* out_re, uint16_t* out_im, uint64_t n, int r16, int r2) {
  std::vector<H> y_re(4096), y_im(4096);
  static thread_local H d_re[16][16], d_im[16][16], t_re[16][16], t_im[16][16],
      o_re[16][16], o_im[16][16];
  for (uint64_t blk = 0; blk < n / 4096; ++blk) {
    for (int q = 0; q < 16; ++q) {
      chunk256(F, in_re, in_im, blk * 16 + q, r16, r2, 4096.0, d_re, d_im);
      for (int i = 0; i < 16; ++i)
        for (int k = 0; k < 16; ++k) {
          y_re[256 * q + i + 16 * k] = d_re[i][k];   // Y_q[m], m = i + 16k
          y_im[256 * q + i + 16 * k] = d_im[i][k];
        }
    }
    // warp p: T[a][q] = Y_q[a+16p] * w_4096^((a+16p)*q); D = T*F;
    // X[(a+16p) + 256k] = D[a][k].   phase = float(m*q) * pi / 2048 (:331-332)
    for (int p = 0; p < 16; ++p) {
      for (int a = 0; a < 16; ++a)
        for (int q = 0; q < 16; ++q) {
          const int m = a + 16 * p;
          const float phase = static_cast<float>(
              (static_cast<double>(static_cast<float>(m * q)) * M_PI) / 2048.0);
          H wc, ws;
          trig_half(phase, wc, ws);
          twiddle_mul(y_re[256 * q + m], y_im[256 * q + m], wc, ws, t_re[a][q], t_im[a][q]);
        }
      tile_product(F, t_re, t_im, o_re, o_im);
      for (int a = 0; a < 16; ++a)
        for (int k = 0; k < 16; ++k) {
          const uint64_t dst = blk * 4096 + (a + 16 * p) + 256 * k;
          out_re[dst] = d2h(o_re[a][k].v);
          out_im[dst] = d2h(o_im[a][k].v);
        }
    }
  }
}

// TensorRadix16 (TensorRadix16.cu:36-214): one L -> 16L combine over the buffer.
void radix16_pass(const Dft16& F, const uint16_t* in_re, const uint16_t* in_im,
                  uint16_t* out_re, uint16_t* out_im, uint64_t n, uint64_t L) {
  const uint64_t combined = 16 * L;
  const uint64_t warps_per_sub = L / 16;
  static thread_local H t_re[16][16], t_im[16][16], d_re[16][16], d_im[16][16];
  for (uint64_t w = 0; w < n / 256; ++w) {
    const uint64_t u = w % warps_per_sub;
    const uint64_t s = w / warps_per_sub;
    for (int a = 0; a < 16; ++a)
      for (int j = 0; j < 16; ++j) {
        const uint64_t i = a + 16 * u;
        const uint64_t g = i + L * j + combined * s;
        // float tmp = float(i*j) / float(combined); float phase = 2.0*M_PI*tmp
        // (TensorRadix16.cu:117-121). i*j is an int product in the reference.
        const float tmp = static_cast<float>(static_cast<int64_t>(i * j)) /
                          static_cast<float>(combined);
        const float phase = static_cast<float>(2.0 * M_PI * static_cast<double>(tmp));
        H wc, ws;
        trig_half(phase, wc, ws);
        const H xr = hdiv(H{h2d(in_re[g])}, 16.0);
        const H xi = hdiv(H{h2d(in_im[g])}, 16.0);
        twiddle_mul(xr, xi, wc, ws, t_re[a][j], t_im[a][j]);
      }
    tile_product(F, t_re, t_im, d_re, d_im);
    for (int a = 0; a < 16; ++a)
      for (int k = 0; k < 16; ++k) {
        const uint64_t dst = (a + 16 * u) + L * k + combined * s;
        out_re[dst] = d2h(d_re[a][k].v);
        out_im[dst] = d2h(d_im[a][k].v);
      }
  }
}

// Radix2Kernel (Radix2.cu:20-77) applied to every pair of sub-FFTs of one pass
// (the host loop is ComputeFFT.h:123-145).
void radix2_pass(const uint16_t* in_re, const uint16_t* in_im, uint16_t* out_re,
                 uint16_t* out_im, uint64_t n, uint64_t L) {
  const H half_{0.5};
  for (uint64_t base = 0; base < n; base += 2 * L)
    for (uint64_t t = 0; t < L; ++t) {
      const float tmp = static_cast<float>(t) / static_cast<float>(L);
      const float phase = static_cast<float>(M_PI * static_cast<double>(tmp));
      H wc, ws;
      trig_half(phase, wc, ws);
      const H p2r{h2d(in_re[base + t + L])}, p2i{h2d(in_im[base + t + L])};
      H mr, mi;
      twiddle_mul(p2r, p2i, wc, ws, mr, mi);
      const H p1r{h2d(in_re[base + t])}, p1i{h2d(in_im[base + t])};
      out_re[base + t] = d2h(hmul(hadd(p1r, mr), half_).v);
      out_im[base + t] = d2h(hmul(hadd(p1i, mi), half_).v);
      out_re[base + t + L] = d2h(hmul(hsub(p1r, mr), half_).v);
      out_im[base + t + L] = d2h(hmul(hsub(p1i, mi), half_).v);
    }
}

inline int ilog2(uint64_t x) {
  int l = 0;
  while ((x >> l) > 1) ++l;
  return l;
}

// ---------------------------------------------------------------------------
// 2. fp64 transforms
// ---------------------------------------------------------------------------
void fft64_inplace(std::vector<double>& re, std::vector<double>& im,
                   const std::vector<double>& wr, const std::vector<double>& wi) {
  const uint64_t n = re.size();
  for (uint64_t i = 1, j = 0; i < n; ++i) {     // bit reversal
    uint64_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      std::swap(re[i], re[j]);
      std::swap(im[i], im[j]);
    }
  }
  for (uint64_t len = 2; len <= n; len <<= 1) {
    const uint64_t step = n / len;
    for (uint64_t i = 0; i < n; i += len)
      for (uint64_t k = 0; k < len / 2; ++k) {
        const double c = wr[k * step], s = wi[k * step];
        const double ur = re[i + k], ui = im[i + k];
        const double xr = re[i + k + len / 2], xi = im[i + k + len / 2];
        const double vr = xr * c - xi * s, vi = xr * s + xi * c;
        re[i + k] = ur + vr;
        im[i + k] = ui + vi;
        re[i + k + len / 2] = ur - vr;
        im[i + k + len / 2] = ui - vi;
      }
  }
}

// binary16 -> double through a 64 Ki-entry table, and per-thread scratch, so that the timed CPU baseline
// (bench.py cpu_baseline) measures the transform and not ldexp() or the allocator.
const double* h2d_table() {
  static const std::vector<double> table = [] {
    std::vector<double> t(65536);
    for (int i = 0; i < 65536; ++i) t[i] = h2d(static_cast<uint16_t>(i));
    return t;
  }();
  return table.data();
}
std::vector<double>& scratch_re() {
  static thread_local std::vector<double> v;
  return v;
}
std::vector<double>& scratch_im() {
  static thread_local std::vector<double> v;
  return v;
}

}  // namespace

extern "C" {

uint16_t orc_f64_to_f16(double x) { return d2h(x); }
double orc_f16_to_f64(uint16_t h) { return h2d(h); }

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// Plan arithmetic of the reference (Plan.h:99-115). mode: 0 = Mode_256,
// 1 = Mode_4096. Returns 0 on success, nonzero for the cases CreatePlan rejects.
int orc_ref_plan(uint64_t n, int mode, int* r16_steps, int* r2_steps,
                 int* results_in_results) {
  if (n == 0 || (n & (n - 1))) return 1;        // Plan.h:85-88
  const int lg = ilog2(n);
  if (lg < 8) return 2;                         // Plan.h:92-96
  if (mode == 1 && n < 4096) return 3;          // Plan.h:102-106
  const int r16 = lg / 4 - 1, r2 = lg % 4;
  const int remaining = (mode == 0) ? (r16 + r2 - 1) : (r16 + r2 - 2);
  if (r16_steps) *r16_steps = r16;
  if (r2_steps) *r2_steps = r2;
  if (results_in_results) *results_in_results = (remaining % 2 == 0) ? 1 : 0;
  return 0;
}

// Digit-reversal gather index (exposed for the permutation tests).
uint64_t orc_ref_gather_index(uint64_t o, int r16_steps, int r2_steps) {
  return gather_index(o, r16_steps, r2_steps);
}

// Whole ComputeFFT(plan, DataHandler) of the reference (ComputeFFT.h:54-151) on a
// DataHandler-shaped buffer of 4*N halves: [in_RE | in_IM | out_RE | out_IM]
// (DataHandler.h:27-35). Like the reference, the input half is used as ping-pong
// scratch; *results_in_results says which half holds the answer afterwards.
int orc_ref_compute_fft(uint64_t n, int mode, uint16_t* buf4n, int* results_in_results) {
  int r16, r2, rir;
  const int rc = orc_ref_plan(n, mode, &r16, &r2, &rir);
  if (rc) return rc;
  static const Dft16 F;
  uint16_t* a_re = buf4n;               // "input" half
  uint16_t* a_im = buf4n + n;
  uint16_t* b_re = buf4n + 2 * n;       // "results" half
  uint16_t* b_im = buf4n + 3 * n;
  if (mode == 0)
    base256(F, a_re, a_im, b_re, b_im, n, r16, r2);
  else
    base4096(F, a_re, a_im, b_re, b_im, n, r16, r2);
  uint16_t *cur_re = b_re, *cur_im = b_im, *nxt_re = a_re, *nxt_im = a_im;
  uint64_t L = (mode == 0) ? 256 : 4096;
  for (int i = (mode == 0) ? 1 : 2; i < r16; ++i) {       // ComputeFFT.h:105-120
    radix16_pass(F, cur_re, cur_im, nxt_re, nxt_im, n, L);
    L *= 16;
    std::swap(cur_re, nxt_re);
    std::swap(cur_im, nxt_im);
  }
  for (int i = 0; i < r2; ++i) {                          // ComputeFFT.h:123-145
    radix2_pass(cur_re, cur_im, nxt_re, nxt_im, n, L);
    L *= 2;
    std::swap(cur_re, nxt_re);
    std::swap(cur_im, nxt_im);
  }
  if (results_in_results) *results_in_results = rir;
  return (cur_re == (rir ? b_re : a_re)) ? 0 : 100;       // self-check of the parity rule
}

// Convenience: batch of planar FFTs, result always delivered to out_*.
// in/out layout: FFT b at in_re + b*stride (halves), same for the others.
int orc_ref_fft(uint64_t n, uint64_t batch, int mode, const uint16_t* in_re,
                const uint16_t* in_im, uint64_t in_stride, uint16_t* out_re,
                uint16_t* out_im, uint64_t out_stride) {
  int rc_all = 0;
#pragma omp parallel for schedule(dynamic)
  for (int64_t b = 0; b < static_cast<int64_t>(batch); ++b) {
    std::vector<uint16_t> buf(4 * n);
    std::memcpy(buf.data(), in_re + b * in_stride, 2 * n);
    std::memcpy(buf.data() + n, in_im + b * in_stride, 2 * n);
    int rir = 1;
    const int rc = orc_ref_compute_fft(n, mode, buf.data(), &rir);
    if (rc) {
#pragma omp atomic write
      rc_all = rc;
      continue;
    }
    const uint16_t* res = buf.data() + (rir ? 2 * n : 0);
    std::memcpy(out_re + b * out_stride, res, 2 * n);
    std::memcpy(out_im + b * out_stride, res + n, 2 * n);
  }
  return rc_all;
}

// fp64 DFT(x)/N, forward sign exp(-2 pi i jk/N). algo 0: naive O(N^2) with exact
// angle reduction (the plumbing reference of BASELINE config 1); algo 1: radix-2
// FFT. threads <= 0: all OpenMP threads.
int orc_dft64(uint64_t n, uint64_t batch, const uint16_t* in_re, const uint16_t* in_im,
              uint64_t in_stride, double* out_re, double* out_im, uint64_t out_stride,
              int algo, int threads) {
  if (n == 0 || (n & (n - 1))) return 1;
#ifdef _OPENMP
  const int nt = threads > 0 ? threads : omp_get_max_threads();
#else
  const int nt = 1;
  (void)threads;
#endif
  std::vector<double> wr(n), wi(n);
  for (uint64_t k = 0; k < n; ++k) {
    const double ang = -2.0 * M_PI * static_cast<double>(k) / static_cast<double>(n);
    wr[k] = std::cos(ang);
    wi[k] = std::sin(ang);
  }
  const double inv = 1.0 / static_cast<double>(n);
#pragma omp parallel for schedule(dynamic) num_threads(nt)
  for (int64_t b = 0; b < static_cast<int64_t>(batch); ++b) {
    const uint16_t* xr = in_re + b * in_stride;
    const uint16_t* xi = in_im + b * in_stride;
    double* yr = out_re + b * out_stride;
    double* yi = out_im + b * out_stride;
    std::vector<double>& ar = scratch_re();
    std::vector<double>& ai = scratch_im();
    ar.resize(n);
    ai.resize(n);
    const double* lut = h2d_table();
    for (uint64_t j = 0; j < n; ++j) {
      ar[j] = lut[xr[j]];
      ai[j] = lut[xi[j]];
    }
    if (algo == 0) {
      for (uint64_t k = 0; k < n; ++k) {
        double sr = 0.0, si = 0.0;
        for (uint64_t j = 0; j < n; ++j) {
          const uint64_t t = (j * k) & (n - 1);
          sr += ar[j] * wr[t] - ai[j] * wi[t];
          si += ar[j] * wi[t] + ai[j] * wr[t];
        }
        yr[k] = sr * inv;
        yi[k] = si * inv;
      }
    } else {
      fft64_inplace(ar, ai, wr, wi);
      for (uint64_t k = 0; k < n; ++k) {
        yr[k] = ar[k] * inv;
        yi[k] = ai[k] * inv;
      }
    }
  }
  return 0;
}

// fp64 in, fp64 out, in place: x <- DFT(x)/N per row of a [batch][n] array (radix-2). The second axis of a 2D oracle
// transform, whose input is the fp64 result of the first axis (no reference counterpart: the reference's oracle is
// cuFFT Z2Z, CuFFTTest.h:218-261).
int orc_fft64_rows(uint64_t n, uint64_t batch, double* re, double* im, uint64_t stride) {
  if (n == 0 || (n & (n - 1))) return 1;
  std::vector<double> wr(n), wi(n);
  for (uint64_t k = 0; k < n; ++k) {
    const double ang = -2.0 * M_PI * static_cast<double>(k) / static_cast<double>(n);
    wr[k] = std::cos(ang);
    wi[k] = std::sin(ang);
  }
  const double inv = 1.0 / static_cast<double>(n);
#pragma omp parallel for schedule(dynamic)
  for (int64_t b = 0; b < static_cast<int64_t>(batch); ++b) {
    std::vector<double>& ar = scratch_re();
    std::vector<double>& ai = scratch_im();
    ar.assign(re + b * stride, re + b * stride + n);
    ai.assign(im + b * stride, im + b * stride + n);
    fft64_inplace(ar, ai, wr, wi);
    for (uint64_t k = 0; k < n; ++k) {
      re[b * stride + k] = ar[k] * inv;
      im[b * stride + k] = ai[k] * inv;
    }
  }
  return 0;
}

// CPU twin of the library's device-side input generator (tensor-fft_amd/csrc/synth.hpp, tfft_synth_uniform; restated
// here, nothing is shared with the product): sample (fft, plane, j) = binary16 of ((h >> 41) - 2^22 + 1/2) 2^-22 with
// h = mix(mix(seed + 0x9E3779B97F4A7C15 (fft + 1)) ^ (2 j + plane)), mix = the splitmix64 finaliser. out = [RE n | IM n]
// per transform.
static inline uint64_t synth_mix64(uint64_t z) {
  z ^= z >> 30;
  z *= 0xbf58476d1ce4e5b9ull;
  z ^= z >> 27;
  z *= 0x94d049bb133111ebull;
  z ^= z >> 31;
  return z;
}
void orc_synth_uniform(uint64_t n, uint64_t batch, uint64_t first_fft, uint64_t seed, uint16_t* out) {
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < static_cast<int64_t>(batch); ++b) {
    const uint64_t s1 = synth_mix64(seed + 0x9E3779B97F4A7C15ull * (first_fft + b + 1));
    for (uint32_t plane = 0; plane < 2; ++plane)
      for (uint64_t j = 0; j < n; ++j) {
        const uint64_t h = synth_mix64(s1 ^ (2 * j + plane));
        const double x = (static_cast<double>(static_cast<int64_t>(h >> 41) - (1 << 22)) + 0.5) / 4194304.0;
        out[(2 * b + plane) * n + j] = d2h(x);
      }
  }
}

// GetRandomWeights (TestingDataCreation.h:15-27): same std:: machinery.
void orc_random_weights(int count, int seed, float* out) {
  std::seed_seq seq = {seed};
  std::default_random_engine gen(seq);
  std::uniform_real_distribution<float> dist(-1.0, 1.0);
  for (int i = 0; i < count; ++i) out[i] = dist(gen);
}

// CreateSineSuperpostionKernel (TestingDataCreation.h:89-117): out = [RE | IM],
// re[n] = sum_f a_f * sinf(2 pi f n / N) summed in double, stored as binary16.
void orc_sine_superposition(uint64_t n, const float* w_re, const float* w_im,
                            int cutoff, uint16_t* out2n) {
#pragma omp parallel for
  for (int64_t t = 0; t < static_cast<int64_t>(n); ++t) {
    double acc_re = 0.0, acc_im = 0.0;
    for (int f = 0; f < cutoff; ++f) {
      const float s = sinf(static_cast<float>((2 * M_PI * f * static_cast<double>(t)) /
                                               static_cast<double>(n)));
      acc_re += w_re[f] * s;     // float * float, accumulated in double
      acc_im += w_im[f] * s;
    }
    out2n[t] = d2h(acc_re);
    out2n[t + n] = d2h(acc_im);
  }
}

// GetLargestDeviation / ComputeAverageDeviation / ComputeSigmaOfDeviation
// (AccuracyCalculator.h:86-148) over `count` = 2N real numbers: pairwise
// ("cascade") summation and the 1/(count-1) in sigma as there. The reference's
// cascade loop stops at tmp == 1 without the last fold (dev[0] += dev[1]), so
// its sums cover only the even-indexed half of the terms; this version does the
// last fold, i.e. returns the true mean/sigma, which are >= the reference's
// figures -- passing the reference thresholds here implies passing them there.
void orc_deviation_stats(const double* a, const double* b, uint64_t count,
                         double* max_dev, double* avg_dev, double* sigma_dev) {
  std::vector<double> d(count);
  double mx = 0.0;
  for (uint64_t i = 0; i < count; ++i) {
    d[i] = std::fabs(a[i] - b[i]);
    mx = std::max(mx, d[i]);
  }
  auto cascade = [&](std::vector<double> v) {
    uint64_t t = count / 2;
    while (t >= 1) {
      for (uint64_t i = 0; i < t; ++i) v[i] += v[i + t];
      if (t == 1) break;
      t /= 2;
    }
    return v[0];
  };
  const double avg = cascade(d) / static_cast<double>(count);
  std::vector<double> sq(count);
  for (uint64_t i = 0; i < count; ++i) sq[i] = (d[i] - avg) * (d[i] - avg);
  const double sig = std::sqrt(cascade(sq) / static_cast<double>(count - 1));
  if (max_dev) *max_dev = mx;
  if (avg_dev) *avg_dev = avg;
  if (sigma_dev) *sigma_dev = sig;
}

}  // extern "C"
