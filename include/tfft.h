/* tfft.h — C ABI of the MI355X (gfx950) tensor-core FFT library (libtfft.so).
 *
 * Drop-in boundary for ONE path of CPestka/Tensor-FFT: the src/base batched,
 * forward, 1/N-scaled, planar-fp16 complex-to-complex 1D FFT. Each entry point
 * names the reference interface it replaces (paths relative to the reference
 * repository root). Only plain pointers and sizes cross this boundary: device
 * pointers are raw HIP device addresses, `stream` is a hipStream_t passed as
 * void*. No torch / C++ types.
 *
 * Data contract (identical to the reference kernels' signature
 * `__half* in_RE, __half* in_IM, __half* out_RE, __half* out_IM`,
 * src/base/TensorFFT256.cu:21-24): split ("planar") IEEE binary16 planes; FFT b
 * of a batch lives at plane_ptr + b * batch_stride halves, where the default
 * batch_stride = 2*N reproduces DataBatchHandler's [fft0_RE|fft0_IM|fft1_RE|..]
 * block (src/base/DataHandler.h:105-114). Result = DFT(x)/N with the sign
 * exp(-2 pi i jk/N) (the reference's "sequential scaling",
 * src/base/TensorFFT4096.cu:169-173, TensorRadix16.cu:132-136, Radix2.cu:67-76).
 */
#ifndef TFFT_H_
#define TFFT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tfft_plan tfft_plan;

/* ABI version of this header. 1: the layouts of rounds 1-3 (tfft_plan_opts began with in_batch_stride, tfft_dist_geometry with n).
 * 2 (round 4 on): both structs begin with struct_size / reserved_ - an incompatible change of every field offset, which is why
 * the number exists: a binding compares tfft_abi_version() of the library it loaded with the TFFT_ABI_VERSION it was built
 * against and refuses to continue on a mismatch (include/tensor_fft.hpp and tensor-fft_amd/capi.py do). Appending fields, new
 * entry points and new flag / variant bits do NOT change it. */
#define TFFT_ABI_VERSION 2
int tfft_abi_version(void);

/* Status codes. 0 = success, mirroring the reference's `std::nullopt == OK`
 * convention (src/base/ComputeFFT.h:147-150); the message of the last failure on
 * the calling thread is available from tfft_last_error(). */
enum {
  TFFT_OK = 0,
  TFFT_ERR_NOT_POW2 = 1,   /* Plan.h:85-88  "Input size has to be a power of 2" */
  TFFT_ERR_TOO_SMALL = 2,  /* Plan.h:92-96  N < 256 for a reference-compatible plan */
  TFFT_ERR_MODE = 3,       /* Plan.h:102-106 Mode_4096 with N < 4096 */
  TFFT_ERR_GEOMETRY = 4,   /* Plan.h:128-133,162-167,184-190 divisibility of warps / r2 block */
  TFFT_ERR_ARG = 5,        /* null / misaligned pointer, bad stride, bad batch */
  TFFT_ERR_DEVICE = 6,     /* not a gfx950 / wave64 / 160 KiB-LDS device (Plan.h:257-296) */
  TFFT_ERR_HIP = 7,        /* a HIP runtime call failed; text in tfft_last_error() */
  TFFT_ERR_WORKSPACE = 8,  /* workspace needed but missing / too small */
  TFFT_ERR_COMM = 9        /* RCCL: library not loadable, a collective call failed, or no communicator where one is needed */
};

/* BaseFFTMode of the reference (src/base/Plan.h:14). */
enum { TFFT_MODE_256 = 0, TFFT_MODE_4096 = 1 };

/* Field-for-field image of `struct Plan<Integer>` (src/base/Plan.h:18-39) as
 * CreatePlan() fills it (Plan.h:77-194). The launch-geometry fields are kept
 * so that code which prints / tunes them still works; the MI355X kernels do
 * not consume them. */
typedef struct tfft_ref_plan {
  uint64_t fft_length;
  int amount_of_r16_steps;
  int amount_of_r2_steps;
  int base_fft_mode;
  int results_in_results;
  int base_fft_warps_per_block;
  int base_fft_blocksize;
  int base_fft_gridsize;
  int base_fft_shared_mem_in_bytes;
  int r16_warps_per_block;
  int r16_blocksize;
  int r16_gridsize;
  int r16_shared_mem_in_bytes;
  int r2_blocksize;
} tfft_ref_plan;

/* Replaces CreatePlan(fft_length, mode, base_fft_warps_per_block,
 * r16_warps_per_block, r2_blocksize) (src/base/Plan.h:77-194). Host only. Same
 * acceptance rules, same derived fields, same warp-count overrides (the
 * warnings the reference prints are returned through tfft_last_error()). */
int tfft_ref_create_plan(uint64_t fft_length, int mode, int base_fft_warps_per_block,
                         int r16_warps_per_block, int r2_blocksize, tfft_ref_plan* out);

/* Replaces PlanWorksOnDevice(plan, device_id) (src/base/Plan.h:257-296): the
 * CC>=8 / warp==32 checks become gfx950 / wave64 / 160 KiB LDS. Returns TFFT_OK
 * or TFFT_ERR_DEVICE / TFFT_ERR_HIP. */
int tfft_device_check(int device_id);

/* Replaces GetMaxNoOptInSharedMem(device_id) (src/base/Plan.h:298-303):
 * sharedMemPerBlock of the device, or -1 on error. */
int tfft_max_no_optin_shared_mem(int device_id);

/* Execution plan for `batch` transforms of length n on `device_id`.
 * n: power of two >= 2 (the reference-compatible shim enforces >= 256).
 * in/out_batch_stride: distance in halves between consecutive FFTs of a plane;
 * 0 selects the DataBatchHandler default 2*n (2*n*inner). Must be a multiple of 8.
 * The plan is immutable after creation and may be shared between host threads;
 * it owns small device-side constant tables (and a workspace if it allocated one).
 * A plan with tfft_plan_workspace_bytes() > 0 has ONE workspace: executions of such a plan
 * must not overlap in time (same stream, or one plan per stream). Single-pass plans
 * (N <= 32768 with a contiguous axis) have no such restriction. */
typedef struct tfft_plan_opts {
  uint32_t struct_size; /* sizeof(tfft_plan_opts) AS THE CALLER WAS COMPILED (TFFT_PLAN_OPTS_INIT / tfft_plan_opts_init set it).
                           From ABI version 2 on (TFFT_ABI_VERSION below: the version that put this field in front) the struct
                           grows only by appending fields, the library reads exactly struct_size bytes and takes every field
                           beyond them as 0: a caller built against THIS header or a later one keeps working against a newer
                           library. Accepted sizes: 72 (the whole struct of ABI 2) and its two documented prefixes, 48 (through
                           output_order) and 64 (+ fourstep_n, fourstep_col0), for bindings that declare only the leading
                           fields (tfft_plan_opts_known_size); 0 and anything else is refused with TFFT_ERR_ARG instead of being
                           read past its end. NOT compatible: binaries built against the ABI-1 header, whose struct began with
                           in_batch_stride (they are refused where the stride reads as an unknown size and cannot be told apart
                           where it reads as 48, 64 or 72: rebuild them; tfft_abi_version() lets a binding check). The
                           reference's boundary is source level (default arguments of CreatePlan, src/base/Plan.h:77-82) and
                           never had this problem; a C ABI does. */
  uint32_t reserved_;   /* must be 0 */
  uint64_t in_batch_stride;
  uint64_t out_batch_stride;
  uint64_t inner;       /* 0/1: plain batched transforms. C >= 8 (power of two): transform along a strided
                           axis, data [batch][n][C] with C independent columns innermost (2D column pass,
                           local passes of a distributed transform) */
  int preserve_input;   /* 0: the input planes may be used as scratch, exactly as the
                           reference does (ComputeFFT.h:89-93,118-119); 1: never written */
  int variant;          /* tuner / experiment knob, 0 = default (what tools/tuner.py writes as the sixth column).
                           Every value accepted here yields CORRECT spectra; bits select between equivalent kernels:
                           N == 4096 kernel: mask of 1 = prefetch the next transform under stages 2/3, 2 = stage
                           the output through LDS (full-row stores), 8 = non-temporal loads/stores, 16 = none of
                           these (default = 2|8; 1 and 2 exclude each other).
                           Any N: 32 = plain autosort chain (no column kernel); 2097152 = do not fuse the
                           radix-16 + radix-2/4 tail into one radix-32/64 pass; 8388608 = no radix-512 column
                           passes; 33554432 = no radix-1024 column passes; 134217728 = among the splits with the
                           fewest passes, the one with the most radix-1024 (then radix-512) passes instead of the
                           measured default; 16777216 = N = 8192..32768 as a column plan instead of the single-pass kernel (2^15: 512 x 64; together with 8388608 | 33554432: 256 x 128 with the cooperative radix-128 final pass); 4194304 = one butterfly per thread in the radix-2/4/8 tail pass
                           (and in the radix-64 pass of a small batch of 2^14), and the packed launch shape of the single-pass
                           kernels (eight working waves per workgroup) for a batch that does not fill the chip, which by default
                           spreads over the CUs with at most one working wave per SIMD (same results bit for bit).
                           268435456 = a final radix-512 pass by the other of its two kernels: the two-round kernel
                           (colfft512r.hpp: 128-column tiles, or 64-column tiles and two 4-wave workgroups per CU with bit
                           524288) where the 8-wave single-round kernel is the default, and vice versa (default: two-round
                           at row pitches of 256 and 512 columns).
                           Column passes: 131072 = per-wave kernel, 524288 = 4-wave cooperative workgroups,
                           262144 = plain (cached) global accesses, 536870912 = non-temporal (streaming) ones; neither:
                           by the plan's footprint, see tfft_plan_cache_policy(); 1048576 = 16-byte stores straight from
                           registers,
                           4096 / 8192 = per-wave kernel with LDS-staged stores / hardware sin-cos twiddles.
                           1073741824 = never the latency form of the radix-256 column pass (collat.hpp: one 64-column
                           block per workgroup of 4 or 8 waves, chosen for passes of at most 64 / 128 blocks): keep the
                           throughput kernels for such passes too (same accuracy bound; the twiddles round differently).
                           Unknown bits are rejected (TFFT_ERR_ARG), see tfft_variant_check().
                           Timing / debugging aids that give WRONG or partial results are NOT part of this field's
                           accepted values: 4 and 64 (N == 4096 kernel: fake stores / no compute), (p << 8), p = 1..15 (run
                           only the first p passes), 128 (skip inter-pass twiddles), 65536 (copy-only column pass).
                           They are honoured only when the environment variable TFFT_DEBUG_VARIANTS=1 is set in the
                           process that creates the plan (tools/pass_breakdown.py, tools/exp_bench.py); otherwise
                           tfft_plan_create() refuses them, so a stale tuner file cannot produce garbage silently. */
  int scale;            /* TFFT_SCALE_* below; 0 = the reference's sequential scaling, result DFT(x)/N */
  int output_order;     /* TFFT_ORDER_* below; 0 = natural order */
  uint64_t fourstep_n;  /* 0 = off. Otherwise (n = 256 or 512, inner = C >= 64 columns): output row k of column c is also
                           multiplied by w_M^(k (fourstep_col0 + c)), M = fourstep_n (a power of two >= n): the twiddle
                           step between the two FFT passes of a four-step transform of length M, fused into the column
                           pass. What TFFT_ORDER_TRANSPOSED plans use internally and what a transform distributed over
                           several GPUs needs in front of its all-to-all (tensor-fft_amd/distributed.py) */
  uint64_t fourstep_col0;
  uint32_t launch_iters; /* launch shape of the plan's grid-stride kernels, a tuner input (tools/tuner.py searches it per
                           (N, batch) and writes it as the seventh column of the tuner file): 0 = the library's measured default
                           per kernel; k = 1 .. 65534: a workgroup takes about k rounds of work (k transforms per wave of the
                           single-pass kernels, k column blocks per workgroup of the column passes) and retires, the hardware
                           dispatcher refilling CUs as they drain (at least one workgroup per CU while there is that much
                           work); TFFT_LAUNCH_PERSISTENT: one workgroup per CU for the whole batch. Never changes results (the
                           kernels stride over the batch): the best shape at batch 65536 is not the best at batch 64. */
  int input_order;      /* TFFT_ORDER_* below; 0 = natural order. TFFT_ORDER_TRANSPOSED: the input is the [N1][N2] matrix
                           that a TRANSPOSED-output plan of the same length leaves behind (read as samples:
                           in[k1 * N2 + k2] = x[k1 + N1 * k2]), and the result is in natural order: two passes over HBM for
                           2^16 <= N <= 2^24 (see TFFT_ORDER_* below) */
} tfft_plan_opts;
enum { TFFT_LAUNCH_PERSISTENT = 65535 };
/* Zero-initialised options that carry the compile-time size: `tfft_plan_opts o = TFFT_PLAN_OPTS_INIT;` */
#define TFFT_PLAN_OPTS_INIT {(uint32_t)sizeof(tfft_plan_opts)}
/* The same at run time, for bindings that build the struct by hand (cgo, ctypes, JNI): zeroes `bytes` bytes at opts and
 * writes struct_size = bytes. TFFT_ERR_ARG when `bytes` is not the size of a released layout. */
int tfft_plan_opts_init(tfft_plan_opts* opts, size_t bytes);
/* Host only: 1 if `bytes` is the size of a layout of tfft_plan_opts this library knows (48, 64 or 72, see struct_size), else 0. */
int tfft_plan_opts_known_size(size_t bytes);

/* tfft_plan_opts.scale — where the 1/N goes. The reference ships "sequential scaling" and keeps the other two as
 * commented-out variants (src/base/TensorFFT256.cu:163-177 "For unscaled results" / "For scaling in one step",
 * Radix2.cu:56-65). All three differ only by exact powers of two, so away from overflow / underflow they give the
 * same significands; what differs is the range of the fp16 intermediates:
 *   SEQUENTIAL  every radix-R stage divides by R: |intermediate| <= max|x| at every stage, no finite input overflows.
 *   NONE        no stage scales: result = DFT(x) (the cuFFT / hipFFT convention). fp16 intermediates grow by the
 *               radix per stage; nothing overflows as long as N * max|x| <= 65504 (worst case: a constant or a pure
 *               tone; white noise of rms s needs about 6 s sqrt(N) <= 65504). Beyond that the result holds inf / nan.
 *   ONCE        one scaling step: every stage unscaled and a single exact factor 1/N applied in fp32 at the last fp32
 *               multiply of the plan (the inter-stage twiddle of its last kernel). Result = DFT(x)/N. Small inputs
 *               keep their precision longest (nothing is pushed towards fp16 subnormals before the last stage); the
 *               stages in front of the scaling step can overflow when 16^s * max|x| > 65504 (s = stages before it). */
enum { TFFT_SCALE_SEQUENTIAL = 0, TFFT_SCALE_NONE = 1, TFFT_SCALE_ONCE = 2 };

/* tfft_plan_opts.output_order / input_order.
 *   NATURAL      out[k] = X[k] (what the reference produces, TensorRadix16.cu:169-213).
 *   TRANSPOSED   N = N1 * N2 (N2 = tfft_plan_transposed_n2(n), N1 = N / N2): out[k1 * N2 + k2] = X[k1 + N1 * k2], i.e.
 *                the [N1][N2] matrix of the four-step algorithm left un-transposed (the order DistributedFFT1D's
 *                "transposed" layout has). For 2^16 <= N <= 2^24 this takes TWO passes over HBM (one strided radix-N1
 *                column pass that also applies w_N^(k1 n2), one contiguous N2-point pass) where natural order needs
 *                three from 2^21 on (2^16 .. 2^20 take two passes either way): for callers that only reduce over the
 *                spectrum, index it through the map above, or work on it pointwise and transform back with a plan whose
 *                INPUT order is TRANSPOSED. Contiguous axis only (inner > 1 is refused). Lengths without a two-pass
 *                split fall back to NATURAL (tfft_plan_transposed_n2(n) == 0). variant: only the column-pass bits
 *                262144 / 524288 and the single-kernel bits of the N2 kernel (1 / 2 / 8 / 16, 1048576) are honoured,
 *                others are refused. Needs a workspace (tfft_plan_workspace_bytes). */
/*   input_order = TRANSPOSED (same N1, N2): the input block is that same [N1][N2] matrix, in[k1 * N2 + k2] = x[k1 + N1 * k2], and
 *                the result is in NATURAL order: N1 contiguous N2-point transforms whose fp32 epilogue applies the four-step
 *                twiddle w_N^(k1 q), then ONE radix-N1 column pass (decimation in time) = two passes over HBM for
 *                2^16 <= N <= 2^24. With tfft_exec_inverse on such a plan, "forward into the transposed order, multiply
 *                spectra pointwise, transform back" costs 2 + 2 passes where natural order needs 3 + 3 from 2^21 on.
 *                Lengths without the layout are refused (TFFT_ERR_ARG), as are TRANSPOSED on both sides, inner > 1,
 *                TFFT_SCALE_ONCE (the plan's last fp32 multiply lies in front of its last stage) and every variant bit
 *                but the column-pass bits 262144 / 524288. The reference has neither order (nor an inverse):
 *                src/base/TensorFFT256.cu:163-177 only comments on scaling. */
enum { TFFT_ORDER_NATURAL = 0, TFFT_ORDER_TRANSPOSED = 1 };

/* Host only. N2 of the TRANSPOSED order for length n (0: no two-pass split, natural order is produced). */
uint64_t tfft_plan_transposed_n2(uint64_t n);

/* Host only. Cache policy tfft_plan_create picks for the column passes of a NATURAL-order plan (n, inner, batch) when the
 * variant names neither: 1 = plain global accesses (the plan's footprint, 12 bytes per sample for input + output + workspace,
 * is small enough for the 256-MiB Infinity Cache to hold a useful part of it between the passes), 0 = non-temporal accesses
 * (larger plans stream; single-kernel lengths and strided axes always do). The thresholds are measured:
 * profiles/r4_cache_policy.txt. The results do not depend on the policy, only the time does. */
int tfft_plan_cache_policy(uint64_t n, uint64_t inner, uint64_t batch);

/* Host only. The planner bits tfft_plan_create gives a NATURAL-order plan whose caller left `variant` at 0. In this order:
 * (1) a wisdom line for (n, nearest batch within three octaves), see tfft_tuning_load() below; (2) 0 when the work (n * batch
 * samples) fills the chip - the splits behind variant 0 were measured at 2^30 samples per launch; (3) otherwise the bits of the
 * split with more, smaller workgroups (a single 2^20-point transform is 16 workgroups of the radix-1024 kernel on 256 CUs: 40 us;
 * as 256 x 256 x 16 with the latency column kernel it takes 18 us): 2^17 ... 2^21 up to 2^20 (2^17, 2^18) / 2^22 (2^19 ... 2^21)
 * samples per launch; 2^18 up to 2^22 samples: the other radix-512 kernel; 2^15 up to 8 transforms: 256 x 128 (8388608 |
 * 33554432 | 16777216: the latency column kernel + a workgroup-cooperative radix-128 pass, 7.3 us for one transform where the
 * single-pass kernel, one CU, takes 11.3); 2^14 / 2^13 up to 4 transforms: 256 x 64 / 256 x 32 the same way (8.4 -> 7.2, 7.6 -> 7.1 us). Measured limits:
 * profiles/r5_small_scan.txt. The
 * within-noise rules of round 4 (two 2^24, a single 2^25) are wisdom lines now (profiles/r5_TunerResults.dat), not code.
 * Independent of the variant, radix-256 column passes of at most 64 blocks (128 from a row pitch of 512 columns on) run as the
 * latency kernel (collat.hpp) unless the variant holds 1073741824.
 * tfft_plan_describe(n, inner, tfft_plan_default_variant(n, inner, batch), ...) is the decomposition such a plan gets. A caller
 * that names any variant bit itself gets exactly that variant. */
int tfft_plan_default_variant(uint64_t n, uint64_t inner, uint64_t batch);

/* ---- Tuner results as plan "wisdom". Replaces CreatePlan(fft_length, tuner_results_file) (src/base/Plan.h:197-255) at the level of
 * the C ABI: the reference reads its launch parameters for one length out of the file its tuner wrote (TunerSingleFFT.cu:10-56,
 * FileWriter.h:250-269: lines `N mode base_wpb r16_wpb r2_blocksize`). This library's tuners (examples/tuner_single_fft.cpp,
 * tools/tuner.py) write the same lines with three more columns, `variant launch_iters batch`, and a process loads them ONCE:
 * tfft_plan_create then gives every natural-order, contiguous-axis plan whose caller left `variant` and `launch_iters` at 0 the
 * line of its length whose batch is nearest on a log scale (at most a factor of 8 away; a line with batch 0 fits any batch). A
 * line with variant 0 keeps the library's default for that shape. Lines without a sixth column (a plain reference tuner file)
 * carry nothing for this library and are skipped. Host only, process-wide, thread-safe; plans created earlier are unaffected.
 * What belongs here rather than in the library's source: every choice whose gain is of the order of the box-to-box spread
 * (profiles/r5_TunerResults.dat: e.g. the split of exactly two 2^24-point transforms). */
int tfft_tuning_load(const char* path, int* lines_taken);     /* TFFT_ERR_ARG: unreadable file or a bad line (nothing is loaded then) */
int tfft_tuning_add(uint64_t n, uint64_t batch, int variant, uint32_t launch_iters);   /* one line; replaces an earlier (n, batch) */
void tfft_tuning_clear(void);
/* 1 and the line's values when a loaded line applies to (n, batch), else 0 */
int tfft_tuning_query(uint64_t n, uint64_t batch, int* variant, uint32_t* launch_iters);

/* Host only: TFFT_OK if `variant` is acceptable to tfft_plan_create for (n, inner): only documented bits, no
 * combination without a compiled kernel, and no WRONG-result debugging bit unless TFFT_DEBUG_VARIANTS=1 is set.
 * CreatePlan(N, tuner_file) of the shims runs it on the file's sixth column. */
int tfft_variant_check(uint64_t n, uint64_t inner, int variant);

int tfft_plan_create(uint64_t n, uint64_t batch, int device_id, const tfft_plan_opts* opts,
                     tfft_plan** out);
void tfft_plan_destroy(tfft_plan* plan);

/* Host-only: the pass decomposition tfft_plan_create would choose, as text ("col:256+tw col:512+tw autosort:64-tw",
 * "k4096:4096", ...: kernel family : radix, "+tw" = applies the next pass's input twiddles, "-tw" = expects them
 * applied). Touches no device, so the planner is testable without a GPU. */
int tfft_plan_describe(uint64_t n, uint64_t inner, int variant, char* buf, size_t bytes);

/* Number of passes over the data one tfft_exec makes (= kernel launches, except that a narrow column pass with a
 * ragged batch takes two) and the bytes of device scratch it needs beyond in/out (0 for N <= 32768 with a contiguous axis). If nonzero, either hand memory in
 * with tfft_plan_set_workspace(), call tfft_plan_prepare() once, or let the first tfft_exec hipMalloc it.
 * IN PLACE (out == in) a plan with an odd number >= 3 of passes needs a third buffer: a library-owned workspace grows to twice
 * tfft_plan_workspace_bytes() at the first such call (a hipMalloc: not under stream capture), a caller's workspace of twice that
 * size is used the same way, and with a smaller caller's workspace the chain starts from a copy of the input instead (one more
 * launch, and the [RE | IM] block layout with batch stride 2 n is then required). The reference's plans say where the spectrum
 * ends up (results_in_results_, src/base/Plan.h:141-145): for 2^18 and 2^21 that is the input half, i.e. in place. */
int tfft_plan_num_launches(const tfft_plan* plan);
size_t tfft_plan_workspace_bytes(const tfft_plan* plan);
int tfft_plan_set_workspace(tfft_plan* plan, void* device_ptr, size_t bytes);
/* Allocates the plan's own workspace NOW (no-op when it needs none or one was handed in), so that no later tfft_exec calls
 * hipMalloc: with it, execution is launches only from the first call on. */
int tfft_plan_prepare(tfft_plan* plan);

/* Replaces ComputeFFT(Plan&, const DataHandler&, int) and ComputeFFT(const Plan&,
 * const DataBatchHandler&, int) (src/base/ComputeFFT.h:54-151, 162-293): enqueues
 * the whole batch on `stream` (hipStream_t, may be NULL = default stream). Like the
 * single-FFT overload it does not synchronise (ComputeFFT.h:49-53); unlike the
 * batch overload it creates no streams. The result is always left in out_*
 * (results_in_results_ == true in reference terms); out may alias in (in place).
 * Pointers must be 16-byte aligned. */
int tfft_exec(const tfft_plan* plan, const void* in_re, const void* in_im, void* out_re,
              void* out_im, void* stream);

/* Inverse transform with the same plan: out = (1/N) sum_j x[j] exp(+2 pi i jk/N). The reference has no
 * inverse; it costs nothing here: forward transform with the RE and IM planes exchanged. */
int tfft_exec_inverse(const tfft_plan* plan, const void* in_re, const void* in_im, void* out_re,
                      void* out_im, void* stream);

/* 2D transform of `batch` images [rows][cols] (BASELINE config "2D 4096 x 4096, batch 64"; the reference has no 2D
 * entry point, this is its 1D path (ComputeFFT.h:162-293) used twice): a row pass over contiguous lines (N = cols,
 * batch * rows transforms) and a column pass along the strided axis (N = rows, inner = cols). Layout: fully planar,
 * in_re / in_im / out_re / out_im each point at batch * rows * cols halves, image after image, row-major.
 * Result = DFT2(x) / (rows * cols). rows, cols: powers of two, cols >= 8.
 * Scratch: tfft_plan2d_workspace_bytes() of device memory (an intermediate image set plus the column plan's own
 * scratch); hand it in with tfft_plan2d_set_workspace() or let the first execution hipMalloc it. 4096 x 4096 runs as two
 * fused passes, chunk by chunk of 4 images through ONE 256-MiB intermediate set (both passes of a chunk back to back: the
 * Infinity Cache still holds part of it), so its workspace does not grow with the batch. Exact in-place execution is allowed. */
typedef struct tfft_plan2d tfft_plan2d;
int tfft_plan2d_create(uint64_t rows, uint64_t cols, uint64_t batch, int device_id, tfft_plan2d** out);
void tfft_plan2d_destroy(tfft_plan2d* plan);
int tfft_plan2d_num_launches(const tfft_plan2d* plan);
size_t tfft_plan2d_workspace_bytes(const tfft_plan2d* plan);
int tfft_plan2d_set_workspace(tfft_plan2d* plan, void* device_ptr, size_t bytes);
int tfft_plan2d_exec(const tfft_plan2d* plan, const void* in_re, const void* in_im, void* out_re, void* out_im,
                     void* stream);
/* inverse 2D transform (planes exchanged on both sides, as tfft_exec_inverse) */
int tfft_plan2d_exec_inverse(const tfft_plan2d* plan, const void* in_re, const void* in_im, void* out_re, void* out_im,
                             void* stream);

/* out[b][a][c] = in[a][b][c] * w_n_tw^((e0 + b) * (a*C + c)), planar binary16, c contiguous (C % 8 == 0);
 * n_tw == 0: pure re-ordering. The pack / twiddle / unpack step around the single all-to-all of a transform
 * distributed over several GPUs; the reference has no counterpart (no multi-device path,
 * src/base/ComputeFFT.h:295-557 is commented out). Not in place. */
int tfft_permute_twiddle(const void* in_re, const void* in_im, void* out_re, void* out_im, uint64_t a,
                         uint64_t b, uint64_t c, uint64_t n_tw, uint64_t e0, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * One transform spread over the GPUs of a node (BASELINE configs[4b]: single N = 2^26 with ONE all-to-all over xGMI).
 * The reference has no counterpart: its multi-GPU code is commented out and ran one independent transform per device
 * (ComputeFFTMultiGPU, src/base/ComputeFFT.h:295-411; DataHandlerMultiGPU, src/base/DataHandler.h:168-403). Shape here: one
 * plan per rank (= per GPU; ranks may be processes, or the devices of one process), four-step FFT with N = N1 N2:
 *
 *   input  ("columns"):    rank p holds x[n1 N2 + p C + c], n1 < N1, c < C = N2 / P, as an [N1][C] matrix per plane
 *   tfft_dist_exec_pre       one radix-N1 column pass with the four-step twiddle fused in -> the plan's send buffers
 *   tfft_dist_exec_exchange  chunk q (K C halves per plane, K = N1 / P) of the send buffers -> rank q's receive buffers:
 *                            one ncclGroupStart / ncclSend + ncclRecv per peer and plane / ncclGroupEnd on `stream`
 *   tfft_dist_exec_post      N2-point row transforms straight out of the receive buffers (no re-order pass where the
 *                            row transform starts with a radix-256 / 512 column pass, see tfft_dist_geometry.reorder)
 *   output ("transposed"): rank q holds X[k1 + N1 k2], k1 = q K + k, as a [K][N2] matrix per plane
 *
 * tfft_dist_exec = the three in order on one stream, so the kernels and the collective are ordered by that stream. A
 * process that drives several devices itself runs the phases device by device and brackets the exchange calls of all its
 * devices with tfft_dist_group_start / _end (RCCL's rule for one thread and several communicators). Every buffer is allocated at
 * plan creation; the exec calls only enqueue. Result = DFT(x) / N like tfft_exec. RCCL is bound with dlopen at the first
 * tfft_dist_* call that needs it; plans for one rank (world = 1) or with comm = NULL never load it. */
typedef struct tfft_dist_plan tfft_dist_plan;

typedef struct tfft_dist_geometry {
  uint32_t struct_size; /* IN: sizeof(tfft_dist_geometry) as the caller was compiled (TFFT_DIST_GEOMETRY_INIT). The library fills
                           at most that many bytes, so an older, shorter struct is never written past its end; 0 or a size
                           below the round-4 layout is refused (TFFT_ERR_ARG) */
  uint32_t reserved_;
  uint64_t n, n1, n2;  /* N = N1 N2 */
  uint64_t cols;       /* C = N2 / world: columns per rank in the column pass */
  uint64_t rows;       /* K = N1 / world: rows per rank in the row pass */
  uint64_t chunk;      /* K C: halves per plane that go to each peer */
  int world, rank;
  int fused;           /* 1: the four-step twiddle rides in the column pass's epilogue (always, today) */
  int reorder;         /* 1: a re-order pass [p'][k][c] -> [k][p' C + c] runs in front of the row transforms; 0: they read the segments in place */
  int local_passes;    /* passes over this rank's N / world samples per transform */
  int slabs;           /* (since round 5; it occupies what was padding) column slabs of the plan: 1, or 2 / 4 with TFFT_DIST_SLABS_* */
} tfft_dist_geometry;
#define TFFT_DIST_GEOMETRY_INIT {(uint32_t)sizeof(tfft_dist_geometry)}

/* Host only: the split tfft_dist_plan_create would choose (TFFT_ERR_ARG when N < 256 * 64 * world). out->struct_size must
 * be set by the caller (see there). */
int tfft_dist_geometry_query(uint64_t n, int world, int rank, tfft_dist_geometry* out);

/* RCCL plumbing for callers that do not bind RCCL themselves. id: TFFT_DIST_ID_BYTES bytes (an ncclUniqueId) made on one rank,
 * carried to the others by the caller (file, socket, MPI, torch.distributed ...). *comm is an ncclComm_t. */
#define TFFT_DIST_ID_BYTES 128
int tfft_dist_unique_id(void* id128);
int tfft_dist_comm_create(int world, int rank, const void* id128, int device_id, void** comm);
int tfft_dist_comm_create_all(int ndev, const int* devices, void** comms);   /* one process, ndev devices (ncclCommInitAll) */
int tfft_dist_comm_destroy(void* comm);
int tfft_dist_comm_info(void* comm, int* count, int* rank);   /* ncclCommCount / ncclCommUserRank; either pointer may be NULL */
int tfft_dist_group_start(void);
int tfft_dist_group_end(void);
/* ncclGetVersion of the RCCL this process has bound (e.g. 22703 = 2.27.3), for run reports */
int tfft_dist_rccl_version(int* version);

/* comm: this rank's ncclComm_t (the caller's own or from tfft_dist_comm_create), or NULL: then tfft_dist_exec_exchange is
 * unavailable and the caller moves the chunks itself between _pre and _post (tfft_dist_plan_buffers; how the tests run
 * several ranks on one GPU, and how another transport than RCCL would plug in). */
/* flags: TFFT_DIST_SELF_VIA_COMM = the rank's own chunk travels through ncclSend / ncclRecv (to itself, inside the same group)
 * instead of a device-to-device copy. No use in production; it lets a box with ONE GPU run the RCCL path of the exchange
 * (world = 1: kernel -> collective -> kernel on one stream). */
/*        TFFT_DIST_CALLER_BUFFERS = the plan allocates NO exchange buffers of its own (4 planes of N / world halves: 512 MiB at
 * N = 2^26 on one rank): the caller hands its own in with tfft_dist_plan_set_buffers before the first execution, e.g. tensors that
 * a framework's own collective can send. */
/*        TFFT_DIST_SLABS_2 / TFFT_DIST_SLABS_4 (round 5) = the exchange overlaps the column pass: the rank's C columns are cut into 2 / 4
 *        slabs, each its own launch of the column kernel and its own ncclSend / ncclRecv group; tfft_dist_exec enqueues the groups on
 *        a second stream of the plan, each behind the event of its slab's column pass, so slab s crosses xGMI while slab s + 1 is
 *        being computed, and the row transforms wait for the last group. The exchange buffers then hold [peer][slab][k][c_s] (a
 *        peer's chunk is still one contiguous K C block, so a caller-run exchange on tfft_dist_plan_buffers is unchanged), the
 *        result is bit-identical to a plan without the flag. Needs N1 = 256, no re-order pass and slabs of whole 128-column
 *        blocks (TFFT_ERR_ARG otherwise). The three separate phase calls run the same slabs one after the other on one stream.
 *        Default (no flag): one slab, everything on the caller's stream, as before. */
enum { TFFT_DIST_SELF_VIA_COMM = 1, TFFT_DIST_CALLER_BUFFERS = 2, TFFT_DIST_SLABS_2 = 4, TFFT_DIST_SLABS_4 = 8 };
int tfft_dist_plan_create(uint64_t n, int world, int rank, int device_id, void* comm, int flags, tfft_dist_plan** out);
void tfft_dist_plan_destroy(tfft_dist_plan* plan);
int tfft_dist_plan_geometry(const tfft_dist_plan* plan, tfft_dist_geometry* out);
/* The exchange buffers (N / world halves each; chunk q at + q * chunk). set_buffers replaces them by caller-owned device
 * memory (16-byte aligned; the four ranges must not overlap: checked as ranges of N / world halves, and nothing is changed
 * when the call fails), e.g. tensors a framework's own collective can send. With one rank and
 * no TFFT_DIST_SELF_VIA_COMM nothing is exchanged: the receive buffers ARE the send buffers and the recv arguments are ignored. */
int tfft_dist_plan_buffers(const tfft_dist_plan* plan, void** send_re, void** send_im, void** recv_re, void** recv_im);
int tfft_dist_plan_set_buffers(tfft_dist_plan* plan, void* send_re, void* send_im, void* recv_re, void* recv_im);
int tfft_dist_exec_pre(const tfft_dist_plan* plan, const void* in_re, const void* in_im, void* stream);
int tfft_dist_exec_exchange(const tfft_dist_plan* plan, void* stream);
int tfft_dist_exec_post(const tfft_dist_plan* plan, void* out_re, void* out_im, void* stream);
int tfft_dist_exec(const tfft_dist_plan* plan, const void* in_re, const void* in_im, void* out_re, void* out_im, void* stream);

/* Host <-> device copies of pageable host memory through a ring of pinned staging buffers (chunked hipMemcpyAsync on a private
 * stream, the host-side copies on a few threads): what DataHandler / DataBatchHandler::CopyDataHostToDevice and
 * CopyResultsDeviceToHost of the C++ shim use in place of the reference's single blocking cudaMemcpy
 * (src/base/DataHandler.h:45-70,116-153). Blocking like it: on return the bytes are where they were sent. The device is the
 * current device. tfft_copy_d2h first waits for the device to finish (as a blocking hipMemcpy does). */
int tfft_copy_h2d(void* dst_device, const void* src_host, size_t bytes);
int tfft_copy_d2h(void* dst_host, const void* src_device, size_t bytes);

/* Layout adapters either side of the path: interleaved (re, im) half2 samples, as cuFFT / hipFFT callers and the
 * reference's comparison code hold them (src/testing/AccuracyCalculator.h:35-48, TestingDataCreation.h half2
 * generators) <-> the planar layout of this library. count = complex samples, a multiple of 8. */
int tfft_deinterleave(const void* in_half2, void* out_re, void* out_im, uint64_t count, void* stream);
int tfft_interleave(const void* in_re, const void* in_im, void* out_half2, uint64_t count, void* stream);

/* Synthetic input born on the device: planes re / im (transform b at + b * batch_stride halves, 0 = 2 n) are filled with
 * uniform(-1, 1) binary16 samples, each a pure function of (seed, first_fft + b, plane, sample index): a counter-based
 * hash, so any sub-batch can be regenerated elsewhere (the CPU-side checker restates the same function) and a benchmark can check
 * sampled transforms of a batch that never existed on the host. The reference creates its signals on the GPU
 * too (src/testing/TestingDataCreation.h:29-147, 152-193), one transform at a time through the host. n % 8 == 0. */
int tfft_synth_uniform(void* re, void* im, uint64_t n, uint64_t batch, uint64_t batch_stride, uint64_t first_fft,
                       uint64_t seed, void* stream);

/* Name of the dominant kernel of this plan (for profiler summaries) and the
 * algorithmic HBM bytes / MFMA flops of one tfft_exec (SURVEY 8d accounting). */
const char* tfft_plan_kernel_name(const tfft_plan* plan);

/* Host only: the column-pass kernel instantiations this library ships, one demangled name per line ("colfft::colfft512_wg_kernel<1, 2,
 * false, true>"), as the library's dispatch table holds them (tfft.hip, TFFT_COL_* lists: a kernel variant is a row of that table,
 * and the launch path can reach no instantiation outside it). Returns the number of rows, or TFFT_ERR_ARG when `bytes` is too
 * small. tests/test_isa_lint.py compares the list with the symbols of the gfx950 code object. No counterpart in the reference. */
int tfft_kernel_list(char* buf, size_t bytes);
double tfft_plan_algorithmic_bytes(const tfft_plan* plan);
double tfft_plan_mfma_flops(const tfft_plan* plan);

/* Message of the last error (or warning) raised on this thread; "" if none. */
const char* tfft_last_error(void);
const char* tfft_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TFFT_H_ */
