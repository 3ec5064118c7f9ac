#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path: batched 1D N=4096 fp16 C2C FFT (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --gpus N --steps K --warmup W          (starts its N ranks itself, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one tfft_exec over one resident batch. At 1 GPU the batch is BASELINE configs[1]: 65536 transforms
(2^28 complex samples, 1 GiB in + 1 GiB out in HBM). With N > 1 GPUs every rank owns 2^21 transforms, the per-GPU
share of BASELINE configs[4a] (batch 2^24 sharded over 8 GPUs: 32 GiB in + 32 GiB out per GPU); the batch shards
with no data-path collective, so scaling is weak and `value` is the aggregate over all ranks.

Input: uniform(-1, 1) binary16 from the library's counter-based hash generator (tfft_synth_uniform, seed 42; transform
index = global index over all ranks), born in HBM. Because every element is a pure function of (seed, transform, plane,
sample), the CPU oracle regenerates any transform of the batch: rank 0 checks sampled transforms of the timed output
against the oracle's fp64 DFT/N, so a broken kernel cannot post a number.

Rank 0 prints ONE JSON line. `roofline` is the HBM roofline of the dominant kernel (algorithmic bytes = 8 B per complex
sample per launch: 4 read + 4 written) with the launch duration measured here by HIP events on the launch
stream. `cpu_baseline` is the CPU oracle (oracle/, a port: the reference has no CPU path, its oracle is cuFFT on the GPU)
timed on this host's cores on a bounded sample of the same input: fp64 radix-2 FFT/N over OpenMP (`value`) and on one
thread, plus the fp64 naive DFT/N at N=256 (BASELINE configs[0]) and at N=4096 on a 64-transform sub-batch.
At 1 GPU `other_configs` carries short measurements of the other BASELINE configs (2^20 x 1024 natural and transposed
order, 2D 4096^2 x 64, single 2^26) and neighbouring lengths, taken AFTER the timed region, each with an oracle check of
a sampled transform / image; they do not enter `value`. With N > 1 GPUs `other_configs` carries BASELINE configs[4b] instead:
ONE transform of N = 2^26 spread over the N GPUs with a single RCCL exchange (tfft_dist_exec), checked by Parseval over all ranks
and four bins per rank against a direct fp64 DFT sum; the entry reports the three phases of the transform separately (column pass,
exchange, row passes: HIP events on the one stream), the communicator's size, the RCCL version and whether the fallback transport
was taken, and under N > 1 the line carries every rank's kernel time.

Failure policy: a failed self-check of the HEADLINE ends the run without a line (a broken kernel cannot post a number). Every
`other_configs` entry, configs[4b] included, is isolated: a failed check or an exception becomes {"error": ...} under the entry's
name, the line is still printed, and the process exits with code 1 (3 when the configs[4b] entry had to be abandoned after
--dist-timeout seconds).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N = 4096
BATCH = 65536                  # BASELINE configs[1], one GPU
BATCH_MULTI = 1 << 21          # BASELINE configs[4a]: 2^24 transforms over 8 GPUs
SEED = 42
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = 2500.0      # dense fp16 MFMA
REL_L2_TOL = 1.5e-3            # the library's stated tolerance against the fp64 DFT/N (tests/test_gpu_parity.py)

RAMP = 100          # untimed launches before the warmup steps (GPU clock ramp, ~35 ms); the cold figure is reported too


def usable_cores(omp_threads):
    """Threads the CPU baseline may really use: OpenMP's count capped by the scheduler affinity and by the cgroup CPU
    quota (a GPU box hands a one-GPU job a share of its host cores; running 128 threads on a 16-core quota only
    measures the throttling)."""
    n = omp_threads
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            tok = open(path).read().split()
            if path.endswith("cpu.max"):
                if tok[0] != "max":
                    n = min(n, max(1, int(int(tok[0]) / int(tok[1]))))
            else:
                q = int(tok[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(seconds_target=8.0):
    """The oracle on this host's cores, on a bounded sample of the benchmark's own input (transforms 0.. of the batch)."""
    import numpy as np
    from oracle import orc

    threads = usable_cores(orc.num_threads())
    probe = 64 * threads
    re, im = orc.synth_uniform(N, probe, 0, SEED)                # the first `probe` transforms of the GPU batch
    out = (np.zeros((probe, N)), np.zeros((probe, N)))           # reused: the loop times transforms, not page faults
    orc.dft64(re, im, threads=threads, out=out)                  # warm
    reps = 0
    t0 = time.perf_counter()
    while True:                                                  # bounded: about seconds_target of CPU work
        orc.dft64(re, im, threads=threads, out=out)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds_target:
            break
    done = reps * probe
    # one thread, same code
    n1 = 256
    out1 = (out[0][:n1], out[1][:n1])
    t0 = time.perf_counter()
    orc.dft64(re[:n1], im[:n1], threads=1, out=out1)
    dt1 = time.perf_counter() - t0
    # naive O(N^2) DFT: N = 4096 on a 64-transform sub-batch (all threads and one thread), N = 256 (configs[0])
    t0 = time.perf_counter()
    orc.dft64(re[:64], im[:64], algo=0, threads=threads)
    dt_naive = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.dft64(re[:4], im[:4], algo=0, threads=1)
    dt_naive1 = time.perf_counter() - t0
    r256, i256 = orc.synth_uniform(256, 4096, 0, SEED)
    t0 = time.perf_counter()
    orc.dft64(r256, i256, algo=0, threads=threads)
    dt_256 = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.dft64(r256[:256], i256[:256], algo=0, threads=1)
    dt_256_1 = time.perf_counter() - t0
    return {
        "value": done * N / dt / 1e9,
        "unit": "Gsamples/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{done} FFTs of N={N} ({done / BATCH:.2f} x the GPU batch of {BATCH}; transforms 0..{probe - 1} of the "
                  f"benchmark input, repeated), oracle fp64 radix-2 FFT/N, OpenMP over the batch, {dt:.1f} s",
        "single_thread": {"value": n1 * N / dt1 / 1e9, "unit": "Gsamples/s", "cores": 1,
                          "sample": f"{n1} FFTs of N={N}, fp64 radix-2 FFT/N, {dt1:.2f} s"},
        "naive_dft_n4096_x64": {"value": 64 * N / dt_naive / 1e9, "unit": "Gsamples/s", "cores": threads,
                                "sample": f"64 transforms of N={N}, fp64 O(N^2) DFT/N, {dt_naive:.2f} s",
                                "single_thread_value": 4 * N / dt_naive1 / 1e9},
        "naive_dft_n256": {"value": 4096 * 256 / dt_256 / 1e9, "unit": "Gsamples/s", "cores": threads,
                           "sample": f"BASELINE configs[0]: 4096 transforms of N=256, fp64 O(N^2) DFT/N, {dt_256:.3f} s",
                           "single_thread_value": 256 * 256 / dt_256_1 / 1e9},
        "note": "the reference has no CPU implementation of this path (its oracle is cuFFT Z2Z on the GPU); this is the "
                "repo's own CPU oracle",
    }


class CheckFailed(Exception):
    """A self-check of a measured output against the CPU oracle (or a size-independent property) failed."""


def _rel_l2(got, exact):
    import numpy as np

    return float(np.linalg.norm(got - exact) / np.linalg.norm(exact))


def check_transforms(torch, orc, y, n, batch, ids, first_fft=0, seed=SEED, perm=None, scale=1.0, in_perm=None):
    """Sampled transforms of an output block ([fft RE | fft IM], stride 2 n) against the oracle's fp64 DFT/N of the
    regenerated input. perm: index map of a transposed-order spectrum; in_perm: the generated block was handed to the plan as
    a transposed-order INPUT, i.e. the signal is block[in_perm]. Returns the worst rel-L2 error; raises beyond the library's
    stated tolerance."""
    import numpy as np

    worst = 0.0
    for b in ids:
        re, im = orc.synth_uniform(n, 1, first_fft + b, seed)
        if in_perm is not None:
            re, im = np.ascontiguousarray(re[:, in_perm]), np.ascontiguousarray(im[:, in_perm])
        e_re, e_im = orc.dft64(re, im)
        exact = (e_re[0] + 1j * e_im[0]) * scale
        o = y[b * 2 * n:(b + 1) * 2 * n].cpu().numpy().astype(np.float64)
        got = o[:n] + 1j * o[n:]
        if perm is not None:
            exact = exact[perm]
        worst = max(worst, _rel_l2(got, exact))
    if not worst < REL_L2_TOL:
        raise CheckFailed(f"self-check failed: N={n} rel-L2 error {worst:.3e} vs the CPU oracle")
    return worst


def other_configs(torch, tf, orc, device):
    """Short measurements of the other BASELINE configs and neighbouring lengths on the same GPU, after the headline
    timing (not part of `value`): Gsamples/s over 20 back-to-back executions each (after 200 ms of untimed launches), inputs resident, workspace preset,
    and an oracle check of a sampled transform (or image) of what was just computed."""
    import numpy as np

    out = {}

    def timed(fn, reps=20):
        # The GPU has idled through the previous entry's CPU-side oracle check: bring it to its steady state first (the headline
        # measurement does the same with its RAMP launches), then time 20 launches. 200 ms, not 50: a rocprofv3 per-dispatch
        # trace of 2^20 x 1024 from idle shows the first (arithmetic-heavy) pass at 1.63, 2.11, 1.91, 1.79, 1.70, 1.63, 1.59 ms
        # before it settles at 1.56-1.57 (profiles/r3_c2_per_dispatch.txt): 50 ms + 10 launches still sat inside that transient
        # and under-reported the steady state by 3-4 % (the reference's protocol: 10 warm-up + 100 timed runs, Bench.h:121-142)
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_warm = 0
        while time.perf_counter() - t0 < 0.2 or n_warm < 3:
            fn()
            n_warm += 1
            if n_warm % 8 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def check_bins(x, y, n, what):
        """Parseval and four spectrum bins of ONE transform against direct fp64 DFT sums on the device (for lengths whose full
        fp64 oracle transform on the host would take tens of seconds)."""
        xs, ys = x[:2 * n].double(), y[:2 * n].double()
        e_in = float((xs * xs).sum()) / n
        e_out = float((ys * ys).sum())
        if not abs(e_out - e_in) / e_in < 5e-3:
            raise CheckFailed(f"self-check failed: {what} Parseval {e_out} vs {e_in}")
        t = torch.arange(n, device="cuda", dtype=torch.float64)
        zr, zi = xs[:n], xs[n:]
        worst = 0.0
        for k in (1, 4097 % n, n // 3, n - 5):
            ph = -2.0 * np.pi * ((t * k) % n) / n
            c, sn = torch.cos(ph), torch.sin(ph)
            er = float((zr * c - zi * sn).sum()) / n
            ei = float((zr * sn + zi * c).sum()) / n
            worst = max(worst, abs(float(y[k]) - er), abs(float(y[n + k]) - ei))
        rms = (e_in / n / 2) ** 0.5
        if not worst < 8 * 2.0 ** -11 * max(rms, 2.0 ** -14):
            raise CheckFailed(f"self-check failed: {what} bins off by {worst:.3e} (spectrum rms {rms:.3e})")
        return f"Parseval + 4 bins against a direct fp64 DFT sum on the device: max |delta| {worst:.2e} (spectrum rms {rms:.2e})"

    def reference_protocol_single():
        """The reference's own benchmark protocol (src/testing/benchmarks/FFTBenchSinlge.cu:11-15, Bench.h:121-142): ONE transform per
        length, N = 2^12, 2^13, ... Reported: DEVICE time per transform, 16 executions captured in one HIP graph and replayed (the
        reference's wall clock around ComputeFFT + synchronise is ~10 us of host time on top, whatever the length;
        examples/bench_single.cpp runs that protocol itself from C++ and prints both). Every length is checked: the full fp64
        oracle transform up to 2^20, Parseval + 4 direct bins beyond."""
        rows = {}
        for lg in range(12, 27):
            n = 1 << lg
            x = torch.empty(2 * n, dtype=torch.float16, device="cuda")
            tf.synth_uniform(x, x[n:], n, 1, seed=SEED + lg)
            y = torch.empty_like(x)
            plan = tf.TfftPlan(n, 1, device, preserve_input=True)
            ws = torch.empty(max(1, plan.workspace_bytes // 2), dtype=torch.float16, device="cuda")
            if plan.workspace_bytes:
                plan.set_workspace(ws)
            plan.exec(x, x[n:], y, y[n:])
            torch.cuda.synchronize()
            st = torch.cuda.Stream()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.stream(st):
                with torch.cuda.graph(gr, stream=st):
                    for _ in range(16):
                        plan.exec(x, x[n:], y, y[n:], stream=st.cuda_stream)
            torch.cuda.synchronize()
            reps = 16 if lg <= 22 else 4
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.03:          # clock ramp
                gr.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    gr.replay()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / reps / 16 * 1e3)
            ts.sort()
            # ... and the reference's protocol as it stands (Bench.h:121-142): wall clock around one eager call + device synchronise,
            # 10 warm-up + 100 timed samples (from Python: ctypes + HIP launch + synchronise, ~15-20 us whatever the transform)
            wall = []
            for k in range(110):
                torch.cuda.synchronize()
                w0 = time.perf_counter()
                plan.exec(x, x[n:], y, y[n:])
                torch.cuda.synchronize()
                if k >= 10:
                    wall.append((time.perf_counter() - w0) * 1e6)
            wall_mean = sum(wall) / len(wall)
            if lg <= 20:
                err = check_transforms(torch, orc, y, n, 1, [0], seed=SEED + lg)
                check = f"vs the fp64 oracle: rel-L2 {err:.2e}"
            else:
                check = check_bins(x, y, n, f"single N=2^{lg}")
            rows[f"2^{lg}"] = {"device_us": ts[len(ts) // 2], "device_us_min": ts[0], "wall_us_eager_call_plus_sync": wall_mean, "passes": plan.num_launches,
                               "plan": tf.plan_describe(n, 1, tf.plan_default_variant(n, 1, 1)), "check": check}
            plan.close()
            del gr
        return {"protocol": "one transform per length (FFTBenchSinlge.cu:11-15); device time per transform, 16 executions per HIP graph, "
                            "median of 7 rounds; beside it the reference's own sample (wall clock of one eager call + synchronise, mean of 100, "
                            "mostly host time); input uniform(-1,1) from the counter-hash generator",
                "lengths": rows}

    cases = (("n256_x_1048576", 256, 1 << 20, "natural"), ("n1024_x_262144", 1024, 1 << 18, "natural"),
             ("n8192_x_32768", 8192, 1 << 15, "natural"), ("n65536_x_4096", 1 << 16, 1 << 12, "natural"),
             ("configs[2]_n2^20_x_1024", 1 << 20, 1024, "natural"),
             ("configs[2]_n2^20_x_1024_transposed_order", 1 << 20, 1024, "transposed"),
             ("configs[2]_n2^20_x_1024_transposed_input", 1 << 20, 1024, "transposed_in"),
             ("n2^24_x_16", 1 << 24, 16, "natural"), ("n2^24_x_16_transposed_input", 1 << 24, 16, "transposed_in"), ("configs[4b]_single_gpu_n2^26_x_1", 1 << 26, 1, "natural"),
             # work that does not fill the chip (the reference's single-transform benchmark, FFTBenchSinlge.cu): the planner's
             # small-work split and the footprint cache policy (tfft_plan_default_variant, tfft_plan_cache_policy)
             ("single_n2^20", 1 << 20, 1, "natural"), ("n2^20_x_16", 1 << 20, 16, "natural"))
    def guarded(name, fn):
        """One failure policy for every entry: a failed check or an exception becomes {"error": ...} under the entry's name, the
        other entries and the headline line are unaffected, and main() exits non-zero after printing the line."""
        try:
            out[name] = fn()
        except Exception as e:      # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"}
            print(f"bench.py: other_configs[{name}]: {out[name]['error']}", file=sys.stderr, flush=True)
        torch.cuda.empty_cache()

    def entry_1d(n, b, order):
        x = torch.empty(b * 2 * n, dtype=torch.float16, device="cuda")
        tf.synth_uniform(x, x[n:], n, b, seed=SEED + n)
        y = torch.empty_like(x)
        plan = tf.TfftPlan(n, b, device, preserve_input=True, output_order="transposed" if order == "transposed" else "natural",
                           input_order="transposed" if order == "transposed_in" else "natural")
        ws = torch.empty(max(1, plan.workspace_bytes // 2), dtype=torch.float16, device="cuda")
        if plan.workspace_bytes:
            plan.set_workspace(ws)
        ms = timed(lambda: plan.exec(x, x[n:], y, y[n:]))
        perm = in_perm = None
        if order == "transposed":
            n2 = tf.transposed_n2(n)
            perm = np.arange(n).reshape(n2, n // n2).T.reshape(-1)         # out[k1 n2 + k2] = X[k1 + n1 k2]
        if order == "transposed_in":
            n2 = tf.transposed_n2(n)
            in_perm = np.arange(n).reshape(n // n2, n2).T.reshape(-1)      # x[k1 + n1 k2] = in[k1 n2 + k2]
        if n <= (1 << 24):
            err = check_transforms(torch, orc, y, n, b, sorted({0, b - 1}), seed=SEED + n, perm=perm, in_perm=in_perm)
            check = f"transforms 0 and {b - 1} vs the fp64 oracle: rel-L2 {err:.2e}"
        else:
            # 2^26: the full fp64 oracle transform takes ~30 s of host time; check Parseval and 4 bins computed directly
            check = check_bins(x, y, n, "N=2^26")
        rep = {"gsamples_per_s": n * b / ms / 1e6, "ms": ms, "passes": plan.num_launches, "check": check}
        if order != "natural":
            rep.update(chunked_roofline(n * b / ms / 1e6, plan.workspace_bytes // (4 * n), "transforms"))
        return rep

    def chunked_roofline(gsamples, chunk, unit):
        """A chunked two-pass plan runs both passes of a chunk back to back through one chunk-sized workspace, so that the second pass
        finds the intermediate in the 256-MiB Infinity Cache. Two ways to price it against the 8 TB/s HBM roofline:
        16 B per sample (both passes' reads and writes counted as memory traffic: SURVEY 8d's two-pass figure) and 8 B per sample
        (the intermediate never leaves the die: only the caller's input and output are HBM traffic). The truth lies between; the
        A/B against whole-batch passes on one box is under profiles/."""
        return {"chunk": chunk, "chunk_unit": unit, "roofline_16B": {"achieved_GBps": gsamples * 16, "frac": gsamples * 16 / HBM_PEAK_GBS},
                "roofline_8B": {"achieved_GBps": gsamples * 8, "frac": gsamples * 8 / HBM_PEAK_GBS}}

    for name, n, b, order in cases:
        guarded(name, lambda: entry_1d(n, b, order))     # noqa: B023 (called at once)

    def entry_2d():
        rows = cols = 4096
        images = 64
        half = images * rows * cols                      # fully planar: all RE images, then all IM images
        x = torch.empty(2 * half, dtype=torch.float16, device="cuda")
        tf.synth_uniform(x[:half], x[half:], rows * cols, images, batch_stride=rows * cols, seed=SEED + 2)
        y = torch.empty_like(x)
        plan2 = tf.TfftPlan2D(rows, cols, images, device)
        ms = timed(lambda: plan2.exec(x[:half], x[half:], y[:half], y[half:]), reps=10)
        b = images - 1
        re, im = orc.synth_uniform(rows * cols, 1, b, SEED + 2)
        a_re, a_im = orc.dft64(re.reshape(rows, cols), im.reshape(rows, cols))
        exact = orc.fft64_rows(np.ascontiguousarray((a_re + 1j * a_im).T)).T
        got = (y[b * rows * cols:(b + 1) * rows * cols].cpu().numpy().astype(np.float64)
               + 1j * y[half + b * rows * cols:half + (b + 1) * rows * cols].cpu().numpy().astype(np.float64)).reshape(rows, cols)
        err = _rel_l2(got, exact)
        if not err < REL_L2_TOL:
            raise CheckFailed(f"self-check failed: 2D 4096x4096 rel-L2 error {err:.3e} vs the CPU oracle")
        rep = {"gsamples_per_s": half / ms / 1e6, "ms": ms, "passes": plan2.num_launches,
               "check": f"image {b} vs the fp64 oracle (rows, then columns): rel-L2 {err:.2e}"}
        rep.update(chunked_roofline(half / ms / 1e6, plan2.workspace_bytes // (4 * rows * cols), "images"))
        return rep

    guarded("configs[3]_2d_4096x4096_x_64", entry_2d)
    guarded("reference_protocol_single", reference_protocol_single)
    return out


def shard(rank, world, total):
    """Contiguous slice [lo, hi) of `total` independent transforms owned by `rank`: the whole multi-GPU story of
    the batched path (FFTs are independent, no data-path collective). bench.py itself runs weak
    scaling (every rank the same per-GPU batch); this helper gives the strong-scaling split and is what places a
    rank's transforms in the global index space of the input generator."""
    per = total // world
    extra = total % world
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


def measured_traffic(kernel_name):
    """HBM bytes per launch from the newest committed PMC summary (tools/summarize_pmc.py; FETCH_SIZE x2 +
    WRITE_SIZE from separate rocprofv3 --pmc passes of this same command), or None."""
    import glob

    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("kernel") == kernel_name and "hbm_bytes_per_dispatch" in d:
            best = (d["hbm_bytes_per_dispatch"]["total"], os.path.relpath(path, ROOT))
    return best


def free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n_ranks, script, script_args, nproc_visible=None):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, as FRESH child processes under
    torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1), before this process has made any GPU call (a
    process that has initialised the GPU must never be replaced or forked into ranks on this pool). The children inherit
    stdout, so rank 0's JSON line is this command's output; the return code is non-zero if any rank failed.
    nproc_visible: GPUs this box has (None: do not check; counting devices does not initialise the GPU)."""
    import subprocess

    if nproc_visible is not None and nproc_visible < n_ranks:
        print(f"bench.py: --gpus {n_ranks} needs {n_ranks} GPUs, this box shows {nproc_visible}", file=sys.stderr, flush=True)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(script_args)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on these hosts (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        print(f"bench.py: the {n_ranks}-rank run failed (torch.distributed.run exit code {rc})", file=sys.stderr, flush=True)
    return rc


_DIST_PROGRESS = {}       # the distributed entry's finished part, for the watchdog in main()


def dist_2pow26(torch, tf, dist, rank, world, local_rank, reps=20, self_via_comm=False):
    """BASELINE configs[4b]: ONE transform of N = 2^26 spread over the `world` GPUs, four-step with a single RCCL exchange
    (tfft_dist_exec: column pass -> ncclSend / ncclRecv group -> row transforms, all on one stream). Every rank calls this;
    returns the report (identical on all ranks). Reported: whole-transform time, the three phases separately (HIP events on the one
    stream around tfft_dist_exec_pre / _exchange / _post, min and max over ranks), the facts of the communicator, and the checks:
    Parseval over all ranks and four spectrum bins per rank against a direct fp64 DFT sum of the (regenerated) input."""
    import numpy as np
    from tensor_fft_amd import capi
    from tensor_fft_amd.distributed import DistributedFFT1D, DistSetupError, HipEngine

    n = 1 << 26
    # The exchange runs inside the C ABI over its own RCCL communicator. DistributedFFT1D makes the ranks AGREE on whether creating
    # it worked (it is the one piece no single-GPU box can rehearse with more than one rank): on failure every rank gets
    # DistSetupError together, and all fall back, in lock-step, to the same plan with the exchange over the torch process group.
    want_native = world > 1 or self_via_comm
    why = None
    try:
        f = DistributedFFT1D(n, engine=HipEngine(local_rank), transport="rccl" if want_native else None, self_via_comm=self_via_comm)
    except DistSetupError as e:
        if world == 1:
            raise
        why = str(e)
        f = DistributedFFT1D(n, engine=HipEngine(local_rank), transport="torch")
    try:
        return _dist_2pow26_body(torch, tf, dist, rank, world, f, n, why, want_native, self_via_comm, reps, capi, local_rank)
    finally:
        f.close()          # the plan, its 512 MiB of buffers and the communicator go whatever happened above


def _agree(torch, dist, world, ok_local):
    """All ranks learn whether ANY of them failed (one MIN all-reduce): a local exception then ends the entry on every rank
    together instead of leaving the others inside the next collective until the watchdog fires."""
    if world == 1:
        return bool(ok_local)
    t = torch.tensor([1 if ok_local else 0], dtype=torch.int32, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t[0]))


def _dist_2pow26_body(torch, tf, dist, rank, world, f, n, why, want_native, self_via_comm, reps, capi, local_rank):
    import numpy as np
    from tensor_fft_amd.distributed import DistributedFFT1D, HipEngine

    g = f.geometry
    n1, n2, c, k = int(g.n1), int(g.n2), int(g.cols), int(g.rows)
    comm_info = None
    if f._comm is not None:
        cnt, me = f._comm.info()
        comm_info = {"ncclCommCount": cnt, "ncclCommUserRank_of_rank0": me}
    try:
        rccl_version = capi.dist_rccl_version() if want_native else None
    except capi.TfftError as e:
        rccl_version = f"unavailable: {e}"
    # the whole signal on every rank (2 x 128 MiB; a pure function of the seed), this rank's columns sliced out of it
    x = torch.empty(2 * n, dtype=torch.float16, device="cuda")
    tf.synth_uniform(x, x[n:], n, 1, seed=SEED + 26)
    in_re = x[:n].view(n1, n2)[:, rank * c:(rank + 1) * c].contiguous().view(-1)
    in_im = x[n:].view(n1, n2)[:, rank * c:(rank + 1) * c].contiguous().view(-1)
    token = torch.zeros(1, device="cuda")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.all_reduce(token)
            torch.cuda.synchronize()

    local_err = None
    ms, ph, re, im = 0.0, {"pre_ms": 0.0, "exchange_ms": 0.0, "post_ms": 0.0}, None, None
    try:
        for _ in range(3):
            re, im = f.forward(in_re, in_im)
        fence()
        t0 = time.perf_counter()
        for _ in range(reps):
            re, im = f.forward(in_re, in_im)
        fence()
        ms = (time.perf_counter() - t0) / reps * 1e3
        # the three phases, each between two HIP events on the stream all of them are enqueued on
        fence()
        ph = f.phase_times(in_re, in_im, reps)
        fence()
        re, im = f.forward(in_re, in_im)
        torch.cuda.synchronize()
    except Exception as e:      # noqa: BLE001
        local_err = f"{type(e).__name__}: {e}"
    if not _agree(torch, dist, world, local_err is None):
        return {"error": local_err or "another rank failed inside the timed section of the distributed transform (its stderr has the reason)"}
    keys = ("pre_ms", "exchange_ms", "post_ms")
    t_max = torch.tensor([ms] + [ph[q] for q in keys], dtype=torch.float64, device="cuda")
    t_min = t_max.clone()
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_min, op=dist.ReduceOp.MIN)
    ms = float(t_max[0])
    phases = {q: {"max_over_ranks": float(t_max[1 + i]), "min_over_ranks": float(t_min[1 + i])} for i, q in enumerate(keys)}
    local_ms = phases["pre_ms"]["max_over_ranks"] + phases["post_ms"]["max_over_ranks"]
    sent = 2 * (world - 1) * int(g.chunk) * 2                       # bytes this rank sends (= receives): both planes
    if self_via_comm:
        sent += 2 * int(g.chunk) * 2                                # (rehearsal: the own chunk travels through RCCL too)
    report = {
        "ms": ms, "gsamples_per_s": n / ms / 1e6, "phases": phases, "local_ms_without_exchange": local_ms,
        "n1": n1, "n2": n2, "columns_per_rank": c, "rows_per_rank": k, "local_passes": int(g.local_passes),
        "reorder_pass": bool(g.reorder), "transport": f.transport, "transport_fallback_taken": why is not None,
        "transport_fallback_reason": why, "communicator": comm_info, "rccl_version": rccl_version,
        "bytes_through_the_collective_per_rank": sent,
        "exchange_GBps_per_rank": (sent / (phases["exchange_ms"]["max_over_ranks"] * 1e-3) / 1e9) if sent else None,
    }
    # ---- checks: Parseval over all ranks, and on EVERY rank four bins of its own slice of the spectrum against direct fp64 sums
    xs = x.double()
    e_in = float((xs * xs).sum()) / n
    e_out = (re.double() ** 2).sum() + (im.double() ** 2).sum()
    if world > 1:
        dist.all_reduce(e_out)
    e_out = float(e_out)
    if not abs(e_out - e_in) / e_in < 5e-3:
        report["error"] = f"self-check failed: distributed N=2^26 Parseval {e_out} vs {e_in}"
        return report
    tt = torch.arange(n, device="cuda", dtype=torch.float64)
    worst = 0.0
    for kk, k2 in ((0, 1), (k // 2, n2 // 3), (k - 1, n2 - 5), (1 % k, 4097)):
        k1 = rank * k + kk
        bin_ = k1 + n1 * k2                                         # X[k1 + N1 k2] lives at [kk][k2] of this rank
        phz = -2.0 * np.pi * ((tt * bin_) % n) / n
        cs, sn = torch.cos(phz), torch.sin(phz)
        er = float((xs[:n] * cs - xs[n:] * sn).sum()) / n
        ei = float((xs[:n] * sn + xs[n:] * cs).sum()) / n
        worst = max(worst, abs(float(re[kk * n2 + k2]) - er), abs(float(im[kk * n2 + k2]) - ei))
    w = torch.tensor([worst], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
    worst = float(w[0])
    rms = (e_in / n / 2) ** 0.5
    if not worst < 8 * 2.0 ** -11 * max(rms, 2.0 ** -14):
        report["error"] = f"self-check failed: distributed N=2^26 bins off by {worst:.3e} (spectrum rms {rms:.3e})"
        return report
    report["check"] = (f"Parseval over all ranks + 4 bins of every rank's own slice against a direct fp64 DFT sum on the device: max |delta| "
                       f"over the ranks {worst:.2e} (spectrum rms {rms:.2e})")
    # ---- the same transform with the exchange overlapped slab by slab (TFFT_DIST_SLABS_*): whole-transform time for S = 2 and 4 beside
    # the S = 1 figure above (whose three phases are separable because they run one after the other); every S must give S = 1's bits
    if f.transport == "rccl" and want_native:
        over = {}
        _DIST_PROGRESS["report"] = dict(report)      # (what the watchdog prints if a rank gets stuck in the part below)
        for slabs in (2, 4):
            _DIST_PROGRESS["stage"] = f"overlapped exchange, {slabs} slabs"
            err_s, ms_s, same = None, 0.0, False
            f2 = None
            try:
                f2 = DistributedFFT1D(n, engine=HipEngine(local_rank), transport="rccl", self_via_comm=self_via_comm, slabs=slabs)
                for _ in range(3):
                    r2, i2 = f2.forward(in_re, in_im)
                fence()
                t0 = time.perf_counter()
                for _ in range(reps):
                    r2, i2 = f2.forward(in_re, in_im)
                fence()
                ms_s = (time.perf_counter() - t0) / reps * 1e3
                same = bool((r2.view(torch.int16) == re.view(torch.int16)).all()) and bool((i2.view(torch.int16) == im.view(torch.int16)).all())
            except Exception as e:      # noqa: BLE001  (a refused geometry is refused on every rank alike; a setup failure is agreed inside the constructor)
                err_s = f"{type(e).__name__}: {e}"
            finally:
                if f2 is not None:
                    f2.close()
            if not _agree(torch, dist, world, err_s is None):
                over[f"slabs_{slabs}"] = {"error": err_s or "another rank failed"}
                break
            t = torch.tensor([ms_s, 1.0 if same else 0.0], dtype=torch.float64, device="cuda")
            tmin = t.clone()
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
            over[f"slabs_{slabs}"] = {"ms": float(t[0]), "gsamples_per_s": n / float(t[0]) / 1e6, "bit_identical_to_one_slab_on_every_rank": bool(float(tmin[1]))}
            if not bool(float(tmin[1])):
                report["error"] = f"self-check failed: {slabs} column slabs do not reproduce the bits of one slab"
        report["overlapped_exchange"] = over
    return report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=0,
                    help="transforms per GPU (default: 65536 = BASELINE configs[1] at 1 GPU, 2^21 = the per-GPU share of "
                         "configs[4a] at N > 1 GPUs)")
    ap.add_argument("--with-dist", action="store_true",
                    help="also run the BASELINE configs[4b] entry (N = 2^26 through tfft_dist_*) when there is only one rank")
    ap.add_argument("--dist-self-via-comm", action="store_true",
                    help="rehearsal aid for a one-GPU box (with --with-dist): the configs[4b] entry creates its RCCL communicator and "
                         "routes the own chunk through ncclSend / ncclRecv, next to the torch process group's own communicator")
    ap.add_argument("--only-dist-entry", action="store_true",
                    help="of the other_configs entries run only configs[4b] (with --with-dist on one rank: the rehearsal the GPU tests use)")
    ap.add_argument("--dist-timeout", type=float, default=300.0,
                    help="seconds the configs[4b] entry may take before the line is printed without it and the run exits with code 3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short extra measurements of the other BASELINE configs (reported under 'other_configs')")
    args = ap.parse_args()

    import torch                     # (importing torch and counting devices do not initialise the GPU)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become the launcher. Nothing has touched the GPU yet; build first so that the ranks
        # only load the prebuilt library.
        import __graft_entry__ as g

        g.build()
        sys.exit(self_launch(args.gpus, os.path.abspath(__file__), sys.argv[1:], nproc_visible=torch.cuda.device_count()))

    import __graft_entry__ as g

    # recompiled here, or the prebuilt library that travelled with the tree (build() is mtime-based)?
    build_mode = "reused: prebuilt libtfft.so newer than its sources" if g.is_current() else "recompiled by this run"
    g.build()
    import tensor_fft_amd as tf

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE={world})")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU (this box shows {torch.cuda.device_count()})")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:      # under torch.distributed.run even a single rank takes the RCCL path
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL: barrier + max only

    tf.device_check(local_rank)
    batch = args.batch or (BATCH if world == 1 else BATCH_MULTI)
    first_fft, _ = shard(rank, world, batch * world)          # this rank's slice of the global transform index space
    # synthetic planar fp16, uniform(-1,1), DataBatchHandler layout [fft_i RE | fft_i IM], born in HBM
    x = torch.empty(batch * 2 * N, dtype=torch.float16, device="cuda")
    tf.synth_uniform(x, x[N:], N, batch, first_fft=first_fft, seed=SEED)
    y = torch.empty_like(x)
    plan = tf.TfftPlan(N, batch, local_rank, preserve_input=True)
    torch.cuda.synchronize()

    def step():
        plan.exec(x, x[N:], y, y[N:])

    token = torch.zeros(1, device="cuda") if dist is not None else None

    def fence():
        """barrier + synchronize. The barrier is a one-element all-reduce on a preallocated tensor (what dist.barrier()
        does, minus its extra device synchronisations, which cost ~1 ms per call and would be charged to the K steps)."""
        torch.cuda.synchronize()
        if dist is not None:
            dist.all_reduce(token)
            torch.cuda.synchronize()

    # The GPU leaves its idle clock state only after a few milliseconds of work. The first K launches are timed as they
    # are (`cold_ms_per_step`, reported, not `value`), then RAMP untimed launches, the contract's W warmup steps and
    # the K timed steps.
    fence()
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0.record()
    for _ in range(args.steps):
        step()
    c1.record()
    fence()
    cold_ms = c0.elapsed_time(c1) / args.steps
    for _ in range(RAMP):
        step()
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    fence()
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps            # launches are back to back on one stream
    kernel_ms_per_rank = [kernel_ms]
    if dist is not None:
        # every rank's own kernel time, so that one slow GPU is visible in the weak-scaling line (the line's kernel_ms is the max)
        mine = torch.tensor([kernel_ms], dtype=torch.float64, device="cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        kernel_ms_per_rank = [float(v[0]) for v in every]
        t = torch.tensor([wall, kernel_ms, cold_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, kernel_ms, cold_ms = float(t[0]), float(t[1]), float(t[2])

    # per-launch spread (SURVEY 8d asks for mean and sigma; the true mean, not the reference's sum / (n - 1),
    # BenchUtil.h:41-48): 20 individually timed launches after the timed region, not part of `value`
    singles = []
    for _ in range(20):
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        step()
        a1.record()
        torch.cuda.synchronize()
        singles.append(a0.elapsed_time(a1))
    single_mean = sum(singles) / len(singles)
    single_sigma = (sum((v - single_mean) ** 2 for v in singles) / (len(singles) - 1)) ** 0.5

    # self-check so a broken kernel cannot post a number: Parseval on every rank's first 64 transforms; on rank 0 sampled
    # transforms of the timed output against the CPU oracle's fp64 DFT/N of the regenerated input
    xs = x[: 64 * 2 * N].float().reshape(64, 2 * N)
    ys = y[: 64 * 2 * N].float().reshape(64, 2 * N)
    par = float((((ys ** 2).sum(1) - (xs ** 2).sum(1) / N).abs() / ((xs ** 2).sum(1) / N)).max())
    if not par < 5e-3:
        raise SystemExit(f"self-check failed: Parseval mismatch {par:.3e}")

    if dist is not None and world > 1:
        # every rank checks one sampled transform of ITS OWN slice against the CPU oracle (a wrong slice on rank 5 must not pass
        # because rank 0's is right); the worst error travels to rank 0, a failure on any rank ends the run on all of them
        from oracle import orc as orc_all

        try:
            mine_err = check_transforms(torch, orc_all, y, N, batch, [(7919 * (rank + 1)) % batch], first_fft=first_fft)
        except CheckFailed:
            mine_err = float("inf")
        worst_rank = torch.tensor([min(mine_err, 1e30)], dtype=torch.float64, device="cuda")
        dist.all_reduce(worst_rank, op=dist.ReduceOp.MAX)
        per_rank_err = float(worst_rank[0])
        if not per_rank_err < REL_L2_TOL:
            raise SystemExit(f"self-check failed: a rank's own slice is off by rel-L2 {per_rank_err:.3e} vs the CPU oracle")
    else:
        per_rank_err = None
    if rank == 0:
        from oracle import orc

        ids = sorted({0, 1, batch // 2 + 3, batch - 1})
        try:
            oracle_err = check_transforms(torch, orc, y, N, batch, ids, first_fft=first_fft)
        except CheckFailed as e:            # the headline itself: no line at all (a broken kernel cannot post a number)
            raise SystemExit(str(e))
        samples_per_step = float(N) * batch * world
        value = samples_per_step * args.steps / wall / 1e9
        alg_bytes = plan.algorithmic_bytes                      # per launch, this rank
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        mfma_tflops = plan.mfma_flops / (kernel_ms * 1e-3) / 1e12
        traffic = measured_traffic(plan.kernel_name) if batch == BATCH else None
        workload = ("BASELINE configs[1]: batched 1D N=4096 fp16 C2C FFT, batch=65536 on one GPU" if (world == 1 and batch == BATCH)
                    else f"BASELINE configs[4a] share: batched 1D N=4096 fp16 C2C FFT, batch={batch} per GPU "
                         f"({batch * world} transforms over {world} GPU(s); configs[4a] = 2^24 over 8)")
        line = {
            "metric": "Gsamples/s + %fp16-MFMA-peak, batched N=4096 fp16 C2C FFT",
            "value": value,
            "unit": "Gsamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {
                "workload": workload + ", planar [RE|IM] blocks resident in HBM, result = DFT(x)/N",
                "n": N,
                "batch_per_gpu": batch,
                "input": f"uniform(-1,1) binary16 from the counter-hash generator (tfft_synth_uniform, seed {SEED}, global "
                         "transform index), reproducible on the CPU",
                "clock_ramp_launches": RAMP,
                "cold_ms_per_step": cold_ms,
                "parallelism": f"batch sharded over {world} GPU(s), no data-path collective",
                "self_check": f"transforms {ids} of rank 0's timed output vs the CPU oracle's fp64 DFT/N: rel-L2 {oracle_err:.2e}; "
                              f"Parseval on 64 transforms per rank: {par:.1e}"
                              + (f"; one sampled transform of EVERY rank's own slice vs the oracle: worst rel-L2 {per_rank_err:.2e}" if per_rank_err is not None else ""),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": plan.kernel_name,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic[0] if traffic else None,
                "traffic_source": traffic[1] if traffic else None,
                "kernel_ms": kernel_ms,
                "kernel_ms_per_rank": {"min": min(kernel_ms_per_rank), "max": max(kernel_ms_per_rank), "all": kernel_ms_per_rank},
                "kernel_ms_single_launches": {"mean": single_mean, "sigma": single_sigma, "n": len(singles)},
                "algorithmic_bytes_per_launch": alg_bytes,
            },
            "mfma": {"tflops": mfma_tflops, "peak": MFMA_PEAK_TFLOPS, "frac": mfma_tflops / MFMA_PEAK_TFLOPS,
                     "flop_per_sample": 384},
        }
        line["runtime"] = {
            "world_size": dist.get_world_size() if dist is not None else 1,
            "backend": dist.get_backend() if dist is not None else None,
            "visible_gpus": torch.cuda.device_count(),
            "device": torch.cuda.get_device_name(local_rank),
            "launcher": "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "none",
            "build_mode": build_mode,
            "library": tf.lib_path() if hasattr(tf, "lib_path") else None,
        }
        if world == 1 and not args.no_other_configs and not args.only_dist_entry and batch == BATCH:
            x = y = None
            torch.cuda.empty_cache()
            line["other_configs"] = other_configs(torch, tf, orc, local_rank)
    else:
        line = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline()          # (before the distributed entry: everything the line needs apart from that entry is then in hand)
    if (world > 1 or args.with_dist) and not args.no_other_configs:
        # BASELINE configs[4b]: every rank takes part (one exchange per transform); rank 0 reports. Same failure policy as every
        # other entry ({"error": ...}, line printed, exit code non-zero), plus a watchdog: this is the one entry that contains a
        # collective no one-GPU box can rehearse with more than one rank, and a rank stuck inside it cannot be interrupted from Python.
        # If the entry has not returned after --dist-timeout seconds, rank 0 prints the line it has (the headline is complete and
        # self-checked by now) with the entry marked as timed out, and every rank leaves with exit code 3.
        x = y = None
        torch.cuda.empty_cache()
        import threading

        def give_up():
            if rank == 0:
                why = f"timed out after {args.dist_timeout} s (a rank is stuck inside the entry; the headline above is unaffected)"
                if "report" in _DIST_PROGRESS:      # the S = 1 transform and its phases were measured and checked: keep them
                    rep = dict(_DIST_PROGRESS["report"])
                    rep["overlapped_exchange"] = {"error": why + f"; stage: {_DIST_PROGRESS.get('stage')}"}
                    rep["error"] = "overlapped exchange: " + why
                else:
                    rep = {"error": why}
                line.setdefault("other_configs", {})["configs[4b]_n2^26_distributed"] = rep
                print(json.dumps(line), flush=True)
            print(f"bench.py: rank {rank}: configs[4b] entry timed out after {args.dist_timeout} s", file=sys.stderr, flush=True)
            os._exit(3)

        dog = threading.Timer(args.dist_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            rep = dist_2pow26(torch, tf, dist, rank, world, local_rank, self_via_comm=args.dist_self_via_comm)
        except Exception as e:      # noqa: BLE001  (the headline line must not depend on this entry)
            rep = {"error": f"{type(e).__name__}: {e}"}
        dog.cancel()
        if rank == 0:
            if "error" in rep:
                print("bench.py: configs[4b] entry: " + rep["error"], file=sys.stderr, flush=True)
            line.setdefault("other_configs", {})["configs[4b]_n2^26_distributed"] = rep
    failed = []
    if rank == 0:
        failed = [k for k, v in line.get("other_configs", {}).items() if isinstance(v, dict) and "error" in v]
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        # one policy for all side entries: the (self-checked) headline line has been printed, the failures are in it, the exit code says so
        print("bench.py: failed other_configs entries: " + ", ".join(failed), file=sys.stderr, flush=True)
        sys.exit(1)


if __name__ == "__main__":
    main()
