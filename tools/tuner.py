"""Autotuner: the MI355X counterpart of the reference's TunerSingleFFT.cu / BenchUtil.h grid search
(src/testing/benchmarks/TunerSingleFFT.cu:10-56, BenchUtil.h:77-150, output FileWriter.h:250-269).

The reference searches (mode, base_fft_warps_per_block, r16_warps_per_block, r2_blocksize) and writes
`N mode a b c` lines that CreatePlan(N, file) reads back (Plan.h:197-255). On MI355X those launch-geometry
knobs do not exist; what can be tuned is the kernel variant of the N = 4096 path and, for other lengths,
the pass decomposition (radix-256 column passes vs the plain radix-16 autosort chain), and for every kernel its launch
shape (tfft_plan_opts.launch_iters: short-lived against persistent workgroups), which depends on the batch. The file keeps
the reference's five columns (still valid input for its own parser, which ignores further tokens) and appends
`variant launch_iters batch` as columns six to eight: one line per (N, batch) with --batches (the MI355X counterpart of
TunerBatchFFTs.cu:10-55). This repo's CreatePlan(N, file) reads all lines of a length and ComputeFFT uses the one whose
batch is nearest to the batch it runs.

    python tools/tuner.py [--out TunerResults.dat] [--min-log2 8] [--max-log2 24] [--samples 20] [--warmup 5]
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def candidates(n):
    """Kernel variants (tfft_plan_opts.variant) tried for length n. Every value here is a documented, CORRECT-result
    variant (tests/test_gpu_round2.py runs each against the oracle); the timing-only debugging bits are not tuner input.
    0 = library default (2|8 at N = 4096); 32 = plain autosort chain; 524288 = 4-wave cooperative column workgroups;
    2097152 = unfused radix-16 + radix-2/4 tail; 1048576 = unstaged column stores; 16777216 = column plan instead of
    the single-pass kernel (2^13..2^15); 8388608 = no radix-512 column passes; 33554432 = no radix-1024 column passes; 134217728 = among the
    splits with the fewest passes, the one with the most wide (radix-1024, then radix-512) passes; 268435456 = a final radix-512
    pass by the OTHER of its two kernels (include/tfft.h: the two-round kernel of colfft512r.hpp with 8-wave workgroups and
    128-column tiles where the single-round kernel is the default, and vice versa; 4-wave workgroups only together with 524288);
    262144 / 536870912 = plain / non-temporal global accesses in the column passes whatever the footprint (default: by footprint,
    tfft_plan_cache_policy)."""
    if n == 4096:
        return [16, 2, 10, 8, 1]
    if n < 8192:
        return [0, 32]
    if n <= 32768:
        return [0, 16777216, 16777216 | 8388608, 16777216 | 8388608 | 33554432, 32]     # (the last but one: 2^15 as 256 x 128, cooperative radix-128 pass)
    return [0, 32, 524288, 2097152, 1048576, 8388608, 33554432, 8388608 | 33554432, 134217728, 268435456, 262144, 536870912]


def iters_candidates():
    """tfft_plan_opts.launch_iters values tried per (N, batch) on the winning variant: 0 = the library's default shape, k = a
    workgroup takes about k rounds and retires, 65535 = persistent workgroups. Launch shapes never change results
    (tests/test_gpu_round3.py runs every value here against the default bit for bit)."""
    return [0, 1, 2, 4, 65535]


def batches_for(n, total_log2):
    """Batch sizes a length is tuned at (the reference tunes one async batch size, TunerBatchFFTs.cu:10-55; the best launch
    shape on MI355X depends on how many workgroups the batch fills the 256 CUs with): 1, 64, 4096 and 2^total / N."""
    big = max(1, (1 << total_log2) // n)
    return sorted({b for b in (1, 64, 4096, big) if b <= big})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="TunerResults.dat")
    ap.add_argument("--min-log2", type=int, default=8)
    ap.add_argument("--max-log2", type=int, default=24)
    ap.add_argument("--samples", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--total-log2", type=int, default=26, help="largest batch = 2^total / N transforms per timing")
    ap.add_argument("--batches", action="store_true", help="tune every length at several batch sizes (1, 64, 4096, 2^total / N): "
                                                           "one file line per (N, batch)")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as g

    g.build()
    import tensor_fft_amd as tf

    def time_plan(n, batch, x, y, v, iters):
        try:
            plan = tf.TfftPlan(n, batch, 0, variant=v, preserve_input=True, launch_iters=iters)
        except tf.TfftError:
            return None
        ws = torch.empty(max(1, plan.workspace_bytes // 2), dtype=torch.float16, device="cuda")
        if plan.workspace_bytes:
            plan.set_workspace(ws)
        ts = []
        for k in range(args.warmup + args.samples):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            plan.exec(x, x[n:], y, y[n:])
            e1.record()
            torch.cuda.synchronize()
            if k >= args.warmup:
                ts.append(e0.elapsed_time(e1) * 1e6)           # ns, like the reference's timer
        avg, sig = statistics.mean(ts), (statistics.stdev(ts) if len(ts) > 1 else 0.0)
        print(f"N=2^{lg} batch={batch} variant={v:2d} iters={iters:5d} launches={plan.num_launches} avg {avg/1e3:9.1f} us "
              f"sigma {sig/1e3:7.1f} us {n * batch / avg:7.2f} Gsamples/s", flush=True)
        return avg

    lines = []
    for lg in range(args.min_log2, args.max_log2 + 1):
        n = 1 << lg
        for batch in (batches_for(n, args.total_log2) if args.batches else [max(1, (1 << args.total_log2) // n)]):
            x = (torch.rand(batch * 2 * n, device="cuda") * 2 - 1).half()
            y = torch.empty_like(x)
            best = None
            for v in candidates(n):
                avg = time_plan(n, batch, x, y, v, 0)
                if avg is not None and (best is None or avg < best[0]):
                    best = (avg, v, 0)
            # Launch shapes only differ when a CU gets more than one round of work: every kernel moves at least 4096 samples per
            # wave and round, so below 256 CUs x 8 waves x 4096 samples all candidates launch the same grid and a "win" is noise
            # (round 3's file carried `256 ... 0 2 1`: launch_iters = 2 at batch 1). Above it a shape has to win twice, by more
            # than 2 % each time, against a re-timed incumbent.
            if n * batch > 256 * 8 * 4096:
                for it in iters_candidates()[1:]:
                    avg = time_plan(n, batch, x, y, best[1], it)
                    if avg is None or not avg < 0.98 * best[0]:
                        continue
                    again = time_plan(n, batch, x, y, best[1], it)
                    base = time_plan(n, batch, x, y, best[1], best[2])
                    if again is not None and base is not None and again < 0.98 * base:
                        best = (min(avg, again), best[1], it)
            mode = 4096 if n >= 4096 else 256
            lines.append(f"{n} {mode} {16 if mode == 4096 else 1} 1 256 {best[1]} {best[2]} {batch}")
    with open(args.out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print(f"wrote {args.out}")


if __name__ == "__main__":
    main()
