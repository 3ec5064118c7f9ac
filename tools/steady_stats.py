"""Steady-state kernel statistics from a `rocprofv3 --kernel-trace` CSV: per kernel the first `--skip-frac` of its dispatches (the
clock ramp / warm-up executions of the profiled command) are dropped and average / min / max / sigma of the rest are written in
rocprofv3's own kernel_stats.csv columns, so that a roofline fraction can be recomputed from profiles/ alone.

    python tools/steady_stats.py "gpurun_out/trace_TAG/*kernel_trace.csv" --warmup 120 --reps 40 --out profiles/TAG_kernel_stats.csv
    python tools/steady_stats.py TRACE --skip 320 --keep 220 ...        (bench.py: 200 cold + 100 ramp + 20 warm-up launches, then 200 timed + 20 single)
Also prints the per-dispatch durations of the first kernel (--list N) the way tools/per_dispatch.py did."""
import argparse
import csv
import glob
import statistics

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--warmup", type=int, default=0)
ap.add_argument("--reps", type=int, default=0)
ap.add_argument("--skip", type=int, default=None, help="dispatches per kernel to drop (instead of warmup / (warmup + reps))")
ap.add_argument("--keep", type=int, default=None, help="dispatches per kernel to keep after the skipped ones (default: all)")
ap.add_argument("--only", default="", help="substring a kernel name must contain")
ap.add_argument("--out", default=None)
ap.add_argument("--list", type=int, default=0)
args = ap.parse_args()
rows = []
for f in sorted(glob.glob(args.trace, recursive=True)):
    with open(f) as fh:
        rows.extend(csv.DictReader(fh))
by = {}
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    by.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = []
total = 0
for name, d in by.items():
    if args.only not in name:
        continue
    if args.skip is not None:
        skip = args.skip
    elif args.warmup + args.reps:
        skip = len(d) * args.warmup // (args.warmup + args.reps)
    else:
        skip = 0
    kept = d[skip:skip + args.keep] if args.keep else d[skip:]
    if not kept:
        continue
    out.append((name, len(kept), sum(kept), statistics.fmean(kept), min(kept), max(kept), statistics.pstdev(kept) if len(kept) > 1 else 0.0, len(d)))
    total += sum(kept)
out.sort(key=lambda t: -t[2])
lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"']
for name, calls, tot, avg, lo, hi, sd, all_calls in out:
    lines.append(f'"{name}",{calls},{tot},{avg:.6f},{100.0 * tot / total:.2f},{lo},{hi},{sd:.6f}')
    print(f"{name[:70]:70s} {calls:5d} of {all_calls:5d} dispatches: avg {avg / 1e3:9.1f} us  min {lo / 1e3:9.1f}  max {hi / 1e3:9.1f}  sigma {sd / 1e3:7.1f}")
if args.out:
    open(args.out, "w").write("\n".join(lines) + "\n")
if args.list and out:
    print("per dispatch (us), all dispatches of", out[0][0][:60], ":", " ".join(f"{v / 1e3:.0f}" for v in by[out[0][0]][:args.list]))
