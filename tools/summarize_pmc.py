"""Turns rocprofv3 --pmc CSVs into profiles/<tag>_pmc_summary.json.

HBM bytes follow MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KiB-units of
1024 B, collected in SEPARATE passes; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane)
coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.

usage: python tools/summarize_pmc.py <tag> [kernel-substring ...] [--dirs GLOB] [--alg-bytes B]
  default dirs: gpurun_out/pmc_<tag>_* (tools/profile_pmc.sh), falling back to gpurun_out/pmc_*.
  With one kernel substring the JSON is the flat round-1 form (bench.py reads `hbm_bytes_per_dispatch.total`);
  with several, one entry per kernel under "kernels".
"""
import argparse
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("needles", nargs="*", default=["fft4096_kernel"])
ap.add_argument("--dirs", default=None)
ap.add_argument("--alg-bytes", type=float, default=0.0, help="algorithmic HBM bytes per dispatch of the (first) kernel")
ap.add_argument("--trace", default=None, help="kernel-trace stats CSV to take average durations from")
ap.add_argument("--skip-frac", type=float, default=0.0,
                help="fraction of every kernel's dispatches (per counter, in dispatch order) that were warm-up executions of the profiled "
                     "command and are dropped: warmup / (warmup + reps) of tools/prof_workload.py")
args = ap.parse_args()
pattern = args.dirs or os.path.join(ROOT, "gpurun_out", f"pmc_{args.tag}_*")
dirs = sorted(glob.glob(pattern)) or sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_*")))
rows = []
for d in dirs:
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            rows.extend(csv.DictReader(fh))

durations = {}
if args.trace:
    for f in sorted(glob.glob(args.trace, recursive=True)):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "AverageNs" in r:
                    durations[r["Name"]] = (float(r["AverageNs"]), int(r["Calls"]))


def summarize(needle):
    vals = collections.defaultdict(list)
    names = set()
    for row in rows:
        if needle in row["Kernel_Name"]:
            vals[(row["Counter_Name"], row["Kernel_Name"])].append((int(row.get("Dispatch_Id", 0) or 0), float(row["Counter_Value"])))
            names.add(row["Kernel_Name"])
    steady = collections.defaultdict(list)
    for (counter, _), v in vals.items():            # steady state only: drop each kernel's warm-up dispatches
        v.sort()
        steady[counter].extend(x for _, x in v[int(len(v) * args.skip_frac):])
    vals = steady
    mean = {k: sum(v) / len(v) for k, v in vals.items()}
    out = {"kernel": needle, "kernel_names": sorted(names), "dispatches_per_counter": {k: len(v) for k, v in vals.items()},
           "warmup_fraction_dropped": args.skip_frac,
           "mean_per_dispatch": mean}
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        rd = mean["FETCH_SIZE"] * 1024 * 2
        wr = mean["WRITE_SIZE"] * 1024
        out["hbm_bytes_per_dispatch"] = {"read_corrected_x2": rd, "write": wr, "total": rd + wr,
                                         "note": "FETCH_SIZE doubled per the gfx950 correction; separate passes"}
        if args.alg_bytes:
            out["hbm_bytes_per_dispatch"]["over_algorithmic"] = (rd + wr) / args.alg_bytes
    if "SQ_WAVE_CYCLES" in mean:
        wc = mean["SQ_WAVE_CYCLES"]
        out["wave_time_shares"] = {k: mean[k] / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS") if k in mean}
    if "SQ_LDS_BANK_CONFLICT" in mean and mean.get("SQ_LDS_IDX_ACTIVE"):
        out["lds_bank_conflict_share_of_lds_active"] = mean["SQ_LDS_BANK_CONFLICT"] / mean["SQ_LDS_IDX_ACTIVE"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and "GRBM_GUI_ACTIVE" in mean:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over the 1024 SIMDs (256 CUs x 4)
        out["mfma_busy_fraction"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (mean["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    for name, (avg, calls) in durations.items():
        if needle in name:
            out.setdefault("kernel_trace", {})[name] = {"average_us": avg / 1e3, "calls": calls}
    return out


if len(args.needles) == 1:
    out = summarize(args.needles[0])
else:
    out = {"kernels": {n: summarize(n) for n in args.needles}}
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
p = os.path.join(ROOT, "profiles", f"{args.tag}_pmc_summary.json")
json.dump(out, open(p, "w"), indent=1, sort_keys=True)
print(open(p).read())
