"""Turns rocprofv3 --pmc CSVs (gpurun_out/pmc_*/.../*counter_collection.csv) into profiles/<tag>_pmc_summary.json.

HBM bytes follow MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KiB-units of
1024 B, collected in SEPARATE passes; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane)
coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
usage: python tools/summarize_pmc.py <tag> [kernel-substring]
"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else "fft4096_kernel"
vals = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if needle in row["Kernel_Name"]:
                vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
mean = {k: sum(v) / len(v) for k, v in vals.items()}
out = {"kernel": needle, "dispatches_per_counter": {k: len(v) for k, v in vals.items()}, "mean_per_dispatch": mean}
if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
    rd = mean["FETCH_SIZE"] * 1024 * 2
    wr = mean["WRITE_SIZE"] * 1024
    out["hbm_bytes_per_dispatch"] = {"read_corrected_x2": rd, "write": wr, "total": rd + wr,
                                     "note": "FETCH_SIZE doubled per the gfx950 correction; separate passes"}
if "SQ_WAVE_CYCLES" in mean:
    wc = mean["SQ_WAVE_CYCLES"]
    out["wave_time_shares"] = {k: mean[k] / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in mean}
if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and "GRBM_GUI_ACTIVE" in mean:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over the 1024 SIMDs (256 CUs x 4)
    out["mfma_busy_fraction"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (mean["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
p = os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json")
json.dump(out, open(p, "w"), indent=1, sort_keys=True)
print(open(p).read())
