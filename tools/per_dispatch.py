"""Per-dispatch durations of a rocprofv3 --kernel-trace CSV (to see launch-to-launch spread): python tools/per_dispatch.py FILE_kernel_trace.csv [name filter]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    name = r.get("Kernel_Name", "")
    if flt in name:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(f"{d:9.1f} us  start {int(r['Start_Timestamp']) % 10**10 / 1e3:12.1f}  {name[:70]}")
