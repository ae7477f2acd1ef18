"""Summarise tools/profile_final.sh's output directory: per-launch averages of every PMC counter for
k_trace dispatches, the corrected HBM traffic (MI355X_MICROARCH.md: KB units, FETCH_SIZE x2 on
gfx950), and the kernel-trace average duration."""
import csv
import glob
import json
import sys
from collections import defaultdict

out = sys.argv[1]
vals = defaultdict(list)
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in vals.items()}
kern = None
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace" in r["Name"]:
            kern = {"name": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
res = {"kernel_trace": kern, "per_launch_averages": avg}
if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
    wb, fb = avg["WRITE_SIZE"] * 1024, avg["FETCH_SIZE"] * 1024
    res["hbm_traffic"] = {"write_bytes": wb, "fetch_bytes_raw": fb, "fetch_bytes_corrected_x2": 2 * fb, "traffic_bytes": wb + 2 * fb}
print(json.dumps(res, indent=1))
