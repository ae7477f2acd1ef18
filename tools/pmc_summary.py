"""Summarise tools/profile_final.sh's output directory: the dominant k_trace instantiation of the run, its
kernel-trace average duration, per-launch averages of every PMC counter, the corrected HBM traffic
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE/WRITE_SIZE in KB; FETCH_SIZE x2 on gfx950) and the per-FRAME figures
bench.py reads back (`per_frame`: traffic_bytes, insts_valu), keyed by the workload (`workload_key`)."""
import csv
import glob
import json
import sys
from collections import defaultdict

out = sys.argv[1]
kern = None
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace" in r["Name"] and (kern is None or float(r["TotalDurationNs"]) > kern["total_ns"]):
            kern = {"name": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                    "max_ns": float(r["MaxNs"]), "total_ns": float(r["TotalDurationNs"])}
vals = defaultdict(list)
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern is not None and r["Kernel_Name"] == kern["name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in vals.items()}
line = None
try:
    for l in open(out + "/stats_bench.log", errors="replace"):
        if l.startswith("{"):
            line = json.loads(l)
except Exception:
    pass
import os
res = {"head": os.environ.get("HEAD_SHA"), "command": "rocprofv3 --kernel-trace [--stats | --pmc <group>] -- python3 bench.py --profile-leg --steps N [--workload W]",
       "kernel_trace": kern, "per_launch_averages": avg}
# the launch's other kernels (k_bin_tiles: the binning kernel in front of every binned launch)
others = {}
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace" not in r["Name"] and r["Name"].startswith(("k_", "void k_")):
            others[r["Name"][:60]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])}
res["other_kernels"] = others
V = line["config"]["frames_per_launch"] if line else 1
if line:
    res["workload_key"] = line["config"].get("workload_key")
    res["workload"] = line["config"]["workload"]
    res["frames_per_launch"] = V
if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
    wb, fb = avg["WRITE_SIZE"] * 1024, avg["FETCH_SIZE"] * 1024
    res["hbm_traffic"] = {"write_bytes": wb, "fetch_bytes_raw": fb, "fetch_bytes_corrected_x2": 2 * fb, "traffic_bytes": wb + 2 * fb}
    res["per_frame"] = {"traffic_bytes": (wb + 2 * fb) / V, "insts_valu": avg.get("SQ_INSTS_VALU", 0.0) / V,
                        "insts_salu": avg.get("SQ_INSTS_SALU", 0.0) / V, "waves": avg.get("SQ_WAVES", 0.0) / V}
if kern and "SQ_INSTS_VALU" in avg:
    # VALU issue-slot utilisation over the kernel-trace duration: wave-instructions x 4 cycles / (1024 SIMDs x t x 2.4 GHz)
    res["valu_issue_frac"] = avg["SQ_INSTS_VALU"] * 4 / (1024 * kern["avg_ns"] * 1e-9 * 2.4e9)
if line:
    res["bench_line_under_rocprof"] = line
print(json.dumps(res, indent=1))
