#!/usr/bin/env python3
"""gpurun_out/prof_g3_<tag>/ (scripts/profile_g3.sh) -> profiles/<tag>_g3_kernel_stats.csv and the "g3" entry of
profiles/traffic_latest.json (per kernel: average launch ms from the trace, SQ_INSTS_VALU per launch), which bench.py
carries into fine_grid.G3 as issue utilisation -- labelled as coming from the profiled run."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
src = os.path.join(ROOT, "gpurun_out", f"prof_g3_{tag}")
newest = lambda pat: sorted(glob.glob(pat), key=os.path.getmtime)[-1]
shutil.copy(newest(os.path.join(src, "trace", "*", "*kernel_stats.csv")), os.path.join(ROOT, "profiles", f"{tag}_g3_kernel_stats.csv"))


def short(name):
    for k in ("gas_optics_mp_kernel", "gas_optics_tree_kernel", "gas_optics_tree_lane_kernel", "moment_up_kernel"):
        if k in name:
            return k
    return None


dur = collections.defaultdict(list)
for r in csv.DictReader(open(newest(os.path.join(src, "trace", "*", "*kernel_trace.csv")))):
    if short(r["Kernel_Name"]):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(newest(os.path.join(src, "pmc_sq", "*", "*counter_collection.csv")))):
    if short(r["Kernel_Name"]):
        cnt[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = lambda v: sum(v) / len(v)
# traffic passes (optional): FETCH_SIZE / WRITE_SIZE in KiB (FETCH doubled on gfx950, as in summarize_profile.py), L2 atomics
for sub in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_TCC_ATOMIC_sum_TCC_EA0_ATOMIC_sum",
            "pmc_TCC_EA0_RDREQ_32B_sum_TCC_EA0_RDREQ_64B_sum_TCC_EA0_RDREQ_128B_sum"):
    files = glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv"))
    if files:
        for r in csv.DictReader(open(sorted(files, key=os.path.getmtime)[-1])):
            if short(r["Kernel_Name"]):
                cnt[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
g3 = {"tag": tag, "command": "python3 scripts/fine_grid.py --dw 0.001 (one 60-layer longwave column, 10^6 lines)", "kernels": {}}
for k, v in dur.items():
    ms = mean(v[1:] if len(v) > 1 else v)                         # (the first launch carries the allocations' page faults)
    e = {"avg_launch_ms": ms, "launches_seen": len(v)}
    if k in cnt:
        e["sq"] = {c: mean(x) for c, x in cnt[k].items()}
        if "SQ_INSTS_VALU" in e["sq"]:
            e["issue_utilisation"] = e["sq"]["SQ_INSTS_VALU"] / (1024 * ms * 1e-3 * 2.4e9 * 0.5)
        if "FETCH_SIZE" in e["sq"] and "WRITE_SIZE" in e["sq"]:
            e["hbm_bytes_per_launch"] = 1024.0 * (2.0 * e["sq"]["FETCH_SIZE"] + e["sq"]["WRITE_SIZE"])
            if "TCC_EA0_RDREQ_128B_sum" in e["sq"]:
                # exact reads (profiles/r4_fetch_calibration.json); the FETCH_SIZE-doubled figure is kept beside it
                e["hbm_bytes_from_fetch_size_doubled"] = e["hbm_bytes_per_launch"]
                e["read_bytes_per_launch"] = (32.0*e["sq"].get("TCC_EA0_RDREQ_32B_sum", 0.0) + 64.0*e["sq"].get("TCC_EA0_RDREQ_64B_sum", 0.0)
                                              + 128.0*e["sq"]["TCC_EA0_RDREQ_128B_sum"])
                e["hbm_bytes_per_launch"] = e["read_bytes_per_launch"] + 1024.0*e["sq"]["WRITE_SIZE"]
            e["hbm_gb_per_s"] = e["hbm_bytes_per_launch"] / (ms * 1e-3) / 1e9
    g3["kernels"][k] = e
path = os.path.join(ROOT, "profiles", "traffic_latest.json")
t = json.load(open(path))
t["g3"] = g3
json.dump(t, open(path, "w"), indent=1)
print(json.dumps(g3, indent=1))
