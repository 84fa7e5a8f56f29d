#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by scripts/profile_round.sh on the GPU box) into the small,
tracked summaries under profiles/: kernel stats, HBM-side traffic per launch (reads from the L2's read requests by size
where that pass exists -- exact; else FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950's vector loads;
WRITE_SIZE as is, in KiB units), SQ counters.

    python scripts/summarize_profile.py r1
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def newest(pattern):
    """gpurun merges successive runs into the same directory: take the most recent file."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:] 


def short(name):
    for k in ("gas_optics_lean_kernel", "gas_optics_mp_kernel", "gas_optics_far_kernel", "gas_optics_kernel", "sw_kernel", "lw_kernel", "clear_sky_kernel", "integrate_rows_kernel",
              "reduce_partials_kernel",
              "fillBufferAligned", "copyBuffer"):
        if k in name:
            return k
    return name[:40]


stats = newest(os.path.join(src, "trace", "*", "*kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
bench = json.load(open(os.path.join(src, "bench_under_trace.json")))

# per-launch averages by (kernel, grid) from the kernel trace
trace = newest(os.path.join(src, "trace", "*", "*kernel_trace.csv"))[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    dur[(short(r["Kernel_Name"]), r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))].append(
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)


def counters(sub, names):
    f = newest(os.path.join(src, sub, "*", "*counter_collection.csv"))
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    if not f:
        return out
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] in names:
            out[(short(r["Kernel_Name"]), r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


fetch = counters("pmc_fetch", {"FETCH_SIZE"})
write = counters("pmc_write", {"WRITE_SIZE"})
rdreq = counters("pmc_rdreq", {"TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"})
sq = counters("pmc_sq", {"SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
                         "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES"})
summary = {"tag": tag, "command": "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras  (%d columns per step in launches of %d)" % (bench["config"]["columns_per_step"], bench["config"]["chunk_columns"]),
           "bench_line_under_trace": {k: bench[k] for k in ("value", "ms_per_step", "kernel_ms_per_step", "config")},
           "kernels": {}}
mean = lambda v: sum(v) / len(v)
for key in sorted(set(fetch) | set(write)):
    name, grid = key
    f = mean(fetch[key]["FETCH_SIZE"]) if key in fetch else None
    w = mean(write[key]["WRITE_SIZE"]) if key in write else None
    entry = {"grid_threads": int(grid), "launches_seen": len(fetch[key]["FETCH_SIZE"]) if key in fetch else None,
             "FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w,
             "hbm_bytes_per_launch": (2.0 * (f or 0.0) + (w or 0.0)) * 1024.0,
             "note": "FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B); WRITE_SIZE checked against the "
                     "memset/solver kernels' known byte counts in this run"}
    if key in rdreq:
        r = {c: mean(v) for c, v in rdreq[key].items()}
        rb = 32.0*r.get("TCC_EA0_RDREQ_32B_sum", 0.0) + 64.0*r.get("TCC_EA0_RDREQ_64B_sum", 0.0) + 128.0*r.get("TCC_EA0_RDREQ_128B_sum", 0.0)
        entry["read_requests_per_launch"] = r
        entry["read_bytes_per_launch"] = rb
        entry["hbm_bytes_from_fetch_size_doubled"] = entry["hbm_bytes_per_launch"]
        entry["hbm_bytes_per_launch"] = rb + (w or 0.0)*1024.0
        entry["note"] = ("reads: the L2's memory-side read requests by size, 32 n32 + 64 n64 + 128 n128 (exact: "
                         "profiles/r4_fetch_calibration.json); FETCH_SIZE doubled is kept beside it; WRITE_SIZE checked against "
                         "the memset/solver kernels' known byte counts")
    if key in sq:
        entry["sq"] = {c: mean(v) for c, v in sq[key].items()}
    summary["kernels"][f"{name}@{grid}"] = entry
json.dump(summary, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)

# what bench.py reads back into roofline.traffic (dominant kernel = SW-band launch = largest grid)
gas = {k: v for k, v in summary["kernels"].items() if k.startswith("gas_optics") and "far_kernel" not in k}
if gas:
    sw_key = max(gas, key=lambda k: gas[k]["grid_threads"])
    lw_key = min(gas, key=lambda k: gas[k]["grid_threads"])
    cfg = bench["config"]
    # the solvers, for roofline_solvers.traffic
    solv = {k.split("@")[0]: v for k, v in summary["kernels"].items() if k.startswith(("sw_kernel", "lw_kernel"))}
    latest = os.path.join(dst, "traffic_latest.json")
    keep = {}
    if os.path.exists(latest):
        keep = {k: v for k, v in json.load(open(latest)).items() if k == "g3"}        # (scripts/summarize_g3.py's entry)
    # (bench.py carries these counters into its line only while the kernels' source is the one they were taken on: ADVICE r4)
    import hashlib
    src_sha = hashlib.sha256(b"".join(open(os.path.join(ROOT, "grtcode_amd", "csrc", "hip", f), "rb").read()
                                      for f in ("k_gas_optics_mp.hip", "mp_general_block.inc", "mp_lean_block.inc", "k_gas_optics_far.hip", "gas_optics_mp_dev.h", "gas_optics_dev.h"))).hexdigest()
    json.dump(dict({"tag": tag, "kernel_source_sha256": src_sha, "cols": cfg["chunk_columns"], "fast": cfg["fast"], "solvers": solv,
                    "gas_optics_sw": gas[sw_key], "gas_optics_lw": gas[lw_key]}, **keep),
              open(latest, "w"), indent=1)
print(json.dumps(summary, indent=1)[:1800])
