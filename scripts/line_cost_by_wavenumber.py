#!/usr/bin/env python3
"""Where the line kernel's time goes, by wavenumber and by layer (a shortwave line costs 1.66x a
longwave line at 1 cm-1 -- why?).

Runs the bench's G1 workload (8 columns per launch, fast = 3) once per band with the INSTRUMENTED instance of the
two-pass first pass (grt_gas_optics_probe, include/grt_ext.h): every workgroup = (cell tile, layer, column) leaves its
entry and exit clocks and its event counts.  A workgroup's clocks include the time it shares its CU with three others,
so a launch's measured duration (HIP events) is attributed to tiles and layers in proportion to their workgroup-cycles.

    python scripts/line_cost_by_wavenumber.py [--cols 8] [--out profiles/r3_sw_cost_by_wavenumber.json]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WORDS = 24


def probe_band(api, wl, go, grid, tag, far_tag, cols, gcols):
    lib = api.load_library()
    L = wl.num_levels - 1
    tau = api.DeviceBuffer(wl.device, 8 * cols * L * int(grid.n))
    # production instance first: timing and the launch geometry
    api.profile_enable(True)
    for _ in range(3):
        api.check(lib.grt_optical_depth_batch(C.byref(go.c), C.byref(gcols), tau.ptr))
    wl.pipe.sync()
    api.profile_read(tag, reset=True)
    for _ in range(5):
        api.check(lib.grt_optical_depth_batch(C.byref(go.c), C.byref(gcols), tau.ptr))
    wl.pipe.sync()
    prod_ms = api.profile_read(tag)[0] / 5
    far_ms = api.profile_read(far_tag)[0] / 5
    info = go.last_launch()
    tile, nslice = int(info["tile"]), int(info["nslice"])
    ntiles = (int(grid.n) + tile - 1) // tile
    nrec = cols * L * ntiles * nslice
    buf = api.DeviceBuffer(wl.device, 8 * WORDS * nrec)
    zeros = np.zeros(WORDS * nrec, dtype=np.uint64)
    api.check(lib.grt_host_to_device(wl.device, buf.ptr, zeros.ctypes.data_as(C.c_void_p), zeros.nbytes))
    api.check(lib.grt_gas_optics_probe(C.byref(go.c), buf.ptr, C.c_uint64(WORDS * nrec)))
    api.profile_read(tag, reset=True)
    api.check(lib.grt_optical_depth_batch(C.byref(go.c), C.byref(gcols), tau.ptr))
    wl.pipe.sync()
    probe_ms = api.profile_read(tag)[0]
    rec = buf.to_host((cols, L, ntiles, nslice, WORDS), dtype=np.uint64).astype(np.float64)
    api.check(lib.grt_gas_optics_probe(C.byref(go.c), None, C.c_uint64(0)))
    buf.free()
    tau.free()
    cyc = rec[..., 1] - rec[..., 0]
    phase = {"prologue": rec[..., 11] - rec[..., 0], "line_loop_until_last_wave": rec[..., 12] - rec[..., 11],
             "epilogue": rec[..., 1] - rec[..., 12]}
    assert np.all(rec[..., 1] > 0) and np.all(cyc > 0), "a workgroup left no record"
    total_cyc = cyc.sum()
    names = ["blocks64", "ring_steps", "near_points", "moment_reductions", "moment_lane_adds", "reg1_steps", "walk_steps"]
    cnt = {n: rec[..., 4 + i] for i, n in enumerate(names)}
    lines = rec[..., 2]
    R = (rec[..., 3].astype(np.uint64) & np.uint64(0xffff)).astype(np.float64)
    corrected = ((rec[..., 3].astype(np.uint64) >> np.uint64(16)) & np.uint64(1)).astype(np.float64)
    w0, dw = float(grid.w0), float(grid.dw)

    inside_names = ["preparation", "moment_reduction_and_adds", "walk_and_queue_pushes", "region1_corrections", "near_field", "rest", "queue_evaluation", "moment_terms"]
    inside = {n: rec[..., 14 + i] for i, n in enumerate(inside_names)}

    def summarise(sel_cyc, sel_lines, sel_cnt, sel_R, sel_corr, sel_inside):
        nl = sel_lines.sum()
        tot_inside = max(sum(v.sum() for v in sel_inside.values()), 1.0)
        blocks = max(sel_cnt["blocks64"].sum(), 1.0)
        ms = probe_ms * sel_cyc.sum() / total_cyc
        return {"ms_of_probe_launch": ms, "ms_of_production_launch": prod_ms * sel_cyc.sum() / total_cyc,
                "share": sel_cyc.sum() / total_cyc, "line_layer_columns": nl,
                "ns_per_line_layer_column": 1e6 * (prod_ms * sel_cyc.sum() / total_cyc) / max(nl, 1.0),
                # VERDICT r2's normalisation: launch time over the lines of the store (a launch = L layers x cols columns)
                "ns_per_line_per_launch": 1e6 * (prod_ms * sel_cyc.sum() / total_cyc) / max(nl / (L * cols), 1.0),
                "workgroup_kcycles_per_64_lines": 1e-3 * sel_cyc.sum() / max(nl / 64.0, 1.0),
                "ring_steps_per_block": sel_cnt["ring_steps"].sum() / blocks,
                "near_points_per_line": sel_cnt["near_points"].sum() / max(nl, 1.0),
                "moment_reductions_per_block": sel_cnt["moment_reductions"].sum() / blocks,
                "moment_lane_adds_per_line": sel_cnt["moment_lane_adds"].sum() / max(nl, 1.0),
                "reg1_correction_steps_per_block": sel_cnt["reg1_steps"].sum() / blocks,
                "walk_steps_per_block": sel_cnt["walk_steps"].sum() / blocks,
                "blocks64_per_workgroup": blocks / max(sel_cyc.size, 1),
                "lines_per_block_worked": nl / blocks,
                "mean_R": float(sel_R.mean()), "corrected_fraction": float(sel_corr.mean()),
                # the waves' own clocks inside the line loop, by phase (shares)
                "wave_clock_shares_in_line_loop": {n: float(v.sum() / tot_inside) for n, v in sel_inside.items()}}

    def pick(idx_tiles=slice(None), idx_layers=slice(None)):
        sub = lambda a: a[:, idx_layers][:, :, idx_tiles]
        return summarise(sub(cyc), sub(lines), {k: sub(v) for k, v in cnt.items()}, sub(R), sub(corrected),
                         {k: sub(v) for k, v in inside.items()})

    # by wavenumber: groups of tiles covering ~1 000 cm-1 (whole band for the longwave in ~250 cm-1 steps)
    span = 1000.0 if grid.n > 10000 else 256.0
    per_group = max(1, int(round(span / (tile * dw))))
    by_w = []
    for t0 in range(0, ntiles, per_group):
        t1 = min(t0 + per_group, ntiles)
        d = pick(slice(t0, t1))
        d.update(w_lo=w0 + t0 * tile * dw, w_hi=w0 + min(t1 * tile, int(grid.n)) * dw)
        by_w.append(d)
    by_layer = []
    for l in range(L):
        d = pick(slice(None), slice(l, l + 1))
        d.update(layer=l)
        by_layer.append(d)
    return {"n": int(grid.n), "tile": tile, "nslice": nslice, "tiles": ntiles, "workgroups": int(nrec),
            "production_first_pass_ms": prod_ms, "production_gather_ms": far_ms, "probe_first_pass_ms": probe_ms,
            "lines_in_store": int(sum(v["v0"].size for v in (wl.lw_lines if tag == 1 else wl.sw_lines).values())),
            "whole_launch": pick(), "by_wavenumber": by_w, "by_layer": by_layer,
            # where a workgroup's own clock goes: prologue (column state, temperature-power table, candidate range, clearing
            # LDS) | the loop over its lines, until the LAST of its four waves is through | epilogue (flush to tau and gmom)
            "workgroup_phases_share": {k: float(v.sum() / cyc.sum()) for k, v in phase.items()},
            "workgroup_kcycles_mean": float(cyc.mean() * 1e-3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cols", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r3_sw_cost_by_wavenumber.json"))
    ap.add_argument("--lw-lines", type=int, default=None)
    ap.add_argument("--sw-lines", type=int, default=None)
    args = ap.parse_args()
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    from grtcode_amd import api, workload as W
    device = api.create_device(0)
    wl = W.G1Workload(device, args.cols, lw_lines=args.lw_lines or W.LW_LINES, sw_lines=args.sw_lines or W.SW_LINES, fast=3)
    (gcols, keep), _ = wl.columns(0, args.cols)
    out = {"what": "cost of the two-pass line kernel's first pass by wavenumber and layer, G1 workload, "
                   f"{args.cols} columns per launch; launch time attributed in proportion to workgroup-cycles (see the script)",
           "lw": probe_band(api, wl, wl.go_lw, wl.grid_lw, 1, 6, args.cols, gcols),
           "sw": probe_band(api, wl, wl.go_sw, wl.grid_sw, 2, 7, args.cols, gcols)}
    lw, sw = out["lw"]["whole_launch"], out["sw"]["whole_launch"]
    out["sw_over_lw_ns_per_line"] = sw["ns_per_line_layer_column"] / lw["ns_per_line_layer_column"]
    wl.destroy()
    os.dup2(saved, 1)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    keys = ("ns_per_line_layer_column", "ring_steps_per_block", "near_points_per_line", "moment_reductions_per_block",
            "moment_lane_adds_per_line", "reg1_correction_steps_per_block", "walk_steps_per_block", "lines_per_block_worked",
            "blocks64_per_workgroup", "mean_R", "corrected_fraction")
    print("band      " + " ".join(f"{k[:14]:>14s}" for k in keys))
    for name, band in (("lw", out["lw"]), ("sw", out["sw"])):
        print(f"{name} all    " + " ".join(f"{band['whole_launch'][k]:14.3f}" for k in keys))
        for d in band["by_wavenumber"]:
            print(f"{d['w_lo']:6.0f}    " + " ".join(f"{d[k]:14.3f}" for k in keys) + f"  {d['ms_of_production_launch']:.3f} ms")
    print("workgroup phases lw", out["lw"]["workgroup_phases_share"], "sw", out["sw"]["workgroup_phases_share"])
    print("production first pass ms: lw", out["lw"]["production_first_pass_ms"], "sw", out["sw"]["production_first_pass_ms"],
          "probe:", out["lw"]["probe_first_pass_ms"], out["sw"]["probe_first_pass_ms"])


if __name__ == "__main__":
    main()
