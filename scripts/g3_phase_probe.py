"""Where the first pass of the 0.001 cm-1 column spends its clocks: the instrumented instance of the twelve-moment tree
form (grt_gas_optics_probe) on ONE 60-layer longwave column of 10^6 lines.

    PYTHONPATH=. python scripts/g3_phase_probe.py [--dw 0.001] [--out profiles/r3_g3_phases.json]
"""
import argparse
import ctypes as C
import json
import sys
import tempfile

import numpy as np

from grtcode_amd import api, synthetic as syn, workload as W

WORDS = 24


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dw", type=float, default=0.001)
    ap.add_argument("--lines", type=int, default=W.LW_LINES)
    ap.add_argument("--out", default=None)
    ap.add_argument("--sw", action="store_true", help="the shortwave band (1 - 50 000 cm-1, 1.5e6 lines) instead of the longwave one")
    args = ap.parse_args()
    lib = api.load_library()
    device = api.create_device(0)
    root = tempfile.mkdtemp(prefix="grt_g3p_")
    files, _ = W.write_tables(root, sw=args.sw)
    spec = (1.0, 50000.0 if args.sw else 3250.0, args.dw)
    lines = W.band_lines(W.SW_LINES if args.sw and args.lines == W.LW_LINES else args.lines, spec, 20261004 if args.sw else 20261003)
    go, grid = W.build_band(device, spec, lines, files, W.NUM_LEVELS)
    col = syn.profile(0, W.NUM_LEVELS)
    for m in W.MOL_ORDER:
        go.set_molecule_ppmv(m, col["ppmv"][m])
    go.set_cfc_ppmv(0, col["cfc_ppmv"][0])
    go.set_cfc_ppmv(1, col["cfc_ppmv"][1])
    go.set_cia_ppmv(0, col["ppmv"][syn.N2])
    go.set_cia_ppmv(1, col["ppmv"][syn.O2])
    L = W.NUM_LEVELS - 1
    opt = api.OpticsObject(L, grid, device)
    go.tune(fast=3)
    go.calculate_optical_depth(col["p"], col["t"], opt)
    api.profile_enable(True)
    for _ in range(2):
        go.calculate_optical_depth(col["p"], col["t"], opt)
    prod = {t: api.profile_read(t) for t in (1, 2, 6, 7)}
    tag = 1 if prod[1][1] else 2
    api.profile_read(tag, reset=True)              # (a reset clears every tag)
    info = go.last_launch()
    tile, nslice = int(info["tile"]), int(info["nslice"])
    ntiles = (int(grid.n) + tile - 1) // tile
    nrec = L * ntiles * nslice
    buf = api.DeviceBuffer(device, 8 * WORDS * nrec)
    zeros = np.zeros(WORDS * nrec, dtype=np.uint64)
    api.check(lib.grt_host_to_device(device, buf.ptr, zeros.ctypes.data_as(C.c_void_p), zeros.nbytes))
    api.check(lib.grt_gas_optics_probe(C.byref(go.c), buf.ptr, C.c_uint64(WORDS * nrec)))
    go.calculate_optical_depth(col["p"], col["t"], opt)
    probe_ms = api.profile_read(tag)[0]
    rec = buf.to_host((L, ntiles, nslice, WORDS), dtype=np.uint64).astype(np.float64)
    api.check(lib.grt_gas_optics_probe(C.byref(go.c), None, C.c_uint64(0)))
    assert np.all(rec[..., 1] > 0), "a workgroup left no record"
    wg = rec[..., 1] - rec[..., 0]
    phases = {"prologue": rec[..., 11] - rec[..., 0], "line_loop": rec[..., 12] - rec[..., 11], "epilogue": rec[..., 1] - rec[..., 12]}
    names = ["preparation", "moment_stores", "walk_and_queue_pushes", "region1_corrections", "near_field_ring", "rest", "queue_evaluation", "moment_terms"]
    inloop = {n: rec[..., 14 + i] for i, n in enumerate(names)}
    tot_in = sum(v.sum() for v in inloop.values())
    cnt = ["blocks64", "ring_steps", "near_points", "moment_reductions", "moment_lane_adds", "reg1_steps", "walk_steps"]
    R = (rec[..., 3].astype(np.uint64) & np.uint64(0xffff)).astype(np.float64)
    out = {"what": "first pass of the tree form, one longwave column at %g cm-1" % args.dw, "ran": info,
           "production_first_pass_ms": prod[tag][0] / max(prod[tag][1], 1), "production_gather_ms": prod[tag + 5][0] / max(prod[tag + 5][1], 1),
           "probe_first_pass_ms": probe_ms,
           "workgroup_clock_shares": {k: float(v.sum() / wg.sum()) for k, v in phases.items()},
           "wave_clock_shares_inside_the_line_loop": {n: float(v.sum() / tot_in) for n, v in inloop.items()},
           "counts_per_block64": {n: float(rec[..., 4 + i].sum() / max(rec[..., 4].sum(), 1)) for i, n in enumerate(cnt)},
           "lines_per_workgroup": float(rec[..., 2].mean()),
           "ring_steps_by_form": {"general": float((rec[..., 5] - rec[..., 22] - rec[..., 23]).sum() / rec[..., 5].sum()),
                                  "no_range_test": float(rec[..., 22].sum() / rec[..., 5].sum()),
                                  "lorentzian_only": float(rec[..., 23].sum() / rec[..., 5].sum())},
           "by_layer": [{"layer": int(l), "R": float(R[l].mean()), "share_of_workgroup_clocks": float(wg[l].sum() / wg.sum()),
                         "ring_steps_per_block": float(rec[l, ..., 5].sum() / max(rec[l, ..., 4].sum(), 1)),
                         "ring_general_share": float((rec[l, ..., 5] - rec[l, ..., 22] - rec[l, ..., 23]).sum() / max(rec[l, ..., 5].sum(), 1)),
                         "near_points_per_line": float(rec[l, ..., 6].sum() / max(rec[l, ..., 2].sum(), 1)),
                         "inside": {n: float(inloop[n][l].sum() / max(sum(inloop[m][l].sum() for m in names), 1)) for n in names}}
                        for l in range(0, L, 6)]}
    text = json.dumps(out, indent=1)
    if args.out:
        open(args.out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    sys.exit(main())
