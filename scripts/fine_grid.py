"""Time the line-by-line optical depths of ONE longwave column on a fine grid (SURVEY §8d grids G2 = 0.1 cm-1,
G3 = 0.001 cm-1, n = 3 249 001) and say which form of the kernel ran.

    PYTHONPATH=. python scripts/fine_grid.py --dw 0.001 [--fast 3] [--lines 1000000] [--reps 3]
"""
import argparse
import json
import tempfile
import time

import numpy as np

from grtcode_amd import api, synthetic as syn, workload as W


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dw", type=float, default=0.001)
    ap.add_argument("--w0", type=float, default=1.0)
    ap.add_argument("--wn", type=float, default=3250.0)
    ap.add_argument("--lines", type=int, default=W.LW_LINES)
    ap.add_argument("--levels", type=int, default=W.NUM_LEVELS)
    ap.add_argument("--fast", type=int, default=3)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--pscale", type=float, default=1.0, help="exploration: pressures scaled by this factor")
    ap.add_argument("--compare", type=int, default=None, help="also run this form and report the largest difference")
    args = ap.parse_args()

    device = api.create_device(0)
    root = tempfile.mkdtemp(prefix="grt_fine_")
    files, _ = W.write_tables(root, sw=False)
    grid_spec = (args.w0, args.wn, args.dw)
    lines = W.band_lines(args.lines, grid_spec, 20261003)
    go, grid = W.build_band(device, grid_spec, lines, files, args.levels)
    col = syn.profile(0, args.levels)
    col["p"] = col["p"] * args.pscale
    for m in W.MOL_ORDER:
        go.set_molecule_ppmv(m, col["ppmv"][m])
    go.set_cfc_ppmv(0, col["cfc_ppmv"][0])
    go.set_cfc_ppmv(1, col["cfc_ppmv"][1])
    go.set_cia_ppmv(0, col["ppmv"][syn.N2])
    go.set_cia_ppmv(1, col["ppmv"][syn.O2])
    opt = api.OpticsObject(args.levels - 1, grid, device)

    def timed(fast):
        go.tune(tile=args.tile, nslice=0, fast=fast)
        go.calculate_optical_depth(col["p"], col["t"], opt)       # warm-up (allocations)
        api.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            go.calculate_optical_depth(col["p"], col["t"], opt)
        dt = (time.perf_counter() - t0) / args.reps
        tags = {t: api.profile_read(t) for t in (1, 2, 6, 7)}
        api.profile_enable(False)
        return dt, go.last_launch(), {k: round(v[0] / max(v[1], 1), 3) for k, v in tags.items() if v[1]}

    dt, info, tags = timed(args.fast)
    out = {"grid": {"w0": args.w0, "wn": args.wn, "dw": args.dw, "n": int(grid.n)}, "layers": args.levels - 1,
           "lines": int(sum(v["v0"].size for v in lines.values())), "asked_fast": args.fast, "ran": info,
           "seconds_per_column": round(dt, 5), "kernel_ms": tags}
    if args.compare is not None:
        tau = opt.read()[0]
        dt2, info2, tags2 = timed(args.compare)
        other = opt.read()[0]
        scale = np.abs(other).max(axis=1, keepdims=True)
        out["compare"] = {"ran": info2, "seconds_per_column": round(dt2, 5), "kernel_ms": tags2,
                          "max_diff_of_layer_max": float(np.max(np.abs(tau - other) / scale))}
    print(json.dumps(out))
    opt.destroy()
    go.destroy()


if __name__ == "__main__":
    main()
