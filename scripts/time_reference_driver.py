#!/usr/bin/env python3
"""The REAL binary an unchanged user runs -- oracle/_ref/grtcode_driver: the reference's framework/src/driver.c main()
and argparse.c compiled unchanged, linked to this library (oracle/Makefile) -- timed on G1-sized bands (VERDICT r2 #5c).

The driver reads ONE HITRAN file for both bands, so the bench's two separate synthetic lists cannot be used as they are:
this list has the bench's line COUNTS per band -- 1.0 M lines in 1-3250 cm-1, 0.5 M more in 3250-50 000 cm-1, i.e.
1.0 M in the longwave band and 1.5 M in the shortwave band -- with the bench's line-parameter distributions.
Grids: LW 1-3250, SW 1-50 000 cm-1 at 1 cm-1, 61 levels, -integrated, 7 molecules, continua, 2 CFCs, 3 CIA pairs.
The driver prints no timings, so runs of N1 and N2 columns are timed whole and the per-column time is the difference
quotient (start-up -- parsing 240 MB of line list, building the stores -- cancels).

    python scripts/time_reference_driver.py [--columns 40 240] [--out profiles/r3_reference_driver_timing.json]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from grtcode_amd import synthetic as syn, workload as W  # noqa: E402

DRIVER = os.path.join(ROOT, "oracle", "_ref", "grtcode_driver")
NAME = {syn.H2O: "H2O", syn.CO2: "CO2", syn.O3: "O3", syn.N2O: "N2O", syn.CO: "CO", syn.CH4: "CH4", syn.O2: "O2"}


def write_par_fast(path, lists):
    """synthetic.write_hitran_par for millions of lines: the same records, formatted array-wise."""
    mol = np.concatenate([np.full(ln["v0"].size, m) for m, ln in lists.items()])
    cat = {k: np.concatenate([ln[k] for ln in lists.values()]) for k in ("v0", "s0", "yair", "yself", "en", "nexp", "delta", "iso")}
    order = np.argsort(cat["v0"], kind="stable")
    with open(path, "w") as f:
        for j in order:
            iso = int(cat["iso"][j])
            iso_c = "0" if iso == 10 else (chr(ord("A") + iso - 11) if iso > 10 else str(iso))
            f.write("%2d%1s%12.6f%10.3E%10.3E%s%s%10.4f%s%s%s\n" % (
                mol[j], iso_c, cat["v0"][j], cat["s0"][j], 0.0, syn._fw(cat["yair"][j], 5, 4), syn._fw(cat["yself"][j], 5, 3),
                cat["en"][j], syn._fw(cat["nexp"][j], 4, 2), syn._fw(cat["delta"][j], 8, 6), " " * 93))


def column_text(c, V):
    col = syn.profile(c, V)
    p, t = col["p"], col["t"]
    pl = 0.5 * (p[:-1] + p[1:])
    lay = lambda x: 0.5e-6 * (x[:-1] + x[1:])
    rows = [("level_pressure", p), ("level_temperature", t), ("layer_pressure", pl), ("layer_temperature", col["t_layer"]),
            ("surface_temperature", [col["t_surf"]]), ("solar_zenith_angle", [np.degrees(np.arccos(col["mu0"]))]),
            ("toa_solar_irradiance", [col["tsi"] * col["mu0"]])]
    rows += [(NAME[m], lay(col["ppmv"][m])) for m in W.MOL_ORDER]
    rows += [("CFC11", lay(col["cfc_ppmv"][0])), ("CFC12", lay(col["cfc_ppmv"][1]))]
    return "column:\n" + "".join(name + ": " + " ".join(repr(float(x)) for x in vals) + "\n" for name, vals in rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--columns", type=int, nargs=2, default=[40, 640])   # (a 600-column difference: the start-up, 0.7 s of parsing, varies by tens of ms)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r4_reference_driver_timing.json"))
    ap.add_argument("--api-timing", action="store_true",
                    help="one more default run of the larger column count under scripts/api_timing_shim.c (LD_PRELOAD): the "
                         "time spent inside each per-column entry point of the library, printed and kept in the record")
    ap.add_argument("--sweep-item-lines", type=int, nargs="*", default=[],
                    help="instead of the timing runs: 100 default columns under the shim for each GRT_ITEM_LINES given "
                         "(lines per piece of the lone-column work list), the library's kernel brackets printed")
    args = ap.parse_args()
    V = W.NUM_LEVELS
    root = tempfile.mkdtemp(prefix="grt_drv_")
    lw = syn.band_line_lists(W.LW_LINES, 1.0, 3250.0, 20261003)
    hi = syn.band_line_lists(W.SW_LINES - W.LW_LINES, 3250.0, 50000.0, 20261004)
    lists = {m: {k: np.concatenate([lw[m][k], hi[m][k]]) for k in lw[m]} for m in W.MOL_ORDER}
    par = os.path.join(root, "lines.par")
    t0 = time.perf_counter()
    write_par_fast(par, lists)
    print(f"wrote {par}: {os.path.getsize(par) / 1e6:.0f} MB in {time.perf_counter() - t0:.0f} s", file=sys.stderr, flush=True)
    files, _ = W.write_tables(root, sw=True)
    runs = {}
    # "3rows": the default arithmetic with the driver's opt-in GRT_FLUX_ROWS=toa,sfc (only the rows `-integrated` integrates
    # cross PCIe: INTEGRATION.md section 7)
    if args.sweep_item_lines:
        shim = os.path.join(root, "libgrt_api_timing.so")
        subprocess.run(["gcc", "-std=gnu99", "-O2", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "scripts", "api_timing_shim.c"), "-ldl", "-o", shim], check=True)
        cols = os.path.join(root, "columns_100.txt")
        with open(cols, "w") as f:
            f.write("".join(column_text(c, V) for c in range(100)))
        cmd = [DRIVER, par, files["solar"], cols, *("-" + NAME[m] for m in W.MOL_ORDER), "-h2o-ctm", files["h2o_dir"],
               "-o3-ctm", files["o3_ctm"], "-CFC-11", files["cfc11"], "-CFC-12", files["cfc12"], "-N2-N2", files["cia_n2n2"],
               "-O2-N2", files["cia_o2n2"], "-O2-O2", files["cia_o2o2"], "-a", "0.2", "-e", "0.98",
               "-w-lw", "1", "-W-lw", "3250", "-r-lw", "1", "-w-sw", "1", "-W-sw", "50000", "-r-sw", "1",
               "-integrated", "-o", os.path.join(root, "out_sweep.txt")]
        for n in args.sweep_item_lines:
            env = dict(os.environ, GRT_GAS_OPTICS_FAST="3", GRT_TIPS_QUIET="1", GRT_HITRAN_CACHE_DIR=root, LD_PRELOAD=shim,
                       GRT_ITEM_LINES=str(n))
            r = subprocess.run(cmd, capture_output=True, text=True, env=env)
            print(f"GRT_ITEM_LINES={n}", file=sys.stderr)
            print("\n".join(ln for ln in r.stderr.splitlines() if "bracket" in ln or "inside" in ln), file=sys.stderr, flush=True)
        return
    for fast in ("3", "3rows", "0"):
        for n in args.columns:
            cols = os.path.join(root, f"columns_{n}.txt")
            with open(cols, "w") as f:
                f.write("".join(column_text(c, V) for c in range(n)))
            cmd = [DRIVER, par, files["solar"], cols, *("-" + NAME[m] for m in W.MOL_ORDER), "-h2o-ctm", files["h2o_dir"],
                   "-o3-ctm", files["o3_ctm"], "-CFC-11", files["cfc11"], "-CFC-12", files["cfc12"], "-N2-N2", files["cia_n2n2"],
                   "-O2-N2", files["cia_o2n2"], "-O2-O2", files["cia_o2o2"], "-a", "0.2", "-e", "0.98",
                   "-w-lw", "1", "-W-lw", "3250", "-r-lw", "1", "-w-sw", "1", "-W-sw", "50000", "-r-sw", "1",
                   "-integrated", "-o", os.path.join(root, f"out_{fast}_{n}.txt")]
            env = dict(os.environ, GRT_GAS_OPTICS_FAST=fast[0], GRT_TIPS_QUIET="1", GRT_HITRAN_CACHE_DIR=root)
            if fast == "3rows":
                env["GRT_FLUX_ROWS"] = "toa,sfc"
            wall = 1e30
            for attempt in range(3):            # (the first run of all also writes the line-list index; best of three)
                t0 = time.perf_counter()
                r = subprocess.run(cmd, capture_output=True, text=True, env=env)
                wall = min(wall, time.perf_counter() - t0)
                if r.returncode != 0:
                    raise SystemExit(r.stderr[-3000:])
            runs[f"fast{fast}_{n}_columns_wall_s"] = wall
            if args.api_timing and fast == "3" and n == args.columns[1]:
                shim = os.path.join(root, "libgrt_api_timing.so")
                subprocess.run(["gcc", "-std=gnu99", "-O2", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                                os.path.join(ROOT, "scripts", "api_timing_shim.c"), "-ldl", "-o", shim], check=True)
                r = subprocess.run(cmd, capture_output=True, text=True, env=dict(env, LD_PRELOAD=shim))
                api_lines = [ln for ln in r.stderr.splitlines() if ln.startswith("api_timing")]
                print("\n".join(api_lines), file=sys.stderr, flush=True)
                runs["api_timing_of_the_default_run"] = {"columns": n, "lines": api_lines}
            print(f"fast={fast} {n} columns: {wall:.2f} s", file=sys.stderr, flush=True)
    n1, n2 = args.columns
    out = {"binary": "oracle/_ref/grtcode_driver (reference framework/src/driver.c + utilities/src/argparse.c unchanged, "
                     "examples/driver_app.c, libgrtcode_hip.so)", "options": "-integrated, LW 1-3250 + SW 1-50000 cm-1 @1 cm-1, 61 levels",
           "lines": {"lw_band": int(sum((ln["v0"] <= 3250.0).sum() for ln in lists.values())),
                     "sw_band": int(sum(ln["v0"].size for ln in lists.values()))}, "runs": runs}
    if args.sweep_item_lines:
        shim = os.path.join(root, "libgrt_api_timing.so")
        subprocess.run(["gcc", "-std=gnu99", "-O2", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "scripts", "api_timing_shim.c"), "-ldl", "-o", shim], check=True)
        cols = os.path.join(root, "columns_100.txt")
        with open(cols, "w") as f:
            f.write("".join(column_text(c, V) for c in range(100)))
        cmd = [DRIVER, par, files["solar"], cols, *("-" + NAME[m] for m in W.MOL_ORDER), "-h2o-ctm", files["h2o_dir"],
               "-o3-ctm", files["o3_ctm"], "-CFC-11", files["cfc11"], "-CFC-12", files["cfc12"], "-N2-N2", files["cia_n2n2"],
               "-O2-N2", files["cia_o2n2"], "-O2-O2", files["cia_o2o2"], "-a", "0.2", "-e", "0.98",
               "-w-lw", "1", "-W-lw", "3250", "-r-lw", "1", "-w-sw", "1", "-W-sw", "50000", "-r-sw", "1",
               "-integrated", "-o", os.path.join(root, "out_sweep.txt")]
        for n in args.sweep_item_lines:
            env = dict(os.environ, GRT_GAS_OPTICS_FAST="3", GRT_TIPS_QUIET="1", GRT_HITRAN_CACHE_DIR=root, LD_PRELOAD=shim,
                       GRT_ITEM_LINES=str(n))
            r = subprocess.run(cmd, capture_output=True, text=True, env=env)
            print(f"GRT_ITEM_LINES={n}", file=sys.stderr)
            print("\n".join(ln for ln in r.stderr.splitlines() if "bracket" in ln or "inside" in ln), file=sys.stderr, flush=True)
        return
    for fast in ("3", "3rows", "0"):
        dt = (runs[f"fast{fast}_{n2}_columns_wall_s"] - runs[f"fast{fast}_{n1}_columns_wall_s"]) / (n2 - n1)
        out[f"fast{fast}"] = {"seconds_per_column": dt, "columns_per_s": 1.0 / dt,
                              "startup_s": runs[f"fast{fast}_{n1}_columns_wall_s"] - n1 * dt}
    out["note"] = ("fast3 = the default of an unchanged driver (production arithmetic); fast0 = GRT_GAS_OPTICS_FAST=0, the reference's "
                   "operation order.  Every column moves 2 x 61 x 53 250 doubles of spectral flux to the host (52 MB: ~1 ms at the "
                   "57 GB/s this box's PCIe delivers) and is integrated there by driver.c:302-326.")
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
