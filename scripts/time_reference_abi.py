"""Where the time of the reference-shaped one-column interface goes (framework/src/driver.c:360-424 call by call) on
the bench workload: wall time of every call, synchronous as the interface is.

    PYTHONPATH=. python scripts/time_reference_abi.py [--fast 3] [--columns 3]
"""
import argparse
import collections
import json
import time

import numpy as np

from grtcode_amd import api, synthetic as syn, workload as W


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fast", type=int, default=0)
    ap.add_argument("--columns", type=int, default=3)
    args = ap.parse_args()
    device = api.create_device(0)
    wl = W.G1Workload(device, 1)
    V = W.NUM_LEVELS
    lw = api.LongwaveObject(V, wl.grid_lw, device)
    sw = api.ShortwaveObject(V, wl.grid_sw, device)
    t = collections.defaultdict(float)

    def timed(name, fn, *a):
        t0 = time.perf_counter()
        r = fn(*a)
        t[name] += time.perf_counter() - t0
        return r
    for name, go, grid in (("lw", wl.go_lw, wl.grid_lw), ("sw", wl.go_sw, wl.grid_sw)):
        go.tune(fast=args.fast)
    bufs = {"lw": (np.zeros((V, wl.grid_lw.n)), np.zeros((V, wl.grid_lw.n))), "sw": (np.zeros((V, wl.grid_sw.n)), np.zeros((V, wl.grid_sw.n)))}
    gas = {"lw": api.OpticsObject(V - 1, wl.grid_lw, device), "sw": api.OpticsObject(V - 1, wl.grid_sw, device)}
    ray = {"lw": api.OpticsObject(V - 1, wl.grid_lw, device), "sw": api.OpticsObject(V - 1, wl.grid_sw, device)}
    for c in range(args.columns + 1):
        if c == 1:
            t.clear()
        col = syn.profile(c, V)
        for name, go in (("lw", wl.go_lw), ("sw", wl.go_sw)):
            def setters():
                for m in W.MOL_ORDER:
                    go.set_molecule_ppmv(m, col["ppmv"][m])
                go.set_cfc_ppmv(0, col["cfc_ppmv"][0])
                go.set_cfc_ppmv(1, col["cfc_ppmv"][1])
                go.set_cia_ppmv(0, col["ppmv"][syn.N2])
                go.set_cia_ppmv(1, col["ppmv"][syn.O2])
            timed(name + " set_ppmv", setters)
            timed(name + " calculate_optical_depth", go.calculate_optical_depth, col["p"], col["t"], gas[name])
            timed(name + " rayleigh_scattering", ray[name].rayleigh, col["p"])
            tot = timed(name + " add_optics", api.add_optics, [gas[name], ray[name]])
            if name == "lw":
                timed("lw calculate_lw_fluxes", lw.fluxes, tot, col["t_surf"], col["t_layer"], col["t"], wl.emis, bufs["lw"])
            else:
                timed("sw calculate_sw_fluxes", sw.fluxes, tot, col["mu0"], 0.5, wl.albedo, wl.albedo, col["tsi"], wl.solar, bufs["sw"])
            timed(name + " destroy_optics", tot.destroy)
    total = sum(t.values())
    print(json.dumps({"fast": args.fast, "ran": {"lw": wl.go_lw.last_launch(), "sw": wl.go_sw.last_launch()},
                      "ms_per_column": {k: round(1e3 * v / args.columns, 3) for k, v in t.items()},
                      "total_ms_per_column": round(1e3 * total / args.columns, 2), "columns_per_s": round(args.columns / total, 2)}))


if __name__ == "__main__":
    main()
