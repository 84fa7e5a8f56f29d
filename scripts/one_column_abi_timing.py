"""Where an unchanged one-column caller's time goes (the call sequence of framework/src/driver.c:360-424 on the G1 bands),
phase by phase, with the caller's flux arrays as they come (pageable) and registered with the HIP runtime.

    PYTHONPATH=. python scripts/one_column_abi_timing.py [--cols 20] [--out profiles/r3_one_column_abi.json]
"""
import argparse
import ctypes as C
import json
import time

import numpy as np

from grtcode_amd import api, synthetic as syn, workload as W


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cols", type=int, default=20)
    ap.add_argument("--out", default=None)
    ap.add_argument("--free", action="store_true",
                    help="no device synchronisation between the phases: what an unchanged caller sees; the phase times "
                         "are then the times the calls themselves take on the host")
    args = ap.parse_args()
    device = api.create_device(0)
    wl = W.G1Workload(device, 1, fast=3)
    V = W.NUM_LEVELS
    lw = api.LongwaveObject(V, wl.grid_lw, device)
    sw = api.ShortwaveObject(V, wl.grid_sw, device)
    objs = {}
    for name, go, grid in (("lw", wl.go_lw, wl.grid_lw), ("sw", wl.go_sw, wl.grid_sw)):
        objs[name] = (go, api.OpticsObject(V - 1, grid, device), api.OpticsObject(V - 1, grid, device), grid,
                      (np.zeros((V, grid.n)), np.zeros((V, grid.n))))
    phases = {}

    def tick(key, t0):
        phases[key] = phases.get(key, 0.0) + (time.perf_counter() - t0)

    def column(c):
        col = syn.profile(c, V)
        for name in ("lw", "sw"):
            go, gas, ray, grid, bufs = objs[name]
            t0 = time.perf_counter()
            for m in W.MOL_ORDER:
                go.set_molecule_ppmv(m, col["ppmv"][m])
            go.set_cfc_ppmv(0, col["cfc_ppmv"][0])
            go.set_cfc_ppmv(1, col["cfc_ppmv"][1])
            go.set_cia_ppmv(0, col["ppmv"][syn.N2])
            go.set_cia_ppmv(1, col["ppmv"][syn.O2])
            tick(name + "_set_ppmv", t0)
            t0 = time.perf_counter()
            go.calculate_optical_depth(col["p"], col["t"], gas)
            if not args.free:
                api.device_synchronize(device)
            tick(name + "_optical_depth", t0)
            t0 = time.perf_counter()
            ray.rayleigh(col["p"])
            tot = api.add_optics([gas, ray])
            if not args.free:
                api.device_synchronize(device)
            tick(name + "_rayleigh_add_optics", t0)
            t0 = time.perf_counter()
            if name == "lw":
                up, dn = lw.fluxes(tot, col["t_surf"], col["t_layer"], col["t"], wl.emis, bufs)
            else:
                up, dn = sw.fluxes(tot, col["mu0"], 0.5, wl.albedo, wl.albedo, col["tsi"], wl.solar, bufs)
            tick(name + "_fluxes_with_download", t0)
            t0 = time.perf_counter()
            dw = grid.dw
            _ = [float(np.sum(0.5 * (r[:-1] + r[1:]) * dw)) for r in (up[0], up[-1], dn[0], dn[-1])]
            tot.destroy()
            tick(name + "_caller_integration_destroy", t0)

    def run(label):
        phases.clear()
        column(0)
        phases.clear()
        t0 = time.perf_counter()
        for c in range(args.cols):
            column(c)
        wall = time.perf_counter() - t0
        return {"label": label, "columns_per_s": args.cols / wall, "ms_per_column": 1e3 * wall / args.cols,
                "ms_per_column_by_phase": {k: round(1e3 * v / args.cols, 3) for k, v in phases.items()}}

    out = {"what": "one-column reference ABI on the G1 bands, fast = 3 (the default of a new object); "
                   + ("no synchronisation between the phases (the calls' own host times; the solver call waits for all)"
                      if args.free else
                      "phases timed with a device synchronisation after each, so their sum is a little above the "
                      "unsynchronised wall time"),
           "runs": [run("caller's flux arrays pageable")]}
    hip = C.CDLL("libamdhip64.so")
    for name in objs:
        for b in objs[name][4]:
            rc = hip.hipHostRegister(C.c_void_p(b.ctypes.data), C.c_size_t(b.nbytes), C.c_uint(0))
            assert rc == 0, rc
    out["runs"].append(run("caller's flux arrays registered (hipHostRegister)"))
    text = json.dumps(out, indent=1)
    if args.out:
        open(args.out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
