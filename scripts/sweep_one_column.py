"""Tile / line-slice sweep of the two-pass line kernel for ONE column per launch (the reference-shaped
calculate_optical_depth): kernel milliseconds from the library's HIP-event brackets.

    PYTHONPATH=. python scripts/sweep_one_column.py"""
import json

from grtcode_amd import api, synthetic as syn, workload as W

device = api.create_device(0)
wl = W.G1Workload(device, 1)
V = W.NUM_LEVELS
col = syn.profile(0, V)
api.profile_enable(True)
out = {}
for name, go, grid, tag, cases in (("lw", wl.go_lw, wl.grid_lw, 1, [(0, 0), (64, 1), (64, 2), (64, 4), (64, 8), (128, 2), (128, 4), (128, 8), (256, 4), (256, 8), (256, 16)]),
                                   ("sw", wl.go_sw, wl.grid_sw, 2, [(0, 0), (64, 1), (128, 1), (128, 2), (256, 1), (256, 2), (256, 4), (512, 2)])):
    for m in W.MOL_ORDER:
        go.set_molecule_ppmv(m, col["ppmv"][m])
    go.set_cfc_ppmv(0, col["cfc_ppmv"][0]); go.set_cfc_ppmv(1, col["cfc_ppmv"][1])
    go.set_cia_ppmv(0, col["ppmv"][syn.N2]); go.set_cia_ppmv(1, col["ppmv"][syn.O2])
    opt = api.OpticsObject(V - 1, grid, device)
    for tile, ns in cases:
        go.tune(tile=tile, nslice=ns, fast=3)
        go.calculate_optical_depth(col["p"], col["t"], opt)
        api.profile_read(tag, reset=True)
        for _ in range(5):
            go.calculate_optical_depth(col["p"], col["t"], opt)
        first, far = api.profile_read(tag)[0] / 5, api.profile_read(tag + 5)[0] / 5
        info = go.last_launch()
        out[f"{name} tile={tile} nslice={ns}"] = {"first_pass_ms": round(first, 3), "gather_ms": round(far, 3), "ran": (info["tile"], info["nslice"])}
        print(name, tile, ns, round(first, 3), round(far, 3), (info["tile"], info["nslice"]), flush=True)
print(json.dumps(out))
