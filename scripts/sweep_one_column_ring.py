"""Tile / line-slice sweep of the ring kernels (fast = 0: reference operation order; fast = 2: fused) for ONE column per
launch -- what an unchanged one-column caller runs with GRT_GAS_OPTICS_FAST=0.

    PYTHONPATH=. python scripts/sweep_one_column_ring.py [fast]"""
import sys

from grtcode_amd import api, synthetic as syn, workload as W

fast = int(sys.argv[1]) if len(sys.argv) > 1 else 0
device = api.create_device(0)
wl = W.G1Workload(device, 1)
V = W.NUM_LEVELS
col = syn.profile(0, V)
api.profile_enable(True)
for name, go, grid, tag in (("lw", wl.go_lw, wl.grid_lw, 1), ("sw", wl.go_sw, wl.grid_sw, 2)):
    for m in W.MOL_ORDER:
        go.set_molecule_ppmv(m, col["ppmv"][m])
    go.set_cfc_ppmv(0, col["cfc_ppmv"][0]); go.set_cfc_ppmv(1, col["cfc_ppmv"][1])
    go.set_cia_ppmv(0, col["ppmv"][syn.N2]); go.set_cia_ppmv(1, col["ppmv"][syn.O2])
    opt = api.OpticsObject(V - 1, grid, device)
    for tile, ns in [(0, 0), (0, 0), (256, 1), (256, 2), (256, 4), (512, 1), (512, 2), (512, 4), (512, 8), (1024, 1), (1024, 2), (1024, 4), (1024, 8), (1024, 16)]:
        try:
            go.tune(tile=tile, nslice=ns, fast=fast)
            go.calculate_optical_depth(col["p"], col["t"], opt)
            api.profile_read(tag, reset=True)
            for _ in range(2):
                go.calculate_optical_depth(col["p"], col["t"], opt)
            ms, n = api.profile_read(tag)
            info = go.last_launch()
            print(name, tile, ns, round(ms / max(n, 1), 3), (info["tile"], info["nslice"], info["fast"]))
        except Exception as e:
            print(name, tile, ns, "failed:", str(e)[:80])
    opt.destroy()
