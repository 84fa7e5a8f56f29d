"""Exploration: wall time of one column of the G1 longwave band through each optical_depth_method."""
import sys, time, tempfile
import numpy as np
from grtcode_amd import api, synthetic as syn, workload as W

device = api.create_device(0)
V = 61
col = syn.profile(0, V)
root = tempfile.mkdtemp(prefix="grt_sweeps_")
files, _ = W.write_tables(root, sw=False)
lines = W.band_lines(int(sys.argv[1]) if len(sys.argv) > 1 else W.LW_LINES, W.LW_GRID, 20261003)
for method, name in ((2, "line_sample (strict)"), (0, "wavenumber_sweep"), (1, "line_sweep")):
    go, grid = W.build_band(device, W.LW_GRID, lines, files, V, method=method)
    for m, x in col["ppmv"].items():
        if m in lines:
            go.set_molecule_ppmv(m, x)
    opt = api.OpticsObject(V - 1, grid, device)
    go.calculate_optical_depth(col["p"], col["t"], opt)     # builds stores
    t0 = time.perf_counter()
    for _ in range(3):
        go.calculate_optical_depth(col["p"], col["t"], opt)
    dt = (time.perf_counter() - t0) / 3
    tau = opt.read()[0]
    print(f"{name}: {dt*1e3:.1f} ms per column, sum tau {tau.sum():.6e}")
    opt.destroy(); go.destroy()
