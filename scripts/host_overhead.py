import time, tempfile
import numpy as np
from grtcode_amd import api, synthetic as syn, workload as W
device = api.create_device(0)
root = tempfile.mkdtemp(prefix="grt_ho_")
files, _ = W.write_tables(root, sw=False)
spec = (1000.0, 1040.0, 1.0)
lines = W.band_lines(7000, spec, 1)
V = W.NUM_LEVELS
go, grid = W.build_band(device, spec, lines, files, V)
col = syn.profile(0, V)
for m in W.MOL_ORDER:
    go.set_molecule_ppmv(m, col["ppmv"][m])
go.set_cfc_ppmv(0, col["cfc_ppmv"][0]); go.set_cfc_ppmv(1, col["cfc_ppmv"][1])
go.set_cia_ppmv(0, col["ppmv"][syn.N2]); go.set_cia_ppmv(1, col["ppmv"][syn.O2])
opt = api.OpticsObject(V - 1, grid, device)
go.tune(fast=3)
for _ in range(5):
    go.calculate_optical_depth(col["p"], col["t"], opt)
api.profile_enable(True)
t0 = time.perf_counter(); n = 200
for _ in range(n):
    go.calculate_optical_depth(col["p"], col["t"], opt)
dt = (time.perf_counter() - t0)/n
k = sum(api.profile_read(t)[0] for t in (1, 2, 6, 7))/n
print(f"tiny band: {dt*1e3:.3f} ms per call, kernels {k:.3f} ms")
t0 = time.perf_counter()
for _ in range(n):
    for m in W.MOL_ORDER:
        go.set_molecule_ppmv(m, col["ppmv"][m])
print(f"set_ppmv x7: {(time.perf_counter()-t0)/n*1e3:.3f} ms")
