"""Full-size G1 column: lean first pass against the general one (GRT_LEAN=0), band by band, worst points per layer.
PYTHONPATH=.:tests python scripts/lean_debug_g1.py [physical]"""
import os, sys
import numpy as np
from grtcode_amd import api, synthetic as syn, workload as W

physical = len(sys.argv) > 1 and sys.argv[1] == "physical"
lib = api.load_library(); device = api.create_device(0)
wl = W.G1Workload(device, 1, physical=physical, spectral=True, fast=3)
(gcols, keep), _ = wl.columns(0, 1)
L = W.NUM_LEVELS - 1
tau = {}
for name, env in (("lean", "1"), ("general", "0")):
    os.environ["GRT_LEAN"] = env
    wl.pipe.run(gcols)
    wl.pipe.sync()
    tau[name] = [api.device_to_host(device, wl.pipe.views(bi)["tau_gas"], (L, n)).copy() for bi, n in ((0, wl.grid_lw.n), (1, wl.grid_sw.n))]
    print(name, wl.go_lw.last_launch(), wl.go_sw.last_launch())
col = syn.profile(0, W.NUM_LEVELS)
for bi, (band, grid, lines) in enumerate((("lw", W.LW_GRID, wl.lw_lines), ("sw", W.SW_GRID, wl.sw_lines))):
    a, b = tau["lean"][bi], tau["general"][bi]
    mx = b.max(axis=1, keepdims=True)
    d = (a - b)/mx
    per = np.abs(d).max(axis=1)
    print(band, "worst", per.max(), "per layer:", " ".join(f"{e:.1e}" for e in per))
    for Lw in np.argsort(-per)[:3]:
        f = int(np.abs(d[Lw]).argmax()); wn = grid[0] + f*grid[2]
        print(f"  layer {Lw} (p {0.5*(col['p'][Lw]+col['p'][Lw+1]):.4f} mb, T {0.5*(col['t'][Lw]+col['t'][Lw+1]):.1f}): {d[Lw, f]:+.2e} at point {f} ({wn:.2f}); tau {b[Lw, f]:.4e} max {mx[Lw, 0]:.4e} at {int(b[Lw].argmax())}")
        lo, hi = max(0, f - 6), min(a.shape[1], f + 7)
        print("     (lean-general)/tau:", " ".join(f"{v:+.1e}" for v in ((a - b)/b)[Lw, lo:hi]))
        cand = []
        for m in lines:
            v = lines[m]["v0"]
            i0 = np.searchsorted(v, wn)
            for i in range(max(0, i0 - 2), min(v.size, i0 + 2)):
                cand.append((abs(v[i] - wn), m, v[i], lines[m]["s0"][i], lines[m]["yair"][i], lines[m]["delta"][i], lines[m]["en"][i], lines[m]["nexp"][i]))
        for dd, m, v, s0, g, dl, en, nn in sorted(cand)[:4]:
            print(f"     line mol {m} at {v:.6f} ({(v - wn)/grid[2]:+.4f} steps) S {s0:.3e} g_air {g:.3f} delta {dl:+.4f} E {en:.1f} n {nn:.2f}")
    big = np.argwhere(np.abs(d) > 1e-6)
    print("  points beyond 1e-6:", len(big), big[:10].tolist())
