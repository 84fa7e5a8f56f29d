"""Where does the lean first pass (GRT_LEAN unset) differ from the general one (GRT_LEAN=0), and from the oracle?
PYTHONPATH=.:tests python scripts/lean_debug.py [case]      case: half | dense | sw
"""
import os, sys, tempfile
import numpy as np
from grtcode_amd import api, synthetic as syn
from scenario import Band
from oracle.bindings import Oracle

case = sys.argv[1] if len(sys.argv) > 1 else "half"
tmp = tempfile.mkdtemp()
if case == "half":
    band = Band(tmp, 2000.0, 2400.0, 0.5, 5000); col = syn.profile(9, 12)
elif case == "dense":
    band = Band(tmp, 900.0, 1100.0, 1.0, 20000); col = syn.profile(2, 13)
elif case == "sw":
    band = Band(tmp, 30000.0, 30600.0, 1.0, 20000, sw=True, with_cfc=False); col = syn.profile(3, 13)
elif case == "det4":       # tests/test_gpu_deterministic.py, case 4
    band = Band(tmp, 500.0, 900.0, 1.0, 8000); col = syn.profile(3, 21)
elif case == "farir":
    band = Band(tmp, 40.0, 400.0, 1.0, 9000); col = syn.profile(0, 25)
elif case.startswith("seed"):      # the cases of tests/test_gpu_moment_kernel.py::test_randomised_grids_profiles_and_launch_shapes
    rng = np.random.default_rng(4242 + int(case[4:]))
    dw = float(rng.choice([0.1, 0.2, 0.25, 0.5, 1.0, 1.25, 1.5]))
    npts = int(rng.integers(150, 900))
    w0 = float(np.round(rng.choice([1.0, 300.0, 2000.0, 9000.0, 30000.0]) + rng.uniform(0, 50), 2))
    span = npts*dw
    if w0 + span > 50000.0:
        w0 = 50000.0 - span
    nlines = int(rng.integers(50, 6000))
    V = int(rng.integers(4, 15))
    band = Band(tmp, w0, w0 + span, dw, nlines, seed=int(rng.integers(1, 10**6)), sw=w0 > 3000.0, with_cfc=w0 < 3000.0)
    col = syn.profile(int(rng.integers(0, 50)), V)
    col["p"] = col["p"]*float(rng.choice([0.3, 1.0, 1.0, 2.5]))
    col["t"] = np.clip(col["t"] + float(rng.uniform(-40, 30)), 150.0, 340.0)
    print(f"{case}: dw {dw} w0 {w0} n {band.nw} lines {nlines} V {V} p_surf {col['p'][-1]:.1f}")
else:
    band = Band(tmp, 300.0, 700.0, 1.0, 6000); col = syn.profile(5, 13)
V = col["p"].size
lib = api.load_library(); device = api.create_device(0); orc = Oracle()
if os.environ.get("DBG_DET"):
    api.check(lib.grt_set_deterministic(1))
want = band.oracle_tau(orc, orc, lib, col)
out = {}
for name, env in (("lean", "1"), ("general", "0")):
    os.environ["GRT_LEAN"] = env
    go, grid = band.gas_optics(device, V, from_file=False)
    go.tune(fast=3, tile=int(os.environ.get("DBG_TILE", "0"))); band.set_column(go, col)
    opt = api.OpticsObject(V - 1, grid, device)
    go.calculate_optical_depth(col["p"], col["t"], opt)
    out[name] = opt.read()[0]; print(name, go.last_launch()); opt.destroy(); go.destroy()
dw = band.dw
mx = want.max(axis=1, keepdims=True)
for name in out:
    err = np.abs(out[name] - want)/mx
    print(f"{name}: worst vs oracle {err.max():.2e}; per layer:", " ".join(f"{e:.1e}" for e in err.max(axis=1)))
diff = (out["lean"] - out["general"])/mx
print("lean - general, per layer worst:", " ".join(f"{e:.1e}" for e in np.abs(diff).max(axis=1)))
for L in np.argsort(-np.abs(diff).max(axis=1))[:3]:
    f = int(np.abs(diff[L]).argmax())
    wn = band.w0 + f*dw
    print(f"layer {L} (p {0.5*(col['p'][L]+col['p'][L+1]):.4f} mb, T {0.5*(col['t'][L]+col['t'][L+1]):.1f}): worst {diff[L, f]:+.2e} at point {f} ({wn:.3f}); tau {want[L, f]:.4e}, layer max {want[L].max():.4e} at point {want[L].argmax()}")
    lo, hi = max(0, f - 8), min(band.nw, f + 9)
    print("    (lean-general)/max around it:", " ".join(f"{v:+.1e}" for v in diff[L, lo:hi]))
    print("    (lean-general)/tau around it:", " ".join(f"{v:+.1e}" for v in ((out['lean'] - out['general'])/want)[L, lo:hi]))
    cand = []
    for m in band.mols:
        v = band.lines[m]["v0"]
        for i in np.argsort(np.abs(v - wn))[:2]:
            cand.append((abs(v[i] - wn), m, v[i], band.lines[m]["s0"][i], band.lines[m]["yair"][i], band.lines[m]["delta"][i], band.lines[m]["en"][i]))
    for d, m, v, s, g, dl, en in sorted(cand)[:4]:
        print(f"    line of molecule {m} at {v:.6f} ({(v - wn)/dw:+.4f} steps), S {s:.3e}, g_air {g:.3f}, delta {dl:+.4f}, E {en:.1f}")
# how many points differ by more than 1e-6 of the layer max, and where relative to tile edges (tile 256 / 64)
big = np.argwhere(np.abs(diff) > 1e-6)
print("points beyond 1e-6:", len(big), "first few (layer, point):", big[:12].tolist())
