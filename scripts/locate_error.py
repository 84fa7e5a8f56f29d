"""Where does the fused form differ most from the oracle (and the moment kernel from the ring kernel) in one of the
randomised cases of tests/test_gpu_moment_kernel.py?   PYTHONPATH=.:tests python scripts/locate_error.py SEED"""
import os, sys, tempfile
import numpy as np
from grtcode_amd import api, synthetic as syn
from scenario import Band
from oracle.bindings import Oracle

seed = int(sys.argv[1])
tmp = tempfile.mkdtemp()
if len(sys.argv) > 2 and sys.argv[2] == "tree":        # the cases of tests/test_gpu_moment_tree.py::test_randomised_fine_grids
    rng = np.random.default_rng(777 + seed)
    dw = float(rng.choice([0.04, 0.02, 0.01, 0.005, 0.0025]))
    npts = int(rng.integers(1500, 9000))
    w0 = float(np.round(rng.choice([50.0, 700.0, 2300.0, 9000.0, 20000.0]) + rng.uniform(0, 50), 2))
    span = npts * dw
    V = int(rng.integers(4, 9))
    nlines = int(rng.integers(40, 1.2e8 / ((V - 1) * 2 * 25 / dw)))
    band = Band(tmp, w0, w0 + npts * dw, dw, nlines, seed=int(rng.integers(1, 10**6)), sw=w0 > 3000.0, with_cfc=w0 < 3000.0)
    col = syn.profile(int(rng.integers(0, 50)), V)
else:
    rng = np.random.default_rng(4242 + seed)
    dw = float(rng.choice([0.1, 0.2, 0.25, 0.5, 1.0, 1.25, 1.5]))
    wide = os.environ.get("GRT_STRESS_WIDE", "")     # soak runs: "1" band anywhere, "2" also grids, pressures, levels
    if wide == "2":
        dw = float(rng.choice([0.05, 0.1, 0.25, 0.5, 1.0, 2.0, 5.0]))
    npts = int(rng.integers(150, 900))
    w0 = float(np.round(rng.choice([1.0, 300.0, 2000.0, 9000.0, 30000.0]) + rng.uniform(0, 50), 2))
    if wide:
        w0 = float(np.round(10.0 ** rng.uniform(0.0, 4.69), 2))
    span = npts * dw
    if w0 + span > 50000.0:
        w0 = 50000.0 - span
    nlines = int(rng.integers(50, 6000))
    V = int(rng.integers(4, 15)) if wide != "2" else int(rng.integers(4, 40))
    band = Band(tmp, w0, w0 + span, dw, nlines, seed=int(rng.integers(1, 10**6)), sw=w0 > 3000.0, with_cfc=w0 < 3000.0)
    col = syn.profile(int(rng.integers(0, 50)), V)
col["p"] = col["p"] * (float(rng.choice([0.3, 1.0, 1.0, 2.5])) if os.environ.get("GRT_STRESS_WIDE", "") != "2" else float(10.0 ** rng.uniform(-1.5, 0.7)))
col["t"] = np.clip(col["t"] + float(rng.uniform(-40, 30)), 150.0, 340.0)
lib = api.load_library(); device = api.create_device(0); orc = Oracle()
want = band.oracle_tau(orc, orc, lib, col)
out = {}
for fast in (2, 3):
    go, grid = band.gas_optics(device, V, from_file=False)
    go.tune(fast=fast); band.set_column(go, col)
    opt = api.OpticsObject(V - 1, grid, device)
    go.calculate_optical_depth(col["p"], col["t"], opt)
    out[fast] = opt.read()[0]; opt.destroy(); go.destroy()
print(f"seed {seed}: dw {dw} w0 {w0} n {band.nw} lines {nlines} V {V} p_surf {col['p'][-1]:.1f} T {col['t'][0]:.0f}-{col['t'][-1]:.0f}")
for fast, tau in out.items():
    err = np.abs(tau - want) / want.max(axis=1, keepdims=True)
    L, f = np.unravel_index(err.argmax(), err.shape)
    wn = band.w0 + f * dw
    print(f"fast {fast}: worst {err.max():.2e} at layer {L} (p {0.5*(col['p'][L]+col['p'][L+1]):.3f} mb, T {0.5*(col['t'][L]+col['t'][L+1]):.1f}), point {f} ({wn:.4f} cm-1); tau there {want[L, f]:.4e}, layer max {want[L].max():.4e} at {band.w0 + want[L].argmax()*dw:.4f}")
    # nearest lines
    best = []
    for m in band.mols:
        v = band.lines[m]["v0"]; k = np.argsort(np.abs(v - wn))[:2]
        best += [(abs(v[i] - wn), m, v[i], band.lines[m]["s0"][i], band.lines[m]["yair"][i]) for i in k]
    for d, m, v, s, g in sorted(best)[:3]:
        print(f"    line of molecule {m} at {v:.6f} (distance {d:.6f} cm-1 = {d/dw:.3f} steps), S {s:.3e}, g_air {g:.3f}")

diff = np.abs(out[3] - out[2]) / want.max(axis=1, keepdims=True)
L, f = np.unravel_index(diff.argmax(), diff.shape)
wn = band.w0 + f * dw
print(f"moment vs ring: worst {diff.max():.2e} at layer {L} (p {0.5*(col['p'][L]+col['p'][L+1]):.3f} mb), point {f} ({wn:.4f}); tau {want[L, f]:.4e}, "
      f"layer max {want[L].max():.4e}; signed (mp-ring)/max {(out[3][L, f]-out[2][L, f])/want[L].max():+.2e}, (mp-oracle)/max {(out[3][L, f]-want[L, f])/want[L].max():+.2e}")
row = (out[3][L] - out[2][L]) / want[L].max()
lo, hi = max(0, f - 8), min(band.nw, f + 9)
print("    (mp-ring)/max around it:", " ".join(f"{v:+.1e}" for v in row[lo:hi]))
for m in band.mols:
    v = band.lines[m]["v0"]; k = np.argsort(np.abs(v - wn))[:1]
    for i in k:
        print(f"    nearest line of molecule {m}: {v[i]:.6f} ({(v[i]-wn)/dw:+.3f} steps), S {band.lines[m]['s0'][i]:.3e}, g_air {band.lines[m]['yair'][i]:.3f} g_self {band.lines[m]['yself'][i]:.3f}")
