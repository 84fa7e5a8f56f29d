"""Parity of the PRODUCTION form (fast = 3: fused arithmetic, far wings by cell moments, two passes / cell hierarchy)
at the sizes the bench runs, and of the integer windows through the production kernels (VERDICT r1 "Next round" 1-2).

  * the bench's own workload (SURVEY §8d grid G1: 1.0 M + 1.5 M lines, 60 layers, LW 1-3250 + SW 1-50000 cm-1 at
    1 cm-1), one column, fast = 3 and fast = 0, against the reference's own C (oracle/_ref, OpenMP) -- on the SURVEY
    line list and on the physically scaled one (synthetic.PHYSICAL_BANDS: OLR ~270 W m-2, surface SW ~0.68 of TOA), because
    the SURVEY list makes a nearly black atmosphere whose fluxes barely depend on tau;
  * three tau metrics (of the layer maximum -- the round-1 one --, pointwise relative, transmission) and spectral fluxes
    point by point (a driver without -integrated writes spectra: driver.c:285-356), worst cases recorded in
    gpurun_out/parity_full_g1.json;
  * integer windows (kernels.c:431-437) BIT-EXACT through fast = 1, 2, 3 and the tree form: the set of grid points an
    isolated line reaches must be the oracle's [s, e], including clipping at 0 and n-1, centres pushed off the grid by
    the pressure shift, and centres halfway between two grid points.
"""
import json
import os

import numpy as np
import pytest

from grtcode_amd import api, synthetic as syn, workload as W
from oracle import reference_column as RC
from scenario import Band, MOLTAB, mol_mass

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# stated bounds; the measured worst cases are written to gpurun_out/parity_full_g1.json
BOUNDS = {
    0: dict(of_layer_max=1e-11, pointwise_rel=1e-9, transmission=1e-12, flux=1e-6, spectral_flux_rel=1e-10),
    # fast = 3, pointwise / transmission: the one deliberate departure -- Humlicek region 1 is carried by the cell moments
    # beyond a line's XLIM0, where the reference switches back to the Lorentzian with a jump of 1.5/XLIM0^2 = 1e-4 of the
    # line's value there; on the SURVEY list (strengths so large that even far wings of the thinnest layers reach tau ~ 1)
    # that shows as <= 1e-4 pointwise and <= 1e-4/e in transmission; on the physically scaled list both are ~1e-6 and 2e-7
    3: dict(of_layer_max=2e-6, pointwise_rel=1.2e-4, transmission=4e-5, flux=1e-3, spectral_flux_rel=1e-5),
}


def record(name, payload):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "parity_full_g1.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[name] = payload
    json.dump(data, open(path, "w"), indent=1)


# GRT_PARITY_PROFILES="0,3,17" in the environment: which synthetic atmospheres (synthetic.profile) -- soak runs
PROFILES = [int(x) for x in os.environ.get("GRT_PARITY_PROFILES", "0").split(",")]


@pytest.mark.parametrize("prof", PROFILES)
@pytest.mark.parametrize("physical", [False, True], ids=["survey_list", "physical_list"])
def test_full_g1_column_against_reference(lib, device, physical, prof):
    """One column of the bench workload at FULL size, LW + SW, production form and reference-order form."""
    kind, chk, orc = RC.checker(omp=True)
    if kind != "reference":
        pytest.skip("needs the prebuilt reference library oracle/_ref/libgrtref_omp.so: the scalar restatement would take "
                    "minutes for 7.7e9 line-shape evaluations (bench.py makes the same comparison for its column 0)")
    RC.set_omp_threads(min(os.cpu_count() or 1, 16))
    wl = W.G1Workload(device, 1, physical=physical, spectral=True)
    col = syn.profile(prof, W.NUM_LEVELS)
    ref = {}
    for band, grid, lines, sw in (("lw", W.LW_GRID, wl.lw_lines, False), ("sw", W.SW_GRID, wl.sw_lines, True)):
        ref[band] = RC.band_column(kind, chk, orc, lib.Q, col, grid, lines, syn.tables(sw=sw), W.MOL_ORDER, MOLTAB,
                                   mol_mass, W.CIA_PAIRS, sw)
    # one reference defect is not reproduced: when a range check inside sw_flux trips on a rounding-level
    # excursion (shortwave.c:318-320) sw_fluxes_kernel stores the PREVIOUS wavenumber's fluxes (:443 ignores the code).
    # Such points -- every level equal to the left neighbour's -- are left out of the spectral comparison below, and in
    # the reference's own integrals they are given the value under test (a handful of 50 000 points; without that the
    # one-point defect shows as ~1e-6 W m-2 in the reference-order comparison)
    r = ref["sw"]
    stale = np.zeros(r["nw"], dtype=bool)
    stale[1:] = np.all(r["up"][:, 1:] == r["up"][:, :-1], axis=0) & np.all(r["dn"][:, 1:] == r["dn"][:, :-1], axis=0)
    assert stale.mean() < 0.01
    want = np.concatenate([ref["lw"]["integ"], ref["sw"]["integ"]])
    if physical and prof == 0:
        assert 250.0 < want[0] < 300.0                     # outgoing longwave, W m-2
        assert 0.62 < want[10] / want[9] < 0.75            # shortwave reaching the surface / incoming
    (gcols, keep), _ = wl.columns(prof, 1)
    L, V = W.NUM_LEVELS - 1, W.NUM_LEVELS
    report = {"checker": kind, "profile": prof, "reference_fluxes_w_m2": want.tolist(), "reference_stale_sw_points": int(stale.sum())}
    for fast in (3, 0):
        wl.go_lw.tune(fast=fast)
        wl.go_sw.tune(fast=fast)
        wl.pipe.run(gcols)
        got = wl.pipe.fluxes(1)[0]
        assert wl.go_lw.last_launch()["fast"] == fast and wl.go_sw.last_launch()["fast"] == fast
        b = BOUNDS[fast]
        rep = {}
        for bi, band in enumerate(("lw", "sw")):
            nw = ref[band]["nw"]
            v = wl.pipe.views(bi)
            tau = api.device_to_host(device, v["tau_gas"], (L, nw))
            m = RC.tau_metrics(tau, ref[band]["tau_gas"])
            up = api.device_to_host(device, v["flux_up"], (V, nw))
            dn = api.device_to_host(device, v["flux_down"], (V, nw))
            fs = max(np.abs(ref[band]["up"]).max(), np.abs(ref[band]["dn"]).max())
            ok = ~stale if band == "sw" else np.ones(nw, dtype=bool)
            if band == "sw":
                m["reference_stale_points"] = int(stale.sum())
            m["spectral_flux_abs_w_m2_per_cm"] = float(max(np.abs(up - ref[band]["up"])[:, ok].max(), np.abs(dn - ref[band]["dn"])[:, ok].max()))
            m["spectral_flux_rel"] = m["spectral_flux_abs_w_m2_per_cm"] / fs
            if band == "sw" and stale.any():
                rows = [np.where(stale, mine, theirs) for mine, theirs in ((up[0], r["up"][0]), (up[-1], r["up"][-1]), (dn[0], r["dn"][0]), (dn[-1], r["dn"][-1]))]
                want = want.copy()
                want[[6, 7, 9, 10]] = [orc.integrate_row(np.ascontiguousarray(x), W.SW_GRID[2]) for x in rows]
            rep[band] = m
            for key in ("of_layer_max", "pointwise_rel", "transmission", "spectral_flux_rel"):
                assert m[key] <= b[key], (physical, fast, band, key, m[key], b[key])
        rep["max_abs_flux_diff_w_m2"] = float(np.abs(got - want).max())
        assert rep["max_abs_flux_diff_w_m2"] <= b["flux"], (physical, fast, rep)
        report[f"fast{fast}"] = rep
        print(f"full G1 column, {'physical' if physical else 'survey'} list, fast={fast}: {json.dumps(rep)}")
    record(("physical_list" if physical else "survey_list") + ("" if prof == 0 else f"_profile{prof}"), report)
    wl.destroy()


# ---- integer windows through the production kernels -------------------------------------------------------------- #
def isolated_lines_band(root, w0, wn, dw, centres, deltas, iso=1):
    """A band whose only absorber is CO2 with a handful of well separated strong lines, no continua / CFC / CIA."""
    band = Band(root, w0, wn, dw, len(centres), mols=[syn.CO2], with_ctm=False, with_cfc=False, with_cia=False, iso_mix=False)
    n = len(centres)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    band.lines[syn.CO2] = dict(v0=np.asarray(centres, dtype=np.float64), s0=np.full(n, 3.0e-20), yair=f32(np.full(n, 0.07)),
                               yself=f32(np.full(n, 0.09)), en=f32(np.full(n, 300.0)), nexp=f32(np.full(n, 0.75)),
                               delta=f32(deltas), iso=np.full(n, iso, dtype=np.int32))
    return band


def window_case(dw):
    """Centres exercising kernels.c:431-437 on a grid from 600 cm-1: clip at index 0, clip at n-1, x.5 (halfway between
    two points), just below / above halfway, pushed off either end of the grid by the pressure shift, and interior."""
    span = 230.0
    w0, wn = 600.0, 600.0 + span
    n = int(np.ceil((wn - w0) / dw)) + 1
    k = lambda idx: w0 + idx * dw
    centres = [w0 + 0.05 * dw,               # low edge: window clipped at 0; shift may push it below -0.5 -> dropped
               k(int(60.0 / dw)) + 0.5 * dw,            # exactly halfway: floor((2x+1)/2) rounds up
               k(int(115.0 / dw)) + 0.4999 * dw,      # just below halfway
               k(int(170.0 / dw)) + 0.5001 * dw,      # just above halfway
               wn - 0.05 * dw]               # high edge: clipped at n-1; shift may push it beyond n-1 -> dropped
    centres = [float("%.6f" % c) for c in centres]
    deltas = [-0.035, 0.013, 0.019, -0.007, 0.035]
    return w0, wn, n, centres, deltas


# (0.002 cm-1: 12 500-point windows, the wave-shared tree gather -- there for the tree form only)
@pytest.mark.parametrize("fast,dw", [(f, d) for f in (1, 2, 3) for d in (1.0, 0.5, 0.05, 0.01, 0.002) if d >= 0.01 or f == 3])
def test_windows_bit_exact_through_production_kernels(tmp_path, oracle, lib, device, dw, fast, monkeypatch):
    w0, wn, n, centres, deltas = window_case(dw)
    band = isolated_lines_band(str(tmp_path), w0, wn, dw, centres, deltas)
    assert band.nw == n
    V = 9
    col = syn.profile(3, V)
    col["p"] = col["p"] * 1.5                          # up to 1.5 atm: shifts of +-0.03 cm-1 move centres across grid points
    go, grid = band.gas_optics(device, V, from_file=False)
    go.tune(fast=fast)
    band.set_column(go, col)
    opt = api.OpticsObject(V - 1, grid, device)
    go.calculate_optical_depth(col["p"], col["t"], opt)
    tau = opt.read()[0]
    info = go.last_launch()
    # the oracle's windows, line by line and layer by layer
    kw = band.oracle_inputs(oracle, lib, col)
    m = kw["mols"][0]
    p_atm = col["p"] * np.float64(np.float32(0.000986923))
    nn, pavg, tavg = oracle.layer_means(p_atm, col["t"])
    ps, ns = oracle.species_means(p_atm, m["x"], nn)
    vnn, snn, gamma, alpha = oracle.line_prep(m["lines"], m["mass"], m["num_iso"], pavg, tavg, ps, m["q"])
    want_tau, ws, we = oracle.line_sample(vnn, snn, gamma, alpha, ns, w0, dw, n, windows=True)
    mask = np.zeros((V - 1, n), dtype=bool)
    dropped = 0
    for i in range(V - 1):
        for j in range(len(centres)):
            if ws[i, j] <= we[i, j]:
                mask[i, ws[i, j]: we[i, j] + 1] = True
            else:
                dropped += 1
    assert np.array_equal(want_tau != 0.0, mask)       # the oracle's own tau is non-zero exactly on its windows
    assert dropped > 0 or dw > 0.05                     # fine grids: the shift pushes edge lines off the grid in some layer
    got_mask = tau != 0.0
    bad = np.argwhere(got_mask != mask)
    assert bad.size == 0, (f"fast={fast} dw={dw} launch={info}: {bad.shape[0]} points differ from the reference's windows, "
                           f"first (layer, index) {bad[:5].tolist()}")
    scale = want_tau.max(axis=1, keepdims=True)
    assert np.max(np.abs(tau - want_tau) / scale) < 2e-6
    if fast == 3 and dw <= 0.05:
        assert info["tree_levels"] > 0, info            # the cell hierarchy ran
    opt.destroy()
    go.destroy()
