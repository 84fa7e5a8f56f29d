"""The tree form of the two-pass moment kernel (fast=3 on windows of more than 200 points a side: far field through
a hierarchy of cells, moment_up_kernel + gas_optics_tree_kernel in k_gas_optics_mp.hip) against the oracle and
against the ring kernel (fast=2, every window point evaluated).  tests/test_moment_tree.py holds the construction
in numpy; here every case states which part of the kernels it is there for."""
import os

import numpy as np
import pytest

from grtcode_amd import api, synthetic as syn
from scenario import Band
from test_gpu_gas_optics import tau_close

pytestmark = pytest.mark.gpu
FAST_TOL = 2e-6
# moment kernel against ring kernel: the moments carry Humlicek region 1 on beyond a line's XLIM0, where the reference
# (and the ring kernel) switch back to the Lorentzian -- 1e-4 of the value there, at most 1e-6 of a layer's largest tau
# (kFoldWrMax in k_gas_optics_mp.hip); everything else they share to ~2e-7
BETWEEN_TOL = 1.2e-6


@pytest.fixture(params=["lane", "wave"], autouse=True)
def gather_form(request, monkeypatch):
    """Both forms of the tree gather on every case: each lane walking its own cells (the library's choice for windows
    of fewer than 8 192 points a side) and the wave-shared scalar walk (its choice beyond; forced here)."""
    monkeypatch.setenv("GRT_TREE_WAVE_MIN", "1000000000" if request.param == "lane" else "1")
    return request.param


def run(band, device, col, fast, tile=0):
    V = col["p"].size
    go, grid = band.gas_optics(device, V, from_file=False)
    go.tune(tile=tile, nslice=0, fast=fast)
    band.set_column(go, col)
    opt = api.OpticsObject(V - 1, grid, device)
    go.calculate_optical_depth(col["p"], col["t"], opt)
    tau = opt.read()[0]
    info = go.last_launch()
    opt.destroy()
    go.destroy()
    return tau, info


def check(band, device, oracle, lib, col, tree=True, tile=0):
    want = band.oracle_tau(oracle, oracle, lib, col)
    got, info = run(band, device, col, 3, tile=tile)
    ring, _ = run(band, device, col, 2)
    e_tree, e_ring, e_between = tau_close(got, want), tau_close(ring, want), tau_close(got, ring)
    print(f"{info}: tree vs oracle {e_tree:.2e}; ring vs oracle {e_ring:.2e}; tree vs ring {e_between:.2e}")
    if tree is not None:
        assert (info["fast"] == 3 and info["tree_levels"] > 0) == tree, info
    assert e_tree < FAST_TOL
    assert e_between < BETWEEN_TOL
    return info


def test_longwave_at_0p02(tmp_path, oracle, lib, device):
    """Windows of 1 250 points a side, eight coarse levels; near fields of a few to ~50 points."""
    band = Band(str(tmp_path), 1000.0, 1060.0, 0.02, 2500)
    info = check(band, device, oracle, lib, syn.profile(2, 9))
    assert info["tree_levels"] == 8


@pytest.mark.parametrize("tile", [0, 64, 256])
def test_0p005_beyond_the_single_level_kernels(tmp_path, oracle, lib, device, tile):
    """Windows of 5 000 points a side (the single-level gather stops at 4 096); Lorentz widths of up to ~20 grid
    steps in the lowest layers, so the fine levels are skipped there and the near field is ~150 points."""
    band = Band(str(tmp_path), 2000.0, 2030.0, 0.005, 1500)
    info = check(band, device, oracle, lib, syn.profile(4, 7), tile=tile)
    assert info["tree_levels"] == 10 and info["halo"] > 50
    # four cells per line: the automatic tile is 1 024 cells, moments straight to global memory, twelve of them
    assert (info["moments"], info["tile"]) == ((12, 1024) if tile == 0 else (8, tile)), info


def test_high_pressure_wide_lorentz_lines(tmp_path, oracle, lib, device):
    """Three atmospheres at 0.01 cm-1: eta up to ~30 grid steps."""
    band = Band(str(tmp_path), 700.0, 740.0, 0.01, 1200)
    col = syn.profile(7, 8)
    col["p"] = col["p"] * 3.0
    check(band, device, oracle, lib, col)


def test_shortwave_region1_reach_of_hundreds_of_points(tmp_path, oracle, lib, device):
    """30 000 cm-1 at 0.02: Doppler widths of two grid steps, Humlicek region 1 reaches ~300 points -- all of it
    inside the ring (the near-field halo is sized for it)."""
    band = Band(str(tmp_path), 30000.0, 30050.0, 0.02, 900, sw=True, with_cfc=False)
    info = check(band, device, oracle, lib, syn.profile(8, 7))
    assert info["halo"] > 250


def test_grid_shorter_than_a_window(tmp_path, oracle, lib, device):
    """2 001 points at 0.005 cm-1: every window is clipped at both ends of the grid (kernels.c:435-437)."""
    band = Band(str(tmp_path), 500.0, 510.0, 0.005, 800, mols=[syn.H2O, syn.CO2], with_cfc=False, with_cia=False)
    check(band, device, oracle, lib, syn.profile(5, 6))


def test_lines_hugging_the_grid_edges_and_beyond(tmp_path, oracle, lib, device):
    """Centres within a few points of index 0 and n-1, and lines outside the grid whose windows reach in."""
    band = Band(str(tmp_path), 1200.0, 1260.0, 0.02, 1500, mols=[syn.H2O, syn.CO2], with_cfc=False, with_cia=False,
                with_ctm=False, line_range=(1190.0, 1270.0))
    for m in band.lines:
        v = band.lines[m]["v0"]
        k = v.size // 4
        v[:k] = np.round(1200.0 + (v[:k] - 1190.0) * 0.002, 6)
        v[-k:] = np.round(1260.0 - (1270.0 - v[-k:]) * 0.002, 6)
        band.lines[m]["v0"] = np.sort(v)
    check(band, device, oracle, lib, syn.profile(5, 9))


def test_no_lines_at_all_leaves_the_continua(tmp_path, oracle, lib, device):
    """An empty line list on a fine grid: zeroed moments, every cell series zero, tau = continua + CFC + CIA."""
    band = Band(str(tmp_path), 900.0, 930.0, 0.02, 0)
    check(band, device, oracle, lib, syn.profile(6, 6))


def test_near_field_wider_than_the_window_falls_back(tmp_path, oracle, lib, device):
    """0.04 cm-1 (625 points a side) under 100 atm: the moment bound asks for a near field beyond the window even with
    twelve moments, so the tree form does not apply and the library falls back to the forms that treat the whole
    window as near."""
    band = Band(str(tmp_path), 500.0, 560.0, 0.04, 1500)
    col = syn.profile(3, 7)
    col["p"] = col["p"] * 100.0
    check(band, device, oracle, lib, col, tree=False)


def test_wide_lorentz_lines_take_twelve_moments_where_eight_do_not_fit(tmp_path, oracle, lib, device):
    """The same grid under 40 atm: with eight moments the near field (7.8 |z|max) is wider than the window, with twelve
    (3.95 |z|max) it fits -- the library prefers the sparse-line form (tiles of 1 024 cells, moments straight to global
    memory) to falling back, dense lines or not."""
    band = Band(str(tmp_path), 500.0, 560.0, 0.04, 1500)
    col = syn.profile(3, 7)
    col["p"] = col["p"] * 40.0
    info = check(band, device, oracle, lib, col)
    assert info["moments"] == 12 and info["tile"] == 1024, info


@pytest.mark.parametrize("seed", range(int(os.environ.get("GRT_STRESS_FIRST", 0)),
                                         int(os.environ.get("GRT_STRESS_FIRST", 0)) + int(os.environ.get("GRT_STRESS_SEEDS", 6))))   # soak runs
def test_randomised_fine_grids(tmp_path, oracle, lib, device, seed):
    rng = np.random.default_rng(777 + seed)
    dw = float(rng.choice([0.04, 0.02, 0.01, 0.005, 0.0025]))
    npts = int(rng.integers(1500, 9000))
    if dw <= 0.005 and seed % 2:
        npts *= 5           # grids several windows wide: the wave-shared gather's common stretches, aligned or not
    w0 = float(np.round(rng.choice([50.0, 700.0, 2300.0, 9000.0, 20000.0]) + rng.uniform(0, 50), 2))
    V = int(rng.integers(4, 9))
    nlines = int(rng.integers(40, 1.2e8 / ((V - 1) * 2 * 25 / dw)))      # keeps the oracle to seconds
    band = Band(str(tmp_path), w0, w0 + npts * dw, dw, nlines, seed=int(rng.integers(1, 10**6)), sw=w0 > 3000.0,
                with_cfc=w0 < 3000.0)
    col = syn.profile(int(rng.integers(0, 50)), V)
    col["p"] = col["p"] * float(rng.choice([0.3, 1.0, 1.0, 2.5]))
    col["t"] = np.clip(col["t"] + float(rng.uniform(-40, 30)), 150.0, 340.0)
    # (high wavenumbers on the finest grids put region 1 beyond what the first pass holds in LDS: those fall back)
    check(band, device, oracle, lib, col, tree=None, tile=int(rng.choice([0, 0, 128, 512])))


@pytest.mark.parametrize("dw,nlines", [(0.02, 4000), (0.005, 700)])
def test_batched_columns_equal_single_columns(tmp_path, device, dw, nlines):
    """grt_optical_depth_batch in the tree form (dense lines: moments through LDS; sparse lines: tiles of 1 024+ cells,
    moments added straight to global memory): every column of a batch equals the one-column call; the halo is the
    batch's largest (columns of very different surface pressure)."""
    import ctypes as C
    from scenario import MOL_ORDER
    band = Band(str(tmp_path), 1500.0, 1530.0, dw, nlines)
    V, ncol = 8, 3
    cols = [syn.profile(20 + c, V) for c in range(ncol)]
    cols[1]["p"] = cols[1]["p"] * 2.0
    cols[2]["p"] = cols[2]["p"] * 0.4
    go, grid = band.gas_optics(device, V, from_file=False)
    go.tune(fast=3)
    single, halos = [], []
    opt = api.OpticsObject(V - 1, grid, device)
    for col in cols:
        band.set_column(go, col)
        go.calculate_optical_depth(col["p"], col["t"], opt)
        single.append(opt.read()[0])
        halos.append(go.last_launch()["halo"])
    opt.destroy()
    gcols, keep = api.make_columns(cols, MOL_ORDER, cfc_order=(0, 1))
    buf = api.DeviceBuffer(device, 8 * ncol * (V - 1) * band.nw)
    api.check(api.load_library().grt_optical_depth_batch(C.byref(go.c), C.byref(gcols), buf.ptr))
    info = go.last_launch()
    batch = buf.to_host((ncol, V - 1, band.nw))
    assert info["tree_levels"] > 0 and info["halo"] == max(halos) and len(set(halos)) > 1, (info, halos)
    for c in range(ncol):
        scale = single[c].max(axis=1, keepdims=True)
        assert np.max(np.abs(batch[c] - single[c]) / scale) < 1e-9      # another halo, another order of additions
    buf.free()
    go.destroy()


def test_full_3m_point_grid_tree_equals_ring(device):
    """BASELINE.json's ~3 M-wavenumber grid at full width (1-3250 cm-1 at 0.001 cm-1, n = 3 249 001, 10^6 lines,
    12 layers to keep the read-back small): the oracle would need hours here, so the check is the size-independent
    one -- the cell hierarchy against the ring kernel, which evaluates every one of the 50 001 window points of
    every line and is itself pinned to the oracle on the small grids above."""
    import tempfile
    from grtcode_amd import workload as W
    V = 13
    root = tempfile.mkdtemp(prefix="grt_g3_")
    files, _ = W.write_tables(root, sw=False)
    spec = (1.0, 3250.0, 0.001)
    go, grid = W.build_band(device, spec, W.band_lines(W.LW_LINES, spec, 20261003), files, V)
    assert grid.n == 3249001
    col = syn.profile(3, V)
    for m in W.MOL_ORDER:
        go.set_molecule_ppmv(m, col["ppmv"][m])
    go.set_cfc_ppmv(0, col["cfc_ppmv"][0])
    go.set_cfc_ppmv(1, col["cfc_ppmv"][1])
    go.set_cia_ppmv(0, col["ppmv"][syn.N2])
    go.set_cia_ppmv(1, col["ppmv"][syn.O2])
    opt = api.OpticsObject(V - 1, grid, device)
    out = {}
    for fast in (3, 2):
        go.tune(fast=fast)
        go.calculate_optical_depth(col["p"], col["t"], opt)
        out[fast] = (opt.read()[0], go.last_launch())
    opt.destroy()
    go.destroy()
    (tree, info), (ring, _) = out[3], out[2]
    assert info["fast"] == 3 and info["tree_levels"] == 12 and info["moments"] == 12, info
    assert np.all(np.isfinite(tree)) and tree.min() >= 0.0
    assert tau_close(tree, ring) < BETWEEN_TOL


def test_full_3m_point_grid_against_the_reference_c(tmp_path, lib, device):
    """The same grid at full width (n = 3 249 001, windows of 50 001 points) against the REFERENCE'S OWN C -- not a
    self-comparison: a line list thin enough for the CPU (16 000 lines x 4 layers x 50 001 points = 3.2e9 line-shape
    evaluations: seconds with OpenMP) still sends the production form down its sparse-line branch -- tiles of 1 024
    cells and more, twelve moments added straight to global memory, twelve coarse levels -- which is the branch the
    10^6-line column takes too (0.3 lines per cell there)."""
    import os
    from oracle import reference_column as RC
    from scenario import Band
    kind, chk, orc = RC.checker(omp=True)
    if kind != "reference":
        pytest.skip("needs the prebuilt reference library oracle/_ref/libgrtref_omp.so (3.2e9 evaluations: a minute for the scalar restatement)")
    RC.set_omp_threads(min(os.cpu_count() or 1, 16))
    # GRT_G3_REF_LINES / GRT_G3_REF_LEVELS: a bigger case for a one-off run (10^6 lines x 4 layers = 2e11 evaluations: ~3
    # minutes of the reference's C on 16 threads; profiles/r3_g3_vs_reference.json)
    nlines = int(os.environ.get("GRT_G3_REF_LINES", 16000))
    V = int(os.environ.get("GRT_G3_REF_LEVELS", 5))
    band = Band(str(tmp_path), 1.0, 3250.0, 0.001, nlines, physical=True)
    assert band.nw == 3249001
    col = syn.profile(int(os.environ.get("GRT_G3_REF_PROFILE", 7)), V)
    go, grid = band.gas_optics(device, V, from_file=False)
    band.set_column(go, col)
    go.tune(fast=3)
    opt = api.OpticsObject(V - 1, grid, device)
    go.calculate_optical_depth(col["p"], col["t"], opt)
    tau = opt.read()[0]
    info = go.last_launch()
    assert info["fast"] == 3 and info["tree_levels"] == 12 and info["moments"] == 12 and info["tile"] >= 1024, info
    beat = None
    if os.environ.get("GRT_G3_REF_OUT"):
        # (a one-off run of many minutes inside one C call: say so once a minute, or the GPU box takes the run for hung)
        import threading
        import time
        stop = threading.Event()

        def heartbeat():
            t0 = time.time()
            while not stop.wait(45.0):
                with open(os.environ["GRT_G3_REF_OUT"] + ".progress", "a") as f:
                    f.write(f"reference C running, {time.time() - t0:.0f} s\n")
        beat = threading.Thread(target=heartbeat, daemon=True)
        beat.start()
    want = band.oracle_tau(chk, orc, lib, col)
    if beat is not None:
        stop.set()
    from oracle.reference_column import tau_metrics
    m = tau_metrics(tau, want)
    print(f"G3 grid, full width, {nlines} lines x {V - 1} layers, tree form vs {kind}: {m}")
    if os.environ.get("GRT_G3_REF_OUT"):
        import json
        json.dump({"grid": {"w0": 1.0, "wn": 3250.0, "dw": 0.001, "n": band.nw}, "lines": int(sum(v["v0"].size for v in band.lines.values())),
                   "layers": V - 1, "ran": info, "checker": kind, "tau_vs_reference": m}, open(os.environ["GRT_G3_REF_OUT"], "w"), indent=1)
    assert m["of_layer_max"] < 2e-6 and m["transmission"] < 4e-5
    opt.destroy()
    go.destroy()
