"""Parity of the line-by-line optical depth path (HIP, through the C ABI) against the oracle.

Tolerances: integer window indices bit-exact; shifted centres bit-exact; S(T), gamma, alpha to
1e-13 relative (ocml vs glibc exp/pow, <= 2 ulp); tau to 1e-11 relative of the layer's largest
tau (same formulas, different summation order + the libm difference above).
"""
import numpy as np
import pytest

from grtcode_amd import api, synthetic as syn
from scenario import Band, rel_err, MOL_ORDER, MOLTAB, mol_mass

pytestmark = pytest.mark.gpu

TAU_TOL = 1e-11


def tau_close(got, want, tol=TAU_TOL):
    scale = np.maximum(np.abs(want).max(axis=1, keepdims=True), 1e-300)
    return np.max(np.abs(got - want) / scale)


@pytest.fixture(scope="module")
def small_band(tmp_path_factory):
    return Band(str(tmp_path_factory.mktemp("band")), 500.0, 700.0, 1.0, 4000)


def test_line_prep_and_windows_bit_exact(small_band, oracle, lib, device):
    col = syn.profile(3, 21)
    go, grid = small_band.gas_optics(device, 21)
    small_band.set_column(go, col)
    got = go.debug_line_prep(col["p"], col["t"])
    p_atm = col["p"] * np.float64(np.float32(0.000986923))
    n, pavg, tavg = oracle.layer_means(p_atm, col["t"])
    kw = small_band.oracle_inputs(oracle, lib, col)
    assert got["v0"].size == sum(m["lines"]["v0"].size for m in kw["mols"])
    assert np.all(np.diff(got["v0"]) >= 0), "merged store must be sorted by centre"
    for slot, m in enumerate(kw["mols"]):
        sel = got["slot"] == slot
        ln = m["lines"]
        assert np.array_equal(got["v0"][sel], ln["v0"])
        ps, ns = oracle.species_means(p_atm, m["x"], n)
        vnn, snn, gamma, alpha = oracle.line_prep(ln, m["mass"], m["num_iso"], pavg, tavg, ps, m["q"])
        _, ws, we = oracle.line_sample(vnn, snn, gamma, alpha, ns, small_band.w0, small_band.dw,
                                       small_band.nw, windows=True)
        assert np.array_equal(got["win_s"][:, sel], ws), "window start indices must be bit-exact"
        assert np.array_equal(got["win_e"][:, sel], we), "window end indices must be bit-exact"
        assert np.array_equal(got["vnn"][:, sel], vnn), "shifted centres must be bit-exact"
        assert rel_err(got["snn"][:, sel], snn) < 1e-13
        assert rel_err(got["gamma"][:, sel], gamma) < 1e-13
        assert rel_err(got["alpha"][:, sel], alpha) < 1e-15
    go.destroy()


@pytest.mark.parametrize("dw,w0,wn,nlines", [(1.0, 500.0, 700.0, 4000), (0.1, 600.0, 640.0, 1500),
                                             (0.5, 1.0, 120.0, 800)])
@pytest.mark.parametrize("tile,nslice", [(0, 0), (64, 1), (256, 3)])
def test_tau_matches_oracle(tmp_path, oracle, lib, device, dw, w0, wn, nlines, tile, nslice):
    band = Band(str(tmp_path), w0, wn, dw, nlines)
    col = syn.profile(1, 13)
    go, grid = band.gas_optics(device, 13)
    go.tune(tile=tile, nslice=nslice)
    band.set_column(go, col)
    opt = api.OpticsObject(12, grid, device)
    go.calculate_optical_depth(col["p"], col["t"], opt)
    tau, omega, g = opt.read()
    want = band.oracle_tau(oracle, oracle, lib, col)
    assert tau.shape == want.shape
    assert np.all(omega == 0) and np.all(g == 0)          # launch.c writes tau only
    assert tau_close(tau, want) < TAU_TOL
    opt.destroy()
    go.destroy()


def test_lines_from_memory_equal_lines_from_file(small_band, device):
    col = syn.profile(2, 9)
    taus = []
    for from_file in (True, False):
        go, grid = small_band.gas_optics(device, 9, from_file=from_file)
        small_band.set_column(go, col)
        opt = api.OpticsObject(8, grid, device)
        go.calculate_optical_depth(col["p"], col["t"], opt)
        taus.append(opt.read()[0])
        opt.destroy()
        go.destroy()
    assert tau_close(taus[0], taus[1]) < 1e-13


def test_window_clipping_and_out_of_grid_lines(tmp_path, oracle, lib, device):
    # lines hugging both grid edges: windows clip at 0 and n-1; pressure shift pushes some
    # centres off the grid, which drops the whole line (kernels.c:433)
    band = Band(str(tmp_path), 100.0, 160.0, 0.25, 600, mols=[syn.H2O, syn.CO2], with_cfc=False,
                with_cia=False, with_ctm=False)
    for m in band.lines:
        v = band.lines[m]["v0"]
        v[: v.size // 3] = np.round(100.0 + (v[: v.size // 3] - 100.0) * 0.004, 6)     # within 0.24 of w0
        v[-(v.size // 3):] = np.round(160.0 - (160.0 - v[-(v.size // 3):]) * 0.004, 6)  # within 0.24 of wn
        band.lines[m]["v0"] = np.sort(v)
    col = syn.profile(5, 7)
    go, grid = band.gas_optics(device, 7, from_file=False)
    band.set_column(go, col)
    opt = api.OpticsObject(6, grid, device)
    go.calculate_optical_depth(col["p"], col["t"], opt)
    tau = opt.read()[0]
    want = band.oracle_tau(oracle, oracle, lib, col)
    assert tau_close(tau, want) < TAU_TOL
    opt.destroy()
    go.destroy()


def test_empty_line_list_leaves_only_continua(tmp_path, oracle, lib, device):
    band = Band(str(tmp_path), 800.0, 1000.0, 1.0, 0, mols=[syn.H2O, syn.O3])
    col = syn.profile(0, 11)
    go, grid = band.gas_optics(device, 11, from_file=False)
    band.set_column(go, col)
    opt = api.OpticsObject(10, grid, device)
    go.calculate_optical_depth(col["p"], col["t"], opt)
    tau = opt.read()[0]
    want = band.oracle_tau(oracle, oracle, lib, col)
    assert tau_close(tau, want) < 1e-13
    assert tau.max() > 0
    opt.destroy()
    go.destroy()


@pytest.mark.parametrize("dw,fast", [(10.0, 0), (10.0, 1), (5.0, 1), (2.0, 1)])
def test_coarse_grids_narrow_windows(tmp_path, oracle, lib, device, dw, fast):
    """Coarse grids (ERA5 runs its shortwave at 10 cm-1: 7-point windows) take the direct-walk path."""
    band = Band(str(tmp_path), 1.0, 3001.0, dw, 5000, sw=True)
    col = syn.profile(6, 17)
    go, grid = band.gas_optics(device, 17, from_file=False)
    go.tune(fast=fast)
    band.set_column(go, col)
    opt = api.OpticsObject(16, grid, device)
    go.calculate_optical_depth(col["p"], col["t"], opt)
    tau = opt.read()[0]
    want = band.oracle_tau(oracle, oracle, lib, col)
    assert tau_close(tau, want) < (2e-6 if fast else TAU_TOL)
    opt.destroy()
    go.destroy()
