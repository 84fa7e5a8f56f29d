"""Parity of Rayleigh, optics combination and the LW/SW solvers (HIP, via the C ABI) vs the oracle.

These kernels restate the reference expression by expression (no FMA contraction), so the only
difference left is ocml vs glibc exp/sqrt rounding (<= 1 ulp per call): fluxes agree to 1e-12
relative to the largest flux in the row; Rayleigh and add_optics use no libm call and are bit-exact.
"""
import numpy as np
import pytest

from grtcode_amd import api, synthetic as syn

pytestmark = pytest.mark.gpu


def row_err(got, want):
    scale = np.maximum(np.abs(want).max(axis=1, keepdims=True), 1e-300)
    return np.max(np.abs(got - want) / scale)


def random_optics(rng, L, n, scatter=True):
    tau = 10.0 ** rng.uniform(-6, 1.5, (L, n))
    omega = rng.uniform(0.0, 0.999, (L, n)) if scatter else np.zeros((L, n))
    g = rng.uniform(-0.5, 0.9, (L, n)) if scatter else np.zeros((L, n))
    return tau, omega, g


def test_rayleigh_bit_exact(oracle, device):
    col = syn.profile(0, 31)
    grid = api.create_spectral_grid(1.0, 50000.0, 10.0)
    o = api.OpticsObject(30, grid, device)
    o.rayleigh(col["p"])
    tau, omega, g = o.read()
    wt, wo, wg = oracle.rayleigh(30, col["p"], 1.0, 10.0, grid.n)
    assert np.array_equal(tau, wt) and np.array_equal(omega, wo) and np.array_equal(g, wg)
    o.destroy()


def test_add_optics_bit_exact_and_gas_only_nan(oracle, device):
    rng = np.random.default_rng(7)
    grid = api.create_spectral_grid(100.0, 400.0, 0.5)
    L, n = 5, grid.n
    sets = [random_optics(rng, L, n) for _ in range(3)]
    objs = []
    for t, om, g in sets:
        o = api.OpticsObject(L, grid, device)
        o.update(t, om, g)
        objs.append(o)
    res = api.add_optics(objs)
    tau, omega, g = res.read()
    wt, wo, wg = oracle.add_optics([s[0] for s in sets], [s[1] for s in sets], [s[2] for s in sets])
    assert np.array_equal(tau, wt) and np.array_equal(omega, wo) and np.array_equal(g, wg)
    res.destroy()
    # gas-only combination: sum(omega*tau) = 0 -> g = 0/0 = NaN, as in the reference (optics.c:144)
    gas = api.OpticsObject(L, grid, device)
    gas.update(sets[0][0], np.zeros((L, n)), np.zeros((L, n)))
    res = api.add_optics([gas])
    assert np.all(np.isnan(res.read()[2]))
    for o in objs + [gas, res]:
        o.destroy()


@pytest.mark.parametrize("K", [8, 9, 23])
def test_add_optics_any_number_of_objects(oracle, lib, device, K):
    """The reference's add_optics has no limit on the number of objects (optics.c:84-124); eight travel as kernel
    arguments, more through a device pointer table -- same sums in the same order, so bit-exact either way.  Also
    exercises the parked-block cache: 23 objects of one size destroyed, evicted oldest-first, then flushed."""
    rng = np.random.default_rng(100 + K)
    grid = api.create_spectral_grid(100.0, 228.0, 0.5)
    L, n = 4, grid.n
    sets = [random_optics(rng, L, n) for _ in range(K)]
    objs = []
    for t, om, g in sets:
        o = api.OpticsObject(L, grid, device)
        o.update(t, om, g)
        objs.append(o)
    res = api.add_optics(objs)
    tau, omega, g = res.read()
    wt, wo, wg = oracle.add_optics([s[0] for s in sets], [s[1] for s in sets], [s[2] for s in sets])
    assert np.array_equal(tau, wt) and np.array_equal(omega, wo) and np.array_equal(g, wg)
    for o in objs + [res]:
        o.destroy()
    api.check(lib.grt_optics_cache_flush())
    api.check(lib.grt_optics_cache_flush())        # idempotent


def test_add_optics_rejects_incompatible(device):
    g1 = api.create_spectral_grid(100.0, 400.0, 0.5)
    g2 = api.create_spectral_grid(100.0, 400.0, 1.0)
    a, b = api.OpticsObject(4, g1, device), api.OpticsObject(4, g2, device)
    with pytest.raises(api.GrtError) as e:
        api.add_optics([a, b])
    assert e.value.code == api.VALUE_ERR
    a.destroy()
    b.destroy()


@pytest.mark.parametrize("L,w0,wn,dw", [(60, 1.0, 3250.0, 1.0), (1, 1.0, 50000.0, 10.0), (12, 500.0, 520.0, 0.01)])
def test_longwave_matches_oracle(oracle, device, L, w0, wn, dw):
    rng = np.random.default_rng(L)
    grid = api.create_spectral_grid(w0, wn, dw)
    n = grid.n
    col = syn.profile(4, L + 1)
    tau, omega, g = random_optics(rng, L, n)
    emis = rng.uniform(0.9, 1.0, n)
    o = api.OpticsObject(L, grid, device)
    o.update(tau, omega, g)
    lw = api.LongwaveObject(L + 1, grid, device)
    up, dn = lw.fluxes(o, col["t_surf"], col["t_layer"], col["t"], emis)
    wu, wd = oracle.lw_fluxes(w0, dw, col["t_surf"], col["t_layer"], col["t"], tau, omega, emis)
    assert row_err(up, wu) < 1e-12 and row_err(dn, wd) < 1e-12
    lw.destroy()
    o.destroy()


@pytest.mark.parametrize("L", [1, 5, 6, 7, 13, 60])
def test_longwave_layer_parallel_form_equals_the_chains(device, monkeypatch, L):
    """calculate_lw_fluxes works the layers' extinctions and Planck terms out first, one thread per (layer, wavenumber),
    and its sweeps read them six layers at a time; GRT_LW_COLUMN_CHAINS=1 is the one-thread-per-wavenumber kernel of
    before.  The fluxes are equal to the last bit."""
    rng = np.random.default_rng(800 + L)
    grid = api.create_spectral_grid(1.0, 3250.0, 1.0)
    n = grid.n
    col = syn.profile(3, L + 1)
    tau, omega, g = random_optics(rng, L, n)
    tau[0, : n // 8] = 1e4          # clamped exponent (longwave.c:177-183)
    emis = rng.uniform(0.9, 1.0, n)
    o = api.OpticsObject(L, grid, device)
    o.update(tau, omega, g)
    lw = api.LongwaveObject(L + 1, grid, device)
    up, dn = (x.copy() for x in lw.fluxes(o, col["t_surf"], col["t_layer"], col["t"], emis))
    monkeypatch.setenv("GRT_LW_COLUMN_CHAINS", "1")
    up1, dn1 = lw.fluxes(o, col["t_surf"], col["t_layer"], col["t"], emis)
    assert np.array_equal(up, up1) and np.array_equal(dn, dn1)
    assert np.all(np.isfinite(up)) and up.max() > 0.0
    lw.destroy()
    o.destroy()


@pytest.mark.parametrize("case", ["random", "huge", "tiny", "near_overflow"])
def test_longwave_extreme_optical_depths(oracle, device, case):
    # the four regimes of longwave/test/test_longwave.c:102-209 (there only checked for SUCCESS)
    grid = api.create_spectral_grid(1.0, 50000.0, 10.0)
    n = grid.n
    rng = np.random.default_rng(1)
    tau = {"random": rng.uniform(0, 1, (1, n)), "huge": np.full((1, n), 1e12), "tiny": np.full((1, n), 1e-12),
           "near_overflow": np.full((1, n), np.log(np.finfo(np.float64).max) - 12.0)}[case]
    omega = np.zeros((1, n))
    o = api.OpticsObject(1, grid, device)
    o.update(tau, omega, omega)
    lw = api.LongwaveObject(2, grid, device)
    T = np.array([250.0, 288.0])
    up, dn = lw.fluxes(o, 290.0, np.array([269.0]), T, np.full(n, 0.98))
    wu, wd = oracle.lw_fluxes(1.0, 10.0, 290.0, np.array([269.0]), T, tau, omega, np.full(n, 0.98))
    assert np.all(np.isfinite(up)) and np.all(np.isfinite(dn))
    assert row_err(up, wu) < 1e-12 and row_err(dn, wd) < 1e-12
    lw.destroy()
    o.destroy()


def test_longwave_rejects_bad_inputs(device):
    grid = api.create_spectral_grid(1.0, 100.0, 1.0)
    o = api.OpticsObject(2, grid, device)
    lw = api.LongwaveObject(3, grid, device)
    T = np.array([250.0, 260.0, 270.0])
    with pytest.raises(api.GrtError) as e:
        lw.fluxes(o, 600.0, T[:2], T, np.full(grid.n, 0.9))           # T_surf out of range
    assert e.value.code == api.RANGE_ERR
    with pytest.raises(api.GrtError) as e:
        lw.fluxes(o, 280.0, T[:2], T, np.full(grid.n, 1.5))           # emissivity > 1
    assert e.value.code == api.RANGE_ERR
    lw4 = api.LongwaveObject(4, grid, device)
    with pytest.raises(api.GrtError) as e:
        lw4.fluxes(o, 280.0, T, np.append(T, 280.0), np.full(grid.n, 0.9))   # level count mismatch
    assert e.value.code == api.VALUE_ERR
    for x in (lw, lw4, o):
        x.destroy()


@pytest.mark.parametrize("L,dw", [(60, 10.0), (1, 10.0), (7, 7.0)])
@pytest.mark.parametrize("scatter", [True, False])
def test_shortwave_matches_oracle(oracle, device, L, dw, scatter):
    rng = np.random.default_rng(100 + L)
    grid = api.create_spectral_grid(1.0, 50000.0, dw)
    n = grid.n
    tau, omega, g = random_optics(rng, L, n, scatter)
    if scatter:
        omega[0, : n // 4] = 1.0          # conservative-scattering branch (shortwave.c:169-179)
        omega[-1, n // 4: n // 2] = 0.0   # no-scattering branch (:114-122)
        tau[L // 2, : n // 8] = 0.0       # "no gas in the layer" branch (:149-158)
    alb = rng.uniform(0.0, 0.6, n)
    solar = rng.uniform(0.0, 1e-4, n)
    o = api.OpticsObject(L, grid, device)
    o.update(tau, omega, g)
    sw = api.ShortwaveObject(L + 1, grid, device)
    up, dn = sw.fluxes(o, 0.6, 0.5, alb, alb, 1360.0, solar)
    wu, wd = oracle.sw_fluxes(omega, g, tau, 0.6, 0.5, alb, alb, 1360.0, solar)
    assert np.all(np.isfinite(up)) and np.all(np.isfinite(dn))
    scale = max(np.abs(wu).max(), np.abs(wd).max())
    assert np.max(np.abs(up - wu)) / scale < 1e-12 and np.max(np.abs(dn - wd)) / scale < 1e-12
    sw.destroy()
    o.destroy()


@pytest.mark.parametrize("L", [1, 5, 6, 7, 13, 60])
def test_shortwave_layer_parallel_form_equals_the_chains(device, monkeypatch, L):
    """calculate_sw_fluxes works the layers' properties out first, one thread per (layer, wavenumber), and its sweeps
    read them six layers at a time; GRT_SW_COLUMN_CHAINS=1 is the one-thread-per-wavenumber kernel of before.  Same
    expressions on the same doubles: the fluxes are equal to the last bit (layer counts around the sweeps' chunk of six)."""
    rng = np.random.default_rng(700 + L)
    grid = api.create_spectral_grid(1.0, 5000.0, 1.0)
    n = grid.n
    tau, omega, g = random_optics(rng, L, n, True)
    omega[0, : n // 4] = 1.0
    omega[-1, n // 4: n // 2] = 0.0
    tau[L // 2, : n // 8] = 0.0
    alb = rng.uniform(0.0, 0.6, n)
    solar = rng.uniform(0.0, 1e-4, n)
    o = api.OpticsObject(L, grid, device)
    o.update(tau, omega, g)
    sw = api.ShortwaveObject(L + 1, grid, device)
    up, dn = (x.copy() for x in sw.fluxes(o, 0.6, 0.5, alb, alb, 1360.0, solar))
    monkeypatch.setenv("GRT_SW_COLUMN_CHAINS", "1")
    up1, dn1 = sw.fluxes(o, 0.6, 0.5, alb, alb, 1360.0, solar)
    assert np.array_equal(up, up1) and np.array_equal(dn, dn1)
    assert np.all(np.isfinite(up)) and up.max() > 0.0
    sw.destroy()
    o.destroy()


@pytest.mark.parametrize("tauval", [1e12, 1e-12, 697.0])
def test_shortwave_extreme_optical_depths(oracle, device, tauval):
    grid = api.create_spectral_grid(1.0, 50000.0, 10.0)
    n = grid.n
    tau = np.full((1, n), tauval)
    omega = np.full((1, n), 0.5)
    g = np.full((1, n), 0.3)
    o = api.OpticsObject(1, grid, device)
    o.update(tau, omega, g)
    sw = api.ShortwaveObject(2, grid, device)
    alb = np.full(n, 0.2)
    solar = np.full(n, 2e-5)
    up, dn = sw.fluxes(o, 0.5, 0.5, alb, alb, 1360.0, solar)
    wu, wd = oracle.sw_fluxes(omega, g, tau, 0.5, 0.5, alb, alb, 1360.0, solar)
    assert np.all(np.isfinite(up)) and np.all(np.isfinite(dn))
    scale = max(np.abs(wu).max(), np.abs(wd).max())
    assert np.max(np.abs(up - wu)) / scale < 1e-12 and np.max(np.abs(dn - wd)) / scale < 1e-12
    sw.destroy()
    o.destroy()


def test_flux_rows_opt_in_copies_only_the_rows_asked_for(oracle, device, monkeypatch):
    """GRT_FLUX_ROWS=toa,sfc,<k>: a caller that integrates three levels gets those rows of
    flux_up / flux_down and keeps the others as it had them -- instead of 2 V n doubles over PCIe per call."""
    L, w0, wn, dw = 12, 1.0, 3250.0, 1.0
    rng = np.random.default_rng(3)
    grid = api.create_spectral_grid(w0, wn, dw)
    n = grid.n
    col = syn.profile(4, L + 1)
    tau, omega, g = random_optics(rng, L, n)
    emis = rng.uniform(0.9, 1.0, n)
    o = api.OpticsObject(L, grid, device)
    o.update(tau, omega, g)
    lw = api.LongwaveObject(L + 1, grid, device)
    full_up, full_dn = lw.fluxes(o, col["t_surf"], col["t_layer"], col["t"], emis)
    monkeypatch.setenv("GRT_FLUX_ROWS", "toa, sfc,5")
    up, dn = np.full((L + 1, n), -7.0), np.full((L + 1, n), -7.0)
    lw.fluxes(o, col["t_surf"], col["t_layer"], col["t"], emis, out=(up, dn))
    rows = [0, L, 5]
    others = [k for k in range(L + 1) if k not in rows]
    assert np.array_equal(up[rows], full_up[rows]) and np.array_equal(dn[rows], full_dn[rows])
    assert np.all(up[others] == -7.0) and np.all(dn[others] == -7.0)
    monkeypatch.setenv("GRT_FLUX_ROWS", "toa,99")                 # outside 0..L: reported, everything copied
    up2, dn2 = lw.fluxes(o, col["t_surf"], col["t_layer"], col["t"], emis)
    assert np.array_equal(up2, full_up) and np.array_equal(dn2, full_dn)
    lw.destroy()
    o.destroy()
