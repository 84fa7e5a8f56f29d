"""The fused form's production kernel (fast=1: far wings by cell moments, k_gas_optics_mp.hip) against the oracle.

Every case states which branch of the kernel it is there for.  Tolerance of the fused form: 2e-6 of each layer's
largest optical depth (the reference's own fp32 rounding in the Humlicek core is ~1e-6 of a line peak; the moment
series itself is cut below 1e-7 of the far-wing value), fluxes 1e-3 W m-2 (BASELINE.json) -- in practice ~1e-6.
The ring kernel (fast=2, every window point evaluated) is compared as well: the two fused forms share everything
but the far wings, so they must agree much more closely than either does with the reference.
"""
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


def run(band, device, col, fast, tile=0, nslice=0, from_file=False):
    V = col["p"].size
    go, grid = band.gas_optics(device, V, from_file=from_file)
    go.tune(tile=tile, nslice=nslice, fast=fast)
    band.set_column(go, col)
    opt = api.OpticsObject(V - 1, grid, device)
    go.calculate_optical_depth(col["p"], col["t"], opt)
    tau = opt.read()[0]
    opt.destroy()
    go.destroy()
    return tau


# the two-pass form's LEAN line loop (fp32 preparation from packed records, round 4) against its general line loop (fp64
# preparation): per-line factors a few fp32 roundings apart -- strength, Lorentz width, 1/Doppler width
LEAN_TOL = 1.5e-6


def check(band, device, oracle, lib, col, **kw):
    want = band.oracle_tau(oracle, oracle, lib, col)
    mp = run(band, device, col, 1, **kw)
    ring = run(band, device, col, 2, **kw)
    lean = run(band, device, col, 3, **kw)              # two-pass form: cell moments through global memory; lean line loop
    os.environ["GRT_LEAN"] = "0"
    try:
        two = run(band, device, col, 3, **kw)           # ... with the general line loop: the one-pass form's arithmetic
    finally:
        del os.environ["GRT_LEAN"]
    e_mp, e_ring, e_between = tau_close(mp, want), tau_close(ring, want), tau_close(mp, ring)
    e_two, e_forms = tau_close(two, want), tau_close(two, mp)
    e_lean, e_lean_two = tau_close(lean, want), tau_close(lean, two)
    print(f"moment kernel vs oracle {e_mp:.2e}; ring kernel vs oracle {e_ring:.2e}; moment vs ring {e_between:.2e}; "
          f"two-pass vs oracle {e_two:.2e}, vs one-pass {e_forms:.2e}; lean two-pass vs oracle {e_lean:.2e}, vs general {e_lean_two:.2e}")
    assert e_lean < FAST_TOL
    assert e_lean_two < LEAN_TOL
    if os.environ.get("GRT_STRESS_STRICT"):             # soak runs: the reference-order form on the same case
        e_strict = tau_close(run(band, device, col, 0, **kw), want)
        print(f"reference-order form vs oracle {e_strict:.2e}")
        assert e_strict < 1e-11
    assert e_mp < FAST_TOL
    assert e_ring < FAST_TOL
    assert e_two < FAST_TOL
    assert e_between < BETWEEN_TOL
    # same arithmetic, other groupings of the fp32 ring sums (where the one-pass form does not apply -- wide windows --
    # fast=1 is the ring kernel itself and the comparison is the one above): a handful of fp32 roundings of a layer's
    # largest tau apart (a soak of 5 000 cases met 3.15e-7 once)
    assert e_forms < (6e-7 if e_between > 1e-12 else BETWEEN_TOL)
    return mp, want


@pytest.mark.parametrize("tile,nslice", [(0, 0), (64, 1), (128, 3), (1024, 2)])
def test_dense_lines_many_per_cell(tmp_path, oracle, lib, device, tile, nslice):
    """~100 lines per grid point: waves whose 64 lines share one or two cells (register reduction of the moments),
    tile edges every 64/128 points, line slices combined with global atomics."""
    band = Band(str(tmp_path), 900.0, 1100.0, 1.0, 20000)
    check(band, device, oracle, lib, syn.profile(2, 13), tile=tile, nslice=nslice)


@pytest.mark.parametrize("w0,prof", [(30000.0, 3), (46000.0, 7), (18000.0, 11)])
def test_region_two_points_of_the_lean_loop_against_the_exact_path(tmp_path, oracle, lib, device, w0, prof):
    """Round 5: the lean loop evaluates Humlicek region 2 (XLIM2 = 6.8 - y <= |x| < XLIM1, RFM_voigt.c:113, :187-199) itself,
    with its own fp32 x and y; the general loop sends the same points through the exact preparation and the class queues.
    At these wavenumbers a grid step is 10-30 Doppler widths, so about half of all lines have their own grid point in
    region 2 (and a quarter in regions 3-4): lean against general within LEAN_TOL, both within FAST_TOL of the oracle --
    at line centres too, where a layer's largest optical depths are."""
    band = Band(str(tmp_path), w0, w0 + 399.0, 1.0, 24000, sw=True, with_cfc=False)
    col = syn.profile(prof, 13)
    mp, want = check(band, device, oracle, lib, col)
    peak = want.argmax(axis=1)
    rows = np.arange(want.shape[0])
    lean = run(band, device, col, 3)
    assert np.max(np.abs(lean[rows, peak] - want[rows, peak]) / want[rows, peak]) < 2e-6


def test_lone_column_of_a_crowded_band_is_cut_by_line_count(tmp_path, oracle, lib, device, monkeypatch):
    """A band whose lines crowd into one tile (real line lists: the infrared end of a shortwave band) as ONE column: the
    launch's work list (GrtGasOpticsArgs.tile_items) cuts that tile into pieces of ~10 000 lines and leaves the empty
    tiles whole; same optical depths as the equal slices it replaces (GRT_TILE_ITEMS=0) and as the oracle."""
    band = Band(str(tmp_path), 1000.0, 2599.0, 1.0, 60000, line_range=(1000.0, 1100.0), with_cfc=False, with_cia=False)
    col = syn.profile(5, 13)
    V = col["p"].size
    want = band.oracle_tau(oracle, oracle, lib, col)
    taus = {}
    for label, env in (("items", None), ("slices", "0")):
        if env is not None:
            monkeypatch.setenv("GRT_TILE_ITEMS", env)
        go, grid = band.gas_optics(device, V, from_file=False)
        go.tune(fast=3)
        band.set_column(go, col)
        opt = api.OpticsObject(V - 1, grid, device)
        go.calculate_optical_depth(col["p"], col["t"], opt)
        taus[label] = opt.read()[0]
        info = go.last_launch()
        if label == "items" and not api.load_library().grt_deterministic():
            assert info["tile"] == 256 and info["nslice"] == 6, info       # 60 000 lines in tile 0: six pieces
            # the list itself (grt_debug_tile_items): every tile's pieces are its candidate range cut without gaps or
            # overlaps, in order; only crowded tiles are cut, into pieces of about 10 000 lines
            items, ranges = go.tile_items()
            assert ranges.shape[0] == 7 and items.shape[0] == 6 + 6
            for t in range(ranges.shape[0]):
                mine = items[items[:, 0] == t]
                assert list(mine[:, 3]) == list(range(len(mine)))
                assert mine[0, 1] == ranges[t, 0] and mine[-1, 2] == ranges[t, 1]
                assert np.array_equal(mine[1:, 1], mine[:-1, 2])
                count = int(ranges[t, 1]) - int(ranges[t, 0])
                assert len(mine) == (1 if count <= 16000 else (count + 5000) // 10000)
            assert int(ranges[0, 1]) - int(ranges[0, 0]) == 60000
        opt.destroy()
        go.destroy()
    assert tau_close(taus["items"], want) < FAST_TOL
    assert tau_close(taus["items"], taus["slices"]) < 6e-7


def test_sparse_lines_many_cells_per_wave(tmp_path, oracle, lib, device):
    """One line every ~10 grid points: a wave's lines span hundreds of cells (per-lane LDS adds of the moments)."""
    band = Band(str(tmp_path), 1.0, 3000.0, 1.0, 300)
    check(band, device, oracle, lib, syn.profile(4, 11))


def test_tables_that_start_and_end_inside_tiles_on_a_grid_of_odd_length(tmp_path, oracle, lib, device):
    """The epilogue (write_tile) reads a spectral table only where it holds anything (GrtTableSpans) and two entries per
    load where the table's row is 16-byte aligned: 621 grid points put every other row off that alignment, the CFC tables
    (700 - 1300 cm-1) begin and end inside tiles, the N2-N2 table (1 - 400) lies wholly outside the grid, the O2 pairs'
    (from 1200) begin inside it."""
    band = Band(str(tmp_path), 690.0, 1310.0, 1.0, 2500)
    assert band.nw % 2 == 1
    check(band, device, oracle, lib, syn.profile(6, 9))
    check(band, device, oracle, lib, syn.profile(3, 9), tile=64, nslice=2)


def test_lines_hugging_the_grid_edges(tmp_path, oracle, lib, device):
    """Windows clipped at index 0 and n-1, centres pushed off the grid by the pressure shift (kernels.c:433-437)."""
    band = Band(str(tmp_path), 100.0, 400.0, 1.0, 900, mols=[syn.H2O, syn.CO2], with_cfc=False, with_cia=False,
                with_ctm=False)
    for m in band.lines:
        v = band.lines[m]["v0"]
        v[: v.size // 3] = np.round(100.0 + (v[: v.size // 3] - 100.0) * 0.004, 6)
        v[-(v.size // 3):] = np.round(400.0 - (400.0 - v[-(v.size // 3):]) * 0.004, 6)
        band.lines[m]["v0"] = np.sort(v)
    check(band, device, oracle, lib, syn.profile(5, 9))


def test_high_pressure_widens_the_near_field(tmp_path, oracle, lib, device):
    """Three atmospheres at the surface: Lorentz widths of several tenths of a grid step, so the moment series
    needs a wider near field (R grows with gamma/wres) in the lowest layers and the smallest in the highest."""
    band = Band(str(tmp_path), 600.0, 800.0, 0.5, 6000)
    col = syn.profile(7, 15)
    col["p"] = col["p"] * 3.0
    check(band, device, oracle, lib, col)


def test_fine_grid_high_wavenumber_region1_beyond_near_field(tmp_path, oracle, lib, device):
    """0.1 cm-1 at 44 000 cm-1: Doppler widths of ~0.06 cm-1, so Humlicek region 1 (|x| < XLIM0) reaches ~90 grid
    steps from the centre -- beyond the near field, where it is applied as a correction to the moment series; the
    near-centre queue receives tens of points per line."""
    band = Band(str(tmp_path), 44000.0, 44200.0, 0.1, 1200, sw=True, with_cfc=False)
    check(band, device, oracle, lib, syn.profile(8, 9))


def test_half_wavenumber_grid_uses_moments_too(tmp_path, oracle, lib, device):
    band = Band(str(tmp_path), 2000.0, 2400.0, 0.5, 5000)
    check(band, device, oracle, lib, syn.profile(9, 12))


def test_window_narrower_than_near_field_falls_back_to_all_near(tmp_path, oracle, lib, device):
    """1.5 cm-1: 17 grid steps per half window; under a (deliberately absurd) 40 atm surface pressure R + 4 > fsteps
    in the lowest layers, where the workgroup treats the whole window as near field (no moments) -- both branches
    in one launch."""
    band = Band(str(tmp_path), 500.0, 1400.0, 1.5, 6000)
    col = syn.profile(3, 15)
    col["p"] = col["p"] * 40.0
    check(band, device, oracle, lib, col)


def test_temperature_exponents_that_are_not_hundredths(tmp_path, oracle, lib, device):
    """HITRAN writes n_air with two decimals and the kernel tabulates (296/T)^(k/100) per workgroup; lines that come in
    through grt_add_molecule_lines may carry anything -- negative, above 1.27, seven digits -- and take the direct
    exponential (a call inside the kernel).  Both in one wave."""
    band = Band(str(tmp_path), 1200.0, 1500.0, 1.0, 3000, mols=[syn.H2O, syn.CO2, syn.CH4])
    rng = np.random.default_rng(5)
    for m in band.lines:
        n = band.lines[m]["nexp"].copy()
        odd = rng.random(n.size) < 0.4
        n[odd] = np.float32(rng.uniform(-0.5, 1.6, odd.sum()))
        band.lines[m]["nexp"] = n.astype(np.float32).astype(np.float64)
    check(band, device, oracle, lib, syn.profile(12, 10))


def test_line_centres_exactly_on_grid_points(tmp_path, oracle, lib, device):
    """x = 0 at a grid point in every layer (no pressure shift): in the thin upper layers the Lorentzian the ring
    adds there, y/(pi y^2), is hundreds of times the true Voigt value the near-centre queue replaces it with -- the
    two must cancel to rounding (one fp32-rounded amplitude used by both)."""
    band = Band(str(tmp_path), 800.0, 1000.0, 1.0, 1500, mols=[syn.H2O, syn.CO2, syn.O3])
    for m in band.lines:
        band.lines[m]["v0"] = np.sort(np.round(band.lines[m]["v0"]))
        band.lines[m]["delta"] = np.zeros_like(band.lines[m]["delta"])
    mp, want = check(band, device, oracle, lib, syn.profile(10, 13))
    # the line centres carry the largest optical depths of every layer: compare there, relative to the value itself
    peak = want.argmax(axis=1)
    rows = np.arange(want.shape[0])
    assert np.max(np.abs(mp[rows, peak] - want[rows, peak]) / want[rows, peak]) < 2e-6


@pytest.mark.parametrize("seed", range(int(os.environ.get("GRT_STRESS_FIRST", 0)),
                                         int(os.environ.get("GRT_STRESS_FIRST", 0)) + int(os.environ.get("GRT_STRESS_SEEDS", 16))))   # soak runs
def test_randomised_grids_profiles_and_launch_shapes(tmp_path, oracle, lib, device, seed):
    """Random band position, grid spacing (windows of 33 to 501 points), line density, number of levels, surface
    pressure, temperature offset, tile size and line slicing: the one- and two-pass moment kernels and the ring kernel
    all within the fused form's tolerance of the oracle, and within rounding of each other."""
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
    band = Band(str(tmp_path), w0, w0 + span, dw, nlines, seed=int(rng.integers(1, 10**6)), sw=w0 > 3000.0,
                with_cfc=w0 < 3000.0)
    col = syn.profile(int(rng.integers(0, 50)), V)
    col["p"] = col["p"] * (float(rng.choice([0.3, 1.0, 1.0, 2.5])) if os.environ.get("GRT_STRESS_WIDE", "") != "2" else float(10.0 ** rng.uniform(-1.5, 0.7)))
    col["t"] = np.clip(col["t"] + float(rng.uniform(-40, 30)), 150.0, 340.0)
    tile = int(rng.choice([0, 0, 64, 128, 256, 512]))
    nslice = int(rng.choice([0, 0, 1, 2, 5]))
    check(band, device, oracle, lib, col, tile=tile, nslice=nslice)


@pytest.mark.parametrize("lines_per_cell,cells,want", [(308, 200, (128, 4)), (30, 1000, (256, 1)), (2, 600, (256, 1))])
def test_one_column_launch_shape_keeps_a_few_thousand_lines_per_workgroup(tmp_path, device, lines_per_cell, cells, want):
    """auto_tune (grt_gas_optics.c) for the reference-shaped one-column call: tiles are narrowed and cut into line slices
    to make more workgroups only while a workgroup keeps ~6 000-8 000 lines -- its fixed costs are paid per tile and slice.
    Measured on G1 (round 4's lean first pass): the longwave band's density wants 128-cell tiles in 4 slices
    (0.56 ms; 64-cell tiles in 2: 0.59), the shortwave band's 256-cell tiles unsliced (1.52 ms; 128-cell tiles: 1.58)."""
    band = Band(str(tmp_path), 1000.0, 1000.0 + cells - 1, 1.0, lines_per_cell * cells, with_cfc=False, with_cia=False, with_ctm=False)
    V = 61
    go, grid = band.gas_optics(device, V, from_file=False)
    go.tune(fast=3)
    col = syn.profile(0, V)
    band.set_column(go, col)
    opt = api.OpticsObject(V - 1, grid, device)
    go.calculate_optical_depth(col["p"], col["t"], opt)
    info = go.last_launch()
    if api.load_library().grt_deterministic():
        want = (want[0], 1)        # (the verification mode never cuts a tile into line slices)
    assert (info["fast"], info["tile"], info["nslice"]) == (3,) + want, info
    opt.destroy()
    go.destroy()
