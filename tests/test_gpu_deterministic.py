"""GRT_DETERMINISTIC=1 / grt_set_deterministic(1) (include/grt_ext.h): every floating-point sum of the line kernels in
one fixed order.  The default mode accumulates with LDS and L2 atomics in the scheduler's order (kernels.c:459's own
`#pragma omp atomic` / CUDA atomicAdd scatter is no more reproducible), so a repeated run moves tau by ~1e-11 of a
layer's largest value in the fused forms; here repeated runs must agree TO THE LAST BIT, in every form the library
has -- reference order, ring, one-pass and two-pass cell moments, the cell hierarchy with eight and with twelve
moments, both of its gathers -- and stay inside the same tolerances against the oracle as the default mode."""
import numpy as np
import pytest

from grtcode_amd import api, synthetic as syn
from scenario import Band, MOL_ORDER, RUN_TO_RUN_FUSED
from test_gpu_gas_optics import tau_close

pytestmark = pytest.mark.gpu


@pytest.fixture()
def deterministic(lib):
    api.check(lib.grt_set_deterministic(1))
    assert lib.grt_deterministic() == 1
    yield lib
    api.check(lib.grt_set_deterministic(-1))


def taus(band, device, col, fast, repeats, tile=0, nslice=0):
    V = col["p"].size
    go, grid = band.gas_optics(device, V, from_file=False)
    go.tune(tile=tile, nslice=nslice, fast=fast)
    band.set_column(go, col)
    opt = api.OpticsObject(V - 1, grid, device)
    out = []
    for _ in range(repeats):
        go.calculate_optical_depth(col["p"], col["t"], opt)
        out.append(opt.read()[0])
    info = go.last_launch()
    opt.destroy()
    go.destroy()
    return out, info


CASES = [
    # (w0, wn, dw, lines, levels, profile, fast, tile, what it is there for)
    (500.0, 900.0, 1.0, 8000, 21, 0, 0, 0, "reference order, line slices forced to one"),
    (500.0, 900.0, 1.0, 8000, 21, 0, 2, 0, "ring kernel"),
    (500.0, 900.0, 1.0, 8000, 21, 0, 1, 0, "one-pass cell moments"),
    (500.0, 900.0, 1.0, 8000, 21, 0, 3, 0, "two-pass cell moments, first pass in phases"),
    (500.0, 900.0, 1.0, 8000, 21, 3, 3, 64, "two-pass, 64-cell tiles: accumulators of three tiles overlap"),
    (1000.0, 1060.0, 0.02, 2500, 9, 2, 3, 0, "cell hierarchy, eight moments in LDS"),
    (2000.0, 2030.0, 0.005, 1500, 7, 4, 3, 0, "cell hierarchy, twelve moments straight to global memory"),
]


@pytest.mark.parametrize("case", CASES, ids=[c[-1].split(",")[0].replace(" ", "_") + f"_{i}" for i, c in enumerate(CASES)])
def test_repeated_runs_are_bit_identical(tmp_path, deterministic, oracle, device, case):
    w0, wn, dw, nlines, V, prof, fast, tile, _ = case
    band = Band(str(tmp_path), w0, wn, dw, nlines)
    col = syn.profile(prof, V)
    runs, info = taus(band, device, col, fast, 4, tile=tile)
    assert info["nslice"] == 1, info
    for r in runs[1:]:
        assert np.array_equal(r, runs[0]), (info, tau_close(r, runs[0]))
    want = band.oracle_tau(oracle, oracle, deterministic, col)
    assert tau_close(runs[0], want) < (1e-11 if fast == 0 else 2e-6)
    # and the default mode computes the same thing in another order (one line slice, as the deterministic mode takes:
    # slices cut the sorted lines into other blocks of 64, i.e. other groupings of the fp32 partial sums -- a few 1e-7)
    api.check(deterministic.grt_set_deterministic(0))
    free, _ = taus(band, device, col, fast, 1, tile=info["tile"], nslice=1)
    assert tau_close(free[0], runs[0]) < (1e-13 if fast == 0 else RUN_TO_RUN_FUSED)


@pytest.mark.parametrize("gather", ["lane", "wave"])
def test_both_tree_gathers(tmp_path, deterministic, device, gather, monkeypatch):
    monkeypatch.setenv("GRT_TREE_WAVE_MIN", "1000000000" if gather == "lane" else "1")
    band = Band(str(tmp_path), 700.0, 740.0, 0.01, 1200)
    runs, info = taus(band, device, syn.profile(7, 8), 3, 3)
    assert info["fast"] == 3 and info["tree_levels"] > 0, info
    assert np.array_equal(runs[1], runs[0]) and np.array_equal(runs[2], runs[0])


def test_environment_variable_is_read_at_every_launch(tmp_path, lib, device, monkeypatch):
    api.check(lib.grt_set_deterministic(-1))
    monkeypatch.setenv("GRT_DETERMINISTIC", "1")
    assert lib.grt_deterministic() == 1
    band = Band(str(tmp_path), 500.0, 700.0, 0.5, 4000)
    runs, info = taus(band, device, syn.profile(1, 11), 3, 3)
    assert info["nslice"] == 1
    assert np.array_equal(runs[1], runs[0]) and np.array_equal(runs[2], runs[0])
    monkeypatch.setenv("GRT_DETERMINISTIC", "0")
    assert lib.grt_deterministic() == 0


def test_pipeline_fluxes_are_bit_identical(tmp_path, deterministic, device):
    """The production pipeline (two-pass line kernel, fused solvers, ordered flux reduction) end to end."""
    lwb = Band(str(tmp_path / "lw"), 1.0, 700.0, 1.0, 6000)
    swb = Band(str(tmp_path / "sw"), 1.0, 9000.0, 1.0, 9000, sw=True)
    V, ncol = 16, 3
    go_lw, grid_lw = lwb.gas_optics(device, V, from_file=False)
    go_sw, grid_sw = swb.gas_optics(device, V, from_file=False)
    go_lw.tune(fast=3)
    go_sw.tune(fast=3)
    emis, alb = np.full(lwb.nw, 0.98), np.full(swb.nw, 0.2)
    solar = api.create_solar_flux(grid_sw, swb.files["solar"])
    gcols, keep = api.make_columns([syn.profile(30 + c, V) for c in range(ncol)], MOL_ORDER, cfc_order=(0, 1))
    pipe = api.Pipeline(go_lw, go_sw, ncol, 5, emis, alb, solar, spectral=False)
    got = []
    for _ in range(3):
        pipe.run(gcols)
        got.append(pipe.fluxes(ncol))
    assert np.array_equal(got[1], got[0]) and np.array_equal(got[2], got[0])
    assert np.all(got[0][:, 0] > 0)
    pipe.destroy()
    go_lw.destroy()
    go_sw.destroy()
