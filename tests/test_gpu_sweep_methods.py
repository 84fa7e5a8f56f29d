"""§8(f)-2: optical_depth_method = wavenumber_sweep (the library default, gas_optics.c:110-113) and line_sweep on
the GPU (k_gas_optics_sweep.hip) against the oracle's restatement of kernels.c:135-406,514-581 -- which is itself
bit-exact against the reference build (tests/test_oracle_golden.py::test_sweep_methods_bit_exact_vs_live_reference).

Tolerance: 1e-11 of each layer's largest optical depth (same formulas in the reference's operation order; what
differs is the order of the additions and ocml-vs-glibc exp/pow, as for line_sample).  Bin membership -- which lines
are "local" and which "remote" for a bin, i.e. the reference's bracket() searches on the per-layer sorted centres --
is integer logic and must agree exactly: a single line on the wrong side shows up as 1e-3, not 1e-11.
"""
import numpy as np
import pytest

from grtcode_amd import api, synthetic as syn
from scenario import Band
from test_gpu_gas_optics import tau_close

pytestmark = pytest.mark.gpu
WAVENUMBER_SWEEP, LINE_SWEEP = 0, 1


def run(band, device, col, method, from_file=False):
    V = col["p"].size
    go, grid = band.gas_optics(device, V, from_file=from_file, method=method)
    assert go.c.optical_depth_method == (WAVENUMBER_SWEEP if method is None else method)
    band.set_column(go, col)
    opt = api.OpticsObject(V - 1, grid, device)
    go.calculate_optical_depth(col["p"], col["t"], opt)
    tau = opt.read()[0]
    opt.destroy()
    go.destroy()
    return tau


@pytest.mark.parametrize("method", [WAVENUMBER_SWEEP, LINE_SWEEP])
@pytest.mark.parametrize("dw,w0,span,nlines", [(1.0, 400.0, 300.0, 6000), (0.1, 600.0, 80.0, 3000),
                                                (0.25, 1.0, 120.0, 2500), (0.02, 900.0, 40.0, 1500), (2.0, 100.0, 600.0, 4000)])
def test_sweep_matches_oracle(tmp_path, oracle, lib, device, method, dw, w0, span, nlines):
    # lines stay 26.5 cm-1 below the top of the grid for line_sweep (see the oracle test: the reference indexes one
    # bin past its arrays for nearer ones)
    top = w0 + span - (26.5 if method == LINE_SWEEP else 0.0)
    band = Band(str(tmp_path), w0, w0 + span, dw, nlines, line_range=(w0, top))
    col = syn.profile(3, 11)
    got = run(band, device, col, method)
    want = band.oracle_tau(oracle, oracle, lib, col, method=method)
    err = tau_close(got, want)
    print(f"method {method} dw {dw}: {err:.2e}")
    assert err < 1e-11


def test_default_method_is_wavenumber_sweep_and_reads_the_hitran_file(tmp_path, oracle, lib, device):
    band = Band(str(tmp_path), 700.0, 900.0, 0.5, 3000)
    col = syn.profile(5, 9)
    got = run(band, device, col, None, from_file=True)               # optical_depth_method == NULL
    want = band.oracle_tau(oracle, oracle, lib, col, method=WAVENUMBER_SWEEP)
    assert tau_close(got, want) < 1e-11


def test_sweeps_with_no_lines_and_with_sparse_lines(tmp_path, oracle, lib, device):
    col = syn.profile(6, 7)
    empty = Band(str(tmp_path / "a"), 800.0, 900.0, 1.0, 0, mols=[syn.H2O, syn.O3])
    sparse = Band(str(tmp_path / "b"), 1.0, 2000.0, 1.0, 40)         # gaps of tens of bins between lines
    for band in (empty, sparse):
        for method in (WAVENUMBER_SWEEP, LINE_SWEEP):
            got = run(band, device, col, method)
            want = band.oracle_tau(oracle, oracle, lib, col, method=method)
            assert tau_close(got, want) < 1e-11


def test_sweep_objects_work_in_the_batched_pipeline(tmp_path, oracle, lib, device):
    from scenario import MOL_ORDER
    V = 9
    band = Band(str(tmp_path), 500.0, 700.0, 1.0, 2000)
    go, grid = band.gas_optics(device, V, from_file=False, method=WAVENUMBER_SWEEP)
    cols = [syn.profile(c, V) for c in (1, 2, 3)]
    pipe = api.Pipeline(go, None, len(cols), -1, np.full(band.nw, 0.98), None, None)
    gcols, keep = api.make_columns(cols, MOL_ORDER, cfc_order=(0, 1))
    pipe.run(gcols)
    pipe.fluxes(len(cols))
    tau = api.device_to_host(device, pipe.views(0)["tau_gas"], (len(cols), V - 1, band.nw))
    for c, col in enumerate(cols):
        want = band.oracle_tau(oracle, oracle, lib, col, method=WAVENUMBER_SWEEP)
        assert tau_close(tau[c], want) < 1e-11
    pipe.destroy()
    go.destroy()


@pytest.mark.parametrize("method", [WAVENUMBER_SWEEP, LINE_SWEEP])
def test_centres_exactly_on_bin_wavenumbers_and_duplicates(tmp_path, oracle, lib, device, method):
    """The reference's bracket() (kernel_utils.c:26-77) stops at whichever equal element its bisection meets first, so
    with line centres EXACTLY on a bin's interpolation wavenumbers -- first point, midpoint, last point -- and repeated
    centres, which lines count as local or remote depends on the search path.  The wave-parallel searches of the
    kernel must give the path's answer (they fall back to the serial search where an element equals the value)."""
    w0, span, dw = 500.0, 120.0, 0.25
    band = Band(str(tmp_path), w0, w0 + span, dw, 900, mols=[syn.CO2, syn.H2O], line_range=(w0, w0 + span - 26.5))
    for m, ln in band.lines.items():
        v = ln["v0"].copy()
        k = np.arange(v.size)
        v[k % 3 == 0] = np.floor(v[k % 3 == 0])                        # on a bin's first grid point (bins are 1 cm-1 wide)
        v[k % 3 == 1] = np.floor(v[k % 3 == 1]) + 0.375                # on its midpoint: (w + (ppb - 1) dw)/2 with ppb = 5
        v[k % 7 == 2] = np.floor(v[k % 7 == 2]) + 0.75                 # on its last grid point
        order = np.argsort(v, kind="stable")
        for key in ln:
            ln[key] = ln[key][order] if key != "v0" else v[order]
        ln["delta"] = np.zeros_like(ln["delta"])                        # no pressure shift: the centres stay where they are
        assert np.sum(np.diff(ln["v0"]) == 0.0) > 50                    # repeated centres
    col = syn.profile(2, 9)
    got = run(band, device, col, method)
    want = band.oracle_tau(oracle, oracle, lib, col, method=method)
    assert tau_close(got, want) < 1e-11
