"""The clouds library (grtcode_amd/csrc/host/grt_clouds.c -> libclouds.a) against tests/cloud_model.py,
an independent numpy restatement of the reference's clouds/ sources.  Host code on both sides: no GPU needed.  The
reference's own clouds/ cannot be built here (netcdf.h) and has no test vectors: this row's parity is UNPINNED, two
implementations by formula agreeing to rounding is what is shown."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from cloud_model import LibcRand, cloud_optics, synthetic_tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "grtcode_amd", "csrc", "host", "grt_clouds.c")


@pytest.fixture(scope="module")
def clouds(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("clouds") / "libclouds_test.so")
    r = subprocess.run(["gcc", "-std=gnu99", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"),
                        SRC, "-o", so, "-lm"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stderr == "", r.stderr
    lib = C.CDLL(so)
    dp = C.POINTER(C.c_double)
    lib.cloud_optics.argtypes = [dp, C.c_int, C.c_int, dp, dp, dp, dp, C.c_double, dp] + [dp] * 6
    lib.calculate_overlap.argtypes = [C.c_int, dp, C.c_double, dp]
    return lib


def ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def column(L, rng):
    cf = np.where(rng.random(L) < 0.5, rng.random(L), 0.0)
    cf[3] = 1.0                                                     # overcast, and a clear layer
    cf[4] = 0.0
    lwc = np.where(cf > 0, 0.3 * rng.random(L), 0.0)
    iwc = np.where(cf > 0, 0.05 * rng.random(L), 0.0)
    iwc[7] = 0.0 if cf[7] > 0 else iwc[7]
    t = np.linspace(205.0, 290.0, L)
    alt = np.log(100.0 * np.linspace(50.0, 1000.0, L)) * 7.3
    return cf, lwc, iwc, t, alt


@pytest.mark.parametrize("seed,edges", [(1, None), (7, [10.0, 200.0, 200.5, 900.0, 2500.0]), (11, [600.0, 900.0, 1200.0])])
def test_cloud_optics_equals_the_numpy_restatement(tmp_path, clouds, seed, edges):
    """Band edges inside, on and outside the grid: the first band is extended down, the last up, and the grid point a
    band's upper limit falls on belongs to the next band (optics_utils.c:140-166)."""
    paths, tables = synthetic_tables(str(tmp_path), seed=seed, band_edges=edges)
    assert clouds.initialize_clouds_lib(paths["beta"].encode(), paths["ice"].encode(), paths["liquid"].encode()) == 0
    rng = np.random.default_rng(seed)
    L, n = 12, 301
    limits = np.concatenate([[0.0], 0.5 + 10.0 * np.arange(1, n), [10.0 * n]])      # n + 1 band limits, as driver.c:476-492 builds
    cf, lwc, iwc, t, alt = column(L, rng)
    overlap = np.zeros(L - 1)
    assert clouds.calculate_overlap(L, ptr(alt), 2.0, ptr(overlap)) == 0
    assert np.allclose(overlap, np.exp(-1.0 * np.abs(alt[:-1] - alt[1:]) / 2.0), rtol=4e-16, atol=0.0)    # (numpy's exp vs libm's)
    rand = LibcRand()
    for call in range(2):                                            # two calls in a row: longwave pass, shortwave pass
        got = [np.full((L, n), -5.0) for _ in range(6)]              # points no band covers keep the caller's values
        rand.seed(100 + seed + call)
        assert clouds.cloud_optics(ptr(limits), n, L, ptr(cf), ptr(lwc), ptr(iwc), ptr(overlap), 10.0, ptr(t), *[ptr(a) for a in got]) == 0
        rand.seed(100 + seed + call)
        want = cloud_optics(tables, rand, limits[:n], cf, lwc, iwc, overlap, 10.0, t, out=[np.full((L, n), -5.0) for _ in range(6)])
        for a, b, name in zip(got, want, ("beta_liquid", "omega_liquid", "g_liquid", "beta_ice", "omega_ice", "g_ice")):
            assert np.allclose(a, b, rtol=1e-13, atol=0.0), (name, np.abs(a - b).max())
        assert np.all(got[0][4] <= 0.0) and np.any(got[0][3] > 0.0)    # the clear layer has no extinction, the overcast one has
        if edges is not None and edges[0] > limits[0]:
            assert np.all(got[1][3][: 3] == got[1][3][0])               # below the first band: its values
    assert clouds.finalize_clouds_lib() == 0


def test_subcolumns_differ_from_band_to_band_and_repeat_with_the_seed(tmp_path, clouds):
    paths, tables = synthetic_tables(str(tmp_path), seed=2)
    clouds.initialize_clouds_lib(paths["beta"].encode(), paths["ice"].encode(), paths["liquid"].encode())
    L, n = 10, 120
    limits = np.concatenate([[0.0], 12.5 + 25.0 * np.arange(1, n), [25.0 * n]])
    cf, lwc, iwc = np.full(L, 0.5), np.full(L, 0.2), np.full(L, 0.02)
    t, overlap = np.full(L, 250.0), np.full(L - 1, 0.3)
    rand = LibcRand()
    runs = []
    for _ in range(2):
        out = [np.zeros((L, n)) for _ in range(6)]
        rand.seed(42)
        clouds.cloud_optics(ptr(limits), n, L, ptr(cf), ptr(lwc), ptr(iwc), ptr(overlap), 10.0, ptr(t), *[ptr(a) for a in out])
        runs.append(out[0].copy())
    assert np.array_equal(runs[0], runs[1])
    cloudy = runs[0] > 0
    assert 0.2 < cloudy.mean() < 0.8 and len({tuple(c) for c in cloudy.T}) > 1      # half-cloudy layers: bands see different subcolumns
    clouds.finalize_clouds_lib()


def test_a_netcdf_path_is_refused_with_advice(tmp_path, clouds):
    bad = tmp_path / "beta.nc"
    bad.write_bytes(b"CDF\x01" + b"\0" * 64)
    code = ("import ctypes as C; lib = C.CDLL(%r); lib.initialize_clouds_lib(%r, b'x', b'y')" % (clouds._name, str(bad).encode()))
    r = subprocess.run(["python3", "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "netcdf_to_dump.py" in r.stderr


# ---- the sampling half of the row, pinned on the reference's own C (task 7) -------------------------------- #
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libstochastic_ref.so")
STOCHASTIC_FIXTURE = os.path.join(ROOT, "tests", "golden", "stochastic_clouds_ref.json")


class _IncompleteBeta(C.Structure):      # clouds/incomplete_beta.h (never read: the bridge asks our loaded tables)
    _fields_ = [("num_shape", C.c_int), ("num_x", C.c_int), ("p", C.c_void_p), ("q", C.c_void_p), ("x", C.c_void_p),
                ("y", C.c_void_p), ("y_inverse", C.c_void_p)]


class _TotalWaterPDF(C.Structure):       # clouds/stochastic_clouds.h:8-13
    _fields_ = [("beta", C.POINTER(_IncompleteBeta)), ("p", C.c_int), ("q", C.c_int)]


def stochastic_cases():
    """Columns for the sampling comparison: random cloud fractions incl. overcast and clear layers, liquid-only and
    ice-only clouds, thick and thin layers (overlap near 1 and near 0)."""
    cases = []
    for seed in (3, 4, 5, 17, 2024):
        rng = np.random.default_rng(seed)
        L = int(rng.integers(6, 40))
        cf = np.where(rng.random(L) < 0.6, rng.random(L), 0.0)
        cf[0], cf[1] = 1.0, 0.0
        lwc = np.where(cf > 0, 0.4 * rng.random(L), 0.0)
        iwc = np.where(cf > 0, 0.06 * rng.random(L), 0.0)
        lwc[2] = 0.0 if cf[2] > 0 and iwc[2] > 0 else lwc[2]
        iwc[3] = 0.0 if cf[3] > 0 and lwc[3] > 0 else iwc[3]
        alt = np.cumsum(10.0 ** rng.uniform(-2.0, 1.0, L))[::-1].copy()
        cases.append(dict(seed=seed, cf=cf, lwc=lwc, iwc=iwc, alt=alt, scale=float(rng.uniform(0.5, 4.0))))
    return cases


def reference_sampler():
    """oracle/_ref/libstochastic_ref.so: the reference's stochastic_clouds.c + our libclouds.a in one object."""
    ref = C.CDLL(REF_SO)
    dp = C.POINTER(C.c_double)
    ref.overlap_parameter.argtypes = [C.c_int, dp, C.c_double, dp]
    ref.overlap_parameter.restype = None
    ref.sample_condensate.argtypes = [_TotalWaterPDF, C.c_int, dp, dp, dp, dp, dp, dp]
    ref.sample_condensate.restype = None
    ref.grt_clouds_sample_subcolumn.argtypes = [C.c_int, dp, dp, dp, dp, dp, dp]
    ref.calculate_overlap.argtypes = [C.c_int, dp, C.c_double, dp]
    return ref


def run_sampler(lib, rand, case, reference, draws=3):
    """`draws` subcolumns in a row after srand(seed): overlap parameter, then ql, qi of every draw."""
    L = case["cf"].size
    overlap = np.zeros(L - 1)
    if reference:
        lib.overlap_parameter(L, ptr(case["alt"]), case["scale"], ptr(overlap))
        beta = _IncompleteBeta()
        pdf = _TotalWaterPDF(C.pointer(beta), 5, 5)                 # clouds_lib.c: construct_water_pdf(&pdf, 5, 5, &beta)
    else:
        assert lib.calculate_overlap(L, ptr(case["alt"]), case["scale"], ptr(overlap)) == 0
    out = [overlap]
    rand.seed(case["seed"])
    for _ in range(draws):
        ql, qi = np.full(L, -1.0), np.full(L, -1.0)
        if reference:
            lib.sample_condensate(pdf, L, ptr(case["cf"]), ptr(case["lwc"]), ptr(case["iwc"]), ptr(overlap), ptr(ql), ptr(qi))
        else:
            assert lib.grt_clouds_sample_subcolumn(L, ptr(case["cf"]), ptr(case["lwc"]), ptr(case["iwc"]), ptr(overlap), ptr(ql), ptr(qi)) == 0
        out += [ql, qi]
    return out


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libstochastic_ref.so not built (no reference tree on this box)")
def test_sampling_equals_the_reference_c_bit_for_bit(tmp_path):
    """clouds/stochastic_clouds.c compiled where it lies: same srand seed, same rand() call order, same subcolumns -- overlap
    parameter and condensate of three successive draws, every layer, to the last bit."""
    paths, _ = synthetic_tables(str(tmp_path), seed=5)
    ref = reference_sampler()
    assert ref.initialize_clouds_lib(paths["beta"].encode(), paths["ice"].encode(), paths["liquid"].encode()) == 0
    rand = LibcRand()
    for case in stochastic_cases():
        want = run_sampler(ref, rand, case, reference=True)
        got = run_sampler(ref, rand, case, reference=False)
        assert any(np.any(w > 0.0) for w in want[1:])
        for w, g in zip(want, got):
            assert np.array_equal(w, g), (case["seed"], np.abs(w - g).max())
    assert ref.finalize_clouds_lib() == 0


def test_sampling_equals_the_committed_reference_vectors(tmp_path, clouds):
    """The same comparison where the reference tree is absent (the GPU box): tests/golden/stochastic_clouds_ref.json holds
    what oracle/_ref/libstochastic_ref.so returned here (tests/golden/make_golden.py writes it); libc's rand() is the
    image's, the same on both boxes."""
    import json
    fixture = json.load(open(STOCHASTIC_FIXTURE))
    paths, _ = synthetic_tables(str(tmp_path), seed=fixture["tables_seed"])
    clouds.grt_clouds_sample_subcolumn.argtypes = [C.c_int] + [C.POINTER(C.c_double)] * 6
    assert clouds.initialize_clouds_lib(paths["beta"].encode(), paths["ice"].encode(), paths["liquid"].encode()) == 0
    rand = LibcRand()
    cases = stochastic_cases()
    assert len(cases) == len(fixture["cases"])
    for case, rec in zip(cases, fixture["cases"]):
        assert rec["seed"] == case["seed"] and np.array_equal(np.array(rec["cf"]), case["cf"])
        got = run_sampler(clouds, rand, case, reference=False)
        for w, g in zip(rec["out"], got):
            assert np.array_equal(np.array([float.fromhex(h) for h in w]), g), case["seed"]
    assert clouds.finalize_clouds_lib() == 0
