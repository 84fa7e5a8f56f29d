"""Drop-in evidence at the boundary: GRTCODE's own callers compile UNCHANGED against include/.

The reference sources are read where they lie (build container only; skipped where /root/reference is
absent, e.g. on the GPU box).  Nothing here runs the hot path."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from grtcode_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "framework")), reason="reference tree not mounted")


def syntax_only(src, extra=()):
    cmd = ["gcc", "-std=gnu99", "-fsyntax-only", "-w", f"-I{ROOT}/include", *extra, os.path.join(REF, src)]
    return subprocess.run(cmd, capture_output=True, text=True)


@needs_ref
def test_reference_driver_compiles_unchanged_against_our_headers():
    # framework/src/driver.c: main(), CLI, column loop, optics combination, flux integration.  Its own
    # driver.h / argparse.h / clouds_lib.h come from the reference tree; every hot-path header from include/.
    r = syntax_only("framework/src/driver.c", [f"-I{REF}/framework/src", f"-I{REF}/clouds", f"-I{REF}/utilities/src"])
    assert r.returncode == 0, r.stderr


@needs_ref
@pytest.mark.parametrize("src", ["fortran-bindings/malloc_structs.c", "longwave/test/test_longwave.c",
                                 "shortwave/test/test_shortwave.c", "utilities/test/test_optics.c",
                                 "utilities/test/test_spectral_grid.c", "utilities/test/test_device.c",
                                 "utilities/test/test_utilities.c", "utilities/test/test_parse_csv.c"])
def test_reference_callers_compile_unchanged(src):
    r = syntax_only(src, [f"-I{REF}/testing_harness/src"])
    assert r.returncode == 0, r.stderr


def test_fortran_helper_built_from_reference_source_works_on_host_structs(lib):
    """oracle/_ref/libgrt_fortran_helpers.so = the reference's malloc_structs.c compiled against include/ and
    linked to our library: struct allocation sizes and spectral_grid_properties through the Fortran path."""
    path = os.path.join(ROOT, "oracle", "_ref", "libgrt_fortran_helpers.so")
    if not os.path.exists(path):
        pytest.skip("helper not built (needs /root/reference at build time)")
    h = C.CDLL(path)
    p = C.c_void_p()
    assert h.malloc_struct(C.byref(p), 0) == 0 and p.value          # GRID
    grid = C.cast(p, C.POINTER(api.SpectralGrid))
    assert lib.create_spectral_grid(grid, C.c_double(1.0), C.c_double(3250.0), C.c_double(0.1)) == 0
    w0, n, dw = C.c_double(), C.c_uint64(), C.c_double()
    assert h.spectral_grid_properties(grid, C.byref(w0), C.byref(n), C.byref(dw)) == 0
    assert (w0.value, n.value, dw.value) == (1.0, 32491, 0.1)
    assert h.malloc_struct(C.byref(p), 0) == api.NON_NULL_ERR       # pointer already set (is_null check)
    assert h.free_struct(C.byref(p)) == 0 and not p.value
    assert h.malloc_struct(C.byref(p), 9) == api.VALUE_ERR


@pytest.mark.gpu
def test_fortran_helper_reads_back_device_optics(device):
    path = os.path.join(ROOT, "oracle", "_ref", "libgrt_fortran_helpers.so")
    if not os.path.exists(path):
        pytest.skip("helper not built (needs /root/reference at build time)")
    h = C.CDLL(path)
    grid = api.create_spectral_grid(10.0, 60.0, 0.5)
    o = api.OpticsObject(3, grid, device)
    rng = np.random.default_rng(0)
    tau, om, g = (rng.uniform(0, 1, (3, grid.n)) for _ in range(3))
    o.update(tau, om, g)
    t2, o2, g2 = (np.zeros((3, grid.n)) for _ in range(3))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    assert h.optical_properties(C.byref(o.c), dp(t2), dp(o2), dp(g2)) == 0     # gmemcpy(..., FROM_DEVICE)
    assert np.array_equal(t2, tau) and np.array_equal(o2, om) and np.array_equal(g2, g)
    o.destroy()


def _run_reference_test(tmp_path, name, link):
    exe = str(tmp_path / name)
    src = [os.path.join(REF, "utilities", "test", name + ".c"), os.path.join(REF, "testing_harness", "src", "test_harness.c")]
    r = subprocess.run(["gcc", "-std=gnu99", "-w", "-O1", f"-I{REF}/testing_harness/src", *link["inc"], *src, *link["lib"],
                        "-lm", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120, cwd=str(tmp_path)).stdout
    return {ln.split()[2].rstrip(":"): ln.strip().endswith("passed.") for ln in out.splitlines() if ln.startswith("Running test")}


@needs_ref
@pytest.mark.parametrize("name,ours_only_fails", [("test_parse_csv", set()), ("test_utilities", set()),
                                                  ("test_spectral_grid", {"test_grid_points"})])
def test_reference_unit_tests_run_against_our_library(tmp_path, name, ours_only_fails):
    """The reference's own host-side unit tests, compiled unchanged where they lie, linked once against OUR library and
    once against the reference's utilities sources: the same tests pass and the same (stale) tests fail --
    except where the reference hands out a HOST_ONLY device and this library deliberately refuses to
    (test_grid_points: create_device(NULL) without a GPU)."""
    libdir = os.path.join(ROOT, "grtcode_amd", "lib")
    ours = _run_reference_test(tmp_path, name, dict(inc=[f"-I{ROOT}/include"],
                                                    lib=[f"-L{libdir}", "-lgrtcode_hip", f"-Wl,-rpath,{libdir}"]))
    os.makedirs(tmp_path / "ref", exist_ok=True)
    ref_src = [os.path.join(REF, "utilities", "src", f) for f in ("utilities.c", "verbosity.c", "spectral_grid.c", "device.c",
                                                                  "parse_csv.c")]
    theirs = _run_reference_test(tmp_path / "ref", name, dict(inc=[f"-I{REF}/utilities/src"], lib=ref_src))
    assert set(ours) == set(theirs) and len(ours) >= 7
    differ = {t for t in ours if ours[t] != theirs[t]}
    assert differ == ours_only_fails, (ours, theirs)
    assert sum(ours.values()) >= 7
