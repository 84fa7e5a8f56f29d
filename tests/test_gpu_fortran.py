"""The Fortran path of the boundary: examples/fortran_gas_optics.f90 `use`s the reference's own binding module
(fortran-bindings/grtcode_fortran.F90) -- compiled UNCHANGED with amdflang -- and its C helper (malloc_structs.c,
compiled UNCHANGED against include/), linked to this library.  The executable is built where the reference tree is
mounted (`make -C oracle ref` -> oracle/_ref/fortran_gas_optics) and travels to the GPU box with the snapshot."""
import os
import subprocess

import numpy as np
import pytest

from grtcode_amd import synthetic as syn
from scenario import Band

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "fortran_gas_optics")


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/fortran_gas_optics not built (needs the reference tree + amdflang)")
def test_fortran_program_on_the_reference_binding_module(tmp_path, oracle, lib):
    V = 9
    k = np.arange(V) / (V - 1)
    col = dict(p=1.0 + 1012.25 * k ** 2, t=210.0 + 80.0 * k,
               ppmv={syn.H2O: 5.0 * 3000.0 ** k, syn.CO2: np.full(V, 400.0)})
    band = Band(str(tmp_path), 600.0, 900.0, 0.5, 3000, mols=[syn.H2O, syn.CO2], with_cfc=False, with_cia=False)
    r = subprocess.run([EXE, band.par, "600", "900", "0.5", band.h2o_dir, band.files["o3_ctm"]],
                       capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, GRT_GAS_OPTICS_FAST="0"))       # reference operation order: 1e-11 below
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    head = [ln for ln in r.stdout.splitlines() if ln.startswith("grid:")][0]
    assert f"n={band.nw} " in head and "molecules=2" in head
    got = np.array([[float(x) for x in ln.split()[2:]] for ln in r.stdout.splitlines() if ln.startswith("layer ")])
    assert got.shape == (V - 1, 3)
    tau_gas = band.oracle_tau(oracle, oracle, lib, col)
    tau_r, om_r, g_r = oracle.rayleigh(V - 1, col["p"], band.w0, band.dw, band.nw)
    want = np.stack([tau_gas.sum(axis=1), (tau_gas + tau_r).sum(axis=1), tau_r.sum(axis=1)], axis=1)
    assert np.max(np.abs(got - want) / want) < 1e-11
