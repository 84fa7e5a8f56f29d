"""Partition sums and isotopologues on the device (VERDICT r1 #1e, #3).

kernels.c:52-66 fills q[layer][iso-1] = 1/Q(mol, T_layer, iso); kernels.c:85 multiplies every line's strength by
q[layer][iso-1]; parse_HITRAN_file.c:382 has put Q(mol, 296, iso) into the strength at load.  Here the oracle is fed
an INDEPENDENT evaluation of Q (numpy interpolation of the same table the product reads through grt_tips_load), so a
wrong isotopologue lookup, a stale Q(296) or a table that is not picked up fails these tests.
"""
import json
import os

import numpy as np
import pytest

from grtcode_amd import api, synthetic as syn
from scenario import Band, MOLTAB, RUN_TO_RUN_FUSED

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
PINS = os.path.join(HERE, "golden", "tips_pins.csv")


def tau_close(got, want):
    return np.max(np.abs(got - want) / np.maximum(np.abs(want).max(axis=1, keepdims=True), 1e-300))


@pytest.fixture()
def clean_tips(lib):
    api.check(lib.grt_tips_reset())
    yield lib
    api.check(lib.grt_tips_reset())


def write_table(path, mols, seed=5):
    """A table with a DISTINCT Q(T) curve for every isotopologue: rows mol,iso,T,Q.  Returns qfunc(mol, T, iso)."""
    rng = np.random.default_rng(seed)
    temps = np.arange(150.0, 351.0, 12.5)
    curves = {}
    with open(path, "w") as f:
        f.write("mol_id,iso,T,Q\n")
        for m in mols:
            for iso in range(1, MOLTAB[m][1] + 1):
                q = (50.0 + 400.0 * rng.random()) * iso * (temps / 296.0) ** (1.0 + rng.random()) * (1.0 + 0.1 * rng.random(temps.size).cumsum() / temps.size)
                q = np.array([float("%.9e" % v) for v in q])
                curves[(m, iso)] = q
                for T, v in zip(temps, q):
                    f.write("%d,%d,%.1f,%.9e\n" % (m, iso, T, v))
    return lambda mol, T, iso: float(np.interp(T, temps, curves[(mol, iso)]))


def test_partition_functions_on_device_reproduce_test_kernels_c(tmp_path, clean_tips, device):
    """With the reference-held values loaded, 1/Q as the device holds it equals q_ref of test_kernels.c:180-189 to the
    six digits that file prints -- through the product's calc_partition_functions path (host prologue -> column state
    -> HBM)."""
    lib = clean_tips
    api.check(lib.grt_tips_load(PINS.encode()))
    k = json.load(open(os.path.join(HERE, "golden", "reference_test_vectors.json")))["test_kernels"]
    T_layer = np.array(k["layer_temperature"])
    t = np.zeros(6)
    t[0] = 228.0
    for i in range(5):
        t[i + 1] = 2.0 * T_layer[i] - t[i]           # layer means (curtis_godson.c:67-68) land on the reference's temperatures
    band = Band(str(tmp_path), 500.0, 560.0, 1.0, 200, mols=[syn.H2O], with_cfc=False, with_cia=False, with_ctm=False)
    go, grid = band.gas_optics(device, 6)
    col = syn.profile(0, 6)
    band.set_column(go, col)
    q = go.debug_partition_functions(col["p"], t)
    assert q.shape == (1, 5, 18)
    ref = np.array(k["q_ref"]).reshape(5, 9)
    for i in range(5):
        for iso in range(9):
            assert "%.5e" % q[0, i, iso] == "%.5e" % ref[i, iso], (i, iso, q[0, i, iso], ref[i, iso])
    assert np.all(q[0, :, 9:] == 0.0)
    go.destroy()


@pytest.mark.parametrize("fast,from_file", [(0, True), (0, False), (3, False), (1, True), (2, False)])
def test_isotopologue_lines_match_oracle_with_independent_Q(tmp_path, clean_tips, oracle, device, fast, from_file):
    lib = clean_tips
    mols = [syn.CO2, syn.O3, syn.H2O]
    qfunc = write_table(str(tmp_path / "q.csv"), mols)
    api.check(lib.grt_tips_load(str(tmp_path / "q.csv").encode()))
    band = Band(str(tmp_path / "b"), 600.0, 800.0, 1.0, 3000, mols=mols, with_cfc=False, with_cia=False, iso_mix=False)
    wanted = {syn.CO2: [1, 2, 3, 10, 11, 12], syn.O3: [1, 5, 9, 10, 17, 18], syn.H2O: [1, 4, 7, 9]}
    for m in mols:
        n = band.lines[m]["v0"].size
        band.lines[m]["iso"] = np.array([wanted[m][j % len(wanted[m])] for j in range(n)], dtype=np.int32)
    syn.write_hitran_par(band.par, band.lines)         # '0' -> 10, 'A' -> 11, 'B' -> 12, ... (parse_HITRAN_file.c:177-194)
    col = syn.profile(4, 15)
    go, grid = band.gas_optics(device, 15, from_file=from_file)
    go.tune(fast=fast)
    band.set_column(go, col)
    opt = api.OpticsObject(14, grid, device)
    go.calculate_optical_depth(col["p"], col["t"], opt)
    tau = opt.read()[0]
    want = band.oracle_tau(oracle, oracle, lib, col, qfunc=qfunc)
    assert tau_close(tau, want) < (1e-11 if fast == 0 else 2e-6)
    # the isotopologue matters: with every line treated as the principal one the result is far off
    band_wrong = Band.__new__(Band)
    band_wrong.__dict__.update(band.__dict__)
    band_wrong.lines = {m: dict(v, iso=np.ones_like(v["iso"])) for m, v in band.lines.items()}
    wrong = band_wrong.oracle_tau(oracle, oracle, lib, col, qfunc=qfunc)
    assert tau_close(wrong, want) > 1e-2
    # the device's q block is the table's
    q = go.debug_partition_functions(col["p"], col["t"])
    tl = 0.5 * (col["t"][:-1] + col["t"][1:])
    for s, m in enumerate(mols):
        for iso in wanted[m]:
            got = q[s, :, iso - 1]
            exp = np.array([1.0 / qfunc(m, T, iso) for T in tl])
            assert np.max(np.abs(got / exp - 1.0)) < 1e-14
    opt.destroy()
    go.destroy()


@pytest.mark.parametrize("fast", [0, 3])
def test_table_loaded_after_add_molecule_rescales_the_store(tmp_path, clean_tips, oracle, device, fast):
    """strengths used to be scaled by Q(296) at add_molecule; a table loaded afterwards mixed the model's
    Q(296) with the table's Q(T).  The store now keeps tabulated strengths and is re-scaled when the provider changes."""
    lib = clean_tips
    mols = [syn.CO2, syn.H2O]
    band = Band(str(tmp_path / "b"), 650.0, 750.0, 0.5, 1500, mols=mols, with_cfc=False, with_cia=False)
    col = syn.profile(2, 9)
    go, grid = band.gas_optics(device, 9)             # lines added while the built-in model is the provider
    go.tune(fast=fast)
    band.set_column(go, col)
    opt = api.OpticsObject(8, grid, device)
    go.calculate_optical_depth(col["p"], col["t"], opt)
    tau_model = opt.read()[0]
    store_model = go.debug_line_strengths()
    tol = 1e-11 if fast == 0 else 2e-6
    assert tau_close(tau_model, band.oracle_tau(oracle, oracle, lib, col)) < tol
    qfunc = write_table(str(tmp_path / "q.csv"), mols, seed=9)
    api.check(lib.grt_tips_load(str(tmp_path / "q.csv").encode()))
    go.calculate_optical_depth(col["p"], col["t"], opt)
    tau_table = opt.read()[0]
    want = band.oracle_tau(oracle, oracle, lib, col, qfunc=qfunc)
    assert tau_close(tau_table, want) < tol
    assert tau_close(tau_table, tau_model) > 1e-2
    store_table = go.debug_line_strengths()
    assert store_table.size == store_model.size and np.any(store_table != store_model)
    api.check(lib.grt_tips_reset())
    go.calculate_optical_depth(col["p"], col["t"], opt)
    # What "restored" means exactly is the line store: the strengths are again, bit for bit, the ones the model gave.
    assert np.array_equal(go.debug_line_strengths(), store_model)
    # tau from the same store: the reference-order form sums in fp64 only (line slices add in any order: ~1e-15); the fused
    # forms add fp32 cell moments and fp64 near fields with atomics in the scheduler's order, which moves tau by ~1e-11 of a
    # layer's largest value from run to run (RUN_TO_RUN_FUSED; bit-identical under GRT_DETERMINISTIC=1, test_gpu_deterministic.py)
    assert tau_close(opt.read()[0], tau_model) < (1e-13 if fast == 0 else RUN_TO_RUN_FUSED)
    opt.destroy()
    go.destroy()
