"""Partition sums Q(mol, T, iso): the reference's provider (gas-optics/src/tips2017.c) is a missing blob, so what
can be pinned are the 50 values its own tests hold -- gas-optics/test/test_tips2017.c:34-65 (five absolute values at
275.234324 K) and test_kernels.c:180-189 (1/Q for H2O isotopologues 1-9 at five temperatures), committed as
tests/golden/tips_pins.csv by tests/golden/harvest_reference_vectors.py.

  * the built-in model (classical rotor x harmonic oscillators, grt_tips.c) must be within 0.5 % of all of them,
    and within 0.5 % on the ratio Q(T)/Q(296) that alone reaches the optical depths;
  * with the pins loaded as a table (grt_tips_load) the provider reproduces them exactly, 1/Q to the six digits
    test_kernels.c prints;
  * the model announces itself once per molecule on stderr; loading / dropping a table moves the generation that
    makes gas-optics objects re-scale their line strengths.
"""
import ctypes as C
import os

import numpy as np
import pytest

from grtcode_amd import api

HERE = os.path.dirname(os.path.abspath(__file__))
PINS = os.path.join(HERE, "golden", "tips_pins.csv")


@pytest.fixture()
def pins():
    a = np.loadtxt(PINS, delimiter=",", skiprows=1)
    return [(int(m), int(i), float(t), float(q)) for m, i, t, q in a]


@pytest.fixture()
def model(lib):
    api.check(lib.grt_tips_reset())
    yield lib
    api.check(lib.grt_tips_reset())


def test_fixture_holds_the_fifty_reference_values(pins):
    assert len(pins) == 50
    assert sum(1 for m, i, t, q in pins if m == 1 and t != 275.234324) == 45
    assert {m for m, i, t, q in pins if t == 275.234324} == {1, 2, 3, 4, 6}


def test_model_within_half_a_percent_of_every_reference_value(model, pins):
    worst = {}
    for m, i, t, q in pins:
        e = abs(model.Q(m, t, i) / q - 1.0)
        worst[m] = max(worst.get(m, 0.0), e)
        assert e < 5e-3, (m, i, t, model.Q(m, t, i), q)
    # the five absolute TIPS-2017 values (test_tips2017.c:34-65): rotor x oscillators gets them to 0.1 %
    for m, i, t, q in pins:
        if t == 275.234324:
            assert abs(model.Q(m, t, i) / q - 1.0) < 1e-3, (m, model.Q(m, t, i), q)


def test_model_ratio_to_296K_within_half_a_percent(model, pins):
    """Only Q(296)/Q(T) reaches tau.  Reference-side ratios: between pinned temperatures of one isotopologue."""
    by_iso = {}
    for m, i, t, q in pins:
        by_iso.setdefault((m, i), []).append((t, q))
    checked = 0
    for (m, i), tq in by_iso.items():
        for a in range(len(tq)):
            for b in range(a + 1, len(tq)):
                want = tq[a][1] / tq[b][1]
                got = model.Q(m, tq[a][0], i) / model.Q(m, tq[b][0], i)
                assert abs(got / want - 1.0) < 5e-3, (m, i, tq[a][0], tq[b][0], got, want)
                checked += 1
    assert checked >= 90
    # the five molecules of test_tips2017.c against HITRAN's own Q(296) (molparam): ratio within 0.1 %
    q296 = {1: 174.58, 2: 286.09, 3: 3483.71, 4: 4984.90, 6: 590.48}
    for m, i, t, q in pins:
        if t == 275.234324:
            assert abs((model.Q(m, t, 1) / model.Q(m, 296.0, 1)) / (q / q296[m]) - 1.0) < 1e-3


def test_table_of_pins_reproduces_them_exactly(model, pins):
    lib = model
    api.check(lib.grt_tips_load(PINS.encode()))
    assert lib.grt_tips_is_table() == 1
    for m, i, t, q in pins:
        assert lib.Q(m, t, i) == q
        assert lib.grt_tips_source(m, i) == 0
    # 1/Q to the six significant digits test_kernels.c:180-189 prints
    import json
    k = json.load(open(os.path.join(HERE, "golden", "reference_test_vectors.json")))["test_kernels"]
    for li, T in enumerate(k["layer_temperature"]):
        for iso in range(9):
            ref = k["q_ref"][li * 9 + iso]
            assert "%.5e" % (1.0 / lib.Q(1, T, iso + 1)) == "%.5e" % ref
    # between pins: linear in T; outside: clamped; species without a row fall back to the model
    q_lo, q_hi = lib.Q(1, 230.92, 1), lib.Q(1, 236.24, 1)
    assert abs(lib.Q(1, 233.58, 1) - 0.5 * (q_lo + q_hi)) < 1e-9 * q_hi
    assert lib.Q(1, 100.0, 1) == q_lo
    assert lib.grt_tips_source(5, 1) == 1 and lib.grt_tips_source(7, 2) == 1
    api.check(lib.grt_tips_reset())
    assert lib.grt_tips_is_table() == 0 and lib.grt_tips_source(1, 1) == 1


def test_sources_and_isotopologue_dependence(model):
    lib = model
    assert lib.grt_tips_source(1, 1) == 1            # H2O: rotor x oscillators
    assert lib.grt_tips_source(27, 1) == 2           # C2H6: no fundamentals tabulated -> rotor alone
    assert lib.grt_tips_source(0, 1) == -1 and lib.grt_tips_source(1, 19) == -1
    # isotopologues differ (the first surrogate ignored iso): HDO and D2O have their own Q296 and modes
    q = [lib.Q(1, 250.0, i) for i in range(1, 10)]
    assert len({round(v, 6) for v in q}) == 9
    r = [lib.Q(1, 250.0, i) / lib.Q(1, 296.0, i) for i in (1, 4, 7)]
    assert r[0] != r[1] != r[2]
    assert lib.Q(2, 250.0, 2) / lib.Q(2, 296.0, 2) != lib.Q(2, 250.0, 1) / lib.Q(2, 296.0, 1)


def test_model_factors_are_memoised_without_changing_a_bit(model):
    """grt_tips.c keeps the model's two costly factors per (mode set, T) in a 2 048-entry table (a column asks for every
    isotopologue of a molecule at each layer's temperature, once per band).  Whatever the order of the calls, and after
    the entries have been pushed out by thousands of other temperatures, Q returns the same doubles."""
    lib = model
    rng = np.random.default_rng(11)
    temps = rng.uniform(150.0, 350.0, 120)
    cases = [(m, i, float(t)) for m in (1, 2, 3, 4, 5, 6, 7) for i in (1, 2, 3) for t in temps]
    first = np.array([lib.Q(m, t, i) for m, i, t in cases])
    order = rng.permutation(len(cases))
    again = np.empty_like(first)
    for k in order:
        m, i, t = cases[k]
        again[k] = lib.Q(m, t, i)
    assert np.array_equal(first, again)
    for t in rng.uniform(150.0, 350.0, 6000):          # evict everything, several times over
        lib.Q(int(rng.integers(1, 8)), float(t), 1)
    third = np.array([lib.Q(m, t, i) for m, i, t in cases])
    assert np.array_equal(first, third)
    # the factors do not depend on the isotopologue where the molecule has one mode set: the ratio to 296 K is shared
    r = [lib.Q(2, 231.7, i) / lib.Q(2, 296.0, i) for i in (1, 2)]
    assert np.isfinite(r).all()


def test_rescale_follows_the_current_provider(model, pins, tmp_path):
    """grt_rescale_strengths (parse_HITRAN_file.c:372-384) uses Q(296) of whatever provider is current: a table loaded
    after add_molecule must not be mixed with strengths scaled by the model -- objects therefore keep raw
    strengths and re-scale when the generation moves."""
    lib = model
    n = 4
    iso = np.array([1, 2, 1, 2], dtype=np.uint8)
    v0 = np.array([500.0, 600.0, 700.0, 800.0])
    en = np.array([100.0, 200.0, 300.0, 400.0], dtype=np.float32)

    def scaled():
        s = np.ones(n)
        lib.grt_rescale_strengths(2, C.c_uint64(n), iso.ctypes.data_as(C.POINTER(C.c_uint8)), v0.ctypes.data_as(api.c_double_p),
                                  en.ctypes.data_as(C.POINTER(C.c_float)), s.ctypes.data_as(api.c_double_p))
        return s
    lib.grt_tips_generation.restype = C.c_ulong
    g0 = lib.grt_tips_generation()
    s_model = scaled()
    table = tmp_path / "q.csv"
    table.write_text("mol_id,iso,T,Q\n2,1,200,100\n2,1,300,200\n2,2,200,1000\n2,2,300,3000\n")
    api.check(lib.grt_tips_load(str(table).encode()))
    assert lib.grt_tips_generation() > g0
    s_table = scaled()
    c2, tref = np.float64(np.float32(-1.4387686)), 296.0
    want = np.array([196.0 if i == 1 else 2920.0 for i in iso]) / (np.exp(c2 * en.astype(np.float64) / tref) * (1.0 - np.exp(c2 * v0 / tref)))
    assert np.allclose(s_table, want, rtol=1e-14)
    assert not np.allclose(s_table, s_model, rtol=1e-3)
    g1 = lib.grt_tips_generation()
    api.check(lib.grt_tips_reset())
    assert lib.grt_tips_generation() > g1
    assert np.array_equal(scaled(), s_model)


def test_bad_tables_are_refused(model, tmp_path):
    lib = model
    for text in ("mol_id,iso,T\n1,1,200\n", "mol_id,iso,T,Q\n1,1,300,10\n1,1,200,20\n", "mol_id,iso,T,Q\n99,1,200,5\n",
                 "mol_id,iso,T,Q\n1,1,200,-5\n"):
        p = tmp_path / "bad.csv"
        p.write_text(text)
        assert lib.grt_tips_load(str(p).encode()) == api.VALUE_ERR
        assert lib.grt_tips_is_table() == 0


def test_model_warns_once_per_molecule(tmp_path):
    """A run on the built-in model differs from a tips2017.c run by more than the flux contract: it must say so
    (ADVICE r1), once per molecule, on stderr whatever the verbosity; GRT_TIPS_QUIET=1 silences it."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from grtcode_amd import api\n"
            "lib = api.load_library()\n"
            "for _ in range(3): lib.Q(2, 250.0, 1); lib.Q(2, 260.0, 2); lib.Q(27, 250.0, 1)\n" % os.path.dirname(HERE))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env={**os.environ, "GRT_TIPS_QUIET": "0"})
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stderr.splitlines() if "partition sums" in l]
    assert len(lines) == 2
    assert "molecule 2 " in lines[0] and "harmonic-oscillator" in lines[0] and "grt_tips_load" in lines[0]
    assert "molecule 27 " in lines[1] and "rigid-rotor" in lines[1] and "percent" in lines[1]
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env={**os.environ, "GRT_TIPS_QUIET": "1"})
    assert "partition sums" not in r.stderr
