#!/usr/bin/env python3
"""Harvest the golden vectors the reference's OWN unit tests hold for the hot path.

Reads the reference's test sources as TEXT (no compilation, no import) and extracts the numeric
input/expected arrays into tests/golden/reference_test_vectors.json:

  gas-optics/test/test_kernels.c      5-layer x 5-line H2O toy problem: inputs, vnn_ref, q_ref (1/Q, pins
                                      TIPS), strength_ref, gamma_ref, alpha_ref, H2O-continuum coefficients
                                      and tau_ref (250 values)
  utilities/test/test_curtis_godson.c + testing_harness/src/circ1.h
                                      CIRC case-1 profile (55 levels) and n/pavg/tavg/ps/ns references
  gas-optics/test/test_tips2017.c     Q(mol, 275.234324 K, iso 1) for H2O, CO2, CH4, N2O, O3

The 50 partition sums among them are also written as tests/golden/tips_pins.csv in grt_tips_load's format.

Only data (numbers) is kept; run where /root/reference is mounted:  python tests/golden/harvest_reference_vectors.py
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
NUM = r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?"


def numbers(text):
    return [float(x) for x in re.findall(NUM, text)]


def c_arrays(src):
    """name -> list of floats for every `name[...] = { ... }` and `.name = { ... }` initialiser (first wins)."""
    out = {}
    for m in re.finditer(r"(?:\.|\b)(\w+)\s*(?:\[[^\]]*\])?\s*=\s*\{([^{}]*)\}", src):
        name, body = m.group(1), m.group(2)
        vals = numbers(body)
        if vals and name not in out:
            out[name] = vals
    return out


def write_tips_pins(vec):
    """tests/golden/tips_pins.csv: every partition sum the reference's tests hold, as a table grt_tips_load
    reads (rows mol_id,iso,T,Q; T ascending per (mol,iso)): Q = 1/q_ref for H2O isotopologues 1-9 at the five
    layer temperatures of test_kernels.c (:180-189, printed to 6 digits) and the five absolute values of
    test_tips2017.c:34-65."""
    ids = dict(H2O=1, CO2=2, O3=3, N2O=4, CH4=6)
    k = vec["test_kernels"]
    rows = []
    niso, temps = k["num_isotopologues"], k["layer_temperature"]
    for i, T in enumerate(temps):
        for iso in range(niso):
            rows.append((1, iso + 1, T, repr(1.0 / k["q_ref"][i * niso + iso])))
    t = vec["test_tips2017"]
    for name, q in t["Q"].items():
        rows.append((ids[name], t["isotopologue"], t["temperature"], repr(q)))
    rows.sort(key=lambda r: (r[0], r[1], r[2]))
    with open(os.path.join(HERE, "tips_pins.csv"), "w") as f:
        f.write("mol_id,iso,T,Q\n")
        for m, iso, T, q in rows:
            f.write("%d,%d,%r,%s\n" % (m, iso, T, q))


def main():
    vec = {}
    src = open(os.path.join(REF, "gas-optics/test/test_kernels.c")).read()
    arr = c_arrays(src)
    mass = re.search(r"\.mass\s*=\s*(" + NUM + r")\s*/\s*(" + NUM + ")", src)
    # the H2O-continuum test declares T0F, CF, T0S, CS, tau_ref in this order inside its function
    ctm = src[src.index("int test_calc_water_vapor_ctm_optical_depth"):src.index("int test_calc_ozone_ctm_optical_depth")]
    ctm_arr = c_arrays(ctm)
    vec["test_kernels"] = {
        "source": "gas-optics/test/test_kernels.c",
        "num_layers": 5, "num_lines": 5, "num_isotopologues": 9, "num_grid_points": 50,
        "level_pressure_atm": arr["level_pressure"], "level_xh2o": arr["level_xh2o"],
        "layer_pressure_atm": arr["pressure"], "layer_temperature": arr["temperature"], "layer_xh2o": arr["xh2o"],
        "center": arr["center"], "delta": arr["delta"], "energy": arr["energy"],
        "gamma_foreign": arr["gamma_foreign"], "gamma_self": arr["gamma_self"],
        "isotopologue": [int(v) for v in arr["isotopologue"]], "n": arr["n"], "strength": arr["strength"],
        "molar_mass": float(mass.group(1)), "avogadro": float(mass.group(2)),
        "vnn_ref": arr["vnn_ref"], "q_ref": arr["q_ref"], "strength_ref": arr["strength_ref"],
        "gamma_ref": arr["gamma_ref"], "alpha_ref": arr["alpha_ref"],
        "ctm_T0F": ctm_arr["T0F"], "ctm_CF": ctm_arr["CF"], "ctm_T0S": ctm_arr["T0S"], "ctm_CS": ctm_arr["CS"],
        "ctm_tau_ref": ctm_arr["tau_ref"],
    }
    circ = c_arrays(open(os.path.join(REF, "testing_harness/src/circ1.h")).read())
    cg_src = open(os.path.join(REF, "utilities/test/test_curtis_godson.c")).read()
    cg = c_arrays(cg_src)
    vec["test_curtis_godson"] = {
        "source": "utilities/test/test_curtis_godson.c + testing_harness/src/circ1.h",
        "mbtoatm": 0.000986923,
        "level_pressure_mb": circ["level_pressure"], "level_temperature": circ["level_temperature"],
        "layer_pressure_mb": circ["layer_pressure"], "H2O_abundance": circ["H2O_abundance"],
        "n_ref": cg["n_ref"], "pavg_ref": cg["pavg_ref"], "tavg_ref": cg["tavg_ref"],
        "ps_ref": cg["ps_ref"], "ns_ref": cg["ns_ref"],
    }
    # CIRC case 1: the only real atmosphere in the reference tree (circ/src/circ1.h:7-24769); the 49 180-point
    # albedo/solar tables are not kept (the CIRC driver accepts a constant albedo: basic-circ-test.c:127-137)
    c1src = open(os.path.join(REF, "circ/src/circ1.h")).read()
    c1 = c_arrays(c1src)
    scal = lambda name: float(re.search(name + r"\s*=\s*(" + NUM + ")", c1src).group(1))
    vec["circ1"] = {
        "source": "circ/src/circ1.h",
        "level_pressure_mb": c1["level_pressure"], "level_temperature": c1["level_temperature"],
        "layer_pressure_mb": c1["layer_pressure"], "layer_temperature": c1["layer_temperature"],
        "surface_temperature": scal("surface_temperature"), "solar_zenith_angle_deg": scal("solar_zenith_angle"),
        "toa_solar_irradiance": scal("toa_solar_irradiance"),
        "abundance": {k.replace("_abundance", ""): c1[k] for k in c1 if k.endswith("_abundance")},
        # LBLRTM broadband reference values quoted by circ/src/basic-circ-test.c:447-495 (sanity magnitudes)
        "lblrtm": {"rlut": 304.27, "rlus": 445.12, "rlds": 288.2, "rsut": 175.0, "rsus": 137.40, "rsdt": 912.79, "rsds": 701.2},
    }
    tips = open(os.path.join(REF, "gas-optics/test/test_tips2017.c")).read()
    temp = float(re.search(r"#define TEMPERATURE\s+(" + NUM + ")", tips).group(1))
    vec["test_tips2017"] = {
        "source": "gas-optics/test/test_tips2017.c", "temperature": temp, "isotopologue": 1,
        "Q": {m.group(1): float(m.group(2)) for m in re.finditer(r"helper\((\w+),\s*(" + NUM + r")\)", tips)},
    }
    write_tips_pins(vec)
    for k, v in vec.items():
        sizes = {a: len(b) for a, b in v.items() if isinstance(b, list)}
        print(k, sizes)
    with open(os.path.join(HERE, "reference_test_vectors.json"), "w") as f:
        json.dump(vec, f, indent=1)


if __name__ == "__main__":
    main()
