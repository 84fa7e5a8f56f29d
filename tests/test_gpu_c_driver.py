"""§8(f)-1: a C caller of the reference-shaped API, linked against the STATIC archives with plain gcc.

examples/circ_driver.c follows framework/src/driver.c's call sequence for one clear-sky column with the command
line of circ/test/test-basic-circ (one HITRAN file for both bands, -H2O ... -O2, continua, CFCs, CIA, -a albedo).
Here it runs the CIRC case 1 column (the numbers of circ/src/circ1.h, held in tests/golden) on synthetic
spectroscopy and its twelve integrated fluxes are compared with the oracle's for the same files.
"""
import os
import subprocess

import numpy as np
import pytest

from grtcode_amd import synthetic as syn
from scenario import Band
from test_gpu_circ_rfmip import NAME, circ1_column
from test_gpu_pipeline import oracle_column

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "grtcode_amd", "lib")
ARCHIVES = ["-lgrtcode_hip_ext", "-lshortwave", "-llongwave", "-lgas_optics", "-lgrtcode_utilities"]


def build_driver(out):
    cmd = ["gcc", "-std=gnu99", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "circ_driver.c"), "-L" + LIBDIR, *ARCHIVES,
           "-L/opt/rocm/lib", "-lamdhip64", "-lstdc++", "-lm", "-Wl,-rpath,/opt/rocm/lib", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def write_column(path, v):
    rows = [("level_pressure", v["level_pressure_mb"]), ("level_temperature", v["level_temperature"]),
            ("layer_pressure", v["layer_pressure_mb"]), ("layer_temperature", v["layer_temperature"]),
            ("surface_temperature", [v["surface_temperature"]]), ("solar_zenith_angle", [v["solar_zenith_angle_deg"]]),
            ("toa_solar_irradiance", [v["toa_solar_irradiance"]])]
    rows += [(k, v["abundance"][k]) for k in ("H2O", "CO2", "O3", "N2O", "CO", "CH4", "O2", "CFC11", "CFC12")]
    with open(path, "w") as f:
        for name, vals in rows:
            f.write(name + ": " + " ".join(repr(float(x)) for x in vals) + "\n")


def test_c_driver_on_static_archives_matches_oracle(tmp_path, oracle, lib):
    col, v = circ1_column()
    user_level, albedo = 20, 0.196
    # one spectroscopy bundle for both bands, as the reference's driver takes it
    swb = Band(str(tmp_path / "data"), 1.0, 20000.0, 2.0, 20000, sw=True)
    lwb = Band(str(tmp_path / "lw_view"), 1.0, 3250.0, 0.5, 0, sw=True)
    lwb.par, lwb.h2o_dir, lwb.files, lwb.tab = swb.par, swb.h2o_dir, swb.files, swb.tab
    lwb.lines = {}
    for m, ln in swb.lines.items():                     # the loader keeps w0 <= v0 <= wn (parse_HITRAN_file.c:340)
        keep = (ln["v0"] >= lwb.w0) & (ln["v0"] <= lwb.wn)
        lwb.lines[m] = {k: a[keep] for k, a in ln.items()}
    assert sum(a["v0"].size for a in lwb.lines.values()) > 1000
    exe = build_driver(str(tmp_path / "circ_driver"))
    write_column(str(tmp_path / "column.txt"), v)
    args = [exe, swb.par, swb.files["solar"], "-p", str(tmp_path / "column.txt"),
            *("-" + NAME[m] for m in swb.mols), "-h2o-ctm", swb.h2o_dir, "-o3-ctm", swb.files["o3_ctm"],
            "-CFC-11", swb.files["cfc11"], "-CFC-12", swb.files["cfc12"],
            "-N2-N2", swb.files["cia_n2n2"], "-O2-N2", swb.files["cia_o2n2"], "-O2-O2", swb.files["cia_o2o2"],
            "-a", repr(albedo), "-flux-at-level", str(user_level),
            "-w-lw", "1", "-W-lw", "3250", "-r-lw", "0.5", "-w-sw", "1", "-W-sw", "20000", "-r-sw", "2"]
    def fluxes(env_extra):
        r = subprocess.run(args, capture_output=True, text=True, timeout=600, env=dict(os.environ, **env_extra))
        assert r.returncode == 0, r.stderr[-2000:]
        line = [s for s in r.stdout.splitlines() if s.startswith("fluxes:")]
        assert len(line) == 1
        out = np.array([float(x) for x in line[0].split()[1:]])
        assert out.size == 12
        return out
    got = fluxes({"GRT_GAS_OPTICS_FAST": "0"})          # reference operation order: the strict comparison below
    # the same unchanged binary as it runs by default -- the production arithmetic (fast = 3) -- and with the one-pass form
    default = fluxes({})
    fast = fluxes({"GRT_GAS_OPTICS_FAST": "1"})
    assert 0.0 < np.max(np.abs(fast - got)) < 1e-4 and 0.0 < np.max(np.abs(default - got)) < 1e-4
    from grtcode_amd import api
    grid_sw = api.create_spectral_grid(swb.w0, swb.wn, swb.dw)
    solar = api.create_solar_flux(grid_sw, swb.files["solar"])
    emis, alb = np.full(lwb.nw, 1.0 - albedo), np.full(swb.nw, albedo)
    for bi, (band, lw) in enumerate(((lwb, True), (swb, False))):
        w = oracle_column(oracle, lib, band, col, lw, emis, alb, solar, user_level)
        assert np.max(np.abs(got[bi * 6: bi * 6 + 6] - w["integ"])) < 1e-6, (got, w["integ"])
    assert got[1] > 300.0 and got[9] > 0.0


def test_c_driver_reports_reference_style_errors(tmp_path):
    exe = build_driver(str(tmp_path / "circ_driver"))
    v = circ1_column()[1]
    write_column(str(tmp_path / "column.txt"), v)
    r = subprocess.run([exe, str(tmp_path / "missing.par"), str(tmp_path / "missing.csv"), "-p",
                        str(tmp_path / "column.txt"), "-H2O"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "missing.par" in r.stderr
