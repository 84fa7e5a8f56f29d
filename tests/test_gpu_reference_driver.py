"""SURVEY §8(f)-1 / VERDICT r1 #4: the reference's OWN framework/src/driver.c -- main(), command line, column loop,
optics combination, flux integration -- compiled unchanged where it lies, linked against this library, run on the GPU.

oracle/_ref/grtcode_driver = driver.c + the reference's utilities/src/argparse.c (both unchanged) + examples/driver_app.c
(ours: the five driver.h:165-203 callbacks on flat text, CIRC column semantics of circ/src/basic-circ-test.c) +
libclouds.a (entry points only) + libgrtcode_hip.so; built by oracle/Makefile in the container, shipped prebuilt.

CIRC case 1 (circ/src/circ1.h numbers from tests/golden) plus a perturbed second column, synthetic spectroscopy:
the fluxes main() writes must equal the oracle's for the same files -- <= 1e-3 W m-2 (the north star's tolerance) as an
unchanged caller runs by default (production arithmetic, fast = 3), <= 1e-6 W m-2 with GRT_GAS_OPTICS_FAST=0 in the
environment (the reference's operation order).
"""
import copy
import os
import subprocess

import numpy as np
import pytest

from grtcode_amd import api, synthetic as syn
from scenario import Band
from test_gpu_c_driver import write_column
from test_gpu_circ_rfmip import NAME, circ1_column
from test_gpu_pipeline import oracle_column

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "grtcode_driver")
needs_driver = pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref/grtcode_driver not built (needs /root/reference at build time)")


def second_column(v):
    w = copy.deepcopy(v)
    w["level_temperature"] = [t + 3.0 for t in v["level_temperature"]]
    w["layer_temperature"] = [t + 3.0 for t in v["layer_temperature"]]
    w["surface_temperature"] = v["surface_temperature"] + 2.0
    w["solar_zenith_angle_deg"] = 30.0
    w["abundance"] = dict(v["abundance"], H2O=[0.7 * x for x in v["abundance"]["H2O"]])
    return w


def as_column(v):
    """The column the driver builds from a dump (basic-circ-test.c semantics) as the checker wants it."""
    p, pl = np.array(v["level_pressure_mb"]), np.array(v["layer_pressure_mb"])
    L = pl.size

    def to_levels(ab):
        ab = np.array(ab)
        out = np.zeros(L + 1)
        out[0], out[L] = ab[0] * 1e6, ab[L - 1] * 1e6
        for i in range(1, L):
            out[i] = (ab[i - 1] + (ab[i] - ab[i - 1]) * (p[i] - pl[i - 1]) / (pl[i] - pl[i - 1])) * 1e6
        return out
    from scenario import MOL_ORDER
    ppmv = {m: to_levels(v["abundance"][NAME[m]]) for m in MOL_ORDER}
    ppmv[syn.N2] = np.full(L + 1, 0.781e6)
    mu0 = float(np.cos(2.0 * np.pi * v["solar_zenith_angle_deg"] / 360.0))
    return dict(p=p, t=np.array(v["level_temperature"]), t_layer=np.array(v["layer_temperature"]),
                t_surf=v["surface_temperature"], ppmv=ppmv, mu0=mu0, tsi=v["toa_solar_irradiance"] / mu0,
                cfc_ppmv={0: to_levels(v["abundance"]["CFC11"]), 1: to_levels(v["abundance"]["CFC12"])})


def parse_output(path):
    out = {}
    for line in open(path):
        if line.startswith("#"):
            continue
        t, c, name, count, *vals = line.split()
        assert int(count) == len(vals)
        out[(int(c), name)] = np.array([float(x) for x in vals])
    return out


@needs_driver
def test_reference_driver_without_gpu_fails_cleanly(tmp_path):
    """No CPU fallback: where no GPU exists the unchanged driver stops at create_device with the library's error text."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    (tmp_path / "c.txt").write_text("column:\nlevel_pressure: 1 500 1000\nlevel_temperature: 220 250 288\nlayer_pressure: 250 750\n"
                                    "layer_temperature: 235 270\nsurface_temperature: 289\nsolar_zenith_angle: 40\ntoa_solar_irradiance: 900\n")
    r = subprocess.run([DRIVER, "none.par", "none.csv", str(tmp_path / "c.txt"), "-o", str(tmp_path / "o.txt")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "Error" in r.stderr


@needs_driver
@pytest.mark.gpu
def test_unchanged_reference_driver_main_matches_oracle(tmp_path, oracle, lib):
    col1, v1 = circ1_column()
    v2 = second_column(v1)
    cols = [col1, as_column(v2)]
    user_level, albedo, emissivity = 20, 0.196, 0.97
    swb = Band(str(tmp_path / "data"), 1.0, 20000.0, 2.0, 20000, sw=True)
    lwb = Band(str(tmp_path / "lw_view"), 1.0, 3250.0, 0.5, 0, sw=True)
    lwb.par, lwb.h2o_dir, lwb.files, lwb.tab = swb.par, swb.h2o_dir, swb.files, swb.tab
    lwb.lines = {}
    for m, ln in swb.lines.items():                     # the loader keeps w0 <= v0 <= wn (parse_HITRAN_file.c:340)
        keep = (ln["v0"] >= lwb.w0) & (ln["v0"] <= lwb.wn)
        lwb.lines[m] = {k: a[keep] for k, a in ln.items()}
    dump = str(tmp_path / "columns.txt")
    with open(dump, "w") as f:
        for k, v in enumerate((v1, v2)):
            f.write("column: %d\n" % k)
            write_column(str(tmp_path / "one.txt"), v)
            f.write(open(str(tmp_path / "one.txt")).read())
    base = [DRIVER, swb.par, swb.files["solar"], dump, *("-" + NAME[m] for m in swb.mols),
            "-h2o-ctm", swb.h2o_dir, "-o3-ctm", swb.files["o3_ctm"], "-CFC-11", swb.files["cfc11"], "-CFC-12", swb.files["cfc12"],
            "-N2-N2", swb.files["cia_n2n2"], "-O2-N2", swb.files["cia_o2n2"], "-O2-O2", swb.files["cia_o2o2"],
            "-a", repr(albedo), "-e", repr(emissivity), "-flux-at-level", str(user_level + 1),
            "-w-lw", "1", "-W-lw", "3250", "-r-lw", "0.5", "-w-sw", "1", "-W-sw", "20000", "-r-sw", "2"]

    def run(extra, env_extra, name):
        out = str(tmp_path / name)
        r = subprocess.run(base + extra + ["-o", out], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, **env_extra))
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
        return parse_output(out)

    grid_sw = api.create_spectral_grid(swb.w0, swb.wn, swb.dw)
    solar = api.create_solar_flux(grid_sw, swb.files["solar"])
    emis, alb = np.full(lwb.nw, emissivity), np.full(swb.nw, albedo)
    want = [(oracle_column(oracle, lib, lwb, c, True, emis, alb, solar, user_level),
             oracle_column(oracle, lib, swb, c, False, emis, alb, solar, user_level)) for c in cols]
    names = (("rlutcsaf", 0, 0), ("rluscsaf", 0, 1), ("rlucsaf_user_level", 0, 2), ("rldscsaf", 0, 4), ("rldcsaf_user_level", 0, 5),
             ("rsutcsaf", 1, 0), ("rsuscsaf", 1, 1), ("rsucsaf_user_level", 1, 2), ("rsdtcsaf", 1, 3), ("rsdscsaf", 1, 4),
             ("rsdcsaf_user_level", 1, 5))
    # GRT_GAS_OPTICS_FAST=0: the reference's operation order; {}: what an unchanged driver gets by default, the production
    # arithmetic (fast = 3), inside the north star's 1e-3 W m-2
    strict = {"GRT_GAS_OPTICS_FAST": "0"}
    for env, tol in ((strict, 1e-6), ({}, 1e-3), ({"GRT_GAS_OPTICS_FAST": "3"}, 1e-3)):
        got = run(["-integrated"], env, "integrated.txt")
        worst = 0.0
        for c in range(2):
            for name, band, k in names:
                assert got[(c, name)].size == 1
                worst = max(worst, abs(got[(c, name)][0] - want[c][band]["integ"][k]))
            assert np.array_equal(got[(c, "level_pressure")], cols[c]["p"])
            assert got[(c, "surface_temperature")][0] == cols[c]["t_surf"]
            assert np.allclose(got[(c, "h2o_vmr")], cols[c]["ppmv"][syn.H2O], rtol=1e-14)
        print(f"reference driver.c main(), {env or 'default (production arithmetic)'}: worst integrated flux difference {worst:.2e} W m-2")
        assert worst < tol
    # the driver's aerosol pass (driver.c:426-472, add_optics of THREE objects): the reference's own aerosol optics are
    # zero (the body of calculate_aerosol_optics is commented out, driver.c:223-238), so its clear-sky fluxes must equal
    # the clear-clean-sky ones of the same run
    got = run(["-integrated", "-aerosols"], strict, "aerosols.txt")
    for c in range(2):
        for name, band, k in names:
            with_aerosols = got[(c, name.replace("csaf", "cs"))]
            assert with_aerosols.size == 1 and with_aerosols[0] == got[(c, name)][0], (c, name)
            assert abs(got[(c, name)][0] - want[c][band]["integ"][k]) < 1e-6
    # one column only (-x/-X as run-rfmip-irf.sh shards: GRTworkflow/run-rfmip-irf.sh:121-122), spectral output
    got = run(["-x", "1", "-X", "1"], strict, "spectral.txt")
    w = want[1]
    assert got[(0, "rlutcsaf")].size == lwb.nw and got[(0, "rsdscsaf")].size == swb.nw
    assert np.max(np.abs(got[(0, "rlutcsaf")] - w[0]["up"][0])) < 1e-10 * np.abs(w[0]["up"]).max()
    assert np.max(np.abs(got[(0, "rldscsaf")] - w[0]["dn"][-1])) < 1e-10 * np.abs(w[0]["dn"]).max()
    assert np.max(np.abs(got[(0, "rsdscsaf")] - w[1]["dn"][-1])) < 1e-10 * np.abs(w[1]["dn"]).max()
    assert np.max(np.abs(got[(0, "rsutcsaf")] - w[1]["up"][0])) < 1e-10 * np.abs(w[1]["dn"]).max()
    assert np.max(np.abs(got[(0, "rlucsaf_user_level")] - w[0]["up"][user_level])) < 1e-10 * np.abs(w[0]["up"]).max()


CLOUDS = os.path.join(ROOT, "oracle", "_ref", "grtcode_driver_clouds_double")


def cloud_double(band_limits, L, cf, lwc, iwc, r_liq, t_layer):
    """tests/support/clouds_double.c in numpy: extinction [1/m], single-scattering albedo, asymmetry per (layer, band)."""
    w = 0.5 * (band_limits[:-1] + band_limits[1:])[None, :]
    r_ice = np.where(t_layer > 250.0, 50.0, 25.0)[:, None]
    cf, lwc, iwc = cf[:, None], lwc[:, None], iwc[:, None]
    liq = (cf * lwc * 1.5e-3 / r_liq * (1.0 + 0.2 * np.exp(-w / 3000.0)), 0.5 + 0.499 * (1.0 - np.exp(-w / 2500.0)) + 0 * cf,
           0.80 + 0.07 * np.exp(-w / 8000.0) + 0 * cf)
    ice = (cf * iwc * 1.2e-3 / r_ice * (1.0 + 0.1 * np.exp(-w / 5000.0)), 0.45 + 0.5 * (1.0 - np.exp(-w / 3500.0)) + 0 * cf,
           0.75 + 0.1 * np.exp(-w / 10000.0) + 0 * cf)
    return liq, ice


@pytest.mark.skipif(not os.path.exists(CLOUDS), reason="oracle/_ref/grtcode_driver_clouds_double not built (needs /root/reference at build time)")
@pytest.mark.gpu
def test_reference_driver_cloud_pass_through_the_library(tmp_path, oracle, lib):
    """SURVEY §8(f)-4, the part that can exist without the reference's netCDF cloud tables: the UNCHANGED driver's cloud
    pass (driver.c:474-597) -- band limits, cloud_optics() filling the liquid / ice Optics_t arrays IN PLACE on the host,
    the thickness scaling of :514-525, add_optics of FOUR objects, the all-sky flux outputs -- runs through this library
    when create_optics hands out host-visible arrays (GRT_OPTICS_HOST_VISIBLE=1).  The clouds library itself is a test
    double with closed-form optics (tests/support/clouds_double.c); the oracle gets the same optics from numpy."""
    col1, v1 = circ1_column()
    L = col1["p"].size - 1
    cf, lwc, iwc = np.zeros(L), np.zeros(L), np.zeros(L)
    cf[38:44], lwc[38:44] = 0.6, 0.2                 # a liquid layer low down (T > 250 K) ...
    cf[14:19], iwc[14:19] = 0.4, 0.05                # ... and an ice layer aloft
    swb = Band(str(tmp_path / "data"), 1.0, 8000.0, 2.0, 8000, sw=True)
    lwb = Band(str(tmp_path / "lw_view"), 1.0, 2000.0, 1.0, 0, sw=True)
    lwb.par, lwb.h2o_dir, lwb.files, lwb.tab = swb.par, swb.h2o_dir, swb.files, swb.tab
    lwb.lines = {m: {k: a[(ln["v0"] >= lwb.w0) & (ln["v0"] <= lwb.wn)] for k, a in ln.items()} for m, ln in swb.lines.items()}
    dump = str(tmp_path / "columns.txt")
    write_column(dump, v1)
    with open(dump, "a") as f:
        for name, vals in (("cloud_fraction", cf), ("liquid_water_content", lwc), ("ice_water_content", iwc)):
            f.write(name + ": " + " ".join(repr(float(x)) for x in vals) + "\n")
    user_level, albedo, emissivity = 20, 0.196, 0.97
    out = str(tmp_path / "cloudy.txt")
    cmd = [CLOUDS, swb.par, swb.files["solar"], dump, *("-" + NAME[m] for m in swb.mols),
           "-h2o-ctm", swb.h2o_dir, "-o3-ctm", swb.files["o3_ctm"], "-CFC-11", swb.files["cfc11"], "-CFC-12", swb.files["cfc12"],
           "-N2-N2", swb.files["cia_n2n2"], "-O2-N2", swb.files["cia_o2n2"], "-O2-O2", swb.files["cia_o2o2"],
           "-a", repr(albedo), "-e", repr(emissivity), "-flux-at-level", str(user_level + 1), "-integrated", "-clouds",
           "-beta-path", "beta.nc", "-ice-path", "ice.nc", "-liquid-path", "liquid.nc",
           "-w-lw", "1", "-W-lw", "2000", "-r-lw", "1", "-w-sw", "1", "-W-sw", "8000", "-r-sw", "2", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, GRT_OPTICS_HOST_VISIBLE="1", GRT_GAS_OPTICS_FAST="0"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "clouds_double: initialised (beta.nc, ice.nc, liquid.nc)" in r.stderr
    got = parse_output(out)
    grid_sw = api.create_spectral_grid(swb.w0, swb.wn, swb.dw)
    solar = api.create_solar_flux(grid_sw, swb.files["solar"])
    # layer thicknesses as the application computes them (basic-circ-test.c:155-166)
    thick = np.abs(np.log(col1["p"][:-1]) - np.log(col1["p"][1:])) * col1["t_layer"] * 8.314462 / (28.9647 * 0.001 * 9.81)
    worst_cloudy = worst_clear = 0.0
    for band, lw, names in ((lwb, True, ("rlut", "rlus", "rlu", None, "rlds", "rld")), (swb, False, ("rsut", "rsus", "rsu", "rsdt", "rsds", "rsd"))):
        emis, alb = np.full(band.nw, emissivity), np.full(band.nw, albedo)
        clear = oracle_column(oracle, lib, band, col1, lw, emis, alb, solar, user_level)
        centres = band.w0 + np.arange(band.nw) * band.dw                      # driver.c:476-492
        limits = np.empty(band.nw + 1)
        limits[1:-1] = 0.5 * (centres[:-1] + centres[1:])
        limits[0] = max(centres[0] - band.dw, 0.0)
        limits[-1] = centres[-1] + band.dw
        (bl, ol, gl), (bi, oi, gi) = cloud_double(limits, L, cf, lwc, iwc, 10.0, col1["t_layer"])
        tl, ti = bl * thick[:, None], bi * thick[:, None]                      # driver.c:514-525
        tr, om_r, g_r = oracle.rayleigh(L, col1["p"], band.w0, band.dw, band.nw)
        z = np.zeros_like(tr)
        tau, omega, g = oracle.add_optics([clear["tau_gas"], tr, tl, ti], [z, om_r, ol, oi], [z, g_r, gl, gi])
        up, dn = (oracle.lw_fluxes(band.w0, band.dw, col1["t_surf"], col1["t_layer"], col1["t"], tau, omega, emis) if lw else
                  oracle.sw_fluxes(omega, g, tau, col1["mu0"], 0.5, alb, alb, col1["tsi"], solar))
        rows = (up[0], up[-1], up[user_level], dn[0], dn[-1], dn[user_level])
        for k, name in enumerate(names):
            if name is None:
                continue
            suffix = "_user_level" if k in (2, 5) else ""
            want = oracle.integrate_row(rows[k], band.dw)
            cloudy = got[(0, name + "af" + suffix)][0]
            worst_cloudy = max(worst_cloudy, abs(cloudy - want))
            worst_clear = max(worst_clear, abs(got[(0, name + "csaf" + suffix)][0] - clear["integ"][k]))
            if k in (0, 3, 4) and name != "rsdt":
                assert abs(cloudy - got[(0, name + "csaf" + suffix)][0]) > 0.5          # the clouds matter
    print(f"reference driver.c cloud pass (clouds double): worst all-sky flux difference {worst_cloudy:.2e}, clear-sky {worst_clear:.2e} W m-2")
    assert worst_cloudy < 1e-6 and worst_clear < 1e-6


@needs_driver
@pytest.mark.gpu
def test_reference_driver_cloud_pass_with_this_repositorys_clouds_library(tmp_path, oracle, lib):
    """SURVEY §8(f)-4 complete: the UNCHANGED driver's cloud pass on libclouds.a of THIS repository
    (grtcode_amd/csrc/host/grt_clouds.c: Pade optics, stochastic condensate sampling, band-to-grid mapping; parameter
    files as GRTDUMP1) -- no test double.  The oracle gets its cloud optics from tests/cloud_model.py, drawing the same
    libc rand() sequence the driver process draws: GRT_CLOUDS_SEED makes initialize_clouds_lib call srand (the reference
    never seeds, and the GPU runtime has drawn from rand() by then); longwave pass first, then shortwave."""
    from cloud_model import LibcRand, cloud_optics, synthetic_tables
    col1, v1 = circ1_column()
    L = col1["p"].size - 1
    cf, lwc, iwc = np.zeros(L), np.zeros(L), np.zeros(L)
    cf[38:44], lwc[38:44] = 0.6, 0.2
    cf[14:19], iwc[14:19] = 0.4, 0.05
    cf[30], lwc[30], iwc[30] = 1.0, 0.05, 0.01                      # an overcast mixed-phase layer
    swb = Band(str(tmp_path / "data"), 1.0, 8000.0, 2.0, 8000, sw=True)
    lwb = Band(str(tmp_path / "lw_view"), 1.0, 2000.0, 1.0, 0, sw=True)
    lwb.par, lwb.h2o_dir, lwb.files, lwb.tab = swb.par, swb.h2o_dir, swb.files, swb.tab
    lwb.lines = {m: {k: a[(ln["v0"] >= lwb.w0) & (ln["v0"] <= lwb.wn)] for k, a in ln.items()} for m, ln in swb.lines.items()}
    paths, tables = synthetic_tables(str(tmp_path), seed=4, band_edges=[50.0, 400.0, 1000.0, 2200.0, 4000.0, 7000.0])
    dump = str(tmp_path / "columns.txt")
    write_column(dump, v1)
    with open(dump, "a") as f:
        for name, vals in (("cloud_fraction", cf), ("liquid_water_content", lwc), ("ice_water_content", iwc)):
            f.write(name + ": " + " ".join(repr(float(x)) for x in vals) + "\n")
    user_level, albedo, emissivity = 20, 0.196, 0.97
    out = str(tmp_path / "cloudy.txt")
    cmd = [DRIVER, swb.par, swb.files["solar"], dump, *("-" + NAME[m] for m in swb.mols),
           "-h2o-ctm", swb.h2o_dir, "-o3-ctm", swb.files["o3_ctm"], "-CFC-11", swb.files["cfc11"], "-CFC-12", swb.files["cfc12"],
           "-N2-N2", swb.files["cia_n2n2"], "-O2-N2", swb.files["cia_o2n2"], "-O2-O2", swb.files["cia_o2o2"],
           "-a", repr(albedo), "-e", repr(emissivity), "-flux-at-level", str(user_level + 1), "-integrated", "-clouds",
           "-beta-path", paths["beta"], "-ice-path", paths["ice"], "-liquid-path", paths["liquid"],
           "-w-lw", "1", "-W-lw", "2000", "-r-lw", "1", "-w-sw", "1", "-W-sw", "8000", "-r-sw", "2", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, GRT_OPTICS_HOST_VISIBLE="1", GRT_GAS_OPTICS_FAST="0", GRT_CLOUDS_SEED="20261004"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    got = parse_output(out)
    grid_sw = api.create_spectral_grid(swb.w0, swb.wn, swb.dw)
    solar = api.create_solar_flux(grid_sw, swb.files["solar"])
    thick = np.abs(np.log(col1["p"][:-1]) - np.log(col1["p"][1:])) * col1["t_layer"] * 8.314462 / (28.9647 * 0.001 * 9.81)
    play = np.array(v1["layer_pressure_mb"])
    overlap = np.exp(-1.0 * np.abs(np.diff(np.log(100.0 * play) * 7.3)) / 2.0)         # driver.c:170-181
    rand = LibcRand()
    rand.seed(20261004)
    worst_cloudy = worst_clear = 0.0
    for band, lw, names in ((lwb, True, ("rlut", "rlus", "rlu", None, "rlds", "rld")), (swb, False, ("rsut", "rsus", "rsu", "rsdt", "rsds", "rsd"))):
        emis, alb = np.full(band.nw, emissivity), np.full(band.nw, albedo)
        clear = oracle_column(oracle, lib, band, col1, lw, emis, alb, solar, user_level)
        centres = band.w0 + np.arange(band.nw) * band.dw
        limits = np.empty(band.nw + 1)
        limits[1:-1] = 0.5 * (centres[:-1] + centres[1:])
        limits[0] = max(centres[0] - band.dw, 0.0)
        limits[-1] = centres[-1] + band.dw
        bl, ol, gl, bi, oi, gi = cloud_optics(tables, rand, limits[:band.nw], cf, lwc, iwc, overlap, 10.0, col1["t_layer"])
        tl, ti = bl * thick[:, None], bi * thick[:, None]
        tr, om_r, g_r = oracle.rayleigh(L, col1["p"], band.w0, band.dw, band.nw)
        z = np.zeros_like(tr)
        tau, omega, g = oracle.add_optics([clear["tau_gas"], tr, tl, ti], [z, om_r, ol, oi], [z, g_r, gl, gi])
        up, dn = (oracle.lw_fluxes(band.w0, band.dw, col1["t_surf"], col1["t_layer"], col1["t"], tau, omega, emis) if lw else
                  oracle.sw_fluxes(omega, g, tau, col1["mu0"], 0.5, alb, alb, col1["tsi"], solar))
        rows = (up[0], up[-1], up[user_level], dn[0], dn[-1], dn[user_level])
        assert np.any(tl > 0) and np.any(ti > 0)
        for k, name in enumerate(names):
            if name is None:
                continue
            suffix = "_user_level" if k in (2, 5) else ""
            want = oracle.integrate_row(rows[k], band.dw)
            cloudy = got[(0, name + "af" + suffix)][0]
            worst_cloudy = max(worst_cloudy, abs(cloudy - want))
            worst_clear = max(worst_clear, abs(got[(0, name + "csaf" + suffix)][0] - clear["integ"][k]))
    print(f"reference driver.c cloud pass (this repository's clouds library): worst all-sky flux difference {worst_cloudy:.2e}, clear-sky {worst_clear:.2e} W m-2")
    assert worst_cloudy < 1e-6 and worst_clear < 1e-6
    assert abs(got[(0, "rlutaf")][0] - got[(0, "rlutcsaf")][0]) > 0.5 and abs(got[(0, "rsdsaf")][0] - got[(0, "rsdscsaf")][0]) > 0.5
