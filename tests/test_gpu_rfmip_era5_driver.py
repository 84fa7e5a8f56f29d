"""SURVEY §8(f)-1, second half: RFMIP-IRF and ERA5 columns through the reference's UNCHANGED
framework/src/driver.c.

oracle/_ref/grtcode_driver_dump = driver.c + utilities/src/argparse.c (both unchanged, compiled where they lie) +
examples/driver_app_dump.c (ours: the five driver.h callbacks with the command lines and column semantics of
rfmip-irf/src/rfmip-irf.c and era5/src/era5.c, on a flat binary dump of the netCDF variables) + this library.

RFMIP-IRF: a synthetic 100-site x 61-level x 2-experiment input; experiment 1 run as the reference's workflow runs it --
two processes with disjoint -x/-X site ranges (GRTworkflow/run-rfmip-irf.sh:103-132) -- then the two shards' integrated
fluxes meet on rank 0 through examples/gather_shards.c (grt_multi_gather_fluxes, file transport); every column against
the oracle fed with what rfmip-irf.c:169-325 must make of the input (Pa -> mb, cos of the zenith angle, level values
of the layer profiles by interpolation in pressure, global means times their units, per-site albedo / emissivity).
ERA5: (time, level, lat, lon) fields, a sub-block by -t/-T -x/-X -y/-Y, mass mixing ratios, mid-layer values, the
greenhouse-gas file, and the reference's cos(zenith) = -1, i.e. no shortwave (era5.c:406-412, driver.c:706)."""
import os
import subprocess

import numpy as np
import pytest

from grtcode_amd import api, synthetic as syn
from grtcode_amd.dumpfile import read_dump, write_dump
from scenario import Band
from test_gpu_pipeline import oracle_column

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "grtcode_driver_dump")
LIBDIR = os.path.join(ROOT, "grtcode_amd", "lib")
needs_driver = pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref/grtcode_driver_dump not built (needs /root/reference at build time)")
MOLS = [syn.H2O, syn.CO2, syn.O3, syn.N2O, syn.CH4, syn.O2]


def test_dump_container_round_trip(tmp_path):
    a = np.arange(2 * 3 * 4 * 5, dtype=np.float64).reshape(2, 3, 4, 5)
    write_dump(str(tmp_path / "d.bin"), {"field": a, "carbon_dioxide_GM": (np.array([284.3, 568.6]), "1e-6"), "scalar": 3.5})
    d = read_dump(str(tmp_path / "d.bin"))
    assert np.array_equal(d["field"][0], a) and d["carbon_dioxide_GM"][1] == "1e-6" and d["scalar"][0].shape == (1,)
    with pytest.raises(ValueError):
        write_dump(str(tmp_path / "e.bin"), {"x": np.zeros((1, 1, 1, 1, 1))})


def bands(tmp_path, w_lw, w_sw, dw_sw, nlines):
    swb = Band(str(tmp_path / "data"), 1.0, w_sw, dw_sw, nlines, mols=MOLS, sw=True)
    lwb = Band(str(tmp_path / "lw_view"), 1.0, w_lw, 1.0, 0, mols=MOLS, sw=True)
    lwb.par, lwb.h2o_dir, lwb.files, lwb.tab = swb.par, swb.h2o_dir, swb.files, swb.tab
    lwb.lines = {m: {k: a[(ln["v0"] >= lwb.w0) & (ln["v0"] <= lwb.wn)] for k, a in ln.items()} for m, ln in swb.lines.items()}
    return lwb, swb


def parse_output(path):
    out = {}
    for line in open(path):
        if line.startswith("#"):
            continue
        t, c, name, count, *vals = line.split()
        out[(int(t), int(c), name)] = np.array([float(x) for x in vals])
    return out


def layers_to_levels(ab, p, pl):
    """rfmip-irf.c:295-308"""
    L = pl.size
    out = np.zeros(L + 1)
    out[0], out[L] = ab[0] * 1e6, ab[L - 1] * 1e6
    for k in range(1, L):
        out[k] = 1e6 * (ab[k - 1] + (ab[k] - ab[k - 1]) * (p[k] - pl[k - 1]) / (pl[k] - pl[k - 1]))
    return out


def driver_flags(lwb, swb, dw_sw):
    return ["-h2o-ctm", swb.h2o_dir, "-o3-ctm", swb.files["o3_ctm"],
            "-N2-N2", swb.files["cia_n2n2"], "-O2-N2", swb.files["cia_o2n2"], "-O2-O2", swb.files["cia_o2o2"],
            "-w-lw", "1", "-W-lw", repr(lwb.wn), "-r-lw", "1", "-w-sw", "1", "-W-sw", repr(swb.wn), "-r-sw", repr(dw_sw)]


@needs_driver
@pytest.mark.gpu
def test_rfmip_irf_100_sites_two_shards_gathered(tmp_path, oracle, lib):
    nsite, V, nexp, experiment = 100, 61, 2, 1
    L = V - 1
    lwb, swb = bands(tmp_path, 600.0, 2400.0, 2.0, 4000)
    # ---- the input as RFMIP-IRF's netCDF file holds it -------------------------------------------------------------
    rng = np.random.default_rng(61)
    base = [syn.profile(c, V) for c in range(nsite)]
    pres_level = np.array([b["p"] * 100.0 for b in base])                                   # Pa
    pres_layer = 0.5 * (pres_level[:, :-1] + pres_level[:, 1:]) * (1.0 + 0.01 * rng.random((nsite, L)))   # not the mid-point
    temp_level = np.array([[b["t"] + 4.0 * e for b in base] for e in range(nexp)])
    temp_layer = np.array([[b["t_layer"] + 4.0 * e for b in base] for e in range(nexp)])
    tsurf = np.array([[b["t_surf"] + 4.5 * e for b in base] for e in range(nexp)])
    sza = np.array([[20.0, 55.0, 100.0, 70.0, 0.0, 89.0][c % 6] for c in range(nsite)])    # every sixth site is in the dark
    tsi = 1360.0 + 0.1 * np.arange(nsite)
    albedo = 0.05 + 0.3 * rng.random(nsite)
    emissivity = 0.9 + 0.1 * rng.random(nsite)
    h2o = np.array([[0.5e-6 * (b["ppmv"][syn.H2O][:-1] + b["ppmv"][syn.H2O][1:]) * (1.0 + 0.2 * e) for b in base] for e in range(nexp)])
    o3 = np.array([[0.5e-6 * (b["ppmv"][syn.O3][:-1] + b["ppmv"][syn.O3][1:]) for b in base] for e in range(nexp)])
    gm = {"carbon_dioxide_GM": ([284.3, 1137.3], "1e-6"), "methane_GM": ([808.2, 1831.5], "1e-9"),
          "nitrous_oxide_GM": ([273.0, 327.0], "1e-9"), "oxygen_GM": ([0.209, 0.209], "1"), "nitrogen_GM": ([0.781, 0.781], "1"),
          "cfc11eq_GM": ([32.1, 809.2], "1e-12"), "cfc12_GM": ([0.0, 520.6], "1e-12")}
    dump = str(tmp_path / "rfmip.dump")
    write_dump(dump, dict({"pres_level": pres_level, "pres_layer": pres_layer, "temp_level": temp_level, "temp_layer": temp_layer,
                           "surface_temperature": tsurf, "solar_zenith_angle": sza, "total_solar_irradiance": tsi,
                           "surface_albedo": albedo, "surface_emissivity": emissivity, "water_vapor": h2o, "ozone": o3},
                          **{k: (np.array(v), u) for k, (v, u) in gm.items()}))
    # ---- the unchanged driver, two shards -----------------------------------------------------------------------------
    flags = ["-H2O", "-CO2", "-O3", "-N2O", "-CH4", "-O2", "-CFC-11-eq", swb.files["cfc11"], "-CFC-12", swb.files["cfc12"]]
    env = dict(os.environ, GRT_GAS_OPTICS_FAST="0")
    shards = [(0, 49), (50, 99)]
    procs = []
    for r, (x, X) in enumerate(shards):
        cmd = [DRIVER, swb.par, swb.files["solar"], dump, str(experiment), *flags, *driver_flags(lwb, swb, 2.0),
               "-x", str(x), "-X", str(X), "-integrated", "-o", str(tmp_path / f"shard{r}.txt")]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=900)
        assert p.returncode == 0, (r, out[-1500:], err[-1500:])
    # ---- one gather of the shards' [columns][12] blocks to rank 0 -------------------------------------------------
    exe = str(tmp_path / "gather_shards")
    r = subprocess.run(["gcc", "-std=gnu99", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "gather_shards.c"), "-L" + LIBDIR, "-lgrtcode_hip", "-lm",
                        "-Wl,-rpath," + LIBDIR, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rdv = tmp_path / "rdv"
    rdv.mkdir()
    gprocs = [subprocess.Popen([exe, str(tmp_path / f"shard{k}.txt"), "-columns", str(nsite), "-ranks", "2", "-rank", str(k),
                                "-rendezvous", str(rdv)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                               env=dict(os.environ, GRT_MULTI_TIMEOUT="120", GRT_MULTI_JOB="rfmip-test")) for k in range(2)]
    gouts = [p.communicate(timeout=300) for p in gprocs]
    for k, p in enumerate(gprocs):
        assert p.returncode == 0, (k, gouts[k][1][-1500:])
    got = np.array([[float(v) for v in line.split(":")[1].split()] for line in gouts[0][0].splitlines() if line.startswith("col ")])
    assert got.shape == (nsite, 12) and gouts[1][0].strip() == "" and os.listdir(rdv) == []
    # ---- what rfmip-irf.c makes of the input, through the oracle -------------------------------------------------
    grid_sw = api.create_spectral_grid(swb.w0, swb.wn, swb.dw)
    solar = api.create_solar_flux(grid_sw, swb.files["solar"])
    scale = lambda name: gm[name][0][experiment] * float(gm[name][1]) * 1e6
    worst, nights = 0.0, 0
    for c in range(nsite):
        p, pl = pres_level[c] * 0.01, pres_layer[c] * 0.01
        ppmv = {syn.H2O: layers_to_levels(h2o[experiment, c], p, pl), syn.O3: layers_to_levels(o3[experiment, c], p, pl),
                syn.CO2: np.full(V, scale("carbon_dioxide_GM")), syn.CH4: np.full(V, scale("methane_GM")),
                syn.N2O: np.full(V, scale("nitrous_oxide_GM")), syn.O2: np.full(V, scale("oxygen_GM")),
                syn.N2: np.full(V, scale("nitrogen_GM"))}
        mu0 = float(np.cos(2.0 * np.pi * sza[c] / 360.0))
        col = dict(p=p, t=temp_level[experiment, c], t_layer=temp_layer[experiment, c], t_surf=tsurf[experiment, c], ppmv=ppmv,
                   mu0=mu0, tsi=tsi[c], cfc_ppmv={0: np.full(V, scale("cfc11eq_GM")), 1: np.full(V, scale("cfc12_GM"))})
        emis, alb = np.full(lwb.nw, emissivity[c]), np.full(swb.nw, albedo[c])
        want = oracle_column(oracle, lib, lwb, col, True, emis, alb, solar)["integ"]
        worst = max(worst, np.max(np.abs(got[c, [0, 1, 4]] - want[[0, 1, 4]])))
        assert got[c, 3] == 0.0                                       # the driver writes no longwave down-at-TOA
        if mu0 > 0.0:
            want = oracle_column(oracle, lib, swb, col, False, emis, alb, solar)["integ"]
            worst = max(worst, np.max(np.abs(got[c, [6, 7, 9, 10]] - want[[0, 1, 3, 4]])))
        else:
            nights += 1
            assert np.all(got[c, 6:] == 0.0)                          # driver.c:706: no shortwave pass in the dark
    print(f"RFMIP-IRF through the unchanged driver.c, 100 sites in two -x/-X shards, gathered: worst flux difference {worst:.2e} W m-2")
    assert nights == sum(1 for c in range(nsite) if c % 6 == 2) and nights > 10
    assert worst < 1e-6
    # the shard files name GLOBAL site indices, and state variables are what the application derived
    second = parse_output(str(tmp_path / "shard1.txt"))
    assert (0, 50, "rlutcsaf") in second and (0, 99, "rlutcsaf") in second and (0, 49, "rlutcsaf") not in second
    assert np.allclose(second[(0, 73, "level_pressure")], pres_level[73] * 0.01, rtol=1e-15)
    assert np.allclose(second[(0, 73, "h2o_vmr")], layers_to_levels(h2o[experiment, 73], pres_level[73] * 0.01, pres_layer[73] * 0.01), rtol=1e-14)


@needs_driver
@pytest.mark.gpu
def test_era5_block_through_the_unchanged_driver(tmp_path, oracle, lib):
    nt, V, nlat, nlon = 2, 31, 3, 4
    L = V - 1
    lwb, swb = bands(tmp_path, 700.0, 1200.0, 4.0, 3000)
    rng = np.random.default_rng(5)
    prof = [[[syn.profile(7 * t + 3 * j + i, V) for i in range(nlon)] for j in range(nlat)] for t in range(nt)]
    field = lambda f: np.array([[[[f(prof[t][j][i])[k] for i in range(nlon)] for j in range(nlat)] for k in range(V)] for t in range(nt)])
    p = field(lambda c: c["p"])                                       # (time, level, lat, lon), mb as the application takes it
    tt = field(lambda c: c["t"])
    q = field(lambda c: c["ppmv"][syn.H2O] * 1e-6 * 18.01528 / 28.97)             # mass mixing ratios
    o3 = field(lambda c: c["ppmv"][syn.O3] * 1e-6 * 48.0 / 28.97)
    skt = np.array([[[prof[t][j][i]["t_surf"] for i in range(nlon)] for j in range(nlat)] for t in range(nt)])
    tisr = 86400.0 * (200.0 + 100.0 * rng.random((nt, nlat, nlon)))
    fal = 0.1 + 0.2 * rng.random((nt, nlat, nlon))
    era = str(tmp_path / "era5.dump")
    write_dump(era, {"p": p, "t": tt, "q": q, "o3": o3, "skt": skt, "tisr": tisr, "fal": fal})
    ghg = str(tmp_path / "ghg.dump")
    years = np.arange(2010, 2016)
    write_dump(ghg, {"ch4": 1.80 + 0.01 * (years - 2010), "co2": 390.0 + 2.0 * (years - 2010), "n2o": 0.323 + 0.001 * (years - 2010),
                     "hfc134aeq": 1.0e-4 * (years - 2000), "cfc12eq": 1.0e-3 + 0.0 * years})
    year, t_sel, ys, xs = 2013, (1, 1), (0, 1), (1, 3)
    out = str(tmp_path / "era5.txt")
    mols = [syn.H2O, syn.O3, syn.CH4, syn.CO2, syn.N2O]
    cmd = [DRIVER, swb.par, swb.files["solar"], era, ghg, "-format", "era5", "-H2O", "-O3", "-CH4", "-CO2", "-N2O",
           "-HFC-134a-eq", swb.files["cfc11"], "-CFC-12-eq", swb.files["cfc12"], *driver_flags(lwb, swb, 4.0),
           "-year", str(year), "-ghg_start_year", "2010", "-t", str(t_sel[0]), "-T", str(t_sel[1]),
           "-y", str(ys[0]), "-Y", str(ys[1]), "-x", str(xs[0]), "-X", str(xs[1]), "-clear", "-integrated", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, GRT_GAS_OPTICS_FAST="0"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    got = parse_output(out)
    assert not any(name.startswith("rs") for _, _, name in got)       # cos(zenith) = -1: the shortwave never runs
    lwb.mols = swb.mols = mols
    lwb.lines = {m: lwb.lines[m] for m in mols}
    iy = year - 2010
    worst = 0.0
    ncell = 0
    for j in range(ys[0], ys[1] + 1):
        for i in range(xs[0], xs[1] + 1):
            t = t_sel[0]
            plev, tlev = p[t, :, j, i], tt[t, :, j, i]
            play = 0.5 * (plev[:-1] + plev[1:])
            tlay = tlev[:-1] + (tlev[1:] - tlev[:-1]) * (play - plev[:-1]) / (plev[1:] - plev[:-1])
            ppmv = {syn.H2O: q[t, :, j, i] * (1e6 * (28.97 / 18.01528)), syn.O3: o3[t, :, j, i] * (1e6 * (28.97 / 48.0)),
                    syn.CH4: np.full(V, 1.80 + 0.01 * iy), syn.CO2: np.full(V, 390.0 + 2.0 * iy), syn.N2O: np.full(V, 0.323 + 0.001 * iy),
                    syn.N2: np.full(V, 0.781e6), syn.O2: np.full(V, 0.21e6)}
            col = dict(p=plev, t=tlev, t_layer=tlay, t_surf=skt[t, j, i], ppmv=ppmv, mu0=-1.0, tsi=0.0,
                       cfc_ppmv={0: np.full(V, 1.0e-4 * (year - 2000)), 1: np.full(V, 1.0e-3)})
            want = oracle_column(oracle, lib, lwb, col, True, np.ones(lwb.nw), None, None)["integ"]
            cell = (j - ys[0]) * (xs[1] - xs[0] + 1) + (i - xs[0])       # the driver's column index within the block
            have = np.array([got[(0, cell, n)][0] for n in ("rlutcsaf", "rluscsaf", "rldscsaf")])
            worst = max(worst, np.max(np.abs(have - want[[0, 1, 4]])))
            assert np.allclose(got[(0, cell, "layer_temperature")], tlay, rtol=1e-14)
            ncell += 1
    print(f"ERA5 block through the unchanged driver.c ({ncell} cells): worst longwave flux difference {worst:.2e} W m-2")
    assert ncell == 6 and worst < 1e-6
