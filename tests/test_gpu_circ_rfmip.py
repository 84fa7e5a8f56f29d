"""Parity on the BASELINE.json configurations that are parity-test cases (not bench lines):

* CIRC case 1 -- the one real atmosphere in the reference tree (circ/src/circ1.h, committed as numbers in
  tests/golden/reference_test_vectors.json): 55 levels / 54 layers, clear sky, LW and SW on the full 1 cm-1
  grids, column prepared exactly as circ/src/basic-circ-test.c does (pressure-interpolated ppmv :51-66,
  cos(SZA) :118-120, TSI/cosz :122-124).  Spectroscopy is synthetic (no HITRAN data ships with the
  reference), so the LBLRTM numbers quoted there are only an order-of-magnitude check.
* RFMIP-IRF as BASELINE.json states it (config 3): 100 columns x 61 levels, LW and SW at 1 cm-1, production form.
"""
import json
import os

import numpy as np
import pytest

from grtcode_amd import api, synthetic as syn
from scenario import Band, MOL_ORDER
from test_gpu_pipeline import oracle_column

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
NAME = {syn.H2O: "H2O", syn.CO2: "CO2", syn.O3: "O3", syn.N2O: "N2O", syn.CO: "CO", syn.CH4: "CH4", syn.O2: "O2"}


def circ1_column():
    v = json.load(open(os.path.join(HERE, "golden", "reference_test_vectors.json")))["circ1"]
    p, pl = np.array(v["level_pressure_mb"]), np.array(v["layer_pressure_mb"])
    L = pl.size

    def to_levels(ab):                                  # basic-circ-test.c:51-66
        ab = np.array(ab)
        out = np.zeros(L + 1)
        out[0], out[L] = ab[0] * 1e6, ab[L - 1] * 1e6
        for i in range(1, L):
            out[i] = (ab[i - 1] + (ab[i] - ab[i - 1]) * (p[i] - pl[i - 1]) / (pl[i] - pl[i - 1])) * 1e6
        return out
    ppmv = {m: to_levels(v["abundance"][NAME[m]]) for m in MOL_ORDER}
    ppmv[syn.N2] = np.full(L + 1, 0.781e6)
    mu0 = float(np.cos(2.0 * np.pi * v["solar_zenith_angle_deg"] / 360.0))
    return dict(p=p, t=np.array(v["level_temperature"]), t_layer=np.array(v["layer_temperature"]),
                t_surf=v["surface_temperature"], ppmv=ppmv, mu0=mu0, tsi=v["toa_solar_irradiance"] / mu0,
                cfc_ppmv={0: to_levels(v["abundance"]["CFC11"]), 1: to_levels(v["abundance"]["CFC12"])}), v


@pytest.mark.parametrize("fast", [0, 1])
def test_circ_case1_lw_sw_full_grids(tmp_path, oracle, lib, device, fast):
    col, v = circ1_column()
    V = col["p"].size
    assert V == 55 and col["p"][0] < col["p"][-1]
    lwb = Band(str(tmp_path / "lw"), 1.0, 3250.0, 1.0, 30000)
    swb = Band(str(tmp_path / "sw"), 1.0, 50000.0, 1.0, 30000, sw=True)
    go_lw, grid_lw = lwb.gas_optics(device, V, from_file=False)
    go_sw, grid_sw = swb.gas_optics(device, V, from_file=False)
    go_lw.tune(fast=fast)
    go_sw.tune(fast=fast)
    emis, alb = np.full(lwb.nw, 1.0 - 0.196), np.full(swb.nw, 0.196)
    solar = api.create_solar_flux(grid_sw, swb.files["solar"])
    pipe = api.Pipeline(go_lw, go_sw, 1, 20, emis, alb, solar)
    gcols, keep = api.make_columns([col], MOL_ORDER, cfc_order=(0, 1))
    pipe.run(gcols)
    got = pipe.fluxes(1)[0]
    tol_tau, tol_flux = (2e-6, 1e-3) if fast else (1e-11, 1e-6)
    for bi, (band, lw) in enumerate(((lwb, True), (swb, False))):
        w = oracle_column(oracle, lib, band, col, lw, emis, alb, solar, 20)
        tau_gas = api.device_to_host(device, pipe.views(bi)["tau_gas"], (V - 1, band.nw))
        scale = np.abs(w["tau_gas"]).max(axis=1, keepdims=True)
        assert np.max(np.abs(tau_gas - w["tau_gas"]) / scale) < tol_tau
        assert np.max(np.abs(got[bi * 6: bi * 6 + 6] - w["integ"])) < tol_flux
    # magnitudes: surface emission sigma*T^4-like and the TOA insolation the CIRC case prescribes
    assert 300.0 < got[1] < 450.0                                   # LW up at the surface (LBLRTM: 445.12)
    assert abs(got[9] - v["toa_solar_irradiance"]) < 0.05 * v["toa_solar_irradiance"]    # SW down at TOA ~ 912.8
    pipe.destroy()
    go_lw.destroy()
    go_sw.destroy()


def test_rfmip_irf_100_columns_61_levels_lw_and_sw_production_form(tmp_path, oracle, lib, device):
    """BASELINE.json config 3 as stated: 100 columns x 61 levels, LW AND SW on the full 1 cm-1 grids (1-3250, 1-50000),
    production form (fast = 3), through the batched pipeline in chunks of 50; every tenth column (10 columns) against
    the CPU checker (the reference's own C where oracle/_ref is there).  Line list with band structure, so the fluxes
    depend on tau (OLR ~ 280, not a black atmosphere)."""
    from oracle import reference_column as RC
    from scenario import full_column
    kind, chk, orc = RC.checker(omp=True)
    RC.set_omp_threads(min(os.cpu_count() or 1, 16))
    V, ncol, chunk = 61, 100, 50
    lwb = Band(str(tmp_path / "lw"), 1.0, 3250.0, 1.0, 30000, physical=True)
    swb = Band(str(tmp_path / "sw"), 1.0, 50000.0, 1.0, 30000, sw=True, physical=True)
    go_lw, grid_lw = lwb.gas_optics(device, V, from_file=False)
    go_sw, grid_sw = swb.gas_optics(device, V, from_file=False)
    go_lw.tune(fast=3)
    go_sw.tune(fast=3)
    emis, alb = np.full(lwb.nw, 0.98), np.full(swb.nw, 0.2)
    solar = api.create_solar_flux(grid_sw, swb.files["solar"])
    pipe = api.Pipeline(go_lw, go_sw, chunk, -1, emis, alb, solar)
    cols = [syn.profile(c, V) for c in range(ncol)]
    got = np.zeros((ncol, api.GRT_FLUXES_PER_COLUMN))
    for first in range(0, ncol, chunk):
        part = cols[first: first + chunk]
        gcols, keep = api.make_columns(part, MOL_ORDER, cfc_order=(0, 1))
        pipe.run(gcols)
        got[first: first + len(part)] = pipe.fluxes(len(part))
    assert go_lw.last_launch()["fast"] == 3 and go_sw.last_launch()["fast"] == 3
    worst = 0.0
    checked = list(range(3, ncol, 10))
    assert len(checked) == 10
    for c in checked:
        for bi, (band, lw) in enumerate(((lwb, True), (swb, False))):
            w = full_column(kind, chk, orc, lib, band, cols[c], lw, emis, alb, solar)
            worst = max(worst, np.max(np.abs(got[c, bi * 6: bi * 6 + 6] - w["integ"])))
    print(f"config 3 (100 x 61, LW+SW @1 cm-1, fast=3): worst integrated-flux difference over 10 columns {worst:.2e} W m-2 ({kind})")
    assert worst < 1e-3
    assert np.all(np.isfinite(got)) and np.all(got[:, 0] > 100.0) and np.all(got[:, 10] > 100.0)
    assert len({round(x, 6) for x in got[:, 0]}) == ncol            # every column really got its own profile
    pipe.destroy()
    go_lw.destroy()
    go_sw.destroy()
