"""BASELINE.json's remaining configurations as parity-test cases (SURVEY.md §8d "Mapping of configs"):

(2) CIRC cases 1-7, clear-sky LW+SW at 1 cm-1: the reference tree holds case 1 only (circ/src/circ1.h); cases 2-7
    need circ-case{2..7}.nc + netCDF, so they are case 1 plus six deterministic perturbations (temperature offsets,
    moisture and ozone scalings, a colder/drier and a warmer/wetter column, a different sun), run as ONE batch.
(4) 1 800 columns = 100 columns x 18 replicas, sharded over 8 ranks with one gather: as stated (61 levels, LW + SW on
    the full 1 cm-1 grids, production form) with 8 processes taking turns on the one GPU and the gather through the C
    entry points' file transport (test_config4_...); plus the property that a column's fluxes do not depend on its
    shard / chunk slot / neighbours.
(5) ERA5-like: longwave at 0.1 cm-1, shortwave at 10 cm-1 (GRTworkflow/run-era5.sh), all 21 CFC species of
    cfcs.h:32-56 active.

Tolerances: production arithmetic (fast=3); optical depths 2e-6 of each layer's maximum, integrated fluxes
1e-3 W m-2 (BASELINE.json north star) -- asserted at 1e-4.
"""
import copy

import numpy as np
import pytest

from grtcode_amd import api, multi, synthetic as syn
from scenario import Band, MOL_ORDER, RUN_TO_RUN_FUSED_FLUX
from test_gpu_circ_rfmip import circ1_column
from test_gpu_gas_optics import tau_close
from test_gpu_pipeline import oracle_column

pytestmark = pytest.mark.gpu
FLUX_TOL = 1e-4


def circ_like_columns():
    base, v = circ1_column()
    cols = [base]
    k = np.linspace(0.0, 1.0, base["p"].size)

    def variant(dt=0.0, h2o=1.0, o3=1.0, co2=1.0, sza=None, tsurf=0.0):
        c = copy.deepcopy(base)
        c["t"] = c["t"] + dt * k
        c["t_layer"] = c["t_layer"] + dt * 0.5 * (k[:-1] + k[1:])
        c["t_surf"] = c["t_surf"] + dt + tsurf
        c["ppmv"][syn.H2O] = c["ppmv"][syn.H2O] * h2o
        c["ppmv"][syn.O3] = c["ppmv"][syn.O3] * o3
        c["ppmv"][syn.CO2] = c["ppmv"][syn.CO2] * co2
        if sza is not None:
            c["mu0"] = float(np.cos(np.deg2rad(sza)))
            c["tsi"] = v["toa_solar_irradiance"] / float(np.cos(np.deg2rad(v["solar_zenith_angle_deg"])))
        return c
    cols += [variant(dt=-25.0, h2o=0.2), variant(dt=8.0, h2o=1.6, tsurf=3.0), variant(o3=1.5, sza=30.0),
             variant(co2=2.0), variant(dt=-10.0, h2o=0.5, o3=0.7, sza=75.0), variant(h2o=3.0, co2=0.7, sza=0.0)]
    return cols


def test_circ_cases_1_to_7_lw_sw_one_batch(tmp_path, oracle, lib, device):
    cols = circ_like_columns()
    V = cols[0]["p"].size
    lwb = Band(str(tmp_path / "lw"), 1.0, 3250.0, 1.0, 12000)
    swb = Band(str(tmp_path / "sw"), 1.0, 50000.0, 1.0, 12000, sw=True)
    go_lw, grid_lw = lwb.gas_optics(device, V, from_file=False)
    go_sw, grid_sw = swb.gas_optics(device, V, from_file=False)
    go_lw.tune(fast=3)
    go_sw.tune(fast=3)
    emis, alb = np.full(lwb.nw, 1.0 - 0.196), np.full(swb.nw, 0.196)
    solar = api.create_solar_flux(grid_sw, swb.files["solar"])
    pipe = api.Pipeline(go_lw, go_sw, len(cols), 20, emis, alb, solar)
    gcols, keep = api.make_columns(cols, MOL_ORDER, cfc_order=(0, 1))
    pipe.run(gcols)
    got = pipe.fluxes(len(cols))
    worst = 0.0
    for ci, col in enumerate(cols):
        for bi, (band, lw) in enumerate(((lwb, True), (swb, False))):
            w = oracle_column(oracle, lib, band, col, lw, emis, alb, solar, 20)
            worst = max(worst, np.max(np.abs(got[ci, bi * 6: bi * 6 + 6] - w["integ"])))
    print(f"CIRC-like batch of 7: worst integrated-flux difference {worst:.2e} W m-2")
    assert worst < FLUX_TOL
    assert len({round(x, 4) for x in got[:, 0]}) == len(cols)       # seven different atmospheres, seven OLRs
    pipe.destroy()
    go_lw.destroy()
    go_sw.destroy()


def test_1800_replicated_columns_in_shards(tmp_path, device):
    base_n, replicas, chunk, V = 100, 18, 64, 21
    band = Band(str(tmp_path), 600.0, 760.0, 1.0, 3000)
    go, grid = band.gas_optics(device, V, from_file=False)
    go.tune(fast=3)
    emis = np.full(band.nw, 0.98)
    pipe = api.Pipeline(go, None, chunk, -1, emis, None, None)
    base = [syn.profile(c, V) for c in range(base_n)]
    order = [(r, c) for r in range(replicas) for c in range(base_n)]           # replica-major, as run-rfmip would
    total = len(order)
    assert total == 1800
    # the 8-rank sharding of the set tiles it exactly (multi.shard is what bench.py and a driver use)
    parts = [multi.shard(total, rank, 8) for rank in range(8)]
    parts = [(first, first + count) for first, count in parts]
    assert parts[0][0] == 0 and parts[-1][1] == total and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    got = np.zeros((total, api.GRT_FLUXES_PER_COLUMN))
    for lo, hi in parts:                                                       # each "rank" runs its shard in chunks
        for first in range(lo, hi, chunk):
            ids = order[first: min(first + chunk, hi)]
            gcols, keep = api.make_columns([base[c] for _, c in ids], MOL_ORDER, cfc_order=(0, 1))
            pipe.run(gcols)
            got[first: first + len(ids)] = pipe.fluxes(len(ids))
    per_col = got.reshape(replicas, base_n, -1)
    # a column's fluxes do not depend on which shard / chunk slot / neighbours it had -- up to the order in which
    # waves add their partial sums (fp32 moments, fp64 tiles): ~1e-8 W m-2 on fluxes of a few hundred
    spread = np.max(np.abs(per_col - per_col[0:1]), axis=0)[:, :6]
    assert np.max(spread) < RUN_TO_RUN_FUSED_FLUX
    assert np.all(per_col[0, :, 0] > 0) and len({round(x, 6) for x in per_col[0, :, 0]}) == base_n
    pipe.destroy()
    go.destroy()


def test_era5_like_fine_longwave_coarse_shortwave_all_cfcs(tmp_path, oracle, lib, device):
    V, ncfc = 17, 21
    lwb = Band(str(tmp_path / "lw"), 700.0, 1300.0, 0.1, 5000)
    swb = Band(str(tmp_path / "sw"), 1.0, 20000.0, 10.0, 5000, sw=True)
    cols = [syn.profile(c, V) for c in (11, 12)]
    for c in cols:                                                             # 21 species, distinct abundances
        c["cfc_ppmv"] = {k: np.full(V, 1.0e-4 * (1 + k)) for k in range(ncfc)}

    def build(band):
        grid = api.create_spectral_grid(band.w0, band.wn, band.dw)
        go = api.GasOpticsObject(V, grid, device, band.par, band.h2o_dir, band.files["o3_ctm"])
        for m in band.mols:
            go.add_molecule_lines(m, band.lines[m])
        for k in range(ncfc):
            go.add_cfc(k, band.files["cfc11" if k % 2 == 0 else "cfc12"])
        for a, b, name in ((0, 0, "cia_n2n2"), (1, 0, "cia_o2n2"), (1, 1, "cia_o2o2")):
            go.add_cia(a, b, band.files[name])
        go.tune(fast=3)
        return go, grid

    def oracle_band(band, col, lw, emis=None, alb=None, solar=None):
        kw = band.oracle_inputs(oracle, lib, col)
        kw["cfcs"] = [(col["cfc_ppmv"][k] * 1e-6, band.table_on_grid(oracle, "cfc11" if k % 2 == 0 else "cfc12"))
                      for k in range(ncfc)]
        tau_gas = oracle.gas_optics(col["p"], col["t"], band.w0, band.dw, band.nw, **kw)
        L = V - 1
        tr, om_r, g_r = oracle.rayleigh(L, col["p"], band.w0, band.dw, band.nw)
        z = np.zeros_like(tau_gas)
        tau, omega, g = oracle.add_optics([tau_gas, tr], [z, om_r], [z, g_r])
        if lw:
            up, dn = oracle.lw_fluxes(band.w0, band.dw, col["t_surf"], col["t_layer"], col["t"], tau, omega, emis)
        else:
            up, dn = oracle.sw_fluxes(omega, g, tau, col["mu0"], 0.5, alb, alb, col["tsi"], solar)
        integ = np.array([oracle.integrate_row(r, band.dw) for r in (up[0], up[-1], dn[0], dn[-1])])
        return tau_gas, integ

    go_lw, grid_lw = build(lwb)
    go_sw, grid_sw = build(swb)
    emis, alb = np.full(lwb.nw, 0.98), np.full(swb.nw, 0.2)
    solar = api.create_solar_flux(grid_sw, swb.files["solar"])
    pipe = api.Pipeline(go_lw, go_sw, len(cols), -1, emis, alb, solar)
    gcols, keep = api.make_columns(cols, MOL_ORDER, cfc_order=tuple(range(ncfc)))
    pipe.run(gcols)
    got = pipe.fluxes(len(cols))
    for bi, (band, lw) in enumerate(((lwb, True), (swb, False))):
        tau_dev = api.device_to_host(device, pipe.views(bi)["tau_gas"], (len(cols), V - 1, band.nw))
        for ci, col in enumerate(cols):
            tau_gas, integ = oracle_band(band, col, lw, emis, alb, solar)
            assert tau_close(tau_dev[ci], tau_gas) < 2e-6
            mine = got[ci, bi * 6: bi * 6 + 6][[0, 1, 3, 4]]
            assert np.max(np.abs(mine - integ)) < FLUX_TOL
    pipe.destroy()
    go_lw.destroy()
    go_sw.destroy()


# ---- config 4 as BASELINE.json states it, as far as one GPU allows ------------------------------------------------- #
RANK_SCRIPT = r"""
import os, sys
import numpy as np
root, tests = sys.argv[1], sys.argv[2]
sys.path.insert(0, root); sys.path.insert(0, tests)
os.environ["GRT_TIPS_QUIET"] = "1"
rank, world, total, work = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
from grtcode_amd import api, multi, synthetic as syn
from scenario import Band, MOL_ORDER, RUN_TO_RUN_FUSED_FLUX
from test_gpu_baseline_configs import replicated_column, CONFIG4
V, chunk = CONFIG4["levels"], CONFIG4["chunk"]
device = api.create_device(0)
lwb = Band(os.path.join(work, f"rank{rank}", "lw"), 1.0, 3250.0, 1.0, CONFIG4["lines"], physical=True)      # same seed -> the same spectroscopy on every rank
swb = Band(os.path.join(work, f"rank{rank}", "sw"), 1.0, 50000.0, 1.0, CONFIG4["lines"], sw=True, physical=True)
go_lw, grid_lw = lwb.gas_optics(device, V, from_file=False)
go_sw, grid_sw = swb.gas_optics(device, V, from_file=False)
go_lw.tune(fast=3); go_sw.tune(fast=3)
emis, alb = np.full(lwb.nw, 0.98), np.full(swb.nw, 0.2)
solar = api.create_solar_flux(grid_sw, swb.files["solar"])
pipe = api.Pipeline(go_lw, go_sw, chunk, -1, emis, alb, solar, spectral=False)       # the production pipeline
m = multi.Multi(multi.FILES, device, rank, world, os.path.join(work, "rdv"))
first, count = m.shard(total)
local = np.zeros((count, 12))
for lo in range(0, count, chunk):
    part = [replicated_column(first + lo + k, V) for k in range(min(chunk, count - lo))]
    gcols, keep = api.make_columns(part, MOL_ORDER, cfc_order=(0, 1))
    pipe.run(gcols)
    local[lo: lo + len(part)] = pipe.fluxes(len(part))
per = -(-total // world)
allf = np.zeros((per * world, 12)) if rank == 0 else None
m.gather_fluxes(local.ctypes.data if count else 0, total, allf.ctypes.data if rank == 0 else 0, False)
if rank == 0:
    np.save(os.path.join(work, "gathered.npy"), allf[:total])
m.destroy(); pipe.destroy(); go_lw.destroy(); go_sw.destroy()
print("rank", rank, "columns", first, first + count)
"""
CONFIG4 = dict(levels=61, chunk=45, lines=8000, base=100, replicas=18)


def replicated_column(index, V):
    """Column `index` of the 1 800-column set: 100 RFMIP-like base columns x 18 replicas, each replica with its own
    temperature perturbation (SURVEY §8d "Column sets"), replica-major as run-rfmip-irf.sh walks its experiments."""
    replica, base = divmod(index, CONFIG4["base"])
    c = syn.profile(base, V)
    rng = np.random.default_rng(5000 + replica)
    dt = rng.uniform(-3.0, 3.0)
    c["t"] = c["t"] + dt
    c["t_layer"] = c["t_layer"] + dt
    c["t_surf"] = c["t_surf"] + dt
    return c


def test_config4_1800_replicated_columns_in_8_rank_shards_with_gather(tmp_path, oracle, lib, device):
    """BASELINE config 4 -- 1 800 synthetic-replicated columns x 61 levels, LW + SW on the full 1 cm-1 grids, production
    form, sharded over 8 ranks (225 columns each) with ONE gather of the [225][12] blocks to rank 0 -- run as far as one
    GPU allows: 8 processes take turns on the card (at most 4 at a time), each its own shard through its own production
    pipeline, and the blocks meet through the library's C entry points grt_multi_* with the file transport (RCCL wants
    a GPU per rank: the driver's 8-GPU runs).  Rank 0's gathered array must hold every column, in order, equal to what a
    single process computes, and within 1e-3 W m-2 of the CPU checker on sampled columns."""
    import os
    import subprocess
    import sys
    from oracle import reference_column as RC
    from scenario import full_column
    world, total = 8, CONFIG4["base"] * CONFIG4["replicas"]
    assert total == 1800 and [multi.shard(total, r, world)[1] for r in range(world)] == [225] * 8
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT)
    (tmp_path / "rdv").mkdir()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = [sys.executable, str(script), root, os.path.join(root, "tests")]
    env = dict(os.environ, GRT_MULTI_TIMEOUT="900")
    for wave in ([7, 6, 5, 4], [3, 2, 1, 0]):                       # rank 0 (which waits for everybody's file) in the last wave
        procs = [subprocess.Popen(args + [str(r), str(world), str(total), str(tmp_path)], stdout=subprocess.PIPE,
                                  stderr=subprocess.PIPE, text=True, env=env) for r in wave]
        for r, p in zip(wave, procs):
            out, err = p.communicate(timeout=900)
            assert p.returncode == 0, (r, out[-500:], err[-2000:])
    got = np.load(str(tmp_path / "gathered.npy"))
    assert got.shape == (total, 12) and np.all(np.isfinite(got)) and np.all(got[:, 0] > 100.0) and np.all(got[:, 10] > 100.0)
    # replicas of a base column differ (their own temperatures) ...
    per_base = got.reshape(CONFIG4["replicas"], CONFIG4["base"], 12)
    assert np.all(np.ptp(per_base[:, :, 0], axis=0) > 1e-3)
    # ... and every gathered row is the column it should be: sampled rows across all eight shards against one process
    V = CONFIG4["levels"]
    lwb = Band(str(tmp_path / "lw"), 1.0, 3250.0, 1.0, CONFIG4["lines"], physical=True)
    swb = Band(str(tmp_path / "sw"), 1.0, 50000.0, 1.0, CONFIG4["lines"], sw=True, physical=True)
    go_lw, grid_lw = lwb.gas_optics(device, V, from_file=False)
    go_sw, grid_sw = swb.gas_optics(device, V, from_file=False)
    go_lw.tune(fast=3)
    go_sw.tune(fast=3)
    emis, alb = np.full(lwb.nw, 0.98), np.full(swb.nw, 0.2)
    solar = api.create_solar_flux(grid_sw, swb.files["solar"])
    sample = [0, 224, 225, 449, 700, 1012, 1349, 1574, 1575, 1799]
    pipe = api.Pipeline(go_lw, go_sw, len(sample), -1, emis, alb, solar, spectral=False)
    gcols, keep = api.make_columns([replicated_column(i, V) for i in sample], MOL_ORDER, cfc_order=(0, 1))
    pipe.run(gcols)
    one = pipe.fluxes(len(sample))
    assert np.max(np.abs(one - got[sample])) < RUN_TO_RUN_FUSED_FLUX       # (atomics: run-to-run ~1e-9)
    kind, chk, orc = RC.checker(omp=True)
    worst = 0.0
    for i in (224, 1575):
        col = replicated_column(i, V)
        for bi, (band, lw) in enumerate(((lwb, True), (swb, False))):
            w = full_column(kind, chk, orc, lib, band, col, lw, emis, alb, solar)
            worst = max(worst, np.max(np.abs(got[i, bi * 6: bi * 6 + 6] - w["integ"])))
    print(f"config 4: 1 800 columns in 8 rank shards, gathered; worst flux difference on sampled columns {worst:.2e} W m-2 ({kind})")
    assert worst < 1e-3
    pipe.destroy()
    go_lw.destroy()
    go_sw.destroy()


def test_config5_era5_like_two_thousand_columns_full_grids_all_cfcs(tmp_path, oracle, lib, device):
    """BASELINE config 5 at a size one GPU does in seconds: 2 048 columns (O(10^3) of the O(10^4)) of 60 layers, longwave 1-3250 cm-1
    at 0.1 cm-1 (n = 32 491: the tree form of the line kernel), shortwave 1-50 000 cm-1 at 10 cm-1 (7-point windows: the
    direct-walk form), all 21 CFC/HCFC species of cfcs.h:32-56, 3 CIA pairs, production pipeline in chunks of 64;
    sampled columns against the CPU checker."""
    import os
    from oracle import reference_column as RC
    kind, chk, orc = RC.checker(omp=True)
    RC.set_omp_threads(min(os.cpu_count() or 1, 16))
    V, ncfc, ncol, chunk = 61, 21, 2048, 64          # (61 levels = the stated 60 layers; round 3 ran 31)
    lwb = Band(str(tmp_path / "lw"), 1.0, 3250.0, 0.1, 6000, physical=True)
    swb = Band(str(tmp_path / "sw"), 1.0, 50000.0, 10.0, 6000, sw=True, physical=True)

    def column(c):
        col = syn.profile(c, V)
        col["cfc_ppmv"] = {k: np.full(V, 0.5e-4 * (1 + k) * (1.0 + 0.1 * (c % 7))) for k in range(ncfc)}
        return col

    def build(band):
        grid = api.create_spectral_grid(band.w0, band.wn, band.dw)
        go = api.GasOpticsObject(V, grid, device, band.par, band.h2o_dir, band.files["o3_ctm"])
        for m in band.mols:
            go.add_molecule_lines(m, band.lines[m])
        for k in range(ncfc):
            go.add_cfc(k, band.files["cfc11" if k % 2 == 0 else "cfc12"])
        for a, b, name in ((0, 0, "cia_n2n2"), (1, 0, "cia_o2n2"), (1, 1, "cia_o2o2")):
            go.add_cia(a, b, band.files[name])
        go.tune(fast=3)
        return go, grid

    go_lw, grid_lw = build(lwb)
    go_sw, grid_sw = build(swb)
    assert grid_lw.n == 32491 and grid_sw.n == 5001
    emis, alb = np.full(lwb.nw, 0.98), np.full(swb.nw, 0.2)
    solar = api.create_solar_flux(grid_sw, swb.files["solar"])
    pipe = api.Pipeline(go_lw, go_sw, chunk, -1, emis, alb, solar, spectral=False)
    got = np.zeros((ncol, api.GRT_FLUXES_PER_COLUMN))
    for first in range(0, ncol, chunk):
        part = [column(c) for c in range(first, first + chunk)]
        gcols, keep = api.make_columns(part, MOL_ORDER, cfc_order=tuple(range(ncfc)))
        pipe.run(gcols)
        got[first: first + chunk] = pipe.fluxes(chunk)
    info = go_lw.last_launch()
    assert info["fast"] == 3 and info["tree_levels"] > 0               # the cell hierarchy carried the 0.1 cm-1 longwave
    assert np.all(np.isfinite(got)) and np.all(got[:, 0] > 100.0) and np.all(got[:, 10] > 100.0)
    worst = 0.0
    for c in (0, 777, 2047):
        col = column(c)
        for bi, (band, lw) in enumerate(((lwb, True), (swb, False))):
            kw = band.oracle_inputs(orc, lib, col)
            kw["cfcs"] = [(col["cfc_ppmv"][k] * 1e-6, band.table_on_grid(orc, "cfc11" if k % 2 == 0 else "cfc12")) for k in range(ncfc)]
            tau_gas = chk.gas_optics(col["p"], col["t"], band.w0, band.dw, band.nw, **kw)
            L = V - 1
            z = np.zeros_like(tau_gas)
            if kind == "reference":
                g = chk.grid(band.w0, band.wn, band.dw)
                tr, om_r, g_r = chk.rayleigh(g, L, col["p"])
                tau, omega, gg = chk.add_optics(g, [tau_gas, tr], [z, om_r], [z, g_r])
                up, dn = (chk.lw_fluxes(g, col["t_surf"], col["t_layer"], col["t"], tau, omega, emis) if lw else
                          chk.sw_fluxes(g, omega, gg, tau, col["mu0"], 0.5, alb, alb, col["tsi"], solar))
            else:
                tr, om_r, g_r = chk.rayleigh(L, col["p"], band.w0, band.dw, band.nw)
                tau, omega, gg = chk.add_optics([tau_gas, tr], [z, om_r], [z, g_r])
                up, dn = (chk.lw_fluxes(band.w0, band.dw, col["t_surf"], col["t_layer"], col["t"], tau, omega, emis) if lw else
                          chk.sw_fluxes(omega, gg, tau, col["mu0"], 0.5, alb, alb, col["tsi"], solar))
            integ = np.array([orc.integrate_row(r, band.dw) for r in (up[0], up[-1], dn[0], dn[-1])])
            worst = max(worst, np.max(np.abs(got[c, bi * 6: bi * 6 + 6][[0, 1, 3, 4]] - integ)))
    print(f"config 5 (ERA5-like, {ncol} columns, LW @0.1 + SW @10 cm-1, 21 CFCs): worst flux difference on sampled columns {worst:.2e} W m-2 ({kind})")
    assert worst < 1e-3
    pipe.destroy()
    go_lw.destroy()
    go_sw.destroy()
