"""The batched device-resident pipeline (grt_pipeline_*) against the oracle's column-by-column
restatement of driver.c:360-424 + 285-356, plus size-independent properties at larger sizes."""
import numpy as np
import pytest

from grtcode_amd import api, synthetic as syn
from scenario import Band, MOL_ORDER

pytestmark = pytest.mark.gpu

FLUX_TOL = 1e-3      # W m-2, BASELINE.json north_star tolerance on broadband fluxes


def oracle_column(orc, lib, band, col, lw, emis=None, alb=None, solar=None, user_level=-1):
    L = col["p"].size - 1
    tau_gas = band.oracle_tau(orc, orc, lib, col)
    tr, om_r, g_r = orc.rayleigh(L, col["p"], band.w0, band.dw, band.nw)
    z = np.zeros_like(tau_gas)
    tau, omega, g = orc.add_optics([tau_gas, tr], [z, om_r], [z, g_r])
    if lw:
        up, dn = orc.lw_fluxes(band.w0, band.dw, col["t_surf"], col["t_layer"], col["t"], tau, omega, emis)
    else:
        up, dn = orc.sw_fluxes(omega, g, tau, col["mu0"], 0.5, alb, alb, col["tsi"], solar)
    rows = [up[0], up[-1], up[user_level] if user_level >= 0 else None,
            dn[0], dn[-1], dn[user_level] if user_level >= 0 else None]
    integ = [orc.integrate_row(r, band.dw) if r is not None else 0.0 for r in rows]
    return dict(tau_gas=tau_gas, tau=tau, omega=omega, g=g, up=up, dn=dn, integ=np.array(integ))


@pytest.fixture(scope="module")
def bands(tmp_path_factory):
    root = tmp_path_factory.mktemp("pipe")
    lw = Band(str(root / "lw"), 1.0, 400.0, 1.0, 3000)
    sw = Band(str(root / "sw"), 1.0, 5000.0, 10.0, 3000, sw=True)
    return lw, sw


def test_pipeline_matches_oracle_per_column(bands, oracle, lib, device):
    lwb, swb = bands
    V, ncol, user_level = 16, 3, 5
    cols = [syn.profile(c, V) for c in range(ncol)]
    go_lw, grid_lw = lwb.gas_optics(device, V)
    go_sw, grid_sw = swb.gas_optics(device, V)
    emis = np.full(lwb.nw, 0.98)
    alb = np.full(swb.nw, 0.2)
    solar = api.create_solar_flux(grid_sw, swb.files["solar"])
    w, y = swb._csv_values(*swb.tab["solar"])
    want_solar = oracle.normalize_solar(swb.w0, swb.dw, oracle.interp_to_grid(swb.w0, swb.dw, swb.nw, w, y))
    assert np.max(np.abs(solar - want_solar)) <= 1e-15 * want_solar.max()
    pipe = api.Pipeline(go_lw, go_sw, ncol, user_level, emis, alb, solar)
    gcols, keep = api.make_columns(cols, MOL_ORDER, cfc_order=(0, 1))
    pipe.run(gcols)
    got = pipe.fluxes(ncol)
    L = V - 1
    for bi, (band, lw) in enumerate(((lwb, True), (swb, False))):
        v = pipe.views(bi)
        tau_gas = api.device_to_host(device, v["tau_gas"], (ncol, L, band.nw))
        tau = api.device_to_host(device, v["tau"], (ncol, L, band.nw))
        omega = api.device_to_host(device, v["omega"], (ncol, L, band.nw))
        up = api.device_to_host(device, v["flux_up"], (ncol, V, band.nw))
        dn = api.device_to_host(device, v["flux_down"], (ncol, V, band.nw))
        for c, col in enumerate(cols):
            w = oracle_column(oracle, lib, band, col, lw, emis, alb, solar, user_level)
            scale = np.abs(w["tau_gas"]).max(axis=1, keepdims=True)
            assert np.max(np.abs(tau_gas[c] - w["tau_gas"]) / scale) < 1e-11
            assert np.max(np.abs(tau[c] - w["tau"]) / np.abs(w["tau"]).max(axis=1, keepdims=True)) < 1e-11
            assert np.max(np.abs(omega[c] - w["omega"])) < 1e-11
            fs = max(np.abs(w["up"]).max(), np.abs(w["dn"]).max())
            assert np.max(np.abs(up[c] - w["up"])) / fs < 1e-10
            assert np.max(np.abs(dn[c] - w["dn"])) / fs < 1e-10
            assert np.max(np.abs(got[c, bi * 6: bi * 6 + 6] - w["integ"])) < FLUX_TOL * 1e-6   # far inside 1e-3 W m-2
    pipe.destroy()
    go_lw.destroy()
    go_sw.destroy()


def test_pipeline_properties_full_width_band(tmp_path, lib, device):
    """Size-independent properties on the full 1 cm-1 longwave grid (n = 3250, 60 layers):
    tau is additive over absorbers and linear in abundance; batched == one-by-one."""
    band = Band(str(tmp_path), 1.0, 3250.0, 1.0, 30000, with_cfc=False, with_cia=False, with_ctm=False)
    V = 61
    col = syn.profile(9, V)

    def tau_for(mols, scale=1.0):
        b = Band.__new__(Band)
        b.__dict__.update(band.__dict__)
        b.mols = mols
        go, grid = b.gas_optics(device, V, from_file=False)
        c = dict(col)
        c["ppmv"] = {k: v * scale for k, v in col["ppmv"].items()}
        b.set_column(go, c)
        opt = api.OpticsObject(V - 1, grid, device)
        go.calculate_optical_depth(c["p"], c["t"], opt)
        t = opt.read()[0]
        opt.destroy()
        go.destroy()
        return t

    t_all = tau_for([syn.CO2, syn.CH4])
    t_a, t_b = tau_for([syn.CO2]), tau_for([syn.CH4])
    scale = t_all.max(axis=1, keepdims=True)
    assert np.max(np.abs(t_all - (t_a + t_b)) / scale) < 1e-12
    # CH4 is trace (ps << p): doubling its abundance doubles N_s, and changes gamma only through ps
    t_b2 = tau_for([syn.CH4], scale=2.0)
    assert np.max(np.abs(t_b2 - 2.0 * t_b) / t_b.max(axis=1, keepdims=True)) < 1e-4
    assert np.all(t_all >= 0) and np.all(np.isfinite(t_all))


def test_pipeline_fast_form_fluxes_within_north_star_tolerance(bands, oracle, lib, device):
    """The production (fused) arithmetic form: integrated fluxes within 1e-3 W m-2 of the oracle
    (BASELINE.json north_star) -- in practice ~1e-6 -- and spectral tau within 2e-6 of the layer maximum."""
    lwb, swb = bands
    V, ncol = 16, 2
    cols = [syn.profile(40 + c, V) for c in range(ncol)]
    go_lw, grid_lw = lwb.gas_optics(device, V)
    go_sw, grid_sw = swb.gas_optics(device, V)
    go_lw.tune(fast=1)
    go_sw.tune(fast=1)
    emis, alb = np.full(lwb.nw, 0.98), np.full(swb.nw, 0.2)
    solar = api.create_solar_flux(grid_sw, swb.files["solar"])
    pipe = api.Pipeline(go_lw, go_sw, ncol, -1, emis, alb, solar)
    gcols, keep = api.make_columns(cols, MOL_ORDER, cfc_order=(0, 1))
    pipe.run(gcols)
    got = pipe.fluxes(ncol)
    for bi, (band, lw) in enumerate(((lwb, True), (swb, False))):
        tau_gas = api.device_to_host(device, pipe.views(bi)["tau_gas"], (ncol, V - 1, band.nw))
        for c, col in enumerate(cols):
            w = oracle_column(oracle, lib, band, col, lw, emis, alb, solar)
            scale = np.abs(w["tau_gas"]).max(axis=1, keepdims=True)
            assert np.max(np.abs(tau_gas[c] - w["tau_gas"]) / scale) < 2e-6
            assert np.max(np.abs(got[c, bi * 6: bi * 6 + 6] - w["integ"])) < FLUX_TOL
            assert got[c, bi * 6 + 2] == 0.0 and got[c, bi * 6 + 5] == 0.0      # no user level requested
    pipe.destroy()
    go_lw.destroy()
    go_sw.destroy()


@pytest.mark.parametrize("user_level", [-1, 0, 7, 15])
def test_fused_production_pipeline_equals_materialised_pipeline(bands, oracle, lib, device, user_level):
    """grt_pipeline_create (production: Rayleigh + optics combination + solver + trapezoid in one kernel per band, nothing
    spectral kept) against grt_pipeline_create_ex(keep_spectra = 1) (tau/omega/g and [level][wavenumber] fluxes as the
    reference's calls leave them, row-wise trapezoid) and against the oracle: same per-wavenumber arithmetic, so the
    integrated fluxes differ only by the order of the spectral sum."""
    lwb, swb = bands
    V, ncol = 16, 3
    cols = [syn.profile(20 + c, V) for c in range(ncol)]
    go_lw, grid_lw = lwb.gas_optics(device, V)
    go_sw, grid_sw = swb.gas_optics(device, V)
    emis, alb = np.full(lwb.nw, 0.98), np.full(swb.nw, 0.2)
    solar = api.create_solar_flux(grid_sw, swb.files["solar"])
    gcols, keep = api.make_columns(cols, MOL_ORDER, cfc_order=(0, 1))
    got = {}
    for spectral in (True, False):
        pipe = api.Pipeline(go_lw, go_sw, ncol, user_level, emis, alb, solar, spectral=spectral)
        pipe.run(gcols)
        got[spectral] = pipe.fluxes(ncol)
        if not spectral:
            pipe.run(gcols)
            # reference-order form: tau is summed in fp64, line slices add with atomics in any order (~1e-16 relative),
            # and the flux reduction itself is ordered; bit-identical in the deterministic mode
            again = pipe.fluxes(ncol)
            assert np.max(np.abs(again - got[False])) < 1e-12 * np.abs(got[False]).max()
            api.check(lib.grt_set_deterministic(1))
            try:
                pipe.run(gcols)
                det = pipe.fluxes(ncol)
                pipe.run(gcols)
                assert np.array_equal(pipe.fluxes(ncol), det)
                assert np.max(np.abs(det - got[False])) < 1e-12 * np.abs(got[False]).max()
            finally:
                api.check(lib.grt_set_deterministic(-1))
            with pytest.raises(api.GrtError):                       # nothing spectral is kept in this form
                ptrs = [api.C.c_void_p() for _ in range(6)]
                api.check(lib.grt_pipeline_views(pipe.p, 0, *[api.C.byref(p) for p in ptrs]))
            assert pipe.views(1)["tau_gas"]                         # ... except tau_gas
        pipe.destroy()
    scale = np.abs(got[True]).max()
    assert np.max(np.abs(got[True] - got[False])) < 1e-12 * scale
    for c, col in enumerate(cols):
        for bi, (band, lw) in enumerate(((lwb, True), (swb, False))):
            w = oracle_column(oracle, lib, band, col, lw, emis, alb, solar, user_level)
            assert np.max(np.abs(got[False][c, bi * 6: bi * 6 + 6] - w["integ"])) < 1e-9
    if user_level < 0:
        assert np.all(got[False][:, [2, 5, 8, 11]] == 0.0)
    go_lw.destroy()
    go_sw.destroy()


@pytest.mark.parametrize("user_level", [-1, 0, 15])
def test_one_sweep_shortwave_solver_equals_the_two_sweeps(bands, lib, device, user_level, monkeypatch):
    """The fused shortwave solver with no flux asked for between top and surface runs ONE sweep from the top (k_shortwave.hip:
    the slab's direct-beam reflectance and its upward diffuse transmission ride along with the reference's downward sweep)
    instead of the reference's two with the layer properties parked in between (GRT_SW_TWO_SWEEPS=1).  The surface fluxes
    are the reference's own operations -- the same doubles; the top's upward flux is the adding method's identity for what
    the upward sweep builds -- the same number to rounding."""
    lwb, swb = bands
    V, ncol = 16, 4
    cols = [syn.profile(40 + c, V) for c in range(ncol)]
    go_lw, grid_lw = lwb.gas_optics(device, V)
    go_sw, grid_sw = swb.gas_optics(device, V)
    emis, alb = np.full(lwb.nw, 0.98), np.full(swb.nw, 0.35)
    solar = api.create_solar_flux(grid_sw, swb.files["solar"])
    gcols, keep = api.make_columns(cols, MOL_ORDER, cfc_order=(0, 1))
    api.check(lib.grt_set_deterministic(1))
    try:
        got = {}
        for two in ("0", "1"):
            monkeypatch.setenv("GRT_SW_TWO_SWEEPS", two)
            pipe = api.Pipeline(go_lw, go_sw, ncol, user_level, emis, alb, solar, spectral=False)
            pipe.run(gcols)
            got[two] = pipe.fluxes(ncol)
            pipe.destroy()
    finally:
        api.check(lib.grt_set_deterministic(-1))
    one, ref = got["0"], got["1"]
    assert np.array_equal(one[:, :6], ref[:, :6])                                   # longwave: untouched
    sw_one, sw_ref = one[:, 6:], ref[:, 6:]                                         # up top, up sfc, up user, down top, down sfc, down user
    assert np.array_equal(sw_one[:, [1, 3, 4]], sw_ref[:, [1, 3, 4]])               # surface up/down, top down: the same doubles
    assert np.max(np.abs(sw_one - sw_ref)) <= 1e-13*np.abs(sw_ref).max()
    assert np.all(sw_ref[:, 0] > 0.0)
    go_lw.destroy()
    go_sw.destroy()


@pytest.mark.parametrize("fast", [0, 3])
def test_tables_part_of_tau_added_by_the_shortwave_solver_is_the_same_doubles(bands, lib, device, fast, monkeypatch):
    """The production pipeline's shortwave gas-optics launch leaves the spectral tables' part of tau (continua, CFC, CIA) to
    the solver kernel, which reads a table entry once per point (GrtGasOpticsArgs.skip_tables, GrtContinua) -- the same
    expressions in the same order: fluxes AND the tau_gas a caller looks at afterwards (grt_pipeline_views completes it)
    are bit for bit those of a run that adds the tables in the gas-optics kernel (GRT_DEFER_CONTINUA=0).  The gas-optics
    object's own entry point still delivers the whole tau after it served a pipeline."""
    lwb, swb = bands
    V, ncol = 16, 3
    cols = [syn.profile(60 + c, V) for c in range(ncol)]
    go_lw, grid_lw = lwb.gas_optics(device, V)
    go_sw, grid_sw = swb.gas_optics(device, V)
    go_lw.tune(fast=fast)
    go_sw.tune(fast=fast)
    emis, alb = np.full(lwb.nw, 0.98), np.full(swb.nw, 0.2)
    solar = api.create_solar_flux(grid_sw, swb.files["solar"])
    gcols, keep = api.make_columns(cols, MOL_ORDER, cfc_order=(0, 1))
    api.check(lib.grt_set_deterministic(1))
    try:
        flux, tau = {}, {}
        for defer in ("1", "0"):
            monkeypatch.setenv("GRT_DEFER_CONTINUA", defer)
            pipe = api.Pipeline(go_lw, go_sw, ncol, -1, emis, alb, solar, spectral=False)
            pipe.run(gcols)
            flux[defer] = pipe.fluxes(ncol)
            tau[defer] = api.device_to_host(device, pipe.views(1)["tau_gas"], (ncol, V - 1, swb.nw)).copy()
            again = api.device_to_host(device, pipe.views(1)["tau_gas"], (ncol, V - 1, swb.nw))
            assert np.array_equal(again, tau[defer])                # (looked at twice: completed once)
            pipe.destroy()
        assert np.array_equal(flux["1"], flux["0"])
        assert np.array_equal(tau["1"], tau["0"])
        with_tables = tau["0"][0]
        swb.set_column(go_sw, cols[0])
        opt = api.OpticsObject(V - 1, grid_sw, device)
        go_sw.calculate_optical_depth(cols[0]["p"], cols[0]["t"], opt)
        direct = opt.read()[0]
        opt.destroy()
        assert np.array_equal(direct, with_tables)
        swb_plain = np.abs(with_tables).max()
        assert swb_plain > 0.0
    finally:
        api.check(lib.grt_set_deterministic(-1))
    go_lw.destroy()
    go_sw.destroy()


def test_two_pipelines_on_two_lanes_equal_one(bands, lib, device):
    """grt_device_use_lane (grt_ext.h): two pipelines with gas-optics objects of their own, each on a stream of its own,
    batches alternating between them without a wait in between -- the fluxes are those of one pipeline run batch by batch."""
    lwb, swb = bands
    V, ncol, nbatch = 16, 3, 4
    emis, alb = np.full(lwb.nw, 0.98), np.full(swb.nw, 0.2)
    batches = [api.make_columns([syn.profile(40 + 3 * b + c, V) for c in range(ncol)], MOL_ORDER, cfc_order=(0, 1)) for b in range(nbatch)]
    made = []
    for lane in (0, 1):
        api.use_lane(device, lane)
        go_lw, _ = lwb.gas_optics(device, V)
        go_sw, grid_sw = swb.gas_optics(device, V)
        go_lw.tune(fast=3)
        go_sw.tune(fast=3)
        solar = api.create_solar_flux(grid_sw, swb.files["solar"])
        made.append((api.Pipeline(go_lw, go_sw, ncol, -1, emis, alb, solar, spectral=False), go_lw, go_sw))
    out = [api.DeviceBuffer(device, 8 * ncol * api.GRT_FLUXES_PER_COLUMN) for _ in range(nbatch)]
    try:
        api.check(lib.grt_set_deterministic(1))                 # (so that "equal" can mean equal)
        for b, (gcols, keep) in enumerate(batches):             # in flight together: no sync between the launches
            api.use_lane(device, b % 2)
            made[b % 2][0].run(gcols, out[b].ptr)
        api.device_synchronize(device)
        two = [o.to_host((ncol, api.GRT_FLUXES_PER_COLUMN)) for o in out]
        api.use_lane(device, 0)
        for b, (gcols, keep) in enumerate(batches):
            made[0][0].run(gcols, out[b].ptr)
            made[0][0].sync()
            assert np.array_equal(out[b].to_host((ncol, api.GRT_FLUXES_PER_COLUMN)), two[b]), b
        assert np.all(two[0][:, 0] > 0) and not np.array_equal(two[0], two[1])
        # a pipeline's own sync() must wait for ITS work whichever lane is selected now: run the lane-1 pipeline,
        # select lane 0, sync the lane-1 pipeline, read -- no device-wide wait in between
        api.use_lane(device, 1)
        made[1][0].run(batches[1][0], out[0].ptr)
        api.use_lane(device, 0)
        made[1][0].sync()
        assert np.array_equal(out[0].to_host((ncol, api.GRT_FLUXES_PER_COLUMN)), two[1])
        # ... and a pipeline is not RUN on another lane than its own (the stream it hands out for the gather would
        # not be the one carrying its kernels): a clean error, nothing queued
        with pytest.raises(Exception, match="lane"):
            made[1][0].run(batches[1][0], out[0].ptr)
    finally:
        api.check(lib.grt_set_deterministic(-1))
        api.use_lane(device, 0)
    for o in out:
        o.free()
    for pipe, go_lw, go_sw in made:
        pipe.destroy()
        go_lw.destroy()
        go_sw.destroy()
