#!/usr/bin/env python3
"""Generate tests/golden/ref_fixtures.npz from the reference's OWN compiled C (oracle/_ref).

Runs only where /root/reference is mounted (`make -C oracle ref` first).  The fixture holds seeded
inputs and the reference's outputs -- numbers only -- so that the oracle restatement stays pinned to
the reference on machines where the reference tree does not exist (the GPU box).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.bindings import Ref  # noqa: E402
from grtcode_amd import synthetic as syn  # noqa: E402


def main():
    ref = Ref()
    out = {}
    rng = np.random.default_rng(42)

    # (i) Voigt K over all five Humlicek regions and both y thresholds (RFM_voigt.c:85-281)
    ys = np.array([1e-7, 1e-3, 0.05, 0.5, 2.0, 5.0, 8.5, 20.0, 69.0, 71.0, 200.0])
    wress = np.array([1e-5, 1e-3, 0.1])
    alpha = 1e-3
    K = np.zeros((ys.size, wress.size, 801))
    for i, y in enumerate(ys):
        for j, wres in enumerate(wress):
            K[i, j] = ref.voigt(1000.0 - 400 * wres, 801, wres, 1000.0 + 0.3 * wres, y * alpha / 0.832554611, alpha)
    out.update(voigt_y=ys, voigt_wres=wress, voigt_alpha=np.array(alpha), voigt_K=K)

    # (ii) prep kernels + line sampling incl. window clipping at both grid edges (kernels.c:34-131,410-465)
    col = syn.profile(7, 7)
    p_atm = col["p"] * np.float64(np.float32(0.000986923))
    n, pavg, tavg = ref.layer_means(p_atm, col["t"])
    x = col["ppmv"][syn.H2O] * 1e-6
    ps, ns = ref.species_means(p_atm, x, n)
    lines = syn.line_list(syn.H2O, 300, 99.5, 150.5)
    lines["iso"] = rng.integers(1, 4, 300).astype(np.int32)
    q = 1.0 / (np.array([174.58, 176.0, 1052.0])[None, :] * (tavg[:, None] / 296.0) ** 1.5)
    mass = float(np.float32(18.010565)) / 6.023e23
    vnn, snn, gamma, alpha_d = ref.line_prep(lines, mass, 3, pavg, tavg, ps, q)
    w0, dw, nw = 100.0, 0.25, 201
    tau = ref.line_sample(vnn, snn, gamma, alpha_d, ns, w0, dw, nw)
    out.update({"ls_" + k: v for k, v in lines.items()})
    out.update(ls_p_atm=p_atm, ls_t=col["t"], ls_x=x, ls_q=q, ls_mass=np.array(mass), ls_n=n, ls_pavg=pavg,
               ls_tavg=tavg, ls_ps=ps, ls_ns=ns, ls_vnn=vnn, ls_snn=snn, ls_gamma=gamma, ls_alpha=alpha_d,
               ls_grid=np.array([w0, dw, nw]), ls_tau=tau)

    # (ii-b) the same prepared lines through the two sweep methods (kernels.c:135-406,514-581) on a grid that
    # extends 29.5 cm-1 above the highest line (line_sweep indexes one bin past its arrays for nearer lines)
    sw_nw = 321
    for method in (0, 1):
        bins = ref.sweep_bins(vnn.shape[0], w0, dw, sw_nw)
        t_sw = ref.sweep(method, bins, vnn, snn, gamma, alpha_d, ns, np.zeros((vnn.shape[0], sw_nw)))
        out["sweep%d_tau" % method] = ref.interpolate(bins, t_sw)
    out.update(sweep_grid=np.array([w0, dw, sw_nw]))

    # (iii) continua / CFC / CIA (kernels.c:469-510,585-630)
    tab = rng.uniform(1e-24, 1e-22, (4, nw))
    tab[2:] = rng.uniform(0.001, 0.03, (2, nw))
    t_h2o = ref.h2o_ctm(np.zeros((6, nw)), tab[1], tavg, ps, ns, tab[3], tab[0], pavg, tab[2])
    xs = rng.uniform(1e-22, 1e-20, nw)
    t_o3 = ref.o3_ctm(np.zeros((6, nw)), xs, ns)
    t_cfc = ref.cfc(np.zeros((6, nw)), n, x, xs)
    t_cia = ref.cia(np.zeros((6, nw)), p_atm, tavg, col["ppmv"][syn.N2] * 1e-6, col["ppmv"][syn.O2] * 1e-6, xs * 1e-24)
    out.update(ct_tab=tab, ct_xs=xs, ct_x1=col["ppmv"][syn.N2] * 1e-6, ct_x2=col["ppmv"][syn.O2] * 1e-6,
               ct_h2o=t_h2o, ct_o3=t_o3, ct_cfc=t_cfc, ct_cia=t_cia)

    # (iv) Rayleigh, add_optics, LW and SW fluxes through the reference's public API
    L = 10
    colw = syn.profile(3, L + 1)
    grid = ref.grid(1.0, 1001.0, 10.0)
    ngrid = grid.n
    r_tau, r_om, r_g = ref.rayleigh(grid, L, colw["p"])
    g_tau = 10.0 ** rng.uniform(-5, 1.5, (L, ngrid))
    z = np.zeros_like(g_tau)
    a_tau, a_om, a_g = ref.add_optics(grid, [g_tau, r_tau], [z, r_om], [z, r_g])
    emis = rng.uniform(0.9, 1.0, ngrid)
    lw_up, lw_dn = ref.lw_fluxes(grid, colw["t_surf"], colw["t_layer"], colw["t"], a_tau, a_om, emis)
    om = rng.uniform(0.0, 0.999, (L, ngrid))
    gg = rng.uniform(-0.5, 0.9, (L, ngrid))
    om[0, :10] = 1.0
    om[-1, 10:20] = 0.0
    alb = rng.uniform(0.0, 0.6, ngrid)
    solar = rng.uniform(0.0, 1e-4, ngrid)
    sw_up, sw_dn = ref.sw_fluxes(grid, om, gg, g_tau, 0.6, 0.5, alb, alb, 1360.0, solar)
    out.update(fx_p=colw["p"], fx_t=colw["t"], fx_tl=colw["t_layer"], fx_ts=np.array(colw["t_surf"]),
               fx_grid=np.array([1.0, 1001.0, 10.0, ngrid]), fx_ray_tau=r_tau, fx_gas_tau=g_tau, fx_add_tau=a_tau,
               fx_add_omega=a_om, fx_add_g=a_g, fx_emis=emis, fx_lw_up=lw_up, fx_lw_dn=lw_dn, fx_sw_omega=om,
               fx_sw_g=gg, fx_alb=alb, fx_solar=solar, fx_sw_up=sw_up, fx_sw_dn=sw_dn)

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_fixtures.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "arrays")
    stochastic_clouds()


def stochastic_clouds():
    """What the reference's clouds/stochastic_clouds.c (oracle/_ref/libstochastic_ref.so, compiled where it lies) returns
    for the cases of tests/test_clouds_library.py::stochastic_cases -- overlap parameter and three successive subcolumns per
    case after srand(seed); doubles as hex strings (bit-exact).  Inputs are regenerated by the test from the same seeds."""
    import json
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_clouds_library as T
    from cloud_model import LibcRand, synthetic_tables
    tables_seed = 5
    with tempfile.TemporaryDirectory() as tmp:
        paths, _ = synthetic_tables(tmp, seed=tables_seed)
        ref = T.reference_sampler()
        assert ref.initialize_clouds_lib(paths["beta"].encode(), paths["ice"].encode(), paths["liquid"].encode()) == 0
        rand = LibcRand()
        cases = []
        for case in T.stochastic_cases():
            out = T.run_sampler(ref, rand, case, reference=True)
            cases.append({"seed": case["seed"], "cf": case["cf"].tolist(), "out": [[float(v).hex() for v in a] for a in out]})
        ref.finalize_clouds_lib()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stochastic_clouds_ref.json")
    json.dump({"what": "overlap_parameter + sample_condensate of /root/reference/clouds/stochastic_clouds.c (built unchanged into "
                       "oracle/_ref/libstochastic_ref.so), three draws per case after srand(seed); beta look-ups from this "
                       "repository's loaded tables (tests/support/beta_bridge.c)",
               "tables_seed": tables_seed, "cases": cases}, open(path, "w"))
    print(path, os.path.getsize(path), "bytes,", len(cases), "cases")


if __name__ == "__main__":
    main()
