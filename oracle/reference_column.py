"""One column of a two-band (longwave + shortwave) workload through the CPU checker.  TEST INFRASTRUCTURE ONLY.

The reference's own C (oracle/_ref/libgrtref_omp.so, OpenMP) where that prebuilt library exists, otherwise our
restatement (oracle/liboracle.so).  Call sequence of framework/src/driver.c:247-270, 381-424, 302-326:
gas optics (line_sample + continua + CFC + CIA) -> Rayleigh -> add_optics -> LW / SW solver -> trapezoid.

Used by bench.py's cpu_baseline leg (timing + the parity of the bench's own column 0) and by tests/ (full-size
parity of the production kernels).  The product never imports this.
"""
import ctypes as C
import os
import time

import numpy as np

from .bindings import Oracle, Ref, ref_available


def set_omp_threads(want):
    """Set the OpenMP thread count of the reference build through libgomp itself (an environment variable set after
    libgomp initialised is ignored) and return what libgomp reports."""
    try:
        gomp = C.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(C.c_int(int(want)))
        return int(gomp.omp_get_max_threads())
    except OSError:
        return 1


def checker(omp=True):
    """(kind, checker, restatement): kind "reference" when oracle/_ref is there, else "port"."""
    orc = Oracle()
    if ref_available(omp=omp):
        return "reference", Ref(omp=omp), orc
    return "port", orc, orc


def band_column(kind, chk, orc, Q, col, grid, lines, tables, mol_order, moltab, mol_mass, cia_pairs, sw, thin=1,
                emissivity=0.98, albedo=0.2, ids=None):
    """One band of one column.  lines: {mol: arrays with tabulated 296 K strengths}; tables: synthetic.tables(sw);
    Q(mol, T, iso) the partition sums to use.  Returns tau_gas [L][n], flux_up/down [V][n], the six integrated
    fluxes in the pipeline's order (up TOA, up surface, 0, down TOA, down surface, 0) and the two stage times."""
    ids = ids or dict(H2O=1, O3=3, N2=22, O2=7)
    w0, wn, dw = grid
    nw = int(np.ceil((wn - w0) / dw)) + 1
    V = col["p"].size
    L = V - 1
    p_atm = col["p"] * np.float64(np.float32(0.000986923))
    _, _, tavg = orc.layer_means(p_atm, col["t"])
    # tables as the CSV files hold them (synthetic.write_csv: "%.6f,%.9e"), so that checker and product see the same numbers
    def on_grid(name):
        w, y = tables[name]
        return orc.interp_to_grid(w0, dw, nw, np.array([float("%.6f" % a) for a in w]), np.array([float("%.9e" % b) for b in y]))
    mols = []
    for m in mol_order:
        ln = {k: v[::thin] for k, v in lines[m].items()}
        niso = moltab[m][1]
        q296 = {int(i): Q(m, 296.0, int(i)) for i in np.unique(ln["iso"])}
        ln["s0"] = orc.rescale_strengths(ln["s0"], ln["en"], ln["v0"], np.array([q296[int(i)] for i in ln["iso"]]))
        q = np.array([[1.0 / Q(m, float(T), k + 1) for k in range(niso)] for T in tavg])
        mols.append(dict(id=m, num_iso=niso, mass=mol_mass(m), lines=ln, x=col["ppmv"][m] * 1e-6, q=q,
                         h2o_ctm=int(m == ids["H2O"]), o3_ctm=int(m == ids["O3"])))
    kw = dict(mols=mols,
              h2o_coefs=[on_grid(k) for k in ("h2o_foreign_296", "h2o_self_296", "h2o_foreign_t", "h2o_self_t")],
              o3_xs=on_grid("o3_ctm"),
              cfcs=[(col["cfc_ppmv"][0] * 1e-6, on_grid("cfc11")), (col["cfc_ppmv"][1] * 1e-6, on_grid("cfc12"))],
              cias=[(col["ppmv"][ids["N2"] if a == 0 else ids["O2"]] * 1e-6, col["ppmv"][ids["N2"] if b == 0 else ids["O2"]] * 1e-6,
                     on_grid(name)) for a, b, name in cia_pairs])
    t0 = time.perf_counter()
    tau_gas = chk.gas_optics(col["p"], col["t"], w0, dw, nw, **kw)
    t_gas = time.perf_counter() - t0
    t0 = time.perf_counter()
    emis, alb = np.full(nw, emissivity), np.full(nw, albedo)
    z = np.zeros_like(tau_gas)
    if kind == "reference":
        g = chk.grid(w0, wn, dw)
        tr, om, gg = chk.rayleigh(g, L, col["p"])
        tau, omega, gsum = chk.add_optics(g, [tau_gas, tr], [z, om], [z, gg])
        if not sw:
            up, dn = chk.lw_fluxes(g, col["t_surf"], col["t_layer"], col["t"], tau, omega, emis)
        else:
            solar = orc.normalize_solar(w0, dw, on_grid("solar"))
            up, dn = chk.sw_fluxes(g, omega, gsum, tau, col["mu0"], 0.5, alb, alb, col["tsi"], solar)
    else:
        tr, om, gg = chk.rayleigh(L, col["p"], w0, dw, nw)
        tau, omega, gsum = chk.add_optics([tau_gas, tr], [z, om], [z, gg])
        if not sw:
            up, dn = chk.lw_fluxes(w0, dw, col["t_surf"], col["t_layer"], col["t"], tau, omega, emis)
        else:
            solar = orc.normalize_solar(w0, dw, on_grid("solar"))
            up, dn = chk.sw_fluxes(omega, gsum, tau, col["mu0"], 0.5, alb, alb, col["tsi"], solar)
    integ = np.array([orc.integrate_row(up[0], dw), orc.integrate_row(up[-1], dw), 0.0,
                      orc.integrate_row(dn[0], dw), orc.integrate_row(dn[-1], dw), 0.0])
    t_rest = time.perf_counter() - t0
    return dict(tau_gas=tau_gas, tau=tau, omega=omega, g=gsum, up=up, dn=dn, integ=integ, t_gas=t_gas, t_rest=t_rest, nw=nw)


def tau_metrics(got, want):
    """How far a tau field is from the checker's, three ways:
      of_layer_max   max |d tau| / (largest tau of that layer)           -- the round-1 metric
      pointwise_rel  max |d tau| / tau  over points with tau > 1e-9 of the layer's largest
      transmission   max |exp(-got) - exp(-want)|                         -- what the solvers see of a layer"""
    d = np.abs(got - want)
    lay_max = np.maximum(np.abs(want).max(axis=1, keepdims=True), 1e-300)
    sel = want > 1e-9 * lay_max
    return {"of_layer_max": float((d / lay_max).max()),
            "pointwise_rel": float((d[sel] / want[sel]).max()) if sel.any() else 0.0,
            "transmission": float(np.abs(np.exp(-got) - np.exp(-want)).max())}
