"""Deterministic synthetic inputs for the line-by-line hot path.

No real spectroscopic data ships with the reference (download-test-data needs
network), so benchmarks and parity tests run on seeded synthetic columns, line
lists in the HITRAN 160-column layout, and smooth cross-section tables.  The
shapes and magnitudes follow the reference's formats (Appendix B of SURVEY.md):
HITRAN .par records (gas-optics/src/parse_HITRAN_file.c:77-100) and two-column
CSV tables with one header line (utilities/src/parse_csv.c:55-166).
"""
import os

import numpy as np

# HITRAN molecule ids (gas-optics/src/molecules.h:32-88)
H2O, CO2, O3, N2O, CO, CH4, O2, N2 = 1, 2, 3, 4, 5, 6, 7, 22

#: fraction of the band's lines given to each absorber (200k/300k/250k/200k/30k/15k/5k of 1.0M)
LINE_SHARE = {H2O: 0.20, CO2: 0.30, O3: 0.25, CH4: 0.20, N2O: 0.03, O2: 0.015, CO: 0.005}


def profile(col=0, num_levels=61):
    """One synthetic column: pressures [mb] TOA->surface, temperatures [K], VMRs [ppmv]."""
    V = num_levels
    k = np.arange(V, dtype=np.float64) / (V - 1)
    p = 0.01 + (1013.25 - 0.01) * k ** 2.5
    rng = np.random.default_rng(1000 + col)
    t = 200.0 + 90.0 * k + rng.uniform(-2.0, 2.0, V)
    t_layer = 0.5 * (t[:-1] + t[1:])
    ppmv = {
        H2O: 4.0 * (1.5e4 / 4.0) ** k,                       # 4 ppmv -> 1.5e4 ppmv log-linear
        O3: 0.02 + 8.0 * np.exp(-0.5 * (np.log(p / 10.0) / 0.9) ** 2),
        CO2: np.full(V, 400.0), CH4: np.full(V, 1.8), N2O: np.full(V, 0.33),
        CO: np.full(V, 0.1), O2: np.full(V, 0.209e6), N2: np.full(V, 0.781e6),
    }
    return dict(p=p, t=t, t_layer=t_layer, t_surf=float(t[-1] + 1.0), ppmv=ppmv,
                emissivity=0.98, albedo=0.2, mu0=0.6, mu_dif=0.5, tsi=1360.0,
                cfc_ppmv={0: np.full(V, 2.3e-4), 1: np.full(V, 5.2e-4)})


#: Band structure of the "physically scaled" line lists: per absorber a floor and Gaussian bands
#: (centre [cm-1], width [cm-1], amplitude) multiplying the uniformly drawn strengths, so that the synthetic
#: atmosphere has windows and bands where the real one does (H2O rotation band / 6.3 um / near-infrared bands,
#: CO2 15 and 4.3 um, O3 9.6 um, CH4 7.7 um, N2O, CO, the O2 A band) instead of being black everywhere:
#: outgoing longwave 270 W m-2, surface shortwave 0.68 of the incoming on column 0 of the full-size lists (cf. the LBLRTM numbers quoted by
#: circ/src/basic-circ-test.c:447-495).  The SURVEY §8(d) list (physical=False) stays the bench workload.
PHYSICAL_BANDS = {
    H2O: (1e-6, [(100.0, 170.0, 0.6), (1595.0, 120.0, 0.4), (3750.0, 180.0, 0.7), (5350.0, 180.0, 0.7), (7250.0, 180.0, 0.7),
                 (8800.0, 180.0, 0.1), (10600.0, 180.0, 0.1), (12200.0, 180.0, 0.01), (13800.0, 180.0, 0.01)]),
    CO2: (1e-8, [(667.0, 30.0, 1.0), (2349.0, 40.0, 3.0), (960.0, 30.0, 1e-4), (1064.0, 30.0, 1e-4), (3700.0, 60.0, 0.1),
                 (5000.0, 80.0, 0.01), (6300.0, 80.0, 1e-3)]),
    O3: (1e-6, [(1042.0, 30.0, 5.0), (701.0, 25.0, 0.3), (2110.0, 30.0, 0.1)]),
    N2O: (1e-6, [(1285.0, 25.0, 10.0), (2224.0, 25.0, 100.0), (589.0, 20.0, 3.0)]),
    CO: (1e-6, [(2143.0, 40.0, 30.0)]),
    CH4: (1e-6, [(1306.0, 50.0, 1.0), (3019.0, 60.0, 1.0), (4300.0, 100.0, 0.3), (6000.0, 100.0, 0.05)]),
    O2: (1e-10, [(1556.0, 50.0, 1e-7), (13120.0, 30.0, 1e-3), (14500.0, 30.0, 1e-4)]),
}


def physical_envelope(mol_id, v0):
    floor, bands = PHYSICAL_BANDS[mol_id]
    e = np.full(v0.shape, floor)
    for c, w, a in bands:
        e += a * np.exp(-0.5 * ((v0 - c) / w) ** 2)
    return e


def line_list(mol_id, n, w0, wn, seed=20261003, physical=False):
    """Synthetic line parameters for one molecule, sorted by centre.

    yair/yself/en/n/delta carry f32-representable values because the reference
    reads those columns through a float (parse_HITRAN_file.c:197-212).
    physical=True: strengths follow PHYSICAL_BANDS (windows and bands) instead of being uniform over the band."""
    rng = np.random.default_rng(seed + 7919 * mol_id)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    v0 = np.sort(np.round(rng.uniform(w0, wn, n), 6))
    return dict(
        v0=v0,
        s0=10.0 ** rng.uniform(-27.0, -20.0, n) * (physical_envelope(mol_id, v0) if physical else 1.0),
        yair=f32(np.round(rng.uniform(0.02, 0.10, n), 4)),
        yself=f32(np.round(rng.uniform(0.1, 0.5, n), 3)),
        en=f32(np.round(rng.uniform(0.0, 4000.0, n), 4)),
        nexp=f32(np.round(rng.uniform(0.4, 0.8, n), 2)),
        delta=f32(np.round(rng.uniform(-0.02, 0.02, n), 6)),
        iso=np.ones(n, dtype=np.int32),
    )


def band_line_lists(total_lines, w0, wn, seed=20261003, physical=False):
    """Per-molecule line lists for one band with the LINE_SHARE split."""
    return {m: line_list(m, max(int(round(total_lines * s)), 1), w0, wn, seed, physical)
            for m, s in LINE_SHARE.items()}


def _fw(v, width, prec):
    """Fortran-style fixed-width float: drop the leading zero when it does not fit (F5.4 -> '.0834')."""
    s = "%.*f" % (prec, v)
    if len(s) > width:
        s = s.replace("0.", ".", 1)
    assert len(s) <= width, (s, width)
    return s.rjust(width)


def write_hitran_par(path, lists):
    """Write {mol_id: lines} as fixed-width HITRAN-2012 160-char records, sorted by centre.

    S is stored as the raw (un-rescaled) 296 K strength, as in a real .par file."""
    recs = []
    for mol, ln in lists.items():
        for j in range(ln["v0"].size):
            recs.append((ln["v0"][j], mol, ln, j))
    recs.sort(key=lambda r: r[0])
    with open(path, "w") as f:
        for v, mol, ln, j in recs:
            iso = int(ln["iso"][j])
            iso_c = "0" if iso == 10 else (chr(ord("A") + iso - 11) if iso > 10 else str(iso))
            s = "%2d%1s%12.6f%10.3E%10.3E" % (mol, iso_c, v, ln["s0"][j], 0.0)
            s += _fw(ln["yair"][j], 5, 4) + _fw(ln["yself"][j], 5, 3) + "%10.4f" % ln["en"][j]
            s += _fw(ln["nexp"][j], 4, 2) + _fw(ln["delta"][j], 8, 6)
            assert len(s) == 67, (len(s), s)
            f.write(s + " " * 93 + "\n")


def read_back_par_values(lists):
    """Round line parameters to what survives the .par text format (so that arrays
    fed directly to a checker equal what a loader parses from write_hitran_par)."""
    out = {}
    for mol, ln in lists.items():
        o = dict(ln)
        o["s0"] = np.array([float("%10.3E" % s) for s in ln["s0"]])
        f32 = lambda a: a.astype(np.float32).astype(np.float64)
        o["yair"] = f32(np.array([float(_fw(v, 5, 4)) for v in ln["yair"]]))
        o["yself"] = f32(np.array([float(_fw(v, 5, 3)) for v in ln["yself"]]))
        o["en"] = f32(np.array([float("%10.4f" % v) for v in ln["en"]]))
        o["nexp"] = f32(np.array([float(_fw(v, 4, 2)) for v in ln["nexp"]]))
        o["delta"] = f32(np.array([float(_fw(v, 8, 6)) for v in ln["delta"]]))
        o["v0"] = np.array([float("%12.6f" % v) for v in ln["v0"]])
        out[mol] = o
    return out


# ---- smooth analytic cross-section tables (CSV: header + "wavenumber,value") ---- #
def _table(w_lo, w_hi, step, fn):
    w = np.arange(w_lo, w_hi + 0.5 * step, step)
    return w, fn(w)


def tables(sw=False):
    """Analytic stand-ins for the data bundle's CSV files (same roles, smooth shapes)."""
    hi = 50000.0 if sw else 3500.0
    step = 50.0 if sw else 5.0
    t = {}
    t["h2o_self_296"] = _table(0.0, hi, step, lambda w: 2.0e-22 * np.exp(-w / 600.0) + 1.0e-27)
    t["h2o_foreign_296"] = _table(0.0, hi, step, lambda w: 4.0e-24 * np.exp(-w / 450.0) + 1.0e-29)
    t["h2o_self_t"] = _table(0.0, hi, step, lambda w: 0.02 + 0.01 * np.exp(-w / 2000.0))
    t["h2o_foreign_t"] = _table(0.0, hi, step, lambda w: 0.002 + 0.001 * np.exp(-w / 2000.0))
    t["o3_ctm"] = _table(10.0, hi, step, lambda w: 1.0e-21 * np.exp(-0.5 * ((w - (30000.0 if sw else 1040.0)) / (4000.0 if sw else 60.0)) ** 2))
    t["cfc11"] = _table(700.0, 1300.0, 1.0, lambda w: 2.0e-18 * np.exp(-0.5 * ((w - 850.0) / 15.0) ** 2))
    t["cfc12"] = _table(700.0, 1300.0, 1.0, lambda w: 3.0e-18 * np.exp(-0.5 * ((w - 920.0) / 12.0) ** 2))
    t["cia_n2n2"] = _table(1.0, 400.0 if not sw else 5000.0, 2.0, lambda w: 1.0e-46 * np.exp(-0.5 * ((w - 100.0) / 60.0) ** 2))
    t["cia_o2n2"] = _table(1200.0, 1900.0 if not sw else 9000.0, 2.0, lambda w: 2.0e-46 * np.exp(-0.5 * ((w - 1560.0) / 80.0) ** 2))
    t["cia_o2o2"] = _table(1200.0, 1900.0 if not sw else 30000.0, 2.0, lambda w: 3.0e-46 * np.exp(-0.5 * ((w - 1560.0) / 80.0) ** 2))
    # Planck-like 5772 K solar shape [arbitrary units]; normalised by create_solar_flux
    t["solar"] = _table(0.5, 50010.0, 10.0, lambda w: w ** 3 / np.expm1(1.4387773538277202 * w / 5772.0))
    return t


def write_csv(path, w, y, header="wavenumber,value", extra_cols=0):
    with open(path, "w") as f:
        f.write(header + ",x" * extra_cols + "\n")
        for a, b in zip(w, y):
            f.write("%.6f,%.9e" % (a, b) + ",0" * extra_cols + "\n")


def write_data_bundle(root, lw_lines, sw_lines, seed=20261003,
                      lw_band=(1.0, 3250.0), sw_band=(1.0, 50000.0)):
    """Lay out a synthetic 'grtcode-data'-style directory; returns dict of paths + the line lists."""
    os.makedirs(os.path.join(root, "water_vapor_continuum"), exist_ok=True)
    out = {"root": root}
    lists = {}
    for band, nlines, rng_ in (("lw", lw_lines, lw_band), ("sw", sw_lines, sw_band)):
        if nlines <= 0:
            continue
        ll = band_line_lists(nlines, rng_[0], rng_[1], seed + (0 if band == "lw" else 1))
        path = os.path.join(root, f"hitran_{band}.par")
        write_hitran_par(path, ll)
        out[f"hitran_{band}"] = path
        lists[band] = read_back_par_values(ll)
    for band in ("lw", "sw"):
        t = tables(sw=(band == "sw"))
        d = os.path.join(root, f"water_vapor_continuum_{band}")
        os.makedirs(d, exist_ok=True)
        write_csv(os.path.join(d, "296MTCKD25_F.csv"), *t["h2o_foreign_296"])
        write_csv(os.path.join(d, "296MTCKD25_S.csv"), *t["h2o_self_296"])
        write_csv(os.path.join(d, "CKDF.csv"), *t["h2o_foreign_t"], extra_cols=2)
        write_csv(os.path.join(d, "CKDS.csv"), *t["h2o_self_t"], extra_cols=2)
        out[f"h2o_ctm_{band}"] = d
        for name in ("o3_ctm", "cfc11", "cfc12", "cia_n2n2", "cia_o2n2", "cia_o2o2"):
            p = os.path.join(root, f"{name}_{band}.csv")
            write_csv(p, *t[name])
            out[f"{name}_{band}"] = p
    p = os.path.join(root, "solar_flux.csv")
    write_csv(p, *tables(sw=True)["solar"])
    out["solar"] = p
    out["lines"] = lists
    return out
