"""Cloud optics as framework/src/driver.c gets them from the clouds library -- an INDEPENDENT numpy restatement, written
from the reference's sources (clouds/clouds_lib.c:84-139, cloud_pade_optics.c:152-213, optics_utils.c:118-169,
stochastic_clouds.c:11-120, incomplete_beta.c:32-64) for tests only.  The reference's clouds/ cannot be compiled here
(it includes netcdf.h) and holds no test vectors, so this is what grtcode_amd/csrc/host/grt_clouds.c is checked
against: two implementations by formula, NOT a pinned oracle (DESIGN.md says "parity unpinned" for this row).

Random numbers: the library draws rand()/RAND_MAX from libc; `LibcRand` draws from the same libc through ctypes, so a
test that seeds with srand(s) and then calls either implementation sees the same sequence."""
import ctypes

import numpy as np

from grtcode_amd.dumpfile import write_dump


class LibcRand:
    def __init__(self):
        self.libc = ctypes.CDLL("libc.so.6")
        self.libc.rand.restype = ctypes.c_int
        self.rand_max = 2147483647

    def seed(self, s):
        self.libc.srand(ctypes.c_uint(s))

    def __call__(self):
        return self.libc.rand() / self.rand_max


def synthetic_tables(root, seed=3, nband=6, band_edges=None):
    """Parameter files in GRTDUMP1 form with the reference files' variable names; smooth made-up numbers (single
    precision where the reference reads floats).  Returns (paths, tables)."""
    rng = np.random.default_rng(seed)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    # incomplete beta: shapes 1..6, regularised I_x(p, q) by quadrature, inverse by interpolation
    shapes, nx = np.arange(1, 7), 201
    x = np.linspace(0.0, 1.0, nx)
    data = np.zeros((6, 6, nx))
    inverse = np.zeros((6, 6, nx))
    fine = np.linspace(0.0, 1.0, 20001)
    for qi, q in enumerate(shapes):
        for pi, p in enumerate(shapes):
            pdf = fine ** (p - 1) * (1.0 - fine) ** (q - 1)
            cdf = np.concatenate([[0.0], np.cumsum(0.5 * (pdf[1:] + pdf[:-1]) * np.diff(fine))])
            cdf /= cdf[-1]
            data[qi, pi] = np.interp(x, fine, cdf)
            inverse[qi, pi] = np.interp(x, cdf, fine)
    beta = {"p": shapes.astype(float), "q": shapes.astype(float), "x": x, "data": data, "inverse": inverse}
    edges = np.array(band_edges if band_edges is not None else np.linspace(10.0, 3000.0, nband + 1))
    nband = edges.size - 1

    def phase(rmin, rmax, nsize, np_, nq_):
        size_edges = np.linspace(rmin, rmax, nsize + 1)
        t = {"Band_limits_lwr": f32(edges[:-1]), "Band_limits_upr": f32(edges[1:]),
             "Effective_Radius_limits_lwr": f32(size_edges[:-1]), "Effective_Radius_limits_upr": f32(size_edges[1:]),
             "Effective_Radius_Ref": f32(0.5 * (size_edges[:-1] + size_edges[1:]))}
        for name, lead, scale in (("ext", 0.1, 0.02), ("ssa", 0.7, 0.05), ("asy", 0.8, 0.03)):
            p = scale * rng.standard_normal((np_, nsize, nband)) * 1e-2
            p[-1] = lead * (1.0 + 0.2 * rng.random((nsize, nband)))          # constant term (the last Horner coefficient)
            q = scale * rng.standard_normal((nq_, nsize, nband)) * 1e-3
            q[-1] = 1.0
            t[f"Pade_{name}_p"], t[f"Pade_{name}_q"] = f32(p), f32(q)
        return t
    tables = {"beta": beta, "ice": phase(5.0, 60.0, 3, 3, 3), "liquid": phase(2.0, 30.0, 2, 3, 2)}
    paths = {}
    for k, t in tables.items():
        paths[k] = f"{root}/{k}.dump"
        write_dump(paths[k], t)
    return paths, tables


def beta_lookup(beta, rows, p, q, at):
    x, y = beta["x"], rows[q - 1, p - 1]
    i = 1
    while i < x.size - 1 and not (x[i] > at):
        i += 1
    slope = (y[i] - y[i - 1]) / (x[i] - x[i - 1])
    return slope * at + (y[i] - slope * x[i])


def ice_size(t):
    for k, d in enumerate((25., 30., 35., 40., 45., 50., 55.)):
        if t > 273.16 - d:
            return (100.6, 80.8, 93.5, 63.9, 42.5, 39.9, 21.6)[k]
    return 20.2


def pade(t, content, radius, band):
    if not content > 0:
        return 0.0, 0.0, 0.0
    lo, hi, ref = t["Effective_Radius_limits_lwr"], t["Effective_Radius_limits_upr"], t["Effective_Radius_Ref"]
    s = next((k for k in range(lo.size) if lo[k] <= radius <= hi[k]), None)
    if s is None:
        return 0.0, 0.0, 0.0
    dr = radius - ref[s]

    def ratio(name):
        num = den = None
        for c in t[f"Pade_{name}_p"][:, s, band]:
            num = c if num is None else c + dr * num
        for c in t[f"Pade_{name}_q"][:, s, band]:
            den = c if den is None else c + dr * den
        return num / den
    return content * ratio("ext"), ratio("ssa"), ratio("asy")


def spread(t, band, w, values, rows):
    lo, hi = t["Band_limits_lwr"][band], t["Band_limits_upr"][band]
    n = w.size
    first = int(np.searchsorted(w, lo, side="left"))
    last = int(np.searchsorted(w, hi, side="right")) - 1
    nband = t["Band_limits_lwr"].size
    for k, v in enumerate(values):
        if band == 0:
            rows[k][:first] = v
        rows[k][first:max(last, first)] = v
        if band == nband - 1:
            rows[k][max(last, 0):] = v


def cloud_optics(tables, rand, w, cf, lwc, iwc, overlap, liquid_radius, temperature, out=None):
    """-> six [L][n] arrays (beta, omega, g of liquid then ice), starting from `out` (the caller's arrays: points no band
    covers keep what they held) or zeros."""
    L, n = cf.size, w.size
    out = [np.zeros((L, n)) for _ in range(6)] if out is None else out
    beta = tables["beta"]
    p = q = 5
    for band in range(tables["liquid"]["Band_limits_lwr"].size):
        rank = np.array([rand() for _ in range(L)])
        decide = np.array([rand() for _ in range(L - 1)])
        for i in range(L - 1):
            if decide[i] <= overlap[i]:
                rank[i + 1] = rank[i]
        for i in range(L):
            ql = qi = 0.0
            if rank[i] > 1.0 - cf[i]:
                qs = beta_lookup(beta, beta["inverse"], p, q, 1.0 - cf[i])
                width = (lwc[i] + iwc[i]) / ((p / (p + q)) * (1.0 - beta_lookup(beta, beta["data"], p + 1, q, qs)) - qs * cf[i])
                total = width * (beta_lookup(beta, beta["inverse"], p, q, rank[i]) - qs)
                frac = lwc[i] / (lwc[i] + iwc[i])
                ql, qi = total * frac, total * (1.0 - frac)
            spread(tables["liquid"], band, w, pade(tables["liquid"], ql, liquid_radius, band), [a[i] for a in out[:3]])
            spread(tables["ice"], band, w, pade(tables["ice"], qi, ice_size(temperature[i]) / 2.0, band), [a[i] for a in out[3:]])
    return out
