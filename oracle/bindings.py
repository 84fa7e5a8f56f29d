"""ctypes bindings for the two CPU checkers.  TEST INFRASTRUCTURE ONLY.

* ``Oracle``  -> oracle/liboracle.so   (our restatement, grt_oracle.c)
* ``Ref``     -> oracle/_ref/libgrtref[_omp].so (the reference's own C sources
  compiled in place by oracle/Makefile; present in the build container and
  shipped prebuilt to the GPU box, absent from git history)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product package (grtcode_amd) never does.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# --------------------------------------------------------------------------- #
# Oracle (restatement)
# --------------------------------------------------------------------------- #
class OrcMolecule(C.Structure):
    _fields_ = [("id", C.c_int), ("num_iso", C.c_int), ("mass", C.c_double),
                ("num_lines", C.c_uint64),
                ("v0", c_double_p), ("s0", c_double_p), ("yair", c_double_p),
                ("yself", c_double_p), ("en", c_double_p), ("nexp", c_double_p),
                ("delta", c_double_p), ("iso", c_int_p), ("x", c_double_p),
                ("q", c_double_p), ("h2o_ctm", C.c_int), ("o3_ctm", C.c_int)]


class OrcBins(C.Structure):              # oracle/grt_oracle.h
    _fields_ = [("num_layers", C.c_int), ("w0", C.c_double), ("wres", C.c_double), ("width", C.c_double),
                ("num_wpoints", C.c_uint64), ("n", C.c_uint64), ("isize", C.c_uint64),
                ("ppb", C.c_int), ("do_interp", C.c_int), ("last_ppb", C.c_int), ("do_last_interp", C.c_int),
                ("w", c_double_p), ("tau", c_double_p), ("l", C.POINTER(C.c_uint64)), ("r", C.POINTER(C.c_uint64))]


class Oracle:
    """numpy-level wrapper over liboracle.so."""

    def __init__(self, path=None):
        path = path or os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle oracle`")
        self.lib = C.CDLL(path)
        self.lib.orc_integrate_row.restype = C.c_double

    # -- layer means -------------------------------------------------------- #
    def layer_means(self, p_atm, t):
        p_atm, t = _f64(p_atm), _f64(t)
        L = p_atm.size - 1
        n, pavg, tavg = np.zeros(L), np.zeros(L), np.zeros(L)
        self.lib.orc_number_densities(C.c_int(L), _dp(p_atm), _dp(n))
        self.lib.orc_pressures_and_temperatures(C.c_int(L), _dp(p_atm), _dp(t), _dp(pavg), _dp(tavg))
        return n, pavg, tavg

    def species_means(self, p_atm, x, n):
        p_atm, x, n = _f64(p_atm), _f64(x), _f64(n)
        L = n.size
        ps, ns = np.zeros(L), np.zeros(L)
        self.lib.orc_partial_pressures_and_number_densities(C.c_int(L), _dp(p_atm), _dp(x), _dp(n), _dp(ps), _dp(ns))
        return ps, ns

    def line_prep(self, lines, mass, num_iso, pavg, tavg, ps, q):
        """lines: dict of v0,delta,s0,en,iso,nexp,yair,yself. Returns (vnn,snn,gamma,alpha) (L,N)."""
        N = lines["v0"].size
        L = pavg.size
        a = {k: _f64(lines[k]) for k in ("v0", "delta", "s0", "en", "nexp", "yair", "yself")}
        iso = np.ascontiguousarray(lines["iso"], dtype=np.int32)
        pavg, tavg, ps, q = _f64(pavg), _f64(tavg), _f64(ps), _f64(q)
        out = [np.zeros((L, N)) for _ in range(4)]
        self.lib.orc_line_prep(C.c_uint64(N), C.c_int(L), C.c_int(num_iso), C.c_double(mass),
                               _dp(a["v0"]), _dp(a["delta"]), _dp(a["s0"]), _dp(a["en"]),
                               iso.ctypes.data_as(c_int_p), _dp(a["nexp"]), _dp(a["yair"]),
                               _dp(a["yself"]), _dp(pavg), _dp(tavg), _dp(ps), _dp(q),
                               *[_dp(o) for o in out])
        return out

    def voigt(self, w_start, npts, wres, center, gamma, alpha):
        K = np.zeros(npts)
        self.lib.orc_voigt(C.c_double(w_start), C.c_uint64(npts), C.c_double(wres),
                           C.c_double(center), C.c_double(gamma), C.c_double(alpha), _dp(K))
        return K

    def line_sample(self, vnn, snn, gamma, alpha, ns, w0, wres, nw, windows=False):
        vnn, snn, gamma, alpha, ns = map(_f64, (vnn, snn, gamma, alpha, ns))
        L, N = vnn.shape
        tau = np.zeros((L, nw))
        ws = np.zeros((L, N), dtype=np.int64)
        we = np.zeros((L, N), dtype=np.int64)
        self.lib.orc_line_sample(C.c_uint64(N), C.c_int(L), _dp(vnn), _dp(snn), _dp(gamma),
                                 _dp(alpha), _dp(ns), C.c_double(w0), C.c_double(wres),
                                 C.c_uint64(nw), _dp(tau),
                                 ws.ctypes.data_as(C.POINTER(C.c_int64)),
                                 we.ctypes.data_as(C.POINTER(C.c_int64)))
        return (tau, ws, we) if windows else tau

    def h2o_ctm(self, tau, CS, T, Ps, N, T0, CF, P, T0F):
        L, nw = tau.shape
        self.lib.orc_h2o_ctm(C.c_uint64(nw), C.c_int(L), _dp(tau), _dp(_f64(CS)), _dp(_f64(T)),
                             _dp(_f64(Ps)), _dp(_f64(N)), _dp(_f64(T0)), _dp(_f64(CF)),
                             _dp(_f64(P)), _dp(_f64(T0F)))
        return tau

    def gas_optics(self, p_mb, t, w0, wres, nw, mols, h2o_coefs=None, o3_xs=None,
                   cfcs=(), cias=(), method=2):
        """mols: list of dict(id,num_iso,mass,lines,x,q,h2o_ctm,o3_ctm);
        cfcs: list of (x_level, xs_grid); cias: list of (x1_level, x2_level, xs_grid);
        method: 0 wavenumber_sweep, 1 line_sweep, 2 line_sample (gas_optics.h:89-94)."""
        p_mb, t = _f64(p_mb), _f64(t)
        V = p_mb.size
        keep = []
        arr = (OrcMolecule * max(len(mols), 1))()
        for k, m in enumerate(mols):
            ln = m["lines"]
            a = {kk: _f64(ln[kk]) for kk in ("v0", "s0", "yair", "yself", "en", "nexp", "delta")}
            iso = np.ascontiguousarray(ln["iso"], dtype=np.int32)
            x, q = _f64(m["x"]), _f64(m["q"])
            keep += [a, iso, x, q]
            arr[k] = OrcMolecule(m["id"], m["num_iso"], m["mass"], ln["v0"].size,
                                 _dp(a["v0"]), _dp(a["s0"]), _dp(a["yair"]), _dp(a["yself"]),
                                 _dp(a["en"]), _dp(a["nexp"]), _dp(a["delta"]),
                                 iso.ctypes.data_as(c_int_p), _dp(x), _dp(q),
                                 int(m.get("h2o_ctm", 0)), int(m.get("o3_ctm", 0)))

        def ptr_array(seq):
            seq = [_f64(s) for s in seq]
            keep.append(seq)
            pa = (c_double_p * max(len(seq), 1))()
            for i, s in enumerate(seq):
                pa[i] = _dp(s)
            return pa

        h2o = ptr_array(h2o_coefs if h2o_coefs is not None else [])
        o3 = _f64(o3_xs) if o3_xs is not None else np.zeros(1)
        cfc_x = ptr_array([c[0] for c in cfcs])
        cfc_xs = ptr_array([c[1] for c in cfcs])
        cia_x1 = ptr_array([c[0] for c in cias])
        cia_x2 = ptr_array([c[1] for c in cias])
        cia_xs = ptr_array([c[2] for c in cias])
        tau = np.zeros((V - 1, nw))
        self.lib.orc_gas_optics_method(C.c_int(method), C.c_int(V), _dp(p_mb), _dp(t), C.c_double(w0),
                                       C.c_double(wres), C.c_uint64(nw), C.c_int(len(mols)), arr, h2o, _dp(o3),
                                       C.c_int(len(cfcs)), cfc_x, cfc_xs,
                                       C.c_int(len(cias)), cia_x1, cia_x2, cia_xs, _dp(tau))
        return tau

    def bracket(self, array, val):
        array = _f64(array)
        l, r = C.c_uint64(0), C.c_uint64(0)
        rc = self.lib.orc_bracket(C.c_uint64(array.size), _dp(array), C.c_double(val), C.byref(l), C.byref(r))
        return rc, l.value, r.value

    def sweep(self, method, vnn, snn, gamma, alpha, ns, w0, wres, nw, bin_width=1.0):
        """One molecule through sort_lines + bin sweep (method 0) or line sweep (1), then the interpolation:
        kernels.c:135-406,514-581 on prepared (layer, line) arrays."""
        vnn, snn, gamma, alpha, ns = (_f64(a).copy() for a in (vnn, snn, gamma, alpha, ns))
        L, N = vnn.shape
        bins = OrcBins()
        self.lib.orc_bins_create(C.byref(bins), C.c_int(L), C.c_double(w0), C.c_uint64(nw), C.c_double(wres),
                                 C.c_double(bin_width))
        tau = np.zeros((L, nw))
        if method == 0:
            self.lib.orc_sort_lines(C.c_uint64(N), C.c_int(L), _dp(vnn), _dp(snn), _dp(gamma), _dp(alpha))
        f = self.lib.orc_bin_sweep if method == 0 else self.lib.orc_line_sweep
        f(C.c_uint64(N), C.c_int(L), _dp(vnn), _dp(snn), _dp(gamma), _dp(alpha), _dp(ns), C.byref(bins), _dp(tau))
        self.lib.orc_interpolate(C.byref(bins), _dp(tau))
        self.lib.orc_bins_destroy(C.byref(bins))
        return tau

    def rayleigh(self, num_layers, p_mb, w0, dw, nw):
        p_mb = _f64(p_mb)
        tau, om, g = (np.zeros((num_layers, nw)) for _ in range(3))
        self.lib.orc_rayleigh(C.c_int(num_layers), _dp(p_mb), C.c_double(w0), C.c_double(dw),
                              C.c_uint64(nw), _dp(tau), _dp(om), _dp(g))
        return tau, om, g

    def add_optics(self, taus, omegas, gs):
        taus, omegas, gs = ([_f64(a) for a in s] for s in (taus, omegas, gs))
        K = len(taus)
        n = taus[0].size

        def pa(seq):
            p = (c_double_p * K)()
            for i, s in enumerate(seq):
                p[i] = _dp(s)
            return p
        tau, om, g = (np.zeros(taus[0].shape) for _ in range(3))
        self.lib.orc_add_optics(C.c_uint64(n), C.c_int(K), pa(taus), pa(omegas), pa(gs),
                                _dp(tau), _dp(om), _dp(g))
        return tau, om, g

    def lw_fluxes(self, w0, wres, T_surf, T_layers, T_levels, tau, omega, emis):
        tau, omega, emis, T_layers, T_levels = map(_f64, (tau, omega, emis, T_layers, T_levels))
        L, nw = tau.shape
        up, dn = np.zeros((L + 1, nw)), np.zeros((L + 1, nw))
        self.lib.orc_lw_fluxes(C.c_int(L + 1), C.c_double(w0), C.c_double(wres), C.c_uint64(nw),
                               C.c_double(T_surf), _dp(T_layers), _dp(T_levels), _dp(tau),
                               _dp(omega), _dp(emis), _dp(up), _dp(dn))
        return up, dn

    def sw_fluxes(self, omega, g, tau, mu_dir, mu_dif, alb_dir, alb_dif, tsi, solar):
        omega, g, tau, alb_dir, alb_dif, solar = map(_f64, (omega, g, tau, alb_dir, alb_dif, solar))
        L, nw = tau.shape
        up, dn = np.zeros((L + 1, nw)), np.zeros((L + 1, nw))
        self.lib.orc_sw_fluxes(C.c_int(L + 1), C.c_uint64(nw), _dp(omega), _dp(g), _dp(tau),
                               C.c_double(mu_dir), C.c_double(mu_dif), _dp(alb_dir), _dp(alb_dif),
                               C.c_double(tsi), _dp(solar), _dp(up), _dp(dn))
        return up, dn

    def integrate_row(self, row, dw):
        row = _f64(row)
        return self.lib.orc_integrate_row(_dp(row), C.c_uint64(row.size), C.c_double(dw))

    def rescale_strengths(self, snn, en, vnn, q296):
        snn = _f64(snn).copy()
        self.lib.orc_rescale_strengths(C.c_uint64(snn.size), _dp(snn), _dp(_f64(en)),
                                       _dp(_f64(vnn)), _dp(_f64(q296)))
        return snn

    def interp_to_grid(self, w0, dw, nw, x, y, constant_extrap=False):
        x, y = _f64(x), _f64(y)
        out = np.zeros(nw)
        self.lib.orc_interp_to_grid(C.c_double(w0), C.c_double(dw), C.c_uint64(nw), _dp(x), _dp(y),
                                    C.c_uint64(x.size), C.c_int(int(constant_extrap)), _dp(out))
        return out

    def normalize_solar(self, w0, dw, c):
        c = _f64(c).copy()
        self.lib.orc_normalize_solar(C.c_double(w0), C.c_double(dw), C.c_uint64(c.size), _dp(c))
        return c


# --------------------------------------------------------------------------- #
# Reference build (oracle/_ref)
# --------------------------------------------------------------------------- #
class RefSpectralGrid(C.Structure):      # utilities/src/spectral_grid.h:32-38
    _fields_ = [("dw", C.c_double), ("n", C.c_uint64), ("wn", C.c_double), ("w0", C.c_double)]


class RefOptics(C.Structure):            # utilities/src/optics.h:30-38
    _fields_ = [("device", C.c_int), ("g", c_double_p), ("grid", RefSpectralGrid),
                ("num_layers", C.c_int), ("omega", c_double_p), ("tau", c_double_p)]


class RefLongwave(C.Structure):          # longwave/src/longwave.h:30-40
    _fields_ = [("num_levels", C.c_int), ("grid", RefSpectralGrid), ("device", C.c_int),
                ("layer_temperature", c_double_p), ("level_temperature", c_double_p),
                ("emissivity", c_double_p), ("flux_up", c_double_p), ("flux_down", c_double_p)]


class RefShortwave(C.Structure):         # shortwave/src/shortwave.h:29-39
    _fields_ = [("num_levels", C.c_int), ("grid", RefSpectralGrid), ("device", C.c_int),
                ("solar_flux", c_double_p), ("sfc_alpha_dir", c_double_p),
                ("sfc_alpha_dif", c_double_p), ("flux_up", c_double_p), ("flux_down", c_double_p)]


class RefSpectralBins(C.Structure):      # gas-optics/src/spectral_bin.h:29-50
    _fields_ = [("num_layers", C.c_int), ("w0", C.c_double), ("wres", C.c_double),
                ("num_wpoints", C.c_uint64), ("n", C.c_uint64), ("width", C.c_double),
                ("isize", C.c_uint64), ("ppb", C.c_int), ("do_interp", C.c_int),
                ("last_ppb", C.c_int), ("do_last_interp", C.c_int),
                ("w", c_double_p), ("tau", c_double_p),
                ("l", C.POINTER(C.c_uint64)), ("r", C.POINTER(C.c_uint64)), ("device", C.c_int)]


class RefLineShapeInputs(C.Structure):   # gas-optics/src/line_shape.h:26-35
    _fields_ = [("w", C.c_double), ("num_wpoints", C.c_uint64), ("wres", C.c_double),
                ("line_center", C.c_double), ("lorentz_hwhm", C.c_double),
                ("doppler_hwhm", C.c_double), ("eta", C.c_double)]


HOST_ONLY = -1


def ref_available(omp=False):
    return os.path.exists(os.path.join(HERE, "_ref", "libgrtref_omp.so" if omp else "libgrtref.so"))


class Ref:
    """numpy-level wrapper over the reference's own compiled C (HOST_ONLY device)."""

    def __init__(self, omp=False):
        path = os.path.join(HERE, "_ref", "libgrtref_omp.so" if omp else "libgrtref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle ref` where /root/reference exists")
        self.lib = C.CDLL(path)
        self.lib.rfm_voigt_line_shape.argtypes = [RefLineShapeInputs, c_double_p]

    def _check(self, rc, what):
        if rc != 0:
            buf = C.create_string_buffer(4096)
            self.lib.grtcode_errstr(C.c_int(rc), buf, C.c_int(4096))
            raise RuntimeError(f"reference {what} returned {rc}: {buf.value.decode(errors='replace')}")

    def grid(self, w0, wn, dw):
        g = RefSpectralGrid()
        self._check(self.lib.create_spectral_grid(C.byref(g), C.c_double(w0), C.c_double(wn), C.c_double(dw)), "create_spectral_grid")
        return g

    def layer_means(self, p_atm, t):
        p_atm, t = _f64(p_atm), _f64(t)
        L = p_atm.size - 1
        n, pavg, tavg = np.zeros(L), np.zeros(L), np.zeros(L)
        self.lib.calc_number_densities(C.c_int(L), _dp(p_atm), _dp(n))
        self.lib.calc_pressures_and_temperatures(C.c_int(L), _dp(p_atm), _dp(t), _dp(pavg), _dp(tavg))
        return n, pavg, tavg

    def species_means(self, p_atm, x, n):
        p_atm, x, n = _f64(p_atm), _f64(x), _f64(n)
        L = n.size
        ps, ns = np.zeros(L), np.zeros(L)
        self.lib.calc_partial_pressures_and_number_densities(C.c_int(L), _dp(p_atm), _dp(x), _dp(n), _dp(ps), _dp(ns))
        return ps, ns

    def line_prep(self, lines, mass, num_iso, pavg, tavg, ps, q):
        N = lines["v0"].size
        L = pavg.size
        a = {k: _f64(lines[k]) for k in ("v0", "delta", "s0", "en", "nexp", "yair", "yself")}
        iso = np.ascontiguousarray(lines["iso"], dtype=np.int32)
        pavg, tavg, ps, q = _f64(pavg), _f64(tavg), _f64(ps), _f64(q)
        vnn, snn, gamma, alpha = (np.zeros((L, N)) for _ in range(4))
        lib = self.lib
        nl, nL = C.c_uint64(N), C.c_int(L)
        lib.calc_line_centers(nl, nL, _dp(a["v0"]), _dp(a["delta"]), _dp(pavg), _dp(vnn))
        lib.calc_line_strengths(nl, nL, C.c_int(num_iso), iso.ctypes.data_as(c_int_p), _dp(a["s0"]),
                                _dp(a["v0"]), _dp(a["en"]), _dp(tavg), _dp(q), _dp(snn))
        lib.calc_lorentz_hw(nl, nL, _dp(a["nexp"]), _dp(a["yair"]), _dp(a["yself"]), _dp(tavg),
                            _dp(pavg), _dp(ps), _dp(gamma))
        lib.calc_doppler_hw(nl, nL, C.c_double(mass), _dp(vnn), _dp(tavg), _dp(alpha))
        return vnn, snn, gamma, alpha

    def voigt(self, w_start, npts, wres, center, gamma, alpha):
        K = np.zeros(npts)
        v = RefLineShapeInputs(w_start, npts, wres, center, gamma, alpha, 0.0)
        self.lib.rfm_voigt_line_shape(v, _dp(K))
        return K

    def _bins(self, L, w0, wres, nw):
        # only the fields calc_optical_depth_line_sample reads (kernels.c:417-438)
        b = RefSpectralBins()
        b.num_layers, b.w0, b.wres, b.num_wpoints = L, w0, wres, nw
        b.device = HOST_ONLY
        return b

    def line_sample(self, vnn, snn, gamma, alpha, ns, w0, wres, nw, tau=None):
        vnn, snn, gamma, alpha, ns = (_f64(a).copy() for a in (vnn, snn, gamma, alpha, ns))
        L, N = vnn.shape
        if tau is None:
            tau = np.zeros((L, nw))
        f = self.lib.calc_optical_depth_line_sample
        f.argtypes = [C.c_uint64, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p,
                      c_double_p, RefSpectralBins, c_double_p, c_double_p, c_double_p]
        self._check(f(N, L, _dp(vnn), _dp(snn), _dp(gamma), _dp(alpha), _dp(ns),
                      self._bins(L, w0, wres, nw), _dp(tau), None, None), "line_sample")
        return tau

    def h2o_ctm(self, tau, CS, T, Ps, N, T0, CF, P, T0F):
        L, nw = tau.shape
        self.lib.calc_water_vapor_ctm_optical_depth(C.c_uint64(nw), C.c_int(L), _dp(tau), _dp(_f64(CS)),
                                                    _dp(_f64(T)), _dp(_f64(Ps)), _dp(_f64(N)), _dp(_f64(T0)),
                                                    _dp(_f64(CF)), _dp(_f64(P)), _dp(_f64(T0F)))
        return tau

    def o3_ctm(self, tau, xs, N):
        L, nw = tau.shape
        self.lib.calc_ozone_ctm_optical_depth(C.c_uint64(nw), C.c_int(L), _dp(_f64(xs)), _dp(_f64(N)), _dp(tau))
        return tau

    def cfc(self, tau, n, x, xs):
        L, nw = tau.shape
        self.lib.calc_cfc_optical_depth(C.c_uint64(nw), C.c_int(L), _dp(_f64(n)), _dp(_f64(x)), _dp(_f64(xs)), _dp(tau))
        return tau

    def cia(self, tau, p, t, x1, x2, xs):
        L, nw = tau.shape
        self.lib.calc_cia_optical_depth(C.c_uint64(nw), C.c_int(L), _dp(_f64(p)), _dp(_f64(t)), _dp(_f64(x1)),
                                        _dp(_f64(x2)), _dp(_f64(xs)), _dp(tau))
        return tau

    def sweep_bins(self, L, w0, wres, nw, bin_width=1.0):
        """spectral_bin.c:30-99 restated on numpy arrays we own (one spare bin of zero padding: the reference's
        line_sweep indexes bin `n` for lines near the top of the grid, kernels.c:345-353,387-403)."""
        b = RefSpectralBins()
        b.num_layers, b.w0, b.wres, b.num_wpoints, b.width = L, w0, wres, nw, bin_width
        b.ppb = int(np.floor(bin_width / wres) + 1)
        b.do_interp = 1 if b.ppb > 3 else 0
        last = nw % b.ppb
        b.last_ppb = b.ppb if last == 0 else last
        b.do_last_interp = 1 if b.last_ppb > 3 else 0
        b.n = nw // b.ppb + (1 if b.ppb != b.last_ppb else 0)
        b.isize = 3 * b.n
        n = b.n
        l = np.zeros(n + 1, dtype=np.uint64)
        r = np.zeros(n + 1, dtype=np.uint64)
        w = np.zeros(3 * (n + 1))
        for i in range(n):
            l[i] = i * b.ppb
            s = b.ppb if i < n - 1 else b.last_ppb
            r[i] = l[i] + s - 1
            w[3 * i] = w0 + b.ppb * i * wres
            w[3 * i + 2] = w[3 * i] + (s - 1) * wres
            w[3 * i + 1] = np.float64(np.float32(0.5)) * (w[3 * i] + w[3 * i + 2])
        taub = np.zeros(L * 3 * n + 64)                     # (layer, n, 3) + padding after the last layer
        b.l = l.ctypes.data_as(C.POINTER(C.c_uint64))
        b.r = r.ctypes.data_as(C.POINTER(C.c_uint64))
        b.w = _dp(w)
        b.tau = _dp(taub)
        b.device = HOST_ONLY
        b._keep = (l, r, w, taub)
        return b

    def sweep(self, method, bins, vnn, snn, gamma, alpha, ns, tau):
        """sort_lines + calc_optical_depth_bin_sweep (method 0) or calc_optical_depth_line_sweep (1)."""
        vnn, snn, gamma, alpha, ns = (_f64(a).copy() for a in (vnn, snn, gamma, alpha, ns))
        L, N = vnn.shape
        if method == 0:
            self._check(self.lib.sort_lines(C.c_uint64(N), C.c_int(L), _dp(vnn), _dp(snn), _dp(gamma), _dp(alpha)),
                        "sort_lines")
        f = self.lib.calc_optical_depth_bin_sweep if method == 0 else self.lib.calc_optical_depth_line_sweep
        f.argtypes = [C.c_uint64, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p,
                      c_double_p, RefSpectralBins, c_double_p]
        self._check(f(N, L, _dp(vnn), _dp(snn), _dp(gamma), _dp(alpha), _dp(ns), bins, _dp(tau)), "sweep")
        return tau

    def interpolate(self, bins, tau):
        for name in ("interpolate", "interpolate_last_bin"):
            f = getattr(self.lib, name)
            f.argtypes = [RefSpectralBins, c_double_p]
            self._check(f(bins, _dp(tau)), name)
        return tau

    def bracket(self, array, val):
        array = _f64(array)
        l, r = C.c_uint64(0), C.c_uint64(0)
        rc = self.lib.bracket(C.c_uint64(array.size), _dp(array), C.c_double(val), C.byref(l), C.byref(r))
        return rc, l.value, r.value

    def gas_optics(self, p_mb, t, w0, wres, nw, mols, h2o_coefs=None, o3_xs=None, cfcs=(), cias=(), method=2):
        """launch.c:40-226 sequencing, executed with the reference's own kernels."""
        p_mb, t = _f64(p_mb), _f64(t)
        p = p_mb * np.float64(np.float32(0.000986923))
        L = p.size - 1
        n, pavg, tavg = self.layer_means(p, t)
        tau = np.zeros((L, nw))
        bins = self.sweep_bins(L, w0, wres, nw) if method != 2 else None
        for m in mols:
            ps, ns = self.species_means(p, m["x"], n)
            if m["lines"]["v0"].size:
                vnn, snn, gamma, alpha = self.line_prep(m["lines"], m["mass"], m["num_iso"], pavg, tavg, ps, m["q"])
                if method == 2:
                    self.line_sample(vnn, snn, gamma, alpha, ns, w0, wres, nw, tau=tau)
                else:
                    self.sweep(method, bins, vnn, snn, gamma, alpha, ns, tau)
            if m.get("h2o_ctm"):
                self.h2o_ctm(tau, h2o_coefs[1], tavg, ps, ns, h2o_coefs[3], h2o_coefs[0], pavg, h2o_coefs[2])
            elif m.get("o3_ctm"):
                self.o3_ctm(tau, o3_xs, ns)
        for x, xs in cfcs:
            self.cfc(tau, n, x, xs)
        for x1, x2, xs in cias:
            self.cia(tau, p, tavg, x1, x2, xs)
        if method != 2:
            self.interpolate(bins, tau)
        return tau

    # -- optics / solvers through the reference's public API ----------------- #
    def _optics(self, grid, tau, omega, g):
        o = RefOptics()
        dev = C.c_int(HOST_ONLY)
        L = tau.shape[0]
        self._check(self.lib.create_optics(C.byref(o), C.c_int(L), C.byref(grid), C.byref(dev)), "create_optics")
        self._check(self.lib.update_optics(C.byref(o), _dp(_f64(tau)), _dp(_f64(omega)), _dp(_f64(g))), "update_optics")
        return o

    def _read_optics(self, o):
        n = o.num_layers * o.grid.n
        shape = (o.num_layers, o.grid.n)
        return tuple(np.ctypeslib.as_array(p, shape=(n,)).reshape(shape).copy() for p in (o.tau, o.omega, o.g))

    def rayleigh(self, grid, num_layers, p_mb):
        p_mb = _f64(p_mb)
        z = np.zeros((num_layers, grid.n))
        o = self._optics(grid, z, z, z)
        self._check(self.lib.rayleigh_scattering(C.byref(o), _dp(p_mb)), "rayleigh_scattering")
        out = self._read_optics(o)
        self.lib.destroy_optics(C.byref(o))
        return out

    def add_optics(self, grid, taus, omegas, gs):
        objs = [self._optics(grid, t, o, g) for t, o, g in zip(taus, omegas, gs)]
        arr = (C.POINTER(RefOptics) * len(objs))(*[C.pointer(o) for o in objs])
        res = RefOptics()
        self._check(self.lib.add_optics(arr, C.c_int(len(objs)), C.byref(res)), "add_optics")
        out = self._read_optics(res)
        for o in objs + [res]:
            self.lib.destroy_optics(C.byref(o))
        return out

    def lw_fluxes(self, grid, T_surf, T_layers, T_levels, tau, omega, emis):
        T_layers, T_levels, emis = (_f64(a).copy() for a in (T_layers, T_levels, emis))
        L = tau.shape[0]
        o = self._optics(grid, tau, omega, np.zeros_like(tau))
        lw = RefLongwave()
        dev = C.c_int(HOST_ONLY)
        self._check(self.lib.create_longwave(C.byref(lw), C.c_int(L + 1), C.byref(grid), C.byref(dev)), "create_longwave")
        up, dn = np.zeros((L + 1, grid.n)), np.zeros((L + 1, grid.n))
        self._check(self.lib.calculate_lw_fluxes(C.byref(lw), C.byref(o), C.c_double(T_surf), _dp(T_layers),
                                                 _dp(T_levels), _dp(emis), _dp(up), _dp(dn)), "calculate_lw_fluxes")
        self.lib.destroy_longwave(C.byref(lw))
        self.lib.destroy_optics(C.byref(o))
        return up, dn

    def sw_fluxes(self, grid, omega, g, tau, mu_dir, mu_dif, alb_dir, alb_dif, tsi, solar):
        alb_dir, alb_dif, solar = (_f64(a).copy() for a in (alb_dir, alb_dif, solar))
        L = tau.shape[0]
        o = self._optics(grid, tau, omega, g)
        sw = RefShortwave()
        dev = C.c_int(HOST_ONLY)
        self._check(self.lib.create_shortwave(C.byref(sw), C.c_int(L + 1), C.byref(grid), C.byref(dev)), "create_shortwave")
        up, dn = np.zeros((L + 1, grid.n)), np.zeros((L + 1, grid.n))
        self._check(self.lib.calculate_sw_fluxes(C.byref(sw), C.byref(o), C.c_double(mu_dir), C.c_double(mu_dif),
                                                 _dp(alb_dir), _dp(alb_dif), C.c_double(tsi), _dp(solar),
                                                 _dp(up), _dp(dn)), "calculate_sw_fluxes")
        self.lib.destroy_shortwave(C.byref(sw))
        self.lib.destroy_optics(C.byref(o))
        return up, dn
