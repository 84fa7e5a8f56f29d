"""ctypes mirror of the C ABI (include/grtcode_hip_api.h, include/grt_ext.h).

Names, argument order and error behaviour follow the reference's C interface so that
tests read like the reference's own: every call returns an int code, non-zero raises
``GrtError`` carrying the text of ``grtcode_errstr``.  Nothing here computes: a missing
shared library is a hard error (``LibraryMissing``), never a fallback.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libgrtcode_hip.so")

HOST_ONLY = -1
NUM_MOLS, NUM_CFCS, NUM_CIAS, MAX_NUM_CIAS = 53, 21, 2, 3
DIR_PATH_LEN, MOL_NAME_LEN, CFC_NAME_LEN, CIA_NAME_LEN = 1024, 8, 16, 8
LINE_SAMPLE = 2
GRT_FLUXES_PER_COLUMN = 12
RETURN_CODES = ["GRTCODE_SUCCESS", "GRTCODE_INVALID_ERR", "GRTCODE_DIVBYZERO_ERR", "GRTCODE_OVERFLOW_ERR",
                "GRTCODE_UNDERFLOW_ERR", "GRTCODE_SENTINEL_ERR", "GRTCODE_NULL_ERR", "GRTCODE_NON_NULL_ERR",
                "GRTCODE_RANGE_ERR", "GRTCODE_VALUE_ERR", "GRTCODE_COMPILER_ERR", "GRTCODE_IO_ERR",
                "GRTCODE_GPU_ERR"]
(SUCCESS, INVALID_ERR, DIVBYZERO_ERR, OVERFLOW_ERR, UNDERFLOW_ERR, SENTINEL_ERR, NULL_ERR, NON_NULL_ERR,
 RANGE_ERR, VALUE_ERR, COMPILER_ERR, IO_ERR, GPU_ERR) = range(13)

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class LibraryMissing(RuntimeError):
    pass


class GrtError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"{RETURN_CODES[code] if 0 <= code < len(RETURN_CODES) else code}: {text}")
        self.code = code


# ---- struct mirrors (field order == include/grtcode_hip_api.h) ---------------------- #
class SpectralGrid(C.Structure):
    _fields_ = [("dw", C.c_double), ("n", C.c_uint64), ("wn", C.c_double), ("w0", C.c_double)]


class Optics(C.Structure):
    _fields_ = [("device", C.c_int), ("g", c_double_p), ("grid", SpectralGrid), ("num_layers", C.c_int),
                ("omega", c_double_p), ("tau", c_double_p)]


class LineParams(C.Structure):
    _fields_ = [("d", c_double_p), ("device", C.c_int), ("en", c_double_p), ("iso", c_int_p),
                ("n", c_double_p), ("num_lines", C.c_uint64), ("snn", c_double_p), ("vnn", c_double_p),
                ("yair", c_double_p), ("yself", c_double_p)]


class Molecule(C.Structure):
    _fields_ = [("device", C.c_int), ("id", C.c_int), ("line_params", LineParams), ("mass", C.c_double),
                ("name", C.c_char * MOL_NAME_LEN), ("num_isotopologues", C.c_int), ("q", c_double_p)]


class CfcCrossSection(C.Structure):
    _fields_ = [("cross_section", c_double_p), ("id", C.c_int), ("name", C.c_char * CFC_NAME_LEN),
                ("num_wpoints", C.c_uint64), ("device", C.c_int)]


class CollisionInducedAbsorption(C.Structure):
    _fields_ = [("id", C.c_int * 2), ("name", C.c_char_p * 2), ("name_buf", C.c_char * (2 * CIA_NAME_LEN)),
                ("cross_section", c_double_p), ("num_wpoints", C.c_uint64), ("device", C.c_int)]


class WaterVaporContinuumCoefs(C.Structure):
    _fields_ = [("coefs", C.POINTER(c_double_p)), ("num_wpoints", C.c_uint64), ("device", C.c_int)]


class OzoneContinuumCoefs(C.Structure):
    _fields_ = [("cross_section", c_double_p), ("num_wpoints", C.c_uint64), ("device", C.c_int)]


class SpectralBins(C.Structure):
    _fields_ = [("num_layers", C.c_int), ("w0", C.c_double), ("wres", C.c_double),
                ("num_wpoints", C.c_uint64), ("n", C.c_uint64), ("width", C.c_double), ("isize", C.c_uint64),
                ("ppb", C.c_int), ("do_interp", C.c_int), ("last_ppb", C.c_int), ("do_last_interp", C.c_int),
                ("w", c_double_p), ("tau", c_double_p), ("l", C.POINTER(C.c_uint64)),
                ("r", C.POINTER(C.c_uint64)), ("device", C.c_int)]


class GasOptics(C.Structure):
    _fields_ = [("device", C.c_int), ("num_levels", C.c_int), ("num_layers", C.c_int),
                ("num_molecules", C.c_int), ("molecule_bit_field", C.c_uint64), ("mols", Molecule * NUM_MOLS),
                ("num_cfcs", C.c_int), ("cfc_bit_field", C.c_uint64), ("cfcs", CfcCrossSection * NUM_CFCS),
                ("x_cfc", c_double_p), ("num_cias", C.c_int), ("cia_bit_field", C.c_uint64),
                ("cia", CollisionInducedAbsorption * MAX_NUM_CIAS), ("x_cia", c_double_p),
                ("h2o_ctm_dir", C.c_char * DIR_PATH_LEN), ("use_h2o_ctm", C.c_int),
                ("h2o_cc", WaterVaporContinuumCoefs), ("o3_ctm_file", C.c_char * DIR_PATH_LEN),
                ("use_o3_ctm", C.c_int), ("o3_cc", OzoneContinuumCoefs), ("grid", SpectralGrid),
                ("bins", SpectralBins), ("hitran_path", C.c_char * DIR_PATH_LEN), ("wcutoff", C.c_double),
                ("optical_depth_method", C.c_int), ("x", c_double_p), ("tau", c_double_p), ("impl", C.c_void_p)]


class Longwave(C.Structure):
    _fields_ = [("num_levels", C.c_int), ("grid", SpectralGrid), ("device", C.c_int),
                ("layer_temperature", c_double_p), ("level_temperature", c_double_p),
                ("emissivity", c_double_p), ("flux_up", c_double_p), ("flux_down", c_double_p)]


class Shortwave(C.Structure):
    _fields_ = [("num_levels", C.c_int), ("grid", SpectralGrid), ("device", C.c_int),
                ("solar_flux", c_double_p), ("sfc_alpha_dir", c_double_p), ("sfc_alpha_dif", c_double_p),
                ("flux_up", c_double_p), ("flux_down", c_double_p)]


class SolarFlux(C.Structure):
    _fields_ = [("grid", SpectralGrid), ("incident_flux", c_double_p), ("n", C.c_uint64)]


class GrtColumns(C.Structure):
    _fields_ = [("ncol", C.c_int), ("num_levels", C.c_int), ("pressure", c_double_p),
                ("temperature", c_double_p), ("layer_temperature", c_double_p),
                ("surface_temperature", c_double_p), ("molecule_ppmv", c_double_p), ("cfc_ppmv", c_double_p),
                ("cia_ppmv", c_double_p), ("cos_zenith", c_double_p), ("total_solar_irradiance", c_double_p)]


#: every symbol include/*.h declares (checked by tests/test_abi_symbols.py against the headers too)
EXPORTS = """
grtcode_errstr grtcode_set_verbosity grtcode_verbosity create_device get_num_gpus
activate is_active angstrom_exponent angstrom_exponent_sample constant_extrapolation linear_sample
interpolate2 integrate2 trapezoid monotonically_increasing copy_str malloc_ptr free_ptr open_file
to_double to_fp_t to_int parse_csv
compare_spectral_grids create_spectral_grid grid_point_index grid_points interpolate_to_grid
add_optics create_optics destroy_optics optics_compatible sample_optics update_optics
create_gas_optics destroy_gas_optics add_molecule set_molecule_ppmv add_cfc set_cfc_ppmv add_cia
set_cia_ppmv calculate_optical_depth get_num_molecules inittips_d Q
create_longwave destroy_longwave calculate_lw_fluxes
create_shortwave destroy_shortwave calculate_sw_fluxes rayleigh_scattering
create_solar_flux destroy_solar_flux disort_shortwave
grt_tips_load grt_tips_reset grt_tips_is_table grt_tips_source grt_sizeof grt_add_molecule_lines grt_gas_optics_tune grt_gas_optics_last_launch grt_hitran_index_stats
grt_optical_depth_batch grt_pipeline_create grt_pipeline_create_ex grt_pipeline_destroy grt_pipeline_run grt_pipeline_sync
grt_pipeline_stream grt_pipeline_views grt_device_malloc grt_device_free grt_device_to_host
grt_host_to_device grt_debug_line_prep grt_debug_partition_functions grt_debug_tile_items grt_debug_voigt grt_debug_line_strengths grt_profile_enable grt_profile_read
grt_set_deterministic grt_deterministic grt_gas_optics_probe grt_optics_cache_flush grt_device_use_lane grt_device_synchronize
grt_multi_shard grt_multi_create grt_multi_destroy grt_multi_gather_fluxes grt_multi_broadcast grt_multi_max
grt_err_begin grt_err_frame grt_log grt_gmalloc grt_gfree grt_gmemset grt_gmemcpy
""".split()

_lib = None


def load_library(path=None):
    """dlopen the C-ABI library; raise LibraryMissing (never fall back) when it is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("GRT_LIB_PATH") or LIB_PATH   # GRT_LIB_PATH: timing experiments only
    if not os.path.exists(path):
        raise LibraryMissing(f"{path} not found: build it with `python -m grtcode_amd.build` "
                             "(hipcc --offload-arch=gfx950); there is no fallback path")
    lib = C.CDLL(path)
    lib.Q.restype = C.c_double
    lib.Q.argtypes = [C.c_int, C.c_double, C.c_int]
    lib.trapezoid.restype = C.c_double
    lib.angstrom_exponent.restype = C.c_double
    lib.angstrom_exponent.argtypes = [C.c_double] * 4
    lib.grt_sizeof.restype = C.c_size_t
    lib.grt_pipeline_stream.restype = C.c_void_p
    lib.grid_point_index.argtypes = [SpectralGrid, C.c_double, C.POINTER(C.c_uint64)]
    lib.create_spectral_grid.argtypes = [C.POINTER(SpectralGrid), C.c_double, C.c_double, C.c_double]
    lib.calculate_lw_fluxes.argtypes = [C.POINTER(Longwave), C.POINTER(Optics), C.c_double, c_double_p,
                                        c_double_p, c_double_p, c_double_p, c_double_p]
    lib.calculate_sw_fluxes.argtypes = [C.POINTER(Shortwave), C.POINTER(Optics), C.c_double, C.c_double,
                                        c_double_p, c_double_p, C.c_double, c_double_p, c_double_p, c_double_p]
    lib.grt_device_malloc.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_size_t]
    lib.grt_device_free.argtypes = [C.c_int, C.c_void_p]
    lib.grt_device_to_host.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.grt_host_to_device.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.grt_pipeline_run.argtypes = [C.c_void_p, C.POINTER(GrtColumns), C.c_void_p]
    lib.grt_pipeline_sync.argtypes = [C.c_void_p]
    lib.grt_pipeline_stream.argtypes = [C.c_void_p]
    lib.grt_pipeline_views.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_void_p)] * 6
    lib.grt_optical_depth_batch.argtypes = [C.POINTER(GasOptics), C.POINTER(GrtColumns), C.c_void_p]
    _lib = lib
    return lib


def check(rc):
    """Raise GrtError for a non-zero return code, with the library's error text."""
    if rc != 0:
        buf = C.create_string_buffer(4096)
        load_library().grtcode_errstr(C.c_int(rc), buf, C.c_int(4096))
        raise GrtError(rc, buf.value.decode(errors="replace").strip())
    return rc


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _opt_double(v):
    return None if v is None else C.byref(C.c_double(v))


# ---- thin object wrappers: same calls, same order, numpy in/out ---------------------- #
def create_spectral_grid(w0, wn, dw):
    g = SpectralGrid()
    check(load_library().create_spectral_grid(C.byref(g), w0, wn, dw))
    return g


def create_device(device_id=None):
    d = C.c_int()
    check(load_library().create_device(C.byref(d), None if device_id is None else C.byref(C.c_int(device_id))))
    return d.value


class DeviceBuffer:
    """A block of device memory owned through grt_device_malloc / grt_device_free."""

    def __init__(self, device, nbytes):
        self.device, self.nbytes = device, nbytes
        self.ptr = C.c_void_p()
        check(load_library().grt_device_malloc(device, C.byref(self.ptr), nbytes))

    def to_host(self, shape, dtype=np.float64, offset=0):
        out = np.empty(shape, dtype=dtype)
        check(load_library().grt_device_to_host(self.device, out.ctypes.data_as(C.c_void_p),
                                                C.c_void_p(self.ptr.value + offset), out.nbytes))
        return out

    def free(self):
        if self.ptr:
            check(load_library().grt_device_free(self.device, self.ptr))
            self.ptr = C.c_void_p()


def device_to_host(device, ptr, shape, dtype=np.float64):
    out = np.empty(shape, dtype=dtype)
    addr = ptr if isinstance(ptr, int) else C.cast(ptr, C.c_void_p).value
    check(load_library().grt_device_to_host(device, out.ctypes.data_as(C.c_void_p), C.c_void_p(addr), out.nbytes))
    return out


class OpticsObject:
    def __init__(self, num_layers, grid, device, _adopt=None):
        self.lib = load_library()
        self.c = _adopt if _adopt is not None else Optics()
        if _adopt is None:
            check(self.lib.create_optics(C.byref(self.c), num_layers, C.byref(grid), C.byref(C.c_int(device))))
        self.shape = (self.c.num_layers, self.c.grid.n)

    def update(self, tau, omega, g):
        tau, omega, g = _f64(tau), _f64(omega), _f64(g)
        check(self.lib.update_optics(C.byref(self.c), _dp(tau), _dp(omega), _dp(g)))

    def read(self):
        return tuple(device_to_host(self.c.device, p, self.shape) for p in (self.c.tau, self.c.omega, self.c.g))

    def rayleigh(self, p_mb):
        p_mb = _f64(p_mb)
        check(self.lib.rayleigh_scattering(C.byref(self.c), _dp(p_mb)))

    def destroy(self):
        check(self.lib.destroy_optics(C.byref(self.c)))


def add_optics(objs):
    lib = load_library()
    arr = (C.POINTER(Optics) * len(objs))(*[C.pointer(o.c) for o in objs])
    res = Optics()
    check(lib.add_optics(arr, len(objs), C.byref(res)))
    return OpticsObject(0, None, 0, _adopt=res)


class GasOpticsObject:
    def __init__(self, num_levels, grid, device, hitran_path="", h2o_ctm_dir=None, o3_ctm_file=None,
                 wcutoff=None, method=LINE_SAMPLE):
        self.lib = load_library()
        self.c = GasOptics()
        enc = lambda s: None if s is None else s.encode()
        check(self.lib.create_gas_optics(C.byref(self.c), num_levels, C.byref(grid), C.byref(C.c_int(device)),
                                         enc(hitran_path), enc(h2o_ctm_dir), enc(o3_ctm_file),
                                         _opt_double(wcutoff),
                                         None if method is None else C.byref(C.c_int(method))))
        self.grid, self.device, self.num_levels = grid, device, num_levels

    def add_molecule(self, mol_id, wmin=None, wmax=None):
        check(self.lib.add_molecule(C.byref(self.c), mol_id, _opt_double(wmin), _opt_double(wmax)))

    def add_molecule_lines(self, mol_id, lines):
        n = lines["v0"].size
        iso = np.ascontiguousarray(lines["iso"], dtype=np.int32)
        a = {k: _f64(lines[k]) for k in ("v0", "s0", "yair", "yself", "en", "nexp", "delta")}
        check(self.lib.grt_add_molecule_lines(C.byref(self.c), mol_id, C.c_uint64(n), iso.ctypes.data_as(c_int_p),
                                              _dp(a["v0"]), _dp(a["s0"]), _dp(a["yair"]), _dp(a["yself"]),
                                              _dp(a["en"]), _dp(a["nexp"]), _dp(a["delta"])))

    def set_molecule_ppmv(self, mol_id, ppmv):
        check(self.lib.set_molecule_ppmv(C.byref(self.c), mol_id, _dp(_f64(ppmv))))

    def add_cfc(self, cfc_id, path):
        check(self.lib.add_cfc(C.byref(self.c), cfc_id, path.encode()))

    def set_cfc_ppmv(self, cfc_id, ppmv):
        check(self.lib.set_cfc_ppmv(C.byref(self.c), cfc_id, _dp(_f64(ppmv))))

    def add_cia(self, s1, s2, path):
        check(self.lib.add_cia(C.byref(self.c), s1, s2, path.encode()))

    def set_cia_ppmv(self, cia_id, ppmv):
        check(self.lib.set_cia_ppmv(C.byref(self.c), cia_id, _dp(_f64(ppmv))))

    def tune(self, tile=0, nslice=0, fast=0):
        check(self.lib.grt_gas_optics_tune(C.byref(self.c), tile, nslice, fast))

    def last_launch(self):
        info = (C.c_longlong * 8)()
        check(self.lib.grt_gas_optics_last_launch(C.byref(self.c), info))
        return dict(zip(("fast", "tile", "nslice", "tree_levels", "halo", "moment_bytes", "moments", "columns_per_launch"), info))

    def tile_items(self):
        """(items [n][4], ranges [tiles][2]) of the last two-pass launch table: grt_debug_tile_items (include/grt_ext.h)."""
        n, tiles = C.c_uint32(), C.c_uint64()
        check(self.lib.grt_debug_tile_items(C.byref(self.c), C.byref(n), None, C.byref(tiles), None))
        items = np.zeros((n.value, 4), dtype=np.uint32)
        ranges = np.zeros((tiles.value, 2), dtype=np.uint32)
        if n.value and tiles.value:
            check(self.lib.grt_debug_tile_items(C.byref(self.c), C.byref(n), items.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                C.byref(tiles), ranges.ctypes.data_as(C.POINTER(C.c_uint32))))
        return items, ranges

    def calculate_optical_depth(self, p_mb, t, optics):
        p_mb, t = _f64(p_mb).copy(), _f64(t).copy()
        check(self.lib.calculate_optical_depth(C.byref(self.c), _dp(p_mb), _dp(t), C.byref(optics.c)))

    def debug_line_prep(self, p_mb, t):
        p_mb, t = _f64(p_mb).copy(), _f64(t).copy()
        n = C.c_uint64()
        check(self.lib.grt_debug_line_prep(C.byref(self.c), _dp(p_mb), _dp(t), C.byref(n), *([None] * 9)))
        N, L = n.value, self.num_levels - 1
        slot = np.zeros(N, dtype=np.uint8)
        v0 = np.zeros(N)
        f = [np.zeros((L, N)) for _ in range(4)]
        ws, we = np.zeros((L, N), dtype=np.int64), np.zeros((L, N), dtype=np.int64)
        if N:
            check(self.lib.grt_debug_line_prep(C.byref(self.c), _dp(p_mb), _dp(t), C.byref(n),
                                               slot.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(v0),
                                               *[_dp(a) for a in f],
                                               ws.ctypes.data_as(C.POINTER(C.c_int64)),
                                               we.ctypes.data_as(C.POINTER(C.c_int64))))
        return dict(slot=slot, v0=v0, vnn=f[0], snn=f[1], gamma=f[2], alpha=f[3], win_s=ws, win_e=we)

    def debug_partition_functions(self, p_mb, t):
        """1/Q(T_layer, iso) as the device's column state holds it: [num_molecules][L][18]."""
        p_mb, t = _f64(p_mb).copy(), _f64(t).copy()
        q = np.zeros((self.c.num_molecules, self.num_levels - 1, 18))
        check(self.lib.grt_debug_partition_functions(C.byref(self.c), _dp(p_mb), _dp(t), _dp(q)))
        return q

    def debug_line_strengths(self):
        """The device line store's strengths (merged store order), after the rescaling of parse_HITRAN_file.c:372-384."""
        n = C.c_uint64(0)
        check(self.lib.grt_debug_line_strengths(C.byref(self.c), C.byref(n), None))
        s0 = np.zeros(n.value)
        if n.value:
            check(self.lib.grt_debug_line_strengths(C.byref(self.c), C.byref(n), _dp(s0)))
        return s0

    def destroy(self):
        check(self.lib.destroy_gas_optics(C.byref(self.c)))


class LongwaveObject:
    def __init__(self, num_levels, grid, device):
        self.lib = load_library()
        self.c = Longwave()
        check(self.lib.create_longwave(C.byref(self.c), num_levels, C.byref(grid), C.byref(C.c_int(device))))

    def fluxes(self, optics, T_surf, T_layers, T_levels, emis, out=None):
        """out: optional (flux_up, flux_down) [V][n] arrays to fill (a C driver allocates them once: driver.c:682-688)."""
        T_layers, T_levels, emis = _f64(T_layers).copy(), _f64(T_levels).copy(), _f64(emis).copy()
        V, n = self.c.num_levels, self.c.grid.n
        up, dn = out if out is not None else (np.zeros((V, n)), np.zeros((V, n)))
        check(self.lib.calculate_lw_fluxes(C.byref(self.c), C.byref(optics.c), T_surf, _dp(T_layers),
                                           _dp(T_levels), _dp(emis), _dp(up), _dp(dn)))
        return up, dn

    def destroy(self):
        check(self.lib.destroy_longwave(C.byref(self.c)))


class ShortwaveObject:
    def __init__(self, num_levels, grid, device):
        self.lib = load_library()
        self.c = Shortwave()
        check(self.lib.create_shortwave(C.byref(self.c), num_levels, C.byref(grid), C.byref(C.c_int(device))))

    def fluxes(self, optics, mu_dir, mu_dif, alb_dir, alb_dif, tsi, solar, out=None):
        alb_dir, alb_dif, solar = _f64(alb_dir).copy(), _f64(alb_dif).copy(), _f64(solar).copy()
        V, n = self.c.num_levels, self.c.grid.n
        up, dn = out if out is not None else (np.zeros((V, n)), np.zeros((V, n)))
        check(self.lib.calculate_sw_fluxes(C.byref(self.c), C.byref(optics.c), mu_dir, mu_dif, _dp(alb_dir),
                                           _dp(alb_dif), tsi, _dp(solar), _dp(up), _dp(dn)))
        return up, dn

    def destroy(self):
        check(self.lib.destroy_shortwave(C.byref(self.c)))


def create_solar_flux(grid, path):
    lib = load_library()
    s = SolarFlux()
    check(lib.create_solar_flux(C.byref(s), C.byref(grid), path.encode()))
    out = np.ctypeslib.as_array(s.incident_flux, shape=(s.n,)).copy()
    check(lib.destroy_solar_flux(C.byref(s)))
    return out


def make_columns(cols, mol_order, cfc_order=(), num_levels=None):
    """Pack a list of synthetic.profile()-style dicts into a GrtColumns struct (+ keep-alive arrays)."""
    V = num_levels or cols[0]["p"].size
    keep = dict(
        p=_f64(np.stack([c["p"] for c in cols])), t=_f64(np.stack([c["t"] for c in cols])),
        tl=_f64(np.stack([c["t_layer"] for c in cols])), ts=_f64([c["t_surf"] for c in cols]),
        mol=_f64(np.stack([np.stack([c["ppmv"][m] for m in mol_order]) for c in cols])) if mol_order else np.zeros(1),
        cfc=_f64(np.stack([np.stack([c["cfc_ppmv"][k] for k in cfc_order]) for c in cols])) if cfc_order else None,
        cia=_f64(np.stack([np.stack([c["ppmv"][22], c["ppmv"][7]]) for c in cols])),   # N2 (CIA_N2=0), O2 (CIA_O2=1)
        mu=_f64([c["mu0"] for c in cols]), tsi=_f64([c["tsi"] for c in cols]))
    gc = GrtColumns(len(cols), V, _dp(keep["p"]), _dp(keep["t"]), _dp(keep["tl"]), _dp(keep["ts"]),
                    _dp(keep["mol"]), _dp(keep["cfc"]) if keep["cfc"] is not None else None,
                    _dp(keep["cia"]), _dp(keep["mu"]), _dp(keep["tsi"]))
    return gc, keep


class Pipeline:
    def __init__(self, lw_gas, sw_gas, max_columns, user_level, emissivity, albedo, solar, spectral=True):
        """spectral=True keeps tau/omega/g and the spectral fluxes (views(): what parity tests read);
        spectral=False is the production form of grt_pipeline_create: fused solvers, integrated fluxes only."""
        self.lib = load_library()
        self.spectral = spectral
        self.p = C.c_void_p()
        self.device = (lw_gas or sw_gas).device
        e = _f64(emissivity) if emissivity is not None else None
        a = _f64(albedo) if albedo is not None else None
        s = _f64(solar) if solar is not None else None
        check(self.lib.grt_pipeline_create_ex(C.byref(self.p), C.byref(lw_gas.c) if lw_gas else None,
                                              C.byref(sw_gas.c) if sw_gas else None, max_columns, user_level,
                                              _dp(e) if e is not None else None, _dp(a) if a is not None else None,
                                              _dp(s) if s is not None else None, int(spectral)))
        self.out = DeviceBuffer(self.device, 8 * GRT_FLUXES_PER_COLUMN * max_columns)
        self.max_columns = max_columns

    def run(self, gcols, out_ptr=None):
        check(self.lib.grt_pipeline_run(self.p, C.byref(gcols), out_ptr if out_ptr is not None else self.out.ptr))

    def sync(self):
        check(self.lib.grt_pipeline_sync(self.p))

    def stream(self):
        return self.lib.grt_pipeline_stream(self.p)

    def fluxes(self, ncol):
        self.sync()
        return self.out.to_host((ncol, GRT_FLUXES_PER_COLUMN))

    def views(self, band):
        ptrs = [C.c_void_p() for _ in range(6)]
        if not self.spectral:
            check(self.lib.grt_pipeline_views(self.p, band, C.byref(ptrs[0]), *([None] * 5)))
            return {"tau_gas": ptrs[0].value}
        check(self.lib.grt_pipeline_views(self.p, band, *[C.byref(p) for p in ptrs]))
        return dict(zip(("tau_gas", "tau", "omega", "g", "flux_up", "flux_down"), [p.value for p in ptrs]))

    def destroy(self):
        self.out.free()
        check(self.lib.grt_pipeline_destroy(C.byref(self.p)))


def debug_voigt(device, fast, w_start, npts, wres, center, gamma, alpha):
    """rfm_voigt_line_shape on the device (grt_debug_voigt): K [npts]."""
    lib = load_library()
    K = np.zeros(npts)
    lib.grt_debug_voigt.argtypes = [C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double, c_double_p]
    check(lib.grt_debug_voigt(device, int(fast), w_start, npts, wres, center, gamma, alpha, _dp(K)))
    return K


def use_lane(device, lane):
    """Calls that follow enqueue on stream `lane` (0..3) of the device: several batches in flight (grt_ext.h)."""
    check(load_library().grt_device_use_lane(device, lane))


def device_synchronize(device):
    check(load_library().grt_device_synchronize(device))


def profile_enable(on=True):
    check(load_library().grt_profile_enable(int(on)))


def profile_read(tag, reset=False):
    ms, n = C.c_double(), C.c_int()
    check(load_library().grt_profile_read(tag, C.byref(ms), C.byref(n), int(reset)))
    return ms.value, n.value
