"""Host-side logic of the C ABI that needs no GPU: error convention, utilities, spectral grid,
CSV + HITRAN readers, the table-onto-grid loader.  Expectations follow the reference's own unit
tests (utilities/test/test_{utilities,spectral_grid,parse_csv,verbosity}.c)."""
import ctypes as C
import os

import numpy as np
import pytest

from grtcode_amd import api, synthetic as syn

c_double_p = C.POINTER(C.c_double)


def dp(a):
    return a.ctypes.data_as(c_double_p)


def test_error_text_and_backtrace(lib):
    g = api.SpectralGrid()
    rc = lib.create_spectral_grid(C.byref(g), 0.5, 100.0, 1.0)      # w0 below MIN_WAVENUMBER
    assert rc == api.RANGE_ERR
    buf = C.create_string_buffer(4096)
    assert lib.grtcode_errstr(rc, buf, 4096) == 0
    text = buf.value.decode()
    assert text.startswith("Error: value (5.000000e-01) less than minimum allowed") and "Backtrace:" in text
    assert "grt_grid.c" in text
    lib.grtcode_errstr(0, buf, 4096)
    assert buf.value.decode() == "No errors."
    assert lib.create_spectral_grid(None, 1.0, 2.0, 1.0) == api.NULL_ERR
    lib.grtcode_set_verbosity(2)
    assert lib.grtcode_verbosity() == 2
    lib.grtcode_set_verbosity(0)


@pytest.mark.parametrize("w0,wn,dw,n", [(1.0, 3250.0, 1.0, 3250), (1.0, 50000.0, 1.0, 50000),
                                        (1.0, 3250.0, 0.1, 32491), (1.0, 3250.0, 0.001, 3249001), (1.0, 10.5, 2.0, 6)])
def test_spectral_grid_size(w0, wn, dw, n):
    assert api.create_spectral_grid(w0, wn, dw).n == n      # ceil((wn-w0)/dw)+1 (spectral_grid.c:61)


def test_spectral_grid_index_and_compare(lib):
    g = api.create_spectral_grid(10.0, 20.0, 0.25)
    idx = C.c_uint64()
    assert lib.grid_point_index(g, 12.75, C.byref(idx)) == 0 and idx.value == 11
    assert lib.grid_point_index(g, 12.80, C.byref(idx)) == api.VALUE_ERR        # off-grid
    assert lib.grid_point_index(g, 25.0, C.byref(idx)) == api.RANGE_ERR
    same = C.c_int()
    g2 = api.create_spectral_grid(10.0, 20.0, 0.5)
    lib.compare_spectral_grids(C.byref(g), C.byref(g2), C.byref(same))
    assert same.value == 0
    lib.compare_spectral_grids(C.byref(g), C.byref(g), C.byref(same))
    assert same.value == 1


def test_interpolate2_edges_and_quirks(lib, oracle):
    x = np.array([2.0, 4.0, 8.0])
    y = np.array([1.0, 3.0, -1.0])
    newx = np.array([1.0, 2.0, 3.0, 4.0, 5.0, 8.0, 9.0])
    out = np.full(newx.size, -7.0)
    lin = C.cast(lib.linear_sample, C.c_void_p)
    const = C.cast(lib.constant_extrapolation, C.c_void_p)
    lib.interpolate2.argtypes = [c_double_p, c_double_p, C.c_size_t, c_double_p, c_double_p, C.c_size_t,
                                 C.c_void_p, C.c_void_p]
    assert lib.interpolate2(dp(x), dp(y), 3, dp(newx), dp(out), newx.size, lin, None) == 0
    # points <= x[0] and > x[-1] untouched; interior linear (utilities.c:145-221)
    assert np.allclose(out, [-7.0, -7.0, 2.0, 3.0, 2.0, -1.0, -7.0])
    assert lib.interpolate2(dp(x), dp(y), 3, dp(newx), dp(out), newx.size, lin, const) == 0
    assert out[0] == 1.0 and out[1] == 1.0 and out[-1] == 3.0      # upper constant extrapolation uses y[n-2]
    bad = np.array([2.0, 2.0, 3.0])
    assert lib.interpolate2(dp(bad), dp(y), 3, dp(newx), dp(out), newx.size, lin, None) == api.VALUE_ERR
    # same arithmetic as the oracle's restatement on a real grid
    g = api.create_spectral_grid(1.0, 12.0, 0.5)
    got = np.zeros(g.n)
    lib.interpolate_to_grid.argtypes = [api.SpectralGrid, c_double_p, c_double_p, C.c_size_t, c_double_p,
                                        C.c_void_p, C.c_void_p]
    assert lib.interpolate_to_grid(g, dp(x), dp(y), 3, dp(got), lin, const) == 0
    assert np.array_equal(got, oracle.interp_to_grid(1.0, 0.5, g.n, x, y, constant_extrap=True))


def test_integrate2_and_trapezoid(lib):
    x = np.linspace(0.0, 1.0, 11)
    y = x ** 2
    s = C.c_double()
    lib.integrate2.argtypes = [c_double_p, c_double_p, C.c_size_t, C.POINTER(C.c_double), C.c_void_p]
    assert lib.integrate2(dp(x), dp(y), 11, C.byref(s), C.cast(lib.trapezoid, C.c_void_p)) == 0
    assert abs(s.value - np.trapezoid(y, x)) < 1e-15
    assert lib.integrate2(dp(x), dp(y), 1, C.byref(s), C.cast(lib.trapezoid, C.c_void_p)) == api.VALUE_ERR


def test_bit_fields_and_conversions(lib):
    b = C.c_uint64(0)
    assert lib.activate(C.byref(b), 5) == 0 and b.value == 32
    assert lib.is_active(b, 5) != 0 and lib.is_active(b, 4) == 0
    assert lib.activate(C.byref(b), 64) == api.RANGE_ERR
    d = C.c_double()
    assert lib.to_double(b"1.5e-3", C.byref(d)) == 0 and d.value == 1.5e-3
    assert lib.to_double(b"abc", C.byref(d)) == api.VALUE_ERR
    assert lib.to_double(b"1e999", C.byref(d)) == api.RANGE_ERR
    i = C.c_int()
    assert lib.to_int(b" 42", C.byref(i)) == 0 and i.value == 42
    assert lib.to_int(b"99999999999", C.byref(i)) == api.VALUE_ERR


def test_parse_csv_layout_and_errors(lib, tmp_path):
    p = tmp_path / "t.csv"
    p.write_text("w,a,b\n1.0,10,100\n2.0,20,200\n3.0,30,300\n")
    nl, nc = C.c_int(), C.c_int()
    out = C.POINTER(C.c_char_p)()
    assert lib.parse_csv(str(p).encode(), C.byref(nl), C.byref(nc), 1, C.byref(out)) == 0
    assert (nl.value, nc.value) == (3, 3)
    toks = [out[i].decode() for i in range(9)]
    assert toks == ["1.0", "2.0", "3.0", "10", "20", "30", "100", "200", "300"]      # column-major (parse_csv.c:29)
    (tmp_path / "ragged.csv").write_text("w,a\n1,2\n3\n")
    assert lib.parse_csv(str(tmp_path / "ragged.csv").encode(), C.byref(nl), C.byref(nc), 1, C.byref(out)) == api.VALUE_ERR
    (tmp_path / "blank.csv").write_text("w,a\n\n1,2\n")
    assert lib.parse_csv(str(tmp_path / "blank.csv").encode(), C.byref(nl), C.byref(nc), 1, C.byref(out)) == api.VALUE_ERR
    (tmp_path / "empty.csv").write_text("")
    assert lib.parse_csv(str(tmp_path / "empty.csv").encode(), C.byref(nl), C.byref(nc), 1, C.byref(out)) == api.VALUE_ERR
    (tmp_path / "hdr.csv").write_text("w,a\n")
    assert lib.parse_csv(str(tmp_path / "hdr.csv").encode(), C.byref(nl), C.byref(nc), 1, C.byref(out)) == api.VALUE_ERR
    assert lib.parse_csv(b"/nonexistent/file.csv", C.byref(nl), C.byref(nc), 1, C.byref(out)) == api.IO_ERR


class HostLines(C.Structure):      # csrc/host/grt_internal.h: GrtHostLines
    _fields_ = [("n", C.c_uint64), ("v0", c_double_p), ("s0", c_double_p)] + \
               [(k, C.POINTER(C.c_float)) for k in ("yair", "yself", "en", "nexp", "delta")] + \
               [("iso", C.POINTER(C.c_uint8))]


def test_hitran_reader_matches_written_records(lib, oracle, tmp_path):
    lists = {syn.H2O: syn.line_list(syn.H2O, 300, 100.0, 900.0), syn.CO2: syn.line_list(syn.CO2, 200, 100.0, 900.0)}
    lists[syn.CO2]["iso"][:] = np.tile([1, 2, 10, 11, 12], 40)          # exercises '0' -> 10 and 'A','B' -> 11, 12
    path = str(tmp_path / "lines.par")
    syn.write_hitran_par(path, lists)
    want = syn.read_back_par_values(lists)
    for mol in (syn.H2O, syn.CO2):
        hl = HostLines()
        assert lib.grt_parse_hitran(path.encode(), mol, C.c_double(200.0), C.c_double(800.0), C.byref(hl)) == 0
        w = want[mol]
        keep = (w["v0"] >= 200.0) & (w["v0"] <= 800.0)
        assert hl.n == keep.sum()
        n = hl.n
        got = {k: np.ctypeslib.as_array(getattr(hl, k), shape=(n,)).copy() for k in ("v0", "s0", "yair", "yself", "en", "nexp", "delta", "iso")}
        assert np.array_equal(got["v0"], w["v0"][keep]) and np.array_equal(got["iso"], w["iso"][keep])
        for k in ("yair", "yself", "en", "nexp", "delta"):
            assert np.array_equal(got[k].astype(np.float64), w[k][keep]), k
        # the reader keeps the tabulated 296 K strengths; the factor of parse_HITRAN_file.c:372-384 is applied when
        # the device store is built (grt_rescale_strengths), with whatever partition sums are current then
        assert np.array_equal(got["s0"], w["s0"][keep])
        q296 = np.array([lib.Q(mol, 296.0, int(i)) for i in w["iso"][keep]])
        s_want = oracle.rescale_strengths(w["s0"][keep], w["en"][keep], w["v0"][keep], q296)
        s_got = got["s0"].copy()
        lib.grt_rescale_strengths(mol, C.c_uint64(n), hl.iso, hl.v0, hl.en, s_got.ctypes.data_as(c_double_p))
        assert np.array_equal(s_got, s_want)
        lib.grt_free_host_lines(C.byref(hl))
    bad = tmp_path / "short.par"
    bad.write_text(" 11  500.000000 1.000E-25\n")
    hl = HostLines()
    assert lib.grt_parse_hitran(str(bad).encode(), 1, C.c_double(1.0), C.c_double(1000.0), C.byref(hl)) == api.VALUE_ERR
    assert lib.grt_parse_hitran(b"/nonexistent.par", 1, C.c_double(1.0), C.c_double(1000.0), C.byref(hl)) == api.IO_ERR


def test_hitran_parse_once_index_equals_per_molecule_scans(lib, tmp_path, monkeypatch):
    """§8(f)-3: the first request for a file indexes every molecule's records; later requests filter from memory.
    Same arrays as one scan per call (GRT_HITRAN_CACHE=0, the reference's behaviour); a rewritten file is re-read."""
    lists = {m: syn.line_list(m, 150 + 10 * m, 50.0, 2500.0) for m in (syn.H2O, syn.CO2, syn.O3, syn.CH4)}
    path = str(tmp_path / "lines.par")
    syn.write_hitran_par(path, lists)

    def read(mol, lo, hi):
        hl = HostLines()
        assert lib.grt_parse_hitran(path.encode(), mol, C.c_double(lo), C.c_double(hi), C.byref(hl)) == 0
        out = {k: np.ctypeslib.as_array(getattr(hl, k), shape=(hl.n,)).copy() if hl.n else np.zeros(0)
               for k in ("v0", "s0", "yair", "yself", "en", "nexp", "delta", "iso")}
        lib.grt_free_host_lines(C.byref(hl))
        return out
    cached = {(m, lo): read(m, lo, hi) for m in lists for lo, hi in ((50.0, 2500.0), (700.0, 900.0))}
    monkeypatch.setenv("GRT_HITRAN_CACHE", "0")
    for (m, lo), got in cached.items():
        want = read(m, lo, 2500.0 if lo == 50.0 else 900.0)
        assert all(np.array_equal(got[k], want[k]) for k in got)
    monkeypatch.delenv("GRT_HITRAN_CACHE")
    assert read(syn.N2O, 50.0, 2500.0)["v0"].size == 0           # a molecule the file does not hold
    # same path, new content: the index is keyed by size and modification time
    lists[syn.H2O] = syn.line_list(syn.H2O, 37, 50.0, 2500.0, seed=5)
    syn.write_hitran_par(path, lists)
    assert read(syn.H2O, 50.0, 2500.0)["v0"].size == 37


def test_hitran_index_on_disk(lib, tmp_path, monkeypatch):
    """GRT_HITRAN_CACHE_DIR: the index of a .par file is also written as a binary file; once the in-memory copy has
    been displaced (two other files) the next request reads that instead of parsing text -- same arrays; a
    truncated index file is ignored and rewritten; a rewritten .par file gets an index of its own."""
    cache = tmp_path / "cache"
    cache.mkdir()
    monkeypatch.setenv("GRT_HITRAN_CACHE_DIR", str(cache))
    files = []
    for k in range(3):
        lists = {m: syn.line_list(m, 120 + 7 * m + k, 50.0, 2500.0, seed=11 + k) for m in (syn.H2O, syn.CO2, syn.CH4)}
        path = str(tmp_path / f"lines{k}.par")
        syn.write_hitran_par(path, lists)
        files.append(path)

    def read(path, mol):
        hl = HostLines()
        assert lib.grt_parse_hitran(path.encode(), mol, C.c_double(100.0), C.c_double(2000.0), C.byref(hl)) == 0
        out = {k: np.ctypeslib.as_array(getattr(hl, k), shape=(hl.n,)).copy()
               for k in ("v0", "s0", "yair", "yself", "en", "nexp", "delta", "iso")}
        lib.grt_free_host_lines(C.byref(hl))
        return out

    def stats():
        st = (C.c_longlong * 3)()
        assert lib.grt_hitran_index_stats(st) == 0
        return list(st)
    s0 = stats()
    first = read(files[0], syn.CO2)
    assert [b - a for a, b in zip(s0, stats())] == [0, 0, 1]          # parsed, and written to the cache directory
    idx = list(cache.glob("*.grtidx"))
    assert len(idx) == 1 and idx[0].stat().st_size > 1000
    read(files[0], syn.H2O)
    assert [b - a for a, b in zip(s0, stats())] == [1, 0, 1]          # from memory
    read(files[1], syn.H2O)
    read(files[2], syn.H2O)                                           # two more files: the first one leaves memory
    s1 = stats()
    again = read(files[0], syn.CO2)
    assert [b - a for a, b in zip(s1, stats())] == [0, 1, 0]          # from its index file
    assert all(np.array_equal(first[k], again[k]) for k in first) and first["v0"].size > 50
    # a damaged index file is not trusted
    blob = idx[0].read_bytes()
    idx[0].write_bytes(blob[: len(blob) // 2])
    read(files[1], syn.CO2)
    read(files[2], syn.CO2)
    s2 = stats()
    third = read(files[0], syn.CO2)
    assert [b - a for a, b in zip(s2, stats())][1:] == [0, 1]         # scanned again ...
    assert idx[0].read_bytes() == blob                                # ... and the index rewritten
    assert all(np.array_equal(first[k], third[k]) for k in first)
    # a same-length index file with one byte changed (an isotopologue code out of range, a flipped strength bit) fails
    # its checksum / content checks and is not trusted either (the kernels index 1/Q by that code)
    for pos in (len(blob) - 3, len(blob) // 2):
        bad = bytearray(blob)
        bad[pos] = 200 if pos == len(blob) - 3 else bad[pos] ^ 0x10
        idx[0].write_bytes(bytes(bad))
        read(files[1], syn.CO2)
        read(files[2], syn.CO2)
        s3 = stats()
        fourth = read(files[0], syn.CO2)
        assert [b - a for a, b in zip(s3, stats())][1:] == [0, 1]
        assert idx[0].read_bytes() == blob
        assert all(np.array_equal(first[k], fourth[k]) for k in first)
    # without the variable nothing is written
    monkeypatch.delenv("GRT_HITRAN_CACHE_DIR")
    lists = {syn.H2O: syn.line_list(syn.H2O, 20, 50.0, 2500.0, seed=3)}
    path = str(tmp_path / "lines3.par")
    syn.write_hitran_par(path, lists)
    read(path, syn.H2O)
    assert len(list(cache.glob("*.grtidx"))) == 3


def test_bad_record_of_another_molecule_does_not_fail_the_one_asked_for(lib, tmp_path, monkeypatch):
    """The reference looks only at the records of the molecule it was asked for (parse_HITRAN_file.c:300-313); the
    parse-once index reads every molecule's records, so a field that does not parse is held against its own molecule
    only: other molecules load, that molecule fails like a scan for it alone does."""
    lists = {m: syn.line_list(m, 40, 100.0, 900.0) for m in (syn.H2O, syn.CO2, syn.CH4)}
    path = str(tmp_path / "lines.par")
    syn.write_hitran_par(path, lists)
    text = open(path).read().splitlines(keepends=True)
    k = next(i for i, ln in enumerate(text) if ln.startswith(" 6"))            # a CH4 record: break its strength field
    text[k] = text[k][:15] + "  NOT-A-NUM" [:10].ljust(10) + text[k][25:]
    assert len(text[k]) == 161
    open(path, "w").write("".join(text))
    for cache in ("1", "0"):
        monkeypatch.setenv("GRT_HITRAN_CACHE", cache)
        for mol, want in ((syn.H2O, 0), (syn.CO2, 0), (syn.CH4, api.VALUE_ERR)):
            hl = HostLines()
            assert lib.grt_parse_hitran(path.encode(), mol, C.c_double(100.0), C.c_double(900.0), C.byref(hl)) == want, (cache, mol)
            if want == 0:
                assert hl.n == 40
                lib.grt_free_host_lines(C.byref(hl))
    # an isotopologue code the kernels could not index is refused for its molecule as well
    text[k] = (" 6" + "Z" + text[k][3:15] + " 1.000E-25" + text[k][25:])
    open(path, "w").write("".join(text))
    monkeypatch.setenv("GRT_HITRAN_CACHE", "1")
    hl = HostLines()
    assert lib.grt_parse_hitran(path.encode(), syn.H2O, C.c_double(100.0), C.c_double(900.0), C.byref(hl)) == 0
    lib.grt_free_host_lines(C.byref(hl))
    assert lib.grt_parse_hitran(path.encode(), syn.CH4, C.c_double(100.0), C.c_double(900.0), C.byref(hl)) == api.VALUE_ERR


def test_table_loader_and_solar_flux(lib, oracle, tmp_path):
    w = np.arange(50.0, 151.0, 10.0)
    y = np.exp(-w / 100.0)
    path = str(tmp_path / "solar.csv")
    syn.write_csv(path, w, y)
    grid = api.create_spectral_grid(1.0, 200.0, 1.0)
    got = api.create_solar_flux(grid, path)
    ww = np.array([float("%.6f" % a) for a in w])
    yy = np.array([float("%.9e" % b) for b in y])
    want = oracle.normalize_solar(1.0, 1.0, oracle.interp_to_grid(1.0, 1.0, grid.n, ww, yy))
    assert np.array_equal(got, want)
    assert abs(np.trapezoid(got, dx=1.0) - 1.0) < 1e-12
    (tmp_path / "three.csv").write_text("w,a,b\n1,2,3\n2,3,4\n")
    s = api.SolarFlux()
    assert lib.create_solar_flux(C.byref(s), C.byref(grid), str(tmp_path / "three.csv").encode()) == api.VALUE_ERR


def test_partition_sum_table_plug_in(lib, tmp_path):
    assert lib.grt_tips_is_table() == 0
    base = lib.Q(1, 250.0, 1)
    p = tmp_path / "tips.csv"
    p.write_text("mol,iso,T,Q\n1,1,200,100.0\n1,1,300,180.0\n2,1,200,200.0\n2,1,300,290.0\n")
    assert lib.grt_tips_load(str(p).encode()) == 0 and lib.grt_tips_is_table() == 1
    assert lib.Q(1, 250.0, 1) == 140.0 and lib.Q(1, 150.0, 1) == 100.0 and lib.Q(1, 400.0, 1) == 180.0
    # species absent from the table: the built-in model, classical rotor x harmonic oscillators (grt_tips.c)
    c2, modes = 1.4387769, (1103.1, 700.9, 1042.1)
    qv = lambda T: np.prod([1.0 / (1.0 - np.exp(-c2 * nu / T)) for nu in modes])
    assert lib.Q(3, 250.0, 1) == pytest.approx(3483.71 * (250.0 / 296.0) ** 1.5 * qv(250.0) / qv(296.0), rel=1e-6)
    (tmp_path / "bad.csv").write_text("mol,iso,T,Q\n1,1,300,1\n1,1,200,2\n")
    assert lib.grt_tips_load(str(tmp_path / "bad.csv").encode()) == api.VALUE_ERR
    assert lib.grt_tips_reset() == 0 and lib.Q(1, 250.0, 1) == base
    assert lib.inittips_d() == 0
