"""The cost-analysis hook (grt_gas_optics_probe, include/grt_ext.h): while a buffer is set an instrumented instance of the
two-pass line kernel runs -- same optical depths -- and every workgroup leaves its record; NULL, or a buffer too small for
the launch, and the production instance runs.  Both forms that have an instrumented instance: the single-level kernel of
coarse grids and the cell hierarchy's twelve-moment form of sparse lines."""
import ctypes as C

import numpy as np
import pytest

from grtcode_amd import api, synthetic as syn
from scenario import Band, RUN_TO_RUN_FUSED

pytestmark = pytest.mark.gpu
WORDS = 24


def optical_depth(go, grid, device, col):
    V = col["p"].size
    opt = api.OpticsObject(V - 1, grid, device)
    go.calculate_optical_depth(col["p"], col["t"], opt)
    tau = opt.read()[0]
    opt.destroy()
    return tau


@pytest.mark.parametrize("case", ["single_level_1cm", "hierarchy_sparse_lines"])
def test_instrumented_instance_same_tau_and_one_record_per_workgroup(tmp_path, device, lib, case):
    if case == "single_level_1cm":
        band, V = Band(str(tmp_path), 600.0, 1100.0, 1.0, 6000), 9
    else:
        band, V = Band(str(tmp_path), 500.0, 512.0, 0.004, 900), 7        # 3 000 cells, 0.3 lines per cell
    go, grid = band.gas_optics(device, V, from_file=False)
    go.tune(fast=3)
    col = syn.profile(2, V)
    band.set_column(go, col)
    # (the instrumented instance is one of the GENERAL line loop: it is compared with that loop -- GRT_LEAN is read at every
    # launch -- not with the lean form, which agrees with it to LEAN_TOL only and which this band's tiles, at the ends of
    # its short grid, take since round 5)
    import os
    production = optical_depth(go, grid, device, col)
    os.environ["GRT_LEAN"] = "0"
    try:
        plain = optical_depth(go, grid, device, col)
    finally:
        del os.environ["GRT_LEAN"]
    info = go.last_launch()
    assert info["fast"] == 3 and (info["tree_levels"] > 0) == (case != "single_level_1cm"), info
    if case != "single_level_1cm":
        assert info["moments"] == 12, info
    L = V - 1
    ntiles = (int(grid.n) + info["tile"] - 1) // info["tile"]
    nrec = L * ntiles * info["nslice"]
    buf = api.DeviceBuffer(device, 8 * WORDS * nrec)
    zeros = np.zeros(WORDS * nrec, dtype=np.uint64)
    api.check(lib.grt_host_to_device(device, buf.ptr, zeros.ctypes.data_as(C.c_void_p), zeros.nbytes))
    api.check(lib.grt_gas_optics_probe(C.byref(go.c), buf.ptr, C.c_uint64(WORDS * nrec)))
    probed = optical_depth(go, grid, device, col)
    rec = buf.to_host((L, ntiles, info["nslice"], WORDS), dtype=np.uint64)
    # same sums, other accumulation order
    peak = np.abs(plain).max(axis=1, keepdims=True)
    assert np.max(np.abs(probed - plain) / peak) < RUN_TO_RUN_FUSED
    # every workgroup: entry < prologue done <= last wave out <= all waves out < exit
    assert np.all(rec[..., 0] > 0) and np.all(rec[..., 1] > rec[..., 0])
    assert np.all(rec[..., 11] >= rec[..., 0]) and np.all(rec[..., 12] >= rec[..., 11]) and np.all(rec[..., 1] >= rec[..., 12])
    # candidate lines: every line of the store is somebody's candidate in every layer; blocks were worked on; clocks were kept
    nlines = sum(v["v0"].size for v in band.lines.values())
    assert np.all(rec[..., 2].sum(axis=(1, 2)) >= nlines)
    assert rec[..., 4].sum() >= L * (nlines // 64)
    assert rec[..., 14:22].sum() > 0
    if case != "single_level_1cm":
        assert rec[..., 22].sum() + rec[..., 23].sum() <= rec[..., 5].sum()      # ring steps by form
    # a buffer too small for the launch: the production instance runs, nothing is written
    api.check(lib.grt_host_to_device(device, buf.ptr, zeros.ctypes.data_as(C.c_void_p), zeros.nbytes))
    api.check(lib.grt_gas_optics_probe(C.byref(go.c), buf.ptr, C.c_uint64(WORDS * nrec - 1)))
    again = optical_depth(go, grid, device, col)
    assert np.max(np.abs(again - production) / peak) < RUN_TO_RUN_FUSED
    assert not buf.to_host((WORDS * nrec,), dtype=np.uint64).any()
    api.check(lib.grt_gas_optics_probe(C.byref(go.c), None, C.c_uint64(0)))
    buf.free()
    go.destroy()
