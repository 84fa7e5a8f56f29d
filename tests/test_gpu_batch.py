"""Batched optical depths equal the one-column ABI call, column by column (bitwise up to the
atomic summation order), and the fast arithmetic form stays within its stated tolerance."""
import ctypes as C

import numpy as np
import pytest

from grtcode_amd import api, synthetic as syn
from scenario import Band, MOL_ORDER

pytestmark = pytest.mark.gpu


def test_batch_equals_single_columns(tmp_path, device):
    band = Band(str(tmp_path), 600.0, 900.0, 0.5, 5000)
    V, ncol = 25, 4
    cols = [syn.profile(10 + c, V) for c in range(ncol)]
    go, grid = band.gas_optics(device, V)
    single = []
    opt = api.OpticsObject(V - 1, grid, device)
    for col in cols:
        band.set_column(go, col)
        go.calculate_optical_depth(col["p"], col["t"], opt)
        single.append(opt.read()[0])
    opt.destroy()
    gcols, keep = api.make_columns(cols, MOL_ORDER, cfc_order=(0, 1))
    buf = api.DeviceBuffer(device, 8 * ncol * (V - 1) * band.nw)
    api.check(api.load_library().grt_optical_depth_batch(C.byref(go.c), C.byref(gcols), buf.ptr))
    batch = buf.to_host((ncol, V - 1, band.nw))
    for c in range(ncol):
        scale = single[c].max(axis=1, keepdims=True)
        assert np.max(np.abs(batch[c] - single[c]) / scale) < 1e-13
    buf.free()
    go.destroy()


def test_fast_form_within_tolerance(tmp_path, oracle, lib, device):
    band = Band(str(tmp_path), 500.0, 800.0, 1.0, 6000)
    col = syn.profile(2, 31)
    go, grid = band.gas_optics(device, 31)
    band.set_column(go, col)
    opt = api.OpticsObject(30, grid, device)
    go.tune(fast=1)
    go.calculate_optical_depth(col["p"], col["t"], opt)
    fast = opt.read()[0]
    want = band.oracle_tau(oracle, oracle, lib, col)
    scale = want.max(axis=1, keepdims=True)
    assert np.max(np.abs(fast - want) / scale) < 2e-6      # fp32 core re-association: ~1e-7 per term
    opt.destroy()
    go.destroy()
