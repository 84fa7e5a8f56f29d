"""The line shape itself on the device: rfm_voigt_line_shape (RFM_voigt.c:85-281)
through the parity hook grt_debug_voigt -- a kernel assembled from the device functions the line kernels use (voigt_x,
voigt_near for Humlicek regions 2-4, the region-1 / far-wing / pure-Lorentz expressions) -- against the committed
220 k-point fixture produced by the reference's own compiled C (tests/golden/ref_fixtures.npz: 11 values of y from 1e-7
to 100 x 3 grid spacings x 801 points; |x| from 0 to 33 000 Doppler widths: all five regions, both y thresholds).

  fast = 0 (reference operation order): every point equal to the reference's double to the last bit, except where the
           one transcendental of the function, exp(-x^2) in the outer sums of region 4, comes from ocml instead of glibc
           (<= 2 ulp of a double);
  fast = 1 (the fused forms' arithmetic: hardware reciprocals, REPWID by one Newton step, Lorentzian + region-1
           correction): <= 1e-6 of the line's peak everywhere and <= 5e-6 pointwise.
"""
import numpy as np
import pytest

from grtcode_amd import api

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fx():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fixtures.npz"))


def cases(fx):
    alpha = float(fx["voigt_alpha"])
    for i, y in enumerate(fx["voigt_y"]):
        for j, wres in enumerate(fx["voigt_wres"]):
            yield (float(y), float(wres), alpha, (1000.0 - 400 * wres, 801, wres, 1000.0 + 0.3 * wres, y * alpha / 0.832554611, alpha),
                   fx["voigt_K"][i, j])


def test_fixture_reaches_every_region(fx):
    ys, seen = fx["voigt_y"], set()
    assert ys.min() <= 1e-6 and ys.max() >= 70.55 and np.any((ys > 8.425) & (ys < 70.55)) and np.any(ys < 8.425)
    for y, wres, alpha, args, K in cases(fx):
        x = np.abs((args[0] + np.arange(801) * wres - args[3]) * 0.832554611 / alpha)
        if y >= 70.55:
            seen.add("lorentz")
            continue
        xlim0 = np.sqrt(15100.0 + y * (40.0 - y * 3.6))
        xlim1 = 0.0 if y >= 8.425 else np.sqrt(164.0 - y * (4.3 + y * 1.8))
        if y <= 1e-6:
            xlim1 = xlim0
        xlim2, xlim3, xlim4 = (xlim0 if y <= 1e-6 else 6.8 - y), 2.4 * y, 18.1 * y + 1.65
        seen |= {"far"} if np.any(x >= xlim0) else set()
        seen |= {"1"} if np.any((x >= xlim1) & (x < xlim0)) else set()
        seen |= {"2"} if np.any((x >= xlim2) & (x < xlim1)) else set()
        seen |= {"3"} if np.any((x < xlim2) & (x < xlim3)) else set()
        seen |= {"4in"} if np.any((x < xlim2) & (x >= xlim3) & (x <= xlim4)) else set()
        seen |= {"4out"} if np.any((x < xlim2) & (x >= xlim3) & (x > xlim4)) else set()
    assert seen == {"lorentz", "far", "1", "2", "3", "4in", "4out"}


def test_device_voigt_reference_order_bit_exact(fx, device):
    total = exact = 0
    for y, wres, alpha, args, want in cases(fx):
        got = api.debug_voigt(device, 0, *args)
        same = got == want
        total += want.size
        exact += int(same.sum())
        # what is not identical differs by the last bits of a double (ocml's exp(-x^2) vs glibc's in region 4)
        assert np.max(np.abs(got - want) / want) <= 5e-16, (y, wres)
    assert total == 11 * 3 * 801
    assert exact >= 0.99 * total, (exact, total)


def test_device_voigt_fused_arithmetic(fx, device):
    worst_peak = worst_rel = 0.0
    for y, wres, alpha, args, want in cases(fx):
        got = api.debug_voigt(device, 1, *args)
        worst_peak = max(worst_peak, float(np.max(np.abs(got - want)) / want.max()))
        worst_rel = max(worst_rel, float(np.max(np.abs(got - want) / want)))
    print(f"fused Voigt vs the reference's: worst {worst_peak:.2e} of a line's peak, {worst_rel:.2e} pointwise")
    assert worst_peak <= 1e-6 and worst_rel <= 5e-6
