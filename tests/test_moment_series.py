"""The error bound the moment kernel (grtcode_amd/csrc/hip/k_gas_optics_mp.hip) relies on, checked in numpy.

Far wings of the lines of one cell c (all lines whose centre index is c; kernels.c:431-437 gives them one window):

    sum_i A_i / ((r - d_i)^2 + e_i^2)  =  sum_{k>=1} M_k r^-(k+1),   M_k = sum_i A_i Im(z_i^k)/e_i,  z = d + i e,

|d| <= 1/2 (offset of the centre from its grid point, in grid steps), e = gamma_L/dw.  The kernel keeps 8 terms and
chooses the near-field radius R per layer from |z|max/(R+1) <= 0.128: R = max(3, ceil(7.8 |z|max) - 1) -- except that
seven points (R = 3) are kept for every e up to 0.3 (round 5, near_radius): the remainder there is no larger than the one
the bound already accepts for a narrow line half-way between two grid points.
"""
import numpy as np
import pytest

K = 8


def radius(eta_max):
    r = max(3, int(np.ceil(7.8 * np.sqrt(0.25 + eta_max ** 2))) - 1)
    return 3 if (r == 4 and eta_max <= 0.3) else r


def moments(A, d, e):
    """The kernel's two-term recurrence: u_k = Re z^k, p_k = Im z^k / e."""
    u, p, M = np.ones_like(d), np.zeros_like(d), []
    for _ in range(K):
        u, p = d * u - e ** 2 * p, d * p + u
        M.append(np.sum(A * p))
    return np.array(M)


def series(M, r):
    u = 1.0 / r
    acc = np.zeros_like(r)
    for k in range(K - 1, -1, -1):      # Horner, as the kernel's far-field gather
        acc = acc * u + M[k]
    return acc * u * u


@pytest.mark.parametrize("eta_max", [0.0, 0.05, 0.115, 0.19, 0.24, 0.3, 0.31, 0.5, 1.15, 3.0])
def test_single_line_worst_case_stays_below_5e7_of_its_far_wing(eta_max):
    R = radius(eta_max)
    r = np.concatenate([np.arange(-250.0, -R), np.arange(R + 1.0, 251.0)])
    worst = 0.0
    for d in (-0.5, -0.25, 0.0, 0.3, 0.5):
        for e in (eta_max, 0.5 * eta_max, 1e-4):
            exact = 1.0 / ((r - d) ** 2 + e ** 2)
            approx = series(moments(np.array([1.0]), np.array([d]), np.array([e])), r)
            worst = max(worst, np.max(np.abs(approx - exact) / exact))
    assert worst < 6e-7, (R, worst)


def test_many_lines_per_cell_aggregate_error_is_smaller_still():
    rng = np.random.default_rng(7)
    n, eta_max = 300, 0.105                       # ~300 lines per grid point, 1 cm-1 grid, surface layer
    d, e = rng.uniform(-0.5, 0.5, n), eta_max * rng.uniform(0, 1, n) ** 2
    A = 10.0 ** rng.uniform(-4, 0, n)
    R = radius(eta_max)
    assert R == 3
    r = np.concatenate([np.arange(-25.0, -R), np.arange(R + 1.0, 26.0)])
    exact = np.sum(A[:, None] / ((r[None, :] - d[:, None]) ** 2 + e[:, None] ** 2), axis=0)
    approx = series(moments(A, d, e), r)
    assert np.max(np.abs(approx - exact) / exact) < 2e-7


def test_fp32_moments_and_gather_cost_about_1e7():
    """The kernel forms the moments and the Horner terms in fp32 (sum over cells in fp64)."""
    rng = np.random.default_rng(9)
    n = 64
    d, e = rng.uniform(-0.5, 0.5, n).astype(np.float32), (0.1 * rng.uniform(0, 1, n)).astype(np.float32)
    A = (10.0 ** rng.uniform(-3, 0, n)).astype(np.float32)
    u, p, M = np.ones(n, np.float32), np.zeros(n, np.float32), []
    for _ in range(K):
        u, p = (d * u - e * e * p).astype(np.float32), (d * p + u).astype(np.float32)
        M.append(np.float32(np.sum((A * p).astype(np.float32), dtype=np.float32)))
    r = np.arange(4.0, 26.0)
    uu = (1.0 / r).astype(np.float32)
    acc = np.zeros_like(uu)
    for k in range(K - 1, -1, -1):
        acc = (acc * uu + M[k]).astype(np.float32)
    approx = (acc * (uu * uu)).astype(np.float64)
    exact = np.sum(A.astype(np.float64)[:, None] / ((r[None, :] - d.astype(np.float64)[:, None]) ** 2
                                                      + e.astype(np.float64)[:, None] ** 2), axis=0)
    assert np.max(np.abs(approx - exact) / exact) < 1e-6


# ---- Humlicek region 1 folded into the moments (near_radius / `corrected` in k_gas_optics_mp.hip) ----------------- #
def region1(x, y):
    """RFM_voigt.c:172-183 without the common factor: (A0 + XQ)/(D0 + XQ (D2 + XQ)) y/pi ~ K/cl."""
    xq, yq = x * x, y * y
    a0 = yq + 0.5
    return (a0 + xq) / (a0 * a0 + xq * ((2 * yq - 1.0) + xq))


def lorentz(x, y):
    return 1.0 / (x * x + y * y)


def correction_series(x, y):
    q, Y = x * x, y * y
    return 1.5 / q ** 2 + (1.25 - 5 * Y) / q ** 3 + (10.5 * Y * Y - 8.75 * Y + 0.875) / q ** 4


@pytest.mark.parametrize("y", [1e-4, 0.05, 0.7, 2.0, 4.0])
def test_region1_minus_lorentzian_is_three_terms_in_inverse_x_squared(y):
    """Beyond X1 = max(13, 8 y) Doppler widths the three terms leave < 2e-6 of the line-shape value itself there
    (which is < 1e-2 of the peak), and the whole correction is what the kernel says it is."""
    x1 = max(13.0, 8.0 * y)
    x = np.linspace(x1, 123.4, 4000)
    exact = region1(x, y) - lorentz(x, y)
    err = np.abs(correction_series(x, y) - exact) / region1(x, y)
    assert err.max() < 2e-6, err.max()
    assert np.all(exact > 0) or y > 0.5                 # (1.5/x^4 leads: region 1 lies above the Lorentzian for small y)


def test_series_beyond_xlim0_costs_at_most_1e7_of_the_peak():
    """The reference switches back to the Lorentzian at XLIM0; the moments go on with region 1.  scipy's wofz gives
    the peak K(0, y) = Re w(i y)."""
    from scipy.special import wofz
    for y, bound in ((4.0, 1.1e-7), (2.0, 3.2e-8), (0.5, 4e-9)):
        xlim0 = np.sqrt(15100.0 + y * (40.0 - y * 3.6))
        spurious = (region1(xlim0, y) - lorentz(xlim0, y)) * y / np.pi          # K1 - K0 with cl = y/pi (repwid = 1)
        peak = wofz(1j * y).real / np.sqrt(np.pi)                                # RSQRPI * Re w
        assert spurious / peak < bound, (y, spurious / peak)


@pytest.mark.parametrize("wr,y,K,sep", [(9.2, 0.6, 8, 7.8), (16.0, 2.5, 8, 7.8), (0.25, 0.02, 12, 3.95), (0.8, 3.9, 12, 3.95)])
def test_cell_moments_with_the_folded_correction_reproduce_region1(wr, y, K, sep):
    """One line, delta = its offset from the cell centre: Lorentzian moments (two-term recurrence) plus
    b4, b6, b8 re-expanded about the centre, Horner in 1/r, against region 1 itself at every far point whose x is
    still inside XLIM0.  wr = Doppler units per grid step."""
    import math
    eta = y / wr
    x1 = max(13.0, 8.0 * y)
    R = max(3, int(np.ceil(sep * np.sqrt(0.25 + eta * eta))) - 1, int(x1 / wr + 1.51))
    worst = 0.0
    for delta in (-0.5, -0.2, 0.0, 0.31, 0.5):
        u, p, m = 1.0, 0.0, []
        for _ in range(K):
            u, p = delta * u - eta * eta * p, delta * p + u
            m.append(p)
        b4, b6 = 1.5 / wr ** 2, (1.25 - 5 * y * y) / wr ** 4
        b8 = (10.5 * y ** 4 - 8.75 * y * y + 0.875) / wr ** 6
        for i in range(2, K):
            m[i] += math.comb(i + 1, 3) * b4 * delta ** (i - 2)
            if i >= 4:
                m[i] += math.comb(i + 1, 5) * b6 * delta ** (i - 4)
            if i >= 6:
                m[i] += math.comb(i + 1, 7) * b8 * delta ** (i - 6)
        r = np.concatenate([-np.arange(R + 1.0, 3000.0), np.arange(R + 1.0, 3000.0)])
        x = (r - delta) * wr
        sel = np.abs(x) < np.sqrt(15100.0 + y * (40.0 - y * 3.6))
        uu = 1.0 / r[sel]
        acc = np.zeros_like(uu)
        for k in range(K - 1, -1, -1):
            acc = acc * uu + m[k]
        approx = acc * uu * uu                     # in units of A = cl/wr^2
        exact = region1(x[sel], y) * wr ** 2
        worst = max(worst, np.max(np.abs(approx - exact) / exact))
    assert worst < 3e-6, worst
