"""The error bound the moment kernel (grtcode_amd/csrc/hip/k_gas_optics_mp.hip) relies on, checked in numpy.

Far wings of the lines of one cell c (all lines whose centre index is c; kernels.c:431-437 gives them one window):

    sum_i A_i / ((r - d_i)^2 + e_i^2)  =  sum_{k>=1} M_k r^-(k+1),   M_k = sum_i A_i Im(z_i^k)/e_i,  z = d + i e,

|d| <= 1/2 (offset of the centre from its grid point, in grid steps), e = gamma_L/dw.  The kernel keeps 8 terms and
chooses the near-field radius R per layer from |z|max/(R+1) <= 0.128: R = max(3, ceil(7.8 |z|max) - 1).
"""
import numpy as np
import pytest

K = 8


def radius(eta_max):
    return max(3, int(np.ceil(7.8 * np.sqrt(0.25 + eta_max ** 2))) - 1)


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


@pytest.mark.parametrize("eta_max", [0.0, 0.05, 0.115, 0.5, 1.15, 3.0])
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
