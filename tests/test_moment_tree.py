"""The cell hierarchy of the fine-grid far field (gas_optics_tree_kernel, k_gas_optics_mp.hip), restated in numpy.

Level 0 is the moment kernel's cell (all lines with centre index c, tests/test_moment_series.py).  A level-l cell
is 2^l consecutive level-0 cells, [j 2^l, (j+1) 2^l); its lines sit within h/2 = 2^(l-1) of its centre
C = j 2^l + 2^(l-1) - 1/2, so in units of h the same series holds:

    sum_i A_i / ((f - x_i)^2 + e_i^2) = (1/h) u^2 (m_1 + u (m_2 + ... )),   u = h/(f - C),
    m_k = M_k / h^k,    M_k = sum_i A_i Im(z_i^k)/e_i,  z_i = (x_i - C) + i e_i,

valid (8 terms, ratio 0.128) where |f - C| >= 7.8 sqrt(h^2/4 + e_max^2) -- or 12 terms, ratio 0.253, 3.95 sqrt(...):
the sparse-line form of the kernels, with half the near field.  Parents come from their two children
by the binomial shift  m'_k = sum_{j<=k} C(k,j) (-+1/4)^(k-j) 2^-j m_j  (the same table at every level).

A target f must receive exactly the cells c with R(c) < |f - c| <= fsteps (kernels.c:435-437: the window of a
line is its centre index +- fsteps): the interval on either side is tiled greedily with the largest aligned,
admissible cells that stay inside the window.
"""
import math

import numpy as np
import pytest

SEP = {8: 7.8, 12: 3.95}      # near field / |z|max: sep^-K = 7e-8 (moment_separation in k_gas_optics_mp.hip)


def shift_tables(K):
    """T[side][k][j]: parent m_(k+1) from child m_(j+1); side 0 = lower child (shift -1/4), 1 = upper (+1/4)."""
    T = np.zeros((2, K, K))
    for side, s in enumerate((-0.25, 0.25)):
        for k in range(1, K + 1):
            for j in range(1, k + 1):
                T[side, k - 1, j - 1] = math.comb(k, j) * s ** (k - j) * 2.0 ** -j
    return T


def level0_moments(nw, c, delta, eta, A, dtype, K=8):
    """[nw][K] moments about the cells' grid points; the kernel's two-term recurrence."""
    M = np.zeros((nw, K))
    u, p = A.astype(dtype), np.zeros_like(A, dtype=dtype)
    d, e2 = delta.astype(dtype), (eta * eta).astype(dtype)
    for k in range(K):
        u, p = (d * u - e2 * p).astype(dtype), (d * p + u).astype(dtype)
        np.add.at(M[:, k], c, p.astype(np.float64))
    return M.astype(dtype)


def build_levels(m0, nlev, dtype):
    K = m0.shape[1]
    T = shift_tables(K).astype(dtype)
    levels = [m0]
    for _ in range(nlev):
        ch = levels[-1]
        if ch.shape[0] % 2:
            ch = np.vstack([ch, np.zeros((1, K), dtype)])
        lo, hi = ch[0::2], ch[1::2]
        levels.append((lo @ T[0].T + hi @ T[1].T).astype(dtype))
    return levels


def cell_value(m, h, d_signed, dtype):
    u = dtype(h) / dtype(d_signed)
    acc = dtype(0)
    for k in range(len(m) - 1, -1, -1):
        acc = dtype(acc * u + m[k])
    return float(dtype(acc * (u * u)) / dtype(h))


def admissible_level(dm, eta_max, sep=7.8):
    """Largest l with (dm + h/2) >= sep sqrt(h^2/4 + eta^2), h = 2^l; dm = distance to the cell's near edge."""
    q = dm * dm - sep ** 2 * eta_max ** 2
    a = (sep ** 2 - 1.0) / 4.0
    disc = dm * dm + 4 * a * q
    if disc < 0:
        return 0
    hmax = 0.999 * (dm + math.sqrt(disc)) / (2 * a)
    return int(math.floor(math.log2(hmax))) if hmax >= 2 else 0


def tree_far_field(f, levels, nw, fsteps, R, eta_max, dtype, counts=None):
    total, lmax = 0.0, len(levels) - 1
    sep = SEP[levels[0].shape[1]]

    def pick(D, align, room):
        l = 0
        if D > R:
            l = min(admissible_level(D - 0.5, eta_max, sep), align, room.bit_length() - 1, lmax)
        return l

    x, e = f + 1, min(f + fsteps, nw - 1)
    while x <= e:
        D = x - f
        l = pick(D, (x & -x).bit_length() - 1, e - x + 1)
        h = 1 << l
        if l > 0 or D > R:
            total += cell_value(levels[l][x >> l], h, -(D - 0.5 + h / 2), dtype)
            if counts is not None:
                counts[l] = counts.get(l, 0) + 1
        x += h
    x, s = f - 1, max(f - fsteps, 0)
    while x >= s:
        D = f - x
        l = pick(D, ((x + 1) & -(x + 1)).bit_length() - 1, x - s + 1)
        h = 1 << l
        if l > 0 or D > R:
            total += cell_value(levels[l][x >> l], h, D - 0.5 + h / 2, dtype)
            if counts is not None:
                counts[l] = counts.get(l, 0) + 1
        x -= h
    return total


def radius(eta_max, K=8):
    return max(3, int(np.ceil(SEP[K] * np.sqrt(0.25 + eta_max ** 2))) - 1)


def exact_far_field(f, c, delta, eta, A, fsteps, R):
    r = f - c
    sel = (np.abs(r) <= fsteps) & (np.abs(r) > R)
    return float(np.sum(A[sel] / ((r[sel] - delta[sel]) ** 2 + eta[sel] ** 2)))


@pytest.mark.parametrize("K", [8, 12])
@pytest.mark.parametrize("eta_max,fsteps,nw", [(0.02, 1000, 5000), (2.5, 1000, 5000), (70.0, 2500, 9000),
                                                 (0.3, 333, 1500)])
def test_tree_equals_the_windowed_sum(eta_max, fsteps, nw, K):
    rng = np.random.default_rng(int(eta_max * 100) + fsteps)
    n = nw // 3
    c = np.sort(rng.integers(0, nw, n))
    delta = rng.uniform(-0.5, 0.5, n)
    eta = eta_max * rng.uniform(0.05, 1, n)
    A = 10.0 ** rng.uniform(-4, 0, n) * eta         # cl/wr^2 ~ S gamma
    R = radius(eta_max, K)
    nlev = int(math.log2(fsteps)) - 1
    targets = np.unique(np.concatenate([rng.integers(0, nw, 40), [0, 1, nw - 1, nw // 2, fsteps, fsteps + 1,
                                                                   nw - fsteps - 1, nw - fsteps]]))
    scale = max(exact_far_field(int(f), c, delta, eta, A, fsteps, -1) for f in targets)   # the layer's largest tau
    for dtype, tol in ((np.float64, 1.5e-7), (np.float32, 6e-7)):
        levels = build_levels(level0_moments(nw, c, delta, eta, A, dtype, K), nlev, dtype)
        worst, counts = 0.0, {}
        for f in targets:
            got = tree_far_field(int(f), levels, nw, fsteps, R, eta_max, dtype, counts)
            want = exact_far_field(int(f), c, delta, eta, A, fsteps, R)
            worst = max(worst, abs(got - want) / scale)
        assert worst < tol, (dtype, worst)
        assert sum(counts.values()) / len(targets) < 40 * (nlev + 1)


def test_window_edges_are_exact():
    """One strong line: the targets at distance fsteps see it, those at fsteps + 1 do not (kernels.c:435-437)."""
    nw, fsteps, cpos = 6000, 1000, 2771
    c, delta, eta, A = np.array([cpos]), np.array([0.37]), np.array([0.01]), np.array([1.0])
    levels = build_levels(level0_moments(nw, c, delta, eta, A, np.float64), 8, np.float64)
    for f in (cpos - fsteps - 1, cpos - fsteps, cpos + fsteps, cpos + fsteps + 1, cpos + 4, cpos - 3):
        got = tree_far_field(f, levels, nw, fsteps, 3, 0.01, np.float64)
        want = exact_far_field(f, c, delta, eta, A, fsteps, 3)
        assert got == pytest.approx(want, rel=2e-7, abs=1e-300)
        assert (got == 0.0) == (want == 0.0)


# ---- the wave-shared form of the gather (windows of 16 384 points and more): near fields are whole 64-point blocks ---- #
def block_far_field(fb, levels, nw, fsteps, R, eta_max, dtype):
    """gas_optics_tree_kernel in numpy for the block fb .. fb + 63: one walk over the cells that lie inside EVERY point's
    window and beyond the block's near field (level chosen for the block's closest point), then every point's own cells
    at the far end of its window.  Returns the 64 far fields (points beyond the grid: 0)."""
    lmax, sep = len(levels) - 1, SEP[levels[0].shape[1]]
    pts = [f for f in range(fb, fb + 64) if f < nw]
    fhi, fhb = pts[-1], fb + 63
    out = {f: 0.0 for f in pts}

    def level(D, align, room):
        return min(admissible_level(D - 0.5, eta_max, sep), align, room.bit_length() - 1, lmax)

    # up: shared [XA, E0s), own [E0s, f + fsteps]
    E0 = min(fb + fsteps, nw - 1) + 1
    XA = min(fhb + R + 1, E0)
    E0s = E0 & ~63 if XA <= (E0 & ~63) else E0
    x = XA
    while x < E0s:
        l = level(x - fhb, (x & -x).bit_length() - 1, E0s - x)
        h = 1 << l
        for f in pts:
            out[f] += cell_value(levels[l][x >> l], h, -((x - f) - 0.5 + h / 2), dtype)
        x += h
    for f in pts:
        e, x = min(f + fsteps, nw - 1), max(E0s, XA)
        while x <= e:
            l = level(x - f, (x & -x).bit_length() - 1, e - x + 1)
            h = 1 << l
            out[f] += cell_value(levels[l][x >> l], h, -((x - f) - 0.5 + h / 2), dtype)
            x += h
    # down: shared (S0s, XB], own [f - fsteps, S0s]
    S0 = max(fhi - fsteps, 0) - 1
    XB = max(fb - R - 1, S0)
    sa = ((S0 + 64) & ~63) - 1
    S0s = sa if XB >= sa else S0
    x = XB
    while x > S0s:
        l = level(fb - x, ((x + 1) & -(x + 1)).bit_length() - 1, x - S0s)
        h = 1 << l
        for f in pts:
            out[f] += cell_value(levels[l][x >> l], h, (f - x) - 0.5 + h / 2, dtype)
        x -= h
    for f in pts:
        s, x = max(f - fsteps, 0), min(S0s, XB)
        while x >= s:
            l = level(f - x, ((x + 1) & -(x + 1)).bit_length() - 1, x - s + 1)
            h = 1 << l
            out[f] += cell_value(levels[l][x >> l], h, (f - x) - 0.5 + h / 2, dtype)
            x -= h
    return out


def exact_far_field_blocks(f, c, delta, eta, A, fsteps, R):
    """The window of kernels.c:435-437 minus the block near field: every 64-point block that c +- R touches."""
    r = f - c
    near = (f >= ((c - R) & ~63)) & (f <= ((c + R) | 63))
    sel = (np.abs(r) <= fsteps) & ~near
    return float(np.sum(A[sel] / ((r[sel] - delta[sel]) ** 2 + eta[sel] ** 2)))


@pytest.mark.parametrize("K", [8, 12])
@pytest.mark.parametrize("eta_max,fsteps,nw", [(0.02, 1000, 5000), (40.0, 2500, 9000), (0.3, 333, 1500), (3.0, 700, 1403)])
def test_wave_shared_walk_with_block_near_fields_equals_the_windowed_sum(eta_max, fsteps, nw, K):
    rng = np.random.default_rng(int(eta_max * 100) + fsteps + K)
    n = nw // 3
    c = np.sort(rng.integers(0, nw, n))
    delta = rng.uniform(-0.5, 0.5, n)
    eta = eta_max * rng.uniform(0.05, 1, n)
    A = 10.0 ** rng.uniform(-4, 0, n) * eta
    R = radius(eta_max, K)
    nlev = int(math.log2(fsteps)) - 1
    blocks = sorted({0, 64, (nw // 2) & ~63, ((nw - 1) & ~63), (fsteps & ~63), ((nw - fsteps) & ~63)} |
                    {int(b) & ~63 for b in rng.integers(0, nw, 4)})
    scale = max(exact_far_field(int(f), c, delta, eta, A, fsteps, -1) for f in range(0, nw, 97))
    levels = build_levels(level0_moments(nw, c, delta, eta, A, np.float64, K), nlev, np.float64)
    worst = 0.0
    for fb in blocks:
        got = block_far_field(fb, levels, nw, fsteps, R, eta_max, np.float64)
        for f, v in got.items():
            want = exact_far_field_blocks(f, c, delta, eta, A, fsteps, R)
            worst = max(worst, abs(v - want) / scale)
    assert worst < 1.5e-7, worst


def test_wave_shared_walk_keeps_the_window_edges_exact():
    """One strong line: with block near fields and a shared walk every point still sees it exactly when it is within
    fsteps of it (kernels.c:435-437) and outside the blocks its near field touches."""
    nw, fsteps, cpos, R = 6000, 1000, 2771, 3
    c, delta, eta, A = np.array([cpos]), np.array([0.37]), np.array([0.01]), np.array([1.0])
    levels = build_levels(level0_moments(nw, c, delta, eta, A, np.float64), 8, np.float64)
    for fb in ((cpos - fsteps - 1) & ~63, (cpos + fsteps) & ~63, cpos & ~63, (cpos & ~63) + 64, (cpos & ~63) - 64):
        got = block_far_field(fb, levels, nw, fsteps, R, 0.01, np.float64)
        for f, v in got.items():
            want = exact_far_field_blocks(f, c, delta, eta, A, fsteps, R)
            assert v == pytest.approx(want, rel=2e-7, abs=1e-300)
            assert (v == 0.0) == (want == 0.0), (fb, f)
