"""
CPU oracle for the spherical-harmonic hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This module is a NumPy restatement of the reference algorithms on the hot path
(SURVEY.md section 8a).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package
``grates_amd`` never does (tests/test_boundary.py checks that).

Parity status: PINNED.  Every function below is checked in
``tests/test_oracle_golden.py`` against golden vectors produced by importing the
reference itself in the build container (``tests/golden/make_golden.py``), see
DESIGN.md section "Oracle".

Each function cites the reference location (relative to /root/reference) whose
arithmetic it follows.  The formulation is kept the same as the reference's
(same recursion, same dgemm-per-row synthesis, same per-parallel F @ Sigma)
because this file is also the CPU baseline timed by bench.py.
"""

import numpy as np

# ----------------------------------------------------------------------------------------------
# index maps (integer, must be bit-exact)                         grates/utilities.py:310-411
# ----------------------------------------------------------------------------------------------


def degree_indices(n, max_order=None):
    """(rows, cols) of all coefficients of degree n, cosines first.  grates/gravityfield.py:15-40"""
    count = n if max_order is None else min(n, max_order)
    rows = np.concatenate((np.full(count + 1, n, dtype=int), np.arange(count, dtype=int)))
    cols = np.concatenate((np.arange(count + 1, dtype=int), np.full(count, n, dtype=int)))
    return rows, cols


def order_indices(max_degree, m):
    """(rows, cols) of all coefficients of order m, cosines first.  grates/gravityfield.py:43-73"""
    rows = np.arange(m, max_degree + 1, dtype=int)
    cols = np.full(rows.size, m, dtype=int)
    if m > 0:
        rows = np.concatenate((rows, np.full(max_degree + 1 - m, m - 1, dtype=int)))
        cols = np.concatenate((cols, np.arange(m, max_degree + 1, dtype=int)))
    return rows, cols


def degreewise_sequence(min_degree, max_degree):
    """
    Degree-wise coefficient order C00, C10, C11, S11, C20, ... as an int array [P, 3] of
    (basis_function 0=c/1=s, degree, order).  grates/gravityfield.py:1291-1332
    """
    seq = []
    for n in range(min_degree, max_degree + 1):
        seq.append((0, n, 0))
        for m in range(1, n + 1):
            seq.append((0, n, m))
            seq.append((1, n, m))
    return np.array(seq, dtype=np.int64).reshape(-1, 3)


def degreewise_array_index(min_degree, max_degree):
    """Row/column of each degree-wise vector entry inside the packed [N+1, N+1] array
    (C_nm at [n, m]; S_nm at [m-1, n]).  grates/utilities.py:336-343"""
    seq = degreewise_sequence(min_degree, max_degree)
    is_s = seq[:, 0] == 1
    rows = np.where(is_s, seq[:, 2] - 1, seq[:, 1])
    cols = np.where(is_s, seq[:, 1], seq[:, 2])
    return rows, cols


def ravel_coefficients(array, min_degree=0, max_degree=None):
    """grates/utilities.py:310-360 (degrees beyond the array are left zero)."""
    if max_degree is None:
        max_degree = array.shape[-1] - 1
    count = (max_degree + 1) ** 2 - min_degree ** 2
    if array.ndim not in (2, 3):
        raise ValueError('Only 2d or 3d spherical harmonic arrays can be raveled.')
    top = min(array.shape[-1] - 1, max_degree)
    out = np.zeros(array.shape[:-2] + (count,), dtype=array.dtype)
    if top >= min_degree:
        rows, cols = degreewise_array_index(min_degree, top)
        out[..., 0:rows.size] = array[..., rows, cols]
    return out


def unravel_coefficients(vector, min_degree=0, max_degree=None):
    """grates/utilities.py:363-411"""
    if max_degree is None:
        max_degree = int(np.sqrt(vector.shape[-1] + min_degree * min_degree) - 1)
    if vector.ndim not in (1, 2):
        raise ValueError('Only 1d or 2d spherical harmonic vectors can be unraveled.')
    out = np.zeros(vector.shape[:-1] + (max_degree + 1, max_degree + 1), dtype=vector.dtype)
    rows, cols = degreewise_array_index(min_degree, max_degree)
    out[..., rows, cols] = vector[..., 0:rows.size]
    return out


# ----------------------------------------------------------------------------------------------
# Legendre / trigonometric tables
# ----------------------------------------------------------------------------------------------


def legendre_functions(max_degree, colat):
    """
    4pi-normalised associated Legendre functions, packed [k, N+1, N+1] (P_nm at [n, m] and
    mirrored into the sine slot [m-1, n]).  grates/utilities.py:13-59.
    Column-by-column recursion; the per-element arithmetic (order of the multiplications) is the
    reference's:  sqrt(..) * cos * P[n-1,m]  -  sqrt(..) * P[n-2,m].
    """
    theta = np.atleast_1d(np.asarray(colat, dtype=float))
    N = max_degree
    P = np.empty((theta.size, N + 1, N + 1))
    P[:, 0, 0] = 1.0
    if N == 0:
        return P
    ct, st = np.cos(theta), np.sin(theta)
    P[:, 1, 0] = np.sqrt(3) * ct
    P[:, 1, 1] = np.sqrt(3) * st
    for n in range(2, N + 1):                                   # sectorials          :41-43
        P[:, n, n] = np.sqrt((2.0 * n + 1.0) / (2.0 * n)) * st * P[:, n - 1, n - 1]
    for n in range(2, N + 1):                                   # first off-diagonal  :45-47
        P[:, n, n - 1] = np.sqrt(2 * n + 1) * ct * P[:, n - 1, n - 1]
    for m in range(0, N - 1):                                   # three-term recursion :49-54
        for n in range(m + 2, N + 1):
            P[:, n, m] = np.sqrt((2.0 * n - 1.0) / (n - m) * (2.0 * n + 1.0) / (n + m)) * ct * P[:, n - 1, m] - \
                np.sqrt((2.0 * n + 1.0) / (2.0 * n - 3.0) * (n - m - 1.0) / (n - m) * (n + m - 1.0) / (n + m)) * P[:, n - 2, m]
    for m in range(1, N + 1):                                   # mirror              :56-57
        P[:, m - 1, m:] = P[:, m:, m]
    return P


def legendre_polynomials(max_degree, colat):
    """Order-0 column with its own coefficient form.  grates/utilities.py:138-151"""
    t = np.cos(np.atleast_1d(np.asarray(colat, dtype=float)))
    P = np.empty((t.size, max_degree + 1))
    P[:, 0] = 1
    if max_degree == 0:
        return P
    P[:, 1] = np.sqrt(3) * t
    for n in range(2, max_degree + 1):
        P[:, n] = np.sqrt((2.0 * n - 1.0) * (2.0 * n + 1.0)) / n * t * P[:, n - 1] - \
            np.sqrt((2.0 * n + 1.0) / (2.0 * n - 3.0)) * (n - 1.0) / n * P[:, n - 2]
    return P


def legendre_functions_per_order(max_degree, order, colat):
    """P_nm for one order, n = m..N, with s = sqrt(1 - t^2).  grates/utilities.py:62-115"""
    if order == 0:
        return legendre_polynomials(max_degree, colat)
    if order > max_degree:
        raise ValueError('order exceeds maximum degree ({0:d} vs. {1:d})'.format(order, max_degree))
    t = np.cos(np.atleast_1d(np.asarray(colat, dtype=float)))
    s = np.sqrt(1 - t ** 2)
    out = np.empty((t.size, max_degree + 1 - order))
    pmm = np.sqrt(3) * s
    for n in range(2, order + 1):
        pmm = np.sqrt((2 * n + 1) / (2 * n)) * s * pmm
    out[:, 0] = pmm
    if out.shape[1] > 1:
        out[:, 1] = np.sqrt(2 * order + 3) * t * out[:, 0]
    for n in range(order + 2, max_degree + 1):
        out[:, n - order] = np.sqrt((2 * n - 1) / (n - order) * (2 * n + 1) / (n + order)) * t * out[:, n - 1 - order] - \
            np.sqrt((2 * n + 1) / (2 * n - 3) * (n - order - 1) / (n - order) * (n + order - 1) / (n + order)) * out[:, n - 2 - order]
    return out


def trigonometric_functions(max_degree, lon):
    """cos(m lon) at [n>=m, m], sin(m lon) at [m-1, n>=m].  grates/utilities.py:249-275"""
    lam = np.atleast_1d(np.asarray(lon, dtype=float))
    cs = np.empty((lam.size, max_degree + 1, max_degree + 1))
    cs[:, :, 0] = 1
    for m in range(1, max_degree + 1):
        cs[:, m:, m] = np.cos(m * lam)[:, np.newaxis]
        cs[:, m - 1, m:] = np.sin(m * lam)[:, np.newaxis]
    return cs


def spherical_harmonics(max_degree, colat, lon):
    """grates/utilities.py:278-307"""
    count = max(np.asarray(colat).size, np.asarray(lon).size)
    Y = np.ones((count, max_degree + 1, max_degree + 1))
    Y *= trigonometric_functions(max_degree, lon)
    Y *= legendre_functions(max_degree, colat)
    return Y


# ----------------------------------------------------------------------------------------------
# ellipsoid geometry                                              grates/utilities.py:414-459
# ----------------------------------------------------------------------------------------------

GRS80_A = 6378137.0
GRS80_F = 298.2572221010 ** -1


def geocentric_radius(latitude, a=GRS80_A, f=GRS80_F):
    e2 = f * (2 - f)
    nu = a / np.sqrt(1 - e2 * np.sin(latitude) ** 2)
    return nu * np.sqrt(np.cos(latitude) ** 2 + (1 - e2) ** 2 * np.sin(latitude) ** 2)


def colatitude(latitude, a=GRS80_A, f=GRS80_F):
    e2 = f * (2 - f)
    nu = a / np.sqrt(1 - e2 * np.sin(latitude) ** 2)
    return np.arccos(nu * (1 - e2) * np.sin(latitude) / geocentric_radius(latitude, a, f))


def geographic_grid(dlon, dlat):
    """meridians, parallels (north -> south), area[nlat, nlon].  grates/grid.py:1146-1151"""
    nlons, nlats = int(360 / dlon), int(180 / dlat)
    meridians = np.linspace(-np.pi + dlon / 180 * np.pi * 0.5, np.pi - dlon / 180 * np.pi * 0.5, nlons)
    parallels = -np.linspace(-np.pi * 0.5 + dlat / 180 * np.pi * 0.5, np.pi * 0.5 - dlat / 180 * np.pi * 0.5, nlats)
    areas = np.tile(2.0 * dlon / 180 * np.pi * np.sin(dlat * 0.5 / 180 * np.pi) * np.cos(parallels)[:, np.newaxis], (1, nlons))
    return meridians, parallels, areas


def gauss_grid(parallel_count, f=GRS80_F):
    """grates/grid.py:1181-1195"""
    from scipy.special import roots_legendre
    zeros, weights, _ = roots_legendre(parallel_count, mu=True)
    dlon = np.pi / parallel_count
    meridians = np.linspace(-np.pi + dlon * 0.5, np.pi - dlon * 0.5, 2 * parallel_count)
    ct = -zeros
    stt = np.sqrt(1 - ct ** 2)
    parallels = np.arctan2(ct, (1 - f) ** 2 * stt)
    areas = np.tile(dlon * weights[:, np.newaxis], (1, meridians.size))
    return meridians, parallels, areas


# ----------------------------------------------------------------------------------------------
# isotropic kernels (the ones the parity tests need)              grates/kernel.py
# ----------------------------------------------------------------------------------------------


class KernelTable:
    """coefficients / inverse_coefficients [npoints, nmax-nmin+1].  grates/kernel.py:85-188"""

    def __init__(self, name, love_numbers=None, rho=1025):
        self.name = name.lower()
        self.k = love_numbers
        self.rho = rho
        if self.name in ('ewh', 'water_height') and love_numbers is None:
            raise ValueError('ewh kernel needs load Love numbers')

    def coefficients(self, nmin, nmax, r, colat):
        r = np.atleast_1d(np.asarray(r, dtype=float))
        if self.name == 'potential':                             # kernel.py:445-449
            return np.ones((r.size, nmax + 1 - nmin))
        if self.name in ('ewh', 'water_height'):                 # kernel.py:403-406
            kn = (4 * np.pi * 6.673e-11 * self.rho) * (1 + self.k[nmin:nmax + 1]) / (2 * np.arange(nmin, nmax + 1, dtype=float) + 1)
            return (kn[:, np.newaxis] * r).T
        raise ValueError("oracle: unsupported kernel '{0}'".format(self.name))

    def inverse_coefficients(self, nmin, nmax, r, colat):        # kernel.py:187-188
        kn = self.coefficients(nmin, nmax, r, colat)
        return np.vstack([np.zeros(kn.shape[0]) if np.allclose(kn[:, k], 0.0) else 1.0 / kn[:, k] for k in range(kn.shape[1])]).T


def gauss_weights(radius_km, max_degree):
    """Jekeli recursion, zero after the first weight below 1e-7.  grates/kernel.py:468-506
    (valid for max_degree <= 1024, the table length the reference allocates up front)."""
    if radius_km < 0:
        raise ValueError('Gaussian filter radius must be positive')
    nmax = 1024
    if max_degree > nmax:
        raise ValueError('oracle gauss_weights restated for max_degree <= 1024 only')
    if radius_km == 0:
        return np.ones(max_degree + 1)
    b = np.log(2.0) / (1 - np.cos(radius_km / 6378.1366))
    wn = np.zeros(nmax + 1)
    wn[0] = 1.0
    wn[1] = (1 + np.exp(-2 * b)) / (1 - np.exp(-2 * b)) - 1 / b
    for n in range(2, nmax + 1):
        wn[n] = -(2 * n - 1) / b * wn[n - 1] + wn[n - 2]
        if wn[n] < 1e-7:
            break
    return wn[0:max_degree + 1]


def kn_table(kernel, max_degree, parallels_or_lat, GM, R, a=GRS80_A, f=GRS80_F):
    """kn[i, n] = (1/k_n(r_i, theta_i)) (R/r_i)^(n+1) GM/R.  grates/gravityfield.py:353-356"""
    colat = colatitude(parallels_or_lat, a, f)
    radius = geocentric_radius(parallels_or_lat, a, f)
    kn = kernel.inverse_coefficients(0, max_degree, radius, colat) * \
        np.power((R / radius)[:, np.newaxis], np.arange(max_degree + 1, dtype=int) + 1) * GM / R
    return colat, radius, kn


def scale_packed_by_degree(T, kn):
    """Multiply a packed [k, N+1, N+1] table by kn[k, n] (degree of every slot).
    grates/gravityfield.py:359-362"""
    N = T.shape[-1] - 1
    T[:, :, 0] *= kn
    for m in range(1, N + 1):
        T[:, m:, m] *= kn[:, m:]
        T[:, m - 1, m:] *= kn[:, m:]
    return T


# ----------------------------------------------------------------------------------------------
# synthesis                                                        grates/gravityfield.py:331-390
# ----------------------------------------------------------------------------------------------

GM_DEFAULT = 3.9860044150e+14
R_DEFAULT = 6.3781363000e+06


def synthesis_regular(anm, meridians, parallels, kernel, GM=GM_DEFAULT, R=R_DEFAULT, a=GRS80_A, f=GRS80_F):
    """Regular-grid synthesis, N+1 dgemms as in the reference.  gravityfield.py:352-368.
    Returns value_array [nlat, nlon]."""
    N = anm.shape[0] - 1
    colat, _, kn = kn_table(kernel, N, parallels, GM, R, a, f)
    Pnm = scale_packed_by_degree(legendre_functions(N, colat), kn)
    Pnm *= anm[np.newaxis, :, :]
    cs = trigonometric_functions(N, meridians)
    values = np.zeros((parallels.size, meridians.size))
    for k in range(N + 1):
        values += Pnm[:, k, :] @ cs[:, k, :].T
    return values


def synthesis_points(anm, longitude, latitude, kernel, GM=GM_DEFAULT, R=R_DEFAULT, a=GRS80_A, f=GRS80_F):
    """Point-list synthesis in blocks of 512.  gravityfield.py:370-388"""
    N = anm.shape[0] - 1
    values = np.zeros(longitude.size)
    for i1 in range(0, longitude.size, 512):
        i2 = min(i1 + 512, longitude.size)
        colat, _, kn = kn_table(kernel, N, latitude[i1:i2], GM, R, a, f)
        Ynm = scale_packed_by_degree(spherical_harmonics(N, colat, longitude[i1:i2]), kn)
        for k in range(N + 1):
            values[i1:i2] += Ynm[:, k, :] @ anm[k, :]
    return values


# ----------------------------------------------------------------------------------------------
# design matrices, analysis                                       grates/grid.py:412-443, 627-790
# ----------------------------------------------------------------------------------------------


def synthesis_matrix_per_order(m, min_degree, max_degree, meridians, parallels, kernel, GM=GM_DEFAULT, R=R_DEFAULT,
                               a=GRS80_A, f=GRS80_F):
    """grid.py:627-663 -- rows ordered parallel-major, columns n = max(m, nmin)..N."""
    colat, _, kn = kn_table(kernel, max_degree, parallels, GM, R, a, f)
    Pm = (legendre_functions_per_order(max_degree, m, colat) * kn[:, m:])[:, max(min_degree - m, 0):]
    if m == 0:
        return np.repeat(Pm, meridians.size, axis=0)
    c = np.cos(m * meridians)
    s = np.sin(m * meridians)
    Ac = (Pm[:, np.newaxis, :] * c[np.newaxis, :, np.newaxis]).reshape(-1, Pm.shape[1])
    As = (Pm[:, np.newaxis, :] * s[np.newaxis, :, np.newaxis]).reshape(-1, Pm.shape[1])
    return Ac, As


def vector_indices(min_degree, max_degree, order, cs=None):
    """grates/gravityfield.py:1226-1262 restricted to what the design matrices use."""
    seq = degreewise_sequence(min_degree, max_degree)
    mask = seq[:, 2] == order
    if cs is not None:
        mask &= seq[:, 0] == (0 if cs == 'c' else 1)
    return np.where(mask)[0]


def synthesis_matrix(min_degree, max_degree, meridians, parallels, kernel, GM=GM_DEFAULT, R=R_DEFAULT,
                     a=GRS80_A, f=GRS80_F):
    """grid.py:412-443"""
    P = (max_degree + 1) ** 2 - min_degree ** 2
    A = np.empty((parallels.size * meridians.size, P))
    A[:, vector_indices(min_degree, max_degree, 0)] = synthesis_matrix_per_order(0, min_degree, max_degree, meridians, parallels, kernel, GM, R, a, f)
    for m in range(1, max_degree + 1):
        idx = np.concatenate((vector_indices(min_degree, max_degree, m, 'c'), vector_indices(min_degree, max_degree, m, 's')))
        A[:, idx] = np.hstack(synthesis_matrix_per_order(m, min_degree, max_degree, meridians, parallels, kernel, GM, R, a, f))
    return A


def analysis_regular(values, area, min_degree, max_degree, meridians, parallels, kernel, GM=GM_DEFAULT, R=R_DEFAULT,
                     a=GRS80_A, f=GRS80_F, orders=None):
    """Area-weighted least squares per order and per cos/sin.  grid.py:665-696, 752-790.
    values, area: flattened [nlat*nlon].  Returns anm [N+1, N+1].
    orders: the orders to solve (default: all, like the reference's loop grid.py:779-785); the orders are independent
    least-squares problems, the others stay zero (bounded samples of the benchmark's CPU baseline)."""
    anm = np.zeros((max_degree + 1, max_degree + 1))
    w = area[:, np.newaxis]
    wanted = set(range(max_degree + 1)) if orders is None else set(int(m) for m in orders)

    def lsq(A):
        return np.linalg.solve((A * w).T @ A, (A * w).T) @ values

    if 0 in wanted:
        anm[min_degree:, 0] = lsq(synthesis_matrix_per_order(0, min_degree, max_degree, meridians, parallels, kernel, GM, R, a, f))
    for m in sorted(wanted - {0}):
        Ac, As = synthesis_matrix_per_order(m, min_degree, max_degree, meridians, parallels, kernel, GM, R, a, f)
        start = max(m, min_degree)
        anm[start:, m] = lsq(Ac)
        anm[m - 1, start:] = lsq(As)
    return anm


def analysis_matrix_regular(area, min_degree, max_degree, meridians, parallels, kernel, GM=GM_DEFAULT, R=R_DEFAULT,
                            a=GRS80_A, f=GRS80_F):
    """Dense analysis operator [P, nlat*nlon] in degree-wise row order, per order and per cos/sin
    solve((A w)^T A, (A w)^T).  grid.py:665-696, 698-730"""
    P = (max_degree + 1) ** 2 - min_degree ** 2
    F = np.empty((P, area.size))
    w = np.ravel(area)[:, np.newaxis]

    def lsq(A):
        return np.linalg.solve((A * w).T @ A, (A * w).T)

    F[vector_indices(min_degree, max_degree, 0), :] = lsq(synthesis_matrix_per_order(0, min_degree, max_degree, meridians, parallels, kernel, GM, R, a, f))
    for m in range(1, max_degree + 1):
        Ac, As = synthesis_matrix_per_order(m, min_degree, max_degree, meridians, parallels, kernel, GM, R, a, f)
        F[vector_indices(min_degree, max_degree, m, 'c'), :] = lsq(Ac)
        F[vector_indices(min_degree, max_degree, m, 's'), :] = lsq(As)
    return F


def window_matrix_regular(values, area, min_degree, max_degree, meridians, parallels, kernel, GM=GM_DEFAULT, R=R_DEFAULT,
                          a=GRS80_A, f=GRS80_F):
    """W = (F * values) A, the grid values as window function.  grid.py:449-475"""
    F = analysis_matrix_regular(area, min_degree, max_degree, meridians, parallels, kernel, GM, R, a, f)
    return (F * np.ravel(values)) @ synthesis_matrix(min_degree, max_degree, meridians, parallels, kernel, GM, R, a, f)


# ----------------------------------------------------------------------------------------------
# covariance propagation                                           grates/grid.py:792-839, 1071-1120
# ----------------------------------------------------------------------------------------------


def covariance_blocks_regular(cov, min_degree, max_degree, meridians, parallels, kernel, rows, GM=GM_DEFAULT, R=R_DEFAULT,
                              a=GRS80_A, f=GRS80_F):
    """F Sigma F^T [nlon, nlon] for the parallels `rows`: the per-parallel product of which covariance_propagation keeps the
    diagonal.  grid.py:825-835"""
    colat, _, kn = kn_table(kernel, max_degree, parallels, GM, R, a, f)
    Pnm = scale_packed_by_degree(legendre_functions(max_degree, colat), kn)
    Pnm = ravel_coefficients(Pnm, min_degree, max_degree)
    cs = ravel_coefficients(trigonometric_functions(max_degree, meridians), min_degree, max_degree)
    out = []
    for k in rows:
        F = cs * Pnm[k:k + 1, :]
        out.append(F @ cov @ F.T)
    return np.stack(out)



def covariance_propagation_regular(cov, min_degree, max_degree, meridians, parallels, kernel, GM=GM_DEFAULT, R=R_DEFAULT,
                                   a=GRS80_A, f=GRS80_F, parallel_range=None):
    """sigma[i*nlon + j] = sqrt(a_ij^T Sigma a_ij), per parallel F @ Sigma then row-dot.
    grid.py:817-839 (the reference forms diag(F Sigma F^T); the row-dot is the same numbers).
    parallel_range=(k0, k1) restricts the computation to a band of parallels (used by the
    bounded CPU baseline in bench.py)."""
    colat, _, kn = kn_table(kernel, max_degree, parallels, GM, R, a, f)
    Pnm = scale_packed_by_degree(legendre_functions(max_degree, colat), kn)
    Pnm = ravel_coefficients(Pnm, min_degree, max_degree)
    cs = ravel_coefficients(trigonometric_functions(max_degree, meridians), min_degree, max_degree)
    k0, k1 = (0, parallels.size) if parallel_range is None else parallel_range
    out = np.zeros((k1 - k0) * meridians.size)
    for k in range(k0, k1):
        F = cs * Pnm[k:k + 1, :]
        out[(k - k0) * meridians.size:(k - k0 + 1) * meridians.size] = np.einsum('ij,ij->i', F @ cov, F)
    return np.sqrt(out)


def covariance_propagation_points(cov, min_degree, max_degree, longitude, latitude, kernel, GM=GM_DEFAULT, R=R_DEFAULT,
                                  a=GRS80_A, f=GRS80_F):
    """Point-list variant, blocks of 256.  grid.py:1096-1120"""
    out = np.zeros(longitude.size)
    for i1 in range(0, longitude.size, 256):
        i2 = min(i1 + 256, longitude.size)
        colat, _, kn = kn_table(kernel, max_degree, latitude[i1:i2], GM, R, a, f)
        Ynm = scale_packed_by_degree(spherical_harmonics(max_degree, colat, longitude[i1:i2]), kn)
        F = ravel_coefficients(Ynm, min_degree, max_degree)
        out[i1:i2] = np.einsum('ij,ij->i', F @ cov, F)
    return np.sqrt(out)


# ----------------------------------------------------------------------------------------------
# filters                                                           grates/filter.py
# ----------------------------------------------------------------------------------------------


def gaussian_filter(anm, radius_km):
    """Degree-wise scaling of degrees >= 2.  filter.py:61-72"""
    N = anm.shape[0] - 1
    wn = gauss_weights(radius_km, N)
    out = anm.copy()
    for n in range(2, N + 1):
        out[degree_indices(n)] *= wn[n]
    return out


def gaussian_matrix(radius_km, min_degree, max_degree):
    """Diagonal matrix, scales every degree >= min_degree (degree 1 too).  filter.py:90-95"""
    wn = gauss_weights(radius_km, max_degree)
    arr = np.zeros((max_degree + 1, max_degree + 1))
    for n in range(min_degree, max_degree + 1):
        arr[degree_indices(n)] = wn[n]
    return np.diag(ravel_coefficients(arr, min_degree, max_degree))


def orderwise_filter(anm, blocks):
    """Per-order block mat-vec, degrees 0-1 restored.  filter.py:175-191.
    blocks = [order0_cos, order1_cos, order1_sin, ...], block m indexed by degree m..nmax."""
    nmax = anm.shape[0] - 1
    block_nmax = blocks[0].shape[0] - 1
    if nmax > block_nmax:
        raise ValueError('DDK filter only implemented for a maximum degree of {1:d} (max_degree={0:d} supplied).'.format(nmax, block_nmax))
    out = anm.copy()
    out[:, 0] = blocks[0][0:nmax + 1, 0:nmax + 1] @ anm[:, 0]
    for m in range(1, nmax + 1):
        out[m:, m] = blocks[2 * m - 1][0:nmax + 1 - m, 0:nmax + 1 - m] @ anm[m:, m]
        out[m - 1, m:] = blocks[2 * m][0:nmax + 1 - m, 0:nmax + 1 - m] @ anm[m - 1, m:]
    out[0:2, 0:2] = anm[0:2, 0:2]
    return out


def orderwise_matrix(blocks, min_degree, max_degree):
    """Scatter the 2N+1 blocks into the dense degree-wise matrix.  filter.py:209-222"""
    count = (max_degree + 1) ** 2
    W = np.zeros((count, count))
    index = np.arange(max_degree + 1, dtype=int) ** 2
    W[np.ix_(index, index)] = blocks[0][0:max_degree + 1, 0:max_degree + 1]
    for m in range(1, max_degree + 1):
        W[np.ix_(index[m:] + 2 * m - 1, index[m:] + 2 * m - 1)] = blocks[2 * m - 1][0:max_degree + 1 - m, 0:max_degree + 1 - m]
        W[np.ix_(index[m:] + 2 * m, index[m:] + 2 * m)] = blocks[2 * m][0:max_degree + 1 - m, 0:max_degree + 1 - m]
    return W[min_degree * min_degree:, min_degree * min_degree:]


DDK_SCALE = {1: 1e14, 2: 1e13, 3: 1e12, 4: 5e11, 5: 1e11, 6: 5e10, 7: 1e10, 8: 5e9}


def ddk_blocks(normal_blocks, level, generic=False):
    """(N_m + diag(w))^-1 N_m with w_n = scale n^4, w_0 = 1.  filter.py:242-257, 332-349"""
    nmax = normal_blocks[0].shape[0] - 1
    if generic:
        if level < 1:
            raise ValueError('DDK level must be at least 1')
        scale = 10 ** (15 - level)
    else:
        if level not in DDK_SCALE:
            raise ValueError('DDK level must be between 1 and 8')
        scale = DDK_SCALE[level]
    weights = scale * np.arange(nmax + 1, dtype=float) ** 4
    weights[0] = 1
    out = []
    for blk in normal_blocks:
        m = nmax + 1 - blk.shape[0]
        out.append(np.linalg.solve(blk + np.diag(weights[m:]), blk))
    return out


def general_matrix_filter(anm, W, min_degree, max_degree):
    """Dense W @ x in degree-wise order; degrees < nmin restored.  filter.py:470-479"""
    nmax_out = min(anm.shape[0] - 1, max_degree)
    x = ravel_coefficients(anm, min_degree, max_degree)
    out = unravel_coefficients(W @ x, min_degree, nmax_out)
    out[0:min_degree, 0:min_degree] = anm[0:min_degree, 0:min_degree]
    return out


# ----------------------------------------------------------------------------------------------
# dense decorrelation filter from a full normal matrix, filter kernels in the space domain
#                                                      grates/filter.py:536-546, 588-598; grates/kernel.py:589-654
# ----------------------------------------------------------------------------------------------


def vdk_matrix(normals, min_degree, max_degree, kaula_scale, kaula_power):
    """W = (N + diag(w))^-1 N with Kaula weights w = kaula_scale n^kaula_power of every degree-wise index.
    grates/filter.py:538-546"""
    weights = np.concatenate([np.full(2 * n + 1, kaula_scale * float(n) ** kaula_power) for n in range(min_degree, max_degree + 1)])
    return np.linalg.solve(normals + np.diag(weights), normals)


def filter_kernel_matrix(K, min_degree, max_degree, kernel):
    """Matrix of grates.filter.FilterKernel AS THE REFERENCE EXECUTES IT (grates/filter.py:590-596): the kernel coefficient
    arrays carry a leading axis of length one, so both factors of
        K2 = (K * kn[np.newaxis, :]) * kn_inverse[:, np.newaxis]
    broadcast along the COLUMNS: K2[i][j] = K[i][j] k_n(j) / k_n(j) -- K itself wherever the kernel coefficient of
    column j is non-zero, zero where it vanishes -- whatever `input_kernel` is (the fixture holds identical values
    for 'potential' and 'ewh').  Kernel coefficients at r = 6378136.3 m."""
    kn = kernel.coefficients(min_degree, max_degree, 6378136.3, 0)[0]
    kinv = kernel.inverse_coefficients(min_degree, max_degree, 6378136.3, 0)[0]
    per_index = lambda t: np.concatenate([np.full(2 * n + 1, t[n - min_degree]) for n in range(min_degree, max_degree + 1)])
    return K * per_index(kn)[np.newaxis, :] * per_index(kinv)[np.newaxis, :]


def anisotropic_kernel_points(K, min_degree, max_degree, source_longitude, source_latitude, longitude, latitude):
    """y(source)^T K y(point) for a list of points.  grates/kernel.py:615-620"""
    ys = ravel_coefficients(spherical_harmonics(max_degree, np.atleast_1d(0.5 * np.pi - source_latitude), np.atleast_1d(source_longitude))[0],
                            min_degree, max_degree)
    v = ys @ K
    out = np.empty(np.size(longitude))
    for k, (lon, lat) in enumerate(zip(np.atleast_1d(longitude), np.atleast_1d(latitude))):
        ye = ravel_coefficients(spherical_harmonics(max_degree, np.atleast_1d(0.5 * np.pi - lat), np.atleast_1d(lon))[0], min_degree, max_degree)
        out[k] = v @ ye
    return out


def anisotropic_kernel_grid(K, min_degree, max_degree, source_longitude, source_latitude, meridians, parallels):
    """The same on meridians x parallels, parallel by parallel.  grates/kernel.py:642-654"""
    ys = ravel_coefficients(spherical_harmonics(max_degree, np.atleast_1d(0.5 * np.pi - source_latitude), np.atleast_1d(source_longitude))[0],
                            min_degree, max_degree)
    v = ys @ K
    pnm = legendre_functions(max_degree, 0.5 * np.pi - np.asarray(parallels, dtype=float))
    cs = trigonometric_functions(max_degree, np.asarray(meridians, dtype=float))
    grid = np.empty((np.size(parallels), np.size(meridians)))
    for k in range(grid.shape[0]):
        grid[k, :] = np.stack([ravel_coefficients(cs[j] * pnm[k], min_degree, max_degree) for j in range(cs.shape[0])]) @ v
    return grid
