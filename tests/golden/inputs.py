"""
Seeded synthetic inputs shared by the golden-vector generator (make_golden.py, run once in the build
container against the imported reference) and by the tests (which regenerate the same inputs and
compare against the stored reference outputs).  Pure NumPy, no reference import.
"""

import numpy as np


def coefficients(seed, max_degree, scale=1e-10):
    """anm [N+1, N+1] ~ N(0,1) * scale  (default_rng(k), N(0,1)*1e-10)."""
    return np.random.default_rng(seed).standard_normal((max_degree + 1, max_degree + 1)) * scale


def spd_covariance(seed, size, scale=1e-22):
    """Seeded SPD matrix (G G^T)/k * scale with k = size + 16."""
    k = size + 16
    G = np.random.default_rng(seed).standard_normal((size, k))
    return (G @ G.T) / k * scale


def orderwise_random_blocks(seed, nmax):
    """[order0_cos, order1_cos, order1_sin, ...] random dense blocks, block m has shape (nmax+1-m)^2."""
    rng = np.random.default_rng(seed)
    blocks = [rng.standard_normal((nmax + 1, nmax + 1)) / (nmax + 1)]
    for m in range(1, nmax + 1):
        blocks.append(rng.standard_normal((nmax + 1 - m, nmax + 1 - m)) / (nmax + 1))
        blocks.append(rng.standard_normal((nmax + 1 - m, nmax + 1 - m)) / (nmax + 1))
    return blocks


def orderwise_normal_blocks(seed, nmax, scale=1e12):
    """Synthetic SPD order-wise normal-equation blocks (G G^T) * scale, reference block shapes."""
    rng = np.random.default_rng(seed)
    sizes = [nmax + 1]
    for m in range(1, nmax + 1):
        sizes += [nmax + 1 - m, nmax + 1 - m]
    blocks = []
    for s in sizes:
        G = rng.standard_normal((s, s + 4))
        blocks.append((G @ G.T) * scale)
    return blocks


def scattered_points(seed, count):
    """Longitude/latitude [rad] of `count` scattered points."""
    rng = np.random.default_rng(seed)
    lon = rng.uniform(-np.pi, np.pi, count)
    lat = np.arcsin(rng.uniform(-1, 1, count))
    return lon, lat


SPECIAL_COLAT = np.array([1e-3, 0.5 * np.pi, np.pi - 1e-3])
