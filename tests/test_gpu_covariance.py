"""
GPU parity of covariance propagation (rows a14 / a15 of SURVEY.md 8a) and of the fp64 MFMA GEMM it runs on.
Tolerance for sigma: 1e-11 relative (square root of a P^2-term sum).
"""

import numpy as np
import pytest

import grates_amd as ga
import inputs
from conftest import relerr
from oracle import shg_oracle as orc

pytestmark = pytest.mark.gpu
TOL_SIGMA = 1e-11


def love():
    return ga.data.load_love_numbers()[0]


@pytest.mark.parametrize('M,N,K', [(1, 1, 1), (16, 16, 4), (128, 128, 16), (130, 70, 33), (257, 129, 100), (64, 300, 7), (500, 3, 511)])
def test_dgemm_against_numpy(M, N, K):
    rng = np.random.default_rng(M * 1000 + N * 10 + K)
    A, B = rng.standard_normal((M, K)), rng.standard_normal((K, N))
    C = ga.engine.to_host(ga.engine.dgemm(A, B))
    assert C.shape == (M, N)
    assert relerr(C, A @ B) < 1e-14


@pytest.mark.parametrize('M,N,K', [(2570, 2700, 515), (2560, 2688, 528), (3001, 2049, 1000)])
def test_dgemm_large_tile_counts(M, N, K):
    """Shapes with more than 384 output tiles stay on the plain MFMA kernel (odd sizes: 8-byte operand loads and a partial
    last K tile; even sizes: 16-byte loads); checked against torch on the device."""
    import torch
    gen = torch.Generator(device='cuda')
    gen.manual_seed(M + N + K)
    A = torch.randn((M, K), dtype=torch.float64, device='cuda', generator=gen)
    B = torch.randn((K, N), dtype=torch.float64, device='cuda', generator=gen)
    C = ga.engine.dgemm(A, B)
    ref = A @ B
    assert float(((C - ref).abs().max() / ref.abs().max()).item()) < 1e-13


def test_dgemm_layout_with_asymmetric_operand():
    # A = I with an asymmetric B catches transposed fragment layouts
    n = 48
    B = np.arange(n * n, dtype=float).reshape(n, n)
    np.testing.assert_array_equal(ga.engine.to_host(ga.engine.dgemm(np.eye(n), B)), B)
    np.testing.assert_array_equal(ga.engine.to_host(ga.engine.dgemm(B, np.eye(n))), B)


def test_golden_regular_grids(golden):
    g = golden('g9_covariance')
    grid = ga.grid.GeographicGrid(2.0, 2.0)
    s = grid.covariance_propagation(inputs.spd_covariance(31, 41 * 41), 0, 40, kernel='ewh')
    assert s.shape == (16200,) and relerr(s, g['n40_2deg_ewh']) < TOL_SIGMA
    np.testing.assert_array_equal(grid.values, s)                      # the method also sets the grid values
    grid = ga.grid.GeographicGrid(5.0, 5.0)
    s = grid.covariance_propagation(inputs.spd_covariance(32, 21 * 21 - 4), 2, 20, kernel='potential')
    assert relerr(s, g['n20_5deg_min2_potential']) < TOL_SIGMA
    cov = inputs.spd_covariance(33, 21 * 21)
    s = grid.covariance_propagation(cov, 0, 20, kernel='ewh')
    assert relerr(s, g['n20_5deg_ewh']) < TOL_SIGMA
    assert relerr(s, g['n20_5deg_ewh_einsum']) < TOL_SIGMA
    band = grid.covariance_propagation(cov, 0, 20, kernel='ewh', parallel_range=(3, 7))
    np.testing.assert_array_equal(band, s[3 * 72:7 * 72])              # latitude-band sharding is bit-reproducible
    with pytest.raises(ValueError):
        grid.covariance_propagation(cov[:-1, :-1], 0, 20)


def test_golden_point_list(golden):
    g = golden('g9_covariance')
    lon, lat = inputs.scattered_points(34, 300)
    grid = ga.grid.IrregularGrid(lon, lat)
    s = grid.covariance_propagation(inputs.spd_covariance(33, 21 * 21), 0, 20, kernel='ewh')
    assert relerr(s, g['points_n20_ewh']) < TOL_SIGMA
    np.testing.assert_array_equal(grid.values, s)


@pytest.mark.parametrize('N,nmin,dlon,dlat', [(3, 0, 30, 30), (12, 3, 7.5, 10), (33, 0, 3, 4), (60, 2, 2, 3)])
def test_against_oracle(N, nmin, dlon, dlat):
    """Ragged tiles: P and the number of grid rows are not multiples of the 128 x 128 block tile."""
    grid = ga.grid.GeographicGrid(dlon, dlat)
    P = (N + 1) ** 2 - nmin ** 2
    cov = inputs.spd_covariance(200 + N, P)
    ref = orc.covariance_propagation_regular(cov, nmin, N, grid.meridians, grid.parallels, orc.KernelTable('ewh', love()))
    s = grid.covariance_propagation(cov, nmin, N, kernel='ewh')
    assert relerr(s, ref) < TOL_SIGMA


def test_properties_identity_and_scaling():
    """Size-independent properties: Sigma = I gives the row norm of the synthesis matrix; sigma scales with sqrt(c)."""
    N = 50
    grid = ga.grid.GeographicGrid(2.0, 2.0)
    P = (N + 1) ** 2
    s1 = grid.covariance_propagation(np.eye(P), 0, N, kernel='potential')
    unit = np.zeros((N + 1, N + 1))
    ker = orc.KernelTable('potential')
    colat, _, kn = orc.kn_table(ker, N, grid.parallels, orc.GM_DEFAULT, orc.R_DEFAULT)
    Pnm = orc.scale_packed_by_degree(orc.legendre_functions(N, colat), kn)
    # sum_p a_p^2 = sum_nm (kn P_nm)^2 (cos^2 + sin^2) = sum over the packed array with order-0 entries counted once
    w = np.ones((N + 1, N + 1))
    w[np.triu_indices(N + 1, 1)] = 0.0                   # sine slots: cos^2 + sin^2 = 1 is already in the cosine slot
    ref = np.sqrt(np.einsum('knm,nm->k', Pnm ** 2, w))
    assert relerr(s1.reshape(90, 180), np.repeat(ref[:, None], 180, axis=1)) < 1e-12
    cov = inputs.spd_covariance(5, P)
    a = grid.covariance_propagation(cov, 0, N, kernel='potential')
    b = grid.covariance_propagation(4.0 * cov, 0, N, kernel='potential')
    np.testing.assert_allclose(b, 2.0 * a, rtol=1e-14)


def test_row_tiled_kernel_on_128_multiple_grid():
    """Grids whose meridian count is a multiple of 128 take the row-tiled kernel (covprop.hip); others the general one."""
    N, nmin = 30, 2
    mer = np.linspace(-np.pi, np.pi, 256, endpoint=False) + np.pi / 256
    par = np.linspace(1.5, -1.5, 9)
    grid = ga.grid.RegularGrid(mer, par)
    P = (N + 1) ** 2 - nmin ** 2
    cov = inputs.spd_covariance(321, P)
    ref = orc.covariance_propagation_regular(cov, nmin, N, mer, par, orc.KernelTable('potential'))
    s = grid.covariance_propagation(cov, nmin, N, kernel='potential')
    assert relerr(s, ref) < TOL_SIGMA


@pytest.mark.parametrize('N,step,nmin', [(20, 5.0, 0), (40, 2.0, 2), (60, 3.0, 0)])
def test_symmetric_shortcut_matches_general_path(N, step, nmin):
    """the upper-triangle variant (half the MFMA work) against the general kernel and, for an exactly symmetric
    matrix, the automatic choice; a non-symmetric matrix keeps the general path under symmetric=None"""
    import torch
    grid = ga.grid.GeographicGrid(step, step)
    P = (N + 1) ** 2 - nmin ** 2
    cov = inputs.spd_covariance(21, P)
    cov = 0.5 * (cov + cov.T)
    full = grid.covariance_propagation(cov, nmin, N, kernel='ewh')
    sym = grid.covariance_propagation(cov, nmin, N, kernel='ewh', symmetric=True)
    auto = grid.covariance_propagation(cov, nmin, N, kernel='ewh', symmetric=None)
    assert relerr(sym, full) < 1e-13
    np.testing.assert_array_equal(auto, sym)
    # only the upper triangle is read
    junk = np.triu(cov) + np.tril(np.full_like(cov, 1e30), -1)
    assert relerr(grid.covariance_propagation(junk, nmin, N, kernel='ewh', symmetric=True), full) < 1e-13
    skew = cov.copy()
    skew[0, 1] *= 1.0 + 1e-9
    np.testing.assert_array_equal(grid.covariance_propagation(skew, nmin, N, kernel='ewh', symmetric=None),
                                  grid.covariance_propagation(skew, nmin, N, kernel='ewh'))
    band = grid.covariance_propagation(cov, nmin, N, kernel='ewh', parallel_range=(3, 11), symmetric=True)
    assert relerr(band, full.reshape(grid.parallels.size, -1)[3:11].ravel()) < 1e-13


@pytest.mark.parametrize('N,step,nmin', [(20, 5.0, 0), (40, 2.0, 2), (60, 3.0, 0), (33, 4.0, 5)])
def test_separable_variant_matches_direct_path_and_oracle(golden, N, step, nmin):
    """method='separable' (latitude / longitude factorisation of the synthesis matrix) against the reference-formulation
    kernel, the oracle and -- for a non-symmetric matrix -- the direct kernel again: nothing in it assumes symmetry."""
    grid = ga.grid.GeographicGrid(step, step)
    P = (N + 1) ** 2 - nmin ** 2
    cov = inputs.spd_covariance(31, P)
    direct = grid.covariance_propagation(cov, nmin, N, kernel='ewh')
    sep = grid.covariance_propagation(cov, nmin, N, kernel='ewh', method='separable')
    assert relerr(sep, direct) < 1e-12
    if N <= 40:
        ref = orc.covariance_propagation_regular(cov, nmin, N, grid.meridians, grid.parallels, orc.KernelTable('ewh', love()))
        assert relerr(sep, ref) < TOL_SIGMA
    band = grid.covariance_propagation(cov, nmin, N, kernel='ewh', parallel_range=(2, 9), method='separable')
    assert relerr(band, direct.reshape(grid.parallels.size, -1)[2:9].ravel()) < 1e-12
    # symmetric matrix: only the slot pairs s <= s' are formed (promised by the caller / detected on the device)
    assert np.array_equal(cov, cov.T)
    half = grid.covariance_propagation(cov, nmin, N, kernel='ewh', method='separable', symmetric=True)
    assert relerr(half, direct) < 1e-12
    auto = grid.covariance_propagation(cov, nmin, N, kernel='ewh', method='separable', symmetric=None)
    np.testing.assert_array_equal(auto, half)
    band = grid.covariance_propagation(cov, nmin, N, kernel='ewh', parallel_range=(2, 9), method='separable', symmetric=True)
    assert relerr(band, direct.reshape(grid.parallels.size, -1)[2:9].ravel()) < 1e-12
    skew = cov + np.triu(np.random.default_rng(3).standard_normal(cov.shape) * np.abs(cov).max() * 1e-3, 1)
    a = grid.covariance_propagation(skew, nmin, N, kernel='potential')
    b = grid.covariance_propagation(skew, nmin, N, kernel='potential', method='separable')
    np.testing.assert_array_equal(grid.covariance_propagation(skew, nmin, N, kernel='potential', method='separable', symmetric=None), b)   # not symmetric: full path
    ok = np.isfinite(a)
    np.testing.assert_array_equal(np.isfinite(b), ok)
    assert relerr(b[ok], a[ok]) < 1e-11
    with pytest.raises(ValueError):
        grid.covariance_propagation(cov, nmin, N, kernel='ewh', method='fast')


def test_separable_variant_golden(golden):
    """the reference's own outputs (g9) through the separable variant"""
    g = golden('g9_covariance')
    grid = ga.grid.GeographicGrid(2.0, 2.0)
    s = grid.covariance_propagation(inputs.spd_covariance(31, 41 * 41), 0, 40, kernel='ewh', method='separable')
    assert relerr(s, g['n40_2deg_ewh']) < TOL_SIGMA
    grid = ga.grid.GeographicGrid(5.0, 5.0)
    s = grid.covariance_propagation(inputs.spd_covariance(32, 21 * 21 - 4), 2, 20, kernel='potential', method='separable')
    assert relerr(s, g['n20_5deg_min2_potential']) < TOL_SIGMA


def test_full_size_properties_config4():
    """BASELINE config 4 at full size (d/o 180, P = 32761, Sigma 8.6 GB, 0.5 degree grid): size-independent properties.
    Sigma = c I gives sigma^2(i, j) = c sum_n (2n + 1) kn[i, n]^2 for every meridian (addition theorem of the 4 pi normalised
    harmonics); sigma scales with the square root of Sigma; the direct kernel (a band) and the separable variant (whole grid)
    agree on a dense random Sigma."""
    import torch
    N = 180
    grid = ga.grid.GeographicGrid(0.5, 0.5)
    P = (N + 1) ** 2
    colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel('ewh'), N, grid.parallels, 3.9860044150e+14, 6.3781363000e+06,
                                                   grid.semimajor_axis, grid.flattening)
    plan = ga.engine.Plan(N, colat, kn, grid.meridians)
    c = 3.0e-20
    cov = torch.zeros((P, P), dtype=torch.float64, device='cuda')
    cov.diagonal().fill_(c)
    closed = np.sqrt(c * (kn ** 2 * (2 * np.arange(N + 1) + 1)[None, :]).sum(axis=1))          # [nlat]
    sep = ga.engine.to_host(plan.covariance_propagation(cov, 0, method='separable')).reshape(360, 720)
    assert relerr(sep, np.repeat(closed[:, None], 720, axis=1)) < 1e-12
    band = ga.engine.to_host(plan.covariance_propagation(cov, 0, 100, 103)).reshape(3, 720)
    assert relerr(band, np.repeat(closed[100:103, None], 720, axis=1)) < 1e-12
    # dense random (non-symmetric) matrix with a dominant diagonal: both paths, and the scaling law
    gen = torch.Generator(device='cuda')
    gen.manual_seed(11)
    cov = torch.rand((P, P), dtype=torch.float64, device='cuda', generator=gen)
    cov.mul_(1e-22 / P)
    cov.diagonal().add_(2e-22)
    direct = plan.covariance_propagation(cov, 0, 178, 181)
    sep = plan.covariance_propagation(cov, 0, method='separable')
    assert float(((sep.reshape(360, 720)[178:181].reshape(-1) - direct).abs().max() / direct.abs().max()).item()) < 1e-12
    # the direct kernel on ALL 360 parallels (556 TFLOP, ~9 s) against the separable result, and bands against the full grid
    full = plan.covariance_propagation(cov, 0)
    assert float(((sep - full).abs().max() / full.abs().max()).item()) < 1e-12
    assert torch.equal(full.reshape(360, 720)[178:181].reshape(-1), direct)
    assert torch.equal(plan.covariance_propagation(cov, 0, 270, 360), full.reshape(360, 720)[270:360].reshape(-1))     # the 4-GPU share of the last rank
    del full
    cov.mul_(9.0)
    sep9 = plan.covariance_propagation(cov, 0, method='separable')
    assert float(((sep9 - 3.0 * sep).abs().max() / sep9.abs().max()).item()) < 1e-13


def test_filtered_covariance_and_blocks_golden(golden):
    """W Sigma W^T ahead of the propagation (SpatialFilter.filter_covariance) and the per-parallel blocks
    F Sigma F^T (RegularGrid.covariance_blocks) against outputs of the reference (tests/golden/g18_operators.npz)."""
    import torch
    g = golden('g18_operators')
    grid = ga.grid.GeographicGrid(5.0, 5.0)
    for nmax, seed in ((12, 53), (20, 54)):
        blocks = orc.ddk_blocks(inputs.orderwise_normal_blocks(seed, nmax), 5)
        flt = ga.filter.OrderWiseFilter(blocks)
        P = (nmax + 1) ** 2 - 4
        cov = inputs.spd_covariance(seed + 10, P)
        filtered = flt.filter_covariance(cov, 2, nmax)
        assert isinstance(filtered, torch.Tensor) and tuple(filtered.shape) == (P, P)
        assert torch.equal(filtered, filtered.T)                               # upper tiles mirrored: exactly symmetric
        if nmax == 12:
            assert relerr(ga.engine.to_host(filtered), g['filtered_cov_n12']) < 1e-12
        sigma = grid.covariance_propagation(filtered, 2, nmax, kernel='ewh')
        assert relerr(sigma, g['filtered_sigma_n{0}_ewh'.format(nmax)]) < 1e-11
        sigma_sym = grid.covariance_propagation(filtered, 2, nmax, kernel='ewh', symmetric=True)
        assert relerr(sigma_sym, g['filtered_sigma_n{0}_ewh'.format(nmax)]) < 1e-11
        gauss = ga.filter.Gaussian(400).filter_covariance(cov, 2, nmax)
        assert relerr(grid.covariance_propagation(gauss, 2, nmax, kernel='potential'), g['gauss_filtered_sigma_n{0}_potential'.format(nmax)]) < 1e-11
    cov = inputs.spd_covariance(33, 21 * 21)
    blocks = ga.engine.to_host(grid.covariance_blocks(cov, 0, 20, kernel='ewh'))
    assert blocks.shape == (36, 72, 72)
    for i in (0, 17, 35):
        assert relerr(blocks[i], g['block_n20_5deg_ewh_{0}'.format(i)]) < 1e-12
    # the diagonal of every block is what covariance_propagation returns; a band equals the same rows of the full result
    sigma = grid.covariance_propagation(cov, 0, 20, kernel='ewh').reshape(36, 72)
    assert relerr(np.sqrt(np.einsum('kii->ki', blocks)), sigma) < 1e-12
    band = ga.engine.to_host(grid.covariance_blocks(cov, 0, 20, kernel='ewh', parallel_range=(16, 19)))
    assert np.array_equal(band, blocks[16:19])


def test_congruence_shapes_and_errors():
    rng = np.random.default_rng(3)
    for n, k in ((1, 1), (5, 9), (130, 67), (257, 300)):
        W, S = rng.standard_normal((n, k)), rng.standard_normal((k, k))
        S = S + S.T
        out = ga.engine.to_host(ga.engine.congruence(W, S))
        assert relerr(out, W @ S @ W.T) < 1e-13 and np.array_equal(out, out.T)
    with pytest.raises(ValueError):
        ga.engine.congruence(np.zeros((3, 4)), np.zeros((5, 5)))
