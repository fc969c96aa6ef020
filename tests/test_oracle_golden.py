"""
Pins the CPU oracle (oracle/shg_oracle.py) to golden vectors produced by the imported reference
(tests/golden/make_golden.py).  Integer index maps: bit-exact.  Floating point: the same arithmetic
is restated, so agreement is at the 1e-15 level; tolerances below leave room for BLAS summation order.
"""

import numpy as np
import pytest

import inputs
from conftest import relerr
from oracle import shg_oracle as orc


def love(golden):
    return golden('g6_kernel')['love_CE'][0]


# ---------------------------------------------------------------- G1/G2
@pytest.mark.parametrize('N', [5, 60, 96, 180])
def test_legendre_functions(golden, N):
    g = golden('g1_legendre')
    colat = g['colat_{0}'.format(N)]
    P = orc.legendre_functions(N, colat)
    ref = g['pnm_{0}'.format(N)]
    assert P.shape == ref.shape
    np.testing.assert_allclose(P, ref, rtol=1e-14, atol=1e-300)
    for m in sorted({0, 1, N // 2, N}):
        ref_m = g['pnm_order_{0}_{1}'.format(N, m)]
        np.testing.assert_allclose(orc.legendre_functions_per_order(N, m, colat), ref_m, rtol=1e-14, atol=1e-300)


def test_legendre_degree_zero_and_raises(golden):
    np.testing.assert_array_equal(orc.legendre_functions(0, np.array([0.3, 1.2])), golden('g1_legendre')['pnm_0'])
    with pytest.raises(ValueError):
        orc.legendre_functions_per_order(3, 4, np.array([0.1]))


# ---------------------------------------------------------------- G3
def test_trigonometric_functions(golden):
    g = golden('g3_trig')
    for N in (20, 96):
        np.testing.assert_allclose(orc.trigonometric_functions(N, g['lon_{0}'.format(N)]), g['cs_{0}'.format(N)], rtol=0, atol=2e-16)
    np.testing.assert_allclose(orc.spherical_harmonics(12, g['ynm_colat'], g['ynm_lon']), g['ynm_12'], rtol=1e-14, atol=1e-16)


# ---------------------------------------------------------------- G4 (bit-exact)
@pytest.mark.parametrize('nmin,nmax', [(0, 5), (2, 5), (0, 60), (2, 96), (0, 180), (3, 3)])
def test_index_maps_bit_exact(golden, nmin, nmax):
    g = golden('g4_index')
    tag = '{0}_{1}'.format(nmin, nmax)
    flat = np.arange((nmax + 1) ** 2, dtype=np.int64).reshape(nmax + 1, nmax + 1)
    np.testing.assert_array_equal(orc.ravel_coefficients(flat, nmin, nmax), g['ravel_' + tag])
    np.testing.assert_array_equal(orc.degreewise_sequence(nmin, nmax), g['seq_' + tag])
    if nmax <= 60:
        vec = np.arange(g['seq_' + tag].shape[0], dtype=np.int64) + 1
        np.testing.assert_array_equal(orc.unravel_coefficients(vec, nmin, nmax), g['unravel_' + tag])
        for m in sorted({0, 1, nmax // 2, nmax}):
            np.testing.assert_array_equal(orc.vector_indices(nmin, nmax, m), g['vidx_{0}_{1}'.format(tag, m)])
            if m > 0:
                np.testing.assert_array_equal(orc.vector_indices(nmin, nmax, m, 'c'), g['vidx_{0}_{1}_c'.format(tag, m)])
                np.testing.assert_array_equal(orc.vector_indices(nmin, nmax, m, 's'), g['vidx_{0}_{1}_s'.format(tag, m)])


def test_index_maps_misc(golden):
    g = golden('g4_index')
    arr3 = np.arange(3 * 36, dtype=np.int64).reshape(3, 6, 6)
    np.testing.assert_array_equal(orc.ravel_coefficients(arr3, 1, 5), g['ravel3d_1_5'])
    np.testing.assert_array_equal(orc.ravel_coefficients(arr3[0], 0, 8), g['ravel_short_0_8'])
    np.testing.assert_array_equal(orc.unravel_coefficients(g['ravel3d_1_5'], 1, 5), g['unravel2d_1_5'])
    for n, mo in ((0, None), (4, None), (7, 3)):
        np.testing.assert_array_equal(np.vstack(orc.degree_indices(n, mo)), g['degidx_{0}_{1}'.format(n, mo)])
    for N, m in ((6, 0), (6, 2), (6, 6)):
        np.testing.assert_array_equal(np.vstack(orc.order_indices(N, m)), g['ordidx_{0}_{1}'.format(N, m)])
    x = np.random.default_rng(0).standard_normal(12 * 12 - 9)
    np.testing.assert_array_equal(orc.ravel_coefficients(orc.unravel_coefficients(x, 3, 11), 3, 11), x)
    with pytest.raises(ValueError):
        orc.ravel_coefficients(np.zeros(4))
    with pytest.raises(ValueError):
        orc.unravel_coefficients(np.zeros((2, 2, 4)))


# ---------------------------------------------------------------- G5
def test_geometry(golden):
    g = golden('g5_geometry')
    for step, tag in ((1.0, '1p0'), (0.25, '0p25')):
        mer, par, area = orc.geographic_grid(step, step)
        np.testing.assert_array_equal(mer, g['meridians_' + tag])
        np.testing.assert_array_equal(par, g['parallels_' + tag])
        np.testing.assert_allclose(orc.colatitude(par), g['colat_' + tag], rtol=0, atol=1e-15)
        np.testing.assert_allclose(orc.geocentric_radius(par), g['radius_' + tag], rtol=1e-15)
        np.testing.assert_allclose(area.sum(axis=1), g['area_rowsum_' + tag], rtol=1e-14)
    mer, par, area = orc.gauss_grid(31)
    np.testing.assert_allclose(mer, g['gauss31_meridians'], rtol=0, atol=1e-15)
    np.testing.assert_allclose(par, g['gauss31_parallels'], rtol=0, atol=1e-15)
    np.testing.assert_allclose(area.ravel(), g['gauss31_area'], rtol=1e-14)


# ---------------------------------------------------------------- G6
def test_kernel_tables(golden):
    g = golden('g6_kernel')
    r, colat = g['r'], g['colat']
    ewh = orc.KernelTable('ewh', love(golden))
    np.testing.assert_allclose(ewh.inverse_coefficients(0, 180, r, colat), g['inv_ewh'], rtol=1e-15)
    np.testing.assert_allclose(ewh.coefficients(2, 40, r, colat), g['coef_ewh'], rtol=1e-15)
    pot = orc.KernelTable('potential')
    np.testing.assert_array_equal(pot.inverse_coefficients(0, 180, r, colat), g['inv_potential'])
    for radius in (0, 200, 300, 500):
        np.testing.assert_allclose(orc.gauss_weights(radius, 200)[np.newaxis, :], g['gauss_{0}'.format(radius)], rtol=1e-15, atol=0)
    assert np.count_nonzero(orc.gauss_weights(300, 200)) == 143   # last non-zero degree 142


# ---------------------------------------------------------------- G7
def test_synthesis_c1(golden):
    g = golden('g7_synthesis')
    filtered = orc.gaussian_filter(inputs.coefficients(1000, 60), 300)
    np.testing.assert_allclose(filtered, g['c1_filtered_anm'], rtol=1e-15, atol=0)
    mer, par, _ = orc.geographic_grid(1.0, 1.0)
    grid = orc.synthesis_regular(filtered, mer, par, orc.KernelTable('ewh', love(golden)))
    assert relerr(grid, g['c1_grid']) < 1e-14


def test_synthesis_c2_unit(golden):
    g = golden('g7_synthesis')
    mer, par, _ = orc.geographic_grid(0.25, 0.25)
    grid = orc.synthesis_regular(inputs.coefficients(1000, 96), mer, par, orc.KernelTable('ewh', love(golden)))
    assert relerr(grid[::9, ::11], g['c2_sample_0']) < 1e-14
    assert relerr(grid.sum(axis=1), g['c2_rowsum_0']) < 1e-13
    assert abs(np.abs(grid).max() - g['c2_maxabs_0'][0]) < 1e-14 * g['c2_maxabs_0'][0]


def test_synthesis_variants(golden):
    g = golden('g7_synthesis')
    mer, par, _ = orc.gauss_grid(61)
    grid = orc.synthesis_regular(inputs.coefficients(7, 60), mer, par, orc.KernelTable('potential'))
    assert relerr(grid, g['gauss61_potential']) < 1e-14
    mer, par, _ = orc.geographic_grid(5.0, 5.0)
    for name in ('potential', 'ewh'):
        grid = orc.synthesis_regular(inputs.coefficients(8, 30), mer, par, orc.KernelTable(name, love(golden)))
        assert relerr(grid, g['n30_5deg_' + name]) < 1e-14
    grid = orc.synthesis_regular(inputs.coefficients(9, 30), mer, par, orc.KernelTable('ewh', love(golden)), GM=3.986004418e14, R=6378137.0)
    assert relerr(grid, g['n30_5deg_gmr']) < 1e-14
    grid = orc.synthesis_regular(inputs.coefficients(13, 25), g['asym_meridians'], g['asym_parallels'], orc.KernelTable('potential'))
    assert relerr(grid, g['asym_potential']) < 1e-14
    lon, lat = inputs.scattered_points(11, 1000)
    vals = orc.synthesis_points(inputs.coefficients(12, 40), lon, lat, orc.KernelTable('ewh', love(golden)))
    assert relerr(vals, g['points_ewh']) < 1e-14


# ---------------------------------------------------------------- G8
def test_analysis(golden):
    g = golden('g8_analysis')
    mer, par, area = orc.gauss_grid(31)
    ewh = orc.KernelTable('ewh', love(golden))
    values = orc.synthesis_regular(inputs.coefficients(22, 30), mer, par, ewh)
    assert relerr(values, g['gauss31_values']) < 1e-14
    anm = orc.analysis_regular(g['gauss31_values'].ravel(), area.ravel(), 2, 30, mer, par, ewh)
    assert relerr(anm, g['gauss31_anm_ewh_2_30']) < 1e-10        # LSQ solve: conditioning amplifies BLAS noise
    mer, par, area = orc.geographic_grid(5.0, 5.0)
    vals = np.random.default_rng(23).standard_normal(par.size * mer.size)
    anm = orc.analysis_regular(vals, area.ravel(), 0, 20, mer, par, orc.KernelTable('potential'))
    assert relerr(anm, g['n20_5deg_random_anm']) < 1e-11
    A = orc.synthesis_matrix(1, 8, mer, par, ewh)
    assert relerr(A, g['n8_5deg_synthesis_matrix']) < 1e-14


@pytest.mark.slow
def test_analysis_n60(golden):
    g = golden('g8_analysis')
    mer, par, area = orc.geographic_grid(1.0, 1.0)
    pot = orc.KernelTable('potential')
    values = orc.synthesis_regular(inputs.coefficients(21, 60), mer, par, pot)
    anm = orc.analysis_regular(values.ravel(), area.ravel(), 0, 60, mer, par, pot)
    assert relerr(anm, g['n60_1deg_anm']) < 1e-10


# ---------------------------------------------------------------- G9
def test_covariance_propagation(golden):
    g = golden('g9_covariance')
    ewh = orc.KernelTable('ewh', love(golden))
    mer, par, _ = orc.geographic_grid(2.0, 2.0)
    s = orc.covariance_propagation_regular(inputs.spd_covariance(31, 41 * 41), 0, 40, mer, par, ewh)
    assert relerr(s, g['n40_2deg_ewh']) < 1e-12
    mer, par, _ = orc.geographic_grid(5.0, 5.0)
    s = orc.covariance_propagation_regular(inputs.spd_covariance(32, 21 * 21 - 4), 2, 20, mer, par, orc.KernelTable('potential'))
    assert relerr(s, g['n20_5deg_min2_potential']) < 1e-12
    cov = inputs.spd_covariance(33, 21 * 21)
    s = orc.covariance_propagation_regular(cov, 0, 20, mer, par, ewh)
    assert relerr(s, g['n20_5deg_ewh']) < 1e-12
    assert relerr(s, g['n20_5deg_ewh_einsum']) < 1e-12
    band = orc.covariance_propagation_regular(cov, 0, 20, mer, par, ewh, parallel_range=(3, 7))
    np.testing.assert_allclose(band, s[3 * mer.size:7 * mer.size], rtol=1e-13)
    lon, lat = inputs.scattered_points(34, 300)
    s = orc.covariance_propagation_points(cov, 0, 20, lon, lat, ewh)
    assert relerr(s, g['points_n20_ewh']) < 1e-12


# ---------------------------------------------------------------- G18
def test_full_matrix_operators(golden):
    """Window matrix, filtered covariance ahead of the propagation and per-parallel covariance blocks."""
    g = golden('g18_operators')
    ewh, pot = orc.KernelTable('ewh', love(golden)), orc.KernelTable('potential')
    mer, par, area = orc.geographic_grid(5.0, 5.0)
    window = np.random.default_rng(51).uniform(0.0, 1.0, area.size)
    assert relerr(orc.window_matrix_regular(window, area, 1, 8, mer, par, pot), g['window_5deg_1_8_potential']) < 1e-12
    gm, gp, ga_ = orc.gauss_grid(13)
    window = (np.random.default_rng(52).uniform(0.0, 1.0, ga_.size) > 0.4).astype(float)
    assert relerr(orc.window_matrix_regular(window, ga_, 0, 10, gm, gp, ewh), g['window_gauss13_0_10_ewh']) < 1e-11
    for nmax, seed in ((12, 53), (20, 54)):
        blocks = orc.ddk_blocks(inputs.orderwise_normal_blocks(seed, nmax), 5)
        W = orc.orderwise_matrix(blocks, 2, nmax)
        cov = inputs.spd_covariance(seed + 10, W.shape[0])
        filtered = W @ cov @ W.T
        if nmax == 12:
            assert relerr(filtered, g['filtered_cov_n12']) < 1e-12
        assert relerr(orc.covariance_propagation_regular(filtered, 2, nmax, mer, par, ewh), g['filtered_sigma_n{0}_ewh'.format(nmax)]) < 1e-11
        G = orc.gaussian_matrix(400, 2, nmax)
        assert relerr(orc.covariance_propagation_regular(G @ cov @ G.T, 2, nmax, mer, par, pot), g['gauss_filtered_sigma_n{0}_potential'.format(nmax)]) < 1e-12
    blocks = orc.covariance_blocks_regular(inputs.spd_covariance(33, 21 * 21), 0, 20, mer, par, ewh, (0, 17, 35))
    for k, i in enumerate((0, 17, 35)):
        assert relerr(blocks[k], g['block_n20_5deg_ewh_{0}'.format(i)]) < 1e-12


# ---------------------------------------------------------------- G10
def test_filters(golden):
    g = golden('g10_filter')
    np.testing.assert_allclose(orc.gaussian_filter(inputs.coefficients(41, 60), 300), g['gaussian_300_n60'], rtol=1e-15, atol=0)
    np.testing.assert_allclose(np.diag(orc.gaussian_matrix(500, 2, 12)), g['gaussian_500_matrix_2_12_diag'], rtol=1e-15)
    for nmax, ngf in ((20, 20), (120, 120), (120, 96)):
        blocks = inputs.orderwise_random_blocks(42, nmax)
        out = orc.orderwise_filter(inputs.coefficients(43, ngf), blocks)
        assert relerr(out, g['orderwise_{0}_{1}'.format(nmax, ngf)]) < 1e-14
    blocks = inputs.orderwise_random_blocks(42, 20)
    np.testing.assert_array_equal(orc.orderwise_matrix(blocks, 0, 20), g['orderwise_20_matrix_0_20'])
    np.testing.assert_array_equal(orc.orderwise_matrix(blocks, 2, 14), g['orderwise_20_matrix_2_14'])
    with pytest.raises(ValueError):
        orc.orderwise_filter(inputs.coefficients(1, 21), blocks)
    for level in (5, 3):
        normals = inputs.orderwise_normal_blocks(44, 20)
        ddk = orc.ddk_blocks(normals, level)
        out = orc.orderwise_filter(inputs.coefficients(45, 20), ddk)
        assert relerr(out, g['ddk{0}_n20'.format(level)]) < 1e-12
        assert relerr(orc.orderwise_matrix(ddk, 2, 20), g['ddk{0}_n20_matrix'.format(level)]) < 1e-12
        gen = orc.ddk_blocks(normals, level, generic=True)
        assert relerr(orc.orderwise_filter(inputs.coefficients(45, 20), gen), g['ddkgeneric{0}_n20'.format(level)]) < 1e-12
    W = np.random.default_rng(46).standard_normal((21 * 21 - 4, 21 * 21 - 4)) / 21
    assert relerr(orc.general_matrix_filter(inputs.coefficients(47, 20), W, 2, 20), g['general_2_20_n20']) < 1e-14
    out14 = orc.general_matrix_filter(inputs.coefficients(47, 14), W, 2, 20)
    assert out14.shape == g['general_2_20_n14'].shape
    assert relerr(out14, g['general_2_20_n14']) < 1e-14


# ---------------------------------------------------------------- G12
def test_vdk_and_filter_kernels(golden):
    g = golden('g12_filter_kernel')
    nmin, nmax = 2, 12
    P = (nmax + 1) ** 2 - nmin ** 2
    normals = inputs.spd_covariance(70, P, scale=1e20)
    W = orc.vdk_matrix(normals, nmin, nmax, 1e18, 2.0)
    assert relerr(W, g['vdk_matrix']) < 1e-12
    assert relerr(orc.general_matrix_filter(inputs.coefficients(71, 12), W, nmin, nmax), g['vdk_filtered_n12']) < 1e-12
    src_lon, src_lat = np.deg2rad(13.0), np.deg2rad(47.5)
    ev_lon = np.deg2rad(np.linspace(-20.0, 50.0, 9))
    ev_lat = np.deg2rad(np.linspace(30.0, 65.0, 6))
    pts_lon, pts_lat = np.meshgrid(ev_lon, ev_lat)
    for name in ('potential', 'ewh'):
        K2 = orc.filter_kernel_matrix(g['vdk_matrix'], nmin, nmax, orc.KernelTable(name, love(golden)))
        pts = orc.anisotropic_kernel_points(K2, nmin, nmax, src_lon, src_lat, pts_lon.ravel(), pts_lat.ravel())
        assert relerr(pts, g['filterkernel_{0}_points'.format(name)]) < 1e-13
        grid = orc.anisotropic_kernel_grid(K2, nmin, nmax, src_lon, src_lat, ev_lon, ev_lat)
        assert relerr(grid.ravel(), g['filterkernel_{0}_points'.format(name)]) < 1e-13
    K = np.random.default_rng(72).standard_normal((P, P)) / nmax
    assert relerr(orc.anisotropic_kernel_grid(K, nmin, nmax, src_lon, src_lat, ev_lon, ev_lat), g['anisotropic_grid']) < 1e-13
    assert relerr(orc.anisotropic_kernel_points(K, nmin, nmax, src_lon, src_lat, ev_lon, ev_lat[0:1].repeat(ev_lon.size)), g['anisotropic_points']) < 1e-13
    blocks = inputs.orderwise_random_blocks(73, nmax)
    Ko = orc.filter_kernel_matrix(orc.orderwise_matrix(blocks, nmin, nmax), nmin, nmax, orc.KernelTable('potential'))
    assert relerr(orc.anisotropic_kernel_points(Ko, nmin, nmax, src_lon, src_lat, pts_lon.ravel(), pts_lat.ravel()), g['filterkernel_orderwise_points']) < 1e-13
