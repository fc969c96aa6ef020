"""
Host-side mirror of the reference interface (kernels, grids, index maps, containers) checked against the
golden vectors of the reference -- no GPU needed, no oracle involved.
"""

import datetime

import numpy as np
import pytest

import grates_amd as ga
import inputs
from conftest import relerr


# ---------------------------------------------------------------- index maps (bit-exact)
@pytest.mark.parametrize('nmin,nmax', [(0, 5), (2, 5), (0, 60), (2, 96), (0, 180), (3, 3)])
def test_index_maps_bit_exact(golden, nmin, nmax):
    g = golden('g4_index')
    tag = '{0}_{1}'.format(nmin, nmax)
    flat = np.arange((nmax + 1) ** 2, dtype=np.int64).reshape(nmax + 1, nmax + 1)
    np.testing.assert_array_equal(ga.utilities.ravel_coefficients(flat, nmin, nmax), g['ravel_' + tag])
    seq = ga.gravityfield.CoefficientSequenceDegreeWise(nmin, nmax)
    np.testing.assert_array_equal(seq.as_array(), g['seq_' + tag])
    assert seq.coefficient_count == (nmax + 1) ** 2 - nmin ** 2
    if nmax <= 60:
        vec = np.arange(seq.coefficient_count, dtype=np.int64) + 1
        np.testing.assert_array_equal(ga.utilities.unravel_coefficients(vec, nmin, nmax), g['unravel_' + tag])
        for m in sorted({0, 1, nmax // 2, nmax}):
            np.testing.assert_array_equal(seq.vector_indices(order=m), g['vidx_{0}_{1}'.format(tag, m)])
            if m > 0:
                np.testing.assert_array_equal(seq.vector_indices(order=m, cs='c'), g['vidx_{0}_{1}_c'.format(tag, m)])
                np.testing.assert_array_equal(seq.vector_indices(order=m, cs='s'), g['vidx_{0}_{1}_s'.format(tag, m)])


def test_index_maps_misc(golden):
    g = golden('g4_index')
    arr3 = np.arange(3 * 36, dtype=np.int64).reshape(3, 6, 6)
    np.testing.assert_array_equal(ga.utilities.ravel_coefficients(arr3, 1, 5), g['ravel3d_1_5'])
    np.testing.assert_array_equal(ga.utilities.ravel_coefficients(arr3[0], 0, 8), g['ravel_short_0_8'])
    np.testing.assert_array_equal(ga.utilities.unravel_coefficients(g['ravel3d_1_5'], 1, 5), g['unravel2d_1_5'])
    for n, mo in ((0, None), (4, None), (7, 3)):
        np.testing.assert_array_equal(np.vstack(ga.gravityfield.degree_indices(n, mo)), g['degidx_{0}_{1}'.format(n, mo)])
    for N, m in ((6, 0), (6, 2), (6, 6)):
        np.testing.assert_array_equal(np.vstack(ga.gravityfield.order_indices(N, m)), g['ordidx_{0}_{1}'.format(N, m)])
    # reference test: ravel -> unravel identity, raises (grates/testing/utilities.py:173-227)
    x = np.random.default_rng(0).standard_normal((4, 9, 9))
    for k in range(4):
        lower_and_upper = ga.utilities.unravel_coefficients(ga.utilities.ravel_coefficients(x[k]))
        np.testing.assert_array_equal(lower_and_upper, x[k])
    with pytest.raises(ValueError):
        ga.utilities.ravel_coefficients(np.zeros(4))
    with pytest.raises(ValueError):
        ga.utilities.unravel_coefficients(np.zeros((2, 2, 4)))
    with pytest.raises(ValueError):
        ga.gravityfield.CoefficientSequenceDegreeWise(0, 3).vector_indices(cs='x')


def test_reorder_indices():
    src = ga.gravityfield.CoefficientSequenceDegreeWise(2, 6)
    for tgt in (ga.gravityfield.CoefficientSequenceDegreeWise(0, 4), ga.gravityfield.CoefficientSequenceOrderWise(1, 8),
                ga.gravityfield.CoefficientSequenceOrderWiseAlternating(0, 5), ga.gravityfield.CoefficientSequenceFlatArray(5)):
        i_src, i_tgt = ga.gravityfield.CoefficientSequence.reorder_indices(src, tgt)
        assert i_src.size == i_tgt.size > 0
        np.testing.assert_array_equal(src.as_array()[i_src], tgt.as_array()[i_tgt])
        assert np.all(np.diff(i_tgt) > 0)
    flat = ga.gravityfield.CoefficientSequenceFlatArray(3).as_array()
    assert flat.tolist()[0:5] == [[0, 0, 0], [1, 1, 1], [1, 2, 1], [1, 3, 1], [0, 1, 0]]


# ---------------------------------------------------------------- geometry / grids
def test_geometry_and_grids(golden):
    g = golden('g5_geometry')
    for step, tag in ((1.0, '1p0'), (0.25, '0p25')):
        grid = ga.grid.GeographicGrid(step, step)
        np.testing.assert_array_equal(grid.meridians, g['meridians_' + tag])
        np.testing.assert_array_equal(grid.parallels, g['parallels_' + tag])
        np.testing.assert_allclose(ga.utilities.colatitude(grid.parallels), g['colat_' + tag], rtol=0, atol=1e-15)
        np.testing.assert_allclose(ga.utilities.geocentric_radius(grid.parallels), g['radius_' + tag], rtol=1e-15)
        np.testing.assert_allclose(grid.area.reshape(grid.parallels.size, -1).sum(axis=1), g['area_rowsum_' + tag], rtol=1e-14)
    grid = ga.grid.GeographicGrid(2.0, 5.0)
    np.testing.assert_array_equal(grid.longitude, g['geo_2_5_lon'])
    np.testing.assert_array_equal(grid.latitude, g['geo_2_5_lat'])
    np.testing.assert_allclose(grid.area, g['geo_2_5_area'], rtol=1e-15)
    assert grid.point_count == grid.size == 180 * 36
    gg = ga.grid.GaussGrid(31)
    np.testing.assert_allclose(gg.meridians, g['gauss31_meridians'], rtol=0, atol=1e-15)
    np.testing.assert_allclose(gg.parallels, g['gauss31_parallels'], rtol=0, atol=1e-15)
    np.testing.assert_allclose(gg.area, g['gauss31_area'], rtol=1e-14)
    rg = ga.grid.RegularGrid(np.linspace(-3.0, 3.0, 7), np.linspace(1.4, -1.4, 5))
    np.testing.assert_allclose(rg.area, g['regular_area'], rtol=1e-15)


def test_grid_value_contract():
    grid = ga.grid.GeographicGrid(30, 30)
    assert grid.values is None
    with pytest.raises(ValueError):
        grid.values = np.zeros((6, 12))
    with pytest.raises(ValueError):
        grid.values = np.zeros(5)
    with pytest.raises(ValueError):
        grid.values = [0.0] * 72
    grid.values = np.arange(72, dtype=float)
    assert grid.value_array.shape == (6, 12)
    grid.epoch = datetime.datetime(2010, 1, 1)
    other = grid.copy()
    assert type(other) is ga.grid.GeographicGrid and other.epoch == grid.epoch
    other.value_array[0, 0] = 99.0
    assert grid.value_array[0, 0] == 0.0
    assert grid.is_compatible(other) and not grid.is_compatible(ga.grid.GeographicGrid(30, 15))
    np.testing.assert_allclose(grid.mean(), np.sum(grid.area * grid.values) / np.sum(grid.area))
    mask = grid.values > 20
    np.testing.assert_allclose(grid.rms(mask), np.sqrt(np.sum(grid.area[mask] * grid.values[mask] ** 2) / np.sum(grid.area[mask])))
    assert grid.std() > 0
    grid.values = None
    with pytest.raises(ValueError):
        grid.to_potential_coefficients(0, 2)
    irr = ga.grid.IrregularGrid(*inputs.scattered_points(1, 10))
    assert irr.point_count == 10 and irr.values is None and not hasattr(irr, 'parallels')
    with pytest.raises(ValueError):
        irr.values = np.zeros(3)
    lon, lat = np.meshgrid(np.linspace(-3, 3, 5), np.linspace(1, -1, 4))
    reg = ga.grid.IrregularGrid(lon.ravel(), lat.ravel()).to_regular()
    assert reg.parallels.size == 4 and reg.meridians.size == 5 and reg.parallels[0] > reg.parallels[-1]


# ---------------------------------------------------------------- kernels
def test_kernel_tables(golden):
    g = golden('g6_kernel')
    r, colat = g['r'], g['colat']
    for name in ('ewh', 'potential', 'geoid', 'obp', 'surface_density', 'anomaly', 'uplift'):
        ker = ga.kernel.get_kernel(name)
        np.testing.assert_allclose(ker.inverse_coefficients(0, 180, r, colat), g['inv_' + name], rtol=2e-13, err_msg=name)
        np.testing.assert_allclose(ker.coefficients(2, 40, r, colat), g['coef_' + name], rtol=2e-13, err_msg=name)
    ker = ga.kernel.get_kernel('deformation')
    np.testing.assert_allclose(ker.coefficients(2, 40, r, colat), g['coef_deformation'], rtol=2e-13)
    ewh = ga.kernel.get_kernel('EWH')
    np.testing.assert_allclose(ewh.coefficients(0, 10), g['ewh_scalar'], rtol=1e-15)
    np.testing.assert_allclose(ewh.coefficient(7, r, colat), g['ewh_coefficient_7'], rtol=1e-15)
    np.testing.assert_array_equal(ewh.inverse_coefficient(0, r, colat), g['ewh_inverse_coefficient_0'])
    np.testing.assert_allclose(ewh.coefficient_array(2, 6), g['ewh_coef_array_2_6'], rtol=1e-15)
    np.testing.assert_allclose(ewh.inverse_coefficient_array(2, 6), g['ewh_inv_array_2_6'], rtol=1e-15)
    for frame in ('CE', 'CM', 'CF'):
        k, h, l = ga.data.load_love_numbers(frame=frame)
        np.testing.assert_array_equal(np.vstack((k[0:257], h[0:257], l[0:257])), g['love_' + frame])
    with pytest.raises(ValueError):
        ga.kernel.get_kernel('nope')
    with pytest.raises(ValueError):
        ga.data.load_love_numbers(frame='xx')


def test_kernel_broadcast_contract():
    # shapes / raises pinned by the reference's own tests (grates/testing/kernel.py:27-73)
    for name in ('ewh', 'obp', 'potential', 'geoid', 'surface_density', 'anomaly', 'uplift'):
        ker = ga.kernel.get_kernel(name)
        r, colat = np.full(5, 6378136.3), np.linspace(0.1, 3.0, 5)
        if name in ('ewh', 'potential', 'surface_density', 'anomaly'):
            assert ker.coefficients(0, 10).shape == (1, 11)
        assert ker.coefficients(2, 10, r, colat).shape == (5, 9)
        assert ker.coefficients(2, 10, 6378136.3, colat).shape == (5, 9)
        assert ker.coefficients(2, 10, r, 0.5).shape == (5, 9)
        assert ker.inverse_coefficients(2, 10, r, colat).shape == (5, 9)
        assert ker.coefficient(3, r, colat).shape == (5,)
        with pytest.raises(ValueError):
            ker.coefficients(0, 10, r, colat[0:3])
        with pytest.raises(ValueError):
            ker.coefficients(0, 10, [1.0, 2.0], colat)


def test_gauss_kernel(golden):
    g = golden('g6_kernel')
    # reference known answers (grates/testing/kernel.py:9-24)
    np.testing.assert_array_equal(ga.kernel.Gauss(0).coefficients(0, 50), np.ones((1, 51)))
    with pytest.raises(ValueError):
        ga.kernel.Gauss(-1)
    for radius in (0, 200, 300, 500):
        np.testing.assert_allclose(ga.kernel.Gauss(radius).coefficients(0, 200), g['gauss_{0}'.format(radius)], rtol=1e-15, atol=0)
    np.testing.assert_allclose(ga.kernel.Gauss(20).coefficients(1000, 1100), g['gauss_20_ext'], rtol=1e-13, atol=0)
    kn = ga.kernel.Gauss(300)
    first = kn.coefficients(0, 1000)
    np.testing.assert_array_almost_equal(kn.coefficients(0, 1100)[:, 0:1001], first, decimal=14)
    np.testing.assert_allclose(kn.evaluate(0, 100, g['gauss_300_psi']), g['gauss_300_eval'], rtol=1e-13)


def test_reference_field_known_answers(golden):
    g = golden('g6_kernel')
    GRS80, WGS84 = ga.gravityfield.GRS80, ga.gravityfield.WGS84
    # known-answer tests of the reference (grates/testing/gravityfield.py:49-59, 80-86)
    np.testing.assert_allclose(GRS80.normal_gravity(6378137.0, np.pi / 2)[0], 9.7803267715, rtol=1e-11)
    np.testing.assert_allclose(GRS80.normal_gravity(6378137.0 * (1 - GRS80.flattening), 0.0)[0], 9.8321863685, rtol=1e-9)
    back = ga.gravityfield.ReferenceField(GM=3986005e8, omega=7292115.0e-11, a=6378137.0, f=GRS80.flattening)
    np.testing.assert_allclose(back.J2, 108263e-8, rtol=1e-14)
    with pytest.raises(ValueError):
        ga.gravityfield.ReferenceField(GM=1.0, omega=1.0, a=1.0)
    np.testing.assert_allclose(GRS80.flattening, g['grs80_flattening'][0], rtol=1e-15)
    np.testing.assert_allclose(GRS80.anm[:, 0], g['grs80_anm_col0'], rtol=1e-14, atol=1e-30)
    np.testing.assert_allclose(WGS84.J2, g['wgs84_J2'][0], rtol=1e-15)
    np.testing.assert_allclose(GRS80.normal_gravity(g['r'], g['colat']), g['normal_gravity'], rtol=1e-13)
    np.testing.assert_allclose(np.array([GRS80.normal_gravity(6378137.0, np.pi / 2)[0], GRS80.normal_gravity(6378137.0 * (1 - GRS80.flattening), 0.0)[0]]),
                               g['normal_gravity_eq_pole'], rtol=1e-13)


# ---------------------------------------------------------------- containers
def make_pc(anm, **kw):
    gf = ga.gravityfield.PotentialCoefficients(**kw)
    gf.anm = anm.copy()
    return gf


def test_potential_coefficients_arithmetic(golden):
    g = golden('g10_filter')
    a = make_pc(inputs.coefficients(60, 8), GM=3.986004418e14, R=6378137.0)
    b = make_pc(inputs.coefficients(61, 12))
    np.testing.assert_allclose((a + b).anm, g['pc_add'], rtol=1e-15)
    np.testing.assert_allclose((b - a).anm, g['pc_sub'], rtol=1e-15)
    np.testing.assert_allclose((a * 2.5).anm, g['pc_mul'], rtol=1e-15)
    np.testing.assert_array_equal(b.slice(min_degree=2, max_degree=10, min_order=1, max_order=6, step_degree=2).anm, g['pc_slice'])
    np.testing.assert_array_equal(b.values, g['pc_values'])
    np.testing.assert_allclose(b.degree_amplitudes(kernel='ewh')[1], g['pc_degree_amplitudes'], rtol=1e-14)
    assert (a + b).GM == a.GM and (b + a).GM == b.GM and (a / 2).anm[3, 1] == a.anm[3, 1] * 0.5
    for bad in ('x', None, np.ones(3)):
        with pytest.raises(TypeError):
            a + bad
        with pytest.raises(TypeError):
            a * bad
    c = b.copy()
    c.anm[2, 0] = 7
    assert b.anm[2, 0] != 7 and c.max_degree == 12
    c.append('s', 14, 3, 1.5)
    assert c.max_degree == 14 and c.anm[2, 14] == 1.5
    c.append('c', 2, 1, -1.0)
    assert c.anm[2, 1] == -1.0
    c.truncate(5)
    assert c.anm.shape == (6, 6)
    v = b.values
    d = ga.gravityfield.PotentialCoefficients()
    d.values = v
    np.testing.assert_array_equal(d.anm, b.anm)
    with pytest.raises(ValueError):
        d.values = np.zeros((2, 2))
    with pytest.raises(ValueError):
        d.values = [1.0]


def test_time_series(golden):
    g = golden('g10_filter')
    series = []
    for e in range(4):
        gf = make_pc(inputs.coefficients(50 + e, 6))
        gf.epoch = datetime.datetime(2010, 1 + e, 15)
        series.append(gf)
    ts = ga.gravityfield.TimeSeries(series[::-1])
    np.testing.assert_array_equal(ts.to_array(), g['timeseries_array'])
    assert ts.epochs() == sorted(ts.epochs()) and len(ts) == 4
    batch = ts.to_coefficient_batch()
    assert batch.shape == (4, 7, 7)
    np.testing.assert_array_equal(batch[2], series[2].anm)
    mid = ts.interpolate_to(datetime.datetime(2010, 1, 30, 12))
    np.testing.assert_allclose(mid.anm, 0.5 * (series[0].anm + series[1].anm), rtol=1e-12)
    with pytest.raises(ValueError):
        ts.interpolate_to(datetime.datetime(2011, 1, 1))
    with pytest.raises(ValueError):
        ga.gravityfield.TimeSeries([ga.gravityfield.PotentialCoefficients()])
    np.testing.assert_allclose((ts - ts * 0.5).to_array(), 0.5 * ts.to_array(), rtol=1e-15)


# ---------------------------------------------------------------- filters: matrices and argument checks (host side)
def test_filter_matrices_and_errors(golden):
    g = golden('g10_filter')
    np.testing.assert_allclose(np.diag(ga.filter.Gaussian(500).matrix(2, 12)), g['gaussian_500_matrix_2_12_diag'], rtol=1e-15)
    blocks = inputs.orderwise_random_blocks(42, 20)
    flt = ga.filter.OrderWiseFilter(blocks)
    np.testing.assert_array_equal(flt.matrix(0, 20), g['orderwise_20_matrix_0_20'])
    np.testing.assert_array_equal(flt.matrix(2, 14), g['orderwise_20_matrix_2_14'])
    with pytest.raises(TypeError):
        flt.filter(np.zeros((3, 3)))
    with pytest.raises(TypeError):
        ga.filter.Gaussian(300).filter('x')
    with pytest.raises(ValueError):
        ga.filter.GeneralMatrix(np.zeros((4, 5)), 0, 1)
    with pytest.raises(ValueError):
        ga.filter.GeneralMatrix(np.zeros((5, 5)), 0, 1)
    W = np.random.default_rng(46).standard_normal((21 * 21 - 4, 21 * 21 - 4)) / 21
    gm = ga.filter.GeneralMatrix(W, 2, 20)
    np.testing.assert_array_equal(gm.matrix(2, 20), W)
    np.testing.assert_array_equal(gm.matrix(3, 18), g['general_matrix_3_18'])
    ga.filter.DDKGeneric._blocked_normals = staticmethod(lambda: inputs.orderwise_normal_blocks(44, 20))
    try:
        with pytest.raises(ValueError):
            ga.filter.DDK(9)
        with pytest.raises(ValueError):
            ga.filter.DDKGeneric(0)
        assert ga.filter.DDKGeneric.normal_equation_matrix().shape == (21 * 21 - 4, 21 * 21 - 4)
    finally:
        ga.filter.DDKGeneric._blocked_normals = staticmethod(lambda: ga.data.ddk_normal_blocks())
    with pytest.raises(FileNotFoundError):
        ga.filter.DDK(5)


def test_file_feeders(golden, tmp_path):
    """GFC / GSM loaders against what the reference parsed from the same synthetic files
    (tests/golden/g13_io.npz; the files are regenerated from their seeds).  Text -> float is bit-exact."""
    import gzip
    import io as _io
    g = golden('g13_io')
    for tag, seed, nmax, header in (('a', 80, 12, True), ('b', 81, 7, False)):
        path = tmp_path / 'model_{0}.gfc'.format(tag)
        path.write_bytes(inputs.gfc_file_text(seed, nmax, header))
        gf = ga.io.loadgfc(str(path))
        np.testing.assert_array_equal(gf.anm, g['gfc_{0}_anm'.format(tag)])
        np.testing.assert_array_equal(np.array([gf.GM, gf.R]), g['gfc_{0}_GM_R'.format(tag)])
        assert gf.epoch is None
    np.testing.assert_array_equal(ga.io.loadgfc(tmp_path / 'model_a.gfc', max_degree=5).anm, g['gfc_a_truncated_anm'])
    # compressed file, open binary stream, open text stream
    with gzip.open(tmp_path / 'model_a.gfc.gz', 'wb') as f:
        f.write(inputs.gfc_file_text(80, 12))
    np.testing.assert_array_equal(ga.io.loadgfc(str(tmp_path / 'model_a.gfc.gz')).anm, g['gfc_a_anm'])
    np.testing.assert_array_equal(ga.io.loadgfc(_io.BytesIO(inputs.gfc_file_text(80, 12))).anm, g['gfc_a_anm'])
    np.testing.assert_array_equal(ga.io.loadgfc(_io.StringIO(inputs.gfc_file_text(80, 12).decode())).anm, g['gfc_a_anm'])
    with pytest.raises(ValueError):
        ga.io.loadgfc(42)

    path = tmp_path / 'GSM-2_2010060-2010090.txt'
    path.write_bytes(inputs.gsm_file_text(82, 10))
    gf = ga.io.loadgsm(str(path))
    np.testing.assert_array_equal(gf.anm, g['gsm_anm'])
    np.testing.assert_array_equal(np.array([gf.GM, gf.R]), g['gsm_GM_R'])
    e = gf.epoch
    np.testing.assert_array_equal(np.array([e.year, e.month, e.day, e.hour, e.minute, e.second]), g['gsm_epoch'])

    # a list of monthly files -> TimeSeries sorted by epoch
    names = []
    for month in (5, 3, 4):
        name = tmp_path / 'GSM-2_2010{0:02d}.txt'.format(month)
        name.write_bytes(inputs.gsm_file_text(90 + month, 10, start='2010-{0:02d}-01T00:00:00.00'.format(month)))
        names.append(str(name))
    ts = ga.io.load_time_series(names, max_degree=8)
    assert [t.month for t in ts.epochs()] == [3, 4, 5]
    assert ts.to_array().shape == (3, 81)
    np.testing.assert_array_equal(ts[0].anm, ga.io.loadgsm(names[1]).anm[0:9, 0:9])
    series = ga.io.load_time_series([str(tmp_path / 'model_a.gfc'), str(tmp_path / 'model_b.gfc')], loader=ga.io.loadgfc,
                                    epochs=[datetime.datetime(2011, 2, 1), datetime.datetime(2011, 1, 1)])
    assert series[0].max_degree == 7 and series[1].max_degree == 12
    with pytest.raises(ValueError):
        ga.io.load_time_series([str(tmp_path / 'model_a.gfc')], loader=ga.io.loadgfc)


def test_sinex_normal_equations_reader(golden, tmp_path):
    """SINEX normal equations (grates/io.py:725-760) against what the reference read from the same
    synthetic files (tests/golden/g14_sinex.npz): matrix, right-hand side, l'Pl and the observation count, bit-exact."""
    import gzip
    g = golden('g14_sinex')
    for tag, seed, nmin, nmax, lower in (('u', 90, 2, 8, False), ('l', 91, 0, 5, True)):
        path = tmp_path / 'normals_{0}.snx'.format(tag)
        path.write_bytes(inputs.sinex_file_text(seed, nmin, nmax, lower))
        N, n, lPl, obs_count = ga.io.loadsinexnormals(str(path))
        np.testing.assert_array_equal(N, g['sinex_{0}_N'.format(tag)])
        np.testing.assert_array_equal(n, g['sinex_{0}_n'.format(tag)])
        np.testing.assert_array_equal(lPl, g['sinex_{0}_lPl'.format(tag)])
        assert obs_count == int(g['sinex_{0}_obs_count'.format(tag)]) and isinstance(obs_count, int)
        np.testing.assert_array_equal(N, N.T)
    # block list: the right-hand side records carry the coefficient numbering (degree-wise here)
    blocks = {b.block_type: b for b in ga.io.loadsinex(str(tmp_path / 'normals_u.snx'))}
    assert set(blocks) == {b'SOLUTION/STATISTICS', b'SOLUTION/NORMAL_EQUATION_VECTOR', b'SOLUTION/NORMAL_EQUATION_MATRIX'}
    rhs = blocks[b'SOLUTION/NORMAL_EQUATION_VECTOR']
    assert rhs.sigmax is None and rhs.parameter_count() == 77
    assert (rhs.basis[0], rhs.degree[0], rhs.order[0]) == ('CN', 2, 0) and (rhs.basis[-1], rhs.degree[-1], rhs.order[-1]) == ('SN', 8, 8)
    assert blocks[b'SOLUTION/STATISTICS'].parameters == 77 and blocks[b'SOLUTION/STATISTICS'].degrees_of_freedom == 12345 - 77
    with gzip.open(tmp_path / 'normals_u.snx.gz', 'wb') as f:
        f.write(inputs.sinex_file_text(90, 2, 8))
    np.testing.assert_array_equal(ga.io.loadsinexnormals(str(tmp_path / 'normals_u.snx.gz'))[0], g['sinex_u_N'])
    # a file without the statistics block is not a normal-equation file
    text = inputs.sinex_file_text(90, 2, 8).decode()
    cut = text[:text.index('+SOLUTION/STATISTICS')] + text[text.index('+SOLUTION/NORMAL_EQUATION_VECTOR'):]
    (tmp_path / 'no_stat.snx').write_text(cut)
    with pytest.raises(ValueError, match='storage schemes 6b or 6c'):
        ga.io.loadsinexnormals(str(tmp_path / 'no_stat.snx'))
    # unsupported parameter type
    (tmp_path / 'bad.snx').write_text(text.replace(' CN     ', ' STAX   ', 1))
    with pytest.raises(ValueError, match='not supported'):
        ga.io.loadsinex(str(tmp_path / 'bad.snx'))


def test_temporal_basis_functions(golden):
    """Design matrices of utilities.Polynomial / Oscillation (grates/utilities.py:462-557) against the reference, bit for bit;
    without a reference epoch the time argument is the modified Julian date."""
    g = golden('g16_time_variable')
    t0 = datetime.datetime(2005, 1, 1)
    epochs = [t0 + datetime.timedelta(days=9.5 * k) for k in range(5)]
    design = np.hstack((ga.utilities.Polynomial(2).design_matrix(epochs), ga.utilities.Oscillation(182.625).design_matrix(epochs)))
    np.testing.assert_array_equal(design, g['design_no_reference'])
    assert design[0, 1] == 53371.0                                                   # MJD of 2005-01-01
    d = ga.utilities.Polynomial(1, t0).design_matrix(epochs)
    np.testing.assert_array_equal(d, np.column_stack((np.ones(5), 9.5 * np.arange(5))))
    assert ga.utilities.Polynomial(0).design_matrix(epochs).shape == (5, 1)
    assert issubclass(ga.utilities.Oscillation, ga.utilities.TemporalBasisFunction)
    with pytest.raises(TypeError):
        ga.utilities.TemporalBasisFunction(None)


def test_time_variable_field_constituents(golden):
    """Trend + annual oscillation + interpolated series (grates/gravityfield.py:784-812, 1054-1140) against the reference's own evaluation
    (g16: the model at epoch 7 of 30), the constituents one by one against their formulas, and the reference import paths of the classes
    that live in grates_amd.extras."""
    g = golden('g16_time_variable')
    gfm = ga.gravityfield

    def field(seed):
        gf = gfm.PotentialCoefficients()
        gf.anm = inputs.coefficients(seed, 20)
        return gf
    t0 = datetime.datetime(2005, 1, 1)
    series = []
    for k in range(6):
        gf = field(120 + k)
        gf.epoch = t0 + datetime.timedelta(days=61 * k)
        series.append(gf)
    trend, cosine, sine = field(110), field(111), field(112)
    model = gfm.TimeVariableGravityField([gfm.Trend(trend, t0), gfm.Oscillation(cosine, sine, 365.25, t0), gfm.TimeSeries(series)])
    epoch = t0 + datetime.timedelta(days=9.5 * 7)
    at7 = model.evaluate_at(epoch)
    assert at7.epoch == epoch
    np.testing.assert_allclose(at7.anm, g['model_anm_at_7'], rtol=0, atol=1e-24)       # values ~1e-10
    days = 66.5
    np.testing.assert_array_equal(gfm.Trend(trend, t0).evaluate_at(epoch).anm, (trend * (days / 365.25)).anm)
    np.testing.assert_array_equal(gfm.Trend(trend, t0, time_scale=1.0).evaluate_at(epoch).anm, (trend * days).anm)
    phase = 2 * np.pi * days / 365.25
    np.testing.assert_allclose(gfm.Oscillation(cosine, sine, 365.25, t0).evaluate_at(epoch).anm, cosine.anm * np.cos(phase) + sine.anm * np.sin(phase),
                               rtol=0, atol=1e-26)
    trend.anm[:] = 0.0                                                                  # the constituents hold copies
    assert np.abs(gfm.Trend(field(110), t0).evaluate_at(epoch).anm).max() > 0
    assert ga.grid.ReuterGrid is ga.extras.ReuterGrid and gfm.SurfaceMasCons is ga.extras.SurfaceMasCons
    assert gfm.AnisotropicBasisFunctions is ga.extras.AnisotropicBasisFunctions
    with pytest.raises(AttributeError):
        gfm.NoSuchClass


def test_reuter_grid_and_latitude_mappings(golden):
    """ReuterGrid point distribution (grates/grid.py:1207-1278) for the three latitude mappings and the ellipsoid <-> sphere
    latitude mappings themselves (:2047-2110) against the reference (tests/golden/g17_reuter.npz)."""
    g = golden('g17_reuter')
    for mapping in ('geocentric', 'authalic', 'conformal'):
        grid = ga.extras.ReuterGrid(18, latitude_mapping=mapping)
        expected = g['reuter18_' + mapping]
        assert grid.point_count == expected.shape[1] == 403
        np.testing.assert_array_equal(grid.longitude, expected[0])
        np.testing.assert_allclose(grid.latitude, expected[1], rtol=0, atol=2e-16)
        np.testing.assert_array_equal(grid.area, expected[2])
        assert abs(grid.area.sum() - 4 * np.pi) < 0.02 * 4 * np.pi                   # Reuter areas tile the sphere approximately
    clone = ga.extras.ReuterGrid(18, latitude_mapping='authalic')
    clone.values = np.arange(403.0)
    twin = clone.copy()
    assert type(twin) is ga.extras.ReuterGrid and twin.is_compatible(clone)
    np.testing.assert_array_equal(twin.latitude, clone.latitude)
    np.testing.assert_array_equal(twin.values, clone.values)
    with pytest.raises(ValueError, match='Unknown latitude mapping'):
        ga.extras.ReuterGrid(5, latitude_mapping='mercator')
    beta = np.linspace(-0.5 * np.pi, 0.5 * np.pi, 37)
    G = ga.grid
    with np.errstate(invalid='ignore'):                                              # q / q0 exceeds 1 by an ulp at the poles, as upstream
        mine = np.vstack((G.geodetic2authalic(beta), G.authalic2geodetic(beta), G.geodetic2conformal(beta), G.conformal2geodetic(beta),
                          G.geodetic2geocentric(beta), G.geocentric2geodetic(beta)))
    np.testing.assert_allclose(mine, g['mappings'], rtol=0, atol=4e-16, equal_nan=True)
    assert G.authalic_radius() == float(g['authalic_radius'])
    inner = beta[1:-1]
    np.testing.assert_allclose(G.geocentric2geodetic(G.geodetic2geocentric(inner)), inner, rtol=0, atol=1e-15)
    np.testing.assert_allclose(G.conformal2geodetic(G.geodetic2conformal(inner)), inner, rtol=0, atol=1e-11)   # series truncated at e^8


def test_grid_subset_and_nearest_neighbour_index():
    """Grid.subset / Grid.nn_index (grates/grid.py:303-356): host helpers around the point lists."""
    grid = ga.grid.GeographicGrid(10.0, 10.0)
    lon, lat = grid.longitude, grid.latitude
    box = (np.abs(lon) < np.radians(45)) & (np.abs(lat) < np.radians(30))
    part = grid.subset(box)
    assert isinstance(part, ga.grid.RegularGrid) and part.point_count == int(box.sum())
    assert part.parallels.size * part.meridians.size == part.point_count
    rng = np.random.default_rng(0)
    ragged = grid.subset(rng.uniform(size=grid.point_count) < 0.3)
    assert type(ragged) is ga.grid.IrregularGrid and ragged.semimajor_axis == grid.semimajor_axis
    slon, slat = inputs.scattered_points(3, 500)
    index = grid.nn_index(slon, slat)
    assert len(index) == grid.point_count
    np.testing.assert_array_equal(np.sort(np.concatenate(index)), np.arange(500))          # every sample exactly once
    xyz = grid.cartesian_coordinates()
    sample = ga.grid.IrregularGrid(slon, slat).cartesian_coordinates()
    k = int(np.argmax([len(i) for i in index]))
    for j in index[k]:
        assert np.argmin(np.sum((xyz - sample[j]) ** 2, axis=1)) == k
