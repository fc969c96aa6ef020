"""
Golden-vector generator.  Run ONCE in the build container (where /root/reference is mounted):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference package (pure Python; netCDF4 / h5py are absent in the image and only used by
file loaders, so empty stand-in modules are registered first -- SURVEY.md 8c), feeds it the seeded
inputs of tests/golden/inputs.py and stores inputs' seeds + reference OUTPUTS as .npz fixtures next to
this script.  Nothing of the reference (source, bytecode, pickles of its classes) is written; only
arrays of numbers.  The GPU box never runs this script and never needs /root/reference.
"""

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import inputs  # noqa: E402

for _name, _attr in (('netCDF4', 'Dataset'), ('h5py', 'File')):
    if _name not in sys.modules:
        _mod = types.ModuleType(_name)
        setattr(_mod, _attr, None)
        sys.modules[_name] = _mod
sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
import grates  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('{0:28s} {1:9.1f} KB'.format(name, os.path.getsize(path) / 1024))


def geographic_colat(step):
    g = grates.grid.GeographicGrid(step, step)
    return grates.utilities.colatitude(g.parallels, g.semimajor_axis, g.flattening)


# ---- G1 / G2: Legendre functions ---------------------------------------------------------------
def g1_g2():
    out = {}
    for N, step, stride in ((5, 10.0, 1), (60, 1.0, 15), (96, 0.25, 72), (180, 0.5, 90)):
        colat = np.concatenate((geographic_colat(step)[::stride], inputs.SPECIAL_COLAT))
        out['colat_{0}'.format(N)] = colat
        out['pnm_{0}'.format(N)] = grates.utilities.legendre_functions(N, colat)
        for m in sorted({0, 1, N // 2, N}):
            out['pnm_order_{0}_{1}'.format(N, m)] = grates.utilities.legendre_functions_per_order(N, m, colat)
    # first 0.5-degree parallel at N=180 (underflow case SURVEY.md 5.3)
    out['pnm_0'] = grates.utilities.legendre_functions(0, np.array([0.3, 1.2]))
    save('g1_legendre', **out)


# ---- G3: trigonometric functions -----------------------------------------------------------------
def g3():
    out = {}
    for N, step, stride in ((20, 15.0, 1), (96, 0.25, 240)):
        lon = grates.grid.GeographicGrid(step, step).meridians[::stride]
        out['lon_{0}'.format(N)] = lon
        out['cs_{0}'.format(N)] = grates.utilities.trigonometric_functions(N, lon)
    lon, lat = inputs.scattered_points(3, 7)
    colat = 0.5 * np.pi - lat
    out['ynm_lon'], out['ynm_colat'] = lon, colat
    out['ynm_12'] = grates.utilities.spherical_harmonics(12, colat, lon)
    save('g3_trig', **out)


# ---- G4: index maps (bit-exact integers) ------------------------------------------------------------
def g4():
    out = {}
    for nmin, nmax in ((0, 5), (2, 5), (0, 60), (2, 96), (0, 180), (3, 3)):
        tag = '{0}_{1}'.format(nmin, nmax)
        # ravel an array whose entries are their own flat index: the result IS the gather map
        flat = np.arange((nmax + 1) ** 2, dtype=np.int64).reshape(nmax + 1, nmax + 1)
        out['ravel_' + tag] = grates.utilities.ravel_coefficients(flat, nmin, nmax)
        seq = grates.gravityfield.CoefficientSequenceDegreeWise(nmin, nmax)
        out['seq_' + tag] = np.array([(c.basis_function, c.degree, c.order) for c in seq.coefficients], dtype=np.int64)
        if nmax <= 60:
            vec = np.arange(seq.coefficient_count, dtype=np.int64) + 1
            out['unravel_' + tag] = grates.utilities.unravel_coefficients(vec, nmin, nmax)
            for m in sorted({0, 1, nmax // 2, nmax}):
                out['vidx_{0}_{1}'.format(tag, m)] = seq.vector_indices(order=m)
                if m > 0:
                    out['vidx_{0}_{1}_c'.format(tag, m)] = seq.vector_indices(order=m, cs='c')
                    out['vidx_{0}_{1}_s'.format(tag, m)] = seq.vector_indices(order=m, cs='s')
    # 3d ravel / 2d unravel, ravel with array smaller than max_degree
    arr3 = np.arange(3 * 36, dtype=np.int64).reshape(3, 6, 6)
    out['ravel3d_1_5'] = grates.utilities.ravel_coefficients(arr3, 1, 5)
    out['ravel_short_0_8'] = grates.utilities.ravel_coefficients(arr3[0], 0, 8)
    out['unravel2d_1_5'] = grates.utilities.unravel_coefficients(out['ravel3d_1_5'], 1, 5)
    for n, mo in ((0, None), (4, None), (7, 3)):
        r, c = grates.gravityfield.degree_indices(n, max_order=mo)
        out['degidx_{0}_{1}'.format(n, mo)] = np.vstack((r, c))
    for N, m in ((6, 0), (6, 2), (6, 6)):
        r, c = grates.gravityfield.order_indices(N, m)
        out['ordidx_{0}_{1}'.format(N, m)] = np.vstack((r, c))
    save('g4_index', **out)


# ---- G5: ellipsoid geometry + grids ------------------------------------------------------------------
def g5():
    out = {}
    for step in (1.0, 0.25):
        g = grates.grid.GeographicGrid(step, step)
        tag = str(step).replace('.', 'p')
        out['parallels_' + tag] = g.parallels
        out['meridians_' + tag] = g.meridians
        out['colat_' + tag] = grates.utilities.colatitude(g.parallels, g.semimajor_axis, g.flattening)
        out['radius_' + tag] = grates.utilities.geocentric_radius(g.parallels, g.semimajor_axis, g.flattening)
        out['area_rowsum_' + tag] = g.area.reshape(g.parallels.size, -1).sum(axis=1)
    g = grates.grid.GeographicGrid(2.0, 5.0)
    out['geo_2_5_meridians'], out['geo_2_5_parallels'], out['geo_2_5_area'] = g.meridians, g.parallels, g.area
    out['geo_2_5_lon'], out['geo_2_5_lat'] = g.longitude, g.latitude
    gg = grates.grid.GaussGrid(31)
    out['gauss31_meridians'], out['gauss31_parallels'], out['gauss31_area'] = gg.meridians, gg.parallels, gg.area
    rg = grates.grid.RegularGrid(np.linspace(-3.0, 3.0, 7), np.linspace(1.4, -1.4, 5))
    out['regular_area'] = rg.area
    save('g5_geometry', **out)


# ---- G6: kernel tables ---------------------------------------------------------------------------------
def g6():
    out = {}
    g = grates.grid.GeographicGrid(1.0, 1.0)
    lat = g.parallels[::24]
    colat = grates.utilities.colatitude(lat, g.semimajor_axis, g.flattening)
    r = grates.utilities.geocentric_radius(lat, g.semimajor_axis, g.flattening)
    out['lat'], out['colat'], out['r'] = lat, colat, r
    for name in ('ewh', 'potential', 'geoid', 'obp', 'surface_density', 'anomaly', 'deformation', 'uplift'):
        ker = grates.kernel.get_kernel(name)
        out['inv_' + name] = ker.inverse_coefficients(0, 180, r, colat)
        out['coef_' + name] = ker.coefficients(2, 40, r, colat)
    ker = grates.kernel.get_kernel('ewh')
    out['ewh_scalar'] = ker.coefficients(0, 10)
    out['ewh_coefficient_7'] = ker.coefficient(7, r, colat)
    out['ewh_inverse_coefficient_0'] = ker.inverse_coefficient(0, r, colat)
    out['ewh_coef_array_2_6'] = ker.coefficient_array(2, 6)      # multi-point call raises upstream (broadcast bug)
    out['ewh_inv_array_2_6'] = ker.inverse_coefficient_array(2, 6)
    for frame in ('CE', 'CM', 'CF'):
        k, h, l = grates.data.load_love_numbers(frame=frame)
        out['love_' + frame] = np.vstack((k[0:257], h[0:257], l[0:257]))
    for radius in (0, 200, 300, 500):
        out['gauss_{0}'.format(radius)] = grates.kernel.Gauss(radius).coefficients(0, 200)
    out['gauss_20_ext'] = grates.kernel.Gauss(20).coefficients(1000, 1100)   # exercises the >1024 extension
    out['gauss_300_psi'] = np.linspace(0, 0.3, 7)
    out['gauss_300_eval'] = grates.kernel.Gauss(300).evaluate(0, 100, out['gauss_300_psi'])
    out['normal_gravity'] = grates.gravityfield.GRS80.normal_gravity(r, colat)
    out['normal_gravity_eq_pole'] = np.array([grates.gravityfield.GRS80.normal_gravity(6378137.0, np.pi / 2)[0],
                                              grates.gravityfield.GRS80.normal_gravity(6378137.0 * (1 - grates.gravityfield.GRS80.flattening), 0.0)[0]])
    out['grs80_flattening'] = np.array([grates.gravityfield.GRS80.flattening])
    out['grs80_anm_col0'] = grates.gravityfield.GRS80.anm[:, 0]
    out['wgs84_J2'] = np.array([grates.gravityfield.WGS84.J2])
    save('g6_kernel', **out)


# ---- G7: synthesis -----------------------------------------------------------------------------------------
def potential_coefficients(anm):
    gf = grates.gravityfield.PotentialCoefficients()
    gf.anm = anm.copy()
    return gf


def g7():
    out = {}
    # C1: d/o 60, Gaussian 300 km, 1 degree, ewh (BASELINE config 1)
    gf = potential_coefficients(inputs.coefficients(1000, 60))
    filtered = grates.filter.Gaussian(300).filter(gf)
    grid = filtered.to_grid(grates.grid.GeographicGrid(1.0, 1.0), kernel='ewh')
    out['c1_filtered_anm'] = filtered.anm
    out['c1_grid'] = grid.value_array
    # C2 unit: d/o 96 -> 0.25 degree, two epochs, strided sample + row sums
    for e in (0, 1):
        gf = potential_coefficients(inputs.coefficients(1000 + e, 96))
        va = gf.to_grid(grates.grid.GeographicGrid(0.25, 0.25), kernel='ewh').value_array
        out['c2_sample_{0}'.format(e)] = va[::9, ::11].copy()
        out['c2_rowsum_{0}'.format(e)] = va.sum(axis=1)
        out['c2_maxabs_{0}'.format(e)] = np.array([np.abs(va).max()])
    # potential kernel on a Gauss grid
    gf = potential_coefficients(inputs.coefficients(7, 60))
    out['gauss61_potential'] = gf.to_grid(grates.grid.GaussGrid(61), kernel='potential').value_array
    # other kernels, small grid
    gf = potential_coefficients(inputs.coefficients(8, 30))
    for name in ('geoid', 'obp', 'potential', 'ewh'):
        out['n30_5deg_' + name] = gf.to_grid(grates.grid.GeographicGrid(5.0, 5.0), kernel=name).value_array
    # non-default GM / R
    gf = grates.gravityfield.PotentialCoefficients(GM=3.986004418e14, R=6378137.0)
    gf.anm = inputs.coefficients(9, 30)
    out['n30_5deg_gmr'] = gf.to_grid(grates.grid.GeographicGrid(5.0, 5.0), kernel='ewh').value_array
    # point-list fallback (no .parallels attribute -> AttributeError path)
    lon, lat = inputs.scattered_points(11, 1000)
    gf = potential_coefficients(inputs.coefficients(12, 40))
    out['points_ewh'] = gf.to_grid(grates.grid.IrregularGrid(lon, lat), kernel='ewh').values
    # non-symmetric regular grid (general path)
    mer = np.array([-3.0, -2.2, -0.4, 0.1, 0.9, 2.5, 3.1])
    par = np.array([1.3, 1.0, 0.2, -0.5, -1.45])
    out['asym_meridians'], out['asym_parallels'] = mer, par
    out['asym_potential'] = potential_coefficients(inputs.coefficients(13, 25)).to_grid(grates.grid.RegularGrid(mer, par), kernel='potential').value_array
    save('g7_synthesis', **out)


# ---- G8: analysis ----------------------------------------------------------------------------------------------
def g8():
    out = {}
    grid = potential_coefficients(inputs.coefficients(21, 60)).to_grid(grates.grid.GeographicGrid(1.0, 1.0), kernel='potential')
    out['n60_1deg_anm'] = grid.to_potential_coefficients(0, 60, kernel='potential').anm
    grid = potential_coefficients(inputs.coefficients(22, 30)).to_grid(grates.grid.GaussGrid(31), kernel='ewh')
    out['gauss31_values'] = grid.value_array
    out['gauss31_anm_ewh_2_30'] = grid.to_potential_coefficients(2, 30, kernel='ewh').anm
    # analysis of arbitrary (non band-limited) values
    g = grates.grid.GeographicGrid(5.0, 5.0)
    g.values = np.random.default_rng(23).standard_normal(g.point_count)
    out['n20_5deg_random_anm'] = g.to_potential_coefficients(0, 20, kernel='potential').anm
    out['n8_5deg_analysis_matrix'] = g.analysis_matrix(1, 8, 'potential')
    out['n8_5deg_synthesis_matrix'] = g.synthesis_matrix(1, 8, 'ewh')
    save('g8_analysis', **out)


# ---- G9: covariance propagation -------------------------------------------------------------------------------------
def g9():
    out = {}
    cov = inputs.spd_covariance(31, 41 * 41)
    out['n40_2deg_ewh'] = grates.grid.GeographicGrid(2.0, 2.0).covariance_propagation(cov, 0, 40, kernel='ewh')
    cov = inputs.spd_covariance(32, 21 * 21 - 4)
    out['n20_5deg_min2_potential'] = grates.grid.GeographicGrid(5.0, 5.0).covariance_propagation(cov, 2, 20, kernel='potential')
    cov = inputs.spd_covariance(33, 21 * 21)
    g = grates.grid.GeographicGrid(5.0, 5.0)
    out['n20_5deg_ewh'] = g.covariance_propagation(cov, 0, 20, kernel='ewh')
    A = g.synthesis_matrix(0, 20, 'ewh')
    out['n20_5deg_ewh_einsum'] = np.sqrt(np.einsum('ij,jk,ik->i', A, cov, A))
    lon, lat = inputs.scattered_points(34, 300)
    out['points_n20_ewh'] = grates.grid.IrregularGrid(lon, lat).covariance_propagation(cov, 0, 20, kernel='ewh')
    save('g9_covariance', **out)


# ---- G10: filters ----------------------------------------------------------------------------------------------------------
def g10():
    out = {}
    gf = potential_coefficients(inputs.coefficients(41, 60))
    out['gaussian_300_n60'] = grates.filter.Gaussian(300).filter(gf).anm
    out['gaussian_500_matrix_2_12_diag'] = np.diag(grates.filter.Gaussian(500).matrix(2, 12))
    for nmax, ngf in ((20, 20), (120, 120), (120, 96)):
        blocks = inputs.orderwise_random_blocks(42, nmax)
        flt = grates.filter.OrderWiseFilter(blocks)
        gf = potential_coefficients(inputs.coefficients(43, ngf))
        out['orderwise_{0}_{1}'.format(nmax, ngf)] = flt.filter(gf).anm
        if nmax == 20:
            out['orderwise_20_matrix_0_20'] = flt.matrix(0, 20)
            out['orderwise_20_matrix_2_14'] = flt.matrix(2, 14)
    # DDK construction from synthetic SPD normals (the real ddk_normal_blocks.npz is absent from the mount)
    for nmax, level in ((20, 5), (20, 3)):
        normals = inputs.orderwise_normal_blocks(44, nmax)
        grates.filter.DDKGeneric._blocked_normals = staticmethod(lambda normals=normals: normals)
        ddk = grates.filter.DDK(level)
        gf = potential_coefficients(inputs.coefficients(45, nmax))
        out['ddk{0}_n{1}'.format(level, nmax)] = ddk.filter(gf).anm
        out['ddk{0}_n{1}_matrix'.format(level, nmax)] = ddk.matrix(2, nmax)
        gen = grates.filter.DDKGeneric(level)
        out['ddkgeneric{0}_n{1}'.format(level, nmax)] = gen.filter(gf).anm
    # dense matrix filter
    W = np.random.default_rng(46).standard_normal((21 * 21 - 4, 21 * 21 - 4)) / 21
    gm = grates.filter.GeneralMatrix(W, 2, 20)
    out['general_2_20_n20'] = gm.filter(potential_coefficients(inputs.coefficients(47, 20))).anm
    out['general_2_20_n14'] = gm.filter(potential_coefficients(inputs.coefficients(47, 14))).anm
    out['general_matrix_3_18'] = gm.matrix(3, 18)
    # time series -> array
    series = []
    import datetime
    for e in range(4):
        gf = potential_coefficients(inputs.coefficients(50 + e, 6))
        gf.epoch = datetime.datetime(2010, 1 + e, 15)
        series.append(gf)
    out['timeseries_array'] = grates.gravityfield.TimeSeries(series[::-1]).to_array()
    # PotentialCoefficients arithmetic
    a = grates.gravityfield.PotentialCoefficients(GM=3.986004418e14, R=6378137.0)
    a.anm = inputs.coefficients(60, 8)
    b = potential_coefficients(inputs.coefficients(61, 12))
    out['pc_add'] = (a + b).anm
    out['pc_sub'] = (b - a).anm
    out['pc_mul'] = (a * 2.5).anm
    out['pc_slice'] = b.slice(min_degree=2, max_degree=10, min_order=1, max_order=6, step_degree=2).anm
    out['pc_values'] = b.values
    d, amp = b.degree_amplitudes(kernel='ewh')
    out['pc_degree_amplitudes'] = amp
    save('g10_filter', **out)


# ---- G12: dense decorrelation filter from a full normal matrix (VDK) and filter kernels in the space domain ----------------
def g12():
    out = {}
    nmin, nmax = 2, 12
    P = (nmax + 1) ** 2 - nmin ** 2
    N = inputs.spd_covariance(70, P, scale=1e20)                    # synthetic SPD normal equation matrix, degree-wise
    vdk = grates.filter.VDK(N, nmin, nmax, 1e18, 2.0)
    out['vdk_matrix'] = vdk.matrix(nmin, nmax)
    # (VDK.filter is unusable upstream: it reads name-mangled attributes of its base class; the matrix is the contract)
    W = vdk.matrix(nmin, nmax)
    gm = grates.filter.GeneralMatrix(W, nmin, nmax)
    out['vdk_filtered_n12'] = gm.filter(potential_coefficients(inputs.coefficients(71, 12))).anm
    # space-domain kernel of an anisotropic filter
    src_lon, src_lat = np.deg2rad(13.0), np.deg2rad(47.5)
    ev_lon = np.deg2rad(np.linspace(-20.0, 50.0, 9))
    ev_lat = np.deg2rad(np.linspace(30.0, 65.0, 6))
    for name, kernel in (('potential', 'potential'), ('ewh', 'ewh')):
        fk = grates.filter.FilterKernel(gm, nmin, nmax, input_kernel=kernel)
        pts_lon, pts_lat = np.meshgrid(ev_lon, ev_lat)
        out['filterkernel_{0}_points'.format(name)] = fk.evaluate(src_lon, src_lat, pts_lon.ravel(), pts_lat.ravel())
        # (upstream FilterKernel.evaluate_grid raises: its matrix carries a leading axis of length 1; evaluate() on the mesh
        #  defines the grid values)
    K = np.random.default_rng(72).standard_normal((P, P)) / nmax
    ak = grates.kernel.AnisotropicKernel(K, nmin, nmax)
    out['anisotropic_grid'] = ak.evaluate_grid(src_lon, src_lat, ev_lon, ev_lat)
    out['anisotropic_points'] = ak.evaluate(src_lon, src_lat, ev_lon, ev_lat[0:1].repeat(ev_lon.size))
    ord_blocks = inputs.orderwise_random_blocks(73, nmax)
    fo = grates.filter.FilterKernel(grates.filter.OrderWiseFilter(ord_blocks), nmin, nmax)
    pts_lon, pts_lat = np.meshgrid(ev_lon, ev_lat)
    out['filterkernel_orderwise_points'] = fo.evaluate(src_lon, src_lat, pts_lon.ravel(), pts_lat.ravel())
    save('g12_filter_kernel', **out)


# ---- G11: block-banded normal equations ("Kalman smoother", lstsq.py) ----------------------------------------------------
def g11():
    from grates import lstsq
    out = {}
    dim, order, epochs = 6, 2, 7
    cf = inputs.var_covariance_function(1, dim, order)
    seq = lstsq.AutoregressiveModelSequence.from_covariance_function(cf)
    for k in range(order + 1):
        model = lstsq.AutoregressiveModel.from_covariance_function(cf[0:k + 1])
        out['var{0}_Q'.format(k)] = model.white_noise_covariance
        if k:
            out['var{0}_coefficients'.format(k)] = np.array(model.coefficients)
        for r in range(k + 1):
            for c in range(r, k + 1):
                out['var{0}_normals_{1}{2}'.format(k, r, c)] = model.normal_equation_block(r, c)
    constraint = seq.normal_equations(epochs)
    out['constraint_matrix'] = constraint.matrix.to_array()
    out['covariance_function_back'] = np.array(seq.covariance_function(3))

    # smoother: observation normals of independent epochs + VAR constraint, variance factors (1, 0.5)
    def observation_system():
        per_epoch = inputs.observation_normals(2, epochs, dim)
        idx = np.arange(0, (epochs + 1) * dim, dim)
        bm = lstsq.BlockMatrix(idx, idx)
        for t, e in enumerate(per_epoch):
            bm[t, t] = e[0]
        return lstsq.NormalEquations(bm, np.vstack([e[1] for e in per_epoch]), sum(e[2] for e in per_epoch), sum(e[3] for e in per_epoch))
    factors = [1.0, 0.5]
    parts = [observation_system(), seq.normal_equations(epochs)]
    combined = lstsq.accumulate_normals(parts, factors)
    out['combined_matrix'] = combined.matrix.to_array()
    out['combined_rhs'] = combined.right_hand_side
    out['combined_lPl'] = np.array(combined.observation_square_sum)
    out['combined_count'] = np.array(combined.observation_count)
    np.random.seed(123)
    x = combined.solve()
    np.random.seed(123)
    signs = np.random.randint(0, 2, size=(epochs * dim, 100))
    signs[signs == 0] = -1
    out['signs'] = signs.astype(np.int8)
    out['solution'] = x
    out['monte_carlo_vectors'] = combined.monte_carlo_vectors
    out['factor'] = combined.matrix.to_array()
    out['posterior_sigma'] = np.array(combined.posterior_sigma(x))
    out['residual_square_sums'] = np.array([p.residual_square_sum(x) for p in parts])
    out['redundancies'] = np.array([p.redundancy(combined, f) for p, f in zip(parts, factors)])
    out['variance_factors'] = lstsq.compute_variance_factors(parts, combined, x, factors)
    combined.compute_covariance(sparse=True)
    out['sparse_inverse'] = combined.matrix.to_array()
    again = lstsq.accumulate_normals([observation_system(), seq.normal_equations(epochs)], factors)
    again.compute_covariance(sparse=False)
    out['full_inverse'] = again.matrix.to_array()

    # general block matrix: ragged blocks (17 = 5 + 5 + 5 + 2), blocks (0, 3) and (1, 2) empty, fill-in at (1, 2)
    n = 17
    A = inputs.spd_covariance(9, n, scale=1.0) + np.eye(n)
    rows, cols = lstsq.BlockMatrix.compute_block_index(A.shape, 5)
    out['ragged_index'] = rows
    A[0:5, 15:17] = 0
    A[15:17, 0:5] = 0
    A[5:10, 10:15] = 0
    A[10:15, 5:10] = 0
    out['ragged_input'] = A
    upper = np.triu(A)
    bm = lstsq.BlockMatrix.from_array(upper, rows, cols)
    b = np.random.default_rng(10).standard_normal((n, 3))
    out['ragged_rhs'] = b
    out['ragged_multiply_symmetric'] = bm.multiply_symmetric(b)
    out['ragged_diag'] = bm.diag()
    sq = bm @ bm
    out['ragged_matmul'] = sq.to_array()
    bm.cholesky()
    out['ragged_factor'] = bm.to_array()
    out['ragged_solve_T'] = bm.solve_triangular(b, transpose=True)
    out['ragged_solve_N'] = bm.solve_triangular(b, transpose=False)
    out['ragged_multiply_N'] = bm.multiply_triangular(b, transpose=False)
    out['ragged_multiply_T'] = bm.multiply_triangular(b, transpose=True)
    sp = bm.copy()
    sp.sparse_inverse()
    out['ragged_sparse_inverse'] = sp.to_array()
    bm.inverse()
    out['ragged_inverse'] = bm.to_array()

    # Tikhonov regularisation
    reg = np.random.default_rng(11).uniform(0.5, 2.0, 12)
    bias = np.random.default_rng(12).standard_normal((12, 1))
    tk = lstsq.TikhonovRegularization(reg, [0, 4, 8, 12], bias)
    out['tikhonov_matrix'] = tk.matrix.to_array()
    out['tikhonov_rhs'] = tk.right_hand_side
    out['tikhonov_lPl'] = np.array(tk.observation_square_sum)
    out['tikhonov_count'] = np.array(tk.observation_count)
    save('g11_lstsq', **out)


# ---- G13: file feeders (synthetic GFC / GSM files written by inputs.py, parsed by the reference) ---------------------------
def g13():
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for tag, seed, nmax, header in (('a', 80, 12, True), ('b', 81, 7, False)):
            path = os.path.join(tmp, 'model_{0}.gfc'.format(tag))
            with open(path, 'wb') as f:
                f.write(inputs.gfc_file_text(seed, nmax, header))
            gf = grates.io.loadgfc(path)
            out['gfc_{0}_anm'.format(tag)] = gf.anm
            out['gfc_{0}_GM_R'.format(tag)] = np.array([gf.GM, gf.R])
        out['gfc_a_truncated_anm'] = grates.io.loadgfc(os.path.join(tmp, 'model_a.gfc'), max_degree=5).anm
        path = os.path.join(tmp, 'GSM-2_2010060-2010090.txt')
        with open(path, 'wb') as f:
            f.write(inputs.gsm_file_text(82, 10))
        gf = grates.io.loadgsm(path)
        out['gsm_anm'] = gf.anm
        out['gsm_GM_R'] = np.array([gf.GM, gf.R])
        out['gsm_epoch'] = np.array([gf.epoch.year, gf.epoch.month, gf.epoch.day, gf.epoch.hour, gf.epoch.minute, gf.epoch.second])
    save('g13_io', **out)


def g14():
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for tag, seed, nmin, nmax, lower in (('u', 90, 2, 8, False), ('l', 91, 0, 5, True)):
            path = os.path.join(tmp, 'normals_{0}.snx'.format(tag))
            with open(path, 'wb') as f:
                f.write(inputs.sinex_file_text(seed, nmin, nmax, lower))
            N, n, lPl, obs_count = grates.io.loadsinexnormals(path)
            out['sinex_{0}_N'.format(tag)] = N
            out['sinex_{0}_n'.format(tag)] = n
            out['sinex_{0}_lPl'.format(tag)] = lPl
            out['sinex_{0}_obs_count'.format(tag)] = np.array(obs_count)
    save('g14_sinex', **out)


def g15():
    out = {}
    lon, lat, values, k_rbf, k_aniso = inputs.basis_function_case(100, 700, 2, 12)
    rbf = grates.gravityfield.RadialBasisFunctions(grates.grid.IrregularGrid(lon, lat), k_rbf, 2, 12)
    rbf.values = values
    out['rbf_anm'] = rbf.to_potential_coefficients().anm
    F = rbf.to_potential_coefficients_matrix()
    out['rbf_matrix_cols10'] = F[:, ::10]                      # every tenth nodal point; F @ values pins the rest
    out['rbf_matrix_times_values'] = F @ values
    out['rbf_grid_ewh'] = rbf.to_grid(grates.grid.GeographicGrid(10.0, 10.0), 'ewh').value_array
    aniso = grates.gravityfield.AnisotropicBasisFunctions(grates.grid.IrregularGrid(lon, lat), k_aniso, 2, 12)
    aniso.values = values
    out['aniso_grid_ewh'] = aniso.to_grid(grates.grid.GeographicGrid(10.0, 10.0), 'ewh').value_array
    out['aniso_grid_potential'] = aniso.to_grid(grates.grid.GeographicGrid(10.0, 10.0), 'potential').value_array
    a = grates.gravityfield.SurfaceMasCons(grates.grid.IrregularGrid(lon, lat), 'ewh')
    b = a.copy()
    a.values = values
    b.values = values[::-1].copy()
    out['mascon_arith'] = ((a + b) * 3 - a / 4.0).values
    save('g15_basis_functions', **out)


def g16():
    import datetime as dt
    out = {}
    gfs = [potential_coefficients(inputs.coefficients(110 + k, 20)) for k in range(3)]
    t0 = dt.datetime(2005, 1, 1)
    series = []
    for k in range(6):
        gf = potential_coefficients(inputs.coefficients(120 + k, 20))
        gf.epoch = t0 + dt.timedelta(days=61 * k)
        series.append(gf)
    model = grates.gravityfield.TimeVariableGravityField([grates.gravityfield.Trend(gfs[0], t0),
                                                          grates.gravityfield.Oscillation(gfs[1], gfs[2], 365.25, t0),
                                                          grates.gravityfield.TimeSeries(series)])
    epochs = [t0 + dt.timedelta(days=9.5 * k) for k in range(30)]
    out['model_anm_at_7'] = model.evaluate_at(epochs[7]).anm
    out['rms_ewh_5deg'] = grates.gravityfield.gridded_rms(model, epochs, 'ewh', grates.grid.GeographicGrid(5.0, 5.0)).values
    binned = grates.gravityfield.TimeSeries(series).bin([t0 + dt.timedelta(days=30), t0 + dt.timedelta(days=200), t0 + dt.timedelta(days=290)],
                                                           func=lambda members: sum(members[1:], members[0]) * (1.0 / len(members)))
    out['binned_anm'] = np.array([d.anm for _, d in binned.items()])
    # detrend: bias + drift + annual oscillation removed from a series of 40 epochs
    rng = np.random.default_rng(130)
    fields = []
    for k in range(40):
        gf = potential_coefficients(inputs.coefficients(140, 15) * (1 + 0.01 * k) + inputs.coefficients(141, 15) * np.sin(2 * np.pi * k * 30.4 / 365.25)
                                    + rng.standard_normal((16, 16)) * 1e-12)
        gf.epoch = t0 + dt.timedelta(days=30.4 * k)
        fields.append(gf)
    ts = grates.gravityfield.TimeSeries(fields)
    out['detrend_parameters'] = ts.detrend([grates.utilities.Polynomial(1, t0), grates.utilities.Oscillation(365.25, t0)])
    out['detrend_residuals'] = ts.to_array()
    out['design_no_reference'] = np.hstack((grates.utilities.Polynomial(2).design_matrix(epochs[:5]), grates.utilities.Oscillation(182.625).design_matrix(epochs[:5])))
    save('g16_time_variable', **out)


def g17():
    out = {}
    for mapping in ('geocentric', 'authalic', 'conformal'):
        grid = grates.grid.ReuterGrid(18, latitude_mapping=mapping)
        out['reuter18_' + mapping] = np.vstack((grid.longitude, grid.latitude, grid.area))
    beta = np.linspace(-0.5 * np.pi, 0.5 * np.pi, 37)
    g = grates.grid
    out['mappings'] = np.vstack((g.geodetic2authalic(beta), g.authalic2geodetic(beta), g.geodetic2conformal(beta), g.conformal2geodetic(beta),
                                 g.geodetic2geocentric(beta), g.geocentric2geodetic(beta)))
    out['authalic_radius'] = np.array(g.authalic_radius())
    gf = potential_coefficients(inputs.coefficients(150, 40))
    out['reuter30_ewh'] = gf.to_grid(grates.grid.ReuterGrid(30), kernel='ewh').values
    save('g17_reuter', **out)


# ---- G18: full-matrix operators: window matrix, filtered covariance, per-parallel covariance blocks ----
def g18():
    out = {}
    # window matrix (grid.py:449-475): grid values in [0, 1] as window function
    g = grates.grid.GeographicGrid(5.0, 5.0)
    g.values = np.random.default_rng(51).uniform(0.0, 1.0, g.point_count)
    out['window_5deg_1_8_potential'] = g.window_matrix(1, 8, 'potential')
    gg = grates.grid.GaussGrid(13)
    gg.values = (np.random.default_rng(52).uniform(0.0, 1.0, gg.point_count) > 0.4).astype(float)
    out['window_gauss13_0_10_ewh'] = gg.window_matrix(0, 10, 'ewh')
    # filtered covariance W Sigma W^T (W = filter.matrix) ahead of the propagation, d/o 12 and d/o 20
    for nmax, seed in ((12, 53), (20, 54)):
        blocks = []
        for k, nb in enumerate(inputs.orderwise_normal_blocks(seed, nmax)):
            m = (k + 1) // 2
            w = 1e11 * np.arange(m, nmax + 1, dtype=float) ** 4
            w[w == 0] = 1.0
            blocks.append(np.linalg.solve(nb + np.diag(w), nb))
        flt = grates.filter.OrderWiseFilter(blocks)
        W = flt.matrix(2, nmax)
        cov = inputs.spd_covariance(seed + 10, W.shape[0])
        filtered = W @ cov @ W.T
        grid = grates.grid.GeographicGrid(5.0, 5.0)
        out['filtered_sigma_n{0}_ewh'.format(nmax)] = grid.covariance_propagation(filtered, 2, nmax, kernel='ewh')
        if nmax == 12:
            out['filtered_cov_n12'] = filtered
        gauss = grates.filter.Gaussian(400).matrix(2, nmax)
        out['gauss_filtered_sigma_n{0}_potential'.format(nmax)] = grid.covariance_propagation(gauss @ cov @ gauss.T, 2, nmax, kernel='potential')
    # per-parallel blocks F Sigma F^T (grid.py:833-835) for three parallels of the 5 degree grid at d/o 20
    cov = inputs.spd_covariance(33, 21 * 21)
    g = grates.grid.GeographicGrid(5.0, 5.0)
    A = g.synthesis_matrix(0, 20, 'ewh')
    nlon = g.meridians.size
    for i in (0, 17, 35):
        F = A[i * nlon:(i + 1) * nlon]
        out['block_n20_5deg_ewh_{0}'.format(i)] = F @ cov @ F.T
    save('g18_operators', **out)


# ---- G19: the public API surface of the path: names, signatures, exception types -----------------------------------
API_MODULES = ('utilities', 'kernel', 'gravityfield', 'grid', 'filter', 'lstsq', 'io', 'data')


def g19():
    """Names and inspect.signature strings of every public callable the reference defines in the modules of the path, the public methods /
    properties of its classes, and the exception TYPE each probe of inputs.api_probes raises -- data about the interface, no code."""
    import inspect
    import json
    import re

    def describe(obj):
        """[[name, kind, repr(default) | None], ...]; a default that is an object instance is recorded as '<instance of Type>'"""
        try:
            sig = inspect.signature(obj)
        except (TypeError, ValueError):
            return None
        out = []
        for prm in sig.parameters.values():
            default = None
            if prm.default is not inspect.Parameter.empty:
                default = re.sub(r' at 0x[0-9a-f]+', '', repr(prm.default))
                if ' object>' in default:
                    default = '<instance of {0}>'.format(type(prm.default).__name__)
            out.append([prm.name, prm.kind.name, default])
        return out

    api = {}
    for modname in API_MODULES:
        mod = getattr(grates, modname)
        entry = {}
        for name, obj in sorted(vars(mod).items()):
            if name.startswith('_') or getattr(obj, '__module__', None) != mod.__name__:
                continue
            if inspect.isclass(obj):
                members = {}
                for mname, member in sorted(vars(obj).items()):
                    if mname.startswith('_') and mname != '__init__':
                        continue
                    if isinstance(member, property):
                        members[mname] = 'property' + ('+setter' if member.fset else '')
                    elif isinstance(member, (staticmethod, classmethod)):
                        members[mname] = [type(member).__name__] + (describe(member.__func__) or [])
                    elif callable(member):
                        members[mname] = describe(member)
                entry[name] = {'kind': 'class', 'bases': [b.__name__ for b in obj.__mro__[1:-1]], 'members': members}
            elif inspect.isfunction(obj):
                entry[name] = {'kind': 'function', 'signature': describe(obj)}
        api[modname] = entry
    raised = {}
    for label, thunk in inputs.api_probes(grates):
        try:
            thunk()
            raised[label] = None
        except Exception as err:            # noqa: BLE001 -- the TYPE is the datum
            raised[label] = type(err).__name__
    path = os.path.join(HERE, 'g19_api.json')
    with open(path, 'w') as f:
        json.dump({'modules': api, 'raises': raised}, f, indent=1, sort_keys=True)
    print('{0:28s} {1:9.1f} KB'.format('g19_api.json', os.path.getsize(path) / 1024))


if __name__ == '__main__':
    only = sys.argv[1:]
    for fn in (g1_g2, g3, g4, g5, g6, g7, g8, g9, g10, g11, g12, g13, g14, g15, g16, g17, g18, g19):
        if not only or fn.__name__ in only:
            fn()
