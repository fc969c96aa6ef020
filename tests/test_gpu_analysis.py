"""
GPU parity of the analysis path (rows a12 / a13 of SURVEY.md 8a): RegularGrid.to_potential_coefficients,
analysis_matrix / synthesis_matrix and the irregular-grid least squares, against golden vectors and the oracle.
The reference solves the normal equations per order with LU; the device builds the operator (A^T W A)^-1 A^T W once from a
Cholesky factor of the same normal matrix.  Measured against the golden vectors (r02, MI355X): 7e-15 ... 1.3e-14 relative,
the reference's own round trip of a band-limited field is 9e-15; the NumPy oracle reproduces the golden vectors bit for
bit (spread 0).  TOL below leaves two orders of magnitude for other grids / conditioning.
"""

import numpy as np
import pytest

import grates_amd as ga
import inputs
from conftest import relerr
from oracle import shg_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-12


def love():
    return ga.data.load_love_numbers()[0]


def make_pc(anm):
    gf = ga.gravityfield.PotentialCoefficients()
    gf.anm = anm.copy()
    return gf


def test_golden_gauss_grid(golden):
    g = golden('g8_analysis')
    grid = ga.grid.GaussGrid(31)
    grid.values = g['gauss31_values'].ravel().copy()
    out = grid.to_potential_coefficients(2, 30, kernel='ewh')
    assert isinstance(out, ga.gravityfield.PotentialCoefficients) and out.anm.shape == (31, 31)
    assert relerr(out.anm, g['gauss31_anm_ewh_2_30']) < TOL
    assert np.all(out.anm[0:2, 0] == 0.0) and out.anm[0, 1] == 0.0           # degrees below min_degree stay zero


def test_golden_random_values_and_matrices(golden):
    g = golden('g8_analysis')
    grid = ga.grid.GeographicGrid(5.0, 5.0)
    grid.values = np.random.default_rng(23).standard_normal(grid.point_count)
    assert relerr(grid.to_potential_coefficients(0, 20, kernel='potential').anm, g['n20_5deg_random_anm']) < TOL
    assert relerr(grid.synthesis_matrix(1, 8, 'ewh'), g['n8_5deg_synthesis_matrix']) < 1e-12
    F = grid.analysis_matrix(1, 8, 'potential')
    assert F.shape == g['n8_5deg_analysis_matrix'].shape
    assert relerr(F, g['n8_5deg_analysis_matrix']) < TOL
    # per-order blocks of the synthesis matrix
    Ac, As = grid.synthesis_matrix_per_order(3, 1, 8, 'ewh', 3.9860044150e+14, 6.3781363000e+06)
    ref_c, ref_s = orc.synthesis_matrix_per_order(3, 1, 8, grid.meridians, grid.parallels, orc.KernelTable('ewh', love()))
    assert relerr(Ac, ref_c) < 1e-12 and relerr(As, ref_s) < 1e-12
    # order 0 (one block, no longitude dependence), an order below min_degree (columns start at min_degree) and the last order
    ker = orc.KernelTable('ewh', love())
    A0 = grid.synthesis_matrix_per_order(0, 1, 8, 'ewh', 3.9860044150e+14, 6.3781363000e+06)
    assert A0.shape == (grid.point_count, 8) and relerr(A0, orc.synthesis_matrix_per_order(0, 1, 8, grid.meridians, grid.parallels, ker)) < 1e-12
    for m, nmin in ((2, 5), (8, 0)):
        Ac, As = grid.synthesis_matrix_per_order(m, nmin, 8, 'ewh', 3.9860044150e+14, 6.3781363000e+06)
        ref_c, ref_s = orc.synthesis_matrix_per_order(m, nmin, 8, grid.meridians, grid.parallels, ker)
        assert Ac.shape == ref_c.shape == (grid.point_count, 9 - max(m, nmin)) and relerr(Ac, ref_c) < 1e-12 and relerr(As, ref_s) < 1e-12
    # point list (IrregularGrid.synthesis_matrix_per_order, grates/grid.py:957-991): the rows of the regular block, point by point
    pts = ga.grid.IrregularGrid(grid.longitude, grid.latitude)
    Pc, Ps = pts.synthesis_matrix_per_order(3, 1, 8, 'ewh', 3.9860044150e+14, 6.3781363000e+06)
    Rc, Rs = orc.synthesis_matrix_per_order(3, 1, 8, grid.meridians, grid.parallels, ker)
    assert relerr(Pc, Rc) < 1e-12 and relerr(Ps, Rs) < 1e-12
    assert relerr(pts.synthesis_matrix_per_order(0, 0, 8, 'potential', 3.9860044150e+14, 6.3781363000e+06),
                  orc.synthesis_matrix_per_order(0, 0, 8, grid.meridians, grid.parallels, orc.KernelTable('potential', love()))) < 1e-12


def test_round_trip_d60_one_degree(golden):
    """The 5.4 s reference case: d/o 60 from a 1 degree grid."""
    g = golden('g8_analysis')
    anm = inputs.coefficients(21, 60)
    grid = make_pc(anm).to_grid(ga.grid.GeographicGrid(1.0, 1.0), kernel='potential')
    out = grid.to_potential_coefficients(0, 60, kernel='potential')
    assert relerr(out.anm, g['n60_1deg_anm']) < TOL
    assert relerr(out.anm, anm) < TOL                                         # band-limited field is recovered


def oracle_orders(values, area, nmin, N, grid, ker, orders):
    """the oracle on a subset of the (independent) per-order least-squares problems and the mask of the entries they fill: the full
    solve at d/o 127 takes the CPU 30 s per epoch, the orders do not interact (grid.py:779-785)"""
    ref = orc.analysis_regular(values, area, nmin, N, grid.meridians, grid.parallels, ker, orders=orders)
    mask = np.zeros((N + 1, N + 1), dtype=bool)
    for m in orders:
        mask[max(m, nmin):, m] = True
        if m:
            mask[m - 1, max(m, nmin):] = True
    return ref, mask


def relerr_masked(got, ref, mask):
    return float(np.max(np.abs(got - ref)[mask]) / np.max(np.abs(ref)))


@pytest.mark.parametrize('N,nmin,dlon,dlat', [(0, 0, 30, 30), (5, 0, 15, 10), (12, 3, 7.5, 6), (40, 2, 2, 2), (63, 0, 2.5, 2.5),
                                              (127, 4, 1.25, 1.25)])      # 127: beyond the fused transform kernel (fold kernel + GEMMs)
def test_against_oracle_batched(N, nmin, dlon, dlat):
    grid = ga.grid.GeographicGrid(dlon, dlat)
    rng = np.random.default_rng(N + 7)
    vals = rng.standard_normal((5, grid.parallels.size, grid.meridians.size))
    ker = orc.KernelTable('ewh', love())
    colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel('ewh'), N, grid.parallels, 3.9860044150e+14, 6.3781363000e+06,
                                                   grid.semimajor_axis, grid.flattening)
    plan = ga.engine.Plan(N, colat, kn, grid.meridians)
    out = ga.engine.to_host(plan.analysis(vals, grid.area, nmin))
    assert out.shape == (5, N + 1, N + 1)
    if N > 100:            # a sample of the orders (first, last, both parities, around the middle) of two epochs
        orders = [0, 1, 2, 3, N // 2, N // 2 + 1, N - 1, N]
        for e in (0, 4):
            ref, mask = oracle_orders(vals[e].ravel(), grid.area, nmin, N, grid, ker, orders)
            assert relerr_masked(out[e], ref, mask) < TOL
        return
    for e in (0, 4):
        ref = orc.analysis_regular(vals[e].ravel(), grid.area, nmin, N, grid.meridians, grid.parallels, ker)
        assert relerr(out[e], ref) < TOL


@pytest.mark.parametrize('N,nmin,dlon,dlat', [(12, 0, 7.5, 6), (12, 3, 7.5, 6), (40, 0, 2, 2), (127, 0, 1.25, 1.25), (127, 4, 1.25, 1.25)])
def test_output_is_written_in_full(N, nmin, dlon, dlat):
    """The library zero-fills the output only when min_degree > 0 (with min_degree 0 the slots write every entry): whatever the
    allocator hands out as the output block -- here a block that held NaNs a moment ago -- the result is the oracle's, zeros
    below min_degree included."""
    import torch
    grid = ga.grid.GeographicGrid(dlon, dlat)
    rng = np.random.default_rng(N + nmin)
    vals = rng.standard_normal((3, grid.parallels.size, grid.meridians.size))
    ker = orc.KernelTable('potential')
    colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel('potential'), N, grid.parallels, 3.9860044150e+14, 6.3781363000e+06,
                                                   grid.semimajor_axis, grid.flattening)
    plan = ga.engine.Plan(N, colat, kn, grid.meridians)
    plan.analysis(vals, grid.area, nmin)                                       # operators built, caches warm
    for _ in range(3):
        poison = torch.full((3, N + 1, N + 1), float('nan'), dtype=torch.float64, device='cuda')
        del poison                                                             # back to the caching allocator: the next block of this size
        out = ga.engine.to_host(plan.analysis(vals, grid.area, nmin))
        assert np.isfinite(out).all()
    if N > 100:
        ref, mask = oracle_orders(vals[2].ravel(), grid.area, nmin, N, grid, ker, [0, 1, 2, N // 2, N - 1, N])
        assert relerr_masked(out[2], ref, mask) < TOL
    else:
        ref = orc.analysis_regular(vals[2].ravel(), grid.area, nmin, N, grid.meridians, grid.parallels, ker)
        assert relerr(out[2], ref) < TOL
    if nmin > 0:
        assert not out[:, :nmin, :nmin].any()                                  # C_nm, n < min_degree (and the S_nm stored there)


def test_operator_follows_weights_and_meridians():
    """The cached operator is rebuilt when the area weights change between calls (same plan, same min_degree), and grids
    without the four-fold meridian symmetry (here: shifted meridians, odd count) take the unfolded longitude transform."""
    N, nmin = 10, 1
    grid = ga.grid.GeographicGrid(9.0, 6.0)
    ker = orc.KernelTable('potential')
    rng = np.random.default_rng(404)
    for meridians in (grid.meridians, grid.meridians + 0.05, np.linspace(-np.pi, np.pi, 35, endpoint=False) + 0.01):
        vals = rng.standard_normal((3, grid.parallels.size, meridians.size))
        area = rng.uniform(0.5, 1.5, grid.parallels.size * meridians.size)
        colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel('potential'), N, grid.parallels, 3.9860044150e+14, 6.3781363000e+06,
                                                       grid.semimajor_axis, grid.flattening)
        plan = ga.engine.Plan(N, colat, kn, meridians)
        first = ga.engine.to_host(plan.analysis(vals, area, nmin))
        other = area * rng.uniform(0.8, 1.2, area.size)
        second = ga.engine.to_host(plan.analysis(vals, other, nmin))            # optimistic pass, verdict "changed", second pass
        again = ga.engine.to_host(plan.analysis(vals, area, nmin))
        for e in range(3):
            assert relerr(first[e], orc.analysis_regular(vals[e].ravel(), area, nmin, N, meridians, grid.parallels, ker)) < TOL
            assert relerr(second[e], orc.analysis_regular(vals[e].ravel(), other, nmin, N, meridians, grid.parallels, ker)) < TOL
        assert np.array_equal(first, again)
        ga.engine.release_scratch()                                              # the scratch kept between calls comes back on demand
        assert np.array_equal(first, ga.engine.to_host(plan.analysis(vals, area, nmin)))


def test_irregular_grid_analysis():
    lon, lat = inputs.scattered_points(77, 400)
    grid = ga.grid.IrregularGrid(lon, lat)
    anm = inputs.coefficients(78, 8)
    grid.values = orc.synthesis_points(anm, lon, lat, orc.KernelTable('potential'))
    out = grid.to_potential_coefficients(0, 8, kernel='potential')
    assert relerr(out.anm, anm) < 1e-8
    A = grid.synthesis_matrix(0, 8, 'potential')
    assert relerr(A @ orc.ravel_coefficients(anm), grid.values) < 1e-12


def test_full_size_d96_half_degree_240_epochs():
    """The 142 s reference case (d/o 96 from a 0.5 degree grid) at the batch size of the headline: 240 epochs.  Properties that
    need no reference run: a band-limited batch comes back (synthesis -> analysis round trip), the analysis is linear, the
    batch result does not depend on the batch composition; two epochs against the NumPy oracle."""
    import torch
    N, B = 96, 240
    grid = ga.grid.GeographicGrid(0.5, 0.5)
    colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel('ewh'), N, grid.parallels, 3.9860044150e+14, 6.3781363000e+06,
                                                   grid.semimajor_axis, grid.flattening)
    plan = ga.engine.Plan(N, colat, kn, grid.meridians)
    batch = np.stack([inputs.coefficients(8000 + e, N) for e in range(B)])
    batch[:, 0, 0] = batch[:, 1, 0] = batch[:, 1, 1] = batch[:, 0, 1] = 0.0      # C00, C10, C11, S11: the analysis starts at degree 2
    grids = plan.synthesis(batch)
    area = grid.area.reshape(grid.parallels.size, grid.meridians.size)
    back = plan.analysis(grids, area, 2)
    ref = torch.from_numpy(batch).to(back.device)
    assert float((back - ref).abs().max() / ref.abs().max()) < TOL
    # linearity on arbitrary (not band-limited) values and independence of the batch composition
    vals = torch.from_numpy(np.random.default_rng(5).standard_normal((3, grid.parallels.size, grid.meridians.size))).to(back.device)
    x = plan.analysis(vals, area, 0)
    combo = plan.analysis((2.0 * vals[0] - 3.0 * vals[1] + 0.5 * vals[2]).unsqueeze(0), area, 0)[0]
    assert float((combo - (2.0 * x[0] - 3.0 * x[1] + 0.5 * x[2])).abs().max() / x.abs().max()) < TOL
    assert torch.equal(plan.analysis(vals[1:2], area, 0)[0], x[1])
    ker = orc.KernelTable('ewh', love())
    got = ga.engine.to_host(x[2])
    ref, mask = oracle_orders(ga.engine.to_host(vals[2]).ravel(), grid.area, 0, N, grid, ker, [0, 1, 2, 47, 48, 95, 96])
    assert relerr_masked(got, ref, mask) < TOL


def test_window_matrix_golden(golden):
    """Grid.window_matrix (grates/grid.py:449-475) against the reference's output; operators stay on the device in between."""
    g = golden('g18_operators')
    grid = ga.grid.GeographicGrid(5.0, 5.0)
    grid.values = np.random.default_rng(51).uniform(0.0, 1.0, grid.point_count)
    W = grid.window_matrix(1, 8, 'potential')
    assert W.shape == (80, 80) and relerr(W, g['window_5deg_1_8_potential']) < TOL
    gg = ga.grid.GaussGrid(13)
    gg.values = (np.random.default_rng(52).uniform(0.0, 1.0, gg.point_count) > 0.4).astype(float)
    assert relerr(gg.window_matrix(0, 10, 'ewh'), g['window_gauss13_0_10_ewh']) < 1e-11
    # a window of ones is the identity on the band (analysis o synthesis)
    grid.values = np.ones(grid.point_count)
    assert relerr(grid.window_matrix(0, 12, 'potential'), np.eye(169)) < TOL
    # device forms agree with the host API
    assert relerr(ga.engine.to_host(grid.synthesis_matrix_device(1, 8, 'ewh')), grid.synthesis_matrix(1, 8, 'ewh')) == 0.0
    F = ga.engine.to_host(grid.analysis_matrix_device(1, 8, 'potential'))
    assert relerr(F, orc.analysis_matrix_regular(grid.area, 1, 8, grid.meridians, grid.parallels, orc.KernelTable('potential'))) < TOL
    irr = ga.grid.IrregularGrid(*inputs.scattered_points(79, 500))
    irr.values = np.random.default_rng(80).uniform(0.0, 1.0, 500)
    Wi = irr.window_matrix(0, 6, 'potential')
    Fi, Ai = irr.analysis_matrix(0, 6, 'potential'), irr.synthesis_matrix(0, 6, 'potential')
    assert relerr(Wi, (Fi * irr.values) @ Ai) < 1e-11


def test_same_weight_tensor_skips_the_comparison_and_stays_correct():
    """engine.Plan.analysis passes area = NULL ("the weights of the previous call", include/shg.h) when it is handed the very
    device tensor of the previous call, unmodified: no device compare, no host synchronisation.  The results must not depend on
    that, and a tensor that was written to in place (or another tensor, or an operator rebuilt through analysis_matrix) must
    be validated again."""
    import torch
    N = 24
    grid = ga.grid.GeographicGrid(5.0, 5.0)
    colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel('potential'), N, grid.parallels, 3.9860044150e+14, 6.3781363000e+06,
                                                   grid.semimajor_axis, grid.flattening)
    plan = ga.engine.Plan(N, colat, kn, grid.meridians)
    shape = (grid.parallels.size, grid.meridians.size)
    vals = torch.from_numpy(np.random.default_rng(77).standard_normal((5,) + shape)).cuda()
    area = ga.engine.to_device(grid.area.reshape(shape))
    first = plan.analysis(vals, area, 0)
    assert plan._analysis_token is not None
    again = plan.analysis(vals, area, 0)                      # fast path
    assert torch.equal(first, again)
    ker = orc.KernelTable('potential', love())
    assert relerr(ga.engine.to_host(again[3]), orc.analysis_regular(ga.engine.to_host(vals[3]).ravel(), grid.area, 0, N, grid.meridians, grid.parallels, ker)) < TOL
    # in-place change of the weights: the version counter moves, the operators are rebuilt
    area[0:6] *= 3.0
    changed = plan.analysis(vals, area, 0)
    fresh = ga.engine.Plan(N, colat, kn, grid.meridians).analysis(vals, area.clone(), 0)
    assert torch.equal(changed, fresh) and not torch.equal(changed, first)
    assert torch.equal(plan.analysis(vals, area, 0), changed)                     # fast path on the new weights
    # operators rebuilt for other weights through analysis_matrix: the next analysis call validates again
    plan.analysis_matrix(grid.area.reshape(shape), 0)
    assert plan._analysis_token is None
    assert torch.equal(plan.analysis(vals, area, 0), changed)
    # another minimum degree is another set of operators
    assert not torch.equal(plan.analysis(vals, area, 2), changed)
    assert torch.equal(plan.analysis(vals, area, 0), changed)
    # raw-pointer writes of this module into the weights (invisible to torch's version counter) withdraw the trust (advisor r03)
    assert plan._analysis_token is not None
    ga.engine.axpby(0.0, area, 2.0, area)                                         # weights doubled through shg_axpby
    assert plan._analysis_token is None
    doubled = plan.analysis(vals, area, 0)
    assert torch.equal(doubled, ga.engine.Plan(N, colat, kn, grid.meridians).analysis(vals, area.clone(), 0))
    # trusted_weights=False always validates (a write through .data is invisible to everything else), True vouches for the tensor
    area.data[0:3] *= 0.5
    assert torch.equal(plan.analysis(vals, area, 0), doubled)                     # stale operators: the documented trap ...
    halved = plan.analysis(vals, area, 0, trusted_weights=False)                  # ... that the explicit validation avoids
    assert not torch.equal(halved, doubled)
    assert torch.equal(halved, ga.engine.Plan(N, colat, kn, grid.meridians).analysis(vals, area.clone(), 0))
    assert torch.equal(plan.analysis(vals, area, 0, trusted_weights=True), halved)
    # a NULL weight pointer on a plan without cached operators is an error, not a fault
    empty = ga.engine.Plan(N, colat, kn, grid.meridians)
    with pytest.raises(Exception):
        ga.engine._lib.call('shg_analysis', empty._handle, ga.engine._ptr(vals), None, 0, 5, ga.engine._ptr(first), ga.engine._stream())


def test_parity_split_and_its_fallback():
    """North-south parity split of the operator product (csrc/analysis.hip): used on mirror-symmetric parallels with mirror-symmetric
    weights (geographic and Gauss grids), with a dropped part of the operators far below the tolerance; weights that break the symmetry
    -- one hemisphere scaled, or one parallel -- show up as a defect of order one and take the full product.  Both against the oracle,
    min_degree 0 and 3 (the parity of the first row of a slot changes with min_degree), odd and even numbers of rows per slot."""
    ker = orc.KernelTable('potential')
    for grid, N in ((ga.grid.GeographicGrid(3.0, 3.0), 40), (ga.grid.GaussGrid(24), 21)):
        nlat, nlon = grid.parallels.size, grid.meridians.size
        colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel('potential'), N, grid.parallels, 3.9860044150e+14, 6.3781363000e+06,
                                                       grid.semimajor_axis, grid.flattening)
        vals = np.random.default_rng(N).standard_normal((3, nlat, nlon))
        plan = ga.engine.Plan(N, colat, kn, grid.meridians)
        assert plan.analysis_info()['parity_split'] is None
        area = grid.area.reshape(nlat, nlon)
        skew = area.copy()
        skew[: nlat // 2] *= 1.25                                               # northern hemisphere weighted up
        one = area.copy()
        one[nlat - 3] *= 0.5                                                    # a single parallel
        for weights, split in ((area, True), (skew, False), (one, False), (area, True)):
            for nmin in (0, 3):
                out = ga.engine.to_host(plan.analysis(vals, weights, nmin))
                info = plan.analysis_info()
                assert info['parity_split'] is split, (N, nmin, info)
                assert (info['parity_defect'] < 1e-12) == split, info
                for e in (0, 2):
                    ref = orc.analysis_regular(vals[e].ravel(), weights.ravel(), nmin, N, grid.meridians, grid.parallels, ker)
                    assert relerr(out[e], ref) < TOL, (N, nmin, split, e)
    # parallels that are not mirror images: never split
    par = np.linspace(1.5, -1.42, 36)
    mer = ga.grid.GeographicGrid(6.0, 6.0).meridians
    grid = ga.grid.RegularGrid(mer, par)
    colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel('potential'), 12, grid.parallels, 3.9860044150e+14, 6.3781363000e+06,
                                                   grid.semimajor_axis, grid.flattening)
    plan = ga.engine.Plan(12, colat, kn, mer)
    vals = np.random.default_rng(3).standard_normal((2, par.size, mer.size))
    area = np.cos(par)[:, np.newaxis] * np.ones((par.size, mer.size))
    out = ga.engine.to_host(plan.analysis(vals, area, 0))
    assert plan.analysis_info()['parity_split'] is False
    assert relerr(out[1], orc.analysis_regular(vals[1].ravel(), area.ravel(), 0, 12, mer, par, ker)) < 1e-11      # (an off-centre cap: less well conditioned)
