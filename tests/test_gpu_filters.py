"""
GPU parity of the filters (rows a16-a19 of SURVEY.md 8a): order-wise block filter (DDK family) and dense matrix
filter against the golden vectors of the reference.  Tolerance: 1e-13 relative.
"""

import numpy as np
import pytest

import grates_amd as ga
import inputs
from conftest import relerr
from oracle import shg_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-13


def make_pc(anm):
    gf = ga.gravityfield.PotentialCoefficients()
    gf.anm = anm.copy()
    return gf


@pytest.mark.parametrize('nmax,ngf', [(20, 20), (120, 120), (120, 96)])
def test_orderwise_filter_golden(golden, nmax, ngf):
    g = golden('g10_filter')
    flt = ga.filter.OrderWiseFilter(inputs.orderwise_random_blocks(42, nmax))
    gf = make_pc(inputs.coefficients(43, ngf))
    out = flt.filter(gf)
    assert out is not gf and out.max_degree == ngf
    np.testing.assert_array_equal(gf.anm, inputs.coefficients(43, ngf))
    assert relerr(out.anm, g['orderwise_{0}_{1}'.format(nmax, ngf)]) < TOL
    np.testing.assert_array_equal(out.anm[0:2, 0:2], gf.anm[0:2, 0:2])       # degrees 0 and 1 restored


def test_orderwise_errors_and_batch():
    flt = ga.filter.OrderWiseFilter(inputs.orderwise_random_blocks(42, 20))
    with pytest.raises(ValueError):
        flt.filter(make_pc(inputs.coefficients(1, 21)))
    with pytest.raises(TypeError):
        flt.filter(np.zeros((3, 3)))
    blocks = inputs.orderwise_random_blocks(42, 20)
    batch = np.stack([inputs.coefficients(300 + e, 17) for e in range(37)])
    out = ga.engine.to_host(flt.filter_batch(batch))
    for e in (0, 13, 36):
        assert relerr(out[e], orc.orderwise_filter(batch[e], blocks)) < TOL
    # the block filter and its dense matrix are the same operator on degrees >= 2 (input without degrees 0, 1)
    W = flt.matrix(2, 17)
    x_in = batch[5].copy()
    x_in[0:2, 0:2] = 0.0
    y = ga.engine.to_host(flt.filter_batch(x_in[np.newaxis]))[0]
    assert relerr(orc.ravel_coefficients(y, 2, 17), W @ orc.ravel_coefficients(x_in, 2, 17)) < TOL


@pytest.mark.parametrize('nmax,ngf,B', [(20, 20, 5), (120, 120, 37), (120, 96, 16), (30, 7, 33), (12, 0, 3), (12, 1, 2)])
def test_order_major_series(golden, nmax, ngf, B):
    """A series kept on the device in order-major layout: conversion from / to the reference arrays is exact, the filter on whole
    per-order matrices gives the values of OrderWiseFilter.filter epoch by epoch (golden fixture where one exists, oracle, and the
    kernel on the reference layout), degrees 0 and 1 restored, padding epochs never leak."""
    import torch
    blocks = inputs.orderwise_random_blocks(42, nmax)
    flt = ga.filter.OrderWiseFilter(blocks)
    batch = np.stack([inputs.coefficients(43 + e, ngf) for e in range(B)])
    series = ga.engine.OrderMajorSeries.from_batch(batch)
    assert series.max_degree == ngf and series.epochs == B and series.padded_epochs % 32 == 0 and series.data.shape[0] == (ngf + 1) ** 2
    np.testing.assert_array_equal(ga.engine.to_host(series.to_batch()), batch)
    # row order: slot 0 = order 0 cosine (degrees 0 .. N), then order 1 cosine, order 1 sine, ...
    host = ga.engine.to_host(series.data)
    np.testing.assert_array_equal(host[0:ngf + 1, 0:B], batch[:, :, 0].T)
    if ngf >= 1:
        np.testing.assert_array_equal(host[ngf + 1:2 * ngf + 1, 0:B], batch[:, 1:, 1].T)          # C_n1, n = 1 .. N
        np.testing.assert_array_equal(host[2 * ngf + 1:3 * ngf + 1, 0:B], batch[:, 0, 1:].T)      # S_n1 at [0][n]
    series.data[:, B:] = 1e300                           # whatever sits in the padding epochs stays there
    out = flt.filter_series(series)
    assert out is not series and out.max_degree == ngf and out.epochs == B
    got = ga.engine.to_host(out.to_batch())
    np.testing.assert_array_equal(ga.engine.to_host(series.to_batch()), batch)                    # input untouched
    for e in sorted({0, B // 2, B - 1}):
        assert relerr(got[e], orc.orderwise_filter(batch[e], blocks)) < TOL
    assert relerr(got, ga.engine.to_host(flt.filter_batch(batch))) < TOL
    np.testing.assert_array_equal(got[:, 0:2, 0:2], batch[:, 0:2, 0:2])                           # degrees 0 and 1 restored
    if (nmax, ngf) in ((20, 20), (120, 120), (120, 96)):
        assert relerr(got[0], golden('g10_filter')['orderwise_{0}_{1}'.format(nmax, ngf)]) < TOL
    if ngf > 0:
        with pytest.raises(ValueError):              # blocks of a lower degree than the series
            ga.filter.OrderWiseFilter(inputs.orderwise_random_blocks(42, ngf - 1)).filter_series(series)


def test_order_major_series_to_grid():
    """DDK-type filter -> synthesis without leaving the device layout: series of degree 120 filtered, synthesised by plans of degree 96
    and 120 (rotation-folded and 4-fold kernels) and by a plan that reads the reference arrays only."""
    blocks = inputs.orderwise_random_blocks(7, 120)
    flt = ga.filter.OrderWiseFilter(blocks)
    B = 9
    batch = np.stack([inputs.coefficients(900 + e, 120) for e in range(B)])
    filtered = flt.filter_series(ga.engine.OrderMajorSeries.from_batch(batch))
    reference = ga.engine.to_host(flt.filter_batch(batch))
    ker = orc.KernelTable('potential')
    for N, step, paths in ((96, 0.25, ('auto', 'fused')), (120, 0.5, ('auto',)), (60, 3.0, ('auto', 'staged'))):
        grid = ga.grid.GeographicGrid(step, step)
        colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel('potential'), N, grid.parallels, 3.9860044150e+14, 6.3781363000e+06,
                                                       grid.semimajor_axis, grid.flattening)
        plan = ga.engine.Plan(N, colat, kn, grid.meridians)
        want = ga.engine.to_host(plan.synthesis(reference[:, :N + 1, :N + 1].copy()))
        check = orc.synthesis_regular(reference[0][:N + 1, :N + 1], grid.meridians, grid.parallels, ker)
        assert relerr(want[0], check) < TOL
        for path in paths:
            try:
                plan.set_path(path)
            except ga._lib.ShgError:
                continue
            got = ga.engine.to_host(plan.synthesis(filtered))
            assert got.shape == want.shape and relerr(got, want) < TOL, (N, path)
    colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel('potential'), 121, grid.parallels, 3.9860044150e+14, 6.3781363000e+06,
                                                   grid.semimajor_axis, grid.flattening)
    with pytest.raises(ValueError):                  # a plan of a higher degree than the series
        ga.engine.Plan(121, colat, kn, grid.meridians).synthesis(filtered)


def test_ddk_from_synthetic_normals(golden):
    """DDK construction + application (config 3 operator) with synthetic SPD normals; the published blocks are absent."""
    g = golden('g10_filter')
    normals = inputs.orderwise_normal_blocks(44, 20)
    ga.filter.DDKGeneric._blocked_normals = staticmethod(lambda: normals)
    try:
        gf = make_pc(inputs.coefficients(45, 20))
        for level in (5, 3):
            ddk = ga.filter.DDK(level)
            assert relerr(ddk.filter(gf).anm, g['ddk{0}_n20'.format(level)]) < 1e-12
            assert relerr(ddk.matrix(2, 20), g['ddk{0}_n20_matrix'.format(level)]) < 1e-12
            assert relerr(ga.filter.DDKGeneric(level).filter(gf).anm, g['ddkgeneric{0}_n20'.format(level)]) < 1e-12
    finally:
        ga.filter.DDKGeneric._blocked_normals = staticmethod(lambda: ga.data.ddk_normal_blocks())


def test_general_matrix_filter(golden):
    g = golden('g10_filter')
    W = np.random.default_rng(46).standard_normal((21 * 21 - 4, 21 * 21 - 4)) / 21
    gm = ga.filter.GeneralMatrix(W, 2, 20)
    out = gm.filter(make_pc(inputs.coefficients(47, 20)))
    assert relerr(out.anm, g['general_2_20_n20']) < TOL
    out14 = gm.filter(make_pc(inputs.coefficients(47, 14)))
    assert out14.anm.shape == g['general_2_20_n14'].shape
    assert relerr(out14.anm, g['general_2_20_n14']) < TOL
    batch = np.stack([inputs.coefficients(400 + e, 20) for e in range(9)])
    res = ga.engine.to_host(gm.filter_batch(batch))
    for e in range(9):
        assert relerr(res[e], orc.general_matrix_filter(batch[e], W, 2, 20)) < TOL


def test_orderwise_filter_above_degree_127():
    """Blocks up to d/o 180: the LDS stage of the block kernel holds 8 orders x 16 epochs up to degree 143, 4 orders up to 295 and 2
    up to 591 (beyond that the kernel refuses); a field of lower degree than the blocks."""
    nmax, T = 180, 70
    normals = inputs.orderwise_normal_blocks(48, nmax)
    blocks = orc.ddk_blocks(normals, 5)
    flt = ga.filter.OrderWiseFilter(blocks)
    batch = np.stack([inputs.coefficients(650 + e, nmax) for e in range(T)])
    out = ga.engine.to_host(flt.filter_batch(batch))
    for e in (0, 33, 69):
        assert relerr(out[e], orc.orderwise_filter(batch[e], blocks)) < TOL
    low = np.stack([inputs.coefficients(660 + e, 150) for e in range(3)])       # a field of lower degree than the blocks
    out_low = ga.engine.to_host(flt.filter_batch(low))
    assert relerr(out_low[2], orc.orderwise_filter(low[2], blocks)) < TOL


def test_config3_ddk5_time_series_d120():
    """BASELINE config 3 at its stated size: DDK5-type filter applied to a d/o-120 series of 240 epochs, block form vs
    full-matrix multiply (14637 x 14637 x 240), both against the oracle on sample epochs."""
    nmax, T = 120, 240
    normals = inputs.orderwise_normal_blocks(44, nmax)
    blocks = orc.ddk_blocks(normals, 5)
    flt = ga.filter.OrderWiseFilter(blocks)
    batch = np.stack([inputs.coefficients(600 + e, nmax) for e in range(T)])
    batch[:, 0:2, 0:2] = 0.0          # GRACE-type series carry no degree 0 / 1; the dense form ignores them as inputs
    out_blocks = ga.engine.to_host(flt.filter_batch(batch))
    for e in (0, 11, 127, 239):
        assert relerr(out_blocks[e], orc.orderwise_filter(batch[e], blocks)) < TOL
    dense = ga.filter.GeneralMatrix(flt.matrix(2, nmax), 2, nmax)             # 14637 x 14637 full normal-type matrix
    out_dense = ga.engine.to_host(dense.filter_batch(batch))
    assert relerr(out_dense, out_blocks) < 1e-12


def test_order_major_layout_above_degree_510():
    """Degrees from 511 on: the row stage of the layout kernels (16 epochs x (N + 2) doubles) exceeds the 64 KB a launch gets without the
    opt-in attribute.  Round 5 launched it regardless, and shg_orderwise_filter -- which takes the order-major kernels for 64 epochs and
    more -- returned the launch error instead of falling back (advisor, round 5).  Pack / unpack round trip, the filter through both
    layouts (blocks 0.5 I: every coefficient of degree >= 2 halved, degrees 0 and 1 kept), ragged epoch count."""
    import torch
    N, B = 520, 70
    rng = np.random.default_rng(520)
    batch = torch.from_numpy(rng.standard_normal((B, N + 1, N + 1))).cuda()
    series = ga.engine.OrderMajorSeries.from_batch(batch)
    assert series.padded_epochs == 96 and torch.equal(series.to_batch(), batch)
    flt = ga.filter.OrderWiseFilter([0.5 * np.eye(N + 1)] + [0.5 * np.eye(N + 1 - m) for m in range(1, N + 1) for _ in (0, 1)])
    out = flt.filter_batch(batch)                                 # 70 epochs: pack -> products -> unpack
    small = flt.filter_batch(batch[0:3])                          # 3 epochs: the kernel on the reference layout
    want = 0.5 * batch
    want[:, 0:2, 0:2] = batch[:, 0:2, 0:2]
    assert torch.equal(out, want) and torch.equal(small, want[0:3])
    assert torch.equal(flt.filter_series(series).to_batch(), want)


def _series(count, max_degree, seed=700):
    import datetime as dt
    fields = []
    for e in range(count):
        gf = make_pc(inputs.coefficients(seed + e, max_degree))
        gf.epoch = dt.datetime(2005, 1, 1) + dt.timedelta(days=30 * e)
        fields.append(gf)
    return ga.gravityfield.TimeSeries(fields)


def test_device_resident_time_series_chain(golden):
    """The reference API carrying the device-resident series: `filter.*.filter(TimeSeries)` filters all epochs in
    one call on an engine.OrderMajorSeries and returns a TimeSeries that stays on the device, `to_grid` reads it through
    shg_synthesis_om.  The chained result is BIT-IDENTICAL to the per-epoch path of the reference API -- filter(field) and
    to_grid(field) one epoch at a time --, for the order-wise (DDK-type) filter and the Gaussian; the golden values of the reference
    at d/o 120; the dense filter to rounding.  Handing out fields ends the residency, nothing else does."""
    g = golden('g10_filter')
    nmax, T = 120, 70                                        # (>= 64 epochs: the batch entry point takes the order-major kernels too)
    ts = _series(T, nmax)
    assert not ts.on_device
    flt = ga.filter.OrderWiseFilter(inputs.orderwise_random_blocks(42, nmax))
    filtered = flt.filter(ts)
    assert isinstance(filtered, ga.gravityfield.TimeSeries) and filtered.on_device and ts.on_device and len(filtered) == T
    assert filtered.epochs() == ts.epochs()
    grid = ga.grid.GeographicGrid(3.0, 3.0)
    grids = ga.engine.to_host(filtered.to_grid(grid, kernel='ewh', as_tensor=True))
    assert filtered.on_device                                # to_grid, to_array, epochs, len: no fields handed out
    arr = filtered.to_array()
    for e in (0, 1, 33, T - 1):
        single = flt.filter(ts[e])
        assert np.array_equal(arr[e], single.values), e                                   # filter: bit-identical to the per-epoch call
        assert np.array_equal(grids[e], single.to_grid(grid, kernel='ewh').value_array), e   # ... and so is the grid
    assert not ts.on_device                                  # ts[e] handed fields out
    gold = make_pc(inputs.coefficients(43, nmax))
    gold.epoch = ts.epochs()[0]
    one = flt.filter(ga.gravityfield.TimeSeries([gold]))
    assert relerr(one[0].anm, g['orderwise_120_120']) < TOL
    # Gaussian on the series: the same products as the per-epoch call
    gauss = ga.filter.Gaussian(300)
    ts = _series(9, 60, seed=900)
    smooth = gauss.filter(ts)
    assert smooth.on_device
    sm = smooth.to_array()
    for e in (0, 8):
        assert np.array_equal(sm[e], gauss.filter(_series(9, 60, seed=900)[e]).values)
    # filters compose on the device; a grid list comes back with the epochs
    both = gauss.filter(ga.filter.OrderWiseFilter(inputs.orderwise_random_blocks(42, 60)).filter(ts))
    out = both.to_grid(ga.grid.GeographicGrid(10.0, 10.0), kernel='potential')
    assert len(out) == 9 and out[3].epoch == ts.epochs()[3] and np.isfinite(out[3].values).all()
    # dense filter on the series: one product with the permuted matrix
    blocks20 = inputs.orderwise_random_blocks(42, 20)
    dense = ga.filter.GeneralMatrix(ga.filter.OrderWiseFilter(blocks20).matrix(2, 20), 2, 20)
    ts20 = _series(5, 20, seed=950)
    d_series = dense.filter(ts20).to_array()
    d_batch = ga.engine.to_host(dense.filter_batch(ts20.to_coefficient_batch()))
    for e in range(5):
        assert relerr(d_series[e], orc.ravel_coefficients(d_batch[e], 0, 20)) < TOL
        assert np.array_equal(d_series[e][0:4], ts20.to_array()[e][0:4])                # degrees 0 and 1 restored
    with pytest.raises(TypeError):
        flt.filter(np.zeros((3, 3)))
    # fields of different degrees: filtered field by field like the reference's call (a padded series would fill the padding)
    mixed = ga.gravityfield.TimeSeries([_series(1, 12, seed=960)[0], _series(2, 20, seed=961)[1]])
    out = ga.filter.OrderWiseFilter(blocks20).filter(mixed)
    assert not mixed.uniform_degree and [f.max_degree for _, f in out.items()] == [12, 20]
    assert np.array_equal(out[0].anm, ga.filter.OrderWiseFilter(blocks20).filter(mixed[0]).anm)


def test_device_resident_time_series_bookkeeping(golden):
    """to_array / to_coefficient_batch / copy / scaling / detrend of a device-resident series equal those of the list of fields
    (g10 `timeseries_array` pins the reference's to_array on the same inputs as tests/test_host_logic.py)."""
    ts = _series(12, 15, seed=40)
    host_array = ts.to_array()
    dev = ga.gravityfield.TimeSeries.from_series(ts.to_coefficient_batch(), ts.epochs())
    assert dev.on_device and len(dev) == 12
    assert np.array_equal(dev.to_array(), host_array)
    assert np.array_equal(dev.to_coefficient_batch(), ts.to_coefficient_batch())
    assert np.array_equal((dev * 2.5).to_array(), (ts * 2.5).to_array())
    assert np.array_equal(dev.copy().to_array(), host_array) and dev.on_device
    both = dev + dev * 0.5
    assert both.on_device and np.array_equal(both.to_array(), (ts + ts * 0.5).to_array())
    assert np.array_equal((dev - dev).to_array(), np.zeros_like(host_array)) and dev.on_device
    basis = [ga.utilities.Polynomial(1, ts.epochs()[0])] if hasattr(ga.utilities, 'Polynomial') else None
    if basis is not None:
        a, b = ts.copy(), dev.copy()
        trend_host, trend_dev = a.detrend(basis), b.detrend(basis)
        assert b.on_device and relerr(trend_dev, trend_host) < 1e-12
        assert relerr(b.to_array(), a.to_array()) < 1e-9        # (residuals: differences of nearly equal numbers, as in g14)
    # fields handed out carry epoch, constants and coefficients; the device copy is rebuilt on demand
    first = dev[0]
    assert not dev.on_device and first.epoch == ts.epochs()[0] and np.array_equal(first.anm, ts[0].anm)
    first.anm[2, 0] = 7.0
    assert dev.to_device() is not None and dev.to_array()[0][4] == 7.0


# ------------------------------------------------------------------------------------------------ SURVEY 8(f) rank 2
def test_vdk_and_filter_kernels_golden(golden):
    """VDK (dense decorrelation filter from a full normal matrix), AnisotropicKernel and FilterKernel against the
    reference's outputs (tests/golden/g12_filter_kernel.npz) and the oracle."""
    g = golden('g12_filter_kernel')
    nmin, nmax = 2, 12
    P = (nmax + 1) ** 2 - nmin ** 2
    normals = inputs.spd_covariance(70, P, scale=1e20)
    vdk = ga.filter.VDK(normals, nmin, nmax, 1e18, 2.0)
    assert relerr(vdk.matrix(nmin, nmax), g['vdk_matrix']) < 1e-11
    gf = ga.gravityfield.PotentialCoefficients()
    gf.anm = inputs.coefficients(71, 12)
    assert relerr(vdk.filter(gf).anm, g['vdk_filtered_n12']) < 1e-11
    with pytest.raises(ValueError):
        ga.filter.VDK(normals[0:5, 0:5], nmin, nmax, 1e18, 2.0)

    src_lon, src_lat = np.deg2rad(13.0), np.deg2rad(47.5)
    ev_lon = np.deg2rad(np.linspace(-20.0, 50.0, 9))
    ev_lat = np.deg2rad(np.linspace(30.0, 65.0, 6))
    pts_lon, pts_lat = np.meshgrid(ev_lon, ev_lat)
    gm = ga.filter.GeneralMatrix(g['vdk_matrix'], nmin, nmax)
    for name in ('potential', 'ewh'):
        fk = ga.filter.FilterKernel(gm, nmin, nmax, input_kernel=name)
        ref = g['filterkernel_{0}_points'.format(name)]
        assert relerr(fk.evaluate(src_lon, src_lat, pts_lon.ravel(), pts_lat.ravel()), ref) < 1e-12
        grid = fk.evaluate_grid(src_lon, src_lat, ev_lon, ev_lat)
        assert grid.shape == (ev_lat.size, ev_lon.size)
        assert relerr(grid.ravel(), ref) < 1e-12
    K = np.random.default_rng(72).standard_normal((P, P)) / nmax
    ak = ga.kernel.AnisotropicKernel(K, nmin, nmax)
    assert relerr(ak.evaluate_grid(src_lon, src_lat, ev_lon, ev_lat), g['anisotropic_grid']) < 1e-12
    assert relerr(ak.evaluate(src_lon, src_lat, ev_lon, ev_lat[0:1].repeat(ev_lon.size)), g['anisotropic_points']) < 1e-12
    fo = ga.filter.FilterKernel(ga.filter.OrderWiseFilter(inputs.orderwise_random_blocks(73, nmax)), nmin, nmax)
    assert relerr(fo.evaluate(src_lon, src_lat, pts_lon.ravel(), pts_lat.ravel()), g['filterkernel_orderwise_points']) < 1e-12
    with pytest.raises(ValueError):
        ga.kernel.AnisotropicKernel(K[0:4], nmin, nmax)
    with pytest.raises(ValueError):
        ak.evaluate(src_lon, src_lat, ev_lon, ev_lat)


def test_anisotropic_kernel_d60_against_oracle():
    """The same evaluation at d/o 60 (P = 3717, beyond the fixture size) against the oracle: Gaussian-filter kernel
    (isotropic, so the kernel only depends on the spherical distance) and a dense random mapping."""
    nmin, nmax = 2, 60
    P = (nmax + 1) ** 2 - nmin ** 2
    src_lon, src_lat = 0.3, -0.4
    lon = np.linspace(-np.pi, np.pi, 24, endpoint=False) + 0.01
    lat = np.linspace(-1.4, 1.4, 15)
    K = np.random.default_rng(5).standard_normal((P, P)) / nmax
    ak = ga.kernel.AnisotropicKernel(K, nmin, nmax)
    ref = orc.anisotropic_kernel_grid(K, nmin, nmax, src_lon, src_lat, lon, lat)
    assert relerr(ak.evaluate_grid(src_lon, src_lat, lon, lat), ref) < 1e-12
    fk = ga.filter.FilterKernel(ga.filter.Gaussian(400), nmin, nmax)
    vals = fk.evaluate_grid(src_lon, src_lat, lon, lat)
    ref = orc.anisotropic_kernel_grid(orc.gaussian_matrix(400, nmin, nmax), nmin, nmax, src_lon, src_lat, lon, lat)
    assert relerr(vals, ref) < 1e-12
    # isotropy: equal spherical distance -> equal kernel value
    mirror = fk.evaluate(src_lon, src_lat, np.array([src_lon + 0.2, src_lon - 0.2]), np.array([src_lat, src_lat]))
    assert abs(mirror[0] - mirror[1]) < 1e-10 * np.abs(vals).max()
