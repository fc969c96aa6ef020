"""
GPU parity of the synthesis path (rows a8-a11 of SURVEY.md 8a): PotentialCoefficients.to_grid and the batched
engine against the reference's golden vectors and against the CPU oracle on seeded inputs.
Tolerance (north_star / SURVEY.md 8d): max|d| / max|ref| <= 1e-12.
"""

import datetime

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


def make_pc(anm, **kw):
    gf = ga.gravityfield.PotentialCoefficients(**kw)
    gf.anm = anm.copy()
    return gf


def test_config1_gaussian_then_grid(golden):
    """BASELINE config 1: d/o 60, 300 km Gaussian, 1 degree grid, ewh."""
    g = golden('g7_synthesis')
    gf = make_pc(inputs.coefficients(1000, 60))
    filtered = ga.filter.Gaussian(300).filter(gf)
    np.testing.assert_allclose(filtered.anm, g['c1_filtered_anm'], rtol=1e-15, atol=0)
    grid_in = ga.grid.GeographicGrid(1.0, 1.0)
    out = filtered.to_grid(grid_in, kernel='ewh')
    assert type(out) is ga.grid.GeographicGrid and out is not grid_in and grid_in.values is None
    assert out.value_array.shape == (180, 360)
    assert relerr(out.value_array, g['c1_grid']) < TOL


def test_config2_unit_two_epochs(golden):
    """d/o 96 -> 0.25 degree, the unit of the headline metric, against reference samples."""
    g = golden('g7_synthesis')
    grid = ga.grid.GeographicGrid(0.25, 0.25)
    batch = np.stack([inputs.coefficients(1000 + e, 96) for e in (0, 1)])
    values = ga.engine.to_host(ga.gravityfield.synthesize(batch, grid, 'ewh'))
    assert values.shape == (2, 720, 1440)
    for e in (0, 1):
        assert relerr(values[e][::9, ::11], g['c2_sample_{0}'.format(e)]) < TOL
        assert relerr(values[e].sum(axis=1), g['c2_rowsum_{0}'.format(e)]) < 1e-11
        assert abs(np.abs(values[e]).max() - g['c2_maxabs_{0}'.format(e)][0]) < TOL * g['c2_maxabs_{0}'.format(e)][0]
    info = ga.engine.cached_plan(96, *_tables(grid, 96, 'ewh')).info()
    assert info['fourfold_symmetry'] and info['nlat'] == 720 and info['nlon'] == 1440


def _tables(grid, N, kernel, GM=3.9860044150e+14, R=6.3781363000e+06):
    colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel(kernel), N, grid.parallels, GM, R, grid.semimajor_axis, grid.flattening)
    return colat, kn, grid.meridians


def test_kernels_grids_and_constants(golden):
    g = golden('g7_synthesis')
    out = make_pc(inputs.coefficients(7, 60)).to_grid(ga.grid.GaussGrid(61), kernel='potential')
    assert type(out) is ga.grid.GaussGrid
    assert relerr(out.value_array, g['gauss61_potential']) < TOL
    gf = make_pc(inputs.coefficients(8, 30))
    for name in ('geoid', 'obp', 'potential', 'ewh'):
        out = gf.to_grid(ga.grid.GeographicGrid(5.0, 5.0), kernel=name)
        assert relerr(out.value_array, g['n30_5deg_' + name]) < TOL, name
    gf = make_pc(inputs.coefficients(9, 30), GM=3.986004418e14, R=6378137.0)
    assert relerr(gf.to_grid(ga.grid.GeographicGrid(5.0, 5.0), kernel='ewh').value_array, g['n30_5deg_gmr']) < TOL


def test_general_path_asymmetric_grid(golden):
    """Meridians without the 4-fold symmetry take the general longitude stage."""
    g = golden('g7_synthesis')
    grid = ga.grid.RegularGrid(g['asym_meridians'], g['asym_parallels'])
    out = make_pc(inputs.coefficients(13, 25)).to_grid(grid, kernel='potential')
    assert relerr(out.value_array, g['asym_potential']) < TOL
    plan = ga.engine.cached_plan(25, *_tables(grid, 25, 'potential'))
    assert not plan.info()['fourfold_symmetry']


@pytest.mark.parametrize('N,dlon,dlat', [(0, 30, 30), (1, 30, 30), (2, 90, 45), (3, 10, 20), (17, 4.5, 3), (33, 2, 2), (64, 1, 2.5), (120, 1, 1)])
def test_against_oracle_shapes(N, dlon, dlat):
    """Seeded inputs, many degree / grid shapes (ragged tiles, tiny grids, nlon % 4 != 0 -> general path)."""
    grid = ga.grid.GeographicGrid(dlon, dlat)
    anm = inputs.coefficients(100 + N, N)
    ref = orc.synthesis_regular(anm, grid.meridians, grid.parallels, orc.KernelTable('ewh', love()))
    out = make_pc(anm).to_grid(grid, 'ewh')
    assert relerr(out.value_array, ref) < TOL


def test_nlon_not_multiple_of_four_and_odd_sizes():
    pot = orc.KernelTable('potential')
    for nlon, nlat in ((6, 3), (10, 7), (37, 19), (50, 1), (1, 5)):
        mer = np.linspace(-np.pi, np.pi, nlon, endpoint=False) + np.pi / nlon
        par = np.linspace(1.5, -1.5, nlat) if nlat > 1 else np.array([0.3])
        grid = ga.grid.RegularGrid(mer, par)
        anm = inputs.coefficients(nlon, 9)
        ref = orc.synthesis_regular(anm, mer, par, pot)
        assert relerr(make_pc(anm).to_grid(grid, 'potential').value_array, ref) < TOL


def test_batch_sizes_and_chunking():
    """Batches that are not multiples of the epoch tile / pass size; results independent of the chunking."""
    grid = ga.grid.GeographicGrid(3.0, 3.0)
    N = 40
    ker = orc.KernelTable('ewh', love())
    batch = np.stack([inputs.coefficients(500 + e, N) for e in range(21)])
    ref = np.stack([orc.synthesis_regular(batch[e], grid.meridians, grid.parallels, ker) for e in range(21)])
    plan = ga.engine.Plan(N, *_tables(grid, N, 'ewh'))
    for chunk in (16, 1, 5, 8, 64):
        plan.set_chunk(chunk)
        for B in (1, 7, 8, 9, 21):
            out = ga.engine.to_host(plan.synthesis(batch[0:B]))
            assert out.shape == (B, 60, 120)
            assert relerr(out, ref[0:B]) < TOL, (chunk, B)
    assert plan.synthesis(batch[0:0]).shape == (0, 60, 120)
    single = ga.engine.to_host(plan.synthesis(batch[3]))
    assert single.shape == (60, 120) and relerr(single, ref[3]) < TOL
    with pytest.raises(ValueError):
        plan.synthesis(np.zeros((2, N, N)))


def test_linearity_and_zero_input():
    grid = ga.grid.GeographicGrid(2.0, 2.0)
    a, b = inputs.coefficients(1, 50), inputs.coefficients(2, 50)
    va = make_pc(a).to_grid(grid, 'potential').value_array
    vb = make_pc(b).to_grid(grid, 'potential').value_array
    vab = make_pc(2.0 * a - 3.0 * b).to_grid(grid, 'potential').value_array
    assert relerr(vab, 2.0 * va - 3.0 * vb) < TOL
    assert np.all(make_pc(np.zeros((51, 51))).to_grid(grid, 'potential').value_array == 0.0)
    # degree-0 only: constant GM/R * (R/r) on every parallel
    c = np.zeros((51, 51))
    c[0, 0] = 1.0
    v = make_pc(c).to_grid(grid, 'potential').value_array
    r = orc.geocentric_radius(grid.parallels)
    np.testing.assert_allclose(v, np.repeat((3.9860044150e+14 / r)[:, None], 180, axis=1), rtol=1e-14)


def test_time_series_to_grid():
    series = []
    for e in range(5):
        gf = make_pc(inputs.coefficients(70 + e, 20))
        gf.epoch = datetime.datetime(2012, 1 + e, 1)
        series.append(gf)
    ts = ga.gravityfield.TimeSeries(series)
    grid = ga.grid.GeographicGrid(10, 10)
    grids = ts.to_grid(grid, 'ewh')
    assert len(grids) == 5 and all(type(x) is ga.grid.GeographicGrid for x in grids)
    ker = orc.KernelTable('ewh', love())
    for gf, out in zip(series, grids):
        assert out.epoch == gf.epoch
        assert relerr(out.value_array, orc.synthesis_regular(gf.anm, grid.meridians, grid.parallels, ker)) < TOL
    t = ts.to_grid(grid, 'ewh', as_tensor=True)
    assert t.is_cuda and tuple(t.shape) == (5, 18, 36)


def test_full_size_properties_config2():
    """BASELINE config 2 at full size (240 x d/o 96 -> 0.25 degree): size-independent properties."""
    import torch
    grid = ga.grid.GeographicGrid(0.25, 0.25)
    B, N = 240, 96
    rng = np.random.default_rng(2024)
    batch = torch.from_numpy(rng.standard_normal((B, N + 1, N + 1)) * 1e-10).cuda()
    out = ga.gravityfield.synthesize(batch, grid, 'ewh')
    assert tuple(out.shape) == (B, 720, 1440) and bool(torch.isfinite(out).all())
    # (1) a few epochs against the oracle
    ker = orc.KernelTable('ewh', love())
    for e in (0, 113, 239):
        ref = orc.synthesis_regular(batch[e].cpu().numpy(), grid.meridians, grid.parallels, ker)
        assert relerr(out[e].cpu().numpy(), ref) < TOL
    # (2) linearity across the batch: synth(sum_b w_b x_b) == sum_b w_b synth(x_b)
    w = torch.from_numpy(rng.standard_normal(B)).cuda()
    combo = ga.gravityfield.synthesize((w[:, None, None] * batch).sum(dim=0, keepdim=True), grid, 'ewh')[0]
    lin = (w[:, None, None] * out).sum(dim=0)
    assert float((combo - lin).abs().max() / lin.abs().max()) < 1e-11
    # (3) permutation of the batch permutes the output (no cross-epoch leakage, chunk boundaries)
    perm = torch.randperm(B, device='cuda')
    out_p = ga.gravityfield.synthesize(batch[perm], grid, 'ewh')
    assert bool((out_p == out[perm]).all())
    # (4) mean over longitude only sees order 0:  mean_j V[i, j] = sum_n kn P_n0 C_n0
    zonal = batch.clone()
    idx = torch.arange(N + 1, device='cuda')
    mask = torch.zeros((N + 1, N + 1), dtype=torch.bool, device='cuda')
    mask[:, 0] = True
    zon = ga.gravityfield.synthesize(zonal * mask, grid, 'ewh')
    assert float((out.mean(dim=2) - zon[:, :, 0]).abs().max() / out.abs().max()) < 1e-12


@pytest.mark.parametrize('N,dlon,dlat', [(0, 30, 30), (1, 30, 30), (2, 90, 45), (3, 10, 20), (17, 4.5, 3), (64, 1, 2.5), (96, 0.5, 0.5), (110, 1, 1), (126, 1.5, 1.5)])
def test_fused_and_staged_paths(N, dlon, dlat):
    """Both synthesis paths (single fused kernel / three staged kernels) against the oracle, ragged batch sizes."""
    grid = ga.grid.GeographicGrid(dlon, dlat)
    ker = orc.KernelTable('ewh', love())
    B = 7
    batch = np.stack([inputs.coefficients(900 + N * 10 + e, N) for e in range(B)])
    ref = np.stack([orc.synthesis_regular(batch[e], grid.meridians, grid.parallels, ker) for e in range(B)])
    plan = ga.engine.Plan(N, *_tables(grid, N, 'ewh'))
    assert plan.info()['fourfold_symmetry'] and plan.info()['fused']
    assert plan.info()['north_south_symmetry'] == (grid.parallels.size % 2 == 0)
    fused32_ok = plan.info()['north_south_symmetry'] and plan.info()['k_slots'] * 48 * 8 <= 160 * 1024
    for path in ('fused', 'fused32', 'staged', 'rot', 'auto'):
        if (path == 'fused32' and not fused32_ok) or (path == 'rot' and not plan.info()['rotation_symmetry']):
            with pytest.raises(ga._lib.ShgError):
                plan.set_path(path)
            continue
        plan.set_path(path)
        assert plan.info()['fused'] == (path != 'staged')
        for nb in (1, 3, 4, 5, 7):
            out = ga.engine.to_host(plan.synthesis(batch[0:nb]))
            assert relerr(out, ref[0:nb]) < TOL, (path, nb)


@pytest.mark.parametrize('N,nlon,nlat,B', [(96, 1440, 720, 5), (96, 1440, 18, 9), (31, 1440, 36, 4), (6, 1440, 16, 1), (1, 2880, 8, 2), (2, 192, 6, 3),
                                          (45, 192, 90, 6), (60, 720, 360, 5), (96, 720, 10, 8), (17, 240, 120, 3), (100, 480, 24, 4)])
def test_rotation_folded_kernel(N, nlon, nlat, B):
    """The rotation-folded kernel (20 images per column where nlon % 160 == 0, 12 where nlon % 96 == 0, 6 where nlon % 48 == 0) against the oracle: long and
    short class lists (padding slots), partial column tiles, a single column tile (waves without work), ragged batch sizes."""
    grid = ga.grid.GeographicGrid(360.0 / nlon, 180.0 / nlat)
    ker = orc.KernelTable('ewh', love())
    batch = np.stack([inputs.coefficients(4000 + N * 10 + e, N) for e in range(B)])
    nref = min(B, 3)
    ref = np.stack([orc.synthesis_regular(batch[e], grid.meridians, grid.parallels, ker) for e in range(nref)])
    plan = ga.engine.Plan(N, *_tables(grid, N, 'ewh'))
    info = plan.info()
    assert info['rotation_symmetry'] and info['fused'] and info['north_south_symmetry']
    outs = {}
    for path in ('auto', 'rot', 'fused'):
        plan.set_path(path)
        outs[path] = ga.engine.to_host(plan.synthesis(batch))
        assert relerr(outs[path][0:nref], ref) < TOL, path
    assert np.array_equal(outs['auto'], outs['rot'])                    # the automatic choice is this kernel
    assert relerr(outs['rot'], outs['fused']) < TOL                     # all epochs against the 4-fold kernel
    plan.set_path('rot')
    for nb in (1, 2, 3):                                                # partial epoch tiles: bit-identical to the full batch
        if nb < B:
            assert np.array_equal(ga.engine.to_host(plan.synthesis(batch[0:nb])), outs['rot'][0:nb])


@pytest.mark.parametrize('N,nlon,nlat,B', [(96, 1440, 720, 5), (96, 1440, 18, 9), (31, 1440, 36, 4), (6, 1440, 16, 1), (1, 2880, 8, 2),
                                          (45, 720, 90, 6), (17, 1440, 10, 3), (100, 1440, 24, 4), (9, 1440, 4, 2), (10, 1440, 4, 5)])
def test_rotation_counts(N, nlon, nlat, B):
    """Every rotation count the meridians allow (3, 6, 9, 10: 6, 12, 18 and 20 images per column of the fundamental domain) against
    the oracle and against each other; counts the meridians do not allow are refused, 0 restores the plan's own choice."""
    grid = ga.grid.GeographicGrid(360.0 / nlon, 180.0 / nlat)
    ker = orc.KernelTable('ewh', love())
    batch = np.stack([inputs.coefficients(7000 + N * 10 + e, N) for e in range(B)])
    nref = min(B, 2)
    ref = np.stack([orc.synthesis_regular(batch[e], grid.meridians, grid.parallels, ker) for e in range(nref)])
    plan = ga.engine.Plan(N, *_tables(grid, N, 'ewh'))
    own = plan.info()['rotations']
    allowed = [R for R in (3, 6, 9, 10) if nlon % (2 * R) == 0 and (nlon // R) % 16 == 0]
    assert own == ([R for R in (10, 9, 6, 3) if R in allowed] + [0])[0]
    outs = {}
    for R in (3, 6, 9, 10, 4, 12):
        if R not in allowed:
            with pytest.raises(ga._lib.ShgError):
                plan.set_rotations(R)
            continue
        plan.set_rotations(R)
        assert plan.info()['rotations'] == R and plan.info()['rotation_symmetry']
        outs[R] = ga.engine.to_host(plan.synthesis(batch))
        assert relerr(outs[R][0:nref], ref) < TOL, R
        assert relerr(outs[R], outs[allowed[0]]) < TOL, R
        if B > 1:
            assert np.array_equal(ga.engine.to_host(plan.synthesis(batch[0:1])), outs[R][0:1])
    plan.set_rotations(0)
    assert plan.info()['rotations'] == own
    assert np.array_equal(ga.engine.to_host(plan.synthesis(batch)), outs[own])


def test_stage_limit_knob():
    """shg_plan_set_stage_limit: a limit on the workgroups in their Legendre stage changes the schedule, never a bit of the result; repeated
    launches find the token counter at zero again (a counter left non-zero would stall the next launch for milliseconds, not break it)."""
    grid = ga.grid.GeographicGrid(0.25, 2.0)
    batch = np.stack([inputs.coefficients(8200 + e, 96) for e in range(9)])
    plan = ga.engine.Plan(96, *_tables(grid, 96, 'ewh'))
    assert plan.info()['rotation_symmetry']
    base = ga.engine.to_host(plan.synthesis(batch))
    for limit in (-7, 3, 1, 0):
        plan.set_stage_limit(limit)
        for call in range(2):
            assert np.array_equal(ga.engine.to_host(plan.synthesis(batch)), base), (limit, call)


def test_rotation_folded_kernel_applicability():
    pot = orc.KernelTable('potential')
    # parallels without the north-south symmetry: plain variant; meridians off the equi-angular raster: other kernels
    mer = np.linspace(-np.pi, np.pi, 480, endpoint=False) + np.pi / 480
    par = np.linspace(1.4, -1.1, 37)
    anm = inputs.coefficients(77, 40)
    grid = ga.grid.RegularGrid(mer, par)
    plan = ga.engine.Plan(40, *_tables(grid, 40, 'potential'))
    assert plan.info()['rotation_symmetry'] and not plan.info()['north_south_symmetry']
    ref = orc.synthesis_regular(anm, mer, par, pot)
    for path in ('auto', 'rot', 'fused'):
        plan.set_path(path)
        assert relerr(ga.engine.to_host(plan.synthesis(anm)), ref) < TOL, path
    shifted = ga.engine.Plan(40, *_tables(ga.grid.RegularGrid(mer + 1e-9, par), 40, 'potential'))
    assert not shifted.info()['rotation_symmetry']
    with pytest.raises(ga._lib.ShgError):
        shifted.set_path('rot')
    for nlon in (360, 180, 1000):                        # nlon / R not a multiple of 16 for R = 6 and R = 3
        g = ga.grid.GeographicGrid(360.0 / nlon, 5.0)
        assert not ga.engine.Plan(10, *_tables(g, 10, 'potential')).info()['rotation_symmetry']
    high = ga.engine.Plan(120, *_tables(ga.grid.GeographicGrid(0.5, 2.0), 120, 'potential'))       # panel beyond the LDS
    assert not high.info()['rotation_symmetry']
    with pytest.raises(ga._lib.ShgError):
        high.set_path('rot')


def test_fused_path_limits():
    grid = ga.grid.GeographicGrid(2, 2)
    plan = ga.engine.Plan(140, *_tables(grid, 140, 'potential'))
    assert plan.info()['fused']                # K = 4 * 80 = 320: too large for the 64-row panel, the 32-row fused kernel takes over
    with pytest.raises(ga._lib.ShgError):
        plan.set_path('fused')
    ker = orc.KernelTable('potential', love())
    batch = np.stack([inputs.coefficients(700 + e, 140) for e in range(5)])
    ref = np.stack([orc.synthesis_regular(batch[e], grid.meridians, grid.parallels, ker) for e in range(5)])
    for path in ('auto', 'fused32', 'staged'):
        plan.set_path(path)
        assert relerr(ga.engine.to_host(plan.synthesis(batch)), ref) < TOL, path
    big = ga.engine.Plan(220, *_tables(grid, 220, 'potential'))
    assert not big.info()['fused']             # K = 448: staged kernels
    with pytest.raises(ga._lib.ShgError):
        big.set_path('fused32')
    odd = ga.grid.GeographicGrid(2, 20)        # 9 parallels: no north-south pairing
    plan = ga.engine.Plan(140, *_tables(odd, 140, 'potential'))
    assert not plan.info()['fused'] and not plan.info()['north_south_symmetry']
    g7 = ga.grid.RegularGrid(np.array([-3.0, -2.2, -0.4, 0.1, 0.9, 2.5, 3.1]), np.array([1.3, 1.0, 0.2, -0.5]))
    plan = ga.engine.Plan(10, *_tables(g7, 10, 'potential'))
    assert not plan.info()['fused'] and not plan.info()['fourfold_symmetry']


def test_point_list_path(golden):
    """Grids without .parallels take the point-list kernel (AttributeError dispatch of the reference)."""
    g = golden('g7_synthesis')
    lon, lat = inputs.scattered_points(11, 1000)
    grid = ga.grid.IrregularGrid(lon, lat)
    out = make_pc(inputs.coefficients(12, 40)).to_grid(grid, kernel='ewh')
    assert type(out) is ga.grid.IrregularGrid and grid.values is None
    assert relerr(out.values, g['points_ewh']) < TOL
    batch = np.stack([inputs.coefficients(12 + e, 40) for e in range(6)])
    vals = ga.engine.to_host(ga.gravityfield.synthesize(batch, grid, 'ewh'))
    ker = orc.KernelTable('ewh', love())
    for e in (0, 5):
        assert relerr(vals[e], orc.synthesis_points(batch[e], lon, lat, ker)) < TOL
    # the same points given as a regular grid agree with the separable path
    reg = ga.grid.GeographicGrid(10, 10)
    irr = ga.grid.IrregularGrid(reg.longitude, reg.latitude)
    a = make_pc(batch[0]).to_grid(reg, 'ewh').values
    b = make_pc(batch[0]).to_grid(irr, 'ewh').values
    assert relerr(a, b) < TOL


@pytest.mark.parametrize('npts,N,B', [(3001, 30, 50), (777, 45, 96), (130, 12, 49)])
def test_point_list_many_epochs(npts, N, B):
    """Long time series on a point list take the GEMM path (rows of the spherical harmonic matrix generated inside the fp64
    MFMA kernel from per-point tables); compared with the oracle and with the recursion kernel that short series use."""
    lon, lat = inputs.scattered_points(21, npts)
    grid = ga.grid.IrregularGrid(lon, lat)
    batch = np.stack([inputs.coefficients(400 + e, N) for e in range(B)])
    vals = ga.engine.to_host(ga.gravityfield.synthesize(batch, grid, 'ewh'))
    assert vals.shape == (B, npts)
    ker = orc.KernelTable('ewh', love())
    for e in (0, B // 2, B - 1):
        assert relerr(vals[e], orc.synthesis_points(batch[e], lon, lat, ker)) < TOL
    short = np.vstack([ga.engine.to_host(ga.gravityfield.synthesize(batch[e0:e0 + 16], grid, 'ewh')) for e0 in range(0, B, 16)])
    assert relerr(vals, short) < 1e-13


def test_monthly_files_to_grids(tmp_path):
    """monthly SDS files -> TimeSeries -> one batched synthesis; every epoch equals the oracle's synthesis
    of the coefficients the file holds."""
    names = []
    for month in range(1, 7):
        name = tmp_path / 'GSM-2_2012{0:02d}.txt'.format(month)
        name.write_bytes(inputs.gsm_file_text(200 + month, 30, start='2012-{0:02d}-01T00:00:00.00'.format(month)))
        names.append(str(name))
    series = ga.io.load_time_series(names[::-1])
    grid = ga.grid.GeographicGrid(2.0, 2.0)
    grids = series.to_grid(grid, kernel='ewh')
    assert [g.epoch.month for g in grids] == [1, 2, 3, 4, 5, 6]
    kernel = orc.KernelTable('ewh', love())
    for k in (0, 3, 5):
        ref = orc.synthesis_regular(ga.io.loadgsm(names[k]).anm, grid.meridians, grid.parallels, kernel)
        assert relerr(grids[k].value_array, ref) < 1e-12


def test_basis_function_representations_golden(golden):
    """RadialBasisFunctions / AnisotropicBasisFunctions / SurfaceMasCons (grates/gravityfield.py:484-785)
    against the reference on 700 scattered nodal points, band 2..12 (tests/golden/g15_basis_functions.npz)."""
    g = golden('g15_basis_functions')
    lon, lat, values, k_rbf, k_aniso = inputs.basis_function_case(100, 700, 2, 12)
    rbf = ga.gravityfield.RadialBasisFunctions(ga.grid.IrregularGrid(lon, lat), k_rbf, 2, 12)
    assert rbf.values.shape == (700,) and not rbf.values.any()
    rbf.values = values
    gf = rbf.to_potential_coefficients()
    assert relerr(gf.anm, g['rbf_anm']) < 1e-12
    F = rbf.to_potential_coefficients_matrix()
    assert F.shape == (165, 700)
    assert relerr(F[:, ::10], g['rbf_matrix_cols10']) < 1e-12
    assert relerr(F @ values, g['rbf_matrix_times_values']) < 1e-12
    assert relerr(rbf.to_grid(ga.grid.GeographicGrid(10.0, 10.0), 'ewh').value_array, g['rbf_grid_ewh']) < 1e-12
    other = rbf.copy()
    other.values[:] = 0.0
    assert rbf.values.any() and rbf.is_compatible(other)

    # small device blocks: the sum over the nodal points is accumulated block by block
    saved = ga.gravityfield._POINT_CHUNK_BYTES
    ga.gravityfield._POINT_CHUNK_BYTES = 8 * 13 * 13 * 96
    try:
        assert relerr(rbf.to_potential_coefficients().anm, g['rbf_anm']) < 1e-12
    finally:
        ga.gravityfield._POINT_CHUNK_BYTES = saved

    aniso = ga.extras.AnisotropicBasisFunctions(ga.grid.IrregularGrid(lon, lat), k_aniso, 2, 12)
    aniso.values = values
    assert relerr(aniso.to_grid(ga.grid.GeographicGrid(10.0, 10.0), 'ewh').value_array, g['aniso_grid_ewh']) < 1e-12
    assert relerr(aniso.to_grid(ga.grid.GeographicGrid(10.0, 10.0), 'potential').value_array, g['aniso_grid_potential']) < 1e-12

    a = ga.extras.SurfaceMasCons(ga.grid.IrregularGrid(lon, lat), 'ewh')
    b = a.copy()
    a.values = values
    b.values = values[::-1].copy()
    np.testing.assert_allclose(((a + b) * 3 - a / 4.0).values, g['mascon_arith'], rtol=1e-15, atol=0)
    with pytest.raises(TypeError):
        a + 1.0
    with pytest.raises(TypeError):
        a * b
    with pytest.raises(ValueError):
        a + ga.extras.SurfaceMasCons(ga.grid.IrregularGrid(lon[:10], lat[:10]), 'ewh')
    # analysis of mascon values = analysis of the same values on the grid itself
    lon2, lat2 = inputs.scattered_points(5, 400)
    pts = ga.grid.IrregularGrid(lon2, lat2)
    pts.values = np.random.default_rng(3).standard_normal(400)
    expected = pts.to_potential_coefficients(0, 8, 'ewh').anm
    assert relerr(ga.extras.SurfaceMasCons(pts, 'ewh').to_potential_coefficients(0, 8).anm, expected) < 1e-13


def test_time_variable_field_and_gridded_rms_golden(golden):
    """Trend + Oscillation + TimeSeries evaluated at 30 epochs, synthesised in batches and reduced to an RMS grid on the
    device (grates/gravityfield.py:788-812, 1054-1172; tests/golden/g16_time_variable.npz)."""
    g = golden('g16_time_variable')

    def field(seed):
        gf = ga.gravityfield.PotentialCoefficients()
        gf.anm = inputs.coefficients(seed, 20)
        return gf
    t0 = datetime.datetime(2005, 1, 1)
    series = []
    for k in range(6):
        gf = field(120 + k)
        gf.epoch = t0 + datetime.timedelta(days=61 * k)
        series.append(gf)
    gfm = ga.gravityfield

    # linear trend per year + annual oscillation + the interpolated series: the constituents the fixture was generated with
    # (grates/gravityfield.py:788-812, 1054-1140)
    model = gfm.TimeVariableGravityField([gfm.Trend(field(110), t0), gfm.Oscillation(field(111), field(112), 365.25, t0), gfm.TimeSeries(series)])
    assert ga.grid.ReuterGrid is ga.extras.ReuterGrid and gfm.SurfaceMasCons is ga.extras.SurfaceMasCons       # reference import paths
    epochs = [t0 + datetime.timedelta(days=9.5 * k) for k in range(30)]
    at7 = model.evaluate_at(epochs[7])
    assert at7.epoch == epochs[7]
    np.testing.assert_allclose(at7.anm, g['model_anm_at_7'], rtol=0, atol=1e-24)       # values ~1e-10
    grid = ga.grid.GeographicGrid(5.0, 5.0)
    for batch in (240, 7, 1):                                                          # one batch / 4 full + 1 ragged / per epoch
        rms = gfm.gridded_rms(model, epochs, 'ewh', grid, batch=batch)
        assert type(rms) is ga.grid.GeographicGrid and rms.values.shape == (36 * 72,)
        assert relerr(rms.values, g['rms_ewh_5deg']) < 1e-12
    # fields with different constants take the per-epoch route
    class Rescaled:
        def evaluate_at(self, epoch):
            gf = model.evaluate_at(epoch)
            if epoch.day % 2:
                other = ga.gravityfield.PotentialCoefficients(GM=gf.GM * (1 + 1e-9), R=gf.R)
                other.anm = gf.anm / (1 + 1e-9)
                gf = other
            return gf
    assert relerr(gfm.gridded_rms(Rescaled(), epochs, 'ewh', grid).values, g['rms_ewh_5deg']) < 1e-12

    binned = gfm.TimeSeries(series).bin([t0 + datetime.timedelta(days=30), t0 + datetime.timedelta(days=200), t0 + datetime.timedelta(days=290)],
                                        func=lambda members: sum(members[1:], members[0]) * (1.0 / len(members)))
    assert binned.epochs()[1] == t0 + datetime.timedelta(days=200)
    np.testing.assert_allclose(np.array([d.anm for _, d in binned.items()]), g['binned_anm'], rtol=0, atol=1e-25)
    with pytest.raises(ValueError):
        gfm.TimeSeries(series).bin([t0, t0 + datetime.timedelta(days=5000)], func=lambda m: m[0])
    with pytest.raises(TypeError):
        gfm.TimeSeries(series).bin([t0])                                               # numpy.mean: as upstream


def test_epoch_rms_kernel_properties():
    """shg_epoch_rms through the C ABI: chained batches = one batch, bit for bit (same order of additions); ragged sizes."""
    import torch
    from grates_amd import engine
    v = torch.randn((37, 100003), dtype=torch.float64, device='cuda')
    whole = engine.epoch_rms(v, None, 37)
    acc = engine.epoch_rms(v[:16], None, 0)
    acc = engine.epoch_rms(v[16:19], acc, 0)
    acc = engine.epoch_rms(v[19:], acc, 37)
    assert torch.equal(whole, acc)
    expected = torch.sqrt((v * v).sum(dim=0) / 37)
    assert float((whole - expected).abs().max()) < 1e-14
    empty = engine.epoch_rms(v[:0], None, 0)
    assert float(empty.abs().max()) == 0.0


def test_detrend_golden(golden):
    """TimeSeries.detrend with bias + drift + annual oscillation over 40 epochs (grates/gravityfield.py:993-1012): estimated
    parameters and residual series against the reference (tests/golden/g16_time_variable.npz); the products run on the device."""
    g = golden('g16_time_variable')
    t0 = datetime.datetime(2005, 1, 1)
    rng = np.random.default_rng(130)
    fields = []
    for k in range(40):
        gf = ga.gravityfield.PotentialCoefficients()
        gf.anm = inputs.coefficients(140, 15) * (1 + 0.01 * k) + inputs.coefficients(141, 15) * np.sin(2 * np.pi * k * 30.4 / 365.25) \
            + rng.standard_normal((16, 16)) * 1e-12
        gf.epoch = t0 + datetime.timedelta(days=30.4 * k)
        fields.append(gf)
    ts = ga.gravityfield.TimeSeries(fields)
    parameters = ts.detrend([ga.utilities.Polynomial(1, t0), ga.utilities.Oscillation(365.25, t0)])
    assert isinstance(parameters, np.ndarray) and parameters.shape == (4, 256)
    assert relerr(parameters, g['detrend_parameters']) < 1e-11
    assert relerr(ts.to_array(), g['detrend_residuals']) < 1e-9                     # residuals are 1e-2 of the signal
    assert np.abs(ts.to_array()).max() < 1e-11


def test_synthesis_on_reuter_grid_golden(golden):
    """to_grid on the points of ReuterGrid(30) through the point-list kernel against the reference (g17_reuter.npz)."""
    g = golden('g17_reuter')
    gf = ga.gravityfield.PotentialCoefficients()
    gf.anm = inputs.coefficients(150, 40)
    grid = ga.extras.ReuterGrid(30)
    out = gf.to_grid(grid, kernel='ewh')
    assert type(out) is ga.extras.ReuterGrid and out.values.shape == g['reuter30_ewh'].shape
    assert relerr(out.values, g['reuter30_ewh']) < 1e-12
