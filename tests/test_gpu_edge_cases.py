"""
Edge cases through the C ABI on the GPU: empty batches, degree 0, single parallel / meridian, tiny grids, error codes.
"""

import numpy as np
import pytest

import grates_amd as ga
import inputs
from conftest import relerr
from oracle import shg_oracle as orc

pytestmark = pytest.mark.gpu


def tables(grid, N, kernel='potential'):
    colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel(kernel), N, grid.parallels, 3.9860044150e+14, 6.3781363000e+06,
                                                   grid.semimajor_axis, grid.flattening)
    return colat, kn, grid.meridians


def test_empty_batches_everywhere():
    import torch
    grid = ga.grid.GeographicGrid(10, 10)
    plan = ga.engine.Plan(8, *tables(grid, 8))
    assert tuple(plan.synthesis(np.zeros((0, 9, 9))).shape) == (0, 18, 36)
    assert tuple(plan.analysis(np.zeros((0, 18, 36)), grid.area, 0).shape) == (0, 9, 9)
    assert tuple(ga.engine.degree_scale(np.zeros((0, 9, 9)), np.ones(9), 2).shape) == (0, 9, 9)
    assert tuple(ga.engine.ravel(np.zeros((0, 9, 9)), 0, 8).shape) == (0, 81)
    assert tuple(ga.engine.unravel(np.zeros((0, 81)), 0, 8).shape) == (0, 9, 9)
    flt = ga.filter.OrderWiseFilter(inputs.orderwise_random_blocks(1, 8))
    assert tuple(flt.filter_batch(np.zeros((0, 9, 9))).shape) == (0, 9, 9)
    assert tuple(ga.engine.dgemm(np.zeros((0, 5)), np.zeros((5, 3))).shape) == (0, 3)
    z = ga.engine.to_host(ga.engine.dgemm(np.zeros((4, 0)), np.zeros((0, 3))))
    assert z.shape == (4, 3) and np.all(z == 0.0)
    assert tuple(plan.covariance_propagation(np.eye(81), 0, 5, 5).shape) == (0,)
    irr = ga.grid.IrregularGrid(np.zeros(0), np.zeros(0), area_element=np.zeros(0))   # (without areas the reference divides by zero too)
    vals = ga.gravityfield.synthesize(np.zeros((2, 9, 9)), irr, 'potential')
    assert tuple(vals.shape) == (2, 0)
    assert torch.cuda.is_available()


def test_degree_zero_and_tiny_grids():
    pot = orc.KernelTable('potential')
    for nlat, nlon in ((1, 4), (1, 1), (2, 8), (5, 4)):
        mer = np.linspace(-np.pi, np.pi, nlon, endpoint=False) + np.pi / nlon
        par = np.linspace(1.2, -1.2, nlat) if nlat > 1 else np.array([0.4])
        grid = ga.grid.RegularGrid(mer, par)
        for N in (0, 1, 3):
            anm = inputs.coefficients(nlat * 10 + N, N)
            gf = ga.gravityfield.PotentialCoefficients()
            gf.anm = anm
            ref = orc.synthesis_regular(anm, mer, par, pot)
            assert relerr(gf.to_grid(grid, 'potential').value_array, ref) < 1e-12, (nlat, nlon, N)
    grid = ga.grid.GeographicGrid(30, 30)
    cov = np.array([[4.0e-20]])
    s = grid.covariance_propagation(cov, 0, 0, kernel='potential')
    r = orc.geocentric_radius(grid.parallels)
    np.testing.assert_allclose(s.reshape(6, 12), np.repeat((2.0e-10 * 3.9860044150e+14 / r)[:, None], 12, axis=1), rtol=1e-13)


def test_min_degree_above_everything_and_status_codes():
    grid = ga.grid.GeographicGrid(10, 10)
    plan = ga.engine.Plan(4, *tables(grid, 4))
    s = ga.engine.to_host(plan.covariance_propagation(np.zeros((0, 0)), 5))         # min_degree = N + 1: empty coefficient set
    assert s.shape == (18 * 36,) and np.all(s == 0.0)
    with pytest.raises(ga._lib.ShgError) as err:
        plan.covariance_propagation(np.eye(25), 0, 3, 99)
    assert err.value.status == -1 and 'band' in str(err.value)
    with pytest.raises(ValueError):
        ga.engine.Plan(4, np.zeros(3), np.zeros((3, 4)), np.zeros(8))                 # kn has the wrong shape
    with pytest.raises(ga._lib.ShgError):
        ga.engine.Plan(4000, np.zeros(1), np.zeros((1, 4001)), np.zeros(4))           # degree out of range


def test_plan_reuse_and_cache():
    ga.engine.clear_plan_cache()               # the cache holds at most 8 plans and evicts the oldest
    grid = ga.grid.GeographicGrid(5, 5)
    gf = ga.gravityfield.PotentialCoefficients()
    gf.anm = inputs.coefficients(3, 20)
    a = gf.to_grid(grid, 'ewh').value_array
    n_plans = len(ga.engine._plan_cache)
    b = gf.to_grid(ga.grid.GeographicGrid(5, 5), 'ewh').value_array                 # equal geometry -> same cached plan
    assert len(ga.engine._plan_cache) == n_plans
    np.testing.assert_array_equal(a, b)
    c = gf.to_grid(grid, 'potential').value_array                                    # other kernel table -> new plan
    assert len(ga.engine._plan_cache) == n_plans + 1 and not np.allclose(a, c)
    ga.engine.clear_plan_cache()
    np.testing.assert_array_equal(gf.to_grid(grid, 'ewh').value_array, a)


def test_shared_plan_on_two_streams_and_threads():
    """A cached plan used from two torch streams and from two host threads: calls on a new stream wait for the
    previous user's work on the device, host threads are serialised -- neither the workspaces nor the lazily built tables race."""
    import threading
    import torch
    grid = ga.grid.GeographicGrid(0.5, 1.0)                    # rotation-folded kernel
    N = 40
    plan = ga.engine.Plan(N, *tables(grid, N, 'ewh'))
    ker = orc.KernelTable('ewh', ga.data.load_love_numbers()[0])
    batches = [np.stack([inputs.coefficients(3000 + 10 * s + e, N) for e in range(6)]) for s in range(2)]
    refs = [np.stack([orc.synthesis_regular(b[e], grid.meridians, grid.parallels, ker) for e in (0, 5)]) for b in batches]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    dev = [torch.from_numpy(b).cuda() for b in batches]
    torch.cuda.synchronize()
    outs = [[], []]
    for rep in range(20):                                      # alternate the streams call by call, different batch sizes
        for s in (0, 1):
            with torch.cuda.stream(streams[s]):
                outs[s].append(plan.synthesis(dev[s][0:6 - (rep % 3)]))
    torch.cuda.synchronize()
    for s in (0, 1):
        for rep, o in enumerate(outs[s]):
            got = ga.engine.to_host(o)
            assert relerr(got[0], refs[s][0]) < 1e-12, (s, rep)
            if got.shape[0] == 6:
                assert relerr(got[5], refs[s][1]) < 1e-12, (s, rep)

    results = {}

    def worker(s):
        with torch.cuda.stream(streams[s]):
            acc = [plan.synthesis(dev[s]) for _ in range(10)]
            streams[s].synchronize()
            results[s] = [ga.engine.to_host(a) for a in acc]

    threads = [threading.Thread(target=worker, args=(s,)) for s in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for s in (0, 1):
        for got in results[s]:
            assert relerr(got[[0, 5]], refs[s]) < 1e-12


def test_error_contract_of_the_reference_on_the_device():
    """every probe of inputs.api_probes -- incl. the ones whose non-raising outcome runs a kernel -- raises the exception type that the
    reference raised when tests/golden/make_golden.py recorded g19_api.json (the CPU half is tests/test_api_signatures.py)"""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g19_api.json')) as f:
        expected = json.load(f)['raises']
    for label, thunk in inputs.api_probes(ga):
        try:
            thunk()
            got = None
        except Exception as err:        # noqa: BLE001
            got = type(err).__name__
        assert got == expected[label], (label, got, expected[label])
