"""
GPU parity of the table functions and index maps (rows a1-a5 of SURVEY.md 8a) against the golden vectors of
the reference and against the oracle.  Everything goes through the C ABI (grates_amd._lib).
"""

import numpy as np
import pytest

import grates_amd as ga
import inputs
from conftest import relerr
from oracle import shg_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('N', [5, 60, 96, 180])
def test_legendre_functions_golden(golden, N):
    g = golden('g1_legendre')
    colat = g['colat_{0}'.format(N)]
    P = ga.utilities.legendre_functions(N, colat)
    ref = g['pnm_{0}'.format(N)]
    assert P.shape == ref.shape
    # same recursion, same association, no FMA contraction; only cos/sin of the device library may differ by an ulp
    np.testing.assert_allclose(P, ref, rtol=5e-13, atol=1e-300)
    assert relerr(P, ref) < 1e-14
    for m in sorted({0, 1, N // 2, N}):
        Pm = ga.utilities.legendre_functions_per_order(N, m, colat)
        np.testing.assert_allclose(Pm, g['pnm_order_{0}_{1}'.format(N, m)], rtol=5e-13, atol=1e-300)


def test_legendre_contracts(golden):
    # shape / raise contracts of the reference tests (grates/testing/utilities.py:64-137)
    assert ga.utilities.legendre_functions(0, 0.3).shape == (1, 1, 1)
    np.testing.assert_array_equal(ga.utilities.legendre_functions(0, np.array([0.3, 1.2])), golden('g1_legendre')['pnm_0'])
    assert ga.utilities.legendre_functions(7, np.linspace(0.1, 3, 11)).shape == (11, 8, 8)
    assert ga.utilities.legendre_functions_per_order(7, 3, np.linspace(0.1, 3, 11)).shape == (11, 5)
    assert ga.utilities.legendre_functions_per_order(7, 7, 0.5).shape == (1, 1)
    with pytest.raises(ValueError):
        ga.utilities.legendre_functions_per_order(3, 4, np.array([0.1]))
    P = ga.utilities.legendre_functions(12, np.array([0.7]))
    for m in range(1, 13):
        np.testing.assert_array_equal(P[0, m - 1, m:], P[0, m:, m])      # mirror into the sine slots
    t = ga.utilities.legendre_functions(4, [0.4], as_tensor=True)
    assert t.is_cuda and tuple(t.shape) == (1, 5, 5)


def test_legendre_underflow_near_pole():
    # N = 180 at the first 0.5-degree parallel: sectorials underflow; no NaN/Inf, same zeros as NumPy
    colat = orc.colatitude(orc.geographic_grid(0.5, 0.5)[1][0:3])
    P = ga.utilities.legendre_functions(180, colat)
    ref = orc.legendre_functions(180, colat)
    assert np.all(np.isfinite(P))
    np.testing.assert_allclose(P, ref, rtol=5e-13, atol=1e-300)


def test_trigonometric_and_spherical_harmonics(golden):
    g = golden('g3_trig')
    for N in (20, 96):
        cs = ga.utilities.trigonometric_functions(N, g['lon_{0}'.format(N)])
        np.testing.assert_allclose(cs, g['cs_{0}'.format(N)], rtol=0, atol=4e-16)
    assert ga.utilities.trigonometric_functions(0, 0.1).shape == (1, 1, 1)
    Y = ga.utilities.spherical_harmonics(12, g['ynm_colat'], g['ynm_lon'])
    np.testing.assert_allclose(Y, g['ynm_12'], rtol=1e-13, atol=1e-15)
    assert ga.utilities.spherical_harmonics(5, 0.3, np.linspace(0, 1, 4)).shape == (4, 6, 6)


@pytest.mark.parametrize('nmin,nmax,na', [(0, 5, 5), (2, 5, 5), (0, 60, 60), (2, 96, 96), (0, 180, 180), (3, 3, 3), (1, 8, 5), (0, 4, 9)])
def test_device_ravel_unravel_bit_exact(nmin, nmax, na):
    rng = np.random.default_rng(nmax * 7 + nmin)
    arr = rng.standard_normal((3, na + 1, na + 1))
    vec = ga.engine.to_host(ga.engine.ravel(arr, nmin, nmax))
    np.testing.assert_array_equal(vec, orc.ravel_coefficients(arr, nmin, nmax))
    P = (nmax + 1) ** 2 - nmin ** 2
    v = rng.standard_normal((3, P))
    back = ga.engine.to_host(ga.engine.unravel(v, nmin, nmax))
    np.testing.assert_array_equal(back, orc.unravel_coefficients(v, nmin, nmax))


def test_device_ravel_golden_index_map(golden):
    g = golden('g4_index')
    for nmin, nmax in ((0, 60), (2, 96), (0, 180)):
        flat = np.arange((nmax + 1) ** 2, dtype=np.float64).reshape(1, nmax + 1, nmax + 1)
        vec = ga.engine.to_host(ga.engine.ravel(flat, nmin, nmax))[0]
        np.testing.assert_array_equal(vec.astype(np.int64), g['ravel_{0}_{1}'.format(nmin, nmax)])


def test_degree_scale_and_gaussian_filter(golden):
    g = golden('g10_filter')
    gf = ga.gravityfield.PotentialCoefficients()
    gf.anm = inputs.coefficients(41, 60)
    out = ga.filter.Gaussian(300).filter(gf)
    assert out is not gf
    np.testing.assert_array_equal(gf.anm, inputs.coefficients(41, 60))
    np.testing.assert_allclose(out.anm, g['gaussian_300_n60'], rtol=1e-15, atol=0)
    batch = np.stack([inputs.coefficients(k, 30) for k in range(5)])
    bw = ga.filter.Butterworth(4, 20)
    res = ga.engine.to_host(bw.filter_batch(batch))
    n = np.maximum(*np.meshgrid(np.arange(31), np.arange(31)))
    np.testing.assert_allclose(res, batch * np.power(1 + (n / 20.0) ** 8, -0.5), rtol=1e-15)
