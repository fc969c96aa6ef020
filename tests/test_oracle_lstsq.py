"""
Pins oracle/lstsq_oracle.py (NumPy restatement of the block-banded normal-equation solver) to the outputs of the
reference stored in tests/golden/g11_lstsq.npz.  Runs on the CPU.
Tolerance: 1e-12 relative to the largest reference entry (identical algorithms, LAPACK on both sides).
"""

import numpy as np

import inputs
from conftest import relerr
from oracle import lstsq_oracle as lo

TOL = 1e-12
DIM, ORDER, EPOCHS = 6, 2, 7


def smoother_parts():
    models = lo.var_sequence(inputs.var_covariance_function(1, DIM, ORDER))
    return [lo.block_diagonal_normals(inputs.observation_normals(2, EPOCHS, DIM)), lo.var_sequence_normals(models, EPOCHS)]


def test_var_models(golden):
    g = golden('g11_lstsq')
    cf = inputs.var_covariance_function(1, DIM, ORDER)
    for k, (coefficients, Q) in enumerate(lo.var_sequence(cf)):
        assert relerr(Q, g['var{0}_Q'.format(k)]) < TOL
        if k:
            assert relerr(np.array(coefficients), g['var{0}_coefficients'.format(k)]) < TOL
        blocks = lo.var_normal_blocks(coefficients, Q)
        scale = max(np.abs(g['var{0}_normals_{1}{2}'.format(k, r, c)]).max() for r, c in blocks)    # B_2 of a VAR(1) process is rounding noise
        for (r, c), blk in blocks.items():
            assert np.abs(blk - g['var{0}_normals_{1}{2}'.format(k, r, c)]).max() < TOL * scale
    constraint = lo.var_sequence_normals(lo.var_sequence(cf), EPOCHS)
    assert relerr(lo.to_array(constraint['matrix']), g['constraint_matrix']) < TOL
    # the constraint normals reproduce the covariance function they were built from (grates/lstsq.py:394-411)
    full = lo.inverse(lo.cholesky(lo.var_sequence_normals(lo.var_sequence(cf), 4)['matrix']))
    back = np.array([full['blocks'][(0, k)] for k in range(4)])
    assert relerr(back, g['covariance_function_back']) < 1e-11
    assert relerr(back[0:3], np.array(cf)) < 1e-11


def test_smoother_solution_and_covariances(golden):
    g = golden('g11_lstsq')
    factors = [1.0, 0.5]
    parts = smoother_parts()
    combined = lo.accumulate(parts, factors)
    assert relerr(lo.to_array(combined['matrix']), g['combined_matrix']) < TOL
    assert relerr(combined['rhs'], g['combined_rhs']) < TOL
    assert abs(combined['lPl'] - g['combined_lPl']) < TOL * abs(g['combined_lPl'])
    assert combined['count'] == int(g['combined_count'])
    x, mc = lo.solve(combined, g['signs'].astype(float))
    assert relerr(x, g['solution']) < TOL
    assert relerr(mc, g['monte_carlo_vectors']) < TOL
    assert relerr(lo.to_array(combined['matrix']), g['factor']) < TOL
    assert abs(lo.posterior_sigma(combined, x) - g['posterior_sigma']) < 1e-11 * abs(g['posterior_sigma'])
    assert relerr(np.array([lo.residual_square_sum(p, x) for p in parts]), g['residual_square_sums']) < 1e-11
    assert relerr(np.array([lo.redundancy(p, mc, f) for p, f in zip(parts, factors)]), g['redundancies']) < 1e-11
    assert relerr(lo.variance_factors(parts, mc, x, factors), g['variance_factors']) < 1e-10
    lo.sparse_inverse(combined['matrix'])
    assert relerr(lo.to_array(combined['matrix']), g['sparse_inverse']) < 1e-11
    again = lo.accumulate(smoother_parts(), factors)
    lo.inverse(lo.cholesky(again['matrix']))
    assert relerr(lo.to_array(again['matrix']), g['full_inverse']) < 1e-11
    # the sparse inverse equals the full inverse on the band
    band = np.abs(np.subtract.outer(np.arange(EPOCHS * DIM) // DIM, np.arange(EPOCHS * DIM) // DIM)) <= ORDER
    upper = np.triu(np.ones_like(band), 0) & band
    np.testing.assert_allclose(g['sparse_inverse'][upper], g['full_inverse'][upper], rtol=0, atol=1e-12)


def test_ragged_block_matrix(golden):
    g = golden('g11_lstsq')
    rows, cols = lo.compute_block_index(g['ragged_input'].shape, 5)
    np.testing.assert_array_equal(rows, g['ragged_index'])
    bm = lo.from_array(np.triu(g['ragged_input']), rows, cols)
    assert sorted(bm['blocks']) == [(0, 0), (0, 1), (0, 2), (1, 1), (1, 3), (2, 2), (2, 3), (3, 3)]
    b = g['ragged_rhs']
    assert relerr(lo.multiply_symmetric(bm, b), g['ragged_multiply_symmetric']) < TOL
    np.testing.assert_array_equal(lo.diag(bm), g['ragged_diag'])
    assert relerr(lo.to_array(lo.matmul(bm, bm)), g['ragged_matmul']) < TOL
    lo.cholesky(bm)
    assert (1, 2) in bm['blocks']                        # fill-in created by the elimination of block row 0
    assert relerr(lo.to_array(bm), g['ragged_factor']) < TOL
    assert relerr(lo.solve_triangular(bm, b, transpose=True), g['ragged_solve_T']) < TOL
    assert relerr(lo.solve_triangular(bm, b, transpose=False), g['ragged_solve_N']) < TOL
    assert relerr(lo.multiply_triangular(bm, b, transpose=False), g['ragged_multiply_N']) < TOL
    assert relerr(lo.multiply_triangular(bm, b, transpose=True), g['ragged_multiply_T']) < TOL
    sp = lo.sparse_inverse(lo.copy_blocks(bm))
    assert relerr(lo.to_array(sp), g['ragged_sparse_inverse']) < 1e-11
    assert relerr(lo.to_array(lo.inverse(bm)), g['ragged_inverse']) < 1e-11


def test_tikhonov(golden):
    g = golden('g11_lstsq')
    reg = np.random.default_rng(11).uniform(0.5, 2.0, 12)
    bias = np.random.default_rng(12).standard_normal((12, 1))
    tk = lo.tikhonov(reg, [0, 4, 8, 12], bias)
    np.testing.assert_array_equal(lo.to_array(tk['matrix']), g['tikhonov_matrix'])
    np.testing.assert_array_equal(tk['rhs'], g['tikhonov_rhs'])
    assert tk['lPl'] == g['tikhonov_lPl'] and tk['count'] == int(g['tikhonov_count'])
    empty = lo.tikhonov(reg, [0, 4, 8, 12])
    assert empty['lPl'] == 0 and not empty['rhs'].any() and empty['count'] == 12
