"""
GPU parity of the block-banded normal-equation solver ("Kalman smoother", SURVEY.md 8f rank 1, BASELINE config 5):
grates_amd.lstsq (host loops of the reference, block operations on fp64 MFMA through the C-ABI) against the golden
vectors of the reference (tests/golden/g11_lstsq.npz) and, at larger sizes, against the NumPy oracle.

Tolerances (fp64, stated per quantity): factors, solutions and products 1e-11 relative to the largest reference
entry; inverses / covariances 1e-10 (triangular solves are GEMMs with explicit inverses of the diagonal factor
blocks; the seeded systems have condition numbers of 1e2 ... 1e4).
"""

import numpy as np
import pytest

import grates_amd as ga
import inputs
from conftest import relerr
from oracle import lstsq_oracle as lo

pytestmark = pytest.mark.gpu
ls = ga.lstsq
TOL = 1e-11
TOL_INV = 1e-10
DIM, ORDER, EPOCHS = 6, 2, 7


def observation_system(seed, epochs, dim):
    per_epoch = inputs.observation_normals(seed, epochs, dim)
    idx = np.arange(0, (epochs + 1) * dim, dim)
    bm = ls.BlockMatrix(idx, idx)
    for t, e in enumerate(per_epoch):
        bm[t, t] = e[0]
    return ls.NormalEquations(bm, np.vstack([e[1] for e in per_epoch]), sum(e[2] for e in per_epoch), sum(e[3] for e in per_epoch))


def test_var_models_golden(golden):
    g = golden('g11_lstsq')
    cf = inputs.var_covariance_function(1, DIM, ORDER)
    for k in range(ORDER + 1):
        model = ls.AutoregressiveModel.from_covariance_function(cf[0:k + 1])
        assert model.order == k and model.dimension == DIM
        assert relerr(model.white_noise_covariance, g['var{0}_Q'.format(k)]) < TOL
        if k:
            assert relerr(np.array(model.coefficients), g['var{0}_coefficients'.format(k)]) < TOL
        # relative to the largest block of the model: the process is VAR(1), so B_2 and the blocks built from it are
        # rounding noise (1e-15 / 1e-30) in the reference as well
        keys = [(r, c) for r in range(k + 1) for c in range(r, k + 1)]
        scale = max(np.abs(g['var{0}_normals_{1}{2}'.format(k, r, c)]).max() for r, c in keys)
        for r, c in keys:
            assert np.abs(model.normal_equation_block(r, c) - g['var{0}_normals_{1}{2}'.format(k, r, c)]).max() < TOL * scale
    seq = ls.AutoregressiveModelSequence.from_covariance_function(cf)
    assert seq.maximum_order == ORDER and seq.dimension == DIM
    constraint = seq.normal_equations(EPOCHS)
    assert constraint.status == 'normal_matrix' and constraint.observation_count == EPOCHS * DIM
    assert relerr(constraint.matrix.to_array(), g['constraint_matrix']) < TOL
    back = np.array(seq.covariance_function(3))
    assert relerr(back, g['covariance_function_back']) < TOL_INV
    # order-one representation and transformed coefficients are pure re-arrangements
    m2 = ls.AutoregressiveModel.from_covariance_function(cf)
    one = m2.order_one_representation()
    assert one.order == 2 * DIM and one.dimension == 2 * DIM       # as upstream: ndarray coefficients become a tuple of rows
    tc = m2.to_transformed_coefficients()
    W_inv = np.linalg.inv(np.linalg.cholesky(m2.white_noise_covariance).T)
    ref = np.hstack([-W_inv @ B for B in m2.coefficients[::-1]] + [W_inv])
    assert relerr(tc, ref) < TOL


def test_smoother_golden(golden):
    g = golden('g11_lstsq')
    factors = [1.0, 0.5]
    seq = ls.AutoregressiveModelSequence.from_covariance_function(inputs.var_covariance_function(1, DIM, ORDER))
    parts = [observation_system(2, EPOCHS, DIM), seq.normal_equations(EPOCHS)]
    combined = ls.accumulate_normals(parts, factors)
    assert relerr(combined.matrix.to_array(), g['combined_matrix']) < TOL
    assert relerr(combined.right_hand_side, g['combined_rhs']) < TOL
    assert abs(combined.observation_square_sum - g['combined_lPl']) < TOL * abs(g['combined_lPl'])
    assert combined.observation_count == int(g['combined_count'])
    np.random.seed(123)                                   # the reference draws the Monte-Carlo signs from numpy's global state
    x = combined.solve()
    assert combined.status == 'cholesky_factor'
    assert x.shape == (EPOCHS * DIM, 1) and isinstance(x, np.ndarray)
    assert relerr(x, g['solution']) < TOL
    assert relerr(combined.monte_carlo_vectors, g['monte_carlo_vectors']) < TOL
    assert relerr(combined.matrix.to_array(), g['factor']) < TOL
    assert abs(combined.posterior_sigma(x) - g['posterior_sigma']) < 1e-10 * abs(g['posterior_sigma'])
    assert relerr(np.array([p.residual_square_sum(x) for p in parts]), g['residual_square_sums']) < TOL_INV
    assert relerr(np.array([p.redundancy(combined, f) for p, f in zip(parts, factors)]), g['redundancies']) < TOL_INV
    assert relerr(ls.compute_variance_factors(parts, combined, x, factors), g['variance_factors']) < 1e-9
    combined.compute_covariance(sparse=True)
    assert combined.status == 'covariance_matrix'
    assert relerr(combined.matrix.to_array(), g['sparse_inverse']) < TOL_INV
    with pytest.raises(ValueError):
        combined.solve()                                  # 'Cholesky factor can only be computed from the normal matrix'
    again = ls.accumulate_normals([observation_system(2, EPOCHS, DIM), seq.normal_equations(EPOCHS)], factors)
    again.compute_covariance(sparse=False)
    assert relerr(again.matrix.to_array(), g['full_inverse']) < TOL_INV
    N, n, lPl, count = parts[0].to_array()
    assert N.shape == (EPOCHS * DIM, EPOCHS * DIM) and n.shape == (EPOCHS * DIM, 1) and count == EPOCHS * (DIM + 5)


def test_ragged_block_matrix_golden(golden):
    g = golden('g11_lstsq')
    rows, cols = ls.BlockMatrix.compute_block_index(g['ragged_input'].shape, 5)
    np.testing.assert_array_equal(rows, g['ragged_index'])
    bm = ls.BlockMatrix.from_array(np.triu(g['ragged_input']), rows, cols)
    assert bm.shape == (4, 4)
    assert bm[0, 3] is None and bm[1, 2] is None and bm[2, 1] is None and bm[0, 1].shape == (5, 5) and bm[2, 3].shape == (5, 2)
    np.testing.assert_array_equal(bm.to_array(), np.triu(g['ragged_input']))
    b = g['ragged_rhs']
    assert relerr(bm.multiply_symmetric(b), g['ragged_multiply_symmetric']) < TOL
    np.testing.assert_array_equal(bm.diag(), g['ragged_diag'])
    assert relerr((bm @ bm).to_array(), g['ragged_matmul']) < TOL
    bm.cholesky()
    assert bm.is_nonzero(1, 2)                             # fill-in
    assert relerr(bm.to_array(), g['ragged_factor']) < TOL
    assert relerr(bm.solve_triangular(b, transpose=True), g['ragged_solve_T']) < TOL
    assert relerr(bm.solve_triangular(b, transpose=False), g['ragged_solve_N']) < TOL
    assert relerr(bm.multiply_triangular(b, transpose=False), g['ragged_multiply_N']) < TOL
    assert relerr(bm.multiply_triangular(b, transpose=True), g['ragged_multiply_T']) < TOL
    import torch
    xt = bm.solve_triangular(torch.from_numpy(b).cuda(), transpose=True)       # device in -> device out
    assert torch.is_tensor(xt) and relerr(xt.cpu().numpy(), g['ragged_solve_T']) < TOL
    sp = bm.copy()
    sp.sparse_inverse()
    assert relerr(sp.to_array(), g['ragged_sparse_inverse']) < TOL_INV
    bm.inverse()
    assert relerr(bm.to_array(), g['ragged_inverse']) < TOL_INV


def test_block_matrix_errors_and_factor_set_directly():
    bm = ls.BlockMatrix([0, 3, 5], [0, 3, 5])
    with pytest.raises(ValueError):
        bm[0, 0] = [[1.0]]                                 # not an ndarray
    with pytest.raises(ValueError):
        bm[0, 0] = np.zeros(3)                             # not two-dimensional
    with pytest.raises(ValueError):
        bm[0, 1] = np.zeros((3, 3))                        # wrong shape
    with pytest.raises(IndexError):
        bm[5, 0] = np.zeros((3, 3))
    with pytest.raises(ValueError):
        ls.BlockMatrix.from_array([[1.0]], [0, 1], [0, 1])
    with pytest.raises(ValueError):
        ls.BlockMatrix.from_array(np.eye(4), [0, 3], [0, 4])
    with pytest.raises(ValueError):
        bm @ np.eye(5)
    # a factor assigned block by block (no cholesky() call) is solved with on-demand inverses of its diagonal blocks
    rng = np.random.default_rng(0)
    U = np.triu(rng.standard_normal((5, 5))) + 3 * np.eye(5)
    f = ls.BlockMatrix.from_array(U, [0, 3, 5], [0, 3, 5])
    b = rng.standard_normal((5, 2))
    assert relerr(f.solve_triangular(b), np.linalg.solve(U, b)) < TOL
    assert relerr(f.solve_triangular(b, transpose=True), np.linalg.solve(U.T, b)) < TOL
    not_pd = ls.BlockMatrix.from_array(-np.eye(5), [0, 3, 5], [0, 3, 5])
    with pytest.raises(np.linalg.LinAlgError):
        not_pd.cholesky()


def test_tikhonov_golden(golden):
    g = golden('g11_lstsq')
    reg = np.random.default_rng(11).uniform(0.5, 2.0, 12)
    bias = np.random.default_rng(12).standard_normal((12, 1))
    tk = ls.TikhonovRegularization(reg, [0, 4, 8, 12], bias)
    np.testing.assert_array_equal(tk.matrix.to_array(), g['tikhonov_matrix'])
    np.testing.assert_array_equal(tk.right_hand_side, g['tikhonov_rhs'])
    assert tk.observation_square_sum == g['tikhonov_lPl'] and tk.observation_count == int(g['tikhonov_count'])


@pytest.mark.parametrize('dim,order,epochs', [(130, 1, 6), (257, 2, 5), (1681, 1, 3)])
def test_smoother_against_oracle(dim, order, epochs):
    """larger blocks (ragged against the 128 x 128 MFMA tiles, and the d/o-40 block size of config 5) against the oracle.
    The d = 1681 observation normals A^T A with A [d + 5, d] have condition numbers around 1e6, so two correct fp64
    factorisations differ by ~ cond * eps = 1e-10 of the largest entry: tolerance 1e-9 there."""
    tol, tol_inv = (1e-9, 1e-8) if dim > 1000 else (TOL, TOL_INV)
    cf = inputs.var_covariance_function(5, dim, order)
    factors = [1.0, 2.0]
    signs = np.random.default_rng(6).integers(0, 2, size=(epochs * dim, 4)) * 2.0 - 1.0
    ref_parts = [lo.block_diagonal_normals(inputs.observation_normals(7, epochs, dim)), lo.var_sequence_normals(lo.var_sequence(cf), epochs)]
    ref = lo.accumulate(ref_parts, factors)
    xr, mcr = lo.solve(ref, signs)
    factor_ref = lo.to_array(ref['matrix'])
    lo.sparse_inverse(ref['matrix'])

    seq = ls.AutoregressiveModelSequence.from_covariance_function(cf)
    parts = [observation_system(7, epochs, dim), seq.normal_equations(epochs)]
    combined = ls.accumulate_normals(parts, factors)
    combined.matrix.cholesky()
    combined.status = 'cholesky_factor'
    assert relerr(combined.matrix.to_array(), factor_ref) < tol
    h = combined.matrix.solve_triangular(combined.right_hand_side, transpose=True)
    x = combined.matrix.solve_triangular(np.hstack((h, signs)))
    assert relerr(x[:, 0:1], xr) < tol
    assert relerr(x[:, 1:], mcr) < tol
    combined.compute_covariance(sparse=True)
    assert relerr(combined.matrix.to_array(), lo.to_array(ref['matrix'])) < tol_inv


def test_config5_chain_of_12_epochs_against_oracle():
    """The bench's spot check as a test: 12 epochs of the seeded config-5 system (d = 1681, N_tt = G G^T / d + 4 I, well conditioned)
    through the partitioned smoother -- solution, every diagonal and every coupling block of the inverse -- against the oracle's
    block Cholesky / sweeps / Takahashi recursion (grates/lstsq.py:698-846), 1e-12."""
    import torch
    import bench
    from grates_amd import distributed as gd
    n, d = 12, 1681
    gen = torch.Generator(device='cuda')
    sets = [bench.smoother_blocks(t, d, gen, torch, ga.engine) for t in range(n)]
    host = [(D.cpu().numpy(), R.cpu().numpy(), b.cpu().numpy()) for D, R, b in sets]
    x, zd, zu = gd.smooth_block_tridiagonal_partitioned([s[0] for s in sets], [s[1] for s in sets[:-1]], torch.cat([s[2] for s in sets], dim=0))
    bm = lo.block_matrix(np.arange(0, (n + 1) * d, d))
    for t, (D, R, _) in enumerate(host):
        bm['blocks'][(t, t)] = D.copy()
        if t + 1 < n:
            bm['blocks'][(t, t + 1)] = R.copy()
    rhs = np.vstack([h[2] for h in host])
    lo.cholesky(bm)
    ref_x = lo.solve_triangular(bm, lo.solve_triangular(bm, rhs, transpose=True))
    lo.sparse_inverse(bm)
    assert relerr(x.cpu().numpy(), ref_x) < 1e-12
    for t in range(n):
        assert relerr(zd[t].cpu().numpy(), bm['blocks'][(t, t)]) < 1e-12, t
        if t + 1 < n:
            assert relerr(zu[t].cpu().numpy(), bm['blocks'][(t, t + 1)]) < 1e-12, t


def test_sinex_normals_solved_on_device(golden, tmp_path):
    """SINEX file -> NormalEquations -> blocked Cholesky solve on the GPU (SURVEY 8f rank 3 feeding rank 1): the solution
    against numpy.linalg.solve of the matrix the reference read from the same file (tests/golden/g14_sinex.npz)."""
    g = golden('g14_sinex')
    path = tmp_path / 'normals_u.snx'
    path.write_bytes(inputs.sinex_file_text(90, 2, 8))
    for block_size in (16, 2048):                                            # 5 ragged blocks / one block
        ne = ga.io.load_normal_equations(str(path), block_size=block_size)
        assert ne.observation_count == int(g['sinex_u_obs_count'])
        np.testing.assert_array_equal(ne.to_array()[0], np.triu(g['sinex_u_N']))
        x = ne.solve()
        expected = np.linalg.solve(g['sinex_u_N'], g['sinex_u_n'])
        assert x.shape == (77, 1) and relerr(x, expected) < 1e-10            # cond(N) ~ 1e3
        assert ne.status == 'cholesky_factor' and ne.monte_carlo_vectors.shape == (77, 100)
        sigma = ne.posterior_sigma(x)
        r = g["sinex_u_lPl"][0] - (g["sinex_u_n"].T @ expected).item()
        assert abs(sigma - np.sqrt(r / (12345 - 77))) < 1e-10 * sigma


def _config5_blocks(t, d, gen, torch):
    """epoch t of the seeded full-size system: N_tt = G G^T / d + 4 I (SPD), N_t,t+1 = R / d (spectral norm ~ 2 / sqrt(d))"""
    gen.manual_seed(50_000 + t)
    G = torch.randn((d, d + 8), dtype=torch.float64, device='cuda', generator=gen)
    D = ga.engine.gemm(G, G, transb=True, alpha=1.0 / d)
    D.diagonal().add_(4.0)
    R = torch.randn((d, d), dtype=torch.float64, device='cuda', generator=gen) / d
    return D, R


def test_full_size_config5_chain():
    """BASELINE config 5 at its stated size: 3650 daily epochs of a d/o-40 state (d = 1681), VAR(1) coupling, solved with the
    1 + 100 right-hand sides of NormalEquations.solve and followed by the sparse inverse (grates/lstsq.py:950-968, 1026-1042).
    The chain is built on the device from per-epoch seeds (3 x 82.5 GB: matrix, coupling blocks, inverses of the diagonal factor
    blocks; the test is skipped, never shortened, when that does not fit) and checked through properties that
    need no reference run: the residual ||N x - n|| / ||n|| with N regenerated from the seeds, symmetry of the covariance
    blocks and the identity (N N^-1)_tt = I, which for a block-tridiagonal N only involves blocks of the sparse inverse."""
    import json
    import os
    import time
    import torch
    d, T = 1681, 3650
    free, _ = torch.cuda.mem_get_info()
    fit = int((free - 30e9) // (3 * d * d * 8))          # 30 GB for the right-hand sides, the scratch and the checks
    if fit < T:
        pytest.skip('config 5 at its stated size needs {0:.0f} GB of free device memory, {1:.0f} GB are free'.format(
            (3 * T * d * d * 8 + 30e9) / 1e9, free / 1e9))
    gen = torch.Generator(device='cuda')
    idx = np.arange(0, (T + 1) * d, d)
    bm = ls.BlockMatrix(idx, idx)
    for t in range(T):
        D, R = _config5_blocks(t, d, gen, torch)
        bm._set_device(t, t, D)
        if t + 1 < T:
            bm._set_device(t, t + 1, R)
    gen.manual_seed(49_999)
    rhs = torch.randn((T * d, 1), dtype=torch.float64, device='cuda', generator=gen)
    signs = torch.randint(0, 2, (T * d, 100), device='cuda', generator=gen).double() * 2.0 - 1.0
    ne = ls.NormalEquations(bm, rhs, 0.0, T * d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x = ne.solve(signs=signs)
    torch.cuda.synchronize()
    t_solve = time.perf_counter() - t0
    xs = torch.cat((x, ne.monte_carlo_vectors[:, :3]), dim=1)
    bs = rhs
    del signs
    ne.monte_carlo_vectors = None

    # N [x, z1, z2, z3] block row by block row against regenerated blocks (rocBLAS products): the residual of the solution, and for
    # the Monte-Carlo vectors z = W^-1 xi (upstream solves only the back substitution for them) z^T N z = |W z|^2 = |xi|^2 = T d
    num = torch.zeros((), dtype=torch.float64, device='cuda')
    quad = torch.zeros(3, dtype=torch.float64, device='cuda')
    prev = None
    for t in range(T):
        D, R = _config5_blocks(t, d, gen, torch)
        own = xs[t * d:(t + 1) * d]
        Nx = D @ own
        if t + 1 < T:
            Nx += R @ xs[(t + 1) * d:(t + 2) * d]
        if prev is not None:
            Nx += prev.t() @ xs[(t - 1) * d:t * d]
        num += ((Nx[:, 0] - bs[t * d:(t + 1) * d, 0]) ** 2).sum()
        quad += (own[:, 1:] * Nx[:, 1:]).sum(dim=0)
        prev = R
    residual = float(torch.sqrt(num) / bs[:, 0].norm())
    quad_defect = float((quad / (T * d) - 1.0).abs().max())
    assert residual < 1e-13 and quad_defect < 1e-13, (residual, quad_defect)

    t0 = time.perf_counter()
    ne.compute_covariance(sparse=True)
    torch.cuda.synchronize()
    t_inv = time.perf_counter() - t0
    eye = torch.eye(d, dtype=torch.float64, device='cuda')
    worst_sym = worst_id = 0.0
    for t in sorted({0, 1, T // 3, T // 2, T - 2, T - 1}):
        Z = bm.device_block(t, t)
        worst_sym = max(worst_sym, float((Z - Z.t()).abs().max() / Z.abs().max()))
        assert float(Z.diagonal().min()) > 0.0
        D, R = _config5_blocks(t, d, gen, torch)
        acc = D @ Z
        if t + 1 < T:
            acc += R @ bm.device_block(t, t + 1).t()
        if t > 0:
            acc += _config5_blocks(t - 1, d, gen, torch)[1].t() @ bm.device_block(t - 1, t)
        worst_id = max(worst_id, float((acc - eye).abs().max()))
    assert worst_sym < 1e-13 and worst_id < 1e-12, (worst_sym, worst_id)
    record = {'path': 'config5 full size: block-banded smoother', 'epochs': T, 'dim': d, 'order': 1, 'right_hand_sides': 101,
              'factor_and_solve_s': round(t_solve, 3), 'sparse_inverse_s': round(t_inv, 3),
              'epochs_per_s': round(T / (t_solve + t_inv), 1), 'residual': residual, 'monte_carlo_quadratic_form_defect': quad_defect,
              'covariance_asymmetry_max': worst_sym, 'identity_defect_max': worst_id}
    print(json.dumps(record))
    del bm, ne, x, xs, bs, rhs, eye, Z, acc, D, R, Nx, own
    torch.cuda.empty_cache()                              # 250 GB go back to the driver: the library allocates outside torch's cache
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, 'config5_full_size.json'), 'w') as f:
            f.write(json.dumps(record) + '\n')
    except OSError:
        pass


def test_block_table_validation():
    """The shg_block_* calls check the compressed-row table before they touch memory: missing diagonal block, columns out of
    order or out of range and NULL blocks are reported as errors (no fault, nothing launched)."""
    import torch
    eng = ga.engine
    bounds = [0, 4, 9, 12]
    blocks = {(i, j): torch.eye(bounds[i + 1] - bounds[i], bounds[j + 1] - bounds[j], dtype=torch.float64, device='cuda') * (4.0 if i == j else 0.5)
              for i, j in ((0, 0), (0, 1), (1, 1), (1, 2), (2, 2))}
    keep = [torch.zeros((s, s), dtype=torch.float64, device='cuda') for s in (4, 5, 3)]
    inverses = np.array([k.data_ptr() for k in keep], dtype=np.uint64)
    good = eng.BlockTable(bounds, blocks)
    assert eng.block_potrf(good, inverses) == 0                                   # SPD: factors
    for mutate in ('swap_columns', 'no_diagonal', 'column_range', 'null_block'):
        table = eng.BlockTable(bounds, {k: v.clone() for k, v in blocks.items()})
        if mutate == 'swap_columns':
            table.colidx[0], table.colidx[1] = table.colidx[1], table.colidx[0]
        elif mutate == 'no_diagonal':
            table.colidx[2] = 2                                                   # row 1 starts with column 2
        elif mutate == 'column_range':
            table.colidx[3] = 7
        else:
            table.address[1] = 0
        with pytest.raises(Exception):
            eng.block_potrf(table, inverses)
    x = torch.ones((12, 2), dtype=torch.float64, device='cuda')
    with pytest.raises(Exception):
        eng.block_solve(good, np.array([0, 0, 0], dtype=np.uint64), False, x)      # no scratch for the inverses


def test_blocks_with_foreign_strides_and_in_place_factor_state():
    """csrc/blockchol.hip reads every block as dense row-major.  Blocks assigned as transposed views (tensor or ndarray) are
    stored contiguously, a table built from a non-contiguous tensor is refused, right-hand sides need a contiguous last
    dimension, and a chain that keeps U_ii^-1 in place of U_ii refuses the operations that would read those blocks as U_ii."""
    import torch
    eng = ga.engine
    rng = np.random.default_rng(5)
    d = 37
    G = rng.standard_normal((2 * d, 2 * d + 3))
    N = G @ G.T / d + 2.0 * np.eye(2 * d)
    idx = np.array([0, d, 2 * d])
    bm = ls.BlockMatrix(idx, idx)
    bm[0, 0] = N[:d, :d]
    bm._set_device(0, 1, torch.from_numpy(np.ascontiguousarray(N[d:, :d])).cuda().t())       # a transposed view of N[1, 0]
    bm[1, 1] = torch.from_numpy(np.ascontiguousarray(N[d:, d:].T)).cuda().t()                # the same through __setitem__
    assert all(bm.device_block(i, j).is_contiguous() for i, j in ((0, 0), (0, 1), (1, 1)))
    rhs = rng.standard_normal((2 * d, 3))
    bm.cholesky()
    x = bm.solve_triangular(bm.solve_triangular(rhs, transpose=True))
    assert relerr(x, np.linalg.solve(N, rhs)) < TOL
    with pytest.raises(ValueError):
        eng.BlockTable(idx, {(0, 0): torch.ones((d, 2 * d), dtype=torch.float64, device='cuda')[:, ::2]})
    wide = torch.ones((2 * d, 6), dtype=torch.float64, device='cuda')
    with pytest.raises(ValueError):
        eng.block_solve(eng.BlockTable(idx, {(i, i): bm.device_block(i, i) for i in (0, 1)}), np.zeros(2, dtype=np.uint64), False, wide[:, ::2])

    # in-place factorisation (what grates_amd.distributed uses for its chains)
    chain = ls.BlockMatrix(idx, idx)
    chain._inverse_in_place = True
    chain[0, 0], chain[0, 1], chain[1, 1] = N[:d, :d], N[:d, d:], N[d:, d:]
    chain.cholesky()
    y = chain.solve_triangular(chain.solve_triangular(rhs, transpose=True))
    assert relerr(y, np.linalg.solve(N, rhs)) < TOL
    twin = chain.copy()                                                   # the copy knows what its diagonal blocks hold
    assert relerr(twin.solve_triangular(twin.solve_triangular(rhs, transpose=True)), y) < 1e-14
    for call in (lambda: chain.multiply_triangular(rhs), lambda: chain.inverse(), lambda: chain._scale(2.0), lambda: chain._axpy(1.0, bm)):
        with pytest.raises(ValueError):
            call()
    chain.sparse_inverse()
    Z = np.linalg.inv(N)
    assert relerr(chain[0, 0], Z[:d, :d]) < TOL_INV and relerr(chain[0, 1], Z[:d, d:]) < TOL_INV
    chain._scale(2.0)                                                     # an ordinary matrix again


# ---- the block-row entry points behind the partitioned smoother: look-ahead, batches of two, short-K products -------------------
def _chain(seed, epochs, d, torch):
    """a symmetric positive definite block-tridiagonal chain on the device (diagonal blocks, coupling blocks) and its dense form"""
    gen = torch.Generator(device='cuda')
    gen.manual_seed(seed)
    diag, upper = [], []
    for t in range(epochs):
        G = torch.randn((d, d + 8), dtype=torch.float64, device='cuda', generator=gen)
        diag.append(G @ G.T / d + 4.0 * torch.eye(d, dtype=torch.float64, device='cuda'))
        if t + 1 < epochs:
            upper.append(torch.randn((d, d), dtype=torch.float64, device='cuda', generator=gen) / d)
    dense = torch.zeros((epochs * d, epochs * d), dtype=torch.float64, device='cuda')
    for t in range(epochs):
        dense[t * d:(t + 1) * d, t * d:(t + 1) * d] = diag[t]
        if t + 1 < epochs:
            dense[t * d:(t + 1) * d, (t + 1) * d:(t + 2) * d] = upper[t]
            dense[(t + 1) * d:(t + 2) * d, t * d:(t + 1) * d] = upper[t].T
    return diag, upper, dense


def _chain_matrix(diag, upper, in_place):
    d = diag[0].shape[0]
    idx = np.arange(0, (len(diag) + 1) * d, d)
    bm = ls.BlockMatrix(idx, idx)
    bm._inverse_in_place = in_place
    for t, b in enumerate(diag):
        bm._set_device(t, t, b.clone())
    for t, b in enumerate(upper):
        bm._set_device(t, t + 1, b.clone())
    return bm


@pytest.mark.parametrize('d', [96, 300, 385, 1681])    # one leaf | look-ahead with three panels | with a ragged fourth one | config 5: 14 panels
def test_factorisation_with_and_without_lookahead(d):
    """shg_block_potrf_rows: the panel sweep with its side streams -- chain rows carrying W and the Schur complement along ('carry',
    forced whatever the queue experiment found) and not ('plain') -- and the automatic choice against the recursive sweep
    ('off') and against the dense Cholesky factor, 1e-12"""
    import torch
    from grates_amd import engine
    epochs = 4 if d < 1000 else 3
    diag, upper, dense = _chain(11, epochs, d, torch)
    reference = torch.linalg.cholesky(dense, upper=True)
    factors = []
    for mode in ('carry', 'plain', 'on', 'off'):
        engine.block_set_lookahead(mode)
        try:
            info = engine.block_lookahead_info()
            assert info['mode'] == engine.LOOKAHEAD_MODES[mode] and 0 <= info['side_queues_apart'] <= 2
            assert info['chain_rows_carry_coupling'] == {'carry': True, 'plain': False, 'off': False, 'on': info['side_queues_apart'] >= 2}[mode]
            bm = _chain_matrix(diag, upper, False)
            bm.cholesky()
        finally:
            engine.block_set_lookahead(True)
        blocks = [bm.device_block(t, t) for t in range(epochs)] + [bm.device_block(t, t + 1) for t in range(epochs - 1)]
        factors.append(blocks)
        for t in range(epochs):
            assert relerr(bm.device_block(t, t).cpu().numpy(), reference[t * d:(t + 1) * d, t * d:(t + 1) * d].cpu().numpy()) < 1e-12, mode
            if t + 1 < epochs:
                assert relerr(bm.device_block(t, t + 1).cpu().numpy(), reference[t * d:(t + 1) * d, (t + 1) * d:(t + 2) * d].cpu().numpy()) < 1e-12, mode
    for other in factors[1:]:
        for a, b in zip(factors[0], other):
            assert relerr(a.cpu().numpy(), b.cpu().numpy()) < 1e-12


@pytest.mark.parametrize('sizes', [(512, 200, 512), (385, 130, 300, 257, 96, 385), (300, 256, 300, 300)])
@pytest.mark.parametrize('in_place', [False, True])
def test_factorisation_of_mixed_block_sizes(sizes, in_place):
    """A chain whose block rows alternate between the panel sweep with its look-ahead (d > 256: the side stream is still growing
    the inverse of row r when row r + 1 starts) and the recursive sweep on the caller's stream (d <= 256): the two must not share
    a work area (advisor r03).  Factor, both inverses' products and the sparse inverse against the dense matrix, 1e-11."""
    import torch
    gen = torch.Generator(device='cuda')
    gen.manual_seed(sum(sizes))
    n = len(sizes)
    bounds = np.concatenate(([0], np.cumsum(sizes)))
    total = int(bounds[-1])
    diag = []
    for d in sizes:
        G = torch.randn((d, d + 8), dtype=torch.float64, device='cuda', generator=gen)
        diag.append(G @ G.T / d + 4.0 * torch.eye(d, dtype=torch.float64, device='cuda'))
    upper = [torch.randn((sizes[t], sizes[t + 1]), dtype=torch.float64, device='cuda', generator=gen) / max(sizes) for t in range(n - 1)]
    dense = torch.zeros((total, total), dtype=torch.float64, device='cuda')
    for t in range(n):
        dense[bounds[t]:bounds[t + 1], bounds[t]:bounds[t + 1]] = diag[t]
        if t + 1 < n:
            dense[bounds[t]:bounds[t + 1], bounds[t + 1]:bounds[t + 2]] = upper[t]
            dense[bounds[t + 1]:bounds[t + 2], bounds[t]:bounds[t + 1]] = upper[t].T
    reference = torch.linalg.cholesky(dense, upper=True).cpu().numpy()
    Z = torch.linalg.inv(dense).cpu().numpy()
    for repeat in range(3):                                                     # (a race shows up in some runs only)
        bm = ls.BlockMatrix(bounds, bounds)
        bm._inverse_in_place = in_place
        for t, b in enumerate(diag):
            bm._set_device(t, t, b.clone())
        for t, b in enumerate(upper):
            bm._set_device(t, t + 1, b.clone())
        bm.cholesky()
        rhs = torch.randn((total, 3), dtype=torch.float64, device='cuda', generator=gen)
        x = bm.solve_triangular(bm.solve_triangular(rhs, transpose=True))
        assert relerr(x.cpu().numpy(), torch.linalg.solve(dense, rhs).cpu().numpy()) < 1e-11, repeat
        for t in range(n):
            if t + 1 < n:
                assert relerr(bm.device_block(t, t + 1).cpu().numpy(), reference[bounds[t]:bounds[t + 1], bounds[t + 1]:bounds[t + 2]]) < 1e-11, (repeat, t)
        bm.sparse_inverse()
        for t in range(n):
            assert relerr(bm.device_block(t, t).cpu().numpy(), Z[bounds[t]:bounds[t + 1], bounds[t]:bounds[t + 1]]) < 1e-11, (repeat, t)


@pytest.mark.parametrize('d,in_place', [(300, True), (300, False), (64, True)])
def test_pair_factorisation_matches_single(d, in_place):
    """shg_block_potrf_rows_pair: two chains of one structure in one pass (every launch a batch of two) against the same two chains
    factored one after the other; the longer chain finishes its further row alone"""
    import torch
    pairs = [_chain(21, 5, d, torch), _chain(22, 4, d, torch)]
    single = []
    for diag, upper, _ in pairs:
        bm = _chain_matrix(diag, upper, in_place)
        bm._cholesky_rows(0, len(diag))
        single.append(bm)
    a, b = (_chain_matrix(diag, upper, in_place) for diag, upper, _ in pairs)
    a._cholesky_rows_pair(b, 0, 3)                       # rows 0 .. 2 of both; block 3 of each is left as Schur complement
    a._cholesky_rows(3, 5)
    b._cholesky_rows(3, 4)
    for one, two, n in ((single[0], a, 5), (single[1], b, 4)):
        for t in range(n):
            assert relerr(two.device_block(t, t).cpu().numpy(), one.device_block(t, t).cpu().numpy()) < 1e-12
            if t + 1 < n:
                assert relerr(two.device_block(t, t + 1).cpu().numpy(), one.device_block(t, t + 1).cpu().numpy()) < 1e-12
    # and the factor is a factor: W^T W = N through the solve of the in-place form
    diag, upper, dense = pairs[0]
    rhs = torch.ones((5 * d, 1), dtype=torch.float64, device='cuda')
    x = a.solve_triangular(a.solve_triangular(rhs.clone(), transpose=True))
    x = x if hasattr(x, 'cpu') else torch.as_tensor(x, device='cuda')
    assert float((dense @ x.reshape(-1, 1) - rhs).abs().max()) < 1e-10


@pytest.mark.parametrize('shape', [(128, 128, 128), (17, 1553, 128), (1553, 128, 96), (128, 300, 64), (100, 100, 100), (1681, 1, 1681), (300, 3, 700),
                                   (700, 8, 257)])
def test_short_k_products(shape):
    """products with K <= 128 and a thin output go through the panel kernel, products with a handful of columns through the streaming
    kernels (csrc/blas.hip): against torch, 1e-13"""
    import torch
    from grates_amd import engine
    M, N, K = shape
    gen = torch.Generator(device='cuda')
    gen.manual_seed(5)
    for transa in (False, True):
        A = torch.randn((K, M) if transa else (M, K), dtype=torch.float64, device='cuda', generator=gen)
        B = torch.randn((K, N), dtype=torch.float64, device='cuda', generator=gen)
        C = torch.randn((M, N), dtype=torch.float64, device='cuda', generator=gen)
        expected = 0.5 * C - 2.0 * ((A.T if transa else A) @ B)
        engine.gemm(A, B, transa=transa, alpha=-2.0, beta=0.5, out=C)
        assert relerr(C.cpu().numpy(), expected.cpu().numpy()) < 1e-13


@pytest.mark.parametrize('n', [1, 31, 32, 33, 100, 1681])
def test_transpose_in_place(n):
    """shg_transpose_in_place (the coupling blocks of a chain walked backwards): bit-exact against torch, also inside a wider array"""
    import torch
    from grates_amd import engine
    A = torch.randn((n, n), dtype=torch.float64, device='cuda')
    expected = A.t().clone()
    assert torch.equal(engine.transpose_in_place(A), expected)
    wide = torch.randn((n, n + 5), dtype=torch.float64, device='cuda')
    before = wide.clone()
    engine.transpose_in_place(wide[:, :n])
    assert torch.equal(wide[:, :n], before[:, :n].t()) and torch.equal(wide[:, n:], before[:, n:])
