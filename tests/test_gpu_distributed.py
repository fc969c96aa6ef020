"""
Epoch-partitioned block-tridiagonal smoother solve (BASELINE config 5) on the GPU: 2 and 3 ranks rehearsed on ONE card
with the gloo backend (every rank maps to cuda:0; with RCCL each rank has its own GPU), compared with the single-process
block Cholesky solve of grates_amd.lstsq and with the NumPy oracle.  Tolerance 1e-10 relative (two different elimination
orders of a system with condition number ~1e3).
"""

import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import relerr

pytestmark = pytest.mark.gpu


def _system(epochs, dim, columns, seed=3):
    """SPD block-tridiagonal system: diagonal blocks G G^T / dim + 3 I, coupling blocks small random"""
    rng = np.random.default_rng(seed)
    diag, upper = [], []
    for t in range(epochs):
        G = rng.standard_normal((dim, dim + 4))
        diag.append(G @ G.T / dim + 3.0 * np.eye(dim))
        upper.append(rng.standard_normal((dim, dim)) * (0.4 / np.sqrt(dim)))
    rhs = rng.standard_normal((epochs * dim, columns))
    return diag, upper, rhs


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, epochs, dim, columns, result_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import grates_amd as ga
    from grates_amd import distributed as gd
    gd.init('gloo')
    diag, upper, rhs = _system(epochs, dim, columns)
    t0, t1 = gd.shard_range(epochs, rank, world)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    x = gd.solve_block_tridiagonal_partitioned([dev(b) for b in diag[t0:t1]], [dev(b) for b in upper[t0:t1]], dev(rhs[t0 * dim:t1 * dim]))
    np.save(os.path.join(result_dir, 'x_{0}.npy'.format(rank)), x.cpu().numpy())
    Zd, Zu = gd.sparse_inverse_block_tridiagonal_partitioned([dev(b) for b in diag[t0:t1]], [dev(b) for b in upper[t0:t1]])
    assert len(Zd) == t1 - t0 and len(Zu) == (t1 - t0 if rank < world - 1 else t1 - t0 - 1)
    np.save(os.path.join(result_dir, 'zd_{0}.npy'.format(rank)), np.stack([b.cpu().numpy() for b in Zd]))
    np.save(os.path.join(result_dir, 'zu_{0}.npy'.format(rank)), np.stack([b.cpu().numpy() for b in Zu]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('world,epochs,dim', [(2, 7, 40), (3, 8, 130), (3, 6, 257)])
def test_partitioned_smoother_solve(world, epochs, dim, tmp_path):
    columns = 3
    mp.spawn(_worker, args=(world, _free_port(), epochs, dim, columns, str(tmp_path)), nprocs=world, join=True)
    x = np.vstack([np.load(os.path.join(str(tmp_path), 'x_{0}.npy'.format(r))) for r in range(world)])
    diag, upper, rhs = _system(epochs, dim, columns)
    N = np.zeros((epochs * dim, epochs * dim))
    for t in range(epochs):
        N[t * dim:(t + 1) * dim, t * dim:(t + 1) * dim] = diag[t]
        if t + 1 < epochs:
            N[t * dim:(t + 1) * dim, (t + 1) * dim:(t + 2) * dim] = upper[t]
            N[(t + 1) * dim:(t + 2) * dim, t * dim:(t + 1) * dim] = upper[t].T
    ref = np.linalg.solve(N, rhs)
    assert relerr(x, ref) < 1e-10
    # single process, same entry point
    import grates_amd as ga
    from grates_amd import distributed as gd
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    one = gd.solve_block_tridiagonal_partitioned([dev(b) for b in diag], [dev(b) for b in upper], dev(rhs)).cpu().numpy()
    assert relerr(one, ref) < 1e-10
    # covariance blocks of the partitioned sparse inverse against the dense inverse
    Zref = np.linalg.inv(N)
    zd = np.concatenate([np.load(os.path.join(str(tmp_path), 'zd_{0}.npy'.format(r))) for r in range(world)])
    zu = np.concatenate([np.load(os.path.join(str(tmp_path), 'zu_{0}.npy'.format(r))) for r in range(world)])
    assert zd.shape == (epochs, dim, dim) and zu.shape == (epochs - 1, dim, dim)
    scale = np.abs(Zref).max()
    for t in range(epochs):
        assert np.abs(zd[t] - Zref[t * dim:(t + 1) * dim, t * dim:(t + 1) * dim]).max() < 1e-10 * scale
        if t + 1 < epochs:
            assert np.abs(zu[t] - Zref[t * dim:(t + 1) * dim, (t + 1) * dim:(t + 2) * dim]).max() < 1e-10 * scale
    Zd1, Zu1 = gd.sparse_inverse_block_tridiagonal_partitioned([dev(b) for b in diag], [dev(b) for b in upper])
    assert relerr(np.stack([b.cpu().numpy() for b in Zd1]), zd) < 1e-10 and relerr(np.stack([b.cpu().numpy() for b in Zu1]), zu) < 1e-10


def test_partitioned_smoother_single_process_single_epoch():
    from grates_amd import distributed as gd
    import torch.distributed as dist
    assert not dist.is_initialized()
    d = [torch.eye(3, dtype=torch.float64, device='cuda')]
    x = gd.solve_block_tridiagonal_partitioned(d, [None], torch.ones((3, 1), dtype=torch.float64, device='cuda'))
    assert relerr(x.cpu().numpy(), np.ones((3, 1))) < 1e-14


# ------------------------------------------------------------------------------------------------
# latitude-band sharded covariance propagation and epoch-sharded synthesis, 3 ranks on one card (gloo)
# ------------------------------------------------------------------------------------------------
def _shard_worker(rank, world, port, result_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import inputs
    import grates_amd as ga
    from grates_amd import distributed as gd
    gd.init('gloo')
    N, nmin = 24, 2
    grid = ga.grid.GeographicGrid(5.0, 4.0)                     # 45 parallels: uneven bands over 3 ranks
    cov = inputs.spd_covariance(77, (N + 1) ** 2 - nmin ** 2)
    for method in ('direct', 'separable'):
        full = gd.covariance_propagation_sharded(grid, cov, nmin, N, kernel='ewh', method=method)
        if rank == 0:
            np.save(os.path.join(result_dir, 'sigma_{0}.npy'.format(method)), full.cpu().numpy())
    batch = np.stack([inputs.coefficients(300 + e, 20) for e in range(7)])
    e0, e1, grids = gd.synthesize_sharded(batch, grid, kernel='ewh')
    np.save(os.path.join(result_dir, 'grids_{0}.npy'.format(rank)), np.concatenate(([e0, e1], grids.cpu().numpy().ravel())))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_sharded_covariance_and_synthesis(tmp_path):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import inputs
    import grates_amd as ga
    world = 3
    mp.spawn(_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    N, nmin = 24, 2
    grid = ga.grid.GeographicGrid(5.0, 4.0)
    cov = inputs.spd_covariance(77, (N + 1) ** 2 - nmin ** 2)
    ref = grid.covariance_propagation(cov, nmin, N, kernel='ewh')
    np.testing.assert_array_equal(np.load(os.path.join(str(tmp_path), 'sigma_direct.npy')), ref)      # band results are bit-reproducible
    assert relerr(np.load(os.path.join(str(tmp_path), 'sigma_separable.npy')), ref) < 1e-12
    batch = np.stack([inputs.coefficients(300 + e, 20) for e in range(7)])
    whole = ga.engine.to_host(ga.gravityfield.synthesize(batch, grid, kernel='ewh'))
    seen = []
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), 'grids_{0}.npy'.format(r)))
        e0, e1 = int(d[0]), int(d[1])
        seen.append((e0, e1))
        np.testing.assert_array_equal(d[2:].reshape(e1 - e0, *whole.shape[1:]), whole[e0:e1])
    assert seen[0][0] == 0 and seen[-1][1] == 7 and all(a[1] == b[0] for a, b in zip(seen, seen[1:]))
