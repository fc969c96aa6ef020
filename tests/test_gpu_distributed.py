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


@pytest.mark.parametrize('world,epochs,dim', [(2, 7, 40), (3, 8, 130), (3, 6, 257), (2, 21, 48)])
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


@pytest.mark.parametrize('sizes', [[5, 9, 130, 7, 64, 3, 17, 40], [33] * 11, [140, 20, 1, 75, 16, 16, 90, 12, 129]])
def test_two_ended_chain_matches_natural_order(sizes):
    """_TwistedChain (both halves of the chain eliminated at once on two streams) against the natural-order chain and the
    dense inverse, ragged block sizes, even and odd epoch counts."""
    from grates_amd import distributed as gd
    rng = np.random.default_rng(len(sizes))
    n = len(sizes)
    bounds = np.concatenate(([0], np.cumsum(sizes)))
    diag = [rng.standard_normal((d, d + 4)) for d in sizes]
    diag = [G @ G.T / G.shape[0] + 3.0 * np.eye(G.shape[0]) for G in diag]
    upper = [rng.standard_normal((sizes[t], sizes[t + 1])) * (0.4 / np.sqrt(max(sizes[t], sizes[t + 1]))) for t in range(n - 1)]
    N = np.zeros((bounds[-1], bounds[-1]))
    for t in range(n):
        N[bounds[t]:bounds[t + 1], bounds[t]:bounds[t + 1]] = diag[t]
        if t + 1 < n:
            N[bounds[t]:bounds[t + 1], bounds[t + 1]:bounds[t + 2]] = upper[t]
            N[bounds[t + 1]:bounds[t + 2], bounds[t]:bounds[t + 1]] = upper[t].T
    rhs = rng.standard_normal((bounds[-1], 4))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dd, du, db = [dev(b) for b in diag], [dev(b) for b in upper], dev(rhs)
    keep = [b.clone() for b in dd + du]
    results = []
    for cls in (gd._TwistedChain, gd._Chain):
        chain = cls(dd, du)
        chain.factor()
        x = chain.solve(db).cpu().numpy()
        zd, zu = chain.sparse_inverse()
        results.append((x, [b.cpu().numpy() for b in zd], [b.cpu().numpy() for b in zu]))
    assert all(torch.equal(a, b) for a, b in zip(keep, dd + du))               # the caller's blocks are not touched
    Z = np.linalg.inv(N)
    scale = np.abs(Z).max()
    for x, zd, zu in results:
        assert relerr(x, np.linalg.solve(N, rhs)) < 1e-11
        assert len(zd) == n and len(zu) == n - 1
        for t in range(n):
            assert np.abs(zd[t] - Z[bounds[t]:bounds[t + 1], bounds[t]:bounds[t + 1]]).max() < 1e-11 * scale
            if t + 1 < n:
                assert np.abs(zu[t] - Z[bounds[t]:bounds[t + 1], bounds[t + 1]:bounds[t + 2]]).max() < 1e-11 * scale
    assert relerr(results[0][0], results[1][0]) < 1e-12


def _dense_chain(diag, upper):
    bounds = np.concatenate(([0], np.cumsum([b.shape[0] for b in diag])))
    N = np.zeros((bounds[-1], bounds[-1]))
    for t, D in enumerate(diag):
        N[bounds[t]:bounds[t + 1], bounds[t]:bounds[t + 1]] = D
        if t + 1 < len(diag):
            N[bounds[t]:bounds[t + 1], bounds[t + 1]:bounds[t + 2]] = upper[t]
            N[bounds[t + 1]:bounds[t + 2], bounds[t]:bounds[t + 1]] = upper[t].T
    return N, bounds


@pytest.mark.parametrize('segments,epochs,dim', [(2, 9, 48), (3, 11, 40), (4, 17, 130), (6, 40, 33)])
def test_segmented_chain_single_rank(segments, epochs, dim):
    """The segmented elimination on one rank (first / last segments eliminated from their free ends, segments between separators with
    both separators as their last block rows and the coupling to the left one carried along as fill-in), every segment on its
    own stream: solution and covariance blocks against the dense solve / inverse."""
    from grates_amd import distributed as gd
    diag, upper, rhs = _system(epochs, dim, 3, seed=epochs)
    upper = upper[:epochs - 1]
    N, bounds = _dense_chain(diag, upper)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dd, du, db = [dev(b) for b in diag], [dev(b) for b in upper], dev(rhs)
    keep = [b.clone() for b in dd + du]
    sc = gd._SegmentedChain(dd, du, db, None, consume=False, segments=segments)
    assert len(sc.segs) == segments and [s['kind'] for s in sc.segs] == ['first'] + ['middle'] * (segments - 2) + ['last']
    assert sc.segs[0]['hi'] - sc.segs[0]['lo'] >= sc.segs[1]['hi'] - sc.segs[1]['lo'] or segments == 2
    x = sc.solve().cpu().numpy()
    zd, zu = sc.covariance()
    assert all(torch.equal(a, b) for a, b in zip(keep, dd + du))                                  # consume=False: the caller's blocks stay
    assert relerr(x, np.linalg.solve(N, rhs)) < 1e-11
    Z = np.linalg.inv(N)
    scale = np.abs(Z).max()
    assert len(zd) == epochs and len(zu) == epochs - 1
    for t in range(epochs):
        assert np.abs(zd[t].cpu().numpy() - Z[bounds[t]:bounds[t + 1], bounds[t]:bounds[t + 1]]).max() < 1e-11 * scale, t
        if t + 1 < epochs:
            assert np.abs(zu[t].cpu().numpy() - Z[bounds[t]:bounds[t + 1], bounds[t + 1]:bounds[t + 2]]).max() < 1e-11 * scale, t
    # covariance only (no right-hand side), through the entry point, blocks consumed
    zd2, zu2 = gd._SegmentedChain([b.clone() for b in dd], [b.clone() for b in du], None, None, consume=True, segments=segments).covariance()
    assert all(torch.equal(a, b) for a, b in zip(zd, zd2)) and all(torch.equal(a, b) for a, b in zip(zu, zu2))
    with pytest.raises(np.linalg.LinAlgError):
        broken = [b.clone() for b in dd]
        broken[epochs // 2] -= 10.0 * torch.eye(dim, dtype=torch.float64, device='cuda')
        gd._SegmentedChain(broken, du, db, None, segments=segments)


def test_two_ended_chain_reports_indefinite_blocks():
    from grates_amd import distributed as gd
    d = [torch.eye(4, dtype=torch.float64, device='cuda') * (1.0 if t != 6 else -1.0) for t in range(9)]
    u = [torch.zeros((4, 4), dtype=torch.float64, device='cuda') for _ in range(8)]
    with pytest.raises(np.linalg.LinAlgError):
        gd._TwistedChain(d, u).factor()


def test_full_size_config5_product_path():
    """BASELINE config 5 at its stated size through the product entry point: 3650 daily epochs of a d/o-40 state (d = 1681),
    solution and covariance blocks from ONE factorisation (smooth_block_tridiagonal_partitioned, world size 1: the chain is
    eliminated from both ends at once on two streams).  The factorisation works in the caller's blocks (consume=True) and keeps
    the inverses of the diagonal factor blocks in those blocks' own storage: 2 x 82.5 GB (skipped, never shortened, on a card
    with less free memory).
    Checked through properties that need no reference run, with the blocks regenerated from their seeds: the residual of the
    solution, symmetry of the covariance blocks and (N N^-1)_tt = I at sample epochs including the ones around the meeting point."""
    import json
    import time
    from grates_amd import distributed as gd
    from test_gpu_lstsq import _config5_blocks
    d, T = 1681, 3650
    free, _ = torch.cuda.mem_get_info()
    if int((free - 30e9) // (2 * d * d * 8)) < T:
        pytest.skip('config 5 at its stated size needs {0:.0f} GB of free device memory, {1:.0f} GB are free'.format(
            (2 * T * d * d * 8 + 30e9) / 1e9, free / 1e9))
    gen = torch.Generator(device='cuda')
    gen.manual_seed(49_999)
    rhs = torch.randn((T * d, 1), dtype=torch.float64, device='cuda', generator=gen)
    # two runs (a first call and a repeated one: nothing is allocated per epoch, so they should take the same time)
    seconds = []
    x = zd = zu = None
    for run in range(2):
        x = zd = zu = None                                # the blocks of the first run go back to the allocator before the second set is built
        diag, upper = [], []
        for t in range(T):
            D, R = _config5_blocks(t, d, gen, torch)
            diag.append(D)
            if t + 1 < T:
                upper.append(R)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x, zd, zu = gd.smooth_block_tridiagonal_partitioned(diag, upper, rhs, consume=True)
        torch.cuda.synchronize()
        seconds.append(time.perf_counter() - t0)
        del diag, upper
    elapsed = seconds[1]
    assert len(zd) == T and len(zu) == T - 1

    num = torch.zeros((), dtype=torch.float64, device='cuda')
    prev = None
    for t in range(T):
        D, R = _config5_blocks(t, d, gen, torch)
        Nx = D @ x[t * d:(t + 1) * d]
        if t + 1 < T:
            Nx += R @ x[(t + 1) * d:(t + 2) * d]
        if prev is not None:
            Nx += prev.t() @ x[(t - 1) * d:t * d]
        num += ((Nx - rhs[t * d:(t + 1) * d]) ** 2).sum()
        prev = R
    residual = float(torch.sqrt(num) / rhs.norm())
    assert residual < 1e-13, residual

    eye = torch.eye(d, dtype=torch.float64, device='cuda')
    worst_sym = worst_id = 0.0
    m = T // 2
    for t in sorted({0, 1, T // 3, m - 1, m, m + 1, T - 2, T - 1}):
        Z = zd[t]
        worst_sym = max(worst_sym, float((Z - Z.t()).abs().max() / Z.abs().max()))
        assert float(Z.diagonal().min()) > 0.0
        D, R = _config5_blocks(t, d, gen, torch)
        acc = D @ Z
        if t + 1 < T:
            acc += R @ zu[t].t()
        if t > 0:
            acc += _config5_blocks(t - 1, d, gen, torch)[1].t() @ zu[t - 1]
        worst_id = max(worst_id, float((acc - eye).abs().max()))
    assert worst_sym < 1e-13 and worst_id < 1e-12, (worst_sym, worst_id)
    record = {'path': 'config5 full size: smooth_block_tridiagonal_partitioned, one rank, two-ended elimination', 'epochs': T, 'dim': d,
              'right_hand_sides': 1, 'first_call_s': round(seconds[0], 3), 'factor_solve_sparse_inverse_s': round(elapsed, 3),
              'epochs_per_s': round(T / elapsed, 1),
              'residual': residual, 'covariance_asymmetry_max': worst_sym, 'identity_defect_max': worst_id}
    print(json.dumps(record))
    del x, zd, zu, Z, acc, D, R, Nx, prev, eye
    torch.cuda.empty_cache()
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, 'config5_two_ended.json'), 'w') as f:
            f.write(json.dumps(record) + '\n')
    except OSError:
        pass


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
    # epoch-sharded analysis of those grids and epoch-sharded filters (block form and dense form), SURVEY 8(e) row 1
    whole = ga.engine.to_host(ga.gravityfield.synthesize(batch, grid, kernel='ewh'))
    a0, a1, anm = gd.analysis_sharded(whole, grid, 0, 20, kernel='ewh')
    flt = ga.filter.OrderWiseFilter(inputs.orderwise_random_blocks(42, 20))
    f0, f1, filtered = gd.filter_sharded(flt, batch)
    g0, g1, dense = gd.filter_sharded(ga.filter.GeneralMatrix(flt.matrix(2, 20), 2, 20), torch.from_numpy(batch).cuda())
    assert (a0, a1) == (e0, e1) == (f0, f1) == (g0, g1)
    np.savez(os.path.join(result_dir, 'epochs_{0}.npz'.format(rank)), span=[e0, e1], anm=anm.cpu().numpy(), filtered=filtered.cpu().numpy(),
             dense=dense.cpu().numpy())
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
    # sharded analysis / filters: every rank's slice equals the slice of the single-process result, bit for bit
    plan_anm = ga.engine.to_host(grid._plan('ewh', 20, 3.9860044150e+14, 6.3781363000e+06).analysis(whole, grid.area.reshape(whole.shape[1:]), 0))
    flt = ga.filter.OrderWiseFilter(inputs.orderwise_random_blocks(42, 20))
    all_filtered = ga.engine.to_host(flt.filter_batch(batch))
    all_dense = ga.engine.to_host(ga.filter.GeneralMatrix(flt.matrix(2, 20), 2, 20).filter_batch(batch))
    assert relerr(plan_anm, batch) < 1e-11                                     # band-limited fields come back
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), 'epochs_{0}.npz'.format(r)))
        e0, e1 = z['span']
        np.testing.assert_array_equal(z['anm'], plan_anm[e0:e1])
        np.testing.assert_array_equal(z['filtered'], all_filtered[e0:e1])
        np.testing.assert_array_equal(z['dense'], all_dense[e0:e1])


def test_bench_rehearsal_world2_and_world4_real_shard_sizes():
    """The 2- and 4-way shardings of BASELINE configs 4 and 5 at their real per-rank sizes, rehearsed on ONE card (two / four rank
    processes under gloo, all mapped to cuda:0; with RCCL each rank has its own GPU) through bench.py itself:
      * covariance: the whole 360 x 720 grid at d/o 180 in four latitude bands of 90 parallels, Sigma (8.6 GB) replicated per rank
        -- the gathered sigma must be bit-identical to the single-process result (checksum and CRC);
      * smoother: d = 1681 with 128 epochs per rank (512 epochs), nested dissection with all_gathers of the separator blocks
        -- solution and covariance blocks are checked inside bench.py (residual over the whole chain across the rank boundaries,
        (N N^-1)_tt = I), the solution checksum must agree with the single-chain run to rounding.
    World size 2 runs the smoother only (256 epochs per rank).  The record (per-rank shard sizes, gathered payload, times) goes to
    gpurun_out/r3_rehearsal_world4.json."""
    import json
    import subprocess
    import sys
    free, _ = torch.cuda.mem_get_info()
    if free < 150e9:
        pytest.skip('four replicas of the d/o-180 covariance matrix and a 512-epoch chain need ~150 GB of free device memory')
    torch.cuda.empty_cache()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = [sys.executable, os.path.join(root, 'bench.py'), '--legs', 'covariance,smoother', '--smoother-epochs', '512', '--smoother-repeats', '1',
              '--cpu-sample', '0', '--cov-repeats', '1', '--cov-extensions', '0', '--steps', '2', '--warmup', '1', '--ramp', '0', '--epochs', '8']
    lines = {}
    for world in (1, 2, 4):
        extra = [] if world == 1 else ['--backend', 'gloo', '--same-device']
        if world == 2:
            extra += ['--legs', 'smoother']                   # (argparse: the last --legs wins)
        env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
        run = subprocess.run(common + ['--gpus', str(world)] + extra, env=env, capture_output=True, text=True, timeout=900)
        assert run.returncode == 0, run.stderr[-2000:]
        lines[world] = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith('{')][-1])
    one, two, four = lines[1], lines[2], lines[4]
    assert four['n_gpus'] == 4 and four['all_checks_ok'] and one['all_checks_ok'] and two['n_gpus'] == 2 and two['all_checks_ok']
    assert two['smoother']['config']['epochs_per_rank'] == [256] * 2 and two['smoother']['check']['residual'] < 1e-13
    assert abs(two['smoother']['check']['solution_checksum'] - one['smoother']['check']['solution_checksum']) < 1e-9 * max(abs(one['smoother']['check']['solution_checksum']), 1.0)
    assert four['covariance']['sigma_crc32'] == one['covariance']['sigma_crc32']
    assert four['covariance']['sigma_checksum'] == one['covariance']['sigma_checksum']
    assert four['smoother']['config']['epochs_per_rank'] == [128] * 4
    assert four['smoother']['check']['residual'] < 1e-13 and four['smoother']['check']['identity_defect_max'] < 1e-12
    a, b = one['smoother']['check']['solution_checksum'], four['smoother']['check']['solution_checksum']
    assert abs(a - b) < 1e-9 * max(abs(a), 1.0)
    d = 1681
    record = {'what': 'bench.py --gpus 4 --backend gloo --same-device: four rank processes on one card, real per-rank shard sizes',
              'covariance': {'bands': four['covariance']['config']['workload'], 'seconds_world4_one_card': four['covariance']['seconds_median'],
                             'seconds_world1': one['covariance']['seconds_median'], 'sigma_crc32': four['covariance']['sigma_crc32'],
                             'gathered_bytes_per_rank': 90 * 720 * 8},
              'smoother': {'epochs_per_rank': four['smoother']['config']['epochs_per_rank'], 'dim': d,
                           'gathered_bytes_per_rank': (5 * d * d + 3 * d + d * d) * 8,
                           'seconds_world4_one_card': four['smoother']['seconds'], 'phases_world4': four['smoother']['phases_s'],
                           'seconds_world1': one['smoother']['seconds'], 'phases_world1': one['smoother']['phases_s'],
                           'seconds_world2_one_card': two['smoother']['seconds'], 'phases_world2': two['smoother']['phases_s'],
                           'residual_world2': two['smoother']['check']['residual'],
                           'residual_world4': four['smoother']['check']['residual'], 'identity_defect_world4': four['smoother']['check']['identity_defect_max'],
                           'solution_checksum_world1': a, 'solution_checksum_world4': b}}
    print(json.dumps(record))
    out_dir = os.path.join(root, 'gpurun_out')
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, 'r3_rehearsal_world4.json'), 'w') as f:
            f.write(json.dumps(record) + '\n')
    except OSError:
        pass
