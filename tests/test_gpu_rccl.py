"""
RCCL test of the sharded paths: fresh rank processes, one GPU each, started through `python -m torch.distributed.run` BEFORE
anything in them has touched a GPU, backend "nccl".  World size 1 runs on the one-GPU box of every round: a launcher-started
rank creates its process group (device_id init, communicator set-up) and every collective of the sharded entry points runs on
device buffers over RCCL; world size 2 needs two GPUs (the gloo rehearsals of tests/test_gpu_distributed.py cover several
ranks on one card).  The second test drives bench.py's rank function the same way.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import relerr

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _launcher_env():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    return env


@pytest.mark.parametrize('world', [1, 2])
def test_sharded_paths_over_rccl(world, tmp_path):
    if torch.cuda.device_count() < world:
        pytest.skip('needs {0} GPUs'.format(world))
    env = _launcher_env()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(HERE, 'rccl_worker.py'), str(tmp_path)]
    done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-4000:]
    record = json.load(open(tmp_path / 'collectives_0.json'))
    assert record['backend'] == 'nccl' and record['world'] == world
    # every collective of the sharded paths ran, on device buffers: sigma bands (2 methods), the segment agreement (all_reduce),
    # separator blocks and boundary couplings of the partitioned smoother
    assert record['all_gather'] >= 2 + 3 and record['all_reduce'] >= 2 and record['on_device'] and not record['on_host']

    sys.path.insert(0, os.path.join(HERE, 'golden'))
    import inputs
    import grates_amd as ga
    N, nmin = 24, 2
    grid = ga.grid.GeographicGrid(5.0, 4.0)
    cov = inputs.spd_covariance(77, (N + 1) ** 2 - nmin ** 2)
    ref = grid.covariance_propagation(cov, nmin, N, kernel='ewh')
    np.testing.assert_array_equal(np.load(tmp_path / 'sigma_direct.npy'), ref)          # band results are bit-reproducible
    assert relerr(np.load(tmp_path / 'sigma_separable.npy'), ref) < 1e-12
    batch = np.stack([inputs.coefficients(300 + e, 20) for e in range(7)])
    whole = ga.engine.to_host(ga.gravityfield.synthesize(batch, grid, kernel='ewh'))
    for r in range(world):
        d = np.load(tmp_path / 'grids_{0}.npy'.format(r))
        e0, e1 = int(d[0]), int(d[1])
        np.testing.assert_array_equal(d[2:].reshape(e1 - e0, *whole.shape[1:]), whole[e0:e1])

    epochs, dim, columns = 8, 130, 3
    rng = np.random.default_rng(3)
    Nm = np.zeros((epochs * dim, epochs * dim))
    for t in range(epochs):
        G = rng.standard_normal((dim, dim + 4))
        Nm[t * dim:(t + 1) * dim, t * dim:(t + 1) * dim] = G @ G.T / dim + 3.0 * np.eye(dim)
        U = rng.standard_normal((dim, dim)) * (0.4 / np.sqrt(dim))
        if t + 1 < epochs:
            Nm[t * dim:(t + 1) * dim, (t + 1) * dim:(t + 2) * dim] = U
            Nm[(t + 1) * dim:(t + 2) * dim, t * dim:(t + 1) * dim] = U.T
    rhs = rng.standard_normal((epochs * dim, columns))
    x = np.vstack([np.load(tmp_path / 'x_{0}.npy'.format(r)) for r in range(world)])
    assert relerr(x, np.linalg.solve(Nm, rhs)) < 1e-10
    Z = np.linalg.inv(Nm)
    zd = np.concatenate([np.load(tmp_path / 'zd_{0}.npy'.format(r)) for r in range(world)])
    zu = np.concatenate([np.load(tmp_path / 'zu_{0}.npy'.format(r)) for r in range(world)])
    scale = np.abs(Z).max()
    for t in range(epochs):
        assert np.abs(zd[t] - Z[t * dim:(t + 1) * dim, t * dim:(t + 1) * dim]).max() < 1e-10 * scale
        if t + 1 < epochs:
            assert np.abs(zu[t] - Z[t * dim:(t + 1) * dim, (t + 1) * dim:(t + 2) * dim]).max() < 1e-10 * scale


def test_bench_rank_function_over_rccl():
    """bench.py as ONE rank under torch.distributed.run with backend nccl: process group with device_id, barriers and
    all_reduces on device tensors (reduce_device = cuda), the all_gather of the covariance leg's sigma bands and the
    smoother's separator gathers, at reduced sizes; the line certifies itself like the N = 1 line."""
    root = os.path.dirname(HERE)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '1', '--backend', 'nccl', '--steps', '2', '--warmup', '1',
           '--ramp', '0', '--epochs', '8', '--cpu-sample', '1', '--cov-parallels', '3', '--cov-repeats', '1', '--cov-extensions', '0',
           '--smoother-epochs', '24', '--smoother-cpu-epochs', '3', '--smoother-repeats', '1', '--legs', 'synthesis,covariance,smoother']
    done = subprocess.run(cmd, env=_launcher_env(), capture_output=True, text=True, timeout=900)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-4000:]
    line = json.loads([ln for ln in done.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['all_checks_ok'] and line['check']['ok']
    assert line['config']['process_group'] == 'nccl world 1'
    assert line['covariance']['check']['ok'] and line['smoother']['check']['ok']
