"""
RCCL smoke test of the sharded paths (advisor r01): two fresh rank processes, one GPU each, started through
`python -m torch.distributed.run` BEFORE anything in them has touched a GPU, backend "nccl".  Skipped on boxes with fewer than
two GPUs (the per-round GPU box has one; the gloo rehearsals of tests/test_gpu_distributed.py cover the logic there).
"""
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


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs')
def test_sharded_paths_over_rccl(tmp_path):
    world = 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(HERE, 'rccl_worker.py'), str(tmp_path)]
    done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-4000:]

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
