"""
Rank program of tests/test_gpu_rccl.py: started by `python -m torch.distributed.run` as a fresh process per GPU (nothing has
touched the GPU before), backend "nccl" = RCCL.  Runs the sharded covariance propagation, the epoch-sharded synthesis and the
epoch-partitioned smoother (solve + sparse inverse) and writes what it got to the directory given on the command line.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, 'golden'))


def main(result_dir):
    import torch
    import inputs
    import grates_amd as ga
    from grates_amd import distributed as gd
    rank, world = gd.init('nccl')
    dist = torch.distributed
    assert dist.is_initialized() and dist.get_backend() == 'nccl' and dist.get_world_size() == world
    counts = {'backend': dist.get_backend(), 'world': world, 'all_gather': 0, 'all_reduce': 0, 'on_device': 0, 'on_host': 0}

    def counted(name, real):
        def call(*a, **k):
            counts[name] += 1
            tensors = [t for x in a for t in (x if isinstance(x, (list, tuple)) else [x]) if torch.is_tensor(t)]
            counts['on_device' if all(t.is_cuda for t in tensors) else 'on_host'] += 1
            return real(*a, **k)
        return call
    dist.all_gather = counted('all_gather', dist.all_gather)
    dist.all_reduce = counted('all_reduce', dist.all_reduce)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()          # noqa: E731

    N, nmin = 24, 2
    grid = ga.grid.GeographicGrid(5.0, 4.0)                                    # 45 parallels: uneven bands
    cov = inputs.spd_covariance(77, (N + 1) ** 2 - nmin ** 2)
    for method in ('direct', 'separable'):
        full = gd.covariance_propagation_sharded(grid, cov, nmin, N, kernel='ewh', method=method)
        if rank == 0:
            np.save(os.path.join(result_dir, 'sigma_{0}.npy'.format(method)), full.cpu().numpy())
    batch = np.stack([inputs.coefficients(300 + e, 20) for e in range(7)])
    e0, e1, grids = gd.synthesize_sharded(batch, grid, kernel='ewh')
    np.save(os.path.join(result_dir, 'grids_{0}.npy'.format(rank)), np.concatenate(([e0, e1], grids.cpu().numpy().ravel())))

    # block-tridiagonal chain, the seeded system of tests/test_gpu_distributed.py
    epochs, dim, columns = 8, 130, 3
    rng = np.random.default_rng(3)
    diag, upper = [], []
    for t in range(epochs):
        G = rng.standard_normal((dim, dim + 4))
        diag.append(G @ G.T / dim + 3.0 * np.eye(dim))
        upper.append(rng.standard_normal((dim, dim)) * (0.4 / np.sqrt(dim)))
    rhs = rng.standard_normal((epochs * dim, columns))
    t0, t1 = gd.shard_range(epochs, rank, world)
    x = gd.solve_block_tridiagonal_partitioned([dev(b) for b in diag[t0:t1]], [dev(b) for b in upper[t0:t1]], dev(rhs[t0 * dim:t1 * dim]))
    Zd, Zu = gd.sparse_inverse_block_tridiagonal_partitioned([dev(b) for b in diag[t0:t1]], [dev(b) for b in upper[t0:t1]])
    np.save(os.path.join(result_dir, 'x_{0}.npy'.format(rank)), x.cpu().numpy())
    np.save(os.path.join(result_dir, 'zd_{0}.npy'.format(rank)), np.stack([b.cpu().numpy() for b in Zd]))
    np.save(os.path.join(result_dir, 'zu_{0}.npy'.format(rank)), np.stack([b.cpu().numpy() for b in Zu]))
    import json
    with open(os.path.join(result_dir, 'collectives_{0}.json'.format(rank)), 'w') as f:
        json.dump(counts, f)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1])
