"""
Multi-GPU sharding of the hot path: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm)
or gloo on the CPU (tests).

The path shards without any data-path collective:
  * synthesis / analysis / filters: epochs are independent -> contiguous epoch ranges per rank, plan tables
    (a few MB) are replicated, the outputs stay sharded;
  * covariance propagation: parallels are independent given the covariance matrix -> contiguous latitude bands
    per rank, Sigma is replicated (loaded or broadcast once, never per call).  The only collective is one
    all_gather of the per-band sigma vectors (M * 8 bytes in total, 2 MB for a 0.5 degree grid) when every rank
    needs the full grid.
"""

import os

import numpy as np


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world == 1:
        return 0, 1
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        local_rank = int(os.environ.get('LOCAL_RANK', str(rank)))
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def shard_range(total, rank, world):
    """Contiguous balanced range [start, stop) of `total` independent units owned by `rank` (sizes differ by <= 1)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError('invalid rank / world size ({0}, {1})'.format(rank, world))
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def latitude_bands(parallel_count, world):
    """[(lat0, lat1)] for every rank: contiguous bands of parallels balanced by count."""
    return [shard_range(parallel_count, r, world) for r in range(world)]


def all_gather_bands(local, band_sizes, group=None):
    """
    Concatenate the per-rank vectors `local` (lengths `band_sizes`, known to every rank) on every rank.
    Uses one all_gather of equally sized (padded) buffers -- works on RCCL and gloo alike.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    if len(band_sizes) != world:
        raise ValueError('band_sizes must have one entry per rank')
    longest = max(band_sizes)
    padded = torch.zeros(longest, dtype=local.dtype, device=local.device)
    padded[0:local.numel()] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[0:n] for p, n in zip(parts, band_sizes)])


def covariance_propagation_sharded(grid, covariance_matrix, min_degree, max_degree, kernel='potential',
                                   GM=3.9860044150e+14, R=6.3781363000e+06, gather=True, group=None):
    """
    Latitude-band sharded RegularGrid.covariance_propagation: this rank propagates its band of parallels on its
    GPU; with gather=True every rank receives the full sigma vector (one RCCL all_gather), otherwise the local band.
    `covariance_matrix` must already be resident on this rank (replicated).
    """
    import torch.distributed as dist
    from . import engine
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    bands = latitude_bands(grid.parallels.size, world)
    lat0, lat1 = bands[rank]
    plan = grid._plan(kernel, max_degree, GM, R)
    local = plan.covariance_propagation(covariance_matrix, min_degree, lat0, lat1)
    if not gather:
        return local
    sizes = [(b1 - b0) * grid.meridians.size for b0, b1 in bands]
    return all_gather_bands(local, sizes, group)


def synthesize_sharded(time_series_batch, grid, kernel='ewh', GM=3.9860044150e+14, R=6.3781363000e+06, group=None):
    """
    Epoch-sharded batched synthesis: `time_series_batch` [T, N+1, N+1] is the full batch (host array) or a callable
    start, stop -> batch slice; returns (start, stop, device tensor of this rank's grids).  No collective.
    """
    import torch.distributed as dist
    from . import gravityfield
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    total = time_series_batch.shape[0] if hasattr(time_series_batch, 'shape') else None
    if total is None:
        raise ValueError('time_series_batch must be an array [T, N+1, N+1]')
    start, stop = shard_range(total, rank, world)
    local = np.ascontiguousarray(time_series_batch[start:stop]) if isinstance(time_series_batch, np.ndarray) else time_series_batch[start:stop]
    return start, stop, gravityfield.synthesize(local, grid, kernel, GM, R)
