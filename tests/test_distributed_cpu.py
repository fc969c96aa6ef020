"""
World-size-2 (and 3) gloo tests of the multi-GPU sharding logic on the CPU: epoch ranges, latitude bands and the
all_gather of per-band sigma vectors.  The GPU kernels themselves are exercised by the -m gpu tests; here the
distributed plumbing is checked with stand-in band vectors so that no GPU is needed.
"""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from grates_amd import distributed as gd


def test_shard_range_partitions():
    for total in (0, 1, 7, 240, 3650):
        for world in (1, 2, 3, 4, 8):
            ranges = [gd.shard_range(total, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            for (a0, a1), (b0, b1) in zip(ranges[:-1], ranges[1:]):
                assert a1 == b0
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1
    assert gd.latitude_bands(360, 4) == [(0, 90), (90, 180), (180, 270), (270, 360)]
    with pytest.raises(ValueError):
        gd.shard_range(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nlat, nlon, result_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    r, w = gd.init('gloo')
    assert (r, w) == (rank, world)
    bands = gd.latitude_bands(nlat, world)
    lat0, lat1 = bands[rank]
    # stand-in for plan.covariance_propagation(cov, nmin, lat0, lat1): sigma of grid point (i, j) = i * nlon + j + 0.5
    local = torch.arange(lat0 * nlon, lat1 * nlon, dtype=torch.float64) + 0.5
    full = gd.all_gather_bands(local, [(b1 - b0) * nlon for b0, b1 in bands])
    # epoch sharding: every epoch is processed exactly once
    start, stop = gd.shard_range(17, rank, world)
    counts = torch.zeros(17, dtype=torch.int64)
    counts[start:stop] = 1
    dist.all_reduce(counts)
    # the epoch shard the sharded synthesis / analysis / filter entry points take from a full batch
    whole = np.arange(17 * 3, dtype=float).reshape(17, 3)
    s0, s1, mine_np = gd._epoch_shard(whole, None)
    t0, t1, mine_t = gd._epoch_shard(torch.from_numpy(whole), None)
    assert (s0, s1) == (t0, t1) == (start, stop) and mine_np.flags['C_CONTIGUOUS']
    np.testing.assert_array_equal(mine_np, whole[start:stop])
    np.testing.assert_array_equal(mine_t.numpy(), whole[start:stop])
    # block gather of the partitioned smoother: every rank contributes a list of blocks, everybody receives all lists
    mine = [torch.full((2, 3), float(rank), dtype=torch.float64), torch.arange(4, dtype=torch.float64).reshape(4, 1) + 10 * rank]
    everyone = gd._gather_blocks(mine)
    assert len(everyone) == world
    for r, blocks in enumerate(everyone):
        assert torch.equal(blocks[0], torch.full((2, 3), float(r), dtype=torch.float64))
        assert torch.equal(blocks[1], torch.arange(4, dtype=torch.float64).reshape(4, 1) + 10 * r)
    np.save(os.path.join(result_dir, 'full_{0}.npy'.format(rank)), full.numpy())
    np.save(os.path.join(result_dir, 'counts_{0}.npy'.format(rank)), counts.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_band_gather_and_epoch_sharding_gloo(world, tmp_path):
    nlat, nlon = 7, 5          # bands of unequal size: the gather pads to the longest band
    mp.spawn(_worker, args=(world, _free_port(), nlat, nlon, str(tmp_path)), nprocs=world, join=True)
    expect = np.arange(nlat * nlon, dtype=float) + 0.5
    for rank in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / 'full_{0}.npy'.format(rank)), expect)
        np.testing.assert_array_equal(np.load(tmp_path / 'counts_{0}.npy'.format(rank)), np.ones(17, dtype=np.int64))


def test_single_process_paths_need_no_process_group():
    local = torch.arange(6, dtype=torch.float64)
    assert gd.all_gather_bands(local, [6]) is local
    os.environ.pop('WORLD_SIZE', None)
    os.environ.pop('RANK', None)
    assert gd.init() == (0, 1)


def test_segment_bounds_and_defaults(monkeypatch):
    """Segments of a rank: contiguous, complete, at least two epochs each (one for the very last segment of the chain), the end
    segments of the chain about 1.5 times as long as the ones between separators."""
    for n_loc, count, first, last in ((456, 2, False, False), (456, 2, True, False), (3650, 2, True, True), (3650, 6, True, True), (9, 4, True, True),
                                      (5, 2, False, True), (3, 1, False, True), (1, 1, False, True)):
        segs = gd._segment_bounds(n_loc, count, first, last)
        assert segs[0][0] == 0 and segs[-1][1] == n_loc and all(a[1] == b[0] for a, b in zip(segs, segs[1:]))
        sizes = [hi - lo for lo, hi in segs]
        assert min(sizes[:-1] + [2]) >= 2 and sizes[-1] >= (1 if last else 2)
    sizes = [hi - lo for lo, hi in gd._segment_bounds(3650, 6, True, True)]
    assert abs(sizes[0] - sizes[-1]) <= 2 and abs(sizes[0] - 1.5 * sizes[2]) <= 3 and len(set(sizes[1:-1])) <= 2
    with pytest.raises(ValueError):
        gd._segment_bounds(3, 2, False, False)
    monkeypatch.delenv('GRATES_AMD_SEGMENTS', raising=False)
    assert gd.default_segments(3650, 1) == 2 and gd.default_segments(456, 8) == 2 and gd.default_segments(7, 1) == 1 and gd.default_segments(3, 2) == 1
    monkeypatch.setenv('GRATES_AMD_SEGMENTS', '6')
    assert gd.default_segments(3650, 1) == 6 and gd.default_segments(20, 1) == 5
