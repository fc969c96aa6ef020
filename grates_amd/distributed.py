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
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  Without that
    environment the process is a single rank without a process group (every sharded entry point then skips its collectives);
    under a launcher a process group is created even for a world of one rank, and the collectives run (over RCCL: communicator
    set-up, device-buffer all_gather -- what tests/test_gpu_rccl.py exercises on a one-GPU box)."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world == 1 and 'RANK' not in os.environ:
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
    if not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    if len(band_sizes) != world:
        raise ValueError('band_sizes must have one entry per rank')
    longest = max(band_sizes)
    padded = torch.zeros(longest, dtype=local.dtype, device=local.device)
    padded[0:local.numel()] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[0:n] for p, n in zip(parts, band_sizes)])


def covariance_propagation_sharded(grid, covariance_matrix, min_degree, max_degree, kernel='potential',
                                   GM=3.9860044150e+14, R=6.3781363000e+06, gather=True, group=None, symmetric=False, method='direct'):
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
    local = plan.covariance_propagation(covariance_matrix, min_degree, lat0, lat1, symmetric=symmetric, method=method)
    if not gather:
        return local
    sizes = [(b1 - b0) * grid.meridians.size for b0, b1 in bands]
    return all_gather_bands(local, sizes, group)


def synthesize_sharded(time_series_batch, grid, kernel='ewh', GM=3.9860044150e+14, R=6.3781363000e+06, group=None):
    """
    Epoch-sharded batched synthesis: `time_series_batch` [T, N+1, N+1] is the full batch (host array) or a callable
    start, stop -> batch slice; returns (start, stop, device tensor of this rank's grids).  No collective.
    """
    from . import gravityfield
    start, stop, local = _epoch_shard(time_series_batch, group)
    return start, stop, gravityfield.synthesize(local, grid, kernel, GM, R)


def _epoch_shard(batch, group):
    """(start, stop, this rank's contiguous slice of the leading axis of `batch`) -- host arrays are sliced before they are uploaded"""
    import torch.distributed as dist
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if not hasattr(batch, 'shape') or len(batch.shape) < 1:
        raise ValueError('an array or tensor with the epochs along its first axis is expected')
    start, stop = shard_range(int(batch.shape[0]), rank, world)
    local = np.ascontiguousarray(batch[start:stop]) if isinstance(batch, np.ndarray) else batch[start:stop]
    return start, stop, local


def analysis_sharded(value_batch, grid, min_degree, max_degree, kernel='potential', GM=3.9860044150e+14, R=6.3781363000e+06, group=None):
    """
    Epoch-sharded batched analysis (RegularGrid.to_potential_coefficients, grates/grid.py:752-790, for a series of grids):
    `value_batch` [T, nlat, nlon] is the full batch (host array or device tensor); this rank analyses its contiguous range of
    epochs on its GPU.  Returns (start, stop, device tensor [stop - start, N+1, N+1]).  No collective: the per-order
    operators depend on the grid only and are built redundantly by every rank.
    """
    start, stop, local = _epoch_shard(value_batch, group)
    nlat, nlon = grid.parallels.size, grid.meridians.size
    if tuple(local.shape[1:]) != (nlat, nlon):
        raise ValueError('value_batch must have shape [T, {0}, {1}], got {2}'.format(nlat, nlon, tuple(value_batch.shape)))
    plan = grid._plan(kernel, max_degree, GM, R)
    return start, stop, plan.analysis(local, grid.area.reshape(nlat, nlon), min_degree)


def filter_sharded(spatial_filter, anm_batch, group=None):
    """
    Epoch-sharded batched filtering: `spatial_filter` is any grates_amd.filter object with `filter_batch` (Gaussian, OrderWiseFilter /
    DDK, GeneralMatrix / VDK; the filter itself -- blocks or the dense matrix -- is replicated on every rank), `anm_batch`
    [T, N+1, N+1] the full series.  Returns (start, stop, device tensor of this rank's filtered epochs).  No collective.
    """
    start, stop, local = _epoch_shard(anm_batch, group)
    return start, stop, spatial_filter.filter_batch(local)


# ---------------------------------------------------------------------------------------------------------------------
# Block-tridiagonal normal equations partitioned over epochs (BASELINE config 5: smoother sharded over the GPUs of a node)
# ---------------------------------------------------------------------------------------------------------------------
def _gather_blocks(tensors, group=None):
    """all_gather of a list of equally shaped device tensors: returns per-rank lists.  RCCL gathers device buffers; under gloo
    (tests, rehearsals on one GPU) the payload is staged through the host."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    flat = torch.cat([t.reshape(-1) for t in tensors]) if tensors else torch.zeros(0, dtype=torch.float64)
    on_host = dist.get_backend(group) == 'gloo'
    send = flat.cpu() if on_host else flat
    parts = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(parts, send, group=group)
    out = []
    for part in parts:
        part = part.to(flat.device) if on_host else part
        blocks, pos = [], 0
        for t in tensors:
            blocks.append(part[pos:pos + t.numel()].reshape(t.shape))
            pos += t.numel()
        out.append(blocks)
    return out


def gather_blocks(tensors, group=None):
    """all_gather of a list of equally shaped device tensors (the same shapes on every rank): per-rank lists, on every rank."""
    return _gather_blocks(tensors, group)


class _Chain:
    """A symmetric positive definite block-tridiagonal matrix (diag[t] = N[t, t], upper[t] = N[t, t+1], device tensors) behind the
    three operations the partitioned smoother needs.  The blocks are copied unless `consume` is set: then the chain works in
    the caller's tensors (which end up holding factor / covariance blocks), as a chain of BASELINE config 5's size must."""

    def __init__(self, diag, upper, consume=False):
        from .lstsq import BlockMatrix
        self.n = len(diag)
        self.matrix = _chain_matrix(BlockMatrix, diag, upper, not consume)

    def factor(self):
        self.matrix.cholesky()

    def solve(self, b):
        """N^-1 b for b [n, k]"""
        return self.matrix.solve_triangular(self.matrix.solve_triangular(b, transpose=True))

    def sparse_inverse(self):
        """(Zdiag, Zupper): the block-tridiagonal part of N^-1"""
        self.matrix.sparse_inverse()
        block = self.matrix.device_block
        return [block(t, t) for t in range(self.n)], [block(t, t + 1) for t in range(self.n - 1)]


def _chain_matrix(BlockMatrix, blocks_d, blocks_u, copy=True):
    index = np.concatenate(([0], np.cumsum([int(b.shape[0]) for b in blocks_d])))
    bm = BlockMatrix(index, index)
    bm._inverse_in_place = True        # solves and sparse inverse only: U_ii^-1 replaces U_ii, a third less memory and no allocations
    for i, b in enumerate(blocks_d):
        bm._set_device(i, i, b.clone() if copy else b)
    for i, b in enumerate(blocks_u):
        bm._set_device(i, i + 1, b.clone() if copy else b)
    return bm


def _transposed(block, in_place):
    """block^T as a contiguous tensor; a square block can be transposed in its own storage"""
    if in_place and block.shape[0] == block.shape[1] and block.is_contiguous():
        from . import engine
        return engine.transpose_in_place(block)
    return block.t().contiguous()


class _TwistedChain:
    """
    The same chain eliminated from both ends at once ("twisted" factorisation: elimination order 0, 1, ..., m-1 and n-1, n-2, ...,
    m+1, then the middle epoch m).  A block-tridiagonal system has no fill-in in this order either, the two half chains share
    nothing but the Schur complement of block m, and each is a plain chain in its own order -- so the top half [0 .. m] and the
    bottom half [n-1 .. m] (reversed, coupling blocks transposed) are two BlockMatrix objects that are factored, swept and
    inverted by two host threads on two HIP streams (shg_block_potrf_rows leaves the common last block as Schur complement;
    the two complements are added and the last row is finished in both).  The epoch-by-epoch factorisation of a d = 1681 block
    keeps a handful of CUs busy (its critical path is a chain of one-workgroup leaf factorisations), so two chains at once take
    about the time of one.  The factor differs from the one of the natural order; solutions and covariance blocks do not
    (up to rounding).
    """

    def __init__(self, diag, upper, consume=False):
        import torch
        from .lstsq import BlockMatrix
        self.torch = torch
        self.n = n = len(diag)
        self.m = m = n // 2
        self.sizes = [int(b.shape[0]) for b in diag]
        self.bounds = np.concatenate(([0], np.cumsum(self.sizes)))
        self.middle = diag[m].clone()                                          # both halves update a copy of their own
        self.top = _chain_matrix(BlockMatrix, diag[:m + 1], upper[:m], not consume)
        # position p of the bottom chain = epoch n - 1 - p; its coupling (p, p + 1) = N[n-1-p, n-2-p] = upper[n-2-p]^T
        self.bottom = _chain_matrix(BlockMatrix, [diag[t] for t in range(n - 1, m, -1)] + [self.middle.clone()],
                                    [_transposed(upper[t - 1], consume) for t in range(n - 1, m, -1)], not consume)
        self.nt, self.nbot = m + 1, n - m
        self.device = diag[0].device
        self.streams = _side_streams(self.device)

    def _both(self, top_job, bottom_job):
        """run the two jobs on two threads / streams; both streams are drained when this returns"""
        from concurrent.futures import ThreadPoolExecutor
        torch = self.torch
        main = torch.cuda.current_stream(self.device)

        def run(job, stream):
            torch.cuda.set_device(self.device)
            stream.wait_stream(main)
            with torch.cuda.stream(stream):
                out = job()
            stream.synchronize()
            return out
        with ThreadPoolExecutor(max_workers=2) as pool:
            futures = [pool.submit(run, job, stream) for job, stream in zip((top_job, bottom_job), self.streams)]
            return [f.result() for f in futures]

    def factor(self):
        from . import engine
        nt, nbot = self.nt, self.nbot
        # the two halves row by row in ONE string of launches, each launch a batch of two (shg_block_potrf_rows_pair); the longer
        # half finishes its extra row alone.  (Blocks of different sizes: two threads on two streams as in the sweeps below.)
        paired = min(nt, nbot) - 1
        if paired >= 1 and len(set(self.sizes)) == 1:
            self.top._cholesky_rows_pair(self.bottom, 0, paired)
            if nt - 1 > paired:
                self.top._cholesky_rows(paired, nt - 1)
            if nbot - 1 > paired:
                self.bottom._cholesky_rows(paired, nbot - 1)
        else:
            self._both(lambda: self.top._cholesky_rows(0, nt - 1), lambda: self.bottom._cholesky_rows(0, nbot - 1))
        # Schur complement of the middle block: N_mm minus the contributions of both halves
        st, sb = self.top.device_block(nt - 1, nt - 1), self.bottom.device_block(nbot - 1, nbot - 1)
        engine.axpby(1.0, sb, 1.0, st)
        engine.axpby(-1.0, self.middle, 1.0, st)
        sb.copy_(st)
        self._both(lambda: self.top._cholesky_rows(nt - 1, nt), lambda: self.bottom._cholesky_rows(nbot - 1, nbot))

    def _rows(self, t):
        return slice(int(self.bounds[t]), int(self.bounds[t + 1]))

    def solve(self, b):
        torch = self.torch
        n, m, k = self.n, self.m, b.shape[1]
        dm = self.sizes[m]
        b_top = b[:int(self.bounds[m + 1])].clone()
        b_bot = torch.cat([b[self._rows(t)] for t in range(n - 1, m, -1)] + [torch.zeros((dm, k), dtype=b.dtype, device=b.device)], dim=0)
        y_top, y_bot = self._both(lambda: self.top.solve_triangular(b_top, transpose=True), lambda: self.bottom.solve_triangular(b_bot, transpose=True))
        # W^T y = b: the middle rows collect the contributions of both halves (the bottom half started from zero there)
        y_top[-dm:] += y_bot[-dm:]
        y_bot[-dm:] = y_top[-dm:]
        x_top, x_bot = self._both(lambda: self.top.solve_triangular(y_top), lambda: self.bottom.solve_triangular(y_bot))
        pos = np.concatenate(([0], np.cumsum([self.sizes[t] for t in range(n - 1, m - 1, -1)])))     # row offsets of the bottom chain
        return torch.cat([x_top] + [x_bot[int(pos[n - 1 - t]):int(pos[n - t])] for t in range(m + 1, n)], dim=0)

    def sparse_inverse(self):
        n, m = self.n, self.m
        top, bot = self.top.device_block, self.bottom.device_block

        def bottom_job():
            # (N^-1)[t, t+1] for t >= m is the transpose of the bottom chain's block (n-2-t, n-1-t): transposed in its own storage
            self.bottom.sparse_inverse()
            return [_transposed(bot(n - 2 - t, n - 1 - t), True) for t in range(m, n - 1)]
        _, lower = self._both(self.top.sparse_inverse, bottom_job)
        zdiag = [top(t, t) for t in range(m + 1)] + [bot(n - 1 - t, n - 1 - t) for t in range(m + 1, n)]
        return zdiag, [top(t, t + 1) for t in range(m)] + lower


_side_stream_cache = {}


def _side_streams(device):
    import torch
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _side_stream_cache:
        _side_stream_cache[key] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
    return _side_stream_cache[key]


def _make_chain(diag, upper, consume=False):
    """two-ended elimination for chains long enough to gain from it (GRATES_AMD_TWISTED=0: natural order throughout)"""
    if len(diag) >= 8 and os.environ.get('GRATES_AMD_TWISTED', '1') != '0':
        return _TwistedChain(diag, upper, consume)
    return _Chain(diag, upper, consume)


_stream_pool = {}


def _streams(device, count):
    """`count` side streams of the device (kept between calls)"""
    import torch
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    pool = _stream_pool.setdefault(key, [])
    while len(pool) < count:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:count]


def _segment_bounds(n_loc, count, first_is_end, last_is_end):
    """Local epoch ranges [lo, hi) of `count` segments of a rank with n_loc epochs.  An epoch of a segment between two separators
    takes about 1.4 times as long as one of the two end segments of the chain (measured at d = 1681: 6.9 against 4.8 ms per
    epoch and stream -- three blocks per row in the sweeps and the Takahashi recursion, but the same latency-bound
    factorisation of the diagonal block), so the end segments get 1.5 times the epochs; every segment keeps at least two
    epochs (interior + separator; the very last one needs only one)."""
    weights = [1.5 if (i == 0 and first_is_end) or (i == count - 1 and last_is_end) else 1.0 for i in range(count)]
    total = sum(weights)
    sizes = [max(2, int(round(n_loc * w / total))) for w in weights]
    # the rounding error goes to the largest segment
    sizes[int(np.argmax(sizes))] += n_loc - sum(sizes)
    if min(sizes) < (1 if last_is_end else 2) or min(sizes[:-1] + [2]) < 2:
        raise ValueError('{0} epochs are too few for {1} segments'.format(n_loc, count))
    bounds = np.concatenate(([0], np.cumsum(sizes)))
    return [(int(bounds[i]), int(bounds[i + 1])) for i in range(count)]


def default_segments(n_loc_min, world):
    """Segments per rank.  A chain is latency bound (the block Cholesky factorisation of one epoch is a chain of ~100 small
    kernels), so a rank works on two segments at once, each on its own stream.  More than two per process did not pay on one
    MI355X (config 5, 3650 x 1681: 7.3 s with 2 segments, 9.3 / 10.1 / 9.3 s with 4 / 6 / 8): the segments between separators
    are factored twice, and the HIP runtime serialises the kernel launches of the threads of one process, so the host becomes
    the pace-maker (~650 000 launches per call)."""
    want = int(os.environ.get('GRATES_AMD_SEGMENTS', '0'))
    if want <= 0:
        want = 2
    return max(1, min(want, n_loc_min // 4))


class _SegmentedChain:
    """
    Nested dissection of a symmetric positive definite block-tridiagonal matrix (the normal equations of a fixed-interval smoother,
    grates/lstsq.py:364-392) whose block rows (epochs) are distributed over the ranks in contiguous ranges and, inside a rank,
    cut into `segments` pieces that are worked on at the same time (one host thread and one HIP stream each).  The last epoch
    of every segment but the very last one is a separator.

      1. Every segment eliminates its interior, all at once (shg_block_potrf_rows up to the separator rows, which collect the Schur
         complement): the first segment of the chain from the top, the last one from the bottom, a segment between two separators
         a and c from the top with a and c as its last two block rows -- the coupling to a is carried along as one fill-in block per
         epoch (W[t, a]), the only extra memory of the scheme.  The forward sweep of the right-hand side goes the same way.
      2. ONE all_gather (RCCL) of five d x d / d x k pieces per segment; every rank factors the separator system (block tridiagonal,
         segments - 1 rows) redundantly: solution and entries of the inverse at the separators.
      3. Back substitution and Takahashi recursion of every segment run over its interior rows only, starting from the values of
         its separator rows (shg_block_solve_rows, shg_block_sparse_inverse_rows).
    Per epoch an end segment costs what the single chain costs (20/3 d^3 for equal block sizes), a segment between separators
    about three times as many flops, most of them in the Takahashi recursion over three blocks per row -- fat products; the
    latency-bound factorisation of the diagonal block happens once per epoch everywhere, so the time per epoch differs by 1.4
    only (`_segment_bounds`).
    One more all_gather of a single d x d block per rank hands the covariance block that couples a rank's last epoch to the next
    rank's first epoch to its owner.  Separator blocks must all have the same size.
    """

    def __init__(self, diag, upper, rhs, group, consume=False, segments=None):
        import torch
        import torch.distributed as dist
        from . import engine
        from .lstsq import BlockMatrix
        self.torch, self.engine, self.group, self.BlockMatrix = torch, engine, group, BlockMatrix
        self.rank = rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.n_loc = n_loc = len(diag)
        self.device = diag[0].device
        self.sizes = sizes = [int(b.shape[0]) for b in diag]
        self.bounds = np.concatenate(([0], np.cumsum(sizes)))
        self.k = k = 0 if rhs is None else int(rhs.shape[1])
        self.kk = max(k, 1)
        if not consume:
            diag, upper = [b.clone() for b in diag], [b.clone() if b is not None else None for b in upper]
        self.diag, self.upper = list(diag), list(upper)
        self.rhs = None if rhs is None else rhs.clone()
        # the same number of segments on every rank (the gathered lists must have the same length): the shortest rank decides, ragged
        # block sizes on ANY rank mean separators at the rank boundaries only, and a caller-given count must be the same everywhere
        self.collective = collective = dist.is_initialized()
        want = int(segments) if segments is not None else -max(int(os.environ.get('GRATES_AMD_SEGMENTS', '0') or 0), 0)    # < 0: the default rule with that wish
        n_min, ragged, seg_lo, seg_hi = n_loc, len(set(sizes)) > 1, want, want
        if collective:
            t = torch.tensor([n_loc, -int(ragged), seg_lo, -seg_hi], dtype=torch.int64, device=self.device if dist.get_backend(group) != 'gloo' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            n_min, ragged, seg_lo, seg_hi = int(t[0].item()), bool(-int(t[1].item())), int(t[2].item()), -int(t[3].item())
        if seg_lo != seg_hi:
            raise ValueError('the ranks were given different segment counts (or GRATES_AMD_SEGMENTS differs between them): {0} .. {1}'.format(seg_lo, seg_hi))
        self.count = count = default_segments(n_min, world) if segments is None else int(segments)
        if ragged and count > 1:
            self.count = count = 1                                   # ragged blocks: separators only at the rank boundaries
        self.K = world * count
        if self.K < 2:
            raise ValueError('a single segment is a plain chain')
        last_rank = rank == world - 1
        self.segs = [dict(lo=lo, hi=hi, g=rank * count + i) for i, (lo, hi) in enumerate(_segment_bounds(n_loc, count, rank == 0, last_rank))]
        for sg in self.segs:
            sg['kind'] = 'first' if sg['g'] == 0 else ('last' if sg['g'] == self.K - 1 else 'middle')
            sg['ni'] = sg['hi'] - sg['lo'] - (0 if sg['kind'] == 'last' else 1)             # interior epochs
            if sg['ni'] < 1:
                raise ValueError('every segment needs an interior epoch')
        # separator size (equal everywhere) and the coupling of the previous rank's last epoch to this rank's first one
        self.d = d = sizes[self.segs[0]['hi'] - 1] if not (last_rank and count == 1) else None
        boundary = upper[n_loc - 1] if not last_rank else None
        if collective:
            shape = (sizes[-1], sizes[-1])
            gathered = _gather_blocks([boundary.contiguous() if boundary is not None else torch.zeros(shape, dtype=torch.float64, device=self.device)], group)
            self.left_of_rank = gathered[rank - 1][0] if rank > 0 else None
            if self.d is None:
                self.d = d = int(self.left_of_rank.shape[0])
        else:
            self.left_of_rank = None
        for sg in self.segs:
            if sg['kind'] != 'last' and sizes[sg['hi'] - 1] != d:
                raise ValueError('all separator blocks must have the same size')
        self.streams = _streams(self.device, len(self.segs))
        # Rehearsals of several ranks on ONE card (tests, bench.py --same-device): with GRATES_AMD_REHEARSAL_TURNS=1 the ranks take
        # turns with their segment work instead of sharing the card, and `busy_s` adds up what a rank spends in its own turns --
        # the time the rank would need on a GPU of its own (the elapsed time of such a run is the sum over the ranks).
        self.turns = world > 1 and os.environ.get('GRATES_AMD_REHEARSAL_TURNS', '0') == '1'
        self.busy_s = 0.0
        # The fill-in blocks W[t, a] of the segments between separators, one allocation per segment made HERE, before the segment
        # threads start: fresh device memory is mapped synchronously (21 us per MB), and allocating 22 MB blocks from inside the
        # threads while the other segments' kernels were running made two such segments 2.5 times slower each (6.8 against
        # 3.5 ms per epoch at d = 1681).
        # (in pieces of at most 1 GB rather than one tensor of 16 GB -- 729 blocks at d = 1681 -- so that the pieces fit the holes
        # the caching allocator has)
        for sg in self.segs:
            if sg['kind'] == 'middle':
                sg['fill'] = fill = [None]
                piece, used = None, 0
                for q in range(1, sg['ni']):
                    dq = sizes[sg['lo'] + q]
                    if piece is None or used + dq > piece.shape[0]:
                        rows = max(dq, min(int((1 << 27) // max(d, 1)), int(self.bounds[sg['hi'] - 1] - self.bounds[sg['lo'] + q])))
                        piece, used = torch.zeros((rows, d), dtype=torch.float64, device=self.device), 0
                    fill.append(piece[used:used + dq])
                    used += dq
        self._factor()

    # ---- helpers
    def zeros(self, r, c):
        return self.torch.zeros((r, c), dtype=self.torch.float64, device=self.device)

    def _rows(self, lo, hi):
        return slice(int(self.bounds[lo]), int(self.bounds[hi]))

    def _left(self, sg):
        """N[a, lo]: coupling of the separator on the left of the segment to its first epoch"""
        return self.upper[sg['lo'] - 1] if sg['lo'] > 0 else self.left_of_rank

    def _each(self, job):
        """job(segment) for all local segments at once, one thread and one stream each; the streams are drained on return"""
        from concurrent.futures import ThreadPoolExecutor
        torch = self.torch
        main = torch.cuda.current_stream(self.device)

        def run(sg, stream):
            torch.cuda.set_device(self.device)
            self.engine.block_set_lookahead(False)         # several chains at once: one queue each (csrc/blas.hip)
            stream.wait_stream(main)
            with torch.cuda.stream(stream):
                out = job(sg)
            stream.synchronize()
            return out
        def all_segments():
            if len(self.segs) == 1:
                return [job(self.segs[0])]
            with ThreadPoolExecutor(max_workers=len(self.segs)) as pool:
                futures = [pool.submit(run, sg, st) for sg, st in zip(self.segs, self.streams)]
                return [f.result() for f in futures]
        return self._in_turn(all_segments)

    def _in_turn(self, work):
        """work() -- at once on every rank, or rank after rank in a rehearsal with turns; its time goes to busy_s"""
        import time
        import torch.distributed as dist
        torch = self.torch
        out = None
        for turn in range(self.world if self.turns else 1):
            if self.turns:
                torch.cuda.synchronize(self.device)
                dist.barrier(self.group)
            if not self.turns or turn == self.rank:
                t0 = time.perf_counter()
                out = work()
                torch.cuda.synchronize(self.device)
                self.busy_s += time.perf_counter() - t0
        if self.turns:
            dist.barrier(self.group)
        return out

    def _chain(self, blocks_d, blocks_u):
        return _chain_matrix(self.BlockMatrix, blocks_d, blocks_u, False)

    # ---- step 1: every segment eliminates its interior; its separator rows collect the Schur complement
    def _build(self, sg):
        """the segment's block matrix: its interior epochs, then its separator rows"""
        lo, hi, ni, d = sg['lo'], sg['hi'], sg['ni'], self.d
        Z = self.zeros
        if sg['kind'] == 'first':
            sg['M'] = self._chain(self.diag[lo:hi], self.upper[lo:hi - 1])
            return
        left = self._left(sg)
        if sg['kind'] == 'last':
            # reversed order: position p = epoch hi - 1 - p, then the separator a on the left (zero block: it collects -S_aa)
            dd = [self.diag[t] for t in range(hi - 1, lo - 1, -1)] + [Z(d, d)]
            uu = [_transposed(self.upper[t - 1], True) for t in range(hi - 1, lo, -1)] + [left.t().contiguous()]
            sg['M'] = self._chain(dd, uu)
            return
        # between two separators a and c: block rows [interior ..., a, c]; a and c start from zero and collect -S_aa, -S_ac, -S_cc,
        # the coupling to a is carried from row to row as fill-in (allocated by the symbolic step of the block Cholesky)
        offset = self.bounds[lo:hi] - self.bounds[lo]
        index = np.concatenate((offset, [offset[-1] + d, offset[-1] + 2 * d]))
        M = sg['M'] = self.BlockMatrix(index, index)
        M._inverse_in_place = True
        for q in range(ni):
            M._set_device(q, q, self.diag[lo + q])
            if q + 1 < ni:
                M._set_device(q, q + 1, self.upper[lo + q])
        M._set_device(0, ni, left.t().contiguous())                       # N[lo, a]
        for q in range(1, ni):                                             # fill-in W[t, a] (allocated before the threads started)
            M._set_device(q, ni, sg['fill'][q])
        M._set_device(ni - 1, ni + 1, self.upper[hi - 2])                  # N[last interior epoch, c]
        for i, j in ((ni, ni), (ni, ni + 1), (ni + 1, ni + 1)):
            M._set_device(i, j, Z(d, d))

    def _eliminate(self):
        """block Cholesky of the interior rows of every segment.  Two chains from the ends of the whole system (segments 'first' and
        'last' of one rank: the single-GPU case) or two segments between separators with equally many epochs have the same
        structure and go through the device together, every launch a batch of two (shg_block_potrf_rows_pair); the others run
        side by side, one thread and one stream each."""
        segs = self.segs
        chains = [sg for sg in segs if sg['kind'] != 'middle']
        between = [sg for sg in segs if sg['kind'] == 'middle']
        pairs, single = [], []
        if len(chains) == 2 and len(set(self.sizes)) == 1:
            pairs.append((chains[0], chains[1], min(chains[0]['ni'], chains[1]['ni'])))
        else:
            single += chains
        while between:
            sg = between.pop(0)
            mate = next((o for o in between if o['ni'] == sg['ni']), None) if len(set(self.sizes)) == 1 else None
            if mate is None:
                single.append(sg)
            else:
                between.remove(mate)
                pairs.append((sg, mate, sg['ni']))
        if not pairs:
            self._each(lambda sg: sg['M']._cholesky_rows(0, sg['ni']))
            return

        def work():
            self.engine.block_set_lookahead(True)
            for a, b, rows in pairs:
                a['M']._cholesky_rows_pair(b['M'], 0, rows)
                for sg in (a, b):
                    if sg['ni'] > rows:
                        sg['M']._cholesky_rows(rows, sg['ni'])
            for sg in single:
                sg['M']._cholesky_rows(0, sg['ni'])
        self._in_turn(work)

    def _reduce(self, sg):
        torch, engine = self.torch, self.engine
        lo, hi, ni, d, kk = sg['lo'], sg['hi'], sg['ni'], self.d, self.kk
        Z = self.zeros
        M = sg['M']
        if sg['kind'] == 'first':
            y = sg['y'] = self.rhs[self._rows(lo, hi)].clone() if self.k else Z(int(self.bounds[hi] - self.bounds[lo]), 1)
            M._solve_rows(y, True, 0, ni)
            return [M.device_block(ni, ni).clone(), Z(d, d), Z(d, d), y[-d:].clone(), Z(d, kk)]
        if sg['kind'] == 'last':
            parts = [self.rhs[self._rows(t, t + 1)] for t in range(hi - 1, lo - 1, -1)] if self.k else [Z(self.sizes[t], 1) for t in range(hi - 1, lo - 1, -1)]
            y = sg['y'] = torch.cat(parts + [Z(d, kk)], dim=0)
            M._solve_rows(y, True, 0, ni)
            return [Z(d, d), M.device_block(ni, ni).clone(), Z(d, d), Z(d, kk), y[-d:].clone()]
        offset = self.bounds[lo:hi] - self.bounds[lo]
        y = sg['y'] = torch.cat(((self.rhs[self._rows(lo, hi - 1)] if self.k else Z(int(offset[-1]), 1)), Z(2 * d, kk)), dim=0)
        M._solve_rows(y, True, 0, ni)
        Acc = self.diag[hi - 1].clone()
        engine.axpby(1.0, M.device_block(ni + 1, ni + 1), 1.0, Acc)
        rc = self.rhs[self._rows(hi - 1, hi)].clone() if self.k else Z(d, 1)
        engine.axpby(1.0, y[-d:], 1.0, rc)
        return [Acc, M.device_block(ni, ni).clone(), M.device_block(ni, ni + 1).clone(), rc, y[-2 * d:-d].clone()]

    # ---- step 2: the separator system, redundantly on every rank
    def _factor(self):
        engine, d, kk, K = self.engine, self.d, self.kk, self.K
        self._each(self._build)
        self._eliminate()
        pieces = self._each(self._reduce)
        flat = [t.contiguous() for p in pieces for t in p]
        if self.collective:
            everyone = _gather_blocks(flat, self.group)
            flat_all = [t for part in everyone for t in part]
        else:
            flat_all = flat
        P = [flat_all[5 * g:5 * g + 5] for g in range(K)]                  # per segment: Acc, Aaa, Aac, rc, ra
        # diag_j = Acc(j) + Aaa(j + 1), coupling (j, j + 1) = Aac(j + 1), right-hand side rc(j) + ra(j + 1)
        nsep = K - 1
        index = np.arange(0, (nsep + 1) * d, d)
        self.reduced = reduced = self.BlockMatrix(index, index)
        red_rhs = self.zeros(nsep * d, kk)
        for j in range(nsep):
            block = P[j][0].clone()
            engine.axpby(1.0, P[j + 1][1], 1.0, block)
            reduced._set_device(j, j, block)
            if j + 1 < nsep:
                reduced._set_device(j, j + 1, P[j + 1][2].clone())
            r = red_rhs[j * d:(j + 1) * d]
            engine.axpby(1.0, P[j][3], 0.0, r)
            engine.axpby(1.0, P[j + 1][4], 1.0, r)
        reduced.cholesky()
        self.x_sep = reduced.solve_triangular(reduced.solve_triangular(red_rhs, transpose=True))
        reduced.sparse_inverse()                                             # Z_SS: blocks (j, j) and (j, j + 1)

    # ---- step 3: solution
    def solve(self):
        d = self.d

        def job(sg):
            M, y, ni = sg['M'], sg['y'], sg['ni']
            j = sg['g'] - 1 if sg['kind'] == 'last' else sg['g']            # the separator in the segment's last row
            y[-d:] = self.x_sep[j * d:(j + 1) * d]
            if sg['kind'] == 'middle':
                y[-2 * d:-d] = self.x_sep[(j - 1) * d:j * d]               # ... and the one on its left in the row before
            M._solve_rows(y, False, 0, ni)
            if sg['kind'] == 'middle':
                return self.torch.cat((y[:-2 * d], y[-d:]), dim=0)          # interior epochs, then the segment's own separator
            if sg['kind'] == 'first':
                return y
            # reversed segment: back into epoch order, without the separator row
            parts, pos = [], 0
            for t in range(sg['hi'] - 1, sg['lo'] - 1, -1):
                parts.append(y[pos:pos + self.sizes[t]])
                pos += self.sizes[t]
            return self.torch.cat(parts[::-1], dim=0)
        return self.torch.cat(self._each(job), dim=0)

    # ---- step 3: covariance blocks
    def covariance(self):
        d = self.d
        sep = self.reduced.device_block

        def job(sg):
            M, ni, lo, hi = sg['M'], sg['ni'], sg['lo'], sg['hi']
            block = M.device_block
            j = sg['g'] - 1 if sg['kind'] == 'last' else sg['g']
            if sg['kind'] == 'middle':                                      # rows [interior ..., a, c]
                block(ni, ni).copy_(sep(j - 1, j - 1))
                block(ni, ni + 1).copy_(sep(j - 1, j))
                block(ni + 1, ni + 1).copy_(sep(j, j))
                M._sparse_inverse_rows(0, ni)
                zd = [block(q, q) for q in range(ni)] + [block(ni + 1, ni + 1)]
                zu = [block(q, q + 1) for q in range(ni - 1)] + [block(ni - 1, ni + 1)]
                return zd, zu, block(0, ni).t().contiguous()                # Z[a, lo]
            block(ni, ni).copy_(sep(j, j))
            M._sparse_inverse_rows(0, ni)
            if sg['kind'] == 'last':
                n = hi - lo
                zd = [block(n - 1 - q, n - 1 - q) for q in range(n)]
                zu = [_transposed(block(n - 2 - q, n - 1 - q), True) for q in range(n - 1)]
                return zd, zu, block(ni - 1, ni).t().contiguous()           # Z[a, lo]
            return [block(q, q) for q in range(hi - lo)], [block(q, q + 1) for q in range(hi - lo - 1)], None
        results = self._each(job)
        zdiag, zupper = [], []
        for i, (zd, zu, to_left) in enumerate(results):
            if i > 0:
                zupper.append(to_left)                                       # coupling of the previous segment's separator to this segment
            zdiag += zd
            zupper += zu
        first_left = results[0][2]
        if self.collective:
            send = first_left if first_left is not None else self.zeros(d, self.sizes[0])
            from_right = _gather_blocks([send.contiguous()], self.group)
            if self.rank < self.world - 1:
                zupper.append(from_right[self.rank + 1][0])
        return zdiag, zupper


def _segmented(diag, upper, rhs, group, consume, segments=None):
    """the segmented chain for this call, or None when the whole chain is one rank's and too short to cut"""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and (default_segments(len(diag), 1) if segments is None else segments) < 2:
        return None
    return _SegmentedChain(diag, upper, rhs, group, consume, segments)


def solve_block_tridiagonal_partitioned(diag, upper, rhs, group=None, consume=False):
    """
    Solve the symmetric positive definite block-tridiagonal system N x = b whose block rows (epochs) are distributed over the
    ranks in contiguous ranges: the fixed-interval smoother of grates.lstsq (NormalEquations.solve on a VAR(1)-constrained
    system; wider bands are the same after grouping `order` epochs into one block row).

    Every rank passes the blocks of ITS epochs t0 .. t1-1 as device tensors:
        diag[k]   N[t0+k, t0+k]      (upper triangle significant, like BlockMatrix)
        upper[k]  N[t0+k, t0+k+1]    (upper[-1] couples to the next rank's first epoch; ignored on the last rank)
        rhs       [sum of block sizes, columns]
    and receives x for its epochs.  Every rank but the last needs at least two epochs.

    Scheme (nested dissection with the last epoch of every rank but the last as separator, _PartitionedChain): each rank factors
    its interior chain with the block Cholesky of grates_amd.lstsq.BlockMatrix (sequential in epochs, all ranks concurrently)
    and solves it for the right-hand side and for its two coupling blocks; ONE all_gather (RCCL) collects the separator blocks
    and the Schur complement pieces, every rank solves the small separator system redundantly and back-substitutes.
    Interior chains of eight epochs and more are eliminated from both ends at once (_TwistedChain).

    consume=True lets the factorisation work in the caller's blocks instead of copies of them (they are overwritten): a chain
    of BASELINE config 5's size (3650 epochs x 2 x 22.6 MB plus the inverses of the diagonal factor blocks) fits the card only once.
    """
    sc = _segmented(diag, upper, rhs, group, consume)
    if sc is None:
        chain = _make_chain(diag, upper[:len(diag) - 1], consume)
        chain.factor()
        return chain.solve(rhs)
    return sc.solve()


def sparse_inverse_block_tridiagonal_partitioned(diag, upper, group=None, consume=False):
    """
    Block-tridiagonal part of N^-1 (the covariance blocks NormalEquations.compute_covariance(sparse=True) leaves in the matrix,
    grates/lstsq.py:823-846, 1026-1042) for a chain whose epochs are distributed over the ranks like in
    solve_block_tridiagonal_partitioned (same arguments without the right-hand side).

    Returns (Zdiag, Zupper): Zdiag[k] = (N^-1)[t0+k, t0+k] and Zupper[k] = (N^-1)[t0+k, t0+k+1] for the epochs of this rank
    (the last rank returns one coupling block less).

    With I the interior epochs of a rank, S the separators and Y = A_II^-1 A_IS (two coupling blocks per rank):
        Z_SS = (A_SS - A_SI A_II^-1 A_IS)^-1       band of the inverse of the separator system (Takahashi, redundantly)
        Z_IS = -Y Z_SS                             only the blocks next to the separators are kept
        Z_II = A_II^-1 + Y Z_SS Y^T                band of A_II^-1 by the Takahashi recursion of the interior chain, plus a
                                                   rank-2d correction per block (two GEMMs per block)
    The only collectives are the two all_gathers of the factorisation and one more of a single d x d block per rank (the
    covariance block that couples a separator to the first epoch of the next rank is computed by that next rank).
    """
    sc = _segmented(diag, upper, None, group, consume)
    if sc is None:
        chain = _make_chain(diag, upper[:len(diag) - 1], consume)
        chain.factor()
        return chain.sparse_inverse()
    return sc.covariance()


def smooth_block_tridiagonal_partitioned(diag, upper, rhs, group=None, consume=False, timings=None):
    """Solution and covariance blocks from ONE factorisation: (x, Zdiag, Zupper) as returned by solve_block_tridiagonal_partitioned
    and sparse_inverse_block_tridiagonal_partitioned (NormalEquations.solve followed by compute_covariance(sparse=True),
    grates/lstsq.py:950-968, 1026-1042).

    timings : dict, optional
        receives the wall-clock seconds of the three phases ('factor_s': interior chains, all_gather and separator system;
        'solve_s': sweeps and back substitution; 'covariance_s': sparse inverse) -- the device is drained after each phase when
        a dict is given, and not otherwise."""
    import time

    def lap(key, since):
        """drain the device and book the phase (only when the caller asked for timings)"""
        if timings is None:
            return since
        pc_torch.cuda.synchronize()
        now = time.perf_counter()
        timings[key] = now - since
        return now
    import torch as pc_torch
    if timings is not None:
        pc_torch.cuda.synchronize()
    t0 = time.perf_counter()
    sc = _segmented(diag, upper, rhs, group, consume)
    if sc is None:
        chain = _make_chain(diag, upper[:len(diag) - 1], consume)
        chain.factor()
        t0 = lap('factor_s', t0)
        x = chain.solve(rhs)
        t0 = lap('solve_s', t0)
        zdiag, zupper = chain.sparse_inverse()
        lap('covariance_s', t0)
        return x, zdiag, zupper
    t0 = lap('factor_s', t0)
    x = sc.solve()
    t0 = lap('solve_s', t0)
    zdiag, zupper = sc.covariance()
    lap('covariance_s', t0)
    if timings is not None:
        timings['segment_work_s'] = sc.busy_s
    return x, zdiag, zupper
