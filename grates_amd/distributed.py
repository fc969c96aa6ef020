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
        block.copy_(block.t().clone())
        return block
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


class _PartitionedChain:
    """
    Nested dissection of a symmetric positive definite block-tridiagonal matrix whose block rows (epochs) are distributed over
    the ranks in contiguous ranges, with the last epoch of every rank but the last as separator: the factored interior chain of
    this rank, Y = A_II^-1 [C_left, C_right, b] for its two coupling blocks and the right-hand side, and the factored
    separator (Schur complement) system, which every rank holds redundantly after ONE all_gather of 5 d x d blocks (and a few
    vectors) per rank.  A second, small all_gather beforehand hands every rank the coupling block to its left separator.
    """

    def __init__(self, diag, upper, rhs, group, consume=False):
        import torch
        import torch.distributed as dist
        from . import engine
        from .lstsq import BlockMatrix
        self.torch, self.engine, self.group = torch, engine, group
        self.rank = rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.n_loc = n_loc = len(diag)
        self.last = last = rank == world - 1
        if n_loc < (1 if last else 2):
            raise ValueError('every rank but the last needs at least two epochs (got {0})'.format(n_loc))
        self.sizes = sizes = [int(b.shape[0]) for b in diag]
        self.bounds = bounds = np.concatenate(([0], np.cumsum(sizes)))
        device = diag[0].device
        self.k = k = 0 if rhs is None else rhs.shape[1]

        def zeros(r, c):
            return torch.zeros((r, c), dtype=torch.float64, device=device)
        self.zeros = zeros

        self.ni = ni = n_loc if (last or world == 1) else n_loc - 1         # interior epochs
        self.n_int = n_int = int(bounds[ni])
        if world == 1:
            self.interior = _make_chain(diag, upper[:n_loc - 1], consume)
            self.interior.factor()
            return

        # coupling to the left separator: N[s_(g-1), t0] lives on the previous rank
        d_sep = sizes[-1]
        boundary = upper[n_loc - 1] if not last else zeros(d_sep, d_sep)
        if not last and boundary.shape[1] != boundary.shape[0]:
            raise ValueError('partitioned solve expects equal block sizes at the rank boundaries')
        boundaries = _gather_blocks([boundary.contiguous()], group)           # collective: every rank takes part
        self.left = left = boundaries[rank - 1][0] if rank > 0 else None      # [d_sep_left, d_first]
        self.right = upper[ni - 1] if not last else None                      # N[t1-2, t1-1]

        self.interior = interior = _make_chain(diag[:ni], upper[:ni - 1], consume)
        interior.factor()
        columns = [rhs[:n_int]] if k else []
        if left is not None:                                                # C_left = left^T in the first interior block rows
            cl = zeros(n_int, left.shape[0])
            cl[:sizes[0]] = left.t()
            columns.append(cl)
        if not last:                                                        # C_right = N[t1-2, t1-1] in the last interior block rows
            cr = zeros(n_int, d_sep)
            cr[int(bounds[ni - 1]):n_int] = upper[ni - 1]
            columns.append(cr)
        W = torch.cat(columns, dim=1).contiguous()
        del columns
        Z = interior.solve(W)
        del W
        self.zb = Z[:, :k]
        pos = k
        self.ZL = self.ZR = None
        if left is not None:
            self.ZL = Z[:, pos:pos + left.shape[0]]
            pos += left.shape[0]
        if not last:
            self.ZR = Z[:, pos:pos + d_sep]
        self.Y = Z[:, k:]                                                   # [Y_left, Y_right]
        zb, ZL, ZR = self.zb, self.ZL, self.ZR

        first = slice(0, sizes[0])
        tail = slice(int(bounds[ni - 1]), n_int)
        # Schur complement pieces of this rank's interior chain (d = separator block size; equal sizes at the boundaries)
        self.d = d = d_sep if not last else left.shape[0]
        kk = max(k, 1)
        has_b = k > 0
        S_ll = engine.gemm(left, ZL[first].contiguous()) if left is not None else zeros(d, d)
        b_l = engine.gemm(left, zb[first].contiguous()) if (left is not None and has_b) else zeros(d, kk)
        S_lr = engine.gemm(left, ZR[first].contiguous()) if (left is not None and not last) else zeros(d, d)
        S_rr = engine.gemm(upper[ni - 1], ZR[tail].contiguous(), transa=True) if not last else zeros(d, d)
        b_r = engine.gemm(upper[ni - 1], zb[tail].contiguous(), transa=True) if (not last and has_b) else zeros(d, kk)
        sep_d = diag[n_loc - 1] if not last else zeros(d, d)
        sep_b = rhs[n_int:] if (not last and has_b) else zeros(d, kk)
        gathered = _gather_blocks([sep_d.contiguous(), S_ll, S_lr, S_rr, sep_b.contiguous(), b_l, b_r], group)

        # separator system (world - 1 block rows), held redundantly on every rank
        self.nsep = nsep = world - 1
        index = np.arange(0, (nsep + 1) * d, d)
        self.reduced = reduced = BlockMatrix(index, index)
        self.red_rhs = red_rhs = zeros(nsep * d, kk)
        for i in range(nsep):
            own, nxt = gathered[i], gathered[i + 1]
            block = own[0].clone()
            engine.axpby(-1.0, own[3], 1.0, block)                           # - S_rr of the chain on its left
            engine.axpby(-1.0, nxt[1], 1.0, block)                           # - S_ll of the chain on its right
            reduced._set_device(i, i, block)
            if i + 1 < nsep:
                coupling = zeros(d, d)
                engine.axpby(-1.0, nxt[2], 0.0, coupling)                    # - S_lr of the chain between separators i and i + 1
                reduced._set_device(i, i + 1, coupling)
            r = red_rhs[i * d:(i + 1) * d]
            engine.axpby(1.0, own[4], 0.0, r)
            engine.axpby(-1.0, own[6], 1.0, r)
            engine.axpby(-1.0, nxt[5], 1.0, r)
        reduced.cholesky()


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
    return _partitioned_solution(_PartitionedChain(diag, upper, rhs, group, consume), rhs)


def _partitioned_solution(pc, rhs):
    if pc.world == 1:
        return pc.interior.solve(rhs)
    engine, d, rank = pc.engine, pc.d, pc.rank
    x_sep = pc.reduced.solve_triangular(pc.reduced.solve_triangular(pc.red_rhs, transpose=True))
    # back substitution of the interior chain
    x_int = pc.zb.clone()
    if pc.left is not None:
        engine.gemm(pc.ZL.contiguous(), x_sep[(rank - 1) * d:rank * d], alpha=-1.0, beta=1.0, out=x_int)
    if not pc.last:
        engine.gemm(pc.ZR.contiguous(), x_sep[rank * d:(rank + 1) * d], alpha=-1.0, beta=1.0, out=x_int)
        return pc.torch.cat((x_int, x_sep[rank * d:(rank + 1) * d]), dim=0)
    return x_int


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
    return _partitioned_covariance(_PartitionedChain(diag, upper, None, group, consume))


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
    pc = _PartitionedChain(diag, upper, rhs, group, consume)
    t0 = lap('factor_s', t0)
    x = _partitioned_solution(pc, rhs)
    t0 = lap('solve_s', t0)
    zdiag, zupper = _partitioned_covariance(pc)
    lap('covariance_s', t0)
    return x, zdiag, zupper


def _partitioned_covariance(pc):
    torch, engine = pc.torch, pc.engine
    ni, bounds = pc.ni, pc.bounds
    if pc.world == 1:
        return pc.interior.sparse_inverse()
    d, rank, last = pc.d, pc.rank, pc.last
    pc.reduced.sparse_inverse()
    sep = pc.reduced.device_block
    # Z_SS restricted to the (at most two) separators next to this rank's chain
    parts = []
    if pc.left is not None:
        row = [sep(rank - 1, rank - 1)] + ([sep(rank - 1, rank)] if not last else [])
        parts.append(torch.cat(row, dim=1))
    if not last:
        row = ([sep(rank - 1, rank).t()] if pc.left is not None else []) + [sep(rank, rank)]
        parts.append(torch.cat(row, dim=1))
    M = torch.cat(parts, dim=0).contiguous()
    dl = d if pc.left is not None else 0
    Y = pc.Y.contiguous()
    P = engine.gemm(Y, M)                                                   # [n_int, dl + dr] = Y Z_SS
    Zdiag, Zupper = pc.interior.sparse_inverse()
    rows = lambda t: slice(int(bounds[t]), int(bounds[t + 1]))          # noqa: E731
    for t in range(ni):
        engine.gemm(P[rows(t)], Y[rows(t)], transb=True, beta=1.0, out=Zdiag[t])
        if t + 1 < ni:
            engine.gemm(P[rows(t)], Y[rows(t + 1)], transb=True, beta=1.0, out=Zupper[t])
    # blocks next to the separators: Z[t, r] = -(Y Z_SS)[t, r],  Z[l, t] = -(Y Z_SS)[t, l]^T
    to_left = (-P[rows(0), :dl]).t().contiguous() if pc.left is not None else pc.zeros(d, pc.sizes[0])
    from_right = _gather_blocks([to_left], pc.group)
    if not last:
        Zupper.append((-P[rows(ni - 1), dl:]).contiguous())
        Zdiag.append(sep(rank, rank))
        Zupper.append(from_right[rank + 1][0])
    return Zdiag, Zupper
