"""
Device engine: thin Python wrappers around libshg.  Arrays live on the GPU as torch tensors (fp64,
contiguous); every routine hands raw device pointers and the current HIP stream to the C ABI.

Nothing in here computes on the CPU: without the library or without a GPU the calls raise.
"""

import ctypes
import hashlib

import numpy as np

from . import _lib


def _torch():
    import torch
    return torch


def require_gpu():
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError('grates_amd: no GPU visible -- the hot path runs on MI355X only (there is no CPU fallback)')
    _lib.load()
    return torch


def device(index=None):
    torch = require_gpu()
    return torch.device('cuda', torch.cuda.current_device() if index is None else index)


def to_device(x, dev=None):
    """fp64 contiguous device tensor from ndarray / tensor / scalar sequence."""
    torch = require_gpu()
    dev = device() if dev is None else dev
    if isinstance(x, torch.Tensor):
        return x.to(device=dev, dtype=torch.float64).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float64))).to(dev)


def to_host(t):
    return t.detach().cpu().numpy()


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(_torch().cuda.current_stream().cuda_stream)


def _host_ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


class Plan:
    """
    Device tables for one (max_degree, parallels, kn table, meridians) configuration
    (shg_plan_create).  `colat` [nlat] geocentric colatitudes, `kn` [nlat, N+1], `meridians` [nlon].
    """

    def __init__(self, max_degree, colat, kn, meridians, device_index=None):
        torch = require_gpu()
        self.device = device(device_index)
        self.max_degree = int(max_degree)
        colat = np.ascontiguousarray(colat, dtype=np.float64)
        kn = np.ascontiguousarray(kn, dtype=np.float64)
        meridians = np.ascontiguousarray(meridians, dtype=np.float64)
        if kn.shape != (colat.size, self.max_degree + 1):
            raise ValueError('kn must have shape (nlat, max_degree + 1), got {0}'.format(kn.shape))
        self.nlat, self.nlon = colat.size, meridians.size
        self._handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.call('shg_plan_create', ctypes.byref(self._handle), self.max_degree, self.nlat, _host_ptr(colat),
                      _host_ptr(kn), self.nlon, _host_ptr(meridians), self.device.index)

    def __del__(self):
        handle = getattr(self, '_handle', None)
        if handle is not None and handle.value:
            try:
                _lib.load().shg_plan_destroy(handle)
            except Exception:
                pass
            self._handle = None             # (module globals may already be gone at interpreter shutdown)

    def info(self):
        arr = (ctypes.c_int64 * 8)()
        _lib.call('shg_plan_info', self._handle, arr)
        return {'max_degree': arr[0], 'nlat': arr[1], 'nlon': arr[2], 'fourfold_symmetry': bool(arr[3] & 1), 'north_south_symmetry': bool(arr[3] & 2),
                'rotation_symmetry': bool(arr[3] & 4), 'epochs_per_pass': arr[4], 'k_slots': arr[5], 'fused': bool(arr[6]), 'path': int(arr[7]) & 0xff, 'rotations': int(arr[7]) >> 8}

    def set_path(self, path):
        """'auto', 'staged' (three kernels, any grid), 'fused' (single kernel on 4-fold symmetric meridians), 'fused32' (32-row
        panels) or 'rot' (rotation-folded kernel on equi-angular meridians with nlon % 96 == 0 or nlon % 48 == 0).  The fused
        kernels use the north-south symmetry of the parallels when the grid has it (their plain variants otherwise)."""
        _lib.call('shg_plan_set_path', self._handle, {'auto': 0, 'staged': 1, 'fused': 2, 'fused32': 5, 'rot': 6}[path])

    def set_stage_limit(self, limit):
        """Rotation-folded kernel only: at most `limit` workgroups in their Legendre stage at once (< 0: sixteenths of the CUs, 0 = off,
        the default).  A tuning knob whose sign differs between boxes (include/shg.h)."""
        _lib.call('shg_plan_set_stage_limit', self._handle, int(limit))

    def set_rotations(self, R):
        """Rotation count of the rotation-folded kernel: 0 (the plan's own choice), 3, 6, 9 or 10; the meridians must be invariant
        under R rotations with nlon / R a multiple of 16 (`info()['rotations']` tells the count in use)."""
        _lib.call('shg_plan_set_rotations', self._handle, int(R))

    def set_chunk(self, epochs_per_pass):
        _lib.call('shg_plan_set_chunk', self._handle, int(epochs_per_pass))

    PROFILE_KINDS = ('pack_coefficients', 'legendre_stage', 'lon_stage', 'covprop', 'analysis_lon', 'analysis_solve', 'k6', 'k7')

    def profile(self, enable=True, kinds=None):
        """Record HIP events around every kernel this plan launches (on the launching stream), or around the kernels of the named
        `kinds` only (PROFILE_KINDS; an event pair costs the stream ~5 us)."""
        if enable and kinds:
            mask = 0
            for name in kinds:
                mask |= 1 << (self.PROFILE_KINDS.index(name) + 1)
            _lib.call('shg_plan_profile', self._handle, mask)
        else:
            _lib.call('shg_plan_profile', self._handle, 1 if enable else 0)

    def profile_read(self):
        """{kernel: (total_ms, launches)} since the last read; synchronises the recorded events."""
        ms = (ctypes.c_double * 8)()
        n = (ctypes.c_int64 * 8)()
        _lib.call('shg_plan_profile_read', self._handle, ms, n)
        return {name: (ms[k], n[k]) for k, name in enumerate(self.PROFILE_KINDS) if n[k] > 0}

    def synthesis(self, anm, out=None):
        """anm [B, N+1, N+1] (or [N+1, N+1]), or an OrderMajorSeries of at least the plan's degree -> grid [B, nlat, nlon] (device tensor)."""
        torch = _torch()
        if isinstance(anm, OrderMajorSeries):
            B = anm.epochs
            if anm.max_degree < self.max_degree:
                raise ValueError('the series holds degrees up to {0}, the plan needs {1}'.format(anm.max_degree, self.max_degree))
            if out is None:
                out = torch.empty((B, self.nlat, self.nlon), dtype=torch.float64, device=self.device)
            elif tuple(out.shape) != (B, self.nlat, self.nlon) or out.dtype != torch.float64 or not out.is_contiguous():
                raise ValueError('out must be a contiguous fp64 tensor of shape {0}'.format((B, self.nlat, self.nlon)))
            Plan._written(out)
            with torch.cuda.device(self.device):
                try:
                    _lib.call('shg_synthesis_om', self._handle, _ptr(anm.data), anm.max_degree, B, anm.padded_epochs, _ptr(out), _stream())
                except _lib.ShgError:            # a plan whose kernel reads the reference arrays only: through them
                    batch = anm.to_batch()
                    n1 = self.max_degree + 1
                    _lib.call('shg_synthesis', self._handle, _ptr(batch[:, :n1, :n1].contiguous()), B, _ptr(out), _stream())
            return out
        x = to_device(anm, self.device)
        single = x.dim() == 2
        if single:
            x = x.unsqueeze(0)
        n1 = self.max_degree + 1
        if x.dim() != 3 or x.shape[1] != n1 or x.shape[2] != n1:
            raise ValueError('coefficient batch must have shape (B, {0}, {0}), got {1}'.format(n1, tuple(x.shape)))
        B = x.shape[0]
        if out is None:
            out = torch.empty((B, self.nlat, self.nlon), dtype=torch.float64, device=self.device)
        elif tuple(out.shape) != (B, self.nlat, self.nlon) or out.dtype != torch.float64 or not out.is_contiguous():
            raise ValueError('out must be a contiguous fp64 tensor of shape {0}'.format((B, self.nlat, self.nlon)))
        Plan._written(out)
        with torch.cuda.device(self.device):
            _lib.call('shg_synthesis', self._handle, _ptr(x), B, _ptr(out), _stream())
        return out[0] if single else out

    def covariance_propagation(self, cov, min_degree, lat0=0, lat1=None, symmetric=False, method='direct'):
        """cov [P, P] degree-wise -> sigma [(lat1-lat0)*nlon] for the band of parallels [lat0, lat1).

        symmetric: False (default) multiplies with the full matrix like the reference; True reads only the upper triangle
        of a symmetric matrix (half the MFMA work); None checks the matrix on the device and takes the shortcut when it is
        exactly symmetric (the result then differs from the general path by summation order only).
        method: 'direct' forms A Sigma like the reference (2 M P^2 flops on the MFMA units); 'separable' uses the
        factorisation of the synthesis matrix into a latitude and a longitude part (about nlon times fewer flops,
        P^2 + 32 P nlat doubles of workspace), same result up to summation order."""
        torch = _torch()
        lat1 = self.nlat if lat1 is None else lat1
        c = to_device(cov, self.device)
        P = (self.max_degree + 1) ** 2 - min_degree ** 2
        if c.dim() != 2 or c.shape[0] != P or c.shape[1] != P:
            raise ValueError('covariance matrix must have shape ({0}, {0}), got {1}'.format(P, tuple(c.shape)))
        out = torch.empty(((lat1 - lat0) * self.nlon,), dtype=torch.float64, device=self.device)
        if method not in ('direct', 'separable'):
            raise ValueError("method must be 'direct' or 'separable'")
        with torch.cuda.device(self.device):
            if symmetric is None:
                defect = torch.zeros(1, dtype=torch.float64, device=self.device)
                _lib.call('shg_symmetry_defect', _ptr(c), P, P, _ptr(defect), _stream())
                symmetric = float(defect.item()) == 0.0
            if method == 'separable':
                _lib.call('shg_covprop_diag_separable_symmetric' if symmetric else 'shg_covprop_diag_separable', self._handle, _ptr(c),
                          int(min_degree), int(lat0), int(lat1), _ptr(out), _stream())
                return out
            _lib.call('shg_covprop_diag_symmetric' if symmetric else 'shg_covprop_diag', self._handle, _ptr(c), int(min_degree), int(lat0), int(lat1),
                      _ptr(out), _stream())
        return out

    _trusting = None          # weak set of the plans that hold a weight token (class attribute, created on first use)

    @classmethod
    def _written(cls, tensor):
        """A library call is about to write into `tensor` through its raw pointer (torch's version counter does not see that): plans
        that trust weights living in the same storage forget their token and validate the weights again on their next analysis."""
        if not cls._trusting:
            return
        try:
            storage = tensor.untyped_storage().data_ptr()
        except Exception:
            return
        for plan in list(cls._trusting):
            w = plan._analysis_weights
            if w is not None and w.untyped_storage().data_ptr() == storage:
                plan._analysis_token = None

    def analysis(self, grid, area, min_degree, trusted_weights=None):
        """grid [B, nlat, nlon], area [nlat, nlon] -> anm [B, N+1, N+1].

        The library validates `area` against the weights its cached operators were built for on every call (a device compare and
        one host synchronisation).  A device tensor that is the very tensor of the previous call (same storage, same torch
        version counter: not written to since) need not be compared again -- the call then passes area = NULL ("the weights of
        the previous call", include/shg.h) and nothing waits for the device.  Writes through `.data` or through raw pointers outside
        this module are invisible to the version counter: `trusted_weights=False` always validates, `True` skips the validation
        whatever the tensor's history (the caller vouches for it); this module's own raw-pointer writes (`gemm(out=)`, `axpby`,
        `Plan.synthesis(out=)`) into the storage of trusted weights withdraw the trust."""
        torch = _torch()
        g = to_device(grid, self.device)
        single = g.dim() == 2
        if single:
            g = g.unsqueeze(0)
        n1 = self.max_degree + 1
        out = torch.empty((g.shape[0], n1, n1), dtype=torch.float64, device=self.device)    # zeroed by the library
        token = None
        if torch.is_tensor(area) and area.is_cuda and area.dtype == torch.float64 and area.is_contiguous():
            token = (area.data_ptr(), area._version, tuple(area.shape), int(min_degree))
        with torch.cuda.device(self.device):
            trusted = token is not None and (token == self._analysis_token if trusted_weights is None else
                                             (bool(trusted_weights) and self._analysis_token is not None and token[2:] == self._analysis_token[2:]))
            if g.shape[0] > 0 and trusted and trusted_weights is not False:
                _lib.call('shg_analysis', self._handle, _ptr(g), None, int(min_degree), g.shape[0], _ptr(out), _stream())
            else:
                a = to_device(area, self.device).reshape(self.nlat, self.nlon)
                self._analysis_token = None
                _lib.call('shg_analysis', self._handle, _ptr(g), _ptr(a), int(min_degree), g.shape[0], _ptr(out), _stream())
                if g.shape[0] > 0:       # (the tensor is kept alive: its address cannot be handed to another one meanwhile)
                    self._analysis_token, self._analysis_weights = token, (area if token is not None else None)
                    if token is not None:
                        import weakref
                        if Plan._trusting is None:
                            Plan._trusting = weakref.WeakSet()
                        Plan._trusting.add(self)
        return out[0] if single else out

    _analysis_token = None
    _analysis_weights = None

    def analysis_info(self):
        """{'parity_split': True | False | None (no operators yet), 'parity_defect': float} of the cached analysis operators (include/shg.h)"""
        info = (ctypes.c_double * 2)()
        _lib.call('shg_analysis_info', self._handle, ctypes.addressof(info))
        return {'parity_split': None if info[0] < 0 else bool(info[0]), 'parity_defect': float(info[1])}

    def analysis_matrix(self, area, min_degree):
        """Dense analysis operator F [P, nlat * nlon] (device tensor) for the area weights `area`: F @ values = analysis(values)."""
        torch = _torch()
        self._analysis_token = None                    # the call may rebuild the cached operators for other weights
        a = to_device(area, self.device).reshape(self.nlat, self.nlon)
        P = (self.max_degree + 1) ** 2 - min_degree ** 2
        out = torch.empty((P, self.nlat * self.nlon), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.call('shg_analysis_matrix', self._handle, _ptr(a), int(min_degree), _ptr(out), _stream())
        return out


def release_scratch():
    """Give the per-stream scratch buffers the library keeps between calls (shg_analysis, split-K block products) back to
    the driver; waits for the device."""
    require_gpu()
    _lib.call('shg_scratch_release')


_plan_cache = {}
_PLAN_CACHE_LIMIT = 8


def cached_plan(max_degree, colat, kn, meridians):
    """Plans are cached by content so that repeated to_grid calls reuse the device tables."""
    torch = require_gpu()
    h = hashlib.blake2b(digest_size=16)
    for a in (np.int64(max_degree), np.int64(torch.cuda.current_device()), colat, kn, meridians):
        h.update(np.ascontiguousarray(a).tobytes())
    key = h.hexdigest()
    plan = _plan_cache.pop(key, None)
    if plan is None:
        if len(_plan_cache) >= _PLAN_CACHE_LIMIT:
            _plan_cache.pop(next(iter(_plan_cache)))           # least recently used: a hit moves its plan to the end
        plan = Plan(max_degree, colat, kn, meridians)
    _plan_cache[key] = plan
    return plan


_grid_plan_cache = {}


def plan_by_grid(scalars, axes, build):
    """The plan of a regular grid found by what defines it -- kernel name, degree, constants (`scalars`, hashable) and the two axes
    (`axes`: parallels, meridians; hashed: 17 KB on the 0.25 degree grid) -- without building the kernel table first; `build()` makes
    the plan on a miss.  Least recently used of _PLAN_CACHE_LIMIT entries goes first."""
    torch = require_gpu()
    h = hashlib.blake2b(digest_size=16)
    h.update(repr((scalars, torch.cuda.current_device())).encode())
    for a in axes:
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    key = h.hexdigest()
    plan = _grid_plan_cache.pop(key, None)
    if plan is None:
        if len(_grid_plan_cache) >= _PLAN_CACHE_LIMIT:
            _grid_plan_cache.pop(next(iter(_grid_plan_cache)))
        plan = build()
    _grid_plan_cache[key] = plan
    return plan


_table_cache = {}


def cached_point_tables(max_degree, latitude, longitude, tag, build):
    """Device copies of the per-point tables (colatitude, longitude, degree factors) of a point list, cached by content like
    the plans: `build()` -> (colat, lon, kn) host arrays is only called for a new (points, kernel / constants tag, degree)."""
    torch = require_gpu()
    h = hashlib.blake2b(digest_size=16)
    h.update(repr((int(max_degree), int(torch.cuda.current_device()), tag)).encode())
    for a in (latitude, longitude):
        h.update(np.ascontiguousarray(a).tobytes())
    key = h.hexdigest()
    entry = _table_cache.get(key)
    if entry is None:
        if len(_table_cache) >= _PLAN_CACHE_LIMIT:
            _table_cache.pop(next(iter(_table_cache)))
        entry = tuple(to_device(a) for a in build())
        _table_cache[key] = entry
    return entry


def clear_plan_cache():
    _plan_cache.clear()
    _table_cache.clear()


# ---------------------------------------------------------------------------------------------------
# table functions / index maps
# ---------------------------------------------------------------------------------------------------

def legendre_functions(max_degree, colat):
    torch = require_gpu()
    th = to_device(np.atleast_1d(colat))
    out = torch.empty((th.numel(), max_degree + 1, max_degree + 1), dtype=torch.float64, device=th.device)
    _lib.call('shg_legendre', int(max_degree), _ptr(th), th.numel(), _ptr(out), _stream())
    return out


def legendre_functions_per_order(max_degree, order, colat):
    torch = require_gpu()
    th = to_device(np.atleast_1d(colat))
    out = torch.empty((th.numel(), max_degree + 1 - order), dtype=torch.float64, device=th.device)
    _lib.call('shg_legendre_order', int(max_degree), int(order), _ptr(th), th.numel(), _ptr(out), _stream())
    return out


def trigonometric_functions(max_degree, lon):
    torch = require_gpu()
    lam = to_device(np.atleast_1d(lon))
    out = torch.empty((lam.numel(), max_degree + 1, max_degree + 1), dtype=torch.float64, device=lam.device)
    _lib.call('shg_trigonometric', int(max_degree), _ptr(lam), lam.numel(), _ptr(out), _stream())
    return out


def synthesis_matrix_order(max_degree, order, min_degree, colat, lon, kn, pointwise):
    """Operator block of one order (device tensors): (cosine block, sine block or None for order 0), rows parallel-major for a
    regular grid (pointwise=False: colat / kn per parallel, lon = meridians) or one row per point (pointwise=True)."""
    torch = require_gpu()
    th, lam, k = to_device(np.atleast_1d(colat)), to_device(np.atleast_1d(lon)), to_device(kn)
    nlat, nlon = th.numel(), lam.numel()
    if k.shape != (nlat, max_degree + 1) or (pointwise and nlon != nlat):
        raise ValueError('synthesis_matrix_order: colat [k], kn [k, max_degree + 1] and, for point lists, lon [k] expected')
    count = max_degree + 1 - max(order, min_degree)
    rows = nlat if pointwise else nlat * nlon
    out_c = torch.empty((rows, max(count, 0)), dtype=torch.float64, device=th.device)
    out_s = torch.empty_like(out_c) if order > 0 else None
    _lib.call('shg_synthesis_matrix_order', int(max_degree), int(order), int(min_degree), _ptr(th), nlat, _ptr(lam), nlon, _ptr(k),
              1 if pointwise else 0, _ptr(out_c), _ptr(out_s) if out_s is not None else None, _stream())
    return out_c, out_s


def synthesis_matrix(max_degree, min_degree, colat, lon, kn):
    """Dense synthesis operator A [npts, P] (device tensor) of the points (colat, lon) with degree factors kn [npts, N+1]."""
    torch = require_gpu()
    th, lam, k = to_device(np.atleast_1d(colat)), to_device(np.atleast_1d(lon)), to_device(kn)
    npts = th.numel()
    if k.shape != (npts, max_degree + 1) or lam.numel() != npts:
        raise ValueError('synthesis_matrix: colat, lon [npts] and kn [npts, max_degree + 1] expected')
    P = (max_degree + 1) ** 2 - min_degree ** 2
    out = torch.empty((npts, P), dtype=torch.float64, device=th.device)
    _lib.call('shg_synthesis_matrix', int(max_degree), int(min_degree), _ptr(th), _ptr(lam), _ptr(k), npts, _ptr(out), _stream())
    return out


# ---------------------------------------------------------------------------------------------------
# sparse block Cholesky (csrc/blockchol.hip): block tables are [nb, nb] uint64 arrays of device addresses (0 = no block)
# ---------------------------------------------------------------------------------------------------

def _table(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(ctypes.c_void_p)


class BlockTable:
    """Stored upper blocks of a block matrix in compressed row form for the shg_block_* calls (include/shg.h):
    `blocks` maps (i, j), j >= i, to a device tensor [rows_i, cols_j]."""

    def __init__(self, bounds, blocks):
        self.bounds = np.ascontiguousarray(bounds, dtype=np.int32)
        nb = self.bounds.size - 1
        keys = sorted(k for k in blocks if k[1] >= k[0])
        for k in keys:                   # csrc/blockchol.hip reads every block as dense row-major with ld = columns
            if not blocks[k].is_contiguous():
                raise ValueError('block {0} is not a contiguous row-major tensor (strides {1})'.format(k, tuple(blocks[k].stride())))
        self.rowptr = np.zeros(nb + 1, dtype=np.int32)
        for i, _ in keys:
            self.rowptr[i + 1] += 1
        np.cumsum(self.rowptr, out=self.rowptr)
        self.colidx = np.array([j for _, j in keys], dtype=np.int32)
        self.address = np.array([blocks[k].data_ptr() for k in keys], dtype=np.uint64)
        self.nb = nb

    def args(self):
        as_ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        return self.nb, as_ptr(self.bounds), as_ptr(self.rowptr), as_ptr(self.colidx), as_ptr(self.address)


def _require_row_major(B, name):
    if B.dim() != 2 or (B.shape[1] > 1 and B.stride(1) != 1):
        raise ValueError('{0}: a two-dimensional tensor with a contiguous last dimension is expected (shape {1}, strides {2})'.format(
            name, tuple(B.shape), tuple(B.stride())))


def block_potrf(table, inverses, first=0, last=None):
    """In-place block Cholesky of the block rows first <= r < last (default: all; later rows are left as the Schur complement);
    returns 0 or the 1-based index of the first non-positive pivot."""
    torch = require_gpu()
    inverses, pi = _table(inverses)
    info = torch.zeros(1, dtype=torch.int32, device=device())
    _lib.call('shg_block_potrf_rows', *table.args(), pi, int(first), int(table.nb if last is None else last), _ptr(info), _stream())
    return int(info.item())


def block_potrf_pair(table0, inverses0, table1, inverses1, first=0, last=None):
    """block_potrf for two matrices of the same structure (bounds, stored blocks) in one pass, every launch serving both
    (shg_block_potrf_rows_pair); returns the two pivot flags."""
    torch = require_gpu()
    for a, b in ((table0.bounds - table0.bounds[0], table1.bounds - table1.bounds[0]), (table0.rowptr, table1.rowptr), (table0.colidx, table1.colidx)):
        if a.shape != b.shape or not np.array_equal(a, b):
            raise ValueError('block_potrf_pair: the two matrices differ in structure')
    inverses0, p0 = _table(inverses0)
    inverses1, p1 = _table(inverses1)
    info = torch.zeros(2, dtype=torch.int32, device=device())
    nb, bounds, rowptr, colidx, address0 = table0.args()
    _lib.call('shg_block_potrf_rows_pair', nb, bounds, rowptr, colidx, address0, p0, table1.args()[4], p1, int(first),
              int(table0.nb if last is None else last), _ptr(info), _stream())
    flags = info.cpu()
    return int(flags[0]), int(flags[1])


LOOKAHEAD_MODES = {False: 0, True: 1, 'off': 0, 'on': 1, 'carry': 2, 'plain': 3}


def block_set_lookahead(mode):
    """The factorisation of a diagonal block overlaps its panel steps on two more streams unless the calling THREAD turns that
    off (shg_block_set_lookahead): threads that factor several matrices at once do better without.  mode: False / 'off', True /
    'on' (chain rows carry their coupling block through the sweep when a second hardware queue was found), 'carry' (always),
    'plain' (never)."""
    _lib.call('shg_block_set_lookahead', LOOKAHEAD_MODES[mode])


def block_lookahead_info():
    """{'mode', 'side_queues_apart', 'chain_rows_carry_coupling'} of the calling thread on the current stream."""
    arr = (ctypes.c_int * 4)()
    _lib.call('shg_block_lookahead_info', _stream(), arr)
    return {'mode': int(arr[0]), 'side_queues_apart': int(arr[1]), 'chain_rows_carry_coupling': bool(arr[2])}


def block_solve(table, inverses, transpose, B):
    """B [n, k] <- W^-1 B or W^-T B in place (device tensor, contiguous last dimension)."""
    _require_row_major(B, 'block_solve')
    inverses, pi = _table(inverses)
    _lib.call('shg_block_solve', *table.args(), pi, 1 if transpose else 0, _ptr(B), B.shape[1], max(B.stride(0), 1), _stream())
    return B


def block_sparse_inverse(table, inverses):
    inverses, pi = _table(inverses)
    _lib.call('shg_block_sparse_inverse', *table.args(), pi, _stream())


def block_solve_rows(table, inverses, transpose, first, last, B):
    """block_solve restricted to the block rows first <= r < last (forward sweep: later rows receive the updates; backward sweep:
    later rows hold the solution already), in place."""
    _require_row_major(B, 'block_solve_rows')
    inverses, pi = _table(inverses)
    _lib.call('shg_block_solve_rows', *table.args(), pi, 1 if transpose else 0, int(first), int(last), _ptr(B), B.shape[1], max(B.stride(0), 1), _stream())
    return B


def block_sparse_inverse_rows(table, inverses, first, last):
    """Takahashi recursion of the block rows last - 1 .. first; the blocks of the later rows hold their entries of the inverse already."""
    inverses, pi = _table(inverses)
    _lib.call('shg_block_sparse_inverse_rows', *table.args(), pi, int(first), int(last), _stream())


def block_inverse(table, inverses):
    inverses, pi = _table(inverses)
    _lib.call('shg_block_inverse', *table.args(), pi, _stream())


def block_multiply(table, mode, B):
    """mode 0: W B, 1: the reference's W^T B (assigning form), 2: N B with the upper blocks of a symmetric N."""
    torch = require_gpu()
    _require_row_major(B, 'block_multiply')
    out = torch.empty((B.shape[0], B.shape[1]), dtype=B.dtype, device=B.device)
    _lib.call('shg_block_multiply', *table.args(), int(mode), _ptr(B), B.shape[1], max(B.stride(0), 1), _ptr(out), max(out.stride(0), 1), _stream())
    return out


def scale_columns(F, w):
    """F[:, j] *= w[j] in place (window function of Grid.window_matrix) on the device."""
    torch = require_gpu()
    F.mul_(to_device(w, F.device).reshape(1, -1))
    return F


def congruence(W, S):
    """W S W^T on the fp64 MFMA GEMM (device tensors or arrays); exactly symmetric for a symmetric S."""
    torch = require_gpu()
    W = W if isinstance(W, torch.Tensor) and W.is_cuda and W.dtype == torch.float64 and W.is_contiguous() else to_device(W)
    S = S if isinstance(S, torch.Tensor) and S.is_cuda and S.dtype == torch.float64 and S.is_contiguous() else to_device(S)
    if W.dim() != 2 or S.dim() != 2 or S.shape[0] != S.shape[1] or W.shape[1] != S.shape[0]:
        raise ValueError('congruence: W [n, k] and a square S [k, k] expected, got {0} and {1}'.format(tuple(W.shape), tuple(S.shape)))
    n, k = W.shape
    out = torch.empty((n, n), dtype=torch.float64, device=W.device)
    work = torch.empty((n, k), dtype=torch.float64, device=W.device)
    _lib.call('shg_congruence', n, k, _ptr(W), max(k, 1), _ptr(S), max(k, 1), _ptr(out), max(n, 1), _ptr(work), _stream())
    return out


def ravel(arr, min_degree, max_degree):
    """arr [B, Na+1, Na+1] device tensor -> [B, P] degree-wise."""
    torch = require_gpu()
    x = to_device(arr)
    B, na = x.shape[0], x.shape[-1] - 1
    P = (max_degree + 1) ** 2 - min_degree ** 2
    out = torch.empty((B, max(P, 0)), dtype=torch.float64, device=x.device)
    _lib.call('shg_ravel', _ptr(x), B, na, int(min_degree), int(max_degree), _ptr(out), _stream())
    return out


def unravel(vec, min_degree, max_degree):
    """vec [B, P] device tensor -> [B, nmax+1, nmax+1]."""
    torch = require_gpu()
    v = to_device(vec)
    B = v.shape[0]
    out = torch.empty((B, max_degree + 1, max_degree + 1), dtype=torch.float64, device=v.device)
    _lib.call('shg_unravel', _ptr(v), B, int(min_degree), int(max_degree), _ptr(out), _stream())
    return out


def degree_scale(anm, weights, first_degree):
    """anm [B, N+1, N+1] scaled by weights[n] for every degree n >= first_degree."""
    torch = require_gpu()
    x = to_device(anm)
    N = x.shape[-1] - 1
    w = to_device(weights)
    if w.numel() != N + 1:
        raise ValueError('weights must have max_degree + 1 entries')
    out = torch.empty_like(x)
    _lib.call('shg_degree_scale', _ptr(w), N, int(first_degree), _ptr(x), x.shape[0], _ptr(out), _stream())
    return out


def synthesis_points(max_degree, colat, lon, kn, anm):
    """Point-list synthesis: colat/lon [npts], kn [npts, N+1], anm [B, N+1, N+1] -> [B, npts]."""
    torch = require_gpu()
    th, lam, k, x = to_device(colat), to_device(lon), to_device(kn), to_device(anm)
    out = torch.empty((x.shape[0], th.numel()), dtype=torch.float64, device=x.device)
    _lib.call('shg_synthesis_points', int(max_degree), _ptr(th), _ptr(lam), _ptr(k), th.numel(), _ptr(x), x.shape[0], _ptr(out), _stream())
    return out


def covprop_points(max_degree, colat, lon, kn, cov, min_degree):
    torch = require_gpu()
    th, lam, k, c = to_device(colat), to_device(lon), to_device(kn), to_device(cov)
    out = torch.empty((th.numel(),), dtype=torch.float64, device=c.device)
    _lib.call('shg_covprop_points', int(max_degree), _ptr(th), _ptr(lam), _ptr(k), th.numel(), _ptr(c), int(min_degree), _ptr(out), _stream())
    return out


def epoch_rms(values, acc=None, count=0):
    """acc [M] (+)= sum over the epochs of values [B, M] squared (device); count > 0 finishes with sqrt(acc / count)."""
    torch = require_gpu()
    v = to_device(values)
    if v.dim() != 2 or not v.is_contiguous():
        v = v.reshape(v.shape[0], -1).contiguous()
    accumulate = acc is not None
    if acc is None:
        acc = torch.empty((v.shape[1],), dtype=torch.float64, device=v.device)
    _lib.call('shg_epoch_rms', _ptr(v), v.shape[0], v.shape[1], int(accumulate), int(count), _ptr(acc), _stream())
    return acc


class OrderMajorSeries:
    """A time series of coefficient sets that stays on the device between operators (the batching of TimeSeries.to_array,
    grates/gravityfield.py:964-980, in the layout the order-wise operators work on): `data` [(N+1)^2, Bpad] with the epochs fastest
    (Bpad = epochs rounded up to 32) and the coefficients of one order and kind (the slots of the DDK block list: order 0 cosine,
    order 1 cosine, order 1 sine, ...) in consecutive rows.  `OrderWiseFilter.filter_series` and `Plan.synthesis` take and return it
    without passing through the reference arrays [B, N+1, N+1]; `from_batch` / `to_batch` convert."""

    def __init__(self, data, max_degree, epochs):
        self.data, self.max_degree, self.epochs = data, int(max_degree), int(epochs)

    @property
    def padded_epochs(self):
        return int(self.data.shape[1])

    @classmethod
    def from_batch(cls, anm):
        torch = require_gpu()
        x = to_device(anm)
        if x.dim() != 3 or x.shape[1] != x.shape[2]:
            raise ValueError('coefficient batch must have shape (B, N+1, N+1), got {0}'.format(tuple(x.shape)))
        B, N = int(x.shape[0]), int(x.shape[1]) - 1
        bpad = -(-max(B, 1) // 32) * 32
        data = torch.empty(((N + 1) ** 2, bpad), dtype=torch.float64, device=x.device)
        _lib.call('shg_order_major_pack', _ptr(x), N, B, _ptr(data), bpad, _stream())
        return cls(data, N, B)

    def to_batch(self):
        torch = require_gpu()
        N, B = self.max_degree, self.epochs
        out = torch.empty((B, N + 1, N + 1), dtype=torch.float64, device=self.data.device)
        _lib.call('shg_order_major_unpack', _ptr(self.data), N, B, self.padded_epochs, _ptr(out), _stream())
        return out

    @property
    def values(self):
        """the coefficients of the epochs that exist, [(N+1)^2, B]: a view of `data` (row stride = padded epochs)"""
        return self.data[:, :self.epochs]

    def like(self, data):
        return OrderMajorSeries(data, self.max_degree, self.epochs)

    def truncated(self, max_degree):
        """the series of the degrees up to `max_degree` (<= the own one): rows gathered on the device"""
        if max_degree == self.max_degree:
            return self
        if max_degree > self.max_degree:
            raise ValueError('the series holds degrees up to {0}'.format(self.max_degree))
        torch = require_gpu()
        N, M = self.max_degree, int(max_degree)
        rows = np.concatenate([order_major_first_row(N, s) + np.arange(M + 1 - ((s + 1) >> 1)) for s in range(2 * M + 1)])
        index = torch.from_numpy(rows.astype(np.int64)).to(self.data.device)
        return OrderMajorSeries(self.data.index_select(0, index), M, self.epochs)

    def to_array(self, min_degree=0):
        """[B, P] device tensor of the degree-wise vectors (TimeSeries.to_array, grates/gravityfield.py:973-980)"""
        torch = require_gpu()
        index = torch.from_numpy(order_major_rows_of_degreewise(self.max_degree, min_degree)).to(self.data.device)
        return self.values.index_select(0, index).t().contiguous()


def order_major_first_row(N, s):
    """first row of slot s (0: order 0 cosine, 2m - 1: order m cosine, 2m: order m sine) in a series of degree N (csrc/filters.hip: om_row)"""
    if s == 0:
        return 0
    m = (s + 1) >> 1
    cos_row = (N + 1) + 2 * ((m - 1) * (N + 1) - m * (m - 1) // 2)
    return cos_row if s & 1 else cos_row + (N + 1 - m)


_om_index_cache = {}


def order_major_rows_of_degreewise(N, min_degree=0):
    """int64 [P]: the row of an order-major series of degree N that holds entry p of the degree-wise vector of the degrees min_degree .. N
    (the order of utilities.ravel_coefficients: C_n0, C_n1, S_n1, C_n2, ... per degree)"""
    key = (int(N), int(min_degree))
    if key not in _om_index_cache:
        rows = []
        for n in range(min_degree, N + 1):
            rows.append(order_major_first_row(N, 0) + n)
            for m in range(1, n + 1):
                rows.append(order_major_first_row(N, 2 * m - 1) + n - m)
                rows.append(order_major_first_row(N, 2 * m) + n - m)
        _om_index_cache[key] = np.asarray(rows, dtype=np.int64)
    return _om_index_cache[key]


def degree_scale_series(series, weights, first_degree=0):
    """degree-wise scaling (Gaussian / Butterworth) of every epoch of an OrderMajorSeries; degrees below `first_degree` are copied"""
    torch = require_gpu()
    w = to_device(weights)
    if w.numel() != series.max_degree + 1:
        raise ValueError('weights must have max_degree + 1 entries')
    out = torch.empty_like(series.data)
    _lib.call('shg_degree_scale_om', _ptr(w), series.max_degree, int(first_degree), _ptr(series.data), series.epochs, series.padded_epochs, _ptr(out), _stream())
    return series.like(out)


def orderwise_filter_series(blocks_packed, block_offsets, block_max_degree, series):
    """OrderWiseFilter.filter of every epoch of an OrderMajorSeries: one matrix product per block, nothing gathered or scattered."""
    torch = require_gpu()
    out = torch.empty_like(series.data)
    _lib.call('shg_orderwise_filter_om', _ptr(blocks_packed), _ptr(block_offsets), int(block_max_degree), series.max_degree,
              _ptr(series.data), series.epochs, series.padded_epochs, _ptr(out), _stream())
    return OrderMajorSeries(out, series.max_degree, series.epochs)


def orderwise_filter(blocks_packed, block_offsets, block_max_degree, anm):
    """blocks_packed: 1d device tensor, block_offsets: int64 device tensor [2Nb+1]; anm [B, N+1, N+1]."""
    torch = require_gpu()
    x = to_device(anm)
    out = torch.empty_like(x)
    _lib.call('shg_orderwise_filter', _ptr(blocks_packed), _ptr(block_offsets), int(block_max_degree), x.shape[-1] - 1,
              _ptr(x), x.shape[0], _ptr(out), _stream())
    return out


def ddk_blocks(normal_blocks, weights):
    """W_k = (N_k + diag(w[m:]))^-1 N_k for a list of order-wise normal blocks (host ndarrays) -> list of ndarrays."""
    torch = require_gpu()
    nb = normal_blocks[0].shape[0] - 1
    sizes = np.array([b.size for b in normal_blocks], dtype=np.int64)
    offsets = np.concatenate(([0], np.cumsum(sizes)[:-1])).astype(np.int64)
    packed = to_device(np.concatenate([np.ascontiguousarray(b, dtype=np.float64).ravel() for b in normal_blocks]))
    off = torch.from_numpy(offsets).to(packed.device)
    work, out = torch.empty_like(packed), torch.empty_like(packed)
    w = to_device(weights)
    _lib.call('shg_ddk_blocks', _ptr(packed), _ptr(off), int(nb), _ptr(w), _ptr(work), _ptr(out), _stream())
    host = to_host(out)
    return [host[o:o + s].reshape(b.shape).copy() for o, s, b in zip(offsets, sizes, normal_blocks)]


def dense_filter(W, X):
    """Y = W @ X, W [P, P], X [P, T] device tensors."""
    torch = require_gpu()
    W, X = to_device(W), to_device(X)
    out = torch.empty_like(X)
    _lib.call('shg_dense_filter', _ptr(W), W.shape[0], _ptr(X), X.shape[1], _ptr(out), _stream())
    return out


def spd_solve(A, B):
    """X = A^-1 B for symmetric positive definite A (device Cholesky)."""
    torch = require_gpu()
    A, B = to_device(A), to_device(B)
    out = torch.empty_like(B)
    _lib.call('shg_spd_solve', _ptr(A), A.shape[0], _ptr(B), B.shape[1], _ptr(out), _stream())
    return out


def dgemm(A, B):
    """C = A @ B on the fp64 MFMA GEMM (row-major device tensors)."""
    torch = require_gpu()
    A, B = to_device(A), to_device(B)
    M, K = A.shape
    N = B.shape[1]
    out = torch.empty((M, N), dtype=torch.float64, device=A.device)
    _lib.call('shg_dgemm', M, N, K, _ptr(A), K, _ptr(B), N, _ptr(out), N, _stream())
    return out


def gemm(A, B, transa=False, transb=False, alpha=1.0, beta=0.0, out=None):
    """out = alpha op(A) op(B) + beta out on the fp64 MFMA GEMM; A, B, out are 2-d row-major device tensors
    (row strides are honoured, the last dimension must be contiguous)."""
    torch = require_gpu()
    A = A if isinstance(A, torch.Tensor) and A.is_cuda and A.dtype == torch.float64 else to_device(A)
    B = B if isinstance(B, torch.Tensor) and B.is_cuda and B.dtype == torch.float64 else to_device(B)
    for t in (A, B):
        if t.dim() != 2 or (t.numel() > 0 and t.shape[1] > 1 and t.stride(1) != 1):
            raise ValueError('gemm operands must be two-dimensional with a contiguous last dimension')
    M = A.shape[1] if transa else A.shape[0]
    K = A.shape[0] if transa else A.shape[1]
    Kb = B.shape[1] if transb else B.shape[0]
    N = B.shape[0] if transb else B.shape[1]
    if K != Kb:
        raise ValueError('gemm: inner dimensions differ ({0} vs {1})'.format(K, Kb))
    if out is None:
        if beta != 0.0:
            raise ValueError('gemm: beta != 0 needs an output tensor')
        out = torch.empty((M, N), dtype=torch.float64, device=A.device)
    elif tuple(out.shape) == (M, N) and not (N > 1 and out.stride(1) != 1):
        Plan._written(out)
    if tuple(out.shape) != (M, N) or (N > 1 and out.stride(1) != 1):
        raise ValueError('gemm: output must be ({0}, {1}) with a contiguous last dimension'.format(M, N))
    _lib.call('shg_gemm', int(transa), int(transb), M, N, K, float(alpha), _ptr(A), max(A.stride(0), 1), _ptr(B), max(B.stride(0), 1),
              float(beta), _ptr(out), max(out.stride(0), 1), _stream())
    return out


def potrf(A, check=True):
    """Upper Cholesky factor U (A = U^T U) of a symmetric positive definite device matrix, in place; the strictly lower
    triangle is zeroed.  Raises numpy.linalg.LinAlgError like scipy.linalg.cholesky when a pivot is not positive."""
    torch = require_gpu()
    if A.dim() != 2 or A.shape[0] != A.shape[1] or (A.shape[1] > 1 and A.stride(1) != 1):
        raise ValueError('potrf: square matrix with a contiguous last dimension expected')
    info = torch.zeros(1, dtype=torch.int32, device=A.device)
    _lib.call('shg_potrf', A.shape[0], _ptr(A), max(A.stride(0), 1), _ptr(info), _stream())
    if check:
        k = int(info.item())
        if k:
            import numpy as np
            raise np.linalg.LinAlgError('{0}-th leading minor of the array is not positive definite'.format(k))
    return A


def trtri(U):
    """Inverse of an upper triangular device matrix (new tensor)."""
    torch = require_gpu()
    if U.dim() != 2 or U.shape[0] != U.shape[1] or (U.shape[1] > 1 and U.stride(1) != 1):
        raise ValueError('trtri: square matrix with a contiguous last dimension expected')
    X = torch.empty((U.shape[0], U.shape[0]), dtype=torch.float64, device=U.device)
    _lib.call('shg_trtri', U.shape[0], _ptr(U), max(U.stride(0), 1), _ptr(X), max(X.stride(0), 1), _stream())
    return X


def transpose_in_place(A):
    """A <- A^T for a square 2-d device tensor with a contiguous last dimension, in its own storage"""
    require_gpu()
    if A.dim() != 2 or A.shape[0] != A.shape[1] or (A.shape[1] > 1 and A.stride(1) != 1):
        raise ValueError('transpose_in_place: a square matrix with a contiguous last dimension is expected')
    _lib.call('shg_transpose_in_place', A.shape[0], _ptr(A), max(A.stride(0), 1), _stream())
    return A


def axpby(alpha, X, beta, Y):
    """Y = alpha X + beta Y in place for 2-d device tensors of equal shape (row strides honoured)."""
    require_gpu()
    if X.dim() != 2 or tuple(X.shape) != tuple(Y.shape):
        raise ValueError('axpby: two-dimensional operands of equal shape expected')
    if X.numel() == 0:
        return Y
    if (X.shape[1] > 1 and X.stride(1) != 1) or (Y.shape[1] > 1 and Y.stride(1) != 1):
        raise ValueError('axpby: contiguous last dimension expected')
    Plan._written(Y)
    _lib.call('shg_axpby', X.shape[0], X.shape[1], float(alpha), _ptr(X), max(X.stride(0), 1), float(beta), _ptr(Y), max(Y.stride(0), 1), _stream())
    return Y
