"""
Spatial filters with the interface of ``grates.filter``: ``Gaussian`` (grates/filter.py:31-95),
``Butterworth`` (:98-130), ``OrderWiseFilter`` (:133-222), ``DDKGeneric`` / ``DDK`` (:225-349),
``BlockedNormalsVDK`` (:352-427), ``GeneralMatrix`` (:430-509), ``VDK`` (:512-546) and ``FilterKernel`` (:575-598).

``filter(gravityfield)`` keeps the reference semantics (new object, input untouched); every filter also
offers ``filter_batch(anm_batch)`` which filters a whole [T, N+1, N+1] stack of epochs in one GPU call --
that is the path the DDK time-series configuration uses.  The coefficient arithmetic runs in libshg
(degree scaling, order-wise block mat-vec, dense fp64 MFMA multiply).
"""

import abc

import numpy as np

from . import data, engine, kernel as _kernel, utilities
from . import gravityfield as _gravityfield
from .gravityfield import PotentialCoefficients, TimeSeries


class SpatialFilter(metaclass=abc.ABCMeta):
    """A filter maps a PotentialCoefficients instance to a filtered copy and can be expressed as a matrix."""

    @abc.abstractmethod
    def filter(self, gravityfield):
        pass

    @abc.abstractmethod
    def matrix(self, min_degree, max_degree):
        pass

    @staticmethod
    def _check(gravityfield):
        if not isinstance(gravityfield, PotentialCoefficients):
            raise TypeError("Filter operation only implemented for instances of 'PotentialCoefficients'")

    def filter_batch(self, anm_batch):
        """Filter a stack [T, N+1, N+1] of coefficient arrays on the GPU; returns a device tensor."""
        raise NotImplementedError

    def filter_covariance(self, covariance_matrix, min_degree, max_degree):
        """
        Covariance matrix of the filtered coefficients, W Sigma W^T with W = self.matrix(min_degree, max_degree) and Sigma in
        degree-wise order -- what scripts on the reference write as ``W @ cov @ W.T`` ahead of
        ``RegularGrid.covariance_propagation`` (grates/filter.py:74-95, 193-222, 481-509; grates/grid.py:792-839).  Extension:
        two fp64 MFMA GEMMs on the device (the second forms the upper tiles only); returns a device tensor [P, P].
        """
        return engine.congruence(self.matrix(min_degree, max_degree), covariance_matrix)

    def _filter_single(self, gravityfield):
        result = gravityfield.copy()
        result.anm = engine.to_host(self.filter_batch(gravityfield.anm[np.newaxis, :, :])[0])
        return result

    def filter_series(self, series):
        """Filter every epoch of an engine.OrderMajorSeries; the default goes through the reference arrays (`filter_batch`), the
        order-wise, degree-wise and dense filters work on the series itself."""
        return engine.OrderMajorSeries.from_batch(self.filter_batch(series.to_batch()))

    def _filter_timeseries(self, series):
        """`filter(TimeSeries)`: the series is filtered on the device, all epochs in one call, and stays there (extension of the
        reference's `filter`, which takes one PotentialCoefficients: grates/filter.py:44-72, 153-191, 456-479).  A series of fields of
        different degrees is filtered field by field, as the reference's call would (every result keeps its field's degree)."""
        if not series.uniform_degree:
            return TimeSeries([self.filter(field) for _, field in series.items()])
        return series._with_series(self.filter_series(series.to_device()))


class _DegreeWiseFilter(SpatialFilter):
    """Filters that scale every coefficient by a factor w_n of its degree."""

    first_degree = 0

    @abc.abstractmethod
    def weights(self, max_degree):
        pass

    def filter_batch(self, anm_batch):
        nmax = anm_batch.shape[-1] - 1
        return engine.degree_scale(anm_batch, self.weights(nmax), self.first_degree)

    def filter_series(self, series):
        return engine.degree_scale_series(series, self.weights(series.max_degree), self.first_degree)

    def filter(self, gravityfield):
        if isinstance(gravityfield, TimeSeries):
            return self._filter_timeseries(gravityfield)
        self._check(gravityfield)
        return self._filter_single(gravityfield)

    def matrix(self, min_degree, max_degree):
        wn = self.weights(max_degree)
        arr = np.zeros((max_degree + 1, max_degree + 1))
        for n in range(min_degree, max_degree + 1):
            arr[_gravityfield.degree_indices(n)] = wn[n]
        return np.diag(utilities.ravel_coefficients(arr, min_degree, max_degree))


class Gaussian(_DegreeWiseFilter):
    """
    Gaussian filter with `radius` in kilometres.  `filter` leaves degrees 0 and 1 untouched
    (grates/filter.py:69-70) while `matrix` scales every degree >= min_degree (:92-95).
    """

    first_degree = 2

    def __init__(self, radius):
        self.radius = radius

    def weights(self, max_degree):
        kn = _kernel.Gauss(self.radius)
        return np.array([kn.coefficient(n)[0] for n in range(max_degree + 1)])


class Butterworth(_DegreeWiseFilter):
    """Butterworth filter on the sphere: w_n = (1 + (n / n_c)^(2 order))^(-1/2) (grates/filter.py:107-130)."""

    def __init__(self, order, cutoff_degree):
        self.order = order
        self.cutoff_degree = cutoff_degree

    def weights(self, max_degree):
        n = np.arange(max_degree + 1, dtype=float)
        return np.power(1 + (n / self.cutoff_degree) ** (2 * self.order), -0.5)


class OrderWiseFilter(SpatialFilter):
    """
    Filter with a block-diagonal matrix: one dense block per order and per cosine / sine
    (Kusche et al. 2009).  `orderwise_blocks` = [order0_cos, order1_cos, order1_sin, ...]; block m is
    indexed by degrees m..nmax (grates/filter.py:148-151).
    """

    def __init__(self, orderwise_blocks):
        self.__array = orderwise_blocks
        self.__nmax = orderwise_blocks[0].shape[0] - 1
        self.__device_blocks = None

    def _device_blocks(self):
        if self.__device_blocks is None:
            torch = engine.require_gpu()
            sizes = np.array([b.size for b in self.__array], dtype=np.int64)
            offsets = np.concatenate(([0], np.cumsum(sizes)[:-1])).astype(np.int64)
            packed = np.concatenate([np.ascontiguousarray(b, dtype=np.float64).ravel() for b in self.__array])
            self.__device_blocks = (engine.to_device(packed), torch.from_numpy(offsets).to(engine.device()))
        return self.__device_blocks

    def filter_batch(self, anm_batch):
        nmax = anm_batch.shape[-1] - 1
        if nmax > self.__nmax:
            raise ValueError('DDK filter only implemented for a maximum degree of {1:d} (max_degree={0:d} supplied).'.format(nmax, self.__nmax))
        packed, offsets = self._device_blocks()
        return engine.orderwise_filter(packed, offsets, self.__nmax, anm_batch)

    def filter_series(self, series):
        """Filter every epoch of an engine.OrderMajorSeries (a time series kept on the device in order-major layout): one matrix
        product per block on whole matrices; returns a new series.  The values are those of `filter` applied epoch by epoch
        (grates/filter.py:153-191), degrees 0 and 1 restored from the input."""
        if series.max_degree > self.__nmax:
            raise ValueError('DDK filter only implemented for a maximum degree of {1:d} (max_degree={0:d} supplied).'.format(series.max_degree, self.__nmax))
        packed, offsets = self._device_blocks()
        return engine.orderwise_filter_series(packed, offsets, self.__nmax, series)

    def filter(self, gravityfield):
        """Filtered copy; degrees 0 and 1 are restored from the input; ValueError above the block degree
        (grates/filter.py:172-191).  A TimeSeries is filtered as a whole on the device (`filter_series`) and stays there."""
        if isinstance(gravityfield, TimeSeries):
            return self._filter_timeseries(gravityfield)
        self._check(gravityfield)
        return self._filter_single(gravityfield)

    def matrix(self, min_degree, max_degree):
        """Dense filter matrix in degree-wise order (grates/filter.py:209-222)."""
        count = (max_degree + 1) * (max_degree + 1)
        W = np.zeros((count, count))
        index = np.arange(max_degree + 1, dtype=int) ** 2
        W[np.ix_(index, index)] = self.__array[0][0:max_degree + 1, 0:max_degree + 1]
        for m in range(1, max_degree + 1):
            for block, shift in ((self.__array[2 * m - 1], 2 * m - 1), (self.__array[2 * m], 2 * m)):
                W[np.ix_(index[m:] + shift, index[m:] + shift)] = block[0:max_degree + 1 - m, 0:max_degree + 1 - m]
        return W[min_degree * min_degree:, min_degree * min_degree:]


def _regularised_blocks(normals, weights):
    """(N_m + diag(w[m:]))^-1 N_m for every order-wise normal block (grates/filter.py:252-255), all blocks in one
    device launch (Cholesky per block: N_m + diag(w) is symmetric positive definite)."""
    return engine.ddk_blocks(normals, weights)


class DDKGeneric(OrderWiseFilter):
    """DDK filter with power-law weights 10^(15 - level) n^4 for any level >= 1 (grates/filter.py:242-257)."""

    def __init__(self, level):
        if level < 1:
            raise ValueError('DDK level must be at least 1 (requested DDK{0:d}).'.format(level))
        normals = DDKGeneric._blocked_normals()
        nmax = normals[0].shape[0] - 1
        weights = 10 ** (15 - level) * np.arange(nmax + 1, dtype=float) ** 4
        weights[0] = 1
        super(DDKGeneric, self).__init__(_regularised_blocks(normals, weights))

    @staticmethod
    def _blocked_normals():
        """Order-wise blocks of the DDK normal equation matrix."""
        return data.ddk_normal_blocks()

    @staticmethod
    def normal_equation_matrix():
        """Dense DDK normal equation matrix in degree-wise order without degrees 0-1 (grates/filter.py:281-297)."""
        normals = DDKGeneric._blocked_normals()
        return OrderWiseFilter(normals).matrix(2, normals[0].shape[0] - 1)


class DDK(OrderWiseFilter):
    """DDK1-DDK8 as used by ICGEM; weights scale[level] n^4 (grates/filter.py:334-349)."""

    SCALE = {1: 1e14, 2: 1e13, 3: 1e12, 4: 5e11, 5: 1e11, 6: 5e10, 7: 1e10, 8: 5e9}

    def __init__(self, level):
        normals = DDKGeneric._blocked_normals()
        nmax = normals[0].shape[0] - 1
        if level not in DDK.SCALE:
            raise ValueError('DDK level must be between 1 and 8 (requested DDK{0}).'.format(level))
        weights = DDK.SCALE[level] * np.arange(nmax + 1, dtype=float) ** 4
        weights[0] = 1
        super(DDK, self).__init__(_regularised_blocks(normals, weights))


class BlockedNormalsVDK(OrderWiseFilter):
    """
    Order-wise (DDK-like) approximation of a VDK filter: the order / basis-function blocks of a full normal
    equation matrix (degree-wise order, degrees min_degree..max_degree) are regularised with Kaula weights
    kaula_scale n^kaula_power (grates/filter.py:382-427).
    """

    def __init__(self, normal_equation_matrix, min_degree, max_degree, kaula_scale, kaula_power):
        weights = kaula_scale * np.arange(max_degree + 1, dtype=float) ** kaula_power
        weights[0] = 1
        seq = _gravityfield.CoefficientSequenceDegreeWise(min_degree, max_degree)
        normals = []
        for m in range(0, max_degree + 1):
            for cs in (('c',) if m == 0 else ('c', 's')):
                idx = seq.vector_indices(order=m, cs=cs)
                size = max_degree + 1 - m
                block = np.zeros((size, size))
                first = max(min_degree - m, 0)
                block[first:, first:] = normal_equation_matrix[np.ix_(idx, idx)]
                normals.append(block)
        super(BlockedNormalsVDK, self).__init__(_regularised_blocks(normals, weights))


class GeneralMatrix(SpatialFilter):
    """
    Filter given by an arbitrary square matrix in degree-wise order for degrees min_degree..max_degree
    (grates/filter.py:445-454).
    """

    def __init__(self, matrix, min_degree, max_degree):
        if matrix.ndim > 2 or matrix.shape[0] != matrix.shape[1]:
            raise ValueError('filter matrix must be square (got {0})'.format(str(matrix.shape)))
        if (max_degree + 1) * (max_degree + 1) - min_degree * min_degree != matrix.shape[0]:
            raise ValueError('filter matrix dimensions do not correspond to min_degree and max_degree (got {0}, {1:d}, {2:d})'.format(str(matrix.shape), min_degree, max_degree))
        self.__W = matrix
        self.__nmin = min_degree
        self.__nmax = max_degree
        self.__device_W = None
        self.__device_W_om = None

    def filter_batch(self, anm_batch):
        """
        [T, Na+1, Na+1] -> [T, min(Na, nmax)+1, ...]: ravel on the device, one dense W @ X multiply on the fp64
        MFMA GEMM for all epochs, unravel, restore degrees below min_degree (grates/filter.py:470-479).
        """
        if self.__device_W is None:
            self.__device_W = engine.to_device(self.__W)
        x_in = engine.to_device(anm_batch)
        na = x_in.shape[-1] - 1
        nmax_out = min(na, self.__nmax)
        X = engine.ravel(x_in, self.__nmin, self.__nmax).T.contiguous()          # [P, T], epoch fastest
        Y = engine.dense_filter(self.__device_W, X)
        P_out = (nmax_out + 1) ** 2 - self.__nmin ** 2
        out = engine.unravel(Y[0:P_out].T.contiguous(), self.__nmin, nmax_out)
        k = min(self.__nmin, nmax_out + 1)
        out[:, 0:k, 0:k] = x_in[:, 0:k, 0:k]
        return out

    def _order_major_matrix(self):
        """W in the row / column order of an order-major series of degree nmax, the identity on the degrees below min_degree (which the
        filter restores from the input): one permuted copy on the device, built on first use"""
        if self.__device_W_om is None:
            torch = engine.require_gpu()
            if self.__device_W is None:
                self.__device_W = engine.to_device(self.__W)
            n_all = (self.__nmax + 1) ** 2
            rows = torch.from_numpy(engine.order_major_rows_of_degreewise(self.__nmax, self.__nmin)).to(self.__device_W.device)
            W_om = torch.zeros((n_all, n_all), dtype=torch.float64, device=self.__device_W.device)
            low = torch.from_numpy(engine.order_major_rows_of_degreewise(self.__nmax, 0)[0:self.__nmin ** 2]).to(rows.device)
            W_om[low, low] = 1.0
            W_om[rows[:, None], rows[None, :]] = self.__device_W
            self.__device_W_om = W_om
        return self.__device_W_om

    def filter_series(self, series):
        """Y = W X on the order-major series itself: the rows and columns of W are permuted once, every call is then ONE product on the
        fp64 MFMA GEMM -- no ravel, no unravel.  (A series of another degree than the filter's goes through the reference arrays.)"""
        if series.max_degree != self.__nmax:
            return super(GeneralMatrix, self).filter_series(series)
        torch = engine.require_gpu()
        out = torch.empty_like(series.data)
        if series.padded_epochs > series.epochs:
            out[:, series.epochs:] = 0.0
        engine.gemm(self._order_major_matrix(), series.values, out=out[:, :series.epochs])
        return series.like(out)

    def filter(self, gravityfield):
        if isinstance(gravityfield, TimeSeries):
            return self._filter_timeseries(gravityfield)
        return self._filter_single(gravityfield)

    def matrix(self, min_degree, max_degree):
        """Filter matrix re-indexed to another degree range (zero where the filter is not defined)."""
        if self.__nmin == min_degree and self.__nmax == max_degree:
            return self.__W.copy()
        target = _gravityfield.CoefficientSequenceDegreeWise(min_degree, max_degree)
        source = _gravityfield.CoefficientSequenceDegreeWise(self.__nmin, self.__nmax)
        W = np.zeros((target.coefficient_count, target.coefficient_count))
        idx_source, idx_target = _gravityfield.CoefficientSequence.reorder_indices(source, target)
        W[np.ix_(idx_target, idx_target)] = self.__W[np.ix_(idx_source, idx_source)].copy()
        return W


class VDK(GeneralMatrix):
    """
    Decorrelation filter from a full normal equation matrix N (degree-wise order, degrees min_degree..max_degree)
    regularised with Kaula weights: W = (N + diag(kaula_scale n^kaula_power))^-1 N (grates/filter.py:536-546), formed
    on the device by a Cholesky solve.  `filter` is the dense matrix filter of GeneralMatrix (the reference's own
    VDK.filter cannot run: it reads name-mangled attributes of its base class).
    """

    def __init__(self, normal_equation_matrix, min_degree, max_degree, kaula_scale, kaula_power):
        normals = np.asarray(normal_equation_matrix, dtype=float)
        count = (max_degree + 1) ** 2 - min_degree ** 2
        if normals.ndim != 2 or normals.shape != (count, count):
            raise ValueError('normal equation matrix does not match degrees {0:d} to {1:d} (got {2})'.format(min_degree, max_degree, str(normals.shape)))
        weights = np.concatenate([np.full(2 * n + 1, kaula_scale * float(n) ** kaula_power) for n in range(min_degree, max_degree + 1)]) \
            if count else np.zeros(0)
        regularised = normals.copy()
        regularised.flat[::count + 1] += weights
        W = engine.to_host(engine.spd_solve(regularised, normals))
        super(VDK, self).__init__(W, min_degree, max_degree)


class FilterKernel(_kernel.AnisotropicKernel):
    """
    Space-domain kernel of a (possibly anisotropic) spatial filter (grates/filter.py:588-598): the filter matrix of
    degrees min_degree..max_degree -- from a SpatialFilter or given directly -- as an AnisotropicKernel.

    As executed by the reference the factors k_n and 1/k_n of `input_kernel` both scale the COLUMNS of the filter matrix
    (its kernel coefficient arrays carry a leading axis of length one), so they cancel wherever k_n is non-zero and
    zero the columns where it vanishes; this class reproduces that result.
    """

    def __init__(self, spatial_filter, min_degree, max_degree, input_kernel='potential'):
        K = spatial_filter.matrix(min_degree, max_degree) if isinstance(spatial_filter, SpatialFilter) else np.asarray(spatial_filter, dtype=float)
        generator = _kernel.get_kernel(input_kernel)
        kn = utilities.ravel_coefficients(generator.coefficient_array(min_degree, max_degree)[0], min_degree, max_degree)
        kn_inverse = utilities.ravel_coefficients(generator.inverse_coefficient_array(min_degree, max_degree)[0], min_degree, max_degree)
        super(FilterKernel, self).__init__(K * (kn * kn_inverse)[np.newaxis, :], min_degree, max_degree)
