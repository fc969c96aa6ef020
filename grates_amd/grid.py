"""
Point distributions on the ellipsoid with the interface of ``grates.grid``: the ``Grid`` base class with
its area-weighted statistics (grates/grid.py:92-507), ``RegularGrid`` (:510-839), ``IrregularGrid``
(:842-1120), ``GeographicGrid`` (:1123-1162) and ``GaussGrid`` (:1165-1204), plus the coordinate helpers the
kernels need.  Reuter / geodesic / mascon grids, basins and point-in-polygon tests are host-side geometry
and out of scope.

Heavy operators -- ``to_potential_coefficients`` (analysis), ``covariance_propagation``,
``synthesis_matrix`` / ``analysis_matrix`` -- run on the GPU through libshg.
"""

import abc

import numpy as np

from . import engine, kernel as _kernel, utilities
from . import gravityfield as _gravityfield

_GM = 3.9860044150e+14
_R = 6.3781363000e+06


class Grid(metaclass=abc.ABCMeta):
    """Base interface for point collections."""

    # interface every grid type provides: copy(), the ellipsoid (semimajor_axis, flattening), per-point longitude / latitude /
    # area, values (None or one value per point) and point_count
    copy = abc.abstractmethod(lambda self: None)
    semimajor_axis, flattening, longitude, latitude, area = (property(abc.abstractmethod(lambda self: None)) for _ in range(5))
    values = abc.abstractmethod(lambda self: None)
    point_count = abc.abstractmethod(lambda self: None)

    @property
    def size(self):
        return self.point_count

    @property
    def colatitude(self):
        return utilities.colatitude(self.latitude, self.semimajor_axis, self.flattening)

    def is_compatible(self, other):
        """True if both grids have numerically equal point coordinates."""
        if self.point_count == other.point_count:
            return np.allclose(self.longitude, other.longitude) and np.allclose(self.latitude, other.latitude)
        return False

    def cartesian_coordinates(self):
        return geodetic2cartesian(self.longitude, self.latitude, h=0, a=self.semimajor_axis, f=self.flattening)

    # ---- area-weighted statistics (grates/grid.py:174-260) -------------------------------------------------
    def __weights(self, mask):
        if mask is None:
            mask = np.ones(self.point_count, dtype=bool)
        areas = self.area
        w = np.ones(np.count_nonzero(mask)) if areas is None else areas[mask]
        return mask, w

    def mean(self, mask=None):
        mask, w = self.__weights(mask)
        return np.sum(w * self.values[mask]) / np.sum(w)

    def rms(self, mask=None):
        mask, w = self.__weights(mask)
        return np.sqrt(np.sum(w * self.values[mask] ** 2) / np.sum(w))

    def std(self, mask=None):
        mask, w = self.__weights(mask)
        centred = self.values[mask] - self.mean(mask)
        return np.sqrt(np.sum(w * centred ** 2) / np.sum(w))

    def distance_matrix(self):
        """Spherical distance [rad] between all pairs of grid points."""
        lon, lat = self.longitude, self.latitude
        return spherical_distance(lon[:, np.newaxis], lat[:, np.newaxis], lon[np.newaxis, :], lat[np.newaxis, :], r=1)

    def subset(self, mask):
        """The points selected by the boolean `mask` as a new grid: a RegularGrid when they still form full parallels and
        meridians, an IrregularGrid otherwise (grates/grid.py:303-327).  Values are not carried over (as upstream)."""
        mask = np.asarray(mask, dtype=bool).ravel()
        points = IrregularGrid(self.longitude[mask], self.latitude[mask], None if self.area is None else self.area[mask],
                               self.semimajor_axis, self.flattening)
        try:
            return points.to_regular()
        except ValueError:
            return points

    def nn_index(self, lon, lat):
        """For every grid point the indices of the sample points (lon, lat) whose nearest grid point it is, by 3D euclidean
        distance (grates/grid.py:329-356)."""
        import scipy.spatial
        sample = IrregularGrid(np.atleast_1d(lon), np.atleast_1d(lat), a=self.semimajor_axis, f=self.flattening).cartesian_coordinates()
        _, nearest = scipy.spatial.cKDTree(self.cartesian_coordinates()).query(sample)
        order = np.argsort(nearest, kind='stable')
        bounds = np.searchsorted(nearest[order], np.arange(self.point_count + 1))
        return [order[bounds[k]:bounds[k + 1]] for k in range(self.point_count)]

    # ---- linear operators ------------------------------------------------------------------------------------------
    @abc.abstractmethod
    def synthesis_matrix_per_order(self, m, min_degree, max_degree, kernel, GM, R):
        pass

    def synthesis_matrix_device(self, min_degree, max_degree, kernel='potential', GM=_GM, R=_R):
        """The dense synthesis operator as a device tensor [points, coefficients] (one generation kernel, shg_synthesis_matrix)."""
        colat, lon, kn = self._point_tables(kernel, max_degree, GM, R)
        return engine.synthesis_matrix(max_degree, min_degree, colat, lon, kn)

    def synthesis_matrix(self, min_degree, max_degree, kernel='potential', GM=_GM, R=_R):
        """Dense operator A (points x coefficients, degree-wise columns) mapping coefficients to grid values
        (grates/grid.py:412-443)."""
        return engine.to_host(self.synthesis_matrix_device(min_degree, max_degree, kernel, GM, R))

    @abc.abstractmethod
    def analysis_matrix(self, min_degree, max_degree, kernel, GM, R):
        pass

    def analysis_matrix_device(self, min_degree, max_degree, kernel='potential', GM=_GM, R=_R):
        """The dense analysis operator as a device tensor [coefficients, points]."""
        return engine.to_device(self.analysis_matrix(min_degree, max_degree, kernel, GM, R))

    def window_matrix(self, min_degree, max_degree, kernel='potential', GM=_GM, R=_R):
        """W = (F * values) A with the grid values as window function (grates/grid.py:472-475); both operators are generated
        on the device and multiplied there, only W is copied back."""
        if self.values is None:
            raise TypeError('grid has no values to use as window function')           # upstream: ndarray *= None
        F = self.analysis_matrix_device(min_degree, max_degree, kernel, GM, R)
        engine.scale_columns(F, self.values)
        return engine.to_host(engine.gemm(F, self.synthesis_matrix_device(min_degree, max_degree, kernel, GM, R)))

    def to_potential_coefficients(self, min_degree, max_degree, kernel='potential', GM=_GM, R=_R):
        """Spherical harmonic analysis through the full analysis matrix (grates/grid.py:498-507)."""
        if self.values is None:
            raise ValueError('grid has no values to propagate to potential coefficients')
        F = engine.to_device(self.analysis_matrix(min_degree, max_degree, kernel, GM, R))
        x = engine.dgemm(F, engine.to_device(self.values).reshape(-1, 1))
        coeffs = _gravityfield.PotentialCoefficients(GM, R)
        coeffs.anm = utilities.unravel_coefficients(engine.to_host(x).ravel(), min_degree, max_degree)
        return coeffs

    def _point_tables(self, kernel, max_degree, GM, R):
        colat, _, kn = _gravityfield.surface_factors(_kernel.get_kernel(kernel), max_degree, self.latitude, GM, R,
                                                     self.semimajor_axis, self.flattening)
        return colat, self.longitude, kn


def _degree_scale_array(kn, max_degree):
    """Device tensor [k, N+1, N+1] holding kn[k, degree of slot]."""
    torch = engine.require_gpu()
    t = engine.to_device(kn)
    idx = torch.arange(max_degree + 1, device=t.device)
    deg = torch.maximum(idx[:, None], idx[None, :])
    return t[:, deg]


class RegularGrid(Grid):
    """
    Regular global point distribution given by meridians (longitudes) and parallels (latitudes, north to
    south) in radians on the ellipsoid (a, f).  `value_array` is (parallels, meridians).
    """

    def __init__(self, meridians, parallels, area_elements=None, a=6378137.0, f=298.2572221010**-1):
        self.parallels = parallels
        self.meridians = meridians
        self.__a = a
        self.__f = f
        if area_elements is None:
            lon_edges = np.concatenate(([-np.pi], self.meridians[0:-1] + 0.5 * np.diff(self.meridians), [np.pi]))
            lat_edges = np.concatenate(([0.5 * np.pi], self.parallels[0:-1] + 0.5 * np.diff(self.parallels), [-0.5 * np.pi]))
            area_elements = 2.0 * (np.sin(np.abs(np.diff(lat_edges)) * 0.5) * np.cos(self.parallels))[:, np.newaxis] * np.diff(lon_edges)
        self.__areas = area_elements
        self.value_array = None
        self.epoch = None

    def copy(self):
        grid = RegularGrid(self.meridians.copy(), self.parallels.copy(), self.__areas.copy(), self.semimajor_axis, self.flattening)
        if self.value_array is not None:
            grid.values = self.values.copy()
        grid.epoch = self.epoch
        return grid

    def to_regular(self, threshold=1e-6):
        """A regular grid is its own regular representation (a copy); the threshold is only validated."""
        if not threshold > 0:
            raise ValueError('threshold should be positive (got {0:e})'.format(threshold))
        return self.copy()

    # ellipsoid and per-point views (points are enumerated parallel by parallel, north to south)
    semimajor_axis = property(lambda self: self.__a)
    flattening = property(lambda self: self.__f)
    point_count = property(lambda self: self.parallels.size * self.meridians.size)
    longitude = property(lambda self: np.tile(self.meridians, self.parallels.size))
    latitude = property(lambda self: np.repeat(self.parallels, self.meridians.size))
    area = property(lambda self: self.__areas.ravel())

    def __get_values(self):
        return None if self.value_array is None else self.value_array.ravel()

    def __set_values(self, val):
        """None clears the grid; a 1-d ndarray with one value per point is reshaped to (parallels, meridians); anything else
        is a ValueError (grates/grid.py:614-625)"""
        if val is None:
            self.value_array = None
            return
        if not isinstance(val, np.ndarray):
            raise ValueError("grid values must be either None or " + str(np.ndarray))
        if val.ndim > 1:
            raise ValueError("unable to assign values of dimension {0:d} to grid".format(val.ndim))
        if val.size != self.point_count:
            raise ValueError("unable to assign values of size {0:d} to grid with {1:d} points".format(val.size, self.point_count))
        self.value_array = np.reshape(val, (self.parallels.size, self.meridians.size))

    values = property(__get_values, __set_values)

    # ---- tables ----------------------------------------------------------------------------------------------------
    def _parallel_tables(self, kernel, max_degree, GM, R):
        return _gravityfield.surface_factors(_kernel.get_kernel(kernel), max_degree, self.parallels, GM, R,
                                             self.semimajor_axis, self.flattening)

    def _plan(self, kernel, max_degree, GM, R):
        colat, _, kn = self._parallel_tables(kernel, max_degree, GM, R)
        return engine.cached_plan(max_degree, colat, kn, self.meridians)

    def synthesis_matrix_per_order(self, m, min_degree, max_degree, kernel, GM, R):
        """Operator block of order m (rows parallel-major): one matrix for m = 0, a (cosine, sine) tuple
        otherwise; columns are degrees max(m, min_degree)..max_degree (grates/grid.py:653-663)."""
        colat, _, kn = self._parallel_tables(kernel, max_degree, GM, R)
        Ac, As = engine.synthesis_matrix_order(max_degree, m, min_degree, colat, self.meridians, kn, pointwise=False)
        return engine.to_host(Ac) if m == 0 else (engine.to_host(Ac), engine.to_host(As))

    def analysis_matrix_device(self, min_degree, max_degree, kernel='potential', GM=_GM, R=_R):
        """The dense analysis operator as a device tensor [coefficients, points]: the cached per-order least-squares operator
        of the plan written out by one kernel (shg_analysis_matrix)."""
        plan = self._plan(kernel, max_degree, GM, R)
        return plan.analysis_matrix(self.area.reshape(self.parallels.size, self.meridians.size), min_degree)

    def analysis_matrix(self, min_degree, max_degree, kernel, GM=_GM, R=_R):
        """Dense analysis operator (coefficients x points): area-weighted least squares per order (grates/grid.py:698-730)."""
        return engine.to_host(self.analysis_matrix_device(min_degree, max_degree, kernel, GM, R))

    def to_potential_coefficients(self, min_degree, max_degree, kernel='potential', GM=_GM, R=_R):
        """Area-weighted least-squares analysis, order by order, on the GPU (grates/grid.py:774-790)."""
        if self.values is None:
            raise ValueError('grid has no values to propagate to potential coefficients')
        plan = self._plan(kernel, max_degree, GM, R)
        anm = plan.analysis(self.value_array, self.area.reshape(self.parallels.size, self.meridians.size), min_degree)
        coeffs = _gravityfield.PotentialCoefficients(GM, R)
        coeffs.anm = engine.to_host(anm)
        return coeffs

    def covariance_propagation(self, covariance_matrix, min_degree, max_degree, kernel='potential', GM=_GM, R=_R,
                               parallel_range=None, symmetric=False, method='direct'):
        """
        Standard deviation of the gridded functional given the coefficient covariance matrix in degree-wise
        order: sqrt(diag(A Sigma A^T)) with A generated on the fly (grates/grid.py:817-839).  Sets and returns
        the grid values.  Extensions: `parallel_range=(i0, i1)` restricts the computation to a latitude band
        and returns only that band without touching the grid values; `symmetric=True` reads only the upper
        triangle of a symmetric matrix (half the work), `symmetric=None` does so when the matrix is found to be
        exactly symmetric; `method='separable'` evaluates the same quadratic forms through the latitude / longitude
        factorisation of A (about nlon times fewer flops; summation order differs, nothing else).
        """
        plan = self._plan(kernel, max_degree, GM, R)
        if parallel_range is not None:
            return engine.to_host(plan.covariance_propagation(covariance_matrix, min_degree, parallel_range[0], parallel_range[1], symmetric=symmetric, method=method))
        sigma = engine.to_host(plan.covariance_propagation(covariance_matrix, min_degree, symmetric=symmetric, method=method))
        self.values = sigma
        return sigma.copy()

    def covariance_blocks(self, covariance_matrix, min_degree, max_degree, kernel='potential', GM=_GM, R=_R, parallel_range=None):
        """
        The full covariance matrix of the points of every parallel, F Sigma F^T [nlon, nlon] with F the rows of the synthesis
        matrix of that parallel -- the product the reference forms per parallel and of which covariance_propagation keeps
        the diagonal (grates/grid.py:833-835).  Returns a device tensor [parallels, nlon, nlon] for the parallels
        [i0, i1) = parallel_range (default: all); per parallel two fp64 MFMA GEMMs (nlon x P x P and nlon x P x nlon).
        """
        torch = engine.require_gpu()
        i0, i1 = (0, self.parallels.size) if parallel_range is None else parallel_range
        nlon = self.meridians.size
        colat, _, kn = self._parallel_tables(kernel, max_degree, GM, R)
        cov = engine.to_device(covariance_matrix)
        P = (max_degree + 1) ** 2 - min_degree ** 2
        if cov.dim() != 2 or cov.shape[0] != P or cov.shape[1] != P:
            raise ValueError('covariance matrix must have shape ({0}, {0}), got {1}'.format(P, tuple(cov.shape)))
        out = torch.empty((i1 - i0, nlon, nlon), dtype=torch.float64, device=cov.device)
        band = max(1, min(i1 - i0, (1 << 28) // max(nlon * P, 1)))               # parallels per pass: 2 GB of synthesis-matrix rows
        for b0 in range(i0, i1, band):
            b1 = min(b0 + band, i1)
            A = engine.synthesis_matrix(max_degree, min_degree, np.repeat(colat[b0:b1], nlon), np.tile(self.meridians, b1 - b0),
                                        np.repeat(kn[b0:b1], nlon, axis=0))
            T = engine.gemm(A, cov)
            for i in range(b0, b1):
                rows = slice((i - b0) * nlon, (i - b0 + 1) * nlon)
                engine.gemm(T[rows], A[rows], transb=True, out=out[i - i0])
        return out


class IrregularGrid(Grid):
    """Arbitrary point list given by longitude / latitude pairs [rad]."""

    def __init__(self, longitude, latitude, area_element=None, a=6378137.0, f=298.2572221010**-1):
        count = longitude.size
        # points: (longitude, latitude, area); without area elements every point gets an equal share of the unit sphere
        self.__points = (longitude, latitude, np.full(count, 4 * np.pi / count) if area_element is None else area_element)
        self.__ellipsoid = (a, f)
        self.__values = None
        self.epoch = None

    def copy(self):
        other = IrregularGrid(*(part.copy() for part in self.__points), *self.__ellipsoid)
        other.values = None if self.__values is None else self.__values.copy()
        other.epoch = self.epoch
        return other

    semimajor_axis = property(lambda self: self.__ellipsoid[0])
    flattening = property(lambda self: self.__ellipsoid[1])
    longitude = property(lambda self: self.__points[0])
    latitude = property(lambda self: self.__points[1])
    area = property(lambda self: self.__points[2])
    point_count = property(lambda self: self.__points[0].size)

    def __set_values(self, val):
        """None, or a 1-d ndarray with one value per point (ValueError otherwise, grates/grid.py:936-947)"""
        if val is not None:
            if not isinstance(val, np.ndarray):
                raise ValueError("grid values must be either None or " + str(np.ndarray))
            if val.ndim > 1:
                raise ValueError("unable to assign values of dimension {0:d} to grid".format(val.ndim))
            if val.size != self.point_count:
                raise ValueError("unable to assign values of size {0:d} to grid with {1:d} points".format(val.size, self.point_count))
        self.__values = val

    values = property(lambda self: self.__values, __set_values)

    def to_regular(self, threshold=1e-6):
        """Coerce into a RegularGrid if the points form meridians x parallels (grates/grid.py:886-914)."""
        if threshold <= 0:
            raise ValueError('threshold should be positive (got {0:e})'.format(threshold))
        threshold /= self.semimajor_axis

        def clusters(sorted_values):
            out, k = [], 0
            while k < sorted_values.size and len(out) < self.point_count:
                out.append(sorted_values[k])
                k += np.searchsorted(sorted_values[k + 1:], sorted_values[k] + threshold) + 1
            return out

        meridians = clusters(np.sort(self.longitude))
        parallels = clusters(np.sort(self.latitude))
        if len(meridians) * len(parallels) != self.point_count:
            raise ValueError('grid cannot be coerced to a regular sampling')
        grid = RegularGrid(np.array(meridians), np.array(parallels[::-1]), a=self.semimajor_axis, f=self.flattening)
        if self.values is not None:
            import scipy.spatial
            tree = scipy.spatial.cKDTree(np.vstack((self.longitude, self.latitude)).T)
            _, index = tree.query(np.vstack((grid.longitude, grid.latitude)).T)
            grid.values = self.values[index]
        return grid

    def synthesis_matrix_per_order(self, m, min_degree, max_degree, kernel, GM, R):
        """Operator block of order m for a point list (grates/grid.py:981-991)."""
        colat, lon, kn = self._point_tables(kernel, max_degree, GM, R)
        Ac, As = engine.synthesis_matrix_order(max_degree, m, min_degree, colat, lon, kn, pointwise=True)
        return engine.to_host(Ac) if m == 0 else (engine.to_host(Ac), engine.to_host(As))

    def analysis_matrix(self, min_degree, max_degree, kernel='potential', GM=_GM, R=_R):
        """(A^T W A)^-1 A^T W with W = diag(area) (grates/grid.py:1015-1017); the normal matrix and the
        right-hand side are formed with the fp64 MFMA GEMM, the normal equations are solved by a device Cholesky."""
        torch = engine.require_gpu()
        A = engine.to_device(self.synthesis_matrix(min_degree, max_degree, kernel, GM, R))
        sw = torch.sqrt(engine.to_device(self.area))
        A = A * sw[:, None]
        At = A.T.contiguous()
        normal = engine.dgemm(At, A)
        return engine.to_host(engine.spd_solve(normal, (At * sw[None, :]).contiguous()))

    def covariance_propagation(self, covariance_matrix, min_degree, max_degree, kernel='potential', GM=_GM, R=_R):
        """Point-list covariance propagation (grates/grid.py:1096-1120).  Sets and returns the grid values."""
        colat, lon, kn = self._point_tables(kernel, max_degree, GM, R)
        sigma = engine.to_host(engine.covprop_points(max_degree, colat, lon, kn, covariance_matrix, min_degree))
        self.values = sigma
        return sigma.copy()


class GeographicGrid(RegularGrid):
    """
    Global geographic grid with step sizes dlon, dlat in degrees; points are pixel centres
    (grates/grid.py:1141-1153).
    """

    def __init__(self, dlon=0.5, dlat=0.5, a=6378137.0, f=298.2572221010**-1):
        self.__dlon = dlon
        self.__dlat = dlat
        nlons = 360 / dlon
        nlats = 180 / dlat
        meridians = np.linspace(-np.pi + dlon / 180 * np.pi * 0.5, np.pi - dlon / 180 * np.pi * 0.5, int(nlons))
        parallels = -np.linspace(-np.pi * 0.5 + dlat / 180 * np.pi * 0.5, np.pi * 0.5 - dlat / 180 * np.pi * 0.5, int(nlats))
        areas = np.tile(2.0 * dlon / 180 * np.pi * np.sin(dlat * 0.5 / 180 * np.pi) * np.cos(parallels)[:, np.newaxis], (1, meridians.size))
        super(GeographicGrid, self).__init__(meridians, parallels, areas, a, f)

    def copy(self):
        grid = GeographicGrid(self.__dlon, self.__dlat, self.semimajor_axis, self.flattening)
        if self.values is not None:
            grid.values = self.values.copy()
        grid.epoch = self.epoch
        return grid


class GaussGrid(RegularGrid):
    """
    Gaussian grid: parallels at the roots of the Legendre polynomial of degree parallel_count (mapped from
    the unit sphere onto the ellipsoid), 2 * parallel_count meridians (grates/grid.py:1181-1195).
    """

    def __init__(self, parallel_count, a=6378137.0, f=298.2572221010**-1):
        from scipy.special import roots_legendre
        zeros, weights, _ = roots_legendre(parallel_count, mu=True)
        dlon = np.pi / parallel_count
        meridians = np.linspace(-np.pi + dlon * 0.5, np.pi - dlon * 0.5, 2 * parallel_count)
        cosine_theta = -zeros
        sine_theta = np.sqrt(1 - cosine_theta ** 2)
        parallels = np.arctan2(cosine_theta, (1 - f) ** 2 * sine_theta)
        areas = np.tile(dlon * weights[:, np.newaxis], (1, meridians.size))
        super(GaussGrid, self).__init__(meridians, parallels, areas, a, f)

    def copy(self):
        grid = GaussGrid(self.parallels.size, self.semimajor_axis, self.flattening)
        if self.value_array is not None:
            grid.values = self.values.copy()
        grid.epoch = self.epoch
        return grid


# -------------------------------------------------------------------------------------------------------
# latitude mappings between the ellipsoid and the sphere (grates/grid.py:2047-2110)
# -------------------------------------------------------------------------------------------------------

def _sine_series(beta, e2, table):
    """beta + sum_j (sum_k c_jk e2^k) sin(2 j beta) for the rows (j, {k: c_jk}) of `table`"""
    result = np.array(beta, dtype=float, copy=True)
    for harmonic, coefficients in table:
        result = result + sum(c * e2 ** k for k, c in coefficients.items()) * np.sin(2 * harmonic * beta)
    return result


def _authalic_q(sin_latitude, e):
    return (1 - e ** 2) * sin_latitude / (1 - e ** 2 * sin_latitude ** 2) - (1 - e ** 2) / (2 * e) * np.log((1 - e * sin_latitude) / (1 + e * sin_latitude))


def authalic_radius(a=6378137.0, f=298.2572221010**-1):
    """radius of the sphere with the surface area of the ellipsoid"""
    return a * np.sqrt(_authalic_q(1.0, np.sqrt(f * (2 - f))) * 0.5)


def geodetic2authalic(latitude, f=298.2572221010**-1):
    if f == 0.0:
        return latitude
    e = np.sqrt(f * (2 - f))
    return np.arcsin(_authalic_q(np.sin(latitude), e) / _authalic_q(1.0, e))


# series in e^2 as the reference evaluates them (including its denominators 181400 and 997920)
_AUTHALIC_TO_GEODETIC = ((1, {1: 1 / 3, 2: 31 / 180, 3: 517 / 5040, 4: 120389 / 181400, 5: 1362254 / 29937600}),
                         (2, {2: 23 / 360, 3: 251 / 3780, 4: 102287 / 1814400, 5: 450739 / 997920}),
                         (3, {3: 761 / 45360, 4: 47561 / 1814400, 5: 434501 / 14968800}),
                         (4, {4: 6059 / 1209600, 5: 625511 / 59875200}),
                         (5, {5: 48017 / 29937600}))
_CONFORMAL_TO_GEODETIC = ((1, {1: 1 / 2, 2: 5 / 24, 3: 1 / 12, 4: 13 / 360}),
                          (2, {2: 7 / 48, 3: 29 / 240, 4: 811 / 11520}),
                          (3, {3: 7 / 120, 4: 81 / 1120}),
                          (4, {4: 4279 / 161280}))


def authalic2geodetic(beta, f=298.2572221010**-1):
    return _sine_series(beta, f * (2 - f), _AUTHALIC_TO_GEODETIC)


def conformal2geodetic(beta, f=298.2572221010**-1):
    return _sine_series(beta, f * (2 - f), _CONFORMAL_TO_GEODETIC)


def geodetic2conformal(latitude, f=298.2572221010**-1):
    e = np.sqrt(f * (2 - f))
    sin_latitude = np.sin(latitude)
    return 2 * np.arctan2(np.sqrt((1 + sin_latitude) * (1 - e * sin_latitude) ** e), np.sqrt((1 - sin_latitude) * (1 + e * sin_latitude) ** e)) - np.pi * 0.5


def geocentric2geodetic(beta, f=298.2572221010**-1):
    return np.arctan2(np.sin(beta), np.cos(beta) * (1 - f) ** 2)


def geodetic2geocentric(latitude, f=298.2572221010**-1):
    return np.arctan2((1 - f) ** 2 * np.sin(latitude), np.cos(latitude))


# -------------------------------------------------------------------------------------------------------
# coordinate helpers (grates/grid.py:1893-2044)
# -------------------------------------------------------------------------------------------------------

def spherical_distance(lon1, lat1, lon2, lat2, r=6378136.3):
    """Great-circle distance on a sphere of radius r (Vincenty form of the arc)."""
    dlon = lon2 - lon1
    y = np.sqrt((np.cos(lat2) * np.sin(dlon)) ** 2 + (np.cos(lat1) * np.sin(lat2) - np.sin(lat1) * np.cos(lat2) * np.cos(dlon)) ** 2)
    x = np.sin(lat1) * np.sin(lat2) + np.cos(lat1) * np.cos(lat2) * np.cos(dlon)
    return np.arctan2(y, x) * r


def spherical2cartesian(r, colatitude, lon):
    xyz = np.empty((np.asarray(lon).size, 3))
    xyz[:, 0] = r * np.sin(colatitude) * np.cos(lon)
    xyz[:, 1] = r * np.sin(colatitude) * np.sin(lon)
    xyz[:, 2] = r * np.cos(colatitude)
    return xyz


def cartesian2spherical(xyz):
    """(r, colatitude, longitude) of cartesian triples (m, 3)."""
    r = np.sqrt(np.sum(xyz ** 2, axis=1))
    colat = np.arctan2(np.sqrt(np.sum(xyz[:, 0:2] ** 2, axis=1)), xyz[:, 2])
    lon = np.arctan2(xyz[:, 1], xyz[:, 0])
    return r, colat, lon


def geodetic2cartesian(lon, lat, h=0, a=6378137.0, f=298.2572221010**-1):
    """Cartesian coordinates (m, 3) from geodetic longitude, latitude [rad] and ellipsoidal height [m]."""
    if f == 0.0:
        return spherical2cartesian(a + h, np.pi * 0.5 - lat, lon)
    e2 = 2 * f - f ** 2
    nu = a / np.sqrt(1 - e2 * np.sin(lat) ** 2)
    return np.vstack(((nu + h) * np.cos(lat) * np.cos(lon), (nu + h) * np.cos(lat) * np.sin(lon), ((1 - e2) * nu + h) * np.sin(lat))).T


def cartesian2geodetic(xyz, a=6378137.0, f=298.2572221010**-1, max_iter=10, threshold=1e-6):
    """Geodetic longitude, latitude [rad] and height [m] by fixed-point iteration of Bowring's equation
    (grates/grid.py:1984-2008)."""
    if f == 0.0:
        r, colat, lon = cartesian2spherical(xyz)
        return lon, np.pi * 0.5 - colat, r - a
    e2 = 2 * f - f ** 2
    z = xyz[:, -1]
    p2 = xyz[:, 0] ** 2 + xyz[:, 1] ** 2
    h_previous = 0
    k = (1 - e2) ** -1
    for _ in range(max_iter):
        c = np.power(p2 + (1 - e2) * z ** 2 * k ** 2, 1.5) / (a * e2)
        k = 1 + (p2 + (1 - e2) * z ** 2 * k ** 3) / (c - p2)
        h = (k ** -1 - (1 - e2)) * np.sqrt(p2 + z ** 2 * k ** 2) / e2
        if np.max(np.abs(h - h_previous)) < threshold:
            break
        h_previous = h
    return np.arctan2(xyz[:, 1], xyz[:, 0]), np.arctan2(k * z, np.sqrt(p2)), h


def __getattr__(name):
    """`grid.ReuterGrid` resolves like in the reference; the class lives in grates_amd.extras (outside the hot path's scope)."""
    if name == 'ReuterGrid':
        from . import extras
        return extras.ReuterGrid
    raise AttributeError('module {0!r} has no attribute {1!r}'.format(__name__, name))
