"""
Harmonic kernels with the interface of ``grates.kernel`` (grates/kernel.py:17-654).

Isotropic kernels only produce the small per-parallel degree-factor table kn[nlat, N+1] that is uploaded into a
device plan; they are evaluated on the host.  ``AnisotropicKernel`` evaluates y(source)^T K y(point) on the GPU
(dense product + synthesis).  ``modulation_transfer`` and ``spatial_resolution`` (diagnostics driven by
scipy.optimize) are out of scope, see DESIGN.md.
"""

import abc

import numpy as np

from . import data, engine, utilities

_GRAVITATIONAL_CONSTANT = 6.673e-11       # value used by the reference, grates/kernel.py:405


def get_kernel(kernel_name):
    """
    Return the kernel registered under `kernel_name` (case-insensitive), e.g. 'ewh', 'potential', 'geoid'.
    Raises ValueError for unknown names (grates/kernel.py:17-67).
    """
    name = kernel_name.lower()
    for aliases, cls in _REGISTRY:
        if name in aliases:
            return cls()
    raise ValueError("Unrecognized kernel '{0:s}'.".format(kernel_name))


class IsotropicKernel(metaclass=abc.ABCMeta):
    """
    Band-limited isotropic kernel.  Kernel coefficients convert the quantity (e.g. water height) into
    potential, inverse coefficients convert potential into the quantity.  Subclasses implement
    `_coefficients(min_degree, max_degree, r, colat)` returning an array (points, degrees).
    """

    @abc.abstractmethod
    def _coefficients(self, min_degree, max_degree, r, colat):
        pass

    def coefficients(self, min_degree, max_degree, r=6378136.3, colat=0):
        """Kernel coefficients (m, max_degree + 1 - min_degree) for radius / colatitude of m points.
        Scalars broadcast against arrays; arrays must have equal shapes (grates/kernel.py:110-124)."""
        kinds = tuple('scalar' if np.isscalar(v) else ('array' if isinstance(v, np.ndarray) else None) for v in (r, colat))
        if None in kinds:
            raise ValueError('input must be either numeric scalar or ndarrays of matching or broadcastable dimensions')
        if kinds == ('array', 'array') and r.shape != colat.shape:
            raise ValueError('shape mismatch in radius and colatitude: objects cannot be broadcast to a single shape')
        # a scalar next to an array is expanded to the array's shape; two scalars stay scalars
        radius = np.full(colat.shape, r) if kinds == ('scalar', 'array') else r
        colatitude = np.full(r.shape, colat) if kinds == ('array', 'scalar') else colat
        return self._coefficients(min_degree, max_degree, radius, colatitude)

    def coefficient(self, n, r=6378136.3, colat=0):
        """Kernel coefficient of degree n for all points, shape (m,)."""
        return self.coefficients(n, n, r, colat).squeeze(axis=1)

    def inverse_coefficient(self, n, r=6378136.3, colat=0):
        """1 / k_n, or zeros if k_n vanishes at every point (grates/kernel.py:144-145)."""
        kn = self.coefficient(n, r, colat)
        return np.zeros(kn.shape) if np.allclose(kn, 0.0) else 1.0 / kn

    def inverse_coefficients(self, min_degree, max_degree, r=6378136.3, colat=0):
        """Inverse kernel coefficients (m, max_degree + 1 - min_degree); a degree whose coefficients are
        all ~0 maps to zeros (grates/kernel.py:187-188)."""
        kn = self.coefficients(min_degree, max_degree, r, colat)
        out = np.empty(kn.shape)
        for k in range(kn.shape[1]):
            out[:, k] = 0.0 if np.allclose(kn[:, k], 0.0) else 1.0 / kn[:, k]
        return out

    def _as_array(self, table, min_degree, max_degree):
        arr = np.zeros((table.shape[0], max_degree + 1, max_degree + 1))
        for n in range(min_degree, max_degree + 1):
            arr[:, n, 0:n + 1] = table[:, n - min_degree, np.newaxis]
            arr[:, 0:n, n] = table[:, n - min_degree, np.newaxis]
        return arr

    def coefficient_array(self, min_degree, max_degree, r=6378136.3, colat=0):
        """Kernel coefficients arranged like a coefficient array (m, max_degree+1, max_degree+1)."""
        return self._as_array(self.coefficients(min_degree, max_degree, r, colat), min_degree, max_degree)

    def inverse_coefficient_array(self, min_degree, max_degree, r=6378136.3, colat=0):
        """Inverse kernel coefficients arranged like a coefficient array."""
        return self._as_array(self.inverse_coefficients(min_degree, max_degree, r, colat), min_degree, max_degree)

    def evaluate(self, min_degree, max_degree, psi, r=6378136.3, colat=0):
        """Kernel in space domain at spherical distance psi [rad] (grates/kernel.py:272-275)."""
        kn = np.zeros(max_degree + 1)
        kn[min_degree:] = self.coefficients(min_degree, max_degree, r, colat)[0, :] * np.sqrt(2 * np.arange(min_degree, max_degree + 1) + 1)
        return utilities.legendre_summation(kn, psi)

    def evaluate_grid(self, min_degree, max_degree, source_longitude, source_latitude, eval_longitude, eval_latitude, r=6378136.3, colat=0):
        """Kernel centred at a source point, evaluated on meridians x parallels (grates/kernel.py:305-308)."""
        from . import grid as _grid
        lon, lat = np.meshgrid(eval_longitude, eval_latitude)
        psi = _grid.spherical_distance(source_longitude, source_latitude, lon, lat, r=1)
        return self.evaluate(min_degree, max_degree, psi, r, colat)


def _degrees(min_degree, max_degree):
    return np.arange(min_degree, max_degree + 1, dtype=float)


def _normal_gravity(r, colat):
    from . import gravityfield
    return gravityfield.GRS80.normal_gravity(r, colat)


class WaterHeight(IsotropicKernel):
    """Equivalent water height [m]: k_n = 4 pi G rho (1 + k'_n) / (2n + 1) r (grates/kernel.py:398-406)."""

    def __init__(self, rho=1025):
        self.__rho = rho
        self.__love_numbers, _, _ = data.load_love_numbers()

    def _coefficients(self, min_degree, max_degree, r=6378136.3, colat=0):
        kn = (4 * np.pi * _GRAVITATIONAL_CONSTANT * self.__rho) * (1 + self.__love_numbers[min_degree:max_degree + 1]) / (2 * _degrees(min_degree, max_degree) + 1)
        return (kn[:, np.newaxis] * r).T


class OceanBottomPressure(IsotropicKernel):
    """Ocean bottom pressure [Pa] (grates/kernel.py:414-421)."""

    def __init__(self):
        self.__love_numbers, _, _ = data.load_love_numbers()

    def _coefficients(self, min_degree, max_degree, r=6378136.3, colat=0):
        kn = (4 * np.pi * _GRAVITATIONAL_CONSTANT) * (1 + self.__love_numbers[min_degree:max_degree + 1]) / (2 * _degrees(min_degree, max_degree) + 1)
        return (kn[:, np.newaxis] * (r / _normal_gravity(r, colat))).T


class SurfaceDensity(IsotropicKernel):
    """Surface density [kg/m^2] (grates/kernel.py:428-435)."""

    def __init__(self):
        self.__love_numbers, _, _ = data.load_love_numbers()

    def _coefficients(self, min_degree, max_degree, r=6378136.3, colat=0):
        kn = (4 * np.pi * _GRAVITATIONAL_CONSTANT) * (1 + self.__love_numbers[min_degree:max_degree + 1]) / (2 * _degrees(min_degree, max_degree) + 1)
        return (kn[:, np.newaxis] * r).T


class Potential(IsotropicKernel):
    """Poisson kernel (disturbing potential): all ones (grates/kernel.py:445-449)."""

    def _coefficients(self, min_degree, max_degree, r=6378136.3, colat=0):
        count = max(np.asarray(r).size, np.asarray(colat).size)
        return np.ones((count, max_degree + 1 - min_degree))


class GravityAnomaly(IsotropicKernel):
    """Gravity anomaly: k_n = r / (n - 1), zero for n = 1 (grates/kernel.py:458-461)."""

    def _coefficients(self, min_degree, max_degree, r=6378136.3, colat=0):
        n = _degrees(min_degree, max_degree)
        kn = np.array([1 / (v - 1) if v != 1 else 0.0 for v in n])
        return (kn[:, np.newaxis] * r).T


class Gauss(IsotropicKernel):
    """
    Gaussian averaging kernel (Jekeli recursion).  Weights are set to zero after the first one that drops
    below 1e-7; a table of 1025 degrees is built up front and extended on demand (grates/kernel.py:468-506,
    including the slightly different Earth radius the reference uses in the extension).
    """

    def __init__(self, radius):
        if radius < 0:
            raise ValueError('Gaussian filter radius must be positive (got {0:f})'.format(radius))
        self.__radius = radius
        table_degree = 1024
        if radius > 0:
            self.__wn = np.zeros(table_degree + 1)
            self.__wn[0] = 1.0
            self.__recurse(1, table_degree, np.log(2.0) / (1 - np.cos(radius / 6378.1366)))
        else:
            self.__wn = np.ones(table_degree + 1)

    def __recurse(self, first, last, b):
        wn = self.__wn
        for n in range(first, last + 1):
            if n == 1:
                wn[1] = (1 + np.exp(-2 * b)) / (1 - np.exp(-2 * b)) - 1 / b
                continue
            wn[n] = -(2 * n - 1) / b * wn[n - 1] + wn[n - 2]
            if wn[n] < 1e-7:
                break

    def _coefficients(self, min_degree, max_degree, r=6378136.3, colat=0):
        have = self.__wn.size - 1
        if max_degree > have:
            if self.__radius > 0:
                old = self.__wn
                self.__wn = np.empty(max_degree + 1)       # the reference leaves the tail uninitialised
                self.__wn[have + 1:] = 0.0                 # past a break; zeros are the defined behaviour here
                self.__wn[0:have + 1] = old
                self.__recurse(have + 1, max_degree, np.log(2.0) / (1 - np.cos(self.__radius / 6378.1363)))
            else:
                self.__wn = np.ones(max_degree + 1)
        count = max(np.asarray(r).size, np.asarray(colat).size)
        return np.tile(self.__wn[min_degree:max_degree + 1], (count, 1))


class GeoidHeight(IsotropicKernel):
    """Geoid height: potential divided by GRS80 normal gravity (grates/kernel.py:516-518)."""

    def _coefficients(self, min_degree, max_degree, r=6378136.3, colat=0):
        return np.tile(_normal_gravity(r, colat)[:, np.newaxis], (1, max_degree + 1 - min_degree))


class UpwardContinuation(IsotropicKernel):
    """Upward continuation (R/r)^(n+1) of another kernel (grates/kernel.py:532-539; the upstream version
    calls the kernel object and raises -- here the wrapped kernel's coefficients are used)."""

    def __init__(self, R=6.3781363000e+06, kernel='potential'):
        self.__kernel = get_kernel(kernel)
        self.__R = R

    def _coefficients(self, min_degree, max_degree, r=6378136.3, colat=0):
        factor = np.power(np.atleast_1d(self.__R / r)[:, np.newaxis], np.arange(min_degree, max_degree + 1, dtype=int) + 1)
        return factor * self.__kernel.coefficients(min_degree, max_degree, r, colat)


class VerticalDeformation(IsotropicKernel):
    """Elastic vertical deformation: gamma / (h'_n / (1 + k'_n)) (grates/kernel.py:551-559)."""

    def __init__(self, frame='CE'):
        k, h, _ = data.load_love_numbers(frame=frame)
        with np.errstate(divide='ignore', invalid='ignore'):
            self.__kn = h / (1 + k)

    def _coefficients(self, min_degree, max_degree, r=6378136.3, colat=0):
        with np.errstate(divide='ignore', invalid='ignore'):
            return _normal_gravity(r, colat)[:, np.newaxis] / self.__kn[min_degree:max_degree + 1]


class Uplift(IsotropicKernel):
    """Approximate uplift after Wahr et al. (2000): 2 gamma / (2n + 1) (grates/kernel.py:572-574)."""

    def _coefficients(self, min_degree, max_degree, r=6378136.3, colat=0):
        return 2 * _normal_gravity(r, colat)[:, np.newaxis] / (2 * _degrees(min_degree, max_degree) + 1)


_REGISTRY = (
    (('ewh', 'water_height'), WaterHeight),
    (('obp', 'ocean_bottom_pressure'), OceanBottomPressure),
    (('potential',), Potential),
    (('geoid', 'geoid_height'), GeoidHeight),
    (('surface_density',), SurfaceDensity),
    (('anomaly', 'gravity_anomaly'), GravityAnomaly),
    (('deformation', 'vertical_derformation'), VerticalDeformation),
    (('uplift',), Uplift),
)


class AnisotropicKernel:
    """
    Possibly anisotropic kernel in the space domain, given by its spherical harmonic mapping K (degree-wise order,
    degrees min_degree..max_degree): kernel(source, point) = y(source)^T K y(point) with the spherical harmonics y
    (grates/kernel.py:589-654).  The row vector y(source)^T K is one dense product on the fp64 MFMA GEMM; the
    evaluation is a spherical harmonic synthesis of that vector (point list or regular grid) with unit degree factors.
    """

    def __init__(self, K, min_degree, max_degree):
        K = np.asarray(K, dtype=float)
        count = (max_degree + 1) ** 2 - min_degree ** 2
        if K.ndim != 2 or K.shape != (count, count):
            raise ValueError('kernel matrix must be square with one row per coefficient of degrees {0:d} to {1:d} (got {2})'.format(min_degree, max_degree, str(K.shape)))
        self.__matrix = K.copy()
        self.__min_degree = min_degree
        self.__max_degree = max_degree
        self.__device_matrix = None

    @property
    def matrix(self):
        return self.__matrix.copy()

    def _source_coefficients(self, source_longitude, source_latitude):
        """y(source)^T K as a coefficient array [1, nmax+1, nmax+1] on the device."""
        N = self.__max_degree
        if self.__device_matrix is None:
            self.__device_matrix = engine.to_device(self.__matrix)
        Y = engine.legendre_functions(N, np.atleast_1d(0.5 * np.pi - np.asarray(source_latitude, dtype=float)).ravel()[0:1])
        Y = Y * engine.trigonometric_functions(N, np.atleast_1d(np.asarray(source_longitude, dtype=float)).ravel()[0:1])
        row = engine.gemm(engine.ravel(Y, self.__min_degree, N), self.__device_matrix)
        return engine.unravel(row, self.__min_degree, N)

    def evaluate(self, source_longitude, source_latitude, eval_longitude, eval_latitude):
        """Kernel centred at the source point, at a list of evaluation points [rad] (grates/kernel.py:615-620)."""
        lon = np.atleast_1d(np.asarray(eval_longitude, dtype=float)).ravel()
        lat = np.atleast_1d(np.asarray(eval_latitude, dtype=float)).ravel()
        if lon.size != lat.size:
            raise ValueError('evaluation longitudes and latitudes differ in size ({0:d} vs {1:d})'.format(lon.size, lat.size))
        anm = self._source_coefficients(source_longitude, source_latitude)
        ones = np.ones((lon.size, self.__max_degree + 1))
        return engine.to_host(engine.synthesis_points(self.__max_degree, 0.5 * np.pi - lat, lon, ones, anm))[0]

    def evaluate_grid(self, source_longitude, source_latitude, eval_longitude, eval_latitude):
        """Kernel centred at the source point on meridians x parallels [rad]; returns (parallels, meridians) like the
        reference (grates/kernel.py:642-654).  One batched synthesis on the regular-grid path."""
        lon = np.atleast_1d(np.asarray(eval_longitude, dtype=float)).ravel()
        lat = np.atleast_1d(np.asarray(eval_latitude, dtype=float)).ravel()
        anm = self._source_coefficients(source_longitude, source_latitude)
        plan = engine.Plan(self.__max_degree, 0.5 * np.pi - lat, np.ones((lat.size, self.__max_degree + 1)), lon)
        return engine.to_host(plan.synthesis(anm))[0]
