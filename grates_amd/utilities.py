"""
Auxiliary functions with the call signatures of ``grates.utilities`` (grates/utilities.py:13-459).

The table functions (Legendre functions, cos/sin tables, spherical harmonics) run on the GPU through
libshg and return NumPy arrays like the reference does (pass ``as_tensor=True`` to keep the result on the
device).  Index maps and ellipsoid geometry are tiny host-side helpers.
"""

import abc
import datetime

import numpy as np

from . import engine


def _result(tensor, as_tensor):
    return tensor if as_tensor else engine.to_host(tensor)


def legendre_functions(max_degree, colat, as_tensor=False):
    """
    Fully normalised associated Legendre functions.  Pnm[:, n, m] holds degree n / order m for all points,
    mirrored into Pnm[:, m-1, n] for m > 0 (grates/utilities.py:13-59).  shape (k, nmax+1, nmax+1)
    """
    return _result(engine.legendre_functions(int(max_degree), np.atleast_1d(colat)), as_tensor)


def legendre_functions_per_order(max_degree, order, colat, as_tensor=False):
    """Legendre functions of one order, degrees order..max_degree (grates/utilities.py:62-115)."""
    if order > max_degree:
        raise ValueError('order exceeds maximum degree ({0:d} vs. {1:d})'.format(order, max_degree))
    return _result(engine.legendre_functions_per_order(int(max_degree), int(order), np.atleast_1d(colat)), as_tensor)


def legendre_polynomials(max_degree, colat, derivative=None):
    """
    Fully normalised Legendre polynomials or their 1st / 2nd derivative with respect to t = cos(colat)
    (grates/utilities.py:118-182).  Small host-side helper (used by kernel evaluation in space domain).
    """
    t = np.cos(np.atleast_1d(colat))
    out = np.empty((t.size, max_degree + 1))
    if derivative not in (None, 1, 2):
        raise ValueError('Derivative must be None, 1 or 2. Got {} instead.'.format(derivative))
    d = 0 if derivative is None else derivative
    # seeds: value / first / second derivative of P0, P1(, P2)
    seeds = {0: (1.0, np.sqrt(3) * t), 1: (0.0, np.sqrt(3)), 2: (0.0, 0.0)}[d]
    out[:, 0] = seeds[0]
    if max_degree == 0:
        return out
    out[:, 1] = seeds[1]
    first = 2
    if d == 2:
        if max_degree == 1:
            return out
        out[:, 2] = 3 * np.sqrt(5)
        first = 3
    for n in range(first, max_degree + 1):
        den, num = {0: (n, n - 1.0), 1: (n - 1.0, n), 2: (n - 2.0, n + 1.0)}[d]
        out[:, n] = np.sqrt((2.0 * n - 1.0) * (2.0 * n + 1.0)) / den * t * out[:, n - 1] - \
            np.sqrt((2.0 * n + 1.0) / (2.0 * n - 3.0)) * num / den * out[:, n - 2]
    return out


def legendre_summation(coefficients, colat, derivative=None):
    """Clenshaw summation of a Legendre series (grates/utilities.py:185-246)."""
    t = np.cos(np.atleast_1d(colat))
    b1 = b2 = 0
    count = coefficients.size
    if derivative is None:
        for k in range(count - 1, 0, -1):
            alpha = np.sqrt((2 * k + 1) * (2 * k + 3)) / (k + 1)
            beta = -np.sqrt((2 * k + 5) / (2 * k + 1)) * (k + 1) / (k + 2)
            b1, b2 = coefficients[k] + alpha * t * b1 + beta * b2, b1
        return coefficients[0] + np.sqrt(3) * t * b1 - 0.5 * np.sqrt(5) * b2
    if derivative == 1:
        for k in range(count - 1, 0, -1):
            alpha = np.sqrt(2 * k + 3) * np.sqrt(2 * k + 1) / k
            beta = -np.sqrt((2 * k + 5) / (2 * k + 1)) * (k + 2) / (k + 1)
            b1, b2 = coefficients[k] + alpha * t * b1 + beta * b2, b1
        return np.sqrt(3) * b1
    if derivative == 2:
        for k in range(count - 1, 1, -1):
            alpha = np.sqrt(2 * k + 3) * np.sqrt(2 * k + 1) / (k - 1)
            beta = -np.sqrt((2 * k + 5) / (2 * k + 1)) * (k + 3) / k
            b1, b2 = coefficients[k] + alpha * t * b1 + beta * b2, b1
        return 3 * np.sqrt(5) * b1
    raise ValueError('Derivative must be None, 1 or 2. Got {} instead.'.format(derivative))


def trigonometric_functions(max_degree, lon, as_tensor=False):
    """cs[:, n, m] = cos(m lon), cs[:, m-1, n] = sin(m lon) (grates/utilities.py:249-275)."""
    return _result(engine.trigonometric_functions(int(max_degree), np.atleast_1d(lon)), as_tensor)


def spherical_harmonics(max_degree, colat, lon, as_tensor=False):
    """Ynm = trigonometric_functions * legendre_functions (grates/utilities.py:278-307)."""
    colat, lon = np.atleast_1d(colat), np.atleast_1d(lon)
    count = max(colat.size, lon.size)
    colat = np.broadcast_to(colat, (count,))
    lon = np.broadcast_to(lon, (count,))
    Y = engine.trigonometric_functions(int(max_degree), lon)
    Y *= engine.legendre_functions(int(max_degree), colat)
    return _result(Y, as_tensor)


# ---------------------------------------------------------------------------------------------------
# degree-wise index maps (host side for ndarrays; engine.ravel / engine.unravel for device batches)
# ---------------------------------------------------------------------------------------------------

def degreewise_array_index(min_degree, max_degree):
    """
    Rows / columns inside the packed (N+1, N+1) coefficient array of every entry of the degree-wise vector
    C00, C10, C11, S11, C20, ...: C_nm sits at [n, m], S_nm at [m-1, n] (grates/utilities.py:336-343).
    """
    rows, cols = [], []
    for n in range(min_degree, max_degree + 1):
        m = np.arange(1, n + 1)
        r = np.empty(2 * n + 1, dtype=np.int64)
        c = np.empty(2 * n + 1, dtype=np.int64)
        r[0], c[0] = n, 0
        r[1::2], c[1::2] = n, m
        r[2::2], c[2::2] = m - 1, n
        rows.append(r)
        cols.append(c)
    if not rows:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    return np.concatenate(rows), np.concatenate(cols)


def ravel_coefficients(array, min_degree=0, max_degree=None):
    """
    Ravel a (m, m) or (k, m, m) coefficient array into degree-wise vector(s) of length
    (max_degree + 1)**2 - min_degree**2; degrees the array does not hold stay zero
    (grates/utilities.py:310-360).
    """
    if max_degree is None:
        max_degree = array.shape[-1] - 1
    if array.ndim not in (2, 3):
        raise ValueError('Only 2d or 3d spherical harmonic arrays can be raveled.')
    count = (max_degree + 1) * (max_degree + 1) - min_degree * min_degree
    out = np.zeros(array.shape[:-2] + (count,), dtype=array.dtype)
    top = min(array.shape[-1] - 1, max_degree)
    if top >= min_degree:
        rows, cols = degreewise_array_index(min_degree, top)
        out[..., 0:rows.size] = array[..., rows, cols]
    return out


def unravel_coefficients(vector, min_degree=0, max_degree=None):
    """Inverse of ravel_coefficients for 1d / 2d input (grates/utilities.py:363-411)."""
    if max_degree is None:
        max_degree = int(np.sqrt(vector.shape[-1] + min_degree * min_degree) - 1)
    if vector.ndim not in (1, 2):
        raise ValueError('Only 1d or 2d spherical harmonic vectors can be unraveled.')
    out = np.zeros(vector.shape[:-1] + (max_degree + 1, max_degree + 1), dtype=vector.dtype)
    rows, cols = degreewise_array_index(min_degree, max_degree)
    out[..., rows, cols] = vector[..., 0:rows.size]
    return out


# ---------------------------------------------------------------------------------------------------
# ellipsoid geometry (grates/utilities.py:414-459)
# ---------------------------------------------------------------------------------------------------

def geocentric_radius(latitude, a=6378137.0, f=298.2572221010**-1):
    """Geocentric radius [m] of points on the ellipsoid surface at geodetic `latitude` [rad]."""
    e2 = f * (2 - f)
    nu = a / np.sqrt(1 - e2 * np.sin(latitude) ** 2)
    return nu * np.sqrt(np.cos(latitude) ** 2 + (1 - e2) ** 2 * np.sin(latitude) ** 2)


def colatitude(latitude, a=6378137.0, f=298.2572221010**-1):
    """Geocentric colatitude [rad] of points on the ellipsoid surface at geodetic `latitude` [rad]."""
    e2 = f * (2 - f)
    nu = a / np.sqrt(1 - e2 * np.sin(latitude) ** 2)
    return np.arccos(nu * (1 - e2) * np.sin(latitude) / geocentric_radius(latitude, a, f))


def kaula_array(min_degree, max_degree, kaula_factor=1e-10, kaula_power=4.0):
    """Kaula-type curve f / n**p as coefficient array, zero below min_degree (grates/utilities.py:560-585)."""
    out = np.zeros((max_degree + 1, max_degree + 1))
    for n in range(min_degree, max_degree + 1):
        value = kaula_factor * np.power(float(n), -float(kaula_power))
        out[n, 0:n + 1] = value
        out[0:n, n] = value
    return out


# ---- temporal basis functions (grates/utilities.py:462-557): design matrices of TimeSeries.detrend -----------------------

def _mjd(epoch):
    """modified Julian date of a datetime with the reference's arithmetic: whole days + seconds / 86400 (microseconds
    dropped; grates/time.py:37-38)"""
    delta = epoch - datetime.datetime(1858, 11, 17)
    return delta.days + delta.seconds / 86400.0


class TemporalBasisFunction(metaclass=abc.ABCMeta):
    """A parametric model of time; `design_matrix(epochs)` returns its [len(epochs), parameter count] design matrix."""

    def __init__(self, reference_epoch):
        self._reference_epoch = reference_epoch

    def _days(self, epochs):
        t = np.array([_mjd(e) for e in epochs])
        if self._reference_epoch is not None:
            t -= _mjd(self._reference_epoch)
        return t

    @abc.abstractmethod
    def design_matrix(self, epochs):
        pass


class Oscillation(TemporalBasisFunction):
    """a cos(2 pi (t - t0) / T) + b sin(2 pi (t - t0) / T), period T in days (without a reference epoch t is the MJD)."""

    def __init__(self, period, reference_epoch=None):
        super().__init__(reference_epoch)
        self.__period = period

    def design_matrix(self, epochs):
        phase = 2 * np.pi / self.__period * self._days(epochs)
        return np.column_stack((np.cos(phase), np.sin(phase)))


class Polynomial(TemporalBasisFunction):
    """sum_k a_k (t - t0)^k for k = 0 .. degree, t in days."""

    def __init__(self, degree, reference_epoch=None):
        super().__init__(reference_epoch)
        self.__degree = degree

    def design_matrix(self, epochs):
        t = self._days(epochs)
        return np.column_stack([np.ones(t.size)] + [t ** k for k in range(1, self.__degree + 1)])
