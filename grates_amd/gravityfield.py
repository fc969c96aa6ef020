"""
Gravity field containers with the interface of ``grates.gravityfield``:
``PotentialCoefficients`` (grates/gravityfield.py:76-481), ``TimeSeries`` (:815-1052), coefficient
sequences (:1175-1471) and the reference fields GRS80 / WGS84 (:1474-1574).

``PotentialCoefficients.to_grid`` and ``TimeSeries.to_grid`` run on the GPU through libshg (batched
Legendre + longitude stage); containers keep ``anm`` as NumPy arrays exactly like the reference, the
batched entry points also accept / return device tensors.
"""

import numpy as np

from . import engine, kernel as _kernel, utilities


def degree_indices(n, max_order=None):
    """Row / column indices of all coefficients of degree n (cosines by increasing order, then sines),
    optionally limited to orders <= max_order (grates/gravityfield.py:15-40)."""
    count = n if max_order is None else min(n, max_order)
    rows = np.concatenate((np.full(count + 1, n, dtype=int), np.arange(count, dtype=int)))
    columns = np.concatenate((np.arange(count + 1, dtype=int), np.full(count, n, dtype=int)))
    return rows, columns


def order_indices(max_degree, m):
    """Row / column indices of all coefficients of order m (cosines by increasing degree, then sines)
    (grates/gravityfield.py:43-73)."""
    degrees = np.arange(m, max_degree + 1, dtype=int)
    if m == 0:                                                 # zonal: cosine terms only, C_n0 at [n, 0]
        return degrees, np.zeros(degrees.size, dtype=int)
    # C_nm at [n, m] followed by S_nm at [m - 1, n]
    return np.concatenate((degrees, np.full(degrees.size, m - 1))), np.concatenate((np.full(degrees.size, m), degrees))


def _degree_array(max_degree):
    idx = np.arange(max_degree + 1)
    return np.maximum(idx[:, np.newaxis], idx[np.newaxis, :])


def _order_array(max_degree):
    idx = np.arange(max_degree + 1)
    rows, cols = idx[:, np.newaxis], idx[np.newaxis, :]
    return np.where(cols <= rows, cols, rows + 1)


def surface_factors(grid_kernel, max_degree, latitude, GM, R, a, f):
    """
    colatitude, radius and kn[i, n] = (1 / k_n(r_i, theta_i)) (R / r_i)^(n+1) GM / R for points on the
    ellipsoid surface at geodetic `latitude` (grates/gravityfield.py:353-356).
    """
    colat = utilities.colatitude(latitude, a, f)
    radius = utilities.geocentric_radius(latitude, a, f)
    kn = grid_kernel.inverse_coefficients(0, max_degree, radius, colat) * \
        np.power((R / radius)[:, np.newaxis], np.arange(max_degree + 1, dtype=int) + 1) * GM / R
    return colat, radius, kn


def _check_operand(left, right, accepted, symbol):
    """The reference's operand check of its arithmetic operators: a TypeError with Python's own wording."""
    if not isinstance(right, accepted):
        raise TypeError("unsupported operand type(s) for {0}: '{1}' and '{2}'".format(symbol, str(type(left)), str(type(right))))


class PotentialCoefficients:
    """
    A set of potential coefficients: ``anm[n, m]`` = C_nm, ``anm[m-1, n]`` = S_nm.

    Parameters
    ----------
    GM : float
        geocentric gravitational constant
    R : float
        reference radius
    max_degree : int
        pre-allocate the coefficient array up to max_degree
    """

    def __init__(self, GM=3.9860044150e+14, R=6.3781363000e+06, max_degree=None):
        self.GM = GM
        self.R = R
        count = 0 if max_degree is None else max_degree + 1
        self.anm = np.zeros((count, count))
        self.epoch = None

    def copy(self):
        """Deep copy."""
        other = PotentialCoefficients(self.GM, self.R)
        other.anm = self.anm.copy()
        other.epoch = self.epoch
        return other

    @property
    def max_degree(self):
        return self.anm.shape[0] - 1

    def slice(self, min_degree=None, max_degree=None, min_order=None, max_order=None, step_degree=1, step_order=1):
        """New instance with everything outside the degree / order ranges set to zero, truncated to max_degree
        (grates/gravityfield.py:132-147)."""
        min_degree = 0 if min_degree is None else min_degree
        max_degree = self.max_degree if max_degree is None else max_degree
        min_order = 0 if min_order is None else min_order
        max_order = max_degree if max_order is None else max_order
        keep = np.logical_and(np.isin(_degree_array(self.max_degree), range(min_degree, max_degree + 1, step_degree)),
                              np.isin(_order_array(self.max_degree), range(min_order, max_order + 1, step_order)))
        other = PotentialCoefficients(self.GM, self.R)
        other.anm = np.where(keep, self.anm, 0.0)
        other.epoch = self.epoch
        other.truncate(max_degree)
        return other

    def append(self, trigonometric_function, degree, order, value):
        """Set a single coefficient, growing the array if needed (grates/gravityfield.py:149-159)."""
        if degree > self.max_degree:
            grown = np.zeros((degree + 1, degree + 1))
            grown[0:self.anm.shape[0], 0:self.anm.shape[1]] = self.anm
            self.anm = grown
        if trigonometric_function in ('c', 'cos', 'cosine'):
            self.anm[degree, order] = value
        elif trigonometric_function in ('s', 'sin', 'sine') and order > 0:
            self.anm[order - 1, degree] = value

    def truncate(self, max_degree):
        """Drop all degrees above max_degree."""
        if max_degree < self.max_degree:
            self.anm = self.anm[0:max_degree + 1, 0:max_degree + 1]

    # ---- arithmetic (grates/gravityfield.py:189-228) ----------------------------------------------------
    def __add__(self, other):
        _check_operand(self, other, PotentialCoefficients, '+')
        # the other set is brought to this set's GM and R first; the sum takes the larger of the two degrees (and, as
        # upstream, the left operand's epoch and constants)
        rescaled = other.anm * ((other.R / self.R) ** _degree_array(other.max_degree) * (other.GM / self.GM))
        if other.max_degree <= self.max_degree:
            total = self.copy()
            total.anm[:rescaled.shape[0], :rescaled.shape[1]] += rescaled
            return total
        total = PotentialCoefficients(self.GM, self.R)
        total.anm = rescaled
        total.anm[:self.anm.shape[0], :self.anm.shape[1]] += self.anm
        total.epoch = self.epoch
        return total

    def __sub__(self, other):
        _check_operand(self, other, PotentialCoefficients, '-')
        return self + other * -1

    def __mul__(self, factor):
        _check_operand(self, factor, (int, float), '*')
        scaled = self.copy()
        scaled.anm *= factor
        return scaled

    def __truediv__(self, divisor):
        _check_operand(self, divisor, (int, float), '/')
        return self * (1.0 / divisor)

    # ---- spectra ---------------------------------------------------------------------------------------------
    def degree_amplitudes(self, max_order=None, kernel='potential'):
        """Degree amplitudes sqrt(sum_m anm^2) / k_n * GM / R (grates/gravityfield.py:248-257)."""
        degrees = np.arange(self.max_degree + 1)
        amplitudes = np.zeros(degrees.size)
        ker = _kernel.get_kernel(kernel)
        for n in degrees:
            amplitudes[n] = np.sum(self.anm[degree_indices(n, max_order=max_order)] ** 2) * np.ravel(ker.inverse_coefficient(n))[0] ** 2
        return degrees, np.sqrt(amplitudes) * self.GM / self.R

    def coefficient_triangle(self, min_degree=2, max_degree=None):
        """Coefficients arranged as S | C triangle for plotting (grates/gravityfield.py:275-281)."""
        max_degree = self.max_degree if max_degree is None else max_degree
        triangle = np.hstack((np.rot90(self.anm, -1), self.anm))
        ones = np.ones(self.anm.shape, dtype=bool)
        mask = np.hstack((np.rot90(np.tril(ones), -1), np.triu(ones, 1)))
        mask[0:min_degree] = True
        return np.ma.masked_array(triangle, mask=mask)[0:max_degree + 1, :]

    def coefficient_amplitudes(self, kernel='potential'):
        """sqrt(C_nm^2 + S_nm^2) after conversion with `kernel` (grates/gravityfield.py:298-311)."""
        ker = _kernel.get_kernel(kernel)
        scaled = np.zeros(self.anm.shape)
        for n in range(self.max_degree + 1):
            idx = degree_indices(n)
            scaled[idx] = self.anm[idx] * self.GM / self.R * ker.inverse_coefficient(n)
        amp = np.zeros(self.anm.shape)
        amp[:, 0] = np.abs(scaled[:, 0])
        for m in range(1, self.max_degree + 1):
            amp[m:, m] = np.sqrt(scaled[m:, m] ** 2 + scaled[m - 1, m:] ** 2)
        return np.ma.masked_array(amp, mask=np.triu(np.ones(amp.shape, dtype=bool), 1))

    def coefficient_phases(self):
        """atan2(S_nm, C_nm) (grates/gravityfield.py:323-329)."""
        phase = np.zeros(self.anm.shape)
        for m in range(1, self.max_degree + 1):
            phase[m:, m] = np.arctan2(self.anm[m - 1, m:], self.anm[m:, m])
        return np.ma.masked_array(phase, mask=np.triu(np.ones(self.anm.shape, dtype=bool), 1))

    # ---- synthesis -----------------------------------------------------------------------------------------------
    def to_grid(self, grid=None, kernel='ewh'):
        """
        Gridded values of the functional `kernel` (default equivalent water height) on `grid` (default
        0.5 degree GeographicGrid).  Returns a deep copy of the grid holding the values; the input grid is
        untouched.  Regular grids (anything with `.parallels`) take the separable GPU path, other point sets
        the point-list GPU path -- the same dispatch rule as grates/gravityfield.py:352-370.
        """
        from . import grid as _grid
        if grid is None:
            grid = _grid.GeographicGrid()
        output = grid.copy()
        values = synthesize(self.anm[np.newaxis, :, :], grid, kernel, self.GM, self.R)
        output.values = engine.to_host(values[0]).ravel()
        return output

    @property
    def values(self):
        """Degree-wise vector of all coefficients."""
        return utilities.ravel_coefficients(self.anm)

    @values.setter
    def values(self, val):
        if val is None:
            self.anm = np.zeros((0, 0))        # upstream refers to a non-existent attribute here
        elif isinstance(val, np.ndarray):
            if val.ndim > 1:
                raise ValueError("unable to assign values of dimension {0:d} to gravity field".format(val.ndim))
            self.anm = utilities.unravel_coefficients(val)
        else:
            raise ValueError("grid values must be either None or " + str(np.ndarray))

    def gravitational_acceleration(self, xyz):
        """
        Gravitational acceleration [m/s^2] at cartesian positions xyz (m, 3).  Host-side helper (needed for the
        GRS80 normal gravity of the geoid / obp / deformation kernels); per-order formulation of
        grates/gravityfield.py:437-481.
        """
        from . import grid as _grid
        r, colat, lon = _grid.cartesian2spherical(xyz)
        N = self.max_degree
        n = np.arange(N + 1, dtype=float)
        legendre = _host_legendre_per_order

        def norm(factor):
            return factor * np.sqrt((2 * n + 1) / (2 * n + 3))

        g = np.empty((xyz.shape[0], 3))
        P_co = legendre(N + 1, 0, colat)
        P_p1 = legendre(N + 1, 1, colat)
        up = np.power(self.R / r[:, np.newaxis], n + 2)
        f_zero = norm(np.sqrt((n + 1) * (n + 1)))
        f_plus = norm(np.sqrt((n + 1) * (n + 2))) * np.sqrt(2)
        g[:, 0] = -((P_p1 * np.cos(lon)[:, np.newaxis]) * f_plus * up) @ self.anm[:, 0]
        g[:, 1] = -((P_p1 * np.sin(lon)[:, np.newaxis]) * f_plus * up) @ self.anm[:, 0]
        g[:, 2] = -2 * ((P_co[:, 1:] * f_zero) * up) @ self.anm[:, 0]
        for m in range(1, N + 1):
            P_m1, P_co = P_co, P_p1
            P_p1 = legendre(N + 1, m + 1, colat)
            nm = n[m:]
            cont = np.power(self.R / r[:, np.newaxis], nm + 2)
            base = np.sqrt((2 * nm + 1) / (2 * nm + 3))
            f_minus = np.sqrt((nm - m + 1) * (nm - m + 2)) * base * (np.sqrt(2) if m == 1 else 1.0)
            f_zero = np.sqrt((nm - m + 1) * (nm + m + 1)) * base
            f_plus = np.sqrt((nm + m + 1) * (nm + m + 2)) * base
            cm1, sm1 = np.cos((m - 1) * lon)[:, np.newaxis], np.sin((m - 1) * lon)[:, np.newaxis]
            c0, s0 = np.cos(m * lon)[:, np.newaxis], np.sin(m * lon)[:, np.newaxis]
            cp1, sp1 = np.cos((m + 1) * lon)[:, np.newaxis], np.sin((m + 1) * lon)[:, np.newaxis]
            C_minus, S_minus = cont * (P_m1[:, 2:] * cm1) * f_minus, cont * (P_m1[:, 2:] * sm1) * f_minus
            C_zero, S_zero = cont * (P_co[:, 1:] * c0) * f_zero, cont * (P_co[:, 1:] * s0) * f_zero
            C_plus, S_plus = cont * (P_p1 * cp1) * f_plus, cont * (P_p1 * sp1) * f_plus
            cnm, snm = self.anm[m:, m], self.anm[m - 1, m:]
            g[:, 0] += (C_minus - C_plus) @ cnm + (S_minus - S_plus) @ snm
            g[:, 1] += (-S_minus - S_plus) @ cnm + (C_minus + C_plus) @ snm
            g[:, 2] += -2 * C_zero @ cnm - 2 * S_zero @ snm
        return g * self.GM / (2 * self.R ** 2)


def _host_legendre_per_order(max_degree, order, colat):
    """Host NumPy per-order Legendre recursion (formulas of grates/utilities.py:62-115, 138-151) for the few
    evaluation points of the reference-field helpers; the GPU version is utilities.legendre_functions_per_order."""
    t = np.cos(np.atleast_1d(colat))
    if order == 0:
        return utilities.legendre_polynomials(max_degree, colat)
    s = np.sqrt(1 - t ** 2)
    out = np.empty((t.size, max_degree + 1 - order))
    pmm = np.sqrt(3) * s
    for n in range(2, order + 1):
        pmm = np.sqrt((2 * n + 1) / (2 * n)) * s * pmm
    out[:, 0] = pmm
    if out.shape[1] > 1:
        out[:, 1] = np.sqrt(2 * order + 3) * t * out[:, 0]
    for n in range(order + 2, max_degree + 1):
        out[:, n - order] = np.sqrt((2 * n - 1) / (n - order) * (2 * n + 1) / (n + order)) * t * out[:, n - 1 - order] - \
            np.sqrt((2 * n + 1) / (2 * n - 3) * (n - order - 1) / (n - order) * (n + order - 1) / (n + order)) * out[:, n - 2 - order]
    return out


def synthesize(anm_batch, grid, kernel='ewh', GM=3.9860044150e+14, R=6.3781363000e+06):
    """
    Batched synthesis: anm_batch [B, N+1, N+1] (ndarray or device tensor) -> device tensor
    [B, nlat, nlon] for regular grids or [B, npts] for point lists.  This is the entry point the
    benchmark times; ``PotentialCoefficients.to_grid`` and ``TimeSeries.to_grid`` are thin wrappers.
    """
    max_degree = anm_batch.max_degree if isinstance(anm_batch, engine.OrderMajorSeries) else anm_batch.shape[-1] - 1
    ker = _kernel.get_kernel(kernel)
    try:
        parallels, meridians = grid.parallels, grid.meridians
    except AttributeError:
        def build():
            colat, _, kn = surface_factors(ker, max_degree, grid.latitude, GM, R, grid.semimajor_axis, grid.flattening)
            return colat, grid.longitude, kn
        colat, lon, kn = engine.cached_point_tables(max_degree, grid.latitude, grid.longitude,
                                                    (str(kernel), float(GM), float(R), float(grid.semimajor_axis), float(grid.flattening)), build)
        if isinstance(anm_batch, engine.OrderMajorSeries):
            anm_batch = anm_batch.to_batch()
        return engine.synthesis_points(max_degree, colat, lon, kn, anm_batch)
    def build():
        colat, _, kn = surface_factors(ker, max_degree, parallels, GM, R, grid.semimajor_axis, grid.flattening)
        return engine.cached_plan(max_degree, colat, kn, meridians)
    # the kernel table of the grid (2.6 ms of NumPy at d/o 96 / 0.25 degree) and the content hash of its 0.6 MB (0.9 ms) are more than the
    # 0.5 ms the device needs for 240 epochs: repeated calls on the same grid find their plan by a hash of the grid's two axes alone
    plan = engine.plan_by_grid(('regular', kernel, int(max_degree), float(GM), float(R), float(grid.semimajor_axis), float(grid.flattening)),
                               (parallels, meridians), build) if isinstance(kernel, str) else build()
    return plan.synthesis(anm_batch)


class TimeSeries:
    """
    Time series of gravity fields of one type, sorted by epoch (grates/gravityfield.py:815-1052).

    Beside the reference's list of fields the series can live ON THE DEVICE as an `engine.OrderMajorSeries` (coefficients of all epochs,
    epochs fastest, laid out for the order-wise operators): `to_device()` builds it once, `filter.*.filter(series)`, `detrend` and
    `to_grid` then work on it without the coefficients ever passing through host arrays, and the fields are only materialised again
    when something asks for them (`series[k]`, `items()`, arithmetic).  Whoever receives field objects may modify them, so handing
    them out ends the device residency (the next device operator packs the fields again).
    """

    def __init__(self, data):
        self.__data = data
        self.__dtype = type(self.__data[0])
        for d in self.__data:
            if not isinstance(d, self.__dtype):
                raise ValueError("Found inconsistent data types (" + str(self.__dtype) + " and " + str(type(d)) + ")")
            if d.epoch is None:
                raise ValueError("At least one data point has no valid time stamp")
        self.__series = None            # engine.OrderMajorSeries in the order of the (sorted) epochs, or None
        self.__meta = None              # (epochs, GM, R) of a series without materialised fields
        self.sort()

    @classmethod
    def from_series(cls, series, epochs, GM=3.9860044150e+14, R=6.3781363000e+06):
        """A device-resident series of PotentialCoefficients: `series` an engine.OrderMajorSeries (or a coefficient batch
        [T, N+1, N+1], packed here) whose epochs carry the time stamps `epochs` (ascending)."""
        epochs = list(epochs)
        if not isinstance(series, engine.OrderMajorSeries):
            series = engine.OrderMajorSeries.from_batch(series)
        if series.epochs != len(epochs) or len(epochs) == 0:
            raise ValueError('one time stamp per epoch of the series expected')
        if any(e is None for e in epochs):
            raise ValueError("At least one data point has no valid time stamp")
        if any(b < a for a, b in zip(epochs, epochs[1:])):
            raise ValueError('the epochs of a device series must be ascending')
        self = cls.__new__(cls)
        self.__data = None
        self.__dtype = PotentialCoefficients
        self.__series = series
        self.__meta = (epochs, GM, R)
        return self

    # ---- the two representations
    def _fields(self, keep_series=False):
        """the list of fields; built from the device series on first use.  Unless `keep_series`, the device copy is dropped: the
        caller may change the fields it gets."""
        if self.__data is None:
            epochs, GM, R = self.__meta
            host = engine.to_host(self.__series.to_batch())
            self.__data = []
            for k, epoch in enumerate(epochs):
                gf = PotentialCoefficients(GM, R)
                gf.anm = host[k].copy()
                gf.epoch = epoch
                self.__data.append(gf)
        if not keep_series:
            self.__series = None
        return self.__data

    @property
    def on_device(self):
        return self.__series is not None

    @property
    def uniform_degree(self):
        """whether all fields have one maximum degree (a device series pads lower ones with zeros: fine for `to_grid`, but a filter with
        dense blocks would fill the padding, where the per-field call of the reference keeps every field's own degree)"""
        return self.__data is None or len({d.max_degree for d in self.__data}) == 1

    def to_device(self):
        """The series as engine.OrderMajorSeries (built on first use; PotentialCoefficients of one GM and R only, fields of a lower
        degree zero-padded).  Stays valid until fields are handed out."""
        if self.__series is None:
            fields = self.__data
            if self.__dtype is not PotentialCoefficients and not all(isinstance(d, PotentialCoefficients) for d in fields):
                raise TypeError('only a series of PotentialCoefficients can be kept on the device')
            GM, R = fields[0].GM, fields[0].R
            if any(d.GM != GM or d.R != R for d in fields):
                raise ValueError("batched operators need a common GM and R for all epochs")
            self.__series = engine.OrderMajorSeries.from_batch(TimeSeries._stack(fields))
            self.__meta = ([d.epoch for d in fields], GM, R)
        return self.__series

    def _constants(self):
        if self.__data is not None:
            return self.__data[0].GM, self.__data[0].R
        return self.__meta[1], self.__meta[2]

    def _with_series(self, series):
        """a new device-resident series with the epochs and constants of this one (the result of a device operator)"""
        GM, R = self._constants()
        return TimeSeries.from_series(series, self.epochs(), GM, R)

    def __len__(self):
        return len(self.__data) if self.__data is not None else self.__series.epochs

    def __getitem__(self, index):
        return self._fields()[index]

    def __setitem__(self, index, value):
        if not isinstance(value, self.__dtype):
            raise ValueError("Inconsistent data types (" + str(self.__dtype) + " and " + str(type(value)) + ")")
        self._fields()[index] = value
        self.sort()

    def copy(self):
        if self.__data is None:
            return self._with_series(self.__series.like(self.__series.data.clone()))
        return TimeSeries([d.copy() for d in self.__data])

    def __add__(self, other):
        if len(self) != len(other):
            raise ValueError("Length of time series differs")
        if isinstance(other, TimeSeries) and self.__series is not None and other.__series is not None and \
                self.__series.max_degree == other.__series.max_degree and self._constants() == other._constants():
            # both on the device, same degree and constants (the reference's rescaling factor (R2 / R1)^n GM2 / GM1 is exactly 1): one sum
            if self.epochs() != other.epochs():
                raise ValueError("Time stamps of elements differ")
            return self._with_series(self.__series.like(self.__series.data + other.__series.data))
        mine = self._fields(keep_series=True)
        theirs = other._fields(keep_series=True) if isinstance(other, TimeSeries) else other      # (no field leaves either series)
        summed = []
        for k in range(len(self)):
            if mine[k].epoch != theirs[k].epoch:
                raise ValueError("Time stamps of elements differ")
            summed.append(mine[k] + theirs[k])
        return TimeSeries(summed)

    def __mul__(self, factor):
        _check_operand(self, factor, (int, float), '*')
        if self.__series is not None:
            return self._with_series(self.__series.like(self.__series.data * factor))
        return TimeSeries([d.copy() * factor for d in self.__data])

    def __truediv__(self, divisor):
        _check_operand(self, divisor, (int, float), '*')          # (upstream reports '*' here as well)
        return self * (1.0 / divisor)

    def __sub__(self, other):
        return self + other * -1

    def sort(self):
        if self.__data is None:
            return                                                  # (a device series is created with ascending epochs)
        order = sorted(range(len(self.__data)), key=lambda k: self.__data[k].epoch)
        if order != list(range(len(order))):
            self.__data[:] = [self.__data[k] for k in order]
            self.__series = None

    def items(self):
        for d in self._fields():
            yield d.epoch, d

    def epochs(self):
        if self.__data is None:
            return list(self.__meta[0])
        return [d.epoch for d in self.__data]

    def interpolate_to(self, epoch):
        """Piecewise linear interpolation to `epoch`; no extrapolation (grates/gravityfield.py:937-947)."""
        data = self._fields(keep_series=True)
        t = np.array([d.epoch for d in data])
        if t.size < 2:
            raise ValueError("at least two data points are required for interpolation")
        if not (t[0] <= epoch <= t[-1]):
            raise ValueError('{0} lies outside the series ({1} .. {2}): no extrapolation'.format(epoch, t[0], t[-1]))
        right = np.searchsorted(t, epoch)                     # first element at or after the epoch (as upstream: side='left')
        before, after = data[right - 1], data[right]
        w = (epoch - before.epoch).total_seconds() / (after.epoch - before.epoch).total_seconds()
        blend = before * (1 - w) + after * w
        blend.epoch = epoch
        return blend

    def evaluate_at(self, epoch):
        return self.interpolate_to(epoch)

    def to_array(self):
        """Time series as (epochs, parameters) array of the degree-wise vectors (grates/gravityfield.py:973-980)."""
        if self.__data is None:
            return engine.to_host(self.__series.to_array())
        count = self.__data[0].values.size
        out = np.empty((len(self.__data), count))
        for k, d in enumerate(self.__data):
            out[k, :] = d.values[0:count]
        return out

    @staticmethod
    def _stack(fields):
        nmax = max(d.max_degree for d in fields)
        out = np.zeros((len(fields), nmax + 1, nmax + 1))
        for k, d in enumerate(fields):
            out[k, 0:d.max_degree + 1, 0:d.max_degree + 1] = d.anm
        return out

    def to_coefficient_batch(self):
        """Stack anm of all epochs into [T, N+1, N+1] (zero-padded to the largest degree): GPU batch layout."""
        if self.__data is None:
            return engine.to_host(self.__series.to_batch())
        return TimeSeries._stack(self.__data)

    def to_grid(self, grid=None, kernel='ewh', as_tensor=False):
        """
        Synthesize every epoch in one batched GPU call (from the device series when there is one: the coefficients never pass
        through host arrays).  Returns a list of grids (copies of `grid` with values and epoch set) or, with as_tensor=True, the
        device tensor [T, nlat, nlon].  All epochs must share GM and R.
        """
        from . import grid as _grid
        if grid is None:
            grid = _grid.GeographicGrid()
        if self.__series is not None:
            GM, R = self._constants()
            values = synthesize(self.__series, grid, kernel, GM, R)
        else:
            GM, R = self.__data[0].GM, self.__data[0].R
            if any(d.GM != GM or d.R != R for d in self.__data):
                raise ValueError("batched synthesis needs a common GM and R for all epochs")
            values = synthesize(self.to_coefficient_batch(), grid, kernel, GM, R)
        if as_tensor:
            return values
        host = engine.to_host(values)
        out = []
        for k, epoch in enumerate(self.epochs()):
            g = grid.copy()
            g.values = host[k].ravel().copy()
            g.epoch = epoch
            out.append(g)
        return out

    def detrend(self, basis_functions):
        """
        Estimate and remove a parametric temporal model in place (grates/gravityfield.py:1002-1012); `basis_functions` are
        `utilities.Polynomial` / `utilities.Oscillation` instances.  The pseudo-inverse of the [T, k] design matrix is host work;
        the two products over all coefficients (k x T x P and T x k x P) run on the fp64 GEMM of the device -- on the device series
        itself when there is one (the residuals stay there), else on the stacked degree-wise vectors of the fields.
        """
        t = self.epochs()
        design = np.hstack([bf.design_matrix(t) for bf in basis_functions])
        pinv = np.linalg.pinv(design)
        if self.__data is None:
            series = self.__series
            obs = series.values                                                       # [P rows, T], rows in order-major order
            trend_t = engine.gemm(obs, pinv, transb=True)                             # [P, k] = obs pinv^T
            engine.gemm(trend_t, design, transb=True, alpha=-1.0, beta=1.0, out=obs)  # obs -= trend^T design^T
            rows = engine.order_major_rows_of_degreewise(series.max_degree)
            return np.ascontiguousarray(engine.to_host(trend_t)[rows].T)
        observations = engine.to_device(self.to_array())
        trend = engine.gemm(pinv, observations)
        engine.gemm(design, trend, alpha=-1.0, beta=1.0, out=observations)
        residuals = engine.to_host(observations)
        for k, d in enumerate(self.__data):
            d.values = residuals[k, :]
        self.__series = None
        return engine.to_host(trend)

    def bin(self, bin_center_epochs, func=np.mean, no_data=np.nan):
        """
        Aggregate the series in bins: every element goes to the nearest bin centre, `func` (default: mean) is applied to the
        elements of a bin, the result carries the centre as epoch (grates/gravityfield.py:1014-1045).  An empty bin is an
        error (upstream fails on it while assigning the epoch; its `no_data` argument is accepted here as there and, as there,
        never used).  As upstream, numpy.mean of PotentialCoefficients ends in a
        TypeError (division by a numpy integer): pass e.g. ``lambda v: sum(v[1:], v[0]) * (1.0 / len(v))``.
        """
        data = self._fields(keep_series=True)
        centers = list(bin_center_epochs)
        nearest = [int(np.argmin([abs((e - c).total_seconds()) for c in centers])) for e in self.epochs()]
        binned = []
        for k, center in enumerate(centers):
            members = [data[i] for i, b in enumerate(nearest) if b == k]
            if not members:
                raise ValueError('no element of the time series falls into the bin centred at {0}'.format(center))
            value = func(members)
            value.epoch = center
            binned.append(value)
        return TimeSeries(binned)

    def append(self, other):
        data = self._fields()
        for _, d in other.items():
            data.append(d)
        self.sort()


def gridded_rms(temporal_gravityfield, epochs, kernel='ewh', base_grid=None, batch=240):
    """
    RMS over `epochs` of a time variable gravity field in the space domain (grates/gravityfield.py:1143-1172).  The fields of up
    to `batch` epochs are synthesised in one call of the batched synthesis and reduced on the device (`shg_epoch_rms`: squares
    added in epoch order); only the RMS grid comes back to the host.
    """
    from .grid import GeographicGrid
    base_grid = GeographicGrid() if base_grid is None else base_grid
    epochs = list(epochs)
    acc = None
    for start in range(0, len(epochs), batch):
        fields = [temporal_gravityfield.evaluate_at(t) for t in epochs[start:start + batch]]
        last = start + len(fields) == len(epochs)
        uniform = all(isinstance(f, PotentialCoefficients) and f.GM == fields[0].GM and f.R == fields[0].R for f in fields)
        if uniform:
            values = synthesize(TimeSeries._stack(fields), base_grid, kernel, fields[0].GM, fields[0].R)
            acc = engine.epoch_rms(values.reshape(len(fields), -1), acc, len(epochs) if last else 0)
        else:                                                  # mixed constants or other representations: one synthesis per field
            for k, field in enumerate(fields):
                values = engine.to_device(field.to_grid(base_grid, kernel=kernel).values).reshape(1, -1)
                acc = engine.epoch_rms(values, acc, len(epochs) if last and k == len(fields) - 1 else 0)
    rms_grid = base_grid.copy()
    rms_grid.values = engine.to_host(acc) if acc is not None else np.full(base_grid.point_count, np.nan)
    return rms_grid


# -------------------------------------------------------------------------------------------------------
# coefficient sequences (grates/gravityfield.py:1175-1471)
# -------------------------------------------------------------------------------------------------------

class CoefficientSequence:
    """Ordered list of (basis_function, degree, order) triples with index lookups."""

    class Coefficient:
        __slots__ = ['degree', 'order', 'basis_function']

        def __init__(self, basis_function, n, m):
            self.degree = n
            self.order = m
            self.basis_function = basis_function

        def __str__(self):
            return 'Coefficient({0}, {1:d}, {2:d})'.format('c' if self.basis_function == 0 else 's', self.degree, self.order)

        __repr__ = __str__

        def __eq__(self, other):
            return self.basis_function == other.basis_function and self.degree == other.degree and self.order == other.order

        def key(self):
            return (int(self.basis_function), int(self.degree), int(self.order))

    def __init__(self, coefficients):
        self.coefficients = tuple(coefficients)
        self.__table = np.array([c.key() for c in self.coefficients], dtype=np.int64).reshape(-1, 3)

    @property
    def coefficient_count(self):
        return len(self.coefficients)

    def as_array(self):
        """Integer array [count, 3] of (basis_function, degree, order)."""
        return self.__table.copy()

    def vector_indices(self, degree=None, order=None, cs=None):
        """Indices of all coefficients matching degree / order / basis function ('c' or 's')."""
        mask = np.ones(self.coefficient_count, dtype=bool)
        if degree is not None:
            mask &= self.__table[:, 1] == degree
        if order is not None:
            mask &= self.__table[:, 2] == order
        if cs is not None:
            if cs in ('c', 'cos', 'cosine'):
                mask &= self.__table[:, 0] == 0
            elif cs in ('s', 'sin', 'sine'):
                mask &= self.__table[:, 0] == 1
            else:
                raise ValueError('basis function not recognized')
        return np.where(mask)[0]

    @staticmethod
    def reorder_indices(source_sequence, target_sequence):
        """Indices of the common coefficients in the source and in the target sequence, sorted by the
        target's ordering rule (grates/gravityfield.py:1283-1288)."""
        source = {c.key(): k for k, c in enumerate(source_sequence.coefficients)}
        target = {c.key(): k for k, c in enumerate(target_sequence.coefficients)}
        common = sorted(set(source) & set(target), key=target_sequence.sort_key)
        return np.array([source[c] for c in common], dtype=int), np.array([target[c] for c in common], dtype=int)


class CoefficientSequenceDegreeWise(CoefficientSequence):
    """C00, C10, C11, S11, C20, C21, S21, C22, S22, ..."""

    def __init__(self, min_degree, max_degree):
        triples = []
        for n in range(min_degree, max_degree + 1):
            triples.append(self.Coefficient(np.int8(0), n, 0))
            for m in range(1, n + 1):
                triples.append(self.Coefficient(np.int8(0), n, m))
                triples.append(self.Coefficient(np.int8(1), n, m))
        super().__init__(triples)

    @staticmethod
    def sort_key(key):
        basis, n, m = key
        return (n, m, basis)


class CoefficientSequenceOrderWiseAlternating(CoefficientSequence):
    """Order by order; within an order by degree with alternating cosine / sine."""

    def __init__(self, min_degree, max_degree):
        triples = [self.Coefficient(np.int8(0), n, 0) for n in range(min_degree, max_degree + 1)]
        for m in range(1, max_degree + 1):
            for n in range(max(min_degree, m), max_degree + 1):
                triples.append(self.Coefficient(np.int8(0), n, m))
                triples.append(self.Coefficient(np.int8(1), n, m))
        super().__init__(triples)

    @staticmethod
    def sort_key(key):
        basis, n, m = key
        return (m, n, basis)


class CoefficientSequenceOrderWise(CoefficientSequence):
    """Order by order; within an order all cosines by degree, then all sines by degree."""

    def __init__(self, min_degree, max_degree):
        triples = [self.Coefficient(np.int8(0), n, 0) for n in range(min_degree, max_degree + 1)]
        for m in range(1, max_degree + 1):
            for basis in (0, 1):
                for n in range(max(min_degree, m), max_degree + 1):
                    triples.append(self.Coefficient(np.int8(basis), n, m))
        super().__init__(triples)

    @staticmethod
    def sort_key(key):
        basis, n, m = key
        return (m, basis, n)


class CoefficientSequenceFlatArray(CoefficientSequence):
    """Row-major order of the packed (N+1, N+1) coefficient array."""

    def __init__(self, max_degree):
        triples = []
        for row in range(max_degree + 1):
            for col in range(max_degree + 1):
                if col <= row:
                    triples.append(self.Coefficient(np.int8(0), row, col))
                else:
                    triples.append(self.Coefficient(np.int8(1), col, row + 1))
        super().__init__(triples)

    @staticmethod
    def sort_key(key):
        basis, n, m = key
        return (n, m) if basis == 0 else (m - 1, n)


# -------------------------------------------------------------------------------------------------------
# space-domain representations on point lists (grates/gravityfield.py:484-785)
# -------------------------------------------------------------------------------------------------------

_POINT_CHUNK_BYTES = 2 << 30      # device table of one block of nodal points


def _point_harmonics_adjoint(points, max_degree, values, upward=None):
    """
    sum_p values[p] Y_nm(theta_p, lambda_p) [(R / r_p)^(n+1)] as a device array [N+1, N+1] (coefficient layout): the adjoint
    of the point-list synthesis.  The harmonics of a block of points are built on the device (Legendre recursion and
    cos / sin tables of the C-ABI), the sum over the points is one transposed fp64 GEMM per block.
    `upward` = reference radius R: scale degree n by (R / r_p)^(n+1).
    """
    from .grid import _degree_scale_array
    torch = engine.require_gpu()
    colat = utilities.colatitude(points.latitude, points.semimajor_axis, points.flattening)
    size = (max_degree + 1) ** 2
    chunk = max(int(_POINT_CHUNK_BYTES // (8 * size)), 1)
    v = engine.to_device(np.ascontiguousarray(values, dtype=float)).reshape(-1, 1)
    out = torch.zeros((size, 1), dtype=torch.float64, device=v.device)
    for start in range(0, colat.size, chunk):
        block = slice(start, min(start + chunk, colat.size))
        Y = engine.trigonometric_functions(max_degree, points.longitude[block])
        Y *= engine.legendre_functions(max_degree, colat[block])
        if upward is not None:
            radius = utilities.geocentric_radius(points.latitude[block], points.semimajor_axis, points.flattening)
            kn = np.power((upward / radius)[:, np.newaxis], np.arange(max_degree + 1, dtype=int) + 1)
            Y *= _degree_scale_array(kn, max_degree)
        engine.gemm(Y.reshape(Y.shape[0], size), v[block], transa=True, beta=1.0, out=out)
    return out.reshape(max_degree + 1, max_degree + 1)


class RadialBasisFunctions:
    """
    Gravity field as radial basis functions at the nodal points of `point_distribution`: shape factors `K` in the
    coefficient layout [N+1, N+1], frequency band min_degree .. max_degree (grates/gravityfield.py:652-785).
    """

    def __init__(self, point_distribution, K, min_degree, max_degree, GM=3.9860044150e+14, R=6.3781363000e+06):
        self.__K = K.copy()
        self.point_distribution = point_distribution.copy()
        self.__min_degree = min_degree
        self.__max_degree = max_degree
        self.GM = GM
        self.R = R
        self.epoch = None
        self.values = np.zeros((self.point_distribution.size))

    def copy(self):
        rbf = RadialBasisFunctions(self.point_distribution.copy(), self.__K, self.__min_degree, self.__max_degree, self.GM, self.R)
        rbf.epoch = self.epoch
        rbf.values = self.values.copy()
        return rbf

    @property
    def values(self):
        return self.point_distribution.values

    @values.setter
    def values(self, val):
        self.point_distribution.values = val

    def is_compatible(self, other):
        return self.point_distribution.is_compatible(other.point_distribution)

    def to_potential_coefficients(self, blocking_factor=256):
        """
        anm = K * sum_p values[p] (R / r_p)^(n+1) Y_nm(p) (grates/gravityfield.py:705-727; all degrees 0 .. max_degree, as
        upstream).  `blocking_factor` is accepted for compatibility: the blocks are sized for the device (2 GB tables).
        """
        coefficients = PotentialCoefficients(self.GM, self.R)
        total = _point_harmonics_adjoint(self.point_distribution, self.__max_degree, self.values, upward=self.R)
        coefficients.anm = engine.to_host(total) * self.__K
        coefficients.epoch = self.epoch
        return coefficients

    def to_potential_coefficients_matrix(self, blocking_factor=256):
        """F [P, points] with coefficients (degree-wise, min_degree .. max_degree) = F values
        (grates/gravityfield.py:729-764)."""
        points = self.point_distribution
        colat = utilities.colatitude(points.latitude, points.semimajor_axis, points.flattening)
        radius = utilities.geocentric_radius(points.latitude, points.semimajor_axis, points.flattening)
        kn = np.power((self.R / radius)[:, np.newaxis], np.arange(self.__max_degree + 1, dtype=int) + 1)
        from .grid import _degree_scale_array
        Y = engine.trigonometric_functions(self.__max_degree, points.longitude)
        Y *= engine.legendre_functions(self.__max_degree, colat)
        Y *= _degree_scale_array(kn, self.__max_degree)
        Y *= engine.to_device(self.__K)
        return engine.to_host(engine.ravel(Y, self.__min_degree, self.__max_degree).T.contiguous())

    def to_grid(self, grid=None, kernel='ewh'):
        """gridded values through the spherical harmonic representation (grates/gravityfield.py:766-785)"""
        from .grid import GeographicGrid
        return self.to_potential_coefficients().to_grid(GeographicGrid() if grid is None else grid, kernel)


# -------------------------------------------------------------------------------------------------------
# reference fields (grates/gravityfield.py:1474-1574)
# -------------------------------------------------------------------------------------------------------

class ReferenceField(PotentialCoefficients):
    """
    Geodetic reference system: level ellipsoid (a, f or J2) rotating with omega, and its normal gravity field
    as even zonal coefficients.  Either f or J2 must be given.
    """

    def __init__(self, GM, omega, a, f=None, J2=None):
        self.omega = omega
        k = np.arange(1, 21, dtype=float)

        def q0_of(e):
            ep = e / np.sqrt(1 - e ** 2)
            return -2 * np.sum(np.power(-1, k) * k * np.power(ep, 2 * k + 1) / ((2 * k + 1) * (2 * k + 3)))

        if J2 is None and f is not None:
            self.flattening = f
            e2 = f * (2 - f)
            e = np.sqrt(e2)
            self.J2 = (e2 - 4 / 15 * (omega ** 2 * a ** 3) / GM * e ** 3 / (2 * q0_of(e))) / 3
        elif f is None and J2 is not None:
            self.J2 = J2
            e, previous = 0.1, np.inf
            while not np.isclose(e, previous, atol=1e-22, rtol=0):
                previous = e
                e = np.sqrt(3 * J2 + 4 / 15 * (omega ** 2 * a ** 3) / GM * e ** 3 / (2 * q0_of(e)))
            e2 = e ** 2
            self.flattening = 1 - np.sqrt(1 - e2)
        else:
            raise ValueError('either flattening f or dynamic force factor J2 must be given for a full definition of the reference field')

        zonals = [1.0]
        n = 1
        while not np.isclose(zonals[-1], 0, atol=1e-22, rtol=0):
            sign = 1 if n % 2 == 0 else -1
            zonals.append(sign * (3 * e2 ** n * (1 - n + 5 * n * self.J2 / e2) / ((2 * n + 1) * (2 * n + 3) * np.sqrt(4 * n + 1))))
            n += 1
        max_degree = (len(zonals) - 1) * 2
        super().__init__(GM, a)
        self.anm = np.zeros((max_degree + 1, max_degree + 1))
        self.anm[0::2, 0] = zonals

    def normal_gravity(self, r, colat):
        """Normal gravity (gravitation + centrifugal) [m/s^2] projected on the ellipsoid normal, at geocentric
        radius r and colatitude colat (grates/gravityfield.py:1560-1570)."""
        from . import grid as _grid
        count = max(np.asarray(r).size, np.asarray(colat).size)
        xyz = np.zeros((count, 3))
        xyz[:, 0] = r * np.sin(colat)
        xyz[:, 2] = r * np.cos(colat)
        _, lat, _ = _grid.cartesian2geodetic(xyz, self.R, self.flattening)
        g = self.gravitational_acceleration(xyz)
        g[:, 0] += self.omega ** 2 * xyz[:, 0]
        return -np.cos(lat) * g[:, 0] - np.sin(lat) * g[:, 2]


WGS84 = ReferenceField(GM=3986004.418e8, omega=7292115.0e-11, a=6378137.0, f=1 / 298.257223563)
GRS80 = ReferenceField(GM=3986005e8, omega=7292115.0e-11, a=6378137.0, J2=108263e-8)


from .timevariable import Oscillation, TimeVariableGravityField, Trend  # noqa: E402,F401  (reference import paths)


def __getattr__(name):
    """`gravityfield.SurfaceMasCons` / `AnisotropicBasisFunctions` resolve like in the reference; the classes live in
    grates_amd.extras (outside the hot path's scope) and are imported on first use."""
    if name in ('SurfaceMasCons', 'AnisotropicBasisFunctions'):
        from . import extras
        return getattr(extras, name)
    raise AttributeError('module {0!r} has no attribute {1!r}'.format(__name__, name))
