"""
Representations and point distributions that SURVEY.md section 2 marks OUT OF SCOPE for the hot path (they are not rows of the
section 8 coverage table): kept as consumers of the in-scope kernels because golden vectors of the reference pin them
(tests/golden/g15_basis_functions.npz, g17_reuter.npz), outside the coverage claim of DESIGN.md.

    SurfaceMasCons              point masses on a grid                        grates/gravityfield.py:484-570
    AnisotropicBasisFunctions   anisotropic kernel functions at nodal points  grates/gravityfield.py:573-649
    ReuterGrid                  Reuter point distribution                     grates/grid.py:1207-1278

All three go through the point-list synthesis and its adjoint (csrc/points.hip) and the regular-grid synthesis.
"""

import numpy as np

from . import engine
from . import utilities
from .gravityfield import PotentialCoefficients, _check_operand, _point_harmonics_adjoint
from .grid import IrregularGrid, authalic2geodetic, conformal2geodetic, geocentric2geodetic


class SurfaceMasCons:
    """
    Point masses / surface elements on a grid whose values are a gravity field functional `kernel`
    (grates/gravityfield.py:484-570).  Arithmetic is point-wise on the values.
    """

    def __init__(self, point_distribution, kernel):
        self.point_distribution = point_distribution
        if self.point_distribution.values is None:
            self.point_distribution.values = np.zeros(self.point_distribution.point_count)
        self.kernel = kernel
        self.epoch = None

    def copy(self):
        other = SurfaceMasCons(self.point_distribution.copy(), self.kernel)
        other.epoch = self.epoch
        return other

    def is_compatible(self, other):
        return self.point_distribution.is_compatible(other.point_distribution)

    @property
    def values(self):
        return self.point_distribution.values

    @values.setter
    def values(self, val):
        self.point_distribution.values = val

    def __combine(self, other, symbol, sign):
        _check_operand(self, other, SurfaceMasCons, symbol)
        if not self.is_compatible(other):
            raise ValueError("point distributions of '" + str(type(self)) + "' instances are not compatible")
        result = self.copy()
        result.values = result.values + sign * other.values
        return result

    def __add__(self, other):
        return self.__combine(other, '+', 1.0)

    def __sub__(self, other):
        return self.__combine(other, '-', -1.0)

    def __mul__(self, other):
        _check_operand(self, other, (int, float), '*')
        result = self.copy()
        result.values = result.values * other
        return result

    def __truediv__(self, other):
        _check_operand(self, other, (int, float), '/')
        return self * (1.0 / other)

    def to_potential_coefficients(self, min_degree, max_degree, GM=3.9860044150e+14, R=6.3781363000e+06):
        """
        Spherical harmonic analysis of the mascon values through the analysis operator of the point distribution.
        (Upstream hands the builtin ``round`` to the grid in place of R, grates/gravityfield.py:570, and cannot run; R is
        passed here.)
        """
        return self.point_distribution.to_potential_coefficients(min_degree, max_degree, self.kernel, GM, R)


class AnisotropicBasisFunctions:
    """
    Gravity field as anisotropic kernel functions at the nodal points: `K` [P, P] acts on the degree-wise vector of the
    point harmonics, band min_degree .. max_degree (grates/gravityfield.py:573-649).
    """

    def __init__(self, point_distribution, K, min_degree, max_degree, GM=3.9860044150e+14, R=6.3781363000e+06):
        self.__K = K.copy()
        self.point_distribution = point_distribution
        self.__min_degree = min_degree
        self.__max_degree = max_degree
        self.GM = GM
        self.R = R
        self.epoch = None
        self.values = np.zeros((self.point_distribution.size))

    @property
    def values(self):
        return self.point_distribution.values

    @values.setter
    def values(self, val):
        self.point_distribution.values = val

    def is_compatible(self, other):
        return self.point_distribution.is_compatible(other.point_distribution)

    def to_potential_coefficients(self):
        """x = K (Y^T values) as potential coefficients (degrees below min_degree zero): the coefficient vector the
        reference forms per block of nodal points inside to_grid (grates/gravityfield.py:637-639)."""
        total = _point_harmonics_adjoint(self.point_distribution, self.__max_degree, self.values)
        y = engine.ravel(total.unsqueeze(0), self.__min_degree, self.__max_degree)
        x = engine.gemm(engine.to_device(self.__K), y.reshape(-1, 1))
        coefficients = PotentialCoefficients(self.GM, self.R)
        coefficients.anm = utilities.unravel_coefficients(engine.to_host(x).ravel(), self.__min_degree, self.__max_degree)
        coefficients.epoch = self.epoch
        return coefficients

    def to_grid(self, grid=None, kernel='ewh'):
        """
        Gridded values: the kernel coefficient vector K Y^T values is synthesised on the parallels of `grid` with the
        kernel factors, the upward continuation (R / r)^(n+1) and GM / R (grates/gravityfield.py:604-649) -- the regular
        grid synthesis of the hot path.
        """
        from .grid import GeographicGrid
        return self.to_potential_coefficients().to_grid(GeographicGrid() if grid is None else grid, kernel)


class ReuterGrid(IrregularGrid):
    """
    Reuter grid of a given level: level + 1 parallels at equal spacing on the unit sphere, on each of them as many points as
    keep the spherical distance to the neighbours near pi / level; poles are single points.  The sphere is mapped onto the
    ellipsoid by `latitude_mapping` ('geocentric', 'authalic' or 'conformal') (grates/grid.py:1207-1278).  The usual nodal
    point distribution of `RadialBasisFunctions` / `SurfaceMasCons`; area elements are those of the unit sphere.
    """

    def __init__(self, level, a=6378137.0, f=298.2572221010**-1, latitude_mapping='geocentric'):
        mappings = {'authalic': authalic2geodetic, 'geocentric': geocentric2geodetic, 'conformal': conformal2geodetic}
        if latitude_mapping.lower() not in mappings:
            raise ValueError('Unknown latitude mapping "{0}".'.format(latitude_mapping))
        dlat = np.pi / level
        polar_cap = 2 * np.pi * (1 - np.cos(dlat * 0.5))
        latitude, counts, areas = [0.5 * np.pi], [1], [polar_cap]
        for k in range(1, level):
            theta = k * dlat
            count = int(2 * np.pi / np.arccos((np.cos(dlat) - np.cos(theta) ** 2) / (np.sin(theta) ** 2)))
            latitude.append(np.pi * 0.5 - theta)
            counts.append(count)
            areas.append(4 * np.pi / count * np.sin(0.5 * dlat) * np.cos(latitude[-1]))
        latitude.append(-0.5 * np.pi)
        counts.append(1)
        areas.append(polar_cap)
        latitude = mappings[latitude_mapping.lower()](np.array(latitude), f)
        lon = [np.zeros(1) if k in (0, level) else np.mod((np.arange(n) + 1.5) * 2 * np.pi / n + np.pi, 2 * np.pi) - np.pi
               for k, n in enumerate(counts)]
        super().__init__(np.concatenate(lon), np.repeat(latitude, counts), np.repeat(np.array(areas), counts), a, f)
        self.__level, self.__mapping = level, latitude_mapping

    def copy(self):
        # (upstream's copy falls back to the geocentric mapping; the mapping is kept here)
        other = ReuterGrid(self.__level, self.semimajor_axis, self.flattening, self.__mapping)
        other.values = None if self.values is None else self.values.copy()
        other.epoch = self.epoch
        return other
