"""
grates_amd -- MI355X-native spherical-harmonic synthesis / analysis, covariance propagation and filter
engine behind the API surface of akvas/grates (PotentialCoefficients / GeographicGrid / kernel /
filter.DDK).  Hot loops are hand-written HIP kernels for gfx950 reached through the C ABI in
include/shg.h; see DESIGN.md.
"""

from . import data
from . import engine
from . import utilities
from . import kernel
from . import gravityfield
from . import grid
from . import filter
from . import lstsq
from . import io
from . import extras

__all__ = ['data', 'engine', 'extras', 'filter', 'gravityfield', 'grid', 'io', 'kernel', 'lstsq', 'utilities']
__version__ = '0.1.0'
