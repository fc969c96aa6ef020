"""
ctypes binding of libshg.so (include/shg.h).  There is no CPU fallback: if the library is missing or a
call fails, an exception is raised.
"""

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libshg.so')      # the one library the package loads; no environment override

c_double_p = ctypes.c_void_p     # device or host pointer passed as integer address
c_plan_p = ctypes.c_void_p


class ShgError(RuntimeError):
    """A libshg call returned a non-zero status."""

    def __init__(self, function, status, message):
        super().__init__('{0} failed with status {1}: {2}'.format(function, status, message))
        self.function = function
        self.status = status


# name -> argument types (all functions return int, except the two string getters)
PROTOTYPES = {
    'shg_plan_create': [ctypes.POINTER(c_plan_p), ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, ctypes.c_int, c_double_p, ctypes.c_int],
    'shg_plan_destroy': [c_plan_p],
    'shg_plan_set_chunk': [c_plan_p, ctypes.c_int],
    'shg_plan_set_path': [c_plan_p, ctypes.c_int],
    'shg_plan_set_rotations': [c_plan_p, ctypes.c_int],
    'shg_plan_set_stage_limit': [c_plan_p, ctypes.c_int],
    'shg_plan_info': [c_plan_p, ctypes.POINTER(ctypes.c_int64)],
    'shg_analysis_info': [c_plan_p, c_double_p],
    'shg_plan_profile': [c_plan_p, ctypes.c_int],
    'shg_plan_profile_read': [c_plan_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)],
    'shg_synthesis': [c_plan_p, c_double_p, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_synthesis_points': [ctypes.c_int, c_double_p, c_double_p, c_double_p, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_legendre': [ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_legendre_order': [ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_trigonometric': [ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_synthesis_matrix_order': [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, ctypes.c_int,
                                   c_double_p, c_double_p, ctypes.c_void_p],
    'shg_ravel': [c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_unravel': [c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_covprop_diag': [c_plan_p, c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_covprop_diag_symmetric': [c_plan_p, c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_covprop_diag_separable': [c_plan_p, c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_covprop_diag_separable_symmetric': [c_plan_p, c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_epoch_rms': [c_double_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_longlong, c_double_p, ctypes.c_void_p],
    'shg_symmetry_defect': [c_double_p, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_covprop_points': [ctypes.c_int, c_double_p, c_double_p, c_double_p, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_degree_scale': [c_double_p, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_orderwise_filter': [c_double_p, c_double_p, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_order_major_pack': [c_double_p, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_void_p],
    'shg_order_major_unpack': [c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_orderwise_filter_om': [c_double_p, c_double_p, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_degree_scale_om': [c_double_p, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_synthesis_om': [c_plan_p, c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_ddk_blocks': [c_double_p, c_double_p, ctypes.c_int, c_double_p, c_double_p, c_double_p, ctypes.c_void_p],
    'shg_dense_filter': [c_double_p, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_spd_solve': [c_double_p, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_dgemm': [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_void_p],
    'shg_gemm': [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_double_p, ctypes.c_int, c_double_p, ctypes.c_int,
                 ctypes.c_double, c_double_p, ctypes.c_int, ctypes.c_void_p],
    'shg_axpby': [ctypes.c_int, ctypes.c_int, ctypes.c_double, c_double_p, ctypes.c_int, ctypes.c_double, c_double_p, ctypes.c_int, ctypes.c_void_p],
    'shg_transpose_in_place': [ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_void_p],
    'shg_potrf': [ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p],
    'shg_trtri': [ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_void_p],
    'shg_analysis': [c_plan_p, c_double_p, c_double_p, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_synthesis_matrix': [ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, c_double_p, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_analysis_matrix': [c_plan_p, c_double_p, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_scratch_release': [],
    'shg_block_potrf': [ctypes.c_int] + [ctypes.c_void_p] * 5 + [c_double_p, ctypes.c_void_p],
    'shg_block_potrf_rows': [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_block_potrf_rows_pair': [ctypes.c_int] + [ctypes.c_void_p] * 7 + [ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_void_p],
    'shg_block_set_lookahead': [ctypes.c_int],
    'shg_block_lookahead_info': [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)],
    'shg_block_solve': [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p],
    'shg_block_sparse_inverse': [ctypes.c_int] + [ctypes.c_void_p] * 6,
    'shg_block_solve_rows': [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p],
    'shg_block_sparse_inverse_rows': [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p],
    'shg_block_inverse': [ctypes.c_int] + [ctypes.c_void_p] * 6,
    'shg_block_multiply': [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_void_p],
    'shg_congruence': [ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, ctypes.c_void_p],
}
STRING_GETTERS = ('shg_last_error', 'shg_version')

_lib = None


def use_library(path):
    """Profiling tools only (tools/timeline.py, tools/gemm_phases.py): load an instrumented build (`make timeline`) instead of
    libshg.so.  Must be called before the first library call of the process; the product never calls it and no environment
    variable selects a library."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError('libshg is already loaded from ' + LIB_PATH)
    LIB_PATH = os.path.abspath(path)


def load():
    """Load libshg.so (once).  Raises ImportError with build instructions when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError('libshg.so not found at {0}: build it with `make -C grates_amd/csrc` or '
                          '`python -c "import __graft_entry__ as g; g.build()"`. There is no CPU fallback.'.format(LIB_PATH))
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)            # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    for name in STRING_GETTERS:
        getattr(lib, name).restype = ctypes.c_char_p
        getattr(lib, name).argtypes = []
    _lib = lib
    return lib


def call(name, *args):
    """Call an int-returning libshg function and raise ShgError on failure."""
    lib = load()
    status = getattr(lib, name)(*args)
    if status != 0:
        raise ShgError(name, status, lib.shg_last_error().decode())
    return status


def version():
    return load().shg_version().decode()
