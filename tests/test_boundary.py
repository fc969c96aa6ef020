"""
CPU checks of the drop-in boundary: libshg.so loads and exports every symbol include/shg.h declares, the
product package never touches oracle/, and compute entry points fail loudly without a GPU.
"""

import ast
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'shg.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(shg_[a-z_0-9]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from grates_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = header_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), 'libshg.so does not export ' + name
    # the ctypes prototype table covers the same set
    assert sorted(list(_lib.PROTOTYPES) + list(_lib.STRING_GETTERS)) == names
    assert 'gfx950' in _lib.version()


def test_error_reporting_without_gpu_calls():
    from grates_amd import _lib
    lib = _lib.load()
    # argument validation happens before any HIP call
    assert lib.shg_plan_set_chunk(None, 4) == -1
    assert b'NULL plan' in lib.shg_last_error()
    with pytest.raises(_lib.ShgError) as err:
        _lib.call('shg_legendre_order', 3, 5, None, 1, None, None)
    assert 'order exceeds maximum degree' in str(err.value)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'grates_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if not fn.endswith('.py'):
                continue
            tree = ast.parse(open(os.path.join(dirpath, fn)).read())
            for node in ast.walk(tree):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    names = [node.module or '']
                for n in names:
                    assert not n.startswith('oracle') and 'shg_oracle' not in n, '{0} imports {1}'.format(fn, n)
    for dirpath, _, files in os.walk(os.path.join(pkg, 'csrc')):
        for fn in files:
            if fn.endswith(('.hip', '.h', '.cpp')):
                assert 'oracle' not in open(os.path.join(dirpath, fn)).read()


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    import grates_amd as ga
    gf = ga.gravityfield.PotentialCoefficients(max_degree=4)
    with pytest.raises(RuntimeError, match='no GPU'):
        gf.to_grid(ga.grid.GeographicGrid(30, 30), 'potential')
    with pytest.raises(RuntimeError, match='no GPU'):
        ga.utilities.legendre_functions(4, np.array([0.3]))
    with pytest.raises(RuntimeError, match='no GPU'):
        ga.filter.Gaussian(300).filter(gf)


def test_header_is_plain_c(tmp_path):
    """include/shg.h is the C ABI: it must compile as C without any HIP / C++ header."""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('gcc not available')
    src = tmp_path / 'use_shg.c'
    src.write_text('#include "shg.h"\nint main(void) { shg_plan* p = 0; return shg_plan_destroy(p) + (int)sizeof(shg_status) * 0; }\n')
    subprocess.run([gcc, '-std=c99', '-Wall', '-Werror', '-fsyntax-only', '-I', os.path.join(ROOT, 'include'), str(src)], check=True)


def test_shipping_library_has_no_experiment_switches(monkeypatch):
    """The timing knock-outs of the profiling builds (SHG_DEBUG, SHG_ROT_X, SHG_GEMM_X, SHG_ANA_X) are compiled out of
    libshg.so: it reads no environment variable at all, and the package loads exactly one library whatever SHG_LIBRARY says."""
    import subprocess
    from grates_amd import _lib
    undefined = subprocess.run(['nm', '-D', '--undefined-only', _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert 'getenv' not in undefined
    blob = open(_lib.LIB_PATH, 'rb').read()
    for name in (b'SHG_DEBUG', b'SHG_TIMELINE_PTR', b'SHG_LIBRARY'):
        assert name not in blob, name
    monkeypatch.setenv('SHG_LIBRARY', '/nonexistent/libother.so')
    import importlib
    fresh = importlib.reload(_lib)
    try:
        assert fresh.LIB_PATH == os.path.join(ROOT, 'grates_amd', 'lib', 'libshg.so')
    finally:
        monkeypatch.delenv('SHG_LIBRARY')
        importlib.reload(_lib)
    for fn in os.listdir(os.path.join(ROOT, 'grates_amd', 'csrc')):
        if fn.endswith(('.hip', '.h')):
            text = open(os.path.join(ROOT, 'grates_amd', 'csrc', fn)).read()
            for m in re.finditer(r'getenv\(', text):
                guard = text.rfind('#ifdef SHG_', 0, m.start())
                assert guard >= 0 and text[guard:m.start()].count('#endif') == 0, fn + ' calls getenv outside an experiment / timeline build'


def test_block_calls_reject_null_inverse_table():
    """A NULL table of inverse scratch matrices is an argument error, not a segfault (checked before any HIP call)."""
    from grates_amd import _lib
    lib = _lib.load()
    bounds = (ctypes.c_int * 2)(0, 3)
    rowptr = (ctypes.c_int * 2)(0, 1)
    colidx = (ctypes.c_int * 1)(0)
    blocks = (ctypes.c_uint64 * 1)(0x1000)
    as_p = lambda a: ctypes.cast(a, ctypes.c_void_p)      # noqa: E731
    for name, tail in (('shg_block_solve', (0, None, 1, 1, None)), ('shg_block_sparse_inverse', (None,)), ('shg_block_inverse', (None,))):
        status = getattr(lib, name)(1, as_p(bounds), as_p(rowptr), as_p(colidx), as_p(blocks), None, *tail)
        assert status == -1 and b'inverses' in lib.shg_last_error(), name
