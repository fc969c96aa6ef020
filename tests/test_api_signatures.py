"""
The Python half of the drop-in boundary as a regression guard: every public callable of the reference's modules on the
path -- names, parameters (name, kind, default) and the exception type of a set of probes, recorded FROM THE IMPORTED REFERENCE in
tests/golden/g19_api.json by tests/golden/make_golden.py -- must exist in grates_amd with the same leading parameters; grates_amd may
only append parameters that have defaults (as_tensor=, parallel_range=, batch=, ...).  What the survey marks out of scope (section 2)
is listed by name below, so that a hot-path name that goes missing fails the test instead of joining a silent skip.
(/root/reference/grates/gravityfield.py:89-390, grid.py:510-839, 1123-1204, filter.py:31-509, utilities.py:13-459)
"""
import inspect
import json
import os
import re
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import inputs  # noqa: E402

import grates_amd as ga  # noqa: E402

with open(os.path.join(HERE, 'golden', 'g19_api.json')) as _f:
    API = json.load(_f)

# public names of the reference that SURVEY.md section 2 marks out of scope (not on the path north_star names); everything else must exist
OUT_OF_SCOPE = {
    'data': {'csr_rl06_mascon_grid', 'gsfc_rl06_mascon_grid'},                                   # mascon product grids (data files absent)
    'grid': {'Basin', 'CSRMasconGridRL06', 'GSFCMasconGridRL06', 'JPLMasconGridRL06', 'GeodesicGrid', 'SpiralGrid', 'GreatCircleSegment',
             'SurfaceElement', 'PolygonSurfaceElement', 'RectangularSurfaceElement', 'spherical_pib', 'spherical_pip', 'winding_number'},
    'io': {'InputFile', 'SINEXBlock', 'SINEXBlockPlaceholder', 'SINEXFile', 'SINEXSphericalHarmonicsVector', 'SINEXStatistics',
           'SINEXSymmetricMatrix', 'read_sinex_block',                                            # the (upstream-broken) SINEX writer's object model
           'loadcsr06mascons', 'loadgsfc06mascons', 'loadrl06mascongrids', 'loadesm', 'loadtn13', 'loadtn14'},      # netCDF / HDF5 / TN readers
    'lstsq': {'UnscentedTransformSymmetric', 'robust_least_squares', 'teigh', 'trsvd'},         # off the smoother path
}
# members of in-scope classes that are out of scope themselves, or upstream defects that cannot be mirrored
OUT_OF_SCOPE_MEMBERS = {
    'grid.Grid': {'create_mask', 'point_neighbours', 'voronoi_cells'},                     # convex hull / Voronoi geometry (scipy.spatial)
    'grid.RegularGrid': {'voronoi_cells'}, 'grid.IrregularGrid': {'voronoi_cells'},          # Basin / polygon geometry
    'kernel.IsotropicKernel': {'modulation_transfer', 'spatial_resolution'}, 'kernel.AnisotropicKernel': {'modulation_transfer', 'spatial_resolution'},
    'lstsq.AutoregressiveModel': {'from_transformed_coefficients'},
    # upstream: `@property def is_nonzero(self, row, column)` (lstsq.py:884-887) cannot be called; here it is the method the docstring describes
    'lstsq.BlockMatrix': {'is_nonzero'},
    # nested sort-key helper classes of the coefficient sequences (an implementation detail of their `sorted` calls)
    'gravityfield.CoefficientSequence': {'ComparableCoefficientSequence'}, 'gravityfield.CoefficientSequenceDegreeWise': {'Comparable'},
    'gravityfield.CoefficientSequenceFlatArray': {'Comparable'}, 'gravityfield.CoefficientSequenceOrderWise': {'Comparable'},
    'gravityfield.CoefficientSequenceOrderWiseAlternating': {'Comparable'},
}


def params_of(obj):
    """as tests/golden/make_golden.py::g19 records them"""
    out = []
    for prm in inspect.signature(obj).parameters.values():
        default = None
        if prm.default is not inspect.Parameter.empty:
            default = re.sub(r' at 0x[0-9a-f]+', '', repr(prm.default))
            if ' object>' in default:
                default = '<instance of {0}>'.format(type(prm.default).__name__)
        out.append([prm.name, prm.kind.name, default])
    return out


def compatible(ref, own):
    """None when `own` accepts every call `ref` accepts with the same meaning, else the reason"""
    if len(own) < len(ref):
        return 'fewer parameters: {0} vs reference {1}'.format([p[0] for p in own], [p[0] for p in ref])
    for r, o in zip(ref, own):
        if r[0] != o[0] or r[1] != o[1]:
            return 'parameter {0} ({1}) where the reference has {2} ({3})'.format(o[0], o[1], r[0], r[1])
        if (r[2] is None) != (o[2] is None):
            return 'parameter {0}: default {1} where the reference has {2}'.format(o[0], o[2], r[2])
        # an instance default built at import time (grid=GeographicGrid()) may be None here and built per call
        if r[2] is not None and r[2] != o[2] and not r[2].startswith('<instance of'):
            return 'parameter {0}: default {1} where the reference has {2}'.format(o[0], o[2], r[2])
    for extra in own[len(ref):]:
        if extra[2] is None and extra[1] not in ('VAR_POSITIONAL', 'VAR_KEYWORD'):
            return 'appended parameter {0} has no default'.format(extra[0])
    return None


def cases():
    for modname, entries in sorted(API['modules'].items()):
        for name, rec in sorted(entries.items()):
            yield modname, name, rec


@pytest.mark.parametrize('modname,name,rec', list(cases()), ids=lambda v: v if isinstance(v, str) else '')
def test_public_callable_matches_reference(modname, name, rec):
    if name in OUT_OF_SCOPE.get(modname, ()):
        pytest.skip('out of scope by SURVEY.md section 2')
    mod = getattr(ga, modname)
    assert hasattr(mod, name), '{0}.{1} of the reference is missing'.format(modname, name)
    obj = getattr(mod, name)
    if rec['kind'] == 'function':
        why = compatible(rec['signature'], params_of(obj))
        assert why is None, '{0}.{1}: {2}'.format(modname, name, why)
        return
    assert inspect.isclass(obj)
    skip = OUT_OF_SCOPE_MEMBERS.get(modname + '.' + name, ())
    for mname, mrec in sorted(rec['members'].items()):
        if mname in skip:
            continue
        assert hasattr(obj, mname), '{0}.{1}.{2} of the reference is missing'.format(modname, name, mname)
        member = inspect.getattr_static(obj, mname)
        if isinstance(mrec, str):                                   # property (+setter)
            assert isinstance(member, property), '{0}.{1}.{2} is a property in the reference'.format(modname, name, mname)
            assert ('+setter' in mrec) <= (member.fset is not None), '{0}.{1}.{2}: the reference property has a setter'.format(modname, name, mname)
            continue
        if mrec is None:
            continue
        if mrec and isinstance(mrec[0], str):                      # ['staticmethod' | 'classmethod', params...]
            assert type(member).__name__ == mrec[0], '{0}.{1}.{2} is a {3} in the reference'.format(modname, name, mname, mrec[0])
            ref, own = mrec[1:], params_of(member.__func__)
        else:
            if mname == '__init__' and '__init__' not in vars(obj) and mrec == [['self', 'POSITIONAL_OR_KEYWORD', None]]:
                continue                                                # the reference's `def __init__(self): pass` against an inherited one
            ref, own = mrec, params_of(getattr(obj, mname))
        why = compatible(ref, own)
        assert why is None, '{0}.{1}.{2}: {3}'.format(modname, name, mname, why)


@pytest.mark.parametrize('label', sorted(API['raises']))
def test_error_contract_matches_reference(label):
    """the same probe on grates_amd raises the exception TYPE the reference raised (or nothing where it raised nothing)"""
    if label in GPU_PROBES:
        pytest.skip('this probe reaches a kernel: replayed under -m gpu (tests/test_gpu_edge_cases.py)')
    thunk = dict(inputs.api_probes(ga))[label]
    expected = API['raises'][label]
    try:
        thunk()
        got = None
    except Exception as err:        # noqa: BLE001
        got = type(err).__name__
    assert got == expected, '{0}: raised {1}, the reference raises {2}'.format(label, got, expected)


# probes whose non-raising outcome runs device code
GPU_PROBES = {'orderwise_filter_ok', 'legendre_scalar_colat'}
