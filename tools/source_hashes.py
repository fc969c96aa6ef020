"""sha256 of the kernel sources behind every bench leg: written into the PMC summaries of profiles/ (tools/pmc_legs_summary.py, tools/pmc_mfma_busy.py)
and compared by bench.py, which reports counter figures of a summary only while the sources they were measured on are unchanged
(there is no git on the GPU box; the commit of the build container travels in tools/.commit, written by tools/stamp_commit.sh)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'grates_amd', 'csrc')
LEG_SOURCES = {
    'synthesis': ('common.h', 'synthesis_rot.hip', 'synthesis_fused.hip'),
    'analysis': ('common.h', 'analysis.hip'),
    'filters': ('common.h', 'filters.hip', 'gemm_tall.hip', 'blas.hip'),
    'filters_block': ('common.h', 'filters.hip'),
    'filters_dense': ('common.h', 'gemm_tall.hip', 'blas.hip'),
    'covariance': ('common.h', 'gemm.hip', 'covprop.hip'),
    'smoother': ('common.h', 'blas.hip', 'blockchol.hip'),
}


def file_hash(name):
    with open(os.path.join(CSRC, name), 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def leg_hashes(leg):
    return {name: file_hash(name) for name in LEG_SOURCES[leg]}


def all_hashes():
    return {leg: leg_hashes(leg) for leg in LEG_SOURCES}


def commit():
    try:
        with open(os.path.join(ROOT, 'tools', '.commit')) as f:
            return f.read().strip()
    except OSError:
        return 'unrecorded'


def unchanged(recorded, leg):
    """True when every source of `leg` still has the hash recorded in a summary ({file: hash})"""
    if not isinstance(recorded, dict) or not recorded:
        return False
    try:
        return all(file_hash(name) == digest for name, digest in recorded.items()) and set(recorded) >= set(LEG_SOURCES[leg])
    except OSError:
        return False
