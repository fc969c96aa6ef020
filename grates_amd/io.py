"""
Real-data feeders with the interface of ``grates.io``: ICGEM GFC files (grates/io.py:130-164),
GRACE / GRACE-FO SDS GSM files (grates/io.py:995-1043) and a loader that turns a list of monthly files into a
``TimeSeries`` ready for the batched GPU paths (``TimeSeries.to_grid``, ``filter_batch``).

Parsing is host work (text files of a few hundred KB); the numbers then live in ``PotentialCoefficients`` /
``TimeSeries`` exactly as if they had been filled by hand.  SINEX normal equations (grates/io.py:686-875) are read
into plain arrays (`loadsinexnormals`) or straight into a ``lstsq.NormalEquations`` whose solve runs on the GPU
(`load_normal_equations`).  TN-13/TN-14 replacement files, the SINEX writer and the mascon readers (netCDF / HDF5)
are not covered.
"""

import bz2
import contextlib
import datetime
import gzip
import io
import os

import numpy as np
import yaml

from .gravityfield import PotentialCoefficients, TimeSeries

__all__ = ['loadgfc', 'loadgsm', 'load_time_series', 'loadsinex', 'loadsinexnormals', 'load_normal_equations']


@contextlib.contextmanager
def _binary_lines(source):
    """Lines (bytes) of a file name (.gz / .bz2 are decompressed on the fly), path object or open binary / text stream."""
    if isinstance(source, os.PathLike):
        source = os.fspath(source)
    owned = isinstance(source, str)
    if owned:
        opener = gzip.open if source.endswith('.gz') else (bz2.open if source.endswith('.bz2') else open)
        stream = opener(source, 'rb')
    elif isinstance(source, (io.BufferedIOBase, io.TextIOBase)):
        stream = source
    else:
        raise ValueError('file_name must be a string, PathLike object or file object')
    if not stream.readable():
        raise ValueError('file stream must be readable')
    text = isinstance(stream, io.TextIOBase)

    def lines():
        for line in stream:
            yield line.encode(stream.encoding or 'utf-8') if text else line
    try:
        yield lines()
    finally:
        if owned:
            stream.close()


def loadgfc(file_name, max_degree=None):
    """
    Potential coefficients from an ICGEM GFC file: ``gfc n m C S`` records, ``radius`` and ``earth_gravity_constant``
    header keys; defaults GM = 3.986004415e14, R = 6378136.3 (grates/io.py:144-164).  Degrees above `max_degree`
    are skipped.
    """
    field = PotentialCoefficients(3.986004415E+14, 6378136.3)
    with _binary_lines(file_name) as lines:
        for line in lines:
            if line.startswith(b'gfc'):
                token = line.split()
                degree, order = int(token[1]), int(token[2])
                if max_degree and degree > max_degree:
                    continue
                field.append('c', degree, order, float(token[3]))
                field.append('s', degree, order, float(token[4]))
            elif line.startswith(b'radius'):
                field.R = float(line.split()[-1])
            elif line.startswith(b'earth_gravity_constant'):
                field.GM = float(line.split()[-1])
    return field


def loadgsm(file_name):
    """
    Potential coefficients from a GRACE / GRACE-FO SDS level-2 file (GSM / GAA-GAD; YAML header, ``GRCOF2`` records).
    GM, R and the maximum degree come from the header.  Like the reference (grates/io.py:1021-1023, where start and
    end are both read from ``time_coverage_start``) the epoch is the start of the data coverage.
    """
    with _binary_lines(file_name) as lines:
        header = b''
        for line in lines:
            if line.startswith(b'# End of YAML header'):
                break
            header += line
        meta = yaml.safe_load(header)['header']
        max_degree = meta['dimensions']['degree']
        attributes = meta['non-standard_attributes']
        field = PotentialCoefficients(attributes['earth_gravity_param']['value'], attributes['mean_equator_radius']['value'])
        anm = np.zeros((max_degree + 1, max_degree + 1))
        for line in lines:
            if line.startswith(b'GRCOF2'):
                token = line.split()
                degree, order = int(token[1]), int(token[2])
                anm[degree, order] = float(token[3])
                if order > 0:
                    anm[order - 1, degree] = float(token[4])
        field.anm = anm
        field.epoch = meta['global_attributes']['time_coverage_start']
    return field


def load_time_series(file_names, loader=loadgsm, epochs=None, max_degree=None):
    """
    A list of monthly files as one ``TimeSeries`` (sorted by epoch): the batch the GPU paths take in one call, e.g.
    ``load_time_series(files).to_grid(grid, 'ewh')``.  `epochs` supplies the epochs for formats without one (GFC);
    `max_degree` truncates every field.
    """
    fields = []
    for k, name in enumerate(file_names):
        field = loader(name)
        if epochs is not None:
            field.epoch = epochs[k]
        if field.epoch is None:
            raise ValueError('{0}: the file carries no epoch; pass epochs='.format(name))
        if isinstance(field.epoch, datetime.date) and not isinstance(field.epoch, datetime.datetime):
            field.epoch = datetime.datetime(field.epoch.year, field.epoch.month, field.epoch.day)
        if max_degree is not None:
            field.truncate(max_degree)
        fields.append(field)
    return TimeSeries(fields)


class SinexBlock:
    """One recognised block of a SINEX file: `block_type` (bytes, without the leading '+' and without the triangle flag of
    matrix blocks, as upstream) plus the fields of its kind --
    vectors: `x`, `sigmax` (None for right-hand sides), `basis` ('CN' / 'SN' per record), `degree`, `order`;
    matrices: `matrix` (symmetric, both triangles filled); statistics: `degrees_of_freedom`, `observation_count`,
    `parameters`, `observation_square_sum`."""

    def __init__(self, block_type, **fields):
        self.block_type = block_type
        self.__dict__.update(fields)

    def parameter_count(self):
        return len(self.x) if hasattr(self, 'x') else None


def _block_records(lines):
    """data records of the block the iterator is inside of (comment lines skipped), up to its '-' line"""
    for line in lines:
        if not line or line.startswith(b'*'):
            continue
        if line.startswith(b'-'):
            return
        yield line


_VECTOR_BLOCKS = (b'+SOLUTION/ESTIMATE', b'+SOLUTION/APRIORI', b'+SOLUTION/NORMAL_EQUATION_VECTOR')
_MATRIX_BLOCKS = (b'+SOLUTION/NORMAL_EQUATION_MATRIX', b'+SOLUTION/MATRIX_ESTIMATE')
_STATISTICS = ((b'NUMBER OF DEGREES OF FREEDOM', 'degrees_of_freedom', True), (b'NUMBER OF OBSERVATIONS', 'observation_count', True),
               (b'NUMBER OF UNKNOWNS', 'parameters', True), (b'WEIGHTED SQUARE SUM OF O-C', 'observation_square_sum', False))


def _sinex_vector(block_type, lines):
    """fixed-column parameter records: type in columns 7-12, degree 14-17, order 22-25, value 47-67, sigma 69-79
    (grates/io.py:520-553)"""
    right_hand_side = block_type.startswith(b'SOLUTION/NORMAL_EQUATION_VECTOR')
    basis, degree, order, x, sigma = [], [], [], [], []
    for line in _block_records(lines):
        kind = line[7:13].strip()
        if kind not in (b'CN', b'SN'):
            raise ValueError('Parameter type <' + kind.decode() + '> not supported.')
        basis.append(kind.decode())
        degree.append(int(line[14:18]))
        order.append(int(line[22:26]))
        x.append(float(line[47:68]))
        if not right_hand_side:
            sigma.append(float(line[69:80]))
    return SinexBlock(block_type, x=np.array(x), sigmax=np.array(sigma) if sigma else None, basis=basis,
                      degree=np.array(degree, dtype=int), order=np.array(order, dtype=int))


def _sinex_matrix(block_type, lines, parameter_count):
    """records `row column v0 [v1 [v2]]` (1-based, one triangle); the other triangle is mirrored (grates/io.py:618-648)"""
    if parameter_count is None:
        raise ValueError('SINEX matrix block before any parameter vector block: parameter count unknown')
    rows, cols, vals = [], [], []
    for line in _block_records(lines):
        token = line.split()
        r, c0 = int(token[0]) - 1, int(token[1]) - 1
        for k, v in enumerate(token[2:]):
            rows.append(r)
            cols.append(c0 + k)
            vals.append(float(v))
    rows, cols = np.array(rows, dtype=int), np.array(cols, dtype=int)
    size = max(parameter_count, int(rows.max()) + 1 if len(rows) else 0, int(cols.max()) + 1 if len(cols) else 0)
    matrix = np.zeros((size, size))
    matrix[rows, cols] = vals
    matrix[cols, rows] = vals
    return SinexBlock(block_type, matrix=matrix)


def _sinex_statistics(block_type, lines):
    fields = {}
    for line in _block_records(lines):
        for label, name, integer in _STATISTICS:
            if line[1:].startswith(label):
                fields[name] = int(float(line[32:])) if integer else float(line[32:])
    missing = [name for _, name, _ in _STATISTICS if name not in fields]
    if missing:
        raise ValueError('SINEX statistics block lacks ' + ', '.join(missing))
    return SinexBlock(block_type, **fields)


def loadsinex(file_name):
    """
    The recognised blocks of a SINEX file as a list (grates/io.py:686-722): parameter vectors (estimate, a-priori,
    right-hand side), symmetric matrices (normal matrix, covariance of the estimate) and the statistics block; other
    blocks are skipped.  Matrix blocks are sized by the first parameter vector that precedes them.
    """
    blocks = []
    parameter_count = None
    with _binary_lines(file_name) as lines:
        first = True
        for line in lines:
            line = line.rstrip()
            if first and line.startswith(b'%'):
                first = False
                continue
            first = False
            if not line or line.startswith(b'*'):
                continue
            if line.startswith(b'%'):
                break
            if not line.startswith(b'+'):
                continue
            if line.startswith(_VECTOR_BLOCKS):
                block = _sinex_vector(line[1:], lines)
            elif line.startswith(_MATRIX_BLOCKS):
                block = _sinex_matrix(line[1:-2], lines, parameter_count)
            elif line.startswith(b'+SOLUTION/STATISTICS'):
                block = _sinex_statistics(line[1:], lines)
            else:
                for _ in _block_records(lines):
                    pass
                continue
            if parameter_count is None:
                parameter_count = block.parameter_count()
            blocks.append(block)
    return blocks


def loadsinexnormals(file_name):
    """
    Normal equations of a SINEX file in storage scheme 6b / 6c (grates/io.py:725-760).

    Returns N [p, p], n [p, 1], lPl [1], obs_count.
    """
    blocks = {b.block_type: b for b in loadsinex(file_name)}
    needed = (b'SOLUTION/NORMAL_EQUATION_MATRIX', b'SOLUTION/NORMAL_EQUATION_VECTOR', b'SOLUTION/STATISTICS')
    if not all(name in blocks for name in needed):
        raise ValueError('SINEX file does not conform to storage schemes 6b or 6c for normal equations.')
    statistics = blocks[b'SOLUTION/STATISTICS']
    return (blocks[b'SOLUTION/NORMAL_EQUATION_MATRIX'].matrix, blocks[b'SOLUTION/NORMAL_EQUATION_VECTOR'].x[:, np.newaxis],
            np.atleast_1d(statistics.observation_square_sum), statistics.observation_count)


def load_normal_equations(file_name, block_size=2048):
    """SINEX normal equations as ``lstsq.NormalEquations`` (dense matrix cut into `block_size` blocks): `solve()` and
    `compute_covariance()` then run the blocked Cholesky kernels on the GPU (extension)."""
    from .lstsq import BlockMatrix, NormalEquations
    N, n, lPl, obs_count = loadsinexnormals(file_name)
    index = BlockMatrix.compute_block_index(N.shape, block_size)
    return NormalEquations(BlockMatrix.from_array(np.triu(N), *index), n, lPl, obs_count)     # upper blocks, as lstsq expects
