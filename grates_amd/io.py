"""
Real-data feeders with the interface of ``grates.io``: ICGEM GFC files (grates/io.py:130-164),
GRACE / GRACE-FO SDS GSM files (grates/io.py:995-1043) and a loader that turns a list of monthly files into a
``TimeSeries`` ready for the batched GPU paths (``TimeSeries.to_grid``, ``filter_batch``).

Parsing is host work (text files of a few hundred KB); the numbers then live in ``PotentialCoefficients`` /
``TimeSeries`` exactly as if they had been filled by hand.  SINEX normal equations, TN-13/TN-14 replacement
files and the mascon readers (netCDF / HDF5) are not covered.
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

__all__ = ['loadgfc', 'loadgsm', 'load_time_series']


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
