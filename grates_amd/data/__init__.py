"""
Package data: ak135 load Love numbers (degrees 0..4096 of the table the reference ships,
grates/data/__init__.py:12-99).
"""

import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'love_numbers_ak135.npz')
_cache = {}


def _table():
    if 'hlk' not in _cache:
        with np.load(_PATH) as f:
            _cache['hlk'] = np.vstack((f['h'], f['l'], f['k'])).T.copy()
    return _cache['hlk']


def import_load_love_numbers(max_degree=None, frame='CE'):
    """
    Load Love numbers (k, h, l) of the elastic Earth model ak135 in frame CE, CM or CF
    (degree-1 terms differ between frames, grates/data/__init__.py:51-62).
    """
    if max_degree is not None and max_degree < 1:
        return np.zeros(1), np.zeros(1), np.zeros(1)
    hlk = _table().copy()
    if max_degree is not None:
        if max_degree + 1 > hlk.shape[0]:
            raise ValueError('load Love numbers are shipped up to degree {0:d} (requested {1:d})'.format(hlk.shape[0] - 1, max_degree))
        hlk = hlk[0:max_degree + 1]
    key = frame.lower()
    if key == 'cm':
        hlk[1, :] -= 1
    elif key == 'cf':
        h1, l1 = hlk[1, 0], hlk[1, 1]
        hlk[1, 0] = (h1 - l1) * 2 / 3
        hlk[1, 1] = (h1 - l1) * -1 / 3
        hlk[1, 2] = (-1 / 3 * h1 - 2 / 3 * l1)
    elif key != 'ce':
        raise ValueError('frame of load love numbers must be one of CM, CE, or CF (got <' + frame + '>)')
    return hlk[:, 2], hlk[:, 0], hlk[:, 1]


def load_love_numbers(max_degree=None, frame='CE'):
    """Return (k, h, l) in the requested frame (cached per frame)."""
    key = frame.lower()
    if key not in ('cm', 'ce', 'cf'):
        raise ValueError('frame of load love numbers must be one of CM, CE, or CF (got <' + frame + '>)')
    if key not in _cache:
        _cache[key] = import_load_love_numbers(frame=frame)
    return _cache[key]


def ddk_normal_blocks():
    """
    Order-wise DDK normal-equation blocks.  The reference reads them from a data blob
    (grates/data/__init__.py:102-117) that is not redistributed here; point the environment variable
    GRATES_DDK_NORMAL_BLOCKS at a ddk_normal_blocks.npz with the reference's keys
    (order0_cos, order1_cos, order1_sin, ...).
    """
    path = os.environ.get('GRATES_DDK_NORMAL_BLOCKS', os.path.join(os.path.dirname(_PATH), 'ddk_normal_blocks.npz'))
    if not os.path.exists(path):
        raise FileNotFoundError('DDK normal blocks not found at {0}; set GRATES_DDK_NORMAL_BLOCKS'.format(path))
    with np.load(path) as f:
        blocks = [f['order0_cos']]
        m = 1
        while 'order{0:d}_cos'.format(m) in f.files:
            blocks.append(f['order{0:d}_cos'.format(m)])
            blocks.append(f['order{0:d}_sin'.format(m)])
            m += 1
        return blocks
