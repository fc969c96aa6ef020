"""
One-off converter (build container only): reads the published ak135 load Love number table shipped with
the reference package (Wang et al. 2012, doi:10.1016/j.cageo.2012.06.022) and stores degrees 0..4096 of the
(h, l, k) columns as a compact .npz.  Data, not code; the reference loads the same table in
grates/data/__init__.py:48-49.

    python grates_amd/data/make_love_numbers.py
"""
import os
import numpy as np

SRC = '/root/reference/grates/data/ak135-LLNs-complete.dat.gz'
MAX_DEGREE = 4096

hlk = np.loadtxt(SRC, skiprows=1, usecols=(1, 2, 3), max_rows=MAX_DEGREE)
hlk = np.vstack((np.zeros((1, 3)), hlk))          # degree 0 row
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'love_numbers_ak135.npz')
np.savez_compressed(out, h=hlk[:, 0], l=hlk[:, 1], k=hlk[:, 2])
print(out, hlk.shape, os.path.getsize(out))
