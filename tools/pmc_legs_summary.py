"""Per-kernel averages of the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_legs.sh -> <out>/<tag>_pmc_traffic.json.
The counters are in KB.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced streaming reads
(16 B per lane, global loads and LDS-DMA alike) and WRITE_SIZE is exact for 16-byte-per-lane stores; other widths are uncalibrated --
both the raw and the doubled fetch figure are kept, `bytes` uses the doubled one (an upper bound where the kernel reads narrower)."""
import csv
import json
import subprocess
import os
import re
import sys


# Kernels whose reads are 8-byte gathers / scalar-width accesses: calibrated on their known byte counts, FETCH_SIZE counts them as they
# are (order-wise filter: 28.1 MB of coefficients + 9.4 MB of blocks per launch against 35.2 MB raw; ravel / unravel: strided 8-byte
# gathers).  Every other kernel of the path reads with 16-byte-per-lane coalesced loads or LDS-DMA: doubled, as the guide prescribes.
NARROW_READERS = ('orderwise_filter_kernel', 'ravel_kernel', 'unravel_kernel', 'gemv_rows_kernel', 'leaf_kernel', 'panel_kernel')


def main(out, tag):
    legs = ('synthesis', 'analysis', 'filters', 'covariance', 'smoother')
    table = {}
    for leg in legs:
        per = {}
        for c in ('FETCH_SIZE', 'WRITE_SIZE'):
            path = os.path.join(out, '{0}_pmc_{1}_{2}.csv'.format(tag, leg, c))
            if not os.path.exists(path):
                continue
            for r in csv.DictReader(open(path)):
                if r['Counter_Name'] != c:
                    continue
                name = re.sub(r'\(.*$', '', r['Kernel_Name']).replace('void ', '').replace('shg::', '').strip()
                if name.startswith('at::') or 'elementwise' in name or name.startswith('void at'):
                    continue
                ent = per.setdefault(name, {'FETCH_SIZE': [], 'WRITE_SIZE': []})
                ent[c].append(float(r['Counter_Value']))
        rows = {}
        for name, ent in per.items():
            f, w = ent['FETCH_SIZE'], ent['WRITE_SIZE']
            if not f or not w:
                continue
            fetch_kb, write_kb = sum(f) / len(f), sum(w) / len(w)
            factor = 1.0 if name.startswith(NARROW_READERS) else 2.0
            rows[name] = {'dispatches': len(f), 'FETCH_SIZE_KB_raw': fetch_kb, 'WRITE_SIZE_KB_raw': write_kb,
                          'fetch_bytes_raw': fetch_kb * 1024.0, 'fetch_bytes_doubled': fetch_kb * 2048.0, 'write_bytes': write_kb * 1024.0,
                          'fetch_factor': factor, 'bytes': fetch_kb * 1024.0 * factor + write_kb * 1024.0}
        table[leg] = rows
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import source_hashes
    summary = {'commit': source_hashes.commit(), 'sources': {leg: source_hashes.leg_hashes(leg) for leg in legs}, 'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes per leg (tools/pmc_legs.sh), averages per dispatch',
               'correction': 'counters in KB; fetch doubled (gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes for wide coalesced reads) except for the '
                             'kernels that gather 8-byte elements (fetch_factor 1, calibrated on their known byte counts); WRITE_SIZE as counted; '
                             'Infinity-Cache hits are counted',
               'legs': table}
    with open(os.path.join(out, tag + '_pmc_traffic.json'), 'w') as f:
        json.dump(summary, f, indent=1)
    for leg, rows in table.items():
        for name, r in sorted(rows.items(), key=lambda kv: -kv[1]['bytes'])[:6]:
            print('{0:11s} {1:60s} n={2:5d} fetch x2 {3:10.1f} MB  write {4:10.1f} MB'.format(leg, name[:60], r['dispatches'], r['fetch_bytes_doubled'] / 1e6, r['write_bytes'] / 1e6))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
