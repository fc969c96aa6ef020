"""
Per-wave stage timeline of the fused synthesis kernel (headline workload: 240 x d/o 96 -> 0.25 deg).

Needs the instrumented library (`make -C grates_amd/csrc timeline`), in which every wave stamps wall_clock64() (100 MHz)
when it starts its tile, finishes the Legendre stage, leaves the barrier, finishes the MFMA bodies of a column block and
has issued the stores of that column block.  Prints the mean length of every stage, the raw timeline of three tiles and
how many waves of the chip are in an epilogue / in the Legendre stage over time.

    python3 tools/timeline.py            (on a GPU box; SHG_DEBUG experiment switches are honoured)
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grates_amd import _lib
if '--release-library' not in sys.argv:
    _lib.use_library(os.path.join(ROOT, 'grates_amd', 'lib', 'libshg_timeline.so'))      # instrumented build (make timeline)
sys.argv = [a for a in sys.argv if a != '--release-library']
import numpy as np, torch
import grates_amd as ga
GM, R = 3.9860044150e+14, 6.3781363000e+06
grid = ga.grid.GeographicGrid(0.25, 0.25)
colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel('ewh'), 96, grid.parallels, GM, R, grid.semimajor_axis, grid.flattening)
plan = ga.engine.Plan(96, colat, kn, grid.meridians)
batch = torch.from_numpy(np.random.default_rng(0).standard_normal((240, 97, 97)) * 1e-10).cuda()
out = torch.empty((240, 720, 1440), dtype=torch.float64, device='cuda')
ntiles = 60 * 45
tl = torch.zeros((ntiles, 8, 16), dtype=torch.int64, device='cuda')      # 8 waves per workgroup (synthesis_rot.hip: SHG_ROT_WAVES)
for _ in range(5): plan.synthesis(batch, out=out)
torch.cuda.synchronize()
os.environ['SHG_TIMELINE_PTR'] = str(tl.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): plan.synthesis(batch, out=out)
e1.record()
torch.cuda.synchronize()
print('avg ms per call (30 calls)', e0.elapsed_time(e1) / 30)
del os.environ['SHG_TIMELINE_PTR']
t = tl.cpu().numpy().astype(np.float64)
if os.environ.get('SHG_TIMELINE_SAVE'):
    np.save(os.environ['SHG_TIMELINE_SAVE'], tl.cpu().numpy())
if (t[:, :, 13] > 0).any():    # shader-clock stamps beside the wall-clock ones at tile start / end: the clock the kernel runs at
    dc = t[:, :, 14] - t[:, :, 13]
    dw = (t[:, :, 12] - t[:, :, 0]) / 100.0
    ok = (t[:, :, 13] > 0) & (t[:, :, 14] > 0) & (dw > 5)
    mhz = dc[ok] / dw[ok]
    print('in-kernel clock MHz: median %.0f p10 %.0f p90 %.0f' % (np.median(mhz), np.percentile(mhz, 10), np.percentile(mhz, 90)))
    t[:, :, 13:] = 0
t0 = t[t > 0].min()
us = (t - t0) / 100.0          # 100 MHz counter
us[t == 0] = np.nan
print('kernel span us', np.nanmax(us))
ph1 = us[:, :, 1] - us[:, :, 0]
bar = us[:, :, 2] - us[:, :, 1]
print('phase1 per wave: mean %.2f  p10 %.2f p90 %.2f max %.2f' % (np.nanmean(ph1), np.nanpercentile(ph1, 10), np.nanpercentile(ph1, 90), np.nanmax(ph1)))
print('barrier wait: mean %.2f max %.2f;  phase1 incl barrier (tile level): mean %.2f' % (np.nanmean(bar), np.nanmax(bar), np.nanmean(us[:, :, 2].max(1) - us[:, :, 0].min(1))))
prev = us[:, :, 2]
for c in range(4):             # units of a wave (rotation-folded kernel: 4 column tiles per wave at 0.25 degree)
    M = us[:, :, 3 + 2 * c] - prev
    E = us[:, :, 4 + 2 * c] - us[:, :, 3 + 2 * c]
    print('cb %d: M mean %.2f p10 %.2f p90 %.2f | E mean %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f' % (c, np.nanmean(M), np.nanpercentile(M, 10), np.nanpercentile(M, 90),
          np.nanmean(E), np.nanpercentile(E, 10), np.nanpercentile(E, 50), np.nanpercentile(E, 90), np.nanmax(E)))
    prev = np.where(np.isnan(us[:, :, 4 + 2 * c]), us[:, :, 3 + 2 * c], us[:, :, 4 + 2 * c])
tile_t = np.nanmax(us[:, :, 12], axis=1) - np.nanmin(us[:, :, 0], axis=1)
print('tile total: mean %.2f p10 %.2f p90 %.2f' % (tile_t.mean(), np.percentile(tile_t, 10), np.percentile(tile_t, 90)))
T = np.arange(0, np.nanmax(us), 4.0)
def count(a, b):
    return [(np.nan_to_num(a, nan=1e30) <= x).sum() - (np.nan_to_num(b, nan=1e30) <= x).sum() for x in T]
inE = sum(np.array(count(us[:, :, 3 + 2 * c], us[:, :, 4 + 2 * c])) for c in range(4))
inP1 = np.array(count(us[:, :, 0], us[:, :, 2]))
print('t(us)   waves in epilogue   waves in phase1 (of 2048)')
for x, a, b in list(zip(T, inE, inP1))[::3]:
    print('%6.0f %6d %6d' % (x, a, b))
np.set_printoptions(linewidth=200, precision=1, suppress=True)
for tile in (300, 1301, 2002):
    base = np.nanmin(us[tile, :, 0])
    print('tile', tile, 'start', round(base, 1), ' rows = waves; cols = start, p1 end, barrier, [M end, E end] x 3, done')
    print(us[tile][:, [0, 1, 2, 3, 4, 5, 6, 7, 8, 12]] - base)
