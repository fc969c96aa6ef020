"""Per-wave stage timeline of the pipelined synthesis kernel (path 'pipe'; 240 x d/o 96 -> 0.25 deg), instrumented library
(`make -C grates_amd/csrc timeline`).  Stamps: 0 start, 1 Legendre stage done, 2 barrier passed, 3 .. 7 end of unit 0 .. 4, 12 done.
    python3 tools/timeline_pipe.py        (SHG_DEBUG knock-outs are honoured: 1 no stores, 2 no Legendre stage, 4 no longitude stage)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grates_amd import _lib
_lib.use_library(os.environ.get('SHG_LIB', os.path.join(ROOT, 'grates_amd', 'lib', 'libshg_timeline.so')))
import numpy as np, torch
import grates_amd as ga
grid = ga.grid.GeographicGrid(0.25, 0.25)
colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel('ewh'), 96, grid.parallels, 3.9860044150e+14, 6.3781363000e+06, grid.semimajor_axis, grid.flattening)
plan = ga.engine.Plan(96, colat, kn, grid.meridians)
plan.set_path(os.environ.get('SHG_PATH', 'pipe'))
batch = torch.from_numpy(np.random.default_rng(0).standard_normal((240, 97, 97)) * 1e-10).cuda()
out = torch.empty((240, 720, 1440), dtype=torch.float64, device='cuda')
tl = torch.zeros((60 * 45 * 8 * 16 + 16,), dtype=torch.int64, device='cuda')
for _ in range(50): plan.synthesis(batch, out=out)
torch.cuda.synchronize()
os.environ['SHG_TIMELINE_PTR'] = str(tl.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): plan.synthesis(batch, out=out)
e1.record()
torch.cuda.synchronize()
print('avg ms per call (30 calls) %.4f' % (e0.elapsed_time(e1) / 30))
del os.environ['SHG_TIMELINE_PTR']
print('panel images that did not arrive in time (30 calls x 2700 tiles):', int(tl[-16].item()))
t = tl[:-16].reshape(60 * 45, 8, 16).cpu().numpy().astype(np.float64)[:, :4, :]
ok = (t[:, :, 13] > 0) & (t[:, :, 14] > 0)
mhz = ((t[:, :, 14] - t[:, :, 13]) / ((t[:, :, 12] - t[:, :, 0]) / 100.0))[ok]
print('in-kernel clock MHz: median %.0f' % np.median(mhz))
t[:, :, 13:] = 0
t0 = t[t > 0].min()
us = (t - t0) / 100.0
us[t == 0] = np.nan
print('kernel span us %.1f' % np.nanmax(us))
print('phase 1 per wave: mean %.2f p10 %.2f p90 %.2f;  incl. barrier (tile): %.2f' % (np.nanmean(us[:, :, 1] - us[:, :, 0]), np.nanpercentile(us[:, :, 1] - us[:, :, 0], 10),
      np.nanpercentile(us[:, :, 1] - us[:, :, 0], 90), np.nanmean(np.nanmax(us[:, :, 2], axis=1) - np.nanmin(us[:, :, 0], axis=1))))
prev = us[:, :, 2]
for u in range(5):
    d = us[:, :, 3 + u] - prev
    print('unit %d: mean %.2f p10 %.2f p90 %.2f' % (u, np.nanmean(d), np.nanpercentile(d, 10), np.nanpercentile(d, 90)))
    prev = us[:, :, 3 + u]
d = us[:, :, 12] - prev
print('flush + last barrier: mean %.2f p90 %.2f' % (np.nanmean(d), np.nanpercentile(d, 90)))
if os.environ.get('SHG_TIMELINE_SAVE'):
    np.save(os.environ['SHG_TIMELINE_SAVE'], tl.cpu().numpy())
start = np.nanmin(us[:, :, 0], axis=1)
for a, b in [(0, 100), (100, 200), (200, 300), (300, 400), (400, 500)]:
    m = (start >= a) & (start < b)
    if m.sum():
        p1 = (np.nanmax(us[:, :, 2], axis=1) - start)[m]
        tt = (np.nanmax(us[:, :, 12], axis=1) - start)[m]
        print('  tiles started in [%d, %d) us: n = %d, phase 1 + barrier %.1f, tile %.1f' % (a, b, m.sum(), p1.mean(), tt.mean()))
tile_t = np.nanmax(us[:, :, 12], axis=1) - np.nanmin(us[:, :, 0], axis=1)
print('tile total: mean %.2f p10 %.2f p90 %.2f' % (tile_t.mean(), np.percentile(tile_t, 10), np.percentile(tile_t, 90)))
