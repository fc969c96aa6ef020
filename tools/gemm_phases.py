"""
Where the K loop of the covariance kernel (gemm_f64_kernel<MODE_COVPROP>, csrc/gemm.hip) spends its cycles: the profiling
build (`make -C grates_amd/csrc timeline`) sums, per wave and on the scalar unit, the shader-clock cycles between four
marks of every K tile -- loads issued | 64 MFMAs issued | next tile staged to LDS | barrier passed -- and the shader clock
against the 100 MHz wall clock.

    python3 tools/gemm_phases.py [parallels [max_degree]]         (on a GPU box)
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

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 180
grid = ga.grid.GeographicGrid(0.5, 0.5)
GM, R = 3.9860044150e+14, 6.3781363000e+06
colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel('ewh'), N, grid.parallels, GM, R, grid.semimajor_axis, grid.flattening)
plan = ga.engine.Plan(N, colat, kn, grid.meridians)
P = (N + 1) ** 2
gen = torch.Generator(device='cuda'); gen.manual_seed(7)
cov = torch.rand((P, P), dtype=torch.float64, device='cuda', generator=gen)
lat0 = 176
M = nb * grid.meridians.size
blocks = -(-M // 128) * -(-P // 128)
tl = torch.zeros((blocks, 4, 8), dtype=torch.int64, device='cuda')
plan.covariance_propagation(cov, 0, lat0, lat0 + nb)
torch.cuda.synchronize()
os.environ['SHG_TIMELINE_PTR'] = str(tl.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
plan.covariance_propagation(cov, 0, lat0, lat0 + nb)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
flop = 2.0 * M * P * P + 2.0 * M * P
print('d/o %d, band of %d parallels: %.2f ms, %.2f TFLOP/s' % (N, nb, ms, flop / ms / 1e9))
t = tl.cpu().numpy().astype(np.float64)
tiles = t[:, :, 7]
ok = tiles > 0
per = t[:, :, 0:4] / np.maximum(tiles, 1)[:, :, None]
names = ('loads issued', '64 MFMAs issued', 'tile staged', 'barrier passed')
for i, n in enumerate(names):
    v = per[:, :, i][ok]
    print('%-16s mean %8.1f  p10 %8.1f  p50 %8.1f  p90 %8.1f cycles per K tile' % (n, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90)))
tot = per.sum(axis=2)[ok]
print('K tile total     mean %8.1f cycles  (64 MFMAs x 64 cycles x 2 waves per SIMD = 8192)' % tot.mean())
clk = (t[:, :, 4] / np.maximum(t[:, :, 5], 1))[ok] * 100.0
print('shader clock     mean %.0f MHz  p10 %.0f  p90 %.0f' % (clk.mean(), np.percentile(clk, 10), np.percentile(clk, 90)))
life = t[:, :, 4][ok]
print('wave lifetime    mean %.0f cycles, K loop share %.3f' % (life.mean(), (tot * tiles[ok]).sum() / life.sum()))
