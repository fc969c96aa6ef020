"""Analysis (d/o 96 <- 0.5 degree grid, 240 epochs) event-timed per call, with the per-kernel times of the plan profile.
    python3 tools/analysis_time.py [library.so | -] [epochs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import grates_amd as ga
if len(sys.argv) > 1 and sys.argv[1] != '-':
    ga._lib.use_library(sys.argv[1])
import bench
N, B = bench.ANA_DEGREE, int(sys.argv[2]) if len(sys.argv) > 2 else 240
grid = ga.grid.GeographicGrid(bench.ANA_GRID_STEP, bench.ANA_GRID_STEP)
nlat, nlon = grid.parallels.size, grid.meridians.size
colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel(bench.KERNEL), N, grid.parallels, bench.GM, bench.R_EARTH, grid.semimajor_axis, grid.flattening)
plan = ga.engine.Plan(N, colat, kn, grid.meridians)
batch = torch.from_numpy(bench.coefficient_batch(20_000, B, N)).cuda()
grids = plan.synthesis(batch)
area = ga.engine.to_device(grid.area.reshape(nlat, nlon))
out = plan.analysis(grids, area, 0)
print('round trip max rel err', float(((out - batch).abs().max() / batch.abs().max()).item()), plan.analysis_info())
for rnd in range(3):
    for _ in range(3): plan.analysis(grids, area, 0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): plan.analysis(grids, area, 0)
    b.record(); torch.cuda.synchronize()
    print('round %d: %.1f us per call' % (rnd, 1e3 * a.elapsed_time(b) / 20), flush=True)
plan.profile(True); plan.profile_read()
for _ in range(10): plan.analysis(grids, area, 0)
torch.cuda.synchronize()
print({k: (round(1e3 * v[0] / max(v[1], 1), 1), v[1]) for k, v in plan.profile_read().items()}, 'us per launch, launches')
