"""Headline synthesis kernel (240 x d/o 96 -> 0.25 degree) against the limit on workgroups in their Legendre stage (shg_plan_set_stage_limit):
interleaved rounds of event-timed launches behind a warm-up of dense products.   python3 tools/stage_limit_sweep.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import grates_amd as ga
import bench
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
grid = ga.grid.GeographicGrid(bench.GRID_STEP, bench.GRID_STEP)
colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel(bench.KERNEL), bench.MAX_DEGREE, grid.parallels, bench.GM, bench.R_EARTH, grid.semimajor_axis, grid.flattening)
plan = ga.engine.Plan(bench.MAX_DEGREE, colat, kn, grid.meridians)
batch = torch.from_numpy(bench.coefficient_batch(1000, bench.EPOCHS, bench.MAX_DEGREE)).cuda()
out = torch.empty((bench.EPOCHS, grid.parallels.size, grid.meridians.size), dtype=torch.float64, device='cuda')
for _ in range(300):
    plan.synthesis(batch, out=out)
limits = (0, 64, 96, 112, 128, 144, 160, 192, 224)
best = {}
for rnd in range(rounds):
    row = []
    for limit in limits:
        plan.set_stage_limit(limit)
        for _ in range(10):
            plan.synthesis(batch, out=out)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(40):
            plan.synthesis(batch, out=out)
        b.record()
        torch.cuda.synchronize()
        us = 1e3 * a.elapsed_time(b) / 40
        row.append(us)
        best[limit] = min(best.get(limit, 1e9), us)
    print('round %d: ' % rnd + '  '.join('%d: %.1f' % (l, u) for l, u in zip(limits, row)), flush=True)
print('best per limit (us per step incl. repack): ' + '  '.join('%d: %.1f' % (l, best[l]) for l in limits))
