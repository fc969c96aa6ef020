"""What distinguishes the boxes on which the Legendre-stage limit helps from those on which it costs?  Per run: the card's plain write rate (torch
fill of 2 GB), its read + write rate (copy of 2 GB), then the headline synthesis kernel without and with the limit (interleaved rounds).
    python3 tools/box_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import grates_amd as ga
import bench

def timed(f, n):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n

grid = ga.grid.GeographicGrid(bench.GRID_STEP, bench.GRID_STEP)
colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel(bench.KERNEL), bench.MAX_DEGREE, grid.parallels, bench.GM, bench.R_EARTH, grid.semimajor_axis, grid.flattening)
plan = ga.engine.Plan(bench.MAX_DEGREE, colat, kn, grid.meridians)
batch = torch.from_numpy(bench.coefficient_batch(1000, bench.EPOCHS, bench.MAX_DEGREE)).cuda()
out = torch.empty((bench.EPOCHS, grid.parallels.size, grid.meridians.size), dtype=torch.float64, device='cuda')
src = torch.empty_like(out)
nbytes = out.numel() * 8
# cold: straight behind the setup
cold_fill = timed(lambda: out.fill_(1.0), 3)
for _ in range(300):
    plan.synthesis(batch, out=out)
fill = min(timed(lambda: out.fill_(1.0), 10) for _ in range(3))
copy = min(timed(lambda: out.copy_(src), 10) for _ in range(3))
res = {0: [], -7: []}
for rnd in range(4):
    for limit in (0, -7):
        plan.set_stage_limit(limit)
        for _ in range(10):
            plan.synthesis(batch, out=out)
        res[limit].append(timed(lambda: plan.synthesis(batch, out=out), 40))
off, on = min(res[0]), min(res[-7])
print('BOX fill %.2f TB/s (cold %.2f)  copy %.2f TB/s (r+w)  synthesis step %.1f us off, %.1f us on -> limit %+.1f %%' % (
    nbytes / fill / 1e9, nbytes / cold_fill / 1e9, 2 * nbytes / copy / 1e9, 1e3 * off, 1e3 * on, 100.0 * (on / off - 1.0)), flush=True)
