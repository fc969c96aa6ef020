"""Experiment-build probe of the rotation-folded synthesis kernel (240 x d/o 96 -> 0.25 deg): kernel time from the plan's
in-library events for a list of (SHG_DEBUG, SHG_STAGGER) settings, interleaved rounds on one card.
    python3 tools/stagger_probe.py [--rounds 4] [--launches 30] dbg:stagger ...      (needs `make -C grates_amd/csrc timeline`)"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grates_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument('settings', nargs='*', default=['0:0'])
ap.add_argument('--rounds', type=int, default=4)
ap.add_argument('--launches', type=int, default=30)
ap.add_argument('--library', default=os.path.join(ROOT, 'grates_amd', 'lib', 'libshg_timeline.so'))
ap.add_argument('--out', default=None)
ap.add_argument('--path', default='auto')
args = ap.parse_args()
_lib.use_library(args.library)
import numpy as np, torch
import grates_amd as ga
grid = ga.grid.GeographicGrid(0.25, 0.25)
colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel('ewh'), 96, grid.parallels, 3.9860044150e+14, 6.3781363000e+06,
                                               grid.semimajor_axis, grid.flattening)
plan = ga.engine.Plan(96, colat, kn, grid.meridians)
plan.set_path(args.path)
batch = torch.from_numpy(np.random.default_rng(0).standard_normal((240, 97, 97)) * 1e-10).cuda()
out = torch.empty((240, 720, 1440), dtype=torch.float64, device='cuda')
for _ in range(200):
    plan.synthesis(batch, out=out)
torch.cuda.synchronize()
times = {s: [] for s in args.settings}
for rnd in range(args.rounds):
    for s in args.settings:
        fields = s.split(':')
        os.environ['SHG_DEBUG'] = fields[0]
        os.environ['SHG_STAGGER'] = fields[1]
        for k in ('SHG_SEM', 'SHG_STAGGER2'):
            os.environ.pop(k, None)
        for extra in fields[2:]:
            k, v = extra.split('=')
            os.environ[k] = v
        for _ in range(5):
            plan.synthesis(batch, out=out)
        plan.profile(True)
        for _ in range(args.launches):
            plan.synthesis(batch, out=out)
        prof = plan.profile_read()
        plan.profile(False)
        ms, n = prof['lon_stage']
        times[s].append(ms / n)
res = {}
for s in args.settings:
    t = sorted(times[s])
    res[s] = {'median_ms': t[len(t) // 2], 'min_ms': t[0], 'max_ms': t[-1]}
    print('dbg:stagger %-12s kernel %.4f ms (min %.4f max %.4f)' % (s, t[len(t) // 2], t[0], t[-1]), flush=True)
if args.out:
    json.dump(res, open(args.out, 'w'), indent=1)
