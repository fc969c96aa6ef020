"""Rotation counts of the rotation-folded synthesis kernel side by side on one card (d/o 96 -> 0.25 deg, 240 epochs): the kernel
time from the in-library events of the plan, interleaved rounds so that clock state and box are shared, every count checked
against the first one.  Usage: python tools/rot_compare.py [R ...] [--rounds 6] [--launches 40] [--library path/to/libshg.so]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('R', nargs='*', type=int, default=[6, 9, 10])
    ap.add_argument('--rounds', type=int, default=6)
    ap.add_argument('--launches', type=int, default=40)
    ap.add_argument('--epochs', type=int, default=240)
    ap.add_argument('--degree', type=int, default=96)
    ap.add_argument('--step', type=float, default=0.25)
    ap.add_argument('--library', default=None)
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    import numpy as np
    import torch
    import grates_amd as ga
    if args.library:
        ga._lib.use_library(args.library)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
    N, B = args.degree, args.epochs
    grid = ga.grid.GeographicGrid(args.step, args.step)
    colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel('ewh'), N, grid.parallels, 3.9860044150e+14, 6.3781363000e+06,
                                                   grid.semimajor_axis, grid.flattening)
    rng = np.random.default_rng(5)
    batch = torch.from_numpy(rng.standard_normal((B, N + 1, N + 1)) * 1e-10).cuda()
    out = torch.empty((B, grid.parallels.size, grid.meridians.size), dtype=torch.float64, device='cuda')
    plans = {}
    for R in args.R:
        plan = ga.engine.Plan(N, colat, kn, grid.meridians)
        plan.set_rotations(R)
        assert plan.info()['rotations'] == R, plan.info()
        plans[R] = plan
    ref = None
    errs = {}
    for R, plan in plans.items():
        plan.synthesis(batch, out=out)
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
            errs[R] = 0.0
        else:
            errs[R] = float(((out - ref).abs().max() / ref.abs().max()).item())
    for _ in range(200):                                      # clocks
        plans[args.R[0]].synthesis(batch, out=out)
    torch.cuda.synchronize()
    times = {R: [] for R in args.R}
    for rnd in range(args.rounds):
        for R, plan in plans.items():
            plan.profile(True)
            for _ in range(args.launches):
                plan.synthesis(batch, out=out)
            prof = plan.profile_read()
            plan.profile(False)
            ms, n = prof['lon_stage']
            times[R].append(ms / n)
    alg = B * 8 * ((N + 1) ** 2 + grid.parallels.size * grid.meridians.size)
    res = {}
    for R in args.R:
        t = sorted(times[R])
        med = t[len(t) // 2]
        res[R] = {'kernel_ms_median': med, 'kernel_ms_min': t[0], 'kernel_ms_max': t[-1], 'frac_of_8TBs_median': alg / (med * 1e-3) / 8e12,
                  'max_rel_diff_vs_first': errs[R]}
        print('R = {0:2d}: kernel {1:.4f} ms (min {2:.4f}, max {3:.4f})  {4:.3f} of 8 TB/s   diff vs R = {5}: {6:.2e}'.format(
            R, med, t[0], t[-1], res[R]['frac_of_8TBs_median'], args.R[0], errs[R]), flush=True)
    if args.out:
        with open(args.out, 'w') as f:
            json.dump({'workload': '{0} x d/o {1} -> {2} deg'.format(B, N, args.step), 'rounds': args.rounds, 'launches_per_round': args.launches,
                       'results': {str(k): v for k, v in res.items()}}, f, indent=1)


if __name__ == '__main__':
    main()
