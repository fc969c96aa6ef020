#!/usr/bin/env python3
"""
Benchmark of the hot path (BASELINE.json): batched synthesis of 240 monthly d/o-96 solutions to a 0.25 degree
GeographicGrid (kernel 'ewh') on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: 240 coefficient sets already resident in HBM ->
240 grids in HBM (shg_synthesis through the C ABI).  With N > 1 the script is launched by
torch.distributed.run, one rank per GPU; epochs are independent, so every rank synthesises its own 240
epochs (weak scaling, no data-path collective) and rank 0 reports the aggregate.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      dominant kernel (lon_stage) against the HBM roofline, kernel time from HIP events recorded
                on the launching stream inside the timed region
  cpu_baseline  the CPU oracle (oracle/shg_oracle.py, same formulation as the reference) timed on a bounded
                sample on this host
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = 'd/o-96 solutions/s to 0.25deg grid + full-cov GFLOP/s at 1/2/4/8 MI355X'
MAX_DEGREE = 96
GRID_STEP = 0.25
EPOCHS = 240
KERNEL = 'ewh'
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_solution(max_degree, nlat, nlon):
    """coefficients read once + grid written once; plan tables amortised over the batch."""
    return 8 * ((max_degree + 1) ** 2 + nlat * nlon)


def cpu_baseline(sample_epochs, batch_host, grid):
    """Oracle synthesis (N+1 dgemms per solution like the reference) on `sample_epochs` solutions."""
    from oracle import shg_oracle as orc
    import grates_amd as ga
    ker = orc.KernelTable(KERNEL, ga.data.load_love_numbers()[0])
    orc.synthesis_regular(batch_host[0], grid.meridians, grid.parallels, ker)      # warm BLAS / page in
    t0 = time.perf_counter()
    for e in range(sample_epochs):
        orc.synthesis_regular(batch_host[e], grid.meridians, grid.parallels, ker)
    dt = time.perf_counter() - t0
    try:
        import threadpoolctl
        threads = max([p.get('num_threads', 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    return {'value': sample_epochs / dt, 'unit': 'solutions/s', 'cores': int(threads), 'kind': 'port',
            'sample': '{0} of the {1} d/o-{2} epochs -> {3} deg grid, NumPy oracle (reference formulation), {4:.1f} s'.format(
                sample_epochs, EPOCHS, MAX_DEGREE, GRID_STEP, dt)}


def pmc_traffic():
    """HBM bytes per lon_stage launch from the committed rocprofv3 --pmc summary, if there is one."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(path) as f:
            return json.load(f).get('lon_stage_bytes_per_launch')
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--epochs', type=int, default=EPOCHS, help='epochs per GPU per step (default: BASELINE config 2)')
    ap.add_argument('--chunk', type=int, default=0, help='epochs per internal pass (0 = library default)')
    ap.add_argument('--cpu-sample', type=int, default=16, help='solutions timed on the CPU baseline (0 = skip)')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit('WORLD_SIZE={0} does not match --gpus {1}'.format(world, args.gpus))
    if args.gpus > 1 and world == 1:
        raise SystemExit('--gpus {0} needs one process per GPU: launch with python -m torch.distributed.run --nproc-per-node {0} bench.py ...'.format(args.gpus))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    import grates_amd as ga

    grid = ga.grid.GeographicGrid(GRID_STEP, GRID_STEP)
    nlat, nlon = grid.parallels.size, grid.meridians.size
    colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel(KERNEL), MAX_DEGREE, grid.parallels,
                                                   3.9860044150e+14, 6.3781363000e+06, grid.semimajor_axis, grid.flattening)
    plan = ga.engine.Plan(MAX_DEGREE, colat, kn, grid.meridians)
    if args.chunk > 0:
        plan.set_chunk(args.chunk)

    # synthetic monthly solutions: default_rng(1000 + e) N(0,1) * 1e-10, distinct per rank
    B = args.epochs
    batch_host = np.stack([np.random.default_rng(1000 + rank * B + e).standard_normal((MAX_DEGREE + 1, MAX_DEGREE + 1)) * 1e-10
                           for e in range(B)])
    batch = torch.from_numpy(batch_host).cuda()
    out = torch.empty((B, nlat, nlon), dtype=torch.float64, device='cuda')

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        plan.synthesis(batch, out=out)
    barrier()
    plan.profile(True)
    plan.profile_read()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.synthesis(batch, out=out)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = plan.profile_read()
    plan.profile(False)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        info = plan.info()
        per_solution = algorithmic_bytes_per_solution(MAX_DEGREE, nlat, nlon)
        lon_ms, lon_launches = prof.get('lon_stage', (0.0, 0))
        launches_per_step = lon_launches / max(args.steps, 1)
        epochs_per_launch = B / max(launches_per_step, 1e-9)
        lon_avg_ms = lon_ms / max(lon_launches, 1)
        achieved = per_solution * epochs_per_launch / (lon_avg_ms * 1e-3) / 1e9 if lon_launches else None
        kernels = {k: {'ms_total': round(v[0], 4), 'launches': int(v[1]), 'avg_us': round(1e3 * v[0] / max(v[1], 1), 3)} for k, v in prof.items()}
        line = {
            'metric': METRIC,
            'value': world * B * args.steps / elapsed,
            'unit': 'solutions/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': 'batch of {0} monthly solutions d/o {1} -> {2} deg GeographicGrid ({3}x{4}), kernel {5}, per GPU'.format(
                B, MAX_DEGREE, GRID_STEP, nlat, nlon, KERNEL),
                'max_degree': MAX_DEGREE, 'epochs_per_gpu': B, 'grid': [nlat, nlon], 'parallelism': 'epochs sharded over {0} GPU(s), no collective'.format(world),
                'epochs_per_pass': info['epochs_per_pass'], 'fourfold_symmetry': info['fourfold_symmetry']},
            'roofline': {
                'kernel': 'lon_stage_kernel<4>', 'bound': 'hbm',
                'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': (achieved / HBM_PEAK_GBS) if achieved else None,
                'traffic': pmc_traffic(),
                'algorithmic_bytes_per_launch': per_solution * epochs_per_launch,
                'avg_launch_ms': lon_avg_ms,
                'whole_path_GBs': per_solution * B * args.steps / elapsed / 1e9,
            },
            'kernels': kernels,
        }
        if world == 1 and args.cpu_sample > 0:
            line['cpu_baseline'] = cpu_baseline(min(args.cpu_sample, B), batch_host, grid)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
