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


COV_DEGREE = 180
COV_GRID_STEP = 0.5
MFMA_F64_PEAK_TFLOPS = 78.6     # MI355X dense fp64 matrix peak (spec); measured 77.3 TFLOP/s (profiles/r01_microbench.txt)


def covariance_leg(args, rank, world, barrier, reduce_device='cuda'):
    """d/o-180 covariance propagation to a 0.5 degree grid (BASELINE config 4), a band of parallels per GPU:
    sigma = sqrt(diag(A Sigma A^T)) with A generated on the fly, A Sigma on fp64 MFMA.  Flops = 2 M P^2 + 2 M P."""
    import torch
    import torch.distributed as dist
    import grates_amd as ga
    N = COV_DEGREE
    grid = ga.grid.GeographicGrid(COV_GRID_STEP, COV_GRID_STEP)
    nlat, nlon = grid.parallels.size, grid.meridians.size
    P = (N + 1) ** 2
    colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel(KERNEL), N, grid.parallels, 3.9860044150e+14, 6.3781363000e+06,
                                                   grid.semimajor_axis, grid.flattening)
    plan = ga.engine.Plan(N, colat, kn, grid.meridians)
    # synthetic SPD covariance, generated on the device (8.59 GB, never shipped): symmetric random + dominant diagonal
    gen = torch.Generator(device='cuda').manual_seed(7)
    cov = torch.rand((P, P), dtype=torch.float64, device='cuda', generator=gen)
    cov = (cov + cov.T) * (0.5e-22 / P)
    cov.diagonal().add_(2e-22)
    band = min(args.cov_parallels, nlat // world)
    lat0 = rank * (nlat // world)                     # each rank works inside its own latitude band of the full sharding
    lat1 = lat0 + band
    plan.covariance_propagation(cov, 0, lat0, lat0 + 1)          # warm-up: builds the plan tables
    barrier()
    plan.profile(True)
    plan.profile_read()
    t0 = time.perf_counter()
    sigma = plan.covariance_propagation(cov, 0, lat0, lat1)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = plan.profile_read()
    plan.profile(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    M = band * nlon
    flops = 2.0 * M * P * P + 2.0 * M * P
    if rank != 0:
        return None
    k_ms, k_n = prof.get('covprop', (0.0, 0))
    achieved = flops / (k_ms * 1e-3) / 1e12 if k_n else None
    out = {
        'metric': 'full-covariance propagation d/o 180 -> 0.5 deg grid', 'value': world * flops / elapsed / 1e9, 'unit': 'GFLOP/s',
        'n_gpus': world, 'config': {'workload': 'd/o {0} (P = {1}, Sigma {2:.2f} GB replicated), {3} of {4} parallels x {5} meridians per GPU'.format(
            N, P, P * P * 8 / 1e9, band, nlat, nlon), 'flops_per_gpu': flops, 'full_grid_flops': 2.0 * nlat * nlon * P * (P + 1.0)},
        'seconds': elapsed,
        'roofline': {'kernel': 'gemm_f64_kernel<MODE_COVPROP>', 'bound': 'mfma', 'achieved': achieved, 'peak': MFMA_F64_PEAK_TFLOPS,
                     'unit': 'TFLOP/s', 'frac': (achieved / MFMA_F64_PEAK_TFLOPS) if achieved else None, 'traffic': None,
                     'avg_launch_ms': k_ms / max(k_n, 1)},
        'sigma_checksum': float(sigma.sum().item()),
    }
    if world == 1:
        # extension, reported beside the headline and not part of `value`: the same band with the upper-triangle shortcut for a
        # symmetric Sigma (half the MFMA work; GFLOP/s still counted with the algorithmic 2 M P^2 + 2 M P of the general product)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sigma_sym = plan.covariance_propagation(cov, 0, lat0, lat1, symmetric=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out['symmetric_shortcut'] = {'seconds': dt, 'algorithmic_GFLOPs': flops / dt / 1e9,
                                     'max_rel_diff_vs_general': float(((sigma_sym - sigma).abs().max() / sigma.abs().max()).item())}
    if world == 1:
        # second extension, also outside `value`: the WHOLE grid through the latitude / longitude factorisation of the synthesis
        # matrix (csrc/covsep.hip: 2 nlat P^2 flops instead of 2 nlat nlon P^2); the band above is compared with its rows
        plan.covariance_propagation(cov, 0, method='separable')      # warm-up with the same workspace sizes (the pool grows by 12 GB)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sigma_sep = plan.covariance_propagation(cov, 0, method='separable')
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rows = sigma_sep.reshape(nlat, nlon)[lat0:lat1].reshape(-1)
        out['separable_variant'] = {'full_grid_seconds': dt, 'full_grid_points': nlat * nlon,
                                    'equivalent_reference_formulation_GFLOPs': 2.0 * nlat * nlon * P * (P + 1.0) / dt / 1e9,
                                    'executed_GFLOP': (2.0 * nlat * P * P + 2.0 * nlat * nlon * (2 * N + 1) ** 2) / 1e9,
                                    'max_rel_diff_vs_general': float(((rows - sigma).abs().max() / sigma.abs().max()).item())}
        del sigma_sep
        # the same for a symmetric Sigma (only the slot pairs s >= s' of B_i are formed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sigma_sep = plan.covariance_propagation(cov, 0, method='separable', symmetric=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rows = sigma_sep.reshape(nlat, nlon)[lat0:lat1].reshape(-1)
        out['separable_variant']['symmetric_full_grid_seconds'] = dt
        out['separable_variant']['symmetric_max_rel_diff_vs_general'] = float(((rows - sigma).abs().max() / sigma.abs().max()).item())
        del sigma_sep
    if world == 1 and args.cov_cpu_parallels > 0:
        from oracle import shg_oracle as orc
        ker = orc.KernelTable(KERNEL, ga.data.load_love_numbers()[0])
        cov_host = cov.cpu().numpy()
        t0 = time.perf_counter()
        ref = orc.covariance_propagation_regular(cov_host, 0, N, grid.meridians, grid.parallels, ker, parallel_range=(lat0, lat0 + args.cov_cpu_parallels))
        dt = time.perf_counter() - t0
        mc = args.cov_cpu_parallels * nlon
        got = sigma[0:mc].cpu().numpy()
        out['cpu_baseline'] = {'value': (2.0 * mc * P * P + 2.0 * mc * P) / dt / 1e9, 'unit': 'GFLOP/s', 'cores': os.cpu_count(), 'kind': 'port',
                               'sample': '{0} of {1} parallels at full P (NumPy oracle, per-parallel F @ Sigma), {2:.1f} s incl. table setup'.format(
                                   args.cov_cpu_parallels, nlat, dt),
                               'max_rel_diff_vs_gpu': float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))}
    return out


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
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--ramp', type=int, default=300, help='untimed launches before the warm-up steps (device clock ramp)')
    ap.add_argument('--epochs', type=int, default=EPOCHS, help='epochs per GPU per step (default: BASELINE config 2)')
    ap.add_argument('--chunk', type=int, default=0, help='epochs per internal pass (0 = library default)')
    ap.add_argument('--path', default='auto', choices=['auto', 'staged', 'fused', 'fused_plain', 'fused32', 'rot', 'rot_plain'], help='synthesis kernel path')
    ap.add_argument('--cpu-sample', type=int, default=16, help='solutions timed on the CPU baseline (0 = skip)')
    ap.add_argument('--cov-parallels', type=int, default=8, help='parallels of the d/o-180 covariance-propagation leg per GPU (0 = skip)')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for N > 1 ('nccl' = RCCL; 'gloo' only to rehearse on one GPU)")
    ap.add_argument('--same-device', action='store_true', help='rehearsal: map every rank to cuda:0 (with --backend gloo)')
    ap.add_argument('--cov-cpu-parallels', type=int, default=1, help='parallels of the covariance CPU baseline (0 = skip)')
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
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    reduce_device = 'cuda' if args.backend == 'nccl' else 'cpu'

    import grates_amd as ga

    grid = ga.grid.GeographicGrid(GRID_STEP, GRID_STEP)
    nlat, nlon = grid.parallels.size, grid.meridians.size
    colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel(KERNEL), MAX_DEGREE, grid.parallels,
                                                   3.9860044150e+14, 6.3781363000e+06, grid.semimajor_axis, grid.flattening)
    plan = ga.engine.Plan(MAX_DEGREE, colat, kn, grid.meridians)
    if args.chunk > 0:
        plan.set_chunk(args.chunk)
    if args.path != 'auto':
        plan.set_path(args.path)

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

    # Device ramp (part of the setup, like the plan build above): after an idle period the first ~100 ms of fp64 MFMA work
    # run at lower clocks (a single launch takes 0.94 ms, launches in a steady stream 0.64-0.67 ms), and the first launches
    # also build the plan's lazily created tables.  The W warm-up steps and the K timed steps follow as the contract says.
    for _ in range(args.ramp):
        plan.synthesis(batch, out=out)
    barrier()
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
        t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        info = plan.info()
        per_solution = algorithmic_bytes_per_solution(MAX_DEGREE, nlat, nlon)
        main_kernel = ('synthesis_rot_kernel' if info['rotation_symmetry'] and args.path in ('auto', 'rot', 'rot_plain') else 'synthesis_fused_kernel') if info['fused'] else 'lon_stage_kernel<4>'
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
                'fused_kernel': info['fused'], 'fourfold_symmetry': info['fourfold_symmetry'], 'rotation_folded_kernel': info['rotation_symmetry'] and args.path in ('auto', 'rot', 'rot_plain'),
                'untimed_ramp_launches': args.ramp},
            'roofline': {
                'kernel': main_kernel, 'bound': 'hbm',
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

    # ---- second half of the metric: full-covariance propagation GFLOP/s (d/o 180 -> 0.5 deg, latitude bands)
    del out, batch
    torch.cuda.empty_cache()
    cov = covariance_leg(args, rank, world, barrier, reduce_device) if args.cov_parallels > 0 else None
    if rank == 0:
        line['covariance'] = cov
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
