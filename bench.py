#!/usr/bin/env python3
"""
Benchmark of the hot path (BASELINE.json): batched synthesis of 240 monthly d/o-96 solutions to a 0.25 degree
GeographicGrid (kernel 'ewh') on MI355X, plus the second half of the metric, the d/o-180 covariance propagation.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: 240 coefficient sets already resident in HBM ->
240 grids in HBM (shg_synthesis through the C ABI).

N > 1: one process per GPU.  Started under torch.distributed.run (RANK / WORLD_SIZE in the environment) the script is a
rank; started bare (`python bench.py --gpus 4`) it launches `python -m torch.distributed.run --nproc-per-node N` on
itself as a child process BEFORE anything touches the GPU and exits with the child's code.
  * synthesis: epochs are independent, every rank synthesises its own 240 epochs (weak scaling, no data-path collective);
    `value` = epochs of all ranks / max-over-ranks time;
  * covariance: the WHOLE 360 x 720 grid, the parallels split into contiguous latitude bands
    (grates_amd.distributed.latitude_bands), Sigma replicated, one all_gather of the per-band sigma (RCCL); the
    checksum of the gathered grid is the same for every N.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      dominant kernel against the HBM roofline, kernel time from HIP events recorded on the launching stream
                inside the timed region
  cpu_baseline  the CPU oracle (oracle/shg_oracle.py, same formulation as the reference) timed on a bounded sample on
                this host (rank 0, N = 1 only)
  covariance    the covariance leg (its own roofline and cpu_baseline objects)
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = 'd/o-96 solutions/s to 0.25deg grid + full-cov GFLOP/s at 1/2/4/8 MI355X'
MAX_DEGREE = 96
GRID_STEP = 0.25
EPOCHS = 240
KERNEL = 'ewh'
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
COV_DEGREE = 180
COV_GRID_STEP = 0.5
MFMA_F64_PEAK_TFLOPS = 78.6     # MI355X dense fp64 matrix peak (spec); measured 77.3 TFLOP/s (profiles/r01_microbench.txt)
GM, R_EARTH = 3.9860044150e+14, 6.3781363000e+06


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--ramp', type=int, default=300, help='untimed launches before the warm-up steps (device clock ramp)')
    ap.add_argument('--epochs', type=int, default=EPOCHS, help='epochs per GPU per step (default: BASELINE config 2)')
    ap.add_argument('--chunk', type=int, default=0, help='epochs per internal pass of the staged path (0 = library default)')
    ap.add_argument('--path', default='auto', choices=['auto', 'staged', 'fused', 'fused_plain', 'fused32', 'rot', 'rot_plain'],
                    help='synthesis kernel path')
    ap.add_argument('--cpu-sample', type=int, default=16, help='solutions timed on the CPU baseline (0 = skip)')
    ap.add_argument('--cov-parallels', type=int, default=-1,
                    help='parallels of the covariance leg over all ranks (-1 = the whole 0.5 degree grid, 360; 0 = skip the leg)')
    ap.add_argument('--cov-repeats', type=int, default=3, help='timed passes of the covariance leg')
    ap.add_argument('--cov-cpu-parallels', type=int, default=1, help='parallels of the covariance CPU baseline (0 = skip)')
    ap.add_argument('--cov-extensions', type=int, default=1, help='1: also time the symmetric and separable variants (N = 1 only)')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for N > 1 ('nccl' = RCCL; 'gloo' only to rehearse on one GPU)")
    ap.add_argument('--same-device', action='store_true', help='rehearsal: map every rank to cuda:0 (with --backend gloo)')
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# launcher
# ---------------------------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks through torch.distributed.run as a child process.
    Nothing in this process has touched the GPU (no torch import so far)."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------------------------
# workloads: the GPU one (product path through the C ABI) and a stand-in of the same shape for the CPU test of the rank
# function (tests/test_bench_ranks.py): same sharding, barriers, gather, checksum and JSON assembly, no GPU
# ---------------------------------------------------------------------------------------------------------------------
class GpuWorkload:
    device = 'cuda'

    def __init__(self, args, rank, world, local_rank):
        import torch
        self.torch = torch
        self.args, self.rank, self.world = args, rank, world
        torch.cuda.set_device(local_rank)

    def synchronize(self):
        self.torch.cuda.synchronize()

    # ---- synthesis
    def setup_synthesis(self):
        import numpy as np
        import grates_amd as ga
        torch = self.torch
        self.ga = ga
        grid = ga.grid.GeographicGrid(GRID_STEP, GRID_STEP)
        self.grid = grid
        self.nlat, self.nlon = grid.parallels.size, grid.meridians.size
        colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel(KERNEL), MAX_DEGREE, grid.parallels, GM, R_EARTH,
                                                       grid.semimajor_axis, grid.flattening)
        self.plan = ga.engine.Plan(MAX_DEGREE, colat, kn, grid.meridians)
        if self.args.chunk > 0:
            self.plan.set_chunk(self.args.chunk)
        if self.args.path != 'auto':
            self.plan.set_path(self.args.path)
        B = self.args.epochs
        # synthetic monthly solutions: default_rng(1000 + e) N(0,1) * 1e-10, distinct per rank
        self.batch_host = np.stack([np.random.default_rng(1000 + self.rank * B + e).standard_normal((MAX_DEGREE + 1, MAX_DEGREE + 1)) * 1e-10
                                    for e in range(B)])
        self.batch = torch.from_numpy(self.batch_host).cuda()
        self.out = torch.empty((B, self.nlat, self.nlon), dtype=torch.float64, device='cuda')
        info = self.plan.info()
        rot = bool(info['rotation_symmetry']) and self.args.path in ('auto', 'rot', 'rot_plain')
        self.kernel_name = ('synthesis_rot_kernel' if rot else 'synthesis_fused_kernel') if info['fused'] else 'lon_stage_kernel<4>'
        self.config = {'fused_kernel': info['fused'], 'fourfold_symmetry': info['fourfold_symmetry'], 'rotation_folded_kernel': rot}

    def synthesis_step(self):
        self.plan.synthesis(self.batch, out=self.out)

    def profile(self, enable):
        self.plan.profile(enable)

    def profile_read(self):
        return self.plan.profile_read()

    def release_synthesis(self):
        del self.out, self.batch
        self.torch.cuda.empty_cache()

    def cpu_baseline(self, sample_epochs):
        """Oracle synthesis (N+1 dgemms per solution like the reference) on `sample_epochs` solutions."""
        from oracle import shg_oracle as orc
        ker = orc.KernelTable(KERNEL, self.ga.data.load_love_numbers()[0])
        grid = self.grid
        orc.synthesis_regular(self.batch_host[0], grid.meridians, grid.parallels, ker)      # warm BLAS / page in
        t0 = time.perf_counter()
        for e in range(sample_epochs):
            orc.synthesis_regular(self.batch_host[e], grid.meridians, grid.parallels, ker)
        dt = time.perf_counter() - t0
        return {'value': sample_epochs / dt, 'unit': 'solutions/s', 'cores': blas_threads(), 'kind': 'port',
                'sample': '{0} of the {1} d/o-{2} epochs -> {3} deg grid, NumPy oracle (reference formulation), {4:.1f} s'.format(
                    sample_epochs, self.args.epochs, MAX_DEGREE, GRID_STEP, dt)}

    # ---- covariance
    def setup_covariance(self):
        ga, torch = self.ga, self.torch
        N = COV_DEGREE
        grid = ga.grid.GeographicGrid(COV_GRID_STEP, COV_GRID_STEP)
        self.cov_grid = grid
        self.cov_nlat, self.cov_nlon = grid.parallels.size, grid.meridians.size
        self.P = (N + 1) ** 2
        colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel(KERNEL), N, grid.parallels, GM, R_EARTH, grid.semimajor_axis, grid.flattening)
        self.cov_plan = ga.engine.Plan(N, colat, kn, grid.meridians)
        # synthetic covariance matrix: Sigma = G G^T / k * 1e-22, G [P, k] standard normal, k = P + 16, generated on
        # the device (seeded Philox stream: the same matrix on every rank; 8.59 GB, never shipped) and multiplied on the fp64 MFMA GEMM
        P, k = self.P, self.P + 16
        gen = torch.Generator(device='cuda').manual_seed(7)
        G = torch.randn((P, k), dtype=torch.float64, device='cuda', generator=gen)
        self.cov = ga.engine.gemm(G, G, transb=True, alpha=1e-22 / k)
        del G
        torch.cuda.empty_cache()
        self.cov_recipe = 'Sigma = G G^T / k * 1e-22, G [P, P + 16] torch.randn(seed 7) on the device (SURVEY 8d recipe with the device generator)'

    def covariance_band(self, lat0, lat1, **kw):
        return self.cov_plan.covariance_propagation(self.cov, 0, lat0, lat1, **kw)

    def cov_profile(self, enable):
        self.cov_plan.profile(enable)

    def cov_profile_read(self):
        return self.cov_plan.profile_read()

    def covariance_cpu(self, lat0, count, sigma_gpu):
        import numpy as np
        from oracle import shg_oracle as orc
        ker = orc.KernelTable(KERNEL, self.ga.data.load_love_numbers()[0])
        grid = self.cov_grid
        cov_host = self.cov.cpu().numpy()
        t0 = time.perf_counter()
        ref = orc.covariance_propagation_regular(cov_host, 0, COV_DEGREE, grid.meridians, grid.parallels, ker, parallel_range=(lat0, lat0 + count))
        dt = time.perf_counter() - t0
        mc = count * self.cov_nlon
        got = sigma_gpu[0:mc].cpu().numpy()
        P = self.P
        return {'value': (2.0 * mc * P * P + 2.0 * mc * P) / dt / 1e9, 'unit': 'GFLOP/s', 'cores': blas_threads(), 'kind': 'port',
                'sample': '{0} of {1} parallels at full P (NumPy oracle, per-parallel F @ Sigma), {2:.1f} s incl. table setup'.format(count, self.cov_nlat, dt),
                'max_rel_diff_vs_gpu': float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))}


def blas_threads():
    """Threads the NumPy BLAS of the CPU baseline runs on (one convention for both legs)."""
    try:
        import threadpoolctl
        return int(max([p.get('num_threads', 1) for p in threadpoolctl.threadpool_info()] or [1]))
    except Exception:
        return int(os.cpu_count() or 1)


def pmc_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary (profiles/), with its origin:
    counters cannot be collected inside a timed run."""
    for name in ('r02_pmc_traffic.json', 'r01_pmc_traffic.json'):
        path = os.path.join(ROOT, 'profiles', name)
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:
            continue
        if kernel_name.split('<')[0] in d.get('kernel', ''):
            return d.get('lon_stage_bytes_per_launch'), 'profiles/' + name + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of ' + d.get('commit', 'an earlier build') + ')'
    return None, None


def algorithmic_bytes_per_solution(max_degree, nlat, nlon):
    """coefficients read once + grid written once; plan tables amortised over the batch."""
    return 8 * ((max_degree + 1) ** 2 + nlat * nlon)


# ---------------------------------------------------------------------------------------------------------------------
# rank function
# ---------------------------------------------------------------------------------------------------------------------
def run_rank(args, workload_factory=GpuWorkload, emit=print):
    """One rank of the benchmark (the whole benchmark when WORLD_SIZE is 1).  Returns the result dict on rank 0."""
    import torch
    import torch.distributed as dist
    from grates_amd import distributed as gd

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = 0 if args.same_device else int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE={0} does not match --gpus {1}'.format(world, args.gpus))
    wl = workload_factory(args, rank, world, local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    reduce_device = wl.device if args.backend == 'nccl' else 'cpu'

    def barrier():
        wl.synchronize()
        if world > 1:
            dist.barrier()
        wl.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed_steps(warmup, steps):
        for _ in range(warmup):
            wl.synthesis_step()
        barrier()
        wl.profile(True)
        wl.profile_read()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            wl.synthesis_step()
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        prof = wl.profile_read()
        wl.profile(False)
        return elapsed, prof

    # ---- synthesis: W warm-up steps, K timed steps between barrier + synchronize pairs, max over ranks.
    # First as the contract states it, straight after the setup (`value_without_ramp`); then again behind `--ramp` untimed
    # launches (`value`): after an idle period the first ~100 ms of fp64 MFMA work run at lower clocks.
    wl.setup_synthesis()
    B = args.epochs
    barrier()
    cold_elapsed, _ = timed_steps(args.warmup, args.steps)
    for _ in range(args.ramp):
        wl.synthesis_step()
    barrier()
    elapsed, prof = timed_steps(args.warmup, args.steps)

    line = None
    if rank == 0:
        per_solution = algorithmic_bytes_per_solution(MAX_DEGREE, wl.nlat, wl.nlon)
        lon_ms, lon_launches = prof.get('lon_stage', (0.0, 0))
        launches_per_step = lon_launches / max(args.steps, 1)
        epochs_per_launch = B / max(launches_per_step, 1e-9)
        lon_avg_ms = lon_ms / max(lon_launches, 1)
        achieved = per_solution * epochs_per_launch / (lon_avg_ms * 1e-3) / 1e9 if lon_launches else None
        traffic, traffic_source = pmc_traffic(wl.kernel_name)
        kernels = {k: {'ms_total': round(v[0], 4), 'launches': int(v[1]), 'avg_us': round(1e3 * v[0] / max(v[1], 1), 3)} for k, v in prof.items()}
        config = {'workload': 'batch of {0} monthly solutions d/o {1} -> {2} deg GeographicGrid ({3}x{4}), kernel {5}, per GPU'.format(
            B, MAX_DEGREE, GRID_STEP, wl.nlat, wl.nlon, KERNEL),
            'max_degree': MAX_DEGREE, 'epochs_per_gpu': B, 'grid': [wl.nlat, wl.nlon],
            'parallelism': 'epochs sharded over {0} GPU(s), no collective'.format(world), 'untimed_ramp_launches': args.ramp}
        config.update(wl.config)
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
            'config': config,
            'value_without_ramp': world * B * args.steps / cold_elapsed,
            'roofline': {
                'kernel': wl.kernel_name, 'bound': 'hbm',
                'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': (achieved / HBM_PEAK_GBS) if achieved else None,
                'traffic': traffic, 'traffic_source': traffic_source,
                'algorithmic_bytes_per_launch': per_solution * epochs_per_launch,
                'avg_launch_ms': lon_avg_ms,
                'whole_path_GBs': per_solution * B * args.steps / elapsed / 1e9,
            },
            'kernels': kernels,
        }
        line['cpu_baseline'] = wl.cpu_baseline(min(args.cpu_sample, B)) if world == 1 and args.cpu_sample > 0 else None

    # ---- second half of the metric: full-covariance propagation GFLOP/s (d/o 180 -> 0.5 deg, latitude bands)
    wl.release_synthesis()
    cov = covariance_leg(args, wl, rank, world, barrier, max_over_ranks, gd) if args.cov_parallels != 0 else None
    if rank == 0:
        line['covariance'] = cov
        emit(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return line


def covariance_leg(args, wl, rank, world, barrier, max_over_ranks, gd):
    """d/o-180 covariance propagation to the 0.5 degree grid (BASELINE config 4): sigma = sqrt(diag(A Sigma A^T)) with A
    generated on the fly, A Sigma on fp64 MFMA.  Flops = 2 M P^2 + 2 M P.  The parallels [0, total) are split into one
    contiguous band per rank, every rank holds all of Sigma, the bands are gathered with one all_gather."""
    import torch
    wl.setup_covariance()
    nlat, nlon, P = wl.cov_nlat, wl.cov_nlon, wl.P
    total = nlat if args.cov_parallels < 0 else min(args.cov_parallels, nlat)
    bands = gd.latitude_bands(total, world)
    lat0, lat1 = bands[rank]
    sizes = [(b1 - b0) * nlon for b0, b1 in bands]
    wl.covariance_band(lat0, min(lat0 + 1, lat1))             # warm-up: builds the plan tables
    barrier()
    wl.cov_profile(True)
    wl.cov_profile_read()
    times = []
    sigma = None
    for _ in range(max(args.cov_repeats, 1)):
        barrier()
        t0 = time.perf_counter()
        local = wl.covariance_band(lat0, lat1)
        sigma = gd.all_gather_bands(local, sizes)             # RCCL all_gather of the per-band sigma (inside the timed region)
        barrier()
        times.append(max_over_ranks(time.perf_counter() - t0))
    prof = wl.cov_profile_read()
    wl.cov_profile(False)
    if rank != 0:
        return None
    M = total * nlon
    flops = 2.0 * M * P * P + 2.0 * M * P
    best, median = min(times), sorted(times)[len(times) // 2]
    k_ms, k_n = prof.get('covprop', (0.0, 0))
    M_rank0 = sizes[0]
    achieved = (2.0 * M_rank0 * P * P + 2.0 * M_rank0 * P) / (k_ms / max(k_n, 1) * 1e-3) / 1e12 if k_n else None
    host = sigma.detach().cpu().numpy()
    out = {
        'metric': 'full-covariance propagation d/o 180 -> 0.5 deg grid', 'value': flops / median / 1e9, 'unit': 'GFLOP/s',
        'n_gpus': world, 'scaling': 'strong',
        'config': {'workload': 'd/o {0} (P = {1}, Sigma {2:.2f} GB replicated), parallels 0..{3} of {4} x {5} meridians in {6} latitude band(s) of {7} parallels, all_gather of sigma'.format(
            COV_DEGREE, P, P * P * 8 / 1e9, total, nlat, nlon, world, [b1 - b0 for b0, b1 in bands]),
            'sigma_recipe': wl.cov_recipe, 'flops': flops, 'full_grid_flops': 2.0 * nlat * nlon * P * (P + 1.0), 'repeats': len(times)},
        'seconds_median': median, 'seconds_min': best, 'seconds_all': times, 'GFLOPs_best': flops / best / 1e9,
        'roofline': {'kernel': 'gemm_f64_kernel<MODE_COVPROP>', 'bound': 'mfma', 'achieved': achieved, 'peak': MFMA_F64_PEAK_TFLOPS,
                     'unit': 'TFLOP/s', 'frac': (achieved / MFMA_F64_PEAK_TFLOPS) if achieved else None, 'traffic': None,
                     'avg_launch_ms': k_ms / max(k_n, 1), 'launches': int(k_n)},
        'sigma_checksum': float(host.sum()), 'sigma_crc32': zlib.crc32(host.tobytes()) & 0xffffffff,
    }
    if world == 1 and args.cov_extensions and hasattr(wl, 'cov_plan'):
        # extensions, reported beside the headline and not part of `value`: the upper-triangle shortcut for a symmetric Sigma (half
        # the MFMA work) on a band of 8 parallels, and the WHOLE grid through the latitude / longitude factorisation of the
        # synthesis matrix (csrc/covsep.hip: 2 nlat P^2 flops instead of 2 nlat nlon P^2)
        ref = sigma[0:8 * nlon]
        wl.synchronize()
        t0 = time.perf_counter()
        sym = wl.covariance_band(0, 8, symmetric=True)
        wl.synchronize()
        dt = time.perf_counter() - t0
        out['symmetric_shortcut'] = {'parallels': 8, 'seconds': dt, 'max_rel_diff_vs_general': float(((sym - ref).abs().max() / ref.abs().max()).item())}
        for symmetric in (False, True):
            wl.covariance_band(0, nlat, method='separable', symmetric=symmetric)      # warm-up with the same workspace sizes
            wl.synchronize()
            t0 = time.perf_counter()
            sep = wl.covariance_band(0, nlat, method='separable', symmetric=symmetric)
            wl.synchronize()
            dt = time.perf_counter() - t0
            key = 'separable_variant_symmetric' if symmetric else 'separable_variant'
            out[key] = {'full_grid_seconds': dt, 'executed_GFLOP': (2.0 * nlat * P * P + 2.0 * nlat * nlon * (2 * COV_DEGREE + 1) ** 2) / 1e9}
            if total == nlat:
                out[key]['max_rel_diff_vs_general'] = float(((sep - sigma).abs().max() / sigma.abs().max()).item())
            del sep
    if world == 1 and args.cov_cpu_parallels > 0 and hasattr(wl, 'covariance_cpu'):
        out['cpu_baseline'] = wl.covariance_cpu(0, args.cov_cpu_parallels, sigma)
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and 'RANK' not in os.environ and int(os.environ.get('WORLD_SIZE', '1')) == 1:
        raise SystemExit(launch_ranks(args, argv))
    run_rank(args)


if __name__ == '__main__':
    main()
