#!/usr/bin/env python3
"""
Benchmark of the hot path (BASELINE.json).  Headline: batched synthesis of 240 monthly d/o-96 solutions to a 0.25 degree
GeographicGrid (kernel 'ewh') on MI355X; beside it one leg per remaining BASELINE configuration:

    covariance   config 4   d/o-180 covariance propagation to the 0.5 degree grid (second half of the metric), fp64 MFMA
    analysis     (a13)      d/o 96 <- 0.5 degree grid, 240 epochs (the reference's 142 s case), HBM
    filters      config 3   DDK5-type filter of a d/o-120 series of 240 epochs: order-wise block form (HBM) and the full
                            normal-matrix multiply 14637^2 x 240 (fp64 MFMA)
    smoother     config 5   3650 daily d/o-40 epochs (d = 1681), block-tridiagonal normal equations: solution and covariance
                            blocks from one factorisation (fp64 MFMA); epochs sharded over the ranks, all_gather of separators

    python bench.py --gpus N --steps K --warmup W [--legs synthesis,covariance,analysis,filters,smoother]

One "step" of the headline = one pass of the hot path over one batch: 240 coefficient sets already resident in HBM ->
240 grids in HBM (shg_synthesis through the C ABI).  `value` is measured exactly as the contract says: W warm-up steps, then K
timed steps between barrier + synchronize pairs, max over ranks.  Leg order (also in `config.leg_order`): synthesis setup, the timed
repeats of the covariance leg, the timed steps of the filters leg (block form, then the dense form: config 3's full-matrix products), THEN
the synthesis contract pass.  Why: behind an idle period -- or behind work that loads only the MFMA units or only the HBM -- launches
6 .. 25 of the synthesis kernel run 4 .. 13 % slower than its steady state (the card's power management settling on the new load;
tools/ramp_probe.py); behind a few products that load both, as the dense filter's do, they run at the steady time.  All of it is
declared, reported work of other BASELINE configurations; nothing runs unreported.  `value_after_ramp` repeats the measurement
behind `--ramp` further untimed launches (with `--legs synthesis` alone it is the only figure taken in the steady state).

N > 1: one process per GPU.  Started under torch.distributed.run (RANK / WORLD_SIZE in the environment) the script is a
rank; started bare (`python bench.py --gpus 4`) it launches `python -m torch.distributed.run --nproc-per-node N` on
itself as a child process BEFORE anything touches the GPU and exits with the child's code.
  * synthesis, analysis, filters: epochs are independent, every rank works on its own 240 epochs (weak scaling, no data-path
    collective); `value` = epochs of all ranks / max-over-ranks time;
  * covariance: the WHOLE 360 x 720 grid, the parallels split into contiguous latitude bands, Sigma replicated, one all_gather of
    the per-band sigma (RCCL); the checksum of the gathered grid is the same for every N (strong scaling);
  * smoother: the 3650 epochs split into contiguous ranges (strong scaling), nested dissection with all_gathers of separator
    blocks (grates_amd.distributed.smooth_block_tridiagonal_partitioned).

Rank 0 prints ONE JSON line with the contract fields plus, for the headline and inside every leg,
  roofline      the leg's dominant kernel(s) against the HBM or fp64 MFMA roofline; kernel time from HIP events recorded on the
                launching stream inside the timed region
  cpu_baseline  the CPU oracle (oracle/, same formulation as the reference) timed on a bounded sample on this host (N = 1 only)
  check         the TIMED OUTPUT BUFFER compared with the oracle on a sample, or a size-independent property of it
The oracle is only the checker and the CPU baseline here; nothing under oracle/ is on the measured path.
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
ANALYSIS_KERNELS = ['analysis_transform_kernel', 'analysis_operator_parity_kernel']
BLOCK_FORM_KERNELS = ['order_major_kernel<true>', 'orderwise_filter_om_kernel', 'order_major_kernel<false>']      # of the filters leg's block form
MFMA_F64_PEAK_TFLOPS = 78.6     # MI355X dense fp64 matrix peak (spec); measured 77.3 TFLOP/s (profiles/r01_microbench.txt)
GM, R_EARTH = 3.9860044150e+14, 6.3781363000e+06
ANA_DEGREE, ANA_GRID_STEP = 96, 0.5
DDK_DEGREE, DDK_LEVEL_SCALE = 120, 1e11          # config 3: DDK5 weights 1e11 n^4 (grates/filter.py:334-349)
SMOOTHER_DIM, SMOOTHER_EPOCHS = 1681, 3650       # config 5: d/o 40 state, ten years of daily solutions
LEGS = ('synthesis', 'covariance', 'analysis', 'filters', 'smoother')
FORBIDDEN_ENVIRONMENT = ('SHG_LIBRARY', 'SHG_DEBUG', 'SHG_TIMELINE_PTR', 'SHG_STAGGER', 'SHG_STAGGER2')


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--legs', default='all', help='comma separated subset of ' + ','.join(LEGS) + " (default: all; 'synthesis' always runs)")
    ap.add_argument('--ramp', type=int, default=300, help='untimed launches before the second measurement of the headline (device clock ramp)')
    ap.add_argument('--idle-pass', type=int, default=1, help='1: also time the contract pass straight behind the idle setup (roofline.value_idle_start)')
    ap.add_argument('--stage-limit-pass', type=int, default=1, help='1: also time the headline with the Legendre-stage limit on (roofline.value_stage_limit_on)')
    ap.add_argument('--api-chain', type=int, default=1, help='1: also time filter(TimeSeries) -> to_grid through the classes (roofline.api_chain_solutions_per_s)')
    ap.add_argument('--epochs', type=int, default=EPOCHS, help='epochs per GPU per step (default: BASELINE config 2)')
    ap.add_argument('--chunk', type=int, default=0, help='epochs per internal pass of the staged path (0 = library default)')
    ap.add_argument('--launch-timeout', type=float, default=1500.0, help='seconds after which `--gpus N` run without a launcher kills its ranks (0 = never)')
    ap.add_argument('--path', default='auto', choices=['auto', 'staged', 'fused', 'fused32', 'rot'], help='synthesis kernel path')
    ap.add_argument('--cpu-sample', type=int, default=16, help='solutions timed on the CPU baseline (0 = skip all CPU baselines)')
    ap.add_argument('--cov-parallels', type=int, default=-1,
                    help='parallels of the covariance leg over all ranks (-1 = the whole 0.5 degree grid, 360; 0 = skip the leg)')
    ap.add_argument('--cov-repeats', type=int, default=3, help='timed passes of the covariance leg')
    ap.add_argument('--cov-cpu-parallels', type=int, default=8, help='parallels of the covariance CPU baseline (0 = skip)')
    ap.add_argument('--cov-extensions', type=int, default=1, help='1: also time the symmetric and separable variants (N = 1 only)')
    ap.add_argument('--smoother-epochs', type=int, default=SMOOTHER_EPOCHS, help='epochs of the smoother leg over all ranks (BASELINE config 5: 3650)')
    ap.add_argument('--smoother-repeats', type=int, default=2, help='timed passes of the smoother leg (first call, repeated call)')
    ap.add_argument('--smoother-cpu-epochs', type=int, default=12, help='epochs of the short chain the CPU oracle solves (baseline + spot check)')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for N > 1 ('nccl' = RCCL; 'gloo' only to rehearse on one GPU)")
    ap.add_argument('--same-device', action='store_true', help='rehearsal: map every rank to cuda:0 (with --backend gloo)')
    args = ap.parse_args(argv)
    wanted = LEGS if args.legs == 'all' else tuple(x.strip() for x in args.legs.split(',') if x.strip())
    unknown = [x for x in wanted if x not in LEGS]
    if unknown:
        ap.error('unknown leg(s) ' + ', '.join(unknown))
    args.leg_set = set(wanted) | {'synthesis'}
    if args.cov_parallels == 0:
        args.leg_set.discard('covariance')
    return args


def refuse_experiment_environment(environ=None):
    """The shipping library has no run-time switches (tests/test_boundary.py); the variables that used to select a profiling
    build or knock parts of a kernel out must not even be present, so that a line can never stem from such a run."""
    environ = os.environ if environ is None else environ
    present = [v for v in FORBIDDEN_ENVIRONMENT if v in environ]
    if present:
        raise SystemExit('bench.py refuses to run with ' + ', '.join(present) + ' set: experiment switches / library overrides are not '
                         'part of the product (unset them; profiling tools live in tools/).')


# ---------------------------------------------------------------------------------------------------------------------
# launcher
# ---------------------------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks through torch.distributed.run as a child process.
    Nothing in this process has touched the GPU (no torch import so far).  The child runs in a session of its own; when it has
    not ended after --launch-timeout seconds the whole session is killed and the exit code is 124.  A rank that fails prints
    its number and traceback (run_rank_reported) and exits non-zero, torch.distributed.run then ends the other ranks and
    returns non-zero itself."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return run_child(cmd, env, args.launch_timeout, '{0} ranks'.format(args.gpus))


def run_child(cmd, env, timeout, what):
    """Run `cmd` in a session of its own and return its exit code; 124 after killing the whole session when it has not ended
    within `timeout` seconds (0 or less: no limit)."""
    import signal
    child = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return child.wait(timeout=timeout if timeout > 0 else None)
    except subprocess.TimeoutExpired:
        sys.stderr.write('bench.py: the {0} did not finish within {1:.0f} s (--launch-timeout): killing them\n'.format(what, timeout))
        sys.stderr.flush()
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(child.pid, sig)
            except ProcessLookupError:
                break
            try:
                child.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        return 124
    except KeyboardInterrupt:
        os.killpg(child.pid, signal.SIGTERM)
        raise


# ---------------------------------------------------------------------------------------------------------------------
# synthetic inputs, seeded per unit so that every world size works on the same data
# ---------------------------------------------------------------------------------------------------------------------
def coefficient_batch(first_seed, count, max_degree):
    import numpy as np
    return np.stack([np.random.default_rng(first_seed + e).standard_normal((max_degree + 1, max_degree + 1)) * 1e-10 for e in range(count)])


def orderwise_normal_blocks(seed, nmax, scale=1e12):
    """Synthetic SPD order-wise normal blocks (G G^T) * 1e12 of the reference's shapes (data/__init__.py:111-117): order 0, then
    cosine and sine block of every order m, each indexed by degree m .. nmax."""
    import numpy as np
    rng = np.random.default_rng(seed)
    sizes = [nmax + 1] + [nmax + 1 - m for m in range(1, nmax + 1) for _ in (0, 1)]
    blocks = []
    for s in sizes:
        G = rng.standard_normal((s, s + 4))
        blocks.append((G @ G.T) * scale)
    return blocks


def smoother_blocks(t, d, gen, torch, engine):
    """Epoch t of the seeded config-5 system: N_tt = G G^T / d + 4 I (SPD), N_t,t+1 = R / d, right-hand side n_t ~ N(0, 1)."""
    gen.manual_seed(50_000 + t)
    G = torch.randn((d, d + 8), dtype=torch.float64, device='cuda', generator=gen)
    D = engine.gemm(G, G, transb=True, alpha=1.0 / d)
    D.diagonal().add_(4.0)
    R = torch.randn((d, d), dtype=torch.float64, device='cuda', generator=gen) / d
    b = torch.randn((d, 1), dtype=torch.float64, device='cuda', generator=gen)
    return D, R, b


# ---------------------------------------------------------------------------------------------------------------------
# workloads: the GPU one (product path through the C ABI) and, in tests/test_bench_ranks.py, a stand-in of the same interface
# for the CPU test of the rank function: same sharding, barriers, gathers, checksums and JSON assembly, no GPU
# ---------------------------------------------------------------------------------------------------------------------
class GpuWorkload:
    device = 'cuda'

    def __init__(self, args, rank, world, local_rank):
        import torch
        self.torch = torch
        self.args, self.rank, self.world = args, rank, world
        torch.cuda.set_device(local_rank)
        import grates_amd as ga
        self.ga = ga

    def synchronize(self):
        self.torch.cuda.synchronize()

    def oracle_kernel(self):
        from oracle import shg_oracle as orc
        return orc, orc.KernelTable(KERNEL, self.ga.data.load_love_numbers()[0])

    # ---- synthesis (headline)
    def setup_synthesis(self):
        ga, torch = self.ga, self.torch
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
        self.batch_host = coefficient_batch(1000 + self.rank * B, B, MAX_DEGREE)
        self.batch = torch.from_numpy(self.batch_host).cuda()
        self.out = torch.empty((B, self.nlat, self.nlon), dtype=torch.float64, device='cuda')
        info = self.plan.info()
        rot = bool(info['rotation_symmetry']) and self.args.path in ('auto', 'rot')
        self.kernel_name = ('synthesis_rot_kernel' if rot else 'synthesis_fused_kernel') if info['fused'] else 'lon_stage_kernel<4>'
        self.config = {'fused_kernel': info['fused'], 'fourfold_symmetry': info['fourfold_symmetry'], 'rotation_folded_kernel': rot,
                       'rotations': info['rotations'] if rot else 0}

    def synthesis_step(self):
        self.plan.synthesis(self.batch, out=self.out)

    def set_stage_limit(self, limit):
        self.plan.set_stage_limit(limit)

    def api_chain(self, ctx):
        """The same 240 x d/o 96 -> 0.25 degree workload through the reference's CLASSES on a device-resident series:
        filter.OrderWiseFilter.filter(TimeSeries) (DDK5-type blocks of d/o 96) -> TimeSeries.to_grid(grid, 'ewh', as_tensor=True); the
        coefficients never pass through host arrays (round 5: the class API moved every epoch over PCIe, <= 7.5 k solutions/s)."""
        import datetime
        import numpy as np
        ga, args = self.ga, self.args
        B = args.epochs
        weights = DDK_LEVEL_SCALE * np.arange(MAX_DEGREE + 1, dtype=float) ** 4
        weights[0] = 1
        flt = ga.filter.OrderWiseFilter(ga.engine.ddk_blocks(orderwise_normal_blocks(44, MAX_DEGREE), weights))
        epochs = [datetime.datetime(2002, 4, 1) + datetime.timedelta(days=30 * e) for e in range(B)]
        ts = ga.gravityfield.TimeSeries.from_series(self.batch, epochs, GM, R_EARTH)
        state = {}

        def step():
            state['grids'] = flt.filter(ts).to_grid(self.grid, KERNEL, as_tensor=True)
        step()
        elapsed, _, step_ms = ctx.timed(step, max(args.warmup // 2, 1), max(args.steps // 2, 1), events=True)
        steps = max(args.steps // 2, 1)
        # one epoch against the per-epoch calls of the same classes (bit-identical by construction: tests/test_gpu_filters.py)
        e = B // 2
        gf = ga.gravityfield.PotentialCoefficients(GM, R_EARTH)
        gf.anm = self.batch_host[e].copy()
        single = flt.filter(gf).to_grid(self.grid, kernel=KERNEL).value_array
        same = bool(np.array_equal(state['grids'][e].cpu().numpy(), single))
        del state['grids']
        self.torch.cuda.empty_cache()
        return {'what': 'OrderWiseFilter.filter(TimeSeries) -> TimeSeries.to_grid(as_tensor=True) on the device-resident series, {0} epochs d/o {1} -> {2} deg'.format(B, MAX_DEGREE, GRID_STEP),
                'solutions_per_s': ctx.world * B * steps / elapsed, 'ms_per_step': 1e3 * elapsed / steps, 'gpu_ms_per_step': step_ms,
                'bit_identical_to_per_epoch_calls': same}

    def profile(self, enable):
        # events around the dominant kernel only: the pair around the 16 us coefficient repack would cost every step another ~5 us
        self.plan.profile(enable, kinds=('lon_stage',) if self.plan.info()['fused'] else None)

    def profile_read(self):
        return self.plan.profile_read()

    def release_synthesis(self):
        del self.out, self.batch
        self.torch.cuda.empty_cache()

    def synthesis_check(self, sample_epochs, timed):
        """The oracle (N+1 dgemms per solution like the reference) on the first `sample_epochs` solutions: compared with the
        grids the timed steps left in the output buffer, and (timed=True) the CPU baseline."""
        import numpy as np
        orc, ker = self.oracle_kernel()
        grid = self.grid
        if timed:
            orc.synthesis_regular(self.batch_host[0], grid.meridians, grid.parallels, ker)      # warm BLAS / page in
        worst = 0.0
        t0 = time.perf_counter()
        refs = [orc.synthesis_regular(self.batch_host[e], grid.meridians, grid.parallels, ker) for e in range(sample_epochs)]
        dt = time.perf_counter() - t0
        for e, ref in enumerate(refs):
            got = self.out[e].cpu().numpy()
            worst = max(worst, float(np.max(np.abs(got - ref)) / np.max(np.abs(ref))))
        check = {'max_rel_err_vs_oracle': worst, 'epochs_checked': sample_epochs, 'tolerance': 1e-12, 'ok': bool(worst < 1e-12),
                 'what': 'grids of the timed output buffer against the NumPy oracle'}
        baseline = None
        if timed:
            baseline = {'value': sample_epochs / dt, 'unit': 'solutions/s', 'cores': blas_threads(), 'kind': 'port',
                        'sample': '{0} of the {1} d/o-{2} epochs -> {3} deg grid, NumPy oracle (reference formulation), {4:.1f} s'.format(
                            sample_epochs, self.args.epochs, MAX_DEGREE, GRID_STEP, dt)}
        return check, baseline

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

    def release_covariance(self):
        del self.cov, self.cov_plan
        self.ga.engine.release_scratch()
        self.torch.cuda.empty_cache()

    def covariance_cpu(self, lat0, count, sigma_gpu):
        import numpy as np
        orc, ker = self.oracle_kernel()
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

    # ---- analysis: RegularGrid.to_potential_coefficients, grates/grid.py:752-790
    def leg_analysis(self, ctx):
        import numpy as np
        ga, torch, args = self.ga, self.torch, self.args
        N, B = ANA_DEGREE, args.epochs
        grid = ga.grid.GeographicGrid(ANA_GRID_STEP, ANA_GRID_STEP)
        nlat, nlon = grid.parallels.size, grid.meridians.size
        colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel(KERNEL), N, grid.parallels, GM, R_EARTH, grid.semimajor_axis, grid.flattening)
        plan = ga.engine.Plan(N, colat, kn, grid.meridians)
        batch_host = coefficient_batch(20_000 + self.rank * B, B, N)
        batch = torch.from_numpy(batch_host).cuda()
        grids = plan.synthesis(batch)                                         # band-limited input fields, resident in HBM
        area = ga.engine.to_device(grid.area.reshape(nlat, nlon))
        state = {}

        def step():
            state['out'] = plan.analysis(grids, area, 0)
        step()                                                                # builds and caches the per-order operators
        elapsed, prof, _ = ctx.timed(step, args.warmup, args.steps, plan)
        out = state['out']
        if ctx.rank != 0:
            return None
        k_lon, k_solve = prof.get('analysis_lon', (0.0, 0)), prof.get('analysis_solve', (0.0, 0))
        per_call_ms = (k_lon[0] + k_solve[0]) / max(args.steps, 1)
        per_epoch = 8 * (nlat * nlon + (N + 1) ** 2)                          # values read once, coefficients written once
        achieved = per_epoch * B / (per_call_ms * 1e-3) / 1e9 if per_call_ms > 0 else None
        # MFMA floor of the transform as built (4-fold folded: 2 nlon/4 (2N+1) flops per row) plus the operator product on the rows that exist
        mfma_flops = 2.0 * B * nlat * (nlon / 4.0) * (2 * N + 1) + 2.0 * B * nlat * (N + 1) ** 2
        roundtrip = float(((out - batch).abs().max() / batch.abs().max()).item())
        leg = {
            'metric': 'analysis d/o {0} <- {1} deg grid ({2}x{3}), area-weighted least squares per order'.format(N, ANA_GRID_STEP, nlat, nlon),
            'value': ctx.world * B * args.steps / elapsed, 'unit': 'epochs/s', 'n_gpus': ctx.world, 'scaling': 'weak',
            'ms_per_step': 1e3 * elapsed / args.steps, 'dtype': 'f64',
            'config': {'workload': '{0} epochs per GPU, grids resident in HBM, kernel {1}, min_degree 0'.format(B, KERNEL), 'max_degree': N,
                       'grid': [nlat, nlon], 'epochs_per_gpu': B},
            'roofline': {'kernel': 'analysis_transform_kernel + analysis_operator_parity_kernel (north-south parity split of the operator product)', 'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS if achieved else None,
                         'traffic': pmc_traffic('analysis', ANALYSIS_KERNELS)[0], 'traffic_source': pmc_traffic('analysis', ANALYSIS_KERNELS)[1],
                         'algorithmic_bytes_per_launch': per_epoch * B, 'avg_launch_ms': per_call_ms,
                         'kernels': {'analysis_transform_kernel': {'avg_us': 1e3 * k_lon[0] / max(k_lon[1], 1), 'launches': int(k_lon[1])},
                                     'analysis_operator_parity_kernel': {'avg_us': 1e3 * k_solve[0] / max(k_solve[1], 1), 'launches': int(k_solve[1])}},
                         'mfma_floor_ms': mfma_flops / (MFMA_F64_PEAK_TFLOPS * 1e12) * 1e3},
            'check': {'roundtrip_max_rel_err': roundtrip, 'tolerance': 1e-11, 'ok': bool(roundtrip < 1e-11),
                      'what': 'coefficients of the timed output buffer against the band-limited input of all {0} epochs'.format(B)},
        }
        if ctx.world == 1 and args.cpu_sample > 0:
            orc, ker = self.oracle_kernel()
            orders = list(range(0, N + 1, 12))                                # 9 of the 97 independent per-order least-squares problems
            values = grids[0].cpu().numpy().ravel()
            t0 = time.perf_counter()
            ref = orc.analysis_regular(values, grid.area, 0, N, grid.meridians, grid.parallels, ker, orders=orders)
            dt = time.perf_counter() - t0
            cost = lambda ms: sum((1 if m == 0 else 2) * (N + 1 - m) for m in ms)      # noqa: E731  columns of the design matrices built
            full = dt * cost(range(N + 1)) / cost(orders)
            got = out[0].cpu().numpy()
            mask = np.zeros((N + 1, N + 1), dtype=bool)
            for m in orders:
                mask[m:, m] = True
                if m:
                    mask[m - 1, m:] = True
            err = float(np.max(np.abs(got - ref)[mask]) / np.max(np.abs(ref)))
            leg['check'].update({'max_rel_err_vs_oracle': err, 'orders_checked': orders, 'ok': bool(roundtrip < 1e-11 and err < 1e-11)})
            leg['cpu_baseline'] = {'value': 1.0 / full, 'unit': 'epochs/s', 'cores': blas_threads(), 'kind': 'port',
                                   'sample': 'orders {0} of one epoch (NumPy oracle: design matrix + weighted normal equations per order, grid.py:665-696), '
                                             '{1:.1f} s, scaled by the design-matrix columns of all orders ({2:.0f} s per epoch)'.format(orders, dt, full)}
        return leg

    # ---- filters: OrderWiseFilter / DDK and GeneralMatrix, grates/filter.py:153-222, 456-479
    def leg_filters(self, ctx):
        return self.leg_filters_report(ctx, self.leg_filters_timed(ctx))

    def leg_filters_timed(self, ctx):
        """The device part of the filters leg: setup, the timed steps of the block form, then those of the dense form (the last thing it
        does on the device).  Returns what the report needs."""
        import numpy as np
        ga, torch, args = self.ga, self.torch, self.args
        nmax, T = DDK_DEGREE, args.epochs
        normals = orderwise_normal_blocks(44, nmax)
        weights = DDK_LEVEL_SCALE * np.arange(nmax + 1, dtype=float) ** 4
        weights[0] = 1
        blocks = ga.engine.ddk_blocks(normals, weights)                      # (N_m + diag w)^-1 N_m on the device, grates/filter.py:334-349
        flt = ga.filter.OrderWiseFilter(blocks)
        batch_host = coefficient_batch(30_000 + self.rank * T, T, nmax)
        batch_host[:, 0:2, 0:2] = 0.0                                          # GRACE-type series carry no degree 0 / 1
        batch = torch.from_numpy(batch_host).cuda()
        state = {}

        def step_block():
            state['block'] = flt.filter_batch(batch)
        step_block()
        el_block, _, ev_block = ctx.timed(step_block, args.warmup, args.steps, events=True)
        # the same operator through the reference's classes on a device-resident series: filter.OrderWiseFilter.filter(TimeSeries) works on
        # the engine.OrderMajorSeries the TimeSeries holds (one product per block on whole matrices, no gather / scatter) and returns a
        # TimeSeries that stays on the device; the series is built once, outside the timed steps, like the batch above
        import datetime
        epochs = [datetime.datetime(2002, 4, 1) + datetime.timedelta(days=30 * e) for e in range(T)]
        ts = ga.gravityfield.TimeSeries.from_series(batch, epochs)

        def step_series():
            state['series'] = flt.filter(ts)
        step_series()
        el_series, _, ev_series = ctx.timed(step_series, args.warmup, args.steps, events=True)
        nmin = 2
        P = (nmax + 1) ** 2 - nmin ** 2
        dense = ga.filter.GeneralMatrix(flt.matrix(nmin, nmax), nmin, nmax)   # 14637 x 14637 full normal-type matrix, 1.7 GB

        def step_dense():
            state['dense'] = dense.filter(ts)                                  # ONE product W_om X on the series (W permuted once, no ravel / unravel)
        step_dense()
        dense_steps = max(args.steps // 2, 1)
        el_dense, _, ev_dense = ctx.timed(step_dense, max(args.warmup // 2, 1), dense_steps, events=True)
        return dict(nmax=nmax, T=T, nmin=nmin, P=P, blocks=blocks, batch_host=batch_host, dense=dense, state=state, el_block=el_block, ev_block=ev_block,
                    el_dense=el_dense, ev_dense=ev_dense, dense_steps=dense_steps, el_series=el_series, ev_series=ev_series)

    def leg_filters_report(self, ctx, st):
        """The line of the filters leg from the timed part's record, the two forms against each other and against the oracle, the CPU baselines."""
        import numpy as np
        args = self.args
        if ctx.rank != 0:
            return None
        nmax, T, nmin, P, blocks, batch_host, dense, state = (st[k] for k in ('nmax', 'T', 'nmin', 'P', 'blocks', 'batch_host', 'dense', 'state'))
        el_block, ev_block, el_dense, ev_dense, dense_steps = (st[k] for k in ('el_block', 'ev_block', 'el_dense', 'ev_dense', 'dense_steps'))
        block_bytes = 8.0 * (sum(b.size for b in blocks) + 2.0 * (nmax + 1) ** 2 * T)
        dense_flops = 2.0 * P * P * T
        dense_batch = state['dense'].to_device().to_batch()
        agree = float(((dense_batch - state['block']).abs().max() / state['block'].abs().max()).item())
        agree_series = float(((state['series'].to_device().to_batch() - state['block']).abs().max() / state['block'].abs().max()).item())
        el_series, ev_series = st['el_series'], st['ev_series']
        out = {
            'block': {
                'metric': 'order-wise (DDK5-type) filter of a d/o-{0} series, block form'.format(nmax), 'value': ctx.world * T * args.steps / el_block,
                'unit': 'epochs/s', 'n_gpus': ctx.world, 'scaling': 'weak', 'ms_per_step': 1e3 * el_block / args.steps, 'dtype': 'f64',
                'config': {'workload': '{0} epochs per GPU, {1} blocks, weights 1e11 n^4'.format(T, len(blocks)), 'max_degree': nmax, 'epochs_per_gpu': T},
                'roofline': {'kernel': 'order_major_kernel<true> + orderwise_filter_om_kernel + order_major_kernel<false> (batches of 64 epochs and more go '
                                       'through the order-major layout inside shg_orderwise_filter)', 'bound': 'hbm',
                             'achieved': block_bytes / (ev_block * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS,
                             'unit': 'GB/s', 'frac': block_bytes / (ev_block * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             'traffic': pmc_traffic('filters', BLOCK_FORM_KERNELS)[0], 'traffic_source': pmc_traffic('filters', BLOCK_FORM_KERNELS)[1],
                             'algorithmic_bytes_per_launch': block_bytes, 'avg_launch_ms': ev_block},
            },
            'order_major': {
                'metric': 'the same filter through filter.OrderWiseFilter.filter(TimeSeries) on the device-resident series (order-major layout, no gather / scatter)', 'value': ctx.world * T * args.steps / el_series,
                'unit': 'epochs/s', 'n_gpus': ctx.world, 'scaling': 'weak', 'ms_per_step': 1e3 * el_series / args.steps, 'dtype': 'f64',
                'roofline': {'kernel': 'orderwise_filter_om_kernel', 'bound': 'hbm', 'achieved': block_bytes / (ev_series * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS,
                             'unit': 'GB/s', 'frac': block_bytes / (ev_series * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             'traffic': pmc_traffic('filters', ['orderwise_filter_om_kernel'])[0],
                             'algorithmic_bytes_per_launch': block_bytes, 'avg_launch_ms': ev_series},
                'check': {'max_rel_diff_vs_block_form': agree_series, 'tolerance': 1e-12, 'ok': bool(agree_series < 1e-12)},
            },
            'dense': {
                'metric': 'the same filter as full normal-matrix multiply W X, W {0} x {0}'.format(P), 'value': ctx.world * T * dense_steps / el_dense,
                'unit': 'epochs/s', 'n_gpus': ctx.world, 'scaling': 'weak', 'ms_per_step': 1e3 * el_dense / dense_steps, 'dtype': 'f64',
                'GFLOPs': dense_flops * dense_steps / el_dense / 1e9,
                'config': {'workload': '{0} epochs per GPU: GeneralMatrix.filter(TimeSeries) = W [{1} x {1}] (+ 4 identity rows) @ X on the device-resident series'.format(T, P), 'max_degree': nmax, 'min_degree': nmin,
                           'epochs_per_gpu': T, 'flops_per_step': dense_flops},
                'roofline': {'kernel': 'gemm_tall_kernel + gemm_tall_fixup_kernel (one product per step on the order-major series: no ravel / unravel)', 'bound': 'mfma',
                             'achieved': dense_flops / (ev_dense * 1e-3) / 1e12, 'peak': MFMA_F64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                             'frac': dense_flops / (ev_dense * 1e-3) / 1e12 / MFMA_F64_PEAK_TFLOPS, 'mfma_busy': pmc_mfma_busy('filters_dense'),
                             'traffic': pmc_traffic('filters', ['gemm_tall_kernel', 'gemm_tall_fixup_kernel'])[0],
                             'traffic_source': pmc_traffic('filters', ['gemm_tall_kernel', 'gemm_tall_fixup_kernel'])[1],
                             'avg_launch_ms': ev_dense},
            },
            'check': {'dense_vs_block_max_rel_diff': agree, 'order_major_vs_block_max_rel_diff': agree_series, 'tolerance': 1e-12,
                      'ok': bool(agree < 1e-12 and agree_series < 1e-12), 'what': 'timed output buffers of the three forms against each other (all epochs)'},
        }
        if ctx.world == 1 and args.cpu_sample > 0:
            orc, _ = self.oracle_kernel()
            host_block = state['block'].cpu().numpy()
            t0 = time.perf_counter()
            refs = [orc.orderwise_filter(batch_host[e], blocks) for e in range(T)]
            dt = time.perf_counter() - t0
            err = max(float(np.max(np.abs(host_block[e] - refs[e])) / np.max(np.abs(refs[e]))) for e in range(T))
            out['block']['cpu_baseline'] = {'value': T / dt, 'unit': 'epochs/s', 'cores': blas_threads(), 'kind': 'port',
                                            'sample': 'all {0} epochs, NumPy oracle (2N+1 block mat-vecs per epoch, filter.py:182-187), {1:.2f} s'.format(T, dt)}
            W = dense.matrix(nmin, nmax)
            X = np.stack([orc.ravel_coefficients(batch_host[e], nmin, nmax) for e in range(T)], axis=1)
            W @ X[:, 0:8]                                                      # warm BLAS
            t0 = time.perf_counter()
            Y = W @ X
            dt = time.perf_counter() - t0
            host_dense = dense_batch.cpu().numpy()
            ref0 = orc.general_matrix_filter(batch_host[0], W, nmin, nmax)
            err_dense = max(float(np.max(np.abs(host_dense[0] - ref0)) / np.max(np.abs(ref0))),
                            float(np.max(np.abs(orc.ravel_coefficients(host_dense[T - 1], nmin, nmax) - Y[:, T - 1])) / np.max(np.abs(Y[:, T - 1]))))
            out['dense']['cpu_baseline'] = {'value': dense_flops / dt / 1e9, 'unit': 'GFLOP/s', 'cores': blas_threads(), 'kind': 'port',
                                            'sample': 'W @ X for all {0} epochs in one dgemm (NumPy oracle, filter.py:473-477), {1:.2f} s'.format(T, dt)}
            out['check'].update({'block_max_rel_err_vs_oracle': err, 'dense_max_rel_err_vs_oracle': err_dense, 'epochs_checked_block': T,
                                 'ok': bool(agree < 1e-12 and agree_series < 1e-12 and err < 1e-12 and err_dense < 1e-12)})
        return out

    # ---- smoother: NormalEquations.solve + compute_covariance(sparse=True), grates/lstsq.py:950-968, 1026-1042
    def leg_smoother(self, ctx):
        import numpy as np
        from grates_amd import distributed as gd
        ga, torch, args = self.ga, self.torch, self.args
        d, T = SMOOTHER_DIM, args.smoother_epochs
        t_first, t_stop = gd.shard_range(T, ctx.rank, ctx.world)
        n_loc = t_stop - t_first
        last = ctx.rank == ctx.world - 1
        free, _ = torch.cuda.mem_get_info()
        per_rank_ranks = ctx.world if args.same_device else 1
        interior_extra = 0 if ctx.world == 1 else 2 * n_loc * d * (2 * d) * 8       # Y and Y Z_SS of the nested dissection
        need = (2 * n_loc * d * d * 8 + interior_extra) * per_rank_ranks + 24e9
        enough = ctx.all_ranks_agree(free >= need and n_loc >= 2)
        if not enough:
            return {'skipped': 'config 5 needs {0:.0f} GB of free device memory per GPU at this world size, {1:.0f} GB are free (or fewer than '
                               'two epochs per rank)'.format(need / 1e9, free / 1e9)} if ctx.rank == 0 else None
        gen = torch.Generator(device='cuda')

        def build():
            diag, upper, rhs = [], [], []
            for t in range(t_first, t_stop):
                D, R, b = smoother_blocks(t, d, gen, torch, ga.engine)
                diag.append(D)
                rhs.append(b)
                if t + 1 < T:
                    upper.append(R)                                            # upper[-1] of a rank but the last couples to the next rank's first epoch
            return diag, upper, torch.cat(rhs, dim=0)

        seconds, phases = [], []
        x = zd = zu = None
        for _ in range(max(args.smoother_repeats, 1)):
            x = zd = zu = None                                                # the previous pass's blocks go back to the allocator first
            diag, upper, rhs = build()
            tm = {}
            ctx.barrier()
            t0 = time.perf_counter()
            x, zd, zu = gd.smooth_block_tridiagonal_partitioned(diag, upper, rhs, consume=True, timings=tm)
            ctx.barrier()
            seconds.append(ctx.max_over_ranks(time.perf_counter() - t0))
            phases.append({k: ctx.max_over_ranks(v) for k, v in sorted(tm.items())})
            del diag, upper
        elapsed = seconds[-1]

        # ---- the timed output: residual of the solution over the whole chain (blocks regenerated from their seeds; the solution of the
        # neighbouring epochs across rank boundaries comes through one small all_gather), identity (N N^-1)_tt = I at sample epochs
        edges = ctx.gather_rows([x[:d, 0].contiguous(), x[-d:, 0].contiguous()])
        before = edges[ctx.rank - 1][1] if ctx.rank > 0 else None
        after = edges[ctx.rank + 1][0] if not last else None
        num = torch.zeros((), dtype=torch.float64, device='cuda')
        den = torch.zeros((), dtype=torch.float64, device='cuda')
        prev = smoother_blocks(t_first - 1, d, gen, torch, ga.engine)[1] if ctx.rank > 0 else None
        for k, t in enumerate(range(t_first, t_stop)):
            D, R, b = smoother_blocks(t, d, gen, torch, ga.engine)
            own = x[k * d:(k + 1) * d]
            Nx = ga.engine.gemm(D, own)
            if t + 1 < T:
                nxt = x[(k + 1) * d:(k + 2) * d] if k + 1 < n_loc else after.reshape(d, 1)
                Nx += ga.engine.gemm(R, nxt.contiguous())
            if prev is not None:
                prv = x[(k - 1) * d:k * d] if k > 0 else before.reshape(d, 1)
                Nx += ga.engine.gemm(prev, prv.contiguous(), transa=True)
            num += ((Nx - b) ** 2).sum()
            den += (b ** 2).sum()
            prev = R
        sums = ctx.sum_over_ranks([float(num.item()), float(den.item()), float(x.sum().item())])
        residual = float(np.sqrt(sums[0] / sums[1]))
        eye = torch.eye(d, dtype=torch.float64, device='cuda')
        worst_id = worst_sym = 0.0
        for k in sorted({1, n_loc // 2, max(n_loc - 3, 1)}):
            if not 1 <= k < n_loc - 1:
                continue
            t = t_first + k
            Z = zd[k]
            worst_sym = max(worst_sym, float(((Z - Z.t()).abs().max() / Z.abs().max()).item()))
            D, R, _ = smoother_blocks(t, d, gen, torch, ga.engine)
            acc = ga.engine.gemm(D, Z) + ga.engine.gemm(R, zu[k], transb=True)
            acc += ga.engine.gemm(smoother_blocks(t - 1, d, gen, torch, ga.engine)[1], zu[k - 1], transa=True)
            worst_id = max(worst_id, float((acc - eye).abs().max().item()))
        worst_id, worst_sym = ctx.max_over_ranks(worst_id), ctx.max_over_ranks(worst_sym)
        del x, zd, zu
        torch.cuda.empty_cache()
        if ctx.rank != 0:
            return None
        flops_factor, flops_inverse = 7.0 / 3.0 * d ** 3, 13.0 / 3.0 * d ** 3
        flops = T * (flops_factor + flops_inverse)
        ph = phases[-1]
        leg = {
            'metric': 'block-tridiagonal normal-equation smoother: solution + covariance blocks of {0} epochs x {1} parameters'.format(T, d),
            'value': T / elapsed, 'unit': 'epochs/s', 'n_gpus': ctx.world, 'scaling': 'strong', 'dtype': 'f64',
            'seconds': elapsed, 'seconds_all': seconds, 'phases_s': ph, 'phases_all': phases,
            'config': {'workload': 'N_tt = G G^T / d + 4 I, N_t,t+1 = R / d seeded per epoch on the device, one right-hand side; '
                                   'grates_amd.distributed.smooth_block_tridiagonal_partitioned(consume=True)', 'epochs': T, 'dim': d,
                       'epochs_per_rank': ctx.all_counts, 'parallelism': 'epochs in {0} contiguous range(s), all_gather of separator blocks'.format(ctx.world),
                       'flops_per_epoch': {'factor (potrf d^3/3 + U^-T A d^3 + Schur update d^3)': flops_factor,
                                           'sparse inverse (U^-1 W d^3 + T Z 2 d^3 + U^-1 U^-T d^3/3 + T Z^T d^3)': flops_inverse},
                       'flops': flops,
                       # which factorisation path the main thread's chain took: chain rows carry their coupling block through the panel
                       # sweep only when the queue experiment found two side streams with hardware queues of their own (csrc/plan.hip)
                       'lookahead': ga.engine.block_lookahead_info()},
            'roofline': {'kernel': 'gemm_ex_kernel + leaf_kernel (block Cholesky, sweeps and Takahashi recursion of one chain)', 'bound': 'mfma',
                         'achieved': flops / elapsed / 1e12, 'peak': MFMA_F64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': flops / elapsed / 1e12 / MFMA_F64_PEAK_TFLOPS,
                         'traffic': None, 'traffic_note': 'MFMA-bound leg of ~650 000 launches; per-kernel HBM-side bytes of a 64-epoch chain: profiles/r04_pmc_traffic.json (legs.smoother)',
                         'factor_TFLOPs': T * flops_factor / ph['factor_s'] / 1e12 if ph.get('factor_s') else None,
                         'factor_frac': T * flops_factor / ph['factor_s'] / 1e12 / MFMA_F64_PEAK_TFLOPS if ph.get('factor_s') else None,
                         'sparse_inverse_TFLOPs': T * flops_inverse / ph['covariance_s'] / 1e12 if ph.get('covariance_s') else None,
                         'note': 'algorithmic flops of the single-chain formulation; with N > 1 the nested dissection executes more'},
            'check': {'residual': residual, 'identity_defect_max': worst_id, 'covariance_asymmetry_max': worst_sym, 'solution_checksum': sums[2],
                      'tolerance': {'residual': 1e-13, 'identity': 1e-12}, 'ok': bool(residual < 1e-13 and worst_id < 1e-12 and worst_sym < 1e-13),
                      'what': '||N x - n|| / ||n|| of the timed solution with N regenerated from the seeds; (N N^-1)_tt = I and symmetry of timed covariance blocks'},
        }
        if ctx.world == 1 and args.cpu_sample > 0 and args.smoother_cpu_epochs >= 2:
            leg['cpu_baseline'], spot = self.smoother_cpu(min(args.smoother_cpu_epochs, T), d, gen)
            leg['check'].update(spot)
            leg['check']['ok'] = bool(leg['check']['ok'] and spot['short_chain_max_rel_err_vs_oracle'] < 1e-9)
        return leg

    def smoother_cpu(self, n, d, gen):
        """The first `n` epochs of the same seeded system as a chain of their own: the oracle (block Cholesky, two sweeps, Takahashi
        recursion in the reference formulation with LAPACK per block) is the CPU baseline, and the product path run on that short
        chain is compared with it."""
        import numpy as np
        from grates_amd import distributed as gd
        from oracle import lstsq_oracle as lo
        ga, torch = self.ga, self.torch
        sets = [smoother_blocks(t, d, gen, torch, ga.engine) for t in range(n)]
        host = [(D.cpu().numpy(), R.cpu().numpy(), b.cpu().numpy()) for D, R, b in sets]
        x, zd, zu = gd.smooth_block_tridiagonal_partitioned([s[0] for s in sets], [s[1] for s in sets[:-1]], torch.cat([s[2] for s in sets], dim=0))
        bm = lo.block_matrix(np.arange(0, (n + 1) * d, d))
        for t, (D, R, _) in enumerate(host):
            bm['blocks'][(t, t)] = D.copy()
            if t + 1 < n:
                bm['blocks'][(t, t + 1)] = R.copy()
        rhs = np.vstack([h[2] for h in host])
        t0 = time.perf_counter()
        lo.cholesky(bm)
        ref_x = lo.solve_triangular(bm, lo.solve_triangular(bm, rhs, transpose=True))
        lo.sparse_inverse(bm)
        dt = time.perf_counter() - t0
        err = float(np.max(np.abs(x.cpu().numpy() - ref_x)) / np.max(np.abs(ref_x)))
        for t in (0, n // 2, n - 1):
            ref = bm['blocks'][(t, t)]
            err = max(err, float(np.max(np.abs(zd[t].cpu().numpy() - ref)) / np.max(np.abs(ref))))
        baseline = {'value': n / dt, 'unit': 'epochs/s', 'cores': blas_threads(), 'kind': 'port',
                    'sample': 'chain of the first {0} epochs at d = {1}: block Cholesky, forward / backward sweep and sparse inverse of the NumPy / LAPACK '
                              'oracle (lstsq.py:698-846), {2:.1f} s'.format(n, d, dt)}
        return baseline, {'short_chain_max_rel_err_vs_oracle': err, 'short_chain_epochs': n, 'short_chain_tolerance': 1e-9}


def blas_threads():
    """Threads the NumPy BLAS of the CPU baseline runs on (one convention for all legs)."""
    try:
        import threadpoolctl
        return int(max([p.get('num_threads', 1) for p in threadpoolctl.threadpool_info()] or [1]))
    except Exception:
        return int(os.cpu_count() or 1)


def pmc_traffic(leg, kernels):
    """HBM-side bytes per launch of the named kernels of a leg (summed) from the committed rocprofv3 --pmc summary of the same workload
    (profiles/r04_pmc_traffic.json, tools/pmc_legs.sh), with its origin: counters cannot be collected inside a timed run.  A kernel
    name matches when the recorded name starts with it.  -> (bytes or None, source or None)"""
    table = rows = summary = None
    for name in PMC_TRAFFIC_SUMMARIES:                                      # the newest summary that holds the leg
        path = os.path.join(ROOT, 'profiles', name)
        try:
            with open(path) as f:
                table = json.load(f)
            rows, summary = table['legs'][leg], name
            break
        except Exception:
            rows = None
    if rows is None:
        return None, None
    # counter figures are quoted only while the kernel sources they were measured on are unchanged (hashes recorded in the summary)
    if not _source_hashes().unchanged(_get(table, 'sources', leg), leg):
        return None, 'profiles/{0}: the kernel sources of this leg have changed since these counters were collected (build of {1}): not quoted'.format(
            summary, table.get('commit', 'unrecorded'))
    total, found = 0.0, []
    for want in kernels:
        hit = [(kname, r) for kname, r in rows.items() if kname.startswith(want)]
        if not hit:
            return None, None
        kname, r = max(hit, key=lambda kv: kv[1]['dispatches'])
        total += r['bytes']
        found.append(kname.split('<')[0])
    return total, 'profiles/' + summary + ' ({0}; {1}; build of {2}): {3}'.format(table.get('source', ''), table.get('correction', ''),
                                                                                     table.get('commit', 'unrecorded'), ' + '.join(found))


PMC_TRAFFIC_SUMMARIES = ('r06_pmc_traffic.json',)
PMC_BUSY_SUMMARIES = ('r06_mfma_busy.json',)


def _source_hashes():
    tools = os.path.join(ROOT, 'tools')
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import source_hashes
    return source_hashes


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or d.get(k) is None:
            return None
        d = d[k]
    return d


LEG_CODES = (('synthesis setup', 'setup'), ('synthesis pass behind the idle setup', 'idle_pass'), ('covariance (timed repeats)', 'covariance'),
             ('filters (timed steps: block form, then dense form)', 'filters'), ('synthesis contract pass', 'CONTRACT_PASS'),
             ('synthesis pass behind', 'ramp+pass'), ('synthesis check + CPU baseline', 'check+cpu'),
             ('covariance extensions + CPU baseline', 'cov_ext+cpu'), ('filters checks + CPU baselines', 'filt_check+cpu'))


def summarise(line):
    """The driver's record keeps the contract's scalar fields and the scalars of `config`, `roofline` and `cpu_baseline` (strings cut
    at ~140 characters); nested objects and further top-level keys are dropped.  So the figures of every leg that the metric names or the
    review reads -- the second half of the metric first: covariance GFLOP/s with its fraction of the fp64 MFMA peak -- are repeated as
    scalars of those three objects, the leg order as ONE short string, the grid as 'nlat x nlon'."""
    cfg, roof = line['config'], line['roofline']
    order = []
    for item in cfg.get('leg_order', []):
        order.append(next((code for text, code in LEG_CODES if item.startswith(text)), item))
    cfg['leg_order'] = '>'.join(order)
    if isinstance(cfg.get('grid'), (list, tuple)):
        cfg['grid'] = 'x'.join(str(v) for v in cfg['grid'])
    cfg['legs'] = ','.join(cfg.get('legs', [])) if isinstance(cfg.get('legs'), (list, tuple)) else cfg.get('legs')
    if isinstance(cfg.get('process_group'), dict):
        cfg['process_group'] = '{backend} world {world}'.format(**cfg['process_group'])
    after = roof.pop('after_ramp', None) or {}
    roof['frac_after_ramp'] = after.get('frac')
    roof['avg_launch_ms_after_ramp'] = after.get('avg_launch_ms')
    roof['value_after_ramp'] = line.get('value_after_ramp')
    cov, ana, flt, smo = (line.get(k) if isinstance(line.get(k), dict) else None for k in ('covariance', 'analysis', 'filters', 'smoother'))
    if cov:
        cfg['covariance_workload'] = 'd/o {0} (P={1}) -> {2} deg grid, {3} of {4} parallels, {5} band(s)'.format(
            COV_DEGREE, _get(cov, 'config', 'P'), COV_GRID_STEP, _get(cov, 'config', 'parallels'), _get(cov, 'config', 'nlat'), cov.get('n_gpus'))
        roof['covariance_GFLOPs'] = cov.get('value')
        roof['covariance_seconds'] = cov.get('seconds_median')
        roof['covariance_kernel_TFLOPs'] = _get(cov, 'roofline', 'achieved')
        roof['covariance_frac'] = _get(cov, 'roofline', 'frac')
        roof['covariance_whole_leg_frac'] = (cov['value'] / 1e3 / MFMA_F64_PEAK_TFLOPS / max(cov.get('n_gpus') or 1, 1)) if cov.get('value') else None
        roof['covariance_mfma_busy'] = _get(cov, 'roofline', 'mfma_busy')
        if isinstance(line.get('cpu_baseline'), dict) and isinstance(cov.get('cpu_baseline'), dict):
            line['cpu_baseline']['covariance_GFLOPs'] = cov['cpu_baseline'].get('value')
            line['cpu_baseline']['covariance_sample'] = cov['cpu_baseline'].get('sample')
    if ana:
        roof['analysis_frac'] = _get(ana, 'roofline', 'frac')
        roof['analysis_epochs_per_s'] = ana.get('value')
        t_, a_ = _get(ana, 'roofline', 'traffic'), _get(ana, 'roofline', 'algorithmic_bytes_per_launch')
        roof['analysis_traffic_ratio'] = (t_ / a_) if t_ and a_ else None
        roof['analysis_ms_per_call'] = _get(ana, 'roofline', 'avg_launch_ms')
    if flt:
        roof['filters_block_frac'] = _get(flt, 'block', 'roofline', 'frac')
        ms_ = _get(flt, 'block', 'roofline', 'avg_launch_ms')
        roof['filters_block_us_per_step'] = 1e3 * ms_ if ms_ else None
        ms_ = _get(flt, 'order_major', 'roofline', 'avg_launch_ms')
        roof['filters_order_major_frac'] = _get(flt, 'order_major', 'roofline', 'frac')
        roof['filters_order_major_us_per_step'] = 1e3 * ms_ if ms_ else None
        roof['filters_dense_frac'] = _get(flt, 'dense', 'roofline', 'frac')
        roof['filters_dense_TFLOPs'] = _get(flt, 'dense', 'roofline', 'achieved')
        roof['filters_dense_mfma_busy'] = _get(flt, 'dense', 'roofline', 'mfma_busy')
    if smo:
        roof['smoother_epochs_per_s'] = smo.get('value')
        roof['smoother_frac'] = _get(smo, 'roofline', 'frac')
        roof['smoother_factor_frac'] = _get(smo, 'roofline', 'factor_frac')
    roof['all_checks_ok'] = line.get('all_checks_ok')
    return line


LINE_LIMIT = 7900          # bytes: the driver keeps the last 8 KB of the output


def compact_line(line, limit=LINE_LIMIT):
    """The printed line within `limit` bytes: floats to 6 significant digits (`value`, `ms_per_step` to 9), one top-level
    `traffic_source` instead of one per leg, strings of the leg objects cut at 72 characters, their notes left out; if that is not
    enough the legs' per-repeat lists, `config` texts, CPU baselines and checks go, in that order (named in `dropped_for_line_limit`).  Contract fields, `config`, `roofline`, `cpu_baseline` are never dropped."""
    keep9 = {'value', 'ms_per_step'}

    kept = ('config', 'roofline', 'cpu_baseline')          # the objects whose scalars the driver keeps: their strings may have 140 characters

    def tidy(o, key=None, depth=0, wide=False):
        if isinstance(o, float):
            if key and 'checksum' in key:
                return o                                    # reproducibility checksums keep every digit
            return float('{0:.{1}g}'.format(o, 9 if key in keep9 and depth <= 1 else 6))
        if isinstance(o, dict):
            return {k: tidy(v, k, depth + 1, wide or (depth == 0 and k in kept)) for k, v in o.items()
                    if not (depth >= 1 and k in ('traffic_note', 'note', 'phases_all', 'flops_per_epoch'))}
        if isinstance(o, (list, tuple)):
            return [tidy(v, key, depth + 1, wide) for v in o]
        limit_ = 140 if wide and depth == 2 else 72
        if isinstance(o, str) and depth > 1 and len(o) > limit_:
            return o[:limit_ - 3] + '...'
        return o

    sources = []

    def strip_sources(o):
        if isinstance(o, dict):
            src = o.pop('traffic_source', None)
            if src and src not in sources:
                sources.append(src)
            for v in o.values():
                strip_sources(v)
    strip_sources(line)
    if sources:
        line['traffic_source'] = sources[0].split(': ')[0][:140]
    line = tidy(line)
    size = lambda o: len(json.dumps(o, separators=(',', ':')))          # noqa: E731
    legs_present = [leg for leg in ('smoother', 'filters', 'analysis', 'covariance') if isinstance(line.get(leg), dict)]

    def nodes(leg):                                  # the leg object and, for the filters leg, its two forms
        top = line[leg]
        return [top] + [v for k, v in top.items() if k in ('block', 'dense', 'order_major') and isinstance(v, dict)]
    # what goes first when the line is too long: per-repeat lists, descriptive texts, kernel tables; whole sub-objects only at the end
    stages = [[(n, 'seconds_all') for leg in legs_present for n in nodes(leg)],
              [(n.get('config'), key) for leg in legs_present for n in nodes(leg) for key in ('workload', 'sigma_recipe', 'parallelism', 'lookahead')],
              [(n.get('roofline'), 'kernels') for leg in legs_present for n in nodes(leg)] + [(line, 'kernels')],
              [(n.get('check'), 'what') for leg in legs_present for n in nodes(leg)] + [(n.get('cpu_baseline'), 'sample') for leg in legs_present for n in nodes(leg)],
              [(n, key) for key in ('config', 'cpu_baseline', 'check') for leg in legs_present for n in nodes(leg)],
              [(line, 'traffic_source')]]
    for stage in stages:
        for target, key in stage:
            if size(line) <= limit:
                break
            if isinstance(target, dict) and key in target:
                target.pop(key)
                line.setdefault('dropped_for_line_limit', []).append(key)
    return line


def pmc_mfma_busy(leg):
    """MFMA-busy share of a leg's dominant kernel from the committed SQ counter summary (tools/pmc_summary.sh, tools/pmc_mfma_busy.py):
    counters cannot be collected inside a timed run.  -> fraction or None"""
    for name in PMC_BUSY_SUMMARIES:
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as f:
                table = json.load(f)
            if not _source_hashes().unchanged(_get(table, 'sources', leg), leg):
                return None                                                  # stale: the kernel has changed since the counter pass
            return float(table['kernels'][leg]['mfma_busy'])
        except Exception:
            pass
    return None


def algorithmic_bytes_per_solution(max_degree, nlat, nlon):
    """coefficients read once + grid written once; plan tables amortised over the batch."""
    return 8 * ((max_degree + 1) ** 2 + nlat * nlon)


# ---------------------------------------------------------------------------------------------------------------------
# rank function
# ---------------------------------------------------------------------------------------------------------------------
class RankContext:
    """Rank, world and the collectives a leg needs: barrier, max / sum over ranks, small gathers."""

    def __init__(self, args, wl, rank, world, dist, torch, reduce_device):
        self.args, self.wl, self.rank, self.world, self.dist, self.torch, self.reduce_device = args, wl, rank, world, dist, torch, reduce_device
        from grates_amd import distributed as gd
        self.all_counts = [b - a for a, b in (gd.shard_range(args.smoother_epochs, r, world) for r in range(world))]

    @property
    def collective(self):
        """collectives run whenever a process group exists -- also for a world of one rank started under a launcher"""
        return self.world > 1 or self.dist.is_initialized()

    def barrier(self):
        self.wl.synchronize()
        if self.collective:
            self.dist.barrier()
        self.wl.synchronize()

    def _reduce(self, values, op):
        if not self.collective:
            return list(values)
        t = self.torch.tensor(list(values), dtype=self.torch.float64, device=self.reduce_device)
        self.dist.all_reduce(t, op=op)
        return [float(v) for v in t.tolist()]

    def max_over_ranks(self, x):
        return self._reduce([x], self.dist.ReduceOp.MAX)[0]

    def sum_over_ranks(self, values):
        return self._reduce(values, self.dist.ReduceOp.SUM)

    def all_ranks_agree(self, flag):
        return self._reduce([1.0 if flag else 0.0], self.dist.ReduceOp.MIN)[0] > 0.5

    def gather_rows(self, tensors):
        """per-rank lists of the given (equally shaped, small) device tensors of every rank"""
        if not self.collective:
            return [list(tensors)]
        from grates_amd import distributed as gd
        return gd.gather_blocks(tensors)

    def timed(self, step, warmup, steps, plan=None, events=False):
        """W warm-up steps, K timed steps between barrier + synchronize pairs, max over ranks -> (seconds, kernel profile, GPU ms per
        step).  Kernel time comes from HIP events over exactly the timed steps, on the stream the kernels are launched on: with a
        plan its in-library events around every kernel, with events=True one event pair around every step (torch's current stream
        is the stream grates_amd.engine hands to every shg_* call)."""
        for _ in range(warmup):
            step()
        self.barrier()
        if plan is not None:
            plan.profile(True)
            plan.profile_read()
        pairs = []
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            if events:
                pair = (self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True))
                pair[0].record()
                step()
                pair[1].record()
                pairs.append(pair)
            else:
                step()
        self.barrier()
        elapsed = self.max_over_ranks(time.perf_counter() - t0)
        prof = {}
        if plan is not None:
            prof = plan.profile_read()
            plan.profile(False)
        step_ms = sum(a.elapsed_time(b) for a, b in pairs) / len(pairs) if pairs else None
        return elapsed, prof, step_ms


def run_rank(args, workload_factory=GpuWorkload, emit=print):
    """One rank of the benchmark (the whole benchmark when WORLD_SIZE is 1).  Returns the result dict on rank 0."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = 0 if args.same_device else int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE={0} does not match --gpus {1}'.format(world, args.gpus))
    wl = workload_factory(args, rank, world, local_rank)
    if (world > 1 or 'RANK' in os.environ) and not dist.is_initialized():       # under a launcher also a world of one rank gets its group
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    reduce_device = wl.device if args.backend == 'nccl' else 'cpu'
    ctx = RankContext(args, wl, rank, world, dist, torch, reduce_device)
    barrier, max_over_ranks = ctx.barrier, ctx.max_over_ranks

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

    # ---- synthesis: W warm-up steps, K timed steps between barrier + synchronize pairs, max over ranks -- straight after the
    # setup, as the contract states it (`value`); then again behind `--ramp` untimed launches (`value_after_ramp`): after an idle
    # period the first ~100 ms of fp64 MFMA work run at lower clocks.
    # Leg order (stated in config.leg_order): the device part of the covariance leg -- ~25 s of declared fp64 MFMA work, reported as its
    # own leg -- runs between the synthesis setup and the synthesis passes, so that the contract pass starts on a card whose clocks
    # have ramped (r03: 0.60 against 0.53 ms per launch when the headline was timed straight after an idle setup).
    wl.setup_synthesis()
    B = args.epochs
    leg_order = ['synthesis setup']
    # the contract pass straight behind the (idle) setup, before any other leg: reported as `roofline.value_idle_start`, so that what the
    # leg order is worth is a number of the record and not a paragraph of DESIGN.md
    idle_elapsed, idle_prof = (None, None)
    if args.idle_pass:
        barrier()
        idle_elapsed, idle_prof = timed_steps(args.warmup, args.steps)
        leg_order.append('synthesis pass behind the idle setup')
    cov_state = None
    if 'covariance' in args.leg_set:
        from grates_amd import distributed as gd
        barrier()
        cov_state = covariance_leg_timed(args, wl, rank, world, barrier, max_over_ranks, gd)
        leg_order.append('covariance (timed repeats)')
    # ... and behind it the timed steps of the filters leg (BASELINE config 3), whose last part is the dense form: ten products that load
    # the fp64 MFMA AND the HBM like the synthesis kernel does.  tools/ramp_probe.py: launches 6 .. 25 of the synthesis kernel take
    # +13 % behind an idle period, +8 % behind covariance propagation (MFMA only), +4 % behind plain copies (HBM only) and their
    # steady time behind as little as three such products (6 ms) -- the card's power management settles on the mix of the load.
    flt_state = None
    if 'filters' in args.leg_set and hasattr(wl, 'leg_filters_timed'):
        barrier()
        flt_state = wl.leg_filters_timed(ctx)
        leg_order.append('filters (timed steps: block form, then dense form)')
    leg_order += ['synthesis contract pass', 'synthesis pass behind {0} more launches'.format(args.ramp), 'synthesis check + CPU baseline']
    barrier()
    elapsed, prof = timed_steps(args.warmup, args.steps)
    for _ in range(args.ramp):
        wl.synthesis_step()
    barrier()
    ramp_elapsed, ramp_prof = timed_steps(args.warmup, args.steps) if args.ramp > 0 else (elapsed, prof)
    # the Legendre-stage limit (shg_plan_set_stage_limit, off by default) measured beside the default in the same steady state: what the
    # knob is worth on THIS box is a number of the record (it took 1.5-2 % off on some boxes and cost 2-4 % on others in round 5)
    limit_elapsed = limit_prof = None
    if args.stage_limit_pass and hasattr(wl, 'set_stage_limit') and wl.config.get('rotation_folded_kernel'):
        wl.set_stage_limit(-7)
        barrier()
        limit_elapsed, limit_prof = timed_steps(args.warmup, args.steps)
        wl.set_stage_limit(0)
    chain = wl.api_chain(ctx) if args.api_chain and hasattr(wl, 'api_chain') else None

    line = None
    if rank == 0:
        per_solution = algorithmic_bytes_per_solution(MAX_DEGREE, wl.nlat, wl.nlon)

        def kernel_rate(p):
            lon_ms, lon_launches = p.get('lon_stage', (0.0, 0))
            launches_per_step = lon_launches / max(args.steps, 1)
            epochs_per_launch = B / max(launches_per_step, 1e-9)
            avg_ms = lon_ms / max(lon_launches, 1)
            rate = per_solution * epochs_per_launch / (avg_ms * 1e-3) / 1e9 if lon_launches else None
            return rate, avg_ms, epochs_per_launch
        achieved, lon_avg_ms, epochs_per_launch = kernel_rate(prof)
        ramp_achieved, ramp_avg_ms, _ = kernel_rate(ramp_prof)
        idle_achieved, idle_avg_ms, _ = kernel_rate(idle_prof) if idle_prof is not None else (None, None, None)
        limit_achieved, limit_avg_ms, _ = kernel_rate(limit_prof) if limit_prof is not None else (None, None, None)
        traffic, traffic_source = pmc_traffic('synthesis', [wl.kernel_name])
        kernels = {k: {'ms_total': round(v[0], 4), 'launches': int(v[1]), 'avg_us': round(1e3 * v[0] / max(v[1], 1), 3)} for k, v in prof.items()}
        config = {'workload': 'batch of {0} monthly solutions d/o {1} -> {2} deg GeographicGrid ({3}x{4}), kernel {5}, per GPU'.format(
            B, MAX_DEGREE, GRID_STEP, wl.nlat, wl.nlon, KERNEL),
            'max_degree': MAX_DEGREE, 'epochs_per_gpu': B, 'grid': [wl.nlat, wl.nlon],
            'parallelism': 'epochs sharded over {0} GPU(s), no collective'.format(world),
            'legs': sorted(args.leg_set), 'ramp_launches_before_value_after_ramp': args.ramp,
            'leg_order': leg_order + (['covariance extensions + CPU baseline'] if cov_state is not None else []) +
                         (['filters checks + CPU baselines'] if flt_state is not None else []) +
                         [n for n in ('analysis', 'filters', 'smoother') if n in args.leg_set and not (n == 'filters' and flt_state is not None)]}
        config.update(wl.config)
        if dist.is_initialized():
            config['process_group'] = {'backend': dist.get_backend(), 'world': dist.get_world_size()}
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
            'value_after_ramp': world * B * args.steps / ramp_elapsed,
            'ms_per_step_after_ramp': 1e3 * ramp_elapsed / args.steps,
            'roofline': {
                'kernel': wl.kernel_name, 'bound': 'hbm',
                'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': (achieved / HBM_PEAK_GBS) if achieved else None,
                'traffic': traffic, 'traffic_source': traffic_source, 'mfma_busy': pmc_mfma_busy('synthesis'),
                'algorithmic_bytes_per_launch': per_solution * epochs_per_launch,
                'avg_launch_ms': lon_avg_ms,
                'whole_path_GBs': per_solution * B * args.steps / elapsed / 1e9,
                'after_ramp': {'achieved': ramp_achieved, 'frac': (ramp_achieved / HBM_PEAK_GBS) if ramp_achieved else None, 'avg_launch_ms': ramp_avg_ms,
                               'whole_path_GBs': per_solution * B * args.steps / ramp_elapsed / 1e9},
                'value_idle_start': (world * B * args.steps / idle_elapsed) if idle_elapsed else None,
                'frac_idle_start': (idle_achieved / HBM_PEAK_GBS) if idle_achieved else None,
                'avg_launch_ms_idle_start': idle_avg_ms,
                # the same steady state as `after_ramp`, with at most 7/16 of the CUs in their Legendre stage at once (the knob is OFF in `value`)
                'value_stage_limit_on': (world * B * args.steps / limit_elapsed) if limit_elapsed else None,
                'avg_launch_ms_stage_limit_on': limit_avg_ms,
                'api_chain_solutions_per_s': chain['solutions_per_s'] if chain else None,
                'api_chain_bit_identical': chain['bit_identical_to_per_epoch_calls'] if chain else None,
            },
            'api_chain': chain,
            'kernels': kernels,
        }
        sample = min(args.cpu_sample, B) if world == 1 else min(args.cpu_sample, 1)
        if sample > 0:
            line['check'], line['cpu_baseline'] = wl.synthesis_check(sample, timed=world == 1)
        else:
            line['check'], line['cpu_baseline'] = None, None

    # ---- the other legs, each between its own barriers
    wl.release_synthesis()
    legs = {}
    if cov_state is not None:
        legs['covariance'] = covariance_leg_report(args, wl, rank, world, cov_state)
        cov_state = None
        if hasattr(wl, 'release_covariance'):
            wl.release_covariance()
    if flt_state is not None:
        legs['filters'] = wl.leg_filters_report(ctx, flt_state)
        flt_state = None
    for name in ('analysis', 'filters', 'smoother'):
        runner = getattr(wl, 'leg_' + name, None)
        if name in args.leg_set and runner is not None and name not in legs:
            barrier()
            legs[name] = runner(ctx)
    if rank == 0:
        line.update(legs)
        checks = [line.get('check')] + [legs[k].get('check') for k in legs if isinstance(legs[k], dict)]
        line['all_checks_ok'] = all(c.get('ok', True) for c in checks if isinstance(c, dict))
        line = compact_line(summarise(line))
        emit(json.dumps(line, separators=(',', ':')))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return line


def covariance_leg(args, wl, rank, world, barrier, max_over_ranks, gd):
    """d/o-180 covariance propagation to the 0.5 degree grid (BASELINE config 4): sigma = sqrt(diag(A Sigma A^T)) with A
    generated on the fly, A Sigma on fp64 MFMA.  Flops = 2 M P^2 + 2 M P.  The parallels [0, total) are split into one
    contiguous band per rank, every rank holds all of Sigma, the bands are gathered with one all_gather."""
    return covariance_leg_report(args, wl, rank, world, covariance_leg_timed(args, wl, rank, world, barrier, max_over_ranks, gd))


def covariance_leg_timed(args, wl, rank, world, barrier, max_over_ranks, gd):
    """The device part of the covariance leg: setup, the timed repeats with their all_gather.  Returns what the report needs."""
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
    return {'times': times, 'prof': prof, 'sigma': sigma, 'bands': bands, 'sizes': sizes, 'total': total}


def covariance_leg_report(args, wl, rank, world, state):
    """The line of the covariance leg from the timed part's record, its extensions (one GPU) and its CPU baseline."""
    if rank != 0:
        return None
    times, prof, sigma, bands, sizes, total = (state[k] for k in ('times', 'prof', 'sigma', 'bands', 'sizes', 'total'))
    nlat, nlon, P = wl.cov_nlat, wl.cov_nlon, wl.P
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
            'P': P, 'parallels': total, 'nlat': nlat, 'sigma_recipe': wl.cov_recipe, 'flops': flops, 'full_grid_flops': 2.0 * nlat * nlon * P * (P + 1.0), 'repeats': len(times)},
        'seconds_median': median, 'seconds_min': best, 'seconds_all': times, 'GFLOPs_best': flops / best / 1e9,
        'roofline': {'kernel': 'gemm_f64_kernel<MODE_COVPROP>', 'bound': 'mfma', 'achieved': achieved, 'peak': MFMA_F64_PEAK_TFLOPS,
                     'unit': 'TFLOP/s', 'frac': (achieved / MFMA_F64_PEAK_TFLOPS) if achieved else None, 'traffic': None, 'mfma_busy': pmc_mfma_busy('covariance'),
                     'traffic_note': 'MFMA-bound; HBM-side bytes of a band of 8 parallels (Sigma read once per 128 rows): profiles/r04_pmc_traffic.json (legs.covariance)',
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
    if world == 1 and args.cov_cpu_parallels > 0 and args.cpu_sample > 0 and hasattr(wl, 'covariance_cpu'):
        out['cpu_baseline'] = wl.covariance_cpu(0, min(args.cov_cpu_parallels, total), sigma)
        err = out['cpu_baseline']['max_rel_diff_vs_gpu']
        out['check'] = {'max_rel_err_vs_oracle': err, 'tolerance': 1e-11, 'ok': bool(err < 1e-11),
                        'what': 'sigma of the first {0} parallel(s) of the timed, gathered output against the NumPy oracle'.format(min(args.cov_cpu_parallels, total))}
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    refuse_experiment_environment()
    if args.gpus > 1 and 'RANK' not in os.environ and int(os.environ.get('WORLD_SIZE', '1')) == 1:
        raise SystemExit(launch_ranks(args, argv))
    run_rank_reported(args)


def run_rank_reported(args, run=None):
    """run_rank; a rank that fails says which one it is, prints its traceback and ends its process with code 1 without waiting for
    collectives that will never complete (the launcher then ends the other ranks and the job returns non-zero)."""
    import traceback
    try:
        return (run or run_rank)(args)
    except Exception as err:                                            # noqa: BLE001 -- (SystemExit passes: it carries its own message)
        rank = os.environ.get('RANK', '0')
        sys.stderr.write('bench.py: rank {0} of {1} FAILED: {2!r}\n{3}'.format(rank, os.environ.get('WORLD_SIZE', '1'), err, traceback.format_exc()))
        sys.stderr.flush()
        sys.stdout.flush()
        if 'RANK' in os.environ:
            os._exit(1)
        raise SystemExit(1)


if __name__ == '__main__':
    main()
