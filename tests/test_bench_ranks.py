"""
bench.py's rank function under gloo with world size 2 on the CPU: epoch-sharded synthesis timing, the covariance leg over
real latitude bands (grates_amd.distributed.latitude_bands) with the all_gather of the per-band sigma, max-over-ranks
timing, one JSON line on rank 0 -- and the launcher that `python bench.py --gpus N` uses when no launcher started it.
The GPU work is replaced by a stand-in workload of the same interface (bench.GpuWorkload): the kernels themselves are
covered by the -m gpu tests, here the multi-rank plumbing of the benchmark is.
"""

import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402


class StubWorkload:
    """CPU stand-in with the interface of bench.GpuWorkload; sigma(i, j) = sqrt(1 + i * nlon + j) for grid point (i, j)."""
    device = 'cpu'

    def __init__(self, args, rank, world, local_rank):
        self.args, self.rank, self.world = args, rank, world
        self.steps_done = 0

    def synchronize(self):
        pass

    def setup_synthesis(self):
        self.nlat, self.nlon = 12, 24
        self.kernel_name = 'stub_kernel'
        self.config = {'stub': True}
        self.x = torch.ones(1000, dtype=torch.float64)

    def synthesis_step(self):
        self.x = self.x * 1.0000001
        self.steps_done += 1

    def profile(self, enable):
        self.profiling = enable
        self.mark = self.steps_done

    def profile_read(self):
        n = self.steps_done - self.mark
        self.mark = self.steps_done
        return {'lon_stage': (0.5 * n, n)} if n else {}

    def release_synthesis(self):
        del self.x

    def synthesis_check(self, sample, timed):
        check = {'max_rel_err_vs_oracle': 0.0, 'epochs_checked': sample, 'ok': True}
        return check, ({'value': 1.0, 'unit': 'solutions/s', 'cores': 1, 'kind': 'port', 'sample': 'stub'} if timed else None)

    def setup_covariance(self):
        self.cov_nlat, self.cov_nlon, self.P = 7, 5, 9
        self.cov_recipe = 'stub'
        self.cov_calls = 0

    def covariance_band(self, lat0, lat1, **kw):
        self.cov_calls += 1
        return torch.sqrt(1.0 + torch.arange(lat0 * self.cov_nlon, lat1 * self.cov_nlon, dtype=torch.float64))

    def cov_profile(self, enable):
        self.cov_mark = self.cov_calls

    def cov_profile_read(self):
        n = self.cov_calls - self.cov_mark
        self.cov_mark = self.cov_calls
        return {'covprop': (2.0 * n, n)} if n else {}

    def leg_smoother(self, ctx):
        """stand-in for the epoch-sharded smoother leg: every rank 'solves' x_t = t for its epochs; exercises the rank context
        (shard sizes, agreement, gathers of boundary rows, sums and maxima over ranks)"""
        from grates_amd import distributed as gd
        T = self.args.smoother_epochs
        t0, t1 = gd.shard_range(T, ctx.rank, ctx.world)
        assert ctx.all_counts[ctx.rank] == t1 - t0 and sum(ctx.all_counts) == T
        assert ctx.all_ranks_agree(True) and not ctx.all_ranks_agree(ctx.rank == 0 and ctx.world > 1)
        x = torch.arange(t0, t1, dtype=torch.float64)
        edges = ctx.gather_rows([x[:1].clone(), x[-1:].clone()])
        if ctx.rank > 0:
            assert float(edges[ctx.rank - 1][1]) == t0 - 1                     # last epoch of the rank before
        if ctx.rank + 1 < ctx.world:
            assert float(edges[ctx.rank + 1][0]) == t1                         # first epoch of the rank behind
        elapsed, _, step_ms = ctx.timed(lambda: None, 1, 2)
        assert step_ms is None
        total = ctx.sum_over_ranks([float(x.sum())])[0]
        longest = ctx.max_over_ranks(float(t1 - t0))
        if ctx.rank != 0:
            return None
        return {'value': T / max(elapsed, 1e-9), 'unit': 'epochs/s', 'n_gpus': ctx.world, 'scaling': 'strong', 'epochs_per_rank': ctx.all_counts,
                'check': {'solution_checksum': total, 'longest_shard': longest, 'ok': total == T * (T - 1) / 2}}


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _rank(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    args = bench.parse_args(['--gpus', str(world), '--steps', '4', '--warmup', '2', '--ramp', '3', '--epochs', '6', '--backend', 'gloo',
                             '--cov-repeats', '2', '--smoother-epochs', '11'])
    lines = []
    result = bench.run_rank(args, workload_factory=StubWorkload, emit=lines.append)
    assert (result is not None) == (rank == 0) and len(lines) == (1 if rank == 0 else 0)
    if rank == 0:
        with open(os.path.join(out_dir, 'line.json'), 'w') as f:
            f.write(lines[0])


def _check_record(line, world):
    """What the driver's record keeps: the contract's scalars and the SCALARS of config / roofline / cpu_baseline.  The whole metric must
    be readable from those: both halves (solutions/s, covariance GFLOP/s with its fraction of the MFMA peak), the leg order as one
    string, the grid, the world -- and the line must fit the 8 KB the driver keeps of the output."""
    text = json.dumps(line, separators=(',', ':'))
    assert len(text) < bench.LINE_LIMIT, len(text)
    cfg, roof = line['config'], line['roofline']
    for obj in (cfg, roof) + ((line['cpu_baseline'],) if line['cpu_baseline'] else ()):
        for k, v in obj.items():
            if k == 'kernels':
                continue
            assert v is None or isinstance(v, (int, float, str, bool)), (k, v)
            assert not isinstance(v, str) or len(v) <= 140, (k, len(v))
    assert cfg['leg_order'].startswith('setup>idle_pass>covariance>') and '>CONTRACT_PASS>' in cfg['leg_order']
    assert cfg['grid'] == '12x24' and 'parallels' in cfg['covariance_workload'] and cfg['legs'].startswith('analysis,covariance')
    if world > 1 or 'process_group' in cfg:
        assert cfg['process_group'] == 'gloo world {0}'.format(world)
    cov = line['covariance']
    assert roof['covariance_GFLOPs'] == pytest.approx(cov['value']) and roof['covariance_seconds'] == pytest.approx(cov['seconds_median'])
    assert roof['covariance_frac'] == pytest.approx(cov['roofline']['frac']) and roof['covariance_kernel_TFLOPs'] > 0
    assert roof['smoother_epochs_per_s'] == pytest.approx(line['smoother']['value']) and roof['all_checks_ok'] is True
    assert 'traffic_source' not in roof


def _expected_sigma():
    return torch.sqrt(1.0 + torch.arange(7 * 5, dtype=torch.float64)).numpy()       # the stand-in's own arithmetic


@pytest.mark.parametrize('world', [2, 3, 7, 8])
def test_rank_function_gloo(world, tmp_path):
    """worlds 7 and 8: the pre-flight of the driver's 8-GPU run -- ragged latitude bands (7 parallels over 7 / 8 ranks: bands of one
    parallel and, for 8 ranks, an EMPTY band), 11 smoother epochs over 8 ranks, the line says n_gpus 8 and the world of its group"""
    mp.spawn(_rank, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    line = json.loads(open(tmp_path / 'line.json').read())
    assert line['metric'] == bench.METRIC and line['unit'] == 'solutions/s' and line['n_gpus'] == world
    assert line['steps'] == 4 and line['warmup'] == 2 and line['scaling'] == 'weak' and line['dtype'] == 'f64'
    assert line['value'] == pytest.approx(world * 6 * 4 / (line['ms_per_step'] * 4e-3))        # whole-job aggregate over all ranks
    assert line['value_after_ramp'] > 0 and line['cpu_baseline'] is None                      # the CPU baseline runs at N = 1 only
    assert line['check']['epochs_checked'] == 1                                               # ... the oracle check of the output does not
    assert line['roofline']['bound'] == 'hbm' and line['roofline']['avg_launch_ms'] == pytest.approx(0.5)
    assert line['roofline']['avg_launch_ms_after_ramp'] == pytest.approx(0.5) and line['roofline']['value_after_ramp'] == line['value_after_ramp']
    assert line['roofline']['value_idle_start'] > 0 and line['roofline']['avg_launch_ms_idle_start'] == pytest.approx(0.5)
    _check_record(line, world)
    sm = line['smoother']
    assert sm['n_gpus'] == world and sum(sm['epochs_per_rank']) == 11 and sm['check']['ok'] and line['all_checks_ok']
    assert sm['check']['longest_shard'] == max(sm['epochs_per_rank'])
    assert 'analysis' not in line and 'filters' not in line                                    # legs the workload does not offer are left out
    cov = line['covariance']
    assert cov['n_gpus'] == world and cov['config']['repeats'] == 2 and len(cov['seconds_all']) == 2
    # the gathered grid is the full grid whatever the number of bands (7 parallels: unequal bands)
    assert cov['sigma_checksum'] == pytest.approx(_expected_sigma().sum(), rel=0, abs=1e-12)
    import zlib
    assert cov['sigma_crc32'] == zlib.crc32(_expected_sigma().tobytes()) & 0xffffffff


def test_rank_function_single_process(monkeypatch):
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    args = bench.parse_args(['--steps', '3', '--warmup', '1', '--ramp', '0', '--epochs', '5', '--cov-repeats', '3'])
    lines = []
    line = bench.run_rank(args, workload_factory=StubWorkload, emit=lines.append)
    assert json.loads(lines[0]) == json.loads(json.dumps(line))
    assert line['n_gpus'] == 1 and line['cpu_baseline']['kind'] == 'port' and line['check']['epochs_checked'] == 5
    _check_record(line, 1)
    assert line['smoother']['epochs_per_rank'] == [bench.SMOOTHER_EPOCHS] and line['all_checks_ok']
    assert line['covariance']['sigma_checksum'] == pytest.approx(_expected_sigma().sum(), rel=0, abs=1e-12)
    assert line['covariance']['seconds_min'] <= line['covariance']['seconds_median']


def test_bare_multi_gpu_invocation_launches_ranks(monkeypatch):
    """`python bench.py --gpus 2` with no launcher: a torch.distributed.run child is started and nothing else happens here."""
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    calls = []
    monkeypatch.setattr(bench, 'run_child', lambda cmd, env, timeout, what: calls.append((cmd, env)) or 0)
    with pytest.raises(SystemExit) as e:
        bench.main(['--gpus', '2', '--steps', '5', '--warmup', '1'])
    assert e.value.code == 0 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '2'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and os.path.basename(cmd[cmd.index('--master-port') + 2]) == 'bench.py'
    assert cmd[-6:] == ['--gpus', '2', '--steps', '5', '--warmup', '1'] and env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    # experiment switches / library overrides in the environment: refused before anything runs
    for name in bench.FORBIDDEN_ENVIRONMENT:
        monkeypatch.setenv(name, '1')
        with pytest.raises(SystemExit) as e:
            bench.main(['--gpus', '1'])
        assert name in str(e.value.code) and len(calls) == 1
        monkeypatch.delenv(name)
    # under a launcher (RANK set) the same command line is a rank, not a launcher
    monkeypatch.setenv('RANK', '0')
    monkeypatch.setenv('WORLD_SIZE', '1')
    with pytest.raises(SystemExit) as e:
        bench.main(['--gpus', '2'])
    assert 'WORLD_SIZE' in str(e.value.code) and len(calls) == 1


def _rank_world_of_one(rank, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    args = bench.parse_args(['--gpus', '1', '--steps', '2', '--warmup', '1', '--ramp', '0', '--epochs', '4', '--backend', 'gloo',
                             '--cov-repeats', '1', '--smoother-epochs', '9'])
    import torch.distributed as dist
    seen = []
    real = dist.all_gather

    def spy(*a, **k):
        seen.append('all_gather')
        return real(*a, **k)
    dist.all_gather = spy
    line = bench.run_rank(args, workload_factory=StubWorkload, emit=lambda text: None)
    assert line['n_gpus'] == 1 and line['all_checks_ok'] and not dist.is_initialized()
    assert seen, 'a world of one rank under a launcher runs its collectives (covariance leg: all_gather of the bands)'
    with open(os.path.join(out_dir, 'ok'), 'w') as f:
        f.write('ok')


def test_world_of_one_under_a_launcher_runs_the_collectives(tmp_path):
    """RANK / WORLD_SIZE = 1 in the environment (torch.distributed.run --nproc-per-node 1): a process group is created and every
    collective of the rank function runs -- the path tests/test_gpu_rccl.py takes over RCCL on the one-GPU box."""
    mp.spawn(_rank_world_of_one, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / 'ok').exists()


def test_launcher_timeout_and_failing_rank(tmp_path):
    import subprocess
    import time
    t0 = time.time()
    rc = bench.run_child([sys.executable, '-c', 'import time; time.sleep(120)'], dict(os.environ), 1.0, 'sleepers')
    assert rc == 124 and time.time() - t0 < 30
    assert bench.run_child([sys.executable, '-c', 'raise SystemExit(3)'], dict(os.environ), 30.0, 'x') == 3
    code = ('import sys; sys.path.insert(0, {0!r}); import bench\n'
            'def boom(args):\n    raise RuntimeError("kernel launch failed")\n'
            'bench.run_rank_reported(bench.parse_args(["--gpus", "2"]), run=boom)\n').format(ROOT)
    done = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, RANK='1', WORLD_SIZE='2'), capture_output=True, text=True, timeout=120)
    assert done.returncode == 1
    assert 'rank 1 of 2 FAILED' in done.stderr and 'kernel launch failed' in done.stderr and 'Traceback' in done.stderr


def test_block_table_compressed_rows():
    """engine.BlockTable: the stored upper blocks of a block matrix in the compressed row form of the shg_block_* calls
    (include/shg.h): rows ascending, columns ascending from the diagonal, blocks below the diagonal left out."""
    from grates_amd import engine
    bounds = [0, 3, 5, 9, 10]
    shape = lambda i, j: (bounds[i + 1] - bounds[i], bounds[j + 1] - bounds[j])          # noqa: E731
    keys = [(2, 3), (0, 0), (1, 1), (0, 2), (3, 3), (2, 2), (1, 0), (0, 1)]                # (1, 0) lies below the diagonal
    blocks = {k: torch.zeros(shape(*k), dtype=torch.float64) for k in keys}
    table = engine.BlockTable(bounds, blocks)
    assert table.nb == 4 and table.bounds.dtype == np.int32 and table.bounds.tolist() == bounds
    assert table.rowptr.tolist() == [0, 3, 4, 6, 7]
    assert table.colidx.tolist() == [0, 1, 2, 1, 2, 3, 3]
    expect = [(0, 0), (0, 1), (0, 2), (1, 1), (2, 2), (2, 3), (3, 3)]
    assert table.address.tolist() == [blocks[k].data_ptr() for k in expect]
    nb, pb, pr, pc, pa = table.args()
    assert nb == 4 and all(isinstance(p, type(pb)) for p in (pr, pc, pa))
