"""
bench.py end to end on the GPU at reduced sizes: every leg runs through the product path, checks its own timed output buffer
against the oracle (or a size-independent property) and the line says so.  The full-size run is the driver's; this keeps the
self-certification of the line from rotting.
"""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_small_sizes():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '1', '--ramp', '2', '--epochs', '12', '--cpu-sample', '2',
           '--cov-parallels', '2', '--cov-repeats', '1', '--cov-extensions', '0', '--smoother-epochs', '24', '--smoother-cpu-epochs', '3', '--smoother-repeats', '1']
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert run.returncode == 0, run.stderr[-3000:]
    line = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith('{')][-1])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
                'roofline', 'cpu_baseline'):
        assert key in line, key
    assert line['n_gpus'] == 1 and line['steps'] == 3 and line['dtype'] == 'f64' and line['vs_baseline'] is None
    assert line['check']['ok'] and line['check']['max_rel_err_vs_oracle'] < 1e-12 and line['check']['epochs_checked'] == 2
    assert line['roofline']['bound'] == 'hbm' and 0 < line['roofline']['frac'] < 1 and line['cpu_baseline']['kind'] == 'port'
    cov = line['covariance']
    assert cov['check']['ok'] and cov['roofline']['bound'] == 'mfma' and cov['cpu_baseline']['cores'] >= 1
    ana = line['analysis']
    assert ana['check']['ok'] and ana['check']['max_rel_err_vs_oracle'] < 1e-11 and ana['roofline']['avg_launch_ms'] > 0
    flt = line['filters']
    assert flt['check']['ok'] and flt['check']['block_max_rel_err_vs_oracle'] < 1e-12 and flt['check']['dense_max_rel_err_vs_oracle'] < 1e-12
    assert flt['block']['roofline']['bound'] == 'hbm' and flt['dense']['roofline']['bound'] == 'mfma'
    sm = line['smoother']
    assert sm['config']['epochs'] == 24 and sm['check']['ok'] and sm['check']['residual'] < 1e-13 and sm['check']['short_chain_max_rel_err_vs_oracle'] < 1e-9
    assert {'factor_s', 'solve_s', 'covariance_s'} <= set(sm['phases_s'])
    assert line['all_checks_ok']
