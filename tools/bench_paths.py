#!/usr/bin/env python3
"""
Timings of every row of the hot-path table at the BASELINE configurations, one JSON object per line:

    python tools/bench_paths.py > profiles/rNN_paths.jsonl

All inputs are synthetic and resident in HBM; times are device times around the C-ABI calls (torch events on the
current stream, which is the stream the kernels are launched on).  The headline metric itself is bench.py.
"""

import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))

import torch  # noqa: E402

import grates_amd as ga  # noqa: E402
import inputs  # noqa: E402

GM, R = 3.9860044150e+14, 6.3781363000e+06


def device_ms(fn, reps=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def emit(name, ms, **kw):
    print(json.dumps(dict(path=name, ms=round(ms, 4), **kw)), flush=True)


def plan_for(grid, N, kernel):
    colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel(kernel), N, grid.parallels, GM, R, grid.semimajor_axis, grid.flattening)
    return ga.engine.Plan(N, colat, kn, grid.meridians)


def main():
    rng = np.random.default_rng(0)

    # config 1: single d/o-60 solution, Gaussian 300 km, 1 degree grid (end-to-end through the drop-in classes, host arrays)
    gf = ga.gravityfield.PotentialCoefficients()
    gf.anm = inputs.coefficients(1000, 60)
    g1 = ga.grid.GeographicGrid(1.0, 1.0)
    flt = ga.filter.Gaussian(300)
    flt.filter(gf).to_grid(g1, 'ewh')
    t0 = time.perf_counter()
    for _ in range(10):
        flt.filter(gf).to_grid(g1, 'ewh')
    emit('config1: Gaussian(300).filter + to_grid d/o 60 -> 1 deg, host in / host out (wall)', (time.perf_counter() - t0) * 100, unit='ms per call')

    # synthesis variants at config 2
    g025 = ga.grid.GeographicGrid(0.25, 0.25)
    batch = torch.from_numpy(rng.standard_normal((240, 97, 97)) * 1e-10).cuda()
    plan = plan_for(g025, 96, 'ewh')
    out = torch.empty((240, 720, 1440), dtype=torch.float64, device='cuda')
    for path in ('rot', 'fused', 'staged'):
        plan.set_path(path)
        ms = device_ms(lambda: plan.synthesis(batch, out=out), reps=40, warmup=10)
        emit('synthesis d/o 96 -> 0.25 deg, 240 epochs, path=' + path, ms, solutions_per_s=round(240 / ms * 1e3), GBs_algorithmic=round(240 * 8369672 / ms / 1e6, 1))
    plan.set_path('auto')
    one = batch[0:1].contiguous()
    ms = device_ms(lambda: plan.synthesis(one), reps=20)
    emit('synthesis d/o 96 -> 0.25 deg, single epoch (latency)', ms)
    ms = device_ms(lambda: ga.engine.epoch_rms(out.reshape(240, -1), None, 240), reps=20)
    emit('RMS over 240 epochs of 0.25 deg grids (shg_epoch_rms, reduction of gridded_rms)', ms, GBs=round(8 * 241 * 720 * 1440 / ms / 1e6, 1))
    del out

    # d/o 180 synthesis (staged path: panel does not fit LDS)
    g05 = ga.grid.GeographicGrid(0.5, 0.5)
    b180 = torch.from_numpy(rng.standard_normal((64, 181, 181)) * 1e-10).cuda()
    p180 = plan_for(g05, 180, 'ewh')
    ms = device_ms(lambda: p180.synthesis(b180))
    emit('synthesis d/o 180 -> 0.5 deg, 64 epochs (32-row fused kernel)', ms, solutions_per_s=round(64 / ms * 1e3))

    # point-list synthesis
    lon, lat = inputs.scattered_points(1, 100000)
    irr = ga.grid.IrregularGrid(lon, lat)
    b40 = rng.standard_normal((16, 41, 41)) * 1e-10
    ms = device_ms(lambda: ga.gravityfield.synthesize(b40, irr, 'ewh'), reps=3, warmup=1)
    emit('point-list synthesis d/o 40, 100000 points x 16 epochs', ms)
    b96 = torch.from_numpy(rng.standard_normal((240, 97, 97)) * 1e-10).cuda()
    ms = device_ms(lambda: ga.gravityfield.synthesize(b96, irr, 'ewh'), reps=3, warmup=1)
    emit('point-list synthesis d/o 96, 100000 points x 240 epochs (GEMM with generated harmonics)', ms, TFLOPs=round(2.0 * 100000 * 97 ** 2 * 240 / ms / 1e9, 1))

    # analysis: d/o 96 from 0.5 degree (reference: 142 s per epoch on 8 cores)
    pa = plan_for(g05, 96, 'potential')
    vals = torch.from_numpy(rng.standard_normal((32, 360, 720))).cuda()
    area = torch.from_numpy(g05.area.reshape(360, 720)).cuda()
    ms = device_ms(lambda: pa.analysis(vals, area, 0), reps=3, warmup=1)
    emit('analysis d/o 96 <- 0.5 deg, 32 epochs', ms, epochs_per_s=round(32 / ms * 1e3, 1))

    # config 3: DDK5-type filter at d/o 120 on 240 epochs: order-wise blocks and dense full-matrix multiply
    nmax, T = 120, 240
    blocks = inputs.orderwise_random_blocks(42, nmax)
    ow = ga.filter.OrderWiseFilter(blocks)
    ts = torch.from_numpy(rng.standard_normal((T, nmax + 1, nmax + 1)) * 1e-10).cuda()
    ms = device_ms(lambda: ow.filter_batch(ts))
    nblock = sum(b.size for b in blocks)
    emit('config3: order-wise DDK filter d/o 120, 240 epochs', ms, epochs_per_s=round(T / ms * 1e3), GBs=round(8 * (nblock + 2 * T * (nmax + 1) ** 2) / ms / 1e6, 1))
    P = (nmax + 1) ** 2 - 4
    W = torch.from_numpy(rng.standard_normal((P, P)) / P).cuda()
    X = torch.from_numpy(rng.standard_normal((P, T))).cuda()
    ms = device_ms(lambda: ga.engine.dense_filter(W, X))
    emit('config3: dense W[14637^2] @ X[14637 x 240] (shg_dense_filter)', ms, TFLOPs=round(2.0 * P * P * T / ms / 1e9, 1))
    normals = inputs.orderwise_normal_blocks(44, nmax)
    w = 1e11 * np.arange(nmax + 1, dtype=float) ** 4
    w[0] = 1
    t0 = time.perf_counter()
    ga.engine.ddk_blocks(normals, w)
    emit('config3: DDK block construction d/o 120 (241 Cholesky solves, incl. host<->device copies, wall)', (time.perf_counter() - t0) * 1e3)

    # fp64 GEMM
    for M, N, K in ((8192, 8192, 8192), (4096, 4096, 4096)):
        A = torch.rand((M, K), dtype=torch.float64, device='cuda') - 0.5
        Bm = torch.rand((K, N), dtype=torch.float64, device='cuda') - 0.5
        ms = device_ms(lambda: ga.engine.dgemm(A, Bm), reps=3)
        emit('shg_dgemm {0}^3'.format(M), ms, TFLOPs=round(2.0 * M * N * K / ms / 1e9, 1))
        del A, Bm

    # covariance propagation d/o 60 -> 1 deg full grid (reference: 6.5 s on 8 cores)
    g1 = ga.grid.GeographicGrid(1.0, 1.0)
    p60 = plan_for(g1, 60, 'ewh')
    cov = torch.from_numpy(inputs.spd_covariance(3, 61 * 61)).cuda()
    ms = device_ms(lambda: p60.covariance_propagation(cov, 0), reps=3, warmup=1)
    emit('covariance propagation d/o 60 -> 1 deg (full grid)', ms, TFLOPs=round(2.0 * 64800 * 3721.0 ** 2 / ms / 1e9, 1))


if __name__ == '__main__':
    main()
