"""How long does the synthesis kernel take launch by launch after (a) an idle period, (b) a stretch of covariance propagation
(dense fp64 MFMA work), (c) a stretch of itself?  One event pair per launch on the launching stream.
Usage: python tools/ramp_probe.py [--launches 120]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--launches', type=int, default=120)
    args = ap.parse_args()
    import numpy as np
    import torch
    import grates_amd as ga
    N, B = 96, 240
    grid = ga.grid.GeographicGrid(0.25, 0.25)
    GM, R = 3.9860044150e+14, 6.3781363000e+06
    colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel('ewh'), N, grid.parallels, GM, R, grid.semimajor_axis, grid.flattening)
    plan = ga.engine.Plan(N, colat, kn, grid.meridians)
    batch = torch.from_numpy(np.random.default_rng(5).standard_normal((B, N + 1, N + 1)) * 1e-10).cuda()
    out = torch.empty((B, grid.parallels.size, grid.meridians.size), dtype=torch.float64, device='cuda')
    cg = ga.grid.GeographicGrid(0.5, 0.5)
    Nc = 120
    ccolat, _, ckn = ga.gravityfield.surface_factors(ga.kernel.get_kernel('ewh'), Nc, cg.parallels, GM, R, cg.semimajor_axis, cg.flattening)
    cplan = ga.engine.Plan(Nc, ccolat, ckn, cg.meridians)
    P = (Nc + 1) ** 2
    G = torch.randn((P, P + 16), dtype=torch.float64, device='cuda')
    cov = ga.engine.gemm(G, G, transb=True, alpha=1e-22 / P)
    del G
    cplan.covariance_propagation(cov, 0, 0, 1)
    plan.synthesis(batch, out=out)
    torch.cuda.synchronize()

    def series(label):
        pairs = []
        for _ in range(args.launches):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            plan.synthesis(batch, out=out)
            b.record()
            pairs.append((a, b))
        torch.cuda.synchronize()
        t = [a.elapsed_time(b) for a, b in pairs]
        print(label, ' '.join('{0:.3f}'.format(x) for x in t[0:12]), '| 25..30:', ' '.join('{0:.3f}'.format(x) for x in t[25:30]),
              '| last 5:', ' '.join('{0:.3f}'.format(x) for x in t[-5:]), '| mean 5..25: {0:.4f}'.format(sum(t[5:25]) / 20), flush=True)

    time.sleep(3.0)
    series('after 3 s idle      ')
    series('after itself        ')
    t0 = time.perf_counter()
    cplan.covariance_propagation(cov, 0, 0, 240)
    torch.cuda.synchronize()
    print('covariance stretch {0:.2f} s'.format(time.perf_counter() - t0))
    series('after covariance    ')
    series('after itself        ')
    other = torch.empty_like(out)
    for _ in range(100):                                     # ~80 ms of plain HBM traffic (2 GB read + 2 GB written per copy)
        other.copy_(out)
    torch.cuda.synchronize()
    series('after HBM copies    ')
    del other
    big = torch.randn((8192, 8192), dtype=torch.float64, device='cuda')
    for _ in range(6):                                       # ~100 ms of plain fp64 MFMA work (square products, L2-resident operands)
        ga.engine.gemm(big, big)
    torch.cuda.synchronize()
    series('after fp64 products ')
    del big
    big = torch.randn((8192, 8192), dtype=torch.float64, device='cuda')
    ga.engine.gemm(big, big)                                 # ~17 ms
    torch.cuda.synchronize()
    series('after ONE product   ')
    del big
    Pn = 14637
    W = torch.randn((Pn, Pn), dtype=torch.float64, device='cuda')
    X = torch.randn((Pn, 240), dtype=torch.float64, device='cuda')
    for _ in range(10):                                      # the dense filter of config 3: 10 x 2 ms
        ga.engine.dense_filter(W, X)
    torch.cuda.synchronize()
    series('after dense filter  ')
    for _ in range(3):
        ga.engine.dense_filter(W, X)
    torch.cuda.synchronize()
    series('after 3 dense filt. ')
    del W, X
    cplan.covariance_propagation(cov, 0, 0, 60)
    torch.cuda.synchronize()
    series('after cov 0.25 s    ')
    time.sleep(0.2)
    series('after 0.2 s idle    ')
    time.sleep(1.0)
    series('after 1 s idle      ')


if __name__ == '__main__':
    main()
