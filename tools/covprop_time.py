"""Covariance propagation d/o 180 -> 0.5 degree grid on a band of parallels, event-timed; for A / B runs of two builds on one box.
    python3 tools/covprop_time.py [library.so | -] [parallels]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import grates_amd as ga
if len(sys.argv) > 1 and sys.argv[1] != '-':
    ga._lib.use_library(sys.argv[1])
import bench
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 48
N = bench.COV_DEGREE
grid = ga.grid.GeographicGrid(bench.COV_GRID_STEP, bench.COV_GRID_STEP)
colat, _, kn = ga.gravityfield.surface_factors(ga.kernel.get_kernel(bench.KERNEL), N, grid.parallels, bench.GM, bench.R_EARTH, grid.semimajor_axis, grid.flattening)
plan = ga.engine.Plan(N, colat, kn, grid.meridians)
P = (N + 1) ** 2
gen = torch.Generator(device='cuda').manual_seed(7)
G = torch.randn((P, 2048), dtype=torch.float64, device='cuda', generator=gen)
cov = ga.engine.gemm(G, G, transb=True, alpha=1e-22 / 2048)
cov.diagonal().add_(1e-22)
del G
lat0 = 150
sig = plan.covariance_propagation(cov, 0, lat0, lat0 + 1)
M = nb * grid.meridians.size
flops = 2.0 * M * P * P + 2.0 * M * P
for rnd in range(3):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    sig = plan.covariance_propagation(cov, 0, lat0, lat0 + nb)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    print('round %d: %.1f ms  %.2f TFLOP/s  checksum %.12e' % (rnd, ms, flops / ms / 1e9, float(sig.sum().item())), flush=True)
