"""Paired elimination of a short config-5 chain (d = 1681) for a kernel trace:
  rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/smoother_trace.py 24
(profiles/r03_panel_sweep_trace.txt is one block row of such a trace, kernels by hardware queue.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import grates_amd as ga
from grates_amd import distributed as gd
import bench
T, d = int(sys.argv[1]), 1681
gen = torch.Generator(device='cuda')
def make():
    diag, upper, rhs = [], [], []
    for t in range(T):
        D, R, b = bench.smoother_blocks(t, d, gen, torch, ga.engine)
        diag.append(D); rhs.append(b)
        if t + 1 < T: upper.append(R)
    return diag, upper, torch.cat(rhs, dim=0)
for rep in range(2):
    diag, upper, rhs = make()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sc = gd._SegmentedChain(diag, upper, rhs, None, consume=True, segments=2)
    torch.cuda.synchronize()
    print('factor', time.perf_counter() - t0, flush=True)
