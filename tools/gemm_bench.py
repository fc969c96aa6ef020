"""Timing of the fp64 MFMA GEMM (shg_dgemm) and of a covariance band: python tools/gemm_bench.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import grates_amd as ga

def timeit(f, reps=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

for M, N, K in ((4096, 4096, 4096), (8192, 8192, 8192), (14641, 240, 14641), (5760, 32761, 32761)):
    A = torch.rand((M, K), dtype=torch.float64, device='cuda') - 0.5
    B = torch.rand((K, N), dtype=torch.float64, device='cuda') - 0.5
    dt = timeit(lambda: ga.engine.dgemm(A, B))
    ref = timeit(lambda: A @ B)
    print('dgemm %6d x %6d x %6d: %8.2f ms  %6.1f TFLOP/s   (torch/rocBLAS %6.1f TFLOP/s)' % (M, N, K, dt * 1e3, 2.0 * M * N * K / dt / 1e12, 2.0 * M * N * K / ref / 1e12), flush=True)
    del A, B
