"""Square and block-sized products through shg_gemm, event-timed: python3 tools/gemm_square_time.py [library.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import grates_amd as ga
if len(sys.argv) > 1:
    ga._lib.use_library(sys.argv[1])
for (M, N, K, ta, tb) in ((1681, 1681, 1681, False, False), (1681, 1681, 1681, False, True), (1681, 1681, 1681, True, False), (4096, 4096, 4096, False, False),
                          (8192, 8192, 8192, False, False), (1681, 128, 1681, False, False)):
    A = torch.rand((K, M) if ta else (M, K), dtype=torch.float64, device='cuda') - 0.5
    B = torch.rand((N, K) if tb else (K, N), dtype=torch.float64, device='cuda') - 0.5
    out = torch.empty((M, N), dtype=torch.float64, device='cuda')
    for _ in range(3): ga.engine.gemm(A, B, transa=ta, transb=tb, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    a.record()
    for _ in range(reps): ga.engine.gemm(A, B, transa=ta, transb=tb, out=out)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print('%5d x %5d x %5d ta=%d tb=%d: %8.3f ms  %5.1f TFLOP/s' % (M, N, K, ta, tb, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
