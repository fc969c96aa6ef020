"""Dense filter product W [14637^2] X [14637 x 240] (config 3) through shg_dense_filter: event-timed, interleaved rounds.
    python3 tools/dense_filter_time.py [library.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import grates_amd as ga
if len(sys.argv) > 1:
    ga._lib.use_library(sys.argv[1])
P, T = 14637, 240
torch.manual_seed(1)
W = torch.rand((P, P), dtype=torch.float64, device='cuda') - 0.5
X = torch.rand((P, T), dtype=torch.float64, device='cuda') - 0.5
ref = W @ X
out = ga.engine.dense_filter(W, X)
print('max rel diff vs torch', float(((out - ref).abs().max() / ref.abs().max()).item()))
for rnd in range(3):
    for _ in range(3): ga.engine.dense_filter(W, X)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): ga.engine.dense_filter(W, X)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print('round %d: %.3f ms  %.1f TFLOP/s = %.3f of 78.6' % (rnd, ms, 2.0 * P * P * T / ms / 1e9, 2.0 * P * P * T / ms / 1e9 / 78.6), flush=True)
