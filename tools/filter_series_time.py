"""Order-wise (DDK-type) filter of 240 epochs at d/o 120: the kernel on the reference layout against the one on the order-major series
(engine.OrderMajorSeries), event-timed on the launching stream, interleaved rounds; plus the conversion kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
import numpy as np, torch
import grates_amd as ga
import inputs
if len(sys.argv) > 1 and sys.argv[1] != '-':
    ga._lib.use_library(sys.argv[1])          # an A/B build (make -C grates_amd/csrc variant ...)
N, B = 120, 240
flt = ga.filter.OrderWiseFilter(inputs.orderwise_random_blocks(42, N))
batch = torch.from_numpy(np.stack([inputs.coefficients(43 + e, N) for e in range(B)])).cuda()
series = ga.engine.OrderMajorSeries.from_batch(batch)
def timed(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n
for rnd in range(3):
    print('round %d: reference layout %.1f us   order-major series %.1f us   pack %.1f us   unpack %.1f us' % (
        rnd, timed(lambda: flt.filter_batch(batch)), timed(lambda: flt.filter_series(series)),
        timed(lambda: ga.engine.OrderMajorSeries.from_batch(batch)), timed(lambda: series.to_batch())), flush=True)
alg = 8.0 * (sum(b.size for b in flt.array) + 2.0 * (N + 1) ** 2 * B) if hasattr(flt, 'array') else None
print('algorithmic bytes', alg)
