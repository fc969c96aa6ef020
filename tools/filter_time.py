"""Time of the order-wise block filter (config 3, block form: d/o 120, 240 epochs) from event pairs on the launching stream.
Usage: python tools/filter_time.py [--library path/to/libshg.so] [--degree 120] [--epochs 240]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--library', default=None)
    ap.add_argument('--degree', type=int, default=120)
    ap.add_argument('--epochs', type=int, default=240)
    ap.add_argument('--runs', type=int, default=200)
    args = ap.parse_args()
    import numpy as np
    import torch
    import grates_amd as ga
    import bench
    if args.library:
        ga._lib.use_library(args.library)
    nmax, T = args.degree, args.epochs
    normals = bench.orderwise_normal_blocks(44, nmax)
    weights = 1e11 * np.arange(nmax + 1, dtype=float) ** 4
    weights[0] = 1
    blocks = ga.engine.ddk_blocks(normals, weights)
    flt = ga.filter.OrderWiseFilter(blocks)
    batch = torch.from_numpy(bench.coefficient_batch(30_000, T, nmax)).cuda()
    out = flt.filter_batch(batch)
    # reference of the tool: the same block products through torch on the device
    ref = batch.clone()
    for kb, blk in enumerate(blocks):
        m = (kb + 1) // 2
        W = torch.from_numpy(np.ascontiguousarray(blk[:nmax + 1 - m, :nmax + 1 - m])).cuda()
        if kb == 0 or kb % 2 == 1:
            x = batch[:, m:, m]
            y = x @ W.T
            y[:, :max(0, 2 - m)] = x[:, :max(0, 2 - m)]
            ref[:, m:, m] = y
        else:
            x = batch[:, m - 1, m:]
            y = x @ W.T
            y[:, :max(0, 2 - m)] = x[:, :max(0, 2 - m)]
            ref[:, m - 1, m:] = y
    err = float(((out - ref).abs().max() / ref.abs().max()).item())
    for _ in range(20):
        flt.filter_batch(batch)
    torch.cuda.synchronize()
    # the device part alone: the launches are queued behind a long product, so that the host is ahead of the device and the span
    # between the two events holds kernels back to back (an event pair per call would also hold the host's time per call)
    big = torch.randn((8192, 8192), dtype=torch.float64, device='cuda')
    packed, offsets = flt._device_blocks()
    outbuf = torch.empty_like(batch)
    t = []
    for _ in range(5):
        ga.engine.gemm(big, big)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.runs):
            ga._lib.call('shg_orderwise_filter', ga.engine._ptr(packed), ga.engine._ptr(offsets), nmax, nmax, ga.engine._ptr(batch), T, ga.engine._ptr(outbuf),
                         ga.engine._stream())
        b.record()
        torch.cuda.synchronize()
        t.append(a.elapsed_time(b) / args.runs)
    t.sort()
    nbytes = 8.0 * (sum(b.size for b in blocks) + 2.0 * (nmax + 1) ** 2 * T)
    med = t[len(t) // 2]
    print('order-wise filter d/o {0}, {1} epochs: median {2:.1f} us (min {3:.1f}), {4:.2f} TB/s = {5:.3f} of 8 TB/s; max rel diff vs torch {6:.1e}'.format(
        nmax, T, med * 1e3, t[0] * 1e3, nbytes / med / 1e9, nbytes / med / 1e9 / 8.0, err))


if __name__ == '__main__':
    main()
