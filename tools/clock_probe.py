"""Clocks and power the card reports (rocm-smi) while the synthesis kernel, the covariance kernel and a plain fp64 product run back to back.
Usage: python tools/clock_probe.py"""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sample(tag, stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=10).stdout
            out.append((tag[0], time.time(), r.strip()[:600]))
        except Exception as e:          # noqa: BLE001
            out.append((tag[0], time.time(), 'error ' + repr(e)))
        time.sleep(0.2)


def main():
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
    big = torch.randn((8192, 8192), dtype=torch.float64, device='cuda')
    plan.synthesis(batch, out=out)
    torch.cuda.synchronize()
    tag, stop, samples = ['idle'], threading.Event(), []
    th = threading.Thread(target=sample, args=(tag, stop, samples))
    th.start()
    time.sleep(1.0)
    for name, fn, seconds in (('synthesis', lambda: plan.synthesis(batch, out=out), 4.0), ('fp64 product', lambda: ga.engine.gemm(big, big), 4.0)):
        tag[0] = name
        t0 = time.time()
        n = 0
        while time.time() - t0 < seconds:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            n += 20
        print(name, 'launches', n, 'ms per launch', 1e3 * (time.time() - t0) / n, flush=True)
    tag[0] = 'idle after'
    time.sleep(1.0)
    stop.set()
    th.join()
    for t, ts, text in samples:
        print(t, text.replace('\n', ' ')[:400])


if __name__ == '__main__':
    main()
