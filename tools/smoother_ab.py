"""Config-5 chain of --epochs epochs (d = 1681) through smooth_block_tridiagonal_partitioned, phases timed; for A / B runs of two
builds on one box.   python tools/smoother_ab.py [--library path/to/libshg.so] [--epochs 256] [--repeats 3]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--library', default=None)
    ap.add_argument('--epochs', type=int, default=256)
    ap.add_argument('--repeats', type=int, default=3)
    args = ap.parse_args()
    import torch
    import grates_amd as ga
    import bench
    from grates_amd import distributed as gd
    if args.library:
        ga._lib.use_library(args.library)
    d, T = bench.SMOOTHER_DIM, args.epochs
    gen = torch.Generator(device='cuda')
    for r in range(args.repeats):
        sets = [bench.smoother_blocks(t, d, gen, torch, ga.engine) for t in range(T)]
        tm = {}
        gd.smooth_block_tridiagonal_partitioned([s[0] for s in sets], [s[1] for s in sets[:-1]], torch.cat([s[2] for s in sets], dim=0), consume=True, timings=tm)
        print('repeat {0}: factor {1:.4f} s  solve {2:.4f} s  covariance {3:.4f} s  -> {4:.1f} epochs/s'.format(
            r, tm['factor_s'], tm['solve_s'], tm['covariance_s'], T / (tm['factor_s'] + tm['solve_s'] + tm['covariance_s'])), flush=True)
        del sets


if __name__ == '__main__':
    main()
