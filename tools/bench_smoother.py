#!/usr/bin/env python3
"""
BASELINE config 5 ("Kalman smoother"): block-banded normal equations of daily d/o-40 solutions (block size 1681) with a
VAR(p) constraint, factorised, solved and sparsely inverted on one MI355X; one JSON object per line.

    python tools/bench_smoother.py [--epochs 32] [--order 1] [--dim 1681] [--cpu-epochs 3]

Synthetic SPD observation normals per epoch + synthetic constraint blocks (the values do not influence the timing).
CPU baseline: the NumPy / SciPy oracle (reference formulation) on --cpu-epochs epochs, per-epoch time compared.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import grates_amd as ga  # noqa: E402

ls = ga.lstsq
eng = ga.engine


def device_ms(fn, reps=3, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def build_system(T, d, p, gen):
    """banded SPD system on the device: diagonal blocks G G^T / d + (2 p + 2) I, off-diagonal blocks small random"""
    idx = np.arange(0, (T + 1) * d, d)
    bm = ls.BlockMatrix(idx, idx)
    for t in range(T):
        G = torch.randn((d, d + 8), dtype=torch.float64, device='cuda', generator=gen)
        D = eng.gemm(G, G, transb=True, alpha=1.0 / d)
        D.diagonal().add_(2.0 * p + 2.0)
        bm._set_device(t, t, D)
        for k in range(1, p + 1):
            if t + k < T:
                bm._set_device(t, t + k, torch.randn((d, d), dtype=torch.float64, device='cuda', generator=gen) / d)
    rhs = torch.randn((T * d, 1), dtype=torch.float64, device='cuda', generator=gen)
    return ls.NormalEquations(bm, rhs, 0.0, T * d)


def emit(**kw):
    print(json.dumps(kw), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--epochs', type=int, default=32)
    ap.add_argument('--order', type=int, default=1)
    ap.add_argument('--dim', type=int, default=1681)
    ap.add_argument('--cpu-epochs', type=int, default=3)
    args = ap.parse_args()
    T, d, p = args.epochs, args.dim, args.order
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1)

    # block primitives at the block size of config 5
    A = build_system(1, d, 0, gen).matrix.device_block(0, 0)
    U = eng.potrf(A.clone())
    X = eng.trtri(U)
    B = torch.randn((d, d), dtype=torch.float64, device='cuda', generator=gen)
    emit(op='shg_potrf', n=d, ms=round(device_ms(lambda: eng.potrf(A.clone(), check=False)), 3), GFLOPs=round(d ** 3 / 3 / 1e6 / device_ms(lambda: eng.potrf(A.clone(), check=False)), 1))
    emit(op='shg_trtri', n=d, ms=round(device_ms(lambda: eng.trtri(U)), 3))
    for ta, tb in ((False, False), (True, False), (False, True)):
        ms = device_ms(lambda: eng.gemm(X, B, transa=ta, transb=tb))
        emit(op='shg_gemm', transa=ta, transb=tb, n=d, ms=round(ms, 3), TFLOPs=round(2.0 * d ** 3 / ms / 1e9, 1))

    # the smoother: factorise, solve (1 + 100 right-hand sides, as upstream), sparse inverse
    ne = build_system(T, d, p, gen)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ne.matrix.cholesky()
    torch.cuda.synchronize()
    t_chol = time.perf_counter() - t0
    ne.status = 'cholesky_factor'
    signs = torch.randint(0, 2, (T * d, 100), device='cuda', generator=gen).double() * 2.0 - 1.0     # device draw: see NormalEquations.solve
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x = ne.solve(signs=signs)
    torch.cuda.synchronize()
    t_solve = time.perf_counter() - t0
    t0 = time.perf_counter()
    ne.compute_covariance(sparse=True)
    torch.cuda.synchronize()
    t_inv = time.perf_counter() - t0
    flops_chol = T * (d ** 3 / 3 + d ** 3 / 3 + p * 2 * d ** 3 + p * (p + 1) * d ** 3)     # potrf + trtri + p panel GEMMs + updates
    emit(path='config5: block-banded smoother', epochs=T, dim=d, order=p, cholesky_s=round(t_chol, 4), solve_101rhs_s=round(t_solve, 4),
         sparse_inverse_s=round(t_inv, 4), ms_per_epoch=round((t_chol + t_solve + t_inv) / T * 1e3, 3),
         epochs_per_s=round(T / (t_chol + t_solve + t_inv), 2), cholesky_TFLOPs=round(flops_chol / t_chol / 1e12, 2),
         solution_norm=float(x.norm()) if torch.is_tensor(x) else float(np.linalg.norm(x)))

    if args.cpu_epochs > 0:
        sys.path.insert(0, ROOT)
        from oracle import lstsq_oracle as lo
        Tc = args.cpu_epochs
        rng = np.random.default_rng(2)
        bm = lo.block_matrix(np.arange(0, (Tc + 1) * d, d))
        for t in range(Tc):
            G = rng.standard_normal((d, d + 8))
            bm['blocks'][(t, t)] = G @ G.T / d + (2.0 * p + 2.0) * np.eye(d)
            for k in range(1, p + 1):
                if t + k < Tc:
                    bm['blocks'][(t, t + k)] = rng.standard_normal((d, d)) / d
        sysc = lo.normals(bm, rng.standard_normal((Tc * d, 1)), 0.0, Tc * d)
        signs = rng.integers(0, 2, size=(Tc * d, 100)) * 2.0 - 1.0
        t0 = time.perf_counter()
        lo.solve(sysc, signs)
        lo.sparse_inverse(sysc['matrix'])
        dt = time.perf_counter() - t0
        emit(path='config5 cpu_baseline (NumPy/SciPy oracle, reference formulation)', epochs=Tc, cores=os.cpu_count(), ms_per_epoch=round(dt / Tc * 1e3, 1),
             epochs_per_s=round(Tc / dt, 3))


if __name__ == '__main__':
    main()
