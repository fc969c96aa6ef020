"""
GPU parity of the dense block operations behind the block-banded normal-equation solver:
shg_gemm / shg_potrf / shg_trtri against NumPy / SciPy on the same seeded inputs.
Tolerances: GEMM 1e-13 relative to max|C|; Cholesky factor and triangular inverse 1e-12 relative for the
well-conditioned seeded matrices used here (cond ~ 1e2).
"""

import numpy as np
import pytest
import scipy.linalg as la
import torch

import grates_amd as ga
from conftest import relerr

pytestmark = pytest.mark.gpu
eng = ga.engine


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize('ta', [False, True])
@pytest.mark.parametrize('tb', [False, True])
@pytest.mark.parametrize('M,N,K', [(1, 1, 1), (5, 7, 3), (130, 257, 33), (300, 200, 128), (128, 128, 16), (257, 129, 500), (64, 1681, 128)])
def test_gemm_transposes(ta, tb, M, N, K):
    rng = np.random.default_rng(M * 1000 + N * 10 + K)
    A = rng.standard_normal((K, M) if ta else (M, K))
    B = rng.standard_normal((N, K) if tb else (K, N))
    ref = (A.T if ta else A) @ (B.T if tb else B)
    out = eng.gemm(dev(A), dev(B), transa=ta, transb=tb).cpu().numpy()
    assert relerr(out, ref) < 1e-13


def test_gemm_seeded_random_shapes():
    """60 seeded shapes across the tile variants of the kernel (128 x 128, 64 x 64, split K), all four transposes, with
    alpha / beta and an output row stride: the paths are chosen from the shape, so the sweep covers their borders."""
    rng = np.random.default_rng(2024)
    for case in range(60):
        M, N = int(rng.integers(1, 1800)), int(rng.integers(1, 1800))
        K = int(rng.choice([1, 3, 16, 17, 100, 128, 300, 511, 512, 900, 2100]))
        if case % 3 == 0:
            M = int(rng.integers(1, 200))
        ta, tb = bool(case & 1), bool(case & 2)
        A = rng.standard_normal((K, M) if ta else (M, K))
        B = rng.standard_normal((N, K) if tb else (K, N))
        C0 = rng.standard_normal((M, N + 3))
        alpha, beta = (1.0, 0.0) if case % 4 else (-0.5, 1.5)
        ref = alpha * ((A.T if ta else A) @ (B.T if tb else B)) + beta * C0[:, 0:N]
        buf = dev(C0)
        eng.gemm(dev(A), dev(B), transa=ta, transb=tb, alpha=alpha, beta=beta, out=buf[:, 0:N])
        got = buf.cpu().numpy()
        assert relerr(got[:, 0:N], ref) < 1e-13, (case, M, N, K, ta, tb)
        np.testing.assert_array_equal(got[:, N:], C0[:, N:])            # the padding columns are untouched


@pytest.mark.parametrize('M,N,K', [(2048, 240, 2048), (2049, 226, 2061), (3000, 240, 2500), (2176, 238, 2063), (4133, 232, 2048), (2500, 178, 2100), (2300, 200, 2049),
                                   (2200, 224, 2064)])
def test_gemm_tall_products(M, N, K):
    """Tall products with 178 .. 240 columns take the whole-width stream-K kernel (gemm_tall.hip): rows and K that are no multiples of
    the tile, a partial last K tile, alpha / beta, row strides of all three operands, and the same bits on every run (the pieces of a
    cut tile are summed in a fixed order)."""
    rng = np.random.default_rng(M + N + K)
    A0 = rng.standard_normal((M, K + 5))
    B0 = rng.standard_normal((K, N + 6))
    C0 = rng.standard_normal((M, N + 2))
    A, B = dev(A0)[:, 0:K], dev(B0)[:, 0:N]
    ref = A0[:, 0:K] @ B0[:, 0:N]
    out = eng.gemm(A, B)
    assert relerr(out.cpu().numpy(), ref) < 1e-13
    again = eng.gemm(A, B)
    assert torch.equal(out, again)
    buf = dev(C0)
    eng.gemm(A, B, alpha=-0.5, beta=1.5, out=buf[:, 0:N])
    got = buf.cpu().numpy()
    assert relerr(got[:, 0:N], -0.5 * ref + 1.5 * C0[:, 0:N]) < 1e-13
    np.testing.assert_array_equal(got[:, N:], C0[:, N:])


def test_gemm_alpha_beta_and_views():
    rng = np.random.default_rng(5)
    A, B, C = rng.standard_normal((150, 90)), rng.standard_normal((90, 70)), rng.standard_normal((150, 70))
    out = dev(C)
    eng.gemm(dev(A), dev(B), alpha=-1.0, beta=1.0, out=out)
    assert relerr(out.cpu().numpy(), C - A @ B) < 1e-13
    out = dev(C)
    eng.gemm(dev(A), dev(B), alpha=0.5, beta=-2.0, out=out)
    assert relerr(out.cpu().numpy(), 0.5 * A @ B - 2 * C) < 1e-13
    # strided views: blocks of a larger matrix as operands and as the output
    big = dev(rng.standard_normal((400, 400)))
    bh = big.cpu().numpy().copy()
    eng.gemm(big[10:140, 5:95], big[200:290, 300:370], alpha=1.0, beta=1.0, out=big[150:280, 100:170])
    bh[150:280, 100:170] += bh[10:140, 5:95] @ bh[200:290, 300:370]
    assert relerr(big.cpu().numpy(), bh) < 1e-13
    # K = 0: C = beta C
    out = dev(C)
    eng.gemm(dev(np.zeros((150, 0))), dev(np.zeros((0, 70))), beta=3.0, out=out)
    np.testing.assert_array_equal(out.cpu().numpy(), 3.0 * C)
    with pytest.raises(ValueError):
        eng.gemm(dev(A), dev(C))
    with pytest.raises(ValueError):
        eng.gemm(dev(A), dev(B), beta=1.0)


def spd(seed, n):
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, n + 8))
    return G @ G.T / n + np.eye(n)


@pytest.mark.parametrize('n', [1, 2, 17, 127, 128, 129, 300, 640, 1681])
def test_potrf_and_trtri(n):
    A = spd(n, n)
    U = eng.potrf(dev(A)).cpu().numpy()
    ref = la.cholesky(A, lower=False)
    assert np.array_equal(np.tril(U, -1), np.zeros_like(U))
    assert relerr(U, ref) < 1e-12
    assert relerr(U.T @ U, A) < 1e-13
    X = eng.trtri(dev(ref)).cpu().numpy()
    assert np.array_equal(np.tril(X, -1), np.zeros_like(X))
    assert relerr(X, la.inv(ref)) < 1e-12
    assert relerr(X @ ref, np.eye(n)) < 1e-12


@pytest.mark.parametrize('n', [257, 400, 900, 1300])
def test_potrf_reproducible_bitwise(n):
    """The row panel U12 = U11^-T A12 is formed in place (the output overwrites an operand): repeated factorisations must
    agree bit for bit, whatever the order in which the workgroups of a launch run."""
    A = dev(spd(n + 1, n))
    first = eng.potrf(A.clone())
    for _ in range(4):
        assert bool((eng.potrf(A.clone()) == first).all())
    assert relerr(first.cpu().numpy(), la.cholesky(spd(n + 1, n), lower=False)) < 1e-12


def test_potrf_only_upper_triangle_referenced_and_strided():
    n = 200
    A = spd(3, n)
    junk = A.copy()
    junk[np.tril_indices(n, -1)] = 1e30                    # scipy.linalg.cholesky(lower=False) never reads it either
    U = eng.potrf(dev(junk)).cpu().numpy()
    assert relerr(U, la.cholesky(A, lower=False)) < 1e-12
    big = dev(np.zeros((300, 300)))
    big[50:250, 20:220] = dev(A)
    eng.potrf(big[50:250, 20:220])
    assert relerr(big[50:250, 20:220].cpu().numpy(), la.cholesky(A, lower=False)) < 1e-12
    assert float(big[0:50].abs().max()) == 0.0 and float(big[:, 220:].abs().max()) == 0.0


def test_potrf_not_positive_definite():
    A = spd(4, 150)
    A[140, 140] = -1.0
    with pytest.raises(np.linalg.LinAlgError):
        eng.potrf(dev(A))
    with pytest.raises(np.linalg.LinAlgError):
        eng.potrf(dev(np.zeros((3, 3))))
    with pytest.raises(ValueError):
        eng.potrf(dev(np.zeros((3, 4))))
