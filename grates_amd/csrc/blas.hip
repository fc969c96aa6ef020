// Dense fp64 building blocks of the block-banded normal-equation solver (grates/lstsq.py:698-883), all on fp64 MFMA:
//   shg_gemm        C = alpha op(A) op(B) + beta C     (op = identity / transpose, batched, optional upper-tiles-only)
//   shg_potrf       A = U^T U in place (upper)         replaces scipy.linalg.cholesky(..., lower=False)   lstsq.py:713
//   shg_trtri       X = U^-1 (upper triangular)        replaces scipy.linalg.solve_triangular / inv       lstsq.py:716, 807, 835, 856
// Triangular solves with a d x d block become GEMMs with the explicit inverse of the d x d factor, which is built
// from exact 128 x 128 diagonal-block inverses by recursive doubling:
//   [U11 U12; 0 U22]^-1 = [X11, -X11 U12 X22; 0, X22].
//
// Blocked right-looking Cholesky, nb = 128: leaf factorisation + leaf inverse in LDS (one workgroup), row panel
// U12 = U11^-T A12 and trailing update A22 -= U12^T U12 as MFMA GEMMs (upper tiles only).
#include <memory>

#include "common.h"

#ifndef SHG_GEMM_TAIL
#define SHG_GEMM_TAIL 1             // tall products in 64-tiles: the rows of the last, mostly empty round as a product of their own (split over K)
#endif
#ifndef SHG_GEMM_TALL
#define SHG_GEMM_TALL 1             // tall products with 178 .. 240 columns: whole-width tiles dealt stream-K (gemm_tall.hip)
#endif
#ifndef SHG_GEMM_STRIPS
#define SHG_GEMM_STRIPS 1          // row-strip workgroup order of tall products with 2 .. 8 column tiles (gemm_ex_kernel)
#endif

namespace shg {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int XK = 16;
constexpr int XLR = 17;      // [row][k] staging: 17-double rows (operand contiguous along k in memory)
// Output tile T x T with T = 128 (wave tile 64 x 64) or, for products with fewer than one 128-tile per CU, T = 64 (wave tile
// 32 x 32: four times as many workgroups).  [k][row] staging: T + 16 doubles per k row.
template <int T> struct GemmExTile {
    static constexpr int LK = T + 16;
    static constexpr int BUF = (T * XLR > XK * (T + 16)) ? T * XLR : XK * (T + 16);      // doubles per operand buffer
    static constexpr int F = T / 32;          // 16 x 16 fragments per wave and dimension = pieces of a thread per operand
};

struct GemmExParams {
    int M, N, K;
    const double* A;
    int lda;
    long long strideA;
    const double* B;
    int ldb;
    long long strideB;
    double* C;
    int ldc;
    long long strideC;
    double alpha, beta;
    int upper_only;           // skip output tiles that lie entirely below the diagonal
    int Ktotal;               // > 0: split-K launch, slice z covers k in [z K, min(Ktotal, (z + 1) K))
    int slices;               // split-K launch of a batch: blockIdx.z = item * slices + slice; strideA/B step the slices,
    long long itemA, itemB;   //   itemA/B the items of the batch (C: the partial products of all of them are contiguous)
    int tri;                  // triangular operands (square, M = K resp. K = N): 1 op(A) upper, 2 op(A) lower, 4 op(B) upper, 8 op(B) lower
    int strip_tiles;          // > 0: 1-d grid in row-strip order -- the `strip_tiles` column tiles of a row strip on ONE XCD, next to each other
    int strip_rows;           //      row tiles of the product (the grid is padded to a multiple of 8 of them)
};

// TA: A is stored [K][M] (op(A) = A^T);  TB: B is stored [N][K] (op(B) = B^T).  Row-major everywhere.
template <bool TA, bool TB, int T>
__global__ __launch_bounds__(256, T == 128 ? 2 : 3) void gemm_ex_kernel(GemmExParams P) {
    constexpr int XM = T, XN = T, XLK = GemmExTile<T>::LK, XBUF = GemmExTile<T>::BUF, F = GemmExTile<T>::F, HALF = T / 2;
    extern __shared__ double gemm_ex_lds[];
    double* As0 = gemm_ex_lds;                  // [2][XBUF]
    double* Bs0 = gemm_ex_lds + 2 * XBUF;       // [2][XBUF]

    // Tall products with a few column tiles (W [14637^2] X [14637 x 240]: four tiles of 64 columns): in the plain 2-d order the column
    // tiles of a row strip are consecutive workgroups, i.e. on DIFFERENT XCDs (round robin), and every one of them streams the strip of
    // op(A) from HBM into its own L2 (W read 4.3 times).  Row-strip order: workgroup L runs on XCD L % 8; the workgroups L, L + 8, ...,
    // L + 8 (c - 1) of one XCD are the c column tiles of row strip 8 (L / (8 c)) + L % 8 and start together.
    int tile_x = blockIdx.x, tile_y = blockIdx.y;
    if (P.strip_tiles > 0) {
        const int L = blockIdx.x, x = L & 7, q = L >> 3;
        tile_x = q % P.strip_tiles;
        tile_y = (q / P.strip_tiles) * 8 + x;
        if (tile_y >= P.strip_rows) return;
    }
    const int m0 = tile_y * XM, n0 = tile_x * XN;
    if (P.upper_only && m0 >= n0 + XN) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fk = lane >> 4;
    // split-K launches (Ktotal > 0) give every z slice its own K range of at most P.K (= Kz here) entries
    const int zslice = P.Ktotal > 0 ? (int)blockIdx.z % P.slices : (int)blockIdx.z;
    const int zitem = P.Ktotal > 0 ? (int)blockIdx.z / P.slices : 0;
    int Kz = P.Ktotal > 0 ? min(P.K, P.Ktotal - zslice * P.K) : P.K;
    const double* A = P.A + (size_t)zslice * P.strideA + (size_t)zitem * P.itemA;
    const double* B = P.B + (size_t)zslice * P.strideB + (size_t)zitem * P.itemB;
    double* C = P.C + (size_t)blockIdx.z * P.strideC;
    // Triangular operands: the K range of this tile shrinks to where neither operand is structurally zero (the inverse of a
    // Cholesky factor times a block: half of the K tiles on average; U^-1 U^-T: a third).  The skipped entries are never read.
    if (P.tri) {
        int kb = 0, ke = Kz;
        if (P.tri & 1) kb = max(kb, m0);                 // op(A)[i][k] = 0 for k < i
        if (P.tri & 2) ke = min(ke, m0 + XM);            // op(A)[i][k] = 0 for k > i
        if (P.tri & 4) ke = min(ke, n0 + XN);            // op(B)[k][j] = 0 for k > j
        if (P.tri & 8) kb = max(kb, n0);                 // op(B)[k][j] = 0 for k < j
        kb &= ~(XK - 1);
        kb = min(kb, max(ke, 0));
        A += TA ? (size_t)kb * P.lda : (size_t)kb;
        B += TB ? (size_t)kb : (size_t)kb * P.ldb;
        Kz = max(ke - kb, 0);
    }

    // staging roles of a thread
    //   k-contiguous operand  ([row][k] in memory):  piece h: row = (tid >> 3) + 32 h, k = (tid & 7) * 2, 2 elements
    //   row-contiguous operand ([k][row] in memory): piece h: k = (tid >> 6) + 4 h, row = (tid & 63) * 2, 2 elements
    //                                                 (T = 64: row = tid & 63, one element)
    // Indices beyond the matrix are clamped to valid addresses (the values only reach rows / columns that are never
    // stored); only the last partial K tile zero-fills.
    const int r_row = tid >> 3, r_k = (tid & 7) * 2;
    const int k_k = __builtin_amdgcn_readfirstlane(tid >> 6), k_row = T == 128 ? (tid & 63) * 2 : (tid & 63);      // k_k is wave-uniform
    double areg[8], breg[8];
    // Addressing of the steady state: fp64 MFMAs and the VALU instructions of all waves of a SIMD share one issue pipe
    // (tools/mfma64_issue.hip), so every load is "uniform 64-bit base, advanced per K tile by scalar instructions, + 32-bit
    // lane offset fixed for the whole kernel" (see gemm.hip for the idiom: the base is pinned in scalar registers, the lane
    // offset is pinned next to its use so that the scalar-base form of the load is selected).
    typedef const double __attribute__((address_space(1))) gdouble_t;
    typedef const char __attribute__((address_space(1))) gbyte_t;
    auto pin = [](unsigned v) {
        asm volatile("" : "+v"(v));
        return v;
    };
    auto at = [](const double* base, unsigned byte_off) {
        unsigned long long b = reinterpret_cast<unsigned long long>(base);
        asm volatile("" : "+s"(b));
        return *reinterpret_cast<gdouble_t*>(reinterpret_cast<gbyte_t*>(b) + byte_off);
    };
    // R-type operand ([row][k]): offsets of the F row pieces (two elements each) relative to the block's first row;
    // K-type operand ([k][row]): offsets of the column pair (T = 128) or the single column (T = 64) relative to the block's first column
    unsigned ar_off[F], br_off[F];
#pragma unroll
    for (int h = 0; h < F; ++h) {
        ar_off[h] = (unsigned)(((size_t)(min(m0 + r_row + 32 * h, P.M - 1) - m0) * P.lda + r_k) * 8);
        br_off[h] = (unsigned)(((size_t)(min(n0 + r_row + 32 * h, P.N - 1) - n0) * P.ldb + r_k) * 8);
    }
    const unsigned ak_off0 = (unsigned)((min(m0 + k_row, P.M - 1) - m0) * 8), ak_off1 = (unsigned)((min(m0 + k_row + 1, P.M - 1) - m0) * 8);
    const unsigned bk_off0 = (unsigned)((min(n0 + k_row, P.N - 1) - n0) * 8), bk_off1 = (unsigned)((min(n0 + k_row + 1, P.N - 1) - n0) * 8);
    const double* a_base = TA ? A + m0 : A + (size_t)m0 * P.lda;           // uniform
    const double* b_base = TB ? B + (size_t)n0 * P.ldb : B + n0;           // uniform

    // register layout of the staged elements: R-type piece h -> [2 h], [2 h + 1];  K-type piece h (k = k_k + 4 h) -> [2 h], [2 h + 1]
    // for T = 128 and [h] for T = 64
    auto fetch_r = [&](double* reg, const double* base, const unsigned* off, int k0) {
        const double* bk = base + k0;
#pragma unroll
        for (int h = 0; h < F; ++h) {
            const unsigned o = pin(off[h]);
            reg[2 * h] = at(bk, o);
            reg[2 * h + 1] = at(bk + 1, o);
        }
    };
    auto fetch_k = [&](double* reg, const double* base, int ld, unsigned off0, unsigned off1, int k0) {
        const unsigned o0 = pin(off0), o1 = pin(off1);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const double* bk = base + (size_t)(k0 + k_k + 4 * h) * ld;
            if (T == 128) {
                reg[2 * h] = at(bk, o0);
                reg[2 * h + 1] = at(bk, o1);
            } else {
                reg[h] = at(bk, o0);
            }
        }
    };
    auto fetch_a = [&](int k0) {
        if (!TA) fetch_r(areg, a_base, ar_off, k0); else fetch_k(areg, a_base, P.lda, ak_off0, ak_off1, k0);
    };
    auto fetch_b = [&](int k0) {
        if (TB) fetch_r(breg, b_base, br_off, k0); else fetch_k(breg, b_base, P.ldb, bk_off0, bk_off1, k0);
    };
    auto fetch_full = [&](int k0) {
        fetch_a(k0);
        fetch_b(k0);
    };
    constexpr int NLOAD_A = !TA ? 2 * F : (T == 128 ? 8 : 4);       // load instructions of a thread per operand and K tile
    constexpr int NLOAD_B = TB ? 2 * F : (T == 128 ? 8 : 4);
    // last partial K tile (runs once, plain per-lane addresses): k clamped, entries beyond K zeroed
    auto lane_ptr = [](const double* base, unsigned byte_off) { return reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + byte_off); };
    auto tail_r = [&](double* reg, const double* base, const unsigned* off, int k0) {
        const int ka = k0 + r_k, kb = ka + 1;                             // a thread's two k indices
        const int ca = min(ka, Kz - 1) - r_k, cb = min(kb, Kz - 1) - r_k;
#pragma unroll
        for (int h = 0; h < F; ++h) {
            const double* row = lane_ptr(base, off[h]);
            const double v0 = row[ca], v1 = row[cb];
            reg[2 * h] = ka < Kz ? v0 : 0.0;
            reg[2 * h + 1] = kb < Kz ? v1 : 0.0;
        }
    };
    auto tail_k = [&](double* reg, const double* base, int ld, unsigned off0, unsigned off1, int k0) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int kk = k0 + k_k + 4 * h;                              // the k row of piece h
            const double* row = base + (size_t)min(kk, Kz - 1) * ld;
            if (T == 128) {
                const double v0 = *lane_ptr(row, off0), v1 = *lane_ptr(row, off1);
                reg[2 * h] = kk < Kz ? v0 : 0.0;
                reg[2 * h + 1] = kk < Kz ? v1 : 0.0;
            } else {
                const double v0 = *lane_ptr(row, off0);
                reg[h] = kk < Kz ? v0 : 0.0;
            }
        }
    };
    auto fetch_tail = [&](int k0) {
        if (!TA) tail_r(areg, a_base, ar_off, k0); else tail_k(areg, a_base, P.lda, ak_off0, ak_off1, k0);
        if (TB) tail_r(breg, b_base, br_off, k0); else tail_k(breg, b_base, P.ldb, bk_off0, bk_off1, k0);
    };
    auto stage_r = [&](double* Xs, const double* reg) {
#pragma unroll
        for (int h = 0; h < F; ++h) {
            Xs[(r_row + 32 * h) * XLR + r_k] = reg[2 * h];
            Xs[(r_row + 32 * h) * XLR + r_k + 1] = reg[2 * h + 1];
        }
    };
    auto stage_k = [&](double* Xs, const double* reg) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            if (T == 128)
                *reinterpret_cast<double2*>(&Xs[(k_k + 4 * h) * XLK + k_row]) = make_double2(reg[2 * h], reg[2 * h + 1]);
            else
                Xs[(k_k + 4 * h) * XLK + k_row] = reg[h];
        }
    };
    auto stage = [&](int buf) {
        if (!TA) stage_r(As0 + buf * XBUF, areg); else stage_k(As0 + buf * XBUF, areg);
        if (TB) stage_r(Bs0 + buf * XBUF, breg); else stage_k(Bs0 + buf * XBUF, breg);
    };

    double4_t acc[F][F];
#pragma unroll
    for (int a = 0; a < F; ++a)
#pragma unroll
        for (int b = 0; b < F; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};

    // K tile = 4 k-steps of F x F MFMAs, organised like the K loop of gemm.hip (see there for the measurements): the fragments
    // of k-step ks + 1 are requested from LDS BEFORE the MFMAs of k-step ks are issued (two fragment sets, the scheduling
    // barriers pin the order), the loop is rotated so that the last k-step of a tile is issued behind the barrier that
    // publishes the next tile (after the request for that tile's first fragments), and the operand loads of the next tile
    // are dealt between the MFMAs of k-steps 0 and 1 instead of being issued in a row in front of them.
    double af0[F], bf0[F], af1[F], bf1[F];
#define SHG_FRAGS(af, bf, buf, ks)                                                                                             \
    do {                                                                                                                       \
        const double* As = As0 + (buf) * XBUF;                                                                                 \
        const double* Bs = Bs0 + (buf) * XBUF;                                                                                 \
        _Pragma("unroll") for (int a = 0; a < F; ++a)                                                                         \
            af[a] = TA ? As[((ks) * 4 + fk) * XLK + wr * HALF + a * 16 + fr] : As[(wr * HALF + a * 16 + fr) * XLR + (ks) * 4 + fk]; \
        _Pragma("unroll") for (int b = 0; b < F; ++b)                                                                         \
            bf[b] = TB ? Bs[(wc * HALF + b * 16 + fr) * XLR + (ks) * 4 + fk] : Bs[((ks) * 4 + fk) * XLK + wc * HALF + b * 16 + fr]; \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
    } while (0)
#define SHG_MFMAS(af, bf)                                                                                                      \
    do {                                                                                                                       \
        _Pragma("unroll") for (int a = 0; a < F; ++a)                                                                         \
            _Pragma("unroll") for (int b = 0; b < F; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
    } while (0)
#define SHG_MFMAS_LOADS(af, bf, nloads)                                                                                        \
    do {                                                                                                                       \
        _Pragma("unroll") for (int a = 0; a < F; ++a)                                                                         \
            _Pragma("unroll") for (int b = 0; b < F; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < (nloads); ++i) {                                                                \
            __builtin_amdgcn_sched_group_barrier(0x008, (F * F) / (nloads) > 0 ? (F * F) / (nloads) : 1, 0);                   \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                                 \
        }                                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
    } while (0)
    // k-steps 0 .. 2 of the tile in `buf` (set 0 holds the fragments of k-step 0); leaves k-step 3 in set 1
#define SHG_TILE_HEAD(buf, knext, prefetch)             \
    do {                                                \
        SHG_FRAGS(af1, bf1, buf, 1);                    \
        if (prefetch) {                                 \
            fetch_a(knext);                             \
            SHG_MFMAS_LOADS(af0, bf0, NLOAD_A);         \
        } else {                                        \
            SHG_MFMAS(af0, bf0);                        \
        }                                               \
        SHG_FRAGS(af0, bf0, buf, 2);                    \
        if (prefetch) {                                 \
            fetch_b(knext);                             \
            SHG_MFMAS_LOADS(af1, bf1, NLOAD_B);         \
        } else {                                        \
            SHG_MFMAS(af1, bf1);                        \
        }                                               \
        SHG_FRAGS(af1, bf1, buf, 3);                    \
        SHG_MFMAS(af0, bf0);                            \
    } while (0)

    if (Kz > 0) {
    const int nfull = Kz / XK;
    const bool has_tail = (Kz % XK) != 0;
    if (nfull > 0)
        fetch_full(0);
    else
        fetch_tail(0);
    stage(0);
    __syncthreads();
    SHG_FRAGS(af0, bf0, 0, 0);
    // branch-free steady state, two K tiles per trip so that the LDS buffer of every access is a literal
#define SHG_STEP(t, buf)                                \
    do {                                                \
        SHG_TILE_HEAD(buf, ((t) + 1) * XK, true);       \
        stage((buf) ^ 1);                               \
        __syncthreads();                                \
        SHG_FRAGS(af0, bf0, (buf) ^ 1, 0);              \
        SHG_MFMAS(af1, bf1);                            \
    } while (0)
    int t = 0;
    for (; t + 2 < nfull; t += 2) {
        SHG_STEP(t, 0);
        SHG_STEP(t + 1, 1);
    }
    if (t + 1 < nfull) SHG_STEP(t, 0);                 // t is even here
    if (nfull > 0) {                                   // last full tile; set 0 holds its first fragments
        if (has_tail) fetch_tail(nfull * XK);
        if ((nfull - 1) & 1) {
            SHG_TILE_HEAD(1, 0, false);
        } else {
            SHG_TILE_HEAD(0, 0, false);
        }
        if (has_tail) {
            stage(nfull & 1);
            __syncthreads();
            if (nfull & 1) {
                SHG_FRAGS(af0, bf0, 1, 0);
            } else {
                SHG_FRAGS(af0, bf0, 0, 0);
            }
        }
        SHG_MFMAS(af1, bf1);
    }
    if (has_tail) {
        if (nfull & 1) {
            SHG_TILE_HEAD(1, 0, false);
        } else {
            SHG_TILE_HEAD(0, 0, false);
        }
        SHG_MFMAS(af1, bf1);
    }
    }   // Kz > 0
#undef SHG_STEP
#undef SHG_TILE_HEAD
#undef SHG_MFMAS_LOADS
#undef SHG_MFMAS
#undef SHG_FRAGS

    // epilogue.  C/D layout: column = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int a = 0; a < F; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = m0 + wr * HALF + a * 16 + fk + 4 * r;
            if (gr >= P.M) continue;
#pragma unroll
            for (int b = 0; b < F; ++b) {
                const int gc = n0 + wc * HALF + b * 16 + fr;
                if (gc >= P.N) continue;
                double* c = C + (size_t)gr * P.ldc + gc;
                const double v = P.alpha * acc[a][b][r];
                *c = P.beta == 0.0 ? v : fma(P.beta, *c, v);
            }
        }
}

__global__ void scale_kernel(int M, int N, double beta, double* __restrict__ C, int ldc, long long strideC) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    double* p = C + (size_t)blockIdx.z * strideC + (size_t)blockIdx.y * ldc + c;
    *p = beta == 0.0 ? 0.0 : beta * *p;
}

__global__ void splitk_reduce_kernel(int M, int N, int slices, double alpha, const double* __restrict__ partial, double beta,
                                     double* __restrict__ C, int ldc, long long strideC) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    const size_t e = (size_t)blockIdx.y * N + c;
    partial += (size_t)blockIdx.z * slices * M * N;
    double s = 0.0;
    for (int z = 0; z < slices; ++z) s += partial[(size_t)z * M * N + e];
    double* out = C + (size_t)blockIdx.z * strideC + (size_t)blockIdx.y * ldc + c;
    *out = beta == 0.0 ? alpha * s : fma(beta, *out, alpha * s);
}

// Products with K <= 128 and a thin output (at most 128 rows or 128 columns): the panel steps of the factorisation, which sit on
// its serial chain between two leaves.  The tiled kernel above spends 8 - 29 us on them (a 128 x 128 x 128 product is ONE
// 128-tile: one CU runs the whole K loop through its LDS pipeline); here every wave owns one 16 x 16 output tile and holds its
// whole K range in registers -- 2 x 32 loads in flight at once, then 32 MFMAs back to back -- and a workgroup is the eight
// row tiles of one 16-column strip (or of one 128-row group).  C may be B (the in-place row panel U12 = X11^T A12: the strip is
// read by this workgroup only, and a barrier separates its last load from its first store).
struct PanelParams {
    int M, N, K;
    const double* A;
    int lda;
    const double* B;
    int ldb;
    double* C;
    int ldc;
    double alpha, beta;
    int upper_only;           // skip tiles entirely below the diagonal
    int a_lower;              // op(A) is lower triangular (square, M = K): k > i is not read
    int b_upper;              // op(B) is upper triangular (square, K = N): k > j is not read
    long long strideA, strideB, strideC;      // batch: item blockIdx.z
};

template <bool TA>
__global__ __launch_bounds__(512) void panel_kernel(PanelParams P) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const int i0 = blockIdx.y * 128 + wave * 16, j0 = blockIdx.x * 16;
    const bool active = i0 < P.M && !(P.upper_only && j0 + 15 < i0);
    P.A += (size_t)blockIdx.z * P.strideA;
    P.B += (size_t)blockIdx.z * P.strideB;
    P.C += (size_t)blockIdx.z * P.strideC;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    double cold[4] = {0.0, 0.0, 0.0, 0.0};
    if (active) {
        int k4end = (P.K + 3) / 4;
        if (P.a_lower) k4end = min(k4end, (i0 + 16 + 3) / 4);
        if (P.b_upper) k4end = min(k4end, (j0 + 16 + 3) / 4);
        const int i = i0 + fr, j = j0 + fr;
        const bool iok = i < P.M, jok = j < P.N;
        double a[32], b[32];
#pragma unroll
        for (int k4 = 0; k4 < 32; ++k4) {
            const int k = 4 * k4 + fk;
            const bool kok = k4 < k4end && k < P.K;
            a[k4] = (kok && iok) ? (TA ? P.A[(size_t)k * P.lda + i] : P.A[(size_t)i * P.lda + k]) : 0.0;
            b[k4] = (kok && jok) ? P.B[(size_t)k * P.ldb + j] : 0.0;
        }
        if (P.beta != 0.0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = i0 + fk + 4 * r;
                if (ci < P.M && jok) cold[r] = P.C[(size_t)ci * P.ldc + j];
            }
        }
#pragma unroll
        for (int k4 = 0; k4 < 32; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k4], b[k4], acc, 0, 0, 0);
    }
    if ((const double*)P.C == P.B) __syncthreads();
    if (active) {
        const int j = j0 + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = i0 + fk + 4 * r;
            if (ci < P.M && j < P.N) P.C[(size_t)ci * P.ldc + j] = P.beta == 0.0 ? P.alpha * acc[r] : fma(P.beta, cold[r], P.alpha * acc[r]);
        }
    }
}

// A block times a handful of right-hand sides, op(A) = A (the backward sweep W x = y of the smoother with one solution vector: N = 1):
// the block is read once from memory and the product is bound by that.  The tiled kernel made 27 workgroups with a K loop of 1681,
// or a split-K launch plus its reduction, out of it (15 us); here one wave takes one output row, its lanes stream along the
// contiguous index k of the block and are summed at the end (sweep of config 5: 0.27 -> 0.08 s).  The transposed product (forward
// sweep) stays with the tiled kernel: with the lanes along the output rows every wave walks down a column strip one row per load,
// and a version with eight waves per 64 rows was three times slower than the split-K launch.
// tri as in GemmExParams (square operands): the structurally zero part of A is not read.
constexpr int kGemvColumns = 8;

struct GemvParams {
    int M, N, K;
    const double* A;
    int lda;
    const double* X;
    int ldx;
    double* Y;
    int ldy;
    double alpha, beta;
    int tri;
};

__global__ __launch_bounds__(256) void gemv_rows_kernel(GemvParams P) {              // op(A) = A
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= P.M) return;
    int kb = 0, ke = P.K;
    if (P.tri & 1) kb = i & ~63;                     // op(A)[i][k] = 0 for k < i
    if (P.tri & 2) ke = min(ke, i + 1);              // ... for k > i
    double s[kGemvColumns];
#pragma unroll
    for (int c = 0; c < kGemvColumns; ++c) s[c] = 0.0;
    const double* a = P.A + (size_t)i * P.lda;
    for (int k = kb + lane; k < ke; k += 64) {
        const double v = ((P.tri & 1) && k < i) ? 0.0 : a[k];
#pragma unroll
        for (int c = 0; c < kGemvColumns; ++c)
            if (c < P.N) s[c] = fma(v, P.X[(size_t)k * P.ldx + c], s[c]);
    }
#pragma unroll
    for (int c = 0; c < kGemvColumns; ++c) {
        if (c >= P.N) break;
        double v = s[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) {
            double* y = P.Y + (size_t)i * P.ldy + c;
            *y = P.beta == 0.0 ? P.alpha * v : fma(P.beta, *y, P.alpha * v);
        }
    }
}

static bool gemv_shape(bool ta, bool tb, int M, int N, int K, int batch, bool upper_only, int tri, const double* A, const double* B, const double* C) {
    return !ta && !tb && batch == 1 && !upper_only && N <= kGemvColumns && M >= 256 && K >= 256 && C != A && C != B && (tri & ~3) == 0 &&
           (tri == 0 || M == K);
}

static int gemv(int M, int N, int K, double alpha, const double* A, int lda, const double* X, int ldx, double beta, double* Y, int ldy, int tri,
                hipStream_t stream) {
    GemvParams P{M, N, K, A, lda, X, ldx, Y, ldy, alpha, beta, tri};
    hipLaunchKernelGGL(gemv_rows_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, stream, P);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

static bool panel_shape(bool tb, int M, int N, int K, int batch, const double* A, const double* B, const double* C) {
    if (tb || batch > 4 || K > 128 || C == A) return false;
    if (C == B) return M <= 128;
    return M <= 128 || N <= 128;
}

static int panel_gemm(bool ta, int M, int N, int K, double alpha, const double* A, int lda, long long strideA, const double* B, int ldb,
                      long long strideB, double beta, double* C, int ldc, long long strideC, int batch, bool upper_only, bool a_lower, bool b_upper,
                      hipStream_t stream) {
    PanelParams P{M, N, K, A, lda, B, ldb, C, ldc, alpha, beta, upper_only ? 1 : 0, a_lower ? 1 : 0, b_upper ? 1 : 0, strideA, strideB, strideC};
    const dim3 grid(ceil_div(N, 16), ceil_div(M, 128), batch);
    if (ta)
        hipLaunchKernelGGL(panel_kernel<true>, grid, dim3(512), 0, stream, P);
    else
        hipLaunchKernelGGL(panel_kernel<false>, grid, dim3(512), 0, stream, P);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

int gemm_ex_tri(bool ta, bool tb, int M, int N, int K, double alpha, const double* A, int lda, long long strideA, const double* B, int ldb,
                long long strideB, double beta, double* C, int ldc, long long strideC, int batch, bool upper_only, int tri, hipStream_t stream);

int gemm_ex(bool ta, bool tb, int M, int N, int K, double alpha, const double* A, int lda, long long strideA, const double* B, int ldb,
            long long strideB, double beta, double* C, int ldc, long long strideC, int batch, bool upper_only, hipStream_t stream) {
    return gemm_ex_tri(ta, tb, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, batch, upper_only, 0, stream);
}

// tri: triangular structure of the operands (see GemmExParams::tri); entries on the zero side are not read
int gemm_ex_tri(bool ta, bool tb, int M, int N, int K, double alpha, const double* A, int lda, long long strideA, const double* B, int ldb,
                long long strideB, double beta, double* C, int ldc, long long strideC, int batch, bool upper_only, int tri, hipStream_t stream) {
    if (M <= 0 || N <= 0 || batch <= 0) return SHG_OK;
    if (SHG_GEMM_TALL && K > 0 && gemm_tall_shape(ta, tb, M, N, K, batch, upper_only, tri, A, lda, B, ldb, C))
        return gemm_tall(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, stream);
    // Tall products in 64-tiles whose last round of workgroups would be mostly empty (W [14637^2] X [14637 x 240]: 916 tiles on 768 workgroup
    // slots -- the card works for four tiles per CU where 3.6 are needed): the rows of the whole rounds first, the remaining rows as a product of
    // their own, which the rules below run as 128-tiles split over K (every CU gets a piece; the partial products are summed in a fixed order).
#ifdef SHG_GEMM_TALL128
    const bool tall128 = K >= 2048 && batch == 1 && !upper_only && tri == 0 && N <= 256 && M >= 8192;
#else
    const bool tall128 = false;
#endif
    if (SHG_GEMM_TAIL && !tall128 && K >= 2048 && batch == 1 && !upper_only && tri == 0 && !ta && (const double*)C != A && (const double*)C != B) {
        const int col_tiles = ceil_div(N, 64), row_tiles = ceil_div(M, 64);
        const long long tiles = (long long)col_tiles * row_tiles;
        constexpr int kSlots = 768;                                      // 64-tile workgroups the card holds at once (three per CU)
        const int rows_per_round = kSlots / std::max(col_tiles, 1);
        if (col_tiles <= 8 && tiles >= 512 && tiles > kSlots && rows_per_round >= 8) {
            const int whole = (row_tiles / rows_per_round) * rows_per_round, rest = row_tiles - whole;
            if (whole > 0 && rest > 0 && rest * 2 <= rows_per_round) {
                const int m_main = whole * 64;
                int rc = gemm_ex_tri(ta, tb, m_main, N, K, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, batch, upper_only, tri, stream);
                if (rc) return rc;
                return gemm_ex_tri(ta, tb, M - m_main, N, K, alpha, A + (size_t)m_main * lda, lda, strideA, B, ldb, strideB, beta, C + (size_t)m_main * ldc, ldc, strideC, batch,
                                   upper_only, tri, stream);
            }
        }
    }
    if (K > 0 && gemv_shape(ta, tb, M, N, K, batch, upper_only, tri, A, B, C)) return gemv(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, tri, stream);
    if (K > 0 && panel_shape(tb, M, N, K, batch, A, B, C) && !(upper_only && M != N))
        return panel_gemm(ta, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, batch, upper_only, (tri & 2) != 0 && M == K,
                          (tri & 4) != 0 && K == N, stream);
    GemmExParams P;
    P.tri = tri;
    P.M = M;
    P.N = N;
    P.K = K;
    P.A = A;
    P.lda = lda;
    P.strideA = strideA;
    P.B = B;
    P.ldb = ldb;
    P.strideB = strideB;
    P.C = C;
    P.ldc = ldc;
    P.strideC = strideC;
    P.alpha = alpha;
    P.beta = beta;
    P.upper_only = upper_only ? 1 : 0;
    if (K <= 0) {                      // empty sum: C = beta C
        hipLaunchKernelGGL(scale_kernel, dim3(ceil_div(N, 256), M, batch), dim3(256), 0, stream, M, N, beta, C, ldc, strideC);
        SHG_HIP(hipGetLastError());
        return SHG_OK;
    }
    P.Ktotal = 0;
    P.slices = 1;
    P.itemA = P.itemB = 0;
    P.strip_tiles = P.strip_rows = 0;
    // 128 x 128 output tiles; products with fewer of them than the card holds at once (512: the K = 128 panel updates of the
    // blocked factorisation, the d = 1681 block products of the smoother alone or as a batch of two) take 64 x 64 tiles, four
    // times as many workgroups: a single partial round of 128-tiles lasts as long as its longest tile (a batch of two
    // 1681^3 products with a triangular operand: 389 us in 392 tiles, 24 TFLOP/s)
    long long work_tiles = (long long)ceil_div(N, 128) * ceil_div(M, 128) * batch;
    if (upper_only) work_tiles = (work_tiles + ceil_div(N, 128)) / 2;
    // (a long K with too few 64-tiles to fill the chip is split below instead; an output that overwrites an operand -- the
    //  in-place row panel U12 = U11^-T A12 of the factorisation -- relies on one workgroup owning a whole column tile of that
    //  operand: 128-row tiles only)
    const long long tiles64 = (long long)ceil_div(N, 64) * ceil_div(M, 64) * batch;
    const bool split_candidate = batch <= 2 && !upper_only && K >= 512 && tiles64 < 256;
    // (and products of at most 64 rows with many column tiles: a 128-row tile would be more than half empty)
    const bool narrow = M <= 64 && !upper_only && tiles64 >= 512;
    const bool small_tiles = (work_tiles < 512 || narrow) && !split_candidate && !tall128 && (const double*)C != A && (const double*)C != B;
    const int XT = small_tiles ? 64 : 128;
    dim3 grid(ceil_div(N, XT), ceil_div(M, XT), batch);
    const size_t lds = (size_t)4 * (small_tiles ? GemmExTile<64>::BUF : GemmExTile<128>::BUF) * sizeof(double);   // 73.7 KB (two workgroups per CU) / 41 KB
    if (SHG_GEMM_STRIPS && batch == 1 && !upper_only && grid.x >= 2 && grid.x <= 8 && grid.y >= 32) {        // tall and skinny: row-strip order
        P.strip_tiles = (int)grid.x;
        P.strip_rows = (int)grid.y;
        grid = dim3((unsigned)(8 * P.strip_tiles * ceil_div((int)grid.y, 8)), 1, 1);
    }
    // Few output tiles and a long K (block times a handful of right-hand sides): split K over grid.z into a workspace of
    // partial products that a second kernel sums in a fixed order (deterministic, unlike atomics).
    const int tiles = (int)(grid.x * grid.y) * batch;
    double* partial = nullptr;
    int slices = 1;
    std::unique_ptr<ScratchLease> lease;            // held until the kernel that sums the partial products is enqueued
    if (!small_tiles && batch <= 2 && !upper_only && tiles < 384 && K >= 512) {                 // fewer than 1.5 workgroups per CU
        slices = std::min(std::min(16, K / 128), std::max(1, 512 / tiles));
#ifdef SHG_GEMM_TALL128
        if (tall128) slices = SHG_GEMM_TALL128;
#endif
        if (slices > 1) {
            const int chunk = round_up(ceil_div(K, slices), XK);
            slices = ceil_div(K, chunk);
            if (slices > 1) {
                lease.reset(new ScratchLease(stream));
                partial = (double*)lease->get(kScratchSplitK, (size_t)batch * slices * M * N * sizeof(double));
            }
            if (slices > 1 && partial != nullptr) {
                P.K = chunk;
                P.Ktotal = K;
                P.slices = slices;
                P.itemA = strideA;
                P.itemB = strideB;
                P.strideA = ta ? (long long)chunk * lda : chunk;
                P.strideB = tb ? chunk : (long long)chunk * ldb;
                P.C = partial;
                P.ldc = N;
                P.strideC = (long long)M * N;
                P.alpha = 1.0;
                P.beta = 0.0;
                P.tri = 0;                              // (K slices and triangular K ranges are not combined)
                grid.z = slices * batch;
            } else {
                slices = 1;
                partial = nullptr;
            }
        }
    }
#define SHG_GEMM_EX(TA_, TB_)                                                                                                          \
    do {                                                                                                                                \
        if (small_tiles) {                                                                                                              \
            SHG_SET_LDS_ONCE((gemm_ex_kernel<TA_, TB_, 64>), lds);                                                                      \
            hipLaunchKernelGGL((gemm_ex_kernel<TA_, TB_, 64>), grid, dim3(256), lds, stream, P);                                        \
        } else {                                                                                                                        \
            SHG_SET_LDS_ONCE((gemm_ex_kernel<TA_, TB_, 128>), lds);                                                                     \
            hipLaunchKernelGGL((gemm_ex_kernel<TA_, TB_, 128>), grid, dim3(256), lds, stream, P);                                       \
        }                                                                                                                               \
    } while (0)
    if (ta) {
        if (tb) SHG_GEMM_EX(true, true); else SHG_GEMM_EX(true, false);
    } else {
        if (tb) SHG_GEMM_EX(false, true); else SHG_GEMM_EX(false, false);
    }
#undef SHG_GEMM_EX
    if (partial) {
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(ceil_div(N, 256), M, batch), dim3(256), 0, stream, M, N, slices, alpha, partial, beta, C, ldc, strideC);
    }
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

// ------------------------------------------------------------------------------------------------
// leaf: one n x n diagonal block (n <= 128) in LDS, one workgroup
//   mode bit 0: factor the upper triangle in place (A = U^T U) and write U back (strictly lower part of the block zeroed)
//   mode bit 1: invert the (factored or given) upper triangular block into Xout[n][ldx]
// ------------------------------------------------------------------------------------------------
constexpr int LEAF = 128;
constexpr int LLD = LEAF + 1;

// 1 / x from the hardware estimate and two Newton steps (the IEEE division is a ~40-instruction dependent chain that
// sits on the critical path of every elimination step); relative error <= 2^-52
__device__ __forceinline__ double fast_reciprocal(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// Thread (ty, tx) = (tid >> 4, tid & 15) of the 32 x 16 thread grid owns the 4 x 8 register tile
//   u[ii][cc] = U[ty + 32 ii][tx + 16 cc]          (cyclic distribution: the work stays balanced as k advances)
// Blocks smaller than 128 are padded with the identity.  Every elimination step broadcasts one row through a double
// buffered LDS line pair (one barrier per two steps) and does its rank-1 updates in registers; the steps are grouped by 32-row
// groups (template parameter) so that all register indices are compile-time constants.  Two waves per SIMD.

// the 32 elimination steps k = 32 KI .. 32 KI + 31 of the factorisation, two per barrier: the owners of the rows k and k + 1
// publish them as they are (row k + 1 not yet updated by step k); every thread forms the updated row k + 1 for the columns
// it needs itself, with exactly the operations its owner would have used, so the result is bit-identical to one step per
// barrier while the serial chain has half as many barrier / LDS round trips
template <int KI>
__device__ __forceinline__ void leaf_factor_group(double (&u)[4][8], double (*rowbuf)[2][LEAF], int* bad, int ty, int tx, int tid) {
    for (int kr = 0; kr < 32; kr += 2) {
        const int k = KI * 32 + kr, buf = (kr >> 1) & 1;
        if (ty == kr || ty == kr + 1) {
            double* dst = rowbuf[buf][ty - kr];
#pragma unroll
            for (int cc = 2 * KI; cc < 8; ++cc) dst[tx + 16 * cc] = u[KI][cc];
        }
        __syncthreads();
        const double* r0 = rowbuf[buf][0];
        const double* r1raw = rowbuf[buf][1];
        double piv0 = r0[k];
        if (!(piv0 > 0.0)) {                     // not positive definite (also catches NaN); the same value in all threads
            if (tid == 0 && *bad == 0) *bad = k + 1;
            piv0 = 1.0;
        }
        const double inv0 = fast_reciprocal(piv0);
        const double f01 = r0[k + 1] * inv0;                             // multiplier of row k + 1 in step k
        double piv1 = fma(-f01, r0[k + 1], r1raw[k + 1]);
        if (!(piv1 > 0.0)) {
            if (tid == 0 && *bad == 0) *bad = k + 2;
            piv1 = 1.0;
        }
        const double inv1 = fast_reciprocal(piv1);
        double c0[8], c1[8];                                             // rows k and k + 1 (after step k) at this thread's columns
#pragma unroll
        for (int cc = 2 * KI; cc < 8; ++cc) {
            c0[cc] = r0[tx + 16 * cc];
            c1[cc] = fma(-f01, c0[cc], r1raw[tx + 16 * cc]);
        }
#pragma unroll
        for (int ii = KI; ii < 4; ++ii) {
            const int i = ty + 32 * ii;
            if (ii > KI || ty > kr) {            // row i > k
                const double f0 = r0[i] * inv0;
                const bool both = ii > KI || ty > kr + 1;                // row i > k + 1: step k + 1 as well
                const double f1 = both ? fma(-f01, r0[i], r1raw[i]) * inv1 : 0.0;
#pragma unroll
                for (int cc = 2 * ii; cc < 8; ++cc)
                    if (cc >= 2 * ii + 2 || tx + 16 * cc >= i) {
                        double v = fma(-f0, c0[cc], u[ii][cc]);
                        if (both) v = fma(-f1, c1[cc], v);
                        u[ii][cc] = v;
                    }
            }
        }
    }
}

// X = U^-1 of the 128 x 128 factor held in LDS (Ul, row-major, row length LLD) by recursive doubling on the MFMA:
//   the eight 16 x 16 diagonal blocks are inverted by one wave each (back substitution, one column per lane, 16 dependent
//   steps), then three levels s = 16, 32, 64 of  X12 = -X11 (U12 X22)  with 16 x 16 MFMA tiles dealt to the 8 waves.
// 16 dependent steps + 6 small products replace the 128 dependent substitution steps of a row-by-row inversion (36 us -> ~12).
// Storage: X is upper triangular like U, so X^T lives in the strictly LOWER triangle of the same LDS array (X[i][j], j > i,
// at Ul[j][i]) and its diagonal in dg[]; U's upper triangle stays intact.  The temporary T = U12 X22 of a level sits in the
// slots of the X12 block it is about to become.
__device__ __forceinline__ double leaf_x_at(const double* Ul, const double* dg, int i, int j) {
    const double off = Ul[(j > i ? j : i) * LLD + i];                 // (a valid address for every lane)
    return j > i ? off : (j == i ? dg[i] : 0.0);
}

__device__ __forceinline__ void leaf_invert_doubling(double* Ul, double* dg, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    // ---- diagonal blocks: wave w inverts block w, lane c < 16 column c
    if (lane < 16) {
        const int b0 = wave * 16, c = lane;
        double x[16];
#pragma unroll
        for (int i = 15; i >= 0; --i) {
            double sum = (i == c) ? 1.0 : 0.0;
#pragma unroll
            for (int j = i + 1; j < 16; ++j) sum = fma(-Ul[(b0 + i) * LLD + b0 + j], x[j], sum);      // x[j] = 0 for j > c
            x[i] = (i <= c) ? sum * dg[b0 + i] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 15; ++i)
            if (i < c) Ul[(b0 + c) * LLD + b0 + i] = x[i];
    }
    __syncthreads();
    // ---- doubling levels
#pragma unroll
    for (int s = 16; s <= 64; s *= 2) {
        const int tps = s / 16;                         // tiles per block side
        const int ntiles = (LEAF / (2 * s)) * tps * tps;   // 4, 8, 16
        double4_t acc[2];
        // T = U12 X22, tile (ti, tj) of pair q -> the slots of X12
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int tau = wave + 8 * rep;
            acc[rep] = (double4_t){0.0, 0.0, 0.0, 0.0};
            if (tau < ntiles) {
                const int q = tau / (tps * tps), ti = (tau / tps) % tps, tj = tau % tps;
                const int r0 = q * 2 * s, c1 = r0 + s;
                for (int k4 = 0; k4 < s / 4; ++k4) {
                    const double a = Ul[(r0 + ti * 16 + fr) * LLD + c1 + k4 * 4 + fk];
                    const double b = leaf_x_at(Ul, dg, c1 + k4 * 4 + fk, c1 + tj * 16 + fr);
                    acc[rep] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[rep], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {                  // (nothing reads the X12 slots before they hold T: no barrier)
            const int tau = wave + 8 * rep;
            if (tau < ntiles) {
                const int q = tau / (tps * tps), ti = (tau / tps) % tps, tj = tau % tps;
                const int r0 = q * 2 * s, c1 = r0 + s;
#pragma unroll
                for (int r = 0; r < 4; ++r) Ul[(c1 + tj * 16 + fr) * LLD + r0 + ti * 16 + fk + 4 * r] = acc[rep][r];      // T[i][j] at the slot of X12[i][j]
            }
        }
        __syncthreads();
        // X12 = -X11 T
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int tau = wave + 8 * rep;
            acc[rep] = (double4_t){0.0, 0.0, 0.0, 0.0};
            if (tau < ntiles) {
                const int q = tau / (tps * tps), ti = (tau / tps) % tps, tj = tau % tps;
                const int r0 = q * 2 * s, c1 = r0 + s;
                for (int k4 = 0; k4 < s / 4; ++k4) {
                    const double a = leaf_x_at(Ul, dg, r0 + ti * 16 + fr, r0 + k4 * 4 + fk);
                    const double b = Ul[(c1 + tj * 16 + fr) * LLD + r0 + k4 * 4 + fk];               // T[k][j]
                    acc[rep] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[rep], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                // all reads of T done
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int tau = wave + 8 * rep;
            if (tau < ntiles) {
                const int q = tau / (tps * tps), ti = (tau / tps) % tps, tj = tau % tps;
                const int r0 = q * 2 * s, c1 = r0 + s;
#pragma unroll
                for (int r = 0; r < 4; ++r) Ul[(c1 + tj * 16 + fr) * LLD + r0 + ti * 16 + fk + 4 * r] = -acc[rep][r];
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(512) void leaf_kernel(int n, double* __restrict__ A, int lda, long long strideA, double* __restrict__ X,
                                                   int ldx, long long strideX, int mode, int* __restrict__ info, int info_base, int info_stride) {
    extern __shared__ double Ul[];                      // [LEAF][LLD] factor, inversion phase only
    __shared__ double rowpair[2][2][LEAF];             // two published rows per barrier, double buffered
    __shared__ double dg[LEAF];
    __shared__ int bad;
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    A += (size_t)blockIdx.x * strideA;
    if (X) X += (size_t)blockIdx.x * strideX;
    if (tid == 0) bad = 0;

    double u[4][8];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            const int i = ty + 32 * ii, c = tx + 16 * cc;
            u[ii][cc] = (i < n && c < n) ? (c >= i ? A[(size_t)i * lda + c] : 0.0) : (i == c ? 1.0 : 0.0);
        }

    if (mode & 1) {
        // right-looking elimination with the row scaling deferred: step k uses the unscaled pivot row,
        //   U[i][c] -= U[k][i] U[k][c] / U[k][k]   (k < i <= c)
        leaf_factor_group<0>(u, rowpair, &bad, ty, tx, tid);
        leaf_factor_group<1>(u, rowpair, &bad, ty, tx, tid);
        leaf_factor_group<2>(u, rowpair, &bad, ty, tx, tid);
        leaf_factor_group<3>(u, rowpair, &bad, ty, tx, tid);
        // scale the rows: U[k][c] /= sqrt(d_k)
        __syncthreads();
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int cc = 0; cc < 8; ++cc)
                if (ty + 32 * ii == tx + 16 * cc) dg[ty + 32 * ii] = u[ii][cc];
        __syncthreads();
        if (bad) {
            // one flag for the batch (info_stride 0: the item shows in the value) or one flag per item
            if (tid == 0 && info) atomicCAS(info + (size_t)blockIdx.x * info_stride, 0, info_base + (info_stride ? 0 : (int)blockIdx.x * LEAF) + bad);
            return;
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const double s = sqrt(dg[ty + 32 * ii]);
            const double rs = 1.0 / s;
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                const int i = ty + 32 * ii, c = tx + 16 * cc;
                if (c > i) u[ii][cc] *= rs;
                if (c == i) u[ii][cc] = s;
                if (i < n && c < n) A[(size_t)i * lda + c] = c >= i ? u[ii][cc] : 0.0;
            }
        }
    }
    if (mode & 2) {
        // X = U^-1 by recursive doubling in LDS (leaf_invert_doubling)
        __syncthreads();                                 // dg is rewritten below
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                const int i = ty + 32 * ii, c = tx + 16 * cc;
                if (c >= i) Ul[i * LLD + c] = u[ii][cc];
                if (i == c) dg[i] = 1.0 / u[ii][cc];
            }
        __syncthreads();
        leaf_invert_doubling(Ul, dg, tid);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                const int i = ty + 32 * ii, c = tx + 16 * cc;
                if (i < n && c < n) X[(size_t)i * ldx + c] = c > i ? Ul[c * LLD + i] : (c == i ? dg[i] : 0.0);
            }
    }
}

static int launch_leaf(int n, double* A, int lda, long long strideA, double* X, int ldx, long long strideX, int batch, int mode, int* info,
                       int info_base, hipStream_t stream, int info_stride = 0) {
    const size_t lds = (mode & 2) ? (size_t)LEAF * LLD * sizeof(double) : 0;
    SHG_SET_LDS_ONCE(leaf_kernel, LEAF * LLD * sizeof(double));
    hipLaunchKernelGGL(leaf_kernel, dim3(batch), dim3(512), lds, stream, n, A, lda, strideA, X, ldx, strideX, mode, info, info_base, info_stride);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

// batch of n x n blocks (n <= 128): A_b = U_b^T U_b in place and X_b = U_b^-1, one workgroup per block (analysis.hip)
int factor_invert_batched(int n, double* A, int lda, long long strideA, double* X, int ldx, long long strideX, int batch, int* info,
                          hipStream_t stream) {
    if (n > LEAF) return fail(SHG_ERR_INVALID, "factor_invert_batched: block size %d exceeds %d", n, LEAF);
    return launch_leaf(n, A, lda, strideA, X, ldx, strideX, batch, 3, info, 0, stream);
}

__global__ void zero_lower_kernel(int n, double* __restrict__ A, int lda, long long strideA) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)n * n) return;
    const int r = (int)(e / n), c = (int)(e % n);
    if (c < r) A[(size_t)blockIdx.y * strideA + (size_t)r * lda + c] = 0.0;
}

static int zero_lower(int n, double* A, int lda, long long strideA, int batch, hipStream_t stream) {
    if (n <= 1) return SHG_OK;
    hipLaunchKernelGGL(zero_lower_kernel, dim3((unsigned)ceil_div64((long long)n * n, 256), batch), dim3(256), 0, stream, n, A, lda, strideA);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

// X = U^-1 from the inverses of the 128 x 128 diagonal blocks (already in X) by recursive doubling: units of size s are
// complete; neighbours (a, a + s) merge into units of 2 s with X12 = -X11 (U12 X22).  work: n * n / 4 doubles.
static int trtri_doubling(int n, const double* U, int ldu, double* X, int ldx, double* work, hipStream_t stream) {
    int rc;
    for (int s = LEAF; s < n; s *= 2) {
        // pairs with a full right unit: one batched launch per product (uniform strides along the diagonal)
        const int nfullpairs = n / (2 * s);
        if (nfullpairs > 0) {
            rc = gemm_ex(false, false, s, s, s, 1.0, U + s, ldu, (long long)2 * s * (ldu + 1), X + (size_t)s * (ldx + 1), ldx,
                         (long long)2 * s * (ldx + 1), 0.0, work, s, (long long)s * s, nfullpairs, false, stream);
            if (rc) return rc;
            rc = gemm_ex(false, false, s, s, s, -1.0, X, ldx, (long long)2 * s * (ldx + 1), work, s, (long long)s * s, 0.0, X + s, ldx,
                         (long long)2 * s * (ldx + 1), nfullpairs, false, stream);
            if (rc) return rc;
        }
        const int a = nfullpairs * 2 * s;                       // a last pair with a ragged right unit
        if (a + s < n) {
            const int n2 = n - (a + s);
            const double* U12 = U + (size_t)a * ldu + (a + s);
            const double* X11 = X + (size_t)a * ldx + a;
            const double* X22 = X + (size_t)(a + s) * ldx + (a + s);
            double* X12 = X + (size_t)a * ldx + (a + s);
            rc = gemm_ex(false, false, s, n2, n2, 1.0, U12, ldu, 0, X22, ldx, 0, 0.0, work, n2, 0, 1, false, stream);
            if (rc) return rc;
            rc = gemm_ex(false, false, s, n2, s, -1.0, X11, ldx, 0, work, n2, 0, 0.0, X12, ldx, 0, 1, false, stream);
            if (rc) return rc;
        }
    }
    return SHG_OK;
}

// X = U^-1 for an upper triangular U [n][ldu]; X [n][ldx] (the strictly lower part of X is set to zero); work: n * 128 doubles
int trtri_upper(int n, const double* U, int ldu, double* X, int ldx, double* work, hipStream_t stream) {
    if (n <= 0) return SHG_OK;
    if (int zrc = zero_fill(X, ldx, n, n, stream)) return zrc;
    // leaves: X_kk = U_kk^-1 for all diagonal blocks at once (the ragged last one separately)
    const int nfullb = n / LEAF, rest = n % LEAF;
    int rc;
    // the leaf kernel reads its block from "A" and writes the inverse to "X": copy U's diagonal blocks through X itself
    // (mode 2 reads A = U diagonal block, which it does not modify)
    if (nfullb > 0) {
        rc = launch_leaf(LEAF, const_cast<double*>(U), ldu, (long long)LEAF * (ldu + 1), X, ldx, (long long)LEAF * (ldx + 1), nfullb, 2, nullptr, 0,
                         stream);
        if (rc) return rc;
    }
    if (rest > 0) {
        const size_t o = (size_t)nfullb * LEAF;
        rc = launch_leaf(rest, const_cast<double*>(U) + o * (ldu + 1), ldu, 0, X + o * (ldx + 1), ldx, 0, 1, 2, nullptr, 0, stream);
        if (rc) return rc;
    }
    return trtri_doubling(n, U, ldu, X, ldx, work, stream);
}

// A = U^T U in place (upper triangle referenced, strictly lower triangle zeroed); work: 128 * 128 doubles; info (device int,
// may be NULL): set to the 1-based index of the first non-positive pivot
int potrf_upper(int n, double* A, int lda, double* work, int* info, hipStream_t stream) {
    int rc;
    for (int k0 = 0; k0 < n; k0 += LEAF) {
        const int kb = std::min(LEAF, n - k0);
        double* Akk = A + (size_t)k0 * lda + k0;
        const int restn = n - k0 - kb;
        rc = launch_leaf(kb, Akk, lda, 0, work, kb, 0, 1, restn > 0 ? 3 : 1, info, k0, stream);
        if (rc) return rc;
        if (restn > 0) {
            double* A12 = Akk + kb;
            // U12 = U11^-T A12 (in place: every workgroup reads only the column tile it overwrites)
            rc = gemm_ex(true, false, kb, restn, kb, 1.0, work, kb, 0, A12, lda, 0, 0.0, A12, lda, 0, 1, false, stream);
            if (rc) return rc;
            // A22 -= U12^T U12 (upper tiles)
            double* A22 = A + (size_t)(k0 + kb) * lda + (k0 + kb);
            rc = gemm_ex(true, false, restn, restn, kb, -1.0, A12, lda, 0, A12, lda, 0, 1.0, A22, lda, 0, 1, true, stream);
            if (rc) return rc;
        }
    }
    return zero_lower(n, A, lda, 0, 1, stream);
}

// A = U^T U in place and X = U^-1 in one recursive sweep (both upper triangular, strictly lower parts zeroed):
//   n <= 128: one leaf (factor + inverse);  else  A11 -> (U11, X11),  U12 = X11^T A12,  A22 -= U12^T U12,  A22 -> (U22, X22),
//   X12 = -X11 U12 X22.
// Every product has K = n / 2, n / 4, ...: fat GEMMs instead of the K = 128 panel products of the blocked right-looking sweep
// (potrf_upper + trtri_upper: 1.65 + 0.83 ms at n = 1681, 48 % of it in panel products that leave most CUs idle).
// work: (n / 2 + 128)^2 doubles.  info as in potrf_upper.
// (batch: the same operation on `batch` matrices at A + b strideA, X + b strideX, with work + b strideW and one flag each)
struct FactorBatch {
    int count;
    long long strideA, strideX, strideW;
    int info_stride;
};


static int potrf_inverse_rec(int n, double* A, int lda, double* X, int ldx, double* work, int* info, int info_base, const FactorBatch& fb,
                             hipStream_t stream) {
    if (n <= LEAF) return launch_leaf(n, A, lda, fb.strideA, X, ldx, fb.strideX, fb.count, 3, info, info_base, stream, fb.info_stride);
    int n1 = ((n / 2 + LEAF - 1) / LEAF) * LEAF;
    if (n1 >= n) n1 = n - LEAF;
    const int n2 = n - n1;
    double *A12 = A + n1, *A22 = A + (size_t)n1 * lda + n1;
    double *X12 = X + n1, *X22 = X + (size_t)n1 * ldx + n1;
    int rc = potrf_inverse_rec(n1, A, lda, X, ldx, work, info, info_base, fb, stream);
    if (rc) return rc;
    // U12 = X11^T A12: A12 is staged in the (still unused) X12 region, the product goes back into A12
    for (int b = 0; b < fb.count; ++b)
        SHG_HIP(hipMemcpy2DAsync(X12 + b * fb.strideX, (size_t)ldx * sizeof(double), A12 + b * fb.strideA, (size_t)lda * sizeof(double),
                                 (size_t)n2 * sizeof(double), n1, hipMemcpyDeviceToDevice, stream));
    rc = gemm_ex_tri(true, false, n1, n2, n1, 1.0, X, ldx, fb.strideX, X12, ldx, fb.strideX, 0.0, A12, lda, fb.strideA, fb.count, false, 2, stream);   // X11^T is lower triangular
    if (rc) return rc;
    rc = gemm_ex(true, false, n2, n2, n1, -1.0, A12, lda, fb.strideA, A12, lda, fb.strideA, 1.0, A22, lda, fb.strideA, fb.count, true, stream);
    if (rc) return rc;
    rc = potrf_inverse_rec(n2, A22, lda, X22, ldx, work, info, info_base + n1, fb, stream);
    if (rc) return rc;
    rc = gemm_ex_tri(false, false, n1, n2, n2, 1.0, A12, lda, fb.strideA, X22, ldx, fb.strideX, 0.0, work, n2, fb.strideW, fb.count, false, 4, stream);    // X22 is upper triangular
    if (rc) return rc;
    return gemm_ex_tri(false, false, n1, n2, n1, -1.0, X, ldx, fb.strideX, work, n2, fb.strideW, 0.0, X12, ldx, fb.strideX, fb.count, false, 1, stream);   // X11 is upper triangular
}

size_t potrf_inverse_work(int n) { return (size_t)(n / 2 + LEAF) * (n / 2 + LEAF); }

// 0: recursive sweep on the caller's stream; 1: panel sweep with look-ahead, chain rows carry their coupling block along when the
// two side streams have hardware queues of their own (a timing experiment, plan.hip); 2: ... carry it along in any case (on the
// stream of the trailing update if there is no second queue); 3: ... never.  Setting of the calling thread.
static thread_local int g_lookahead = 1;
void potrf_inverse_set_lookahead(int mode) { g_lookahead = mode; }
int potrf_inverse_lookahead_mode() { return g_lookahead; }

// The same factor and inverse by 128-column panels with a look-ahead of one panel.  The serial chain of the recursive sweep is
// 14 leaves AND every product between them (1.89 ms at n = 1681: 0.79 ms of leaves, 1.0 ms of products that are too small to
// fill the chip).  Here the caller's stream carries only what the next leaf waits for,
//   leaf k (U_kk, X_kk)  ->  U_k,k+1 = X_kk^T A_k,k+1  ->  A_k+1,k+1 -= U_k,k+1^T U_k,k+1  ->  leaf k + 1 ...
// and a side stream follows one panel behind with the rest of step k,
//   U_k,k+2.. = X_kk^T A_k,k+2..,   A_k+1..,k+2.. -= U_k,k+1..^T U_k,k+2..,
// and then grows the inverse by one block column (left-looking; nothing in the factorisation waits for it):
//   X_0..k-1,k = -X_0..k-1,0..k-1 (U_0..k-1,k X_kk).
// Events order the two: the side stream starts step k when leaf k and U_k,k+1 exist, the caller's stream forms U_k+1,k+2 when
// the side stream has finished the trailing update of step k, and joins it at the end.  Every block receives its updates in a
// fixed order, so the result does not depend on how the streams interleave.
// With a Coupling (a chain row: the block row continues to the right of A) a second side stream, on a hardware queue of its own,
// carries the same steps through the coupling block and the next diagonal block; it needs row panel k of U (an event of the
// first side stream) and is joined at the end of the sweep only.
// One matrix alone: 1.27 - 1.40 ms instead of 1.89 ms (n = 1681).  It does NOT pay for matrices that are factored from several
// host threads at once: the card overlaps the kernels of two to three queues, not of four or six (two threads: 2.2 ms per matrix
// each, against 2.0 ms with the recursive sweep; three: 3.5 against 2.2) -- concurrent chains are factored as a BATCH instead
// (FactorBatch: every launch serves all of them), which keeps the two queues and halves the launches per matrix.
static int potrf_inverse_lookahead(int n, double* A, int lda, double* X, int ldx, double* work, int* info, const FactorBatch& fb,
                                   const Coupling* cp, hipStream_t stream) {
    ScratchLease lease(stream);
    hipStream_t side[2];
    hipEvent_t to_side, from_side[2];
    int rc = lease.side(side, &to_side, from_side);
    if (rc) return rc;
    // (trailing update and inverse on ONE side stream: behind the event that the caller's stream waits for, the inverse delays
    //  nobody but the next trailing update)
    hipStream_t trail = side[0], inverse = side[0];
    // the columns to the right of A (Coupling) on a queue of their own if there is one
    hipStream_t couple = lease.sides_apart() >= 2 || g_lookahead == 2 ? side[1] : side[0];
    hipEvent_t row_ready = nullptr;
    if (cp && (rc = lease.event(2, &row_ready)) != SHG_OK) return rc;
    const int nb = ceil_div(n, LEAF), nbatch = fb.count;
    const long long sA = fb.strideA, sX = fb.strideX, sW = fb.strideW;
    auto at = [&](double* P, int ld, int i, int j) { return P + (size_t)i * LEAF * ld + (size_t)j * LEAF; };
    // the strictly lower parts of both results are zero: nothing below reads them or writes them (the leaves write the lower
    // parts of their own diagonal blocks, with zeros), so the inverse's stream clears them while the first leaf runs
    SHG_HIP(hipEventRecord(to_side, stream));
    SHG_HIP(hipStreamWaitEvent(trail, to_side, 0));
    rc = zero_lower(n, A, lda, sA, nbatch, inverse);
    if (!rc) rc = zero_lower(n, X, ldx, sX, nbatch, inverse);
    if (rc) return rc;
    for (int k = 0; k < nb; ++k) {
        const int k0 = k * LEAF, kb = std::min(LEAF, n - k0);
        const int k1 = k0 + kb, kb1 = std::min(LEAF, n - k1);           // next panel
        const int k2 = k1 + kb1, rest2 = n - k2;                        // the panels behind it
        double *Akk = at(A, lda, k, k), *Xkk = at(X, ldx, k, k);
        rc = launch_leaf(kb, Akk, lda, sA, Xkk, ldx, sX, nbatch, 3, info, k0, stream, fb.info_stride);
        if (rc) return rc;
        if (k > 0) SHG_HIP(hipStreamWaitEvent(stream, from_side[0], 0));             // step k - 1 of the trailing update is complete
        if (kb1 > 0) {
            double* Ak1 = at(A, lda, k, k + 1);
            rc = gemm_ex_tri(true, false, kb, kb1, kb, 1.0, Xkk, ldx, sX, Ak1, lda, sA, 0.0, Ak1, lda, sA, nbatch, false, 2, stream);     // in place: one column tile
            if (rc) return rc;
            rc = gemm_ex(true, false, kb1, kb1, kb, -1.0, Ak1, lda, sA, Ak1, lda, sA, 1.0, at(A, lda, k + 1, k + 1), lda, sA, nbatch, true, stream);
            if (rc) return rc;
        }
        SHG_HIP(hipEventRecord(to_side, stream));
        if (rest2 > 0 || cp) SHG_HIP(hipStreamWaitEvent(trail, to_side, 0));
        if (rest2 > 0) {
            double *Ak1 = at(A, lda, k, k + 1), *Ak2 = at(A, lda, k, k + 2);
            rc = gemm_ex_tri(true, false, kb, rest2, kb, 1.0, Xkk, ldx, sX, Ak2, lda, sA, 0.0, Ak2, lda, sA, nbatch, false, 2, trail);   // in place: kb <= 128 rows
            if (rc) return rc;
            if (cp && couple != trail) SHG_HIP(hipEventRecord(row_ready, trail));               // row panel k of U is complete
            rc = gemm_ex(true, false, kb1, rest2, kb, -1.0, Ak1, lda, sA, Ak2, lda, sA, 1.0, at(A, lda, k + 1, k + 2), lda, sA, nbatch, false, trail);
            if (rc) return rc;
            rc = gemm_ex(true, false, rest2, rest2, kb, -1.0, Ak2, lda, sA, Ak2, lda, sA, 1.0, at(A, lda, k + 2, k + 2), lda, sA, nbatch, true, trail);
            if (rc) return rc;
        }
        if (cp) {
            // the same three products for the columns to the right of A: rows k of B, the rows of B below, and S
            if (couple != trail) {
                SHG_HIP(hipStreamWaitEvent(couple, to_side, 0));
                if (rest2 > 0) SHG_HIP(hipStreamWaitEvent(couple, row_ready, 0));
            }
            double* Bk = cp->B + (size_t)k0 * cp->ldb;
            rc = gemm_ex_tri(true, false, kb, cp->nc, kb, 1.0, Xkk, ldx, sX, Bk, cp->ldb, cp->strideB, 0.0, Bk, cp->ldb, cp->strideB, nbatch, false, 2, couple);
            if (rc) return rc;
            if (n - k1 > 0) {
                rc = gemm_ex(true, false, n - k1, cp->nc, kb, -1.0, at(A, lda, k, k + 1), lda, sA, Bk, cp->ldb, cp->strideB, 1.0, cp->B + (size_t)k1 * cp->ldb,
                             cp->ldb, cp->strideB, nbatch, false, couple);
                if (rc) return rc;
            }
            // S -= W^T W for the rows of W that have become final, in two pieces (after half of the panels and after the last one):
            // a rank-128 update per step reads and writes all of S for 8 K-steps of work
            const int half = nb / 2;                                 // (nb >= 3 here)
            if (k == half - 1 || k == nb - 1) {
                const int r0 = k == nb - 1 ? half * LEAF : 0;        // rows r0 .. k1 - 1 of W
                double* Wr = cp->B + (size_t)r0 * cp->ldb;
                rc = gemm_ex(true, false, cp->nc, cp->nc, k1 - r0, -1.0, Wr, cp->ldb, cp->strideB, Wr, cp->ldb, cp->strideB, 1.0, cp->S, cp->lds, cp->strideS, nbatch,
                             true, couple);
                if (rc) return rc;
            }
        }
        SHG_HIP(hipEventRecord(from_side[0], trail));
        if (k > 0) {
            // (U_0..k-1,k is complete: its last block is U_k-1,k of the caller's stream, the others are older)
            SHG_HIP(hipStreamWaitEvent(inverse, to_side, 0));
            double* U0k = at(A, lda, 0, k);
            rc = gemm_ex_tri(false, false, k0, kb, kb, 1.0, U0k, lda, sA, Xkk, ldx, sX, 0.0, work, kb, sW, nbatch, false, 4, inverse);    // X_kk is upper triangular
            if (rc) return rc;
            rc = gemm_ex_tri(false, false, k0, kb, k0, -1.0, X, ldx, sX, work, kb, sW, 0.0, at(X, ldx, 0, k), ldx, sX, nbatch, false, 1, inverse);   // and so is X_0..k-1,0..k-1
            if (rc) return rc;
        }
    }
    SHG_HIP(hipStreamWaitEvent(stream, from_side[0], 0));
    if (cp && couple != trail) {
        SHG_HIP(hipEventRecord(from_side[1], couple));
        SHG_HIP(hipStreamWaitEvent(stream, from_side[1], 0));
    }
    if (cp && cp->inverse_done) {
        SHG_HIP(hipEventRecord(cp->inverse_done, inverse));          // B and S are complete; the last block column of the inverse may still be growing
        return SHG_OK;
    }
    SHG_HIP(hipEventRecord(from_side[1], inverse));
    SHG_HIP(hipStreamWaitEvent(stream, from_side[1], 0));
    return SHG_OK;
}

// whether potrf_inverse_batch takes the coupling block and the next diagonal block along (Coupling)
static bool takes_lookahead(int n) { return n > 2 * LEAF && g_lookahead; }

bool potrf_inverse_carries_coupling(int n, hipStream_t stream) {
    if (!takes_lookahead(n) || g_lookahead == 3) return false;
    // only with a hardware queue of its own for the coupling columns: on the stream of the trailing update they make it the
    // pace-maker of the sweep (2.24 against 1.87 ms per pair of d = 1681 blocks) -- unless the caller insists (mode 2)
    ScratchLease lease(stream);
    hipStream_t side[2];
    hipEvent_t to_side, from_side[2];
    return lease.side(side, &to_side, from_side) == SHG_OK && (lease.sides_apart() >= 2 || g_lookahead == 2);
}

int potrf_inverse_batch(int n, double* A, int lda, long long strideA, double* X, int ldx, long long strideX, double* work, long long strideW,
                        int* info, int info_stride, int batch, const Coupling* cp, hipStream_t stream) {
    if (n <= 0 || batch <= 0) return SHG_OK;
    const FactorBatch fb{batch, strideA, strideX, strideW, info_stride};
    if (takes_lookahead(n)) return potrf_inverse_lookahead(n, A, lda, X, ldx, work, info, fb, cp, stream);
    if (cp) return fail(SHG_ERR_INVALID, "potrf_inverse_batch: a coupling block needs the look-ahead sweep");
    for (int b = 0; b < batch; ++b)
        if (int zrc = zero_fill(X + b * strideX, ldx, n, n, stream)) return zrc;
    const int rc = potrf_inverse_rec(n, A, lda, X, ldx, work, info, 0, fb, stream);
    if (rc) return rc;
    return zero_lower(n, A, lda, strideA, batch, stream);
}

int potrf_inverse_upper(int n, double* A, int lda, double* X, int ldx, double* work, int* info, hipStream_t stream) {
    return potrf_inverse_batch(n, A, lda, 0, X, ldx, 0, work, 0, info, 0, 1, nullptr, stream);
}

}  // namespace shg

using namespace shg;

extern "C" int shg_gemm(int transa, int transb, int M, int N, int K, double alpha, const double* A, int lda, const double* B, int ldb,
                        double beta, double* C, int ldc, void* stream) {
    SHG_REQUIRE(M >= 0 && N >= 0 && K >= 0, "shg_gemm: negative dimension");
    if (M == 0 || N == 0) return SHG_OK;
    SHG_REQUIRE(C != nullptr && ldc >= N, "shg_gemm: bad output");
    if (K > 0) {
        SHG_REQUIRE(A && B, "shg_gemm: NULL pointer");
        SHG_REQUIRE(lda >= (transa ? M : K) && ldb >= (transb ? K : N), "shg_gemm: leading dimension too small");
    }
    return gemm_ex(transa != 0, transb != 0, M, N, K, alpha, A, lda, 0, B, ldb, 0, beta, C, ldc, 0, 1, false, (hipStream_t)stream);
}

namespace shg {
__global__ void axpby_kernel(int rows, int cols, double alpha, const double* __restrict__ X, int ldx, double beta, double* __restrict__ Y,
                             int ldy) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= cols) return;
    double* y = Y + (size_t)r * ldy + c;
    const double ax = alpha * X[(size_t)r * ldx + c];
    *y = beta == 0.0 ? ax : fma(beta, *y, ax);
}
}  // namespace shg

// A <- A^T for a square matrix in its own storage: 32 x 32 tiles through LDS, the tile pairs (bi, bj), bi < bj, swapped by one
// workgroup each (the coupling blocks of a chain that is walked backwards, 22.6 MB at d = 1681: one pass instead of the copy and the
// strided copy back of block.copy_(block.t().clone()))
namespace shg {
__global__ __launch_bounds__(256) void transpose_in_place_kernel(int n, double* __restrict__ A, int lda) {
    __shared__ double upper[32][33], lower[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int r = bi * 32 + k, c = bj * 32 + tx;              // tile (bi, bj)
        upper[k][tx] = (r < n && c < n) ? A[(size_t)r * lda + c] : 0.0;
        const int r2 = bj * 32 + k, c2 = bi * 32 + tx;            // tile (bj, bi)
        lower[k][tx] = (r2 < n && c2 < n) ? A[(size_t)r2 * lda + c2] : 0.0;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int r = bi * 32 + k, c = bj * 32 + tx;
        if (r < n && c < n) A[(size_t)r * lda + c] = lower[tx][k];
        const int r2 = bj * 32 + k, c2 = bi * 32 + tx;
        if (bi != bj && r2 < n && c2 < n) A[(size_t)r2 * lda + c2] = upper[tx][k];
    }
}
}  // namespace shg

extern "C" int shg_transpose_in_place(int n, double* A, int lda, void* stream) {
    SHG_REQUIRE(n >= 0, "shg_transpose_in_place: negative size");
    if (n <= 1) return SHG_OK;
    SHG_REQUIRE(A != nullptr && lda >= n, "shg_transpose_in_place: bad matrix");
    hipLaunchKernelGGL(shg::transpose_in_place_kernel, dim3(shg::ceil_div(n, 32), shg::ceil_div(n, 32)), dim3(256), 0, (hipStream_t)stream, n, A, lda);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

extern "C" int shg_axpby(int rows, int cols, double alpha, const double* X, int ldx, double beta, double* Y, int ldy, void* stream) {
    SHG_REQUIRE(rows >= 0 && cols >= 0, "shg_axpby: negative size");
    if (rows == 0 || cols == 0) return SHG_OK;
    SHG_REQUIRE(X && Y && ldx >= cols && ldy >= cols, "shg_axpby: bad operands");
    SHG_REQUIRE(rows <= 65535 * 1024, "shg_axpby: too many rows");
    for (int r0 = 0; r0 < rows; r0 += 65535) {
        const int nr = std::min(65535, rows - r0);
        hipLaunchKernelGGL(axpby_kernel, dim3(ceil_div(cols, 256), nr), dim3(256), 0, (hipStream_t)stream, nr, cols, alpha,
                           X + (size_t)r0 * ldx, ldx, beta, Y + (size_t)r0 * ldy, ldy);
    }
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

extern "C" int shg_potrf(int n, double* A, int lda, int* info, void* stream_) {
    SHG_REQUIRE(n >= 0, "shg_potrf: negative size");
    if (n == 0) return SHG_OK;
    SHG_REQUIRE(A != nullptr && lda >= n, "shg_potrf: bad matrix");
    hipStream_t stream = (hipStream_t)stream_;
    double* work = nullptr;
    if (workspace_alloc((void**)&work, (size_t)LEAF * LEAF * sizeof(double), stream) != hipSuccess)
        return fail(SHG_ERR_NOMEM, "shg_potrf: workspace allocation failed");
    if (info && (zero_fill(info, stream) != SHG_OK)) return SHG_ERR_HIP;
    const int rc = potrf_upper(n, A, lda, work, info, stream);
    (void)hipFreeAsync(work, stream);
    return rc;
}

extern "C" int shg_trtri(int n, const double* U, int ldu, double* X, int ldx, void* stream_) {
    SHG_REQUIRE(n >= 0, "shg_trtri: negative size");
    if (n == 0) return SHG_OK;
    SHG_REQUIRE(U && X && ldu >= n && ldx >= n, "shg_trtri: bad matrix");
    SHG_REQUIRE(U != X, "shg_trtri: in-place inversion is not supported");
    hipStream_t stream = (hipStream_t)stream_;
    double* work = nullptr;
    if (workspace_alloc((void**)&work, (size_t)n * n / 2 * sizeof(double) + 1024, stream) != hipSuccess)
        return fail(SHG_ERR_NOMEM, "shg_trtri: workspace allocation failed");
    const int rc = trtri_upper(n, U, ldu, X, ldx, work, stream);
    (void)hipFreeAsync(work, stream);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// Congruence transform  C = W S W^T  (filtered covariance matrix: W a filter matrix, S the coefficient covariance;
// users of the reference write `W @ S @ W.T` with SpatialFilter.matrix(), grates/filter.py:74-95, 193-222, 481-509, ahead of
// RegularGrid.covariance_propagation, grates/grid.py:792-839).  Two fp64 MFMA GEMMs; the second one forms the upper tiles only
// and the strictly lower triangle is mirrored, so a symmetric S gives an exactly symmetric C.
// ------------------------------------------------------------------------------------------------
namespace shg {
__global__ __launch_bounds__(256) void mirror_upper_kernel(int n, double* __restrict__ C, int ldc) {
    __shared__ double tile[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x;                 // tile (bi, bj) of the upper triangle, bj >= bi
    if (bj < bi) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int r = bi * 32 + k, c = bj * 32 + tx;
        tile[k][tx] = (r < n && c < n) ? C[(size_t)r * ldc + c] : 0.0;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int r = bj * 32 + k, c = bi * 32 + tx;            // mirrored position
        if (r < n && c < n && r > c) C[(size_t)r * ldc + c] = tile[tx][k];
    }
}
}  // namespace shg

extern "C" int shg_congruence(int n, int k, const double* W, int ldw, const double* S, int lds, double* C, int ldc, double* work, void* stream_) {
    SHG_REQUIRE(n >= 0 && k >= 0, "shg_congruence: negative dimension");
    if (n == 0) return SHG_OK;
    SHG_REQUIRE(W && S && C && work, "shg_congruence: NULL pointer");
    SHG_REQUIRE(ldw >= k && lds >= k && ldc >= n, "shg_congruence: leading dimension too small");
    hipStream_t stream = (hipStream_t)stream_;
    // work [n][k] = W S;  C = work W^T (upper tiles), then the mirror image
    int rc = shg::gemm_ex(false, false, n, k, k, 1.0, W, ldw, 0, S, lds, 0, 0.0, work, k, 0, 1, false, stream);
    if (rc) return rc;
    rc = shg::gemm_ex(false, true, n, n, k, 1.0, work, k, 0, W, ldw, 0, 0.0, C, ldc, 0, 1, true, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(shg::mirror_upper_kernel, dim3(shg::ceil_div(n, 32), shg::ceil_div(n, 32)), dim3(256), 0, stream, n, C, ldc);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}
