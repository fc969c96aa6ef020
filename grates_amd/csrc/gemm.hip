// fp64 MFMA GEMM (v_mfma_f64_16x16x4_f64) in two flavours that share one main loop:
//   MODE_PLAIN    C[M][N] = A[M][K] B[K][N]                      shg_dgemm, shg_dense_filter (grates/filter.py:473-474)
//   MODE_COVPROP  sigma2[r] = sum_c (A Sigma)[r][c] A[r][c]        shg_covprop_diag            (grates/grid.py:833-835)
//                 with the rows of A = synthesis matrix generated on the fly:
//                 A[(i, j)][p] = PK[i][p] * CS[slot(p)][j]         (grates/grid.py:825-834: F = cs * Pnm[k])
//                 A (68 GB at d/o 180 / 0.5 deg) is never materialised, only M doubles leave the kernel.
//   MODE_SYNTH    C[M][N] = A[M][K] X[K][N] with the same generated A     point-list synthesis of many epochs (points.hip)
//
// Block tile 128 x 128, BK = 16, 4 waves as 2 x 2, wave tile 64 x 64 (16 accumulators).  Global -> register
// prefetch of the next K tile overlaps the 64 MFMAs of the current one; one barrier per K tile.
#include "common.h"

namespace shg {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDA = 18;     // As[row][k], 18-double rows: the 16 rows x 2 k a half wave reads in one LDS cycle fall into 32 distinct bank pairs
constexpr int LDB = 144;    // Bs[k][col], 128 + 16 pad

enum { MODE_PLAIN = 0, MODE_COVPROP = 1, MODE_SYNTH = 2 };      // SYNTH: A generated like COVPROP, C stored like PLAIN

struct GemmParams {
    int M, N, K;
    const double* A;
    int lda;
    const double* B;
    int ldb;
    double* C;
    int ldc;
    // covariance propagation
    const double* pkd;      // [nlat][ldp]  kn-scaled Legendre functions in degree-wise order (min_degree 0)
    int ldp;
    const double* csr;      // [2N+1][nlon] 1, cos(lon), sin(lon), cos(2 lon), ... : row = rank of the coefficient inside its degree
    const int* rslot;       // [ldp] rank inside the degree of every degree-wise index
    int ldcs;               // leading dimension of csr (nlon, or the point count for point lists)
    long long idiv, jmod;   // flat row R -> table row R / idiv, table column R % jmod (regular grid: both nlon)
    int p_off;              // min_degree^2: first degree-wise index covered by the covariance matrix
    long long row0;         // first flat grid row (lat0 * nlon) of the band
    double* partial;        // [gridDim.x][M] per-column-block partial row sums
    int pkt;                // generated A: Legendre table stored transposed, pkd[p][row] (point lists)
};

// SYM (covariance mode only): Sigma is symmetric and only its upper triangle is used,
//   sigma2[r] = sum_c [ sum_{p<c} 2 A[r][p] Sigma[p][c] + A[r][c] Sigma[c][c] ] A[r][c]:
// column block n0 runs its K loop over p < n0 + 128 only (half the MFMAs on average); the accumulators are doubled once when
// the loop reaches the diagonal block, inside which the Sigma tile is weighted 2 / 1 / 0 (p < c / p = c / p > c) while it is
// staged.  Column blocks are taken from the last (longest) to the first.
// PKT (generated A only): the Legendre table is stored transposed, pkd[p][row] with ldp = number of rows -- the layout of
// point lists, where every row of A has its own table row and consecutive lanes (rows) then read consecutive addresses.
template <int MODE, bool VEC, bool SYM = false, bool PKT = false>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmParams P) {      // 2 waves per SIMD: two blocks per CU
    extern __shared__ double gemm_lds[];
    double (*As)[BM * LDA] = reinterpret_cast<double (*)[BM * LDA]>(gemm_lds);                       // [2][BM * LDA]
    double (*Bs)[BK * LDB] = reinterpret_cast<double (*)[BK * LDB]>(gemm_lds + 2 * BM * LDA);        // [2][BK * LDB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fk = lane >> 4;
    // wave-uniform half of the K tile a thread generates in COVPROP mode: keeps the degree / rank bookkeeping on the scalar unit
    const int khalf = __builtin_amdgcn_readfirstlane(tid >> 7);
    // PLAIN: column block fastest.  COVPROP: row block fastest -- the blocks resident at one time then walk the same
    // 33 MB column panel of Sigma together and it is fetched from HBM once instead of once per row block.
    const int m0 = (MODE == MODE_PLAIN ? blockIdx.y : blockIdx.x) * BM;
    const int colblock = MODE == MODE_PLAIN ? (int)blockIdx.x : (SYM ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y);
    const int n0 = colblock * BN;
    const int Keff = SYM ? min(P.K, n0 + BN) : P.K;                // SYM: rows of Sigma up to the end of the diagonal block

    // ---- operand fetch.  Full K tiles are fetched without any bounds test: row / column indices beyond the matrix are
    //      clamped to valid addresses (the garbage only reaches rows / columns that are never stored), so the main loop
    //      is branch free and the compiler can keep the prefetch in flight across the MFMAs.  Only the last partial
    //      K tile zero-fills.
    // PLAIN A: 128 rows x 8 pieces of 2 doubles; piece h of a thread: row = (tid >> 3) + 32 h, k = (tid & 7) * 2
    // COVPROP A: element h of a thread: row = tid & 127 (fixed), k = (tid >> 7) + 2 h, generated from two small tables
    // B: 16 k-rows x 64 pieces of 2 doubles; piece h: k = (tid >> 6) + 4 h, col = (tid & 63) * 2
    // Addressing: fp64 MFMAs and the VALU instructions of all waves of a SIMD share one issue pipe (tools/mfma64_issue.hip),
    // so the K loop keeps address arithmetic on the scalar unit: every load is "uniform 64-bit base (advanced per K tile by
    // scalar instructions) + per-lane 32-bit byte offset (fixed for the whole kernel)".
    double areg[8];
    double creg[8];             // COVPROP: cos/sin factors; the product with areg is formed when the tile is staged, i.e.
                                // after the MFMAs of the current tile, so that the loads stay in flight across them
    double2 breg[4];
    const int a_kk = (tid & 7) * 2;
    const int b_k = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform first k row of the B pieces
    const int b_col = (tid & 63) * 2;
    // (the empty asm pins the uniform base in scalar registers -- without it the compiler folds base + lane offset into one
    //  loop-invariant 64-bit vector address and adds the scalar part with a VALU instruction per load; the address goes
    //  through an integer, so the pointer is rebuilt in the global address space explicitly.  The 32 -> 64 bit extension of
    //  the lane offset has to stay in the basic block of the load for the scalar-base form to be selected: `pin` below.)
    auto pin = [](unsigned v) {
        asm volatile("" : "+v"(v));
        return v;
    };
    typedef const double __attribute__((address_space(1))) gdouble_t;
    typedef const char __attribute__((address_space(1))) gbyte_t;
    typedef int __attribute__((address_space(4))) crank_t;
    typedef double gdouble2_v __attribute__((ext_vector_type(2)));
    typedef const gdouble2_v __attribute__((address_space(1))) gdouble2_t;
    auto at = [](const double* base, unsigned byte_off) {
        unsigned long long b = reinterpret_cast<unsigned long long>(base);
        asm volatile("" : "+s"(b));
        return *reinterpret_cast<gdouble_t*>(reinterpret_cast<gbyte_t*>(b) + byte_off);
    };
    auto at2 = [](const double* base, unsigned byte_off) {
        unsigned long long b = reinterpret_cast<unsigned long long>(base);
        asm volatile("" : "+s"(b));
        const gdouble2_v v = *reinterpret_cast<gdouble2_t*>(reinterpret_cast<gbyte_t*>(b) + byte_off);
        return make_double2(v.x, v.y);
    };
    // PLAIN A: per-lane offsets of the four row pieces relative to row m0 (clamped rows stay inside the matrix)
    unsigned a_voff[4] = {0, 0, 0, 0};
    const double* a_base = nullptr;                                  // uniform: &A[m0][0]
    if (MODE == MODE_PLAIN) {
        a_base = P.A + (size_t)m0 * P.lda;
#pragma unroll
        for (int h = 0; h < 4; ++h) a_voff[h] = (unsigned)(((size_t)(min(m0 + (tid >> 3) + 32 * h, P.M - 1) - m0) * P.lda + a_kk) * 8);
    }
    // COVPROP A: per-lane offsets into the two tables
    unsigned pk_voff = 0, cs_voff = 0;
    const double* pk_base = nullptr;                                 // uniform: table row of the block's first grid row
    if (MODE != MODE_PLAIN) {
        const long long R = P.row0 + min(m0 + (tid & 127), P.M - 1);
        // (64-bit division runs on the vector unit even for uniform operands: readfirstlane brings the quotient back)
        const int gi0 = __builtin_amdgcn_readfirstlane((int)((P.row0 + m0) / P.idiv));
        if (PKT) {
            pk_base = P.pkd + (size_t)P.p_off * P.ldp + (P.row0 + m0);       // + k * ldp
            pk_voff = (unsigned)((R - (P.row0 + m0)) * 8);
        } else {
            pk_base = P.pkd + (size_t)gi0 * P.ldp + P.p_off;
            pk_voff = (unsigned)((R / P.idiv - gi0) * P.ldp * 8);    // at most 127 table rows
        }
        cs_voff = (unsigned)((R % P.jmod) * 8);
    }
    const unsigned b_voff0 = (unsigned)((VEC ? min(n0 + b_col, P.N - 2) : min(n0 + b_col, P.N - 1)) * 8);
    const unsigned b_voff1 = (unsigned)(min(n0 + b_col + 1, P.N - 1) * 8);

    // COVPROP: ranks (inside their degree) of the eight degree-wise indices a thread generates in the next K tile, read from
    // the rank table with scalar loads (constant address space: uniform loads from it are scalar loads)
    int rank[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto load_ranks = [&](int k0) {
        const crank_t* rs = reinterpret_cast<const crank_t*>(reinterpret_cast<unsigned long long>(P.rslot));
#pragma unroll
        for (int h = 0; h < 8; ++h) rank[h] = rs[min(k0 + khalf + 2 * h, P.K - 1) + P.p_off];      // (the table ends at p_off + K)
    };
    if (MODE != MODE_PLAIN) load_ranks(0);
    auto fetch_full = [&](int k0) {
        if (MODE == MODE_PLAIN) {
            const double* ak = a_base + k0;                           // uniform
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const unsigned off = pin(a_voff[h]);
                if (VEC) {
                    const double2 t = at2(ak, off);
                    areg[2 * h] = t.x;
                    areg[2 * h + 1] = t.y;
                } else {
                    areg[2 * h] = at(ak, off);
                    areg[2 * h + 1] = at(ak + 1, off);
                }
            }
        } else {
            // degree-wise index p = n^2 + r: the rank r inside the degree selects the cos/sin row (table lookup on the scalar unit)
            const double* pk = pk_base + k0 + khalf;                  // uniform
            const unsigned poff = pin(pk_voff), coff = pin(cs_voff);
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                areg[h] = PKT ? at(pk_base + (size_t)(k0 + khalf + 2 * h) * P.ldp, poff) : at(pk + 2 * h, poff);
                creg[h] = at(P.csr + (size_t)rank[h] * P.ldcs, coff);
            }
            load_ranks(k0 + BK);                                      // for the next K tile: a whole tile ahead of their use
        }
        const unsigned boff0 = pin(b_voff0), boff1 = pin(b_voff1);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const double* brow = P.B + (size_t)(k0 + b_k + 4 * h) * P.ldb;      // uniform
            if (VEC)
                breg[h] = at2(brow, boff0);
            else
                breg[h] = make_double2(at(brow, boff0), at(brow, boff1));
        }
    };
    // last partial K tile: same addresses with k clamped, entries beyond K zeroed
    auto fetch_tail = [&](int k0) {
        if (MODE == MODE_PLAIN) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int k1 = k0 + a_kk;                               // a_voff already points at column a_kk
                const double* arow = reinterpret_cast<const double*>(reinterpret_cast<const char*>(a_base) + a_voff[h]) - a_kk;     // per lane
                const double v0 = arow[min(k1, P.K - 1)];
                const double v1 = arow[min(k1 + 1, P.K - 1)];
                areg[2 * h] = k1 < P.K ? v0 : 0.0;
                areg[2 * h + 1] = k1 + 1 < P.K ? v1 : 0.0;
            }
        } else {
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const int gk = k0 + 2 * h + khalf;
                const int kc = min(gk, P.K - 1);
                areg[h] = gk < P.K ? (PKT ? at(pk_base + (size_t)kc * P.ldp, pk_voff) : at(pk_base + kc, pk_voff)) : 0.0;
                creg[h] = at(P.csr + (size_t)reinterpret_cast<const crank_t*>(reinterpret_cast<unsigned long long>(P.rslot))[kc + P.p_off] * P.ldcs, cs_voff);
            }
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int gk = k0 + b_k + 4 * h;
            const double* brow = P.B + (size_t)min(gk, P.K - 1) * P.ldb;
            const double x = at(brow, b_voff0);
            const double y = VEC ? at(brow + 1, b_voff0) : at(brow, b_voff1);
            breg[h] = gk < P.K ? make_double2(x, y) : make_double2(0.0, 0.0);
        }
    };
    auto stage = [&](int buf) {
        if (MODE == MODE_PLAIN) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int row = (tid >> 3) + 32 * h;
                As[buf][row * LDA + a_kk] = areg[2 * h];
                As[buf][row * LDA + a_kk + 1] = areg[2 * h + 1];
            }
        } else {
            const int row = tid & 127, kb = khalf;
#pragma unroll
            for (int h = 0; h < 8; ++h) As[buf][row * LDA + 2 * h + kb] = areg[h] * creg[h];
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) *reinterpret_cast<double2*>(&Bs[buf][(b_k + 4 * h) * LDB + b_col]) = breg[h];
    };
    // SYM: weights of the Sigma tile rows k0 + b_k + 4 h inside the diagonal block (applied to breg before staging)
    auto weight_diagonal = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int p = k0 + b_k + 4 * h, c = n0 + b_col;
            breg[h].x *= p < c ? 2.0 : (p == c ? 1.0 : 0.0);
            breg[h].y *= p < c + 1 ? 2.0 : (p == c + 1 ? 1.0 : 0.0);
        }
    };

    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};

    // fragments of k-step ks + 1 are read from LDS while the 16 MFMAs of k-step ks run (two named fragment sets)
    auto compute = [&](int buf) {
        const double* Ab = As[buf] + (wr * 64 + fr) * LDA + fk;
        const double* Bb = Bs[buf] + fk * LDB + wc * 64 + fr;
        double af0[4], bf0[4], af1[4], bf1[4];
#define SHG_FRAGS(af, bf, ks)                                                              \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) af[a] = Ab[a * 16 * LDA + (ks) * 4];     \
    _Pragma("unroll") for (int b = 0; b < 4; ++b) bf[b] = Bb[(ks) * 4 * LDB + b * 16]
#define SHG_MFMA16(af, bf)                                                                 \
    _Pragma("unroll") for (int a = 0; a < 4; ++a)                                         \
        _Pragma("unroll") for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0)
        SHG_FRAGS(af0, bf0, 0);
        __builtin_amdgcn_sched_barrier(0);
        SHG_FRAGS(af1, bf1, 1);
        SHG_MFMA16(af0, bf0);
        __builtin_amdgcn_sched_barrier(0);
        SHG_FRAGS(af0, bf0, 2);
        SHG_MFMA16(af1, bf1);
        __builtin_amdgcn_sched_barrier(0);
        SHG_FRAGS(af1, bf1, 3);
        SHG_MFMA16(af0, bf0);
        __builtin_amdgcn_sched_barrier(0);
        SHG_MFMA16(af1, bf1);
#undef SHG_FRAGS
#undef SHG_MFMA16
    };

    const int nfull = Keff / BK;
    const bool has_tail = (Keff % BK) != 0;
    const int tdiag = n0 / BK;                         // SYM: first K tile of the diagonal block
    if (nfull > 0)
        fetch_full(0);
    else
        fetch_tail(0);
    if (SYM && tdiag == 0) weight_diagonal(0);
    stage(0);
    __syncthreads();
    // branch-free steady state, two K tiles per trip so that the LDS buffer of every access is a literal
    auto step = [&](int t, int buf) {
        fetch_full((t + 1) * BK);
        if (SYM && t == tdiag && t > 0) {              // all rows above the diagonal block are accumulated: they count twice
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] *= 2.0;
        }
        compute(buf);
        if (SYM && t + 1 >= tdiag) weight_diagonal((t + 1) * BK);
        stage(buf ^ 1);
        __syncthreads();
    };
    int t = 0;
    for (; t + 2 < nfull; t += 2) {
        step(t, 0);
        step(t + 1, 1);
    }
    if (t + 1 < nfull) step(t, 0);                     // t is even here
    if (nfull > 0) {
        if (has_tail) fetch_tail(nfull * BK);
        if (SYM && nfull - 1 == tdiag && tdiag > 0) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] *= 2.0;
        }
        compute((nfull - 1) & 1);
        if (has_tail) {
            if (SYM) weight_diagonal(nfull * BK);
            stage(nfull & 1);
        }
        __syncthreads();
    }
    if (has_tail) {
        if (SYM && nfull == tdiag && tdiag > 0) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] *= 2.0;
        }
        compute(nfull & 1);
    }
    __syncthreads();

    // ---- epilogue.  C/D layout: column = lane & 15, row = (lane >> 4) + 4 * reg
    if (MODE != MODE_COVPROP) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = m0 + wr * 64 + a * 16 + fk + 4 * r;
                if (gr >= P.M) continue;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int gc = n0 + wc * 64 + b * 16 + fr;
                    if (gc < P.N) P.C[(size_t)gr * P.ldc + gc] = acc[a][b][r];
                }
            }
    } else {
        // row-dot of the (A Sigma) tile with the matching A tile, reduced over the 128 columns of the block
        double* red = As[0];                // reuse: [128 rows][2 column halves]
        int colslot[4];
        bool colok[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int gc = n0 + wc * 64 + b * 16 + fr;
            colok[b] = gc < P.N;
            colslot[b] = P.rslot[(colok[b] ? gc : 0) + P.p_off];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr * 64 + a * 16 + fk + 4 * r;
                const bool rok = m0 + row < P.M;
                const long long R = P.row0 + m0 + (rok ? row : 0);
                const long long gi = R / P.idiv;
                const int gj = (int)(R % P.jmod);
                double s = 0.0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (rok && colok[b]) {
                        const int pf = n0 + wc * 64 + b * 16 + fr + P.p_off;
                        const double pkv = PKT ? P.pkd[(size_t)pf * P.ldp + R] : P.pkd[gi * P.ldp + pf];
                        const double aval = pkv * P.csr[(size_t)colslot[b] * P.ldcs + gj];
                        s = fma(acc[a][b][r], aval, s);
                    }
                }
                // reduce over the 16 lanes that hold the 16 columns of a tile row
                s += __shfl_xor(s, 1);
                s += __shfl_xor(s, 2);
                s += __shfl_xor(s, 4);
                s += __shfl_xor(s, 8);
                if (fr == 0) red[row * 2 + wc] = s;
            }
        __syncthreads();
        if (tid < BM && m0 + tid < P.M) P.partial[(size_t)colblock * P.M + m0 + tid] = red[tid * 2] + red[tid * 2 + 1];
    }
}

__global__ void covprop_reduce_kernel(int M, int nparts, const double* __restrict__ partial, double* __restrict__ sigma) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    double s = 0.0;
    for (int c = 0; c < nparts; ++c) s += partial[(size_t)c * M + r];
    sigma[r] = sqrt(s);                                     // grates/grid.py:837-839
}

// PKD[i][p] = PK[(m, n)][i] rearranged to the degree-wise index p (min_degree 0); rslot[p] = rank inside the degree
__global__ void covprop_pkd_kernel(int N, int nlat, int ldlat, const double* __restrict__ pk, double* __restrict__ pkd,
                                   int* __restrict__ rslot) {
    const int P = (N + 1) * (N + 1);
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long long)nlat * P) return;
    const int i = (int)(tid / P), p = (int)(tid % P);
    int n = (int)sqrt((double)p);
    while ((n + 1) * (n + 1) <= p) ++n;
    while (n * n > p) --n;
    const int r = p - n * n;
    const int m = (r + 1) >> 1;
    pkd[tid] = pk[(size_t)(order_offset(N, m) + n - m) * ldlat + i];
    if (i == 0) rslot[p] = r;
}

// CSR[r][j]: r = 0 -> 1, r = 2m-1 -> cos(m lon_j), r = 2m -> sin(m lon_j)
__global__ void cs_table_kernel(int N, int nlon, const double* __restrict__ lon, double* __restrict__ csr) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long long)(2 * N + 1) * nlon) return;
    const int r = (int)(tid / nlon), j = (int)(tid % nlon);
    const int m = (r + 1) >> 1;
    const double arg = (double)m * lon[j];
    csr[tid] = (r == 0) ? 1.0 : ((r & 1) ? cos(arg) : sin(arg));
}

int covprop_build_cs_table(shg_plan* p, hipStream_t stream) {
    if (p->cs_slot) return SHG_OK;
    const long long n = (long long)(2 * p->N + 1) * p->nlon;
    if (hipMalloc((void**)&p->cs_slot, (size_t)n * sizeof(double)) != hipSuccess) return fail(SHG_ERR_NOMEM, "cos/sin table allocation failed");
    hipLaunchKernelGGL(cs_table_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, stream, p->N, p->nlon, p->lon, p->cs_slot);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

static int launch_gemm(int mode, const GemmParams& P, hipStream_t stream, bool symmetric = false) {
    const dim3 grid = mode == MODE_PLAIN ? dim3(ceil_div(P.N, BN), ceil_div(P.M, BM)) : dim3(ceil_div(P.M, BM), ceil_div(P.N, BN));
    const size_t lds = (size_t)(2 * BM * LDA + 2 * BK * LDB) * sizeof(double);      // 71.7 KB: two blocks per CU
    // 16-byte operand loads need even leading dimensions / sizes and 16-byte aligned bases
    const bool vec = (P.ldb % 2 == 0) && (P.N % 2 == 0) && ((uintptr_t)P.B % 16 == 0) &&
                     (mode != MODE_PLAIN || ((P.lda % 2 == 0) && (P.K % 2 == 0) && ((uintptr_t)P.A % 16 == 0)));
#define SHG_GEMM_LAUNCH(M_, V_)                                                                                                \
    do {                                                                                                                        \
        SHG_HIP(hipFuncSetAttribute((const void*)gemm_f64_kernel<M_, V_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((gemm_f64_kernel<M_, V_>), grid, dim3(256), lds, stream, P);                                         \
    } while (0)
    if (mode == MODE_PLAIN) {
        if (vec) SHG_GEMM_LAUNCH(MODE_PLAIN, true); else SHG_GEMM_LAUNCH(MODE_PLAIN, false);
    } else if (mode == MODE_SYNTH) {
        // (point lists: transposed Legendre table)
        if (vec) {
            SHG_HIP(hipFuncSetAttribute((const void*)gemm_f64_kernel<MODE_SYNTH, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((gemm_f64_kernel<MODE_SYNTH, true, false, true>), grid, dim3(256), lds, stream, P);
        } else {
            SHG_HIP(hipFuncSetAttribute((const void*)gemm_f64_kernel<MODE_SYNTH, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((gemm_f64_kernel<MODE_SYNTH, false, false, true>), grid, dim3(256), lds, stream, P);
        }
    } else {
        if (symmetric) {
            SHG_HIP(hipFuncSetAttribute((const void*)gemm_f64_kernel<MODE_COVPROP, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((gemm_f64_kernel<MODE_COVPROP, false, true>), grid, dim3(256), lds, stream, P);
        } else if (P.pkt) {
            SHG_HIP(hipFuncSetAttribute((const void*)gemm_f64_kernel<MODE_COVPROP, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((gemm_f64_kernel<MODE_COVPROP, false, false, true>), grid, dim3(256), lds, stream, P);
        } else if (vec) {
            SHG_GEMM_LAUNCH(MODE_COVPROP, true);
        } else {
            SHG_GEMM_LAUNCH(MODE_COVPROP, false);
        }
    }
#undef SHG_GEMM_LAUNCH
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

// C[M][N] = A X for M rows whose A entries are products of two table entries; point lists: pkdT [P][M] (transposed Legendre
// table, ldp = M), csr [2N+1][M]
int synth_generic(const double* pkd, int ldp, const double* csr, int ldcs, const int* rslot, long long idiv, long long jmod, int M,
                  const double* X, int K, int N, double* C, hipStream_t stream) {
    GemmParams G = {};
    G.M = M;
    G.N = N;
    G.K = K;
    G.B = X;
    G.ldb = N;
    G.C = C;
    G.ldc = N;
    G.pkd = pkd;
    G.ldp = ldp;
    G.csr = csr;
    G.ldcs = ldcs;
    G.rslot = rslot;
    G.idiv = idiv;
    G.jmod = jmod;
    G.p_off = 0;
    G.row0 = 0;
    G.pkt = 1;
    return launch_gemm(MODE_SYNTH, G, stream);
}

// sigma[r] = sqrt(a_r^T Sigma a_r) for M rows whose A entries are products of two table entries
int covprop_generic(const double* pkd, int ldp, const double* csr, int ldcs, const int* rslot, long long idiv, long long jmod,
                    long long row0, int M, const double* cov, int Pn, int p_off, double* partial, double* sigma, shg_plan* prof,
                    hipStream_t stream, bool symmetric, bool transposed_table) {
    GemmParams G = {};
    G.pkt = transposed_table ? 1 : 0;
    G.M = M;
    G.N = Pn;
    G.K = Pn;
    G.B = cov;
    G.ldb = Pn;
    G.pkd = pkd;
    G.ldp = ldp;
    G.csr = csr;
    G.ldcs = ldcs;
    G.rslot = rslot;
    G.idiv = idiv;
    G.jmod = jmod;
    G.p_off = p_off;
    G.row0 = row0;
    G.partial = partial;
    {
        ProfileScope ps(prof, 3, stream);
        int rc = launch_gemm(MODE_COVPROP, G, stream, symmetric);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(covprop_reduce_kernel, dim3(ceil_div(M, 256)), dim3(256), 0, stream, M, ceil_div(Pn, BN), partial, sigma);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

}  // namespace shg

namespace shg {
int gemm_ex(bool ta, bool tb, int M, int N, int K, double alpha, const double* A, int lda, long long strideA, const double* B, int ldb,
            long long strideB, double beta, double* C, int ldc, long long strideC, int batch, bool upper_only, hipStream_t stream);   // blas.hip
int covprop_rows(shg_plan* p, const double* cov, int Pn, int p_off, int lat0, int lat1, double* partial, hipStream_t stream);
}

using namespace shg;

extern "C" int shg_dgemm(int M, int N, int K, const double* A, int lda, const double* B, int ldb, double* C, int ldc, void* stream_) {
    SHG_REQUIRE(M >= 0 && N >= 0 && K >= 0, "shg_dgemm: negative dimension");
    if (M == 0 || N == 0) return SHG_OK;
    SHG_REQUIRE(C != nullptr, "shg_dgemm: NULL output pointer");
    SHG_REQUIRE(ldc >= N, "shg_dgemm: leading dimension too small");
    if (K == 0) {                                     // empty sum: C = 0
        SHG_HIP(hipMemset2DAsync(C, (size_t)ldc * sizeof(double), 0, (size_t)N * sizeof(double), M, (hipStream_t)stream_));
        return SHG_OK;
    }
    SHG_REQUIRE(A && B, "shg_dgemm: NULL pointer");
    SHG_REQUIRE(lda >= K && ldb >= N, "shg_dgemm: leading dimension too small");
    // too few output tiles to fill the chip (e.g. the dense filter: 14637 x 240): the general kernel splits K
    if (ceil_div(M, BM) * ceil_div(N, BN) < 384 && K >= 512)
        return gemm_ex(false, false, M, N, K, 1.0, A, lda, 0, B, ldb, 0, 0.0, C, ldc, 0, 1, false, (hipStream_t)stream_);
    GemmParams P = {};
    P.M = M;
    P.N = N;
    P.K = K;
    P.A = A;
    P.lda = lda;
    P.B = B;
    P.ldb = ldb;
    P.C = C;
    P.ldc = ldc;
    return launch_gemm(MODE_PLAIN, P, (hipStream_t)stream_);
}

extern "C" int shg_dense_filter(const double* W, int Pn, const double* X, int T, double* Y, void* stream_) {
    SHG_REQUIRE(Pn >= 0 && T >= 0, "shg_dense_filter: negative dimension");
    return shg_dgemm(Pn, T, Pn, W, Pn, X, T, Y, T, stream_);
}

static int covprop_diag_impl(shg_plan* p, const double* cov, int nmin, int lat0, int lat1, double* sigma, void* stream_, bool symmetric);

extern "C" int shg_covprop_diag(shg_plan* p, const double* cov, int nmin, int lat0, int lat1, double* sigma, void* stream_) {
    return covprop_diag_impl(p, cov, nmin, lat0, lat1, sigma, stream_, false);
}

extern "C" int shg_covprop_diag_symmetric(shg_plan* p, const double* cov, int nmin, int lat0, int lat1, double* sigma, void* stream_) {
    return covprop_diag_impl(p, cov, nmin, lat0, lat1, sigma, stream_, true);
}

namespace shg {
// max |S[p][c] - S[c][p]| over 32 x 32 tile pairs (upper tiles compared with their mirror images through LDS)
__global__ __launch_bounds__(256) void symmetry_defect_kernel(int n, const double* __restrict__ S, int ld, double* __restrict__ out) {
    __shared__ double tile[32][33];
    __shared__ double red[256];
    const int bi = blockIdx.y, bj = blockIdx.x;
    double worst = 0.0;
    if (bj >= bi) {
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        for (int k = ty; k < 32; k += 8) {
            const int r = bj * 32 + k, c = bi * 32 + tx;                 // mirror tile (bj, bi), row r, column c
            tile[k][tx] = (r < n && c < n) ? S[(size_t)r * ld + c] : 0.0;
        }
        __syncthreads();
        for (int k = ty; k < 32; k += 8) {
            const int r = bi * 32 + k, c = bj * 32 + tx;
            if (r < n && c < n) worst = fmax(worst, fabs(S[(size_t)r * ld + c] - tile[tx][k]));
        }
    }
    red[threadIdx.x] = worst;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + w]);
        __syncthreads();
    }
    if (threadIdx.x == 0 && red[0] > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(red[0]));
}
}  // namespace shg

extern "C" int shg_symmetry_defect(const double* S, int n, int ld, double* defect, void* stream_) {
    SHG_REQUIRE(n >= 0 && ld >= n, "shg_symmetry_defect: bad size");
    SHG_REQUIRE(defect != nullptr, "shg_symmetry_defect: NULL output");
    hipStream_t stream = (hipStream_t)stream_;
    SHG_HIP(hipMemsetAsync(defect, 0, sizeof(double), stream));
    if (n == 0) return SHG_OK;
    SHG_REQUIRE(S != nullptr, "shg_symmetry_defect: NULL matrix");
    const int nt = ceil_div(n, 32);
    hipLaunchKernelGGL(shg::symmetry_defect_kernel, dim3(nt, nt), dim3(256), 0, stream, n, S, ld, defect);     // non-negative doubles order like their bit patterns
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

static int covprop_diag_impl(shg_plan* p, const double* cov, int nmin, int lat0, int lat1, double* sigma, void* stream_, bool symmetric) {
    SHG_REQUIRE(p != nullptr, "shg_covprop_diag: NULL plan");
    SHG_REQUIRE(nmin >= 0 && nmin <= p->N + 1, "shg_covprop_diag: min_degree %d out of range", nmin);
    SHG_REQUIRE(lat0 >= 0 && lat1 <= p->nlat && lat0 <= lat1, "shg_covprop_diag: bad band [%d, %d)", lat0, lat1);
    if (lat0 == lat1) return SHG_OK;
    SHG_REQUIRE(sigma != nullptr, "shg_covprop_diag: NULL output");
    hipStream_t stream = (hipStream_t)stream_;
    PlanGuard guard(p, stream);
    const int Pfull = (p->N + 1) * (p->N + 1);
    const int Pn = Pfull - nmin * nmin;
    const long long M = (long long)(lat1 - lat0) * p->nlon;
    SHG_REQUIRE(M < (1LL << 31), "shg_covprop_diag: band too large");
    SHG_REQUIRE(Pn == 0 || cov != nullptr, "shg_covprop_diag: NULL covariance");

    int rc = build_pk_table(p, stream);
    if (rc) return rc;
    rc = covprop_build_cs_table(p, stream);
    if (rc) return rc;
    if (!p->pk_deg) {
        if (hipMalloc((void**)&p->pk_deg, (size_t)p->nlat * Pfull * sizeof(double)) != hipSuccess ||
            hipMalloc((void**)&p->rslot, (size_t)Pfull * sizeof(int)) != hipSuccess)
            return fail(SHG_ERR_NOMEM, "covariance propagation tables: allocation failed");
        hipLaunchKernelGGL(covprop_pkd_kernel, dim3((unsigned)ceil_div64((long long)p->nlat * Pfull, 256)), dim3(256), 0, stream, p->N,
                           p->nlat, p->ldlat, p->pk, p->pk_deg, p->rslot);
        SHG_HIP(hipGetLastError());
    }
    const int ncolblocks = std::max(1, ceil_div(Pn, BN));
    const size_t need = (size_t)ncolblocks * M;
    if (need > p->cov_partial_size) {
        if (p->cov_partial) {
            SHG_HIP(hipStreamSynchronize(stream));
            (void)hipFree(p->cov_partial);
            p->cov_partial = nullptr;
            p->cov_partial_size = 0;                 // a failed grow must not leave the old size behind
        }
        if (hipMalloc((void**)&p->cov_partial, need * sizeof(double)) != hipSuccess) return fail(SHG_ERR_NOMEM, "covariance propagation workspace (%zu doubles)", need);
        p->cov_partial_size = need;
    }
    if (Pn == 0) {
        SHG_HIP(hipMemsetAsync(sigma, 0, M * sizeof(double), stream));
        return SHG_OK;
    }
    // Two kernels: covprop_rows (covprop.hip) keeps row tiles inside one parallel (operand tiles are plain table rows) but
    // pads every parallel to a multiple of 128 meridians; the general kernel generates A element-wise and wastes nothing.
    const int padded = round_up(p->nlon, 128);
    if (!symmetric && (padded - p->nlon) * 25 <= p->nlon) {   // padding waste <= 4 %
        rc = covprop_rows(p, cov, Pn, nmin * nmin, lat0, lat1, p->cov_partial, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(covprop_reduce_kernel, dim3(ceil_div((int)M, 256)), dim3(256), 0, stream, (int)M, ncolblocks, p->cov_partial, sigma);
        SHG_HIP(hipGetLastError());
        return SHG_OK;
    }
    return covprop_generic(p->pk_deg, Pfull, p->cs_slot, p->nlon, p->rslot, p->nlon, p->nlon, (long long)lat0 * p->nlon, (int)M, cov, Pn,
                           nmin * nmin, p->cov_partial, sigma, p, stream, symmetric, false);
}
