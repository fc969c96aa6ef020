// Spherical-harmonic analysis on a regular grid: area-weighted least squares per order and per cos / sin,
//   x = (A^T W A)^-1 A^T W v,   A[(i, j)][n] = kn[i][n] P_nm(theta_i) cos|sin(m lon_j),  W = diag(area)
// (replaces RegularGrid.__analysis_matrix_per_order / to_potential_coefficients, grates/grid.py:665-696, 774-790).
//
// The design matrix is never formed.  Its separable structure gives
//   A^T W v  [n] = sum_i PK_m[n][i] g_s[i],      g_s[i]  = sum_j area[i][j] T_s(lon_j) v[i][j]      (longitude transform)
//   A^T W A [n][n'] = sum_i PK_m[n][i] w2_s[i] PK_m[n'][i],   w2_s[i] = sum_j area[i][j] T_s(lon_j)^2
// Stages (all on the device, batched over epochs):
//   1. weight_transpose   WVt[j][(b, i)] = area[i][j] v[b][i][j]
//   2. fp64 MFMA GEMM     Gt[s][(b, i)] = sum_j T_s(lon_j) WVt[j][(b, i)]          (gemm.hip)
//      (grids with the four-fold meridian symmetry: weight_fold + GEMMs over a quarter of the meridians, see below)
//   3. weight_squares     w2[s][i]
//   4. analysis_solve     one workgroup per slot s = (m, cos|sin): normal matrix, Cholesky, right-hand sides of all
//                         epochs, forward / backward substitution, scatter into anm.
#include "common.h"

namespace shg {

int covprop_build_cs_table(shg_plan* p, hipStream_t stream);   // gemm.hip

constexpr int kAnaEpochChunk = 256;   // epochs per pass: one pass for the usual batches (workspace 0.65 GB at 0.5 degree); 64 measured 1.98 ms per 240 epochs

__global__ __launch_bounds__(256) void weight_transpose_kernel(int nb, int nlat, int nlon, const double* __restrict__ v,
                                                               const double* __restrict__ area, double* __restrict__ wvt) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    const int j0 = blockIdx.x * 32;
    const long long r0 = (long long)blockIdx.y * 32;              // flat (b, i) row
    const long long rows = (long long)nb * nlat;
    for (int k = ty; k < 32; k += 8) {
        const long long r = r0 + k;
        const int j = j0 + tx;
        double x = 0.0;
        if (r < rows && j < nlon) x = v[r * nlon + j] * area[(size_t)(r % nlat) * nlon + j];
        tile[k][tx] = x;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int j = j0 + k;
        const long long r = r0 + tx;
        if (j < nlon && r < rows) wvt[(size_t)j * rows + r] = tile[tx][k];
    }
}

// Longitude fold for grids with the four-fold meridian symmetry (plan flag sym4).  With mu_c = lon[nlon/2 + c] in (0, pi/2)
// the four meridians  j1 = nlon/2 + c (mu),  j2 = nlon/2 - 1 - c (-mu),  j3 = c (mu - pi),  j4 = nlon - 1 - c (pi - mu)  carry
//   cos m lon = cos m mu * (1, 1, (-1)^m, (-1)^m),      sin m lon = sin m mu * (1, -1, (-1)^m, -(-1)^m),
// so the longitude transform needs only the quarter domain once the weighted values w = area * v are folded:
//   F[0] = (w1 + w2) - (w3 + w4)   cos, odd m          F[1] = (w1 - w2) - (w3 - w4)   sin, odd m
//   F[2] = (w1 + w2) + (w3 + w4)   cos, even m         F[3] = (w1 - w2) + (w3 - w4)   sin, even m
// each stored [nlon/4][(b, i)]: a quarter of the multiply-adds of the plain transform, the same bytes read.
__global__ __launch_bounds__(256) void weight_fold_kernel(int nb, int nlat, int nlon, const double* __restrict__ v,
                                                          const double* __restrict__ area, double* __restrict__ F) {
    __shared__ double tile[4][32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    const int nq = nlon / 4, h = nlon / 2;
    const int c0 = blockIdx.x * 32;
    const long long r0 = (long long)blockIdx.y * 32;              // flat (b, i) row
    const long long rows = (long long)nb * nlat;
    for (int k = ty; k < 32; k += 8) {
        const long long r = r0 + k;
        const int c = c0 + tx;
        double w1 = 0.0, w2 = 0.0, w3 = 0.0, w4 = 0.0;
        if (r < rows && c < nq) {
            const double* vr = v + r * nlon;
            const double* ar = area + (size_t)(r % nlat) * nlon;
            w1 = vr[h + c] * ar[h + c];
            w2 = vr[h - 1 - c] * ar[h - 1 - c];
            w3 = vr[c] * ar[c];
            w4 = vr[nlon - 1 - c] * ar[nlon - 1 - c];
        }
        const double p12 = w1 + w2, p34 = w3 + w4, q12 = w1 - w2, q34 = w3 - w4;
        tile[0][k][tx] = p12 - p34;
        tile[1][k][tx] = q12 - q34;
        tile[2][k][tx] = p12 + p34;
        tile[3][k][tx] = q12 + q34;
    }
    __syncthreads();
    const size_t plane = (size_t)nq * rows;
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k;
        const long long r = r0 + tx;
        if (c < nq && r < rows)
            for (int g = 0; g < 4; ++g) F[g * plane + (size_t)c * rows + r] = tile[g][tx][k];
    }
}

// Fold and longitude transform in one kernel (degrees up to 126): gt[slot][(b, i)] = sum_c T_slot(mu_c) F_group(slot)[c][(b, i)]
// without the folded planes ever reaching memory -- the values are read once (with their area weights), folded in registers
// and multiplied from LDS.  A workgroup owns 64 rows (b, i) and ALL slots: wave g the group g = cos even | cos odd | sin even |
// sin odd with MT tiles of 16 orders x 4 tiles of 16 rows (every trig fragment serves four row tiles, every value fragment MT
// order tiles: 7 LDS reads per 12 MFMAs at d/o 96).  Order 0 needs no trig: its sum over the columns is kept by the loader
// threads and reduced at the end, which leaves the four groups with N/2 or (N+1)/2 orders each.  The quarter domain is walked in
// chunks of 8 columns through two LDS stages; per chunk every thread loads two columns of one row in their four images (and
// the weights), folds them, and one row of the trig chunk.  Operand fragments: A = T[order][c] (LDS [c][order]), B = F[c][row]
// (LDS [group][c][row]), D row = order fk + 4 reg, column = row fr.
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
#ifndef SHG_ANA_PARITY
#define SHG_ANA_PARITY 1     // 0: A/B builds without the north-south parity split of the operator product
#endif
#ifndef SHG_ANA_X
#define SHG_ANA_X 0    // experiment switches (timing only): 1 no MFMAs, 2 no global loads after the first chunk, 4 / 8 / 16 no weight / trig / value loads
#endif
constexpr int kAtKC = 8, kAtRows = 64, kAtFP = kAtRows + 2;

// slot of order index k of group g: cos 2 (k + 1) | cos 2 k + 1 | sin 2 (k + 1) | sin 2 k + 1
__device__ __forceinline__ int analysis_slot(int g, int k) { return g == 0 ? 4 * k + 3 : (g == 1 ? 4 * k + 1 : (g == 2 ? 4 * k + 4 : 4 * k + 2)); }

// trig table in the order the kernel stages it: T[chunk][k][row], row = (group g, order index), T = cos | sin (order * mu_c),
// zero beyond the quarter domain and beyond a group's orders (built once per plan and degree)
__global__ void analysis_trig_kernel(int N, int nlon, int mt, const double* __restrict__ cs, double* __restrict__ T) {
    const int TR = 4 * mt * 16;
    const int nq = nlon / 4;
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;                                // column of the padded quarter domain
    if (row >= TR) return;
    const int g = row / (mt * 16), k = row % (mt * 16);
    const int count = (g & 1) ? (N + 1) / 2 : N / 2;
    double value = 0.0;
    if (c < nq && k < count) {
        const int slot = g == 0 ? 4 * k + 3 : (g == 1 ? 4 * k + 1 : (g == 2 ? 4 * k + 4 : 4 * k + 2));
        value = cs[(size_t)slot * nlon + nlon / 2 + c];
    }
    T[((size_t)(c / kAtKC) * kAtKC + c % kAtKC) * TR + row] = value;
}

template <int MT, bool ROWW>
__global__ __launch_bounds__(256) void analysis_transform_kernel(int nb, int nlat, int nlon, int N, const double* __restrict__ v,
                                                                 const double* __restrict__ area, const double* __restrict__ trig,
                                                                 double* __restrict__ gt) {
    constexpr int TR = 4 * MT * 16;                     // rows of the trig chunk: group g, tile t, order 16 t + m
    constexpr int TP = TR + 2;
    __shared__ __attribute__((aligned(16))) double TL[2][kAtKC][TP];
    __shared__ double FL[2][4][kAtKC][kAtFP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const int nq = nlon / 4, h = nlon / 2;
    const long long rows = (long long)nb * nlat;
    const long long r0 = (long long)blockIdx.x * kAtRows;
    const int count[4] = {N / 2, (N + 1) / 2, N / 2, (N + 1) / 2};
    // loader role: row lrow, columns c0 + 2 cp, c0 + 2 cp + 1
    const int lrow = tid >> 2, cp = tid & 3;
    const long long lr = r0 + lrow;
    const bool row_ok = lr < rows;
    const double* vrow = v + (row_ok ? lr : rows - 1) * nlon;
    const double* arow = area + (size_t)((row_ok ? lr : rows - 1) % nlat) * nlon;
    // trig role: MT 16-byte pieces of the chunk's [8][TR] block, lane-contiguous in memory; piece q covers rows 2 pr, 2 pr + 1 of
    // column pk
    const double wrow = ROWW ? arow[0] : 0.0;               // weights constant along the parallel: one value per thread

    // The values come from HBM, the weights and the trig rows from L2: the values of a chunk are requested two chunks ahead
    // (two register sets, used alternately), weights and trig rows one chunk ahead -- and in that order of age, because the
    // memory counter retires in order: what is waited for must be older than what may stay in flight.
    struct Images {
        double2_t x1, x2, x3, x4;                         // the four images of two columns (x2, x4: descending)
    };
    Images va, vb;
    double order0 = 0.0;                                  // sum of this thread's columns of its row: the order-0 transform
    double2_t w1, w2, w3, w4;
    double2_t tq[MT];
    // Columns come in aligned pairs (c even; nlon is a multiple of 4, so every image of a pair is a 16-byte load inside the
    // row); a pair that starts beyond the quarter domain is not loaded, a column beyond it is zeroed when it is staged.
    auto fetch_values = [&](Images& x, int c0) {
        const int c = c0 + 2 * cp;
        const int cc = c < nq ? c : 0;
        if (SHG_ANA_X & 16) return;
        x.x1 = *reinterpret_cast<const double2_t*>(vrow + h + cc);
        x.x2 = *reinterpret_cast<const double2_t*>(vrow + h - 2 - cc);
        x.x3 = *reinterpret_cast<const double2_t*>(vrow + cc);
        x.x4 = *reinterpret_cast<const double2_t*>(vrow + nlon - 2 - cc);
    };
    auto fetch_weights = [&](int c0) {
        const int c = c0 + 2 * cp;
        const int cc = c < nq ? c : 0;
        if (!ROWW && !(SHG_ANA_X & 4)) {
            w1 = *reinterpret_cast<const double2_t*>(arow + h + cc);
            w2 = *reinterpret_cast<const double2_t*>(arow + h - 2 - cc);
            w3 = *reinterpret_cast<const double2_t*>(arow + cc);
            w4 = *reinterpret_cast<const double2_t*>(arow + nlon - 2 - cc);
        }
        if (!(SHG_ANA_X & 8)) {
            const double* block = trig + (size_t)(c0 / kAtKC) * (kAtKC * TR);
#pragma unroll
            for (int q = 0; q < MT; ++q) tq[q] = *reinterpret_cast<const double2_t*>(block + 2 * (q * 256 + tid));
        }
    };
    // Rows beyond the batch may hold anything -- they only reach accumulator columns that the epilogue does not store; the
    // trig table is zero beyond the quarter domain.
    auto stage = [&](const Images& x, int c0, int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // ascending images hold column c + j in component j, descending ones in component 1 - j
            double a1, a2, a3, a4;
            if (ROWW) {
                a1 = (j ? x.x1.y : x.x1.x) * wrow, a2 = (j ? x.x2.x : x.x2.y) * wrow;
                a3 = (j ? x.x3.y : x.x3.x) * wrow, a4 = (j ? x.x4.x : x.x4.y) * wrow;
            } else {
                a1 = j ? x.x1.y * w1.y : x.x1.x * w1.x, a2 = j ? x.x2.x * w2.x : x.x2.y * w2.y;
                a3 = j ? x.x3.y * w3.y : x.x3.x * w3.x, a4 = j ? x.x4.x * w4.x : x.x4.y * w4.y;
            }
            const double p12 = a1 + a2, p34 = a3 + a4, q12 = a1 - a2, q34 = a3 - a4;
            order0 += (c0 + 2 * cp + j < nq) ? p12 + p34 : 0.0;
            FL[buf][0][2 * cp + j][lrow] = p12 + p34;       // cos, even orders
            FL[buf][1][2 * cp + j][lrow] = p12 - p34;       // cos, odd
            FL[buf][2][2 * cp + j][lrow] = q12 + q34;       // sin, even
            FL[buf][3][2 * cp + j][lrow] = q12 - q34;       // sin, odd
        }
#pragma unroll
        for (int q = 0; q < MT; ++q) {
            const int piece = q * 256 + tid;                // piece of two rows: column piece / (TR / 2), rows 2 (piece % (TR / 2))
            *reinterpret_cast<double2_t*>(&TL[buf][piece / (TR / 2)][2 * (piece % (TR / 2))]) = tq[q];
        }
    };

    double4_t acc[MT][4];                               // [order tile][row tile] of this wave's group
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[t][nt] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const int g = wave;
    const int ntile = (count[g] + 15) / 16;

    const int nchunk = (nq + kAtKC - 1) / kAtKC;
    // products of one chunk: per k-step MT trig fragments and four value fragments, read one k-step ahead of their MFMAs
    auto products = [&](int buf) {
        double fa[2][MT], fb[2][4];
        auto read_step = [&](int ks, int set) {
#pragma unroll
            for (int t = 0; t < MT; ++t) fa[set][t] = TL[buf][4 * ks + fk][(g * MT + t) * 16 + fr];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) fb[set][nt] = FL[buf][g][4 * ks + fk][16 * nt + fr];
        };
        read_step(0, 0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks + 1 < 2) read_step(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < MT; ++t)
                if (t < ntile && !(SHG_ANA_X & 1)) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ks & 1][t], fb[ks & 1][nt], acc[t][nt], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // one step: weights of chunk ch + 1 and values of chunk ch + 2 requested, products of chunk ch, chunk ch + 1 staged
    auto step = [&](int ch, Images& next, Images& after) {
        const int buf = ch & 1;
        if (ch + 1 < nchunk && !(SHG_ANA_X & 2)) fetch_weights((ch + 1) * kAtKC);
        if (ch + 2 < nchunk && !(SHG_ANA_X & 2)) fetch_values(after, (ch + 2) * kAtKC);
        products(buf);
        if (ch + 1 < nchunk) stage(next, (ch + 1) * kAtKC, buf ^ 1);
        __syncthreads();
    };
    fetch_values(va, 0);
    fetch_weights(0);
    if (nchunk > 1) fetch_values(vb, kAtKC);
    stage(va, 0, 0);
    __syncthreads();
    for (int ch = 0; ch < nchunk; ch += 2) {
        step(ch, vb, va);                         // chunk ch + 1 sits in vb, chunk ch + 2 goes to va
        if (ch + 1 < nchunk) step(ch + 1, va, vb);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const long long r = r0 + 16 * nt + fr;
        if (r < rows) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int k = 16 * t + fk + 4 * reg;
                    if (k < count[g]) gt[(size_t)analysis_slot(g, k) * rows + r] = acc[t][nt][reg];
                }
        }
    }
    // order 0: the four loader threads of a row sit in adjacent lanes
    order0 += __shfl_xor(order0, 1);
    order0 += __shfl_xor(order0, 2);
    if (cp == 0 && row_ok) gt[lr] = order0;
}

// w2[s][i] = sum_j area[i][j] cs[s][j]^2 : one workgroup per parallel
__global__ __launch_bounds__(256) void weight_squares_kernel(int S, int nlat, int nlon, const double* __restrict__ area,
                                                             const double* __restrict__ cs, double* __restrict__ w2) {
    __shared__ double arow[256];
    const int i = blockIdx.x;
    for (int s0 = 0; s0 < S; s0 += 256) {
        const int s = s0 + threadIdx.x;
        double acc = 0.0;
        for (int j0 = 0; j0 < nlon; j0 += 256) {
            __syncthreads();
            arow[threadIdx.x] = (j0 + (int)threadIdx.x < nlon) ? area[(size_t)i * nlon + j0 + threadIdx.x] : 0.0;
            __syncthreads();
            if (s < S) {
                const int jn = min(256, nlon - j0);
                const double* c = cs + (size_t)s * nlon + j0;
                for (int jj = 0; jj < jn; ++jj) acc = fma(arow[jj] * c[jj], c[jj], acc);
            }
        }
        if (s < S) w2[(size_t)s * nlat + i] = acc;
    }
}

// blas.hip
int gemm_ex(bool ta, bool tb, int M, int N, int K, double alpha, const double* A, int lda, long long strideA, const double* B, int ldb,
            long long strideB, double beta, double* C, int ldc, long long strideC, int batch, bool upper_only, hipStream_t stream);
int factor_invert_batched(int n, double* A, int lda, long long strideA, double* X, int ldx, long long strideX, int batch, int* info,
                          hipStream_t stream);
int potrf_upper(int n, double* A, int lda, double* work, int* info, hipStream_t stream);
int trtri_upper(int n, const double* U, int ldu, double* X, int ldx, double* work, hipStream_t stream);

// Per slot s = (m, cos | sin) the least-squares solution is x_s = H_s g_s with the operator
//   H_s = (PK_m W2_s PK_m^T)^-1 PK_m      [d_s][nlat],   d_s = N + 1 - max(m, nmin) degrees,
// which depends on the plan, the area weights and nmin only.  It is built once (batched over the 2N+1 slots: MFMA GEMMs for
// the normal matrices, one workgroup per slot for the Cholesky factor U_s and its inverse, N_s^-1 PK_m = U^-1 (U^-T PK_m) as
// two more GEMMs), cached in the plan under a checksum of the weights, and every call is then the longitude transform plus
// ONE batched GEMM.  Slots are padded to N + 1 rows (zero rows of PK, unit diagonal in the normal matrix).

// PKs[s][a][i] = PK[(m, n0 + a)][i] (a < d_s, else 0);  PKw = PKs * w2[s][i]
__global__ void analysis_gather_kernel(int N, int nmin, int nlat, int ldlat, const double* __restrict__ pk, const double* __restrict__ w2,
                                       double* __restrict__ pks, double* __restrict__ pkw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int a = blockIdx.y, s = blockIdx.z;
    if (i >= nlat) return;
    const int m = (s + 1) >> 1;
    const int n0 = max(m, nmin);
    const int d = N + 1 - n0;
    const size_t o = ((size_t)s * (N + 1) + a) * nlat + i;
    const double v = a < d ? pk[(size_t)(order_offset(N, m) + n0 - m + a) * ldlat + i] : 0.0;
    pks[o] = v;
    pkw[o] = v * w2[(size_t)s * nlat + i];
}

__global__ void analysis_pad_kernel(int N, int nmin, double* __restrict__ nmat) {
    const int s = blockIdx.x, a = threadIdx.x;
    const int m = (s + 1) >> 1;
    const int d = N + 1 - max(m, nmin);
    if (a <= N && a >= d) nmat[((size_t)s * (N + 1) + a) * (N + 1) + a] = 1.0;
}

// anm[b][n][m] / anm[b][m-1][n] <- X[s][a][b]
__global__ void analysis_scatter_kernel(int N, int nmin, int nb, int b0, const double* __restrict__ X, double* __restrict__ anm) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int a = blockIdx.y, s = blockIdx.z;
    if (b >= nb) return;
    const int m = (s + 1) >> 1;
    const bool sine = s > 0 && (s & 1) == 0;
    const int n0 = max(m, nmin);
    if (a >= N + 1 - n0) return;
    const int n = n0 + a;
    double* out = anm + (size_t)(b0 + b) * (N + 1) * (N + 1);
    out[sine ? (size_t)(m - 1) * (N + 1) + n : (size_t)n * (N + 1) + m] = X[((size_t)s * (N + 1) + a) * nb + b];
}

// x_s = H_s g_s for all slots and epochs, written straight into anm (degrees up to 127, even parallel count).  A slot of order m
// has only d_s = N + 1 - max(m, nmin) rows: a batched GEMM on the padded [N+1][nlat] operators computes (N+1)(2N+1) rows
// where (N+1)^2 - nmin^2 are wanted, and rounds N + 1 = 97 up to a 128-row tile on top.  Here a workgroup takes one slot and
// 64 epochs: ceil(d_s / 16) row tiles of 16, wave w the epochs 16 w .. 16 w + 15; operands through two LDS stages of 16
// parallels ([k][row] and [k][epoch], so that the fragment reads are conflict free), results scattered into
// anm[b][n][m] / anm[b][m-1][n] by the lanes that hold them (the scatter kernel's pattern, without the round trip through X).
constexpr int kOpKC = 16, kOpCols = 64, kOpRows = 128;

__global__ __launch_bounds__(256) void analysis_operator_kernel(int N, int nmin, int nlat, int nb, int b0, int ngroups, const double* __restrict__ H,
                                                                const double* __restrict__ gt, double* __restrict__ anm) {
    __shared__ double AL[2][kOpKC][kOpRows + 2];
    __shared__ double BL[2][kOpKC][kOpCols + 2];
    // Workgroups go to the 8 XCDs round robin by their linear index.  All epoch groups of a slot read the same operator H_s: XCD x takes
    // the slots x, x + 8, ... with their epoch groups next to each other, so that H_s comes in from memory once per XCD's L2 (slot-major
    // over all XCDs, the order of round 3, every epoch group fetched it again: 108 instead of 27 MB at d/o 96, 240 epochs).
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int s = 8 * (seq / ngroups) + xcd, cb = (seq % ngroups) * kOpCols;
    if (s > 2 * N) return;
    const int m = (s + 1) >> 1;
    const bool sine = s > 0 && (s & 1) == 0;
    const int n0 = max(m, nmin), R = N + 1;
    const int d = R - n0;
    if (d <= 0) return;
    const int ntile = (d + 15) >> 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const double* Hs = H + (size_t)s * R * nlat;
    const double* gs = gt + (size_t)s * nb * nlat;                 // [nb][nlat]
    const int kp = tid & 7, lrow = tid >> 3;                       // loader: 16-byte piece kp of row lrow (+ 32 per pass)
    double2_t ra[4], rb[2];
    auto fetch = [&](int k0) {
        const int k = k0 + 2 * kp;
        const bool kok = k < nlat;                                 // nlat is even: a pair is inside or outside
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = lrow + 32 * p;
            ra[p] = (kok && row < d) ? *reinterpret_cast<const double2_t*>(Hs + (size_t)row * nlat + k) : (double2_t){0.0, 0.0};
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int col = cb + lrow + 32 * p;
            rb[p] = (kok && col < nb) ? *reinterpret_cast<const double2_t*>(gs + (size_t)col * nlat + k) : (double2_t){0.0, 0.0};
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (lrow + 32 * p < 16 * ntile) {
                AL[buf][2 * kp][lrow + 32 * p] = ra[p].x;
                AL[buf][2 * kp + 1][lrow + 32 * p] = ra[p].y;
            }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            BL[buf][2 * kp][lrow + 32 * p] = rb[p].x;
            BL[buf][2 * kp + 1][lrow + 32 * p] = rb[p].y;
        }
    };
    double4_t acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const int nchunk = (nlat + kOpKC - 1) / kOpKC;
    fetch(0);
    stage(0);
    __syncthreads();
    for (int ch = 0; ch < nchunk; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunk) fetch((ch + 1) * kOpKC);
        double fa[2][8], fb[2];
        auto read_step = [&](int kk, int set) {
            fb[set] = BL[buf][4 * kk + fk][16 * wave + fr];
#pragma unroll
            for (int t = 0; t < 8; ++t) fa[set][t] = AL[buf][4 * kk + fk][(t < ntile ? 16 * t : 0) + fr];
        };
        read_step(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk + 1 < 4) read_step(kk + 1, (kk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (t < ntile) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[kk & 1][t], fb[kk & 1], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ch + 1 < nchunk) stage(buf ^ 1);
        __syncthreads();
    }
    const int b = cb + 16 * wave + fr;
    if (b < nb) {
        double* out = anm + (size_t)(b0 + b) * R * R;
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t < ntile) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int a = 16 * t + fk + 4 * reg;
                    if (a < d) {
                        const int n = n0 + a;
                        out[sine ? (size_t)(m - 1) * R + n : (size_t)n * R + m] = acc[t][reg];
                    }
                }
            }
    }
}

// ---- North-south parity split of the operator product (round 6).
// On parallels that are mirror images of each other (colat[nlat-1-i] = pi - colat[i], equal kernel factors) with mirror-symmetric weights,
// P_nm(pi - theta) = (-1)^(n-m) P_nm(theta) makes the normal equations of a slot decouple by the parity of n - m, and the operator
// inherits the symmetry: H_s[n][nlat-1-i] = (-1)^(n-m) H_s[n][i].  Then
//     x_n = sum_i H_s[n][i] g[i] = sum_{i < nlat/2} Hp[n][i] (g[i] +- g[nlat-1-i]),      Hp[n][i] = (H_s[n][i] +- H_s[n][nlat-1-i]) / 2,
// with + for even and - for odd n - m: half the products and half the operator bytes.  Hp is formed from the operator AS BUILT (from the
// reference's own colatitudes, whose mirror images differ by ~1e-14 rad near the poles): the part of H_s that does not have the symmetry,
// (H_s[n][i] -+ H_s[n][nlat-1-i]) / 2, is dropped, and its largest entry relative to the largest entry of H is measured while Hp is
// formed -- the split is used when that defect is below kParityDefectMax (asymmetric weights or parallels show up as a defect of order
// one and keep the full product).
//   Hp [S][2 RE][nlat/2], RE = ceil((N+1)/2): rows j < dE of a slot hold the even rows a = e0 + 2 j (n = n0 + a), rows RE + j the odd ones.
constexpr double kParityDefectMax = 5e-12;

__device__ __forceinline__ void atomic_max_positive(double* addr, double v) {       // v >= 0: the bit patterns order like the values
    atomicMax(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__builtin_bit_cast(long long, v));
}

__global__ __launch_bounds__(256) void analysis_parity_kernel(int N, int nmin, int nlat, const double* __restrict__ H, double* __restrict__ Hp,
                                                              double* __restrict__ norms /* [2]: max |H|, max |dropped part| */) {
    const int R = N + 1, RE = (R + 1) / 2, nh = nlat / 2;
    const int s = blockIdx.y, row = blockIdx.x;                   // row of Hp: parity = row / RE, j = row % RE
    const int m = (s + 1) >> 1, n0 = max(m, nmin), d = R - n0;
    const int odd = row / RE, j = row % RE;
    const int e0 = (n0 - m) & 1;                                   // first even row of the slot
    const int a = (odd ? 1 - e0 : e0) + 2 * j;
    double* out = Hp + ((size_t)s * 2 * RE + row) * nh;
    double big = 0.0, drop = 0.0;
    for (int i = threadIdx.x; i < nh; i += 256) {
        double v = 0.0;
        if (d > 0 && a < d) {
            const double hn = H[((size_t)s * R + a) * nlat + i], hs = H[((size_t)s * R + a) * nlat + nlat - 1 - i];
            v = odd ? 0.5 * (hn - hs) : 0.5 * (hn + hs);
            big = fmax(big, fmax(fabs(hn), fabs(hs)));
            drop = fmax(drop, fabs(odd ? 0.5 * (hn + hs) : 0.5 * (hn - hs)));
        }
        out[i] = v;
    }
    for (int o = 32; o > 0; o >>= 1) {
        big = fmax(big, __shfl_xor(big, o));
        drop = fmax(drop, __shfl_xor(drop, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomic_max_positive(norms, big);
        atomic_max_positive(norms + 1, drop);
    }
}

// x_s = Hp_s (g_s +- mirrored g_s): the operator product on the northern half.  A workgroup takes one slot and 64 epochs as
// analysis_operator_kernel does; the loader forms both combinations of a chunk of 16 northern parallels and their mirror images from
// one read of the transform's output, the even row tiles multiply the sums, the odd ones the differences.
__global__ __launch_bounds__(256) void analysis_operator_parity_kernel(int N, int nmin, int nlat, int nb, int b0, int ngroups, const double* __restrict__ Hp,
                                                                       const double* __restrict__ gt, double* __restrict__ anm) {
    __shared__ double AL[2][kOpKC][kOpRows + 2];
    __shared__ double BL[2][2][kOpKC][kOpCols + 2];               // [stage][sum | difference]
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int s = 8 * (seq / ngroups) + xcd, cb = (seq % ngroups) * kOpCols;
    if (s > 2 * N) return;
    const int m = (s + 1) >> 1;
    const bool sine = s > 0 && (s & 1) == 0;
    const int n0 = max(m, nmin), R = N + 1, RE = (R + 1) / 2, nh = nlat / 2;
    const int d = R - n0;
    if (d <= 0) return;
    const int e0 = (n0 - m) & 1;
    const int dE = (d - e0 + 1) / 2, dO = d - dE;
    const int tE = (dE + 15) >> 4, ntile = tE + ((dO + 15) >> 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const double* Hs = Hp + (size_t)s * 2 * RE * nh;
    const double* gs = gt + (size_t)s * nb * nlat;                 // [nb][nlat]
    const int kp = tid & 7, lrow = tid >> 3;                       // loader: 16-byte piece kp of row lrow (+ 32 per pass)
    // The product is bound by the latency of its operand loads, not by its MFMAs (one chunk ahead: 3 TB/s of operands at two workgroups
    // per CU): the operands of a chunk are requested TWO chunks ahead, in two register sets used in turn.
    struct Operands {
        double2_t ra[4], rn[2], rs[2];
    };
    Operands oa, ob;
    auto fetch = [&](Operands& o, int k0) {
        const int k = k0 + 2 * kp;
        const bool kok = k < nh;                                   // nh is even: a pair is inside or outside
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = lrow + 32 * p;                         // row of the stage: even tiles first, then the odd ones
            const bool even = row < 16 * tE;
            const int j = even ? row : row - 16 * tE;
            const bool ok = kok && (even ? j < dE : j < dO);
            o.ra[p] = ok ? *reinterpret_cast<const double2_t*>(Hs + (size_t)((even ? 0 : RE) + j) * nh + k) : (double2_t){0.0, 0.0};
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int col = cb + lrow + 32 * p;
            const bool ok = kok && col < nb;
            o.rn[p] = ok ? *reinterpret_cast<const double2_t*>(gs + (size_t)col * nlat + k) : (double2_t){0.0, 0.0};
            o.rs[p] = ok ? *reinterpret_cast<const double2_t*>(gs + (size_t)col * nlat + nlat - 2 - k) : (double2_t){0.0, 0.0};      // mirrors of k + 1, k
        }
    };
    auto stage = [&](const Operands& o, int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (lrow + 32 * p < 16 * ntile) {
                AL[buf][2 * kp][lrow + 32 * p] = o.ra[p].x;
                AL[buf][2 * kp + 1][lrow + 32 * p] = o.ra[p].y;
            }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            BL[buf][0][2 * kp][lrow + 32 * p] = o.rn[p].x + o.rs[p].y;
            BL[buf][0][2 * kp + 1][lrow + 32 * p] = o.rn[p].y + o.rs[p].x;
            BL[buf][1][2 * kp][lrow + 32 * p] = o.rn[p].x - o.rs[p].y;
            BL[buf][1][2 * kp + 1][lrow + 32 * p] = o.rn[p].y - o.rs[p].x;
        }
    };
    double4_t acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const int nchunk = (nh + kOpKC - 1) / kOpKC;
    auto products = [&](int buf) {
        double fa[2][8], fb[2][2];
        auto read_step = [&](int kk, int set) {
            fb[set][0] = BL[buf][0][4 * kk + fk][16 * wave + fr];
            fb[set][1] = BL[buf][1][4 * kk + fk][16 * wave + fr];
#pragma unroll
            for (int t = 0; t < 8; ++t) fa[set][t] = AL[buf][4 * kk + fk][(t < ntile ? 16 * t : 0) + fr];
        };
        read_step(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk + 1 < 4) read_step(kk + 1, (kk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (t < ntile) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[kk & 1][t], t < tE ? fb[kk & 1][0] : fb[kk & 1][1], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // chunk ch sits in stage ch & 1; `next` holds chunk ch + 1, `after` is free for chunk ch + 2
    auto step = [&](int ch, Operands& next, Operands& after) {
        if (ch + 2 < nchunk) fetch(after, (ch + 2) * kOpKC);
        products(ch & 1);
        if (ch + 1 < nchunk) stage(next, (ch & 1) ^ 1);
        __syncthreads();
    };
    fetch(oa, 0);
    if (nchunk > 1) fetch(ob, kOpKC);
    stage(oa, 0);
    __syncthreads();
    for (int ch = 0; ch < nchunk; ch += 2) {
        step(ch, ob, oa);                          // chunk ch + 1 is in ob, chunk ch + 2 goes to oa
        if (ch + 1 < nchunk) step(ch + 1, oa, ob);
    }
    const int b = cb + 16 * wave + fr;
    if (b < nb) {
        double* out = anm + (size_t)(b0 + b) * R * R;
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t < ntile) {
                const bool even = t < tE;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int j = 16 * (even ? t : t - tE) + fk + 4 * reg;
                    if (j < (even ? dE : dO)) {
                        const int n = n0 + (even ? e0 : 1 - e0) + 2 * j;
                        out[sine ? (size_t)(m - 1) * R + n : (size_t)n * R + m] = acc[t][reg];
                    }
                }
            }
    }
}

__global__ __launch_bounds__(256) void analysis_zero_kernel(long long n, double* __restrict__ x) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) x[e] = 0.0;
}

// *varies becomes non-zero when the weights are not constant along every parallel
__global__ __launch_bounds__(256) void analysis_rowconst_kernel(int nlat, int nlon, const double* __restrict__ area, int* __restrict__ varies) {
    int d = 0;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < (long long)nlat * nlon; e += (long long)gridDim.x * 256)
        d |= __builtin_bit_cast(long long, area[e]) != __builtin_bit_cast(long long, area[(e / nlon) * nlon]) ? 1 : 0;
    if (d) atomicOr(varies, 1);
}

// number of entries in which the area weights differ from the ones the cached operator was built for (bitwise compare)
__global__ __launch_bounds__(256) void analysis_compare_kernel(long long n, const double* __restrict__ a, const double* __restrict__ b,
                                                               int* __restrict__ diff) {
    int d = 0;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256)
        d |= __builtin_bit_cast(long long, a[e]) != __builtin_bit_cast(long long, b[e]) ? 1 : 0;
    if (d) atomicOr(diff, 1);
}

// F[row(s, a)][(i, j)] = H_s[a][i] area[i][j] T_s(lon_j): the dense analysis operator in degree-wise row order
// (grates/grid.py:698-730 assembles it order by order from M x (N + 1 - m) design matrices)
__global__ __launch_bounds__(256) void analysis_matrix_kernel(int N, int nmin, int nlat, int nlon, const double* __restrict__ H,
                                                              const double* __restrict__ area, const double* __restrict__ cs,
                                                              double* __restrict__ F) {
    const int i = blockIdx.x, a = blockIdx.y, s = blockIdx.z;
    const int m = (s + 1) >> 1;
    const bool sine = s > 0 && (s & 1) == 0;
    const int n0 = max(m, nmin);
    if (a >= N + 1 - n0) return;
    const int n = n0 + a;
    const size_t row = (size_t)n * n - (size_t)nmin * nmin + (m == 0 ? 0 : 2 * m - 1 + (sine ? 1 : 0));
    const double h = H[((size_t)s * (N + 1) + a) * nlat + i];
    double* out = F + row * ((size_t)nlat * nlon) + (size_t)i * nlon;
    for (int j = threadIdx.x; j < nlon; j += 256) out[j] = h * area[(size_t)i * nlon + j] * cs[(size_t)s * nlon + j];
}

// p->ana_Hp from p->ana_H where the parallels are mirror images of each other; p->ana_parity tells whether the product may use it
static int build_parity_operator(shg_plan* p, int nmin, hipStream_t stream) {
    const int N = p->N, S = 2 * N + 1, nlat = p->nlat, R = N + 1, RE = (R + 1) / 2;
    p->ana_parity = false;
    p->ana_parity_defect = -1.0;
    if (!p->sym_ns || nlat % 4 != 0 || R > kOpRows) return SHG_OK;
    const size_t count = (size_t)S * 2 * RE * (nlat / 2);
    if (!p->ana_Hp && hipMalloc((void**)&p->ana_Hp, count * sizeof(double)) != hipSuccess) return fail(SHG_ERR_NOMEM, "analysis operator allocation failed");
    double* norms = nullptr;
    if (workspace_alloc((void**)&norms, 2 * sizeof(double), stream) != hipSuccess) return fail(SHG_ERR_NOMEM, "shg_analysis: workspace allocation failed");
    SHG_HIP(hipMemsetAsync(norms, 0, 2 * sizeof(double), stream));
    hipLaunchKernelGGL(analysis_parity_kernel, dim3(2 * RE, S), dim3(256), 0, stream, N, nmin, nlat, p->ana_H, p->ana_Hp, norms);
    double host[2] = {0.0, 0.0};
    SHG_HIP(hipMemcpyAsync(host, norms, sizeof(host), hipMemcpyDeviceToHost, stream));
    SHG_HIP(hipStreamSynchronize(stream));
    (void)hipFreeAsync(norms, stream);
    p->ana_parity_defect = host[0] > 0.0 ? host[1] / host[0] : 0.0;
    p->ana_parity = p->ana_parity_defect <= kParityDefectMax;
    return SHG_OK;
}

// builds p->ana_H for the weights w2 (device, [S][nlat]); returns SHG_ERR_INVALID if a normal matrix is not positive definite
static int build_analysis_operator(shg_plan* p, const double* w2, int nmin, hipStream_t stream) {
    const int N = p->N, S = 2 * N + 1, nlat = p->nlat, R = N + 1;
    const size_t slab = (size_t)S * R * nlat;
    if (!p->ana_H && hipMalloc((void**)&p->ana_H, slab * sizeof(double)) != hipSuccess) return fail(SHG_ERR_NOMEM, "analysis operator allocation failed");
    double *pks = nullptr, *pkw = nullptr, *nmat = nullptr, *uinv = nullptr, *work = nullptr;
    int* info = nullptr;
    if (workspace_alloc((void**)&pks, slab * sizeof(double), stream) != hipSuccess || workspace_alloc((void**)&pkw, slab * sizeof(double), stream) != hipSuccess ||
        workspace_alloc((void**)&nmat, (size_t)S * R * R * sizeof(double), stream) != hipSuccess ||
        workspace_alloc((void**)&uinv, (size_t)S * R * R * sizeof(double), stream) != hipSuccess ||
        workspace_alloc((void**)&work, ((size_t)R * R + 128 * 128) * sizeof(double), stream) != hipSuccess ||
        workspace_alloc((void**)&info, sizeof(int), stream) != hipSuccess) {
        for (void* q : {(void*)pks, (void*)pkw, (void*)nmat, (void*)uinv, (void*)work, (void*)info})
            if (q) (void)hipFreeAsync(q, stream);
        return fail(SHG_ERR_NOMEM, "analysis operator workspace allocation failed");
    }
    if (zero_fill(info, stream) != SHG_OK) return SHG_ERR_HIP;
    hipLaunchKernelGGL(analysis_gather_kernel, dim3(ceil_div(nlat, 128), R, S), dim3(128), 0, stream, N, nmin, nlat, p->ldlat, p->pk, w2, pks, pkw);
    // normal matrices N_s = PKw_s PKs_s^T
    int rc = gemm_ex(false, true, R, R, nlat, 1.0, pkw, nlat, (long long)R * nlat, pks, nlat, (long long)R * nlat, 0.0, nmat, R, (long long)R * R, S,
                     false, stream);
    if (!rc) {
        hipLaunchKernelGGL(analysis_pad_kernel, dim3(S), dim3(std::max(64, round_up(R, 64))), 0, stream, N, nmin, nmat);
        if (R <= 128) {
            rc = factor_invert_batched(R, nmat, R, (long long)R * R, uinv, R, (long long)R * R, S, info, stream);
        } else {
            for (int s = 0; s < S && !rc; ++s) {
                rc = potrf_upper(R, nmat + (size_t)s * R * R, R, work + (size_t)R * R, info, stream);
                if (!rc) rc = trtri_upper(R, nmat + (size_t)s * R * R, R, uinv + (size_t)s * R * R, R, work, stream);
            }
        }
    }
    // H_s = U_s^-1 (U_s^-T PKs_s)
    if (!rc) rc = gemm_ex(true, false, R, nlat, R, 1.0, uinv, R, (long long)R * R, pks, nlat, (long long)R * nlat, 0.0, pkw, nlat, (long long)R * nlat, S, false, stream);
    if (!rc) rc = gemm_ex(false, false, R, nlat, R, 1.0, uinv, R, (long long)R * R, pkw, nlat, (long long)R * nlat, 0.0, p->ana_H, nlat, (long long)R * nlat, S, false, stream);
    int bad = 0;
    if (!rc) {
        SHG_HIP(hipMemcpyAsync(&bad, info, sizeof(int), hipMemcpyDeviceToHost, stream));
        SHG_HIP(hipStreamSynchronize(stream));
    }
    (void)hipFreeAsync(pks, stream);
    (void)hipFreeAsync(pkw, stream);
    (void)hipFreeAsync(nmat, stream);
    (void)hipFreeAsync(uinv, stream);
    (void)hipFreeAsync(work, stream);
    (void)hipFreeAsync(info, stream);
    if (rc) return rc;
    if (bad) return fail(SHG_ERR_INVALID, "shg_analysis: a normal matrix is not positive definite (grid does not resolve the requested degrees)");
    return build_parity_operator(p, nmin, stream);
}

}  // namespace shg

using namespace shg;

// The cached operator H (p->ana_H) belongs to one set of area weights and one minimum degree.  The weights are compared
// entry by entry on the device with the copy kept beside the operator; the operator is rebuilt when they differ.
static int analysis_tables(shg_plan* p, hipStream_t stream) {
    const int rc = build_pk_table(p, stream);
    return rc ? rc : covprop_build_cs_table(p, stream);
}

static bool analysis_operator_cached(const shg_plan* p, int nmin) { return p->ana_H && p->ana_area && p->ana_nmin == nmin; }

static int rebuild_analysis_operator(shg_plan* p, const double* area, int nmin, hipStream_t stream) {
    const int N = p->N, S = 2 * N + 1, nlat = p->nlat, nlon = p->nlon;
    const size_t na = (size_t)nlat * nlon;
    p->ana_nmin = -1;
    if (!p->ana_area && hipMalloc((void**)&p->ana_area, na * sizeof(double)) != hipSuccess) return fail(SHG_ERR_NOMEM, "analysis weight copy allocation failed");
    double* w2 = nullptr;
    if (workspace_alloc((void**)&w2, (size_t)S * nlat * sizeof(double), stream) != hipSuccess) return fail(SHG_ERR_NOMEM, "shg_analysis: workspace allocation failed");
    hipLaunchKernelGGL(weight_squares_kernel, dim3(nlat), dim3(256), 0, stream, S, nlat, nlon, area, p->cs_slot, w2);
    int rc = build_analysis_operator(p, w2, nmin, stream);
    (void)hipFreeAsync(w2, stream);
    if (rc) return rc;
    SHG_HIP(hipMemcpyAsync(p->ana_area, area, na * sizeof(double), hipMemcpyDeviceToDevice, stream));
    {   // weights that are constant along every parallel (geographic and Gauss grids) need not be streamed by the transform kernel
        ScratchLease lease(stream);
        int* varies = (int*)lease.get(kScratchAnaFlag, sizeof(int));
        if (!varies) return fail(SHG_ERR_NOMEM, "shg_analysis: workspace allocation failed");
        int host = 1;
        if (zero_fill(varies, stream) != SHG_OK) return SHG_ERR_HIP;
        hipLaunchKernelGGL(analysis_rowconst_kernel, dim3(256), dim3(256), 0, stream, nlat, nlon, area, varies);
        SHG_HIP(hipMemcpyAsync(&host, varies, sizeof(int), hipMemcpyDeviceToHost, stream));
        SHG_HIP(hipStreamSynchronize(stream));
        p->ana_rowconst = host == 0;
    }
    p->ana_nmin = nmin;
    return SHG_OK;
}

// *diff (device, zeroed here) becomes non-zero when `area` is not the set of weights the cached operator was built for
static int launch_weight_compare(shg_plan* p, const double* area, int* diff, hipStream_t stream) {
    if (zero_fill(diff, stream) != SHG_OK) return SHG_ERR_HIP;
    hipLaunchKernelGGL(analysis_compare_kernel, dim3(256), dim3(256), 0, stream, (long long)p->nlat * p->nlon, area, p->ana_area, diff);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

static int ensure_analysis_operator(shg_plan* p, const double* area, int nmin, hipStream_t stream) {
    int rc = analysis_tables(p, stream);
    if (rc) return rc;
    if (analysis_operator_cached(p, nmin)) {
        int* diff = nullptr;
        if (workspace_alloc((void**)&diff, sizeof(int), stream) != hipSuccess) return fail(SHG_ERR_NOMEM, "shg_analysis: workspace allocation failed");
        int host = 0;
        rc = launch_weight_compare(p, area, diff, stream);
        hipError_t e = rc ? hipSuccess : hipMemcpyAsync(&host, diff, sizeof(int), hipMemcpyDeviceToHost, stream);
        if (!rc && e == hipSuccess) e = hipStreamSynchronize(stream);
        (void)hipFreeAsync(diff, stream);
        if (rc) return rc;
        if (e != hipSuccess) return fail(SHG_ERR_HIP, "shg_analysis: weight comparison failed: %s", hipGetErrorString(e));
        if (host == 0) return SHG_OK;
    }
    return rebuild_analysis_operator(p, area, nmin, stream);
}

extern "C" int shg_analysis_info(const shg_plan* p, double info[2]) {
    SHG_REQUIRE(p != nullptr && info != nullptr, "shg_analysis_info: NULL argument");
    info[0] = p->ana_nmin < 0 ? -1.0 : (p->ana_parity && SHG_ANA_PARITY ? 1.0 : 0.0);
    info[1] = p->ana_parity_defect;
    return SHG_OK;
}

extern "C" int shg_analysis_matrix(shg_plan* p, const double* area, int nmin, double* F, void* stream_) {
    SHG_REQUIRE(p != nullptr, "shg_analysis_matrix: NULL plan");
    SHG_REQUIRE(nmin >= 0 && nmin <= p->N, "shg_analysis_matrix: min_degree %d out of range", nmin);
    SHG_REQUIRE(area && F, "shg_analysis_matrix: NULL pointer");
    hipStream_t stream = (hipStream_t)stream_;
    PlanGuard guard(p, stream);
    const int rc = ensure_analysis_operator(p, area, nmin, stream);
    if (rc) return rc;
    const int N = p->N;
    hipLaunchKernelGGL(analysis_matrix_kernel, dim3(p->nlat, N + 1, 2 * N + 1), dim3(256), 0, stream, N, nmin, p->nlat, p->nlon, p->ana_H, area,
                       p->cs_slot, F);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

// the fused transform kernel's trig table for mt order tiles per group (built once per plan)
static int ensure_transform_table(shg_plan* p, int mt, hipStream_t stream) {
    if (p->ana_trig && p->ana_trig_mt == mt) return SHG_OK;
    const int TR = 4 * mt * 16;
    const int ncol = ceil_div(p->nlon / 4, kAtKC) * kAtKC;
    if (p->ana_trig) {
        SHG_HIP(hipStreamSynchronize(stream));
        (void)hipFree(p->ana_trig);
        p->ana_trig = nullptr;
    }
    if (hipMalloc((void**)&p->ana_trig, (size_t)ncol * TR * sizeof(double)) != hipSuccess) return fail(SHG_ERR_NOMEM, "analysis trig table allocation failed");
    hipLaunchKernelGGL(analysis_trig_kernel, dim3(ceil_div(TR, 128), ncol), dim3(128), 0, stream, p->N, p->nlon, mt, p->cs_slot, p->ana_trig);
    SHG_HIP(hipGetLastError());
    p->ana_trig_mt = mt;
    return SHG_OK;
}

// fold kernel + one batched GEMM per parity (degrees above 126: the fused kernel's accumulators do not fit)
static int folded_transform(shg_plan* p, const double* grid, const double* area, int nb, double* wvt, double* gt, hipStream_t stream) {
    const int N = p->N, nlat = p->nlat, nlon = p->nlon;
    const long long rows = (long long)nb * nlat;
    // slots in gt: 0 = order 0, 2m-1 = cos m, 2m = sin m.  Per parity and cos | sin the slots are 4 apart, so each
    // group is a GEMM on strided rows of the trig table (columns nlon/2 ... of cs_slot = the quarter domain) whose
    // output rows land in slot order; the two groups of a parity differ by one slot and form one batched call.
    const int nq = nlon / 4;
    const long long plane = (long long)nq * rows;
    hipLaunchKernelGGL(weight_fold_kernel, dim3(ceil_div(nq, 32), (unsigned)ceil_div64(rows, 32)), dim3(256), 0, stream, nb, nlat,
               nlon, grid, area, wvt);
    ProfileScope ps(p, 4, stream);
    int rc;
    const double* quarter = p->cs_slot + nlon / 2;
    rc = gemm_ex(false, false, 1, (int)rows, nq, 1.0, quarter, nlon, 0, wvt + 2 * plane, (int)rows, 0, 0.0, gt, (int)rows, 0, 1, false, stream);
    const int odd = (N + 1) / 2, even = N / 2;    // orders 1, 3, ... / 2, 4, ...
    if (!rc && odd)
        rc = gemm_ex(false, false, odd, (int)rows, nq, 1.0, quarter + (size_t)nlon, 4 * nlon, nlon, wvt, (int)rows, plane, 0.0,
             gt + rows, 4 * (int)rows, rows, 2, false, stream);
    if (!rc && even)
        rc = gemm_ex(false, false, even, (int)rows, nq, 1.0, quarter + (size_t)3 * nlon, 4 * nlon, nlon, wvt + 2 * plane, (int)rows,
             plane, 0.0, gt + 3 * rows, 4 * (int)rows, rows, 2, false, stream);
    return rc;
}

// The fused transform kernel for `nb` epochs starting at `values` -> gt [S][nb][nlat] (degrees up to 126, four-fold meridian symmetry)
static int launch_fused_transform(shg_plan* p, const double* values, const double* area, int nb, double* gt, hipStream_t stream) {
    const int N = p->N, nlat = p->nlat, nlon = p->nlon;
    const unsigned blocks = (unsigned)ceil_div64((long long)nb * nlat, kAtRows);
    const int mt = N <= 64 ? 2 : (N <= 96 ? 3 : 4);           // (N + 1) / 2 orders per group at most: 32 | 48 | 64
    const int rc = ensure_transform_table(p, mt, stream);
    if (rc) return rc;
#define SHG_ANA_LAUNCH(MT_, RW_)                                                                                                       \
    hipLaunchKernelGGL((analysis_transform_kernel<MT_, RW_>), dim3(blocks), dim3(256), 0, stream, nb, nlat, nlon, N, values, area, p->ana_trig, gt)
    if (p->ana_rowconst) {
        if (mt == 2) SHG_ANA_LAUNCH(2, true); else if (mt == 3) SHG_ANA_LAUNCH(3, true); else SHG_ANA_LAUNCH(4, true);
    } else {
        if (mt == 2) SHG_ANA_LAUNCH(2, false); else if (mt == 3) SHG_ANA_LAUNCH(3, false); else SHG_ANA_LAUNCH(4, false);
    }
#undef SHG_ANA_LAUNCH
    return SHG_OK;
}

// one pass over all epochs with the operator in p->ana_H (workspaces sized for `chunk` epochs)
static int analysis_pass(shg_plan* p, const double* grid, const double* area, int nmin, int B, int chunk, bool folded, double* wvt, double* gt,
                         double* X, double* anm, hipStream_t stream) {
    const int N = p->N, S = 2 * N + 1, nlat = p->nlat, nlon = p->nlon, R = N + 1;
    // degrees below min_degree stay zero; with min_degree 0 the slots write every entry of the output (C_nm at [n][m], n >= m, and S_nm at
    // [m-1][n], n >= m >= 1, tile the (N+1) x (N+1) array exactly)
    if (nmin > 0)
        hipLaunchKernelGGL(analysis_zero_kernel, dim3((unsigned)std::min<long long>(ceil_div64((long long)B * R * R, 2 * 256), 4096)), dim3(256), 0, stream,
                           (long long)B * R * R, anm);
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = std::min(chunk, B - b0);
        const long long rows = (long long)nb * nlat;
        int rc;
        if (folded) {
            // slots in gt: 0 = order 0, 2m-1 = cos m, 2m = sin m.  Per parity and cos | sin the slots are 4 apart, so each
            // group is a GEMM on strided rows of the trig table (columns nlon/2 ... of cs_slot = the quarter domain) whose
            // output rows land in slot order; the two groups of a parity differ by one slot and form one batched call.
            if (N <= 126) {
                ProfileScope ps(p, 4, stream);
                rc = launch_fused_transform(p, grid + (size_t)b0 * nlat * nlon, area, nb, gt, stream);
            } else {
                rc = folded_transform(p, grid + (size_t)b0 * nlat * nlon, area, nb, wvt, gt, stream);
            }
        } else {
            hipLaunchKernelGGL(weight_transpose_kernel, dim3(ceil_div(nlon, 32), (unsigned)ceil_div64(rows, 32)), dim3(256), 0, stream, nb,
                               nlat, nlon, grid + (size_t)b0 * nlat * nlon, area, wvt);
            ProfileScope ps(p, 4, stream);
            rc = shg_dgemm(S, (int)rows, nlon, p->cs_slot, nlon, wvt, (int)rows, gt, (int)rows, stream);
        }
        if (rc) return rc;
        ProfileScope ps(p, 5, stream);
        if (p->ana_parity && SHG_ANA_PARITY) {
            hipLaunchKernelGGL(analysis_operator_parity_kernel, dim3((unsigned)(8 * ceil_div(S, 8) * ceil_div(nb, kOpCols))), dim3(256), 0, stream, N, nmin, nlat, nb, b0,
                               ceil_div(nb, kOpCols), p->ana_Hp, gt, anm);
        } else if (R <= kOpRows && nlat % 2 == 0) {
            hipLaunchKernelGGL(analysis_operator_kernel, dim3((unsigned)(8 * ceil_div(S, 8) * ceil_div(nb, kOpCols))), dim3(256), 0, stream, N, nmin, nlat, nb, b0,
                               ceil_div(nb, kOpCols), p->ana_H, gt, anm);
        } else {
            // X_s [R][nb] = H_s [R][nlat] gt_s^T   (gt_s is [nb][nlat])
            rc = gemm_ex(false, true, R, nb, nlat, 1.0, p->ana_H, nlat, (long long)R * nlat, gt, nlat, rows, 0.0, X, nb, (long long)R * nb, S, false, stream);
            if (rc) return rc;
            hipLaunchKernelGGL(analysis_scatter_kernel, dim3(ceil_div(nb, 64), R, S), dim3(64), 0, stream, N, nmin, nb, b0, X, anm);
        }
    }
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

extern "C" int shg_analysis(shg_plan* p, const double* grid, const double* area, int nmin, int B, double* anm, void* stream_) {
    SHG_REQUIRE(p != nullptr, "shg_analysis: NULL plan");
    SHG_REQUIRE(B >= 0 && nmin >= 0, "shg_analysis: negative size");
    if (B == 0) return SHG_OK;
    SHG_REQUIRE(grid && anm, "shg_analysis: NULL pointer");
    SHG_REQUIRE(nmin <= p->N, "shg_analysis: min_degree %d out of range", nmin);
    hipStream_t stream = (hipStream_t)stream_;
    PlanGuard guard(p, stream);
    const bool trusted = area == nullptr;               // the weights of the previous call: the plan's own copy, nothing to compare
    if (trusted) {
        SHG_REQUIRE(analysis_operator_cached(p, nmin), "shg_analysis: area == NULL, but the plan holds no operators for min_degree %d", nmin);
        area = p->ana_area;
    }
    const int N = p->N, S = 2 * N + 1, nlat = p->nlat, nlon = p->nlon;
    int rc = analysis_tables(p, stream);
    if (rc) return rc;
    // With a cached operator the pass is queued at once and the comparison of the weights with the ones the operator was
    // built for rides along on the stream: its verdict is read after the pass (the host never waits in the middle of the
    // call), and only weights that did change cost a rebuild and a second pass.
    const bool optimistic = !trusted && analysis_operator_cached(p, nmin);
    if (!optimistic && !trusted && (rc = rebuild_analysis_operator(p, area, nmin, stream)) != SHG_OK) return rc;

    const int chunk = std::min(B, kAnaEpochChunk);
    const int R = N + 1;
    const bool folded = p->sym4 && 4LL * chunk * nlat < (1LL << 29);     // ldc = 4 * rows as int
    // per-stream scratch kept between calls (stream_scratch): a hipFreeAsync of the 0.5 GB fold buffer alone costs 0.2 ms
    ScratchLease lease(stream);
    double* wvt = folded && N <= 126 ? nullptr : (double*)lease.get(kScratchAnaFold, (size_t)nlon * chunk * nlat * sizeof(double));
    double* gt = (double*)lease.get(kScratchAnaTransform, (size_t)S * chunk * nlat * sizeof(double));
    const bool direct = R <= kOpRows && nlat % 2 == 0;          // operator kernel writes anm itself
    double* X = direct ? nullptr : (double*)lease.get(kScratchAnaSolution, (size_t)S * R * chunk * sizeof(double));
    int* diff = (int*)lease.get(kScratchAnaFlag, sizeof(int));
    if ((!wvt && !(folded && N <= 126)) || !gt || (!X && !direct) || !diff) return fail(SHG_ERR_NOMEM, "shg_analysis: workspace allocation failed");
    if (optimistic) rc = launch_weight_compare(p, area, diff, stream);
    if (!rc) rc = analysis_pass(p, grid, area, nmin, B, chunk, folded, wvt, gt, X, anm, stream);
    if (!rc && optimistic) {
        int changed = 0;
        hipError_t e = hipMemcpyAsync(&changed, diff, sizeof(int), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) rc = fail(SHG_ERR_HIP, "shg_analysis: weight comparison failed: %s", hipGetErrorString(e));
        if (!rc && changed) {
            rc = rebuild_analysis_operator(p, area, nmin, stream);
            if (!rc) rc = analysis_pass(p, grid, area, nmin, B, chunk, folded, wvt, gt, X, anm, stream);
        }
    }
    return rc;
}
