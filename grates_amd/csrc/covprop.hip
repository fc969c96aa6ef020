// Covariance propagation on a regular grid, fp64 MFMA:  sigma^2(i, j) = a_ij^T Sigma a_ij        (grates/grid.py:817-839)
//
// Row tiles never cross a parallel: all 128 rows of a tile are meridians j0..j0+127 of ONE parallel i, so the rows of
// the synthesis matrix factor as  A[(i, j)][p] = PK[i][p] * CS[rank(p)][j]  with a tile-uniform PK.  The kernel uses
//     (A Sigma)[r][c] = sum_p CS[rank(p)][j_r] * (PK[i][p] Sigma[p][c])
// i.e. the A operand tile is 16 contiguous 1 KB rows of the small cos/sin table (no gathers, no multiplies) and PK is
// folded into the Sigma tile while it is staged in LDS.  Both operand tiles are k-major [16][128 + 16 pad] and every
// MFMA fragment read is conflict free.  Tile 128 x 128, BK = 16, 4 waves x (64 x 64), 2 workgroups per CU,
// register prefetch of the next K tile, one barrier per K tile.  Epilogue: row-dot with the regenerated A tile, reduced over
// the 128 columns; per-column-block partial sums are summed in a fixed order by covprop_reduce_kernel (gemm.hip).
#include "common.h"

namespace shg {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int CT = 128;          // tile edge
constexpr int CK = 16;           // K slots per stage
constexpr int CLD = 144;         // k-major LDS rows: 128 + 16 pad -> the 4 k-rows of a fragment read use disjoint banks

struct CovParams {
    int P, p_off, nlon, ldp, lat0, tiles_per_parallel, M;   // P = size of Sigma, M = band rows = nparallels * nlon
    const double* cov;        // [P][P]
    const double* pkd;        // [nlat][ldp]  kn P_nm in degree-wise order (min_degree 0)
    const double* csr;        // [2N+1][nlon]
    double* partial;          // [column blocks][M]
};

__device__ inline void degree_rank(int pf, int& n, int& r) {      // degree-wise index -> (degree, rank inside the degree)
    n = (int)sqrt((double)pf);
    while ((n + 1) * (n + 1) <= pf) ++n;
    while (n * n > pf) --n;
    r = pf - n * n;
}

__global__ __launch_bounds__(256, 2) void covprop_rows_kernel(CovParams P) {
    extern __shared__ double cov_lds[];
    double (*As)[CK * CLD] = reinterpret_cast<double (*)[CK * CLD]>(cov_lds);                   // [2][CK * CLD]
    double (*Bs)[CK * CLD] = reinterpret_cast<double (*)[CK * CLD]>(cov_lds + 2 * CK * CLD);    // [2][CK * CLD]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fk = lane >> 4;
    // row tile fastest: the workgroups resident at one time walk the same column panel of Sigma together
    const int i = P.lat0 + blockIdx.x / P.tiles_per_parallel;
    const int j0 = (blockIdx.x % P.tiles_per_parallel) * CT;
    const int n0 = blockIdx.y * CT;
    const double* pk = P.pkd + (size_t)i * P.ldp + P.p_off;

    // staging: a stage is 16 k-rows x 128 doubles = 1024 pieces of 2 doubles; piece h of a thread: k = (tid >> 6) + 4 h,
    // col = (tid & 63) * 2.  Column indices beyond the table / matrix are clamped (their products only reach rows /
    // columns that are never stored), so full K tiles are fetched without any bounds test.
    const int st_col = (tid & 63) * 2;
    const int st_k = tid >> 6;
    const double* a0 = P.csr + min(j0 + st_col, P.nlon - 1);
    const double* a1 = P.csr + min(j0 + st_col + 1, P.nlon - 1);
    const double* b0 = P.cov + min(n0 + st_col, P.P - 1);
    const double* b1 = P.cov + min(n0 + st_col + 1, P.P - 1);
    double2 areg[4], breg[4];
    // (degree, rank) of the degree-wise index of each of this thread's 4 k-rows, advanced by 16 per K tile
    int deg[4], rank[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) degree_rank(st_k + 4 * h + P.p_off, deg[h], rank[h]);
    auto fetch = [&](int k0, bool tail) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int k = k0 + st_k + 4 * h;
            const int kc = tail ? min(k, P.P - 1) : k;
            const int rc = tail && k >= P.P ? 0 : rank[h];
            const size_t ra = (size_t)rc * P.nlon, rb = (size_t)kc * P.P;
            const double s = pk[kc];                   // PK[i][p] folded into the Sigma tile
            double2 va = make_double2(a0[ra], a1[ra]);
            double2 vb = make_double2(b0[rb] * s, b1[rb] * s);
            if (tail && k >= P.P) {
                va = make_double2(0.0, 0.0);
                vb = make_double2(0.0, 0.0);
            }
            areg[h] = va;
            breg[h] = vb;
            rank[h] += CK;                             // next K tile
            while (rank[h] > 2 * deg[h]) {
                rank[h] -= 2 * deg[h] + 1;
                ++deg[h];
            }
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            *reinterpret_cast<double2*>(&As[buf][(st_k + 4 * h) * CLD + st_col]) = areg[h];
            *reinterpret_cast<double2*>(&Bs[buf][(st_k + 4 * h) * CLD + st_col]) = breg[h];
        }
    };

    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};

    auto compute = [&](int buf) {
        const double* Ab = As[buf] + fk * CLD + wr * 64 + fr;
        const double* Bb = Bs[buf] + fk * CLD + wc * 64 + fr;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            double af[4], bf[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) af[a] = Ab[ks * 4 * CLD + a * 16];
#pragma unroll
            for (int b = 0; b < 4; ++b) bf[b] = Bb[ks * 4 * CLD + b * 16];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
    };
    const int nfull = P.P / CK;
    const bool has_tail = (P.P % CK) != 0;
    fetch(0, nfull == 0);
    stage(0);
    __syncthreads();
    for (int t = 0; t + 1 < nfull; ++t) {             // branch-free steady state
        fetch((t + 1) * CK, false);
        compute(t & 1);
        stage((t + 1) & 1);
        __syncthreads();
    }
    if (nfull > 0) {
        if (has_tail) fetch(nfull * CK, true);
        compute((nfull - 1) & 1);
        if (has_tail) stage(nfull & 1);
        __syncthreads();
    }
    if (has_tail) compute(nfull & 1);
    __syncthreads();

    // ---- epilogue: sum_c (A Sigma)[r][c] * A[r][c] over the 128 columns of this block.  C/D layout: column = lane & 15,
    //      row = (lane >> 4) + 4 * reg.  A[r][c] = PK[i][c] * CS[rank(c)][j_r]
    double* red = As[0];                               // [128 rows][2 column halves]
    double colpk[4];
    const double* colcs[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int c = n0 + wc * 64 + b * 16 + fr;
        const bool ok = c < P.P;
        int n, r;
        degree_rank((ok ? c : 0) + P.p_off, n, r);
        colpk[b] = ok ? pk[c] : 0.0;
        colcs[b] = P.csr + (size_t)r * P.nlon;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wr * 64 + a * 16 + fk + 4 * r;
            const int j = min(j0 + row, P.nlon - 1);
            double s = 0.0;
#pragma unroll
            for (int b = 0; b < 4; ++b) s = fma(acc[a][b][r], colpk[b] * colcs[b][j], s);
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            s += __shfl_xor(s, 8);
            if (fr == 0) red[row * 2 + wc] = s;
        }
    __syncthreads();
    if (tid < CT && j0 + tid < P.nlon)
        P.partial[(size_t)blockIdx.y * P.M + (size_t)(i - P.lat0) * P.nlon + j0 + tid] = red[tid * 2] + red[tid * 2 + 1];
}

int covprop_rows(shg_plan* p, const double* cov, int Pn, int p_off, int lat0, int lat1, double* partial, hipStream_t stream) {
    CovParams C;
    C.P = Pn;
    C.p_off = p_off;
    C.nlon = p->nlon;
    C.ldp = (p->N + 1) * (p->N + 1);
    C.lat0 = lat0;
    C.tiles_per_parallel = ceil_div(p->nlon, CT);
    C.M = (lat1 - lat0) * p->nlon;
    C.cov = cov;
    C.pkd = p->pk_deg;
    C.csr = p->cs_slot;
    C.partial = partial;
    const size_t lds = (size_t)4 * CK * CLD * sizeof(double);            // 73.7 KB: two workgroups per CU
    SHG_HIP(hipFuncSetAttribute((const void*)covprop_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid((unsigned)((lat1 - lat0) * C.tiles_per_parallel), (unsigned)ceil_div(Pn, CT));
    ProfileScope ps(p, 3, stream);
    hipLaunchKernelGGL(covprop_rows_kernel, grid, dim3(256), lds, stream, C);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

}  // namespace shg
