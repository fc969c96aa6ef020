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
//   3. weight_squares     w2[s][i]
//   4. analysis_solve     one workgroup per slot s = (m, cos|sin): normal matrix, Cholesky, right-hand sides of all
//                         epochs, forward / backward substitution, scatter into anm.
#include "common.h"

namespace shg {

int covprop_build_cs_table(shg_plan* p, hipStream_t stream);   // gemm.hip

constexpr int kAnaEpochChunk = 64;

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

struct SolveParams {
    int N, nmin, nlat, ldlat, nb, b0, B;
    const double* pk;      // [(m, n)][ldlat]
    const double* w2;      // [S][nlat]
    const double* gt;      // [S][nb * nlat]
    double* wsN;           // [S][(N+1)^2] normal matrices / Cholesky factors
    double* wsR;           // [S][(N+1) * nb] right-hand sides / solutions
    int factor;            // 1: build and factor the normal matrix (first epoch chunk), 0: reuse
    double* anm;           // [B][N+1][N+1]
};

__global__ __launch_bounds__(256) void analysis_solve_kernel(SolveParams P) {
    const int s = blockIdx.x;
    const int m = (s + 1) >> 1;
    const bool sine = s > 0 && (s & 1) == 0;
    const int n0 = max(m, P.nmin);
    const int d = P.N + 1 - n0;
    if (d <= 0) return;
    const int tid = threadIdx.x;
    const double* pk = P.pk + (size_t)(order_offset(P.N, m) + n0 - m) * P.ldlat;       // row a <-> degree n0 + a
    const double* w2 = P.w2 + (size_t)s * P.nlat;
    double* L = P.wsN + (size_t)s * (P.N + 1) * (P.N + 1);                             // [d][d], lower triangle used
    double* R = P.wsR + (size_t)s * (P.N + 1) * P.nb;                                  // [d][nb]

    if (P.factor) {
        // ---- normal matrix, lower triangle:  L[a][c] = sum_i pk[a][i] w2[i] pk[c][i]
        for (int e = tid; e < d * d; e += 256) {
            const int a = e / d, c = e % d;
            if (c > a) continue;
            const double* pa = pk + (size_t)a * P.ldlat;
            const double* pc = pk + (size_t)c * P.ldlat;
            double acc = 0.0;
            for (int i = 0; i < P.nlat; ++i) acc = fma(pa[i] * w2[i], pc[i], acc);
            L[a * d + c] = acc;
        }
        __syncthreads();
        // ---- Cholesky factorisation in place (right-looking)
        for (int k = 0; k < d; ++k) {
            if (tid == 0) L[k * d + k] = sqrt(L[k * d + k]);
            __syncthreads();
            const double piv = L[k * d + k];
            for (int r = k + 1 + tid; r < d; r += 256) L[r * d + k] /= piv;
            __syncthreads();
            const int t = d - k - 1;
            for (int e = tid; e < t * t; e += 256) {
                const int r = k + 1 + e / t, c = k + 1 + e % t;
                if (c <= r) L[r * d + c] = fma(-L[r * d + k], L[c * d + k], L[r * d + c]);
            }
            __syncthreads();
        }
    }
    // ---- right-hand sides of all epochs of the chunk:  R[a][b] = sum_i pk[a][i] gt[s][b][i]
    const double* gt = P.gt + (size_t)s * P.nb * P.nlat;
    for (int e = tid; e < d * P.nb; e += 256) {
        const int a = e / P.nb, b = e % P.nb;
        const double* pa = pk + (size_t)a * P.ldlat;
        const double* g = gt + (size_t)b * P.nlat;
        double acc = 0.0;
        for (int i = 0; i < P.nlat; ++i) acc = fma(pa[i], g[i], acc);
        R[a * P.nb + b] = acc;
    }
    __syncthreads();
    // ---- L y = r, L^T x = y: one thread per epoch
    for (int b = tid; b < P.nb; b += 256) {
        for (int k = 0; k < d; ++k) {
            double acc = R[k * P.nb + b];
            for (int j = 0; j < k; ++j) acc = fma(-L[k * d + j], R[j * P.nb + b], acc);
            R[k * P.nb + b] = acc / L[k * d + k];
        }
        for (int k = d - 1; k >= 0; --k) {
            double acc = R[k * P.nb + b];
            for (int j = k + 1; j < d; ++j) acc = fma(-L[j * d + k], R[j * P.nb + b], acc);
            R[k * P.nb + b] = acc / L[k * d + k];
        }
        double* out = P.anm + (size_t)(P.b0 + b) * (P.N + 1) * (P.N + 1);
        for (int a = 0; a < d; ++a) {
            const int n = n0 + a;
            out[sine ? (size_t)(m - 1) * (P.N + 1) + n : (size_t)n * (P.N + 1) + m] = R[a * P.nb + b];
        }
    }
}

}  // namespace shg

using namespace shg;

extern "C" int shg_analysis(shg_plan* p, const double* grid, const double* area, int nmin, int B, double* anm, void* stream_) {
    SHG_REQUIRE(p != nullptr, "shg_analysis: NULL plan");
    SHG_REQUIRE(B >= 0 && nmin >= 0, "shg_analysis: negative size");
    if (B == 0) return SHG_OK;
    SHG_REQUIRE(grid && area && anm, "shg_analysis: NULL pointer");
    hipStream_t stream = (hipStream_t)stream_;
    const int N = p->N, S = 2 * N + 1, nlat = p->nlat, nlon = p->nlon;
    int rc = build_pk_table(p, stream);
    if (rc) return rc;
    rc = covprop_build_cs_table(p, stream);
    if (rc) return rc;
    SHG_HIP(hipMemsetAsync(anm, 0, (size_t)B * (N + 1) * (N + 1) * sizeof(double), stream));

    const int chunk = std::min(B, kAnaEpochChunk);
    double *wvt = nullptr, *gt = nullptr, *w2 = nullptr, *wsN = nullptr, *wsR = nullptr;
    if (hipMallocAsync((void**)&wvt, (size_t)nlon * chunk * nlat * sizeof(double), stream) != hipSuccess ||
        hipMallocAsync((void**)&gt, (size_t)S * chunk * nlat * sizeof(double), stream) != hipSuccess ||
        hipMallocAsync((void**)&w2, (size_t)S * nlat * sizeof(double), stream) != hipSuccess ||
        hipMallocAsync((void**)&wsN, (size_t)S * (N + 1) * (N + 1) * sizeof(double), stream) != hipSuccess ||
        hipMallocAsync((void**)&wsR, (size_t)S * (N + 1) * chunk * sizeof(double), stream) != hipSuccess)
        return fail(SHG_ERR_NOMEM, "shg_analysis: workspace allocation failed");
    hipLaunchKernelGGL(weight_squares_kernel, dim3(nlat), dim3(256), 0, stream, S, nlat, nlon, area, p->cs_slot, w2);

    for (int b0 = 0; b0 < B && rc == SHG_OK; b0 += chunk) {
        const int nb = std::min(chunk, B - b0);
        const long long rows = (long long)nb * nlat;
        hipLaunchKernelGGL(weight_transpose_kernel, dim3(ceil_div(nlon, 32), (unsigned)ceil_div64(rows, 32)), dim3(256), 0, stream, nb,
                           nlat, nlon, grid + (size_t)b0 * nlat * nlon, area, wvt);
        {
            ProfileScope ps(p, 4, stream);
            rc = shg_dgemm(S, (int)rows, nlon, p->cs_slot, nlon, wvt, (int)rows, gt, (int)rows, stream);
        }
        if (rc) break;
        SolveParams Q;
        Q.N = N;
        Q.nmin = nmin;
        Q.nlat = nlat;
        Q.ldlat = p->ldlat;
        Q.nb = nb;
        Q.b0 = b0;
        Q.B = B;
        Q.pk = p->pk;
        Q.w2 = w2;
        Q.gt = gt;
        Q.wsN = wsN;
        Q.wsR = wsR;
        Q.factor = b0 == 0 ? 1 : 0;
        Q.anm = anm;
        ProfileScope ps(p, 5, stream);
        hipLaunchKernelGGL(analysis_solve_kernel, dim3(S), dim3(256), 0, stream, Q);
    }
    (void)hipFreeAsync(wvt, stream);
    (void)hipFreeAsync(gt, stream);
    (void)hipFreeAsync(w2, stream);
    (void)hipFreeAsync(wsN, stream);
    (void)hipFreeAsync(wsR, stream);
    if (rc) return rc;
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}
