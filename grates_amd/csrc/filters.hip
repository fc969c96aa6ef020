// Order-wise block filter (DDK family):  replaces OrderWiseFilter.filter, grates/filter.py:180-191.
//   for every order m and cosine / sine:  y[m..N] = W_block[0:N+1-m, 0:N+1-m] x[m..N]   for all epochs at once,
//   degrees 0 and 1 copied from the input.
// HBM-bound integer-indexed streaming: each block matrix is read once per tile of 32 epochs, coefficient
// vectors are staged in LDS.
#include "common.h"

namespace shg {

constexpr int kFiltEpochs = 64;     // epochs per workgroup = lanes of a wave

// One workgroup = one order block x 64 epochs, 16 waves (the work per block is small and made of dependent scalar-load /
// LDS round trips: many waves per SIMD hide them).  The coefficient vectors of the 64 epochs are gathered into LDS
// (xs[k][epoch]); every wave then takes groups of four block rows: the row index is wave-uniform, so the block entries come
// through scalar loads and feed the FMAs as scalar operands, and one LDS read of xs[c][epoch] serves four FMAs.
// HBM/L2 bound: bytes = 8 (sum of block sizes x epoch groups + 2 P B).
__global__ __launch_bounds__(1024) void orderwise_filter_kernel(int Nb, int N, int B, const double* __restrict__ blocks,
                                                               const long long* __restrict__ block_off,
                                                               const double* __restrict__ in, double* __restrict__ out) {
    extern __shared__ double xs[];                 // [n][kFiltEpochs]
    const int kb = blockIdx.x;                     // 0: order 0 cos, 2m-1: order m cos, 2m: order m sin
    const int m = (kb + 1) >> 1;
    const bool sine = kb > 0 && (kb & 1) == 0;
    const int n = N + 1 - m;                       // coefficients of this order in the field
    const int ld = Nb + 1 - m;                     // leading dimension of the stored block
    const int e = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y * kFiltEpochs + e;
    const size_t E = (size_t)(N + 1) * (N + 1);
    const double* W = blocks + block_off[kb];

    // element k (degree m + k) of this order: cos at [m+k][m], sin at [m-1][m+k]
    auto pos = [&](int k) -> size_t { return sine ? (size_t)(m - 1) * (N + 1) + (m + k) : (size_t)(m + k) * (N + 1) + m; };
    // gather: 8 elements per thread in flight at a time (the addresses are a whole coefficient array apart from lane to lane,
    // so every element is its own memory transaction: a loop that waits for each one is bound by their latency)
    for (int k0 = 0; k0 < n; k0 += 128) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + wave + 16 * j;                     // indices clamped, not tested: no branch around the loads
            v[j] = in[(size_t)min(b, B - 1) * E + pos(min(k, n - 1))];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + wave + 16 * j;
            if (k < n) xs[k * kFiltEpochs + e] = b < B ? v[j] : 0.0;
        }
    }
    __syncthreads();
    for (int r0 = 4 * wave; r0 < n; r0 += 64) {       // rows r0 .. r0 + 3 (clamped: a duplicate row is computed, not stored)
        const double* w0 = W + (size_t)r0 * ld;
        const double* w1 = W + (size_t)min(r0 + 1, n - 1) * ld;
        const double* w2 = W + (size_t)min(r0 + 2, n - 1) * ld;
        const double* w3 = W + (size_t)min(r0 + 3, n - 1) * ld;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int c = 0;
        for (; c + 4 <= n; c += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double x = xs[(c + u) * kFiltEpochs + e];
                s0 = fma(w0[c + u], x, s0);
                s1 = fma(w1[c + u], x, s1);
                s2 = fma(w2[c + u], x, s2);
                s3 = fma(w3[c + u], x, s3);
            }
        }
        for (; c < n; ++c) {
            const double x = xs[c * kFiltEpochs + e];
            s0 = fma(w0[c], x, s0);
            s1 = fma(w1[c], x, s1);
            s2 = fma(w2[c], x, s2);
            s3 = fma(w3[c], x, s3);
        }
        if (b < B) {
            const double s[4] = {s0, s1, s2, s3};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = r0 + i;
                if (r < n) out[(size_t)b * E + pos(r)] = (m + r <= 1) ? xs[r * kFiltEpochs + e] : s[i];    // filter.py:189
            }
        }
    }
}

}  // namespace shg

using namespace shg;

extern "C" int shg_orderwise_filter(const double* blocks_packed, const int64_t* block_off, int Nb, int N, const double* anm_in, int B,
                                    double* anm_out, void* stream_) {
    SHG_REQUIRE(Nb >= 0 && N >= 0 && B >= 0, "shg_orderwise_filter: negative size");
    SHG_REQUIRE(N <= Nb, "DDK filter only implemented for a maximum degree of %d (max_degree=%d supplied).", Nb, N);
    if (B == 0) return SHG_OK;
    SHG_REQUIRE(blocks_packed && block_off && anm_in && anm_out, "shg_orderwise_filter: NULL pointer");
    SHG_REQUIRE(anm_in != anm_out, "shg_orderwise_filter: in-place operation is not supported");
    const size_t lds = (size_t)(N + 1) * kFiltEpochs * sizeof(double);
    // the coefficient vectors of an order for 64 epochs sit in LDS: 64 KB by default, up to the 160 KB of a CU on request
    SHG_REQUIRE(lds <= 160 * 1024, "shg_orderwise_filter: degree %d exceeds the LDS staging of the block kernel (max degree 319)", N);
    if (lds > 64 * 1024) SHG_HIP(hipFuncSetAttribute((const void*)orderwise_filter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(orderwise_filter_kernel, dim3(2 * N + 1, ceil_div(B, kFiltEpochs)), dim3(1024), lds, (hipStream_t)stream_, Nb, N, B,
                       blocks_packed, (const long long*)block_off, anm_in, anm_out);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

// ------------------------------------------------------------------------------------------------
// DDK block construction:  W_k = (N_k + diag(w[m:]))^-1 N_k  for every order-wise normal block
// (replaces the 2 Nb + 1 dense solves of DDK.__init__ / DDKGeneric.__init__, grates/filter.py:252-255, 344-347).
// N_k + D is symmetric positive definite (normal matrix plus positive power-law weights): one workgroup per
// block does an in-place Cholesky factorisation and 2 triangular solves with all columns of N_k as right-hand sides.
// ------------------------------------------------------------------------------------------------
namespace shg {

// in-place right-looking Cholesky of the lower triangle of L [d][d] by one workgroup of 256 threads
__device__ void cholesky_inplace(double* L, int d, int tid) {
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

// X <- (L L^T)^-1 X for the ncol columns of X [d][ldx]: one thread per column
__device__ void cholesky_solve_columns(const double* L, double* X, int d, int ncol, int ldx, int tid) {
    for (int col = tid; col < ncol; col += 256) {
        for (int k = 0; k < d; ++k) {
            double acc = X[(size_t)k * ldx + col];
            for (int j = 0; j < k; ++j) acc = fma(-L[k * d + j], X[(size_t)j * ldx + col], acc);
            X[(size_t)k * ldx + col] = acc / L[k * d + k];
        }
        for (int k = d - 1; k >= 0; --k) {
            double acc = X[(size_t)k * ldx + col];
            for (int j = k + 1; j < d; ++j) acc = fma(-L[j * d + k], X[(size_t)j * ldx + col], acc);
            X[(size_t)k * ldx + col] = acc / L[k * d + k];
        }
    }
}

__global__ __launch_bounds__(256) void ddk_blocks_kernel(int Nb, const double* __restrict__ normals, const long long* __restrict__ off,
                                                         const double* __restrict__ weights, double* __restrict__ work,
                                                         double* __restrict__ out) {
    const int kb = blockIdx.x;
    const int m = (kb + 1) >> 1;
    const int d = Nb + 1 - m;
    const int tid = threadIdx.x;
    const double* Nk = normals + off[kb];
    double* L = work + off[kb];
    double* X = out + off[kb];
    for (int e = tid; e < d * d; e += 256) {
        const int r = e / d, c = e % d;
        L[e] = Nk[e] + (r == c ? weights[m + r] : 0.0);
        X[e] = Nk[e];
    }
    __syncthreads();
    cholesky_inplace(L, d, tid);
    cholesky_solve_columns(L, X, d, d, d, tid);
}

// X = A^-1 B for one symmetric positive definite A [n][n] and B [n][k]; column tiles of B go to different workgroups,
// each of which factors its own copy of A (small systems: least-squares normal equations of point lists).
__global__ __launch_bounds__(256) void spd_solve_kernel(int n, int k, const double* __restrict__ A, const double* __restrict__ Bm,
                                                        double* __restrict__ work, double* __restrict__ X) {
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * 256;
    const int nc = min(256, k - c0);
    double* L = work + (size_t)blockIdx.x * n * n;
    for (int e = tid; e < n * n; e += 256) L[e] = A[e];
    for (int e = tid; e < n * nc; e += 256) {
        const int r = e / nc, c = e % nc;
        X[(size_t)r * k + c0 + c] = Bm[(size_t)r * k + c0 + c];
    }
    __syncthreads();
    cholesky_inplace(L, n, tid);
    cholesky_solve_columns(L, X + c0, n, nc, k, tid);
}

}  // namespace shg

extern "C" int shg_ddk_blocks(const double* normals_packed, const int64_t* block_off, int Nb, const double* weights, double* work,
                              double* blocks_out, void* stream_) {
    SHG_REQUIRE(Nb >= 0, "shg_ddk_blocks: negative degree");
    SHG_REQUIRE(normals_packed && block_off && weights && work && blocks_out, "shg_ddk_blocks: NULL pointer");
    hipLaunchKernelGGL(shg::ddk_blocks_kernel, dim3(2 * Nb + 1), dim3(256), 0, (hipStream_t)stream_, Nb, normals_packed,
                       (const long long*)block_off, weights, work, blocks_out);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

extern "C" int shg_spd_solve(const double* A, int n, const double* Bm, int k, double* X, void* stream_) {
    SHG_REQUIRE(n >= 0 && k >= 0, "shg_spd_solve: negative size");
    if (n == 0 || k == 0) return SHG_OK;
    SHG_REQUIRE(A && Bm && X, "shg_spd_solve: NULL pointer");
    hipStream_t stream = (hipStream_t)stream_;
    const int nblocks = ceil_div(k, 256);
    double* work = nullptr;
    if (hipMallocAsync((void**)&work, (size_t)nblocks * n * n * sizeof(double), stream) != hipSuccess)
        return fail(SHG_ERR_NOMEM, "shg_spd_solve: workspace allocation failed");
    hipLaunchKernelGGL(shg::spd_solve_kernel, dim3(nblocks), dim3(256), 0, stream, n, k, A, Bm, work, X);
    (void)hipFreeAsync(work, stream);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}
