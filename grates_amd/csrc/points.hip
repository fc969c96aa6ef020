// Point-list paths (irregular grids): every point carries its own colatitude, longitude and kn row.
//   shg_synthesis_points   replaces grates/gravityfield.py:370-388 (blocks of 512 points, spherical_harmonics + dgemv)
//   shg_covprop_points     replaces grates/grid.py:1096-1120       (blocks of 256 points, F Sigma F^T diagonal)
// Synthesis: lane <-> point, column recursion of P_nm in registers, coefficients of an order staged in LDS,
// 16 epochs per pass.  Covariance: per-point tables feed the same fp64 MFMA kernel as the regular grid.
#include "common.h"

namespace shg {

void recursion_tables(int N, std::vector<double>& a, std::vector<double>& b);   // plan.hip

constexpr int kPtEpochs = 16;

// dst[c][r] = src[r][c] through 32 x 32 LDS tiles (coalesced on both sides).  Used for knT[n][pt] = kn[pt][n] -- inside the
// recursion kernel consecutive lanes are consecutive points, so the degree factors are read along the points (with the
// point-major layout of the interface every lane would touch its own cache line, N + 1 times over) -- and for the operand /
// result layouts of the GEMM path.
__global__ __launch_bounds__(256) void transpose_kernel(int rows, int cols, const double* __restrict__ src, size_t ld_src,
                                                        double* __restrict__ dst, size_t ld_dst) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < rows && c < cols) ? src[(size_t)r * ld_src + c] : 0.0;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (c < cols && r < rows) dst[(size_t)c * ld_dst + r] = tile[tx][k];
    }
}

// One wave = 64 points x up to 16 epochs.  For every order m the coefficients C_nm, S_nm (n = m..N) of the 16 epochs are first
// gathered into LDS with ordinary vector loads (many in flight; fetched through wave-uniform scalar loads instead, 32 dependent
// cache misses per (n, m) made the kernel 80 times slower), then the degree loop reads them as LDS broadcasts.
__global__ __launch_bounds__(64) void synthesis_points_kernel(int N, int npts, int B, const double* __restrict__ colat,
                                                              const double* __restrict__ lon, const double* __restrict__ knT,
                                                              const double* __restrict__ arec, const double* __restrict__ brec,
                                                              const double* __restrict__ anm, double* __restrict__ values) {
    extern __shared__ double slab[];                 // [N + 1 - m][2][kPtEpochs]
    const int lane = threadIdx.x;
    const int pt = blockIdx.x * 64 + lane;
    const int b0 = blockIdx.y * kPtEpochs;
    const bool ok = pt < npts;
    const int q = ok ? pt : 0;
    const double th = colat[q], lam = lon[q];
    const double t = cos(th), st = sin(th);
    const double* knp = knT + q;                                 // + n * npts
    const size_t E = (size_t)(N + 1) * (N + 1);
    double acc[kPtEpochs];
#pragma unroll
    for (int bb = 0; bb < kPtEpochs; ++bb) acc[bb] = 0.0;
    double pmm = 1.0;
    for (int m = 0; m <= N; ++m) {
        const int cnt = N + 1 - m;
        __syncthreads();                                         // the previous order's slab has been consumed
        for (int idx = lane; idx < cnt * 2 * kPtEpochs; idx += 64) {
            const int e = idx % kPtEpochs, cs = (idx / kPtEpochs) & 1, k = idx / (2 * kPtEpochs);
            const double* a = anm + (size_t)min(b0 + e, B - 1) * E;
            double v = 0.0;
            if (cs == 0)
                v = a[(size_t)(m + k) * (N + 1) + m];
            else if (m >= 1)
                v = a[(size_t)(m - 1) * (N + 1) + m + k];
            slab[idx] = v;
        }
        __syncthreads();
        if (m == 1)
            pmm = sqrt(3.0) * st;
        else if (m >= 2)
            pmm = sqrt((2.0 * m + 1.0) / (2.0 * m)) * st * pmm;
        const double arg = (double)m * lam;
        const double cm = cos(arg), sm = sin(arg);
        double p1 = pmm, p2 = 0.0;
        const int off = order_offset(N, m);
        for (int n = m; n <= N; ++n) {
            if (n > m) {
                const double p = (arec[off + n - m] * t) * p1 - brec[off + n - m] * p2;     // wave-uniform table entries
                p2 = p1;
                p1 = p;
            }
            const double pk = p1 * knp[(size_t)n * npts];
            const double yc = pk * cm, ys = pk * sm;
            const double* c = slab + (size_t)(n - m) * 2 * kPtEpochs;
#pragma unroll
            for (int bb = 0; bb < kPtEpochs; ++bb) {
                acc[bb] = fma(yc, c[bb], acc[bb]);
                acc[bb] = fma(ys, c[kPtEpochs + bb], acc[bb]);
            }
        }
    }
    if (ok) {
#pragma unroll
        for (int bb = 0; bb < kPtEpochs; ++bb)
            if (b0 + bb < B) values[(size_t)(b0 + bb) * npts + pt] = acc[bb];
    }
}

// per-point tables of the generated-operand GEMM (point-list covariance propagation and synthesis of long series):
// pkT[p][pt] (degree-wise index p, min_degree 0; lanes = points, so every store runs along the points), csr[r][pt], rslot[p]
__global__ void synth_point_tables_kernel(int N, int npts, const double* __restrict__ colat, const double* __restrict__ lon,
                                          const double* __restrict__ knT, size_t ld_kn, const double* __restrict__ arec,
                                          const double* __restrict__ brec, double* __restrict__ pkT, double* __restrict__ csr,
                                          int* __restrict__ rslot) {
    const int pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= npts) return;
    const double th = colat[pt], lam = lon[pt];
    const double t = cos(th), st = sin(th);
    double pmm = 1.0;
    for (int m = 0; m <= N; ++m) {
        if (m == 1)
            pmm = sqrt(3.0) * st;
        else if (m >= 2)
            pmm = sqrt((2.0 * m + 1.0) / (2.0 * m)) * st * pmm;
        const double arg = (double)m * lam;
        if (m == 0) {
            csr[pt] = 1.0;
        } else {
            csr[(size_t)(2 * m - 1) * npts + pt] = cos(arg);
            csr[(size_t)(2 * m) * npts + pt] = sin(arg);
        }
        double p1 = pmm, p2 = 0.0;
        const int off = order_offset(N, m);
        for (int n = m; n <= N; ++n) {
            if (n > m) {
                const double p = (arec[off + n - m] * t) * p1 - brec[off + n - m] * p2;     // host-built factors, as everywhere
                p2 = p1;
                p1 = p;
            }
            const double pk = p1 * knT[(size_t)n * ld_kn + pt];
            const int base = n * n;
            if (m == 0) {
                pkT[(size_t)base * npts + pt] = pk;
            } else {
                pkT[(size_t)(base + 2 * m - 1) * npts + pt] = pk;
                pkT[(size_t)(base + 2 * m) * npts + pt] = pk;
            }
            if (pt == 0) {
                if (m == 0) {
                    rslot[base] = 0;
                } else {
                    rslot[base + 2 * m - 1] = 2 * m - 1;
                    rslot[base + 2 * m] = 2 * m;
                }
            }
        }
    }
}

// A[pt][p - p0] = pkT[p][pt] csr[rslot[p]][pt] through 32 x 32 LDS tiles: reads run along the points, writes along the coefficients
__global__ __launch_bounds__(256) void synthesis_matrix_kernel(int npts, int Pn, int p0, const double* __restrict__ pkT, const double* __restrict__ csr,
                                                               const int* __restrict__ rslot, double* __restrict__ A, size_t lda) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    const int q0 = blockIdx.x * 32, t0 = blockIdx.y * 32;         // coefficient / point tile
    for (int k = ty; k < 32; k += 8) {
        const int q = q0 + k, pt = t0 + tx;
        double v = 0.0;
        if (q < Pn && pt < npts) v = pkT[(size_t)(p0 + q) * npts + pt] * csr[(size_t)rslot[p0 + q] * npts + pt];
        tile[k][tx] = v;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int pt = t0 + k, q = q0 + tx;
        if (pt < npts && q < Pn) A[(size_t)pt * lda + q] = tile[tx][k];
    }
}

int synth_generic(const double* pkd, int ldp, const double* csr, int ldcs, const int* rslot, long long idiv, long long jmod, int M,
                  const double* X, int K, int N, double* C, hipStream_t stream);      // gemm.hip

int covprop_generic(const double* pkd, int ldp, const double* csr, int ldcs, const int* rslot, long long idiv, long long jmod,
                    long long row0, int M, const double* cov, int Pn, int p_off, double* partial, double* sigma, shg_plan* prof,
                    hipStream_t stream, bool symmetric, bool transposed_table, const unsigned* csoff = nullptr, int pk_rows = 0, int ldcov = 0);

}  // namespace shg

using namespace shg;

// points per pass: 2 GB of Legendre table, and at most 65535 point tiles of 32 (grid.y of the generation and transpose kernels)
static int point_chunk(int npts, int Pfull) {
    const long long by_memory = std::max<long long>(128, ((1LL << 31) / 8 / Pfull) / 128 * 128);
    return (int)std::min<long long>(std::min<long long>(npts, 65535LL * 32), by_memory);
}

extern "C" int shg_synthesis_points(int N, const double* colat, const double* lon, const double* kn, int npts, const double* anm,
                                    int B, double* values, void* stream_) {
    SHG_REQUIRE(N >= 0 && npts >= 0 && B >= 0, "shg_synthesis_points: negative size");
    if (npts == 0 || B == 0) return SHG_OK;
    SHG_REQUIRE(colat && lon && kn && anm && values, "shg_synthesis_points: NULL pointer");
    hipStream_t stream = (hipStream_t)stream_;
    std::vector<double> a, b;
    recursion_tables(N, a, b);                       // same host-built factors as the plans (reference expression order)
    double* tab = nullptr;
    if (hipMallocAsync((void**)&tab, 2 * a.size() * sizeof(double), stream) != hipSuccess) return fail(SHG_ERR_NOMEM, "shg_synthesis_points: table allocation failed");
    SHG_HIP(hipMemcpyAsync(tab, a.data(), a.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    SHG_HIP(hipMemcpyAsync(tab + a.size(), b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    SHG_HIP(hipStreamSynchronize(stream));           // the host vectors go out of scope
    double* knT = nullptr;
    if (hipMallocAsync((void**)&knT, (size_t)(N + 1) * npts * sizeof(double), stream) != hipSuccess) {
        (void)hipFreeAsync(tab, stream);
        return fail(SHG_ERR_NOMEM, "shg_synthesis_points: table allocation failed");
    }
    hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(npts, 32), ceil_div(N + 1, 32)), dim3(256), 0, stream, npts, N + 1, kn, (size_t)(N + 1), knT, (size_t)npts);
    // Many epochs: values = Y X as one fp64 MFMA GEMM per chunk of points, with the rows of the spherical harmonic matrix
    // Y[pt][p] = PK[p][pt] cs[rank(p)][pt] generated inside the kernel from per-point tables (MODE_SYNTH of gemm.hip): the
    // recursion runs once per point instead of once per point and group of 16 epochs.
    if (B >= 48) {
        const int Pfull = (N + 1) * (N + 1);
        const int chunk = point_chunk(npts, Pfull);
        double *X = nullptr, *xr = nullptr, *pkT = nullptr, *csr = nullptr, *Cc = nullptr;
        int* rslot = nullptr;
        int rc = SHG_OK;
        if (hipMallocAsync((void**)&xr, (size_t)B * Pfull * sizeof(double), stream) != hipSuccess ||
            hipMallocAsync((void**)&X, (size_t)B * Pfull * sizeof(double), stream) != hipSuccess ||
            hipMallocAsync((void**)&pkT, (size_t)chunk * Pfull * sizeof(double), stream) != hipSuccess ||
            hipMallocAsync((void**)&csr, (size_t)(2 * N + 1) * chunk * sizeof(double), stream) != hipSuccess ||
            hipMallocAsync((void**)&Cc, (size_t)chunk * B * sizeof(double), stream) != hipSuccess ||
            hipMallocAsync((void**)&rslot, (size_t)Pfull * sizeof(int), stream) != hipSuccess)
            rc = fail(SHG_ERR_NOMEM, "shg_synthesis_points: workspace allocation failed");
        if (rc == SHG_OK) rc = shg_ravel(anm, B, N, 0, N, xr, stream);
        if (rc == SHG_OK) {
            hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(B, 32), ceil_div(Pfull, 32)), dim3(256), 0, stream, B, Pfull, xr, (size_t)Pfull, X, (size_t)B);
            for (int c0 = 0; c0 < npts && rc == SHG_OK; c0 += chunk) {
                const int nc = std::min(chunk, npts - c0);
                hipLaunchKernelGGL(synth_point_tables_kernel, dim3(ceil_div(nc, 64)), dim3(64), 0, stream, N, nc, colat + c0, lon + c0, knT + c0,
                                   (size_t)npts, tab, tab + a.size(), pkT, csr, rslot);
                rc = synth_generic(pkT, nc, csr, nc, rslot, 1, (long long)1 << 40, nc, X, Pfull, B, Cc, stream);
                if (rc) break;
                hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(nc, 32), ceil_div(B, 32)), dim3(256), 0, stream, nc, B, Cc, (size_t)B, values + c0, (size_t)npts);
            }
        }
        for (void* q : {(void*)X, (void*)xr, (void*)pkT, (void*)csr, (void*)Cc, (void*)rslot, (void*)knT, (void*)tab})
            if (q) (void)hipFreeAsync(q, stream);
        if (rc) return rc;
        SHG_HIP(hipGetLastError());
        return SHG_OK;
    }
    hipLaunchKernelGGL(synthesis_points_kernel, dim3(ceil_div(npts, 64), ceil_div(B, kPtEpochs)), dim3(64), (size_t)(N + 1) * 2 * kPtEpochs * sizeof(double), stream, N, npts, B, colat, lon,
                       knT, tab, tab + a.size(), anm, values);
    (void)hipFreeAsync(knT, stream);
    (void)hipFreeAsync(tab, stream);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

extern "C" int shg_covprop_points(int N, const double* colat, const double* lon, const double* kn, int npts, const double* cov,
                                  int nmin, double* sigma, void* stream_) {
    SHG_REQUIRE(N >= 0 && npts >= 0 && nmin >= 0 && nmin <= N + 1, "shg_covprop_points: bad size");
    if (npts == 0) return SHG_OK;
    SHG_REQUIRE(colat && lon && kn && sigma, "shg_covprop_points: NULL pointer");
    hipStream_t stream = (hipStream_t)stream_;
    const int Pfull = (N + 1) * (N + 1);
    const int Pn = Pfull - nmin * nmin;
    if (Pn == 0) {
        SHG_HIP(hipMemsetAsync(sigma, 0, (size_t)npts * sizeof(double), stream));
        return SHG_OK;
    }
    SHG_REQUIRE(cov != nullptr, "shg_covprop_points: NULL covariance");
    // per-point tables, Legendre table transposed (pkT[p][point]): consecutive lanes of the generated-operand kernel are
    // consecutive points (43 -> 55 TFLOP/s against the point-major table)
    std::vector<double> a, b;
    recursion_tables(N, a, b);
    double *pkT = nullptr, *csr = nullptr, *partial = nullptr, *knT = nullptr, *tab = nullptr;
    int* rslot = nullptr;
    const int ncolblocks = ceil_div(Pn, 128);
    if (hipMallocAsync((void**)&pkT, (size_t)npts * Pfull * sizeof(double), stream) != hipSuccess ||
        hipMallocAsync((void**)&csr, (size_t)(2 * N + 1) * npts * sizeof(double), stream) != hipSuccess ||
        hipMallocAsync((void**)&rslot, (size_t)Pfull * sizeof(int), stream) != hipSuccess ||
        hipMallocAsync((void**)&partial, (size_t)ncolblocks * npts * sizeof(double), stream) != hipSuccess ||
        hipMallocAsync((void**)&knT, (size_t)(N + 1) * npts * sizeof(double), stream) != hipSuccess ||
        hipMallocAsync((void**)&tab, 2 * a.size() * sizeof(double), stream) != hipSuccess)
        return fail(SHG_ERR_NOMEM, "shg_covprop_points: workspace allocation failed");
    SHG_HIP(hipMemcpyAsync(tab, a.data(), a.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    SHG_HIP(hipMemcpyAsync(tab + a.size(), b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    SHG_HIP(hipStreamSynchronize(stream));           // the host vectors go out of scope
    hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(npts, 32), ceil_div(N + 1, 32)), dim3(256), 0, stream, npts, N + 1, kn, (size_t)(N + 1), knT, (size_t)npts);
    hipLaunchKernelGGL(synth_point_tables_kernel, dim3(ceil_div(npts, 64)), dim3(64), 0, stream, N, npts, colat, lon, knT, (size_t)npts, tab,
                       tab + a.size(), pkT, csr, rslot);
    int rc = covprop_generic(pkT, npts, csr, npts, rslot, 1, (long long)1 << 40, 0, npts, cov, Pn, nmin * nmin, partial, sigma, nullptr, stream, false, true);
    for (void* q : {(void*)pkT, (void*)csr, (void*)rslot, (void*)partial, (void*)knT, (void*)tab}) (void)hipFreeAsync(q, stream);
    if (rc) return rc;
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

// Dense synthesis operator A [npts][Pn] (degree-wise columns from min_degree on) of a point list: one generation kernel per
// chunk of points, straight into the caller's matrix (replaces Grid.synthesis_matrix, grates/grid.py:412-443, which stacks
// per-order blocks on the host).
extern "C" int shg_synthesis_matrix(int N, int nmin, const double* colat, const double* lon, const double* kn, int npts, double* A, void* stream_) {
    SHG_REQUIRE(N >= 0 && npts >= 0 && nmin >= 0 && nmin <= N + 1, "shg_synthesis_matrix: bad size");
    const int Pfull = (N + 1) * (N + 1), Pn = Pfull - nmin * nmin;
    if (npts == 0 || Pn == 0) return SHG_OK;
    SHG_REQUIRE(colat && lon && kn && A, "shg_synthesis_matrix: NULL pointer");
    hipStream_t stream = (hipStream_t)stream_;
    std::vector<double> a, b;
    recursion_tables(N, a, b);
    const int chunk = point_chunk(npts, Pfull);
    double *pkT = nullptr, *csr = nullptr, *knT = nullptr, *tab = nullptr;
    int* rslot = nullptr;
    int rc = SHG_OK;
    if (hipMallocAsync((void**)&pkT, (size_t)chunk * Pfull * sizeof(double), stream) != hipSuccess ||
        hipMallocAsync((void**)&csr, (size_t)(2 * N + 1) * chunk * sizeof(double), stream) != hipSuccess ||
        hipMallocAsync((void**)&rslot, (size_t)Pfull * sizeof(int), stream) != hipSuccess ||
        hipMallocAsync((void**)&knT, (size_t)(N + 1) * npts * sizeof(double), stream) != hipSuccess ||
        hipMallocAsync((void**)&tab, 2 * a.size() * sizeof(double), stream) != hipSuccess)
        rc = fail(SHG_ERR_NOMEM, "shg_synthesis_matrix: workspace allocation failed");
    if (rc == SHG_OK) {
        hipError_t e = hipMemcpyAsync(tab, a.data(), a.size() * sizeof(double), hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(tab + a.size(), b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);           // the host vectors go out of scope
        if (e != hipSuccess) rc = fail(SHG_ERR_HIP, "shg_synthesis_matrix: table upload failed: %s", hipGetErrorString(e));
    }
    if (rc == SHG_OK) {
        hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(npts, 32), ceil_div(N + 1, 32)), dim3(256), 0, stream, npts, N + 1, kn, (size_t)(N + 1), knT, (size_t)npts);
        for (int c0 = 0; c0 < npts; c0 += chunk) {
            const int nc = std::min(chunk, npts - c0);
            hipLaunchKernelGGL(synth_point_tables_kernel, dim3(ceil_div(nc, 64)), dim3(64), 0, stream, N, nc, colat + c0, lon + c0, knT + c0,
                               (size_t)npts, tab, tab + a.size(), pkT, csr, rslot);
            hipLaunchKernelGGL(synthesis_matrix_kernel, dim3(ceil_div(Pn, 32), ceil_div(nc, 32)), dim3(256), 0, stream, nc, Pn, nmin * nmin, pkT, csr,
                               rslot, A + (size_t)c0 * Pn, (size_t)Pn);
        }
    }
    for (void* q : {(void*)pkT, (void*)csr, (void*)rslot, (void*)knT, (void*)tab})
        if (q) (void)hipFreeAsync(q, stream);
    if (rc) return rc;
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}
