// Table functions exposed for parity checks and the degree-wise index maps applied on the device:
//   shg_legendre / shg_legendre_order / shg_trigonometric   (grates/utilities.py:13-115, 249-275)
//   shg_ravel / shg_unravel                                 (grates/utilities.py:310-411)
//   shg_degree_scale                                        (grates/filter.py:61-72)
// The file is built with -ffp-contract=off: the recursions below keep the reference's expression order
// and produce the same IEEE results as the NumPy code for the same cos/sin inputs.
#include "common.h"

namespace shg {

__device__ inline double rec_a(int ni, int mi) {
    const double n = ni, m = mi;
    if (ni == mi + 1) return sqrt((double)(2 * ni + 1));
    return sqrt((2.0 * n - 1.0) / (n - m) * (2.0 * n + 1.0) / (n + m));
}
__device__ inline double rec_b(int ni, int mi) {
    const double n = ni, m = mi;
    if (ni == mi + 1) return 0.0;
    return sqrt((2.0 * n + 1.0) / (2.0 * n - 3.0) * (n - m - 1.0) / (n - m) * (n + m - 1.0) / (n + m));
}

// sectorials P_nn of every point: thread <-> point
__global__ void legendre_sectorial_kernel(int N, int k, const double* __restrict__ colat, double* __restrict__ pnm) {
    const int pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= k) return;
    const size_t E = (size_t)(N + 1) * (N + 1);
    double* P = pnm + pt * E;
    const double st = sin(colat[pt]);
    double pv = 1.0;
    P[0] = pv;
    for (int n = 1; n <= N; ++n) {
        if (n == 1)
            pv = sqrt(3.0) * st;
        else
            pv = sqrt((2.0 * n + 1.0) / (2.0 * n)) * st * pv;
        P[(size_t)n * (N + 1) + n] = pv;
    }
}

// column m of every point: thread <-> (point, m); also mirrors into the sine slots [m-1][n]
__global__ void legendre_column_kernel(int N, int k, const double* __restrict__ colat, double* __restrict__ pnm) {
    const int m = blockIdx.y;
    const int pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= k) return;
    const size_t E = (size_t)(N + 1) * (N + 1);
    double* P = pnm + pt * E;
    const double t = cos(colat[pt]);
    double p1 = P[(size_t)m * (N + 1) + m], p2 = 0.0;
    if (m >= 1) P[(size_t)(m - 1) * (N + 1) + m] = p1;
    for (int n = m + 1; n <= N; ++n) {
        const double p = (rec_a(n, m) * t) * p1 - rec_b(n, m) * p2;
        p2 = p1;
        p1 = p;
        P[(size_t)n * (N + 1) + m] = p;
        if (m >= 1) P[(size_t)(m - 1) * (N + 1) + n] = p;
    }
}

// legendre_functions_per_order: thread <-> point                         grates/utilities.py:85-115, 138-151
__global__ void legendre_order_kernel(int N, int m, int k, const double* __restrict__ colat, double* __restrict__ pm) {
    const int pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= k) return;
    const int cnt = N + 1 - m;
    double* out = pm + (size_t)pt * cnt;
    const double t = cos(colat[pt]);
    if (m == 0) {
        double p2 = 1.0;
        out[0] = p2;
        if (N == 0) return;
        double p1 = sqrt(3.0) * t;
        out[1] = p1;
        for (int ni = 2; ni <= N; ++ni) {
            const double n = ni;
            const double p = sqrt((2.0 * n - 1.0) * (2.0 * n + 1.0)) / n * t * p1 - sqrt((2.0 * n + 1.0) / (2.0 * n - 3.0)) * (n - 1.0) / n * p2;
            p2 = p1;
            p1 = p;
            out[ni] = p;
        }
        return;
    }
    const double s = sqrt(1.0 - t * t);
    double pv = sqrt(3.0) * s;
    for (int ni = 2; ni <= m; ++ni) {
        const double n = ni;
        pv = sqrt((2.0 * n + 1.0) / (2.0 * n)) * s * pv;
    }
    out[0] = pv;
    if (cnt > 1) out[1] = sqrt((double)(2 * m + 3)) * t * out[0];
    for (int ni = m + 2; ni <= N; ++ni) {
        const double n = ni, mm = m;
        out[ni - m] = sqrt((2.0 * n - 1.0) / (n - mm) * (2.0 * n + 1.0) / (n + mm)) * t * out[ni - 1 - m] -
                      sqrt((2.0 * n + 1.0) / (2.0 * n - 3.0) * (n - mm - 1.0) / (n - mm) * (n + mm - 1.0) / (n + mm)) * out[ni - 2 - m];
    }
}

__global__ void trigonometric_kernel(int N, int k, const double* __restrict__ lon, double* __restrict__ cs) {
    const int m = blockIdx.y;
    const int pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= k) return;
    double* T = cs + (size_t)pt * (N + 1) * (N + 1);
    if (m == 0) {
        for (int n = 0; n <= N; ++n) T[(size_t)n * (N + 1)] = 1.0;
        return;
    }
    const double arg = (double)m * lon[pt];
    const double c = cos(arg), s = sin(arg);
    for (int n = m; n <= N; ++n) {
        T[(size_t)n * (N + 1) + m] = c;
        T[(size_t)(m - 1) * (N + 1) + n] = s;
    }
}

// degree n and in-degree rank r of degree-wise vector position q (offset by nmin^2)
__device__ inline void degreewise_decode(long long q, int& n, int& r) {
    n = (int)sqrt((double)q);
    while ((long long)(n + 1) * (n + 1) <= q) ++n;
    while ((long long)n * n > q) --n;
    r = (int)(q - (long long)n * n);
}

__global__ void ravel_kernel(int B, int Na, int nmin, int P, const double* __restrict__ arr, double* __restrict__ vec) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long long)B * P) return;
    const int b = (int)(tid / P);
    const int q = (int)(tid % P);
    int n, r;
    degreewise_decode((long long)q + (long long)nmin * nmin, n, r);
    double v = 0.0;
    if (n <= Na) {
        const int m = (r + 1) >> 1;
        const bool sine = (r > 0) && ((r & 1) == 0);
        const int row = sine ? m - 1 : n, col = sine ? n : m;
        v = arr[((size_t)b * (Na + 1) + row) * (Na + 1) + col];
    }
    vec[tid] = v;
}

__global__ void unravel_kernel(int B, int nmin, int nmax, int P, const double* __restrict__ vec, double* __restrict__ arr) {
    const int E = (nmax + 1) * (nmax + 1);
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long long)B * E) return;
    const int b = (int)(tid / E);
    const int e = (int)(tid % E);
    const int row = e / (nmax + 1), col = e % (nmax + 1);
    int n, m, sine;
    if (col <= row) {
        n = row; m = col; sine = 0;
    } else {
        n = col; m = row + 1; sine = 1;
    }
    double v = 0.0;
    if (n >= nmin) {
        const int q = n * n - nmin * nmin + (m == 0 ? 0 : (sine ? 2 * m : 2 * m - 1));
        v = vec[(size_t)b * P + q];
    }
    arr[tid] = v;
}

__global__ void degree_scale_kernel(int B, int N, int nfirst, const double* __restrict__ w, const double* __restrict__ in,
                                    double* __restrict__ out) {
    const int E = (N + 1) * (N + 1);
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long long)B * E) return;
    const int e = (int)(tid % E);
    const int n = max(e / (N + 1), e % (N + 1));
    out[tid] = (n >= nfirst) ? in[tid] * w[n] : in[tid];
}

}  // namespace shg

using namespace shg;

extern "C" int shg_legendre(int N, const double* colat, int k, double* pnm, void* stream_) {
    SHG_REQUIRE(N >= 0 && k >= 0, "shg_legendre: bad size (N=%d, k=%d)", N, k);
    if (k == 0) return SHG_OK;
    SHG_REQUIRE(colat && pnm, "shg_legendre: NULL pointer");
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(legendre_sectorial_kernel, dim3(ceil_div(k, 64)), dim3(64), 0, stream, N, k, colat, pnm);
    hipLaunchKernelGGL(legendre_column_kernel, dim3(ceil_div(k, 64), N + 1), dim3(64), 0, stream, N, k, colat, pnm);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

extern "C" int shg_legendre_order(int N, int m, const double* colat, int k, double* pm, void* stream_) {
    SHG_REQUIRE(N >= 0 && k >= 0 && m >= 0, "shg_legendre_order: bad size");
    SHG_REQUIRE(m <= N, "order exceeds maximum degree (%d vs. %d)", m, N);
    if (k == 0) return SHG_OK;
    SHG_REQUIRE(colat && pm, "shg_legendre_order: NULL pointer");
    hipLaunchKernelGGL(legendre_order_kernel, dim3(ceil_div(k, 64)), dim3(64), 0, (hipStream_t)stream_, N, m, k, colat, pm);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

namespace shg {
// rows of the per-order operator block: thread <-> (row, column); pm [nlat][N + 1 - m] from legendre_order_kernel
__global__ void order_block_kernel(int N, int m, int c0, int cnt, int nlat, int nlon, int pointwise, const double* __restrict__ pm,
                                   const double* __restrict__ kn, const double* __restrict__ lon, double* __restrict__ out_cos,
                                   double* __restrict__ out_sin) {
    const long long rows = pointwise ? nlat : (long long)nlat * nlon;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * cnt) return;
    const long long row = e / cnt;
    const int c = (int)(e - row * cnt);
    const int i = pointwise ? (int)row : (int)(row / nlon);
    const int j = pointwise ? (int)row : (int)(row - (long long)i * nlon);
    const int n = m + c0 + c;
    const double p = pm[(size_t)i * (N + 1 - m) + c0 + c] * kn[(size_t)i * (N + 1) + n];
    if (m == 0) {
        out_cos[e] = p;
        return;
    }
    const double arg = (double)m * lon[j];
    out_cos[e] = p * cos(arg);
    out_sin[e] = p * sin(arg);
}
}  // namespace shg

extern "C" int shg_synthesis_matrix_order(int N, int m, int nmin, const double* colat, int nlat, const double* lon, int nlon, const double* kn,
                                          int pointwise, double* out_cos, double* out_sin, void* stream_) {
    SHG_REQUIRE(N >= 0 && m >= 0 && nmin >= 0 && nlat >= 0 && (pointwise || nlon >= 0), "shg_synthesis_matrix_order: bad size");
    SHG_REQUIRE(m <= N, "order exceeds maximum degree (%d vs. %d)", m, N);
    const int c0 = nmin > m ? nmin - m : 0, cnt = N + 1 - m - c0;
    const long long rows = pointwise ? nlat : (long long)nlat * nlon;
    if (rows == 0 || cnt <= 0) return SHG_OK;
    SHG_REQUIRE(colat && kn && out_cos && (m == 0 || (lon && out_sin)), "shg_synthesis_matrix_order: NULL pointer");
    SHG_REQUIRE(rows * cnt < (1LL << 31) * 256, "shg_synthesis_matrix_order: problem too large");
    hipStream_t stream = (hipStream_t)stream_;
    double* pm = nullptr;
    if (workspace_alloc((void**)&pm, (size_t)nlat * (N + 1 - m) * sizeof(double), stream) != hipSuccess)
        return fail(SHG_ERR_NOMEM, "shg_synthesis_matrix_order: workspace allocation failed");
    hipLaunchKernelGGL(legendre_order_kernel, dim3(ceil_div(nlat, 64)), dim3(64), 0, stream, N, m, nlat, colat, pm);
    hipLaunchKernelGGL(order_block_kernel, dim3((unsigned)ceil_div64(rows * cnt, 256)), dim3(256), 0, stream, N, m, c0, cnt, nlat, nlon, pointwise,
                       pm, kn, lon, out_cos, out_sin);
    const hipError_t e = hipGetLastError();
    (void)hipFreeAsync(pm, stream);
    if (e != hipSuccess) return fail(SHG_ERR_HIP, "shg_synthesis_matrix_order: launch failed: %s", hipGetErrorString(e));
    return SHG_OK;
}

extern "C" int shg_trigonometric(int N, const double* lon, int k, double* cs, void* stream_) {
    SHG_REQUIRE(N >= 0 && k >= 0, "shg_trigonometric: bad size");
    if (k == 0) return SHG_OK;
    SHG_REQUIRE(lon && cs, "shg_trigonometric: NULL pointer");
    hipLaunchKernelGGL(trigonometric_kernel, dim3(ceil_div(k, 64), N + 1), dim3(64), 0, (hipStream_t)stream_, N, k, lon, cs);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

extern "C" int shg_ravel(const double* arr, int B, int Na, int nmin, int nmax, double* vec, void* stream_) {
    SHG_REQUIRE(B >= 0 && Na >= 0 && nmin >= 0 && nmax >= nmin - 1, "shg_ravel: bad size");
    const long long P = (long long)(nmax + 1) * (nmax + 1) - (long long)nmin * nmin;
    if (B == 0 || P <= 0) return SHG_OK;
    SHG_REQUIRE(arr && vec, "shg_ravel: NULL pointer");
    SHG_REQUIRE(P * B < (1LL << 40), "shg_ravel: problem too large");
    hipLaunchKernelGGL(ravel_kernel, dim3((unsigned)ceil_div64(P * B, 256)), dim3(256), 0, (hipStream_t)stream_, B, Na, nmin, (int)P, arr, vec);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

extern "C" int shg_unravel(const double* vec, int B, int nmin, int nmax, double* arr, void* stream_) {
    SHG_REQUIRE(B >= 0 && nmin >= 0 && nmax >= 0, "shg_unravel: bad size");
    const long long P = (long long)(nmax + 1) * (nmax + 1) - (long long)nmin * nmin;
    if (B == 0) return SHG_OK;
    SHG_REQUIRE(arr && (vec || P <= 0), "shg_unravel: NULL pointer");
    const long long total = (long long)B * (nmax + 1) * (nmax + 1);
    hipLaunchKernelGGL(unravel_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream_, B, nmin, nmax,
                       (int)(P > 0 ? P : 0), vec, arr);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

extern "C" int shg_degree_scale(const double* w, int N, int nfirst, const double* in, int B, double* out, void* stream_) {
    SHG_REQUIRE(B >= 0 && N >= 0, "shg_degree_scale: bad size");
    if (B == 0) return SHG_OK;
    SHG_REQUIRE(w && in && out, "shg_degree_scale: NULL pointer");
    const long long total = (long long)B * (N + 1) * (N + 1);
    hipLaunchKernelGGL(degree_scale_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream_, B, N, nfirst, w, in, out);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}
