// Order-wise block filter (DDK family):  replaces OrderWiseFilter.filter, grates/filter.py:180-191.
//   for every order m and cosine / sine:  y[m..N] = W_block[0:N+1-m, 0:N+1-m] x[m..N]   for all epochs at once,
//   degrees 0 and 1 copied from the input.
// HBM-bound integer-indexed streaming: each block matrix is read once per tile of 32 epochs, coefficient
// vectors are staged in LDS.
#include "common.h"

namespace shg {

constexpr int kFiltEpochs = 32;     // epochs per thread block (x dimension of the 8 x 32 thread tile)

__global__ __launch_bounds__(256) void orderwise_filter_kernel(int Nb, int N, int B, const double* __restrict__ blocks,
                                                               const long long* __restrict__ block_off,
                                                               const double* __restrict__ in, double* __restrict__ out) {
    extern __shared__ double xs[];                 // [n][kFiltEpochs]
    const int kb = blockIdx.x;                     // 0: order 0 cos, 2m-1: order m cos, 2m: order m sin
    const int m = (kb + 1) >> 1;
    const bool sine = kb > 0 && (kb & 1) == 0;
    const int n = N + 1 - m;                       // coefficients of this order in the field
    const int ld = Nb + 1 - m;                     // leading dimension of the stored block
    const int e = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int b = blockIdx.y * kFiltEpochs + e;
    const size_t E = (size_t)(N + 1) * (N + 1);
    const double* W = blocks + block_off[kb];

    // element k (degree m + k) of this order: cos at [m+k][m], sin at [m-1][m+k]
    auto pos = [&](int k) -> size_t { return sine ? (size_t)(m - 1) * (N + 1) + (m + k) : (size_t)(m + k) * (N + 1) + m; };
    for (int k = ty; k < n; k += 8) xs[k * kFiltEpochs + e] = (b < B) ? in[(size_t)b * E + pos(k)] : 0.0;
    __syncthreads();
    for (int r = ty; r < n; r += 8) {
        const double* wrow = W + (size_t)r * ld;
        double s = 0.0;
        for (int c = 0; c < n; ++c) s = fma(wrow[c], xs[c * kFiltEpochs + e], s);
        if (b < B) {
            const int degree = m + r;
            out[(size_t)b * E + pos(r)] = (degree <= 1) ? xs[r * kFiltEpochs + e] : s;    // filter.py:189
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
    hipLaunchKernelGGL(orderwise_filter_kernel, dim3(2 * N + 1, ceil_div(B, kFiltEpochs)), dim3(256), lds, (hipStream_t)stream_, Nb, N, B,
                       blocks_packed, (const long long*)block_off, anm_in, anm_out);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}
