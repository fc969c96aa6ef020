// Reductions over the epochs of a batch of grids: the consumer side of the batched synthesis
//   (gridded_rms, grates/gravityfield.py:1143-1172: rms[i] = sqrt(sum_t grid_t[i]^2 / T)).
// HBM bound: every grid value is read once (8 B M per epoch), one thread per grid point, consecutive lanes = consecutive
// points; the squares are added in epoch order (the reference's `rms_values += values**2`), no FMA.
#include "common.h"

namespace shg {

__global__ __launch_bounds__(256) void epoch_rms_kernel(int B, long long M, const double* __restrict__ values, int accumulate, long long count,
                                                        double* __restrict__ acc) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const double* v = values + i;
    double s = accumulate ? acc[i] : 0.0;
    int b = 0;
    for (; b + 4 <= B; b += 4) {                    // four independent loads in flight, added in order
        const double x0 = __builtin_nontemporal_load(v + (size_t)b * M), x1 = __builtin_nontemporal_load(v + (size_t)(b + 1) * M);
        const double x2 = __builtin_nontemporal_load(v + (size_t)(b + 2) * M), x3 = __builtin_nontemporal_load(v + (size_t)(b + 3) * M);
        s = s + x0 * x0;
        s = s + x1 * x1;
        s = s + x2 * x2;
        s = s + x3 * x3;
    }
    for (; b < B; ++b) {
        const double x = v[(size_t)b * M];
        s = s + x * x;
    }
    acc[i] = count > 0 ? sqrt(s / (double)count) : s;
}

}  // namespace shg

extern "C" int shg_epoch_rms(const double* values, int B, long long M, int accumulate, long long count, double* acc, void* stream_) {
    SHG_REQUIRE(B >= 0 && M >= 0 && count >= 0, "shg_epoch_rms: negative size");
    if (M == 0) return SHG_OK;
    SHG_REQUIRE(acc != nullptr && (values != nullptr || B == 0), "shg_epoch_rms: NULL pointer");
    SHG_REQUIRE(M <= (long long)0x7fffffff * 256, "shg_epoch_rms: %lld points per epoch exceed the launch grid", M);
    hipLaunchKernelGGL(shg::epoch_rms_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, B, M, values, accumulate,
                       count, acc);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}
