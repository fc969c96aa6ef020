// Micro-benchmarks that calibrate the rooflines used in DESIGN.md (fp64 MFMA rate, fp64 VALU FMA rate,
// both pipes together, HBM streaming write / copy).  Build: hipcc -O3 --offload-arch=gfx950 microbench.hip -o microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>   // 0 mfma only, 1 valu only, 2 waves 0-3 mfma + waves 4-7 valu
__global__ __launch_bounds__(512) void pipes_kernel(int iters, double* out, double seed) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = MODE == 0 || (MODE == 2 && wave < 4);
    if (do_mfma) {
        double4_t acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = (double4_t){0, 0, 0, 0};
        double a = seed + threadIdx.x, b = seed * 0.5 + threadIdx.x;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        double s = 0;
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (s == 12345.678) out[threadIdx.x] = s;
    } else {
        double x[8];
        for (int i = 0; i < 8; ++i) x[i] = seed + i + threadIdx.x;
        const double c = seed * 1e-9, d = 1.0 - seed * 1e-12;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = fma(x[i], d, c);
        }
        double s = 0;
        for (int i = 0; i < 8; ++i) s += x[i];
        if (s == 12345.678) out[threadIdx.x] = s;
    }
}

__global__ void fill_kernel(double2* __restrict__ dst, size_t n2, double v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n2; i += stride) dst[i] = make_double2(v, v);
}
__global__ void copy_kernel(const double2* __restrict__ src, double2* __restrict__ dst, size_t n2) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n2; i += stride) dst[i] = src[i];
}

template <typename F>
float time_ms(F f, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    double* out; CK(hipMalloc(&out, 4096));
    const int cus = prop.multiProcessorCount;
    const int iters = 4000;
    for (int wpc = 1; wpc <= 2; ++wpc) {     // blocks per CU
        const int blocks = cus * wpc;
        float t0 = time_ms([&] { hipLaunchKernelGGL(pipes_kernel<0>, dim3(blocks), dim3(256), 0, 0, iters, out, 1.0); }, 5);
        double mfma_flops = (double)blocks * 4 * iters * 4 * (16.0 * 16 * 4 * 2);
        printf("mfma_f64 only   (%d blk/CU x 4 waves): %.3f ms  %.1f TFLOP/s\n", wpc, t0, mfma_flops / t0 / 1e9);
        float t1 = time_ms([&] { hipLaunchKernelGGL(pipes_kernel<1>, dim3(blocks), dim3(256), 0, 0, iters, out, 1.0); }, 5);
        double valu_flops = (double)blocks * 256 * iters * 32 * 2.0;
        printf("valu fma_f64 only (%d blk/CU x 4 waves): %.3f ms  %.1f TFLOP/s\n", wpc, t1, valu_flops / t1 / 1e9);
        float t2 = time_ms([&] { hipLaunchKernelGGL(pipes_kernel<2>, dim3(blocks), dim3(512), 0, 0, iters, out, 1.0); }, 5);
        printf("both (4 mfma + 4 valu waves per block, %d blk/CU): %.3f ms  mfma %.1f + valu %.1f = %.1f TFLOP/s\n", wpc, t2,
               mfma_flops / t2 / 1e9, valu_flops / t2 / 1e9, (mfma_flops + valu_flops) / t2 / 1e9);
    }
    {
        float t1 = time_ms([&] { hipLaunchKernelGGL(pipes_kernel<1>, dim3(cus * 2), dim3(512), 0, 0, iters, out, 1.0); }, 5);
        double valu_flops = (double)cus * 2 * 512 * iters * 32 * 2.0;
        printf("valu fma_f64 only (2 blk/CU x 8 waves): %.3f ms  %.1f TFLOP/s\n", t1, valu_flops / t1 / 1e9);
    }
    const size_t bytes = (size_t)4 << 30;
    double2 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    const size_t n2 = bytes / sizeof(double2);
    float tf = time_ms([&] { hipLaunchKernelGGL(fill_kernel, dim3(cus * 16), dim3(256), 0, 0, a, n2, 1.5); }, 5);
    printf("HBM fill 4 GiB (16 B/lane stores): %.3f ms  %.2f TB/s\n", tf, bytes / tf / 1e9);
    float tc = time_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(cus * 16), dim3(256), 0, 0, a, b, n2); }, 5);
    printf("HBM copy 4 GiB: %.3f ms  %.2f TB/s (read+write)\n", tc, 2.0 * bytes / tc / 1e9);
    return 0;
}
