// fp64 MFMA issue-rate experiments: operands from registers vs. one LDS read per MFMA, 1 or 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: constant operands, 4 acc; 1: 8 different register operands; 2: A operand from LDS ring (3 ahead); 3: like 2, single accumulator chain of 4 groups
__global__ __launch_bounds__(512) void k(int iters, double* out, double seed) {
    __shared__ double lds[64 * 80];
    for (int i = threadIdx.x; i < 64 * 80; i += blockDim.x) lds[i] = seed + i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    double4_t acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (double4_t){0, 0, 0, 0};
    double b[8], a[8];
    for (int i = 0; i < 8; ++i) { b[i] = seed * (i + 1) + lane; a[i] = seed - i + lane; }
    const double* ap = lds + (lane >> 4) * 80 + (lane & 15);
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int s = 0; s < 16; ++s) acc[s & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc[s & 3], 0, 0, 0);
    } else if (MODE == 1) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int s = 0; s < 16; ++s) acc[s & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s & 7], b[(s * 3) & 7], acc[s & 3], 0, 0, 0);
    } else {
        double ar[4];
        for (int s = 0; s < 4; ++s) ar[s] = ap[s * 320];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                acc[s & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[s & 3], b[s & 7], acc[s & 3], 0, 0, 0);
                ar[s & 3] = ap[((s + 4) & 15) * 320 + (it & 1) * 16];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    double r = 0;
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (r == 12345.678) out[threadIdx.x] = r;
}

template <int MODE>
void run(const char* name, int threads, double* out) {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount, iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, iters, out, 1.0);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, iters, out, 1.0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)(threads / 64) / 4.0 * iters * 16;
    printf("%-44s %d waves/SIMD: %.3f ms, %.1f TFLOP/s, %.1f cycles/MFMA/SIMD @2.4GHz\n", name, threads / 256, ms,
           (double)blocks * (threads / 64) * iters * 16 * 2048.0 / ms / 1e9, ms * 1e-3 * 2.4e9 / mfma_per_simd);
}

int main() {
    double* out; (void)hipMalloc(&out, 4096);
    for (int threads : {256, 512}) {
        if (threads == 256) { run<0>("constant operands", 256, out); run<1>("8 rotating register operands", 256, out); run<2>("A operand via LDS ring", 256, out); }
        else { run<0>("constant operands", 512, out); run<1>("8 rotating register operands", 512, out); run<2>("A operand via LDS ring", 512, out); }
    }
    return 0;
}
