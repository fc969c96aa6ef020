// Which feature of the synthesis longitude loop costs MFMA issue slots?  Builds the loop up step by step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int NF = 52, STRIDE = 80, AHEAD = 4;

// MODE 0: small LDS, 8 B regs; 1: 133 KB panel addressing; 2: + 52 stationary B registers; 3: + 16 accumulators (4 groups x 4 row tiles per k-step)
template <int MODE>
__global__ __launch_bounds__(512) void k(int iters, double* out, const double* tab, double seed) {
    extern __shared__ double lds[];
    const int nlds = (MODE == 0 ? 64 : 208) * STRIDE;
    for (int i = threadIdx.x; i < nlds; i += blockDim.x) lds[i] = seed + i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int fr = lane & 15, fk = lane >> 4;
    double bf[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) bf[f] = MODE >= 2 ? tab[f * 64 + lane] : seed * ((f & 7) + 1) + lane;
    const int span = MODE == 0 ? 16 : NF;      // k-steps addressed in LDS
    if (MODE <= 2) {
        double4_t acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = (double4_t){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
            const double* ap = lds + fk * STRIDE + (it & 3) * 16 + fr;
            double ar[AHEAD];
#pragma unroll
            for (int s = 0; s < AHEAD; ++s) ar[s] = ap[(s % span) * 4 * STRIDE];
#pragma unroll
            for (int s = 0; s < NF; ++s) {
                acc[s & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[s % AHEAD], bf[MODE >= 2 ? s : (s & 7)], acc[s & 3], 0, 0, 0);
                if (s + AHEAD < NF) ar[s % AHEAD] = ap[((s + AHEAD) % span) * 4 * STRIDE];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        double r = 0;
        for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (r == 12345.678) out[threadIdx.x] = r;
    } else {
        double4_t acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0, 0, 0, 0};
        for (int it = 0; it < iters / 4; ++it) {
            const double* ap = lds + fk * STRIDE + fr;
#pragma unroll
            for (int s = 0; s < NF; ++s) {
                double a4[4];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) a4[rt] = ap[s * 4 * STRIDE + rt * 16];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) acc[s & 3][rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[rt], bf[s], acc[s & 3][rt], 0, 0, 0);
            }
        }
        double r = 0;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r += acc[i][j][0] + acc[i][j][3];
        if (r == 12345.678) out[threadIdx.x] = r;
    }
}

template <int MODE>
void run(const char* name, double* out, const double* tab) {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount, iters = 400, threads = 512;
    const size_t lds = (size_t)(MODE == 0 ? 64 : 208) * STRIDE * 8;
    (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), lds, 0, iters, out, tab, 1.0);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), lds, 0, iters, out, tab, 1.0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = 2.0 * iters * NF;
    printf("%-60s %.3f ms, %.1f cycles/MFMA/SIMD @2.4GHz (%s)\n", name, ms, ms * 1e-3 * 2.4e9 / mfma_per_simd, hipGetErrorString(hipGetLastError()));
}

int main() {
    double *out, *tab; (void)hipMalloc(&out, 4096); (void)hipMalloc(&tab, NF * 64 * 8 + 4096); (void)hipMemset(tab, 0, NF * 64 * 8);
    run<0>("8 waves, 40 KB LDS, 8 B regs, A via LDS ring", out, tab);
    run<1>("+ 133 KB panel addressing", out, tab);
    run<2>("+ 52 stationary B registers", out, tab);
    run<3>("16 accumulators, 4 A reads + 4 MFMAs per k-step (v2 shape)", out, tab);
    return 0;
}
