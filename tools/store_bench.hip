// Store-engine ceilings of the synthesis output pattern (240 x 720 x 1440 doubles = 1.99 GB, rows of 11520 bytes):
// persistent workgroups (one per CU) walk the (epoch tile, parallel tile) tiles; a wave owns one epoch of the tile and writes, per
// column tile and image, 16 rows x 16 columns (128 bytes per row piece) -- with 8-byte stores (4 rows x 128 B per instruction),
// 16-byte stores (8 rows x 128 B) or, for reference, fully contiguous 1 KB per instruction.  No arithmetic.
//   hipcc -O3 --offload-arch=gfx950 tools/store_bench.hip -o tools/scratch/store_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double2_t __attribute__((ext_vector_type(2)));
constexpr int NLAT = 720, NLON = 1440, B = 240, R = 10, ND = 72, NR = 144;

#ifndef NT
#define NT 1
#endif
template <typename V>
__device__ __forceinline__ void STORE(V v, V* p) {
#if NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
template <int STYLE, int WAVES>      // STYLE 0: 8 B stores, 1: 16 B stores, 2: contiguous 1 KB per instruction
__global__ __launch_bounds__(64 * WAVES) void store_kernel(double* G, int ntiles) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    if (lds && threadIdx.x == 1023456) lds[0] = 1.0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int bt = tile / 45, it = tile % 45;
        const int epoch = bt * 4 + (wave & 3);
        double* Ge = G + (size_t)epoch * NLAT * NLON;
        if (STYLE == 3) {
            // pairs of adjacent column tiles in one wave: 4 rows x 256 B per instruction; the half tile behind them as 8 rows x 64 B
            for (int pr = (WAVES == 8 ? wave >> 2 : 0); pr < 3; pr += (WAVES == 8 ? 2 : 1)) {
#pragma unroll
                for (int t = 0; t < 2 * R; ++t) {
                    const int k = t < R ? t : t - R;
                    int w = NLON / 2 + k * NR - (t < R ? 0 : ND);
                    w = w >= NLON ? w - NLON : w;
                    if (pr < 2) {
                        const int col0 = t < R ? w + 32 * pr : w + ND - 32 * pr - 32;
#pragma unroll
                        for (int h = 0; h < 4; ++h) {
                            const int s = (lane >> 4) + 4 * h;
                            const int row = s < 8 ? it * 8 + s : NLAT - 1 - (it * 8 + s - 8);
                            STORE((double2_t){(double)t, (double)h}, (double2_t*)(Ge + (size_t)row * NLON + col0 + 2 * (lane & 15)));
                        }
                    } else {
                        const int col0 = t < R ? w + 64 : w + ND - 72;
#pragma unroll
                        for (int h = 0; h < 1; ++h) {
                            const int s = lane >> 2;
                            const int row = s < 8 ? it * 8 + s : NLAT - 1 - (it * 8 + s - 8);
                            STORE((double2_t){(double)t, (double)h}, (double2_t*)(Ge + (size_t)row * NLON + col0 + 2 * (lane & 3)));
                        }
                    }
                }
            }
            continue;
        }
        for (int ct = (WAVES == 8 ? wave >> 2 : 0); ct < 5; ct += (WAVES == 8 ? 2 : 1)) {
            const int ncol = ct == 4 ? 8 : 16;
#pragma unroll
            for (int t = 0; t < 2 * R; ++t) {
                const int k = t < R ? t : t - R;
                int w = NLON / 2 + k * NR - (t < R ? 0 : ND);
                w = w >= NLON ? w - NLON : w;
                const int col0 = t < R ? w + 16 * ct : w + ND - 16 * ct - ncol;
                if (STYLE == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int s = fk + 4 * r;
                        const int row = s < 8 ? it * 8 + s : NLAT - 1 - (it * 8 + s - 8);
                        if (fr < ncol) STORE((double)(t + r), Ge + (size_t)row * NLON + col0 + fr);
                    }
                } else if (STYLE == 1) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int s = fk + ((fr & 1) ? 8 : 0) + 4 * h;
                        const int row = s < 8 ? it * 8 + s : NLAT - 1 - (it * 8 + s - 8);
                        if ((fr & ~1) < ncol) STORE((double2_t){(double)t, (double)h}, (double2_t*)(Ge + (size_t)row * NLON + col0 + (fr & ~1)));
                    }
                } else {
                    // same bytes, contiguous: 2 x 1 KB per image and wave
                    double* base = Ge + ((size_t)(it * 5 + ct) * 2 * R + t) * 256;
                    if (ct < 4 || true) {
                        STORE((double2_t){(double)t, 1.0}, (double2_t*)(base) + lane);
                        STORE((double2_t){(double)t, 2.0}, (double2_t*)(base + 128) + lane);
                    }
                }
            }
        }
    }
}

template <int STYLE, int WAVES>
void run(const char* name, double* G, int grid, size_t lds) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)store_kernel<STYLE, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((store_kernel<STYLE, WAVES>), dim3(grid), dim3(64 * WAVES), lds, 0, G, 2700);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((store_kernel<STYLE, WAVES>), dim3(grid), dim3(64 * WAVES), lds, 0, G, 2700);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double bytes = (double)B * NLAT * NLON * 8 * (STYLE == 2 ? 1.0 : 1.0);
    printf("%-44s grid %4d lds %3zu KB: %.4f ms  %.2f TB/s\n", name, grid, lds >> 10, ms, bytes / (ms * 1e-3) / 1e12);
}

int main() {
    double* G;
    hipMalloc(&G, (size_t)B * NLAT * NLON * 8 + (1 << 20));
    // fewer workgroups than CUs: what ONE CU can store when the HBM is not the limit (time x grid / 256 = time of a CU's share)
    for (size_t lds : {(size_t)150 << 10}) {
        const int per_cu = lds > (80 << 10) ? 1 : lds > (40 << 10) ? 2 : 4;
        for (int mult : {1}) {
            const int grid = 256 * per_cu * mult;
            run<0, 4>("8 B stores, 4 waves", G, grid, lds);
            run<0, 8>("8 B stores, 8 waves", G, grid, lds);
            run<1, 4>("16 B stores, 4 waves", G, grid, lds);
            run<1, 8>("16 B stores, 8 waves", G, grid, lds);
            run<3, 4>("16 B stores, 256 B runs, 4 waves", G, grid, lds);
            run<3, 8>("16 B stores, 256 B runs, 8 waves", G, grid, lds);
            run<2, 4>("contiguous 1 KB, 4 waves", G, grid, lds);
            run<2, 8>("contiguous 1 KB, 8 waves", G, grid, lds);
        }
    }
    return 0;
}
