// Which property of the synthesis kernel's store pattern costs the gap to a plain fill (tools/fill_bench.hip: 5.9 - 6.0 TB/s)?  One persistent
// workgroup of 8 waves per CU (the kernel's shape), 2 GB (240 x 720 x 1440 doubles, rows of 11520 bytes), 16-byte stores, no arithmetic:
//   0  contiguous: the workgroup's 8 waves write 8 KB in a row, pieces in address order                                  (the fill, at this shape)
//   1  contiguous per wave, the four epochs of a tile (8.3 MB apart) on different waves                                   (store_bench style 2)
//   2  the kernel's pattern: per (wave = epoch, column tile) 20 images x 16 rows x 128 B, image-major                      (store_bench style 1)
//   3  the same pieces ROW-major: for every row the 20 images of the column tile one after the other
//   4  the kernel's pattern with the waves of a workgroup on ONE epoch (four column tiles at a time)
//   hipcc -O3 --offload-arch=gfx950 tools/store_bench2.hip -o tools/scratch/store_bench2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double2_t __attribute__((ext_vector_type(2)));
constexpr int NLAT = 720, NLON = 1440, B = 240, R = 10, ND = 72, NR = 144;
#ifndef NT
#define NT 1
#endif
__device__ __forceinline__ void st(double2_t v, double2_t* p) {
#if NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ int image_col0(int t, int ct) {
    const int k = t < R ? t : t - R;
    int w = NLON / 2 + k * NR - (t < R ? 0 : ND);
    w = w >= NLON ? w - NLON : w;
    const int ncol = ct == 4 ? 8 : 16;
    return t < R ? w + 16 * ct : w + ND - 16 * ct - ncol;
}
template <int STYLE>
__global__ __launch_bounds__(512) void store_kernel(double* G, int ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int bt = tile / 45, it = tile % 45;
        if (STYLE == 0) {
            // the tile's bytes (4 epochs x 16 rows x 11520 B = 737 280 B) as one contiguous piece at tile * 737280
            double2_t* base = (double2_t*)((char*)G + (size_t)tile * 737280);
            for (int i = threadIdx.x; i < 737280 / 16; i += 512) st((double2_t){1.0, 2.0}, base + i);
            continue;
        }
        if (STYLE == 1) {
            const int epoch = bt * 4 + (wave & 3);
            double2_t* base = (double2_t*)((char*)G + (size_t)epoch * NLAT * NLON * 8 + (size_t)it * 184320 + (size_t)(wave >> 2) * 92160);
            for (int i = lane; i < 92160 / 16; i += 64) st((double2_t){1.0, 2.0}, base + i);
            continue;
        }
        const int ct0 = STYLE == 4 ? wave : wave >> 2, ctstep = STYLE == 4 ? 8 : 2;
        for (int e = 0; e < (STYLE == 4 ? 4 : 1); ++e) {
            const int epoch = bt * 4 + (STYLE == 4 ? e : (wave & 3));
            double* Ge = G + (size_t)epoch * NLAT * NLON;
            for (int ct = ct0; ct < 5; ct += ctstep) {
                const int ncol = ct == 4 ? 8 : 16;
                if (STYLE == 3) {
                    // lanes: 8 lanes x 16 B = one 128-byte piece; 8 pieces (images t0 .. t0 + 7) per instruction, one row at a time
#pragma unroll 4
                    for (int s = 0; s < 16; ++s) {
                        const int row = s < 8 ? it * 8 + s : NLAT - 1 - (it * 8 + s - 8);
                        for (int t0 = 0; t0 < 2 * R; t0 += 8) {
                            const int t = t0 + (lane >> 3);
                            if (t < 2 * R && 2 * (lane & 7) < ncol) st((double2_t){(double)t, 1.0}, (double2_t*)(Ge + (size_t)row * NLON + image_col0(t, ct) + 2 * (lane & 7)));
                        }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 2 * R; ++t) {
                        const int col0 = image_col0(t, ct);
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int s = fk + ((fr & 1) ? 8 : 0) + 4 * h;
                            const int row = s < 8 ? it * 8 + s : NLAT - 1 - (it * 8 + s - 8);
                            if ((fr & ~1) < ncol) st((double2_t){(double)t, (double)h}, (double2_t*)(Ge + (size_t)row * NLON + col0 + (fr & ~1)));
                        }
                    }
                }
            }
        }
    }
}
template <int STYLE>
void run(const char* name, double* G) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel<STYLE>, dim3(256), dim3(512), 0, 0, G, 2700);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(store_kernel<STYLE>, dim3(256), dim3(512), 0, 0, G, 2700);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 20;
    printf("%-78s %.4f ms  %.2f TB/s\n", name, ms, (double)B * NLAT * NLON * 8 / (ms * 1e-3) / 1e12);
}
int main() {
    double* G;
    hipMalloc(&G, (size_t)B * NLAT * NLON * 8 + (1 << 20));
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("0 contiguous per workgroup (tile bytes in address order)", G);
        run<1>("1 contiguous per wave, four epochs per workgroup", G);
        run<2>("2 kernel pattern: 20 images x 16 rows x 128 B per (epoch, column tile), image-major", G);
        run<3>("3 the same pieces row-major (8 images of one row per instruction)", G);
        run<4>("4 kernel pattern, the 8 waves of a workgroup on one epoch at a time", G);
    }
    return 0;
}
