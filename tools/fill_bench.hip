// What the card's write path gives for a plain contiguous fill of 2 GB, by launch shape: many small workgroups (what an elementwise library
// kernel does) against persistent ones (what tools/store_bench.hip measures: one or a few workgroups per CU walking the buffer).
//   hipcc -O3 --offload-arch=gfx950 tools/fill_bench.hip -o tools/scratch/fill_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double2_t __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ void st(double2_t v, double2_t* p) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
// every thread 4 x 16 bytes, a workgroup a contiguous 16 KB piece; grid-stride over the pieces
template <bool NT>
__global__ __launch_bounds__(256) void fill_kernel(double2_t* out, size_t pieces) {
    for (size_t p = blockIdx.x; p < pieces; p += gridDim.x) {
        double2_t* base = out + p * 1024 + threadIdx.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) st<NT>((double2_t){1.0, 2.0}, base + 256 * j);
    }
}
template <bool NT>
void run(const char* name, double2_t* out, size_t bytes, unsigned grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t pieces = bytes / 16384;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fill_kernel<NT>, dim3(grid), dim3(256), 0, 0, out, pieces);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(fill_kernel<NT>, dim3(grid), dim3(256), 0, 0, out, pieces);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-14s grid %7u: %.4f ms  %.2f TB/s\n", name, grid, ms / 20, bytes / (ms / 20 * 1e-3) / 1e12);
}
int main() {
    const size_t bytes = (size_t)240 * 720 * 1440 * 8;
    double2_t* out;
    hipMalloc(&out, bytes + (1 << 20));
    for (unsigned grid : {256u, 512u, 1024u, 2048u, 4096u, 16384u, (unsigned)(bytes / 16384)}) {
        run<false>("plain", out, bytes, grid);
        run<true>("non-temporal", out, bytes, grid);
    }
    return 0;
}
