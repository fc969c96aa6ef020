// Read-side ceilings of the analysis input pattern (240 x 360 x 720 doubles = 498 MB, rows of 5760 bytes): workgroups of four waves own
// 64 rows, a wave 16 of them; the quarter domain is walked in steps, every step reading the four images of the step's columns of every
// row (x3 ascending from the row start, x1 ascending from the middle, x2 descending from the middle, x4 descending from the row end).
// Variants: bytes that one load instruction takes from one row (64: 16 rows x 64 B as the MFMA operand layout has it, 128: 8 rows x
// 128 B, 256), a fully contiguous reference (the same bytes as 1 KB runs), and the same with the transform's output written behind
// the reads (193 slots x 64 rows per workgroup, 128-byte pieces).  No arithmetic beyond a checksum.
//   hipcc -O3 --offload-arch=gfx950 tools/load_bench.hip -o tools/scratch/load_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double2_t __attribute__((ext_vector_type(2)));
constexpr int NLAT = 360, NLON = 720, B = 240;
constexpr long long ROWS = (long long)B * NLAT;

// PIECE: bytes per row and instruction (64 .. 1024); DEPTH: steps in flight; CONTIG: ignore the row structure
template <int PIECE, int DEPTH, bool CONTIG, bool WRITES>
__global__ __launch_bounds__(256, 2) void load_kernel(const double* __restrict__ V, double* __restrict__ out, double* __restrict__ gt) {
    constexpr int LPR = PIECE / 16;            // lanes per row
    constexpr int RPI = 64 / LPR;              // rows per instruction
    constexpr int NI = 16 / RPI;               // instructions per image and step (16 rows of a wave)
    constexpr int STEPCOLS = PIECE / 8;        // columns of a step
    constexpr int NSTEP = (180 + STEPCOLS - 1) / STEPCOLS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.x * 64 + wave * 16;
    const int lrow = lane / LPR, lcol = (lane % LPR) * 2;
    double2_t acc = {0.0, 0.0};
    double2_t buf[DEPTH][4][NI];
    auto issue = [&](int step, int slot) {
        const int c0 = step * STEPCOLS;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const long long row = r0 + i * RPI + lrow;
            const double* p = V + row * NLON;
            int c = c0 + lcol;
            if (c > 178) c = 178;
            if (CONTIG) {
                const double* q = V + (r0 * NLON) + ((size_t)(step * 4) * NI + i) * 128 + lane * 2;      // 1 KB runs inside the wave's 16 rows
                buf[slot][0][i] = *(const double2_t*)(q);
                buf[slot][1][i] = *(const double2_t*)(q + NI * 128);
                buf[slot][2][i] = *(const double2_t*)(q + 2 * NI * 128);
                buf[slot][3][i] = *(const double2_t*)(q + 3 * NI * 128);
            } else {
                buf[slot][0][i] = *(const double2_t*)(p + c);
                buf[slot][1][i] = *(const double2_t*)(p + 360 + c);
                buf[slot][2][i] = *(const double2_t*)(p + 358 - c);
                buf[slot][3][i] = *(const double2_t*)(p + 718 - c);
            }
        }
    };
    auto consume = [&](int slot) {
#pragma unroll
        for (int im = 0; im < 4; ++im)
#pragma unroll
            for (int i = 0; i < NI; ++i) acc += buf[slot][im][i];
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(d < NSTEP ? d : NSTEP - 1, d);
    for (int s0 = 0; s0 < NSTEP; s0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int s = s0 + d;
            if (s < NSTEP) {
                const int sn = s + DEPTH - 1 < NSTEP ? s + DEPTH - 1 : NSTEP - 1;
                issue(sn, (d + DEPTH - 1) % DEPTH);
                consume(d);
            }
        }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y;
    if (WRITES) {
        // the transform's output: 193 slots x the workgroup's 64 rows, gt[slot][row]; a wave writes 16 rows (128 B) of 4 slots per instruction
        const int fr = lane & 15, fk = lane >> 4;
        for (int s0 = 0; s0 < 193; s0 += 4) {
            const int slot = s0 + fk;
            if (slot < 193) gt[(size_t)slot * ROWS + r0 + fr] = acc.x + slot;
        }
    }
}

template <int PIECE, int DEPTH, bool CONTIG, bool WRITES>
static void run(const char* name, const double* V, double* out, double* gt) {
    const int blocks = (int)(ROWS / 64);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((load_kernel<PIECE, DEPTH, CONTIG, WRITES>), dim3(blocks), dim3(256), 0, 0, V, out, gt);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((load_kernel<PIECE, DEPTH, CONTIG, WRITES>), dim3(blocks), dim3(256), 0, 0, V, out, gt);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)ROWS * NLON * 8 + (WRITES ? (double)ROWS * 193 * 8 : 0.0);
    printf("%-28s %8.1f us  %6.2f TB/s\n", name, 1e3 * ms / 10, bytes / (ms / 10 * 1e-3) / 1e12);
    fflush(stdout);
}

int main() {
    double *V, *out;
    const size_t n = (size_t)ROWS * NLON;
    hipMalloc(&V, n * 8 + 65536);
    hipMalloc(&out, (size_t)(ROWS / 64) * 256 * 8);
    hipMemset(V, 0, n * 8 + 65536);
    double* gt;
    hipMalloc(&gt, (size_t)ROWS * 193 * 8);
    run<64, 2, false, false>("64 B pieces, depth 2", V, out, gt);
    run<64, 3, false, false>("64 B pieces, depth 3", V, out, gt);
    run<128, 2, false, false>("128 B pieces, depth 2", V, out, gt);
    run<256, 2, false, false>("256 B pieces, depth 2", V, out, gt);
    run<128, 2, true, false>("contiguous, depth 2", V, out, gt);
    run<64, 2, false, true>("64 B pieces + gt writes", V, out, gt);
    run<128, 2, false, true>("128 B pieces + gt writes", V, out, gt);
    run<128, 2, true, true>("contiguous + gt writes", V, out, gt);
    return 0;
}
