// Tall-and-skinny fp64 product C = alpha A B + beta C with A [M x K] row-major (K-contiguous), B [K x N] row-major and N <= 240:
// the dense form of the DDK filters, W [14637^2] X [14637 x 240] (grates/filter.py:473-474, `GeneralMatrix.filter`: one numpy matmul per
// epoch in the reference).
//
// Why a kernel of its own.  In square output tiles (blas.hip: gemm_ex_kernel) this shape pays three ways: the 240 columns fill 3.75
// tiles of 64 (6 % of the MFMAs work on padding), the 916 tiles do not divide over the 768 workgroup slots of the card, and W crosses
// LDS although no two waves of a workgroup share a row of it.  Here
//   * an output tile is 128 rows x ALL columns (NF = ceil(N / 16) fragments of 16: no column padding beyond the last fragment);
//   * the eight waves of a workgroup own 16 rows each (NF accumulator fragments = 8 NF registers): their rows of A go from global memory straight
//     into the A-operand registers of the MFMAs -- lane (fr, fk) loads A[row fr][k0 + 2 fk .. + 1] and A[row fr][k0 + 8 + 2 fk .. + 1]
//     (two 16-byte loads, each instruction covering 64 contiguous bytes of 16 rows) and uses its four values in the four k-steps of
//     the K tile; B only has to present the SAME k to the lane, i.e. the k-step s of lane group fk reads row 2 fk + {0, 1, 8, 9}[s];
//   * the K tile of B (16 rows x N) is shared by the eight waves through LDS, copied there by LDS-DMA (no registers, no ds_write),
//     double-buffered, rows 16 NF + 8 doubles apart (conflict-free fragment reads); one barrier per K tile (4 NF = 60 MFMAs of a wave; K tiles of 32 rows measured no faster);
//   * the work is dealt stream-K: the (tile, K tile) iterations of the whole product form one sequence that is cut into G equal
//     ranges, one per workgroup (one per CU).  A range that covers a whole tile writes C itself; the at most two partial ends of a
//     range go to a workspace and a second kernel sums the pieces of every cut tile in the order of the ranges (a fixed order:
//     the result does not depend on which workgroup ran when).
#include <memory>

#include "common.h"

namespace shg {

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int kTallWaves = 8;                              // one workgroup of eight waves per CU
constexpr int kTallRows = 16 * kTallWaves;                 // rows of an output tile
constexpr int kTallK = 16;                                 // K tile
constexpr int kTallRowsPerWave = kTallK / kTallWaves;      // rows of a B tile that one wave copies
constexpr int kTallGroups = 256;                           // ranges of the iteration sequence = workgroups
#ifndef SHG_TALL_X
#define SHG_TALL_X 0          // timing experiments (wrong results): 1 the B tile is copied once per segment, 2 the A operand is loaded once per segment, 4 no fragment reads, 8 no barriers
#endif

struct TallParams {
    int M, N, K;
    const double* A;
    int lda;
    const double* B;
    int ldb;
    double* C;
    int ldc;
    double alpha, beta;
    double* pieces;           // [2 G][128][16 NF]: partial tiles, slot 2 w (first segment of range w) / 2 w + 1 (a later one)
    long long iters;          // tiles * nk
    int nk;                   // WHOLE K tiles of an output tile (the partial one behind them belongs to the segment that holds the last whole one)
    int groups;               // G
};

__host__ __device__ inline long long tall_range_start(long long w, long long iters, long long groups) { return w * iters / groups; }

// the range that holds iteration i
__device__ inline int tall_owner(long long i, long long iters, int groups) {
    long long w = i * groups / iters;
    if (w > groups - 1) w = groups - 1;
    while (w + 1 < groups && tall_range_start(w + 1, iters, groups) <= i) ++w;
    while (w > 0 && tall_range_start(w, iters, groups) > i) --w;
    return (int)w;
}

// LDS-DMA of one 16-byte piece per lane of `mask`: LDS address = lds_addr + 16 * lane (see synthesis_rot.hip: glds16).  The execution
// mask is set inside the statement: a branch around a partial copy would end the scheduling region in the middle of the MFMAs.
// exec is restored by the statement itself.  M0 is left overwritten and cannot be declared: hipcc treats it as reserved (a clobber entry
// only draws -Winline-asm "clobber list contains reserved registers" and changes nothing) and reloads it in front of every use of its
// own; gfx9 LDS instructions do not read it (synthesis_rot.hip: glds16).
__device__ __forceinline__ void tall_glds16(const double* gbase, unsigned lane_off, unsigned lds_addr, unsigned long long mask) {
    unsigned long long saved;
    asm volatile(
        "s_mov_b64 %0, exec\n\t"
        "s_mov_b64 exec, %4\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b64 exec, %0"
        : "=&s"(saved)
        : "v"(lane_off), "s"(gbase), "s"(lds_addr), "s"(mask)
        : "memory");
}

template <int NF>
__global__ __launch_bounds__(64 * kTallWaves, 1) void gemm_tall_kernel(TallParams P) {
    constexpr int NC = 16 * NF, LS = NC + 8;              // LS = 8 mod 16: the rows 2 fk and 2 fk + 2 of a fragment read are 16 banks apart
    constexpr int NDMA = (8 * NF + 63) / 64;              // DMA instructions per row of the B tile (16-byte pieces, 64 per instruction)
    constexpr int NCOPY = kTallRowsPerWave * NDMA;        // DMA instructions of a wave per K tile
    static_assert(2 * NCOPY < NF, "the copies of a K tile are dealt between the MFMAs of its first k-step");
    constexpr int LOADS_AT = 2 * NCOPY + 1 < NF ? 2 * NCOPY + 1 : NF - 1;      // the A loads behind the last copy
    extern __shared__ double tall_lds[];                  // [2][16][LS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const long long w = blockIdx.x;
    long long it = tall_range_start(w, P.iters, P.groups);
    const long long it_end = tall_range_start(w + 1, P.iters, P.groups);
    const unsigned lds_base = (unsigned)(size_t)tall_lds;
    const bool has_tail = (P.K % kTallK) != 0;
    // lanes of DMA instruction q that hold a piece of a row (pieces beyond N: never copied, their columns are never stored)
    unsigned long long dma_mask[NDMA];
#pragma unroll
    for (int q = 0; q < NDMA; ++q) {
        const int n = min(8 * NF, P.N / 2) - 64 * q;
        dma_mask[q] = n >= 64 ? ~0ull : n > 0 ? (1ull << n) - 1 : 0ull;
    }
    const unsigned dma_lane = (unsigned)lane * 16u;

    typedef const double2_t __attribute__((address_space(1))) gdouble2_t;
    typedef const char __attribute__((address_space(1))) gbyte_t;
    auto at2 = [](const double* base, unsigned byte_off) {
        unsigned long long b = reinterpret_cast<unsigned long long>(base);
        asm volatile("" : "+s"(b));
        return *reinterpret_cast<gdouble2_t*>(reinterpret_cast<gbyte_t*>(b) + byte_off);
    };

    for (int seg = 0; it < it_end; ++seg) {
        const int tile = (int)(it / P.nk), ka = (int)(it - (long long)tile * P.nk);
        const int kb = (int)min((long long)P.nk, (long long)ka + (it_end - it));
        const int m0 = tile * kTallRows, r0 = m0 + wave * 16;
        const double* arow = P.A + (size_t)m0 * P.lda;                     // uniform
        const unsigned aoff = (unsigned)(((size_t)(min(r0 + fr, P.M - 1) - m0) * P.lda + 2 * fk) * 8);

        // A operand of the whole K tile t: element s of areg belongs to k = 16 t + 2 fk + {0, 1, 8, 9}[s]
        auto load_a = [&](int t, double4_t& areg) {
            const double2_t lo = at2(arow + t * kTallK, aoff), hi = at2(arow + t * kTallK + 8, aoff);
            areg = (double4_t){lo.x, lo.y, hi.x, hi.y};
        };
        // copy i of the wave's share of the B tile of K tile t into LDS buffer buf (rows beyond K: the last row again -- finite values
        // that meet zeros of A)
        auto dma_b = [&](int t, int buf, int i) {
            const int jr = kTallRowsPerWave * wave + i / NDMA, q = i % NDMA;
            const double* src = P.B + (size_t)min(t * kTallK + jr, P.K - 1) * P.ldb + 128 * q;
            tall_glds16(src, dma_lane, lds_base + (unsigned)(((buf * kTallK + jr) * LS) * 8) + 1024u * q, dma_mask[q]);
        };

        double4_t acc[NF];
#pragma unroll
        for (int b = 0; b < NF; ++b) acc[b] = (double4_t){0.0, 0.0, 0.0, 0.0};
        typedef const volatile double __attribute__((address_space(3))) lds_double_t;
        lds_double_t* bl = (lds_double_t*)tall_lds + (2 * fk) * LS + fr;   // the lane's corner of a B tile
        // the fragments of k-step s + 1 are requested before the MFMAs of k-step s are issued (two fragment sets; the scheduling
        // barriers keep hipcc from hoisting all four sets in front of the first MFMA, which does not fit the register file)
        auto frags = [&](lds_double_t* Bs, int s, double (&bf)[NF]) {
            const int ro = ((s & 1) + 8 * (s >> 1)) * LS;
            if (SHG_TALL_X & 4) {
#pragma unroll
                for (int b = 0; b < NF; ++b) asm volatile("" : "+v"(bf[b]));
                return;
            }
            // (volatile: hipcc would pair the reads into ds_read2_b64 -- 8 LDS cycles per instruction on 32 banks where two
            //  ds_read_b64 take 2 + 2 on 64 banks)
#pragma unroll
            for (int b = 0; b < NF; ++b) bf[b] = *(lds_double_t*)(Bs + ro + 16 * b);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mfmas = [&](const double4_t& areg, int s, const double (&bf)[NF]) {
#pragma unroll
            for (int b = 0; b < NF; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[s], bf[b], acc[b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        // k-step 0 with the copies and the A loads of K tile tn dealt between its MFMAs: an MFMA holds the pipe for 64 cycles, the
        // instructions of a copy issue in its shadow -- in one run in front of the MFMAs they would idle the pipe of a SIMD whose two
        // waves leave the barrier together
        auto mfmas_deal = [&](const double4_t& areg, const double (&bf)[NF], int tn, int nbuf, double4_t& next) {
#pragma unroll
            for (int b = 0; b < NF; ++b) {
                acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[0], bf[b], acc[b], 0, 0, 0);
                if (b % 2 == 1 && b / 2 < NCOPY && !(SHG_TALL_X & 1)) {          // (hipcc would move the statements in front of the MFMAs)
                    __builtin_amdgcn_sched_barrier(0);
                    dma_b(tn, nbuf, b / 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (b == LOADS_AT && !(SHG_TALL_X & 2)) {
                    __builtin_amdgcn_sched_barrier(0);
                    load_a(tn, next);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // end of a K tile: the wave's DMA pieces and its next A operand have arrived (the empty statement makes hipcc place ITS wait for
        // the A loads here -- it cannot see the explicit one -- instead of in front of the first MFMA that follows loads of the tile after)
        auto publish = [&](double4_t& next) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(next.x), "+v"(next.y), "+v"(next.z), "+v"(next.w));
            if (!(SHG_TALL_X & 8)) __syncthreads();
        };
        // One whole K tile out of buffer `buf` (set 0 holds the fragments of its k-step 0), K tile tn on its way into the other buffer.
        // The loop is rotated across the barrier: the last k-step is issued behind the barrier that publishes the next tile and behind
        // the request for that tile's first fragments.  No branches: the tile after the last one of a segment is that tile again.
        double bf0[NF], bf1[NF];
#pragma unroll
        for (int b = 0; b < NF; ++b) bf0[b] = bf1[b] = (SHG_TALL_X & 4) ? 1.0 + 0.37 * fr + 0.11 * b : 0.0;     // (no fragment reads: operands that toggle the multipliers like real ones)
        auto tile_step = [&](int buf, const double4_t& areg, double4_t& next, int tn) {
            lds_double_t* Bs = bl + buf * kTallK * LS;
            frags(Bs, 1, bf1);
            mfmas_deal(areg, bf0, tn, buf ^ 1, next);
            frags(Bs, 2, bf0);
            mfmas(areg, 1, bf1);
            frags(Bs, 3, bf1);
            mfmas(areg, 2, bf0);
            publish(next);
            frags(bl + (buf ^ 1) * kTallK * LS, 0, bf0);
            mfmas(areg, 3, bf1);
        };

        double4_t a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
        if (ka < kb) {
            int t = ka;
#pragma unroll
            for (int i = 0; i < NCOPY; ++i) dma_b(t, 0, i);
            load_a(t, a0);
            publish(a0);
            frags(bl, 0, bf0);
            while (true) {
                tile_step(0, a0, a1, min(t + 1, kb - 1));
                if (++t >= kb) break;
                tile_step(1, a1, a0, min(t + 1, kb - 1));
                if (++t >= kb) break;
            }
        }
        if (has_tail && kb == P.nk) {                      // the partial K tile behind the whole ones: entries of A beyond K are zero
            __syncthreads();                               // (the fragment requests of the loop's last step are served)
#pragma unroll
            for (int i = 0; i < NCOPY; ++i) dma_b(P.nk, 0, i);
            const double* row = reinterpret_cast<const double*>(reinterpret_cast<const char*>(arow) + aoff) - 2 * fk;
            double v[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = P.nk * kTallK + 2 * fk + (s & 1) + 8 * (s >> 1);
                const double x = row[min(k, P.K - 1)];
                v[s] = k < P.K ? x : 0.0;
            }
            a0 = (double4_t){v[0], v[1], v[2], v[3]};
            publish(a0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                frags(bl, s, bf0);
                mfmas(a0, s, bf0);
            }
        }

        // C/D layout: column = lane & 15, row = (lane >> 4) + 4 * element
        if (ka == 0 && kb == P.nk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gr = r0 + fk + 4 * i;
                if (gr >= P.M) continue;
#pragma unroll
                for (int b = 0; b < NF; ++b) {
                    const int gc = 16 * b + fr;
                    if (gc >= P.N) continue;
                    double* c = P.C + (size_t)gr * P.ldc + gc;
                    const double v = P.alpha * acc[b][i];
                    *c = P.beta == 0.0 ? v : fma(P.beta, *c, v);
                }
            }
        } else {
            double* piece = P.pieces + ((size_t)2 * w + (seg > 0 ? 1 : 0)) * kTallRows * NC;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int b = 0; b < NF; ++b) piece[(size_t)(wave * 16 + fk + 4 * i) * NC + 16 * b + fr] = acc[b][i];
        }
        __syncthreads();                                   // the next segment's first copies overwrite buffer 0
        it += kb - ka;
    }
}

// sums the pieces of every tile that was cut between ranges, in the order of the ranges.  grid (tiles, rows of a tile / 2): a block
// handles two rows, a thread two adjacent columns (N is even, the pieces are 16-byte aligned); the loads of up to four pieces are in
// flight together
__global__ __launch_bounds__(256) void gemm_tall_fixup_kernel(TallParams P, int NC) {
    const int tile = blockIdx.x, row = 2 * blockIdx.y + (threadIdx.x >> 7), gr = tile * kTallRows + row;
    const int col = 2 * (threadIdx.x & 127);
    if (gr >= P.M || col >= P.N) return;
    const long long t0 = (long long)tile * P.nk, t1 = t0 + P.nk;
    const int w_first = tall_owner(t0, P.iters, P.groups), w_last = tall_owner(t1 - 1, P.iters, P.groups);
    if (w_first == w_last) return;                         // written by its only range
    const size_t at = (size_t)row * NC + col;
    auto piece_of = [&](int w) {
        const long long start = tall_range_start(w, P.iters, P.groups);
        return reinterpret_cast<const double2_t*>(P.pieces + ((size_t)2 * w + (start < t0 ? 1 : 0)) * kTallRows * NC + at);
    };
    double2_t sum = {0.0, 0.0};
    int w = w_first;
    for (; w + 3 <= w_last; w += 4) {                      // (ranges are never empty here: iters >= groups)
        const double2_t p0 = *piece_of(w), p1 = *piece_of(w + 1), p2 = *piece_of(w + 2), p3 = *piece_of(w + 3);
        sum += p0;
        sum += p1;
        sum += p2;
        sum += p3;
    }
    for (; w <= w_last; ++w) sum += *piece_of(w);
    double* c = P.C + (size_t)gr * P.ldc + col;
    const double v0 = P.alpha * sum.x, v1 = P.alpha * sum.y;
    c[0] = P.beta == 0.0 ? v0 : fma(P.beta, c[0], v0);
    c[1] = P.beta == 0.0 ? v1 : fma(P.beta, c[1], v1);
}

bool gemm_tall_shape(bool ta, bool tb, int M, int N, int K, int batch, bool upper_only, int tri, const double* A, int lda, const double* B, int ldb,
                     const double* C) {
    if (ta || tb || batch != 1 || upper_only || tri != 0) return false;
    if (N <= 176 || N > 240 || (N & 1) || (ldb & 1) || (reinterpret_cast<size_t>(B) & 15)) return false;     // 12 .. 15 column fragments; 16-byte pieces of the rows of B
    if (M < 2048 || K < 2048 || (long long)lda * kTallRows * 8 >= (1ll << 32)) return false;
    return C != A && C != B;
}

template <int NF>
static int gemm_tall_launch(TallParams P, hipStream_t stream) {
    constexpr int NC = 16 * NF;
    const int tiles = ceil_div(P.M, kTallRows);
    P.iters = (long long)tiles * P.nk;
    P.groups = (int)std::min<long long>(kTallGroups, P.iters);
    ScratchLease lease(stream);
    P.pieces = (double*)lease.get(kScratchSplitK, (size_t)2 * P.groups * kTallRows * NC * sizeof(double));
    if (P.pieces == nullptr) return fail(SHG_ERR_HIP, "gemm_tall: no workspace for the partial tiles");
    const size_t lds = (size_t)2 * kTallK * (NC + 8) * sizeof(double);
    SHG_SET_LDS_ONCE((gemm_tall_kernel<NF>), lds);
    hipLaunchKernelGGL((gemm_tall_kernel<NF>), dim3(P.groups), dim3(64 * kTallWaves), lds, stream, P);
    hipLaunchKernelGGL(gemm_tall_fixup_kernel, dim3(tiles, kTallRows / 2), dim3(256), 0, stream, P, NC);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

int gemm_tall(int M, int N, int K, double alpha, const double* A, int lda, const double* B, int ldb, double beta, double* C, int ldc, hipStream_t stream) {
    TallParams P;
    P.M = M;
    P.N = N;
    P.K = K;
    P.A = A;
    P.lda = lda;
    P.B = B;
    P.ldb = ldb;
    P.C = C;
    P.ldc = ldc;
    P.alpha = alpha;
    P.beta = beta;
    P.nk = K / kTallK;
    P.pieces = nullptr;
    P.iters = 0;
    P.groups = 0;
    switch (ceil_div(N, 16)) {                     // column fragments of a tile
        case 12: return gemm_tall_launch<12>(P, stream);
        case 13: return gemm_tall_launch<13>(P, stream);
        case 14: return gemm_tall_launch<14>(P, stream);
        case 15: return gemm_tall_launch<15>(P, stream);
    }
    return fail(SHG_ERR_INVALID, "gemm_tall: %d columns", N);
}

}  // namespace shg
