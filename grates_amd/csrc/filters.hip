// Order-wise block filter (DDK family):  replaces OrderWiseFilter.filter, grates/filter.py:180-191.
//   for every order m and cosine / sine:  y[m..N] = W_block[0:N+1-m, 0:N+1-m] x[m..N]   for all epochs at once,
//   degrees 0 and 1 copied from the input.
// HBM-bound integer-indexed streaming: each block matrix is read once per tile of 32 epochs, coefficient
// vectors are staged in LDS.
#include "common.h"

namespace shg {

#ifndef SHG_FILT_X
#define SHG_FILT_X 0      // experiment switches (timing only, -DSHG_EXPERIMENT builds): 1 no gather, 2 no products, 4 no scatter
#endif
#ifndef SHG_FILT_GROUP
#define SHG_FILT_GROUP 0  // orders per workgroup: 0 = the largest of 8, 4, 2 whose LDS stage fits
#endif

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));

constexpr int kOwEpochs = 16;       // epochs per workgroup = one MFMA column tile
constexpr int kOwPitch = 17;        // doubles per LDS row xs[k][epoch]: lanes along k (sine gather / scatter) and lanes along the epochs
                                    // (cosine gather / scatter, MFMA fragments) both spread over the banks
#ifndef SHG_FILT_WAVES
#define SHG_FILT_WAVES 16
#endif
constexpr int kOwWaves = SHG_FILT_WAVES;    // waves per workgroup
constexpr int kOwThreads = 64 * kOwWaves;
constexpr int kOwMaxUnits = 96 / kOwWaves;  // (order, row tile) units per wave
#ifndef SHG_FILT_DEPTH
#define SHG_FILT_DEPTH 16
#endif
constexpr int kOwDepth = SHG_FILT_DEPTH;    // loads of a thread in flight in the gather
__device__ __host__ inline int ow_ceil_div(int a, int b) { return (a + b - 1) / b; }

// One workgroup = G consecutive orders (all cosine or all sine blocks) x 16 epochs, 16 waves.
//   Why G orders together: in the coefficient array of an epoch ([n][m]: C_nm at [n][m], S_nm at [m-1][n], utilities.py:310-411) the
//   cosine coefficients of an order are a COLUMN piece, 8 (N+1) bytes apart -- one 64-byte sector per element whichever way the
//   lanes run (round 3: 3.6 cycles per element and CU for the gather and again for the scatter, 0.84 TB/s; every sector fetched
//   and written back once per order that has 8 bytes in it).  The same sectors hold the cosine coefficients of the neighbouring
//   orders: with G = 8 orders per workgroup the lanes run along the orders and a sector is touched once.
//   gather   cosine: lanes (order, epoch): 64-byte runs [n][m0 .. m0+7]; sine: lanes along the degree, whole runs [m-1][m .. N];
//            staged in LDS as xs[order][k][epoch]
//   product  Y_m [n x 16 epochs] = W_m [n x n] X_m on the fp64 MFMA: unit = (order, tile of 16 rows), dealt to the waves;
//            A fragments (block entries) straight from L2 with one 16-byte buffer load per lane and pair of k-steps (16 rows x 64
//            bytes per wave-instruction, rows clamped, beyond the packed blocks the range check returns 0), B fragments from xs;
//            the rows k >= n of xs are zero, so that whatever the A fragments hold beyond their block's columns does not count.
//            (Round 3 formed the products with v_fma_f64 and scalar loads of the block entries: 40 of its 74 us.)
//   scatter  the results replace the input in xs once every wave has finished reading it; then like the gather.
// HBM/L2 bound: bytes = 8 (sum of block sizes x epoch groups from L2 + 2 P B).
template <int G>
__global__ __launch_bounds__(kOwThreads) void orderwise_filter_kernel(int Nb, int N, int B, int ngc, int ngs, int negroups, int plane, const double* __restrict__ blocks,
                                                               const long long* __restrict__ block_off,
                                                               const double* __restrict__ in, double* __restrict__ out) {
    extern __shared__ double xs[];                 // [G][plane]: plane j holds xs[k][epoch] of order m0 + j, k < npad, pitch 17
    // Workgroup -> (order group, epoch group).  Every epoch group of an order group reads the same blocks (G blocks, ~1 MB at d/o 120),
    // and all blocks together (9.4 MB) do not fit the L2 of an XCD (4 MB): dealt out group-major over all XCDs (the first version of
    // this kernel) the blocks streamed in from beyond the L2 once per epoch group -- 141 MB for 240 epochs, 40 of its 80 us.
    // Workgroups go to the XCDs round robin by their linear index: XCD x = blockIdx % 8 takes the order groups x, x + 8, ... one
    // after the other, all epoch groups of one before the next, so that its L2 holds the blocks of the one or two order groups
    // its CUs are working on.
    const int ngrp = ngc + ngs;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int grp = 8 * (seq / negroups) + xcd, eg = seq % negroups;
    if (grp >= ngrp) return;
    // groups in the order c0, s0, c1, s1, ...: the long blocks first
    const bool sine = grp < 2 * ngs ? (grp & 1) != 0 : false;
    const int gi = grp < 2 * ngs ? grp >> 1 : grp - ngs;
    const int m0 = sine ? 1 + G * gi : G * gi;     // first order of the group
    const int n0 = N + 1 - m0;                     // coefficients of order m0 in the field
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = eg * kOwEpochs;
    const size_t E = (size_t)(N + 1) * (N + 1);
    const int ld_in = N + 1;

    // the rows k >= n of every plane are zero (the gather fills the others): at most 7 + j of them below the padded length
    {
        const int npad = 8 * ow_ceil_div(n0, 8);
        for (int i = tid; i < G * 16 * kOwPitch; i += kOwThreads) {
            const int j = i / (16 * kOwPitch), k = npad - 1 - (i / kOwPitch) % 16, e = i % kOwPitch;
            if (k >= n0 - j && k >= 0) xs[j * plane + k * kOwPitch + e] = 0.0;
        }
    }

    // ---- gather
    if (!(SHG_FILT_X & 1)) {
        if (!sine) {
            // (row n, epoch e, order j), j fastest: C_n,m0+j at [n][m0 + j]
            const int total = n0 * kOwEpochs * G;
            for (int i0 = 0; i0 < total; i0 += kOwDepth * kOwThreads) {
                double v[kOwDepth];
#pragma unroll
                for (int u = 0; u < kOwDepth; ++u) {
                    const int i = min(i0 + u * kOwThreads + tid, total - 1);
                    const int j = i % G, e = (i / G) % kOwEpochs, n = m0 + i / (G * kOwEpochs);
                    v[u] = in[(size_t)min(b0 + e, B - 1) * E + (size_t)n * ld_in + min(m0 + j, n)];      // (clamped, not tested: no branch around the loads)
                }
#pragma unroll
                for (int u = 0; u < kOwDepth; ++u) {
                    const int i = i0 + u * kOwThreads + tid;
                    const int j = i % G, e = (i / G) % kOwEpochs, n = m0 + i / (G * kOwEpochs);
                    if (i < total && m0 + j <= n) xs[j * plane + (n - m0 - j) * kOwPitch + e] = b0 + e < B ? v[u] : 0.0;
                }
            }
        } else {
            // (order j, epoch e, degree k), k fastest: S_m+k,m at [m - 1][m + k], m = m0 + j
            const int total = G * kOwEpochs * n0;
            for (int i0 = 0; i0 < total; i0 += kOwDepth * kOwThreads) {
                double v[kOwDepth];
#pragma unroll
                for (int u = 0; u < kOwDepth; ++u) {
                    const int i = min(i0 + u * kOwThreads + tid, total - 1);
                    const int k = i % n0, e = (i / n0) % kOwEpochs, m = min(m0 + i / (n0 * kOwEpochs), N);
                    v[u] = in[(size_t)min(b0 + e, B - 1) * E + (size_t)(m - 1) * ld_in + min(m + k, N)];
                }
#pragma unroll
                for (int u = 0; u < kOwDepth; ++u) {
                    const int i = i0 + u * kOwThreads + tid;
                    const int k = i % n0, e = (i / n0) % kOwEpochs, j = i / (n0 * kOwEpochs);
                    if (i < total && m0 + j + k <= N) xs[j * plane + k * kOwPitch + e] = b0 + e < B ? v[u] : 0.0;
                }
            }
        }
    }
    __syncthreads();

    // ---- products
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(blocks), 0, (int)((block_off[2 * Nb] + 1) * 8), 0x00020000);
    const int fr = lane & 15, fk = lane >> 4;
    double4_t res[kOwMaxUnits];
    int unit_j[kOwMaxUnits], unit_r0[kOwMaxUnits];
    {
        // unit u of the group = (order j, row tile rt) in order of j; this wave takes the units wave, wave + waves, ...  The orders
        // of a group have T or T - 1 row tiles (n_j = n0 - j, G <= 16): the first jf orders T, the others T - 1.
        const int T = ow_ceil_div(n0, 16), jf = min((n0 - 1) % 16 + 1, G);
#pragma unroll
        for (int q = 0; q < kOwMaxUnits; ++q) {
            const int u = wave + kOwWaves * q;
            int j, rt;
            if (u < jf * T) {
                j = u / T;
                rt = u % T;
            } else if (T > 1) {
                j = jf + (u - jf * T) / (T - 1);
                rt = (u - jf * T) % (T - 1);
            } else {
                j = G;
                rt = 0;
            }
            const bool ok = j < G && j < n0 && 16 * rt < n0 - j;
            unit_j[q] = ok ? j : -1;
            unit_r0[q] = 16 * rt;
        }
    }
#pragma unroll
    for (int q = 0; q < kOwMaxUnits; ++q) {
        double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        const int j = unit_j[q];
        if (j >= 0 && !(SHG_FILT_X & 2)) {
            const int m = m0 + j, nj = n0 - j, ld = Nb + 1 - m;
            const int kb = m == 0 ? 0 : 2 * m - (sine ? 0 : 1);                     // 0: order 0 cos, 2m-1: order m cos, 2m: order m sin
            // (the whole offset in the vector operand: the range check of a raw buffer does not cover the scalar offset)
            const unsigned voff = (unsigned)(block_off[kb] * 8) + (unsigned)((min(unit_r0[q] + fr, nj - 1) * ld + 2 * fk) * 8);
            const double* xb = xs + j * plane + 2 * fk * kOwPitch + fr;
            // columns c + 2 fk, c + 2 fk + 1 of the block rows (A fragments of two k-steps), rows c + 2 fk, c + 2 fk + 1 of xs; the A
            // fragments come from L2 four pairs of k-steps ahead of their use (a ring of four registers pairs)
            auto lda = [&](int c) { return __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, voff + (unsigned)c * 8u, 0, 0)); };
            auto step = [&](const double2_t a, int c) {
                const double x0 = xb[c * kOwPitch], x1 = xb[(c + 1) * kOwPitch];
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, x0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, x1, acc1, 0, 0, 0);
            };
            double2_t a0 = lda(0), a1 = lda(8), a2 = lda(16), a3 = lda(24);
            for (int c = 0; c < nj; c += 32) {
                step(a0, c);
                a0 = lda(c + 32);
                if (c + 8 < nj) step(a1, c + 8);
                a1 = lda(c + 40);
                if (c + 16 < nj) step(a2, c + 16);
                a2 = lda(c + 48);
                if (c + 24 < nj) step(a3, c + 24);
                a3 = lda(c + 56);
            }
        }
        res[q] = acc0 + acc1;
    }
    __syncthreads();                                  // every wave has read what it needs of xs
#pragma unroll
    for (int q = 0; q < kOwMaxUnits; ++q) {
        const int j = unit_j[q];
        if (j >= 0) {
            const int m = m0 + j, nj = n0 - j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = unit_r0[q] + fk + 4 * r;                            // C/D layout: row = fk + 4 reg, column = fr
                if (row < nj && m + row > 1) xs[j * plane + row * kOwPitch + fr] = res[q][r];     // degrees 0 and 1 keep the input (filter.py:189)
            }
        }
    }
    __syncthreads();

    // ---- scatter
    if (!(SHG_FILT_X & 4)) {
        if (!sine) {
            const int total = n0 * kOwEpochs * G;
            for (int i = tid; i < total; i += kOwThreads) {
                const int j = i % G, e = (i / G) % kOwEpochs, n = m0 + i / (G * kOwEpochs);
                if (m0 + j <= n && b0 + e < B) out[(size_t)(b0 + e) * E + (size_t)n * ld_in + m0 + j] = xs[j * plane + (n - m0 - j) * kOwPitch + e];
            }
        } else {
            const int total = G * kOwEpochs * n0;
            for (int i = tid; i < total; i += kOwThreads) {
                const int k = i % n0, e = (i / n0) % kOwEpochs, j = i / (n0 * kOwEpochs);
                const int m = m0 + j;
                if (m + k <= N && b0 + e < B) out[(size_t)(b0 + e) * E + (size_t)(m - 1) * ld_in + m + k] = xs[j * plane + k * kOwPitch + e];
            }
        }
    }
}

}  // namespace shg

using namespace shg;

// LDS plane of one order: round_up(N + 1, 8) rows of 17 doubles, padded so that planes lie 16 / G + 1 (mod 16) doubles apart:
// the lanes (order, epoch) of the cosine gather then hit distinct banks
static int orderwise_plane(int N, int G) {
    const int rows = round_up(N + 1, 8) * kOwPitch;
    return rows + ((1 + 16 / G - rows) % 16 + 16) % 16;
}

template <int G>
static int launch_orderwise(int Nb, int N, int B, const double* blocks, const long long* block_off, const double* in, double* out, hipStream_t stream) {
    const int plane = orderwise_plane(N, G);
    const size_t lds = (size_t)G * plane * sizeof(double);
    if (lds > 64 * 1024) SHG_HIP(hipFuncSetAttribute((const void*)orderwise_filter_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int ngc = ceil_div(N + 1, G), ngs = ceil_div(N, G);       // groups of cosine (orders 0 .. N) and sine (1 .. N) blocks
    const int negroups = ceil_div(B, kOwEpochs);
    hipLaunchKernelGGL(orderwise_filter_kernel<G>, dim3((unsigned)(8 * ceil_div(ngc + ngs, 8) * negroups)), dim3(kOwThreads), lds, stream, Nb, N, B, ngc, ngs, negroups, plane,
                       blocks, block_off, in, out);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

#ifndef SHG_FILT_VIA_SERIES
#define SHG_FILT_VIA_SERIES 1
#endif
constexpr int kSeriesMinEpochs = 64;
#ifndef SHG_OM_EPOCH_GROUPS
#define SHG_OM_EPOCH_GROUPS 1     // groups of 32 epochs per workgroup of orderwise_filter_om_kernel (1 with three waves per SIMD: 25.8 us; 2 with two: 28.5)
#endif
// LDS of order_major_kernel: one row of the reference arrays for 16 epochs (above 64 KB the launch needs the opt-in attribute)
constexpr size_t kOrderMajorMaxLds = 160 * 1024;
static size_t order_major_lds_bytes(int N) { return (size_t)16 * (N + 2) * sizeof(double); }
extern "C" int shg_order_major_pack(const double* anm, int N, int B, double* om, int Bpad, void* stream_);
extern "C" int shg_order_major_unpack(const double* om, int N, int B, int Bpad, double* anm, void* stream_);
extern "C" int shg_orderwise_filter_om(const double* blocks_packed, const int64_t* block_off, int Nb, int N, const double* om_in, int B, int Bpad,
                                       double* om_out, void* stream_);

extern "C" int shg_orderwise_filter(const double* blocks_packed, const int64_t* block_off, int Nb, int N, const double* anm_in, int B,
                                    double* anm_out, void* stream_) {
    SHG_REQUIRE(Nb >= 0 && N >= 0 && B >= 0, "shg_orderwise_filter: negative size");
    SHG_REQUIRE(N <= Nb, "DDK filter only implemented for a maximum degree of %d (max_degree=%d supplied).", Nb, N);
    if (B == 0) return SHG_OK;
    SHG_REQUIRE(blocks_packed && block_off && anm_in && anm_out, "shg_orderwise_filter: NULL pointer");
    SHG_REQUIRE(anm_in != anm_out, "shg_orderwise_filter: in-place operation is not supported");
    hipStream_t stream = (hipStream_t)stream_;
    const long long* off = (const long long*)block_off;
    // Batches of many epochs go through the order-major layout (below): pack 14 us + one product per block on whole matrices 30 us +
    // unpack 12 us at d/o 120 and 240 epochs, against 63 us for the kernel that gathers from and scatters to the reference layout --
    // the same products in the same order, bit-identical results.
    if (SHG_FILT_VIA_SERIES && B >= kSeriesMinEpochs) {
        const int Bpad = round_up(B, 32);
        const size_t bytes = (size_t)(N + 1) * (N + 1) * Bpad * sizeof(double);
        ScratchLease lease(stream);
        double* om_in = (double*)lease.get(kScratchSeriesIn, bytes);
        double* om_out = (double*)lease.get(kScratchSeriesOut, bytes);
        // (only where the row stage of the layout kernels fits the LDS; whatever fails on this way, the block kernel below still serves)
        if (om_in && om_out && order_major_lds_bytes(N) <= kOrderMajorMaxLds) {
            int rc = shg_order_major_pack(anm_in, N, B, om_in, Bpad, stream_);
            if (!rc) rc = shg_orderwise_filter_om(blocks_packed, block_off, Nb, N, om_in, B, Bpad, om_out, stream_);
            if (!rc) rc = shg_order_major_unpack(om_out, N, B, Bpad, anm_out, stream_);
            if (!rc) return rc;
        }
    }
    // orders per workgroup: as many as the LDS stage allows (8: degree <= 143, 4: <= 295, 2: <= 591); every wave holds the results of
    // its (order, row tile) units in registers
    auto fits = [&](int G) {
        return (size_t)G * orderwise_plane(N, G) * sizeof(double) <= 160 * 1024 && ceil_div(G * ceil_div(N + 1, 16), kOwWaves) <= kOwMaxUnits;
    };
    const int want = SHG_FILT_GROUP;
    if ((want == 0 || want == 8) && fits(8)) return launch_orderwise<8>(Nb, N, B, blocks_packed, off, anm_in, anm_out, stream);
    if ((want == 0 || want == 4) && fits(4)) return launch_orderwise<4>(Nb, N, B, blocks_packed, off, anm_in, anm_out, stream);
    if (fits(2)) return launch_orderwise<2>(Nb, N, B, blocks_packed, off, anm_in, anm_out, stream);
    return fail(SHG_ERR_UNSUPPORTED, "shg_orderwise_filter: degree %d exceeds the LDS staging of the block kernel (max degree 591)", N);
}

// ------------------------------------------------------------------------------------------------
// Order-major series (round 5): a time series of coefficient sets that STAYS on the device between operators, laid out for them.
//
//   om [P = (N+1)^2 rows][Bpad epochs]   row of (slot s, degree offset k) = om_row(N, s) + k,   s = 0: order 0 cosine,
//                                        2m - 1: order m cosine, 2m: order m sine (the order of the DDK block list), k = n - m
//
// Epochs are the fastest index (Bpad = B rounded up to 32: every row is a whole number of 256-byte runs), the coefficients of one order
// are one contiguous [n x Bpad] matrix.  The order-wise filter is then Y_s = W_s X_s on whole matrices -- no gather, no
// scatter (35 of the 58 us of orderwise_filter_kernel on the reference layout) --, and the coefficient repack of the fused
// synthesis kernels reads four consecutive epochs of a degree with one 32-byte access.
// shg_order_major_pack / _unpack convert from / to the reference layout anm [B][N+1][N+1] (C_nm at [n][m], S_nm at [m-1][n]).
// ------------------------------------------------------------------------------------------------
namespace shg {

__host__ __device__ inline long long om_row(int N, int s) {        // first row of slot s
    if (s == 0) return 0;
    const int m = (s + 1) >> 1;
    const long long cos_row = (long long)(N + 1) + 2LL * ((long long)(m - 1) * (N + 1) - (long long)m * (m - 1) / 2);
    return (s & 1) ? cos_row : cos_row + (N + 1 - m);
}

// One workgroup = row r of the reference arrays of 16 epochs: [r][c <= r] are C_r,c (slot of order c cosine, k = r - c), [r][c > r]
// are S_c,r+1 (slot of order r + 1 sine, k = c - r - 1).  Reads whole rows, writes 128-byte pieces (16 epochs of one coefficient).
template <bool PACK>
__global__ __launch_bounds__(256) void order_major_kernel(int N, int B, int Bpad, const double* __restrict__ src, double* __restrict__ dst) {
    extern __shared__ double tile[];               // [16 epochs][N + 2]
    const int r = blockIdx.x, b0 = blockIdx.y * 16, tid = threadIdx.x;
    const int ld = N + 2;
    const size_t E = (size_t)(N + 1) * (N + 1);
    auto om_index = [&](int c) {
        const int s = c <= r ? (c == 0 ? 0 : 2 * c - 1) : 2 * (r + 1);
        const int k = c <= r ? r - c : c - r - 1;
        return (size_t)(om_row(N, s) + k) * Bpad;
    };
    if (PACK) {
        for (int i = tid; i < 16 * (N + 1); i += 256) {
            const int e = i / (N + 1), c = i % (N + 1);
            tile[e * ld + c] = b0 + e < B ? src[(size_t)(b0 + e) * E + (size_t)r * (N + 1) + c] : 0.0;
        }
        __syncthreads();
        for (int i = tid; i < 16 * (N + 1); i += 256) {
            const int c = i >> 4, e = i & 15;
            if (b0 + e < Bpad) dst[om_index(c) + b0 + e] = tile[e * ld + c];
        }
    } else {
        for (int i = tid; i < 16 * (N + 1); i += 256) {
            const int c = i >> 4, e = i & 15;
            tile[e * ld + c] = b0 + e < B ? src[om_index(c) + b0 + e] : 0.0;
        }
        __syncthreads();
        for (int i = tid; i < 16 * (N + 1); i += 256) {
            const int e = i / (N + 1), c = i % (N + 1);
            if (b0 + e < B) dst[(size_t)(b0 + e) * E + (size_t)r * (N + 1) + c] = tile[e * ld + c];
        }
    }
}

// Y_s [n x Bpad] = W_s [n x n] X_s [n x Bpad], one workgroup per (slot s, slice of 32 EG epochs); wave w takes the row tiles w, w + 4, ...
// (a pass of four row tiles at a time).  The K loop is a pipeline over chunks of kOmChunk rows of the slice: the chunk in use sits in one of
// two LDS buffers, the next one in registers (written to the other buffer behind the products), the one after that is in flight -- and so
// are the block entries of those chunks (one 16-byte range-checked buffer load per lane and step of eight columns).  The barrier of the
// pipeline waits for the LDS only (s_waitcnt lgkmcnt(0); s_barrier): __syncthreads would drain the loads in flight.
// History of this kernel (kernel time per 240 epochs at d/o 120; 66 MB of algorithmic traffic = 8 us of HBM time, 9 us of MFMA time):
//   round 5: one workgroup per (slot, row tile), rows of X from L2 inside the K loop, three steps in flight                       30 us
//   round 6: the slice of X staged in the LDS at once, then all products -- whatever the staging (4 or 16 loads in flight), the
//            window of block entries (4 steps or a whole row tile) or the slice width (32 | 64 epochs)                            27-29 us
//            (every workgroup of a round stages at the same time -- 32 MB in one burst, nothing computes --, then every workgroup computes
//            and the HBM idles: SQ counters show the MFMA pipe busy 33 % of the time, the waves waiting half of theirs)
//   this form: loads and products of a workgroup overlap, workgroups in different phases share a CU                                25.8 us
//            (32 epochs per workgroup, three waves per SIMD; 64 epochs at two waves per SIMD 28.5, 32 epochs at four 29.5)
// Knock-outs of this form (SHG_OM_X): without the products 15.3 us, without the loads of X 23.5, without the block entries 24.5: the
// kernel takes the SUM of its memory time (66 MB at 4.3 TB/s) and its MFMA time (338 000 MFMAs = 9 us on 1024 SIMDs), not their maximum.
// The products and their order are those of orderwise_filter_kernel, the results bit-identical: step c multiplies W[row][c + 2 fk],
// W[row][c + 2 fk + 1] with the rows c + 2 fk, c + 2 fk + 1 of X in two MFMAs per 16 epochs, the two partial sums are added at the end.  A
// lane reads two adjacent epochs (16 bytes) of a row of X -- the B operands of two MFMAs whose columns are the even and the odd epochs of
// a group of 32 -- and therefore also OWNS two adjacent epochs of four result rows: it stores 16 bytes.  Degrees 0 and 1 keep the input
// (filter.py:189).
// Workgroup b runs on XCD b % 8: all slices of a slot go to one XCD (they share the slot's block in its L2), the slots -- by decreasing
// length, i.e. in their own order -- in a snake over the XCDs so that every XCD gets about the same arithmetic, the long ones first.
#ifndef SHG_OM_X
#define SHG_OM_X 0      // experiment switches (timing only, wrong results): 1 no products, 2 no stores, 4 no loads of X, 8 no loads of block entries
#endif
constexpr int kOmChunk = 32;           // rows of the slice per pipeline stage = four steps of eight columns
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifndef SHG_OM_WAVES_PER_EU
#define SHG_OM_WAVES_PER_EU 3
#endif
template <int EG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SHG_OM_WAVES_PER_EU, SHG_OM_WAVES_PER_EU))) void orderwise_filter_om_kernel(int Nb, int N, int Bpad, int nslice, const double* __restrict__ blocks,
                                                                  const long long* __restrict__ block_off, const double* __restrict__ in,
                                                                  double* __restrict__ out) {
    constexpr int PR = 16 * EG;                                                 // double2 per staged row
    constexpr int RPP = 256 / PR;                                               // rows per pass of the loader threads
    constexpr int NLD = kOmChunk / RPP;                                         // loads of a thread per chunk
    constexpr int NST = kOmChunk / 8;                                           // steps per chunk
    __shared__ __attribute__((aligned(16))) double2_t om_stage[2][kOmChunk * PR];
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int rank = seq / nslice, slice = seq % nslice;
    const int s = 8 * rank + ((rank & 1) ? 7 - xcd : xcd);
    if (s > 2 * N) return;
    const int m = (s + 1) >> 1, n = N + 1 - m, ld = Nb + 1 - m;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const int ldx = Bpad / 2;                                                   // double2 per row of the series
    const int p0 = slice * PR;                                                  // first pair of epochs of the slice
    const int npair = min(PR, ldx - p0);                                        // pairs that exist (a multiple of 16)
    const size_t row_first = (size_t)om_row(N, s);
    const double2_t* X = reinterpret_cast<const double2_t*>(in + row_first * Bpad) + p0;
    double2_t* Y = reinterpret_cast<double2_t*>(out + row_first * Bpad) + p0;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(blocks), 0, (int)((block_off[2 * Nb] + 1) * 8), 0x00020000);
    const int Kpad = (n + 7) & ~7, ntile = (n + 15) >> 4;
    const int nchunk = (Kpad + kOmChunk - 1) / kOmChunk;
    const double2_t zero2 = {0.0, 0.0};
    const int lj = tid % PR, lr = tid / PR;                                     // loader: pair lj of the rows lr, lr + RPP, ...
    const bool pair_ok = lj < npair;
    struct Chunk {
        double2_t x[NLD];                   // this thread's part of the chunk's rows of X
        double2_t a[NST];                   // this lane's block entries of the chunk's steps
    };
    for (int t0 = 0; t0 < ntile; t0 += 4) {
        // (the tiles of a pass go to the waves in an order that turns with the workgroup: wave w of every workgroup runs on SIMD w, and a slot of 5 row
        //  tiles has 2, 1, 1, 1 of them for its four waves -- unturned, SIMD 0 carried 1.25 times the average and SIMD 3 0.7 times)
        const int tile = t0 + ((wave + rank + slice) & 3);
        const bool active = tile < ntile;
        const int r0 = 16 * min(tile, ntile - 1);                               // (idle waves load the last tile's entries: the same count of loads in every wave)
        const unsigned voff = (unsigned)(block_off[s] * 8) + (unsigned)((min(r0 + fr, n - 1) * ld + 2 * fk) * 8);
        auto fetch = [&](Chunk& c, int j) {                                     // (beyond the packed blocks the range check returns 0; rows beyond the block are zero)
            const int k0 = j * kOmChunk;
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                const int k = k0 + lr + u * RPP;
                c.x[u] = (k < n && pair_ok && !(SHG_OM_X & 4)) ? X[(size_t)k * ldx + lj] : zero2;
            }
#pragma unroll
            for (int d = 0; d < NST; ++d)
                c.a[d] = (SHG_OM_X & 8) ? (double2_t){1.0, 1.0} : __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, voff + (unsigned)(k0 + 8 * d) * 8u, 0, 0));
        };
        auto stage = [&](const Chunk& c, int buf) {
#pragma unroll
            for (int u = 0; u < NLD; ++u) om_stage[buf][(lr + u * RPP) * PR + lj] = c.x[u];
        };
        double4_t acc[EG][2][2];
#pragma unroll
        for (int q = 0; q < EG; ++q) acc[q][0][0] = acc[q][0][1] = acc[q][1][0] = acc[q][1][1] = (double4_t){0.0, 0.0, 0.0, 0.0};
        auto products = [&](const double2_t (&a)[NST], int j, int buf) {
            const int steps = min(NST, (Kpad - j * kOmChunk) / 8);
#pragma unroll
            for (int d = 0; d < NST; ++d) {
                if (d < steps) {
                    const double2_t* xr = om_stage[buf] + (8 * d + 2 * fk) * PR + fr;
#pragma unroll
                    for (int q = 0; q < EG; ++q) {
                        const double2_t x0 = xr[16 * q], x1 = xr[PR + 16 * q];
                        acc[q][0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[d].x, x0.x, acc[q][0][0], 0, 0, 0);      // even epochs, k-step 0
                        acc[q][1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[d].x, x0.y, acc[q][1][0], 0, 0, 0);      // odd epochs
                        acc[q][0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[d].y, x1.x, acc[q][0][1], 0, 0, 0);      // k-step 1
                        acc[q][1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[d].y, x1.y, acc[q][1][1], 0, 0, 0);
                    }
                }
            }
        };
        // chunk j sits in stage j & 1, its block entries in `cur_a`; `next` holds chunk j + 1, `after` takes chunk j + 2
        double2_t cur_a[NST];
        auto step = [&](int j, Chunk& next, Chunk& after) {
            if (j + 2 < nchunk) fetch(after, j + 2);
            if (active && !(SHG_OM_X & 1)) products(cur_a, j, j & 1);
            if (j + 1 < nchunk) {
                stage(next, (j & 1) ^ 1);
#pragma unroll
                for (int d = 0; d < NST; ++d) cur_a[d] = next.a[d];
            }
            lds_barrier();
        };
        Chunk ca, cb;
        fetch(ca, 0);
        if (nchunk > 1) fetch(cb, 1);
        lds_barrier();                               // (a second pass: every wave is done with the stages of the first)
        stage(ca, 0);
#pragma unroll
        for (int d = 0; d < NST; ++d) cur_a[d] = ca.a[d];
        lds_barrier();
        for (int j = 0; j < nchunk; j += 2) {
            step(j, cb, ca);                         // chunk j + 1 is in cb, chunk j + 2 goes to ca
            if (j + 1 < nchunk) step(j + 1, ca, cb);
        }
        if (active && !(SHG_OM_X & 2)) {
#pragma unroll
            for (int q = 0; q < EG; ++q) {
                if (16 * q < npair) {
                    const double4_t even = acc[q][0][0] + acc[q][0][1], odd = acc[q][1][0] + acc[q][1][1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * tile + fk + 4 * r;                     // C/D layout: row = fk + 4 reg, column = fr
                        if (row < n) {
                            const size_t at = (size_t)row * ldx + 16 * q + fr;
                            Y[at] = m + row > 1 ? (double2_t){even[r], odd[r]} : X[at];
                        }
                    }
                }
            }
        }
    }
}

}  // namespace shg

namespace shg {
// degree-wise scaling of an order-major series: row (slot s, k) holds degree n = m + k of all epochs; rows below `nfirst` are copied
__global__ __launch_bounds__(256) void degree_scale_om_kernel(int N, int nfirst, int Bpad, const double* __restrict__ w, const double* __restrict__ in,
                                                              double* __restrict__ out) {
    const int s = blockIdx.x, m = (s + 1) >> 1, n = N + 1 - m;
    const size_t first = (size_t)om_row(N, s) * Bpad;
    const int pairs = Bpad / 2;
    const double2_t* x = reinterpret_cast<const double2_t*>(in + first);
    double2_t* y = reinterpret_cast<double2_t*>(out + first);
    for (int e = blockIdx.y * 256 + threadIdx.x; e < n * pairs; e += gridDim.y * 256) {
        const int degree = m + e / pairs;
        const double f = degree >= nfirst ? w[degree] : 1.0;
        const double2_t v = x[e];
        y[e] = degree >= nfirst ? (double2_t){v.x * f, v.y * f} : v;
    }
}
}  // namespace shg

extern "C" int shg_degree_scale_om(const double* w, int N, int nfirst, const double* om_in, int B, int Bpad, double* om_out, void* stream_) {
    SHG_REQUIRE(N >= 0 && B >= 0 && Bpad >= B && Bpad % 32 == 0, "shg_degree_scale_om: need N >= 0, 0 <= B <= Bpad, Bpad a multiple of 32");
    if (B == 0) return SHG_OK;
    SHG_REQUIRE(w && om_in && om_out, "shg_degree_scale_om: NULL pointer");
    hipLaunchKernelGGL(shg::degree_scale_om_kernel, dim3(2 * N + 1, 4), dim3(256), 0, (hipStream_t)stream_, N, nfirst, Bpad, w, om_in, om_out);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

extern "C" int shg_order_major_pack(const double* anm, int N, int B, double* om, int Bpad, void* stream_) {
    SHG_REQUIRE(N >= 0 && B >= 0 && Bpad >= B && Bpad % 32 == 0, "shg_order_major_pack: need N >= 0, 0 <= B <= Bpad, Bpad a multiple of 32");
    if (B == 0) return SHG_OK;
    SHG_REQUIRE(anm && om, "shg_order_major_pack: NULL pointer");
    const size_t lds = order_major_lds_bytes(N);
    SHG_REQUIRE(lds <= kOrderMajorMaxLds, "shg_order_major_pack: degree %d exceeds the LDS row stage of the layout kernel (max degree %d)", N, (int)(kOrderMajorMaxLds / 128) - 2);
    if (lds > 64 * 1024) SHG_SET_LDS_ONCE(shg::order_major_kernel<true>, kOrderMajorMaxLds);
    hipLaunchKernelGGL(shg::order_major_kernel<true>, dim3(N + 1, Bpad / 16), dim3(256), lds, (hipStream_t)stream_, N, B, Bpad, anm, om);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

extern "C" int shg_order_major_unpack(const double* om, int N, int B, int Bpad, double* anm, void* stream_) {
    SHG_REQUIRE(N >= 0 && B >= 0 && Bpad >= B && Bpad % 32 == 0, "shg_order_major_unpack: need N >= 0, 0 <= B <= Bpad, Bpad a multiple of 32");
    if (B == 0) return SHG_OK;
    SHG_REQUIRE(anm && om, "shg_order_major_unpack: NULL pointer");
    const size_t lds = order_major_lds_bytes(N);
    SHG_REQUIRE(lds <= kOrderMajorMaxLds, "shg_order_major_unpack: degree %d exceeds the LDS row stage of the layout kernel (max degree %d)", N, (int)(kOrderMajorMaxLds / 128) - 2);
    if (lds > 64 * 1024) SHG_SET_LDS_ONCE(shg::order_major_kernel<false>, kOrderMajorMaxLds);
    hipLaunchKernelGGL(shg::order_major_kernel<false>, dim3(N + 1, Bpad / 16), dim3(256), lds, (hipStream_t)stream_, N, B, Bpad, om, anm);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

extern "C" int shg_orderwise_filter_om(const double* blocks_packed, const int64_t* block_off, int Nb, int N, const double* om_in, int B, int Bpad,
                                       double* om_out, void* stream_) {
    SHG_REQUIRE(Nb >= 0 && N >= 0 && B >= 0 && Bpad >= B && Bpad % 32 == 0, "shg_orderwise_filter_om: negative size, or Bpad not a multiple of 32 >= B");
    SHG_REQUIRE(N <= Nb, "DDK filter only implemented for a maximum degree of %d (max_degree=%d supplied).", Nb, N);
    if (B == 0) return SHG_OK;
    SHG_REQUIRE(blocks_packed && block_off && om_in && om_out, "shg_orderwise_filter_om: NULL pointer");
    SHG_REQUIRE(om_in != om_out, "shg_orderwise_filter_om: in-place operation is not supported");
    hipStream_t stream = (hipStream_t)stream_;
    constexpr int EG = SHG_OM_EPOCH_GROUPS;            // slices of 32 EG epochs
    const int nslice = ceil_div(Bpad, 32 * EG);
    hipLaunchKernelGGL(shg::orderwise_filter_om_kernel<EG>, dim3((unsigned)(8 * ceil_div(2 * N + 1, 8) * nslice)), dim3(256), 0, stream, Nb, N, Bpad, nslice,
                       blocks_packed, (const long long*)block_off, om_in, om_out);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

// ------------------------------------------------------------------------------------------------
// DDK block construction:  W_k = (N_k + diag(w[m:]))^-1 N_k  for every order-wise normal block
// (replaces the 2 Nb + 1 dense solves of DDK.__init__ / DDKGeneric.__init__, grates/filter.py:252-255, 344-347).
// N_k + D is symmetric positive definite (normal matrix plus positive power-law weights): one workgroup per
// block does an in-place Cholesky factorisation and 2 triangular solves with all columns of N_k as right-hand sides.
// ------------------------------------------------------------------------------------------------
namespace shg {

// in-place right-looking Cholesky of the lower triangle of L [d][d] by one workgroup of 256 threads
__device__ void cholesky_inplace(double* L, int d, int tid) {
    for (int k = 0; k < d; ++k) {
        if (tid == 0) L[k * d + k] = sqrt(L[k * d + k]);
        __syncthreads();
        const double piv = L[k * d + k];
        for (int r = k + 1 + tid; r < d; r += 256) L[r * d + k] /= piv;
        __syncthreads();
        const int t = d - k - 1;
        for (int e = tid; e < t * t; e += 256) {
            const int r = k + 1 + e / t, c = k + 1 + e % t;
            if (c <= r) L[r * d + c] = fma(-L[r * d + k], L[c * d + k], L[r * d + c]);
        }
        __syncthreads();
    }
}

// X <- (L L^T)^-1 X for the ncol columns of X [d][ldx]: one thread per column
__device__ void cholesky_solve_columns(const double* L, double* X, int d, int ncol, int ldx, int tid) {
    for (int col = tid; col < ncol; col += 256) {
        for (int k = 0; k < d; ++k) {
            double acc = X[(size_t)k * ldx + col];
            for (int j = 0; j < k; ++j) acc = fma(-L[k * d + j], X[(size_t)j * ldx + col], acc);
            X[(size_t)k * ldx + col] = acc / L[k * d + k];
        }
        for (int k = d - 1; k >= 0; --k) {
            double acc = X[(size_t)k * ldx + col];
            for (int j = k + 1; j < d; ++j) acc = fma(-L[j * d + k], X[(size_t)j * ldx + col], acc);
            X[(size_t)k * ldx + col] = acc / L[k * d + k];
        }
    }
}

__global__ __launch_bounds__(256) void ddk_blocks_kernel(int Nb, const double* __restrict__ normals, const long long* __restrict__ off,
                                                         const double* __restrict__ weights, double* __restrict__ work,
                                                         double* __restrict__ out) {
    const int kb = blockIdx.x;
    const int m = (kb + 1) >> 1;
    const int d = Nb + 1 - m;
    const int tid = threadIdx.x;
    const double* Nk = normals + off[kb];
    double* L = work + off[kb];
    double* X = out + off[kb];
    for (int e = tid; e < d * d; e += 256) {
        const int r = e / d, c = e % d;
        L[e] = Nk[e] + (r == c ? weights[m + r] : 0.0);
        X[e] = Nk[e];
    }
    __syncthreads();
    cholesky_inplace(L, d, tid);
    cholesky_solve_columns(L, X, d, d, d, tid);
}

// X = A^-1 B for one symmetric positive definite A [n][n] and B [n][k]; column tiles of B go to different workgroups,
// each of which factors its own copy of A (small systems: least-squares normal equations of point lists).
__global__ __launch_bounds__(256) void spd_solve_kernel(int n, int k, const double* __restrict__ A, const double* __restrict__ Bm,
                                                        double* __restrict__ work, double* __restrict__ X) {
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * 256;
    const int nc = min(256, k - c0);
    double* L = work + (size_t)blockIdx.x * n * n;
    for (int e = tid; e < n * n; e += 256) L[e] = A[e];
    for (int e = tid; e < n * nc; e += 256) {
        const int r = e / nc, c = e % nc;
        X[(size_t)r * k + c0 + c] = Bm[(size_t)r * k + c0 + c];
    }
    __syncthreads();
    cholesky_inplace(L, n, tid);
    cholesky_solve_columns(L, X + c0, n, nc, k, tid);
}

}  // namespace shg

extern "C" int shg_ddk_blocks(const double* normals_packed, const int64_t* block_off, int Nb, const double* weights, double* work,
                              double* blocks_out, void* stream_) {
    SHG_REQUIRE(Nb >= 0, "shg_ddk_blocks: negative degree");
    SHG_REQUIRE(normals_packed && block_off && weights && work && blocks_out, "shg_ddk_blocks: NULL pointer");
    hipLaunchKernelGGL(shg::ddk_blocks_kernel, dim3(2 * Nb + 1), dim3(256), 0, (hipStream_t)stream_, Nb, normals_packed,
                       (const long long*)block_off, weights, work, blocks_out);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

extern "C" int shg_spd_solve(const double* A, int n, const double* Bm, int k, double* X, void* stream_) {
    SHG_REQUIRE(n >= 0 && k >= 0, "shg_spd_solve: negative size");
    if (n == 0 || k == 0) return SHG_OK;
    SHG_REQUIRE(A && Bm && X, "shg_spd_solve: NULL pointer");
    hipStream_t stream = (hipStream_t)stream_;
    const int nblocks = ceil_div(k, 256);
    double* work = nullptr;
    if (hipMallocAsync((void**)&work, (size_t)nblocks * n * n * sizeof(double), stream) != hipSuccess)
        return fail(SHG_ERR_NOMEM, "shg_spd_solve: workspace allocation failed");
    hipLaunchKernelGGL(shg::spd_solve_kernel, dim3(nblocks), dim3(256), 0, stream, n, k, A, Bm, work, X);
    (void)hipFreeAsync(work, stream);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}
