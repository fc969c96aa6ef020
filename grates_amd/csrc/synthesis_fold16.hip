// Fused batched synthesis for grids whose meridians have the full 16-fold symmetry of an equi-angular cell-centred
// grid with nlon % 16 == 0 (grates/grid.py:1146-1151: every GeographicGrid of 0.25 or 0.5 degree):
//     lon -> -lon   and   lon -> lon + k pi / 4,  k = 0 .. 7.
//
// Same structure as synthesis_fused.hip (one workgroup = 4 epochs x 16 parallels; Legendre stage on MFMA into an LDS panel,
// longitude stage on MFMA out of it) with a longitude stage that evaluates trigonometric sums only on the fundamental
// domain mu_c = (c + 1/2) dlon in (0, pi/8), c = 0 .. nlon/16 - 1, and forms the 16 images of every column in registers:
// four radix-2 steps of a decimation-in-frequency FFT done by the epilogue, the remaining DFT of nlon/16 points by MFMA.
//
// Orders m >= 1 fall into classes r = min(m mod 8, 8 - m mod 8) in {0, .., 4} with sign s_m = +1 (m mod 8 <= 4) or -1, because
//     cos(m k pi/4) = cos(r k pi/4),   sin(m k pi/4) = s_m sin(r k pi/4).
// With the panel holding A_m = sum_n C_nm PK_nm and B'_m = s_m sum_n S_nm PK_nm (the sign is folded into the coefficient
// repack) and the table T1 = cos(m mu), T2 = s_m sin(m mu), the sums per class
//     CA = A T1,  SA = A T2,  CB = B' T1,  SB = B' T2          (r = 0 and r = 4 need CA and SB only)
// give, for sign s = +-1 and k = 0 .. 7,
//     f(s mu + k pi/4) = sum_r cos(r k pi/4) (CA_r + s SB_r) + sin(r k pi/4) (CB_r - s SA_r).
// 16 accumulators per (16 rows x 16 columns): 84 MFMAs at d/o 96 where the 4-fold kernel issues 192 for the same outputs;
// the order-0 term is the start value of CA_0.
//
// Operands of the longitude stage: A fragments (A_m, B'_m) come from the panel with one ds_read_b128 per k-step, B fragments
// (T1, T2) are streamed from L2 by LDS-DMA (global_load_lds_dwordx4, 1 KB per wave-instruction) into a private ring of six
// 1 KB slots per wave, five pieces ahead of their use, and read back with one ds_read_b128: no registers are tied up by the
// prefetch and the loop needs no compile-time knowledge of the class lengths.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.h"

#ifndef SHG_STORE_AUX
#define SHG_STORE_AUX 2          // nt: the grids are streamed out and never re-read (see synthesis_fused.hip)
#endif

namespace shg {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));

constexpr int kRingDepth = 5;                       // trig pieces in flight per wave
constexpr int kRingSlots = 6;                       // ring slots (1 KB each) per wave
constexpr int kRingDoubles = 8 * kRingSlots * 128;  // the rings of the 8 waves sit at the start of the LDS (DMA offsets < 64 KB)
constexpr int kEpilogueStores = 32;                 // store instructions of one epilogue (pair stores): 16 images x 2

struct Fold16Params {
    int N, nlat, nlon, B, nit, nh;
    int n16;                  // columns of the fundamental domain = nlon / 16
    int nct;                  // column tiles of 16
    int npieces;              // trig pieces (k-steps) per column tile = sum of cls_nk
    int cls_nk[5];            // k-steps (4 orders each) of the classes, in the order r = 0, 4, 1, 2, 3
    int cls_slot[5];          // first panel slot of each class
    int cls_cnt[5];           // orders in each class (the slots up to 4 * cls_nk are zero padding)
    int nslot;                // panel slots of the orders >= 1; order 0 sits in slot nslot
    int dbg;                  // experiment switches (SHG_DEBUG): 1 no stores, 2 no Legendre stage, 4 no longitude stage
    int Qtot;
    const double* cpk4;       // repacked coefficients (see synthesis_fused.hip), S_nm negated for m mod 8 in {5, 6, 7}
    const double* pkf;
    const int4* itemtab;
    int nrec, ntrip;
#ifdef SHG_TIMELINE
    unsigned long long* tl;
#endif
    const int* blockmap;
    const int* badmap;
    const double* trig16;     // [nct][npieces][64 lanes][2]: (cos(m mu), s_m sin(m mu)) of order slot 4 ks + lane / 16, column 16 ct + lane % 16
    double* G;
};

#ifdef SHG_TIMELINE
#define F16_STAMP(ev)                                                                                         \
    do {                                                                                                      \
        if (P.tl && lane == 0) P.tl[((size_t)blockIdx.x * 8 + wave) * 16 + (ev)] = wall_clock64();            \
    } while (0)
#else
#define F16_STAMP(ev)
#endif

// LDS-DMA of one 1 KB piece: lane l copies 16 bytes from gbase + lane_off to LDS address lds_addr + 16 l.
// M0 is saved and restored (the compiler reserves it); s_nop 4 covers a scalar write of the base just before the statement.
__device__ __forceinline__ void glds16(const double* gbase, unsigned lane_off, unsigned lds_addr) {
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_off), "s"(gbase), "s"(lds_addr)
        : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <bool NS>
__global__ __launch_bounds__(512) void synthesis_fold16_kernel(Fold16Params P) {
    extern __shared__ __attribute__((aligned(16))) double As[];   // rings [8][kRingSlots][64][2], then panel [nslot + 1][64 rows][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbt = (P.B + 3) >> 2;
    const int bt = P.blockmap ? P.blockmap[2 * blockIdx.x] : (int)(blockIdx.x % nbt);
    const int it = P.blockmap ? P.blockmap[2 * blockIdx.x + 1] : (int)(blockIdx.x / nbt);
    const int i0 = it * 16;                             // plain layout: first parallel of the block
    const int i0n = it * 8;                             // NS layout: first northern parallel of the block
    const int fr = lane & 15, fk = lane >> 4;
    F16_STAMP(0);

    double2_t* const panel = reinterpret_cast<double2_t*>(As + kRingDoubles);      // [(slot * 64 + row)]

    // ---- zero the padding slots of the panel
    for (int c = 0; c < 5; ++c)
        for (int s = P.cls_slot[c] + P.cls_cnt[c]; s < P.cls_slot[c] + 4 * P.cls_nk[c]; ++s)
            if (tid < 64) panel[s * 64 + tid] = (double2_t){0.0, 0.0};

    // ---- trig stream of this wave: units u = wave + 8 q, (row tile, column tile) = (u & 3, u >> 2); the pieces of unit q
    //      are the npieces consecutive KB of column tile u >> 2.  The issue side runs kRingDepth pieces ahead of the consumer
    //      and keeps issuing (re-reading the last piece) when the stream is exhausted, so that the count of DMAs in flight
    //      is the same at every wait.
    const int nunits = 4 * P.nct;
    const int nq = wave < nunits ? (nunits - wave + 7) >> 3 : 0;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)As;
    const unsigned ring_lds = lds0 + (unsigned)wave * (kRingSlots * 1024);
    const unsigned lane_off = (unsigned)lane * 16u;
    const size_t ct_stride = (size_t)P.npieces * 128;
    const double* iptr = P.trig16 + (size_t)(nq > 0 ? wave >> 2 : 0) * ct_stride;
    int ileft = P.npieces, iq = 0, islot = 0;
    auto issue_piece = [&]() {
        glds16(iptr, lane_off, ring_lds + (unsigned)islot * 1024u);
        islot = islot + 1 == kRingSlots ? 0 : islot + 1;
        if (ileft > 1) {
            --ileft;
            iptr += 128;
        } else if (iq + 1 < nq) {
            ++iq;
            ileft = P.npieces;
            iptr = P.trig16 + (size_t)((wave + 8 * iq) >> 2) * ct_stride;
        }
    };
#pragma unroll
    for (int d = 0; d < kRingDepth; ++d) issue_piece();

    // ---- phase 1: Legendre stage (see synthesis_fused.hip).  Orders are distributed over the 8 waves; the result of order m
    //      is written as one 16-byte pair (A_m, B'_m) per panel row.
    if (!(P.dbg & 2)) {
        constexpr int ASTRIDE = NS ? 128 : 64;
        const int bad = NS ? P.badmap[it] : -1;
        const double* pkb = P.pkf + ((size_t)it * P.Qtot * 64 + lane) * 2;
        int mode = NS && bad >= 0 ? 1 : 0;
        int prow = lane;
        const double* cf = NS ? P.cpk4 + ((size_t)bt * P.Qtot * 64 + lane) * 2
                              : P.cpk4 + ((size_t)bt * P.Qtot * 32 + fk * 8 + (fr & 7)) * 2;
        const bool arow = NS || fr < 8;
        double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};

#define F16_P1_ISSUE(rec, ALO, AHI, BLO, BHI)                                                \
    do {                                                                                     \
        ALO = *reinterpret_cast<const double2*>(cf + (size_t)(rec).x * ASTRIDE);             \
        BLO = *reinterpret_cast<const double2*>(pkb + (size_t)(rec).x * 128);                \
        AHI = *reinterpret_cast<const double2*>(cf + (size_t)(rec).y * ASTRIDE);             \
        BHI = *reinterpret_cast<const double2*>(pkb + (size_t)(rec).y * 128);                \
    } while (0)

#define F16_P1_CONSUME(rec, ALO, AHI, BLO, BHI)                                                                     \
    do {                                                                                                            \
        const bool lo_ = arow && ((rec).w & 1);                                                                     \
        const bool hi_ = arow && ((rec).w & 2);                                                                     \
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lo_ ? ALO.x : 0.0, BLO.x, acc0, 0, 0, 0);                       \
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lo_ ? ALO.y : 0.0, BLO.y, acc1, 0, 0, 0);                       \
        if ((rec).w & 2) {                                                                                          \
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(hi_ ? AHI.x : 0.0, BHI.x, acc0, 0, 0, 0);                   \
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(hi_ ? AHI.y : 0.0, BHI.y, acc1, 0, 0, 0);                   \
        }                                                                                                           \
        if ((rec).w & 4) {                                  /* last item of an order: see synthesis_fused.hip */     \
            double vc_ = acc0[0] + acc1[0], vs_ = acc0[1] + acc1[1];                                                \
            if (NS) {                                                                                               \
                const double oc_ = acc0[2] + acc1[2], os_ = acc0[3] + acc1[3];                                      \
                const double rc_ = swap_half_row(fr < 8 ? vc_ : oc_), rs_ = swap_half_row(fr < 8 ? vs_ : os_);      \
                vc_ = (mode == 0 && fr >= 8) ? rc_ - oc_ : vc_ + rc_;                                               \
                vs_ = (mode == 0 && fr >= 8) ? rs_ - os_ : vs_ + rs_;                                               \
            }                                                                                                       \
            if (!NS || mode == 0 || fr < 8) panel[(rec).z * 64 + prow] = (double2_t){vc_, vs_};                     \
            acc0 = (double4_t){0.0, 0.0, 0.0, 0.0};                                                                 \
            acc1 = (double4_t){0.0, 0.0, 0.0, 0.0};                                                                 \
        }                                                                                                           \
    } while (0)

        double2 xal = {0, 0}, xah = {0, 0}, xbl = {0, 0}, xbh = {0, 0};
        double2 yal = {0, 0}, yah = {0, 0}, ybl = {0, 0}, ybh = {0, 0};
        double2 zal = {0, 0}, zah = {0, 0}, zbl = {0, 0}, zbh = {0, 0};
        double2 wal = {0, 0}, wah = {0, 0}, wbl = {0, 0}, wbh = {0, 0};
        const int4* recs = P.itemtab + (size_t)wave * P.nrec;
        for (int pass = 0; pass < (mode == 0 ? 1 : 2); ++pass) {
            if (pass == 1) {                                          // mirrored parallels of a polar block: their own table
                mode = 2;
                prow = lane + 8;
                pkb = P.pkf + ((size_t)(P.nit + bad) * P.Qtot * 64 + lane) * 2;
            }
            int4 c0 = recs[0], c1 = recs[1], c2 = recs[2];
            int4 n0 = recs[3], n1 = recs[4], n2 = recs[5], n3 = recs[6];
            F16_P1_ISSUE(c0, xal, xah, xbl, xbh);
            F16_P1_ISSUE(c1, yal, yah, ybl, ybh);
            F16_P1_ISSUE(c2, zal, zah, zbl, zbh);
            for (int trip = 0; trip < P.ntrip; ++trip) {
                const int4 a3 = n0, a4 = n1, a5 = n2, a6 = n3;
                const int4* nr = recs + 4 * trip + 7;
                n0 = nr[0];
                n1 = nr[1];
                n2 = nr[2];
                n3 = nr[3];
                F16_P1_ISSUE(a3, wal, wah, wbl, wbh);
                F16_P1_CONSUME(c0, xal, xah, xbl, xbh);
                F16_P1_ISSUE(a4, xal, xah, xbl, xbh);
                F16_P1_CONSUME(c1, yal, yah, ybl, ybh);
                F16_P1_ISSUE(a5, yal, yah, ybl, ybh);
                F16_P1_CONSUME(c2, zal, zah, zbl, zbh);
                F16_P1_ISSUE(a6, zal, zah, zbl, zbh);
                F16_P1_CONSUME(a3, wal, wah, wbl, wbh);
                c0 = a4;
                c1 = a5;
                c2 = a6;
            }
        }
#undef F16_P1_ISSUE
#undef F16_P1_CONSUME
    }
    F16_STAMP(1);
    __syncthreads();          // panel complete; from here on it is read-only and the waves run independently
    F16_STAMP(2);

    auto grid_row = [&](int s) { return NS ? (s < 8 ? i0n + s : P.nlat - 1 - (i0n + s - 8)) : i0 + s; };
    auto slot_valid = [&](int s) { return NS ? i0n + (s & 7) < P.nh : i0 + s < P.nlat; };

    // ---- phase 2: longitude stage
    const bool pair_stores = (P.n16 & 1) == 0;
    const int grid_bytes = P.nlat * P.nlon * 8;        // one epoch's grid; < 2^31 (checked on the host)
    const int n2 = P.nlon >> 1, n8 = P.nlon >> 3;
    const double2_t* const ringp = reinterpret_cast<const double2_t*>(As) + wave * (kRingSlots * 64) + lane;   // + slot * 64
    int cslot = 0;
    for (int q = 0; q < nq && !(P.dbg & 4); ++q) {
        const int u = wave + 8 * q, rt = u & 3, ct = u >> 2;
        double4_t acc[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) acc[a] = (double4_t){0.0, 0.0, 0.0, 0.0};
        {
            // order 0 does not depend on the longitude: start value of CA_0 (C/D layout: row = fk + 4 reg, all columns)
            const double2_t* z = panel + P.nslot * 64 + rt * 16 + fk;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[0][r] = z[4 * r].x;
        }
        const double2_t* prow = panel + rt * 16 + fr + fk * 64;          // + slot * 64
        int itu = 0;
#define F16_FETCH(SLOT)                                                                                   \
    issue_piece();                                                                                        \
    if ((itu < kRingDepth || (P.dbg & 64)) && pair_stores)                                                                  \
        wait_vmcnt<kRingDepth + kEpilogueStores>(); /* the stores of the previous epilogue are younger than these pieces */ \
    else                                                                                                  \
        wait_vmcnt<kRingDepth>();                                                                         \
    ++itu;                                                                                                \
    const double2_t t_ = ringp[cslot * 64];                                                               \
    cslot = cslot + 1 == kRingSlots ? 0 : cslot + 1;                                                      \
    const double2_t ab_ = prow[(SLOT) * 64]
#define F16_STEP2(C, A0)                                                                                  \
    for (int i = 0; i < P.cls_nk[C]; ++i) {                                                               \
        F16_FETCH(P.cls_slot[C] + 4 * i);                                                                 \
        acc[A0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ab_.x, t_.x, acc[A0], 0, 0, 0);                    \
        acc[A0 + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ab_.y, t_.y, acc[A0 + 1], 0, 0, 0);            \
    }
#define F16_STEP4(C, A0)                                                                                  \
    for (int i = 0; i < P.cls_nk[C]; ++i) {                                                               \
        F16_FETCH(P.cls_slot[C] + 4 * i);                                                                 \
        acc[A0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ab_.x, t_.x, acc[A0], 0, 0, 0);                    \
        acc[A0 + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ab_.x, t_.y, acc[A0 + 1], 0, 0, 0);            \
        acc[A0 + 2] = __builtin_amdgcn_mfma_f64_16x16x4f64(ab_.y, t_.x, acc[A0 + 2], 0, 0, 0);            \
        acc[A0 + 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(ab_.y, t_.y, acc[A0 + 3], 0, 0, 0);            \
    }
        // accumulators: 0 CA_0, 1 SB_0, 2 CA_4, 3 SB_4, then (CA, SA, CB, SB) of r = 1, 2, 3
        F16_STEP2(0, 0)
        F16_STEP2(1, 2)
        F16_STEP4(2, 4)
        F16_STEP4(3, 8)
        F16_STEP4(4, 12)
#undef F16_FETCH
#undef F16_STEP2
#undef F16_STEP4
        F16_STAMP(3 + 2 * min(q, 3));

        // ---- epilogue: 16 images per column.  X_r(s) = CA_r + s SB_r, Y_r(s) = CB_r - s SA_r;
        //      f_k = E_k + O_k, f_(k+4) = E_k - O_k with E_k = X_0 + (-1)^k X_4 + cos(k pi/2) X_2 + sin(k pi/2) Y_2 and
        //      O_0 = X_1 + X_3, O_1 = h ((X_1 - X_3) + (Y_1 + Y_3)), O_2 = Y_1 - Y_3, O_3 = h ((Y_1 + Y_3) - (X_1 - X_3)), h = sqrt(1/2).
        //      Image t = k (s = +1, ascending columns) or 8 + k (s = -1, descending columns) replaces accumulator t.
        constexpr double h = 0.70710678118654752440;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double ca0 = acc[0][r], sb0 = acc[1][r], ca4 = acc[2][r], sb4 = acc[3][r];
            const double ca1 = acc[4][r], sa1 = acc[5][r], cb1 = acc[6][r], sb1 = acc[7][r];
            const double ca2 = acc[8][r], sa2 = acc[9][r], cb2 = acc[10][r], sb2 = acc[11][r];
            const double ca3 = acc[12][r], sa3 = acc[13][r], cb3 = acc[14][r], sb3 = acc[15][r];
#pragma unroll
            for (int sgn = 0; sgn < 2; ++sgn) {
                const double x0 = sgn ? ca0 - sb0 : ca0 + sb0, x4 = sgn ? ca4 - sb4 : ca4 + sb4;
                const double x1 = sgn ? ca1 - sb1 : ca1 + sb1, y1 = sgn ? cb1 + sa1 : cb1 - sa1;
                const double x2 = sgn ? ca2 - sb2 : ca2 + sb2, y2 = sgn ? cb2 + sa2 : cb2 - sa2;
                const double x3 = sgn ? ca3 - sb3 : ca3 + sb3, y3 = sgn ? cb3 + sa3 : cb3 - sa3;
                const double uu = x0 + x4, vv = x0 - x4;
                const double e0 = uu + x2, e2 = uu - x2, e1 = vv + y2, e3 = vv - y2;
                const double a = x1 + x3, b = x1 - x3, c = y1 + y3, d = y1 - y3;
                const double bc = b + c, cb = c - b;
                acc[8 * sgn + 0][r] = e0 + a;
                acc[8 * sgn + 4][r] = e0 - a;
                acc[8 * sgn + 2][r] = e2 + d;
                acc[8 * sgn + 6][r] = e2 - d;
                acc[8 * sgn + 1][r] = fma(h, bc, e1);
                acc[8 * sgn + 5][r] = fma(-h, bc, e1);
                acc[8 * sgn + 3][r] = fma(h, cb, e3);
                acc[8 * sgn + 7][r] = fma(-h, cb, e3);
            }
        }
        const int b = bt * 4 + rt;
        const bool epoch_ok = b < P.B && !(P.dbg & 1);
        double* const Gb = P.G + (size_t)min(b, P.B - 1) * P.nlat * P.nlon;
        if (pair_stores) {
            // lanes (2 q, 2 q + 1) hold adjacent columns: after the exchange every lane owns two rows x two adjacent columns and
            // stores 16 bytes.  Byte offset = lane part (row, column inside the tile) + wave-uniform part (image, column tile);
            // lanes outside the grid carry an offset beyond the buffer and are dropped.
            // ALWAYS kEpilogueStores store instructions: the vmcnt bookkeeping of the trig stream counts them.
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(Gb, 0, grid_bytes, 0x00020000);
            const int par = fr & 1, ce = fr & ~1;
            const int sa = fk + (par ? 8 : 0), sb = sa + 4;
            const bool col_ok = epoch_ok && ct * 16 + ce < P.n16;
            const unsigned row_a = col_ok && slot_valid(sa) ? (unsigned)grid_row(sa) * (unsigned)P.nlon * 8u : 0x80000000u;
            const unsigned row_b = col_ok && slot_valid(sb) ? (unsigned)grid_row(sb) * (unsigned)P.nlon * 8u : 0x80000000u;
            const unsigned asc = (unsigned)ce * 8u, desc = (unsigned)(14 - ce) * 8u;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int k = t & 7;
                const bool ascending = t < 8;
                // first column of the tile's run: s = +1: (n2 + k n8) mod nlon + 16 ct;  s = -1: (n2 + k n8 - n16) mod nlon + n16 - 16 ct - 16
                int w = n2 + k * n8 - (ascending ? 0 : P.n16);
                w = w >= P.nlon ? w - P.nlon : w;
                const int soff = (ascending ? w + 16 * ct : w + P.n16 - 16 * ct - 16) * 8;
                const int soff_dbg = (P.dbg & 16) ? (soff & ~127) : (P.dbg & 32) ? (soff & ~63) : soff;      // timing experiments: aligned (wrong) columns
                double a_lo, a_hi, b_lo, b_hi;
                pair_exchange(acc[t][0], acc[t][2], 0xAAAAAAAAAAAAAAAAull, a_lo, a_hi);
                pair_exchange(acc[t][1], acc[t][3], 0xAAAAAAAAAAAAAAAAull, b_lo, b_hi);
                const double2_t va = ascending ? (double2_t){a_lo, a_hi} : (double2_t){a_hi, a_lo};
                const double2_t vb = ascending ? (double2_t){b_lo, b_hi} : (double2_t){b_hi, b_lo};
                // The wave-uniform part goes into the vector offset, not into the scalar offset operand of the store: with a
                // REGISTER soffset hipcc assumes that a 16-byte store's data registers may be overwritten by the very next VALU
                // instruction (the documented exemption of the gfx9 store-data hazard) and schedules one there; on gfx950 that
                // corrupted the low dword of the stored value in some lanes of some launches.
                const unsigned lane_col = (ascending ? asc : desc) + (unsigned)soff_dbg;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, va), rsrc, row_a + lane_col, 0, SHG_STORE_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, vb), rsrc, row_b + lane_col, 0, SHG_STORE_AUX);
            }
        } else {
            const int c = ct * 16 + fr;
            if (epoch_ok && c < P.n16) {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int k = t & 7;
                    int w = n2 + k * n8 - (t < 8 ? 0 : P.n16);
                    w = w >= P.nlon ? w - P.nlon : w;
                    const int j = t < 8 ? w + c : w + P.n16 - 1 - c;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (slot_valid(fk + 4 * r)) Gb[(size_t)grid_row(fk + 4 * r) * P.nlon + j] = acc[t][r];
                }
            }
        }
        F16_STAMP(4 + 2 * min(q, 3));
    }
    wait_vmcnt<0>();          // the prefetched pieces of the (padded) stream must have landed before the LDS is released
    F16_STAMP(12);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------

// class index (position in the order 0, 4, 1, 2, 3) and sign of order m >= 1
static inline void order_class(int m, int& cls, int& sign) {
    const int rho = m & 7;
    const int r = rho <= 4 ? rho : 8 - rho;
    sign = rho <= 4 ? 1 : -1;
    static const int pos[5] = {0, 2, 3, 4, 1};         // r -> position
    cls = pos[r];
}

// True when the meridians are lon_j = -pi + (j + 1/2) 2 pi / nlon to within a few ulp of pi: every one of the 16 images
// s mu_c + k pi/4 of the fundamental domain mu_c = lon[nlon/2 + c] is a grid column.
bool has_sixteenfold_symmetry(int nlon, const double* lon) {
    if (nlon < 256 || nlon % 16 != 0) return false;
    const double tol = 2e-15;
    const double pi = 3.14159265358979323846;
    const int n2 = nlon / 2, n8 = nlon / 8, n16 = nlon / 16;
    for (int c = 0; c < n16; ++c) {
        const double mu = lon[n2 + c];
        for (int k = 0; k < 8; ++k) {
            const int jp = (n2 + k * n8 + c) % nlon, jm = (n2 + k * n8 - 1 - c) % nlon;
            double dp = lon[jp] - (mu + k * (pi / 4)), dm = lon[jm] - (-mu + k * (pi / 4));
            dp -= 2 * pi * std::round(dp / (2 * pi));
            dm -= 2 * pi * std::round(dm / (2 * pi));
            if (std::fabs(dp) > tol || std::fabs(dm) > tol) return false;
        }
    }
    return true;
}

// class layout of the panel / trig stream for degree N; returns the panel slots (without the order-0 slot)
int fold16_layout(int N, int nk[5], int slot[5], int cnt[5], std::vector<int>* order_slot) {
    for (int c = 0; c < 5; ++c) cnt[c] = 0;
    for (int m = 1; m <= N; ++m) {
        int c, s;
        order_class(m, c, s);
        cnt[c]++;
    }
    int s = 0;
    for (int c = 0; c < 5; ++c) {
        nk[c] = (cnt[c] + 3) / 4;
        slot[c] = s;
        s += 4 * nk[c];
    }
    if (order_slot) {
        order_slot->assign(N + 1, 0);
        int next[5];
        for (int c = 0; c < 5; ++c) next[c] = slot[c];
        for (int m = 1; m <= N; ++m) {
            int c, sg;
            order_class(m, c, sg);
            (*order_slot)[m] = next[c]++;
        }
        (*order_slot)[0] = s;
    }
    return s;
}

static size_t fold16_lds_bytes(int nslot) { return (size_t)kRingDoubles * 8 + (size_t)(nslot + 1) * 1024; }

int fold16_applicable(const shg_plan* p) {
    if (!p->sym16) return 0;
    if ((long long)p->nlat * p->nlon * 8 >= (1LL << 31)) return 0;
    int nk[5], slot[5], cnt[5];
    const int nslot = fold16_layout(p->N, nk, slot, cnt, nullptr);
    return fold16_lds_bytes(nslot) <= 160 * 1024 ? 1 : 0;
}

// trig stream [nct][npieces][64][2] (+ one spare piece), built on the host like the other cos/sin tables (grates/utilities.py:272-273)
int build_trig16(shg_plan* p, const double* lon_h) {
    const int N = p->N, nlon = p->nlon, n16 = nlon / 16, nct = ceil_div(n16, 16);
    int nk[5], slot[5], cnt[5];
    std::vector<int> order_slot;
    const int nslot = fold16_layout(N, nk, slot, cnt, &order_slot);
    const int npieces = nk[0] + nk[1] + nk[2] + nk[3] + nk[4];
    std::vector<int> slot_order(nslot, -1);
    for (int m = 1; m <= N; ++m) slot_order[order_slot[m]] = m;
    std::vector<double> tab(((size_t)nct * npieces + 1) * 128, 0.0);
    for (int ct = 0; ct < nct; ++ct)
        for (int ks = 0; ks < npieces; ++ks)
            for (int l = 0; l < 64; ++l) {
                const int m = slot_order[4 * ks + (l >> 4)], c = 16 * ct + (l & 15);
                if (m < 0 || c >= n16) continue;
                int cls, sg;
                order_class(m, cls, sg);
                const double arg = (double)m * lon_h[nlon / 2 + c];
                double* dst = &tab[(((size_t)ct * npieces + ks) * 64 + l) * 2];
                dst[0] = std::cos(arg);
                dst[1] = sg * std::sin(arg);
            }
    SHG_HIP(hipMalloc((void**)&p->trig16, tab.size() * sizeof(double)));
    SHG_HIP(hipMemcpy(p->trig16, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    return SHG_OK;
}

int synthesis_fold16(shg_plan* p, const double* anm, int B, double* grid, hipStream_t stream) {
    if (!fold16_applicable(p)) return fail(SHG_ERR_UNSUPPORTED, "16-fold synthesis kernel not applicable to this plan");
    const bool ns = p->sym_ns && p->path != 7;
    int rc = build_pkf_table(p, ns, true, stream);
    if (rc) return rc;
    const int nbt = ceil_div(B, 4);
    const int nit = ns ? ceil_div(p->nlat / 2, 8) : ceil_div(p->nlat, 16);
    rc = pack_coefficients_fused(p, ns, true, anm, B, stream);
    if (rc) return rc;
    Fold16Params P;
    P.N = p->N;
    P.nlat = p->nlat;
    P.nlon = p->nlon;
    P.B = B;
    P.nit = nit;
    P.nh = p->nlat / 2;
    P.n16 = p->nlon / 16;
    P.nct = ceil_div(P.n16, 16);
    P.nslot = fold16_layout(p->N, P.cls_nk, P.cls_slot, P.cls_cnt, nullptr);
    P.npieces = P.cls_nk[0] + P.cls_nk[1] + P.cls_nk[2] + P.cls_nk[3] + P.cls_nk[4];
    const char* dbg_env = getenv("SHG_DEBUG");
    P.dbg = dbg_env ? atoi(dbg_env) : 0;
    P.Qtot = p->Qtot;
    P.cpk4 = p->cpk4;
    P.pkf = p->pkf;
    P.itemtab = reinterpret_cast<const int4*>(p->itemtab_d);
    P.nrec = p->itemtab_nrec;
    P.ntrip = p->itemtab_ntrip;
    P.badmap = p->badmap_d;
    P.blockmap = nullptr;
    if (!(P.dbg & 2048)) {
        rc = build_blockmap(p, nbt, nit, stream);
        if (rc) return rc;
        P.blockmap = p->blockmap_d;
    }
    P.trig16 = p->trig16;
    P.G = grid;
#ifdef SHG_TIMELINE
    P.tl = getenv("SHG_TIMELINE_PTR") ? (unsigned long long*)strtoull(getenv("SHG_TIMELINE_PTR"), nullptr, 0) : nullptr;
#endif
    const size_t lds = fold16_lds_bytes(P.nslot);
    const dim3 grid_dim((unsigned)(nbt * nit));
    ProfileScope ps(p, 2, stream);
    if (ns) {
        SHG_HIP(hipFuncSetAttribute((const void*)synthesis_fold16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((synthesis_fold16_kernel<true>), grid_dim, dim3(512), lds, stream, P);
    } else {
        SHG_HIP(hipFuncSetAttribute((const void*)synthesis_fold16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((synthesis_fold16_kernel<false>), grid_dim, dim3(512), lds, stream, P);
    }
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

}  // namespace shg
