// Fused batched synthesis for grids whose meridians are invariant under lon -> -lon and under R rotations lon -> lon + 2 pi k / R
// (an equi-angular cell-centred grid, grates/grid.py:1146-1151, with nlon a multiple of 2 R): every column of the fundamental
// domain mu_c = (c + 1/2) dlon, c = 0 .. nlon / (2 R) - 1, has 2 R images s mu_c + 2 pi k / R, s = +-1, k = 0 .. R - 1.
//
// Same structure as synthesis_fused.hip (one workgroup = 4 epochs x 16 parallels; Legendre stage on MFMA into an LDS panel,
// longitude stage on MFMA out of it), but the longitude stage evaluates its trigonometric sums on the fundamental domain only
// and forms the 2 R images of a column in registers: the epilogue does the radix-R step of a decimation-in-frequency FFT, the
// MFMAs the remaining DFT of nlon / (2 R) points.
//
// Orders m >= 1 fall into classes r = min(m mod R, R - m mod R) with sign s_m = +1 (m mod R <= R / 2) or -1, because
//     cos(2 pi m k / R) = cos(2 pi r k / R),   sin(2 pi m k / R) = s_m sin(2 pi r k / R).
// With the panel holding A_m = sum_n C_nm PK_nm and B'_m = s_m sum_n S_nm PK_nm (the sign is folded into the coefficient
// repack) and the table T1 = cos(m mu), T2 = s_m sin(m mu), the sums per class
//     CA = A T1,  SA = A T2,  CB = B' T1,  SB = B' T2          (r = 0 and r = R / 2 need CA and SB only)
// give
//     f(s mu + 2 pi k / R) = sum_r cos(2 pi r k / R) (CA_r + s SB_r) + sin(2 pi r k / R) (CB_r - s SA_r).
//
// R is chosen so that the images of a 16-column tile are whole 128-byte lines of the grid: nlon / R must be a multiple of 16.
//   R = 6 (12 images; nlon % 96 == 0: the 0.25 degree grid): classes 0, 3 (two sums) and 1, 2 (four sums) = 12 accumulators per
//          16 rows x 16 columns, 80 MFMAs at d/o 96 where the 4-fold kernel of synthesis_fused.hip issues 144 for the same outputs.
// Measured on the way (d/o 96 -> 0.25 degree): R = 8 (16 images, 84 MFMAs for 16 x 16 x 16 outputs) computes faster but its
// images start at multiples of nlon / 8 = 180 columns, i.e. at 32-byte instead of 128-byte boundaries, and the partial lines
// cost more HBM time than the MFMAs save (0.9 - 1.3 ms against 0.63 ms with the same stores forced onto line boundaries).
//
// Operands of the longitude stage: A fragments (A_m, B'_m) come from the panel with one ds_read_b128 per k-step, B fragments
// (T1, T2) are streamed from L2 by LDS-DMA (global_load_lds_dwordx4, 1 KB per wave-instruction) into a private ring of six
// 1 KB slots per wave, five pieces ahead of their use, and read back with one ds_read_b128: no registers are tied up by the
// prefetch and the loop needs no compile-time knowledge of the class lengths.  The fragments of k-step p + 1 are read while
// the MFMAs of k-step p run (two named register sets).
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

#ifndef SHG_ROT_PREFERENCE
#define SHG_ROT_PREFERENCE 10, 9, 6, 3
#endif
#ifndef SHG_ROT_WAVES
#define SHG_ROT_WAVES 8          // waves per workgroup.  12 (three per SIMD; the kernel needs 143 registers, no spill at 168; ring
#endif                           // depth 3 to fit the LDS) measured the same 0.525 ms: the fp64 issue pipe is the bound, not occupancy
#ifndef SHG_RING_DEPTH
#define SHG_RING_DEPTH (SHG_ROT_WAVES > 8 ? 3 : 5)
#endif
constexpr int kWaves = SHG_ROT_WAVES;
constexpr int kColStride = kWaves / 4;              // column tiles of a wave: wave >> 2, + kColStride, ...
constexpr int kRingDepth = SHG_RING_DEPTH;         // trig pieces in flight per wave
constexpr int kRingSlots = kRingDepth + 1;          // ring slots (1 KB each) per wave
constexpr int kRingDoubles = kWaves * kRingSlots * 128;  // the rings of the waves sit at the start of the LDS (DMA offsets < 64 KB)
constexpr int kMaxClasses = kRotMaxClasses;

// class layout for R rotations: classes r = 0 .. R / 2; in the K sequence the two-sum classes come first (r = 0, and r = R / 2 for
// even R), then r = 1, 2, ... with four sums each: 2 R accumulators per 16 rows x 16 columns, which the epilogue turns into the
// 2 R images in place
template <int R>
struct RotTraits {
    static constexpr int kClasses = R / 2 + 1, kTwo = R % 2 == 0 ? 2 : 1, kAcc = 2 * R;
    // accumulators of class r: (CA, SB) at a(r) for the two-sum classes, (CA, SA, CB, SB) at a(r) for the others
    static constexpr int acc_of(int r) { return r == 0 ? 0 : (R % 2 == 0 && 2 * r == R) ? 2 : 2 * kTwo + 4 * (r - 1); }
};

struct RotParams {
    int N, nlat, nlon, B, nit, nh;
    int nd;                   // columns of the fundamental domain = nlon / (2 R)
    int nct;                  // column tiles of 16
    int npieces;              // trig pieces (k-steps) per column tile = sum of cls_nk
    int cls_nk[kMaxClasses];  // k-steps (4 orders each) of the classes in K order; their panel slots follow each other: class c starts at 4 * (nk[0] + .. + nk[c-1])
    int cls_cnt[kMaxClasses]; // orders in each class (the slots up to 4 * cls_nk are zero padding)
    int nslot;                // panel slots of the orders >= 1 = 4 * npieces; order 0 sits in slot nslot
#ifdef SHG_EXPERIMENT
    int dbg;                  // experiment switches (SHG_DEBUG): 1 no stores, 2 no Legendre stage, 4 no longitude stage
    int stagger;              // SHG_STAGGER: the first 256 workgroups start up to this many 10 ns ticks late (spread in 32 steps)
    int stagger2;             // SHG_STAGGER2: waves 4 .. 7 enter the longitude stage this many 10 ns ticks behind waves 0 .. 3
#endif
    int Qtot;
    const double* cpk4;       // repacked coefficients (see synthesis_fused.hip), S_nm negated where s_m = -1
    const double* pkf;
    const int4* itemtab;
    int nrec, ntrip;
    int* sem;                 // tokens of the Legendre stage in use (device-wide counter, 0 between launches)
    int sem_limit;            // at most this many workgroups run their Legendre stage at the same time (0 = no limit)
#ifdef SHG_TIMELINE
    unsigned long long* tl;
#endif
    const int* blockmap;
    const int* badmap;
    const double* trig;       // [column tile][npieces][64 lanes][2]: (cos(m mu), s_m sin(m mu)) of order slot 4 ks + lane / 16, column 16 ct + lane % 16
    double* G;
};

#ifdef SHG_TIMELINE
#define ROT_STAMP(ev)                                                                                         \
    do {                                                                                                      \
        if (P.tl && lane == 0) P.tl[((size_t)blockIdx.x * kWaves + wave) * 16 + (ev)] = wall_clock64();            \
        if (P.tl && lane == 0 && ((ev) == 0 || (ev) == 12))                                                   \
            P.tl[((size_t)blockIdx.x * kWaves + wave) * 16 + ((ev) == 0 ? 13 : 14)] = __builtin_amdgcn_s_memtime();  \
    } while (0)
#else
#define ROT_STAMP(ev)
#endif

// LDS-DMA of one 1 KB piece: lane l copies 16 bytes from gbase + lane_off to LDS address lds_addr + 16 l.
// M0 is overwritten and not restored: hipcc treats it as a reserved register that it loads right in front of every use of
// its own (it refuses it in a clobber list for that reason), and gfx9 LDS instructions do not read it -- saving and restoring it
// around every piece cost three scalar instructions and five idle cycles per k-step.  The s_nop covers the scalar write of M0
// in front of its use by the vector memory instruction.
__device__ __forceinline__ void glds16(const double* gbase, unsigned lane_off, unsigned lds_addr) {
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(lane_off), "s"(gbase), "s"(lds_addr)
        : "memory");
}

#ifndef SHG_ROT_X
#define SHG_ROT_X 0          // experiment switches (timing only): 1 no issue-side stream bookkeeping, 2 no consumer-side bookkeeping
#endif

typedef int int4_s __attribute__((ext_vector_type(4)));

// 8-byte non-temporal buffer store with the wave-uniform part of the address in the SCALAR offset (no vector add per store), as
// inline asm.  The scalar offset travels through M0, written by an SALU instruction of the statement itself: hipcc reloads
// spilled scalars with v_readlane right in front of an asm statement, and an SGPR written by the VALU needs five wait states
// before a vector memory instruction may read it -- an SALU read does not.  (M0 is free: gfx9 LDS instructions do not read it,
// and every LDS-DMA of the trig stream sets it in its own statement.)  Stores of at most 8 bytes have no store-data hazard.
__device__ __forceinline__ void store_b64_soff(double v, int4_s rsrc, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_store_dwordx2 %0, %1, %2, m0 offen nt"
                 :
                 : "v"(v), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// The 2 R images of one (row, column) from its sums, in place: acc[t] <- image t, t = k (s = +1, ascending columns) or R + k
// (s = -1, descending columns).  X_r(s) = CA_r + s SB_r, Y_r(s) = CB_r - s SA_r.
template <int R>
__device__ __forceinline__ void rot_images(double4_t* acc, int r);

// R = 6, accumulators 0 CA_0, 1 SB_0, 2 CA_3, 3 SB_3, 4-7 (CA, SA, CB, SB)_1, 8-11 (CA, SA, CB, SB)_2.  With g = sqrt(3) / 2:
//   E_0 = X_0 + X_2,  E_1 = X_0 - X_2 / 2 + g Y_2,  E_2 = X_0 - X_2 / 2 - g Y_2        (classes 0, 2: period 3 in k)
//   O_0 = X_1 + X_3,  O_1 = X_1 / 2 + g Y_1 - X_3,  O_2 = -X_1 / 2 + g Y_1 + X_3       (classes 1, 3: O_(k+3) = -O_k)
//   f_k = E_k + O_k,  f_(k+3) = E_k - O_k,  k = 0, 1, 2.
template <>
__device__ __forceinline__ void rot_images<6>(double4_t* acc, int r) {
    constexpr double g = 0.86602540378443864676;
    const double ca0 = acc[0][r], sb0 = acc[1][r], ca3 = acc[2][r], sb3 = acc[3][r];
    const double ca1 = acc[4][r], sa1 = acc[5][r], cb1 = acc[6][r], sb1 = acc[7][r];
    const double ca2 = acc[8][r], sa2 = acc[9][r], cb2 = acc[10][r], sb2 = acc[11][r];
#pragma unroll
    for (int sgn = 0; sgn < 2; ++sgn) {
        const double x0 = sgn ? ca0 - sb0 : ca0 + sb0, x3 = sgn ? ca3 - sb3 : ca3 + sb3;
        const double x1 = sgn ? ca1 - sb1 : ca1 + sb1, y1 = sgn ? cb1 + sa1 : cb1 - sa1;
        const double x2 = sgn ? ca2 - sb2 : ca2 + sb2, y2 = sgn ? cb2 + sa2 : cb2 - sa2;
        const double t = fma(-0.5, x2, x0);
        const double e0 = x0 + x2, e1 = fma(g, y2, t), e2 = fma(-g, y2, t);
        const double u = fma(0.5, x1, -x3);
        const double o0 = x1 + x3, o1 = fma(g, y1, u), o2 = fma(g, y1, -u);
        acc[6 * sgn + 0][r] = e0 + o0;
        acc[6 * sgn + 3][r] = e0 - o0;
        acc[6 * sgn + 1][r] = e1 + o1;
        acc[6 * sgn + 4][r] = e1 - o1;
        acc[6 * sgn + 2][r] = e2 + o2;
        acc[6 * sgn + 5][r] = e2 - o2;
    }
}

// R = 3, accumulators 0 CA_0, 1 SB_0, 2-5 (CA, SA, CB, SB)_1:  f_0 = X_0 + X_1,  f_1,2 = X_0 - X_1 / 2 +- g Y_1.
template <>
__device__ __forceinline__ void rot_images<3>(double4_t* acc, int r) {
    constexpr double g = 0.86602540378443864676;
    const double ca0 = acc[0][r], sb0 = acc[1][r];
    const double ca1 = acc[2][r], sa1 = acc[3][r], cb1 = acc[4][r], sb1 = acc[5][r];
#pragma unroll
    for (int sgn = 0; sgn < 2; ++sgn) {
        const double x0 = sgn ? ca0 - sb0 : ca0 + sb0;
        const double x1 = sgn ? ca1 - sb1 : ca1 + sb1, y1 = sgn ? cb1 + sa1 : cb1 - sa1;
        const double t = fma(-0.5, x1, x0);
        acc[3 * sgn + 0][r] = x0 + x1;
        acc[3 * sgn + 1][r] = fma(g, y1, t);
        acc[3 * sgn + 2][r] = fma(-g, y1, t);
    }
}

// R = 9, accumulators 0 CA_0, 1 SB_0, 2 + 4 (r - 1) .. (CA, SA, CB, SB)_r, r = 1 .. 4.  With c_j = cos(40 j deg), s_j = sin(40 j deg)
// (c_3 = -1/2, s_3 = sqrt(3) / 2) and C_k = X_0 + sum_r c_(rk mod 9) X_r, S_k = sum_r s_(rk mod 9) Y_r:
//   f_0 = X_0 + X_1 + X_2 + X_3 + X_4,   f_k = C_k + S_k,   f_(9-k) = C_k - S_k,   k = 1 .. 4.
template <>
__device__ __forceinline__ void rot_images<9>(double4_t* acc, int r) {
    constexpr double c1 = 0.76604444311897803520, c2 = 0.17364817766693034885, c4 = -0.93969262078590838405;
    constexpr double s1 = 0.64278760968653932632, s2 = 0.98480775301220805937, s3 = 0.86602540378443864676, s4 = 0.34202014332566873304;
    const double ca0 = acc[0][r], sb0 = acc[1][r];
    double ca[4], sa[4], cb[4], sb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ca[j] = acc[2 + 4 * j][r];
        sa[j] = acc[3 + 4 * j][r];
        cb[j] = acc[4 + 4 * j][r];
        sb[j] = acc[5 + 4 * j][r];
    }
#pragma unroll
    for (int sgn = 0; sgn < 2; ++sgn) {
        const double x0 = sgn ? ca0 - sb0 : ca0 + sb0;
        double x[4], y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x[j] = sgn ? ca[j] - sb[j] : ca[j] + sb[j];
            y[j] = sgn ? cb[j] + sa[j] : cb[j] - sa[j];
        }
        const double x1 = x[0], x2 = x[1], x3 = x[2], x4 = x[3], y1 = y[0], y2 = y[1], y3 = y[2], y4 = y[3];
        const double t = (x1 + x2) + x4, x03 = x0 + x3;
        const double h = fma(-0.5, x3, x0), u = s3 * y3;
        const double C1 = fma(c1, x1, fma(c2, x2, fma(c4, x4, h)));
        const double C2 = fma(c2, x1, fma(c4, x2, fma(c1, x4, h)));
        const double C3 = fma(-0.5, t, x03);
        const double C4 = fma(c4, x1, fma(c1, x2, fma(c2, x4, h)));
        const double S1 = fma(s1, y1, fma(s2, y2, fma(s4, y4, u)));
        const double S2 = fma(s2, y1, fma(s4, y2, fma(-s1, y4, -u)));
        const double S3 = s3 * ((y1 - y2) + y4);
        const double S4 = fma(s4, y1, fma(-s1, y2, fma(-s2, y4, u)));
        acc[9 * sgn + 0][r] = x03 + t;
        acc[9 * sgn + 1][r] = C1 + S1;
        acc[9 * sgn + 8][r] = C1 - S1;
        acc[9 * sgn + 2][r] = C2 + S2;
        acc[9 * sgn + 7][r] = C2 - S2;
        acc[9 * sgn + 3][r] = C3 + S3;
        acc[9 * sgn + 6][r] = C3 - S3;
        acc[9 * sgn + 4][r] = C4 + S4;
        acc[9 * sgn + 5][r] = C4 - S4;
    }
}

// R = 10, accumulators 0 CA_0, 1 SB_0, 2 CA_5, 3 SB_5, 4 + 4 (r - 1) .. (CA, SA, CB, SB)_r, r = 1 .. 4.  With c1 = cos 36, c2 = cos 72,
// s1 = sin 36, s2 = sin 72 (degrees): the even classes have period 5 in k, the odd ones change sign after 5 steps,
//   E_0 = X_0 + X_2 + X_4,  E_1,4 = (X_0 + c2 X_2 - c1 X_4) +- (s2 Y_2 + s1 Y_4),  E_2,3 = (X_0 - c1 X_2 + c2 X_4) +- (s1 Y_2 - s2 Y_4)
//   O_0 = X_1 + X_3 + X_5,  O_1,4 = (s1 Y_1 + s2 Y_3) +- (c1 X_1 - c2 X_3 - X_5),  O_2,3 = (s2 Y_1 - s1 Y_3) +- (c2 X_1 - c1 X_3 + X_5)
//   f_k = E_k + O_k,  f_(k+5) = E_k - O_k,  k = 0 .. 4.
template <>
__device__ __forceinline__ void rot_images<10>(double4_t* acc, int r) {
    constexpr double c1 = 0.80901699437494742410, c2 = 0.30901699437494742410;
    constexpr double s1 = 0.58778525229247312917, s2 = 0.95105651629515357212;
    const double ca0 = acc[0][r], sb0 = acc[1][r], ca5 = acc[2][r], sb5 = acc[3][r];
    double ca[4], sa[4], cb[4], sb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ca[j] = acc[4 + 4 * j][r];
        sa[j] = acc[5 + 4 * j][r];
        cb[j] = acc[6 + 4 * j][r];
        sb[j] = acc[7 + 4 * j][r];
    }
#pragma unroll
    for (int sgn = 0; sgn < 2; ++sgn) {
        const double x0 = sgn ? ca0 - sb0 : ca0 + sb0, x5 = sgn ? ca5 - sb5 : ca5 + sb5;
        double x[4], y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x[j] = sgn ? ca[j] - sb[j] : ca[j] + sb[j];
            y[j] = sgn ? cb[j] + sa[j] : cb[j] - sa[j];
        }
        const double x1 = x[0], x2 = x[1], x3 = x[2], x4 = x[3], y1 = y[0], y2 = y[1], y3 = y[2], y4 = y[3];
        const double e0 = (x0 + x2) + x4;
        const double a1 = fma(c2, x2, fma(-c1, x4, x0)), b1 = fma(s2, y2, s1 * y4);
        const double a2 = fma(-c1, x2, fma(c2, x4, x0)), b2 = fma(s1, y2, -(s2 * y4));
        const double e1 = a1 + b1, e4 = a1 - b1, e2 = a2 + b2, e3 = a2 - b2;
        const double o0 = (x1 + x3) + x5;
        const double p = fma(c1, x1, fma(-c2, x3, -x5)), q = fma(s1, y1, s2 * y3);
        const double u = fma(c2, x1, fma(-c1, x3, x5)), v = fma(s2, y1, -(s1 * y3));
        const double o1 = q + p, o4 = q - p, o2 = v + u, o3 = v - u;
        acc[10 * sgn + 0][r] = e0 + o0;
        acc[10 * sgn + 5][r] = e0 - o0;
        acc[10 * sgn + 1][r] = e1 + o1;
        acc[10 * sgn + 6][r] = e1 - o1;
        acc[10 * sgn + 2][r] = e2 + o2;
        acc[10 * sgn + 7][r] = e2 - o2;
        acc[10 * sgn + 3][r] = e3 + o3;
        acc[10 * sgn + 8][r] = e3 - o3;
        acc[10 * sgn + 4][r] = e4 + o4;
        acc[10 * sgn + 9][r] = e4 - o4;
    }
}

// Trig stream and fragment registers of one consumer wave; in the persistent kernel they live across tiles.
struct RotStream {
    const double* iptr;       // next piece to issue
    const double* ibase;      // first piece of the column tile being issued
    const double* ifirst;     // first piece of the wave's first column tile (wave >> 2), ilast: of its last one
    const double* ilast;
    int irem;                 // pieces of the current column tile left to issue
    unsigned im0, ring_lds;   // LDS address of the next ring slot to fill / of the wave's first slot
    int cslot, pf;            // ring slot / k-step (inside its unit) of the next fetch
    double2_t tx, abx;        // fragments of the next k-step to compute
};

// The issue side walks the column tiles of its wave on its own: (wave >> 2), + 2, ..., then again from the start for the next tile.
// EP = epochs (= row tiles) of the workgroup: eight waves, one workgroup per CU
constexpr int EP = 4;
__device__ __forceinline__ void rot_stream_init(RotStream& S, const RotParams& P, const double* As, int wave) {
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)As;
    S.ring_lds = lds0 + (unsigned)wave * (kRingSlots * 1024);
    S.im0 = S.ring_lds;
    const int ct0 = min(wave / EP, P.nct - 1);
    const int nq = (P.nct - ct0 + kColStride - 1) / kColStride;     // column tiles of this wave
    S.ifirst = P.trig + (size_t)ct0 * P.npieces * 128;
    S.ilast = S.ifirst + (size_t)(nq - 1) * kColStride * P.npieces * 128;
    S.ibase = S.ifirst;
    S.iptr = S.ifirst;
    S.irem = P.npieces;
    S.cslot = 0;
    S.pf = 0;
    S.tx = (double2_t){0.0, 0.0};
    S.abx = (double2_t){0.0, 0.0};
}

__device__ __forceinline__ void rot_issue_piece(RotStream& S, const RotParams& P, unsigned lane_off) {
    glds16(S.iptr, lane_off, S.im0);
#if defined(SHG_ROT_X) && (SHG_ROT_X & 1)
    return;                                        // experiment: no stream bookkeeping (timing only, wrong results)
#endif
    S.im0 = S.im0 + 1024 == S.ring_lds + kRingSlots * 1024 ? S.ring_lds : S.im0 + 1024;
    const bool more = S.irem > 1;
    const double* nbase = S.ibase == S.ilast ? S.ifirst : S.ibase + kColStride * P.npieces * 128;
    S.ibase = more ? S.ibase : nbase;
    S.iptr = more ? S.iptr + 128 : nbase;
    S.irem = more ? S.irem - 1 : P.npieces;
}

// Longitude stage + epilogue of one wave for the tile (bt, it).  A unit = (row tile wave & 3, column tile ct), ct = wave >> 2,
// (wave >> 2) + kColStride, ...  (Taking the column tiles of a row tile from a common counter, so that the older wave of a SIMD, which
// wins every MFMA issue slot and runs ~1.3 times faster, takes more of them, measured 16 % slower: the four waves that work
// on the same column tile then drift apart and no longer share the trig pieces in the L1.)
// On entry (S.tx, S.abx) hold the fragments of the first k-step; on exit those of the first k-step of column tile wave >> 2
// again (trig part; the panel part is re-read by the caller once the next panel is there).
template <bool NS, int R>
__device__ __forceinline__ void rot_phase2(const RotParams& P, const double* As, const double2_t* panel, RotStream& S, int wave, int lane,
                                           int b0, int it) {
    constexpr int PR = 16 * EP;                        // panel rows
    using T = RotTraits<R>;
    constexpr int kImages = 2 * R;
    const int fr = lane & 15, fk = lane >> 4;
    const int i0 = it * 16, i0n = it * 8;
    const unsigned lane_off = (unsigned)lane * 16u;
    auto grid_row = [&](int s) { return NS ? (s < 8 ? i0n + s : P.nlat - 1 - (i0n + s - 8)) : i0 + s; };
    auto slot_valid = [&](int s) { return NS ? i0n + (s & 7) < P.nh : i0 + s < P.nlat; };

    const int grid_bytes = P.nlat * P.nlon * 8;        // one epoch's grid; < 2^31 (checked on the host)
    const int n2 = P.nlon >> 1, nR = P.nlon / R;
    const double2_t* const ringp = reinterpret_cast<const double2_t*>(As) + wave * (kRingSlots * 64) + lane;   // + slot * 64
    // all units of a wave lie in the same row tile
    const int rt = wave % EP;
    const double2_t* const prow = panel + rt * 16 + fr + fk * PR;        // + 4 PR p: k-step p of the flat class sequence
    // Fragments of the next k-step of the flat (unit, k-step) sequence -> (T_, AB_): one more trig piece issued, the piece of
    // this k-step waited for, ring slot and panel rows read.  Branch-free and unconditional (after the last k-step of the last
    // unit it re-reads valid memory), so that hipcc keeps exact lgkmcnt counts across the loops: the MFMAs of k-step p then
    // wait for their own fragments only (lgkmcnt(2)), not for the reads of k-step p + 1 issued just before them.
    // The wait is always the strict one (all but the kRingDepth youngest operations done): stores of the previous epilogue
    // that are still in flight are waited for too, which measured no different from counting them out.
#define ROT_FETCH(T_, AB_)                                                                                \
    do {                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        if (!(SHG_ROT_X & 8)) rot_issue_piece(S, P, lane_off);                                            \
        if (!(SHG_ROT_X & 8)) wait_vmcnt<kRingDepth>();                                                   \
        if (!(SHG_ROT_X & 4)) T_ = ringp[S.cslot * 64];                                                   \
        if (!(SHG_ROT_X & 2)) S.cslot = S.cslot + 1 == kRingSlots ? 0 : S.cslot + 1;                      \
        AB_ = prow[S.pf * (4 * PR)];                                                                           \
        if (!(SHG_ROT_X & 2)) S.pf = S.pf + 1 == P.npieces ? 0 : S.pf + 1;                                \
        __builtin_amdgcn_sched_barrier(0);                                                                \
    } while (0)
#ifdef SHG_ROT_SETPRIO
#define ROT_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define ROT_PRIO(x)
#endif
#define ROT_MFMA2(A0, T_, AB_)                                                                            \
    ROT_PRIO(1);                                                                                          \
    acc[A0] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.x, T_.x, acc[A0], 0, 0, 0);                        \
    ROT_PRIO(2);                                                                                          \
    acc[A0 + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.y, T_.y, acc[A0 + 1], 0, 0, 0);                \
    ROT_PRIO(0)
#define ROT_MFMA4(A0, T_, AB_)                                                                            \
    ROT_PRIO(1);                                                                                          \
    acc[A0] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.x, T_.x, acc[A0], 0, 0, 0);                        \
    ROT_PRIO(2);                                                                                          \
    acc[A0 + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.x, T_.y, acc[A0 + 1], 0, 0, 0);                \
    acc[A0 + 2] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.y, T_.x, acc[A0 + 2], 0, 0, 0);                \
    acc[A0 + 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.y, T_.y, acc[A0 + 3], 0, 0, 0);                \
    ROT_PRIO(0)
    // first k-step of a class: the accumulators start from the constant 0 operand of the MFMA instead of being cleared with
    // eight VALU moves each (96 per unit; a VALU instruction costs the fp64 MFMA stream ~10 cycles).  Accumulator 0 (CA of
    // class 0) starts from the order-0 values, which are loaded into it beforehand.
#define ROT_MFMA2_FIRST(A0, T_, AB_)                                                                      \
    acc[A0] = (A0) == 0 ? __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.x, T_.x, acc[A0], 0, 0, 0)              \
                        : __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.x, T_.x, kZero4, 0, 0, 0);             \
    acc[A0 + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.y, T_.y, kZero4, 0, 0, 0)
#define ROT_MFMA4_FIRST(A0, T_, AB_)                                                                      \
    acc[A0] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.x, T_.x, kZero4, 0, 0, 0);                         \
    acc[A0 + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.x, T_.y, kZero4, 0, 0, 0);                     \
    acc[A0 + 2] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.y, T_.x, kZero4, 0, 0, 0);                     \
    acc[A0 + 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.y, T_.y, kZero4, 0, 0, 0)
    // one class: k-steps in pairs on the register sets (tx, abx) / (ty, aby); the current fragments are in (tx, abx) on entry
    // and on exit, the fragments of the next k-step are fetched before the MFMAs of the current one are issued
#define ROT_CLASS(C, MF, A0, NACC)                                                                        \
    {                                                                                                     \
        const int nk_ = P.cls_nk[C];                                                                      \
        int i_ = 0;                                                                                       \
        if (nk_ >= 2) {                                                                                   \
            ROT_FETCH(ty, aby);                                                                           \
            MF##_FIRST(A0, tx, abx);                                                                      \
            ROT_FETCH(tx, abx);                                                                           \
            MF(A0, ty, aby);                                                                              \
            i_ = 2;                                                                                       \
        } else if (nk_ == 1) {                                                                            \
            ROT_FETCH(ty, aby);                                                                           \
            MF##_FIRST(A0, tx, abx);                                                                      \
            tx = ty;                                                                                      \
            abx = aby;                                                                                    \
            i_ = 1;                                                                                       \
        } else {                                                                                          \
            _Pragma("unroll") for (int z_ = ((A0) == 0 ? 1 : 0); z_ < (NACC); ++z_) acc[(A0) + z_] = kZero4; \
        }                                                                                                 \
        for (; i_ + 2 <= nk_; i_ += 2) {                                                                  \
            ROT_FETCH(ty, aby);                                                                           \
            MF(A0, tx, abx);                                                                              \
            ROT_FETCH(tx, abx);                                                                           \
            MF(A0, ty, aby);                                                                              \
        }                                                                                                 \
        if (i_ < nk_) {                                                                                   \
            ROT_FETCH(ty, aby);                                                                           \
            MF(A0, tx, abx);                                                                              \
            tx = ty;                                                                                      \
            abx = aby;                                                                                    \
        }                                                                                                 \
    }
    const double4_t kZero4 = {0.0, 0.0, 0.0, 0.0};
    double2_t tx = S.tx, abx = S.abx, ty = {0.0, 0.0}, aby = {0.0, 0.0};
    const int ct0 = wave / EP;
    for (int ct = ct0, q = 0; ct < P.nct; ct += kColStride, ++q) {
        (void)q;
        double4_t acc[T::kAcc];
        {
            // order 0 does not depend on the longitude: start value of CA_0 (C/D layout: row = fk + 4 reg, all columns)
            const double2_t* z = panel + P.nslot * PR + rt * 16 + fk;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[0][r] = z[4 * r].x;
        }
#pragma unroll
        for (int c = 0; c < T::kClasses; ++c) {
            if (c < T::kTwo) {
                ROT_CLASS(c, ROT_MFMA2, 2 * c, 2)
            } else {
                ROT_CLASS(c, ROT_MFMA4, 2 * T::kTwo + 4 * (c - T::kTwo), 4)
            }
        }
        ROT_STAMP(3 + 2 * min(q, 3));

        // ---- epilogue: the 2 R images of every column, in place
#pragma unroll
        for (int r = 0; r < 4; ++r) rot_images<R>(acc, r);      // (unconditional: a run-time switch around an in-place update of the
                                                                //  accumulator vectors makes hipcc copy every vector, ~60 moves per row)
        const int b = b0 + rt;
        const bool epoch_ok = b < P.B && !SHG_DBG(P, 1);
        double* const Gb = P.G + (size_t)min(b, P.B - 1) * P.nlat * P.nlon;
        {
            // lanes (2 q, 2 q + 1) hold adjacent columns: after the exchange every lane owns two rows x two adjacent columns and
            // stores 16 bytes.  Byte offset = lane part (row, column inside the tile) + wave-uniform part (image, column tile);
            // lanes outside the grid carry an offset beyond the buffer and are dropped.
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(Gb, 0, grid_bytes, 0x00020000);
            const int par = fr & 1, ce = fr & ~1;
            const int sa = fk + (par ? 8 : 0), sb = sa + 4;
            const bool col_ok = epoch_ok && ct * 16 + ce < P.nd;
            const unsigned row_a = col_ok && slot_valid(sa) ? (unsigned)grid_row(sa) * (unsigned)P.nlon * 8u : 0x80000000u;
            const unsigned row_b = col_ok && slot_valid(sb) ? (unsigned)grid_row(sb) * (unsigned)P.nlon * 8u : 0x80000000u;
            const unsigned asc = (unsigned)ce * 8u, desc = (unsigned)(14 - ce) * 8u;
#pragma unroll
            for (int t = 0; t < kImages; ++t) {
                const int k = t < R ? t : t - R;
                const bool ascending = t < R;
                // first column of the tile's run: s = +1: (n2 + k nR) mod nlon + 16 ct;  s = -1: (n2 + k nR - nd) mod nlon + nd - 16 ct - 16
                int w = n2 + k * nR - (ascending ? 0 : P.nd);
                w = w >= P.nlon ? w - P.nlon : w;
                const int soff = (ascending ? w + 16 * ct : w + P.nd - 16 * ct - 16) * 8;
                double a_lo, a_hi, b_lo, b_hi;
                pair_exchange(acc[t][0], acc[t][2], 0xAAAAAAAAAAAAAAAAull, a_lo, a_hi);
                pair_exchange(acc[t][1], acc[t][3], 0xAAAAAAAAAAAAAAAAull, b_lo, b_hi);
                const double2_t va = ascending ? (double2_t){a_lo, a_hi} : (double2_t){a_hi, a_lo};
                const double2_t vb = ascending ? (double2_t){b_lo, b_hi} : (double2_t){b_hi, b_lo};
                // The wave-uniform part goes into the vector offset, not into the scalar offset operand of the store: with a
                // REGISTER soffset hipcc assumes that a 16-byte store's data registers may be overwritten by the very next VALU
                // instruction (the documented exemption of the gfx9 store-data hazard) and schedules one there; on gfx950 that
                // corrupted the low dword of the stored value in some lanes of some launches.
                const unsigned lane_col = (ascending ? asc : desc) + (unsigned)soff;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, va), rsrc, row_a + lane_col, 0, SHG_STORE_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, vb), rsrc, row_b + lane_col, 0, SHG_STORE_AUX);
            }
        }
        ROT_STAMP(4 + 2 * min(q, 3));
    }
    S.tx = tx;
    S.abx = abx;
#undef ROT_CLASS
#undef ROT_MFMA2
#undef ROT_MFMA4
#undef ROT_MFMA2_FIRST
#undef ROT_MFMA4_FIRST
#undef ROT_FETCH
}

__device__ __forceinline__ void panel_put(double2_t* panel, int index, double2_t v) { panel[index] = v; }

// Phase 1: Legendre stage of tile (bt, it) (see synthesis_fused.hip).  The orders of the tile are dealt to the waves that call this
// (work-item records `recs`, one list per wave); the result of order m is one 16-byte pair (A_m, B'_m) per panel row, written to
// `panel` -- the LDS panel of the tile, or (persistent kernel) the image of the NEXT tile's panel in global memory.
template <bool NS, typename PanelPtr>
__device__ __forceinline__ void rot_phase1(const RotParams& P, PanelPtr panel, const int4* recs_wave, int bt, int it, int lane) {
    const int fr = lane & 15, fk = lane >> 4;
    // Everything that is the same for all lanes stays on the scalar unit: the work-item records come through scalar loads
    // (constant address space), every operand load is "scalar base + lane offset fixed for the kernel", the flags of a record
    // steer uniform branches, and the two accumulators of an order start from the MFMA's constant-zero operand instead of
    // being cleared with 16 VALU moves.  Measured before: ~9 VALU instructions per MFMA in this stage (64-bit vector address
    // arithmetic from records that sat in vector registers, flag tests, selects, clears); VALU instructions share the issue
    // pipe with the fp64 MFMAs, so the stage was bound by their sum (2 x (6.5 k + 9 k) cycles per SIMD and tile = the 14 us
    // it took), not by the L2 -> L1 rate it had been attributed to.
    constexpr int ASTRIDE = NS ? 128 : 64;
    typedef int int4_v __attribute__((ext_vector_type(4)));
    typedef const int4_v __attribute__((address_space(4))) crec_t;
    // operand loads: buffer loads "descriptor of the tile's table (scalar) + lane offset (fixed) + octet offset (scalar)": not a
    // single vector instruction per load besides the load itself (the global-load form needs a register copy per load to keep
    // its scalar base)
    auto ld16 = [](__amdgpu_buffer_rsrc_t table, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(table, voff, soff, 0));
    };
    const int bad = NS ? P.badmap[it] : -1;
    const int it_tab = SHG_DBG(P, 16) ? 7 : it, bt_tab = SHG_DBG(P, 32) ? 0 : bt;       // experiment: every workgroup reads the same tables
    __amdgpu_buffer_rsrc_t pku = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(P.pkf + (size_t)it_tab * P.Qtot * 128), 0, 0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t cfu =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(NS ? P.cpk4 + (size_t)bt_tab * P.Qtot * 128 : P.cpk4 + (size_t)bt_tab * P.Qtot * 64), 0, 0xffffffffu, 0x00020000);
    const unsigned pk_voff = (unsigned)lane * 16u;
    const unsigned cf_voff = NS ? (unsigned)lane * 16u : (unsigned)(fk * 8 + (fr & 7)) * 16u;
    int mode = NS && bad >= 0 ? 1 : 0;
    int prow = lane;
    const bool arow = NS || fr < 8;
    double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    const double4_t zero4 = {0.0, 0.0, 0.0, 0.0};
    double sgm = (mode == 0 && fr >= 8) ? -1.0 : 1.0;                 // sign of the lane's own part in the north / south combination
    bool fresh = true;                                              // uniform: the next MFMA pair opens an order

#define ROT_P1_ISSUE(rec, ALO, AHI, BLO, BHI)                                                \
do {                                                                                     \
    ALO = ld16(cfu, cf_voff, (unsigned)(rec).x * (ASTRIDE * 8u));                        \
    BLO = ld16(pku, pk_voff, (unsigned)(rec).x * 1024u);                                 \
    AHI = ld16(cfu, cf_voff, (unsigned)(rec).y * (ASTRIDE * 8u));                        \
    BHI = ld16(pku, pk_voff, (unsigned)(rec).y * 1024u);                                 \
} while (0)

#define ROT_P1_CONSUME(rec, ALO, AHI, BLO, BHI)                                                                     \
do {                                                                                                            \
    if ((rec).w & 1) {                                                                                          \
        const double ax_ = NS ? ALO.x : (arow ? ALO.x : 0.0), ay_ = NS ? ALO.y : (arow ? ALO.y : 0.0);          \
        if (fresh) {                                                                                            \
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ax_, BLO.x, zero4, 0, 0, 0);                            \
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ay_, BLO.y, zero4, 0, 0, 0);                            \
        } else {                                                                                                \
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ax_, BLO.x, acc0, 0, 0, 0);                             \
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ay_, BLO.y, acc1, 0, 0, 0);                             \
        }                                                                                                       \
        fresh = false;                                                                                          \
    }                                                                                                           \
    if ((rec).w & 2) {                                                                                          \
        const double ax_ = NS ? AHI.x : (arow ? AHI.x : 0.0), ay_ = NS ? AHI.y : (arow ? AHI.y : 0.0);          \
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ax_, BHI.x, acc0, 0, 0, 0);                                 \
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ay_, BHI.y, acc1, 0, 0, 0);                                 \
    }                                                                                                           \
    if ((rec).w & 4) {                                  /* last item of an order: see synthesis_fused.hip */     \
        double vc_ = acc0[0] + acc1[0], vs_ = acc0[1] + acc1[1];                                                \
        if (NS) {                                                                                               \
            const double oc_ = acc0[2] + acc1[2], os_ = acc0[3] + acc1[3];                                      \
            /* a lane sends x = (E of slots 0-7 | O of slots 8-15) and receives the other part r: E + O = r + x on */ \
            /* the northern slots, E - O = r - x on the mirrored ones: one exact fma with the lane's sign           */ \
            const double xc_ = fr < 8 ? vc_ : oc_, xs_ = fr < 8 ? vs_ : os_;                                   \
            const double rc_ = swap_half_row(xc_), rs_ = swap_half_row(xs_);                                   \
            vc_ = fma(sgm, xc_, rc_);                                                                           \
            vs_ = fma(sgm, xs_, rs_);                                                                           \
        }                                                                                                       \
        if (!NS || mode == 0 || fr < 8) panel_put(panel, (rec).z * 64 + prow, (double2_t){vc_, vs_});           \
        fresh = true;                                                                                           \
    }                                                                                                           \
} while (0)

    double2 xal = {0, 0}, xah = {0, 0}, xbl = {0, 0}, xbh = {0, 0};
    double2 yal = {0, 0}, yah = {0, 0}, ybl = {0, 0}, ybh = {0, 0};
    double2 zal = {0, 0}, zah = {0, 0}, zbl = {0, 0}, zbh = {0, 0};
    double2 wal = {0, 0}, wah = {0, 0}, wbl = {0, 0}, wbh = {0, 0};
    crec_t* recs = reinterpret_cast<crec_t*>(reinterpret_cast<unsigned long long>(recs_wave));
    for (int pass = 0; pass < (mode == 0 ? 1 : 2); ++pass) {
        if (pass == 1) {                                          // mirrored parallels of a polar block: their own table
            mode = 2;
            sgm = 1.0;
            prow = lane + 8;
            pku = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(P.pkf + (size_t)(P.nit + bad) * P.Qtot * 128), 0, 0xffffffffu, 0x00020000);
        }
        int4_v c0 = recs[0], c1 = recs[1], c2 = recs[2];
        int4_v n0 = recs[3], n1 = recs[4], n2 = recs[5], n3 = recs[6];
        // (scheduling barriers: the order of the first loads must be the order the loop consumes them in, see pipe_phase1)
        ROT_P1_ISSUE(c0, xal, xah, xbl, xbh);
        __builtin_amdgcn_sched_barrier(0);
        ROT_P1_ISSUE(c1, yal, yah, ybl, ybh);
        __builtin_amdgcn_sched_barrier(0);
        ROT_P1_ISSUE(c2, zal, zah, zbl, zbh);
        __builtin_amdgcn_sched_barrier(0);
        for (int trip = 0; trip < P.ntrip; ++trip) {
            const int4_v a3 = n0, a4 = n1, a5 = n2, a6 = n3;
            crec_t* nr = recs + 4 * trip + 7;
            n0 = nr[0];
            n1 = nr[1];
            n2 = nr[2];
            n3 = nr[3];
            ROT_P1_ISSUE(a3, wal, wah, wbl, wbh);
            ROT_P1_CONSUME(c0, xal, xah, xbl, xbh);
            ROT_P1_ISSUE(a4, xal, xah, xbl, xbh);
            ROT_P1_CONSUME(c1, yal, yah, ybl, ybh);
            ROT_P1_ISSUE(a5, yal, yah, ybl, ybh);
            ROT_P1_CONSUME(c2, zal, zah, zbl, zbh);
            ROT_P1_ISSUE(a6, zal, zah, zbl, zbh);
            ROT_P1_CONSUME(a3, wal, wah, wbl, wbh);
            c0 = a4;
            c1 = a5;
            c2 = a6;
        }
    }
#undef ROT_P1_ISSUE
#undef ROT_P1_CONSUME
}

// Tokens of the Legendre stage.  A workgroup does not store while it runs its Legendre stage, and workgroups that share the HBM fall
// into step (tools/timeline.py: the whole chip in the Legendre stage at once, the HBM idle, then the whole chip storing at twice the
// rate the HBM takes).  A quarter of the CUs saturate the HBM's write path (tools/store_bench.hip), so it is enough that never more
// than `limit` workgroups are in the stage at once: one lane takes a token before the stage and returns it behind it.  The wait is
// bounded to a few milliseconds (a counter left non-zero by an aborted launch must not hang the next one); the counter is 0 again
// when the grid has drained.
__device__ __forceinline__ void legendre_token_acquire(int* sem, int limit) {
    // Take with an atomic add (any number of workgroups get their token in the same round trip; a compare-and-swap lets one through
    // per round trip: measured 250 us for the first 128), give back when the add shows that the limit was passed.  A waiter only LOOKS
    // (an add of 0: atomics execute at the memory side, a plain agent-scope load is served by this XCD's L2 and saw a returned token
    // ~100 us late) and adds again when it has seen a free token, so that the counter is not inflated by the waiters.
    for (int spin = 0; spin < 600; ++spin) {
        if (__hip_atomic_fetch_add(sem, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < limit) return;
        (void)__hip_atomic_fetch_add(sem, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        do {                               // (long naps: a few hundred waiters polling one word every microsecond keep the releases from it)
            __builtin_amdgcn_s_sleep(127);
            ++spin;
        } while (spin < 600 && __hip_atomic_fetch_add(sem, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= limit);
    }
    (void)__hip_atomic_fetch_add(sem, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // (gave up waiting: the release still pairs)
}
__device__ __forceinline__ void legendre_token_release(int* sem) { (void)__hip_atomic_fetch_add(sem, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <bool NS, int R>
__global__ __launch_bounds__(128 * EP) void synthesis_rot_kernel(RotParams P) {
    constexpr int NW = 2 * EP, PR = 16 * EP;           // waves, panel rows
    using T = RotTraits<R>;
    extern __shared__ __attribute__((aligned(16))) double As[];   // rings [8][kRingSlots][64][2], then panel [nslot + 1][64 rows][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbt = (P.B + 3) >> 2;
    const int tile = (int)blockIdx.x;
    if (tile >= nbt * P.nit) return;
    const int bt = P.blockmap ? P.blockmap[2 * tile] : tile % nbt;
    const int it = P.blockmap ? P.blockmap[2 * tile + 1] : tile / nbt;
    const int fr = lane & 15, fk = lane >> 4;
#ifdef SHG_EXPERIMENT
    if (P.stagger > 0 && blockIdx.x < 256) {
        const long long ticks = (long long)P.stagger * (long long)(((blockIdx.x >> 3) * 13) & 31) / 32;
        const long long t0 = wall_clock64();
        for (int i = 0; i < 100000 && (long long)wall_clock64() - t0 < ticks; ++i) __builtin_amdgcn_s_sleep(8);
    }
#endif
    ROT_STAMP(0);

    double2_t* const panel = reinterpret_cast<double2_t*>(As + NW * kRingSlots * 128);      // [(slot * PR + row)], behind the rings of the waves

    // ---- zero the padding slots of the panel
    {
        int s0 = 0;
        for (int c = 0; c < T::kClasses; ++c) {
            for (int s = s0 + P.cls_cnt[c]; s < s0 + 4 * P.cls_nk[c]; ++s)
                if (tid < PR) panel[s * PR + tid] = (double2_t){0.0, 0.0};
            s0 += 4 * P.cls_nk[c];
        }
    }

    // ---- trig stream of this wave.  The issue side runs kRingDepth pieces ahead of the consumer and never stops (when a wave has
    //      no further column tile it re-reads pieces), so that the count of DMAs in flight is the same at every wait.
    RotStream S;
    rot_stream_init(S, P, As, wave);
#pragma unroll
    for (int d = 0; d < kRingDepth; ++d) rot_issue_piece(S, P, (unsigned)lane * 16u);

    // ---- phase 1: Legendre stage.  Orders are distributed over the waves of the workgroup.
    if (P.sem_limit > 0) {
        if (tid == 0) legendre_token_acquire(P.sem, P.sem_limit);
        __syncthreads();
    }
    if (!SHG_DBG(P, 2)) rot_phase1<NS>(P, panel, P.itemtab + (size_t)wave * P.nrec, bt, it, lane);
    ROT_STAMP(1);
    __syncthreads();          // panel complete; from here on it is read-only and the waves run independently
    if (P.sem_limit > 0 && tid == 0) legendre_token_release(P.sem);
    ROT_STAMP(2);
#ifdef SHG_EXPERIMENT
    if (P.stagger2 > 0 && wave >= EP) {
        const long long t0 = wall_clock64();
        for (int i = 0; i < 100000 && (long long)wall_clock64() - t0 < P.stagger2; ++i) __builtin_amdgcn_s_sleep(4);
    }
#endif

    // ---- phase 2: longitude stage
    if (wave / EP < P.nct && !SHG_DBG(P, 4)) {
        const double2_t* const ringp = reinterpret_cast<const double2_t*>(As) + wave * (kRingSlots * 64) + lane;
        rot_issue_piece(S, P, (unsigned)lane * 16u);       // fragments of the first k-step
        wait_vmcnt<kRingDepth>();
        S.tx = ringp[0];
        S.cslot = 1;
        S.abx = panel[(wave % EP) * 16 + fr + fk * PR];
        S.pf = P.npieces > 1 ? 1 : 0;
        rot_phase2<NS, R>(P, As, panel, S, wave, lane, bt * 4, it);
        // The prefetched pieces of the stream must have landed before the LDS is released -- but not the stores: the 4 R stores of
        // the last unit are the youngest operations of the wave (its last LDS-DMA was issued in the last k-step, before them), and
        // the counter runs in order, so "at most 4 R outstanding" means every DMA is done.  The wave ends with its stores in flight
        // instead of waiting for HBM to acknowledge them.
        wait_vmcnt<4 * R>();
    } else {
        wait_vmcnt<0>();
    }
    ROT_STAMP(12);
}


// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------

// class position in the K sequence (two-sum classes first: r = 0, then r = R / 2 for even R, then r = 1, 2, ...) and sign of
// order m >= 1 for R rotations
static inline void order_class(int R, int m, int& cls, int& sign) {
    const int rho = m % R;
    const int r = 2 * rho <= R ? rho : R - rho;
    sign = 2 * rho <= R ? 1 : -1;
    if (R % 2 == 0)
        cls = r == 0 ? 0 : 2 * r == R ? 1 : r + 1;
    else
        cls = r;
}
static inline int rot_classes(int R) { return R / 2 + 1; }

// sign of the sine coefficients of order m in the repacked coefficient table (0 = no rotation kernel)
int rot_sigma_negative(int R, int m) { return R > 0 && 2 * (m % R) > R ? 1 : 0; }

// True when the meridians are lon_j = -pi + (j + 1/2) 2 pi / nlon to within a few ulp of pi: every one of the 2 R images
// s mu_c + 2 pi k / R of the fundamental domain mu_c = lon[nlon/2 + c] is a grid column.
bool has_rotation_symmetry(int nlon, const double* lon, int R) {
    if (nlon < 192 || nlon % (2 * R) != 0 || nlon % 2 != 0) return false;
    if ((nlon / R) % 16 != 0) return false;              // images of a column tile = whole 128-byte lines (nd = nlon / (2 R) is then even: 16-byte pair stores)
    const double tol = 3e-15;
    const long double pi = 3.141592653589793238462643383279502884L;
    const int n2 = nlon / 2, nR = nlon / R, nd = nlon / (2 * R);
    for (int c = 0; c < nd; ++c) {
        const long double mu = lon[n2 + c];
        for (int k = 0; k < R; ++k) {
            const int jp = (n2 + k * nR + c) % nlon, jm = (n2 + k * nR - 1 - c + nlon) % nlon;
            long double dp = lon[jp] - (mu + 2 * pi * k / R), dm = lon[jm] - (-mu + 2 * pi * k / R);
            dp -= 2 * pi * std::round((double)(dp / (2 * pi)));
            dm -= 2 * pi * std::round((double)(dm / (2 * pi)));
            if (std::fabs((double)dp) > tol || std::fabs((double)dm) > tol) return false;
        }
    }
    return true;
}

// class layout of the panel / trig stream for degree N; returns the panel slots (without the order-0 slot)
int rot_layout(int R, int N, int nk[kMaxClasses], int cnt[kMaxClasses], std::vector<int>* order_slot) {
    const int nc = rot_classes(R);
    for (int c = 0; c < kMaxClasses; ++c) cnt[c] = nk[c] = 0;
    for (int m = 1; m <= N; ++m) {
        int c, s;
        order_class(R, m, c, s);
        cnt[c]++;
    }
    int s = 0, slot[kMaxClasses];
    for (int c = 0; c < nc; ++c) {
        nk[c] = (cnt[c] + 3) / 4;
        slot[c] = s;
        s += 4 * nk[c];
    }
    if (order_slot) {
        order_slot->assign(N + 1, 0);
        int next[kMaxClasses];
        for (int c = 0; c < nc; ++c) next[c] = slot[c];
        for (int m = 1; m <= N; ++m) {
            int c, sg;
            order_class(R, m, c, sg);
            (*order_slot)[m] = next[c]++;
        }
        (*order_slot)[0] = s;
    }
    return s;
}

int rot_kernel_waves() { return kWaves; }

static size_t rot_lds_bytes(int nslot) { return (size_t)kRingDoubles * 8 + (size_t)(nslot + 1) * 1024; }      // rings, panel

static bool rot_fits(int R, int N) {
    int nk[kMaxClasses], cnt[kMaxClasses];
    return rot_lds_bytes(rot_layout(R, N, nk, cnt, nullptr)) <= 160 * 1024;
}

int rot_applicable(const shg_plan* p) {
    if (p->rotR == 0 || p->N < 1) return 0;
    if ((long long)p->nlat * p->nlon * 8 >= (1LL << 31)) return 0;
    return rot_fits(p->rotR, p->N) ? 1 : 0;
}

// The rotation count of a plan: the first of kRotPreference that the meridians allow and whose panel fits the LDS at degree N.
// 10 rotations (20 images; the 0.25 degree grid: 72 columns in the fundamental domain) need 92 MFMAs per 16 rows x 16 columns x 20
// images at d/o 96, 9 rotations (18 images, 80 columns) 102, 6 rotations (12 images, 120 columns) 80.  The 0.5 degree grid (nlon = 720) allows 9
// and 3: 0.167 against 0.222 ms per 240 epochs at d/o 96.
int rot_choose(int nlon, const double* lon_h, int N) {
    static const int kRotPreference[] = {SHG_ROT_PREFERENCE};
    int fallback = 0;
    for (int R : kRotPreference)
        if (has_rotation_symmetry(nlon, lon_h, R)) {
            if (N < 1 || rot_fits(R, N)) return R;
            if (!fallback) fallback = R;
        }
    return fallback;
}

// trig stream [nct][npieces][64][2] (+ one spare piece), built on the host like the other cos/sin tables (grates/utilities.py:272-273)
int build_rot_trig(shg_plan* p, const double* lon_h) {
    if (p->rot_trig) {
        (void)hipFree(p->rot_trig);
        p->rot_trig = nullptr;
    }
    const int R = p->rotR, N = p->N, nlon = p->nlon, nd = nlon / (2 * R), nct = ceil_div(nd, 16);
    int nk[kMaxClasses], cnt[kMaxClasses];
    std::vector<int> order_slot;
    const int nslot = rot_layout(R, N, nk, cnt, &order_slot);
    const int npieces = nslot / 4;
    std::vector<int> slot_order(nslot, -1);
    for (int m = 1; m <= N; ++m) slot_order[order_slot[m]] = m;
    std::vector<double> tab(((size_t)nct * npieces + 1) * 128, 0.0);
    for (int ct = 0; ct < nct; ++ct)
        for (int ks = 0; ks < npieces; ++ks)
            for (int l = 0; l < 64; ++l) {
                const int m = slot_order[4 * ks + (l >> 4)], c = 16 * ct + (l & 15);
                if (m < 0 || c >= nd) continue;
                int cls, sg;
                order_class(R, m, cls, sg);
                const double arg = (double)m * lon_h[nlon / 2 + c];
                double* dst = &tab[(((size_t)ct * npieces + ks) * 64 + l) * 2];
                dst[0] = std::cos(arg);
                dst[1] = sg * std::sin(arg);
            }
    SHG_HIP(hipMalloc((void**)&p->rot_trig, tab.size() * sizeof(double)));
    SHG_HIP(hipMemcpy(p->rot_trig, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    return SHG_OK;
}

template <int R>
static int launch_rot(bool ns, const RotParams& P, size_t lds, dim3 grid_dim, hipStream_t stream) {
    if (ns) {
        SHG_HIP(hipFuncSetAttribute((const void*)synthesis_rot_kernel<true, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((synthesis_rot_kernel<true, R>), grid_dim, dim3(64 * kWaves), lds, stream, P);
    } else {
        SHG_HIP(hipFuncSetAttribute((const void*)synthesis_rot_kernel<false, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((synthesis_rot_kernel<false, R>), grid_dim, dim3(64 * kWaves), lds, stream, P);
    }
    return SHG_OK;
}

int synthesis_rot(shg_plan* p, const double* anm, int B, double* grid, hipStream_t stream) {
    if (!rot_applicable(p)) return fail(SHG_ERR_UNSUPPORTED, "rotation-folded synthesis kernel not applicable to this plan");
    const int R = p->rotR;
    const bool ns = p->sym_ns;
    int rc = build_pkf_table(p, ns, R, stream);
    if (rc) return rc;
    const int nbt = ceil_div(B, 4);
    const int nit = ns ? ceil_div(p->nlat / 2, 8) : ceil_div(p->nlat, 16);
    rc = pack_coefficients_fused(p, ns, R, anm, B, stream);
    if (rc) return rc;
    RotParams P;
    P.N = p->N;
    P.nlat = p->nlat;
    P.nlon = p->nlon;
    P.B = B;
    P.nit = nit;
    P.nh = p->nlat / 2;
    P.nd = p->nlon / (2 * R);
    P.nct = ceil_div(P.nd, 16);
    P.nslot = rot_layout(R, p->N, P.cls_nk, P.cls_cnt, nullptr);
    P.npieces = P.nslot / 4;
#ifdef SHG_EXPERIMENT
    P.dbg = experiment_switches();
    P.stagger = getenv("SHG_STAGGER") ? atoi(getenv("SHG_STAGGER")) : 0;
    P.stagger2 = getenv("SHG_STAGGER2") ? atoi(getenv("SHG_STAGGER2")) : 0;
#endif
    P.Qtot = p->Qtot;
    P.cpk4 = p->cpk4;
    P.pkf = p->pkf;
    P.itemtab = reinterpret_cast<const int4*>(p->itemtab_d);
    P.nrec = p->itemtab_nrec;
    P.ntrip = p->itemtab_ntrip;
    P.sem = p->sem_d;                         // allocated and zeroed by rot_set_stage_limit; 0 again whenever a grid has drained
    P.sem_limit = p->sem_d ? p->stage_limit : 0;
#ifdef SHG_EXPERIMENT
    if (getenv("SHG_SEM") && p->sem_d) P.sem_limit = atoi(getenv("SHG_SEM"));
#endif
    P.badmap = p->badmap_d;
    P.blockmap = nullptr;
    if (!SHG_DBG(P, 2048)) {
        rc = build_blockmap(p, nbt, nit, stream);
        if (rc) return rc;
        P.blockmap = p->blockmap_d;
    }
    P.trig = p->rot_trig;
    P.G = grid;
#ifdef SHG_TIMELINE
    P.tl = getenv("SHG_TIMELINE_PTR") ? (unsigned long long*)strtoull(getenv("SHG_TIMELINE_PTR"), nullptr, 0) : nullptr;
#endif
    const size_t lds = rot_lds_bytes(P.nslot);
    const dim3 grid_dim((unsigned)(nbt * nit));
    ProfileScope ps(p, 2, stream);
    switch (R) {
        case 10: rc = launch_rot<10>(ns, P, lds, grid_dim, stream); break;
        case 9: rc = launch_rot<9>(ns, P, lds, grid_dim, stream); break;
        case 6: rc = launch_rot<6>(ns, P, lds, grid_dim, stream); break;
        case 3: rc = launch_rot<3>(ns, P, lds, grid_dim, stream); break;
        default: return fail(SHG_ERR_UNSUPPORTED, "rotation-folded synthesis kernel: no kernel for %d rotations", R);
    }
    if (rc) return rc;
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

// Limit on the workgroups that run their Legendre stage at once (see legendre_token_acquire).  OFF by default: over six boxes of the
// pool a limit of 7/16 of the CUs took 1.5 - 2 % off the kernel on the slower three and cost 2 - 4 % on the faster three (DESIGN.md
// Appendix A) -- a tuning knob, not a default.  limit > 0: that many workgroups; < 0: that many sixteenths of the device's CUs;
// 0: no limit.  The counter is allocated and zeroed here, synchronously, so that no launch can see it uninitialised.
int rot_set_stage_limit(shg_plan* p, int limit) {
    if (limit < 0) {
        int dev = 0, cus = 0;
        SHG_HIP(hipGetDevice(&dev));
        SHG_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        limit = std::max(1, cus * -limit / 16);
    }
    if (limit > 0 && !p->sem_d) {
        if (hipMalloc((void**)&p->sem_d, 256) != hipSuccess) return fail(SHG_ERR_NOMEM, "token counter allocation failed");
        SHG_HIP(hipMemset(p->sem_d, 0, 256));
        SHG_HIP(hipDeviceSynchronize());
    }
    p->stage_limit = limit;
    return SHG_OK;
}

}  // namespace shg
