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
    double* ring;             // fed pipeline: images of the panels of `ring_tiles` tiles ((nslot + 1) KB each, padding slots zero)
    int ring_tiles;
    int fed_first;            // tiles 0 .. fed_first - 1 are not produced (every consumer workgroup makes the panel of its first tile itself)
    int* produced;            // [ring_tiles] tile + 1 once the image of that tile's panel is complete
    int* consumed;            // [ring_tiles] tile + 1 once the consumer has the image in its LDS (-1: slot given up for this launch)
    int* sem;                 // tokens of the Legendre stage in use (device-wide counter, 0 between launches)
    int sem_limit;            // at most this many workgroups run their Legendre stage at the same time (0 = no limit)
    const int2* itemtab2;     // pipelined kernel: packed work items of its four waves (see build_item_table)
    int nrec2, ntrip2;
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
// EP = epochs (= row tiles) of the workgroup: 4 (eight waves, one workgroup per CU) or 2 (four waves, two workgroups per CU)
template <int EP>
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
template <bool NS, int R, int EP>
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

// Where a panel row goes: into the LDS panel of the tile, or (producer kernel of the fed pipeline) write-through into the image of a
// tile's panel in global memory, from where another workgroup's LDS-DMA takes it (sc1 = aux 16: cdna_hip_programming.md guideline 16, R1)
struct GlobalPanel {
    __amdgpu_buffer_rsrc_t rsrc;
    bool skip;
};
__device__ __forceinline__ void panel_put(double2_t* panel, int index, double2_t v) { panel[index] = v; }
// the 32-row panel of a workgroup that keeps two of the tile's four epochs (index = slot * 64 + epoch * 16 + row of the full panel)
struct HalfPanel {
    double2_t* rows;
    int half;
};
__device__ __forceinline__ void panel_put(const HalfPanel& panel, int index, double2_t v) {
    const int row = index & 63;
    if ((row >> 5) == panel.half) panel.rows[(index >> 6) * 32 + (row & 31)] = v;
}
__device__ __forceinline__ void panel_put(const GlobalPanel& panel, int index, double2_t v) {
    if (panel.skip) {                       // (experiment builds: the producer without its stores)
        asm volatile("" ::"v"(v));
        return;
    }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, v), panel.rsrc, (unsigned)index * 16u, 0, 16);
}

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

template <bool NS, int R, int EP>
__global__ __launch_bounds__(128 * EP) void synthesis_rot_kernel(RotParams P) {
    constexpr int NW = 2 * EP, PR = 16 * EP;           // waves, panel rows
    using T = RotTraits<R>;
    extern __shared__ __attribute__((aligned(16))) double As[];   // rings [8][kRingSlots][64][2], then panel [nslot + 1][64 rows][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbt = (P.B + 3) >> 2;
    // EP == 2: the workgroups b and b + 8 (one XCD) are the two halves of a tile: epochs 0, 1 and 2, 3
    const int tile = EP == 4 ? (int)blockIdx.x : (int)((blockIdx.x >> 4) * 8 + (blockIdx.x & 7));
    const int half = EP == 4 ? 0 : (int)((blockIdx.x >> 3) & 1);
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
    rot_stream_init<EP>(S, P, As, wave);
#pragma unroll
    for (int d = 0; d < kRingDepth; ++d) rot_issue_piece(S, P, (unsigned)lane * 16u);

    // ---- phase 1: Legendre stage.  Orders are distributed over the waves of the workgroup.
    if (P.sem_limit > 0) {
        if (tid == 0) legendre_token_acquire(P.sem, P.sem_limit);
        __syncthreads();
    }
    if (!SHG_DBG(P, 2)) {
        if (EP == 4) {
            rot_phase1<NS>(P, panel, P.itemtab + (size_t)wave * P.nrec, bt, it, lane);
        } else {                  // two of the eight item lists per wave; the rows of the other two epochs are computed and dropped
            const HalfPanel hp = {panel, half};
            rot_phase1<NS>(P, hp, P.itemtab + (size_t)wave * P.nrec, bt, it, lane);
            rot_phase1<NS>(P, hp, P.itemtab + (size_t)(wave + 4) * P.nrec, bt, it, lane);
        }
    }
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
        rot_phase2<NS, R, EP>(P, As, panel, S, wave, lane, bt * 4 + half * 2, it);
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

// =====================================================================================================================
// Pipelined variant (round 5): ONE wave per SIMD, the accumulators of a unit twice.
//
// What bounds synthesis_rot_kernel is not its instruction count but when it stores: every wave issues the 40 stores of a
// unit in one burst at the end of the unit, all eight waves of a workgroup at about the same time, no wave stores anything
// during the Legendre stage or the k-loop of a unit, and a wave's next k-loop waits for the acknowledgement of its burst
// (one in-order vmcnt for stores and trig pieces).  Measured (tools/timeline.py, round 5): the no-store kernel takes 0.38 ms,
// the stores alone 0.37 ms, together 0.50 ms; staggering workgroups or waves moves nothing (the chip falls back into step).
//
// Here a wave owns 512 registers: the 2 R images of unit u stay in registers while unit u + 1 accumulates into a second set,
// and leave during that k-loop, a group of images behind every class of the K sequence (8-byte stores straight from the
// accumulator registers, 4 rows x 128 bytes per instruction, wave-uniform address part in the scalar offset: no lane exchange
// and no vector address arithmetic).  The trig stream no longer shares the in-order counter with the stores inside the
// k-loop: the four waves of a workgroup (one row tile = one epoch each) walk the column tiles together, the pieces of column tile
// ct + 1 are copied into the second half of a double buffer by LDS-DMA at the start of unit ct, and the only wait for them
// stands at the end of the unit (all of the unit's stores are younger than they: vmcnt(63) never waits for a recent store),
// followed by the workgroup's one barrier per unit.
// =====================================================================================================================
#ifndef SHG_LEGENDRE_TOKENS
#define SHG_LEGENDRE_TOKENS -7      // > 0: that many; < 0: that many sixteenths of the device's CUs; 0: no limit
#endif
// Workgroups that may run their Legendre stage at once.  Measured (round 5, 240 x d/o 96 -> 0.25 degree, three boxes, alternating
// processes): 96 .. 128 of 256 take 1.5 - 2 % off the kernel (0.497 / 0.472 / 0.497 against 0.505 / 0.480 / 0.508 ms), 64 cost 9 %.
constexpr int kLegendreTokens = SHG_LEGENDRE_TOKENS;
constexpr int kPipeWaves = 4;
#ifndef SHG_PIPE_X
#define SHG_PIPE_X 0           // experiment switches of the pipelined kernel's Legendre stage (timing only): 1 no arithmetic, 2 no operand loads
#endif

typedef int int8_v __attribute__((ext_vector_type(8)));

// Legendre stage of the pipelined kernel.  With one wave per SIMD nothing hides the end of an order -- the last MFMA's latency, the
// exchange between the hemispheres, the write of the panel row -- unless the wave itself has other work: every wave runs TWO
// independent item lists (streams A and B, the lists of waves w and w + 4 of an eight-wave dealing) interleaved item by item,
// each with its own accumulator pair and four operand sets (three items in flight per stream).  Records are packed into two words
// (x = first octet | second octet << 16, y = panel slot | flags << 16; four records per 32-byte scalar load).
constexpr int kPipeChunks = 8;          // the images still parked when a tile begins leave in this many groups during its Legendre stage

template <bool NS, class Flush>
__device__ __forceinline__ void pipe_phase1(const RotParams& P, double2_t* panel, const int2* recs_a, const int2* recs_b, int bt, int it, int lane,
                                            bool pending, Flush&& flush_chunk) {
    const int fr = lane & 15, fk = lane >> 4;
    constexpr int ASTRIDE = NS ? 128 : 64;
    typedef const int8_v __attribute__((address_space(4))) crec_t;
    auto ld16 = [](__amdgpu_buffer_rsrc_t table, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(table, voff, soff, 0));
    };
    const int bad = NS ? P.badmap[it] : -1;
    __amdgpu_buffer_rsrc_t pku = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(P.pkf + (size_t)it * P.Qtot * 128), 0, 0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t cfu =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(NS ? P.cpk4 + (size_t)bt * P.Qtot * 128 : P.cpk4 + (size_t)bt * P.Qtot * 64), 0, 0xffffffffu, 0x00020000);
    const unsigned pk_voff = (unsigned)lane * 16u;
    const unsigned cf_voff = NS ? (unsigned)lane * 16u : (unsigned)(fk * 8 + (fr & 7)) * 16u;
    int mode = NS && bad >= 0 ? 1 : 0;
    int prow = lane;
    const bool arow = NS || fr < 8;
    double4_t accA0 = {0.0, 0.0, 0.0, 0.0}, accA1 = accA0, accB0 = accA0, accB1 = accA0;
    const double4_t zero4 = {0.0, 0.0, 0.0, 0.0};
    double sgm = (mode == 0 && fr >= 8) ? -1.0 : 1.0;
    bool freshA = true, freshB = true;

#define PIPE_P1_ISSUE(rx, S)                                                                      \
    do {                                                                                          \
        const unsigned lo_ = (unsigned)(rx) & 0xffffu, hi_ = (unsigned)(rx) >> 16;                \
        if (SHG_PIPE_X & 2) break;          /* experiment: no operand loads */                     \
        al##S = ld16(cfu, cf_voff, lo_ * (ASTRIDE * 8u));                                         \
        bl##S = ld16(pku, pk_voff, lo_ * 1024u);                                                  \
        ah##S = ld16(cfu, cf_voff, hi_ * (ASTRIDE * 8u));                                         \
        bh##S = ld16(pku, pk_voff, hi_ * 1024u);                                                  \
    } while (0)

#define PIPE_P1_CONSUME(ry, S, Q)                                                                                   \
    do {                                                                                                            \
        const int fl_ = (ry) >> 16;                                                                                 \
        if (SHG_PIPE_X & 1) {               /* experiment: loads only */                                            \
            asm volatile("" ::"v"(al##S.x), "v"(al##S.y), "v"(ah##S.x), "v"(ah##S.y), "v"(bl##S.x), "v"(bl##S.y), "v"(bh##S.x), "v"(bh##S.y)); \
            break;                                                                                                  \
        }                                                                                                           \
        if (fl_ & 1) {                                                                                              \
            const double ax_ = NS ? al##S.x : (arow ? al##S.x : 0.0), ay_ = NS ? al##S.y : (arow ? al##S.y : 0.0);  \
            if (fresh##Q) {                                                                                         \
                acc##Q##0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ax_, bl##S.x, zero4, 0, 0, 0);                     \
                acc##Q##1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ay_, bl##S.y, zero4, 0, 0, 0);                     \
            } else {                                                                                                \
                acc##Q##0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ax_, bl##S.x, acc##Q##0, 0, 0, 0);                 \
                acc##Q##1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ay_, bl##S.y, acc##Q##1, 0, 0, 0);                 \
            }                                                                                                       \
            fresh##Q = false;                                                                                       \
        }                                                                                                           \
        if (fl_ & 2) {                                                                                              \
            const double ax_ = NS ? ah##S.x : (arow ? ah##S.x : 0.0), ay_ = NS ? ah##S.y : (arow ? ah##S.y : 0.0);  \
            acc##Q##0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ax_, bh##S.x, acc##Q##0, 0, 0, 0);                     \
            acc##Q##1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ay_, bh##S.y, acc##Q##1, 0, 0, 0);                     \
        }                                                                                                           \
        if (fl_ & 4) {                                  /* last item of an order: see rot_phase1 */                  \
            double vc_ = acc##Q##0[0] + acc##Q##1[0], vs_ = acc##Q##0[1] + acc##Q##1[1];                            \
            if (NS) {                                                                                               \
                const double oc_ = acc##Q##0[2] + acc##Q##1[2], os_ = acc##Q##0[3] + acc##Q##1[3];                  \
                const double xc_ = fr < 8 ? vc_ : oc_, xs_ = fr < 8 ? vs_ : os_;                                   \
                const double rc_ = swap_half_row(xc_), rs_ = swap_half_row(xs_);                                   \
                vc_ = fma(sgm, xc_, rc_);                                                                           \
                vs_ = fma(sgm, xs_, rs_);                                                                           \
            }                                                                                                       \
            if (!NS || mode == 0 || fr < 8) panel[((ry) & 0xffff) * 64 + prow] = (double2_t){vc_, vs_};             \
            fresh##Q = true;                                                                                        \
        }                                                                                                           \
    } while (0)

    double2 al0 = {0, 0}, ah0 = {0, 0}, bl0 = {0, 0}, bh0 = {0, 0}, al1 = {0, 0}, ah1 = {0, 0}, bl1 = {0, 0}, bh1 = {0, 0};
    double2 al2 = {0, 0}, ah2 = {0, 0}, bl2 = {0, 0}, bh2 = {0, 0}, al3 = {0, 0}, ah3 = {0, 0}, bl3 = {0, 0}, bh3 = {0, 0};
    double2 al4 = {0, 0}, ah4 = {0, 0}, bl4 = {0, 0}, bh4 = {0, 0}, al5 = {0, 0}, ah5 = {0, 0}, bl5 = {0, 0}, bh5 = {0, 0};
    double2 al6 = {0, 0}, ah6 = {0, 0}, bl6 = {0, 0}, bh6 = {0, 0}, al7 = {0, 0}, ah7 = {0, 0}, bl7 = {0, 0}, bh7 = {0, 0};
    crec_t* ra = reinterpret_cast<crec_t*>(reinterpret_cast<unsigned long long>(recs_a));
    crec_t* rb = reinterpret_cast<crec_t*>(reinterpret_cast<unsigned long long>(recs_b));
    for (int pass = 0; pass < (mode == 0 ? 1 : 2); ++pass) {
        if (pass == 1) {                                          // mirrored parallels of a polar block: their own table
            mode = 2;
            sgm = 1.0;
            prow = lane + 8;
            pku = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(P.pkf + (size_t)(P.nit + bad) * P.Qtot * 128), 0, 0xffffffffu, 0x00020000);
        }
        // per stream: records of the current trip (items 0 .. 3; items 0 .. 2 are in flight in sets 0 .. 2 / 4 .. 6) and of the next one
        int8_v ca = ra[0], cb = rb[0], na = ra[1], nb = rb[1];
        int chunk = pass == 0 && pending ? 0 : kPipeChunks, next_at = 0;
        // (the scheduling barriers keep the order of the first loads: left alone hipcc sorts them so that the set the loop consumes first is
        //  loaded LAST, and its wait-count analysis then drains every load at the head of every trip -- s_waitcnt vmcnt(4) instead of (24))
        PIPE_P1_ISSUE(ca[0], 0);
        __builtin_amdgcn_sched_barrier(0);
        PIPE_P1_ISSUE(cb[0], 4);
        __builtin_amdgcn_sched_barrier(0);
        PIPE_P1_ISSUE(ca[2], 1);
        __builtin_amdgcn_sched_barrier(0);
        PIPE_P1_ISSUE(cb[2], 5);
        __builtin_amdgcn_sched_barrier(0);
        PIPE_P1_ISSUE(ca[4], 2);
        __builtin_amdgcn_sched_barrier(0);
        PIPE_P1_ISSUE(cb[4], 6);
        __builtin_amdgcn_sched_barrier(0);
        for (int trip = 0; trip < P.ntrip2; ++trip) {
            if (chunk < kPipeChunks && trip >= next_at) {          // (uniform) the next group of parked images leaves
                flush_chunk(chunk);
                ++chunk;
                next_at = (chunk * P.ntrip2) / kPipeChunks;
            }
            PIPE_P1_ISSUE(ca[6], 3);
            PIPE_P1_CONSUME(ca[1], 0, A);
            PIPE_P1_ISSUE(cb[6], 7);
            PIPE_P1_CONSUME(cb[1], 4, B);
            PIPE_P1_ISSUE(na[0], 0);
            PIPE_P1_CONSUME(ca[3], 1, A);
            PIPE_P1_ISSUE(nb[0], 4);
            PIPE_P1_CONSUME(cb[3], 5, B);
            PIPE_P1_ISSUE(na[2], 1);
            PIPE_P1_CONSUME(ca[5], 2, A);
            PIPE_P1_ISSUE(nb[2], 5);
            PIPE_P1_CONSUME(cb[5], 6, B);
            PIPE_P1_ISSUE(na[4], 2);
            PIPE_P1_CONSUME(ca[7], 3, A);
            PIPE_P1_ISSUE(nb[4], 6);
            PIPE_P1_CONSUME(cb[7], 7, B);
            ca = na;
            cb = nb;
            na = ra[trip + 2];
            nb = rb[trip + 2];
        }
        for (; chunk < kPipeChunks; ++chunk) flush_chunk(chunk);
    }
#undef PIPE_P1_ISSUE
#undef PIPE_P1_CONSUME
}

// A value parked in the accumulator half of the register file.  gfx950 gives a wave 512 registers, but vector ALU instructions
// address only the first 256 (the architectural VGPRs); the others (AGPRs) can be MFMA accumulators and the DATA of loads and
// stores.  The images of a unit wait for their stores there: the empty statement ties its result to an AGPR, hipcc moves the
// value across with v_accvgpr_write and the architectural registers are free for the next unit's sums.
__device__ __forceinline__ double park(double v) {
    double a;
    asm volatile("; park" : "=a"(a) : "0"(v));
    return a;
}
// 8-byte store as store_b64_soff with its data in AGPRs
__device__ __forceinline__ void store_b64_soff_acc(double v, int4_s rsrc, unsigned voff, unsigned soff) {
    // (s_mov + s_nop 3 = five wait states: hipcc may reload the DESCRIPTOR from a spill lane with v_readlane right in front of the
    //  statement too, and a VALU-written SGPR must not be read by a vector memory instruction earlier)
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 3\n\tbuffer_store_dwordx2 %0, %1, %2, m0 offen nt"
                 :
                 : "a"(v), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

template <bool NS, int R>
__device__ __forceinline__ void pipe_unit(const RotParams& P, double (&Y)[2 * R][4], const double2_t* tb, const double2_t* prow,
                                          double4_t z4, bool pending, int4_s rs, const unsigned (&va)[4], const unsigned (&vd)[4], int ct_prev) {
    double4_t X[2 * R];
    using T = RotTraits<R>;
    const double4_t kZero4 = {0.0, 0.0, 0.0, 0.0};
    const int n2 = P.nlon >> 1, nR = P.nlon / R;
    int pf = 0;
    double2_t tx, abx, ty = {0.0, 0.0}, aby = {0.0, 0.0};
#define PIPE_FETCH(T_, AB_)                        \
    do {                                           \
        T_ = tb[pf * 64];                          \
        AB_ = prow[pf * 256];                      \
        pf = pf + 1 < P.npieces ? pf + 1 : pf;     /* (the fetch behind the last k-step re-reads the last one) */ \
    } while (0)
#define PIPE_MFMA2(A0, T_, AB_)                                                                     \
    X[A0] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.x, T_.x, X[A0], 0, 0, 0);                      \
    X[A0 + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.y, T_.y, X[A0 + 1], 0, 0, 0)
#define PIPE_MFMA4(A0, T_, AB_)                                                                     \
    X[A0] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.x, T_.x, X[A0], 0, 0, 0);                      \
    X[A0 + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.x, T_.y, X[A0 + 1], 0, 0, 0);              \
    X[A0 + 2] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.y, T_.x, X[A0 + 2], 0, 0, 0);              \
    X[A0 + 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.y, T_.y, X[A0 + 3], 0, 0, 0)
#define PIPE_MFMA2_FIRST(A0, T_, AB_)                                                               \
    X[A0] = (A0) == 0 ? __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.x, T_.x, X[A0], 0, 0, 0)            \
                      : __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.x, T_.x, kZero4, 0, 0, 0);         \
    X[A0 + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.y, T_.y, kZero4, 0, 0, 0)
#define PIPE_MFMA4_FIRST(A0, T_, AB_)                                                               \
    X[A0] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.x, T_.x, kZero4, 0, 0, 0);                     \
    X[A0 + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.x, T_.y, kZero4, 0, 0, 0);                 \
    X[A0 + 2] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.y, T_.x, kZero4, 0, 0, 0);                 \
    X[A0 + 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(AB_.y, T_.y, kZero4, 0, 0, 0)
    // one class: k-steps in pairs on the register sets (tx, abx) / (ty, aby), as in rot_phase2
#define PIPE_CLASS(C, MF, A0, NACC)                                                                       \
    if (!SHG_DBG(P, 8)) {                                                                                 \
        const int nk_ = P.cls_nk[C];                                                                      \
        int i_ = 0;                                                                                       \
        if (nk_ >= 2) {                                                                                   \
            PIPE_FETCH(ty, aby);                                                                          \
            MF##_FIRST(A0, tx, abx);                                                                      \
            PIPE_FETCH(tx, abx);                                                                          \
            MF(A0, ty, aby);                                                                              \
            i_ = 2;                                                                                       \
        } else if (nk_ == 1) {                                                                            \
            PIPE_FETCH(ty, aby);                                                                          \
            MF##_FIRST(A0, tx, abx);                                                                      \
            tx = ty;                                                                                      \
            abx = aby;                                                                                    \
            i_ = 1;                                                                                       \
        } else {                                                                                          \
            _Pragma("unroll") for (int z_ = ((A0) == 0 ? 1 : 0); z_ < (NACC); ++z_) X[(A0) + z_] = kZero4; \
        }                                                                                                 \
        for (; i_ + 2 <= nk_; i_ += 2) {                                                                  \
            PIPE_FETCH(ty, aby);                                                                          \
            MF(A0, tx, abx);                                                                              \
            PIPE_FETCH(tx, abx);                                                                          \
            MF(A0, ty, aby);                                                                              \
        }                                                                                                 \
        if (i_ < nk_) {                                                                                   \
            PIPE_FETCH(ty, aby);                                                                          \
            MF(A0, tx, abx);                                                                              \
            tx = ty;                                                                                      \
            abx = aby;                                                                                    \
        }                                                                                                 \
    }
    PIPE_FETCH(tx, abx);                                     // fragments of the first k-step
    X[0] = z4;                                               // order 0 does not depend on the longitude: start value of CA_0 (rows fk + 4 reg)
#pragma unroll
    for (int c = 0; c < T::kClasses; ++c) {
        if (c < T::kTwo) {
            PIPE_CLASS(c, PIPE_MFMA2, 2 * c, 2)
        } else {
            PIPE_CLASS(c, PIPE_MFMA4, 2 * T::kTwo + 4 * (c - T::kTwo), 4)
        }
        if (pending) {
            // the images t = c, c + classes, ... of the previous unit leave
#pragma unroll
            for (int t = c; t < 2 * R; t += T::kClasses) {
                const int k = t < R ? t : t - R;
                const bool ascending = t < R;
                int w = n2 + k * nR - (ascending ? 0 : P.nd);
                w = w >= P.nlon ? w - P.nlon : w;
                const unsigned soff = (unsigned)(ascending ? w + 16 * ct_prev : w + P.nd - 16 * ct_prev - 16) * 8u;
#pragma unroll
                for (int r = 0; r < 4; ++r) store_b64_soff_acc(Y[t][r], rs, ascending ? va[r] : vd[r], soff);
            }
        }
    }
    if (!SHG_DBG(P, 8)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) rot_images<R>(X, r);
        // the images wait for their stores in the accumulator registers
#pragma unroll
        for (int t = 0; t < 2 * R; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) Y[t][r] = park(X[t][r]);
    }
#undef PIPE_CLASS
#undef PIPE_MFMA2
#undef PIPE_MFMA4
#undef PIPE_MFMA2_FIRST
#undef PIPE_MFMA4_FIRST
#undef PIPE_FETCH
}

// the images t = T0, T0 + STEP, ... of a unit as 8-byte stores (ncol = 16, or 8 for a half tile at the end of the fundamental domain)
template <int R, int T0, int STEP>
__device__ __forceinline__ void pipe_flush(const double (&Y)[2 * R][4], int4_s rs, const unsigned (&va)[4], const unsigned (&vd)[4], int nlon, int nd, int ct, int ncol) {
    const int n2 = nlon >> 1, nR = nlon / R;
#pragma unroll
    for (int t = T0; t < 2 * R; t += STEP) {
        const int k = t < R ? t : t - R;
        const bool ascending = t < R;
        int w = n2 + k * nR - (ascending ? 0 : nd);
        w = w >= nlon ? w - nlon : w;
        const unsigned soff = (unsigned)(ascending ? w + 16 * ct : w + nd - 16 * ct - ncol) * 8u;
#pragma unroll
        for (int r = 0; r < 4; ++r) store_b64_soff_acc(Y[t][r], rs, ascending ? va[r] : vd[r], soff);
    }
}

// =====================================================================================================================
// Fed pipeline (path 8): the Legendre stage as a kernel of its own that runs BESIDE the pipelined longitude kernel on every CU.
//
// Measured (round 5, one card): the longitude stage + stores of either rotation-folded kernel alone take 0.437 ms, 1.08 x the time the
// HBM needs for the grids (0.405 ms, tools/store_bench.hip); the Legendre stage adds 0.07 - 0.11 ms because no wave of the workgroup
// stores while it runs.  The pipelined kernel leaves 112 registers per SIMD free: one more wave, of another kernel.  So
// `legendre_panel_kernel` (four waves of <= 104 registers, no LDS to speak of) computes the panels of the tiles in the consumer's
// order into a ring of panel images in global memory (write-through stores; the ring lives in L2 / Infinity Cache), and the persistent
// consumer takes each image into its LDS by LDS-DMA and goes straight to the longitude stage: its stores never stop, and the
// Legendre stage's MFMAs fill the gaps of the longitude stage's on the same SIMDs.
// Hand-off per tile, agent scope (cdna_hip_programming.md guideline 16, R1): producer -- sc1 stores, every wave drains, barrier, one
// lane stores produced[slot] = tile + 1; consumer -- one lane polls that word, acquire, barrier, LDS-DMA, barrier, consumed[slot] =
// tile + 1 (the ring slot may be overwritten).  Both waits are BOUNDED and nobody depends on them: a consumer that does not get its
// image in time computes the panel itself (and poisons the slot for the rest of the launch: consumed = -1), a producer that does not
// get its slot in time skips its tile.  Results do not depend on whether, or how far, the two kernels overlap.
// =====================================================================================================================
constexpr int kFedPollNaps = 400;          // x ~0.9 us: the consumer's patience for one panel image (the producer's for one ring slot)

__device__ __forceinline__ bool wait_word_equals(int* word, int want, int naps) {
    for (int i = 0; i < naps; ++i) {
        if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want) return true;
        __builtin_amdgcn_s_sleep(32);
    }
    return false;
}

template <bool NS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(104))) void legendre_panel_kernel(RotParams P) {
    __shared__ int go;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x + P.fed_first;            // (the consumers make the panels of their first tiles themselves: nothing to wait for at the start)
    const int nbt = (P.B + 3) >> 2;
    const int bt = P.blockmap ? P.blockmap[2 * tile] : tile % nbt;
    const int it = P.blockmap ? P.blockmap[2 * tile + 1] : tile / nbt;
    const int slot = tile % P.ring_tiles;
    // This kernel has a third of the longitude kernel's arithmetic and should run AHEAD of it (up to the ring's length): its waves take
    // every issue slot they can use, the longitude kernel's wave on the same SIMD fills the rest (priority, then age)
    __builtin_amdgcn_s_setprio(3);
    if (tid == 0) go = tile < P.ring_tiles + P.fed_first || wait_word_equals(P.consumed + slot, tile - P.ring_tiles + 1, kFedPollNaps) ? 1 : 0;
    __syncthreads();
    if (!go) return;                                   // (the consumer of this tile computes the panel itself)
    const size_t image_doubles = (size_t)(P.nslot + 1) * 128;
    GlobalPanel panel = {__builtin_amdgcn_make_buffer_rsrc(P.ring + (size_t)slot * image_doubles, 0, (unsigned)(image_doubles * 8), 0x00020000), (bool)SHG_DBG(P, 256)};
    // the eight item lists of the rotation-folded kernel, two per wave
    rot_phase1<NS>(P, panel, P.itemtab + (size_t)wave * P.nrec, bt, it, lane);
    rot_phase1<NS>(P, panel, P.itemtab + (size_t)(wave + 4) * P.nrec, bt, it, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores ...
    __syncthreads();
    if (tid == 0) __hip_atomic_store(P.produced + slot, tile + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ... before ONE lane says so
}

#ifdef SHG_TIMELINE
#define PIPE_STAMP(ev)                                                                                        \
    do {                                                                                                      \
        if (P.tl && lane == 0) P.tl[((size_t)tile * kWaves + wave) * 16 + (ev)] = wall_clock64();                  \
        if (P.tl && lane == 0 && ((ev) == 0 || (ev) == 12))                                                   \
            P.tl[((size_t)tile * kWaves + wave) * 16 + ((ev) == 0 ? 13 : 14)] = __builtin_amdgcn_s_memtime();       \
    } while (0)
#else
#define PIPE_STAMP(ev)
#endif

template <bool NS, int R, bool FED>
__global__ __launch_bounds__(64 * kPipeWaves, 1) void synthesis_pipe_kernel(RotParams P) {
    using T = RotTraits<R>;
    extern __shared__ __attribute__((aligned(16))) double As[];   // trig buffers [2][npieces][64][2], then panel [nslot + 1][64 rows][2]
    __shared__ int fed_ready;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // = row tile = epoch of the tile
    const int rt = wave;
    const int nbt = (P.B + 3) >> 2;
    const int ntiles = nbt * P.nit;
    const int fr = lane & 15, fk = lane >> 4;
    const int tb_doubles = P.npieces * 128;
    double2_t* const panel = reinterpret_cast<double2_t*>(As + 2 * tb_doubles);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)As;
    const unsigned lane_off = (unsigned)lane * 16u;
    const int grid_bytes = P.nlat * P.nlon * 8;
    const int nct = P.nct, ctl = nct - 1;
    const int ncol_last = min(16, P.nd - 16 * ctl);                 // columns of the last column tile: 16, or 8 (half tile)
    const double2_t* const prow = panel + rt * 16 + fr + fk * 64;  // + 256 p: k-step p
    const double2_t* const tb0 = reinterpret_cast<const double2_t*>(As) + lane;
    const double2_t* const tb1 = tb0 + P.npieces * 64;
    // The trig pieces of a unit are issued before the unit's 8 R stores: "at most min(8 R, 63) operations outstanding" means they have landed
    constexpr int kYounger = 8 * R < 63 ? 8 * R : 63;

    // trig pieces of column tile ct -> buffer (ct & 1), dealt to the four waves
    auto issue_trig = [&](int ct) {
        const double* src = P.trig + (size_t)ct * P.npieces * 128;
        const unsigned dst = lds0 + (unsigned)(ct & 1) * (unsigned)tb_doubles * 8u;
        for (int j = wave; j < P.npieces; j += kPipeWaves) glds16(src + (size_t)j * 128, lane_off, dst + (unsigned)j * 1024u);
    };

    // ---- zero the padding slots of the panel (the Legendre stage never writes them)
    {
        int s0 = 0;
        for (int c = 0; c < T::kClasses; ++c) {
            for (int s = s0 + P.cls_cnt[c]; s < s0 + 4 * P.cls_nk[c]; ++s)
                if (tid < 64) panel[s * 64 + tid] = (double2_t){0.0, 0.0};
            s0 += 4 * P.cls_nk[c];
        }
    }

#ifdef SHG_EXPERIMENT
    if (P.stagger > 0) {             // experiment: the workgroups start up to P.stagger ticks apart (32 steps)
        const long long ticks = (long long)P.stagger * (long long)(((blockIdx.x >> 3) * 13) & 31) / 32;
        const long long t0 = wall_clock64();
        for (int i = 0; i < 100000 && (long long)wall_clock64() - t0 < ticks; ++i) __builtin_amdgcn_s_sleep(8);
    }
#endif
    double Y[T::kAcc][4];            // images of the previous unit, parked in AGPRs
#pragma unroll
    for (int t = 0; t < T::kAcc; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) Y[t][r] = park(0.0);
    bool have_last = false;          // the images of the previous tile's last unit are still parked

    // The workgroup is persistent: it takes the tiles blockIdx.x, blockIdx.x + gridDim.x, ... of the (XCD-aware) tile order, so that the
    // images of a tile's last unit can leave during the Legendre stage of the next tile: the chip's store stream never stops.
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int bt = P.blockmap ? P.blockmap[2 * tile] : tile % nbt;
        const int it = P.blockmap ? P.blockmap[2 * tile + 1] : tile / nbt;
        PIPE_STAMP(0);
        issue_trig(0);            // (buffer 0 is free: every wave has passed the barrier behind the last unit that read it)
        auto grid_row_of = [&](int it_, int s_) { return NS ? (s_ < 8 ? it_ * 8 + s_ : P.nlat - 1 - (it_ * 8 + s_ - 8)) : it_ * 16 + s_; };
        auto slot_valid_of = [&](int it_, int s_) { return NS ? it_ * 8 + (s_ & 7) < P.nh : it_ * 16 + s_ < P.nlat; };
        auto descriptor_of = [&](int bt_) {
            const int b_ = min(bt_ * 4 + rt, P.B - 1);
            const unsigned long long g_ = (unsigned long long)(P.G + (size_t)b_ * P.nlat * P.nlon);
            return (int4_s){(int)(unsigned)g_, (int)(unsigned)((g_ >> 32) & 0xffffu), grid_bytes, 0x00020000};
        };
        // what the parked images of the previous tile's last unit need to leave: its grid (descriptor) and its rows, from the tile order again
        // (scalars carried around the tile loop end up in vector registers, which the store's descriptor operand cannot be)
        const int tile_prev = max(tile - (int)gridDim.x, 0);
        const int bt_prev = P.blockmap ? P.blockmap[2 * tile_prev] : tile_prev % nbt;
        const int it_prev = P.blockmap ? P.blockmap[2 * tile_prev + 1] : tile_prev / nbt;
        const int4_s rs_last = descriptor_of(bt_prev);
        unsigned va_last[4], vd_last[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int sl = fk + 4 * r;
            const unsigned ro = (unsigned)grid_row_of(it_prev, sl) * (unsigned)P.nlon * 8u;
            const bool ok = slot_valid_of(it_prev, sl) && fr < ncol_last;
            va_last[r] = ok ? ro + (unsigned)fr * 8u : 0x80000000u;
            vd_last[r] = ok ? ro + (unsigned)(ncol_last - 1 - fr) * 8u : 0x80000000u;
        }

        // ---- phase 1: Legendre stage, the orders dealt to the four waves (two lists each); the parked images leave in groups meanwhile
        auto flush_chunk = [&](int c) {
            switch (c) {
                case 0: pipe_flush<R, 0, kPipeChunks>(Y, rs_last, va_last, vd_last, P.nlon, P.nd, ctl, ncol_last); break;
                case 1: pipe_flush<R, 1, kPipeChunks>(Y, rs_last, va_last, vd_last, P.nlon, P.nd, ctl, ncol_last); break;
                case 2: pipe_flush<R, 2, kPipeChunks>(Y, rs_last, va_last, vd_last, P.nlon, P.nd, ctl, ncol_last); break;
                case 3: pipe_flush<R, 3, kPipeChunks>(Y, rs_last, va_last, vd_last, P.nlon, P.nd, ctl, ncol_last); break;
                case 4: pipe_flush<R, 4, kPipeChunks>(Y, rs_last, va_last, vd_last, P.nlon, P.nd, ctl, ncol_last); break;
                case 5: pipe_flush<R, 5, kPipeChunks>(Y, rs_last, va_last, vd_last, P.nlon, P.nd, ctl, ncol_last); break;
                case 6: pipe_flush<R, 6, kPipeChunks>(Y, rs_last, va_last, vd_last, P.nlon, P.nd, ctl, ncol_last); break;
                default: pipe_flush<R, 7, kPipeChunks>(Y, rs_last, va_last, vd_last, P.nlon, P.nd, ctl, ncol_last); break;
            }
        };
        bool self_made = !FED;
        if (FED) {
            // the image of this tile's panel: wait for it (one lane, bounded), acquire, LDS-DMA; the parked images leave meanwhile
            const int slot = tile % P.ring_tiles;
            if (tid == 0) {
                const bool ok = tile >= P.fed_first && wait_word_equals(P.produced + slot, tile + 1, kFedPollNaps);
                if (ok) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                fed_ready = ok ? 1 : 0;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (holds the barrier until the invalidate has completed)
            }
            __syncthreads();
            if (fed_ready) {
                const double* image = P.ring + (size_t)slot * (size_t)(P.nslot + 1) * 128;
                const unsigned dst = lds0 + 2u * (unsigned)tb_doubles * 8u;
                if (!SHG_DBG(P, 512))
                    for (int j = wave; j <= P.nslot; j += kPipeWaves) glds16(image + (size_t)j * 128, lane_off, dst + (unsigned)j * 1024u);
                if (have_last)
                    for (int c = 0; c < kPipeChunks; ++c) flush_chunk(c);
            } else {
                self_made = true;
#ifdef SHG_TIMELINE
                if (P.tl && tid == 0) atomicAdd(reinterpret_cast<unsigned long long*>(P.tl) + ((size_t)ntiles * kWaves * 16), 1ull);       // images not in time
#endif
            }
        }
        if (self_made) {
            if (P.sem_limit > 0) {
                if (tid == 0) legendre_token_acquire(P.sem, P.sem_limit);
                __syncthreads();
            }
            if (!SHG_DBG(P, 2))
                pipe_phase1<NS>(P, panel, P.itemtab2 + (size_t)wave * P.nrec2, P.itemtab2 + (size_t)(wave + kPipeWaves) * P.nrec2, bt, it, lane, have_last, flush_chunk);
            else if (have_last)
                for (int c = 0; c < kPipeChunks; ++c) flush_chunk(c);
        }
        PIPE_STAMP(1);
        // the trig pieces of column tile 0 (and the panel image) were issued before the 8 R stores of the parked images (if there were any)
        if (have_last) wait_vmcnt<kYounger>(); else wait_vmcnt<0>();
        __syncthreads();          // panel and trig buffer 0 complete
        if (self_made && P.sem_limit > 0 && tid == 0) legendre_token_release(P.sem);
        if (FED && tid == 0)      // the ring slot is free again -- or given up for this launch, if its image did not arrive in time
            __hip_atomic_store(P.consumed + tile % P.ring_tiles, fed_ready || tile < P.fed_first ? tile + 1 : -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        PIPE_STAMP(2);

        // ---- phase 2: longitude stage, wave = row tile
        const bool epoch_ok = bt * 4 + rt < P.B && !SHG_DBG(P, 1);
        const int4_s rs = descriptor_of(bt);
        // lane parts of the store offsets: rows fk + 4 reg, column fr of a whole tile (ascending / mirrored images)
        unsigned va[4], vd[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int sl = fk + 4 * r;
            const unsigned ro = (unsigned)grid_row_of(it, sl) * (unsigned)P.nlon * 8u;
            const bool ok = slot_valid_of(it, sl);
            va[r] = ok ? ro + (unsigned)fr * 8u : 0x80000000u;
            vd[r] = ok ? ro + (unsigned)(15 - fr) * 8u : 0x80000000u;
        }
        double4_t z4;
        {
            const double2_t* z = panel + P.nslot * 64 + rt * 16 + fk;         // order 0: rows fk + 4 reg
#pragma unroll
            for (int r = 0; r < 4; ++r) z4[r] = z[4 * r].x;
        }
        if (!SHG_DBG(P, 4)) {
            for (int ct = 0; ct < nct; ++ct) {
                // unit ct: its sums accumulate while the images of unit ct - 1 leave; the trig pieces of unit ct + 1 arrive in the other buffer
                if (ct + 1 < nct) issue_trig(ct + 1);
                pipe_unit<NS, R>(P, Y, (ct & 1) ? tb1 : tb0, prow, z4, epoch_ok && ct > 0, rs, va, vd, ct - 1);
                PIPE_STAMP(3 + min(ct, 4));
                if (epoch_ok && ct > 0) wait_vmcnt<kYounger>(); else wait_vmcnt<0>();
                __syncthreads();          // every wave is done with this unit's trig buffer (and, behind the last unit, with the panel)
            }
        }
        have_last = epoch_ok && !SHG_DBG(P, 4);
        PIPE_STAMP(12);
    }
    // ---- the images of the very last unit
    if (have_last) {
        const int done = (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x;           // tiles of this workgroup - 1
        const int tile_last = (int)blockIdx.x + done * (int)gridDim.x;
        const int bt_l = P.blockmap ? P.blockmap[2 * tile_last] : tile_last % nbt;
        const int it_l = P.blockmap ? P.blockmap[2 * tile_last + 1] : tile_last / nbt;
        const int b_ = min(bt_l * 4 + rt, P.B - 1);
        const unsigned long long g_ = (unsigned long long)(P.G + (size_t)b_ * P.nlat * P.nlon);
        const int4_s rs_l = {(int)(unsigned)g_, (int)(unsigned)((g_ >> 32) & 0xffffu), grid_bytes, 0x00020000};
        unsigned va_l[4], vd_l[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int sl = fk + 4 * r;
            const int row = NS ? (sl < 8 ? it_l * 8 + sl : P.nlat - 1 - (it_l * 8 + sl - 8)) : it_l * 16 + sl;
            const bool ok = (NS ? it_l * 8 + (sl & 7) < P.nh : it_l * 16 + sl < P.nlat) && fr < ncol_last;
            const unsigned ro = (unsigned)row * (unsigned)P.nlon * 8u;
            va_l[r] = ok ? ro + (unsigned)fr * 8u : 0x80000000u;
            vd_l[r] = ok ? ro + (unsigned)(ncol_last - 1 - fr) * 8u : 0x80000000u;
        }
        pipe_flush<R, 0, 1>(Y, rs_l, va_l, vd_l, P.nlon, P.nd, ctl, ncol_last);
    }
    // every LDS-DMA of this wave was waited for at the end of its unit; the stores may still be in flight when the wave ends
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
static int launch_rot(shg_plan* p, bool ns, int mode, const RotParams& P, size_t lds, dim3 grid_dim, hipStream_t stream) {
#define SHG_LAUNCH_PIPE(NS_, FED_)                                                                                                        \
    do {                                                                                                                                  \
        SHG_HIP(hipFuncSetAttribute((const void*)synthesis_pipe_kernel<NS_, R, FED_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((synthesis_pipe_kernel<NS_, R, FED_>), grid_dim, dim3(64 * kPipeWaves), lds, stream, P);                        \
    } while (0)
    if (mode == 2) {
        if (ns) SHG_LAUNCH_PIPE(true, true); else SHG_LAUNCH_PIPE(false, true);
    } else if (mode == 1) {
        if (ns) SHG_LAUNCH_PIPE(true, false); else SHG_LAUNCH_PIPE(false, false);
    } else if (mode == 3) {           // two workgroups of two epochs per CU
        if (ns) {
            SHG_HIP(hipFuncSetAttribute((const void*)synthesis_rot_kernel<true, R, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((synthesis_rot_kernel<true, R, 2>), grid_dim, dim3(256), lds, stream, P);
        } else {
            SHG_HIP(hipFuncSetAttribute((const void*)synthesis_rot_kernel<false, R, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((synthesis_rot_kernel<false, R, 2>), grid_dim, dim3(256), lds, stream, P);
        }
    } else if (ns) {
        SHG_HIP(hipFuncSetAttribute((const void*)synthesis_rot_kernel<true, R, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((synthesis_rot_kernel<true, R, 4>), grid_dim, dim3(64 * kWaves), lds, stream, P);
    } else {
        SHG_HIP(hipFuncSetAttribute((const void*)synthesis_rot_kernel<false, R, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((synthesis_rot_kernel<false, R, 4>), grid_dim, dim3(64 * kWaves), lds, stream, P);
    }
#undef SHG_LAUNCH_PIPE
    return SHG_OK;
}

// LDS of the pipelined kernel: two trig buffers of one column tile each, then the panel
static size_t pipe_lds_bytes(int nslot) { return (size_t)(2 * (nslot / 4) + nslot + 1) * 1024; }

int pipe_applicable(const shg_plan* p) {
    if (!rot_applicable(p)) return 0;
    int nk[kMaxClasses], cnt[kMaxClasses];
    return pipe_lds_bytes(rot_layout(p->rotR, p->N, nk, cnt, nullptr)) <= 160 * 1024 ? 1 : 0;
}

static int synthesis_rot_launch(shg_plan* p, int mode, const double* anm, int B, double* grid, hipStream_t stream) {
    const bool pipe = mode == 1 || mode == 2;
    if (!(pipe ? pipe_applicable(p) : rot_applicable(p))) return fail(SHG_ERR_UNSUPPORTED, "rotation-folded synthesis kernel not applicable to this plan");
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
    if (!p->sem_d) {
        if (hipMalloc((void**)&p->sem_d, 256) != hipSuccess) return fail(SHG_ERR_NOMEM, "token counter allocation failed");
        SHG_HIP(hipMemset(p->sem_d, 0, 256));
    }
    P.sem = p->sem_d;
    P.sem_limit = kLegendreTokens;
    if (kLegendreTokens < 0) {
        int dev = 0, cus = 0;
        SHG_HIP(hipGetDevice(&dev));
        SHG_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        P.sem_limit = std::max(1, cus * -kLegendreTokens / 16);
    }
    if (mode != 0) P.sem_limit = 0;          // (the other kernels: measured without gain)
#ifdef SHG_EXPERIMENT
    if (getenv("SHG_SEM")) P.sem_limit = atoi(getenv("SHG_SEM"));
#endif
    P.itemtab2 = reinterpret_cast<const int2*>(p->itemtab2_d);
    P.nrec2 = p->itemtab2_nrec;
    P.ntrip2 = p->itemtab2_ntrip;
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
    // mode 3: four rings of kRingSlots KB, then the 32-row panel
    const size_t lds = pipe ? pipe_lds_bytes(P.nslot) : mode == 3 ? (size_t)4 * kRingSlots * 1024 + (size_t)(P.nslot + 1) * 512 : rot_lds_bytes(P.nslot);
    unsigned nwg_pipe = (unsigned)(nbt * nit);
    if (pipe) {                                       // persistent workgroups, one per CU (the kernel's LDS admits no second one)
        int dev = 0, cus = 0;
        SHG_HIP(hipGetDevice(&dev));
        SHG_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        nwg_pipe = std::min(nwg_pipe, (unsigned)std::max(cus, 1));
    }
    P.fed_first = 0;
    P.ring = nullptr;
    P.ring_tiles = 0;
    P.produced = P.consumed = nullptr;
    if (mode == 2) {
        // the ring of panel images, the hand-off words, the producer's stream; the words are zeroed before every launch
        constexpr int kRingTiles = 512;
        const size_t image_bytes = (size_t)(P.nslot + 1) * 1024;
        if (p->ring_bytes < image_bytes * kRingTiles) {
            SHG_HIP(hipDeviceSynchronize());
            if (p->ring_d) (void)hipFree(p->ring_d);
            p->ring_d = nullptr;
            p->ring_bytes = 0;
            if (hipMalloc((void**)&p->ring_d, image_bytes * kRingTiles) != hipSuccess) return fail(SHG_ERR_NOMEM, "panel ring allocation failed");
            p->ring_bytes = image_bytes * kRingTiles;
        }
        if (p->ring_image_bytes != image_bytes) {                // (the padding slots of the images are never written: they must read as zero)
            SHG_HIP(hipMemsetAsync(p->ring_d, 0, p->ring_bytes, stream));
            p->ring_image_bytes = image_bytes;
        }
        if (!p->handoff_d && hipMalloc((void**)&p->handoff_d, 2 * kRingTiles * sizeof(int)) != hipSuccess) return fail(SHG_ERR_NOMEM, "hand-off words allocation failed");
        if (!p->side_stream) {
            SHG_HIP(hipStreamCreateWithFlags(&p->side_stream, hipStreamNonBlocking));
            SHG_HIP(hipEventCreateWithFlags(&p->fork_event, hipEventDisableTiming));
            SHG_HIP(hipEventCreateWithFlags(&p->join_event, hipEventDisableTiming));
        }
        P.ring = p->ring_d;
        P.ring_tiles = kRingTiles;
        P.produced = p->handoff_d;
        P.consumed = p->handoff_d + kRingTiles;
        P.fed_first = (int)std::min<unsigned>(nwg_pipe, (unsigned)(nbt * nit));
    }
    // (mode 3: the two halves of tile 8 j + x are the workgroups 16 j + x and 16 j + 8 + x)
    const dim3 grid_dim(pipe ? nwg_pipe : mode == 3 ? (unsigned)(ceil_div(nbt * nit, 8) * 16) : (unsigned)(nbt * nit));
    ProfileScope ps(p, 2, stream);          // (fed pipeline: from the fork to the join, i.e. both kernels)
    if (mode == 2) {
        SHG_HIP(hipMemsetAsync(p->handoff_d, 0, 2 * P.ring_tiles * sizeof(int), stream));
        SHG_HIP(hipEventRecord(p->fork_event, stream));
        SHG_HIP(hipStreamWaitEvent(p->side_stream, p->fork_event, 0));
        const int produced_tiles = nbt * nit - P.fed_first;
        if (produced_tiles > 0) {
            if (ns)
                hipLaunchKernelGGL(legendre_panel_kernel<true>, dim3((unsigned)produced_tiles), dim3(256), 0, p->side_stream, P);
            else
                hipLaunchKernelGGL(legendre_panel_kernel<false>, dim3((unsigned)produced_tiles), dim3(256), 0, p->side_stream, P);
            SHG_HIP(hipGetLastError());
        }
        SHG_HIP(hipEventRecord(p->join_event, p->side_stream));
    }
    switch (R) {
        case 10: rc = launch_rot<10>(p, ns, mode, P, lds, grid_dim, stream); break;
        case 9: rc = launch_rot<9>(p, ns, mode, P, lds, grid_dim, stream); break;
        case 6: rc = launch_rot<6>(p, ns, mode, P, lds, grid_dim, stream); break;
        case 3: rc = launch_rot<3>(p, ns, mode, P, lds, grid_dim, stream); break;
        default: return fail(SHG_ERR_UNSUPPORTED, "rotation-folded synthesis kernel: no kernel for %d rotations", R);
    }
    if (mode == 2) SHG_HIP(hipStreamWaitEvent(stream, p->join_event, 0));

    if (rc) return rc;
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

int synthesis_rot(shg_plan* p, const double* anm, int B, double* grid, hipStream_t stream) { return synthesis_rot_launch(p, 0, anm, B, grid, stream); }
int synthesis_pipe(shg_plan* p, const double* anm, int B, double* grid, hipStream_t stream) { return synthesis_rot_launch(p, 1, anm, B, grid, stream); }
int synthesis_fed(shg_plan* p, const double* anm, int B, double* grid, hipStream_t stream) { return synthesis_rot_launch(p, 2, anm, B, grid, stream); }
int synthesis_rot_halves(shg_plan* p, const double* anm, int B, double* grid, hipStream_t stream) { return synthesis_rot_launch(p, 3, anm, B, grid, stream); }

}  // namespace shg
