// Fused batched synthesis for grids with 4-fold symmetric meridians (every equi-angular grid):
// ONE kernel per batch, nothing but coefficients in and grids out touches HBM.
//
// Block = 4 epochs x 16 parallels (64 panel rows), 8 waves.
//   phase 1 (Legendre stage on MFMA):  for every order m
//        D[(epoch, c/s)][parallel] = sum_n coef[(epoch, c/s)][n] * PK_m[n][parallel]
//     A operand = repacked coefficients (rows 0-3 C_nm of the 4 epochs, rows 4-7 S_nm, rows 8-15 zero),
//     B operand = plan table PK_m[n][i] = kn[i][n] P_nm(theta_i)  (what the reference forms at
//     grates/gravityfield.py:358-362), both read as MFMA fragments straight from L2.
//     The result lands in the LDS panel As[slot(m, c/s)][row = epoch * 16 + parallel].
//   phase 2 (longitude stage on MFMA): wave w owns column tile cb * 8 + w of every block of 8 column tiles
//     (64 panel rows x 16 quarter-columns).  A fragments come from the read-only LDS panel, B fragments
//     (cos/sin table) straight from L2 through a register ring that runs one body (16 MFMAs) ahead; there is
//     no barrier after the panel is complete.  The four longitude images of every quarter-column are formed
//     in registers and stored with 16-byte stores.
#include <algorithm>
#include <cstdlib>
#include <utility>

#include "common.h"

#ifndef SHG_STORE_AUX
#define SHG_STORE_AUX 2          // cache policy bits of the grid stores: 2 = nt (streaming data that is not re-read; the tables
                                // stay in L2: 0.67 ms against 0.72 ms with 0 or 1 = sc0, 0.74 ms with 16 = sc1; 3, 18, 19 like 2)
#endif

namespace shg {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));

constexpr int kPanelStride = 80;    // 64 rows + 16 pad: the 4 k-rows of one fragment read fall on disjoint LDS banks

// ------------------------------------------------------------------------------------------------
// plan-time table  PK[(m, n)][i] = kn[i][n] * P_nm(theta_i)   (order-major packed, parallel fastest)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void pk_table_kernel(int N, int ldlat, const double* __restrict__ ct,
                                                      const double* __restrict__ pmm, const double* __restrict__ knT,
                                                      const double* __restrict__ arec, const double* __restrict__ brec,
                                                      double* __restrict__ pk) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int m = blockIdx.y;
    const int off = order_offset(N, m);
    const double t = ct[i];
    double p1 = pmm[(size_t)m * ldlat + i], p2 = 0.0;
    pk[(size_t)off * ldlat + i] = p1 * knT[(size_t)m * ldlat + i];
    for (int n = m + 1; n <= N; ++n) {
        const int idx = off + n - m;
        const double p = (arec[idx] * t) * p1 - brec[idx] * p2;      // grates/utilities.py:52-54, no contraction
        p2 = p1;
        p1 = p;
        pk[(size_t)idx * ldlat + i] = p * knT[(size_t)n * ldlat + i];
    }
}

// The same table in MFMA-fragment order.  Order m owns ceil((N+1-m)/8) "row octets" (8 degrees = 2 k-steps) starting at
// octet qoff[m]; octet o of parallel tile it is 64 lanes x 2 doubles, lane = fk * 16 + fr holding
//   PK[(m, m + 8 j + 4 s + fk)][16 it + fr]   for s = 0, 1       (zero beyond degree N)
// so that one 16-byte load per lane fetches the B fragments of two k-steps and a wave reads 1 KB contiguous.
__global__ __launch_bounds__(64) void pkf_table_kernel(int N, int ldlat, int nit, int Qtot, const int* __restrict__ qoff,
                                                       const double* __restrict__ ct, const double* __restrict__ pmm,
                                                       const double* __restrict__ knT, const double* __restrict__ arec,
                                                       const double* __restrict__ brec, double* __restrict__ pkf) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= nit * 16) return;
    const int m = blockIdx.y;
    const int off = order_offset(N, m);
    const double t = ct[i];
    double* dst = pkf + (((size_t)(i >> 4) * Qtot + qoff[m]) * 64 + (i & 15)) * 2;      // + octet * 128 + fk * 32 + s
    double p1 = pmm[(size_t)m * ldlat + i], p2 = 0.0;
    dst[0] = p1 * knT[(size_t)m * ldlat + i];
    for (int n = m + 1; n <= N; ++n) {
        const int idx = off + n - m, nl = n - m;
        const double p = (arec[idx] * t) * p1 - brec[idx] * p2;      // grates/utilities.py:52-54, no contraction
        p2 = p1;
        p1 = p;
        dst[(size_t)(nl >> 3) * 128 + (nl & 3) * 32 + ((nl >> 2) & 1)] = p * knT[(size_t)n * ldlat + i];
    }
}

// coefficients of 4 epochs in MFMA-fragment order:  cpk4[bt][octet][fk * 8 + c/s * 4 + epoch][s]  (A operand rows 0-3 =
// C_nm of the 4 epochs, rows 4-7 = S_nm; rows 8-15 are zero and not stored)
// rotR != 0 (rotation-folded kernel, synthesis_rot.hip): S_nm of the orders with 2 (m mod R) > R are stored negated
__global__ __launch_bounds__(256) void pack_coefficients4_kernel(int N, int B, int Qtot, int rotR, const int* __restrict__ qoff,
                                                                  const double* __restrict__ anm, double* __restrict__ cpk4) {
    const int E = (N + 1) * (N + 1);
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int bt = blockIdx.y;
    const int r = e / (N + 1), c = e % (N + 1);
    int m, nl, cs;
    if (c <= r) {
        m = c;
        nl = r - c;
        cs = 0;
    } else {
        m = r + 1;
        nl = c - (r + 1);
        cs = 1;
    }
    double* dst = cpk4 + (((size_t)bt * Qtot + qoff[m] + (nl >> 3)) * 32 + (nl & 3) * 8 + cs * 4) * 2 + ((nl >> 2) & 1);
    const double sg = (rotR && cs && 2 * (m % rotR) > rotR) ? -1.0 : 1.0;
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) dst[bb * 2] = (bt * 4 + bb < B) ? sg * anm[(size_t)(bt * 4 + bb) * E + e] : 0.0;
}

// North-south symmetric grids (colat[nlat-1-i] = pi - colat[i], equal kn rows): P_nm(pi - theta) = (-1)^(n-m) P_nm(theta), so
// the Legendre stage only needs the northern parallels.  A block then covers 8 northern parallels and their 8 mirror
// images; the sums over even and odd n - m are formed side by side in one MFMA tile,
//   A rows 0-7 : coefficients of degrees m + 2 j      (4 epochs x cos/sin),   B columns 0-7 : PK[m + 2 j][8 parallels]
//   A rows 8-15: coefficients of degrees m + 2 j + 1,                          B columns 8-15: PK[m + 2 j + 1][8 parallels]
// (k-step = 4 degree pairs j; the off-diagonal 8 x 8 blocks of the product are discarded), and north / south panel
// rows are E + O / E - O.  Half the MFMAs and half the PK bytes of the plain layout.  Order m owns ceil((N+1-m)/16)
// octets (16 degrees = 2 k-steps); octet = 64 lanes x 2 doubles, lane = (pair & 3) * 16 + parity * 8 + parallel.
__global__ __launch_bounds__(64) void pkf_ns_table_kernel(int N, int nlat, int ldlat, int nh, int nit, int Qtot, const int* __restrict__ qoff,
                                                          const int* __restrict__ badmap, const double* __restrict__ ct,
                                                          const double* __restrict__ pmm, const double* __restrict__ knT,
                                                          const double* __restrict__ arec, const double* __restrict__ brec,
                                                          double* __restrict__ pkf) {
    const int i = blockIdx.x * 64 + threadIdx.x;                    // northern parallel
    if (i >= nh) return;
    // z = 0: northern parallels, tile i / 8.  z = 1: the mirrored parallels of the blocks flagged in badmap get their own
    // tile (nit + rank) in the same lane slots
    int row = i, tile = i >> 3;
    if (blockIdx.z == 1) {
        const int rank = badmap[i >> 3];
        if (rank < 0) return;
        row = nlat - 1 - i;
        tile = nit + rank;
    }
    const int m = blockIdx.y;
    const int off = order_offset(N, m);
    const double t = ct[row];
    double* dst = pkf + (((size_t)tile * Qtot + qoff[m]) * 64 + (i & 7)) * 2;
    double p1 = pmm[(size_t)m * ldlat + row], p2 = 0.0;
    dst[0] = p1 * knT[(size_t)m * ldlat + row];
    for (int n = m + 1; n <= N; ++n) {
        const int idx = off + n - m, nl = n - m;
        const double p = (arec[idx] * t) * p1 - brec[idx] * p2;      // grates/utilities.py:52-54, no contraction
        p2 = p1;
        p1 = p;
        const int j = nl >> 1, kstep = j >> 2;
        dst[(size_t)(kstep >> 1) * 128 + ((j & 3) * 16 + (nl & 1) * 8) * 2 + (kstep & 1)] = p * knT[(size_t)n * ldlat + row];
    }
}

// scatter form of the NS repack (32-row kernel, synthesis_fused32.hip)
__global__ __launch_bounds__(256) void pack_coefficients4_ns_kernel(int N, int B, int Qtot, const int* __restrict__ qoff,
                                                                     const double* __restrict__ anm, double* __restrict__ cpk4) {
    const int E = (N + 1) * (N + 1);
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int bt = blockIdx.y;
    const int r = e / (N + 1), c = e % (N + 1);
    int m, nl, cs;
    if (c <= r) {
        m = c;
        nl = r - c;
        cs = 0;
    } else {
        m = r + 1;
        nl = c - (r + 1);
        cs = 1;
    }
    const int j = nl >> 1, kstep = j >> 2;
    double* dst = cpk4 + (((size_t)bt * Qtot + qoff[m] + (kstep >> 1)) * 64 + (j & 3) * 16 + (nl & 1) * 8 + cs * 4) * 2 + (kstep & 1);
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) dst[bb * 2] = (bt * 4 + bb < B) ? anm[(size_t)(bt * 4 + bb) * E + e] : 0.0;
}

// Gather form of the NS repack: one thread per 16-byte element of the fragment-ordered table (fully coalesced writes, every
// element written, padding included: no zero-fill of the workspace); the two degrees of an element are read from the epoch's
// coefficient triangle, which stays in L2.  octinfo[octet] = order | (octet index inside the order) << 8.
// Workgroups go to the 8 XCDs round robin by their linear index: XCD x takes the epoch tiles x, x + 8, ... with all their octets one after
// the other, so that the coefficient triangles of an epoch tile (4 x 75 KB at d/o 96; every 64-byte sector of their cosine columns is
// shared by eight orders) are fetched into one L2 only.
__global__ __launch_bounds__(256) void pack_coefficients4_ns_gather_kernel(int N, int B, int Qtot, int rotR, int nbt, const int* __restrict__ octinfo,
                                                                            const double* __restrict__ anm, double* __restrict__ cpk4) {
    const int nx = (Qtot * 64 + 255) / 256;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int bt = 8 * (seq / nx) + xcd;
    const int t = (seq % nx) * 256 + threadIdx.x;                  // (octet, lane)
    if (bt >= nbt || t >= Qtot * 64) return;
    const int oct = t >> 6, lane = t & 63;
    const int info = octinfo[oct];
    const int m = info & 255, ol = info >> 8;
    const int fk = lane >> 4, row = lane & 15;
    const int par = row >> 3, cs = (row >> 2) & 1, b = bt * 4 + (row & 3);
    const size_t E = (size_t)(N + 1) * (N + 1);
    double v[2];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
        const int nl = 2 * ((2 * ol + s_) * 4 + fk) + par;         // degree offset n - m
        const int n = m + nl;
        const bool ok = n <= N && b < B && !(cs == 1 && m == 0);
        const size_t e = cs == 0 ? (size_t)n * (N + 1) + m : (size_t)(m - 1) * (N + 1) + n;
        v[s_] = ok ? anm[(size_t)b * E + e] : 0.0;
        if (rotR && cs && 2 * (m % rotR) > rotR) v[s_] = -v[s_];
    }
    *reinterpret_cast<double2*>(cpk4 + ((size_t)bt * Qtot * 64 + t) * 2) = make_double2(v[0], v[1]);
}

// The same repack from an order-major series (filters.hip: om [(N_s + 1)^2 rows][Bpad epochs], N_s >= N): the four epochs of an element
// are 32 consecutive bytes of one row.
__host__ __device__ inline long long om_row_of(int Ns, int s) {
    if (s == 0) return 0;
    const int m = (s + 1) >> 1;
    const long long cos_row = (long long)(Ns + 1) + 2LL * ((long long)(m - 1) * (Ns + 1) - (long long)m * (m - 1) / 2);
    return (s & 1) ? cos_row : cos_row + (Ns + 1 - m);
}
__global__ __launch_bounds__(256) void pack_coefficients4_ns_gather_om_kernel(int N, int B, int Qtot, int rotR, int nbt, const int* __restrict__ octinfo,
                                                                               const double* __restrict__ om, int Ns, int Bpad, double* __restrict__ cpk4) {
    const int nx = (Qtot * 64 + 255) / 256;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int bt = 8 * (seq / nx) + xcd;
    const int t = (seq % nx) * 256 + threadIdx.x;                  // (octet, lane)
    if (bt >= nbt || t >= Qtot * 64) return;
    const int oct = t >> 6, lane = t & 63;
    const int info = octinfo[oct];
    const int m = info & 255, ol = info >> 8;
    const int fk = lane >> 4, row = lane & 15;
    const int par = row >> 3, cs = (row >> 2) & 1, b = bt * 4 + (row & 3);
    const long long slot_row = om_row_of(Ns, m == 0 ? 0 : 2 * m - 1 + cs);
    double v[2];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
        const int nl = 2 * ((2 * ol + s_) * 4 + fk) + par;         // degree offset n - m
        const bool ok = m + nl <= N && b < B && !(cs == 1 && m == 0);
        v[s_] = ok ? om[(size_t)(slot_row + nl) * Bpad + b] : 0.0;
        if (rotR && cs && 2 * (m % rotR) > rotR) v[s_] = -v[s_];
    }
    *reinterpret_cast<double2*>(cpk4 + ((size_t)bt * Qtot * 64 + t) * 2) = make_double2(v[0], v[1]);
}

struct FusedParams {
    int N, nlat, nlon, ldlat, K, ncol, B, nit, Ppk, ncb;   // nit = 16-parallel tiles, ncb = column blocks (8 tiles each)
    int goff[5];              // first K slot of each (cos/sin, m even/odd) group; every group is a multiple of 16 slots
    int gcount[4];            // used slots per group (the rest up to goff[g+1] is zero padding)
    int ns, nh;               // north-south symmetric variant: blocks of 8 northern parallels + mirrors; nh = nlat / 2
    int slot0;                // panel slot of order 0 when it is folded out of the K loop (start value of the accumulators), or -1
#ifdef SHG_EXPERIMENT
    int dbg;                  // experiment switches (SHG_DEBUG): 1 no stores, 2 no Legendre phase, 4 no longitude phase, 8 no longitude MFMAs
#endif
    int Qtot;                 // row octets of the fragment-ordered tables
    const int* qoff;          // [N+2]
    const double* cpk4;       // [nbt][Qtot][32][2]
    const double* pkf;        // [nit][Qtot][64][2]
    const int4* itemtab;      // [8 waves][nrec] work items of the Legendre stage (see build_item_table)
    int nrec, ntrip;
#ifdef SHG_TIMELINE
    unsigned long long* tl;   // instrumented build (make timeline): [workgroups][8 waves][16] timestamps of wall_clock64()
#endif
    const int* blockmap;      // [blocks][2] (epoch tile, parallel tile) of every workgroup, or NULL for the plain order
    const int* badmap;        // NS variant: [nit] -1, or rank of the block among those whose mirrored parallels need their own table
    const double* trig;       // [ncb * 8][K][16]
    double* G;
};

// One work item of phase 1: two row octets (4 k-steps, 16 degrees) of order m starting at octet j0 of that order.
struct LegendreItem {
    int m, j0;
    __device__ bool valid(int N) const { return m <= N; }
    __device__ LegendreItem next(int N, int octet_degrees = 8) const {
        LegendreItem r = {m, j0 + 2};
        if (r.j0 * octet_degrees >= N + 1 - m) {
            r.m = m + 8;
            r.j0 = 0;
        }
        return r;
    }
};

// Instrumented build (`make timeline` -> lib/libshg_timeline.so, read by tools/timeline.py): every wave records when it
// passes the stages of its tile.  Compiled out of the product library.
#ifdef SHG_TIMELINE
#define SHG_STAMP(ev)                                                                                         \
    do {                                                                                                      \
        if (P.tl && lane == 0) P.tl[((size_t)blockIdx.x * 8 + wave) * 16 + (ev)] = wall_clock64();            \
    } while (0)
#else
#define SHG_STAMP(ev)
#endif

template <bool NS>
__global__ __launch_bounds__(512) void synthesis_fused_kernel(FusedParams P) {
    extern __shared__ double As[];                     // panel [K][kPanelStride]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave-uniform: keeps the order loops on the scalar unit
    const int nbt = (P.B + 3) >> 2;
    // XCD-aware block order (blockmap, built on the host): workgroup b runs on XCD b % 8; every XCD gets a contiguous
    // range of epoch tiles (their coefficient slabs, ~2.6 MB, then stay in its 4 MB L2) and walks it parallel-tile major,
    // so that the PK slab of a parallel tile is fetched once per XCD and reused by its epoch tiles back to back
    const int bt = P.blockmap ? P.blockmap[2 * blockIdx.x] : (int)(blockIdx.x % nbt);
    const int it = P.blockmap ? P.blockmap[2 * blockIdx.x + 1] : (int)(blockIdx.x / nbt);
    const int i0 = it * 16;                             // plain layout: first parallel of the block
    const int i0n = it * 8;                             // NS layout: first northern parallel of the block
    const int fr = lane & 15, fk = lane >> 4;
    SHG_STAMP(0);

    // ---- zero the padding slots of the panel
    for (int g = 0; g < 4; ++g)
        for (int s = P.goff[g] + P.gcount[g]; s < P.goff[g + 1]; ++s)
            if (tid < 64) As[s * kPanelStride + tid] = 0.0;

    // cos/sin fragments (B operand of the longitude stage) of the first body: fetched ahead of the Legendre stage, so that the
    // longitude stage starts without an exposed L2 round trip
    const double* tbase = P.trig + ((size_t)wave * P.K + fk) * 16 + fr;      // + cb * cb_stride + body * 256 + u * 64
    double ring0 = tbase[0], ring1 = tbase[64], ring2 = tbase[128], ring3 = tbase[192];      // B fragments of the current body

    // ---- phase 1: Legendre stage.  Orders are distributed over the 8 waves; items of 4 k-steps are double
    //      buffered in two named register sets so that the fragments of item t+1 are in flight while item t runs.
    if (!SHG_DBG(P, 2)) {
        // plain layout: octet = 8 degrees, A rows 8-15 are zero (not stored);  NS layout: octet = 16 degrees, all 16 rows used
        constexpr int ASTRIDE = NS ? 128 : 64;                        // doubles per octet of the coefficient table
        // Bookkeeping on the scalar unit, as in synthesis_rot.hip (see there for the measurement): records through scalar loads,
        // operand loads "scalar base + fixed lane offset", uniform branches on the record flags, the accumulators of an order
        // started from the MFMA's constant-zero operand.
        typedef int int4_v __attribute__((ext_vector_type(4)));
        typedef const int4_v __attribute__((address_space(4))) crec_t;
        auto ld16 = [](__amdgpu_buffer_rsrc_t table, unsigned voff, unsigned soff) {      // descriptor (scalar) + lane offset + octet offset (scalar)
            return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(table, voff, soff, 0));
        };
        const int bad = NS ? P.badmap[it] : -1;                       // block-uniform
        __amdgpu_buffer_rsrc_t pku = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(P.pkf + (size_t)it * P.Qtot * 128), 0, 0xffffffffu, 0x00020000);
        const __amdgpu_buffer_rsrc_t cfu =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(NS ? P.cpk4 + (size_t)bt * P.Qtot * 128 : P.cpk4 + (size_t)bt * P.Qtot * 64), 0, 0xffffffffu, 0x00020000);
        const unsigned pk_voff = (unsigned)lane * 16u;
        const unsigned cf_voff = NS ? (unsigned)lane * 16u : (unsigned)(fk * 8 + (fr & 7)) * 16u;
        int mode = NS && bad >= 0 ? 1 : 0;
        int prow = lane;                                              // panel row written by this lane
        const bool arow = NS || fr < 8;
        double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        const double4_t zero4 = {0.0, 0.0, 0.0, 0.0};
        double sgm = (mode == 0 && fr >= 8) ? -1.0 : 1.0;                 // sign of the lane's own part in the north / south combination
        bool fresh = true;                                            // uniform: the next MFMA pair opens an order

        // 16-byte fragment loads: A and B of two k-steps per load, 1 KB (B) / 512 B or 1 KB (A) contiguous per wave
#define SHG_P1_ISSUE(rec, ALO, AHI, BLO, BHI)                                                \
    do {                                                                                     \
        ALO = ld16(cfu, cf_voff, (unsigned)(rec).x * (ASTRIDE * 8u));                        \
        BLO = ld16(pku, pk_voff, (unsigned)(rec).x * 1024u);                                 \
        AHI = ld16(cfu, cf_voff, (unsigned)(rec).y * (ASTRIDE * 8u));                        \
        BHI = ld16(pku, pk_voff, (unsigned)(rec).y * 1024u);                                 \
    } while (0)

#define SHG_P1_CONSUME(rec, ALO, AHI, BLO, BHI)                                                                     \
    do {                                                                                                            \
        if ((rec).w & 1) {                                  /* item valid (uniform) */                              \
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
        if ((rec).w & 2) {                                  /* second octet present (uniform) */                    \
            const double ax_ = NS ? AHI.x : (arow ? AHI.x : 0.0), ay_ = NS ? AHI.y : (arow ? AHI.y : 0.0);          \
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ax_, BHI.x, acc0, 0, 0, 0);                                 \
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ay_, BHI.y, acc1, 0, 0, 0);                                 \
        }                                                                                                           \
        if ((rec).w & 4) {                                  /* last item of an order */                             \
            /* C/D layout: row = (lane >> 4) + 4 reg, col = lane & 15: reg 0 = cosine part of epoch (lane >> 4),  */ \
            /* reg 1 = sine part; panel row = epoch * 16 + parallel slot = lane.                                  */ \
            /* NS: regs 0, 1 = even-degree sums E (valid in columns 0-7), regs 2, 3 = odd-degree sums O (valid in */ \
            /* columns 8-15) of the same 8 parallels: lanes fr and fr + 8 exchange them; slots 0-7 get E + O      */ \
            /* (northern parallels), slots 8-15 E - O (their mirror images)                                       */ \
            /* mode 0: both hemispheres from the northern table; mode 1 / 2 (blocks near the poles): northern / mirrored */ \
            /* parallels from their own tables, E + O each, written by the lanes fr < 8                            */ \
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
            if (!NS || mode == 0 || fr < 8) {                                                                       \
                As[((rec).z & 0xFFFF) * kPanelStride + prow] = vc_;                   /* cosine slot */            \
                if ((rec).z >> 16) As[(((rec).z >> 16) - 1) * kPanelStride + prow] = vs_;   /* sine slot (m >= 1) */ \
            }                                                                                                       \
            fresh = true;                                                                                           \
        }                                                                                                           \
    } while (0)

        double2 xal = {0, 0}, xah = {0, 0}, xbl = {0, 0}, xbh = {0, 0};     // register sets X, Y, Z, W
        double2 yal = {0, 0}, yah = {0, 0}, ybl = {0, 0}, ybh = {0, 0};
        double2 zal = {0, 0}, zah = {0, 0}, zbl = {0, 0}, zbh = {0, 0};
        double2 wal = {0, 0}, wah = {0, 0}, wbl = {0, 0}, wbh = {0, 0};
        // Counted loop, four items per trip, three items in flight (with 16-byte fragment loads and the north-south layout
        // the stage is bound by the L2 latency of a wave's own item chain).  Loads are issued unconditionally (an
        // exhausted sequence re-reads a valid item and its MFMAs see a zero A operand): no branch around loads and a single
        // loop exit keep the compiler's vmcnt bookkeeping exact.
        // The work items of a wave (orders wave, wave + 8, ...; two octets each) are the same for every workgroup: they come
        // as records {first octet, second octet, panel slots, flags} from a table built with the plan, fetched four records
        // per trip with scalar loads one trip ahead.  Counted loop, four items per trip, three items in flight; exhausted
        // sequences are padded with records that re-read a valid octet with all flags clear (zero A operand): no branch
        // around loads and a single loop exit keep the compiler's vmcnt bookkeeping exact.
        crec_t* recs = reinterpret_cast<crec_t*>(reinterpret_cast<unsigned long long>(P.itemtab + (size_t)wave * P.nrec));
        for (int pass = 0; pass < (mode == 0 ? 1 : 2); ++pass) {
            if (pass == 1) {                                          // mirrored parallels of a polar block: their own table
                mode = 2;
                sgm = 1.0;
                prow = lane + 8;
                pku = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(P.pkf + (size_t)(P.nit + bad) * P.Qtot * 128), 0, 0xffffffffu, 0x00020000);
            }
            int4_v c0 = recs[0], c1 = recs[1], c2 = recs[2];
            int4_v n0 = recs[3], n1 = recs[4], n2 = recs[5], n3 = recs[6];
            SHG_P1_ISSUE(c0, xal, xah, xbl, xbh);
            SHG_P1_ISSUE(c1, yal, yah, ybl, ybh);
            SHG_P1_ISSUE(c2, zal, zah, zbl, zbh);
            for (int trip = 0; trip < P.ntrip; ++trip) {
                const int4_v a3 = n0, a4 = n1, a5 = n2, a6 = n3;
                crec_t* nr = recs + 4 * trip + 7;
                n0 = nr[0];
                n1 = nr[1];
                n2 = nr[2];
                n3 = nr[3];
                SHG_P1_ISSUE(a3, wal, wah, wbl, wbh);
                SHG_P1_CONSUME(c0, xal, xah, xbl, xbh);
                SHG_P1_ISSUE(a4, xal, xah, xbl, xbh);
                SHG_P1_CONSUME(c1, yal, yah, ybl, ybh);
                SHG_P1_ISSUE(a5, yal, yah, ybl, ybh);
                SHG_P1_CONSUME(c2, zal, zah, zbl, zbh);
                SHG_P1_ISSUE(a6, zal, zah, zbl, zbh);
                SHG_P1_CONSUME(a3, wal, wah, wbl, wbh);
                c0 = a4;
                c1 = a5;
                c2 = a6;
            }
        }
#undef SHG_P1_ISSUE
#undef SHG_P1_CONSUME
    }
    SHG_STAMP(1);
    __syncthreads();          // panel complete; from here on it is read-only and the waves run independently
    SHG_STAMP(2);

    // grid row of panel row slot s (0..15) of this block.  NS: slots 0-7 = northern parallels, 8-15 = their mirror images
    auto grid_row = [&](int s) { return NS ? (s < 8 ? i0n + s : P.nlat - 1 - (i0n + s - 8)) : i0 + s; };
    auto slot_valid = [&](int s) { return NS ? i0n + (s & 7) < P.nh : i0 + s < P.nlat; };

    // ---- phase 2: longitude stage
    const int nbody = P.K >> 4;                        // bodies of 4 k-steps; every group is a whole number of bodies
    const bool pair_stores = (P.ncol & 1) == 0;
    const int grid_bytes = P.nlat * P.nlon * 8;        // one epoch's grid; < 2^31 (checked on the host)
    const int par = fr & 1;
    const size_t cb_stride = (size_t)8 * P.K * 16;
    for (int ccb = 0; ccb < P.ncb && !SHG_DBG(P, 4); ++ccb) {
        double4_t acc[4][4];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[gg][rt] = (double4_t){0.0, 0.0, 0.0, 0.0};
        if (P.slot0 >= 0) {
            // order 0 does not depend on the longitude: its panel entry is the start value of the cosine / even-order sums
            // (C/D layout: row = fk + 4 reg of row tile rt, the same value in all 16 columns)
            const double* z = As + (size_t)P.slot0 * kPanelStride + fk;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[0][rt][r] = z[rt * 16 + 4 * r];
        }
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            for (int body = P.goff[gg] >> 4; body < (P.goff[gg + 1] >> 4) && !SHG_DBG(P, 8); ++body) {
                // next body of the flat (column block, body) sequence; clamped at the very end
                int nb_ = body + 1, ncb_ = ccb;
                if (nb_ == nbody) {
                    nb_ = 0;
                    ncb_ = min(ccb + 1, P.ncb - 1);
                }
                const double* tn = tbase + (size_t)ncb_ * cb_stride + (size_t)nb_ * 256;
                // B fragments of the NEXT body: issued now, consumed a whole body (16 MFMAs) later
                const double nx0 = tn[0], nx1 = tn[64], nx2 = tn[128], nx3 = tn[192];
                const double* ap = As + (size_t)(body * 16 + fk) * kPanelStride + fr;
                double a0[4], a1[4];
#define SHG_READ_A(dst, u)                                                     \
    _Pragma("unroll") for (int rt = 0; rt < 4; ++rt) dst[rt] = ap[(u) * 4 * kPanelStride + rt * 16]
#define SHG_MFMA4(src, ring)                                                   \
    _Pragma("unroll") for (int rt = 0; rt < 4; ++rt) acc[gg][rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(src[rt], ring, acc[gg][rt], 0, 0, 0)
                // hand-scheduled: the A fragments of k-step u+1 are read while k-step u runs
                SHG_READ_A(a0, 0);
                SHG_READ_A(a1, 1);
                __builtin_amdgcn_sched_barrier(0);
                SHG_MFMA4(a0, ring0);
                __builtin_amdgcn_sched_barrier(0);
                SHG_READ_A(a0, 2);
                SHG_MFMA4(a1, ring1);
                __builtin_amdgcn_sched_barrier(0);
                SHG_READ_A(a1, 3);
                SHG_MFMA4(a0, ring2);
                __builtin_amdgcn_sched_barrier(0);
                SHG_MFMA4(a1, ring3);
                __builtin_amdgcn_sched_barrier(0);
                ring0 = nx0;
                ring1 = nx1;
                ring2 = nx2;
                ring3 = nx3;
#undef SHG_READ_A
#undef SHG_MFMA4
            }
        }
        // epilogue of this column block: four longitude images per quarter-column
        //   lon_j: E + O,  -lon_j (column nlon-1-j): E - O,  -pi - lon_j (nlon/2-1-j): Ee - Eo - Oe + Oo,
        //   lon_j + pi (nlon/2+j): Ee - Eo + Oe - Oo
        SHG_STAMP(3 + 2 * ccb);
        const int jt = (ccb * 8 + wave) * 16;
        if (jt >= P.ncol) continue;                               // uniform per wave (padding tile)
        // 16-byte store offsets of this lane inside one epoch's grid, formed once per column block: rows of the two panel slots
        // it owns after the neighbour swap, columns of the four longitude images.  0x80000000 marks a lane outside the grid:
        // any sum with a column offset stays beyond the buffer size (< 2^31 bytes, checked on the host).
        unsigned pair_off_a = 0x80000000u, pair_off_b = 0x80000000u, pair_col[4] = {0, 0, 0, 0};
        if (pair_stores) {
            const int jc = jt + (fr & ~1);
            const int sa = fk + (par ? 8 : 0), sb = sa + 4;             // panel row slots of this lane
            if (jc < P.ncol && slot_valid(sa)) pair_off_a = (unsigned)grid_row(sa) * (unsigned)P.nlon * 8u;
            if (jc < P.ncol && slot_valid(sb)) pair_off_b = (unsigned)grid_row(sb) * (unsigned)P.nlon * 8u;
            pair_col[0] = (unsigned)jc * 8u;
            pair_col[1] = (unsigned)(P.nlon - 2 - jc) * 8u;
            pair_col[2] = (unsigned)(P.nlon / 2 - 2 - jc) * 8u;
            pair_col[3] = (unsigned)(P.nlon / 2 + jc) * 8u;
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int b = bt * 4 + rt;
            if (b >= P.B) continue;
            double img[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double ee = acc[0][rt][r], eo = acc[1][rt][r], oe = acc[2][rt][r], oo = acc[3][rt][r];
                const double s1 = ee + eo, s2 = ee - eo, d1 = oe + oo, d2 = oe - oo;
                img[0][r] = s1 + d1;
                img[1][r] = s1 - d1;
                img[2][r] = s2 - d2;
                img[3][r] = s2 + d2;
            }
            if (SHG_DBG(P, 1)) {
                if (img[0][0] != 1.2345e-300) continue;
            }
            if (pair_stores) {
                // lanes (2q, 2q+1) hold adjacent columns: swap halves so that every lane owns two rows x two
                // adjacent columns and stores 16 bytes; one store instruction then writes 8 rows x one full 128-byte line
                // (a transposed product with 4 adjacent columns per lane needs no exchange but writes 16 rows x 64 bytes per
                // instruction and measured 12 % slower in round 1).
                // Branch-free buffer stores: lanes outside the grid carry an offset beyond the buffer and are dropped.
                const __amdgpu_buffer_rsrc_t rsrc =
                    __builtin_amdgcn_make_buffer_rsrc(P.G + (size_t)b * P.nlat * P.nlon, 0, grid_bytes, 0x00020000);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    double a_lo, a_hi, b_lo, b_hi;
                    pair_exchange(img[t][0], img[t][2], 0xAAAAAAAAAAAAAAAAull, a_lo, a_hi);
                    pair_exchange(img[t][1], img[t][3], 0xAAAAAAAAAAAAAAAAull, b_lo, b_hi);
                    const bool ascending = t == 0 || t == 3;
                    const double2_t va = ascending ? (double2_t){a_lo, a_hi} : (double2_t){a_hi, a_lo};
                    const double2_t vb = ascending ? (double2_t){b_lo, b_hi} : (double2_t){b_hi, b_lo};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, va), rsrc, pair_off_a + pair_col[t], 0, SHG_STORE_AUX);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, vb), rsrc, pair_off_b + pair_col[t], 0, SHG_STORE_AUX);
                }
            } else {
                const int j = jt + fr;
                if (j >= P.ncol) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (!slot_valid(fk + 4 * r)) continue;
                    const int i = grid_row(fk + 4 * r);
                    double* row = P.G + ((size_t)b * P.nlat + i) * P.nlon;
                    row[j] = img[0][r];
                    row[P.nlon - 1 - j] = img[1][r];
                    row[P.nlon / 2 - 1 - j] = img[2][r];
                    row[P.nlon / 2 + j] = img[3][r];
                }
            }
        }
        SHG_STAMP(4 + 2 * ccb);
    }
    SHG_STAMP(12);
}

static size_t fused_lds_bytes(int K) { return (size_t)K * kPanelStride * sizeof(double); }

// non-zero when the fused kernel applies: 4-fold symmetric meridians and a panel that fits the 160 KiB LDS
int fused_chunk_for(const shg_plan* p) {
    if (!p->sym4 || (p->K & 15)) return 0;
    if ((long long)p->nlat * p->nlon * 8 >= (1LL << 31)) return 0;          // 32-bit store offsets inside one epoch's grid
    return fused_lds_bytes(p->K) <= 160 * 1024 ? 16 : 0;
}

int build_pk_table(shg_plan* p, hipStream_t stream) {
    if (p->pk) return SHG_OK;
    const size_t n = ((size_t)packed_count(p->N) + 4) * p->ldlat;
    if (hipMalloc((void**)&p->pk, n * sizeof(double)) != hipSuccess) return fail(SHG_ERR_NOMEM, "PK table allocation failed (%zu doubles)", n);
    SHG_HIP(hipMemsetAsync(p->pk, 0, n * sizeof(double), stream));
    hipLaunchKernelGGL(pk_table_kernel, dim3(p->ldlat / 64, p->N + 1), dim3(64), 0, stream, p->N, p->ldlat, p->ct, p->pmm, p->knT,
                       p->arec, p->brec, p->pk);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

// fragment-ordered table of the fused kernel (and the octet offsets both fragment-ordered tables share)
static int build_item_table(shg_plan* p, int od, const std::vector<int>& qoff, int rotR, hipStream_t stream);
int rot_kernel_waves();          // synthesis_rot.hip: waves per workgroup of the rotation-folded kernel

// rotR != 0: the work items carry the panel slots of the rotation-folded kernel (synthesis_rot.hip) instead of those of the 4-fold one
int build_pkf_table(shg_plan* p, bool ns, int rotR, hipStream_t stream) {
    const int variant = ns ? 2 : 1;
    if (p->pkf && p->pkf_variant == variant && p->itemtab_rot == rotR) return SHG_OK;
    if (p->pkf) {                                       // the other layout was built before (explicit path switch)
        SHG_HIP(hipDeviceSynchronize());
        (void)hipFree(p->pkf);
        p->pkf = nullptr;
    }
    const int N = p->N;
    const int od = ns ? 16 : 8;                         // degrees per octet
    const int nit = ns ? ceil_div(p->nlat / 2, 8) : ceil_div(p->nlat, 16);
    std::vector<int> qoff(N + 2);
    int q = 0;
    for (int m = 0; m <= N; ++m) {
        qoff[m] = q;
        q += (N + 1 - m + od - 1) / od;
    }
    qoff[N + 1] = q;
    p->Qtot = q;
    if (!p->qoff && hipMalloc((void**)&p->qoff, qoff.size() * sizeof(int)) != hipSuccess) return fail(SHG_ERR_NOMEM, "octet table allocation failed");
    SHG_HIP(hipMemcpy(p->qoff, qoff.data(), qoff.size() * sizeof(int), hipMemcpyHostToDevice));
    const size_t n = (size_t)(nit + (ns ? p->ns_nbad : 0)) * q * 128;
    if (hipMalloc((void**)&p->pkf, n * sizeof(double)) != hipSuccess) return fail(SHG_ERR_NOMEM, "PK table allocation failed (%zu doubles)", n);
    SHG_HIP(hipMemsetAsync(p->pkf, 0, n * sizeof(double), stream));
    if (ns) {
        if (!p->badmap_d) {
            if (hipMalloc((void**)&p->badmap_d, p->ns_badmap.size() * sizeof(int)) != hipSuccess) return fail(SHG_ERR_NOMEM, "block map allocation failed");
            SHG_HIP(hipMemcpy(p->badmap_d, p->ns_badmap.data(), p->ns_badmap.size() * sizeof(int), hipMemcpyHostToDevice));
        }
        hipLaunchKernelGGL(pkf_ns_table_kernel, dim3(ceil_div(p->nlat / 2, 64), N + 1, 2), dim3(64), 0, stream, N, p->nlat, p->ldlat, p->nlat / 2, nit, q,
                           p->qoff, p->badmap_d, p->ct, p->pmm, p->knT, p->arec, p->brec, p->pkf);
    }
    else
        hipLaunchKernelGGL(pkf_table_kernel, dim3(p->ldlat / 64, N + 1), dim3(64), 0, stream, N, p->ldlat, nit, q, p->qoff, p->ct, p->pmm,
                           p->knT, p->arec, p->brec, p->pkf);
    SHG_HIP(hipGetLastError());
    const int rc_items = build_item_table(p, od, qoff, rotR, stream);
    if (rc_items) return rc_items;
    p->pkf_variant = variant;
    return SHG_OK;
}

// Work items of the Legendre stage per wave (8 waves, 12 in the rotation-folded kernel; the orders are dealt to the waves longest first, each to the wave
// with the fewest items so far): records
//   x, y = first / second octet of the item in the fragment-ordered tables (y = x when the order has no second octet left)
//   z    = panel slot of the cosine part | (panel slot of the sine part + 1) << 16   (0 in the upper half: order 0)
//   w    = bit 0 item valid, bit 1 second octet valid, bit 2 last item of its order
// padded per wave to 4 * ntrip + 8 records (the kernel runs ntrip trips of four items and prefetches one trip ahead).
static int build_item_table(shg_plan* p, int od, const std::vector<int>& qoff, int rotR, hipStream_t stream) {
    const int N = p->N;
    std::vector<int> slot16;
    if (rotR) {
        int nk[kRotMaxClasses], cn[kRotMaxClasses];
        rot_layout(rotR, N, nk, cn, &slot16);
    }
    const int nw = rotR ? rot_kernel_waves() : 8;        // waves that share the orders of a tile
    std::vector<std::vector<int>> rec(nw);
    size_t longest = 0;
    for (int m = 0; m <= N; ++m) {                       // orders by decreasing length, each to the wave with the fewest items so far
        {
            int w = 0;
            for (int v = 1; v < nw; ++v)
                if (rec[v].size() < rec[w].size()) w = v;
            const int cnt = N + 1 - m, q = (cnt + od - 1) / od;
            // panel slots (+1 for the sine part, 0 = none).  With order 0 folded out of the K loop (fold0) it sits behind the groups.
            const int* go = p->fold0 ? p->goff_f : p->goff;
            const int even_shift = p->fold0 ? 1 : 0;
            const int slot_c = m == 0 && p->fold0 ? p->K_f : go[m & 1] + (m >> 1) - ((m & 1) ? 0 : even_shift);
            const int slot_s = m >= 1 ? go[2 + (m & 1)] + ((m & 1) ? (m >> 1) : (m >> 1) - 1) + 1 : 0;
            const int zrec = rotR ? slot16[m] : (slot_c | (slot_s << 16));
            for (int j0 = 0; j0 < q; j0 += 2) {
                const int o0 = qoff[m] + j0, o1 = o0 + (j0 + 1 < q ? 1 : 0);
                const int flags = 1 | ((j0 + 1) * od < cnt ? 2 : 0) | (j0 + 2 >= q ? 4 : 0);
                rec[w].insert(rec[w].end(), {o0, o1, zrec, flags});
            }
        }
    }
    for (int w = 0; w < nw; ++w) longest = std::max(longest, rec[w].size() / 4);
    // octet -> (order, octet inside the order) for the gather repack
    std::vector<int> octinfo((size_t)qoff[N + 1], 0);
    for (int m = 0; m <= N; ++m)
        for (int o = qoff[m]; o < qoff[m + 1]; ++o) octinfo[o] = m | ((o - qoff[m]) << 8);
    if (p->octinfo_d) {
        SHG_HIP(hipDeviceSynchronize());
        (void)hipFree(p->octinfo_d);
        p->octinfo_d = nullptr;
    }
    if (hipMalloc((void**)&p->octinfo_d, std::max<size_t>(octinfo.size(), 1) * sizeof(int)) != hipSuccess) return fail(SHG_ERR_NOMEM, "octet table allocation failed");
    SHG_HIP(hipMemcpy(p->octinfo_d, octinfo.data(), octinfo.size() * sizeof(int), hipMemcpyHostToDevice));
    const int ntrip = (int)((longest + 3) / 4), nrec = 4 * ntrip + 8;
    std::vector<int> table((size_t)nw * nrec * 4, 0);
    for (int w = 0; w < nw; ++w) {
        const int pad = rec[w].empty() ? 0 : rec[w][0];              // a valid octet for the padding records
        for (int t = 0; t < nrec; ++t)
            for (int c = 0; c < 4; ++c) {
                const size_t src = (size_t)t * 4 + c;
                table[((size_t)w * nrec + t) * 4 + c] = src < rec[w].size() ? rec[w][src] : (c < 2 ? pad : 0);
            }
    }
    if (p->itemtab_d) {
        SHG_HIP(hipDeviceSynchronize());
        (void)hipFree(p->itemtab_d);
        p->itemtab_d = nullptr;
    }
    if (hipMalloc((void**)&p->itemtab_d, table.size() * sizeof(int)) != hipSuccess) return fail(SHG_ERR_NOMEM, "work item table allocation failed");
    SHG_HIP(hipMemcpyAsync(p->itemtab_d, table.data(), table.size() * sizeof(int), hipMemcpyHostToDevice, stream));
    SHG_HIP(hipStreamSynchronize(stream));                             // the host vector goes out of scope
    p->itemtab_nrec = nrec;
    p->itemtab_rot = rotR;
    p->itemtab_ntrip = ntrip;
    return SHG_OK;
}

// (epoch tile, parallel tile) of every workgroup in XCD-aware order, cached in the plan per (epoch tiles, parallel tiles)
int build_blockmap(shg_plan* p, int nbt, int nit, hipStream_t stream) {
    if (p->blockmap_d && p->blockmap_nbt == nbt && p->blockmap_nit == nit) return SHG_OK;
    constexpr int XCDS = 8;
    const int total = nbt * nit;
    std::vector<int> map((size_t)total * 2);
    // XCD k owns the linear range [k total / 8, (k + 1) total / 8) of s = bt * nit + it and serves it parallel-tile major
    std::vector<std::vector<int>> per_xcd(XCDS);
    for (int k = 0; k < XCDS; ++k) {
        const long long s0 = (long long)k * total / XCDS, s1 = (long long)(k + 1) * total / XCDS;
        std::vector<std::pair<int, int>> items;            // (it, bt)
        for (long long s_ = s0; s_ < s1; ++s_) items.emplace_back((int)(s_ % nit), (int)(s_ / nit));
        std::sort(items.begin(), items.end());
        for (auto& e : items) {
            per_xcd[k].push_back(e.second);
            per_xcd[k].push_back(e.first);
        }
    }
    // workgroup b -> XCD b % 8, its (b / 8)-th item; XCDs with fewer items than others take the leftovers in order
    std::vector<size_t> next(XCDS, 0);
    for (int b = 0; b < total; ++b) {
        int k = b % XCDS;
        for (int tries = 0; tries < XCDS && next[k] >= per_xcd[k].size(); ++tries) k = (k + 1) % XCDS;
        map[2 * (size_t)b] = per_xcd[k][next[k]];
        map[2 * (size_t)b + 1] = per_xcd[k][next[k] + 1];
        next[k] += 2;
    }
    if (p->blockmap_d) {
        SHG_HIP(hipStreamSynchronize(stream));
        (void)hipFree(p->blockmap_d);
        p->blockmap_d = nullptr;
    }
    if (hipMalloc((void**)&p->blockmap_d, map.size() * sizeof(int)) != hipSuccess) return fail(SHG_ERR_NOMEM, "block map allocation failed");
    SHG_HIP(hipMemcpy(p->blockmap_d, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice));
    p->blockmap_nbt = nbt;
    p->blockmap_nit = nit;
    return SHG_OK;
}

// Repack of the coefficient batch into MFMA-fragment order (workspace p->cpk4: [nbt][Qtot][32][2], NS: [nbt][Qtot][64][2]);
// rotR: sign convention of the rotation-folded kernel (0 = none).
int pack_coefficients_fused(shg_plan* p, bool ns, int rotR, const double* anm, int B, hipStream_t stream) {
    const int nbt = ceil_div(B, 4);
    const int variant = ns ? 4 : 2;
    const size_t need = (size_t)nbt * p->Qtot * (ns ? 128 : 64);
    if (need > p->cpk4_size) {
        if (p->cpk4) {
            SHG_HIP(hipStreamSynchronize(stream));
            (void)hipFree(p->cpk4);
            p->cpk4 = nullptr;
            p->cpk4_size = 0;
            p->cpk4_zeroed = 0;
        }
        if (hipMalloc((void**)&p->cpk4, need * sizeof(double)) != hipSuccess) return fail(SHG_ERR_NOMEM, "coefficient workspace allocation failed");
        p->cpk4_size = need;
    }
    // sine slots of order 0 and the padding rows of the octets are never written by the scatter kernel and must read as
    // zero (layouts differ per variant)
    if (need > 0 && (p->cpk4_variant != variant || p->cpk4_zeroed < need)) {
        SHG_HIP(hipMemsetAsync(p->cpk4, 0, p->cpk4_size * sizeof(double), stream));
        p->cpk4_variant = variant;
        p->cpk4_zeroed = p->cpk4_size;
    }
    const int E = (p->N + 1) * (p->N + 1);
    ProfileScope ps(p, 0, stream);
    if (p->om_src) {
        if (!ns) return fail(SHG_ERR_UNSUPPORTED, "synthesis from an order-major series needs parallels symmetric about the equator (the gather repack)");
        hipLaunchKernelGGL(pack_coefficients4_ns_gather_om_kernel, dim3((unsigned)(8 * ceil_div(nbt, 8) * ceil_div(p->Qtot * 64, 256))), dim3(256), 0, stream, p->N, B,
                           p->Qtot, rotR, nbt, p->octinfo_d, p->om_src, p->om_N, p->om_Bpad, p->cpk4);
    } else if (ns)
        hipLaunchKernelGGL(pack_coefficients4_ns_gather_kernel, dim3((unsigned)(8 * ceil_div(nbt, 8) * ceil_div(p->Qtot * 64, 256))), dim3(256), 0, stream, p->N, B,
                           p->Qtot, rotR, nbt, p->octinfo_d, anm, p->cpk4);
    else
        hipLaunchKernelGGL(pack_coefficients4_kernel, dim3(ceil_div(E, 256), nbt), dim3(256), 0, stream, p->N, B, p->Qtot, rotR, p->qoff, anm,
                           p->cpk4);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

int synthesis_fused(shg_plan* p, const double* anm, int B, double* grid, hipStream_t stream) {
    if (fused_chunk_for(p) == 0) return fail(SHG_ERR_UNSUPPORTED, "fused synthesis not applicable to this plan");
    const bool ns = p->sym_ns;                      // north-south symmetric variant where the parallels allow it
    int rc = build_pkf_table(p, ns, 0, stream);
    if (rc) return rc;
    const int nbt = ceil_div(B, 4);
    const int nit = ns ? ceil_div(p->nlat / 2, 8) : ceil_div(p->nlat, 16);
    rc = pack_coefficients_fused(p, ns, 0, anm, B, stream);
    if (rc) return rc;
    FusedParams P;
    P.N = p->N;
    P.nlat = p->nlat;
    P.nlon = p->nlon;
    P.ldlat = p->ldlat;
    const bool fold = p->fold0;
    P.K = fold ? p->K_f : p->K;
    P.slot0 = fold ? p->K_f : -1;
    P.ncol = p->ncol;
    P.B = B;
    P.nit = nit;
    P.ns = ns ? 1 : 0;
    P.nh = p->nlat / 2;
    P.Ppk = packed_count(p->N);
    P.ncb = ceil_div(p->ncoltiles, 8);
    for (int g = 0; g < 5; ++g) P.goff[g] = fold ? p->goff_f[g] : p->goff[g];
    const int N = p->N;
    const int cnt[4] = {N / 2 + (fold ? 0 : 1), (N + 1) / 2, N / 2, (N + 1) / 2};
    for (int g = 0; g < 4; ++g) P.gcount[g] = cnt[g];
#ifdef SHG_EXPERIMENT
    P.dbg = experiment_switches();
#endif
    P.Qtot = p->Qtot;
    P.qoff = p->qoff;
    P.cpk4 = p->cpk4;
    P.pkf = p->pkf;
    P.badmap = p->badmap_d;
    P.itemtab = reinterpret_cast<const int4*>(p->itemtab_d);
    P.nrec = p->itemtab_nrec;
    P.ntrip = p->itemtab_ntrip;
    P.blockmap = nullptr;
    if (!SHG_DBG(P, 2048)) {                             // SHG_DEBUG bit 11: plain block order (experiment switch)
        rc = build_blockmap(p, nbt, nit, stream);
        if (rc) return rc;
        P.blockmap = p->blockmap_d;
    }
    P.trig = fold ? p->trig_f : p->trig;
    P.G = grid;
#ifdef SHG_TIMELINE
    P.tl = getenv("SHG_TIMELINE_PTR") ? (unsigned long long*)strtoull(getenv("SHG_TIMELINE_PTR"), nullptr, 0) : nullptr;
#endif
    const size_t lds = fused_lds_bytes(fold ? p->K_f + 1 : p->K);
    const dim3 grid_dim((unsigned)(nbt * P.nit));
    ProfileScope ps(p, 2, stream);
    if (ns) {
        SHG_HIP(hipFuncSetAttribute((const void*)synthesis_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((synthesis_fused_kernel<true>), grid_dim, dim3(512), lds, stream, P);
    } else {
        SHG_HIP(hipFuncSetAttribute((const void*)synthesis_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((synthesis_fused_kernel<false>), grid_dim, dim3(512), lds, stream, P);
    }
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

}  // namespace shg
