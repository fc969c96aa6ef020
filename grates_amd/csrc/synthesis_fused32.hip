// Fused batched synthesis, two workgroups per CU (north-south symmetric grids with 4-fold symmetric meridians).
//
// Same algorithm as synthesis_fused.hip (Legendre stage on MFMA into an LDS panel, longitude stage on MFMA out of it,
// four longitude images and 16-byte stores in the epilogue) with half-size workgroups:
//   workgroup = 4 epochs x (4 northern parallels + their 4 mirror images) = 32 panel rows, 4 waves, K * 48 * 8 bytes of
//   LDS (79.9 KB at d/o 96), so that TWO workgroups are resident per CU and the Legendre phase (L2 bound), the longitude
//   phase (MFMA bound) and the epilogue (store bound) of different workgroups can overlap.
// Measured (r01): 0.92 ms per 240-epoch launch against 0.80 ms of the 64-row kernel in the same process -- the Legendre
// phase costs twice as much per output (half of the MFMA columns are padding with 4 parallels per block) and the overlap
// does not pay for it, with or without staggering the two resident sets.  Kept as an explicit path
// (shg_plan_set_path(plan, 5)) for the next round's experiments; the automatic choice is the 64-row kernel.
// Longitude stage: wave w owns column tiles cb * 8 + 2 w and cb * 8 + 2 w + 1 of every column block (2 column tiles x
// 2 row tiles x 4 groups = 16 accumulators), so a body is again 16 MFMAs per 4 k-steps.
#include <cstdlib>

#include "common.h"

namespace shg {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

constexpr int kPanelStride32 = 48;   // 32 rows + 16 pad: k-rows of a fragment read fall on disjoint LDS bank halves

// PK table of the 4-parallel blocks in MFMA-fragment order (see pkf_ns_table_kernel): lane = (pair & 3) * 16 + parity * 8 + (i & 3),
// lanes with (lane & 7) >= 4 stay zero.  z = 1: mirrored parallels of the blocks flagged in badmap (tile nit + rank).
__global__ __launch_bounds__(64) void pkf32_table_kernel(int N, int nlat, int ldlat, int nh, int nit, int Qtot, const int* __restrict__ qoff,
                                                         const int* __restrict__ badmap, const double* __restrict__ ct,
                                                         const double* __restrict__ pmm, const double* __restrict__ knT,
                                                         const double* __restrict__ arec, const double* __restrict__ brec,
                                                         double* __restrict__ pkf) {
    const int i = blockIdx.x * 64 + threadIdx.x;                    // northern parallel
    if (i >= nh) return;
    int row = i, tile = i >> 2;
    if (blockIdx.z == 1) {
        const int rank = badmap[i >> 2];
        if (rank < 0) return;
        row = nlat - 1 - i;
        tile = nit + rank;
    }
    const int m = blockIdx.y;
    const int off = order_offset(N, m);
    const double t = ct[row];
    double* dst = pkf + (((size_t)tile * Qtot + qoff[m]) * 64 + (i & 3)) * 2;
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

struct Fused32Params {
    int N, nlat, nlon, K, ncol, B, nit, ncb, nh, Qtot;
    int goff[5];
    int gcount[4];
#ifdef SHG_EXPERIMENT
    int dbg;
#endif
    const int* qoff;
    const int* badmap;        // [nit]
    const double* cpk4;       // [nbt][Qtot][64][2]  (pack_coefficients4_ns_kernel)
    const double* pkf;        // [nit + nbad][Qtot][64][2]
    const double* trig;       // [ncb * 8][K][16]
    double* G;
};

__device__ inline double swap_lane8(double x) {                     // lane +- 8 inside the row of 16 (DPP row_ror:8)
    const long long bits = __builtin_bit_cast(long long, x);
    const int lo = (int)bits, hi = (int)(bits >> 32);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xF, 0xF, true);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi2 << 32) | (unsigned int)lo2);
}

struct Item32 {                                                     // two octets (32 degrees) of order m starting at octet j0
    int m, j0;
    __device__ bool valid(int N) const { return m <= N; }
    __device__ Item32 next(int N) const {
        Item32 r = {m, j0 + 2};
        if (r.j0 * 16 >= N + 1 - m) {
            r.m = m + 4;                                            // orders are dealt to the 4 waves
            r.j0 = 0;
        }
        return r;
    }
};

__global__ __launch_bounds__(256, 2) void synthesis_fused32_kernel(Fused32Params P) {
    extern __shared__ double As[];                     // panel [K][kPanelStride32], row = epoch * 8 + slot (0-3 north, 4-7 south)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbt = (P.B + 3) >> 2;
    const int bt = blockIdx.x % nbt;                   // epoch tile fastest: neighbouring blocks share the PK slab
    const int it = blockIdx.x / nbt;
    const int i0n = it * 4;
    const int fr = lane & 15, fk = lane >> 4;

    // zero the padding slots of the panel
    for (int g = 0; g < 4; ++g)
        for (int s = P.goff[g] + P.gcount[g]; s < P.goff[g + 1]; ++s)
            if (tid < 32) As[s * kPanelStride32 + tid] = 0.0;

    // ---- phase 1: Legendre stage (see synthesis_fused.hip; NS layout, 4 parallels per block: columns 0-3 / 8-11 of the
    //      MFMA tile carry the even / odd degree sums of the 4 northern parallels)
    if (!SHG_DBG(P, 2)) {
        const int bad = P.badmap[it];
        const double* pkb = P.pkf + ((size_t)it * P.Qtot * 64 + lane) * 2;
        const double* cf = P.cpk4 + ((size_t)bt * P.Qtot * 64 + lane) * 2;
        int mode = bad >= 0 ? 1 : 0;
        // panel row written by this lane: lanes fr < 4 hold E (epoch fk, parallel fr), lanes 8-11 hold O
        int prow = fk * 8 + (fr < 8 ? fr : fr - 4);               // mode 0: north slots 0-3 from lanes 0-3, south slots 4-7 from lanes 8-11
        double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};

#define SHG_P1_ISSUE(item, ALO, AHI, BLO, BHI)                                               \
    do {                                                                                     \
        const int q_ = (P.N + 16 - (item).m) >> 4;                                           \
        const int o0_ = P.qoff[(item).m] + (item).j0;                                        \
        const int o1_ = o0_ + ((item).j0 + 1 < q_ ? 1 : 0);                                  \
        ALO = *reinterpret_cast<const double2*>(cf + (size_t)o0_ * 128);                     \
        BLO = *reinterpret_cast<const double2*>(pkb + (size_t)o0_ * 128);                    \
        AHI = *reinterpret_cast<const double2*>(cf + (size_t)o1_ * 128);                     \
        BHI = *reinterpret_cast<const double2*>(pkb + (size_t)o1_ * 128);                    \
    } while (0)

#define SHG_P1_CONSUME(item, nxt, ALO, AHI, BLO, BHI)                                                               \
    do {                                                                                                            \
        const bool lo_ = (item).m <= P.N;                                                                           \
        const bool hi_ = lo_ && ((item).j0 + 1) * 16 < P.N + 1 - (item).m;                                          \
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lo_ ? ALO.x : 0.0, BLO.x, acc0, 0, 0, 0);                       \
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lo_ ? ALO.y : 0.0, BLO.y, acc1, 0, 0, 0);                       \
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(hi_ ? AHI.x : 0.0, BHI.x, acc0, 0, 0, 0);                       \
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(hi_ ? AHI.y : 0.0, BHI.y, acc1, 0, 0, 0);                       \
        if ((nxt).m != (item).m && (item).m <= P.N) {                                                               \
            /* regs 0, 1: even-degree sums E (cos, sin) in columns 0-3; regs 2, 3: odd-degree sums O in columns 8-11 */ \
            const int m_ = (item).m;                                                                                \
            double vc_ = acc0[0] + acc1[0], vs_ = acc0[1] + acc1[1];                                                \
            const double oc_ = acc0[2] + acc1[2], os_ = acc0[3] + acc1[3];                                          \
            const double rc_ = swap_lane8(fr < 8 ? vc_ : oc_), rs_ = swap_lane8(fr < 8 ? vs_ : os_);                \
            vc_ = (mode == 0 && fr >= 8) ? rc_ - oc_ : vc_ + rc_;                                                   \
            vs_ = (mode == 0 && fr >= 8) ? rs_ - os_ : vs_ + rs_;                                                   \
            if ((fr & 7) < 4 && (mode == 0 || fr < 8)) {                                                            \
                As[(P.goff[m_ & 1] + (m_ >> 1)) * kPanelStride32 + prow] = vc_;                                     \
                if (m_ >= 1)                                                                                        \
                    As[(P.goff[2 + (m_ & 1)] + ((m_ & 1) ? (m_ >> 1) : (m_ >> 1) - 1)) * kPanelStride32 + prow] = vs_; \
            }                                                                                                       \
            acc0 = (double4_t){0.0, 0.0, 0.0, 0.0};                                                                 \
            acc1 = (double4_t){0.0, 0.0, 0.0, 0.0};                                                                 \
        }                                                                                                           \
    } while (0)

        double2 xal = {0, 0}, xah = {0, 0}, xbl = {0, 0}, xbh = {0, 0};
        double2 yal = {0, 0}, yah = {0, 0}, ybl = {0, 0}, ybh = {0, 0};
        int nitems = 0;
        for (int m = wave; m <= P.N; m += 4) nitems += (P.N + 1 - m + 31) >> 5;
        for (int pass = 0; pass < (bad >= 0 ? 2 : 1); ++pass) {
            if (pass == 1) {                                          // mirrored parallels of a polar block: their own table
                mode = 2;
                prow = fk * 8 + 4 + fr;                               // lanes 0-3 now hold E + O of the mirrored parallels
                pkb = P.pkf + ((size_t)(P.nit + bad) * P.Qtot * 64 + lane) * 2;
            }
            Item32 cur = {wave, 0};
            const Item32 first = cur;
            if (nitems > 0) SHG_P1_ISSUE(cur, xal, xah, xbl, xbh);
            for (int trip = 0; trip < (nitems + 1) / 2; ++trip) {
                const Item32 nx = cur.next(P.N);
                const Item32 ld1 = nx.valid(P.N) ? nx : first;
                SHG_P1_ISSUE(ld1, yal, yah, ybl, ybh);
                SHG_P1_CONSUME(cur, nx, xal, xah, xbl, xbh);
                const Item32 nn = nx.next(P.N);
                const Item32 ld2 = nn.valid(P.N) ? nn : first;
                SHG_P1_ISSUE(ld2, xal, xah, xbl, xbh);
                SHG_P1_CONSUME(nx, nn, yal, yah, ybl, ybh);
                cur = nn;
            }
        }
#undef SHG_P1_ISSUE
#undef SHG_P1_CONSUME
    }
    __syncthreads();          // panel complete; from here on it is read-only and the waves run independently

    // ---- phase 2: longitude stage.  Row tile rt = panel rows 16 rt .. 16 rt + 15 = epochs 2 rt, 2 rt + 1.
    const int nbody = P.K >> 4;
    const bool pair_stores = (P.ncol & 1) == 0;
    const int par = fr & 1;
    // fragments of column tile (cb * 8 + 2 wave + ct): + cb * cb_stride + ct * ct_stride + body * 256 + u * 64
    const double* tbase = P.trig + ((size_t)(2 * wave) * P.K + fk) * 16 + fr;
    const size_t ct_stride = (size_t)P.K * 16;
    const size_t cb_stride = (size_t)8 * P.K * 16;
    double ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;     // B fragments of the current body: column tiles a and b
    ra0 = tbase[0];
    ra1 = tbase[64];
    ra2 = tbase[128];
    ra3 = tbase[192];
    rb0 = tbase[ct_stride];
    rb1 = tbase[ct_stride + 64];
    rb2 = tbase[ct_stride + 128];
    rb3 = tbase[ct_stride + 192];
    for (int ccb = 0; ccb < P.ncb && !SHG_DBG(P, 4); ++ccb) {
        double4_t acc[4][2][2];                        // [group][column tile][row tile]
#pragma unroll
        for (int gg = 0; gg < 4; ++gg)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) acc[gg][ct][rt] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            for (int body = P.goff[gg] >> 4; body < (P.goff[gg + 1] >> 4) && !SHG_DBG(P, 8); ++body) {
                int nb_ = body + 1, ncb_ = ccb;        // next body of the flat (column block, body) sequence; clamped at the very end
                if (nb_ == nbody) {
                    nb_ = 0;
                    ncb_ = min(ccb + 1, P.ncb - 1);
                }
                const double* tn = tbase + (size_t)ncb_ * cb_stride + (size_t)nb_ * 256;
                const double na0 = tn[0], na1 = tn[64], na2 = tn[128], na3 = tn[192];
                const double nb0 = tn[ct_stride], nb1 = tn[ct_stride + 64], nb2 = tn[ct_stride + 128], nb3 = tn[ct_stride + 192];
                const double* ap = As + (size_t)(body * 16 + fk) * kPanelStride32 + fr;
                double a0[2], a1[2];
#define SHG_READ_A(dst, u)                                                     \
    _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) dst[rt] = ap[(u) * 4 * kPanelStride32 + rt * 16]
#define SHG_MFMA4(src, fa, fb)                                                 \
    _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) {                         \
        acc[gg][0][rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(src[rt], fa, acc[gg][0][rt], 0, 0, 0); \
        acc[gg][1][rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(src[rt], fb, acc[gg][1][rt], 0, 0, 0); \
    }
                SHG_READ_A(a0, 0);
                SHG_READ_A(a1, 1);
                __builtin_amdgcn_sched_barrier(0);
                SHG_MFMA4(a0, ra0, rb0);
                __builtin_amdgcn_sched_barrier(0);
                SHG_READ_A(a0, 2);
                SHG_MFMA4(a1, ra1, rb1);
                __builtin_amdgcn_sched_barrier(0);
                SHG_READ_A(a1, 3);
                SHG_MFMA4(a0, ra2, rb2);
                __builtin_amdgcn_sched_barrier(0);
                SHG_MFMA4(a1, ra3, rb3);
                __builtin_amdgcn_sched_barrier(0);
                ra0 = na0;
                ra1 = na1;
                ra2 = na2;
                ra3 = na3;
                rb0 = nb0;
                rb1 = nb1;
                rb2 = nb2;
                rb3 = nb3;
#undef SHG_READ_A
#undef SHG_MFMA4
            }
        }
        // epilogue of this column block: four longitude images per quarter-column (see synthesis_fused.hip).
        // D row = fk + 4 reg of row tile rt = panel row 16 rt + fk + 4 reg: reg 0 / 1 = north / south slot fk of epoch 2 rt,
        // reg 2 / 3 = the same of epoch 2 rt + 1.
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int jt = (ccb * 8 + 2 * wave + ct) * 16;
            if (jt >= P.ncol) continue;                               // uniform per wave (padding tile)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                double img[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double ee = acc[0][ct][rt][r], eo = acc[1][ct][rt][r], oe = acc[2][ct][rt][r], oo = acc[3][ct][rt][r];
                    const double s1 = ee + eo, s2 = ee - eo, d1 = oe + oo, d2 = oe - oo;
                    img[0][r] = s1 + d1;
                    img[1][r] = s1 - d1;
                    img[2][r] = s2 - d2;
                    img[3][r] = s2 + d2;
                }
                if (SHG_DBG(P, 1)) {
                    if (img[0][0] != 1.2345e-300) continue;
                }
                const bool rowok = i0n + fk < P.nh;
                const int inorth = i0n + fk, isouth = P.nlat - 1 - inorth;
                if (pair_stores) {
                    // lanes (2q, 2q+1) hold adjacent columns: the even lane keeps epoch 2 rt (regs 0, 1), the odd lane takes
                    // epoch 2 rt + 1 (regs 2, 3); after the swap every lane owns 2 rows x 2 adjacent columns of one epoch
                    const int jc = jt + (fr & ~1);
                    const int b = bt * 4 + 2 * rt + par;
                    const bool ok = rowok && jc < P.ncol && b < P.B;
                    double* rown = P.G + ((size_t)(ok ? b : 0) * P.nlat + (ok ? inorth : 0)) * P.nlon;
                    double* rows = P.G + ((size_t)(ok ? b : 0) * P.nlat + (ok ? isouth : 0)) * P.nlon;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        double n_lo, n_hi, s_lo, s_hi;
                        pair_exchange(img[t][0], img[t][2], 0xAAAAAAAAAAAAAAAAull, n_lo, n_hi);
                        pair_exchange(img[t][1], img[t][3], 0xAAAAAAAAAAAAAAAAull, s_lo, s_hi);
                        int col;
                        bool ascending;
                        if (t == 0) { col = jc; ascending = true; }
                        else if (t == 1) { col = P.nlon - 2 - jc; ascending = false; }
                        else if (t == 2) { col = P.nlon / 2 - 2 - jc; ascending = false; }
                        else { col = P.nlon / 2 + jc; ascending = true; }
                        const double2_t vn = ascending ? (double2_t){n_lo, n_hi} : (double2_t){n_hi, n_lo};
                        const double2_t vs = ascending ? (double2_t){s_lo, s_hi} : (double2_t){s_hi, s_lo};
                        if (ok) {                                         // streaming data: non-temporal stores
                            __builtin_nontemporal_store(vn, reinterpret_cast<double2_t*>(rown + col));
                            __builtin_nontemporal_store(vs, reinterpret_cast<double2_t*>(rows + col));
                        }
                    }
                } else {
                    const int j = jt + fr;
                    if (j >= P.ncol || !rowok) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int b = bt * 4 + 2 * rt + (r >> 1);
                        if (b >= P.B) continue;
                        double* row = P.G + ((size_t)b * P.nlat + ((r & 1) ? isouth : inorth)) * P.nlon;
                        row[j] = img[0][r];
                        row[P.nlon - 1 - j] = img[1][r];
                        row[P.nlon / 2 - 1 - j] = img[2][r];
                        row[P.nlon / 2 + j] = img[3][r];
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void pack_coefficients4_ns_kernel(int N, int B, int Qtot, const int* __restrict__ qoff,
                                                                     const double* __restrict__ anm, double* __restrict__ cpk4);   // synthesis_fused.hip

static size_t fused32_lds_bytes(int K) { return (size_t)K * kPanelStride32 * sizeof(double); }

// non-zero when this variant applies: both symmetries and the panel in the 160 KiB LDS (degrees up to ~210; two workgroups
// per CU up to degree 96)
int fused32_applicable(const shg_plan* p) {
    if (!p->sym4 || !p->sym_ns || (p->K & 15)) return 0;
    if ((long long)p->nlat * p->nlon * 8 >= (1LL << 31)) return 0;
    return fused32_lds_bytes(p->K) <= 160 * 1024 ? 1 : 0;
}

int build_pkf32_table(shg_plan* p, hipStream_t stream) {
    if (p->pkf32) return SHG_OK;
    const int N = p->N, nh = p->nlat / 2, nit = ceil_div(nh, 4);
    // octet offsets (16 degrees per octet) and the polar-block map at 4-parallel granularity
    std::vector<int> qoff(N + 2);
    int q = 0;
    for (int m = 0; m <= N; ++m) {
        qoff[m] = q;
        q += (N + 1 - m + 15) / 16;
    }
    qoff[N + 1] = q;
    std::vector<int> badmap(nit, -1);
    int nbad = 0;
    for (int i = 0; i < nh; ++i)
        if (p->ns_badrow[i] && badmap[i >> 2] < 0) badmap[i >> 2] = nbad++;
    if (hipMalloc((void**)&p->qoff32, qoff.size() * sizeof(int)) != hipSuccess || hipMalloc((void**)&p->badmap32_d, badmap.size() * sizeof(int)) != hipSuccess)
        return fail(SHG_ERR_NOMEM, "octet / block map allocation failed");
    SHG_HIP(hipMemcpy(p->qoff32, qoff.data(), qoff.size() * sizeof(int), hipMemcpyHostToDevice));
    SHG_HIP(hipMemcpy(p->badmap32_d, badmap.data(), badmap.size() * sizeof(int), hipMemcpyHostToDevice));
    p->Qtot32 = q;
    p->nbad32 = nbad;
    const size_t n = (size_t)(nit + nbad) * q * 128;
    if (hipMalloc((void**)&p->pkf32, n * sizeof(double)) != hipSuccess) return fail(SHG_ERR_NOMEM, "PK table allocation failed (%zu doubles)", n);
    SHG_HIP(hipMemsetAsync(p->pkf32, 0, n * sizeof(double), stream));
    hipLaunchKernelGGL(pkf32_table_kernel, dim3(ceil_div(nh, 64), N + 1, 2), dim3(64), 0, stream, N, p->nlat, p->ldlat, nh, nit, q, p->qoff32, p->badmap32_d,
                       p->ct, p->pmm, p->knT, p->arec, p->brec, p->pkf32);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

int synthesis_fused32(shg_plan* p, const double* anm, int B, double* grid, hipStream_t stream) {
    if (!fused32_applicable(p)) return fail(SHG_ERR_UNSUPPORTED, "two-workgroup fused synthesis not applicable to this plan");
    int rc = build_pkf32_table(p, stream);
    if (rc) return rc;
    const int nbt = ceil_div(B, 4);
    const int N = p->N;
    const int nh = p->nlat / 2, nit = ceil_div(nh, 4);
    const size_t need = (size_t)nbt * p->Qtot32 * 128;
    if (need > p->cpk4_size) {
        if (p->cpk4) {
            SHG_HIP(hipStreamSynchronize(stream));
            (void)hipFree(p->cpk4);
            p->cpk4 = nullptr;
            p->cpk4_size = 0;                        // a failed grow must not leave the old size behind
            p->cpk4_zeroed = 0;
        }
        if (hipMalloc((void**)&p->cpk4, need * sizeof(double)) != hipSuccess) return fail(SHG_ERR_NOMEM, "coefficient workspace allocation failed");
        p->cpk4_size = need;
    }
    if (need > 0 && (p->cpk4_variant != 4 || p->cpk4_zeroed < need)) {      // same coefficient layout as the NS variant of synthesis_fused.hip
        SHG_HIP(hipMemsetAsync(p->cpk4, 0, p->cpk4_size * sizeof(double), stream));
        p->cpk4_variant = 4;
        p->cpk4_zeroed = p->cpk4_size;
    }
    Fused32Params P;
    P.N = N;
    P.nlat = p->nlat;
    P.nlon = p->nlon;
    P.K = p->K;
    P.ncol = p->ncol;
    P.B = B;
    P.nit = nit;
    P.ncb = ceil_div(p->ncoltiles, 8);
    P.nh = nh;
    P.Qtot = p->Qtot32;
    for (int g = 0; g < 5; ++g) P.goff[g] = p->goff[g];
    const int cnt[4] = {N / 2 + 1, (N + 1) / 2, N / 2, (N + 1) / 2};
    for (int g = 0; g < 4; ++g) P.gcount[g] = cnt[g];
#ifdef SHG_EXPERIMENT
    P.dbg = experiment_switches();
#endif
    P.qoff = p->qoff32;
    P.badmap = p->badmap32_d;
    P.cpk4 = p->cpk4;
    P.pkf = p->pkf32;
    P.trig = p->trig;
    P.G = grid;
    const int E = (N + 1) * (N + 1);
    {
        ProfileScope ps(p, 0, stream);
        hipLaunchKernelGGL(pack_coefficients4_ns_kernel, dim3(ceil_div(E, 256), nbt), dim3(256), 0, stream, N, B, p->Qtot32, p->qoff32, anm, p->cpk4);
    }
    const size_t lds = fused32_lds_bytes(p->K);
    ProfileScope ps(p, 2, stream);
    SHG_HIP(hipFuncSetAttribute((const void*)synthesis_fused32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(synthesis_fused32_kernel, dim3((unsigned)(nbt * nit)), dim3(256), lds, stream, P);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

}  // namespace shg
