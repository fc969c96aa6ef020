// Batched spherical-harmonic synthesis on a regular grid (replaces grates/gravityfield.py:352-368).
//
// Separable formulation, three kernels per pass of `chunk` epochs:
//   pack_coefficients   anm[b][n][m]  ->  cpk[(n,m)][c/s][b]   (epoch fastest, so that one wave of the
//                       Legendre stage fetches the coefficients of 8 epochs with wave-uniform scalar loads)
//   legendre_stage      lane <-> parallel, one order m per wave:  column recursion of P_nm in registers,
//                       A_m(i) = sum_n kn[i][n] P_nm(theta_i) C_nm,  B_m(i) likewise with S_nm
//                       F[b][slot(m,c/s)][i]   (i fastest: coalesced stores, MFMA A-operand friendly)
//   lon_stage           G[b][i][j] = sum_slot F[b][slot][i] T[slot][j] on v_mfma_f64_16x16x4_f64.
//                       With 4-fold symmetric meridians only a quarter of the columns is computed per
//                       (cos/sin, m even/odd) group and the four images are formed in the epilogue.
#include "common.h"

namespace shg {

typedef double double4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// pack_coefficients
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_coefficients_kernel(int N, int nb, int Bpad, const double* __restrict__ anm,
                                                                 double* __restrict__ cpk) {
    const int E = (N + 1) * (N + 1);
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int b0 = blockIdx.y * kEpochTile;
    const int r = e / (N + 1), c = e % (N + 1);
    int idx, cs;
    if (c <= r) {                       // C_nm at [n][m]
        idx = order_offset(N, c) + r - c;
        cs = 0;
    } else {                            // S_nm at [m-1][n]
        idx = order_offset(N, r + 1) + c - (r + 1);
        cs = 1;
    }
    double v[kEpochTile];
#pragma unroll
    for (int bb = 0; bb < kEpochTile; ++bb) v[bb] = (b0 + bb < nb) ? anm[(size_t)(b0 + bb) * E + e] : 0.0;
    double* dst = cpk + ((size_t)idx * 2 + cs) * Bpad + b0;
#pragma unroll
    for (int bb = 0; bb < kEpochTile; ++bb) dst[bb] = v[bb];
}

// ------------------------------------------------------------------------------------------------
// legendre_stage: one wave = 64 parallels x one order x 8 epochs
// ------------------------------------------------------------------------------------------------
struct SlotMap {
    int sym4;
    int N;
    int goff[4];
    __device__ int cosine(int m) const { return sym4 ? goff[m & 1] + (m >> 1) : m; }
    __device__ int sine(int m) const { return sym4 ? goff[2 + (m & 1)] + ((m & 1) ? (m >> 1) : (m >> 1) - 1) : N + m; }
};

__global__ __launch_bounds__(64) void legendre_stage_kernel(int N, int ldlat, int K, int Bpad, SlotMap map,
                                                            const double* __restrict__ ct, const double* __restrict__ pmm,
                                                            const double* __restrict__ knT, const double* __restrict__ arec,
                                                            const double* __restrict__ brec, const double* __restrict__ cpk,
                                                            double* __restrict__ F) {
    const int i = blockIdx.x * kLatTile + threadIdx.x;
    const int m = blockIdx.y;
    const int b0 = blockIdx.z * kEpochTile;
    const int off = order_offset(N, m);

    const double t = ct[i];
    double p1 = pmm[(size_t)m * ldlat + i];     // P_mm
    double p2 = 0.0;
    double accC[kEpochTile], accS[kEpochTile];
#pragma unroll
    for (int bb = 0; bb < kEpochTile; ++bb) accC[bb] = accS[bb] = 0.0;

    {
        const double pk = p1 * knT[(size_t)m * ldlat + i];
        const double* c = cpk + (size_t)off * 2 * Bpad + b0;
#pragma unroll
        for (int bb = 0; bb < kEpochTile; ++bb) {
            accC[bb] = fma(pk, c[bb], accC[bb]);
            accS[bb] = fma(pk, c[Bpad + bb], accS[bb]);
        }
    }
    for (int n = m + 1; n <= N; ++n) {
        const int idx = off + n - m;
        const double a = arec[idx], bq = brec[idx];
        // reference arithmetic, no contraction (file is built with -ffp-contract=off):
        //   sqrt(..) * cos * P[n-1] - sqrt(..) * P[n-2]                         grates/utilities.py:52-54
        const double p = (a * t) * p1 - bq * p2;
        p2 = p1;
        p1 = p;
        const double pk = p * knT[(size_t)n * ldlat + i];
        const double* c = cpk + (size_t)idx * 2 * Bpad + b0;
#pragma unroll
        for (int bb = 0; bb < kEpochTile; ++bb) {
            accC[bb] = fma(pk, c[bb], accC[bb]);
            accS[bb] = fma(pk, c[Bpad + bb], accS[bb]);
        }
    }
    const int sc = map.cosine(m);
#pragma unroll
    for (int bb = 0; bb < kEpochTile; ++bb) F[((size_t)(b0 + bb) * K + sc) * ldlat + i] = accC[bb];
    if (m >= 1) {
        const int ss = map.sine(m);
#pragma unroll
        for (int bb = 0; bb < kEpochTile; ++bb) F[((size_t)(b0 + bb) * K + ss) * ldlat + i] = accS[bb];
    }
}

// ------------------------------------------------------------------------------------------------
// lon_stage: block tile 64 rows x 64 columns (per group), 4 waves as 2 x 2, wave tile 32 x 32.
// ------------------------------------------------------------------------------------------------
constexpr int kLdsStride = 80;   // doubles per k-row: 64 + 16 pad -> the 4 k-rows of one fragment read hit disjoint banks
constexpr int kChunkSlots = 16;  // K slots staged per barrier pair

struct LonParams {
    int nlat, nlon, ldlat, K, ncol, nrt;   // nrt = row tiles (16 rows) per epoch
    int total_rt;                          // row tiles in this pass
    int goff[5];
    const double* F;
    const double* trig;                    // [coltile][K][16]
    double* G;                             // [nb][nlat][nlon]
};

template <int NG>
__global__ __launch_bounds__(256) void lon_stage_kernel(LonParams P) {
    __shared__ double As[kChunkSlots * kLdsStride];
    __shared__ double Bs[kChunkSlots * kLdsStride];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int rt_block = blockIdx.x * 4;          // first 16-row tile of this block
    const int ct_block = blockIdx.y * 4;          // first 16-column tile of this block

    double4_t acc[NG][2][2];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[g][a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};

    // staging assignment: a chunk row (one K slot) of 64 doubles = 32 x 16-byte pieces; 16 slots -> 512 pieces, 2 per thread
    const int ld_piece = tid & 31;                // which 16-byte piece of the 64-double row
    const int ld_slot0 = tid >> 5;                // 0..7, second piece is slot + 8
    const int ld_tile = ld_piece >> 3;            // 16-row / 16-column tile inside the block tile
    const int ld_in = (ld_piece & 7) * 2;         // double offset inside the tile

    // A source: F[(b*K + k)*ldlat + i0 + ld_in]
    const int rt_a = rt_block + ld_tile;
    const bool a_ok = rt_a < P.total_rt;
    const int b_a = a_ok ? rt_a / P.nrt : 0;
    const int i_a = a_ok ? (rt_a % P.nrt) * 16 : 0;
    const double* a_src = P.F + ((size_t)b_a * P.K) * P.ldlat + i_a + ld_in;
    // B source: trig[(ct*K + k)*16 + ld_in]
    const int ct_b = ct_block + ld_tile;
    const bool b_ok = ct_b * 16 < P.ncol;
    const double* b_src = P.trig + ((size_t)(b_ok ? ct_b : 0) * P.K) * 16 + ld_in;

    const int fr = lane & 15, fk = lane >> 4;

#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int k_begin = P.goff[g], k_end = P.goff[g + 1];
        for (int k0 = k_begin; k0 < k_end; k0 += kChunkSlots) {
            const int nslots = min(kChunkSlots, k_end - k0);
            __syncthreads();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int s = ld_slot0 + 8 * h;
                double2 va = make_double2(0.0, 0.0), vb = make_double2(0.0, 0.0);
                if (s < nslots) {
                    if (a_ok) va = *reinterpret_cast<const double2*>(a_src + (size_t)(k0 + s) * P.ldlat);
                    if (b_ok) vb = *reinterpret_cast<const double2*>(b_src + (size_t)(k0 + s) * 16);
                }
                *reinterpret_cast<double2*>(&As[s * kLdsStride + ld_tile * 16 + ld_in]) = va;
                *reinterpret_cast<double2*>(&Bs[s * kLdsStride + ld_tile * 16 + ld_in]) = vb;
            }
            __syncthreads();
            const int nsteps = nslots >> 2;
            for (int ks = 0; ks < nsteps; ++ks) {
                const int row = (ks * 4 + fk) * kLdsStride;
                const double a0 = As[row + (wr * 2 + 0) * 16 + fr];
                const double a1 = As[row + (wr * 2 + 1) * 16 + fr];
                const double b0 = Bs[row + (wc * 2 + 0) * 16 + fr];
                const double b1 = Bs[row + (wc * 2 + 1) * 16 + fr];
                acc[g][0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[g][0][0], 0, 0, 0);
                acc[g][0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[g][0][1], 0, 0, 0);
                acc[g][1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[g][1][0], 0, 0, 0);
                acc[g][1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[g][1][1], 0, 0, 0);
            }
        }
    }

    // epilogue: C/D layout of v_mfma_f64_16x16x4: column = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int rt = rt_block + wr * 2 + a;
        if (rt >= P.total_rt) continue;
        const int b = rt / P.nrt;
        const int i0 = (rt % P.nrt) * 16;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int j = (ct_block + wc * 2 + c) * 16 + fr;
            if (j >= P.ncol) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + fk + 4 * r;
                if (i >= P.nlat) continue;
                double* row = P.G + ((size_t)b * P.nlat + i) * P.nlon;
                if (NG == 4) {
                    const double ee = acc[0][a][c][r], eo = acc[1 % NG][a][c][r];
                    const double oe = acc[2 % NG][a][c][r], oo = acc[3 % NG][a][c][r];
                    const double s1 = ee + eo, s2 = ee - eo, d1 = oe + oo, d2 = oe - oo;
                    row[j] = s1 + d1;                       // lon_j
                    row[P.nlon - 1 - j] = s1 - d1;          // -lon_j
                    row[P.nlon / 2 - 1 - j] = s2 - d2;      // -pi - lon_j
                    row[P.nlon / 2 + j] = s2 + d2;          // lon_j + pi
                } else {
                    row[j] = acc[0][a][c][r];
                }
            }
        }
    }
}

}  // namespace shg

using namespace shg;

static int synthesis_dispatch(shg_plan* p, const double* anm, int B, double* grid, hipStream_t stream);

extern "C" int shg_synthesis(shg_plan* p, const double* anm, int B, double* grid, void* stream_) {
    SHG_REQUIRE(p != nullptr, "shg_synthesis: NULL plan");
    SHG_REQUIRE(B >= 0, "shg_synthesis: negative batch size %d", B);
    if (B == 0) return SHG_OK;
    SHG_REQUIRE(anm != nullptr && grid != nullptr, "shg_synthesis: NULL array pointer");
    hipStream_t stream = (hipStream_t)stream_;
    PlanGuard guard(p, stream);
    return synthesis_dispatch(p, anm, B, grid, stream);
}

extern "C" int shg_synthesis_om(shg_plan* p, const double* om, int Ns, int B, int Bpad, double* grid, void* stream_) {
    SHG_REQUIRE(p != nullptr, "shg_synthesis_om: NULL plan");
    SHG_REQUIRE(B >= 0 && Bpad >= B && Bpad % 32 == 0, "shg_synthesis_om: need 0 <= B <= Bpad, Bpad a multiple of 32");
    SHG_REQUIRE(Ns >= p->N, "shg_synthesis_om: the series holds degrees up to %d, the plan needs %d", Ns, p->N);
    if (B == 0) return SHG_OK;
    SHG_REQUIRE(om != nullptr && grid != nullptr, "shg_synthesis_om: NULL array pointer");
    hipStream_t stream = (hipStream_t)stream_;
    PlanGuard guard(p, stream);
    const bool fused_ns = p->sym_ns && (p->path >= 6 || p->path == 2 || (p->path == 0 && (rot_applicable(p) || fused_chunk_for(p) != 0)));
    SHG_REQUIRE(fused_ns, "shg_synthesis_om: only the fused kernels on north-south symmetric parallels read an order-major series (unpack it for other plans)");
    p->om_src = om;
    p->om_N = Ns;
    p->om_Bpad = Bpad;
    const int rc = synthesis_dispatch(p, om, B, grid, stream);      // (the coefficient pointer is not used: the repack reads p->om_src)
    p->om_src = nullptr;
    return rc;
}

static int synthesis_dispatch(shg_plan* p, const double* anm, int B, double* grid, hipStream_t stream) {
    // degrees beyond the 64-row panel (d/o 127 ... ~210): the 32-row fused kernel still beats the three-kernel path
    if (p->path >= 6 || (p->path == 0 && rot_applicable(p))) return synthesis_rot(p, anm, B, grid, stream);
    if (p->path == 5 || (p->path == 0 && fused_chunk_for(p) == 0 && fused32_applicable(p))) return synthesis_fused32(p, anm, B, grid, stream);
    if (p->path >= 2 || (p->path == 0 && fused_chunk_for(p) != 0)) return synthesis_fused(p, anm, B, grid, stream);
    int rc = plan_alloc_workspace(p);
    if (rc) return rc;

    const int N = p->N;
    const int E = (N + 1) * (N + 1);
    const int Bpad = round_up(p->chunk, kEpochTile);
    SlotMap map;
    map.sym4 = p->sym4 ? 1 : 0;
    map.N = N;
    for (int g = 0; g < 4; ++g) map.goff[g] = p->goff[g < p->ngroups ? g : 0];

    for (int c0 = 0; c0 < B; c0 += p->chunk) {
        const int nb = std::min(p->chunk, B - c0);
        const int nbt = ceil_div(nb, kEpochTile);
        {
            ProfileScope ps(p, 0, stream);
            hipLaunchKernelGGL(pack_coefficients_kernel, dim3(ceil_div(E, 256), nbt), dim3(256), 0, stream, N, nb, Bpad,
                               anm + (size_t)c0 * E, p->cpk);
        }
        {
            ProfileScope ps(p, 1, stream);
            hipLaunchKernelGGL(legendre_stage_kernel, dim3(p->ldlat / kLatTile, N + 1, nbt), dim3(64), 0, stream, N, p->ldlat,
                               p->K, Bpad, map, p->ct, p->pmm, p->knT, p->arec, p->brec, p->cpk, p->F);
        }
        LonParams L;
        L.nlat = p->nlat;
        L.nlon = p->nlon;
        L.ldlat = p->ldlat;
        L.K = p->K;
        L.ncol = p->ncol;
        L.nrt = ceil_div(p->nlat, 16);
        L.total_rt = nb * L.nrt;
        for (int g = 0; g < 5; ++g) L.goff[g] = p->goff[g];
        L.F = p->F;
        L.trig = p->trig;
        L.G = grid + (size_t)c0 * p->nlat * p->nlon;
        dim3 grid_dim(ceil_div(L.total_rt, 4), ceil_div(p->ncoltiles, 4));
        ProfileScope ps(p, 2, stream);
        if (p->sym4)
            hipLaunchKernelGGL(lon_stage_kernel<4>, grid_dim, dim3(256), 0, stream, L);
        else
            hipLaunchKernelGGL(lon_stage_kernel<1>, grid_dim, dim3(256), 0, stream, L);
    }
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}
