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
//   phase 2 (longitude stage on MFMA): panel (64 x K) times the cos/sin table, streamed through a
//     double-buffered LDS ring in chunks of CHUNK K-slots, 8 column tiles (128 quarter-columns) at a time;
//     the four longitude images of every quarter-column are formed in registers and stored.
#include "common.h"

namespace shg {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int kPanelStride = 80;    // 64 rows + 16 pad: k-rows of one fragment read fall on disjoint LDS banks
constexpr int kTrigStride = 144;    // 128 columns + 16 pad

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

// coefficients of 4 epochs side by side:  cpk4[bt][(m, n)][c/s][4]
__global__ __launch_bounds__(256) void pack_coefficients4_kernel(int N, int B, const double* __restrict__ anm,
                                                                  double* __restrict__ cpk4) {
    const int E = (N + 1) * (N + 1);
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int bt = blockIdx.y;
    const int r = e / (N + 1), c = e % (N + 1);
    int idx, cs;
    if (c <= r) {
        idx = order_offset(N, c) + r - c;
        cs = 0;
    } else {
        idx = order_offset(N, r + 1) + c - (r + 1);
        cs = 1;
    }
    double v[4];
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) v[bb] = (bt * 4 + bb < B) ? anm[(size_t)(bt * 4 + bb) * E + e] : 0.0;
    double* dst = cpk4 + (((size_t)bt * packed_count(N) + idx) * 2 + cs) * 4;
    *reinterpret_cast<double2*>(dst) = make_double2(v[0], v[1]);
    *reinterpret_cast<double2*>(dst + 2) = make_double2(v[2], v[3]);
}

struct FusedParams {
    int N, nlat, nlon, ldlat, K, ncol, B, nit, Ppk, ncb;   // nit = 16-parallel tiles, ncb = column blocks (8 tiles each)
    int goff[5];
    int gcount[4];            // used slots per group (the rest up to goff[g+1] is zero padding)
    const double* cpk4;
    const double* pk;
    const double* trig;       // [ncb * 8][K][16]
    double* G;
};

template <int CHUNK>
__global__ __launch_bounds__(512) void synthesis_fused_kernel(FusedParams P) {
    extern __shared__ double lds[];
    double* As = lds;                                  // [K][kPanelStride]
    double* Bs = lds + (size_t)P.K * kPanelStride;     // [2][CHUNK][kTrigStride]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int bt = blockIdx.x % ((P.B + 3) / 4);       // epoch tile fastest: neighbouring blocks share the PK slab
    const int it = blockIdx.x / ((P.B + 3) / 4);
    const int i0 = it * 16;
    const int fr = lane & 15, fk = lane >> 4;

    // ---- zero the padding slots of the panel
    for (int g = 0; g < 4; ++g)
        for (int s = P.goff[g] + P.gcount[g]; s < P.goff[g + 1]; ++s)
            if (tid < 64) As[s * kPanelStride + tid] = 0.0;

    // ---- phase 1: Legendre stage, orders distributed over the 8 waves
    {
        const double* pkcol = P.pk + i0 + fr;
        const double* cf = P.cpk4 + (size_t)bt * P.Ppk * 8 + fr;
        const bool arow = fr < 8;
        for (int m = wave; m <= P.N; m += 8) {
            const int off = order_offset(P.N, m);
            const int cnt = P.N + 1 - m;
            double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
            for (int k0 = 0; k0 < cnt; k0 += 16) {
                double a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int nl = k0 + u * 4 + fk;
                    const bool ok = nl < cnt;
                    a[u] = (ok && arow) ? cf[(size_t)(off + nl) * 8] : 0.0;
                    b[u] = ok ? pkcol[(size_t)(off + nl) * P.ldlat] : 0.0;
                }
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], acc1, 0, 0, 0);
            }
            // C/D layout: row = (lane >> 4) + 4 reg, col = lane & 15  ->  reg 0 = C part of epoch (lane >> 4),
            // reg 1 = S part; panel row = epoch * 16 + parallel = lane
            const int sc = P.goff[m & 1] + (m >> 1);
            As[sc * kPanelStride + lane] = acc0[0] + acc1[0];
            if (m >= 1) {
                const int ss = P.goff[2 + (m & 1)] + ((m & 1) ? (m >> 1) : (m >> 1) - 1);
                As[ss * kPanelStride + lane] = acc0[1] + acc1[1];
            }
        }
    }

    // ---- phase 2: longitude stage
    const int wr = wave >> 2, wc = wave & 3;           // 2 x 4 waves, wave tile 32 rows x 32 quarter-columns
    // staging of one chunk: CHUNK slots x 8 tiles x 128 B = CHUNK * 64 pieces of 16 B; 512 threads
    constexpr int kPieces = CHUNK * 64 / 512;          // 2 (CHUNK 16) or 1 (CHUNK 8)
    const int st_q = (tid & 7) * 2;                    // double offset inside the 16-column row
    const int st_tile = (tid >> 3) & 7;
    const int st_slot = tid >> 6;                      // 0..7 (+8 for the second piece)

    // flat list of chunks: (cb, g, k0) in lexicographic order
    int cb = 0, g = 0, k0 = P.goff[0];
    auto advance = [&](int& cb_, int& g_, int& k0_) {
        k0_ += CHUNK;
        while (k0_ >= P.goff[g_ + 1]) {       // also skips empty groups (tiny degrees)
            ++g_;
            if (g_ == 4) {
                g_ = 0;
                ++cb_;
                if (cb_ >= P.ncb) return;
            }
            k0_ = P.goff[g_];
        }
    };
    double2 stage[kPieces];
    auto load_chunk = [&](int cb_, int g_, int k0_) {
#pragma unroll
        for (int h = 0; h < kPieces; ++h) {
            const int s = st_slot + 8 * h;
            stage[h] = make_double2(0.0, 0.0);
            if (cb_ < P.ncb && k0_ + s < P.goff[g_ + 1])
                stage[h] = *reinterpret_cast<const double2*>(P.trig + ((size_t)(cb_ * 8 + st_tile) * P.K + k0_ + s) * 16 + st_q);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int h = 0; h < kPieces; ++h) {
            const int s = st_slot + 8 * h;
            *reinterpret_cast<double2*>(&Bs[((size_t)buf * CHUNK + s) * kTrigStride + st_tile * 16 + st_q]) = stage[h];
        }
    };

    load_chunk(cb, g, k0);
    store_chunk(0);
    __syncthreads();          // panel (phase 1) and chunk 0 visible

    double4_t acc[4][2][2];
    int buf = 0;
    for (int ccb = 0; ccb < P.ncb; ++ccb) {
#pragma unroll
        for (int gg = 0; gg < 4; ++gg)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[gg][a][c] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            for (int kk = P.goff[gg]; kk < P.goff[gg + 1]; kk += CHUNK) {
                // prefetch the next chunk of the flat sequence into registers
                int ncb_ = cb, ng_ = g, nk_ = k0;
                advance(ncb_, ng_, nk_);
                load_chunk(ncb_, ng_, nk_);
                const int nsteps = min(CHUNK, P.goff[gg + 1] - kk) >> 2;
                const double* Bb = Bs + (size_t)buf * CHUNK * kTrigStride;
                for (int ks = 0; ks < nsteps; ++ks) {
                    const double* arow = As + (size_t)(kk + ks * 4 + fk) * kPanelStride + fr;
                    const double* brow = Bb + (size_t)(ks * 4 + fk) * kTrigStride + wc * 32 + fr;
                    const double a0 = arow[(wr * 2 + 0) * 16], a1 = arow[(wr * 2 + 1) * 16];
                    const double b0 = brow[0], b1 = brow[16];
                    acc[gg][0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[gg][0][0], 0, 0, 0);
                    acc[gg][0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[gg][0][1], 0, 0, 0);
                    acc[gg][1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[gg][1][0], 0, 0, 0);
                    acc[gg][1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[gg][1][1], 0, 0, 0);
                }
                store_chunk(buf ^ 1);
                __syncthreads();
                buf ^= 1;
                cb = ncb_;
                g = ng_;
                k0 = nk_;
            }
        }
        // epilogue of this column block: four longitude images per quarter-column
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int b = bt * 4 + wr * 2 + a;
            if (b >= P.B) continue;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int j = (ccb * 8 + wc * 2 + c) * 16 + fr;
                if (j >= P.ncol) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + fk + 4 * r;
                    if (i >= P.nlat) continue;
                    double* row = P.G + ((size_t)b * P.nlat + i) * P.nlon;
                    const double ee = acc[0][a][c][r], eo = acc[1][a][c][r], oe = acc[2][a][c][r], oo = acc[3][a][c][r];
                    const double s1 = ee + eo, s2 = ee - eo, d1 = oe + oo, d2 = oe - oo;
                    row[j] = s1 + d1;
                    row[P.nlon - 1 - j] = s1 - d1;
                    row[P.nlon / 2 - 1 - j] = s2 - d2;
                    row[P.nlon / 2 + j] = s2 + d2;
                }
            }
        }
    }
}

static size_t fused_lds_bytes(int K, int chunk) { return ((size_t)K * kPanelStride + 2 * (size_t)chunk * kTrigStride) * sizeof(double); }

// 0: not applicable, 16 / 8: chunk size to use
int fused_chunk_for(const shg_plan* p) {
    if (!p->sym4) return 0;
    const size_t limit = 160 * 1024;
    if (fused_lds_bytes(p->K, 16) <= limit) return 16;
    if (fused_lds_bytes(p->K, 8) <= limit) return 8;
    return 0;
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

int synthesis_fused(shg_plan* p, const double* anm, int B, double* grid, hipStream_t stream) {
    const int chunk = fused_chunk_for(p);
    if (chunk == 0) return fail(SHG_ERR_UNSUPPORTED, "fused synthesis not applicable to this plan");
    int rc = build_pk_table(p, stream);
    if (rc) return rc;
    const int nbt = ceil_div(B, 4);
    const int Ppk = packed_count(p->N);
    const size_t need = (size_t)nbt * Ppk * 8;
    if (need > p->cpk4_size) {
        if (p->cpk4) {
            SHG_HIP(hipStreamSynchronize(stream));
            (void)hipFree(p->cpk4);
            p->cpk4 = nullptr;
        }
        if (hipMalloc((void**)&p->cpk4, need * sizeof(double)) != hipSuccess) return fail(SHG_ERR_NOMEM, "coefficient workspace allocation failed");
        p->cpk4_size = need;
        SHG_HIP(hipMemsetAsync(p->cpk4, 0, need * sizeof(double), stream));   // sine slots of order 0 stay zero
    }
    const int E = (p->N + 1) * (p->N + 1);
    {
        ProfileScope ps(p, 0, stream);
        hipLaunchKernelGGL(pack_coefficients4_kernel, dim3(ceil_div(E, 256), nbt), dim3(256), 0, stream, p->N, B, anm, p->cpk4);
    }
    FusedParams P;
    P.N = p->N;
    P.nlat = p->nlat;
    P.nlon = p->nlon;
    P.ldlat = p->ldlat;
    P.K = p->K;
    P.ncol = p->ncol;
    P.B = B;
    P.nit = ceil_div(p->nlat, 16);
    P.Ppk = Ppk;
    P.ncb = ceil_div(p->ncoltiles, 8);
    for (int g = 0; g < 5; ++g) P.goff[g] = p->goff[g];
    const int N = p->N;
    const int cnt[4] = {N / 2 + 1, (N + 1) / 2, N / 2, (N + 1) / 2};
    for (int g = 0; g < 4; ++g) P.gcount[g] = cnt[g];
    P.cpk4 = p->cpk4;
    P.pk = p->pk;
    P.trig = p->trig;
    P.G = grid;
    const size_t lds = fused_lds_bytes(p->K, chunk);
    const dim3 grid_dim((unsigned)(nbt * P.nit));
    ProfileScope ps(p, 2, stream);
    if (chunk == 16) {
        SHG_HIP(hipFuncSetAttribute((const void*)synthesis_fused_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(synthesis_fused_kernel<16>, grid_dim, dim3(512), lds, stream, P);
    } else {
        SHG_HIP(hipFuncSetAttribute((const void*)synthesis_fused_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(synthesis_fused_kernel<8>, grid_dim, dim3(512), lds, stream, P);
    }
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

}  // namespace shg
