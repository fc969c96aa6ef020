// Covariance propagation on a regular grid through the separable structure of the synthesis matrix
//   (extension; same result as shg_covprop_diag, which follows the reference's formulation grates/grid.py:817-839).
//
// A row of the synthesis matrix factorises: a_(i,j)[(n, s)] = PK_n,m(s)(theta_i) * t_s(lon_j) with the slot s = rank of a
// coefficient inside its degree (0: order 0, 2m-1: cos m, 2m: sin m).  Hence
//     sigma^2(i, j) = t(j)^T B_i t(j),      B_i[s][s'] = sum_{n, n'} PK_n,s(i) Sigma[(n, s)][(n', s')] PK_n',s'(i)
// and the work drops from 2 M P^2 flops (5.6e14 at d/o 180 / 0.5 deg) to 2 nlat P^2 + 2 nlat nlon S^2 (8.4e11):
//   1. Sigma is permuted once into slot-major order (rows and columns), so that the coefficients of a slot are contiguous;
//   2. G_s' = Sigma''[:, slot s'] PK_s'          one fp64 MFMA GEMM per slot   [P x n_s'] [n_s' x nlat]
//   3. B_i[s][s'] = sum_k PK_s[k][i] G_s'[(s, k)][i]                           (contraction kernel, written per parallel)
//   4. Y_i = B_i T                              one batched GEMM               [S x S] [S x nlon]
//   5. sigma(i, j) = sqrt(sum_s T[s][j] Y_i[s][j])
// Everything is HBM / L2 bound except steps 2 and 4.  Workspace: P^2 + 32 P nlat + nlat S^2 + nlat S nlon doubles.
#include <vector>

#include "common.h"

namespace shg {

int gemm_ex(bool ta, bool tb, int M, int N, int K, double alpha, const double* A, int lda, long long strideA, const double* B, int ldb,
            long long strideB, double beta, double* C, int ldc, long long strideC, int batch, bool upper_only, hipStream_t stream);   // blas.hip
int covprop_build_cs_table(shg_plan* p, hipStream_t stream);   // gemm.hip

constexpr int kSepSlotChunk = 32;      // slots whose G matrices are alive at one time

constexpr int kPermSegment = 8192;     // doubles of a source row staged in LDS at one time (64 KB: two workgroups per CU)

// out[a][b] = cov[perm[a]][perm[b]]; one workgroup per output row.  The source row is read in contiguous segments into LDS,
// the gather happens there: HBM sees coalesced reads and writes only (a gather straight from memory touches one 64-byte
// sector per element: 8.8 ms for 8.6 GB at d/o 180).  `pairs` lists, segment by segment, the output positions b whose source
// perm[b] lies in the segment as (b, perm[b] - segment start), b ascending: slot-major order makes them runs of consecutive b.
__global__ __launch_bounds__(1024) void covsep_permute_kernel(int Pn, const int* __restrict__ perm, const int2* __restrict__ pairs,
                                                              const int* __restrict__ pair_off, const double* __restrict__ cov,
                                                              double* __restrict__ out) {
    __shared__ double seg[kPermSegment];
    const int a = blockIdx.x;
    const double* src = cov + (size_t)perm[a] * Pn;
    double* dst = out + (size_t)a * Pn;
    for (int s0 = 0, k = 0; s0 < Pn; s0 += kPermSegment, ++k) {
        const int len = min(kPermSegment, Pn - s0);
        for (int e = threadIdx.x; e < len; e += 1024) seg[e] = __builtin_nontemporal_load(src + s0 + e);
        __syncthreads();
        const int e1 = pair_off[k + 1];
        for (int e = pair_off[k] + threadIdx.x; e < e1; e += 1024) {
            const int2 bq = pairs[e];
            dst[bq.x] = seg[bq.y];
        }
        __syncthreads();
    }
}

// Bm[i][s][sc0 + c] = sum_k PK_s[k][i] G_c[(soff[s] + k)][i] for a tile of 32 parallels x 32 slots of the chunk.
// Reads run along the parallels (contiguous in both tables), the result tile is transposed through LDS so that the writes
// run along the slots.
// half != 0 (symmetric Sigma): G_c holds only the rows of the slots >= c (the shorter ones: a quarter of the G traffic of
// the general case); the entries s > c are written twice their value, the entries s < c as zero, so that t^T Bm t is unchanged.
__global__ __launch_bounds__(256) void covsep_contract_kernel(int N, int nmin, int Pn, int nb, int ldlat, int lat0, int S, int sc0, int nsc,
                                                              int half, const int* __restrict__ soff, const double* __restrict__ pk,
                                                              const double* __restrict__ G, double* __restrict__ Bm) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
    const int i = blockIdx.x * 32 + tx;                            // parallel inside the band
    const int s = blockIdx.y;                                      // row slot
    const int c0 = blockIdx.z * 32;                                // first column slot (inside the chunk) of this tile
    const int m = (s + 1) >> 1;
    const int n0 = max(m, nmin);
    const int ns = N - n0 + 1;                                     // coefficients of slot s (may be <= 0)
    const int ic = min(i, nb - 1);
    const double* pkrow = pk + ((size_t)(order_offset(N, m) + n0 - m)) * ldlat + lat0 + ic;
    const size_t slab = (size_t)Pn * nb;                           // one G matrix
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const double* g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = G + (size_t)min(c0 + ty + 8 * q, nsc - 1) * slab + (size_t)soff[s] * nb + ic;
    const bool skip = half && s < sc0 + c0;                        // the whole tile lies above the diagonal
    for (int k = 0; k < (skip ? 0 : ns); ++k) {
        const double pv = pkrow[(size_t)k * ldlat];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = fma(pv, g[q][(size_t)k * nb], acc[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = sc0 + c0 + ty + 8 * q;
        tile[tx][ty + 8 * q] = !half || s == c ? acc[q] : (s > c ? 2.0 * acc[q] : 0.0);        // rows s < c of G_c do not exist
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ii = blockIdx.x * 32 + ty + 8 * q;               // parallel
        const int c = c0 + tx;                                     // column slot inside the chunk
        if (ii < nb && c < nsc) Bm[((size_t)ii * S + s) * S + sc0 + c] = tile[ty + 8 * q][tx];
    }
}

// sigma[i][j] = sqrt(sum_s T[s][j] Y[i][s][j])
__global__ __launch_bounds__(256) void covsep_reduce_kernel(int S, int nlon, int nb, const double* __restrict__ cs, const double* __restrict__ Y,
                                                            double* __restrict__ sigma) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= nlon) return;
    const double* y = Y + (size_t)i * S * nlon + j;
    double acc = 0.0;
    for (int s = 0; s < S; ++s) acc = fma(cs[(size_t)s * nlon + j], y[(size_t)s * nlon], acc);
    sigma[(size_t)i * nlon + j] = sqrt(acc);
}

}  // namespace shg

using namespace shg;

static int covprop_diag_separable(shg_plan* p, const double* cov, int nmin, int lat0, int lat1, double* sigma, void* stream_, bool half) {
    SHG_REQUIRE(p != nullptr, "shg_covprop_diag_separable: NULL plan");
    SHG_REQUIRE(nmin >= 0 && nmin <= p->N + 1, "shg_covprop_diag_separable: min_degree %d out of range", nmin);
    SHG_REQUIRE(lat0 >= 0 && lat1 <= p->nlat && lat0 <= lat1, "shg_covprop_diag_separable: bad band [%d, %d)", lat0, lat1);
    const int nb = lat1 - lat0;
    if (nb == 0) return SHG_OK;
    SHG_REQUIRE(sigma != nullptr, "shg_covprop_diag_separable: NULL output");
    hipStream_t stream = (hipStream_t)stream_;
    PlanGuard guard(p, stream);
    const int N = p->N, S = 2 * N + 1, nlon = p->nlon;
    const int Pn = (N + 1) * (N + 1) - nmin * nmin;
    if (Pn == 0) {
        SHG_HIP(hipMemsetAsync(sigma, 0, (size_t)nb * nlon * sizeof(double), stream));
        return SHG_OK;
    }
    SHG_REQUIRE(cov != nullptr, "shg_covprop_diag_separable: NULL covariance");
    int rc = build_pk_table(p, stream);
    if (rc) return rc;
    rc = covprop_build_cs_table(p, stream);
    if (rc) return rc;

    // slot-major order of the degree-wise coefficients
    std::vector<int> soff(S + 1, 0), perm((size_t)Pn);
    for (int s = 0; s < S; ++s) {
        const int m = (s + 1) >> 1, n0 = std::max(m, nmin);
        const int ns = std::max(N - n0 + 1, 0);
        for (int k = 0; k < ns; ++k) perm[(size_t)soff[s] + k] = (n0 + k) * (n0 + k) - nmin * nmin + s;
        soff[s + 1] = soff[s] + ns;
    }
    // output positions grouped by the segment of the source row they come from
    const int nseg = ceil_div(Pn, kPermSegment);
    std::vector<int> pair_off(nseg + 1, 0);
    std::vector<int2> pairs((size_t)Pn);
    for (int b = 0; b < Pn; ++b) ++pair_off[perm[b] / kPermSegment + 1];
    for (int k = 0; k < nseg; ++k) pair_off[k + 1] += pair_off[k];
    {
        std::vector<int> fill(pair_off.begin(), pair_off.end() - 1);
        for (int b = 0; b < Pn; ++b) {
            const int k = perm[b] / kPermSegment;
            pairs[(size_t)fill[k]++] = make_int2(b, perm[b] - k * kPermSegment);
        }
    }
    int *soff_d = nullptr, *perm_d = nullptr, *pair_off_d = nullptr;
    int2* pairs_d = nullptr;
    double *Sp = nullptr, *G = nullptr, *Bm = nullptr, *Y = nullptr;
    if (workspace_alloc((void**)&soff_d, soff.size() * sizeof(int), stream) != hipSuccess ||
        workspace_alloc((void**)&perm_d, perm.size() * sizeof(int), stream) != hipSuccess ||
        workspace_alloc((void**)&pairs_d, pairs.size() * sizeof(int2), stream) != hipSuccess ||
        workspace_alloc((void**)&pair_off_d, pair_off.size() * sizeof(int), stream) != hipSuccess ||
        workspace_alloc((void**)&Sp, (size_t)Pn * Pn * sizeof(double), stream) != hipSuccess ||
        workspace_alloc((void**)&G, (size_t)kSepSlotChunk * Pn * nb * sizeof(double), stream) != hipSuccess ||
        workspace_alloc((void**)&Bm, (size_t)nb * S * S * sizeof(double), stream) != hipSuccess ||
        workspace_alloc((void**)&Y, (size_t)nb * S * nlon * sizeof(double), stream) != hipSuccess)
        return fail(SHG_ERR_NOMEM, "shg_covprop_diag_separable: workspace allocation failed (%.1f GB)",
                    ((double)Pn * Pn + (double)kSepSlotChunk * Pn * nb + (double)nb * S * S + (double)nb * S * nlon) * 8e-9);
    SHG_HIP(hipMemcpyAsync(soff_d, soff.data(), soff.size() * sizeof(int), hipMemcpyHostToDevice, stream));
    SHG_HIP(hipMemcpyAsync(perm_d, perm.data(), perm.size() * sizeof(int), hipMemcpyHostToDevice, stream));
    SHG_HIP(hipMemcpyAsync(pairs_d, pairs.data(), pairs.size() * sizeof(int2), hipMemcpyHostToDevice, stream));
    SHG_HIP(hipMemcpyAsync(pair_off_d, pair_off.data(), pair_off.size() * sizeof(int), hipMemcpyHostToDevice, stream));
    SHG_HIP(hipStreamSynchronize(stream));                              // the host vectors go out of scope at return
    hipLaunchKernelGGL(covsep_permute_kernel, dim3(Pn), dim3(1024), 0, stream, Pn, perm_d, pairs_d, pair_off_d, cov, Sp);

    {
        ProfileScope ps(p, 3, stream);
        for (int sc0 = 0; sc0 < S && rc == SHG_OK; sc0 += kSepSlotChunk) {
            const int nsc = std::min(kSepSlotChunk, S - sc0);
            for (int c = 0; c < nsc && rc == SHG_OK;) {
                const int s = sc0 + c, m = (s + 1) >> 1, n0 = std::max(m, nmin), ns = N - n0 + 1;
                // the cosine and the sine slot of an order have the same coefficients count and the same PK rows: one batched launch
                const int pair = (s >= 1 && (s & 1) == 1 && c + 1 < nsc) ? 2 : 1;
                double* Gc = G + (size_t)c * Pn * nb;
                const int row0 = half ? soff[s] : 0;                      // symmetric Sigma: only the rows of this slot and the later ones
                if (ns <= 0)
                    SHG_HIP(hipMemsetAsync(Gc, 0, (size_t)pair * Pn * nb * sizeof(double), stream));
                else
                    rc = gemm_ex(false, false, Pn - row0, nb, ns, 1.0, Sp + (size_t)row0 * Pn + soff[s], Pn, ns,
                                 p->pk + (size_t)(order_offset(N, m) + n0 - m) * p->ldlat + lat0, p->ldlat, 0, 0.0, Gc + (size_t)row0 * nb, nb,
                                 (long long)Pn * nb, pair, false, stream);
                c += pair;
            }
            if (rc) break;
            hipLaunchKernelGGL(covsep_contract_kernel, dim3(ceil_div(nb, 32), S, ceil_div(nsc, 32)), dim3(256), 0, stream, N, nmin, Pn, nb, p->ldlat,
                               lat0, S, sc0, nsc, half ? 1 : 0, soff_d, p->pk, G, Bm);
        }
        if (rc == SHG_OK)
            rc = gemm_ex(false, false, S, nlon, S, 1.0, Bm, S, (long long)S * S, p->cs_slot, nlon, 0, 0.0, Y, nlon, (long long)S * nlon, nb, false, stream);
        if (rc == SHG_OK)
            hipLaunchKernelGGL(covsep_reduce_kernel, dim3(ceil_div(nlon, 256), nb), dim3(256), 0, stream, S, nlon, nb, p->cs_slot, Y, sigma);
    }
    (void)hipFreeAsync(soff_d, stream);
    (void)hipFreeAsync(perm_d, stream);
    (void)hipFreeAsync(pairs_d, stream);
    (void)hipFreeAsync(pair_off_d, stream);
    (void)hipFreeAsync(Sp, stream);
    (void)hipFreeAsync(G, stream);
    (void)hipFreeAsync(Bm, stream);
    (void)hipFreeAsync(Y, stream);
    if (rc) return rc;
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

extern "C" int shg_covprop_diag_separable(shg_plan* p, const double* cov, int nmin, int lat0, int lat1, double* sigma, void* stream) {
    return covprop_diag_separable(p, cov, nmin, lat0, lat1, sigma, stream, false);
}

// Sigma symmetric (not checked: shg_symmetry_defect): B_i is symmetric as well, only the slot pairs s >= s' are formed --
// half of the GEMM work and a quarter of the G traffic.  Reads the entries (p, q) of Sigma with slot(p) >= slot(q).
extern "C" int shg_covprop_diag_separable_symmetric(shg_plan* p, const double* cov, int nmin, int lat0, int lat1, double* sigma, void* stream) {
    return covprop_diag_separable(p, cov, nmin, lat0, lat1, sigma, stream, true);
}
