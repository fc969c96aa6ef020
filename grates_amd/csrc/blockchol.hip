// Sparse block Cholesky on the device: factorisation, triangular solves, sparse (Takahashi) and full inverse and the
// factor / symmetric products of a symmetric positive definite block matrix of which the upper blocks are stored
// ("Kalman smoother" normal equations: grates/lstsq.py:698-882, 950-968, 1026-1042).
//
// The matrix has nb block rows / columns with boundaries bounds[0 .. nb].  The stored blocks (j >= i only) come in compressed
// row form: row i owns the entries rowptr[i] .. rowptr[i + 1] - 1, colidx[e] is the block column (ascending within a row, the
// diagonal block first) and blk[e] the device address of the block, row-major [rows_i][cols_j].  The table is O(stored
// blocks), so a chain of thousands of epochs (BASELINE config 5: 3650) costs nothing to describe.  The caller
// (grates_amd/lstsq.py) owns the blocks, has allocated the fill-in of the factor (symbolic step on the host) and passes one
// scratch matrix `inv[i]` [rows_i][rows_i] per block row that receives U_ii^-1: every triangular solve with a diagonal factor
// block is a GEMM with that inverse.  One C call walks the whole matrix; all block operations are fp64 MFMA GEMMs
// (blas.hip: gemm_ex) and a recursive Cholesky factorisation that yields the inverse of the factor in the same sweep
// (potrf_inverse_upper), enqueued on the caller's stream without host synchronisation.
//
// Formulation (right-looking / outer-product forms; the results equal the reference's left-looking loops up to summation
// order):
//   factor   for r = 0 .. nb-1:  U_rr = chol(A_rr);  W_rc = U_rr^-T A_rc (c > r);  A_cd -= W_rc^T W_rd (r < c <= d)
//   W^T x=b  for r ascending:    x_r = U_rr^-T b_r;  b_c -= W_rc^T x_r (c > r)
//   W x = b  for r descending:   b_r -= sum_c W_rc x_c (c > r);  x_r = U_rr^-1 b_r
//   Z = (W^T W)^-1 on the pattern of W (Takahashi), r descending with T_rk = U_rr^-1 W_rk:
//            Z_rj = -sum_{k > r} T_rk Z_kj (j > r),   Z_rr = U_rr^-1 U_rr^-T - sum_{k > r} T_rk Z_rk^T
#include <algorithm>
#include <cstdint>
#include <vector>

#include "common.h"

namespace shg {

int gemm_ex(bool ta, bool tb, int M, int N, int K, double alpha, const double* A, int lda, long long strideA, const double* B, int ldb,
            long long strideB, double beta, double* C, int ldc, long long strideC, int batch, bool upper_only, hipStream_t stream);
int gemm_ex_tri(bool ta, bool tb, int M, int N, int K, double alpha, const double* A, int lda, long long strideA, const double* B, int ldb,
                long long strideB, double beta, double* C, int ldc, long long strideC, int batch, bool upper_only, int tri, hipStream_t stream);
int potrf_inverse_batch(int n, double* A, int lda, long long strideA, double* X, int ldx, long long strideX, double* work, long long strideW,
                        int* info, int info_stride, int batch, const Coupling* cp, hipStream_t stream);
bool potrf_inverse_carries_coupling(int n, hipStream_t stream);
void potrf_inverse_set_lookahead(int mode);
int potrf_inverse_lookahead_mode();
size_t potrf_inverse_work(int n);

namespace {

struct BlockView {
    int nb;
    const int* bounds;
    const int* rowptr;
    const int* colidx;
    double* const* blk;
    int size(int i) const { return bounds[i + 1] - bounds[i]; }
    int begin(int i) const { return rowptr[i]; }
    int end(int i) const { return rowptr[i + 1]; }
    // block (i, j), j >= i, or NULL
    double* at(int i, int j) const {
        const int* lo = colidx + rowptr[i];
        const int* hi = colidx + rowptr[i + 1];
        const int* it = std::lower_bound(lo, hi, j);
        return (it != hi && *it == j) ? blk[it - colidx] : nullptr;
    }
    int max_size() const {
        int m = 0;
        for (int i = 0; i < nb; ++i) m = std::max(m, size(i));
        return m;
    }
};

// C = alpha op(A) op(B) + beta C for whole blocks (leading dimension = column count of the block)
inline int gemm(bool ta, bool tb, int M, int N, int K, double alpha, const double* A, int lda, const double* B, int ldb, double beta, double* C,
                int ldc, bool upper, hipStream_t s) {
    return gemm_ex(ta, tb, M, N, K, alpha, A, lda, 0, B, ldb, 0, beta, C, ldc, 0, 1, upper, s);
}

// the same with a triangular operand: the inverse of a diagonal factor block is upper triangular, so op(A) = U^-1 is upper (1),
// op(A) = U^-T lower (2), op(B) = U^-T lower (8) -- the products skip the K tiles that are structurally zero
inline int gemm_tri(bool ta, bool tb, int M, int N, int K, double alpha, const double* A, int lda, const double* B, int ldb, double beta, double* C,
                    int ldc, int tri, hipStream_t s) {
    return gemm_ex_tri(ta, tb, M, N, K, alpha, A, lda, 0, B, ldb, 0, beta, C, ldc, 0, 1, false, tri, s);
}

// Workspaces of one call: the stream's own grow-only scratch buffers (ScratchLease, plan.hip), one slot per request in the
// order of the requests; the stream orders the calls that share them.  (Not hipMallocAsync / hipFreeAsync per call: the free
// keeps the calling thread until the stream has caught up, see plan.hip, and a chain of per-row calls -- the segmented
// smoother -- then runs at the pace of the host.)
struct Scratch {
    ScratchLease lease;
    int next = kScratchBlocks;
    explicit Scratch(hipStream_t s) : lease(s) {}
    double* get(size_t doubles) { return (double*)lease.get(next++, std::max<size_t>(doubles, 1) * sizeof(double)); }
};

int check(const BlockView& V, double* const* inv, const char* who) {
    SHG_REQUIRE(V.nb >= 0 && V.bounds && V.rowptr && (V.nb == 0 || (V.colidx && V.blk)), "%s: NULL block table", who);
    SHG_REQUIRE(V.nb == 0 || V.rowptr[0] == 0, "%s: rowptr must start at 0", who);
    SHG_REQUIRE(V.nb == 0 || inv != nullptr, "%s: NULL table of scratch matrices for the inverses of the diagonal factor blocks", who);
    for (int i = 0; i < V.nb; ++i) {
        SHG_REQUIRE(V.size(i) > 0, "%s: empty block row %d", who, i);
        SHG_REQUIRE(V.end(i) > V.begin(i) && V.colidx[V.begin(i)] == i && V.blk[V.begin(i)] != nullptr, "%s: diagonal block %d is missing", who, i);
        for (int e = V.begin(i) + 1; e < V.end(i); ++e)
            SHG_REQUIRE(V.colidx[e] > V.colidx[e - 1] && V.colidx[e] < V.nb && V.blk[e] != nullptr,
                        "%s: block row %d: columns must ascend from the diagonal, inside the matrix, with non-NULL blocks", who, i);
        SHG_REQUIRE(inv[i] != nullptr, "%s: scratch for the inverse of diagonal block %d is missing", who, i);
    }
    return SHG_OK;
}

}  // namespace

__global__ void mirror_upper_kernel(int n, double* __restrict__ C, int ldc);       // blas.hip: C[c][r] = C[r][c] for c > r

__global__ void merge_info_kernel(int* __restrict__ dst, const int* __restrict__ src, int offset) {
    if (*dst == 0 && *src != 0) *dst = *src + offset;
}

}  // namespace shg

using namespace shg;

// A = W^T W in place (upper blocks); inv[r] <- U_rr^-1.  *info (device int, may be NULL): 1-based index of the first
// non-positive pivot (counted over the whole matrix), 0 on success.
// Only the block rows first <= r < last are eliminated: the rows before `first` count as factored already (their updates have
// been applied), the rows from `last` on receive the updates and are left as the Schur complement of what has been
// eliminated.  Two chains that meet in a common last block (the two halves of a block-tridiagonal system, each walked from
// its free end) are factored this way; the caller adds the two complements and finishes with the last row.
// `count` matrices of the same structure (shg_block_potrf_rows_pair: the two chains) go through every launch together, as a
// batch of two whose items lie (second block - first block) apart: the step of one chain is a string of launches that are too
// small to fill the card, and the card overlaps the queues of two or three host threads only in part.
static int potrf_rows(int count, const BlockView* V, double* const* const* blk, double* const* const* inv, int first, int last, int* info,
                      int info_stride, hipStream_t stream) {
    const BlockView& V0 = V[0];
    const int* colidx = V0.colidx;
    const int* bounds = V0.bounds;
    // (doubles between the two items of a batch: the blocks are separate allocations, so the distance is taken between addresses)
    auto apart = [&](const double* const* p) { return count > 1 ? (long long)(((intptr_t)p[1] - (intptr_t)p[0]) / (intptr_t)sizeof(double)) : 0LL; };
    Scratch scratch(stream);
    const int dmax = V0.max_size();
    const size_t wsize = potrf_inverse_work(dmax), bsize = (size_t)dmax * dmax;
    // two work areas in turn, like the copies of the diagonal block below: after a chain row the side stream of the look-ahead is
    // still growing the last block column of that row's inverse THROUGH its work area when the next row starts -- and a next row
    // that does not take the look-ahead (128 < d <= 256, or no chain row) works on the caller's stream
    double* work_of[2] = {scratch.get(wsize * count), scratch.get(wsize * count)};
    double* panel = scratch.get(bsize * count);
    int* info_blk = (int*)scratch.get(count);
    SHG_REQUIRE(work_of[0] && work_of[1] && panel && info_blk, "shg_block_potrf: workspace allocation failed");
    // inv[r] == blk[diagonal r]: the caller keeps U_rr^-1 INSTEAD of U_rr (nothing but shg_block_multiply needs the diagonal
    // factor blocks once their inverses exist): the block is factored in a scratch copy and its inverse goes where it was
    // (two copies in turn: the inverse of row r is still being completed from copy r % 2, on the side stream of the look-ahead,
    //  while row r + 1 is factored in the other one; `inverse_done` says when a copy is free again)
    double* diag_copy[2] = {nullptr, nullptr};
    hipEvent_t inverse_done[2] = {nullptr, nullptr};
    bool pending[2] = {false, false};
    int rc = SHG_OK;
    if ((rc = scratch.lease.event(0, &inverse_done[0])) != SHG_OK || (rc = scratch.lease.event(1, &inverse_done[1])) != SHG_OK) return rc;
    auto join = [&](int which) -> int {            // the caller's stream waits for the inverse that was left growing
        if (pending[which]) SHG_HIP(hipStreamWaitEvent(stream, inverse_done[which], 0));
        pending[which] = false;
        return SHG_OK;
    };
    for (int b = 0; b < count; ++b)
        if (info && (rc = zero_fill(info + (size_t)b * info_stride, stream)) != SHG_OK) return rc;
    for (int r = first; r < last; ++r) {
        const int dr = V0.size(r);
        const int e0 = V0.begin(r), e1 = V0.end(r);
        const int turn = (r - first) & 1;
        double* Arr[2] = {blk[0][e0], count > 1 ? blk[1][e0] : nullptr};
        const double* Xrr[2] = {inv[0][r], count > 1 ? inv[1][r] : nullptr};
        const bool in_place = Xrr[0] == Arr[0];
        SHG_REQUIRE(count == 1 || (Xrr[1] == Arr[1]) == in_place, "shg_block_potrf_rows_pair: block row %d keeps its inverse in place in one matrix only", r);
        if ((rc = join(turn)) != SHG_OK) return rc;
        if (in_place) {
            if (!diag_copy[turn]) diag_copy[turn] = scratch.get(bsize * count);
            SHG_REQUIRE(diag_copy[turn] != nullptr, "shg_block_potrf: workspace allocation failed");
            for (int b = 0; b < count; ++b) {
                SHG_HIP(hipMemcpyAsync(diag_copy[turn] + b * bsize, Arr[b], (size_t)dr * dr * sizeof(double), hipMemcpyDeviceToDevice, stream));
                Arr[b] = diag_copy[turn] + b * bsize;
            }
        }
        for (int b = 0; b < count; ++b)
            if ((rc = zero_fill(info_blk + b, stream)) != SHG_OK) return rc;
        const long long sA = apart(Arr), sX = apart(Xrr);
        // a chain row (one coupling block, to the next block row): W_r,r+1 and the Schur complement of the next diagonal block come
        // out of the panel sweep over the diagonal block itself (Coupling, common.h), and the next row starts while the inverse of
        // this one is still being completed
        const bool chain_row = e1 - e0 == 2 && colidx[e0 + 1] == r + 1 && potrf_inverse_carries_coupling(dr, stream);
        Coupling cp{};
        if (chain_row) {
            const int dc = V0.size(r + 1);
            const double* Wn[2] = {blk[0][e0 + 1], count > 1 ? blk[1][e0 + 1] : nullptr};
            const double* Sn[2] = {V[0].at(r + 1, r + 1), count > 1 ? V[1].at(r + 1, r + 1) : nullptr};
            cp = Coupling{const_cast<double*>(Wn[0]), dc, dc, apart(Wn), const_cast<double*>(Sn[0]), dc, apart(Sn), inverse_done[turn]};
        }
        rc = potrf_inverse_batch(dr, Arr[0], dr, sA, const_cast<double*>(Xrr[0]), dr, sX, work_of[turn], (long long)wsize, info_blk, 1, count, chain_row ? &cp : nullptr,
                                 stream);   // factor and inverse in one sweep
        if (rc) return rc;
        if (info)
            for (int b = 0; b < count; ++b)     // first failure wins
                hipLaunchKernelGGL(merge_info_kernel, dim3(1), dim3(1), 0, stream, info + (size_t)b * info_stride, info_blk + b, bounds[r] - bounds[0]);
        if (chain_row) {
            pending[turn] = true;
            continue;
        }
        // W_rc = U_rr^-T A_rc, through the panel scratch (the product cannot overwrite its own operand)
        for (int e = e0 + 1; e < e1; ++e) {
            const int dc = V0.size(colidx[e]);
            const double* Arc[2] = {blk[0][e], count > 1 ? blk[1][e] : nullptr};
            rc = gemm_ex_tri(true, false, dr, dc, dr, 1.0, Xrr[0], dr, sX, Arc[0], dc, apart(Arc), 0.0, panel, dc, (long long)bsize, count, false, 2, stream);
            if (rc) return rc;
            for (int b = 0; b < count; ++b)
                SHG_HIP(hipMemcpyAsync(blk[b][e], panel + b * bsize, (size_t)dr * dc * sizeof(double), hipMemcpyDeviceToDevice, stream));
        }
        // trailing update A_cd -= W_rc^T W_rd
        for (int e = e0 + 1; e < e1; ++e) {
            const int c = colidx[e];
            for (int f = e; f < e1; ++f) {
                const int d = colidx[f];
                const double* Acd[2] = {V[0].at(c, d), count > 1 ? V[1].at(c, d) : nullptr};
                SHG_REQUIRE(Acd[0] != nullptr && (count == 1 || Acd[1] != nullptr), "shg_block_potrf: fill-in block (%d, %d) was not allocated", c, d);
                const double* Wc[2] = {blk[0][e], count > 1 ? blk[1][e] : nullptr};
                const double* Wd[2] = {blk[0][f], count > 1 ? blk[1][f] : nullptr};
                rc = gemm_ex(true, false, V0.size(c), V0.size(d), dr, -1.0, Wc[0], V0.size(c), apart(Wc), Wd[0], V0.size(d), apart(Wd), 1.0,
                             const_cast<double*>(Acd[0]), V0.size(d), apart(Acd), count, c == d, stream);
                if (rc) return rc;
            }
        }
    }
    if ((rc = join(0)) != SHG_OK || (rc = join(1)) != SHG_OK) return rc;
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

extern "C" int shg_block_potrf_rows(int nb, const int* bounds, const int* rowptr, const int* colidx, double* const* blk, double* const* inv,
                                    int first, int last, int* info, void* stream_) {
    const BlockView V{nb, bounds, rowptr, colidx, blk};
    int rc = check(V, inv, "shg_block_potrf");
    if (rc) return rc;
    SHG_REQUIRE(first >= 0 && first <= last && last <= nb, "shg_block_potrf_rows: rows %d .. %d outside 0 .. %d", first, last, nb);
    return potrf_rows(1, &V, &blk, &inv, first, last, info, 0, (hipStream_t)stream_);
}

// shg_block_potrf_rows for two matrices of the same structure (one block table, two sets of blocks), info[0] and info[1]
extern "C" int shg_block_potrf_rows_pair(int nb, const int* bounds, const int* rowptr, const int* colidx, double* const* blk0, double* const* inv0,
                                         double* const* blk1, double* const* inv1, int first, int last, int* info, void* stream_) {
    const BlockView V[2] = {{nb, bounds, rowptr, colidx, blk0}, {nb, bounds, rowptr, colidx, blk1}};
    int rc = check(V[0], inv0, "shg_block_potrf_rows_pair");
    if (!rc) rc = check(V[1], inv1, "shg_block_potrf_rows_pair");
    if (rc) return rc;
    SHG_REQUIRE(first >= 0 && first <= last && last <= nb, "shg_block_potrf_rows_pair: rows %d .. %d outside 0 .. %d", first, last, nb);
    double* const* blk[2] = {blk0, blk1};
    double* const* inv[2] = {inv0, inv1};
    return potrf_rows(2, V, blk, inv, first, last, info, 1, (hipStream_t)stream_);
}

// The factorisation of a diagonal block takes a look-ahead on two more streams (blas.hip) unless the calling thread turns it
// off: a caller that factors several matrices from several threads at once does better without (and better still with
// shg_block_potrf_rows_pair).  The setting belongs to the calling thread.
extern "C" int shg_block_set_lookahead(int mode) {
    SHG_REQUIRE(mode >= 0 && mode <= 3, "shg_block_set_lookahead: mode %d not in 0 .. 3", mode);
    potrf_inverse_set_lookahead(mode);
    return SHG_OK;
}

extern "C" int shg_block_lookahead_info(void* stream_, int which[4]) {
    SHG_REQUIRE(which != nullptr, "shg_block_lookahead_info: NULL argument");
    hipStream_t stream = (hipStream_t)stream_;
    ScratchLease lease(stream);
    hipStream_t side[2];
    hipEvent_t to_side, from_side[2];
    const int rc = lease.side(side, &to_side, from_side);
    if (rc) return rc;
    which[0] = potrf_inverse_lookahead_mode();
    which[1] = lease.sides_apart();
    which[2] = potrf_inverse_carries_coupling(1 << 20, stream) ? 1 : 0;       // what a chain row of a large block does under this setting
    which[3] = 0;
    return SHG_OK;
}

extern "C" int shg_block_potrf(int nb, const int* bounds, const int* rowptr, const int* colidx, double* const* blk, double* const* inv, int* info,
                               void* stream) {
    return shg_block_potrf_rows(nb, bounds, rowptr, colidx, blk, inv, 0, nb, info, stream);
}

// Solve W x = b (transpose == 0) or W^T x = b (transpose != 0) with the block factor; B [n][k] row-major with leading
// dimension ldb holds b on entry and x on exit.
// Only the block rows first <= r < last take part (shg_block_solve: all).  Forward sweep (transpose): the rows before `first` count
// as done, the rows from `last` on receive the updates b_c -= W_rc^T x_r and are left as the reduced right-hand side of what
// has been eliminated.  Backward sweep: the rows from `last` on hold the solution already (the caller put it there), the rows
// last - 1 .. first are substituted.  A chain that was factored up to its last block row (shg_block_potrf_rows) -- a separator
// shared with its neighbour, whose solution comes from the separator system -- is swept this way.
extern "C" int shg_block_solve_rows(int nb, const int* bounds, const int* rowptr, const int* colidx, double* const* blk, double* const* inv,
                                    int transpose, int first, int last, double* B, int k, int ldb, void* stream_) {
    const BlockView V{nb, bounds, rowptr, colidx, blk};
    int rc = check(V, inv, "shg_block_solve");
    if (rc) return rc;
    SHG_REQUIRE(first >= 0 && first <= last && last <= nb, "shg_block_solve_rows: rows %d .. %d outside 0 .. %d", first, last, nb);
    SHG_REQUIRE(k >= 0, "shg_block_solve: negative number of right-hand sides");
    if (k == 0 || nb == 0) return SHG_OK;
    SHG_REQUIRE(B != nullptr && ldb >= k, "shg_block_solve: bad right-hand side");
    hipStream_t stream = (hipStream_t)stream_;
    Scratch scratch(stream);
    double* tmp = scratch.get((size_t)V.max_size() * k);
    SHG_REQUIRE(tmp != nullptr, "shg_block_solve: workspace allocation failed");
    auto rows = [&](int i) { return B + (size_t)(bounds[i] - bounds[0]) * ldb; };
    if (transpose) {
        for (int r = first; r < last; ++r) {
            const int dr = V.size(r);
            rc = gemm_tri(true, false, dr, k, dr, 1.0, inv[r], dr, rows(r), ldb, 0.0, tmp, k, 2, stream);             // x_r = U_rr^-T b_r
            if (rc) return rc;
            SHG_HIP(hipMemcpy2DAsync(rows(r), (size_t)ldb * sizeof(double), tmp, (size_t)k * sizeof(double), (size_t)k * sizeof(double), dr,
                                     hipMemcpyDeviceToDevice, stream));
            for (int e = V.begin(r) + 1; e < V.end(r); ++e) {
                const int c = colidx[e];
                const double* Wrc = blk[e];
                rc = gemm(true, false, V.size(c), k, dr, -1.0, Wrc, V.size(c), rows(r), ldb, 1.0, rows(c), ldb, false, stream);   // b_c -= W_rc^T x_r
                if (rc) return rc;
            }
        }
    } else {
        for (int r = last - 1; r >= first; --r) {
            const int dr = V.size(r);
            for (int e = V.begin(r) + 1; e < V.end(r); ++e) {
                const int c = colidx[e];
                const double* Wrc = blk[e];
                rc = gemm(false, false, dr, k, V.size(c), -1.0, Wrc, V.size(c), rows(c), ldb, 1.0, rows(r), ldb, false, stream);  // b_r -= W_rc x_c
                if (rc) return rc;
            }
            rc = gemm_tri(false, false, dr, k, dr, 1.0, inv[r], dr, rows(r), ldb, 0.0, tmp, k, 1, stream);             // x_r = U_rr^-1 b_r
            if (rc) return rc;
            SHG_HIP(hipMemcpy2DAsync(rows(r), (size_t)ldb * sizeof(double), tmp, (size_t)k * sizeof(double), (size_t)k * sizeof(double), dr,
                                     hipMemcpyDeviceToDevice, stream));
        }
    }
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

extern "C" int shg_block_solve(int nb, const int* bounds, const int* rowptr, const int* colidx, double* const* blk, double* const* inv,
                               int transpose, double* B, int k, int ldb, void* stream) {
    return shg_block_solve_rows(nb, bounds, rowptr, colidx, blk, inv, transpose, 0, nb, B, k, ldb, stream);
}

// Z = (W^T W)^-1 on the pattern of the factor W held by the blocks, in place (upper blocks); inv[r] = U_rr^-1 on entry.
// Only the block rows last - 1 .. first are processed (shg_block_sparse_inverse: all): the stored blocks of the rows from `last` on
// must hold their entries of the inverse already -- the Takahashi recursion of a row only reads entries of later rows.  For
// a chain whose last block row is a separator shared with a neighbour (its entry of the inverse comes from the separator system).
extern "C" int shg_block_sparse_inverse_rows(int nb, const int* bounds, const int* rowptr, const int* colidx, double* const* blk,
                                             double* const* inv, int first, int last, void* stream_) {
    const BlockView V{nb, bounds, rowptr, colidx, blk};
    int rc = check(V, inv, "shg_block_sparse_inverse");
    if (rc) return rc;
    SHG_REQUIRE(first >= 0 && first <= last && last <= nb, "shg_block_sparse_inverse_rows: rows %d .. %d outside 0 .. %d", first, last, nb);
    hipStream_t stream = (hipStream_t)stream_;
    Scratch scratch(stream);
    const int dmax = V.max_size();
    // T_rk = U_rr^-1 W_rk of the current block row: as many scratch blocks as the densest row has off-diagonal blocks
    int most = 0;
    for (int r = 0; r < nb; ++r) most = std::max(most, V.end(r) - V.begin(r) - 1);
    const int most_slot = std::max(most, 1);                                      // one more scratch block behind the T blocks
    double* tbuf = scratch.get((size_t)(most_slot + 1) * dmax * dmax);
    SHG_REQUIRE(tbuf != nullptr, "shg_block_sparse_inverse: workspace allocation failed");
    auto T = [&](int e, int r) { return tbuf + (size_t)(e - V.begin(r) - 1) * dmax * dmax; };       // scratch of entry e of row r
    for (int r = last - 1; r >= first; --r) {
        const int dr = V.size(r);
        const int e0 = V.begin(r), e1 = V.end(r);
        for (int e = e0 + 1; e < e1; ++e) {
            const int dk = V.size(colidx[e]);
            rc = gemm_tri(false, false, dr, dk, dr, 1.0, inv[r], dr, blk[e], dk, 0.0, T(e, r), dk, 1, stream);
            if (rc) return rc;
            rc = zero_fill(blk[e], dk, dk, dr, stream);
            if (rc) return rc;
        }
        if (inv[r] == blk[e0]) {                                                                                       // (inverse kept in the diagonal block itself)
            double* zrr = T(e0 + 1 + most_slot, r);
            rc = gemm_ex_tri(false, true, dr, dr, dr, 1.0, inv[r], dr, 0, inv[r], dr, 0, 0.0, zrr, dr, 0, 1, true, 9, stream);
            if (rc) return rc;
            SHG_HIP(hipMemcpyAsync(blk[e0], zrr, (size_t)dr * dr * sizeof(double), hipMemcpyDeviceToDevice, stream));
        } else {
            rc = gemm_ex_tri(false, true, dr, dr, dr, 1.0, inv[r], dr, 0, inv[r], dr, 0, 0.0, blk[e0], dr, 0, 1, true, 9, stream);   // Z_rr = U^-1 U^-T ...
        }
        if (rc) return rc;
        // (the diagonal block is symmetric: its upper triangle is computed -- half the flops of its d^3 products -- and mirrored
        //  below, once the row is complete)
        for (int f = e1 - 1; f >= e0; --f) {                                                                           // ... and the row, last block first
            const int j = colidx[f];
            double* Zrj = blk[f];
            for (int e = e0 + 1; e < e1; ++e) {
                const int k = colidx[e];
                const double* Zkj = k <= j ? V.at(k, j) : V.at(j, k);                                                  // Z_kj = Z_jk^T for k > j
                if (!Zkj) continue;
                if (k <= j)
                    rc = gemm(false, false, dr, V.size(j), V.size(k), -1.0, T(e, r), V.size(k), Zkj, V.size(j), 1.0, Zrj, V.size(j), j == r, stream);
                else
                    rc = gemm(false, true, dr, V.size(j), V.size(k), -1.0, T(e, r), V.size(k), Zkj, V.size(k), 1.0, Zrj, V.size(j), j == r, stream);
                if (rc) return rc;
            }
        }
        if (dr > 1) {
            hipLaunchKernelGGL(mirror_upper_kernel, dim3(ceil_div(dr, 32), ceil_div(dr, 32)), dim3(256), 0, stream, dr, blk[e0], dr);
            SHG_HIP(hipGetLastError());
        }
    }
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

extern "C" int shg_block_sparse_inverse(int nb, const int* bounds, const int* rowptr, const int* colidx, double* const* blk, double* const* inv,
                                        void* stream) {
    return shg_block_sparse_inverse_rows(nb, bounds, rowptr, colidx, blk, inv, 0, nb, stream);
}

// Full inverse (W^T W)^-1 from the factor, in place, upper blocks: every block (i, j), j >= i, must be allocated.
//   X = W^-1 by block back substitution (X_jj = U_jj^-1, X_ij = -U_ii^-1 sum_{i < k <= j} W_ik X_kj), then Z_ij = sum_{k >= j} X_ik X_jk^T.
extern "C" int shg_block_inverse(int nb, const int* bounds, const int* rowptr, const int* colidx, double* const* blk, double* const* inv,
                                 void* stream_) {
    const BlockView V{nb, bounds, rowptr, colidx, blk};
    int rc = check(V, inv, "shg_block_inverse");
    if (rc) return rc;
    for (int i = 0; i < nb; ++i) SHG_REQUIRE(inv[i] != blk[rowptr[i]], "shg_block_inverse: needs the diagonal factor blocks (inverse kept in their place)");
    for (int i = 0; i < nb; ++i)                        // ascending columns from the diagonal: a full row has exactly nb - i entries
        SHG_REQUIRE(V.end(i) - V.begin(i) == nb - i, "shg_block_inverse: block row %d is not fully allocated", i);
    hipStream_t stream = (hipStream_t)stream_;
    Scratch scratch(stream);
    const int dmax = V.max_size();
    double* acc = scratch.get((size_t)dmax * dmax);
    SHG_REQUIRE(acc != nullptr, "shg_block_inverse: workspace allocation failed");
    // X = W^-1, column by column from the right; column j only needs the X blocks of column j below row i
    for (int j = nb - 1; j >= 0; --j) {
        const int dj = V.size(j);
        for (int i = j - 1; i >= 0; --i) {
            const int di = V.size(i);
            // acc = sum_{i < k <= j} W_ik X_kj  with X_jj = inv[j] (W_ij itself is consumed here and replaced by X_ij)
            rc = gemm(false, false, di, dj, dj, 1.0, V.at(i, j), dj, inv[j], dj, 0.0, acc, dj, false, stream);
            if (rc) return rc;
            for (int k = i + 1; k < j; ++k) {
                rc = gemm(false, false, di, dj, V.size(k), 1.0, V.at(i, k), V.size(k), V.at(k, j), dj, 1.0, acc, dj, false, stream);     // X_kj already final
                if (rc) return rc;
            }
            rc = gemm(false, false, di, dj, di, -1.0, inv[i], di, acc, dj, 0.0, V.at(i, j), dj, false, stream);
            if (rc) return rc;
        }
    }
    // careful with the order above: column j uses W_ik (k < j) of the columns to its left, which are still untouched factors,
    // and X_kj of its own column.  Now the diagonal blocks become X_jj and Z = X X^T row by row, left to right.
    for (int j = 0; j < nb; ++j)
        SHG_HIP(hipMemcpyAsync(V.at(j, j), inv[j], (size_t)V.size(j) * V.size(j) * sizeof(double), hipMemcpyDeviceToDevice, stream));
    for (int i = 0; i < nb; ++i) {
        const int di = V.size(i);
        for (int j = i; j < nb; ++j) {
            const int dj = V.size(j);
            // Z_ij = sum_{k >= j} X_ik X_jk^T: X_ij (k = j) is read and overwritten by the same product -> through the scratch
            rc = gemm(false, true, di, dj, dj, 1.0, V.at(i, j), dj, V.at(j, j), dj, 0.0, acc, dj, false, stream);
            if (rc) return rc;
            for (int k = j + 1; k < nb; ++k) {
                rc = gemm(false, true, di, dj, V.size(k), 1.0, V.at(i, k), V.size(k), V.at(j, k), V.size(k), 1.0, acc, dj, false, stream);
                if (rc) return rc;
            }
            // rows below i still need X_ij?  No: Z_i'j' with i' > i uses X_i'k only.  But Z_ij' (j' > j) of this row uses X_ik, k >= j' > j: not X_ij.
            SHG_HIP(hipMemcpyAsync(V.at(i, j), acc, (size_t)di * dj * sizeof(double), hipMemcpyDeviceToDevice, stream));
        }
    }
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

// Products with the stored blocks, V [n][k] = op B [n][k]:
//   mode 0: V = W B (upper triangular factor);  mode 1: V_i = W_ji^T B_j of the LAST stored block j <= i (the reference's
//   transposed branch assigns instead of accumulating, grates/lstsq.py:743: kept);  mode 2: V = N B, N symmetric, upper blocks stored
extern "C" int shg_block_multiply(int nb, const int* bounds, const int* rowptr, const int* colidx, double* const* blk, int mode, const double* B,
                                  int k, int ldb, double* Vout, int ldv, void* stream_) {
    const BlockView V{nb, bounds, rowptr, colidx, blk};
    SHG_REQUIRE(nb >= 0 && bounds && rowptr && (nb == 0 || (colidx && blk)), "shg_block_multiply: NULL block table");
    SHG_REQUIRE(mode >= 0 && mode <= 2 && k >= 0, "shg_block_multiply: bad mode / size");
    if (k == 0 || nb == 0) return SHG_OK;
    SHG_REQUIRE(B && Vout && ldb >= k && ldv >= k && B != Vout, "shg_block_multiply: bad operands");
    for (int i = 0; i < nb; ++i)
        for (int e = V.begin(i); e < V.end(i); ++e)
            SHG_REQUIRE(colidx[e] >= i && colidx[e] < nb && (e == V.begin(i) || colidx[e] > colidx[e - 1]) && blk[e] != nullptr,
                        "shg_block_multiply: block row %d: columns must ascend within the upper triangle", i);
    hipStream_t stream = (hipStream_t)stream_;
    const int n = bounds[nb] - bounds[0];
    if (int zrc = zero_fill(Vout, ldv, k, n, stream)) return zrc;
    auto rb = [&](int i) { return B + (size_t)(bounds[i] - bounds[0]) * ldb; };
    auto rv = [&](int i) { return Vout + (size_t)(bounds[i] - bounds[0]) * ldv; };
    int rc = SHG_OK;
    // rows ascending and columns ascending within a row: in mode 1 the LAST stored block (j, i), j <= i, assigns V_i, as upstream
    for (int i = 0; i < nb && !rc; ++i) {
        for (int e = V.begin(i); e < V.end(i) && !rc; ++e) {
            const int j = colidx[e];
            const double* Aij = blk[e];
            if (mode == 1) {
                rc = gemm(true, false, V.size(j), k, V.size(i), 1.0, Aij, V.size(j), rb(i), ldb, 0.0, rv(j), ldv, false, stream);
                continue;
            }
            rc = gemm(false, false, V.size(i), k, V.size(j), 1.0, Aij, V.size(j), rb(j), ldb, 1.0, rv(i), ldv, false, stream);
            if (!rc && mode == 2 && j > i) rc = gemm(true, false, V.size(j), k, V.size(i), 1.0, Aij, V.size(j), rb(i), ldb, 1.0, rv(j), ldv, false, stream);
        }
    }
    if (rc) return rc;
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}
