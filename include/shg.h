/*
 * shg.h -- C ABI of libshg: spherical-harmonic synthesis / analysis / covariance propagation / filter
 *          kernels for AMD MI355X (gfx950, CDNA4).
 *
 * This is the drop-in boundary for the hot path of akvas/grates.  The reference is
 * pure Python and has no FFI of its own; every entry point below names the reference function whose inner
 * loop it replaces (paths relative to the reference checkout).  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative shg_status; shg_last_error() gives the message of
 *     the last failure on the calling thread.  No exception crosses the boundary.
 *   - pointers named *_h are HOST pointers (read during the call, not retained); all other array
 *     pointers are DEVICE pointers to contiguous fp64 / int arrays owned by the caller
 *     (e.g. torch.Tensor.data_ptr()).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are asynchronous with
 *     respect to the host; a plan may have only one operation in flight at a time (it owns workspace).
 *   - coefficient arrays use the reference packing  anm[n][m] = C_nm (m <= n),  anm[m-1][n] = S_nm (m >= 1)
 *     (grates/gravityfield.py:156-159); "degree-wise" vectors use  index(C,n,0) = n^2 - nmin^2,
 *     index(C,n,m) = n^2 + 2m - 1 - nmin^2,  index(S,n,m) = n^2 + 2m - nmin^2  (grates/utilities.py:331-343).
 *   - grids are [nlat][nlon] row-major, parallels north to south (grates/grid.py:609-625, 1146-1151).
 */
#ifndef SHG_H
#define SHG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct shg_plan shg_plan;

typedef enum {
    SHG_OK = 0,
    SHG_ERR_INVALID = -1,   /* bad argument (NULL pointer, negative size, degree mismatch ...) */
    SHG_ERR_HIP = -2,       /* a HIP runtime call failed                                        */
    SHG_ERR_NOMEM = -3,     /* workspace allocation failed                                      */
    SHG_ERR_UNSUPPORTED = -4
} shg_status;

/* ------------------------------------------------------------------------------------------------
 * Plan: per-(degree, parallels, kernel table, meridians) tables held on the device.
 *   replaces the per-call table rebuild of PotentialCoefficients.to_grid
 *   (grates/gravityfield.py:353-365: colatitude/radius -> kn, legendre_functions, trigonometric_functions).
 *
 *   N        maximum degree
 *   colat_h  [nlat] geocentric colatitude of every parallel                  (grates/utilities.py:438-459)
 *   kn_h     [nlat][N+1] per-parallel degree factors (1/k_n)(R/r)^(n+1) GM/R  (grates/gravityfield.py:356)
 *   lon_h    [nlon] longitude of every meridian
 * ------------------------------------------------------------------------------------------------ */
int shg_plan_create(shg_plan** out, int N, int nlat, const double* colat_h, const double* kn_h,
                    int nlon, const double* lon_h, int device);
int shg_plan_destroy(shg_plan* plan);

/* Number of epochs processed per internal pass (workspace is sized for it).  Default 16. */
int shg_plan_set_chunk(shg_plan* plan, int epochs_per_pass);

/* Synthesis path: 0 = automatic, 1 = three-kernel path (pack, Legendre stage, longitude stage; any grid, any degree),
 * 2 = single fused kernel (4-fold symmetric meridians, degree <= 126), 5 = fused kernel with 32-row panels (both symmetries,
 * degree <= ~210; chosen automatically above degree 126), 6 = single fused kernel for equi-angular cell-centred meridians that
 * folds the longitude stage over 6 (nlon a multiple of 96) or 3 (nlon a multiple of 48) rotations and the reflection of the
 * meridian set (degree <= ~110; the automatic choice where it applies).  Kernels 2 and 6 use the north-south symmetry of the
 * parallels when the grid has it and their plain variant otherwise; the variant is not a choice of the caller. */
int shg_plan_set_path(shg_plan* plan, int path);

/* Kernel 6 only: at most `limit` workgroups run their Legendre stage (in which they do not store) at the same time; limit < 0 = that many
 * sixteenths of the device's CUs, 0 = no limit (the default).  A tuning knob: on MI355X -7 took 1.5-2 % off the d/o-96 / 0.25 degree
 * kernel on some boxes and cost 2-4 % on others.  Waits for the device the first time a limit is set. */
int shg_plan_set_stage_limit(shg_plan* plan, int limit);

/* Rotation count R of kernel 6: the longitude sums are evaluated on nlon / (2 R) columns and the 2 R images of every column are
 * formed in registers.  0 = the plan's own choice (the default: 10 where nlon / 10 is a multiple of 16 -- the 0.25 degree grid --,
 * else 9 -- the 0.5 degree grid --, else 6, else 3), or one of 3, 6, 9, 10; the meridians must be equi-angular and cell-centred with nlon a multiple of 2 R and
 * nlon / R a multiple of 16, and the panel of that count must fit the LDS.  Waits for the device (the tables are rebuilt).
 * shg_plan_info reports the count in use in bits 8.. of which[7]. */
int shg_plan_set_rotations(shg_plan* plan, int R);

/* Introspection: which[0]=N, [1]=nlat, [2]=nlon, [3]=bit 0: 4-fold longitude symmetry, bit 1: parallels symmetric about the
 * equator, bit 2: the rotation-folded kernel (path 6) applies, [4]=epochs per pass, [5]=K slots of the longitude stage, [6]=1 if synthesis uses the fused kernel, [7]=path | rotation count of kernel 6 << 8. */
int shg_plan_info(const shg_plan* plan, int64_t which[8]);

/* Per-kernel timing with HIP events recorded on the caller's stream around every kernel a plan launches.
 * kinds: 0 pack_coefficients, 1 legendre_stage, 2 lon_stage, 3 covprop, 4 analysis_lon, 5 analysis_solve.
 * enable: 0 = off, 1 = every kind, otherwise a mask of kinds shifted by one (bit k + 1 = kind k; an event pair costs the stream ~5 us).
 * shg_plan_profile_read synchronises the recorded events, returns accumulated milliseconds and launch
 * counts per kind since the last read, and resets the accumulators. */
#define SHG_PROFILE_KINDS 8
int shg_plan_profile(shg_plan* plan, int enable);
int shg_plan_profile_read(shg_plan* plan, double ms[SHG_PROFILE_KINDS], int64_t launches[SHG_PROFILE_KINDS]);

/* ------------------------------------------------------------------------------------------------
 * Synthesis  V[b][i][j] = sum_nm kn[i][n] P_nm(theta_i) (C_nm[b] cos m lon_j + S_nm[b] sin m lon_j)
 *   replaces PotentialCoefficients.to_grid, regular-grid branch   (grates/gravityfield.py:352-368)
 *   anm  [B][N+1][N+1]      grid  [B][nlat][nlon]
 * ------------------------------------------------------------------------------------------------ */
int shg_synthesis(shg_plan* plan, const double* anm, int B, double* grid, void* stream);

/* Point-list synthesis, one thread block per block of points
 *   replaces PotentialCoefficients.to_grid, AttributeError branch  (grates/gravityfield.py:370-388)
 *   colat, lon [npts]; kn [npts][N+1]; anm [B][N+1][N+1]; values [B][npts]                            */
int shg_synthesis_points(int N, const double* colat, const double* lon, const double* kn, int npts,
                         const double* anm, int B, double* values, void* stream);

/* RMS over the epochs of a batch of grids, the reduction of gridded_rms (grates/gravityfield.py:1164-1170):
 *   acc[i] = (accumulate ? acc[i] : 0) + sum_b values[b][i]^2   (epoch order, no FMA);   count > 0: acc[i] = sqrt(acc[i] / count).
 * values [B][M] (grids of a batch, flattened), acc [M].  Batches are chained with accumulate = 1, the last call passes the
 * number of epochs as count. */
int shg_epoch_rms(const double* values, int B, long long M, int accumulate, long long count, double* acc, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Parity hooks for the table functions
 *   shg_legendre          utilities.legendre_functions            (grates/utilities.py:13-59)
 *                         colat [k] -> pnm [k][N+1][N+1] (packed, sine slots mirrored)
 *   shg_legendre_order    utilities.legendre_functions_per_order   (grates/utilities.py:62-115)
 *                         colat [k] -> pm [k][N+1-m]   (uses s = sqrt(1 - t^2))
 *   shg_trigonometric     utilities.trigonometric_functions        (grates/utilities.py:249-275)
 *                         lon [k] -> cs [k][N+1][N+1]
 * ------------------------------------------------------------------------------------------------ */
int shg_legendre(int N, const double* colat, int k, double* pnm, void* stream);
int shg_legendre_order(int N, int m, const double* colat, int k, double* pm, void* stream);
int shg_trigonometric(int N, const double* lon, int k, double* cs, void* stream);

/* Operator block of one order m (RegularGrid.synthesis_matrix_per_order, grates/grid.py:627-663, and
 * IrregularGrid.synthesis_matrix_per_order, grates/grid.py:957-991): columns = degrees max(m, nmin) .. N,
 *   out_cos[row][c] = kn[i][n] P_nm(colat_i) cos(m lon_j),   out_sin likewise with sin   (n = max(m, nmin) + c),
 * with P_nm from the per-order recursion (s = sqrt(1 - t^2), like shg_legendre_order).
 *   pointwise = 0: regular grid, colat / kn rows of the nlat parallels, lon of the nlon meridians, row = i * nlon + j;
 *   pointwise = 1: point list, colat / lon / kn rows of the nlat points (nlon is ignored), row = i.
 * m = 0: only out_cos is written (cos 0 = 1); out_sin may be NULL. */
int shg_synthesis_matrix_order(int N, int m, int nmin, const double* colat, int nlat, const double* lon, int nlon, const double* kn,
                               int pointwise, double* out_cos, double* out_sin, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Degree-wise ravel / unravel of batches (integer index maps applied on the device)
 *   replaces utilities.ravel_coefficients / unravel_coefficients   (grates/utilities.py:310-411)
 *   and TimeSeries.to_array                                        (grates/gravityfield.py:964-980)
 *   arr [B][Na+1][Na+1]  <->  vec [B][(nmax+1)^2 - nmin^2]; degrees > Na read as zero / are dropped.
 * ------------------------------------------------------------------------------------------------ */
int shg_ravel(const double* arr, int B, int Na, int nmin, int nmax, double* vec, void* stream);
int shg_unravel(const double* vec, int B, int nmin, int nmax, double* arr /* [B][nmax+1][nmax+1], zero-filled */, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Covariance propagation  sigma[i*nlon + j] = sqrt(a_ij^T Sigma a_ij),  a_ij = row of the synthesis matrix
 *   replaces RegularGrid.covariance_propagation                     (grates/grid.py:817-839)
 *   cov    [P][P] degree-wise order, P = (N+1)^2 - nmin^2 (only N == plan degree is accepted)
 *   sigma  [(lat1-lat0)*nlon]  for the band of parallels lat0 <= i < lat1 (latitude-band sharding)
 *   The A rows are generated on the fly (never materialised); A*Sigma runs on v_mfma_f64_16x16x4_f64.
 * ------------------------------------------------------------------------------------------------ */
int shg_covprop_diag(shg_plan* plan, const double* cov, int nmin, int lat0, int lat1, double* sigma, void* stream);
/* The same for a symmetric Sigma of which only the upper triangle is read: half of the MFMA work
 *   sigma2 = sum_c [ sum_{p<c} 2 a_p Sigma_pc + a_c Sigma_cc ] a_c
 * (an extension; the reference multiplies with the full matrix, grates/grid.py:833).  shg_symmetry_defect writes
 * max |S[p][c] - S[c][p]| to *defect (device double): 0 means the shortcut reproduces the general result up to summation order. */
int shg_covprop_diag_symmetric(shg_plan* plan, const double* cov, int nmin, int lat0, int lat1, double* sigma, void* stream);
int shg_symmetry_defect(const double* S, int n, int ld, double* defect, void* stream);
/* The same result through the separable structure of the synthesis matrix (an extension, see csrc/covsep.hip):
 *   sigma^2(i, j) = t(j)^T B_i t(j),  B_i[s][s'] = sum_{n,n'} PK_n,s(i) Sigma[(n,s)][(n',s')] PK_n',s'(i)
 * 2 nlat P^2 flops instead of 2 nlat nlon P^2 (d/o 180, 0.5 deg: 8e11 instead of 5.6e14); differs from shg_covprop_diag by
 * summation order only.  Workspace: about P^2 + 32 P nlat doubles (17 GB at d/o 180 for the full grid). */
int shg_covprop_diag_separable(shg_plan* plan, const double* cov, int nmin, int lat0, int lat1, double* sigma, void* stream);
/* cov symmetric (caller's promise, cf. shg_symmetry_defect): B_i is symmetric too and only its slot pairs s >= s' are formed,
 * half the work of shg_covprop_diag_separable. */
int shg_covprop_diag_separable_symmetric(shg_plan* plan, const double* cov, int nmin, int lat0, int lat1, double* sigma, void* stream);

/* Point-list variant  (grates/grid.py:1096-1120): colat, lon [npts]; kn [npts][N+1]; sigma [npts] */
int shg_covprop_points(int N, const double* colat, const double* lon, const double* kn, int npts,
                       const double* cov, int nmin, double* sigma, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Filters
 *   shg_degree_scale      Gaussian.filter / Butterworth.filter degree-wise scaling (grates/filter.py:61-72)
 *                         anm_out[b][slot of degree n] = w[n] * anm_in   for n >= nfirst, copied below
 *   shg_orderwise_filter  OrderWiseFilter.filter                      (grates/filter.py:175-191)
 *                         blocks packed back to back in the reference list order
 *                         [order0_cos, order1_cos, order1_sin, ...], block k row-major with leading
 *                         dimension block_dim[k] = Nb+1-m; block_off [2Nb+1] offsets in doubles.
 *                         Coefficients of degree N <= Nb are filtered with the leading (N+1-m)^2 sub-block;
 *                         degrees 0 and 1 are restored from the input.
 *   shg_dense_filter      GeneralMatrix.filter core  Y = W X           (grates/filter.py:473-474)
 *                         W [P][P], X [P][T] (epoch fastest), Y [P][T]; fp64 MFMA GEMM.
 * ------------------------------------------------------------------------------------------------ */
int shg_degree_scale(const double* w /* [N+1] device */, int N, int nfirst, const double* anm_in, int B,
                     double* anm_out, void* stream);
int shg_orderwise_filter(const double* blocks_packed, const int64_t* block_off, int Nb, int N,
                         const double* anm_in, int B, double* anm_out, void* stream);
int shg_dense_filter(const double* W, int P, const double* X, int T, double* Y, void* stream);

/* Order-major series: a time series of coefficient sets kept on the device between operators -- the batching the reference does with
 * TimeSeries.to_array (grates/gravityfield.py:964-980), in the layout the order-wise operators want:
 *   om [(N+1)^2 rows][Bpad]   epochs fastest (Bpad = B rounded up to a multiple of 32), row of (slot s, k = n - m) = first_row(s) + k with
 *   the slots in the order of the DDK block list (s = 0: order 0 cosine, 2m - 1: order m cosine, 2m: order m sine; grates/filter.py:153-191).
 *   shg_order_major_pack / _unpack   from / to the reference arrays anm [B][N+1][N+1]
 *   shg_orderwise_filter_om          OrderWiseFilter.filter of all epochs: Y_s = W_s X_s per slot on whole matrices (no gather / scatter);
 *                                    degrees 0 and 1 keep the input, N <= Nb as in shg_orderwise_filter
 *   shg_degree_scale_om              shg_degree_scale (Gaussian / Butterworth) of all epochs of a series
 *   shg_synthesis_om                 shg_synthesis of a series of degree Ns >= the plan's N (higher degrees are not read); fused kernels on
 *                                    parallels symmetric about the equator only (SHG_ERR_INVALID otherwise: unpack the series) */
int shg_order_major_pack(const double* anm, int N, int B, double* om, int Bpad, void* stream);
int shg_order_major_unpack(const double* om, int N, int B, int Bpad, double* anm, void* stream);
int shg_orderwise_filter_om(const double* blocks_packed, const int64_t* block_off, int Nb, int N, const double* om_in, int B, int Bpad,
                            double* om_out, void* stream);
int shg_degree_scale_om(const double* w /* [N+1] device */, int N, int nfirst, const double* om_in, int B, int Bpad, double* om_out, void* stream);
int shg_synthesis_om(shg_plan* plan, const double* om, int Ns, int B, int Bpad, double* grid, void* stream);

/* DDK block construction  W_k = (N_k + diag(w[m:]))^-1 N_k  for all 2Nb+1 order-wise normal blocks
 *   replaces the dense solves of DDK.__init__ / DDKGeneric.__init__      (grates/filter.py:252-255, 344-347)
 *   normals_packed / blocks_out / work: blocks back to back as in shg_orderwise_filter (work: scratch of the
 *   same size); weights [Nb+1] device array (w_0 = 1, w_n = scale n^4). */
int shg_ddk_blocks(const double* normals_packed, const int64_t* block_off, int Nb, const double* weights, double* work,
                   double* blocks_out, void* stream);

/* X = A^-1 B for a symmetric positive definite A [n][n], B / X [n][k] (Cholesky on the device)
 *   replaces the normal-equation solve of IrregularGrid.analysis_matrix   (grates/grid.py:1015-1017) */
int shg_spd_solve(const double* A, int n, const double* B, int k, double* X, void* stream);

/* General fp64 MFMA GEMM  C[M][N] = A[M][K] B[K][N]  (row-major, leading dimensions in elements). */
int shg_dgemm(int M, int N, int K, const double* A, int lda, const double* B, int ldb, double* C, int ldc, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Dense block operations of the block-banded normal-equation solver ("Kalman smoother", grates/lstsq.py:698-883).
 * The reference loops over the non-zero blocks of a BlockMatrix in Python and calls NumPy / SciPy LAPACK per block;
 * the replacement keeps those loops on the host and runs every block operation on the device:
 *   shg_gemm    C = alpha op(A) op(B) + beta C, row-major, transX != 0 -> operand stored transposed
 *               replaces the `@` products of blocks          (grates/lstsq.py:679, 711, 743-748, 770-774, 805, 815, 846, 860-882)
 *   shg_potrf   A = U^T U in place, upper triangle referenced, strictly lower triangle zeroed on exit; *info (device int,
 *               may be NULL) receives the 1-based index of the first non-positive pivot, 0 on success
 *               replaces scipy.linalg.cholesky(lower=False)  (grates/lstsq.py:713; numpy.linalg.cholesky lstsq.py:197)
 *   shg_trtri   X = U^-1 for an upper triangular U (X != U; strictly lower triangle of X zeroed).  Triangular solves
 *               with a factor block are GEMMs with this inverse
 *               replaces scipy.linalg.solve_triangular / inv (grates/lstsq.py:716, 807, 817, 835, 839, 856, 868)
 * ------------------------------------------------------------------------------------------------ */
int shg_gemm(int transa, int transb, int M, int N, int K, double alpha, const double* A, int lda, const double* B, int ldb,
             double beta, double* C, int ldc, void* stream);
int shg_potrf(int n, double* A, int lda, int* info, void* stream);
/* Y = alpha X + beta Y on [rows][cols] blocks: _scale / _axpy of blocks and vectors (grates/lstsq.py:889-903, 1107-1117) */
int shg_axpby(int rows, int cols, double alpha, const double* X, int ldx, double beta, double* Y, int ldy, void* stream);
int shg_transpose_in_place(int n, double* A, int lda, void* stream);      /* A <- A^T, square, in its own storage */
int shg_trtri(int n, const double* U, int ldu, double* X, int ldx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Sparse block Cholesky of the block-banded normal equations ("Kalman smoother"), one call per operation: the blocks stay in
 * HBM, the call walks them on the device (csrc/blockchol.hip).  nb block rows / columns with boundaries bounds[0..nb]; the
 * stored blocks (j >= i) in compressed row form: entries rowptr[i] .. rowptr[i+1]-1 of row i, colidx[e] ascending from the
 * diagonal, blk[e] = device address of block (i, colidx[e]) (row-major [rows_i][cols_j]);
 * inv[i] = scratch [rows_i][rows_i] holding U_ii^-1 (written by shg_block_potrf, read by the others).  inv[i] may be the
 * diagonal block itself: then U_ii^-1 is kept INSTEAD of U_ii (solve and sparse inverse need nothing else; a third less
 * memory for a long chain), and shg_block_multiply / shg_block_inverse, which read U_ii, must not be used on that factor.
 *   shg_block_potrf           N = W^T W in place, fill-in allocated by the caller       (grates/lstsq.py:698-717)
 *   shg_block_potrf_rows      the same for the block rows first <= r < last only: earlier rows count as factored, later rows are
 *                             left as the Schur complement (two half chains of a tridiagonal system on two streams)
 *   shg_block_solve           W x = b / W^T x = b for B [n][k] in place                 (grates/lstsq.py:778-821, 950-968)
 *   shg_block_sparse_inverse  (W^T W)^-1 on the pattern of W (Takahashi), in place      (grates/lstsq.py:823-846, 1026-1042)
 *   shg_block_inverse         full inverse, upper blocks, in place                      (grates/lstsq.py:848-882)
 *   shg_block_multiply        V = W B (mode 0), the reference's W^T B (1), N B for a symmetric N (2)   (grates/lstsq.py:719-776)
 * ------------------------------------------------------------------------------------------------ */
int shg_block_potrf(int nb, const int* bounds, const int* rowptr, const int* colidx, double* const* blk, double* const* inv, int* info, void* stream);
int shg_block_potrf_rows(int nb, const int* bounds, const int* rowptr, const int* colidx, double* const* blk, double* const* inv, int first, int last,
                         int* info, void* stream);
/* shg_block_potrf_rows for TWO matrices of the same structure (one table, two sets of blocks and inverses): every launch
 * serves both, info[0] and info[1] receive the two pivot flags.  The two half chains of a two-ended elimination go through the
 * device as one string of launches instead of two that the card overlaps only in part. */
int shg_block_potrf_rows_pair(int nb, const int* bounds, const int* rowptr, const int* colidx, double* const* blk0, double* const* inv0,
                              double* const* blk1, double* const* inv1, int first, int last, int* info, void* stream);
/* The factorisation of a diagonal block larger than 256 overlaps its 128-column panel steps on two more streams of the device
 * (look-ahead; the caller's stream waits for them before the call returns control of the data).  A THREAD that factors matrices
 * beside other threads turns it off for itself: the card does not overlap that many queues (csrc/blas.hip).  The first such
 * factorisation on a stream synchronises that stream once: the side streams are chosen by a timing experiment so that they do
 * not share a hardware queue with it or with each other (csrc/plan.hip); every later call only enqueues. */
int shg_block_set_lookahead(int mode);
/* mode (of the calling thread): 0 = off (recursive sweep on the caller's stream), 1 = on (the default): panel sweep with look-ahead;
 * a chain row -- a block row whose only coupling is to the next one, as in the block-tridiagonal normal equations of a smoother,
 * grates/lstsq.py:364-392 -- carries its coupling block and the next diagonal block through the sweep on the second side stream
 * WHEN that stream has a hardware queue of its own (found by the timing experiment); 2 = on, chain rows always carry them;
 * 3 = on, chain rows never do (the coupling block is formed after the sweep, by one product).  The results of the variants agree to
 * rounding (different summation orders); which one mode 1 takes on this stream is reported by
 * shg_block_lookahead_info: which[0] = mode, [1] = side streams with a hardware queue of their own (0 .. 2), [2] = 1 if chain
 * rows carry their coupling along under the current mode.  (Synchronises the stream once, like the first factorisation.) */
int shg_block_lookahead_info(void* stream, int which[4]);
int shg_block_solve(int nb, const int* bounds, const int* rowptr, const int* colidx, double* const* blk, double* const* inv, int transpose, double* B,
                    int k, int ldb, void* stream);
int shg_block_sparse_inverse(int nb, const int* bounds, const int* rowptr, const int* colidx, double* const* blk, double* const* inv, void* stream);
/* Row ranges (chains cut into segments at separator epochs, grates_amd/distributed.py): the forward sweep eliminates the rows
 * first <= r < last only and leaves the reduced right-hand side in the later rows; the backward sweep and the Takahashi recursion
 * process the rows last - 1 .. first and take the solution / the entries of the inverse of the later rows as given. */
int shg_block_solve_rows(int nb, const int* bounds, const int* rowptr, const int* colidx, double* const* blk, double* const* inv, int transpose,
                         int first, int last, double* B, int k, int ldb, void* stream);
int shg_block_sparse_inverse_rows(int nb, const int* bounds, const int* rowptr, const int* colidx, double* const* blk, double* const* inv,
                                  int first, int last, void* stream);
int shg_block_inverse(int nb, const int* bounds, const int* rowptr, const int* colidx, double* const* blk, double* const* inv, void* stream);
int shg_block_multiply(int nb, const int* bounds, const int* rowptr, const int* colidx, double* const* blk, int mode, const double* B, int k, int ldb,
                       double* V, int ldv, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Analysis (area-weighted least squares per order)
 *   replaces RegularGrid.to_potential_coefficients                    (grates/grid.py:665-696, 752-790)
 *   grid [B][nlat][nlon]; area [nlat][nlon]; anm [B][N+1][N+1] (degrees < nmin left zero)
 * The per-order operators depend on the plan, the weights and nmin only; they are cached in the plan and every call compares
 * `area` with the weights they were built for (on the device; the verdict costs one host synchronisation at the end of the call).
 * area == NULL: "the weights of the previous call" -- the cached operators are used as they are and nothing is compared or
 * waited for; an error if the plan holds no operators for this nmin.
 * ------------------------------------------------------------------------------------------------ */
int shg_analysis(shg_plan* plan, const double* grid, const double* area, int nmin, int B, double* anm, void* stream);

/* How the cached operators of a plan are applied: info[0] = 1 when the operator product uses the north-south parity split (parallels
 * that are mirror images of each other, mirror-symmetric weights: x_n = sum over the northern parallels of Hp[n][i] (g[i] +- g[mirror i]),
 * half the products), 0 for the full product, -1 when the plan holds no operators yet; info[1] = the largest entry of the operators
 * that the split drops relative to their largest entry (the split is used below 5e-12; asymmetric weights give ~1). */
int shg_analysis_info(const shg_plan* plan, double info[2]);

/* ------------------------------------------------------------------------------------------------
 * Full-matrix forms of the operators
 *   shg_synthesis_matrix  dense synthesis operator of a point list, A [npts][Pn], Pn = (N+1)^2 - nmin^2 degree-wise
 *                         columns: A[p][c] = kn[p][n] P_nm(colat_p) cos|sin(m lon_p)          (grates/grid.py:412-443)
 *                         colat, lon [npts], kn [npts][N+1] device arrays
 *   shg_analysis_matrix   dense analysis operator of a regular grid, F [Pn][nlat * nlon]: the least-squares operator of
 *                         shg_analysis written out (F v = analysis of v)                      (grates/grid.py:698-730)
 *   shg_congruence        C [n][n] = W [n][k] S [k][k] W^T: filtered covariance matrix `W @ S @ W.T` built from
 *                         SpatialFilter.matrix() (grates/filter.py:74-95, 193-222, 481-509) ahead of the covariance
 *                         propagation (grates/grid.py:792-839); two fp64 MFMA GEMMs, the second on the upper tiles only;
 *                         work [n][k] scratch
 * ------------------------------------------------------------------------------------------------ */
int shg_synthesis_matrix(int N, int nmin, const double* colat, const double* lon, const double* kn, int npts, double* A, void* stream);
int shg_analysis_matrix(shg_plan* plan, const double* area, int nmin, double* F, void* stream);
int shg_congruence(int n, int k, const double* W, int ldw, const double* S, int lds, double* C, int ldc, double* work, void* stream);

/* Some operations keep their scratch buffers per stream between calls (the split-K workspace of the block products, the
 * buffers of shg_analysis: freeing stream-ordered memory costs more than these calls take).  This gives them back; it waits
 * for the device first. */
int shg_scratch_release(void);
const char* shg_last_error(void);
const char* shg_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SHG_H */
