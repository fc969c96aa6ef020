// fp64 MFMA GEMM (v_mfma_f64_16x16x4_f64) in two flavours that share one main loop:
//   MODE_PLAIN    C[M][N] = A[M][K] B[K][N]                      shg_dgemm, shg_dense_filter (grates/filter.py:473-474)
//   MODE_COVPROP  sigma2[r] = sum_c (A Sigma)[r][c] A[r][c]        shg_covprop_diag            (grates/grid.py:833-835)
//                 with the rows of A = synthesis matrix generated on the fly:
//                 A[(i, j)][p] = PK[i][p] * CS[slot(p)][j]         (grates/grid.py:825-834: F = cs * Pnm[k])
//                 A (68 GB at d/o 180 / 0.5 deg) is never materialised, only M doubles leave the kernel.
//   MODE_SYNTH    C[M][N] = A[M][K] X[K][N] with the same generated A     point-list synthesis of many epochs (points.hip)
//
// Block tile 128 x 128, BK = 16, 4 waves as 2 x 2, wave tile 64 x 64 (16 accumulators).  Global -> register
// prefetch of the next K tile overlaps the 64 MFMAs of the current one; one barrier per K tile.
#include "common.h"

namespace shg {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16;
#ifndef SHG_GEMM_X
#define SHG_GEMM_X 0                    // experiment switches (timing only, results are wrong): 1 no A products, 2 no barrier
#endif                                  // in the K loop, 4 no operand loads in the K loop, 8 no LDS stores in the K loop, 16 Sigma tile not staged, 32 A tile not
                                        // staged, 64 no Legendre factor
#ifndef SHG_GEMM_PRIO
#define SHG_GEMM_PRIO 1                 // wave priority raised while a tile is staged (1: until the barrier, 2: through it, 3: through the last k-step)
#endif
#ifndef SHG_GEMM_INTERLEAVE
#define SHG_GEMM_INTERLEAVE 1           // operand loads dealt between the MFMAs (0: experiment switch, loads in a row)
#endif
#ifndef SHG_LDA
#define SHG_LDA 17
#endif
constexpr int LDA = SHG_LDA;     // As[row][k], 17-double rows: neither the fragment reads (16 rows x 2 k per half wave) nor the staging writes (32 consecutive rows per half wave) have a bank conflict; 18 made the writes 2-way (+0.7 %)
constexpr int LDB = 144;    // Bs[k][col], 128 + 16 pad

enum { MODE_PLAIN = 0, MODE_COVPROP = 1, MODE_SYNTH = 2 };      // SYNTH: A generated like COVPROP, C stored like PLAIN

struct GemmParams {
    int M, N, K;
    const double* A;
    int lda;
    const double* B;
    int ldb;
    double* C;
    int ldc;
    // covariance propagation
    const double* pkd;      // [nlat][ldp]  kn-scaled Legendre functions in degree-wise order (min_degree 0)
    int ldp;
    int pk_rows;            // rows of pkd (regular grids: bounds the buffer descriptor of the Legendre table)
    const double* csr;      // [2N+1][nlon] 1, cos(lon), sin(lon), cos(2 lon), ... : row = rank of the coefficient inside its degree
    const int* rslot;       // [ldp] rank inside the degree of every degree-wise index
    const unsigned* csoff;  // [ldp + 16] byte offset rank * ldcs * 8 of the cos/sin row of every degree-wise index (regular grids)
    int ldcs;               // leading dimension of csr (nlon, or the point count for point lists)
    long long idiv, jmod;   // flat row R -> table row R / idiv, table column R % jmod (regular grid: both nlon)
    int p_off;              // min_degree^2: first degree-wise index covered by the covariance matrix
    long long row0;         // first flat grid row (lat0 * nlon) of the band
    double* partial;        // [gridDim.x][M] per-column-block partial row sums
    int pkt;                // generated A: Legendre table stored transposed, pkd[p][row] (point lists)
#ifdef SHG_TIMELINE
    unsigned long long* tl; // profiling build: [blocks][4 waves][8] cycle sums of the K-loop phases (tools/gemm_phases.py)
#endif
};

// profiling build only: shader-clock cycles spent in the phases of the K loop, summed per wave on the scalar unit
#ifdef SHG_TIMELINE
#define SHG_TL_MARK(i)                                          \
    do {                                                        \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        tl_sum[i] += now_ - tl_prev;                            \
        tl_prev = now_;                                         \
    } while (0)
#else
#define SHG_TL_MARK(i)
#endif

// SYM (covariance mode only): Sigma is symmetric and only its upper triangle is used,
//   sigma2[r] = sum_c [ sum_{p<c} 2 A[r][p] Sigma[p][c] + A[r][c] Sigma[c][c] ] A[r][c]:
// column block n0 runs its K loop over p < n0 + 128 only (half the MFMAs on average); the accumulators are doubled once when
// the loop reaches the diagonal block, inside which the Sigma tile is weighted 2 / 1 / 0 (p < c / p = c / p > c) while it is
// staged.  Column blocks are taken from the last (longest) to the first.
// PKT (generated A only): the Legendre table is stored transposed, pkd[p][row] with ldp = number of rows -- the layout of
// point lists, where every row of A has its own table row and consecutive lanes (rows) then read consecutive addresses.
template <int MODE, bool VEC, bool SYM = false, bool PKT = false>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmParams P) {      // 2 waves per SIMD: two blocks per CU
    extern __shared__ double gemm_lds[];
    double (*As)[BM * LDA] = reinterpret_cast<double (*)[BM * LDA]>(gemm_lds);                       // [2][BM * LDA]
    double (*Bs)[BK * LDB] = reinterpret_cast<double (*)[BK * LDB]>(gemm_lds + 2 * BM * LDA);        // [2][BK * LDB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fk = lane >> 4;
    // wave-uniform half of the K tile a thread generates in COVPROP mode: keeps the degree / rank bookkeeping on the scalar unit
    const int khalf = __builtin_amdgcn_readfirstlane(tid >> 7);
    // regular-grid covariance propagation: operands come through buffer loads (see fetch_full)
    // (not the symmetric variant: with its weighting code hipcc runs out of registers, spills the prefetched operands inside the
    //  loop and the result of the spilled build is wrong in the last row quad of every wave tile; it keeps the global-load form)
    constexpr bool BUF = (MODE == MODE_COVPROP) && !PKT && !SYM;
    // ... and with 16-byte aligned Sigma rows (VEC) the Sigma tile does not pass through registers at all: every wave copies four
    // tile rows (1 KB each) straight into LDS with buffer loads that write to LDS, right behind the barrier that frees the buffer
    // (the same for the B operand of a plain product: PDMA)
    constexpr bool PDMA = (MODE == MODE_PLAIN) && VEC;
    constexpr bool DMA = (BUF && VEC) || PDMA;
    // PLAIN: column block fastest.  COVPROP: row block fastest -- the blocks resident at one time then walk the same
    // 33 MB column panel of Sigma together and it is fetched from HBM once instead of once per row block.
    const int m0 = (MODE == MODE_PLAIN ? blockIdx.y : blockIdx.x) * BM;
    const int colblock = MODE == MODE_PLAIN ? (int)blockIdx.x : (SYM ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y);
    const int n0 = colblock * BN;
    const int Keff = SYM ? min(P.K, n0 + BN) : P.K;                // SYM: rows of Sigma up to the end of the diagonal block

    // ---- operand fetch.  Full K tiles are fetched without any bounds test: row / column indices beyond the matrix are
    //      clamped to valid addresses (the garbage only reaches rows / columns that are never stored), so the main loop
    //      is branch free and the compiler can keep the prefetch in flight across the MFMAs.  Only the last partial
    //      K tile zero-fills.
    // PLAIN A: 128 rows x 8 pieces of 2 doubles; piece h of a thread: row = (tid >> 3) + 32 h, k = (tid & 7) * 2
    // COVPROP A: element h of a thread: row = tid & 127 (fixed), k = (tid >> 7) + 2 h, generated from two small tables
    // B: 16 k-rows x 64 pieces of 2 doubles; piece h: k = (tid >> 6) + 4 h, col = (tid & 63) * 2
    // Addressing: fp64 MFMAs and the VALU instructions of all waves of a SIMD share one issue pipe (tools/mfma64_issue.hip),
    // so the K loop keeps address arithmetic on the scalar unit: every load is "uniform 64-bit base (advanced per K tile by
    // scalar instructions) + per-lane 32-bit byte offset (fixed for the whole kernel)".
    double areg[8];
    double creg[8];             // COVPROP: cos/sin factors; the product with areg is formed when the tile is staged, i.e.
                                // after the MFMAs of the current tile, so that the loads stay in flight across them
    double2 breg[4];
    const int a_kk = (tid & 7) * 2;
    const int b_k = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform first k row of the B pieces
    const int b_col = (tid & 63) * 2;
    // (the empty asm pins the uniform base in scalar registers -- without it the compiler folds base + lane offset into one
    //  loop-invariant 64-bit vector address and adds the scalar part with a VALU instruction per load; the address goes
    //  through an integer, so the pointer is rebuilt in the global address space explicitly.  The 32 -> 64 bit extension of
    //  the lane offset has to stay in the basic block of the load for the scalar-base form to be selected: `pin` below.)
    auto pin = [](unsigned v) {
        asm volatile("" : "+v"(v));
        return v;
    };
    typedef const double __attribute__((address_space(1))) gdouble_t;
    typedef const char __attribute__((address_space(1))) gbyte_t;
    typedef int __attribute__((address_space(4))) crank_t;
    typedef double gdouble2_v __attribute__((ext_vector_type(2)));
    typedef const gdouble2_v __attribute__((address_space(1))) gdouble2_t;
    auto at = [](const double* base, unsigned byte_off) {
        unsigned long long b = reinterpret_cast<unsigned long long>(base);
        asm volatile("" : "+s"(b));
        return *reinterpret_cast<gdouble_t*>(reinterpret_cast<gbyte_t*>(b) + byte_off);
    };
    auto at2 = [](const double* base, unsigned byte_off) {
        unsigned long long b = reinterpret_cast<unsigned long long>(base);
        asm volatile("" : "+s"(b));
        const gdouble2_v v = *reinterpret_cast<gdouble2_t*>(reinterpret_cast<gbyte_t*>(b) + byte_off);
        return make_double2(v.x, v.y);
    };
    // PLAIN A: per-lane offsets of the four row pieces relative to row m0 (clamped rows stay inside the matrix)
    unsigned a_voff[4] = {0, 0, 0, 0};
    const double* a_base = nullptr;                                  // uniform: &A[m0][0]
    if (MODE == MODE_PLAIN) {
        a_base = P.A + (size_t)m0 * P.lda;
#pragma unroll
        for (int h = 0; h < 4; ++h) a_voff[h] = (unsigned)(((size_t)(min(m0 + (tid >> 3) + 32 * h, P.M - 1) - m0) * P.lda + a_kk) * 8);
    }
    // COVPROP A: per-lane offsets into the two tables
    unsigned pk_voff = 0, cs_voff = 0;
    const double* pk_base = nullptr;                                 // uniform: table row of the block's first grid row
    if (MODE != MODE_PLAIN) {
        const long long R = P.row0 + min(m0 + (tid & 127), P.M - 1);
        // (64-bit division runs on the vector unit even for uniform operands: readfirstlane brings the quotient back)
        const int gi0 = __builtin_amdgcn_readfirstlane((int)((P.row0 + m0) / P.idiv));
        if (PKT) {
            pk_base = P.pkd + (size_t)P.p_off * P.ldp + (P.row0 + m0);       // + k * ldp
            pk_voff = (unsigned)((R - (P.row0 + m0)) * 8);
        } else {
            pk_base = P.pkd + (size_t)gi0 * P.ldp + P.p_off;
            pk_voff = (unsigned)((R / P.idiv - gi0) * P.ldp * 8);    // at most 127 table rows
        }
        cs_voff = (unsigned)((R % P.jmod) * 8);
    }
    const unsigned b_voff0 = (unsigned)((VEC ? min(n0 + b_col, (P.N + 1) / 2 * 2 - 2) : min(n0 + b_col, P.N - 1)) * 8);
    const unsigned b_voff1 = (unsigned)(min(n0 + b_col + 1, P.N - 1) * 8);

    // COVPROP: ranks (inside their degree) of the eight degree-wise indices a thread generates in the K tile it fetches next,
    // read from the rank table with scalar loads (constant address space: uniform loads from it are scalar loads)
    int rank[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto load_ranks = [&](int k0) {
        if (BUF) return;
        const crank_t* rs = reinterpret_cast<const crank_t*>(reinterpret_cast<unsigned long long>(P.rslot));
#pragma unroll
        for (int h = 0; h < 8; ++h) rank[h] = rs[min(k0 + khalf + 2 * h, P.K - 1) + P.p_off];      // (the table ends at p_off + K)
    };
    if (MODE != MODE_PLAIN) load_ranks(0);
    // BUF: every operand load of a full K tile is a buffer load "descriptor (scalar) + lane offset (vector, fixed for the whole
    // kernel) + scalar offset", so that nothing but the load itself is issued per operand: the general form above spends ~140
    // scalar instructions per K tile on 64-bit row addresses, ~2000 cycles in which the wave issues no MFMA -- as long as the
    // 64 MFMAs of the other wave of the SIMD take, which leaves no slack and idles the pipe 16 % of the time.
    //   cos/sin: descriptor of the table, lane offset = meridian, scalar offset = rank * ldcs * 8 read as ONE 16-dword scalar load
    //            from the offset table (entries k0 + khalf + 2 h of the tile sit at the even positions)
    //   PK:      descriptor of the block's first table row, lane offset = table row of the lane (+ 16 h: immediate), scalar
    //            offset (k0 + khalf) * 8
    //   Sigma:   descriptor rebuilt per tile at row k0 (the matrix exceeds 4 GB), lane offset = column, scalar offset = row * ldb * 8
    typedef unsigned uint16_v __attribute__((ext_vector_type(16), aligned(4)));
    typedef const uint16_v __attribute__((address_space(4))) coff16_t;
    uint16_v csoffs = {};
    auto load_offsets = [&](int k0) {           // (the table is padded by 16 entries: no clamping)
        if (BUF) csoffs = *reinterpret_cast<coff16_t*>(reinterpret_cast<unsigned long long>(P.csoff + (P.p_off + k0 + khalf)));
    };
    auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    };
    const __amdgpu_buffer_rsrc_t rs_cs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(P.csr), 0, 0xffffffffu, 0x00020000);
    // (the Legendre descriptor ends with the table: the A operand is requested one tile ahead as well, and the request for the
    //  tile after the last one of the last parallel reads as zero instead of leaving the allocation)
    const unsigned pk_bytes = BUF ? (unsigned)(((long long)P.pk_rows * P.ldp - (pk_base - P.pkd)) * 8) : 0u;
    const __amdgpu_buffer_rsrc_t rs_pk = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(pk_base), 0, pk_bytes, 0x00020000);
    unsigned brow_soff[4] = {0, 0, 0, 0};
    if (BUF || PDMA) {
#pragma unroll
        for (int h = 0; h < 4; ++h) brow_soff[h] = (unsigned)((size_t)(b_k + 4 * h) * P.ldb * 8);
    }
    if (BUF) load_offsets(0);
    // (BUF) rows h0 .. h1-1 of the generated A operand of the K tile at k0, and its Sigma rows.  The Sigma descriptor ends with
    // the matrix: rows beyond K read as zero, so a tile may be requested speculatively (the loop asks one tile ahead of need)
    auto fetch_A = [&](int k0, int h0, int h1) {
        const unsigned ksoff = (unsigned)(k0 + khalf) * 8u;          // uniform
#pragma unroll
        for (int h = h0; h < h1; ++h) {
            areg[h] = bload(rs_pk, pk_voff + 16 * h, ksoff);
            creg[h] = bload(rs_cs, cs_voff, csoffs[2 * h]);
        }
    };
    const unsigned b_row_bytes = (unsigned)P.ldb * 8u;
    const int b_rows_cap = (int)(0xffffffffu / max(b_row_bytes, 1u));        // rows that fit the 32-bit range of a descriptor
    typedef void __attribute__((address_space(3))) lds_void_t;
    auto fetch_B = [&](int k0, int buf) {
        const int left = max(P.K - k0, 0);                           // uniform: rows from k0 to the end of Sigma (32-bit scalar arithmetic)
        const unsigned records = left > b_rows_cap ? 0xffffffffu : (unsigned)left * b_row_bytes;
        const __amdgpu_buffer_rsrc_t rs_b =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(P.B + (size_t)min(k0, P.K) * P.ldb), 0, records, 0x00020000);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            if (DMA) {
                // lane l delivers columns 2 l, 2 l + 1 of tile row b_k + 4 h to the LDS row base + 16 l
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void_t*)(Bs[buf] + (b_k + 4 * h) * LDB), 16, b_voff0, brow_soff[h], 0, 0);
            } else if (VEC) {
                const uint4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs_b, b_voff0, brow_soff[h], 0);
                breg[h] = __builtin_bit_cast(double2, v);
            } else {
                breg[h] = make_double2(bload(rs_b, b_voff0, brow_soff[h]), bload(rs_b, b_voff1, brow_soff[h]));
            }
        }
    };
    // The operand loads of a full K tile in three parts, each issued between the MFMAs of one k-step (SHG_TILE_HEAD):
    //   PLAIN          part 0: A            part 1: B
    //   generated A    part 0: A rows 0-3   part 1: A rows 4-7   part 2: Sigma / X
    //   BUF            behind the barrier, under the last k-step of the previous tile: Sigma and A rows 0-3; part 0: A rows 4-7
    constexpr int NLOAD0 = MODE == MODE_PLAIN ? (VEC ? 4 : 8) : 8;
    constexpr int NLOADA = 8;                                        // BUF: loads of A rows 0-3
    constexpr int NLOAD1 = MODE == MODE_PLAIN ? (VEC ? 0 : 8) : (BUF ? 0 : 8);      // (plain product with 16-byte loads: B goes straight to LDS)
    constexpr int NLOAD2 = (MODE == MODE_PLAIN || BUF) ? 0 : (VEC ? 4 : 8);
    constexpr int NLOADB = VEC ? 4 : 8;
    auto fetch_old_B = [&](int k0) {
        const unsigned boff0 = pin(b_voff0), boff1 = pin(b_voff1);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const double* brow = P.B + (size_t)(k0 + b_k + 4 * h) * P.ldb;      // uniform
            if (VEC)
                breg[h] = at2(brow, boff0);
            else
                breg[h] = make_double2(at(brow, boff0), at(brow, boff1));
        }
    };
    auto fetch_part = [&](int k0, int part) {
        if (MODE == MODE_PLAIN) {
            if (part == 0) {
                const double* ak = a_base + k0;                       // uniform
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const unsigned off = pin(a_voff[h]);
                    if (VEC) {
                        const double2 t = at2(ak, off);
                        areg[2 * h] = t.x;
                        areg[2 * h + 1] = t.y;
                    } else {
                        areg[2 * h] = at(ak, off);
                        areg[2 * h + 1] = at(ak + 1, off);
                    }
                }
            } else if (part == 1 && !PDMA) {
                fetch_old_B(k0);
            }
        } else if (BUF) {
            if (part == 0) fetch_A(k0, 4, 8);        // rows 0-3 and Sigma were requested behind the previous barrier
        } else if (part < 2) {
            // degree-wise index p = n^2 + r: the rank r inside the degree selects the cos/sin row (table lookup on the scalar unit)
            const double* pk = pk_base + k0 + khalf;                  // uniform
            const unsigned poff = pin(pk_voff), coff = pin(cs_voff);
#pragma unroll
            for (int h = 4 * part; h < 4 * part + 4; ++h) {
                areg[h] = PKT ? at(pk_base + (size_t)(k0 + khalf + 2 * h) * P.ldp, poff) : at(pk + 2 * h, poff);
                creg[h] = at(P.csr + (size_t)rank[h] * P.ldcs, coff);
            }
        } else {
            fetch_old_B(k0);
        }
    };
    auto fetch_full = [&](int k0) {
        fetch_part(k0, 0);
        fetch_part(k0, 1);
        fetch_part(k0, 2);
        if (BUF) {
            fetch_A(k0, 0, 4);
            fetch_B(k0, 0);
        }
        if (PDMA) fetch_B(k0, 0);
    };
    // last partial K tile: same addresses with k clamped, entries beyond K zeroed
    auto fetch_tail = [&](int k0) {
        if (MODE == MODE_PLAIN) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int k1 = k0 + a_kk;                               // a_voff already points at column a_kk
                const double* arow = reinterpret_cast<const double*>(reinterpret_cast<const char*>(a_base) + a_voff[h]) - a_kk;     // per lane
                const double v0 = arow[min(k1, P.K - 1)];
                const double v1 = arow[min(k1 + 1, P.K - 1)];
                areg[2 * h] = k1 < P.K ? v0 : 0.0;
                areg[2 * h + 1] = k1 + 1 < P.K ? v1 : 0.0;
            }
        } else {
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const int gk = k0 + 2 * h + khalf;
                const int kc = min(gk, P.K - 1);
                areg[h] = gk < P.K ? (PKT ? at(pk_base + (size_t)kc * P.ldp, pk_voff) : at(pk_base + kc, pk_voff)) : 0.0;
                creg[h] = at(P.csr + (size_t)reinterpret_cast<const crank_t*>(reinterpret_cast<unsigned long long>(P.rslot))[kc + P.p_off] * P.ldcs, cs_voff);
            }
        }
        if (DMA) return;                          // the Sigma rows of the partial tile are in LDS already (rows beyond K read as zero)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int gk = k0 + b_k + 4 * h;
            const double* brow = P.B + (size_t)min(gk, P.K - 1) * P.ldb;
            const double x = at(brow, b_voff0);
            const double y = VEC ? at(brow + 1, b_voff0) : at(brow, b_voff1);
            breg[h] = gk < P.K ? make_double2(x, y) : make_double2(0.0, 0.0);
        }
    };
    typedef double __attribute__((address_space(3))) lds_store_t;
    typedef double dbl2_v __attribute__((ext_vector_type(2)));
    typedef dbl2_v __attribute__((address_space(3))) lds_store2_t;
    lds_store_t* a_stage_base = (lds_store_t*)(As[0] + (MODE == MODE_PLAIN ? (tid >> 3) * LDA + a_kk : (tid & 127) * LDA + khalf));
    lds_store_t* b_stage_base = (lds_store_t*)(Bs[0] + b_k * LDB + b_col);
    asm volatile("" : "+v"(a_stage_base));
    asm volatile("" : "+v"(b_stage_base));
    auto stage = [&](int buf) {
        if (MODE == MODE_PLAIN) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                lds_store_t* dst = a_stage_base + buf * (BM * LDA) + 32 * h * LDA;
                dst[0] = areg[2 * h];
                dst[1] = areg[2 * h + 1];
            }
        } else if (!(SHG_GEMM_X & 32)) {
#pragma unroll
            for (int h = 0; h < 8; ++h) a_stage_base[buf * (BM * LDA) + 2 * h] = (SHG_GEMM_X & 1) ? areg[h] : ((SHG_GEMM_X & 64) ? creg[h] : areg[h] * creg[h]);
        }
        if ((SHG_GEMM_X & 16) || DMA) return;
#pragma unroll
        for (int h = 0; h < 4; ++h) *(lds_store2_t*)(b_stage_base + buf * (BK * LDB) + 4 * h * LDB) = (dbl2_v){breg[h].x, breg[h].y};
    };
    // SYM: weights of the Sigma tile rows k0 + b_k + 4 h inside the diagonal block (applied to breg before staging)
    auto weight_diagonal = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int p = k0 + b_k + 4 * h, c = n0 + b_col;
            breg[h].x *= p < c ? 2.0 : (p == c ? 1.0 : 0.0);
            breg[h].y *= p < c + 1 ? 2.0 : (p == c + 1 ? 1.0 : 0.0);
        }
    };

    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};

    // K tile = 4 k-steps of 16 MFMAs.  The fragments of k-step ks + 1 are requested from LDS BEFORE the MFMAs of k-step ks are
    // issued (two fragment sets; the scheduling barriers pin that order -- left alone, hipcc issues the reads after the MFMAs
    // that free their registers and every k-step then waits a full LDS round trip on an idle pipe), and the loop is rotated:
    // the last k-step of a tile is issued after the barrier that publishes the next tile, behind the request for that tile's
    // first fragments, so the LDS latency after the barrier is covered as well.
    double af0[4], bf0[4], af1[4], bf1[4];
    // Every fragment is one ds_read_b64 "lane base + 16-bit immediate": the bases are opaque to the compiler (it would split the
    // 72 KB of offsets differently and add to the base with a VALU instruction per k-step) and the B reads are volatile, which
    // keeps them from being paired into ds_read2_b64, whose 8-bit offsets need such an add as well.  An integer VALU instruction
    // costs the MFMA pipe more than a tenth of an MFMA (profiles/r01_mfma64_issue.txt).
    typedef double __attribute__((address_space(3))) lds_double_t;       // 32-bit LDS addresses: the bases stay ds_read / ds_write operands
    const lds_double_t* a_frag_base = (const lds_double_t*)(As[0] + (wr * 64 + fr) * LDA + fk);
    const lds_double_t* b_frag_base = (const lds_double_t*)(Bs[0] + fk * LDB + wc * 64 + fr);
    asm volatile("" : "+v"(a_frag_base));
    asm volatile("" : "+v"(b_frag_base));
#define SHG_FRAGS(af, bf, buf, ks)                                                                                   \
    do {                                                                                                             \
        const lds_double_t* Ab_ = a_frag_base + (buf) * (BM * LDA) + (ks) * 4;                                       \
        const volatile lds_double_t* Bb_ = b_frag_base + (buf) * (BK * LDB) + (ks) * 4 * LDB;                        \
        _Pragma("unroll") for (int a = 0; a < 4; ++a) af[a] = Ab_[a * 16 * LDA];                                     \
        _Pragma("unroll") for (int b = 0; b < 4; ++b) bf[b] = Bb_[b * 16];                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)
#define SHG_MFMA16(af, bf)                                                                                           \
    do {                                                                                                             \
        _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                                \
            _Pragma("unroll") for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)
    // 16 MFMAs with `nloads` operand loads (issued in front of them in the source) dealt between them, two MFMAs per load: a
    // wave issues in order, and 24 loads in a row keep it away from the MFMA pipe for ~1800 cycles (the address unit takes one
    // load per ~16 cycles and serves eight waves)
#define SHG_MFMA16_LOADS(af, bf, nloads)                                                                             \
    do {                                                                                                             \
        _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                                \
            _Pragma("unroll") for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < (SHG_GEMM_INTERLEAVE ? (nloads) : 0); ++i) {                           \
            __builtin_amdgcn_sched_group_barrier(0x008, (nloads) > 8 ? 1 : ((nloads) > 4 ? 2 : 4), 0);                \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                       \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)
    // k-steps 0 .. 2 of the tile in `buf` (set 0 holds the fragments of k-step 0); leaves k-step 3 in set 1.  BUF: the A operand
    // of the next tile (at knext) is requested under k-steps 0 and 1
#define SHG_TILE_HEAD(buf, knext, prefetch) \
    do {                                    \
        SHG_FRAGS(af1, bf1, buf, 1);        \
        if (prefetch) {                     \
            fetch_part(knext, 0);           \
            SHG_MFMA16_LOADS(af0, bf0, NLOAD0); \
        } else {                            \
            SHG_MFMA16(af0, bf0);           \
        }                                   \
        SHG_FRAGS(af0, bf0, buf, 2);        \
        if ((prefetch) && NLOAD1 > 0) {     \
            fetch_part(knext, 1);           \
            SHG_MFMA16_LOADS(af1, bf1, NLOAD1); \
        } else {                            \
            SHG_MFMA16(af1, bf1);           \
        }                                   \
        SHG_FRAGS(af1, bf1, buf, 3);        \
        if ((prefetch) && NLOAD2 > 0) {     \
            fetch_part(knext, 2);           \
            SHG_MFMA16_LOADS(af0, bf0, NLOAD2); \
        } else {                            \
            SHG_MFMA16(af0, bf0);           \
        }                                   \
    } while (0)

    const int nfull = Keff / BK;
    const bool has_tail = (Keff % BK) != 0;
    const int tdiag = n0 / BK;                         // SYM: first K tile of the diagonal block
    auto double_accumulators = [&]() {                 // SYM: all rows above the diagonal block are accumulated: they count twice
        // The MFMAs of the previous k-step were issued just before (rotated loop) and hipcc does not separate a VALU read from the
        // last passes of an MFMA across the loop back edge: without the explicit wait the fourth row quad of the last accumulators
        // is read before it is written (observed: sigma = NaN in rows 60-63 of the wave tiles; 32 idle cycles are not enough).
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] *= 2.0;
    };
#ifdef SHG_TIMELINE
    unsigned long long tl_sum[4] = {0, 0, 0, 0};
    const unsigned long long tl_start = __builtin_readcyclecounter(), tl_wall = wall_clock64();
    unsigned long long tl_prev = tl_start;
#endif
    if (nfull > 0) {
        fetch_full(0);
    } else {
        fetch_tail(0);
        if (DMA) fetch_B(0, 0);
    }
    if (SYM && tdiag == 0) weight_diagonal(0);
    stage(0);
    // (scalar loads return out of order: while one is in flight every wait for LDS data degrades to lgkmcnt(0).  The ranks of
    //  the tile after next are therefore requested right in front of the barrier, which waits for lgkmcnt(0) anyway.)
    if (MODE != MODE_PLAIN) {
        load_ranks(BK);
        load_offsets(BK);
    }
    __syncthreads();
    SHG_FRAGS(af0, bf0, 0, 0);
    if (BUF) {                                         // tile 1: Sigma and the first half of A
        fetch_B(BK, 1);
        fetch_A(BK, 0, 4);
    }
    if (PDMA) fetch_B(BK, 1);
    // branch-free steady state, two K tiles per trip so that the LDS buffer of every access is a literal
#define SHG_STEP(t, buf)                                                      \
    do {                                                                      \
        SHG_TL_MARK(0);                                                       \
        if (SYM && (t) == tdiag && (t) > 0) double_accumulators();            \
        SHG_TILE_HEAD(buf, ((t) + 1) * BK, !(SHG_GEMM_X & 4));               \
        SHG_TL_MARK(1);                                                       \
        if (SYM && (t) + 1 >= tdiag) weight_diagonal(((t) + 1) * BK);         \
        if (SHG_GEMM_PRIO) __builtin_amdgcn_s_setprio(3);                     \
        if (!(SHG_GEMM_X & 8)) stage((buf) ^ 1);                              \
        if (MODE != MODE_PLAIN) {                                             \
            load_ranks(((t) + 2) * BK);                                       \
            load_offsets(((t) + 2) * BK);                                     \
        }                                                                     \
        SHG_TL_MARK(2);                                                       \
        if (SHG_GEMM_PRIO == 1) __builtin_amdgcn_s_setprio(0);                \
        if (!(SHG_GEMM_X & 2)) __syncthreads();                               \
        if (SHG_GEMM_PRIO == 2) __builtin_amdgcn_s_setprio(0);                \
        SHG_FRAGS(af0, bf0, (buf) ^ 1, 0);                                    \
        if (BUF && !(SHG_GEMM_X & 4)) {                                       \
            fetch_B(((t) + 2) * BK, buf);                                     \
            fetch_A(((t) + 2) * BK, 0, 4);                                    \
            SHG_MFMA16_LOADS(af1, bf1, NLOADB + NLOADA);                      \
        } else if (PDMA && !(SHG_GEMM_X & 4)) {                               \
            fetch_B(((t) + 2) * BK, buf);                                     \
            SHG_MFMA16_LOADS(af1, bf1, NLOADB);                               \
        } else {                                                              \
            SHG_MFMA16(af1, bf1);                                             \
        }                                                                     \
        if (SHG_GEMM_PRIO == 3) __builtin_amdgcn_s_setprio(0);                \
        SHG_TL_MARK(3);                                                       \
    } while (0)
    int t = 0;
    for (; t + 2 < nfull; t += 2) {
        SHG_STEP(t, 0);
        SHG_STEP(t + 1, 1);
    }
    if (t + 1 < nfull) SHG_STEP(t, 0);                 // t is even here
    if (nfull > 0) {                                   // last full tile; set 0 holds its first fragments
        if (has_tail) fetch_tail(nfull * BK);
        if (SYM && nfull - 1 == tdiag && tdiag > 0) double_accumulators();
        if ((nfull - 1) & 1) {
            SHG_TILE_HEAD(1, 0, false);
        } else {
            SHG_TILE_HEAD(0, 0, false);
        }
        if (has_tail) {
            if (SYM) weight_diagonal(nfull * BK);
            stage(nfull & 1);
            __syncthreads();
            if (nfull & 1) {
                SHG_FRAGS(af0, bf0, 1, 0);
            } else {
                SHG_FRAGS(af0, bf0, 0, 0);
            }
        }
        SHG_MFMA16(af1, bf1);
    }
    if (has_tail) {
        if (SYM && nfull == tdiag && tdiag > 0) double_accumulators();
        if (nfull & 1) {
            SHG_TILE_HEAD(1, 0, false);
        } else {
            SHG_TILE_HEAD(0, 0, false);
        }
        SHG_MFMA16(af1, bf1);
    }
#undef SHG_STEP
#undef SHG_MFMA16_LOADS
#undef SHG_TILE_HEAD
#undef SHG_FRAGS
#undef SHG_MFMA16
    __syncthreads();

#ifdef SHG_TIMELINE
    if (P.tl && lane == 0) {
        const size_t blin = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        unsigned long long* o = P.tl + (blin * 4 + wave) * 8;
        for (int i = 0; i < 4; ++i) o[i] = tl_sum[i];
        o[4] = __builtin_readcyclecounter() - tl_start;
        o[5] = wall_clock64() - tl_wall;
        o[6] = tl_wall;
        o[7] = (unsigned long long)nfull;
    }
#endif
    // ---- epilogue.  C/D layout: column = lane & 15, row = (lane >> 4) + 4 * reg
    if (MODE != MODE_COVPROP) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = m0 + wr * 64 + a * 16 + fk + 4 * r;
                if (gr >= P.M) continue;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int gc = n0 + wc * 64 + b * 16 + fr;
                    if (gc < P.N) P.C[(size_t)gr * P.ldc + gc] = acc[a][b][r];
                }
            }
    } else {
        // row-dot of the (A Sigma) tile with the matching A tile, reduced over the 128 columns of the block
        double* red = As[0];                // reuse: [128 rows][2 column halves]
        int colslot[4];
        bool colok[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int gc = n0 + wc * 64 + b * 16 + fr;
            colok[b] = gc < P.N;
            colslot[b] = P.rslot[(colok[b] ? gc : 0) + P.p_off];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr * 64 + a * 16 + fk + 4 * r;
                const bool rok = m0 + row < P.M;
                const long long R = P.row0 + m0 + (rok ? row : 0);
                const long long gi = R / P.idiv;
                const int gj = (int)(R % P.jmod);
                double s = 0.0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (rok && colok[b]) {
                        const int pf = n0 + wc * 64 + b * 16 + fr + P.p_off;
                        const double pkv = PKT ? P.pkd[(size_t)pf * P.ldp + R] : P.pkd[gi * P.ldp + pf];
                        const double aval = pkv * P.csr[(size_t)colslot[b] * P.ldcs + gj];
                        s = fma(acc[a][b][r], aval, s);
                    }
                }
                // reduce over the 16 lanes that hold the 16 columns of a tile row
                s += __shfl_xor(s, 1);
                s += __shfl_xor(s, 2);
                s += __shfl_xor(s, 4);
                s += __shfl_xor(s, 8);
                if (fr == 0) red[row * 2 + wc] = s;
            }
        __syncthreads();
        if (tid < BM && m0 + tid < P.M) P.partial[(size_t)colblock * P.M + m0 + tid] = red[tid * 2] + red[tid * 2 + 1];
    }
}

__global__ void covprop_reduce_kernel(int M, int nparts, const double* __restrict__ partial, double* __restrict__ sigma) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    double s = 0.0;
    for (int c = 0; c < nparts; ++c) s += partial[(size_t)c * M + r];
    sigma[r] = sqrt(s);                                     // grates/grid.py:837-839
}

// PKD[i][p] = PK[(m, n)][i] rearranged to the degree-wise index p (min_degree 0); rslot[p] = rank r inside the degree,
// csoff[p] = r * nlon * 8 = byte offset of the cos/sin table row of that rank
__global__ void covprop_pkd_kernel(int N, int nlat, int ldlat, int nlon, const double* __restrict__ pk, double* __restrict__ pkd,
                                   int* __restrict__ rslot, unsigned* __restrict__ csoff) {
    const int P = (N + 1) * (N + 1);
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long long)nlat * P) return;
    const int i = (int)(tid / P), p = (int)(tid % P);
    int n = (int)sqrt((double)p);
    while ((n + 1) * (n + 1) <= p) ++n;
    while (n * n > p) --n;
    const int r = p - n * n;
    const int m = (r + 1) >> 1;
    pkd[tid] = pk[(size_t)(order_offset(N, m) + n - m) * ldlat + i];
    if (i == 0) {
        rslot[p] = r;
        csoff[p] = (unsigned)r * (unsigned)nlon * 8u;
    }
}

// CSR[r][j]: r = 0 -> 1, r = 2m-1 -> cos(m lon_j), r = 2m -> sin(m lon_j)
__global__ void cs_table_kernel(int N, int nlon, const double* __restrict__ lon, double* __restrict__ csr) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long long)(2 * N + 1) * nlon) return;
    const int r = (int)(tid / nlon), j = (int)(tid % nlon);
    const int m = (r + 1) >> 1;
    const double arg = (double)m * lon[j];
    csr[tid] = (r == 0) ? 1.0 : ((r & 1) ? cos(arg) : sin(arg));
}

int covprop_build_cs_table(shg_plan* p, hipStream_t stream) {
    if (p->cs_slot) return SHG_OK;
    const long long n = (long long)(2 * p->N + 1) * p->nlon;
    if (hipMalloc((void**)&p->cs_slot, (size_t)n * sizeof(double)) != hipSuccess) return fail(SHG_ERR_NOMEM, "cos/sin table allocation failed");
    hipLaunchKernelGGL(cs_table_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, stream, p->N, p->nlon, p->lon, p->cs_slot);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

static int launch_gemm(int mode, const GemmParams& P, hipStream_t stream, bool symmetric = false) {
    const dim3 grid = mode == MODE_PLAIN ? dim3(ceil_div(P.N, BN), ceil_div(P.M, BM)) : dim3(ceil_div(P.M, BM), ceil_div(P.N, BN));
    const size_t lds = (size_t)(2 * BM * LDA + 2 * BK * LDB) * sizeof(double);      // 71.7 KB: two blocks per CU
    // 16-byte operand loads need even leading dimensions / sizes and 16-byte aligned bases
    // (an odd N is fine as long as the rows are even: the pair that starts at the last column reads one element of padding,
    //  which only reaches an accumulator column that is never stored)
    const bool vec = (P.ldb % 2 == 0) && (P.N % 2 == 0 || P.ldb > P.N) && ((uintptr_t)P.B % 16 == 0) &&
                     (mode != MODE_PLAIN || ((P.lda % 2 == 0) && (P.K % 2 == 0) && ((uintptr_t)P.A % 16 == 0)));
#define SHG_GEMM_LAUNCH(M_, V_)                                                                                                \
    do {                                                                                                                        \
        SHG_HIP(hipFuncSetAttribute((const void*)gemm_f64_kernel<M_, V_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((gemm_f64_kernel<M_, V_>), grid, dim3(256), lds, stream, P);                                         \
    } while (0)
    if (mode == MODE_PLAIN) {
        if (vec) SHG_GEMM_LAUNCH(MODE_PLAIN, true); else SHG_GEMM_LAUNCH(MODE_PLAIN, false);
    } else if (mode == MODE_SYNTH) {
        // (point lists: transposed Legendre table)
        if (vec) {
            SHG_HIP(hipFuncSetAttribute((const void*)gemm_f64_kernel<MODE_SYNTH, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((gemm_f64_kernel<MODE_SYNTH, true, false, true>), grid, dim3(256), lds, stream, P);
        } else {
            SHG_HIP(hipFuncSetAttribute((const void*)gemm_f64_kernel<MODE_SYNTH, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((gemm_f64_kernel<MODE_SYNTH, false, false, true>), grid, dim3(256), lds, stream, P);
        }
    } else {
        if (symmetric) {
            SHG_HIP(hipFuncSetAttribute((const void*)gemm_f64_kernel<MODE_COVPROP, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((gemm_f64_kernel<MODE_COVPROP, false, true>), grid, dim3(256), lds, stream, P);
        } else if (P.pkt) {
            SHG_HIP(hipFuncSetAttribute((const void*)gemm_f64_kernel<MODE_COVPROP, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((gemm_f64_kernel<MODE_COVPROP, false, false, true>), grid, dim3(256), lds, stream, P);
        } else if (vec) {
            SHG_GEMM_LAUNCH(MODE_COVPROP, true);
        } else {
            SHG_GEMM_LAUNCH(MODE_COVPROP, false);
        }
    }
#undef SHG_GEMM_LAUNCH
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

// C[M][N] = A X for M rows whose A entries are products of two table entries; point lists: pkdT [P][M] (transposed Legendre
// table, ldp = M), csr [2N+1][M]
int synth_generic(const double* pkd, int ldp, const double* csr, int ldcs, const int* rslot, long long idiv, long long jmod, int M,
                  const double* X, int K, int N, double* C, hipStream_t stream) {
    GemmParams G = {};
    G.M = M;
    G.N = N;
    G.K = K;
    G.B = X;
    G.ldb = N;
    G.C = C;
    G.ldc = N;
    G.pkd = pkd;
    G.ldp = ldp;
    G.csr = csr;
    G.ldcs = ldcs;
    G.rslot = rslot;
    G.idiv = idiv;
    G.jmod = jmod;
    G.p_off = 0;
    G.row0 = 0;
    G.pkt = 1;
    return launch_gemm(MODE_SYNTH, G, stream);
}

// sigma[r] = sqrt(a_r^T Sigma a_r) for M rows whose A entries are products of two table entries
int covprop_generic(const double* pkd, int ldp, const double* csr, int ldcs, const int* rslot, long long idiv, long long jmod,
                    long long row0, int M, const double* cov, int Pn, int p_off, double* partial, double* sigma, shg_plan* prof,
                    hipStream_t stream, bool symmetric, bool transposed_table, const unsigned* csoff, int pk_rows, int ldcov) {
    GemmParams G = {};
    G.csoff = csoff;
    if (!transposed_table && !csoff) return fail(SHG_ERR_INVALID, "covprop_generic: the regular-grid kernel needs the cos/sin offset table");
    G.pkt = transposed_table ? 1 : 0;
    G.M = M;
    G.N = Pn;
    G.K = Pn;
    G.B = cov;
    G.ldb = ldcov > 0 ? ldcov : Pn;
    G.pkd = pkd;
    G.ldp = ldp;
    G.pk_rows = pk_rows;
    G.csr = csr;
    G.ldcs = ldcs;
    G.rslot = rslot;
    G.idiv = idiv;
    G.jmod = jmod;
    G.p_off = p_off;
    G.row0 = row0;
    G.partial = partial;
#ifdef SHG_TIMELINE
    G.tl = getenv("SHG_TIMELINE_PTR") ? (unsigned long long*)strtoull(getenv("SHG_TIMELINE_PTR"), nullptr, 0) : nullptr;
#endif
    {
        ProfileScope ps(prof, 3, stream);
        int rc = launch_gemm(MODE_COVPROP, G, stream, symmetric);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(covprop_reduce_kernel, dim3(ceil_div(M, 256)), dim3(256), 0, stream, M, ceil_div(Pn, BN), partial, sigma);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

}  // namespace shg

namespace shg {
int gemm_ex(bool ta, bool tb, int M, int N, int K, double alpha, const double* A, int lda, long long strideA, const double* B, int ldb,
            long long strideB, double beta, double* C, int ldc, long long strideC, int batch, bool upper_only, hipStream_t stream);   // blas.hip
int covprop_rows(shg_plan* p, const double* cov, int Pn, int p_off, int lat0, int lat1, double* partial, hipStream_t stream);
}

using namespace shg;

extern "C" int shg_dgemm(int M, int N, int K, const double* A, int lda, const double* B, int ldb, double* C, int ldc, void* stream_) {
    SHG_REQUIRE(M >= 0 && N >= 0 && K >= 0, "shg_dgemm: negative dimension");
    if (M == 0 || N == 0) return SHG_OK;
    SHG_REQUIRE(C != nullptr, "shg_dgemm: NULL output pointer");
    SHG_REQUIRE(ldc >= N, "shg_dgemm: leading dimension too small");
    if (K == 0) {                                     // empty sum: C = 0
        SHG_HIP(hipMemset2DAsync(C, (size_t)ldc * sizeof(double), 0, (size_t)N * sizeof(double), M, (hipStream_t)stream_));
        return SHG_OK;
    }
    SHG_REQUIRE(A && B, "shg_dgemm: NULL pointer");
    SHG_REQUIRE(lda >= K && ldb >= N, "shg_dgemm: leading dimension too small");
    // too few output tiles to fill the chip (e.g. the dense filter: 14637 x 240): the general kernel splits K
    if (ceil_div(M, BM) * ceil_div(N, BN) < 384 && K >= 512)
        return gemm_ex(false, false, M, N, K, 1.0, A, lda, 0, B, ldb, 0, 0.0, C, ldc, 0, 1, false, (hipStream_t)stream_);
    GemmParams P = {};
    P.M = M;
    P.N = N;
    P.K = K;
    P.A = A;
    P.lda = lda;
    P.B = B;
    P.ldb = ldb;
    P.C = C;
    P.ldc = ldc;
    return launch_gemm(MODE_PLAIN, P, (hipStream_t)stream_);
}

extern "C" int shg_dense_filter(const double* W, int Pn, const double* X, int T, double* Y, void* stream_) {
    SHG_REQUIRE(Pn >= 0 && T >= 0, "shg_dense_filter: negative dimension");
    return shg_dgemm(Pn, T, Pn, W, Pn, X, T, Y, T, stream_);
}

static int covprop_diag_impl(shg_plan* p, const double* cov, int nmin, int lat0, int lat1, double* sigma, void* stream_, bool symmetric);

extern "C" int shg_covprop_diag(shg_plan* p, const double* cov, int nmin, int lat0, int lat1, double* sigma, void* stream_) {
    return covprop_diag_impl(p, cov, nmin, lat0, lat1, sigma, stream_, false);
}

extern "C" int shg_covprop_diag_symmetric(shg_plan* p, const double* cov, int nmin, int lat0, int lat1, double* sigma, void* stream_) {
    return covprop_diag_impl(p, cov, nmin, lat0, lat1, sigma, stream_, true);
}

namespace shg {
// max |S[p][c] - S[c][p]| over 32 x 32 tile pairs (upper tiles compared with their mirror images through LDS)
__global__ __launch_bounds__(256) void symmetry_defect_kernel(int n, const double* __restrict__ S, int ld, double* __restrict__ out) {
    __shared__ double tile[32][33];
    __shared__ double red[256];
    const int bi = blockIdx.y, bj = blockIdx.x;
    double worst = 0.0;
    if (bj >= bi) {
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        for (int k = ty; k < 32; k += 8) {
            const int r = bj * 32 + k, c = bi * 32 + tx;                 // mirror tile (bj, bi), row r, column c
            tile[k][tx] = (r < n && c < n) ? S[(size_t)r * ld + c] : 0.0;
        }
        __syncthreads();
        for (int k = ty; k < 32; k += 8) {
            const int r = bi * 32 + k, c = bj * 32 + tx;
            if (r < n && c < n) worst = fmax(worst, fabs(S[(size_t)r * ld + c] - tile[tx][k]));
        }
    }
    red[threadIdx.x] = worst;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + w]);
        __syncthreads();
    }
    if (threadIdx.x == 0 && red[0] > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(red[0]));
}
}  // namespace shg

extern "C" int shg_symmetry_defect(const double* S, int n, int ld, double* defect, void* stream_) {
    SHG_REQUIRE(n >= 0 && ld >= n, "shg_symmetry_defect: bad size");
    SHG_REQUIRE(defect != nullptr, "shg_symmetry_defect: NULL output");
    hipStream_t stream = (hipStream_t)stream_;
    if (zero_fill(defect, 1, 1, 1, stream) != SHG_OK) return SHG_ERR_HIP;
    if (n == 0) return SHG_OK;
    SHG_REQUIRE(S != nullptr, "shg_symmetry_defect: NULL matrix");
    const int nt = ceil_div(n, 32);
    hipLaunchKernelGGL(shg::symmetry_defect_kernel, dim3(nt, nt), dim3(256), 0, stream, n, S, ld, defect);     // non-negative doubles order like their bit patterns
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

static int covprop_diag_impl(shg_plan* p, const double* cov, int nmin, int lat0, int lat1, double* sigma, void* stream_, bool symmetric) {
    SHG_REQUIRE(p != nullptr, "shg_covprop_diag: NULL plan");
    SHG_REQUIRE(nmin >= 0 && nmin <= p->N + 1, "shg_covprop_diag: min_degree %d out of range", nmin);
    SHG_REQUIRE(lat0 >= 0 && lat1 <= p->nlat && lat0 <= lat1, "shg_covprop_diag: bad band [%d, %d)", lat0, lat1);
    if (lat0 == lat1) return SHG_OK;
    SHG_REQUIRE(sigma != nullptr, "shg_covprop_diag: NULL output");
    hipStream_t stream = (hipStream_t)stream_;
    PlanGuard guard(p, stream);
    const int Pfull = (p->N + 1) * (p->N + 1);
    const int Pn = Pfull - nmin * nmin;
    const long long M = (long long)(lat1 - lat0) * p->nlon;
    SHG_REQUIRE(M < (1LL << 31), "shg_covprop_diag: band too large");
    SHG_REQUIRE(Pn == 0 || cov != nullptr, "shg_covprop_diag: NULL covariance");

    int rc = build_pk_table(p, stream);
    if (rc) return rc;
    rc = covprop_build_cs_table(p, stream);
    if (rc) return rc;
    if (!p->pk_deg) {
        if (hipMalloc((void**)&p->pk_deg, ((size_t)p->nlat * Pfull + 64) * sizeof(double)) != hipSuccess ||      // + padding: scalar loads run a tile ahead
            hipMalloc((void**)&p->rslot, ((size_t)2 * Pfull + 48) * sizeof(int)) != hipSuccess)      // ranks | cos/sin row offsets + 48 entries of padding
            return fail(SHG_ERR_NOMEM, "covariance propagation tables: allocation failed");
        SHG_HIP(hipMemsetAsync(p->rslot, 0, ((size_t)2 * Pfull + 48) * sizeof(int), stream));
        SHG_HIP(hipMemsetAsync(p->pk_deg + (size_t)p->nlat * Pfull, 0, 64 * sizeof(double), stream));
        hipLaunchKernelGGL(covprop_pkd_kernel, dim3((unsigned)ceil_div64((long long)p->nlat * Pfull, 256)), dim3(256), 0, stream, p->N,
                           p->nlat, p->ldlat, p->nlon, p->pk, p->pk_deg, p->rslot, reinterpret_cast<unsigned*>(p->rslot + Pfull));
        SHG_HIP(hipGetLastError());
    }
    const int ncolblocks = std::max(1, ceil_div(Pn, BN));
    const size_t need = (size_t)ncolblocks * M;
    if (need > p->cov_partial_size) {
        if (p->cov_partial) {
            SHG_HIP(hipStreamSynchronize(stream));
            (void)hipFree(p->cov_partial);
            p->cov_partial = nullptr;
            p->cov_partial_size = 0;                 // a failed grow must not leave the old size behind
        }
        if (hipMalloc((void**)&p->cov_partial, need * sizeof(double)) != hipSuccess) return fail(SHG_ERR_NOMEM, "covariance propagation workspace (%zu doubles)", need);
        p->cov_partial_size = need;
    }
    if (Pn == 0) {
        SHG_HIP(hipMemsetAsync(sigma, 0, M * sizeof(double), stream));
        return SHG_OK;
    }
    // Two kernels: covprop_rows (covprop.hip) keeps row tiles inside one parallel (operand tiles are plain table rows) but
    // pads every parallel to a multiple of 128 meridians; the general kernel generates A element-wise and wastes nothing.
    const int padded = round_up(p->nlon, 128);
    if (!symmetric && (padded - p->nlon) * 25 <= p->nlon) {   // padding waste <= 4 %
        rc = covprop_rows(p, cov, Pn, nmin * nmin, lat0, lat1, p->cov_partial, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(covprop_reduce_kernel, dim3(ceil_div((int)M, 256)), dim3(256), 0, stream, (int)M, ncolblocks, p->cov_partial, sigma);
        SHG_HIP(hipGetLastError());
        return SHG_OK;
    }
    // The general kernel copies Sigma tiles straight into LDS when the rows of Sigma are 16-byte aligned.  An odd dimension
    // (d/o 180: 32761) or an odd base address gets a copy with rows of even length first: 2 x 8.6 GB of traffic, ~3 ms, against
    // seconds of MFMA work; below ~2 TFLOP of work the copy does not pay and the register path reads the matrix as it is.
    const double* sigma_in = cov;
    int ld_in = Pn;
    const bool aligned = (Pn % 2 == 0) && ((uintptr_t)cov % 16 == 0);
    if (!symmetric && !aligned && (double)M * Pn * Pn > 1e12) {
        const int ld_pad = Pn + (Pn & 1) + ((Pn + (Pn & 1)) % 512 == 0 ? 2 : 0);        // even, not a multiple of 4 KB
        const size_t need_pad = (size_t)Pn * ld_pad;
        if (need_pad > p->cov_pad_size) {
            if (p->cov_pad) {
                SHG_HIP(hipStreamSynchronize(stream));
                (void)hipFree(p->cov_pad);
                p->cov_pad = nullptr;
                p->cov_pad_size = 0;
            }
            if (hipMalloc((void**)&p->cov_pad, need_pad * sizeof(double)) == hipSuccess) p->cov_pad_size = need_pad;
            else (void)hipGetLastError();                                                // no room for the copy: register path
        }
        if (p->cov_pad) {
            SHG_HIP(hipMemcpy2DAsync(p->cov_pad, (size_t)ld_pad * sizeof(double), cov, (size_t)Pn * sizeof(double), (size_t)Pn * sizeof(double), Pn,
                                     hipMemcpyDeviceToDevice, stream));
            sigma_in = p->cov_pad;
            ld_in = ld_pad;
        }
    }
    return covprop_generic(p->pk_deg, Pfull, p->cs_slot, p->nlon, p->rslot, p->nlon, p->nlon, (long long)lat0 * p->nlon, (int)M, sigma_in, Pn,
                           nmin * nmin, p->cov_partial, sigma, p, stream, symmetric, false, reinterpret_cast<const unsigned*>(p->rslot + Pfull), p->nlat,
                           ld_in);
}
