// Shared declarations of libshg (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "../../include/shg.h"

namespace shg {

int fail(int code, const char* fmt, ...);
// Zero fills as kernels of this library (plan.hip), for the paths that several host threads feed at once (the chains of the
// block-banded smoother, one thread and one stream each): on ROCm 7.2 a 4-byte hipMemsetAsync issued from four threads on
// four streams left 0x80808080 in its target about four runs in five (the pivot flag of blockchol.hip's factorisation; with a
// kernel in its place, never).  rows x cols doubles with leading dimension ld (ld == cols: one contiguous run).
int zero_fill(double* p, long long ld, long long cols, long long rows, hipStream_t stream);
int zero_fill(int* p, hipStream_t stream);
// grow-only scratch of a stream (plan.hip), kept until shg_scratch_release(); one slot per buffer that is live at the same time
enum ScratchSlot { kScratchSplitK = 0, kScratchAnaFold = 1, kScratchAnaTransform = 2, kScratchAnaSolution = 3, kScratchAnaFlag = 4,
                   kScratchSeriesIn = 5, kScratchSeriesOut = 6 /* order-major copies of a batch in the reference layout (filters.hip) */,
                   kScratchBlocks = 8 /* and up: the workspaces of one block-matrix call (blockchol.hip) */ };
class ScratchLease {          // the scratch buffers of one stream, held while the operations that use them are enqueued (plan.hip)
public:
    explicit ScratchLease(hipStream_t stream);
    ~ScratchLease();
    ScratchLease(const ScratchLease&) = delete;
    ScratchLease& operator=(const ScratchLease&) = delete;
    void* get(int slot, size_t bytes);
    // two more streams of the same device for work that runs beside the leased stream (the look-ahead of blas.hip's
    // factorisation), with the events that order them; created on first use, kept with the scratch buffers
    int side(hipStream_t streams[2], hipEvent_t* to_side, hipEvent_t from_side[2]);
    int event(int i, hipEvent_t* ev);       // further events of the stream (0 .. 7), created on first use
    int sides_apart() const;                // how many of the side streams have a hardware queue of their own (0 .. 2)
private:
    void* owner_;
};
// The block row of a chain continues to the right of its diagonal block A [n x n] with ONE coupling block B [n x nc] (which
// becomes W = U^-T B) above the next diagonal block S [nc x nc] (which becomes the Schur complement S - W^T W).  The panel
// sweep over A simply carries on through B and S: step k solves its 128 rows of B, subtracts their products from the rows of B
// below and from S -- on the side stream, under the leaves, instead of two d^3 products (and their launches) after the sweep.
struct Coupling {
    double* B;
    int ldb, nc;
    long long strideB;
    double* S;
    int lds;
    long long strideS;
    hipEvent_t inverse_done;      // recorded on the side stream behind the last product of the inverse; the caller's stream does
                                  // not wait for it (lookahead_join does)
};

// gemm_tall.hip: C = alpha A B + beta C for tall A [M x K] and at most 240 columns (the dense filter product)
bool gemm_tall_shape(bool ta, bool tb, int M, int N, int K, int batch, bool upper_only, int tri, const double* A, int lda, const double* B, int ldb,
                     const double* C);
int gemm_tall(int M, int N, int K, double alpha, const double* A, int lda, const double* B, int ldb, double beta, double* C, int ldc, hipStream_t stream);

void stream_scratch_release();
hipError_t workspace_alloc(void** ptr, size_t bytes, hipStream_t stream);    // hipMallocAsync from a pool that keeps freed memory cached

#define SHG_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t _e = (call);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return shg::fail(SHG_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e),   \
                             __FILE__, __LINE__);                                                  \
    } while (0)

#define SHG_REQUIRE(cond, ...)                                                                     \
    do {                                                                                           \
        if (!(cond)) return shg::fail(SHG_ERR_INVALID, __VA_ARGS__);                               \
    } while (0)

// Knock-out switches for timing experiments (results are wrong with any of them on) exist in -DSHG_EXPERIMENT builds only
// (`make timeline` -> libshg_timeline.so, never loaded by the package): the shipping library reads no environment variable.
#ifdef SHG_EXPERIMENT
#define SHG_DBG(P, bits) ((P).dbg & (bits))
inline int experiment_switches() {
    const char* e = getenv("SHG_DEBUG");
    return e ? atoi(e) : 0;
}
#else
#define SHG_DBG(P, bits) 0
#if (defined(SHG_ROT_X) && SHG_ROT_X) || (defined(SHG_PIPE_X) && SHG_PIPE_X) || (defined(SHG_GEMM_X) && SHG_GEMM_X) || (defined(SHG_ANA_X) && SHG_ANA_X) || (defined(SHG_FILT_X) && SHG_FILT_X) || (defined(SHG_TALL_X) && SHG_TALL_X)
#error "SHG_ROT_X / SHG_GEMM_X / SHG_ANA_X / SHG_FILT_X are timing experiments: build with -DSHG_EXPERIMENT (make timeline)"
#endif
#endif

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel and device instead of in front of every launch: the call goes
// through the runtime's process-wide lock, and a smoother chain issues ~100 launches per epoch from several host threads.
#define SHG_SET_LDS_ONCE(fn, bytes)                                                                                        \
    do {                                                                                                                   \
        static std::atomic<unsigned long long> done_{0};                                                                   \
        int dev_ = 0;                                                                                                      \
        (void)hipGetDevice(&dev_);                                                                                         \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                                                               \
        if (!(done_.load(std::memory_order_acquire) & bit_)) {                                                             \
            SHG_HIP(hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)));     \
            done_.fetch_or(bit_, std::memory_order_release);                                                               \
        }                                                                                                                  \
    } while (0)

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// packed order-major index of (n, m), n >= m:  off(m) + n - m,  off(m) = m (N+1) - m (m-1) / 2
__host__ __device__ inline int order_offset(int N, int m) { return m * (N + 1) - (m * (m - 1)) / 2; }
__host__ __device__ inline int packed_count(int N) { return (N + 1) * (N + 2) / 2; }

// Exchange between the lanes 2q and 2q+1 that turns "4 rows x 1 column per lane" into "2 rows x 2 adjacent columns per lane":
//   even lanes: lo = x (own),                  hi = x of the odd neighbour
//   odd lanes:  lo = y of the even neighbour,  hi = y (own)
// One v_cndmask_b32 with a DPP source (quad_perm [1, 0, 3, 2]) per 32-bit half: the select of the value to send, the DPP move
// and the select of the received value collapse into one instruction.  fp64 MFMAs and VALU instructions of all waves
// of a SIMD share one issue pipe (tools/mfma64_issue.hip), so every VALU instruction saved here is MFMA time.
// `odd` = lane mask of the odd lanes.  All lanes must be active.
__device__ inline void pair_exchange(double x, double y, unsigned long long odd, double& lo, double& hi) {
    const long long xb = __builtin_bit_cast(long long, x), yb = __builtin_bit_cast(long long, y);
    const int x0 = (int)xb, x1 = (int)(xb >> 32), y0 = (int)yb, y1 = (int)(yb >> 32);
    int l0, l1, h0, h1;
    asm volatile(
        "s_nop 1\n\t"                                                   // VALU write -> DPP read of the same VGPR: 2 wait states
        "s_mov_b64 vcc, %[odd]\n\t"
        "v_cndmask_b32_dpp %[h0], %[x0], %[y0], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"     // odd ? y : x of the neighbour
        "v_cndmask_b32_dpp %[h1], %[x1], %[y1], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_not_b64 vcc, vcc\n\t"
        "v_cndmask_b32_dpp %[l0], %[y0], %[x0], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"     // even ? x : y of the neighbour
        "v_cndmask_b32_dpp %[l1], %[y1], %[x1], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
        : [l0] "=&v"(l0), [l1] "=&v"(l1), [h0] "=&v"(h0), [h1] "=&v"(h1)
        : [x0] "v"(x0), [x1] "v"(x1), [y0] "v"(y0), [y1] "v"(y1), [odd] "s"(odd)
        : "vcc", "scc");
    lo = __builtin_bit_cast(double, ((long long)l1 << 32) | (unsigned int)l0);
    hi = __builtin_bit_cast(double, ((long long)h1 << 32) | (unsigned int)h0);
}


// value of the lane 8 positions away inside the row of 16 lanes (DPP row_ror:8)
__device__ inline double swap_half_row(double x) {
    const long long bits = __builtin_bit_cast(long long, x);
    const int lo = (int)bits, hi = (int)(bits >> 32);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xF, 0xF, true);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi2 << 32) | (unsigned int)lo2);
}

constexpr int kRotMaxClasses = 6; // classes r = 0 .. R / 2 of the rotation-folded synthesis kernel, R <= 10
constexpr int kEpochTile = 8;    // epochs handled together by one wave of the Legendre stage
constexpr int kLatTile = 64;     // parallels per wave of the Legendre stage (lane <-> parallel)

}  // namespace shg

// Device-resident tables of one (degree, parallels, kn, meridians) configuration.
struct shg_plan {
    int device = 0;
    int N = 0, nlat = 0, nlon = 0;
    int ldlat = 0;          // nlat rounded up to 64: leading dimension of per-parallel tables / F
    bool sym4 = false;      // 4-fold longitude symmetry path
    int rotR = 0;           // rotations of the meridian set used by the rotation-folded kernel (synthesis_rot.hip): 10, 9, 6, 3 or 0 = not applicable
    std::vector<double> lon_host;   // meridians as given (the trig stream of that kernel is rebuilt when shg_plan_set_rotations changes R)
    double* rot_trig = nullptr; // [column tiles][k-steps][64 lanes][2] cos / signed sin stream of that kernel
    bool sym_ns = false;    // parallels (colatitude and kn rows) symmetric about the equator
    int ngroups = 1;        // 4 (sym4) or 1
    int goff[5] = {0, 0, 0, 0, 0};   // first K slot of each group (multiples of 4), goff[ngroups] = K
    int K = 0;              // total K slots of the longitude stage (multiple of 4)
    int ncol = 0;           // output columns per group: nlon/4 (sym4) or nlon
    int ncoltiles = 0;      // ceil(ncol / 16)
    int chunk = 16;         // epochs per pass

    // device tables
    double* ct = nullptr;       // [ldlat] cos(colat)
    double* st = nullptr;       // [ldlat] sin(colat)
    double* pmm = nullptr;      // [N+1][ldlat] sectorial seeds P_mm(theta_i)
    double* knT = nullptr;      // [N+1][ldlat] kn transposed
    double* arec = nullptr;     // [packed] recursion factor a_nm
    double* brec = nullptr;     // [packed] recursion factor b_nm
    double* trig = nullptr;     // [ncoltiles][K][16] cos/sin table, column-tile major
    // 64-row fused kernel: order 0 (constant along a parallel) leaves the K loop and becomes the start value of the cosine /
    // even-order accumulators, when that saves a body of 16 slots (d/o 96: 13 -> 12 bodies).  Slot K_f of the panel holds it.
    bool fold0 = false;
    int goff_f[5] = {0, 0, 0, 0, 0};
    int K_f = 0;
    double* trig_f = nullptr;   // [ncoltiles][K_f][16]
    double* lon = nullptr;      // [nlon]
    double* colat = nullptr;    // [nlat]
    // covariance-propagation tables (built lazily)
    double* pk_deg = nullptr;   // [nlat][Pfull] kn-scaled P_nm in degree-wise order (nmin = 0)
    double* cs_slot = nullptr;  // [2N+1][nlon] cos/sin per slot (0, 1c, 1s, 2c, 2s, ...)
    int* rslot = nullptr;       // [Pfull] rank inside its degree of every degree-wise index
    double* cov_partial = nullptr;   // [column blocks][band rows] partial row sums of the covariance kernel
    size_t cov_partial_size = 0;
    double* cov_pad = nullptr;       // covariance matrix copied to rows of even length (16-byte aligned rows for the LDS copy of the kernel)
    size_t cov_pad_size = 0;
    // workspace
    double* cpk = nullptr;      // [packed][2][chunk_pad] repacked coefficients of one pass
    double* F = nullptr;        // [chunk][K][ldlat] output of the Legendre stage
    int chunk_alloc = 0;
    // fused synthesis path (synthesis_fused.hip)
    double* pk = nullptr;       // [packed + 4][ldlat] kn-scaled Legendre table (two-kernel variant), built on first use
    double* pkf = nullptr;      // [nit][Qtot][64 lanes][2] the same table in MFMA-fragment order (fused kernel), built on first use
    int* qoff = nullptr;        // [N+2] first row-octet of every order in the fragment-ordered tables; qoff[N+1] = Qtot
    int Qtot = 0;
    std::vector<int> ns_badmap; // per block of 8 northern parallels: -1, or rank among the blocks whose mirrored parallels get their own table
    int ns_nbad = 0;
    int* badmap_d = nullptr;
    int* octinfo_d = nullptr;   // [Qtot] order | octet-in-order << 8 of every octet of the fragment-ordered tables
    int* itemtab_d = nullptr;   // work items of the fused kernel's Legendre stage, [8 waves][nrec][4]
    int itemtab_nrec = 0, itemtab_ntrip = 0;
    int itemtab_rot = -1;       // panel slot convention of the work items: 0 = 4-fold kernel, R = rotation-folded kernel
    const double* om_src = nullptr;   // set for the duration of shg_synthesis_om: the coefficient repack reads this order-major series
    int om_N = 0, om_Bpad = 0;
    int* sem_d = nullptr;       // token counter of the rotation-folded kernel's Legendre stage (synthesis_rot.hip), allocated by rot_set_stage_limit
    int stage_limit = 0;        // workgroups that may run their Legendre stage at once (0 = no limit, the default)
    int* blockmap_d = nullptr;  // XCD-aware (epoch tile, parallel tile) order of the fused kernel's workgroups
    int blockmap_nbt = 0, blockmap_nit = 0;
    std::vector<char> ns_badrow;    // per northern parallel: mirror image deviates too much to share the northern table
    // two-workgroup fused synthesis (synthesis_fused32.hip): blocks of 4 northern parallels
    double* pkf32 = nullptr;
    int* qoff32 = nullptr;
    int* badmap32_d = nullptr;
    int Qtot32 = 0, nbad32 = 0;
    int pkf_variant = 0;        // 1 plain fragment order, 2 north-south symmetric fragment order
    double* cpk4 = nullptr;     // repacked coefficients of the whole batch: [ceil(B/4)][Qtot][32][2] (fused) or [ceil(B/8)][packed][2][8]
    size_t cpk4_size = 0;
    int cpk4_variant = 0;       // layout the workspace was last zero-initialised for
    size_t cpk4_zeroed = 0;
    // analysis operator cache (analysis.hip): H[S][N+1][nlat] for the area weights ana_area and min degree ana_nmin
    double* ana_H = nullptr;
    double* ana_Hp = nullptr;   // north-south parity form of the operator [S][2 ceil((N+1)/2)][nlat/2] (analysis.hip), valid when ana_parity
    bool ana_parity = false;
    double ana_parity_defect = -1.0;   // largest dropped entry / largest entry of H when ana_Hp was formed (-1: not formed)
    double* ana_area = nullptr; // [nlat][nlon] copy of the area weights the operator was built for (compared on the device per call)
    int ana_nmin = -1;
    bool ana_rowconst = false;  // the weights of the cached operator are constant along every parallel (geographic and Gauss grids)
    double* ana_trig = nullptr; // trig table of the fused transform kernel in chunk order [chunk][8 columns][4 groups x MT x 16 orders], zero padded
    int ana_trig_mt = 0;
    int path = 0;               // 0 auto, 1 three-kernel path, 2 fused 4-fold kernel, 5 fused kernel with 32-row panels (two workgroups per CU), 6 rotation-folded fused kernel

    // users of the plan are serialised (PlanGuard): its tables are built lazily and its workspaces are per plan, not per stream
    std::mutex mtx;
    hipStream_t last_stream = nullptr;
    bool used = false;
    hipEvent_t order_event = nullptr;

    // optional per-kernel event timing (shg_plan_profile)
    bool profiling = false;
    unsigned profile_mask = ~0u;            // kinds that are timed while profiling is on
    std::vector<hipEvent_t> prof_events;    // pairs (start, stop)
    std::vector<int> prof_kinds;
    size_t prof_used = 0;                   // number of pairs in use
};

namespace shg {
int plan_alloc_workspace(shg_plan* p);
int fused_chunk_for(const shg_plan* p);
int build_pk_table(shg_plan* p, hipStream_t stream);
int build_pkf_table(shg_plan* p, bool ns, int rotR, hipStream_t stream);
int build_blockmap(shg_plan* p, int nbt, int nit, hipStream_t stream);
int pack_coefficients_fused(shg_plan* p, bool ns, int rotR, const double* anm, int B, hipStream_t stream);
int synthesis_fused(shg_plan* p, const double* anm, int B, double* grid, hipStream_t stream);
bool has_rotation_symmetry(int nlon, const double* lon, int R);
int rot_layout(int R, int N, int nk[kRotMaxClasses], int cnt[kRotMaxClasses], std::vector<int>* order_slot);
int rot_choose(int nlon, const double* lon_h, int N);
int rot_applicable(const shg_plan* p);
int build_rot_trig(shg_plan* p, const double* lon_h);
int synthesis_rot(shg_plan* p, const double* anm, int B, double* grid, hipStream_t stream);
int rot_set_stage_limit(shg_plan* p, int limit);
int fused32_applicable(const shg_plan* p);
int synthesis_fused32(shg_plan* p, const double* anm, int B, double* grid, hipStream_t stream);

// Entry points that work on a plan hold this for their duration: one host thread at a time, and a call on another stream than
// the previous one first waits (on the device) for everything that call has enqueued -- two streams or threads sharing a cached
// plan can then not overwrite each other's packed coefficients, panels or partial sums, and a reallocation never frees a
// buffer another stream still reads.  No cost while the stream stays the same.
struct PlanGuard {
    shg_plan* p;
    std::unique_lock<std::mutex> lock;
    PlanGuard(shg_plan* plan, hipStream_t stream) : p(plan), lock(plan->mtx) {
        if (p->used && p->last_stream != stream) {
            bool ordered = false;
            if (p->order_event || hipEventCreateWithFlags(&p->order_event, hipEventDisableTiming) == hipSuccess)
                ordered = hipEventRecord(p->order_event, p->last_stream) == hipSuccess && hipStreamWaitEvent(stream, p->order_event, 0) == hipSuccess;
            if (!ordered) {                      // e.g. the previous stream no longer exists
                (void)hipGetLastError();
                (void)hipDeviceSynchronize();
            }
        }
        p->last_stream = stream;
        p->used = true;
    }
};

// RAII event pair around one kernel launch (no-op unless profiling is enabled on the plan)
struct ProfileScope {
    shg_plan* p;
    hipStream_t stream;
    hipEvent_t stop = nullptr;
    ProfileScope(shg_plan* plan, int kind, hipStream_t s);
    ~ProfileScope();
};
}  // namespace shg
