// Plan creation: host-side generation of the small per-plan tables (recursion factors, sectorial
// seeds, cos/sin tables) and their upload.  Table arithmetic keeps the reference's expression order
// (grates/utilities.py:41-54) so that the Legendre values are bit-identical to the NumPy ones.
#include <cmath>
#include <cstring>
#include <string>

#include <map>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>

#include "common.h"

namespace shg {

static thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

// Stream-ordered scratch memory.  The default memory pool of a device gives freed memory back to the driver at the next
// synchronisation unless its release threshold is raised; with the threshold at its maximum hipMallocAsync / hipFreeAsync
// of per-call workspaces cost microseconds instead of a driver allocation each.
hipError_t workspace_alloc(void** ptr, size_t bytes, hipStream_t stream) {
    static thread_local int prepared_device = -1;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev != prepared_device) {
        hipMemPool_t pool;
        if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
            uint64_t threshold = UINT64_MAX;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &threshold);
        }
        prepared_device = dev;
    }
    return hipMallocAsync(ptr, bytes, stream);
}

__global__ void zero_fill_kernel(double* __restrict__ p, long long ld, long long cols, long long count) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (long long)gridDim.x * blockDim.x)
        p[ld == cols ? e : (e / cols) * ld + e % cols] = 0.0;
}

__global__ void zero_int_kernel(int* __restrict__ p) { *p = 0; }

int zero_fill(double* p, long long ld, long long cols, long long rows, hipStream_t stream) {
    if (rows <= 0 || cols <= 0) return SHG_OK;
    if (rows == 1) ld = cols;
    const long long count = rows * cols;
    const unsigned blocks = (unsigned)std::min<long long>(ceil_div64(count, 256), 256 * 16);
    hipLaunchKernelGGL(zero_fill_kernel, dim3(blocks), dim3(256), 0, stream, p, ld, cols, count);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

int zero_fill(int* p, hipStream_t stream) {
    hipLaunchKernelGGL(zero_int_kernel, dim3(1), dim3(1), 0, stream, p);
    SHG_HIP(hipGetLastError());
    return SHG_OK;
}

// Scratch that lives as long as its stream is in use: grow-only buffers per (stream, slot), handed to successive
// operations of that stream (which the stream orders, so they may share them).  For workspaces inside chains of thousands of
// small operations: on ROCm 7.2 hipFreeAsync keeps the calling thread until the stream has caught up (0.2 - 1.4 ms per call
// in the factorisation of a d = 1681 block; 0.2 ms for a 0.5 GB buffer even on an idle stream), which makes the host the
// pace-maker of such a chain.  A ScratchLease holds the stream's buffers exclusively while the operations that use them are
// being enqueued, so that two threads that feed the same stream cannot interleave their use of one buffer; leases nest
// inside a thread (shg_analysis holds one while gemm_ex takes its own).  shg_scratch_release() gives the buffers back.
namespace {
struct ScratchBuffer {
    void* ptr = nullptr;
    size_t size = 0;
};
struct StreamScratch {
    std::recursive_mutex mtx;
    int device = 0;
    hipStream_t stream = nullptr;
    std::map<int, ScratchBuffer> slots;
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t to_side = nullptr, from_side[2] = {nullptr, nullptr};
    hipEvent_t more[8] = {};
    int side_apart = 0;              // how many of the side streams run on hardware queues of their own (found by experiment)
};
// Keyed by (device, stream): the default stream is the null handle on every device, and a buffer allocated on one device must
// never be handed to a kernel of another.  Entries are created once and never erased, so a pointer to one stays valid.
// Lock order: g_scratch_mutex is only ever held for the map lookup itself and never while a stream mutex is taken -- a nested
// lease (shg_analysis -> gemm_ex) and a concurrent shg_scratch_release() can then not wait for each other in a cycle.
std::mutex g_scratch_mutex;
std::map<std::pair<int, hipStream_t>, std::unique_ptr<StreamScratch>> g_scratch;
}  // namespace

ScratchLease::ScratchLease(hipStream_t stream) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        std::unique_ptr<StreamScratch>& e = g_scratch[std::make_pair(dev, stream)];
        if (!e) {
            e.reset(new StreamScratch);
            e->device = dev;
            e->stream = stream;
        }
        owner_ = e.get();
    }
    static_cast<StreamScratch*>(owner_)->mtx.lock();
}

ScratchLease::~ScratchLease() { static_cast<StreamScratch*>(owner_)->mtx.unlock(); }

void* ScratchLease::get(int slot, size_t bytes) {
    ScratchBuffer& e = static_cast<StreamScratch*>(owner_)->slots[slot];
    if (e.size < bytes) {
        if (e.ptr) (void)hipFree(e.ptr);             // waits for the device: whatever still used the old buffer is done
        e.ptr = nullptr;
        e.size = 0;
        const size_t want = std::max(bytes + bytes / 4, (size_t)1 << 20);
        if (hipMalloc(&e.ptr, want) != hipSuccess) {
            e.ptr = nullptr;
            (void)hipGetLastError();
            return nullptr;
        }
        e.size = want;
    }
    return e.ptr;
}

// Side streams that really run beside the leased stream.  The runtime spreads its streams over a few hardware queues (four by
// default; the streams of one queue run one after the other), PyTorch alone has created dozens before this library is first
// called, and nothing in the API says which queue a stream got -- two streams created here one after the other shared one as
// often as not, and the look-ahead they were meant to carry ran in series.  So the queues are told apart by experiment, once per
// leased stream: a one-thread kernel that spins for 0.2 ms on one stream, a time stamp on the other; the stamp is earlier than
// the end of the spin exactly when the two streams do not share a queue.  (Priorities would separate the queues too, but with
// one stream above and one below the ordinary priority every kernel of the leased stream took 40 - 55 us on this card.)
__global__ void spin_kernel(long long ticks, long long* __restrict__ out) {
    const long long t0 = wall_clock64();
    for (int i = 0; i < 100000 && wall_clock64() - t0 < ticks; ++i) __builtin_amdgcn_s_sleep(8);       // (bounded whatever the counter does)
    out[0] = wall_clock64();
}

__global__ void stamp_kernel(long long* __restrict__ out) { out[0] = wall_clock64(); }

static bool queues_apart(hipStream_t a, hipStream_t b, long long* probe) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, a, 20000LL, probe);         // 0.2 ms of the 100 MHz counter
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, b, probe + 1);
    long long host[2] = {0, 0};
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess ||
        hipMemcpy(host, probe, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return host[1] < host[0];
}

int ScratchLease::side(hipStream_t streams[2], hipEvent_t* to_side, hipEvent_t from_side[2]) {
    StreamScratch* e = static_cast<StreamScratch*>(owner_);
    if (!e->to_side) {
        long long* probe = nullptr;
        SHG_HIP(hipMalloc((void**)&probe, 2 * sizeof(long long)));
        hipStream_t found[2] = {nullptr, nullptr};
        int nfound = 0;
        std::vector<hipStream_t> rejected;
        for (int attempt = 0; attempt < 12 && nfound < 2; ++attempt) {
            hipStream_t s = nullptr;
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) break;
            const bool ok = queues_apart(e->stream, s, probe) && (nfound == 0 || (queues_apart(found[0], s, probe) && queues_apart(s, found[0], probe)));
            if (ok)
                found[nfound++] = s;
            else
                rejected.push_back(s);
        }
        // (fewer than two such streams: the result is the same, the streams just take turns)
        for (int i = nfound; i < 2; ++i) {
            if (rejected.empty()) {
                (void)hipFree(probe);
                return fail(SHG_ERR_HIP, "side streams could not be created");
            }
            found[i] = rejected.back();
            rejected.pop_back();
        }
        for (hipStream_t s : rejected) (void)hipStreamDestroy(s);
        (void)hipFree(probe);
        hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
        bool created = true;
        for (int i = 0; i < 3 && created; ++i) created = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) == hipSuccess;
        if (!created) {                                  // nothing is kept: the next call starts over
            for (hipEvent_t x : ev)
                if (x) (void)hipEventDestroy(x);
            for (hipStream_t s : found) (void)hipStreamDestroy(s);
            (void)hipGetLastError();
            return fail(SHG_ERR_HIP, "events of the side streams could not be created");
        }
        e->side_apart = nfound;
        for (int i = 0; i < 2; ++i) {
            e->side[i] = found[i];
            e->from_side[i] = ev[i];
        }
        e->to_side = ev[2];
    }
    for (int i = 0; i < 2; ++i) {
        streams[i] = e->side[i];
        from_side[i] = e->from_side[i];
    }
    *to_side = e->to_side;
    return SHG_OK;
}

int ScratchLease::sides_apart() const { return static_cast<StreamScratch*>(owner_)->side_apart; }

int ScratchLease::event(int i, hipEvent_t* ev) {
    StreamScratch* e = static_cast<StreamScratch*>(owner_);
    SHG_REQUIRE(i >= 0 && i < 8, "ScratchLease::event: index %d", i);
    if (!e->more[i]) SHG_HIP(hipEventCreateWithFlags(&e->more[i], hipEventDisableTiming));
    *ev = e->more[i];
    return SHG_OK;
}

void stream_scratch_release() {
    std::vector<StreamScratch*> entries;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        for (auto& kv : g_scratch) entries.push_back(kv.second.get());
    }
    int current = 0;
    (void)hipGetDevice(&current);
    for (StreamScratch* e : entries) {
        std::lock_guard<std::recursive_mutex> hold(e->mtx);
        if (e->slots.empty() && !e->to_side) continue;
        (void)hipSetDevice(e->device);
        for (auto& b : e->slots)
            if (b.second.ptr) (void)hipFree(b.second.ptr);
        e->slots.clear();
        // the side streams of the look-ahead and every event of the entry (the caller has drained the device): the next
        // factorisation on this stream repeats the queue experiment
        for (int i = 0; i < 2; ++i) {
            if (e->side[i]) (void)hipStreamDestroy(e->side[i]);
            if (e->from_side[i]) (void)hipEventDestroy(e->from_side[i]);
            e->side[i] = nullptr;
            e->from_side[i] = nullptr;
        }
        if (e->to_side) (void)hipEventDestroy(e->to_side);
        e->to_side = nullptr;
        for (hipEvent_t& x : e->more) {
            if (x) (void)hipEventDestroy(x);
            x = nullptr;
        }
        e->side_apart = 0;
    }
    (void)hipSetDevice(current);
}

// a_nm / b_nm of  P_nm = (a_nm t) P_(n-1)m - b_nm P_(n-2)m   in packed order-major layout.
//   n == m     : unused (seed), a = b = 0
//   n == m + 1 : a = sqrt(2n + 1), b = 0                                    utilities.py:45-47
//   n >= m + 2 : a = sqrt((2n-1)/(n-m) (2n+1)/(n+m)),
//                b = sqrt((2n+1)/(2n-3) (n-m-1)/(n-m) (n+m-1)/(n+m))        utilities.py:49-54
void recursion_tables(int N, std::vector<double>& a, std::vector<double>& b) {
    a.assign(packed_count(N), 0.0);
    b.assign(packed_count(N), 0.0);
    for (int m = 0; m <= N; ++m) {
        for (int ni = m + 1; ni <= N; ++ni) {
            const int idx = order_offset(N, m) + ni - m;
            const double n = ni, mm = m;
            if (ni == m + 1) {
                a[idx] = std::sqrt((double)(2 * ni + 1));
            } else {
                a[idx] = std::sqrt((2.0 * n - 1.0) / (n - mm) * (2.0 * n + 1.0) / (n + mm));
                b[idx] = std::sqrt((2.0 * n + 1.0) / (2.0 * n - 3.0) * (n - mm - 1.0) / (n - mm) * (n + mm - 1.0) / (n + mm));
            }
        }
    }
}

static int upload(double** dst, const std::vector<double>& src) {
    SHG_HIP(hipMalloc((void**)dst, std::max<size_t>(src.size(), 1) * sizeof(double)));
    if (!src.empty()) SHG_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(double), hipMemcpyHostToDevice));
    return SHG_OK;
}

// True when meridians obey lon[nlon-1-j] = -lon[j], lon[nlon/2-1-j] = -pi - lon[j], lon[nlon/2+j] = lon[j] + pi
// to within a few ulp of pi (any equi-angular cell-centred grid, grates/grid.py:1149, 1186).
static bool has_fourfold_symmetry(int nlon, const double* lon) {
    if (nlon < 4 || nlon % 4 != 0) return false;
    const double tol = 2e-15;
    const double pi = 3.14159265358979323846;
    for (int j = 0; j < nlon / 4; ++j) {
        if (std::fabs(lon[nlon - 1 - j] + lon[j]) > tol) return false;
        if (std::fabs(lon[nlon / 2 - 1 - j] + pi + lon[j]) > tol) return false;
        if (std::fabs(lon[nlon / 2 + j] - pi - lon[j]) > tol) return false;
    }
    return true;
}

// True when the parallels are symmetric about the equator: colat[nlat-1-i] = pi - colat[i] and equal degree factors, up to
// the rounding of the caller's geometry (grates/gravityfield.py:353-356 evaluates both hemispheres independently; near the
// poles colatitude = arccos(...) is ill conditioned and mirrored values differ by up to 1.5e-14 rad on a 0.25 degree grid).
// Blocks of 8 northern parallels whose mirror images deviate by more than 1e-13 / (N + 1) rad or 5e-14 relative in kn are
// flagged in `badmap`: the fused kernel evaluates their southern rows from a table of their own instead of (-1)^(n-m)
// times the northern one, which keeps the deviation from an independent evaluation below ~1e-13 of the field maximum.
static bool has_north_south_symmetry(int N, int nlat, const double* colat, const double* kn, std::vector<int>& badmap, int& nbad,
                                     std::vector<char>& badrow) {
    badmap.clear();
    badrow.clear();
    nbad = 0;
    if (nlat < 2 || nlat % 2 != 0) return false;
    const double pi = 3.14159265358979323846;
    const int nh = nlat / 2;
    badmap.assign(ceil_div(nh, 8), -1);
    badrow.assign(nh, 0);
    for (int i = 0; i < nh; ++i) {
        const int mi = nlat - 1 - i;
        const double dtheta = std::fabs(colat[i] + colat[mi] - pi);
        if (dtheta > 1e-11) return false;
        bool bad = dtheta * (N + 1) > 1e-13;
        for (int n = 0; n <= N; ++n) {
            const double a = kn[(size_t)i * (N + 1) + n], b = kn[(size_t)mi * (N + 1) + n];
            const double d = std::fabs(a - b), s = std::max(std::fabs(a), std::fabs(b));
            if (d > 1e-10 * s) return false;
            if (d > 5e-14 * s) bad = true;
        }
        badrow[i] = bad ? 1 : 0;
        if (bad && badmap[i >> 3] < 0) badmap[i >> 3] = nbad++;
    }
    return true;
}

int plan_alloc_workspace(shg_plan* p) {
    if (p->chunk_alloc == p->chunk && p->F) return SHG_OK;
    if (p->cpk) (void)hipFree(p->cpk);
    if (p->F) (void)hipFree(p->F);
    p->cpk = p->F = nullptr;
    const int chunk_pad = round_up(p->chunk, kEpochTile);
    const size_t ncpk = (size_t)packed_count(p->N) * 2 * chunk_pad;
    const size_t nF = (size_t)chunk_pad * p->K * p->ldlat;
    if (hipMalloc((void**)&p->cpk, ncpk * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&p->F, nF * sizeof(double)) != hipSuccess)
        return fail(SHG_ERR_NOMEM, "workspace allocation failed (%zu + %zu doubles)", ncpk, nF);
    // padding slots / padding epochs must hold finite numbers: they meet zero table rows in the MFMA
    SHG_HIP(hipMemset(p->cpk, 0, ncpk * sizeof(double)));
    SHG_HIP(hipMemset(p->F, 0, nF * sizeof(double)));
    p->chunk_alloc = p->chunk;
    return SHG_OK;
}

ProfileScope::ProfileScope(shg_plan* plan, int kind, hipStream_t s) : p(plan), stream(s) {
    if (!p || !p->profiling || !((p->profile_mask >> kind) & 1)) return;
    if (p->prof_used * 2 + 2 > p->prof_events.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        p->prof_events.push_back(a);
        p->prof_events.push_back(b);
        p->prof_kinds.push_back(kind);
    }
    p->prof_kinds[p->prof_used] = kind;
    (void)hipEventRecord(p->prof_events[p->prof_used * 2], stream);
    stop = p->prof_events[p->prof_used * 2 + 1];
    p->prof_used++;
}

ProfileScope::~ProfileScope() {
    if (stop) (void)hipEventRecord(stop, stream);
}

}  // namespace shg

using namespace shg;

extern "C" int shg_plan_profile(shg_plan* p, int enable) {
    SHG_REQUIRE(p != nullptr, "shg_plan_profile: NULL plan");
    // enable: 0 off, 1 every kind, otherwise a mask of kinds shifted by one (bit k + 1 = kind k): an event pair costs the stream ~5 us
    p->profiling = enable != 0;
    p->profile_mask = enable == 1 ? ~0u : (unsigned)enable >> 1;
    return SHG_OK;
}

extern "C" int shg_plan_profile_read(shg_plan* p, double ms[SHG_PROFILE_KINDS], int64_t launches[SHG_PROFILE_KINDS]) {
    SHG_REQUIRE(p != nullptr && ms != nullptr && launches != nullptr, "shg_plan_profile_read: NULL argument");
    for (int k = 0; k < SHG_PROFILE_KINDS; ++k) {
        ms[k] = 0.0;
        launches[k] = 0;
    }
    for (size_t e = 0; e < p->prof_used; ++e) {
        SHG_HIP(hipEventSynchronize(p->prof_events[e * 2 + 1]));
        float t = 0.f;
        SHG_HIP(hipEventElapsedTime(&t, p->prof_events[e * 2], p->prof_events[e * 2 + 1]));
        const int k = p->prof_kinds[e];
        if (k >= 0 && k < SHG_PROFILE_KINDS) {
            ms[k] += t;
            launches[k] += 1;
        }
    }
    p->prof_used = 0;
    return SHG_OK;
}

extern "C" const char* shg_last_error(void) { return g_last_error.c_str(); }
extern "C" int shg_scratch_release(void) {
    SHG_HIP(hipDeviceSynchronize());
    shg::stream_scratch_release();
    return SHG_OK;
}

extern "C" const char* shg_version(void) { return "libshg 0.1 (gfx950)"; }

extern "C" int shg_plan_create(shg_plan** out, int N, int nlat, const double* colat_h, const double* kn_h,
                               int nlon, const double* lon_h, int device) {
    SHG_REQUIRE(out != nullptr, "shg_plan_create: out is NULL");
    *out = nullptr;
    SHG_REQUIRE(N >= 0 && N <= 2047, "shg_plan_create: degree %d out of range [0, 2047]", N);
    SHG_REQUIRE(nlat > 0 && nlon > 0, "shg_plan_create: empty grid (%d x %d)", nlat, nlon);
    SHG_REQUIRE(colat_h && kn_h && lon_h, "shg_plan_create: NULL table pointer");
    SHG_HIP(hipSetDevice(device));

    shg_plan* p = new shg_plan();
    p->device = device;
    p->N = N;
    p->nlat = nlat;
    p->nlon = nlon;
    p->ldlat = round_up(nlat, kLatTile);
    p->sym4 = has_fourfold_symmetry(nlon, lon_h);
    p->sym_ns = has_north_south_symmetry(N, nlat, colat_h, kn_h, p->ns_badmap, p->ns_nbad, p->ns_badrow);
    // rotation-folded kernel: the largest rotation count the meridians allow (0.25 degree grid: 10, 0.5 degree grid: 3)
    p->rotR = rot_choose(nlon, lon_h, N);
    p->lon_host.assign(lon_h, lon_h + nlon);

    // ---- K slots of the longitude stage
    if (p->sym4) {
        const int cnt[4] = {N / 2 + 1, (N + 1) / 2, N / 2, (N + 1) / 2};   // cos even, cos odd, sin even, sin odd
        p->ngroups = 4;
        p->goff[0] = 0;
        // groups padded to whole bodies of 4 MFMA k-steps (16 slots): the fused kernel then needs no remainder handling
        for (int g = 0; g < 4; ++g) p->goff[g + 1] = p->goff[g] + round_up(cnt[g], 16);
        p->ncol = nlon / 4;
    } else {
        p->ngroups = 1;
        p->goff[0] = 0;
        p->goff[1] = round_up(2 * N + 1, 4);
        p->ncol = nlon;
    }
    p->K = p->goff[p->ngroups];
    p->ncoltiles = ceil_div(p->ncol, 16);

    // ---- per-parallel tables, padded to ldlat with zeros
    std::vector<double> ct(p->ldlat, 0.0), st(p->ldlat, 0.0);
    std::vector<double> pmm((size_t)(N + 1) * p->ldlat, 0.0), knT((size_t)(N + 1) * p->ldlat, 0.0);
    for (int i = 0; i < nlat; ++i) {
        ct[i] = std::cos(colat_h[i]);
        st[i] = std::sin(colat_h[i]);
        double pv = 1.0;                                             // P_00
        pmm[i] = pv;
        for (int n = 1; n <= N; ++n) {
            if (n == 1)
                pv = std::sqrt(3.0) * st[i];                         // utilities.py:39
            else
                pv = std::sqrt((2.0 * n + 1.0) / (2.0 * n)) * st[i] * pv;   // utilities.py:42-43
            pmm[(size_t)n * p->ldlat + i] = pv;
        }
        for (int n = 0; n <= N; ++n) knT[(size_t)n * p->ldlat + i] = kn_h[(size_t)i * (N + 1) + n];
    }
    std::vector<double> a, b;
    recursion_tables(N, a, b);

    // ---- cos/sin table [coltile][K][16]
    // padded to whole column blocks of the fused kernel plus one chunk of rows (its prefetch is unconditional)
    std::vector<double> trig(((size_t)round_up(p->ncoltiles, 8) * p->K + 16) * 16, 0.0);
    auto put = [&](int slot, int m, bool sine) {
        for (int j = 0; j < p->ncol; ++j) {
            const double arg = (double)m * lon_h[j];                 // utilities.py:272-273: cos(m * lon)
            trig[((size_t)(j / 16) * p->K + slot) * 16 + (j % 16)] = sine ? std::sin(arg) : std::cos(arg);
        }
    };
    if (p->sym4) {
        for (int m = 0; m <= N; ++m) {
            put(p->goff[m & 1] + m / 2, m, false);
            if (m >= 1) put(p->goff[2 + (m & 1)] + ((m & 1) ? m / 2 : m / 2 - 1), m, true);
        }
    } else {
        for (int m = 0; m <= N; ++m) {
            put(m, m, false);
            if (m >= 1) put(N + m, m, true);
        }
    }

    // the same table without order 0 (see fold0 in common.h)
    std::vector<double> trig_f;
    if (p->sym4) {
        const int cnt_f[4] = {N / 2, (N + 1) / 2, N / 2, (N + 1) / 2};
        p->goff_f[0] = 0;
        for (int g = 0; g < 4; ++g) p->goff_f[g + 1] = p->goff_f[g] + round_up(cnt_f[g], 16);
        p->K_f = p->goff_f[4];
        p->fold0 = p->K_f < p->K;
        if (p->fold0) {
            trig_f.assign(((size_t)round_up(p->ncoltiles, 8) * p->K_f + 16) * 16, 0.0);
            for (int m = 1; m <= N; ++m)
                for (int sine = 0; sine < 2; ++sine) {
                    int slot;
                    if (!sine) {
                        if (m & 1) slot = p->goff_f[1] + m / 2; else slot = p->goff_f[0] + m / 2 - 1;
                    } else {
                        if (m & 1) slot = p->goff_f[3] + m / 2; else slot = p->goff_f[2] + m / 2 - 1;
                    }
                    for (int j = 0; j < p->ncol; ++j) {
                        const double arg = (double)m * lon_h[j];
                        trig_f[((size_t)(j / 16) * p->K_f + slot) * 16 + (j % 16)] = sine ? std::sin(arg) : std::cos(arg);
                    }
                }
        }
    }

    int rc = SHG_OK;
    if ((p->fold0 && (rc = upload(&p->trig_f, trig_f))) || (rc = upload(&p->ct, ct)) || (rc = upload(&p->st, st)) || (rc = upload(&p->pmm, pmm)) ||
        (rc = upload(&p->knT, knT)) || (rc = upload(&p->arec, a)) || (rc = upload(&p->brec, b)) ||
        (rc = upload(&p->trig, trig)) ||
        (rc = upload(&p->lon, std::vector<double>(lon_h, lon_h + nlon))) ||
        (rc = upload(&p->colat, std::vector<double>(colat_h, colat_h + nlat)))) {
        shg_plan_destroy(p);
        return rc;
    }
    if (rot_applicable(p) && (rc = build_rot_trig(p, lon_h))) {
        shg_plan_destroy(p);
        return rc;
    }
    *out = p;
    return SHG_OK;
}

extern "C" int shg_plan_destroy(shg_plan* p) {
    if (!p) return SHG_OK;
    double* ptrs[] = {p->ct, p->st, p->pmm, p->knT, p->arec, p->brec, p->trig, p->trig_f, p->lon, p->colat,
                      p->pk_deg, p->cs_slot, p->cpk, p->F, p->pk, p->pkf, p->pkf32, p->cpk4, p->cov_partial, p->cov_pad, p->ana_H, p->ana_Hp, p->ana_area, p->ana_trig, p->rot_trig};
    if (p->rslot) (void)hipFree(p->rslot);
    if (p->qoff) (void)hipFree(p->qoff);
    if (p->badmap_d) (void)hipFree(p->badmap_d);
    if (p->blockmap_d) (void)hipFree(p->blockmap_d);
    if (p->sem_d) (void)hipFree(p->sem_d);
    if (p->itemtab_d) (void)hipFree(p->itemtab_d);
    if (p->octinfo_d) (void)hipFree(p->octinfo_d);
    if (p->qoff32) (void)hipFree(p->qoff32);
    if (p->badmap32_d) (void)hipFree(p->badmap32_d);
    for (double* q : ptrs)
        if (q) (void)hipFree(q);
    for (hipEvent_t e : p->prof_events) (void)hipEventDestroy(e);
    if (p->order_event) (void)hipEventDestroy(p->order_event);
    delete p;
    return SHG_OK;
}

extern "C" int shg_plan_set_chunk(shg_plan* p, int epochs_per_pass) {
    SHG_REQUIRE(p != nullptr, "shg_plan_set_chunk: NULL plan");
    SHG_REQUIRE(epochs_per_pass >= 1 && epochs_per_pass <= 4096, "shg_plan_set_chunk: %d out of range", epochs_per_pass);
    p->chunk = epochs_per_pass;
    return SHG_OK;
}

extern "C" int shg_plan_set_path(shg_plan* p, int path) {
    SHG_REQUIRE(p != nullptr, "shg_plan_set_path: NULL plan");
    SHG_REQUIRE(path == 0 || path == 1 || path == 2 || path == 5 || path == 6, "shg_plan_set_path: path %d not in {0, 1, 2, 5, 6}", path);
    SHG_REQUIRE(path != 5 || fused32_applicable(p), "shg_plan_set_path: the two-workgroup fused kernel needs both grid symmetries and K <= 416 (K = %d)", p->K);
    SHG_REQUIRE(path != 2 || fused_chunk_for(p) != 0, "shg_plan_set_path: fused kernel not applicable (needs 4-fold symmetric meridians and K <= 224, K = %d)", p->K);
    SHG_REQUIRE(path < 6 || rot_applicable(p), "shg_plan_set_path: rotation-folded kernel not applicable (needs equi-angular meridians with nlon %% 96 == 0 or nlon %% 48 == 0, nlon >= 192, and a panel within the LDS)");
    p->path = path;
    return SHG_OK;
}

extern "C" int shg_plan_set_stage_limit(shg_plan* p, int limit) {
    SHG_REQUIRE(p != nullptr, "shg_plan_set_stage_limit: NULL plan");
    std::unique_lock<std::mutex> lock(p->mtx);
    return rot_set_stage_limit(p, limit);
}

extern "C" int shg_plan_set_rotations(shg_plan* p, int R) {
    SHG_REQUIRE(p != nullptr, "shg_plan_set_rotations: NULL plan");
    std::unique_lock<std::mutex> lock(p->mtx);
    const int want = R == 0 ? rot_choose(p->nlon, p->lon_host.data(), p->N) : R;
    SHG_REQUIRE(R == 0 || ((R == 3 || R == 6 || R == 9 || R == 10) && has_rotation_symmetry(p->nlon, p->lon_host.data(), R)),
                "shg_plan_set_rotations: the meridians are not invariant under %d rotations in whole 128-byte lines (R in {3, 6, 9, 10}, nlon %% 2R == 0, (nlon / R) %% 16 == 0)", R);
    if (want == p->rotR) return SHG_OK;
    const int before = p->rotR;
    p->rotR = want;
    if (want != 0 && !rot_applicable(p)) {
        p->rotR = before;
        return fail(SHG_ERR_UNSUPPORTED, "shg_plan_set_rotations: the panel of %d rotations at degree %d does not fit the LDS", want, p->N);
    }
    SHG_HIP(hipDeviceSynchronize());               // the trig stream of the old count may still be in use
    if (rot_applicable(p)) {
        const int rc = build_rot_trig(p, p->lon_host.data());
        if (rc) return rc;
    }
    return SHG_OK;
}

extern "C" int shg_plan_info(const shg_plan* p, int64_t which[8]) {
    SHG_REQUIRE(p != nullptr && which != nullptr, "shg_plan_info: NULL argument");
    which[0] = p->N;
    which[1] = p->nlat;
    which[2] = p->nlon;
    which[3] = (p->sym4 ? 1 : 0) | (p->sym_ns ? 2 : 0) | (rot_applicable(p) ? 4 : 0);
    which[4] = p->chunk;
    which[5] = p->K;
    which[6] = (p->path >= 2 || (p->path == 0 && (rot_applicable(p) || fused_chunk_for(p) != 0 || fused32_applicable(p)))) ? 1 : 0;
    which[7] = p->path | (rot_applicable(p) ? p->rotR << 8 : 0);
    return SHG_OK;
}
