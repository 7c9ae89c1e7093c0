// wgbsseg.hip — host side of libwgbsseg.so: the C ABI of include/wgbsseg.h over the gfx950 kernels of
// seg_kernels.h.  One context = one GPU (own streams, grow-only scratch in HBM).  No CPU compute path exists
// here: without a HIP device every entry point fails.
//
// Build (see wgbs_tools_amd/build.py):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared wgbsseg.hip -o libwgbsseg.so
// -ffp-contract=off is REQUIRED: the likelihood term must round exactly like the reference's x86-64 build
// (no fused multiply-add anywhere; SURVEY.md 7 hard part 2).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <emmintrin.h>
#endif

#include "../../include/wgbsseg.h"
#include "seg_kernels.h"
#include "plain_dp.h"
#include "stitch.h"
#include "add_loci.h"
#include "table_io.h"

namespace {

inline double wall_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// WGBSSEG_PROFILE=1: where the host time of a call goes (allocations, upload), on stderr
inline bool profiling() { static const bool on = getenv("WGBSSEG_PROFILE") && atoi(getenv("WGBSSEG_PROFILE")); return on; }
std::atomic<long long> g_alloc_us(0), g_alloc_bytes(0), g_alloc_calls(0);

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        const double t0 = profiling() ? wall_s() : 0.0;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { e = hipMalloc(&p, bytes); want = bytes; }
        if (e == hipSuccess) cap = want;
        if (profiling()) { g_alloc_us += (long long)((wall_s() - t0) * 1e6); g_alloc_bytes += (long long)want; g_alloc_calls += 1; }
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

void set_err(char* err, size_t errlen, const char* fmt, ...)
{
    if (!err || !errlen) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, errlen, fmt, ap);
    va_end(ap);
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            set_err(err, errlen, "HIP error %d (%s) at %s:%d: %s", (int)e__, hipGetErrorString(e__), \
                    __FILE__, __LINE__, #expr);                                                    \
            return WGBSSEG_E_HIP;                                                                  \
        }                                                                                          \
    } while (0)

hipError_t set_kernel_attributes();      // defined below, next to the kernel launchers
inline int ceil_pow2(int v) { int r = 1; while (r < v) r <<= 1; return r; }
inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

}  // namespace

struct PinnedBuf {          // grow-only page-locked host buffer (fast, truly asynchronous D2H)
    void* p = nullptr;
    size_t cap = 0;
    bool ensure(size_t bytes)
    {
        if (bytes <= cap) return true;
        const double t0 = profiling() ? wall_s() : 0.0;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return false; }
        cap = want;
        if (profiling()) { g_alloc_us += (long long)((wall_s() - t0) * 1e6); g_alloc_bytes += (long long)want; g_alloc_calls += 1; }
        return true;
    }
    bool ensure_exact(size_t bytes)
    {
        if (bytes <= cap) return true;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { p = nullptr; return false; }
        cap = bytes;
        return true;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

#define WG_MAX_CPUS 8192
struct NearCpus { std::vector<unsigned long> bits; bool valid = false; };      // affinity mask of the CPUs next to a device

struct wgbsseg_ctx {
    int device = 0;
    NearCpus near_cpus;          // the upload threads run there (worked out once, at create)
    hipStream_t sA = nullptr, sB = nullptr, sC = nullptr;     // scoring (+ everything else) | recurrence, traceback | the scan pass
    hipStream_t sA2 = nullptr;                                // second scoring stream: medium / wide tiles beside the narrow ones (stage loop)
    // inputs
    DevBuf betas_own, loci_own;
    const uint8_t* betas = nullptr;
    int64_t pitch = 0, n_total = 0;
    int32_t elem = 1;        // bytes per count of the resident rows: 1 = .beta / .bin (uint8 pairs), 2 = .lbeta (uint16 pairs; block sums only)
    int64_t site_base = 0;   // absolute 0-based index of resident site 0 (messages only; all call coordinates are resident-relative)
    int32_t n_samples = 0;
    const uint32_t* loci = nullptr;
    int64_t n_loci = 0;
    // scratch
    DevBuf chunks, wtile, carry, W16, cum32, back16, chunk_pairs, status, tile_tot, tile_base, tile_chunk;
    DevBuf plan_cbase, plan_cum0, plan_tbase, plan_pairs, plan_tiles, plan_cnt, tilesA, tilesB, tilesM, umax16;
    std::vector<PinnedBuf> pinned;
    std::vector<PinnedBuf> up_stage;   // two page-locked staging pieces per upload thread (set_betas_host)
    DevBuf cost[3], stage_ctr, dpstate, tmp_borders, nb, boff, out_borders[2], edges, dbg_a, dbg_b, dbg_c, lookup;
    int out_par = 0;              // which of the two result buffers the batch in flight writes: the lists of the previous batch may still be on their way home (below)
    // early delivery (segment_regions' first batch): k_copy_out (on the scan stream) writes the chunks' border lists into the page-locked result ...
    hipEvent_t evS = nullptr;     // the scan stream's last command of the batch in flight (the copy of its verdict)
    hipEvent_t evD = nullptr;     // ... while the host already rehearses the junctions on the lists' edges and runs the follow-up batch; evD: the lists are home
    bool out_pending = false;
    bool batch_open = false;      // a batch is between its first launch and its clean end (an error return leaves it set: the next batch drains the device first)
    std::vector<wg_d2> h_lookup;   // host copy of the k-scaled log tables of the call in flight (source of an async upload)
    float lookup_pc = -1.0f;       // what the device copy `lookup` was built for: pseudo count and exponent rows of the narrow / wide / medium class
    int lookup_rows[3] = {-1, -1, -1};
    // events
    hipEvent_t ev[13] = {};  // [0] batch begins, [1] windows done, [7] statistics copied, [3] plan done, [4]-[6] traceback / borders; the scan pass (stream C): [12] inputs uploaded,
                             // [8]..[9] k_validate, [10]..[11] k_scan (jobs with wide tiles only), [2] its verdict may be copied
    PinnedBuf h_status;      // page-locked landing area of the status words: D2H copies that really are asynchronous
    PinnedBuf h_job, h_pieces, h_sb;   // ... the staging areas of a job's tables on their way up (chunk table, window tiles, cleared status | k_validate's pieces)
    PinnedBuf h_out;         // ... and of a batch's CSR offsets (a small batch: of its border lists too)
    std::vector<hipEvent_t> ev_cost0, ev_cost1, ev_dp0, ev_dp1, ev_fork, ev_join;     // per stage
    // last-call info
    wgbsseg_timings tim = {};
    int64_t last_sites = 0, last_pairs = 0;
    int32_t last_stages = 0;
    bool last_valid = false;
    long long cost_budget_bytes = 0;
    int force_stages = 0;
    bool counted_live = false;
    double stage_gate_wide_evals = 100000.0;                  // jobs with medium / wide tiles are gated from this many evaluations per step on
    bool stage_gate_shared = false;                           // (tests) gate even when other contexts live on the device, and whatever the job's tiles
    int stage_gate = 768;                                     // > 0: the stages of a staged all-narrow job alternate between the two scoring streams behind k_stage_gate; the value = tiles of slack
    double stage_min_evals_per_step = 21000.0;                // staging for few chunks only from this many evaluations per step of the longest chunk
    int last_stage_pct = -1;                                  // length of a staged job's LAST stage (whose recurrence nothing hides) in percent of the other stages'
    int force_dp_mode = 0;   // WGBSSEG_DP_MODE: 1 = 32-step batches (wide-window path), 2 = the same with 15 worker waves
    int force_ns = 0;
    int force_ti = 0;
    double last_block_sums_ms = 0.0;
    int64_t table_blocks = 0;  // > 0: dbg_b still holds the [n_samples][table_blocks] ratio table of the last mode-3 block reduction
    bool accumulate = false;   // add to `tim` instead of resetting it (region-level calls span several batches)
    // site ranges whose `meth <= cov` check has run in the API call in flight (sorted, disjoint): the follow-up batches of a
    // region-level call hold junction patches only, all inside chunks its first batch has validated.  Never kept across
    // API calls: device-resident betas handed over by pointer may change between them.
    std::vector<std::pair<int64_t, int64_t>> validated;
    DevBuf scan_pieces, divcheck, plan_sb, bs_desc;
    std::vector<int32_t> h_stage_bounds;
    // the short division core of the narrow scoring tiles: verified on the device per pseudo count (k_check_div)
    float divs_pc = -1.0f;     // pseudo count the verdict below is for
    bool divs_ok = false;
    int64_t last_dp_chunks = 0, last_dp_stride = 0;
    float divs_m_pc = -1.0f;   // the same for the operand pairs of a medium tile (ntotal <= 255 * WG_MEDIUM_WMAX)
    bool divs_m_ok = false;
    bool divs_enabled = true;  // WGBSSEG_DIV_SHORT=0: always the 8-instruction core
    bool bs_general = false;   // WGBSSEG_BLOCK_SUMS_GENERAL=1: never the streaming block-sums kernel
    int64_t scan_piece_sites = 4096;    // WGBSSEG_SCAN_PIECE_SITES: sites per wave task of k_validate (multiple of 1024)
};

namespace {

bool device_local_cpus(int device, NearCpus* out);      // (below, with the streaming upload)
inline void pin_to(const NearCpus& n);

// Pageable host rows (typically memory-mapped .beta files) -> HBM: dst + r * dst_pitch <- rows[r][0 .. row_bytes).
// One thread drives ~33 GB/s of that (page faults + the copy into page-locked staging); a few threads, each with its
// own stream and two 4 MB staging pieces, fill the link (measured 46 GB/s with 4; more threads only contend).
int upload_rows(wgbsseg_ctx* c, uint8_t* dst, int64_t dst_pitch, const uint8_t* const* rows, int64_t n_rows, int64_t row_bytes,
                const char* what, char* err, size_t errlen)
{
    const double t0 = wall_s();
    int64_t piece = 4 << 20;
    { const char* e = getenv("WGBSSEG_UPLOAD_PIECE_KB"); if (e && atoi(e) >= 64) piece = (int64_t)atoi(e) << 10; }
    const int64_t pieces_per_row = (row_bytes + piece - 1) / piece, n_pieces = pieces_per_row * n_rows;
    int T = 4;
    { const char* e = getenv("WGBSSEG_UPLOAD_THREADS"); if (e && atoi(e) > 0) T = atoi(e); }
    T = (int)std::min<int64_t>(std::min<int64_t>(T, 64), std::max<int64_t>(1, std::min<int64_t>(n_pieces / 2, (row_bytes * n_rows) >> 24)));   // small inputs: the plain copy
    if (T <= 1) {
        for (int64_t r = 0; r < n_rows; r++)
            HIP_TRY(hipMemcpyAsync(dst + r * dst_pitch, rows[r], (size_t)row_bytes, hipMemcpyHostToDevice, c->sA));
        HIP_TRY(hipStreamSynchronize(c->sA));
    } else {
        if (c->up_stage.size() < (size_t)(2 * T)) c->up_stage.resize((size_t)(2 * T));
        std::atomic<int64_t> next(0);
        std::vector<hipError_t> terr((size_t)T, hipSuccess);
        auto worker = [&](int t) {
            pin_to(c->near_cpus);
            hipError_t e = hipSetDevice(c->device);
            // (its two staging pieces are allocated — first touched — by the worker itself, on the device's CPUs: they sit on that socket)
            for (int k = 0; k < 2 && e == hipSuccess; k++) if (!c->up_stage[(size_t)(2 * t + k)].ensure_exact((size_t)piece)) e = hipErrorOutOfMemory;
            hipStream_t st = nullptr;
            hipEvent_t ev[2] = {nullptr, nullptr};
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            for (int k = 0; k < 2 && e == hipSuccess; k++) e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
            bool busy[2] = {false, false};
            int k = 0;
            while (e == hipSuccess) {
                const int64_t it = next.fetch_add(1);
                if (it >= n_pieces) break;
                const int64_t r = it / pieces_per_row, o = (it % pieces_per_row) * piece, b = std::min<int64_t>(piece, row_bytes - o);
                void* stg = c->up_stage[(size_t)(2 * t + k)].p;
                if (busy[k]) { e = hipEventSynchronize(ev[k]); if (e != hipSuccess) break; }      // its previous copy has left the piece
                memcpy(stg, rows[r] + o, (size_t)b);
                e = hipMemcpyAsync(dst + r * dst_pitch + o, stg, (size_t)b, hipMemcpyHostToDevice, st);
                if (e == hipSuccess) e = hipEventRecord(ev[k], st);
                busy[k] = true;
                k ^= 1;
            }
            if (st) { const hipError_t e2 = hipStreamSynchronize(st); if (e == hipSuccess) e = e2; (void)hipStreamDestroy(st); }
            for (auto& x : ev) if (x) (void)hipEventDestroy(x);
            terr[(size_t)t] = e;
        };
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back(worker, t);
        for (auto& x : th) x.join();
        for (int t = 0; t < T; t++) HIP_TRY(terr[(size_t)t]);
    }
    if (profiling()) fprintf(stderr, "[wgbsseg] %s to the device: %.1f ms, %.1f GB/s (%d upload threads)\n", what, (wall_s() - t0) * 1e3,
                             (double)row_bytes * n_rows / (wall_s() - t0) * 1e-9, T);
    return WGBSSEG_OK;
}

}  // namespace

extern "C" {

int wgbsseg_version(void) { return WGBSSEG_VERSION; }

int wgbsseg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

}  // extern "C"
namespace {
int create_ctx(int device, bool scan_low_priority, wgbsseg_ctx** out, char* err, size_t errlen);
std::atomic<int> g_live_ctx[64];      // contexts alive per device (gated stages want theirs alone on its device)
}
extern "C" {

int wgbsseg_create(int device, wgbsseg_ctx** out, char* err, size_t errlen)
{
    return create_ctx(device, true, out, err, errlen);
}

}  // extern "C"
namespace {
// scan_low_priority: the scan stream below the scoring streams (right where k_validate competes with the kernels of its OWN batch only; the shares of a group that
// double up on a device get it at the scoring streams' priority: wgbsseg_group_create)
int create_ctx(int device, bool scan_low_priority, wgbsseg_ctx** out, char* err, size_t errlen)
{
    if (!out) { set_err(err, errlen, "out is NULL"); return WGBSSEG_E_ARG; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_err(err, errlen, "no HIP device available (%s); libwgbsseg has no CPU fallback", hipGetErrorString(e));
        return WGBSSEG_E_HIP;
    }
    if (device < 0 || device >= n) { set_err(err, errlen, "device %d out of range (0..%d)", device, n - 1); return WGBSSEG_E_ARG; }
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (!strstr(prop.gcnArchName, "gfx950")) {
        set_err(err, errlen, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
        return WGBSSEG_E_HIP;
    }
    HIP_TRY(set_kernel_attributes());        // per device, checked: k_dp / k_cost ask for more than 64 KB of dynamic LDS
    wgbsseg_ctx* c = new (std::nothrow) wgbsseg_ctx();
    if (!c) { set_err(err, errlen, "out of host memory"); return WGBSSEG_E_NOMEM; }
    c->device = device;
    (void)device_local_cpus(device, &c->near_cpus);
    HIP_TRY(hipStreamCreateWithFlags(&c->sA, hipStreamNonBlocking));
    {   // the recurrence stream outranks the scoring stream: its 483 latency-bound workgroups should never queue behind
        // the hundreds of thousands of throughput-bound scoring tiles
        int lo_p = 0, hi_p = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);
        HIP_TRY(hipStreamCreateWithPriority(&c->sB, hipStreamNonBlocking, hi_p));
        // ... and the scan stream ranks below it: k_validate starts with the batch, beside the windows pass, and must not take the wavefront
        // slots of the kernels that are on the way to the first scoring tile (measured, hg19 x 32: scoring begins 0.36 ms into the batch
        // instead of 0.46, profiles/r06_front_ab.txt)
        const char* sp = getenv("WGBSSEG_SCAN_PRIO");          // (A/B) 0: the scan stream at the scoring stream's priority
        if (scan_low_priority && !(sp && atoi(sp) == 0)) HIP_TRY(hipStreamCreateWithPriority(&c->sC, hipStreamNonBlocking, lo_p));
    }
    if (!c->sC) HIP_TRY(hipStreamCreateWithFlags(&c->sC, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->sA2, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->evD, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->evS, hipEventDisableTiming));
    for (auto& v : c->ev) HIP_TRY(hipEventCreate(&v));
    const char* b = getenv("WGBSSEG_COST_BUDGET_MB");
    c->cost_budget_bytes = (b && atoll(b) > 0 ? atoll(b) : 6144LL) << 20;
    const char* fs = getenv("WGBSSEG_FORCE_STAGES");
    c->force_stages = fs ? atoi(fs) : 0;
    { const char* e = getenv("WGBSSEG_STAGE_GATE"); if (e) c->stage_gate = std::max(0, atoi(e)); }
    { const char* e = getenv("WGBSSEG_STAGE_GATE_SHARED"); c->stage_gate_shared = e && atoi(e) != 0; }
    { const char* e = getenv("WGBSSEG_STAGE_GATE_WIDE_EVALS"); if (e) c->stage_gate_wide_evals = atof(e); }
    { const char* e = getenv("WGBSSEG_STAGE_MIN_EVALS"); if (e) c->stage_min_evals_per_step = atof(e); }
    { const char* e = getenv("WGBSSEG_LAST_STAGE_PCT"); if (e) c->last_stage_pct = std::min(800, std::max(5, atoi(e))); }      // (A/B, tests; default: chosen per job)
    { const char* e = getenv("WGBSSEG_DP_MODE"); c->force_dp_mode = e ? std::min(2, std::max(0, atoi(e))) : 0; }
    const char* fn = getenv("WGBSSEG_NS");
    c->force_ns = fn ? atoi(fn) : 0;
    const char* ft = getenv("WGBSSEG_TI");
    c->force_ti = ft ? atoi(ft) : 0;
    const char* bg = getenv("WGBSSEG_BLOCK_SUMS_GENERAL");
    if (bg) c->bs_general = atoi(bg) != 0;
    const char* dv = getenv("WGBSSEG_DIV_SHORT");
    if (dv) c->divs_enabled = atoi(dv) != 0;
    const char* sp = getenv("WGBSSEG_SCAN_PIECE_SITES");
    if (sp && atoi(sp) >= 1024) c->scan_piece_sites = (int64_t)(atoi(sp) & ~1023);
    g_live_ctx[device & 63].fetch_add(1, std::memory_order_relaxed);
    c->counted_live = true;
    *out = c;
    return WGBSSEG_OK;
}
}  // namespace
extern "C" {

void wgbsseg_destroy(wgbsseg_ctx* c)
{
    if (!c) return;
    if (c->counted_live) g_live_ctx[c->device & 63].fetch_sub(1, std::memory_order_relaxed);
    const double t0 = wall_s();
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    DevBuf* all[] = {&c->betas_own, &c->loci_own, &c->chunks, &c->wtile, &c->carry, &c->W16, &c->cum32, &c->back16, &c->chunk_pairs, &c->tile_tot, &c->tile_base, &c->tile_chunk,
                     &c->status, &c->plan_cbase, &c->plan_cum0, &c->plan_tbase, &c->plan_pairs, &c->plan_tiles, &c->plan_cnt, &c->tilesA, &c->tilesB, &c->tilesM, &c->umax16,
                     &c->cost[0], &c->cost[1], &c->cost[2], &c->stage_ctr, &c->dpstate, &c->tmp_borders, &c->nb, &c->boff, &c->out_borders[0], &c->out_borders[1], &c->edges,
                     &c->dbg_a, &c->dbg_b, &c->dbg_c, &c->lookup, &c->scan_pieces, &c->divcheck, &c->plan_sb, &c->bs_desc};
    for (auto* b : all) b->release();
    for (auto& pb : c->pinned) pb.release();
    for (auto& pb : c->up_stage) pb.release();
    c->h_status.release();
    c->h_out.release();
    c->h_job.release();
    c->h_pieces.release();
    c->h_sb.release();
    for (auto& v : c->ev) if (v) (void)hipEventDestroy(v);
    for (auto* vec : {&c->ev_cost0, &c->ev_cost1, &c->ev_dp0, &c->ev_dp1, &c->ev_fork, &c->ev_join}) for (auto v : *vec) (void)hipEventDestroy(v);
    if (c->sA) (void)hipStreamDestroy(c->sA);
    if (c->sB) (void)hipStreamDestroy(c->sB);
    if (c->sC) (void)hipStreamDestroy(c->sC);
    if (c->sA2) (void)hipStreamDestroy(c->sA2);
    if (c->evD) (void)hipEventDestroy(c->evD);
    if (c->evS) (void)hipEventDestroy(c->evS);
    delete c;
    if (profiling()) fprintf(stderr, "[wgbsseg] destroy: %.1f ms\n", (wall_s() - t0) * 1e3);
}

static int set_rows_host(wgbsseg_ctx* c, const uint8_t* const* samples, int64_t n_samples, int64_t n_sites, int elem, char* err, size_t errlen)
{
    if (!c || !samples || n_samples < 1 || n_sites < 1) { set_err(err, errlen, "bad arguments to set_betas_host"); return WGBSSEG_E_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    const int64_t pitch = round_up(2 * elem * n_sites, 256) + 256;      // slack: vector loads may run past the last site
    HIP_TRY(c->betas_own.ensure((size_t)pitch * (size_t)n_samples));
    for (int64_t s = 0; s < n_samples; s++)
        if (!samples[s]) { set_err(err, errlen, "samples[%lld] is NULL", (long long)s); return WGBSSEG_E_ARG; }
    const int rc = upload_rows(c, c->betas_own.as<uint8_t>(), pitch, samples, n_samples, 2 * elem * n_sites, "betas", err, errlen);
    if (rc != WGBSSEG_OK) return rc;
    c->betas = c->betas_own.as<uint8_t>();
    c->pitch = pitch; c->n_total = n_sites; c->n_samples = (int32_t)n_samples; c->elem = elem;
    c->last_valid = false;
    return WGBSSEG_OK;
}

int wgbsseg_set_betas_host(wgbsseg_ctx* c, const uint8_t* const* samples, int64_t n_samples, int64_t n_sites,
                           char* err, size_t errlen)
{
    return set_rows_host(c, samples, n_samples, n_sites, 1, err, errlen);
}

int wgbsseg_set_lbetas_host(wgbsseg_ctx* c, const uint16_t* const* samples, int64_t n_samples, int64_t n_sites,
                            char* err, size_t errlen)
{
    return set_rows_host(c, reinterpret_cast<const uint8_t* const*>(samples), n_samples, n_sites, 2, err, errlen);
}

int wgbsseg_set_betas_device(wgbsseg_ctx* c, const void* base, int64_t n_samples, int64_t pitch_bytes, int64_t n_sites,
                             char* err, size_t errlen)
{
    if (!c || !base || n_samples < 1 || n_sites < 1 || pitch_bytes < 2 * n_sites) { set_err(err, errlen, "bad arguments to set_betas_device"); return WGBSSEG_E_ARG; }
    if (((uintptr_t)base & 15) || (pitch_bytes & 15)) { set_err(err, errlen, "device betas base and pitch must be multiples of 16 bytes"); return WGBSSEG_E_ARG; }
    c->betas = reinterpret_cast<const uint8_t*>(base);
    c->pitch = pitch_bytes; c->n_total = n_sites; c->n_samples = (int32_t)n_samples; c->elem = 1;
    c->last_valid = false;
    return WGBSSEG_OK;
}

int wgbsseg_set_loci_host(wgbsseg_ctx* c, const uint32_t* loci, int64_t n_sites, char* err, size_t errlen)
{
    if (!c || !loci || n_sites < 1) { set_err(err, errlen, "bad arguments to set_loci_host"); return WGBSSEG_E_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->loci_own.ensure((size_t)n_sites * 4));
    const uint8_t* row = reinterpret_cast<const uint8_t*>(loci);
    const int rc = upload_rows(c, c->loci_own.as<uint8_t>(), 0, &row, 1, n_sites * 4, "loci", err, errlen);
    if (rc != WGBSSEG_OK) return rc;
    c->loci = c->loci_own.as<uint32_t>();
    c->n_loci = n_sites;
    c->last_valid = false;
    return WGBSSEG_OK;
}

int wgbsseg_set_loci_device(wgbsseg_ctx* c, const void* loci, int64_t n_sites, char* err, size_t errlen)
{
    if (!c || !loci || n_sites < 1) { set_err(err, errlen, "bad arguments to set_loci_device"); return WGBSSEG_E_ARG; }
    c->loci = reinterpret_cast<const uint32_t*>(loci);
    c->n_loci = n_sites;
    c->last_valid = false;
    return WGBSSEG_OK;
}

}  // extern "C"

namespace {

// Host copy of the chunk table + job-wide offsets; uploads it and builds the JobView.
struct Job {
    std::vector<ChunkDesc> h;
    std::vector<ScanPiece> pieces;      // what k_validate reads: the batch's sites not yet validated in this API call, cut into wave tasks
    int64_t val_sites = 0;
    std::vector<int64_t> wtile_off;     // exclusive prefix of 256-site tiles per chunk (+ total), then (as int32 pairs) the chunk of every 256th tile
    int64_t sites = 0, carry_entries = 0, units = 0, n_hint = 0;
    int64_t sites_padded = 0;           // the job-site arrays (W16, cum32, back16) hold every chunk from a multiple of 8 entries on: 16-byte vectors of them are aligned
    int32_t max_len = 0;
    JobStatus st0;          // source of an async H2D copy: must outlive the call's stream work
    JobView v = {};
};

int build_job(wgbsseg_ctx* c, const int64_t* start0, const int32_t* len, int64_t n_chunks, Job& job, bool need_loci,
              char* err, size_t errlen)
{
    if (!c) { set_err(err, errlen, "ctx is NULL"); return WGBSSEG_E_ARG; }
    if (!c->betas) { set_err(err, errlen, "betas not set"); return WGBSSEG_E_STATE; }
    if (c->elem != 1) { set_err(err, errlen, "the resident rows are uint16 (.lbeta): segment, scan and prefix sums read uint8 .beta data only (as the reference's segmentor, segmentor.cpp:166-176)"); return WGBSSEG_E_STATE; }
    if (need_loci && (!c->loci || c->n_loci != c->n_total)) { set_err(err, errlen, "loci not set or length differs from the betas (%lld vs %lld)", (long long)c->n_loci, (long long)c->n_total); return WGBSSEG_E_STATE; }
    if (!start0 || !len || n_chunks < 1 || n_chunks > 0x7fffffff) { set_err(err, errlen, "bad chunk list"); return WGBSSEG_E_ARG; }
    job.h.resize((size_t)n_chunks);
    job.wtile_off.resize((size_t)n_chunks + 1);
    int64_t so = 0, co = 0, wt = 0, uo = 0, total = 0;
    for (int64_t i = 0; i < n_chunks; i++) {
        if (len[i] < 1 || len[i] > (1 << 30) || start0[i] < 0 || start0[i] + len[i] > c->n_total) {
            set_err(err, errlen, "chunk %lld = [%lld, +%d) is empty or outside the %lld sites of the beta files",
                    (long long)i, (long long)start0[i], (int)len[i], (long long)c->n_total);
            return WGBSSEG_E_ARG;
        }
        ChunkDesc& d = job.h[(size_t)i];
        d.start0 = start0[i]; d.len = len[i]; d.site_off = so; d.carry_off = co; d.unit_off = uo; d.nG = (int32_t)(((start0[i] + len[i] - 1) >> WG_CARRY_SHIFT) - (start0[i] >> WG_CARRY_SHIFT) + 1);
        job.wtile_off[(size_t)i] = wt;
        wt += (len[i] + WG_WIN_TILE - 1) / WG_WIN_TILE;
        so += round_up((int64_t)len[i], 8);
        total += len[i];
        uo += (len[i] + 15) / 16;
        co += (int64_t)d.nG * c->n_samples;
        job.max_len = std::max(job.max_len, len[i]);
    }
    job.wtile_off[(size_t)n_chunks] = wt;
    {   // hint table for k_window: chunk of tile 256*h, for h = 0 .. tiles/256 + 1 (int32, packed behind the prefix)
        const int64_t nh = (wt >> 8) + 2;
        job.n_hint = nh;
        job.wtile_off.resize((size_t)n_chunks + 1 + (size_t)((nh + 1) / 2));
        int32_t* hint = reinterpret_cast<int32_t*>(job.wtile_off.data() + n_chunks + 1);
        int64_t cix = 0;
        for (int64_t h = 0; h < nh; h++) {
            const int64_t tile = h << 8;
            while (cix + 1 < n_chunks && job.wtile_off[(size_t)cix + 1] <= tile) cix++;
            hint[h] = (int32_t)cix;
        }
    }
    job.sites = total; job.sites_padded = so; job.carry_entries = co; job.units = uo;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->chunks.ensure(sizeof(ChunkDesc) * (size_t)n_chunks));
    HIP_TRY(c->carry.ensure(sizeof(uint2) * (size_t)co));
    HIP_TRY(c->status.ensure(sizeof(JobStatus)));
    HIP_TRY(c->wtile.ensure(8 * job.wtile_off.size()));
    memset(&job.st0, 0, sizeof(job.st0));
    job.st0.first_bad = ~0ULL;
    // the job's tables go up through ONE page-locked staging area (round 6): a copy out of pageable memory is staged by the runtime and
    // blocks the caller for ~15 us — four of them in front of every batch, the follow-up batches of junction patches included
    const size_t b_chunks = sizeof(ChunkDesc) * (size_t)n_chunks, b_wtile = 8 * job.wtile_off.size();
    const size_t o_wtile = (size_t)round_up((int64_t)b_chunks, 64), o_status = o_wtile + (size_t)round_up((int64_t)b_wtile, 64), o_end = o_status + 64;
    if (c->h_job.ensure(o_end)) {
        char* h = reinterpret_cast<char*>(c->h_job.p);
        memcpy(h, job.h.data(), b_chunks);
        memcpy(h + o_wtile, job.wtile_off.data(), b_wtile);
        memcpy(h + o_status, &job.st0, sizeof(job.st0));
        HIP_TRY(hipMemcpyAsync(c->chunks.p, h, b_chunks, hipMemcpyHostToDevice, c->sA));
        HIP_TRY(hipMemcpyAsync(c->wtile.p, h + o_wtile, b_wtile, hipMemcpyHostToDevice, c->sA));
        HIP_TRY(hipMemcpyAsync(c->status.p, h + o_status, sizeof(job.st0), hipMemcpyHostToDevice, c->sA));
    } else {
        HIP_TRY(hipMemcpyAsync(c->chunks.p, job.h.data(), b_chunks, hipMemcpyHostToDevice, c->sA));
        HIP_TRY(hipMemcpyAsync(c->wtile.p, job.wtile_off.data(), b_wtile, hipMemcpyHostToDevice, c->sA));
        HIP_TRY(hipMemcpyAsync(c->status.p, &job.st0, sizeof(job.st0), hipMemcpyHostToDevice, c->sA));
    }
    JobView& v = job.v;
    v.betas = c->betas; v.pitch = c->pitch; v.n_total = c->n_total; v.loci = c->loci;
    v.chunks = c->chunks.as<ChunkDesc>(); v.carry = c->carry.as<uint2>();
    v.n_samples = c->n_samples; v.n_chunks = (int32_t)n_chunks;
    return WGBSSEG_OK;
}

bool interval_covered(const std::vector<std::pair<int64_t, int64_t>>& v, int64_t lo, int64_t hi)
{
    // v sorted by first site: the last interval that begins at or before lo (false negatives only when intervals overlap)
    auto it = std::upper_bound(v.begin(), v.end(), lo, [](int64_t x, const std::pair<int64_t, int64_t>& iv) { return x < iv.first; });
    return it != v.begin() && (--it)->second >= hi;
}

// The sites of a batch that still need the `meth <= cov` pass in this API call, as wave tasks of k_validate.  Chunks of a
// genome arrive in order and tile their regions: they fuse into one run per region; junction patches (same batch or a
// follow-up batch) fall inside runs and drop out.  Arbitrary chunk lists work too (overlaps are validated twice).
int plan_validation(wgbsseg_ctx* c, Job& job, bool fresh_call, char* err, size_t errlen)
{
    if (fresh_call) c->validated.clear();
    std::vector<std::pair<int64_t, int64_t>> runs;
    bool sorted = true;
    for (const ChunkDesc& d : job.h) {
        const int64_t lo = d.start0, hi = d.start0 + d.len;
        if (!runs.empty() && lo == runs.back().second) { runs.back().second = hi; continue; }
        if (interval_covered(c->validated, lo, hi) || (sorted && interval_covered(runs, lo, hi))) continue;
        if (!runs.empty() && lo < runs.back().first) sorted = false;
        runs.emplace_back(lo, hi);
    }
    const int64_t P = c->scan_piece_sites;
    job.pieces.clear(); job.val_sites = 0;
    for (const auto& r : runs) {
        job.val_sites += r.second - r.first;
        for (int64_t a = r.first; a < r.second; ) {
            const int64_t b = std::min(r.second, (a / P + 1) * P);          // cut on absolute multiples of P: aligned streams
            job.pieces.push_back(ScanPiece{a, (int32_t)(b - a), 0});
            a = b;
        }
    }
    if (!runs.empty()) {
        std::vector<std::pair<int64_t, int64_t>>& V = c->validated;
        V.insert(V.end(), runs.begin(), runs.end());
        std::sort(V.begin(), V.end());
        size_t w = 0;
        for (size_t i = 1; i < V.size(); i++) {
            if (V[i].first <= V[w].second) V[w].second = std::max(V[w].second, V[i].second);
            else V[++w] = V[i];
        }
        V.resize(w + 1);
        HIP_TRY(c->scan_pieces.ensure(sizeof(ScanPiece) * job.pieces.size()));
        const size_t b_pieces = sizeof(ScanPiece) * job.pieces.size();
        const void* src = job.pieces.data();
        if (c->h_pieces.ensure(b_pieces)) { memcpy(c->h_pieces.p, src, b_pieces); src = c->h_pieces.p; }      // (page-locked: the copy does not block, see build_job)
        HIP_TRY(hipMemcpyAsync(c->scan_pieces.p, src, b_pieces, hipMemcpyHostToDevice, c->sA));
    }
    return WGBSSEG_OK;
}

// The scan pass.  k_validate (per piece, read-only): the `meth <= cov` check of read_beta_file (segmentor.cpp:179-188) over the sites of the
// batch this API call has not checked yet; it needs nothing but the beta bytes, so a batch launches it first of all, beside the windows pass
// (round 6).  k_scan (per chunk, with carries + the same check): only for a job with wide scoring tiles — the one consumer of the carries —
// or a caller that wants them.
int launch_validate(wgbsseg_ctx* c, const Job& job, hipStream_t s, char* err, size_t errlen)
{
    if (job.pieces.empty()) return WGBSSEG_OK;
    const int64_t tasks = (int64_t)job.pieces.size() * job.v.n_samples;
    const int64_t vb = (tasks + (WG_BLOCK / 64) - 1) / (WG_BLOCK / 64);
    if (vb > 0x7fffffff) { set_err(err, errlen, "too many (piece, sample) rows"); return WGBSSEG_E_ARG; }
    hipLaunchKernelGGL(k_validate, dim3((unsigned)vb), dim3(WG_BLOCK), 0, s, job.v, c->status.as<JobStatus>(),
                       c->scan_pieces.as<ScanPiece>(), (int64_t)job.pieces.size());
    HIP_TRY(hipGetLastError());
    return WGBSSEG_OK;
}

int launch_scan(wgbsseg_ctx* c, const Job& job, hipStream_t s, char* err, size_t errlen)
{
    const int64_t rows = (int64_t)job.v.n_chunks * job.v.n_samples;      // wave tasks
    const int64_t blocks = (rows + (WG_BLOCK / 64) - 1) / (WG_BLOCK / 64);
    if (blocks > 0x7fffffff) { set_err(err, errlen, "too many (chunk, sample) rows"); return WGBSSEG_E_ARG; }
    hipLaunchKernelGGL(k_scan, dim3((unsigned)blocks), dim3(WG_BLOCK), 0, s, job.v, c->status.as<JobStatus>());
    HIP_TRY(hipGetLastError());
    return WGBSSEG_OK;
}

int report_bad_site(const wgbsseg_ctx* c, const JobStatus& st, char* err, size_t errlen)
{
    const long long s = (long long)(st.first_bad >> 40), site = (long long)(st.first_bad & ((1ULL << 40) - 1));
    uint8_t mc[2] = {0, 0};
    (void)hipMemcpy(mc, c->betas + s * c->pitch + 2 * site, 2, hipMemcpyDeviceToHost);
    set_err(err, errlen, "invalid data: sample %lld (0-based, argument order), site %lld (0-based): meth %d > cov %d",
            s, site + (long long)c->site_base, (int)mc[0], (int)mc[1]);
    return WGBSSEG_E_METH_GT_COV;
}

template <int TI, int FAST, int SPLIT>
hipError_t launch_cost(const JobView& v, const StageView& sv, const CostArgs& a, const TileDesc* td, int64_t tiles, double* cost, size_t lds, hipStream_t s)
{
    const int64_t padded = round_up(tiles, 8 * (int64_t)a.xcd_group);      // whole groups for every XCD; workgroups behind the last tile leave at once
    if (TI > 64 || a.NS >= v.n_samples)      // every sample in LDS at once: the form without partial sums across sample groups
        hipLaunchKernelGGL((k_cost<TI, FAST, SPLIT, true>), dim3((unsigned)padded), dim3(WG_BLOCK), lds, s, v, sv, a, td, tiles, cost, padded);
    else
        hipLaunchKernelGGL((k_cost<(TI > 64 ? 64 : TI), FAST, SPLIT, false>), dim3((unsigned)padded), dim3(WG_BLOCK), lds, s, v, sv, a, td, tiles, cost, padded);
    return hipGetLastError();
}

// narrow tiles: TI start sites (64 / 32 / 16); wide tiles: WG_WIDE_TS start sites x WG_WIDE_TK end sites
template <int FAST>
hipError_t launch_cost_ti(int TI, bool wide, const JobView& v, const StageView& sv, const CostArgs& a, const TileDesc* td, int64_t tiles, double* cost, size_t lds, hipStream_t s)
{
    if (TI < 0) return launch_cost<WG_MEDIUM_TS, FAST, 2>(v, sv, a, td, tiles, cost, lds, s);                     // medium tiles (TI = -1)
    if (wide) return launch_cost<WG_WIDE_TS, FAST == 3 ? 2 : FAST, 1>(v, sv, a, td, tiles, cost, lds, s);      // (the short division is a form of the tiles with few operand pairs)
    if (TI == 128) return launch_cost<128, FAST, 0>(v, sv, a, td, tiles, cost, lds, s);
    if (TI == 64) return launch_cost<64, FAST, 0>(v, sv, a, td, tiles, cost, lds, s);
    if (TI == 32) return launch_cost<32, FAST, 0>(v, sv, a, td, tiles, cost, lds, s);
    return launch_cost<16, FAST, 0>(v, sv, a, td, tiles, cost, lds, s);
}

// Kernels that ask for more than 64 KB of dynamic LDS need the attribute on EVERY device they run on: set once per
// context, right after hipSetDevice (wgbsseg_create), and checked.
template <int FAST>
hipError_t set_cost_attrs()
{
    const void* fns[] = {reinterpret_cast<const void*>(&k_cost<128, FAST, 0, true>),
                         reinterpret_cast<const void*>(&k_cost<64, FAST, 0, true>), reinterpret_cast<const void*>(&k_cost<32, FAST, 0, true>),
                         reinterpret_cast<const void*>(&k_cost<16, FAST, 0, true>), reinterpret_cast<const void*>(&k_cost<WG_WIDE_TS, FAST == 3 ? 2 : FAST, 1, true>),
                         reinterpret_cast<const void*>(&k_cost<WG_MEDIUM_TS, FAST, 2, true>),
                         reinterpret_cast<const void*>(&k_cost<64, FAST, 0, false>), reinterpret_cast<const void*>(&k_cost<32, FAST, 0, false>),
                         reinterpret_cast<const void*>(&k_cost<16, FAST, 0, false>), reinterpret_cast<const void*>(&k_cost<WG_WIDE_TS, FAST == 3 ? 2 : FAST, 1, false>),
                         reinterpret_cast<const void*>(&k_cost<WG_MEDIUM_TS, FAST, 2, false>)};
    for (const void* f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
hipError_t set_kernel_attributes()
{
    hipError_t e = set_cost_attrs<0>();
    if (e == hipSuccess) e = set_cost_attrs<1>();
    if (e == hipSuccess) e = set_cost_attrs<2>();
    if (e == hipSuccess) e = set_cost_attrs<3>();
    const void* dps[] = {reinterpret_cast<const void*>(&k_dp<7, 64>), reinterpret_cast<const void*>(&k_dp<7, 64, true>),
                         reinterpret_cast<const void*>(&k_dp<7, 32>),
                         reinterpret_cast<const void*>(&k_dp<15, 32>)};
    for (const void* f : dps)
        if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e;
}

void grow_events(std::vector<hipEvent_t>& v, size_t n)
{
    while (v.size() < n) { hipEvent_t e; (void)hipEventCreate(&e); v.push_back(e); }
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {
typedef std::function<int32_t*(int64_t)> BorderAlloc;       // total border count -> destination (NULL: too small)

// Early delivery of a batch's result (the first batch of a region-level call; the destination must be page-locked, i.e. device-visible):
// the call returns when the CSR offsets, the lists of the items from `n_lead` on (junction patches) and the EDGES of the leading items'
// lists (the chunks: first / last WG_EDGE borders each) are on the host; the leading items' lists are still being written by k_copy_out
// and are complete after wait_pending_output().
struct EarlyOut {
    int64_t n_lead = 0;              // in: leading items whose lists may come late (0: no early delivery)
    bool dest_device_visible = false;   // in (set by the caller's BorderAlloc): the destination it handed out is page-locked memory the device can write
    const int32_t* edges = nullptr;  // out: [n_lead][front | back][WG_EDGE], the back right-aligned (valid until the context's next batch)
    bool pending = false;            // out: the leading lists are still on their way
};
int wait_pending_output(wgbsseg_ctx* c, char* err, size_t errlen)
{
    if (!c->out_pending) return WGBSSEG_OK;
    c->out_pending = false;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->evD));
    return WGBSSEG_OK;
}

int segment_chunks_impl(wgbsseg_ctx* c, const int64_t* chunk_start0, const int32_t* chunk_len, int64_t n_chunks,
                        const wgbsseg_params* P, const BorderAlloc& alloc, int64_t* borders_off, char* err, size_t errlen, bool allow_plain = true, EarlyOut* early = nullptr);

// One chunk whose loci are not ascending, the reference's loops as written (csrc/plain_dp.h): ascending borders incl. 0 and len.
int plain_segment_chunk(wgbsseg_ctx* c, int64_t start0, int32_t n, const wgbsseg_params* P, std::vector<int32_t>& borders, char* err, size_t errlen)
{
    const int32_t W = (int32_t)std::min<int64_t>(P->max_cpg, n);
    // Ring of R rows of W doubles: the rows of the band in flight + the W - 1 rows before it that its steps still read.  ~1 GB of
    // ring while that holds at least W + 255 rows (W <= ~11,400); deeper windows take the floor W + 255 rows, i.e. up to W * n * 8
    // bytes (28.8 GB for max_cpg >= chunk = 60,000 — within an MI355X's 288 GB; E_NOMEM below when the device cannot give it).
    // WGBSSEG_PLAIN_RING_ROWS caps the rows (tests: the banded ring with R < n on chunks small enough for the suite).
    int64_t budget_rows = std::max<int64_t>((int64_t)W + 255, (1LL << 30) / ((int64_t)W * 8));
    if (const char* e = getenv("WGBSSEG_PLAIN_RING_ROWS")) {
        const long long v = atoll(e);
        if (v > 0) budget_rows = std::max<int64_t>((int64_t)W, (int64_t)v);     // R >= W keeps the band >= 1 row
    }
    const int32_t R = (int32_t)std::min<int64_t>(n, budget_rows);
    const int32_t band = R >= n ? n : R - (W - 1);
    DevBuf buf, M, T, bad;
    struct Free { DevBuf& a; DevBuf& b; DevBuf& c; DevBuf& d; ~Free() { a.release(); b.release(); c.release(); d.release(); } } fr{buf, M, T, bad};
    if (buf.ensure((size_t)W * (size_t)R * 8) != hipSuccess || M.ensure((size_t)(n + 1) * 8) != hipSuccess || T.ensure((size_t)(n + 1) * 4) != hipSuccess ||
        bad.ensure(8) != hipSuccess) {
        (void)hipGetLastError();
        set_err(err, errlen, "out of device memory for the plain recurrence of a chunk with non-ascending loci (%d sites, max_cpg %d)", (int)n, (int)W);
        return WGBSSEG_E_NOMEM;
    }
    JobStatus st = {};
    st.first_bad = ~0ULL;
    HIP_TRY(hipMemcpyAsync(bad.p, &st.first_bad, 8, hipMemcpyHostToDevice, c->sA));
    PlainArgs A = {c->betas, c->pitch, c->n_total, c->n_samples, c->loci, start0, n, W, P->max_bp, P->pseudo_count, P->pseudo_count + P->pseudo_count,
                   R, buf.as<double>(), M.as<double>(), T.as<int32_t>(), bad.as<unsigned long long>()};
    for (int32_t b0 = 0; b0 < n; b0 += band) {
        const int32_t b1 = (int32_t)std::min<int64_t>(n, (int64_t)b0 + band);
        hipLaunchKernelGGL(k_plain_rows, dim3((unsigned)((b1 - b0 + WG_BLOCK - 1) / WG_BLOCK)), dim3(WG_BLOCK), 0, c->sA, A, b0, b1);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_plain_dp, dim3(1), dim3(WG_BLOCK), 0, c->sA, A, b0, b1);
        HIP_TRY(hipGetLastError());
    }
    std::vector<int32_t> hT((size_t)n + 1);
    HIP_TRY(hipMemcpyAsync(hT.data(), T.p, (size_t)(n + 1) * 4, hipMemcpyDeviceToHost, c->sA));
    HIP_TRY(hipMemcpyAsync(&st.first_bad, bad.p, 8, hipMemcpyDeviceToHost, c->sA));
    HIP_TRY(hipStreamSynchronize(c->sA));
    if (st.first_bad != ~0ULL) return report_bad_site(c, st, err, errlen);
    borders.clear();                                          // segmentor.cpp:50-58: i = n; push; while i > 0: i = max(0, T[i]); push — printed in reverse
    int32_t i = n;
    borders.push_back(i);
    while (i > 0) {
        const int32_t t = hT[(size_t)i];
        const int32_t nx = t > 0 ? t : 0;
        if (nx >= i) { set_err(err, errlen, "internal: back-pointer %d at step %d of a plain recurrence", (int)t, (int)i); return WGBSSEG_E_STATE; }
        i = nx;
        borders.push_back(i);
    }
    std::reverse(borders.begin(), borders.end());
    return WGBSSEG_OK;
}

// A batch in which k_window found chunks with non-ascending loci: those chunks take the plain path, the others the batch path (again,
// without them), and the border lists are put back in the caller's order.
int segment_chunks_with_disorder(wgbsseg_ctx* c, const int64_t* chunk_start0, const int32_t* chunk_len, int64_t n_chunks, const wgbsseg_params* P,
                                 const wgbsseg_params* Peff, const Job& job, const BorderAlloc& alloc, int64_t* borders_off, char* err, size_t errlen)
{
    DevBuf flags;
    struct Free { DevBuf& a; ~Free() { a.release(); } } fr{flags};
    HIP_TRY(flags.ensure((size_t)n_chunks * 4));
    HIP_TRY(hipMemsetAsync(flags.p, 0, (size_t)n_chunks * 4, c->sA));
    hipLaunchKernelGGL(k_find_disorder, dim3((unsigned)n_chunks), dim3(WG_BLOCK), 0, c->sA, job.v, flags.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    std::vector<uint32_t> hf((size_t)n_chunks);
    HIP_TRY(hipMemcpyAsync(hf.data(), flags.p, (size_t)n_chunks * 4, hipMemcpyDeviceToHost, c->sA));
    HIP_TRY(hipStreamSynchronize(c->sA));
    std::vector<int64_t> os, ooff;
    std::vector<int32_t> ol, ob;
    std::vector<int64_t> oidx;
    for (int64_t i = 0; i < n_chunks; i++) if (!hf[(size_t)i]) { os.push_back(chunk_start0[i]); ol.push_back(chunk_len[i]); oidx.push_back(i); }
    if (!os.empty()) {
        ooff.resize(os.size() + 1);
        const int rc = segment_chunks_impl(c, os.data(), ol.data(), (int64_t)os.size(), P,
                                           [&](int64_t total) { ob.resize((size_t)std::max<int64_t>(1, total)); return ob.data(); }, ooff.data(), err, errlen, false);
        if (rc != WGBSSEG_OK) return rc;
    }
    std::vector<std::vector<int32_t>> plain((size_t)n_chunks);
    for (int64_t i = 0; i < n_chunks; i++)
        if (hf[(size_t)i]) {
            wgbsseg_params Pc = *Peff;                         // (max_cpg is already cut to the longest chunk of the call; the kernels cut it to this chunk)
            const int rc = plain_segment_chunk(c, chunk_start0[i], chunk_len[i], &Pc, plain[(size_t)i], err, errlen);
            if (rc != WGBSSEG_OK) return rc;
        }
    int64_t total = 0;
    size_t oi = 0;
    for (int64_t i = 0; i < n_chunks; i++) {
        borders_off[i] = total;
        if (hf[(size_t)i]) total += (int64_t)plain[(size_t)i].size();
        else { total += ooff[oi + 1] - ooff[oi]; oi++; }
    }
    borders_off[n_chunks] = total;
    int32_t* out = alloc(total);
    if (!out) { set_err(err, errlen, "borders_out too small: need %lld ints", (long long)total); return WGBSSEG_E_CAPACITY; }
    oi = 0;
    for (int64_t i = 0; i < n_chunks; i++) {
        if (hf[(size_t)i]) memcpy(out + borders_off[i], plain[(size_t)i].data(), plain[(size_t)i].size() * 4);
        else { memcpy(out + borders_off[i], ob.data() + ooff[oi], (size_t)(ooff[oi + 1] - ooff[oi]) * 4); oi++; }
    }
    c->last_valid = false;
    return WGBSSEG_OK;
}

int segment_chunks_impl(wgbsseg_ctx* c, const int64_t* chunk_start0, const int32_t* chunk_len, int64_t n_chunks,
                        const wgbsseg_params* P, const BorderAlloc& alloc, int64_t* borders_off, char* err, size_t errlen, bool allow_plain, EarlyOut* early)
{
    if (!P || !borders_off) { set_err(err, errlen, "NULL params/borders pointer"); return WGBSSEG_E_ARG; }
    if (P->max_bp == 0) { set_err(err, errlen, "max_bp must be >= 1 (the reference reads uninitialised loci when it is 0: segmentor.cpp:38,114)"); return WGBSSEG_E_ARG; }
    if (P->max_cpg < 1) { set_err(err, errlen, "max_cpg must be >= 1"); return WGBSSEG_E_ARG; }
    // The reference sizes its ring to any max_cpg (segmentor.cpp:92-95).  What bounds it here: a block's counts must stay exact in
    // the float sums of segmentor.cpp:122-123 (255 * max_cpg < 2^24: beyond that the REFERENCE's own sums round),
    // and a window is stored in 16 bits.  A window never exceeds its chunk (segmentor.cpp:110: j < n - i), so max_cpg counts
    // only up to the longest chunk of the call.
    int32_t longest = 1;
    if (chunk_len) for (int64_t i = 0; i < n_chunks; i++) longest = std::max(longest, chunk_len[i]);
    wgbsseg_params Peff = *P;
    Peff.max_cpg = std::min<uint32_t>(P->max_cpg, (uint32_t)longest);
    if ((uint64_t)Peff.max_cpg * 255u >= (1u << 24) || Peff.max_cpg > WGBSSEG_MAX_CPG) {
        set_err(err, errlen, "max_cpg %u with chunks of up to %d sites unsupported: blocks of more than %d sites (255 * sites >= 2^24) do not keep their counts exact "
                "in the float sums of the reference itself (segmentor.cpp:122-123), and windows are stored in 16 bits", P->max_cpg, (int)longest, WGBSSEG_MAX_CPG);
        return WGBSSEG_E_ARG;
    }
    const wgbsseg_params* const P0 = P;
    P = &Peff;
    static const bool host_marks = getenv("WGBSSEG_PROFILE") && atoi(getenv("WGBSSEG_PROFILE")) >= 2;
    double hm[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // (WGBSSEG_PROFILE=2) host clock: entry, tables up, windows queued, statistics here, scoring queued, everything queued, results here, return
    if (host_marks) hm[0] = wall_s();
    if (!(P->pseudo_count >= 0.0f)) { set_err(err, errlen, "pseudo_count must be >= 0"); return WGBSSEG_E_ARG; }
    // A batch that failed half way may have left work behind on the scoring / scan streams that reads the staging areas: such a context is drained before it is
    // used again.  (A clean one is NOT synchronised stream by stream here: the follow-up batch of an early delivery starts while k_copy_out still writes the first
    // batch's lists, and a stream synchronisation — a marker of its own in a hardware queue the streams share — waited ~0.13 ms behind that kernel.)
    if (c && c->batch_open) {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipDeviceSynchronize());
        c->out_pending = false;
    }
    if (c) c->batch_open = true;
    Job job;
    int rc = build_job(c, chunk_start0, chunk_len, n_chunks, job, true, err, errlen);
    if (rc != WGBSSEG_OK) return rc;
    rc = plan_validation(c, job, !c->accumulate, err, errlen);
    if (rc != WGBSSEG_OK) return rc;
    c->last_valid = false;
    const int nC = (int)n_chunks;
    const int64_t J = job.sites;
    JobView& v = job.v;

    const int64_t Jp = job.sites_padded;                         // every chunk's slice of the job-site arrays begins at a multiple of 8 entries
    const int64_t nT = job.wtile_off[(size_t)nC];                // 1024-site tiles of the windows pass
    if (nT > 0x7fffffff) { set_err(err, errlen, "too many sites in one call"); return WGBSSEG_E_ARG; }
    HIP_TRY(c->W16.ensure((size_t)Jp * 2 + 16));
    HIP_TRY(c->cum32.ensure((size_t)Jp * 4 + 16));
    HIP_TRY(c->back16.ensure((size_t)Jp * 2 + 16));
    HIP_TRY(c->chunk_pairs.ensure((size_t)nC * 8));
    HIP_TRY(c->umax16.ensure((size_t)job.units * 2));
    HIP_TRY(c->tile_tot.ensure((size_t)nT * 4));
    HIP_TRY(c->tile_base.ensure((size_t)nT * 4));
    HIP_TRY(c->tile_chunk.ensure((size_t)nT * 16));          // (one int4 record per tile)
    v.W16 = c->W16.as<uint16_t>(); v.cum32 = c->cum32.as<uint32_t>(); v.back16 = c->back16.as<uint16_t>();
    v.chunk_pairs = c->chunk_pairs.as<int64_t>(); v.umax16 = c->umax16.as<uint16_t>();
    if (!c->h_status.ensure(4 * sizeof(JobStatus))) { set_err(err, errlen, "out of page-locked host memory"); return WGBSSEG_E_NOMEM; }

    // ---- the scan pass starts with the batch, on its own stream (round 6) ---------------------------------------
    // k_validate needs the beta bytes and the list of pieces, nothing from the windows: it runs beside k_window (HBM-bound beside
    // instruction-bound) instead of behind it, and the scoring kernel of a job without wide tiles — which needs nothing from the scan
    // but its verdict, read at the end of the batch — no longer queues behind either.  (Rounds 1-5: windows 0.28 ms, then the scan
    // 0.32 ms, then the tile plan: scoring began 0.71 ms into the hg19 x 32 batch.)
    if (host_marks) hm[1] = wall_s();
    HIP_TRY(hipEventRecord(c->ev[0], c->sA));
    HIP_TRY(hipEventRecord(c->ev[12], c->sA));                   // chunk table, pieces and the cleared status block are on the device
    hipStream_t const sS = c->sC;
    // where k_validate starts (WGBSSEG_SCAN_AFTER, A/B): 0 with the batch, beside the windows pass; 1 behind the windows pass, beside the tile plan; 2 behind the
    // tile plan, beside the first scoring tiles only (the latency-bound kernels of the front — row offsets, stage plan, tile descriptors — keep the memory system to themselves)
    static const int scan_after = getenv("WGBSSEG_SCAN_AFTER") ? atoi(getenv("WGBSSEG_SCAN_AFTER")) : 0;
    bool validate_queued = false;
    HIP_TRY(hipStreamWaitEvent(sS, c->ev[12], 0));
    if (!scan_after) {
        validate_queued = true;
        HIP_TRY(hipEventRecord(c->ev[8], sS));
        rc = launch_validate(c, job, sS, err, errlen);
        if (rc != WGBSSEG_OK) return rc;
        HIP_TRY(hipEventRecord(c->ev[9], sS));
    }

    // ---- window extents ------------------------------------------------------------------------------------------
    // a pseudo count this context has not scored with yet: may the narrow tiles use the short division core?  Every operand
    // pair they can form is tried on the device (0.1 ms, once); the verdict arrives with the window statistics.
    const bool check_div = c->divs_enabled && wg_term_mode(P->pseudo_count) == 2 && c->divs_pc != P->pseudo_count;
    if (check_div) {
        HIP_TRY(c->divcheck.ensure(4));
        HIP_TRY(hipMemsetAsync(c->divcheck.p, 0, 4, c->sA));
        const int max_total = 255 * WG_NARROW_WMAX;
        hipLaunchKernelGGL(k_check_div, dim3((unsigned)max_total + 1), dim3(WG_BLOCK), 0, c->sA, P->pseudo_count, P->pseudo_count + P->pseudo_count,
                           max_total, c->divcheck.as<unsigned int>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(reinterpret_cast<JobStatus*>(c->h_status.p) + 1, c->divcheck.p, 4, hipMemcpyDeviceToHost, c->sA));
    }
    // the search's first step when a wavefront has windows of 64 sites and more: 2^floor(log2(max_cpg - 1))
    int top_step = 1;
    while (2 * (int64_t)top_step <= (int64_t)P->max_cpg - 1) top_step *= 2;
    // loci of a 1024-site tile, of the site before it and of everything a PROBE of its searches can touch, in LDS (<= 48 KB; deeper windows search in L2)
    const int64_t win_want = 1 + (int64_t)WG_WIN_TILE + 2 * (int64_t)std::max(top_step, 32) + 8;
    const int win_lds = win_want <= 12288 ? (int)win_want : 0;
    // medium tiles (round 3): units of a non-narrow group whose windows stay <= WG_MEDIUM_WMAX = 252 sites (block counts < 2^16): CpG islands
    const int wm_env = getenv("WGBSSEG_MEDIUM_WMAX") ? std::max(0, std::min(WG_MEDIUM_WMAX, atoi(getenv("WGBSSEG_MEDIUM_WMAX")))) : WG_MEDIUM_WMAX;   // 0: no medium class (A/B, tests; read per call)
    const int WMED = wm_env > WG_NARROW_WMAX ? wm_env : 0;
    const int64_t* const d_wtile = c->wtile.as<int64_t>();
    hipLaunchKernelGGL(k_window, dim3((unsigned)nT), dim3(WG_BLOCK), (size_t)win_lds * 4, c->sA, v, c->status.as<JobStatus>(),
                       d_wtile, reinterpret_cast<const int32_t*>(d_wtile + nC + 1), P->max_cpg, P->max_bp, win_lds, top_step, std::max(WMED, WG_NARROW_WMAX),
                       c->tile_tot.as<uint32_t>(), c->tile_chunk.as<int4>());
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_window_scan, dim3((unsigned)nC), dim3(WG_BLOCK), 0, c->sA, v, c->status.as<JobStatus>(), d_wtile, c->tile_tot.as<uint32_t>(), c->tile_base.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev[1], c->sA));
    if (scan_after == 1) {
        validate_queued = true;
        HIP_TRY(hipStreamWaitEvent(sS, c->ev[1], 0));
        HIP_TRY(hipEventRecord(c->ev[8], sS));
        rc = launch_validate(c, job, sS, err, errlen);
        if (rc != WGBSSEG_OK) return rc;
        HIP_TRY(hipEventRecord(c->ev[9], sS));
    }
    JobStatus* hst = reinterpret_cast<JobStatus*>(c->h_status.p);
    HIP_TRY(hipMemcpyAsync(&hst[0], c->status.p, sizeof(JobStatus), hipMemcpyDeviceToHost, c->sA));
    HIP_TRY(hipEventRecord(c->ev[7], c->sA));
    // (the row offsets are not in the statistics: this pass runs while the host reads them and plans the tiles)
    hipLaunchKernelGGL(k_window_cum, dim3((unsigned)((nT + WG_CUM_TILES - 1) / WG_CUM_TILES)), dim3(WG_BLOCK), 0, c->sA, (const uint16_t*)v.W16, v.cum32, (const int4*)c->tile_chunk.as<int4>(),
                       (const uint32_t*)c->tile_base.as<uint32_t>(), nT);
    HIP_TRY(hipGetLastError());
    if (host_marks) hm[2] = wall_s();
    HIP_TRY(hipEventSynchronize(c->ev[7]));                    // window statistics are here; the scan pass is still running
    if (host_marks) hm[3] = wall_s();
    const JobStatus st = hst[0];
    if (check_div) {
        c->divs_ok = *reinterpret_cast<const unsigned int*>(&hst[1]) == 0u;
        c->divs_pc = P->pseudo_count;
        if (profiling()) fprintf(stderr, "[wgbsseg] short division core for pseudo count %g: %s\n", (double)P->pseudo_count, c->divs_ok ? "verified on every operand pair of a narrow tile" : "NOT exact, the full core stays");
    }
    // wide tiles read the carries of k_scan: behind k_validate on the scan stream, as soon as the windows are known
    if (st.wide_units && !st.loci_disorder && !st.overflow) {
        HIP_TRY(hipStreamWaitEvent(sS, c->ev[1], 0));
        HIP_TRY(hipEventRecord(c->ev[10], sS));
        rc = launch_scan(c, job, sS, err, errlen);
        if (rc != WGBSSEG_OK) return rc;
        HIP_TRY(hipEventRecord(c->ev[11], sS));
    }
    // (k_scan checks every byte k_validate would: a job with wide tiles that has not queued k_validate yet does without it)
    const bool validate_late = !validate_queued && !(st.wide_units && !st.loci_disorder && !st.overflow);
    auto close_scan_stream = [&]() -> hipError_t {             // the scan's verdict, read at the end of the batch
        hipError_t e = hipEventRecord(c->ev[2], sS);
        if (e == hipSuccess) e = hipMemcpyAsync(&hst[2], c->status.p, sizeof(JobStatus), hipMemcpyDeviceToHost, sS);
        if (e == hipSuccess) e = hipEventRecord(c->evS, sS);
        return e;
    };
    auto queue_validate_now = [&]() -> int {
        HIP_TRY(hipEventRecord(c->ev[8], sS));
        const int r = launch_validate(c, job, sS, err, errlen);
        if (r != WGBSSEG_OK) return r;
        HIP_TRY(hipEventRecord(c->ev[9], sS));
        return WGBSSEG_OK;
    };
    if (!validate_queued && !validate_late) { HIP_TRY(hipEventRecord(c->ev[8], sS)); HIP_TRY(hipEventRecord(c->ev[9], sS)); }
    if (!validate_late) HIP_TRY(close_scan_stream());
    if (st.loci_disorder || st.overflow) {
        JobStatus st2;                                        // the scan's verdict takes precedence, as it always has
        if (validate_late) { rc = queue_validate_now(); if (rc != WGBSSEG_OK) return rc; HIP_TRY(close_scan_stream()); }
        HIP_TRY(hipStreamSynchronize(sS));
        HIP_TRY(hipStreamSynchronize(c->sA));
        HIP_TRY(hipMemcpyAsync(&st2, c->status.p, sizeof(st2), hipMemcpyDeviceToHost, c->sA));
        HIP_TRY(hipStreamSynchronize(c->sA));
        if (st2.first_bad != ~0ULL) return report_bad_site(c, st2, err, errlen);
        if (st.loci_disorder) {
            // loci not ascending inside a chunk: the reference bars such an extension and leaves the site out of its running sums
            // (segmentor.cpp:114-117).  The batch path rests on ascending loci (windows, prefix sums); the chunks concerned take
            // the plain path (csrc/plain_dp.h: the reference's loops as written), the rest of the batch runs again without them.
            if (allow_plain) return segment_chunks_with_disorder(c, chunk_start0, chunk_len, n_chunks, P0, &Peff, job, alloc, borders_off, err, errlen);
            set_err(err, errlen, "internal: loci not ascending inside chunk %u (0-based sites [%lld, +%d)) of a batch that was filtered for such chunks",
                    st.loci_disorder - 1, (long long)(job.h[st.loci_disorder - 1].start0 + c->site_base), (int)job.h[st.loci_disorder - 1].len);
            return WGBSSEG_E_LOCI_ORDER;
        }
        set_err(err, errlen, "a chunk scores more than 2^32 blocks; use a smaller chunk_size"); return WGBSSEG_E_ARG;
    }
    const int Wmax = (int)st.max_window;
    const int64_t total_pairs = (int64_t)st.total_pairs;

    // ---- tiling of the scoring kernel ----------------------------------------------------------------------
    // Narrow tiles (class A): TI aligned start sites whose windows are all <= WA = min(widest window, WG_NARROW_WMAX);
    // LDS per workgroup: fast log tables + NS sample rows of TI+61 tile-local prefixes (one packed dword each) + small
    // per-tile arrays.  The kernel is a long dependent chain per evaluation, so resident wavefronts matter: pick the
    // shape that maximises (workgroups per CU) x (lane occupancy of the block rounds).
    // Wide tiles (class B): 16 start sites x 128 end sites, for every 16-site unit that shares an aligned TI-group with
    // a window > WA (CpG islands; everything in deep mode); prefixes of starts and ends as (meth, cov) dword pairs.
    const int Nsmp = (int)c->n_samples;
    const double Favg = (double)total_pairs / (double)std::max<int64_t>(1, J);
    const int WA = std::min(Wmax, WG_NARROW_WMAX);
    const int TKB = WG_WIDE_TK;
    // exponent rows of the k-scaled log tables (pseudo count >= 1): narrow tiles score blocks of <= WG_NARROW_WMAX sites,
    // wide tiles blocks up to the job's widest window
    // The guard-free form of the term (pseudo count >= 1) rests on block totals below 2^21 (csrc/exact_log2.h: p < 1 after the
    // three float roundings, table rows down to 2^-23): the narrow tiles (blocks of <= 60 sites) always have them, the wide
    // tiles only while 255 * (the job's widest window) < 2^21 — windows beyond 8224 sites (max_cpg > 8000: round 3) score with
    // the general guarded form, which assumes nothing about the totals.
    const int term_modeA = wg_term_mode(P->pseudo_count);
    const int term_modeB = (term_modeA == 2 && 255.0 * std::max(Wmax, 1) >= 0x1p21) ? 1 : term_modeA;
    const bool ks = term_modeA == 2, ksB = term_modeB == 2;
    const int rowsA = ks ? wg_lookup_rows(P->pseudo_count, 255.0 * WG_NARROW_WMAX) : 0;
    const int rowsB = ksB ? wg_lookup_rows(P->pseudo_count, 255.0 * std::max(Wmax, 1)) : 0;
    const int rowsM = ks ? wg_lookup_rows(P->pseudo_count, 255.0 * WG_MEDIUM_WMAX) : 0;
    if (rowsA > WG_KY_KMIN + 1 || rowsB > WG_KY_KMIN + 1 || rowsM > WG_KY_KMIN + 1) { set_err(err, errlen, "internal: %d / %d / %d lookup rows", rowsA, rowsB, rowsM); return WGBSSEG_E_ARG; }
    // tile class: 0 narrow (ti starts), 1 wide, 2 medium
    // block -> start map of the narrow / medium tiles: a byte per eight blocks + forward steps.  (A byte per block — no steps — was measured for
    // the 128-start tiles of small cohorts, whose LDS has room: x8 scoring 6.73 ms against 6.59 with the coarse map, x16 12.50 / 12.07: the
    // workgroup per CU it costs outweighs the steps it saves; profiles/r04_cost_rec_ab.txt.)
    auto cmap_shift = [&](int, int) { return 3; };
    auto lds_for = [&](int ti, int cls, int ns) -> size_t {
        const int wm = cls == 2 ? WG_MEDIUM_WMAX : WG_NARROW_WMAX;
        const size_t rows = cls == 1 ? ((((size_t)ns * (WG_WIDE_TK + 1 + WG_WIDE_TS + 1) + 1) & ~(size_t)1) * 8)    // P of the ends + P of the starts, (meth, cov) as two dwords
                                     : ((((size_t)ns * (ti + wm + 1) + 3) & ~(size_t)3) * 4);             // tile-local prefixes, packed in one dword
        // guard-free kernels: just the two lookup tables, sized to the pseudo count and the tile class; otherwise the general fast tables
        const size_t tabs = (cls == 1 ? ksB : ks) ? (size_t)(cls == 1 ? rowsB : cls == 2 ? rowsM : rowsA) * (16 + 64) * sizeof(wg_d2) : sizeof(wg_fast_tables);
        return tabs + rows + (size_t)(ti + 1) * 16 + 32 +      // one 16-byte record per start (+ the closing one), 8 ints
               (cls == 1 ? 0 : ((size_t)ti * wm >> cmap_shift(ti, cls)) + 8);      // (+ the block -> start map)
    };
    int TI = 64, NSA = 1, NSB = 1, NSM = 1;
    const int ti128_max_n = 16;
    {
        double best = -1;
        for (int ti = 128; ti >= 16; ti >>= 1) {
            if (c->force_ti > 0 && ti != c->force_ti) continue;
            const int ns_opts[5] = {Nsmp, 32, 16, 8, 4};
            for (int ns : ns_opts) {
                if (ns > Nsmp) continue;
                // 128-start tiles: half the per-tile overhead for small cohorts; only with every sample in LDS at once (one group),
                // and not by default above 16 samples, where their larger rows cost a workgroup per CU
                if (ti == 128 && (ns != Nsmp || (c->force_ti != 128 && Nsmp > ti128_max_n))) continue;
                if (c->force_ns > 0 && ns != std::min(c->force_ns, Nsmp)) continue;
                const size_t l = lds_for(ti, 0, ns);
                if (l > 64 * 1024) continue;
                // workgroups per CU the score counts on: LDS is handed out in granules of 1280 bytes; 5 for the 64-start tiles and below (the form
                // with partial sums across sample groups has 95 VGPRs = 5 per CU; the one-group form has 63, but letting it count 7 makes small
                // cohorts pick 64-start tiles, measured slower than 128-start ones: x8 scoring 6.92 vs 6.59 ms, profiles/r04_cost_rec_ab.txt)
                const int wgs = (int)std::min<size_t>(ti == 128 ? 8 : 5, (160 * 1024) / (size_t)round_up((int64_t)l, 1280));
                const double q = ti * std::min<double>(Favg, WA), eff = q / (256.0 * std::ceil(q / 256.0));
                const double groups = std::ceil((double)Nsmp / ns);
                const double score = wgs * eff / (1.0 + 0.02 * (groups - 1)) * (1.0 + 0.04 * (ti / 16));   // bias to big tiles (less staging)
                if (score > best) { best = score; TI = ti; NSA = ns; }
            }
        }
        if (best < 0) { TI = 16; NSA = 1; }
        best = -1;
        const int ns_opts[6] = {Nsmp, 32, 16, 8, 4, 1};
        for (int ns : ns_opts) {
            if (ns > Nsmp) continue;
            if (c->force_ns > 0 && ns != std::min(c->force_ns, Nsmp)) continue;
            const size_t l = lds_for(WG_WIDE_TS, 1, ns);
            if (l > 64 * 1024) continue;
            const int wgs = (int)std::min<size_t>(4, (160 * 1024) / (size_t)round_up((int64_t)l, 1280));      // 103 VGPRs: 4 workgroups per CU at most
            const double groups = std::ceil((double)Nsmp / ns);
            const double score = wgs / (1.0 + 0.02 * (groups - 1));
            if (score > best) { best = score; NSB = ns; }
        }
        best = -1;
        for (int ns : ns_opts) {                                   // medium tiles: rows of 269 dwords per sample
            if (ns > Nsmp) continue;
            const size_t l = lds_for(WG_MEDIUM_TS, 2, ns);
            if (l > 64 * 1024) continue;
            const int wgs = (int)std::min<size_t>(5, (160 * 1024) / (size_t)round_up((int64_t)l, 1280));
            const double groups = std::ceil((double)Nsmp / ns);
            const double score = wgs / (1.0 + 0.02 * (groups - 1));
            if (score > best) { best = score; NSM = ns; }
        }
    }
    CostArgs caA, caB;
    memset(&caA, 0, sizeof(caA));
    caA.pc = P->pseudo_count; caA.pc2 = P->pseudo_count + P->pseudo_count;
    caA.xcd_group = 64;
    caA.cmap = 3;
    caB = caA;
    CostArgs caM = caA;
    caA.cmap = cmap_shift(TI, 0);
    caA.NS = NSA; caA.rows = rowsA;
    caB.NS = NSB; caB.rows = rowsB;
    caM.NS = NSM; caM.rows = rowsM;
    if (ks) {   // the k-scaled tables of the tile classes (same IEEE operations as on the device): built and uploaded when the pseudo count or a
                // class's exponent rows differ from what the context holds — a follow-up batch of the same call, the next call of a bench, reuse them
        const bool same = c->lookup_pc == P->pseudo_count && c->lookup_rows[0] == rowsA && c->lookup_rows[1] == rowsB && c->lookup_rows[2] == rowsM && c->lookup.p;
        if (!same) {
            static const wg_log_tables host_tabs = WG_LOG_TABLES_INIT;
            c->h_lookup.resize((size_t)(rowsA + rowsB + rowsM) * 80);
            wg_d2* dst = c->h_lookup.data();
            for (int rows : {rowsA, rowsB, rowsM}) {
                for (int x = 0; x < rows * 16; x++) dst[x] = wg_ks_iy_entry(&host_tabs, rows, x);
                for (int x = 0; x < rows * 64; x++) dst[rows * 16 + x] = wg_ks_ky_entry(&host_tabs, rows, x);
                dst += (size_t)rows * 80;
            }
            HIP_TRY(c->lookup.ensure(c->h_lookup.size() * sizeof(wg_d2)));
            HIP_TRY(hipMemcpyAsync(c->lookup.p, c->h_lookup.data(), c->h_lookup.size() * sizeof(wg_d2), hipMemcpyHostToDevice, c->sA));   // h_lookup lives in the context
            c->lookup_pc = P->pseudo_count; c->lookup_rows[0] = rowsA; c->lookup_rows[1] = rowsB; c->lookup_rows[2] = rowsM;
        }
        caA.tab = c->lookup.as<wg_d2>();
        caB.tab = caA.tab + (size_t)rowsA * 80;
        caM.tab = caB.tab + (size_t)rowsB * 80;
    }
    const size_t ldsA = (size_t)round_up((int64_t)lds_for(TI, 0, NSA), 16);
    const size_t ldsB = (size_t)round_up((int64_t)lds_for(WG_WIDE_TS, 1, NSB), 16);
    const size_t ldsM = (size_t)round_up((int64_t)lds_for(WG_MEDIUM_TS, 2, NSM), 16);
    const int term_mode = term_modeA;

    // ---- stages: bound the scored-block buffer and overlap scoring (stream A) with the recurrence (stream B) --
    int n_stages = 1;
    {
        const long long bytes = total_pairs * 8;
        n_stages = (int)std::max<long long>(1, (bytes + c->cost_budget_bytes - 1) / c->cost_budget_bytes);
        // Few chunks (one rank's share of a sharded genome): the recurrence occupies a fraction of the CUs, so let it
        // chase the scoring kernel stage by stage (measured: 71 chunks 10.1 -> 8.7 ms, 132 chunks 16.2 -> 15.0 ms).  With
        // hundreds of chunks k_dp alone (3.9 ms per 60k-site chunk, all chunks at once) beats k_dp competing for CUs.
        // (the junction patches that ride along in the batch are a few hundred sites each: they do not count)
        int n_long = 0;
        for (const ChunkDesc& d : job.h) n_long += d.len >= 8192;
        // ... unless the recurrence is the longer of the two anyway (a share of a SMALL cohort): beside the scoring kernel a step of the recurrence
        // takes ~50 ns against ~21 alone, so staging pays only when the scoring lasts longer than the ~30 ns per step it costs: at 7e11
        // evaluations/s, from ~21,000 evaluations per step of the longest chunk on (round 6, one GPU's share of 8, 71 chunks: x 8 3.69 ms in eight
        // stages, 2.63 in one; x 32 4.09 against 5.08; profiles/r06_stage_gate_ab.txt)
        const bool worth = (double)total_pairs * c->n_samples >= (double)job.max_len * c->stage_min_evals_per_step;
        if (job.max_len >= 8192) n_stages = std::max(n_stages, n_long <= 160 && worth ? 8 : 1);
        // Many chunks, all windows <= 64: one stage scores and k_dp<7,64> follows alone (1.7 ms exposed).  (Two uneven stages — the
        // recurrence of the first part of every chunk beside the scoring of the rest — were measured in round 2: 27.1-27.3 ms against 27.0
        // whatever the split, profiles/r02_tail_split_sweep.txt: the scoring kernel loses what the recurrence no longer shows.)
        if (c->force_stages > 0) n_stages = c->force_stages;
        n_stages = std::min<int>(n_stages, std::max(1, (job.max_len + 63) / 64));
    }
    // Gated stages (k_stage_gate): the two scoring streams swap roles from stage to stage — the narrow tiles of an even stage on A with its medium / wide tiles
    // beside them on A2, an odd stage the other way round (a second PAIR of streams was measured: six streams of one priority share hardware queues, every
    // share lost 10-15 %)
    const bool all_narrow = Wmax <= WG_NARROW_WMAX;
    // (one context per device: the streams of several contexts share hardware queues, where a gate could sit ahead of the very launch another context's gate waits for —
    // the waits are bounded, but nothing would be gained)
    // A job with medium / wide tiles (its recurrences are the 32-step kernels with the full LDS footprint, which lived on the drains: a share of 8, x 32 with islands,
    // 5.95 -> 6.4 ms under the gate) is gated only when it is clearly scoring-bound (x 100 with islands: 14.0 -> 13.3 ms): from 100,000 evaluations per step on.
    const double evals_per_step = (double)total_pairs * c->n_samples / std::max<double>(1.0, (double)job.max_len);
    const bool gated = c->stage_gate > 0 && n_stages > 1 && (all_narrow || evals_per_step >= c->stage_gate_wide_evals || c->stage_gate_shared) &&
                       (c->stage_gate_shared || g_live_ctx[c->device & 63].load(std::memory_order_relaxed) == 1);
    std::vector<int32_t>& sb = c->h_stage_bounds;              // (lives in the context: source of an async upload)
    {
        // equal stages but the last (gated jobs only): its recurrence is the only one that runs with the chip to itself, at twice the pace of the others.
        // Which length: the recurrences of the other stages take ~50 ns per step beside the scoring, so while the scoring of a stage lasts no longer than 1.5 x its
        // recurrence (at 7e11 evaluations/s: up to ~52,000 evaluations per step of the longest chunk) the chain of recurrences is what the step waits for, and a last
        // stage three times the others' shortens it (a share of 8, x 32: 4.08 -> 3.79 ms with the gate; equal stages + gate alone: 4.2); a scoring-bound job keeps equal
        // stages (x 200: 20.4 -> 18.35 ms with the gate, 18.7 with a last stage twice the others').  profiles/r06_stage_gate_ab.txt
        const int pct = !gated ? 100 : (c->last_stage_pct > 0 ? c->last_stage_pct : (evals_per_step <= 52500.0 ? 300 : 100));
        const double parts = n_stages > 1 ? (double)(n_stages - 1) + pct / 100.0 : 1.0;
        const int S = (int)round_up((int64_t)std::ceil((double)job.max_len / parts), 64);
        n_stages = std::min<int>(n_stages, (job.max_len + S - 1) / S);
        sb.resize((size_t)n_stages + 1);
        for (int q = 0; q < n_stages; q++) sb[(size_t)q] = (int32_t)std::min<int64_t>((int64_t)q * S, job.max_len);
        sb[(size_t)n_stages] = (int32_t)job.max_len;
    }
    HIP_TRY(c->plan_sb.ensure(sb.size() * 4));
    {
        const void* src = sb.data();
        if (c->h_sb.ensure(sb.size() * 4)) { memcpy(c->h_sb.p, src, sb.size() * 4); src = c->h_sb.p; }      // (page-locked: the copy does not block)
        HIP_TRY(hipMemcpyAsync(c->plan_sb.p, src, sb.size() * 4, hipMemcpyHostToDevice, c->sA));
    }
    HIP_TRY(c->plan_cbase.ensure((size_t)n_stages * nC * 8));
    HIP_TRY(c->plan_cum0.ensure((size_t)n_stages * nC * 4));
    HIP_TRY(c->plan_tbase.ensure((size_t)n_stages * (nC + 1) * 8 * 3));
    HIP_TRY(c->plan_cnt.ensure((size_t)n_stages * nC * 4 * 3));
    HIP_TRY(c->plan_pairs.ensure((size_t)n_stages * 8));
    HIP_TRY(c->plan_tiles.ensure((size_t)n_stages * 8 * 3));
    PlanArgs pa = {c->plan_sb.as<int32_t>(), TI, WA, TKB, n_stages, WMED};
    uint32_t* cntA = c->plan_cnt.as<uint32_t>();
    uint32_t* cntB = cntA + (size_t)n_stages * nC;
    uint32_t* cntM = cntB + (size_t)n_stages * nC;
    int64_t* tbaseA = c->plan_tbase.as<int64_t>();
    int64_t* tbaseB = tbaseA + (size_t)n_stages * (nC + 1);
    int64_t* tbaseM = tbaseB + (size_t)n_stages * (nC + 1);
    std::vector<int64_t> stage_pairs((size_t)n_stages), stage_tiles((size_t)n_stages * 3, 0);
    // No window beyond the narrow tiles' (every default-parameter genome outside CpG islands): every aligned group of TI starts is ONE narrow
    // tile, so the host knows the tile counts without asking the device (round 6: one host round trip less in front of the scoring of every
    // batch); the scored blocks per stage — what sizes the cost buffer — from the windows' statistics: all of them in one stage, at most
    // (sites of the stage) x (widest window) otherwise.
    bool check_div_m = false;
    if (all_narrow) {
        for (int stg = 0; stg < n_stages; stg++) {
            int64_t nt = 0, sites_in = 0;
            for (const ChunkDesc& d : job.h) {
                const int64_t s0 = sb[(size_t)stg], s1 = std::min<int64_t>(sb[(size_t)stg + 1], d.len);
                if (s1 > s0) { nt += (s1 - s0 + TI - 1) / TI; sites_in += s1 - s0; }
            }
            stage_tiles[3 * (size_t)stg] = nt;
            stage_pairs[(size_t)stg] = n_stages == 1 ? total_pairs : std::min<int64_t>(total_pairs, sites_in * std::max(Wmax, 1));
        }
        hipLaunchKernelGGL(k_stage_plan, dim3((unsigned)n_stages), dim3(WG_PLAN_BLOCK), 0, c->sA, v, pa, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr,
                           c->plan_cbase.as<int64_t>(), c->plan_cum0.as<uint32_t>(), tbaseA, tbaseB, tbaseM, c->plan_pairs.as<int64_t>(), c->plan_tiles.as<int64_t>());
        HIP_TRY(hipGetLastError());
    } else {
        hipLaunchKernelGGL(k_tile_count, dim3((unsigned)nC, (unsigned)n_stages), dim3(WG_BLOCK), 0, c->sA, v, pa, cntA, cntB, cntM);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_stage_plan, dim3((unsigned)n_stages), dim3(WG_PLAN_BLOCK), 0, c->sA, v, pa, (const uint32_t*)cntA, (const uint32_t*)cntB, (const uint32_t*)cntM,
                           c->plan_cbase.as<int64_t>(), c->plan_cum0.as<uint32_t>(), tbaseA, tbaseB, tbaseM, c->plan_pairs.as<int64_t>(), c->plan_tiles.as<int64_t>());
        HIP_TRY(hipGetLastError());
        // medium tiles and a pseudo count whose short division has not been tried on THEIR operand pairs yet (0 <= nmeth <= ntotal <=
        // 255 * 252: 2.1e9 pairs, ~2 ms, once per context and pseudo count, and only for a job that has windows > 60 at all)
        check_div_m = c->divs_enabled && term_modeA == 2 && WMED > 0 && c->divs_m_pc != P->pseudo_count;
        if (check_div_m) {
            HIP_TRY(c->divcheck.ensure(4));
            HIP_TRY(hipMemsetAsync(c->divcheck.p, 0, 4, c->sA));
            const int max_total = 255 * WG_MEDIUM_WMAX;
            hipLaunchKernelGGL(k_check_div, dim3((unsigned)max_total + 1), dim3(WG_BLOCK), 0, c->sA, P->pseudo_count, P->pseudo_count + P->pseudo_count,
                               max_total, c->divcheck.as<unsigned int>());
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(reinterpret_cast<JobStatus*>(c->h_status.p) + 3, c->divcheck.p, 4, hipMemcpyDeviceToHost, c->sA));
        }
        HIP_TRY(hipMemcpyAsync(stage_pairs.data(), c->plan_pairs.p, (size_t)n_stages * 8, hipMemcpyDeviceToHost, c->sA));
        HIP_TRY(hipMemcpyAsync(stage_tiles.data(), c->plan_tiles.p, (size_t)n_stages * 24, hipMemcpyDeviceToHost, c->sA));
        HIP_TRY(hipStreamSynchronize(c->sA));
        if (check_div_m) {
            c->divs_m_ok = *reinterpret_cast<const unsigned int*>(&hst[3]) == 0u;
            c->divs_m_pc = P->pseudo_count;
            if (profiling()) fprintf(stderr, "[wgbsseg] short division core for pseudo count %g on the operand pairs of a MEDIUM tile: %s\n", (double)P->pseudo_count, c->divs_m_ok ? "verified on every pair" : "NOT exact, the full core stays");
        }
    }
    std::vector<int64_t> tileA0((size_t)n_stages + 1, 0), tileB0((size_t)n_stages + 1, 0), tileM0((size_t)n_stages + 1, 0);
    for (int stg = 0; stg < n_stages; stg++) {
        tileA0[(size_t)stg + 1] = tileA0[(size_t)stg] + stage_tiles[3 * (size_t)stg];
        tileB0[(size_t)stg + 1] = tileB0[(size_t)stg] + stage_tiles[3 * (size_t)stg + 1];
        tileM0[(size_t)stg + 1] = tileM0[(size_t)stg] + stage_tiles[3 * (size_t)stg + 2];
    }
    if (tileA0[(size_t)n_stages] > 0x7fffffffLL || tileB0[(size_t)n_stages] > 0x7fffffffLL || tileM0[(size_t)n_stages] > 0x7fffffffLL) { set_err(err, errlen, "too many scoring tiles in one call"); return WGBSSEG_E_ARG; }
    HIP_TRY(c->tilesA.ensure((size_t)std::max<int64_t>(1, tileA0[(size_t)n_stages]) * sizeof(TileDesc)));
    HIP_TRY(c->tilesB.ensure((size_t)std::max<int64_t>(1, tileB0[(size_t)n_stages]) * sizeof(TileDesc)));
    HIP_TRY(c->tilesM.ensure((size_t)std::max<int64_t>(1, tileM0[(size_t)n_stages]) * sizeof(TileDesc)));
    for (int stg = 0; stg < n_stages; stg++) {
        hipLaunchKernelGGL(k_tile_emit, dim3((unsigned)nC), dim3(WG_BLOCK), 0, c->sA, v, pa, stg, tbaseA, tbaseB, tbaseM,
                           c->tilesA.as<TileDesc>() + tileA0[(size_t)stg], c->tilesB.as<TileDesc>() + tileB0[(size_t)stg],
                           c->tilesM.as<TileDesc>() + tileM0[(size_t)stg]);
        HIP_TRY(hipGetLastError());
    }
    if (gated) {
        HIP_TRY(c->stage_ctr.ensure((size_t)n_stages * 4));
        HIP_TRY(hipMemsetAsync(c->stage_ctr.p, 0, (size_t)n_stages * 4, c->sA));
    }
    HIP_TRY(hipEventRecord(c->ev[3], c->sA));
    if (gated) HIP_TRY(hipStreamWaitEvent(c->sA2, c->ev[3], 0));
    if (validate_late) {                                        // (WGBSSEG_SCAN_AFTER=2) k_validate behind the tile plan
        HIP_TRY(hipStreamWaitEvent(sS, c->ev[3], 0));
        rc = queue_validate_now();
        if (rc != WGBSSEG_OK) return rc;
        HIP_TRY(close_scan_stream());
    }
    if (host_marks) hm[4] = wall_s();
    int64_t max_stage_pairs = 1;
    for (auto x : stage_pairs) max_stage_pairs = std::max(max_stage_pairs, x);
    const int nbuf = n_stages > 1 ? (gated && n_stages > 2 ? 3 : 2) : 1;      // (gated: a stage's scoring must not wait for the recurrence two stages back when its gate opens)
    for (int b = 0; b < nbuf; b++) HIP_TRY(c->cost[b].ensure((size_t)max_stage_pairs * 8));
    // k_dp: 64-step batches when no window of the job exceeds 64 sites; otherwise 32-step batches with a second pending
    // register per lane and, for blocks longer than 128 sites, a ring of pending maxima per chunk in global memory
    int dp_mode = Wmax > 512 ? 2 : (Wmax > 64 ? 1 : 0);           // 0: <3,64>  1: <3,32>  2: <15,32> (deep windows: more workers)
    if (c->force_dp_mode > dp_mode) dp_mode = c->force_dp_mode;
    const int ringN = dp_mode ? ceil_pow2(Wmax + 128) : 0;
    const int64_t state_stride = round_up(WG_DP_STATE_HDR + (int64_t)ringN + (ringN + 1) / 2, 2);   // doubles per chunk
    HIP_TRY(c->dpstate.ensure((size_t)nC * (size_t)state_stride * 8));
    HIP_TRY(c->tmp_borders.ensure((size_t)(Jp + nC) * 4));
    HIP_TRY(c->nb.ensure((size_t)nC * 4));
    // the batch's result on the device: the CSR offsets of the chunks' border lists, then the lists — one buffer, so that a small batch
    // (a follow-up batch of junction patches) comes home in ONE copy
    const size_t out_head = (size_t)round_up((int64_t)(nC + 1) * 8, 16);
    c->out_par ^= 1;                                             // (the other buffer may still be feeding k_copy_out of the previous batch)
    DevBuf& outb = c->out_borders[c->out_par];
    HIP_TRY(outb.ensure(out_head + (size_t)(J + nC) * 4));
    int64_t* const d_boff = outb.as<int64_t>();
    int32_t* const d_bord = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(outb.p) + out_head);
    grow_events(c->ev_cost0, n_stages); grow_events(c->ev_cost1, n_stages);
    grow_events(c->ev_dp0, n_stages); grow_events(c->ev_dp1, n_stages);
    grow_events(c->ev_fork, n_stages); grow_events(c->ev_join, n_stages);
    StageView sv;
    sv.cbase = c->plan_cbase.as<int64_t>(); sv.cum0 = c->plan_cum0.as<uint32_t>(); sv.tbaseA = tbaseA; sv.tbaseB = tbaseB;
    sv.sb = c->plan_sb.as<int32_t>();
    // LDS of k_dp: two arranged batches (64 steps x 64 lanes, or 32 steps x 64 lanes x {A, B}) + M ring + fetched ring entries + flags + ring of windows / row offsets
    const int dp_wlean_off = (getenv("WGBSSEG_DP_WLEAN") && atoi(getenv("WGBSSEG_DP_WLEAN")) == 0) ? 1 : 0;      // 0: narrow batches of a wide job on the generic step (A/B, tests; read per call)
    DpArgs da = {ringN, {dp_wlean_off, 0, 0}};
#ifdef WGBSSEG_DP_TIMING
    if (const char* e = getenv("WGBSSEG_DP_DEBUG")) da.pad[1] = atoi(e);      // timing builds only: see k_dp (results are WRONG in these modes)
#endif
    c->last_dp_chunks = nC; c->last_dp_stride = state_stride;
    const size_t lds_dp = 2 * 4096 * 8 + 128 * 8 + 64 * 12 + 16 + 1024 * 6;
    if (st.wide_units) HIP_TRY(hipStreamWaitEvent(c->sA, c->ev[2], 0));      // wide tiles read the carries: scoring after the scan
    if (st.wide_units && gated) HIP_TRY(hipStreamWaitEvent(c->sA2, c->ev[2], 0));
    // Two scoring streams.  A scoring launch ends in a tail of partly filled workgroup slots (1280 on the chip), and a stage
    // with more than one tile class pays one per class: there the medium and wide tiles go to a second stream, beside the
    // narrow ones (they write disjoint rows of the cost buffer), forked off the first stream when the stage may begin
    // and joined before the stage's end is recorded (ev_fork / ev_join).  Measured, islands x32: scoring 29.49 -> 28.70 ms, x8 9.35 -> 9.14 ms.
    // Alternating the STAGES of a staged job between the streams was measured too and is not done: the one-eighth share
    // (8 stages) scored in 3.40 instead of 3.49 ms but its recurrences, which run beside the next stage's scoring, fell behind
    // (step 4.56 -> 4.79 ms); x200 / x512 gained 0.1-0.5 %.
    const bool two_cost_streams = true;
    for (int stg = 0; stg < n_stages; stg++) {
        hipStream_t const sP = gated && (stg & 1) ? c->sA2 : c->sA;
        sv.stage = stg;
        double* cbuf = c->cost[stg % nbuf].as<double>();
        const bool side = two_cost_streams && stage_tiles[3 * (size_t)stg] > 0 && (stage_tiles[3 * (size_t)stg + 1] > 0 || stage_tiles[3 * (size_t)stg + 2] > 0);
        hipStream_t const sSide = side ? (gated && (stg & 1) ? c->sA : c->sA2) : sP;
        if (stg >= nbuf) HIP_TRY(hipStreamWaitEvent(sP, c->ev_dp1[stg - nbuf], 0));   // buffer free again
        if (gated) {
            const int64_t before = stg > 0 ? stage_tiles[3 * (size_t)stg - 3] + stage_tiles[3 * (size_t)stg - 2] + stage_tiles[3 * (size_t)stg - 1] : 0;      // tiles of every class
            if (before > 0) {
                // the wait's bound: ten times what the stage before should take at 5e11 evaluations/s (its scored blocks are an upper bound for an all-narrow job), between
                // 5 ms and 2 s — a gate that opens early lets two stages interleave (the recurrence of the first starts late); one that waits for a launch which some other
                // context's work holds up (a context created on this device while the batch is in flight) gives up after a time in proportion to the job
                const double est_s = 10.0 * (double)stage_pairs[(size_t)stg - 1] * c->n_samples / 5e11;
                const long long max_ticks = (long long)(1e8 * std::min(2.0, std::max(0.005, est_s)));
                hipLaunchKernelGGL(k_stage_gate, dim3(1), dim3(64), 0, sP, (const uint32_t*)(c->stage_ctr.as<uint32_t>() + (stg - 1)),
                                   (uint32_t)std::max<int64_t>(1, before - c->stage_gate), max_ticks);
                HIP_TRY(hipGetLastError());
            }
            caA.finished = caB.finished = caM.finished = c->stage_ctr.as<uint32_t>() + stg;
        }
        HIP_TRY(hipEventRecord(c->ev_cost0[stg], sP));
        if (side) { HIP_TRY(hipEventRecord(c->ev_fork[stg], sP)); HIP_TRY(hipStreamWaitEvent(sSide, c->ev_fork[stg], 0)); }
        if (stage_tiles[3 * (size_t)stg] > 0) {
            const TileDesc* td = c->tilesA.as<TileDesc>() + tileA0[(size_t)stg];
            const int64_t nt = stage_tiles[3 * (size_t)stg];
            const bool divs = c->divs_enabled && c->divs_ok && c->divs_pc == P->pseudo_count;
            hipError_t e = term_mode == 2 ? (divs ? launch_cost_ti<3>(TI, false, v, sv, caA, td, nt, cbuf, ldsA, sP)
                                                  : launch_cost_ti<2>(TI, false, v, sv, caA, td, nt, cbuf, ldsA, sP))
                         : (term_mode == 1 ? launch_cost_ti<1>(TI, false, v, sv, caA, td, nt, cbuf, ldsA, sP)
                                           : launch_cost_ti<0>(TI, false, v, sv, caA, td, nt, cbuf, ldsA, sP));
            HIP_TRY(e);
        }
        if (stage_tiles[3 * (size_t)stg + 2] > 0) {                  // medium tiles: the narrow tiles' arithmetic on rows of 269 entries
            const TileDesc* td = c->tilesM.as<TileDesc>() + tileM0[(size_t)stg];
            const int64_t nt = stage_tiles[3 * (size_t)stg + 2];
            const bool divs = c->divs_enabled && c->divs_m_ok && c->divs_m_pc == P->pseudo_count;
            hipError_t e = term_mode == 2 ? (divs ? launch_cost_ti<3>(-1, false, v, sv, caM, td, nt, cbuf, ldsM, sSide)
                                                  : launch_cost_ti<2>(-1, false, v, sv, caM, td, nt, cbuf, ldsM, sSide))
                         : (term_mode == 1 ? launch_cost_ti<1>(-1, false, v, sv, caM, td, nt, cbuf, ldsM, sSide)
                                           : launch_cost_ti<0>(-1, false, v, sv, caM, td, nt, cbuf, ldsM, sSide));
            HIP_TRY(e);
        }
        if (stage_tiles[3 * (size_t)stg + 1] > 0) {
            const TileDesc* td = c->tilesB.as<TileDesc>() + tileB0[(size_t)stg];
            const int64_t nt = stage_tiles[3 * (size_t)stg + 1];
            hipError_t e = term_modeB == 2 ? launch_cost_ti<2>(TI, true, v, sv, caB, td, nt, cbuf, ldsB, sSide)
                         : (term_modeB == 1 ? launch_cost_ti<1>(TI, true, v, sv, caB, td, nt, cbuf, ldsB, sSide)
                                           : launch_cost_ti<0>(TI, true, v, sv, caB, td, nt, cbuf, ldsB, sSide));
            HIP_TRY(e);
        }
        if (side) { HIP_TRY(hipEventRecord(c->ev_join[stg], sSide)); HIP_TRY(hipStreamWaitEvent(sP, c->ev_join[stg], 0)); }
        HIP_TRY(hipEventRecord(c->ev_cost1[stg], sP));
        HIP_TRY(hipStreamWaitEvent(c->sB, c->ev_cost1[stg], 0));
        HIP_TRY(hipEventRecord(c->ev_dp0[stg], c->sB));
        // worker waves per chunk, measured: 64-step batches (no window > 64): 7 (whole genome 3 -> 2.14 ms, 7 -> 1.77 ms,
        // 5 / 11 / 15 -> 2.8-3.1 ms); 32-step batches (islands): round 1 measured 3 workers 4.8 ms, 7 -> 5.8 ms; round 3 (the batch loops written per role
        // since round 2: 127 instead of 174 VGPRs) 3 -> 4.28 ms, 7 -> 3.52, with the lean step in the narrow batches 3.34 (11 workers: 5.0): 7 is the default now.  64-step batches again in round 3: 5 -> 1.83, 7 -> 1.66, 11 -> 1.65 ms.
        // 16-step batches at the footprint of one scoring workgroup when the recurrence of a stage runs beside the scoring of
        // the next one: whenever the call is staged.  Windows <= 60 (no wide tile in the job): the step with the batched bookkeeping, 6.75 instead
        // of 10 VALU instructions.
        // the LAST stage's recurrence has the chip to itself: the 64-step-batch kernel is twice as fast there
        const bool dp16 = dp_mode == 0 && n_stages > 1 && stg + 1 < n_stages;
        if (dp16)                            hipLaunchKernelGGL((k_dp16<3>), dim3((unsigned)nC), dim3(64 * 4), (size_t)(2 * 16 * 64 * 8 + WG_DP_META_RING * 6), c->sB, v, sv, cbuf, c->dpstate.as<double>(), state_stride);
        else if (dp_mode == 0 && Wmax <= WG_NARROW_WMAX)
                                             hipLaunchKernelGGL((k_dp<7, 64, true>), dim3((unsigned)nC), dim3(64 * 8), lds_dp, c->sB, v, sv, cbuf, da, c->dpstate.as<double>(), state_stride);
        else if (dp_mode == 0)               hipLaunchKernelGGL((k_dp<7, 64>), dim3((unsigned)nC), dim3(64 * 8), lds_dp, c->sB, v, sv, cbuf, da, c->dpstate.as<double>(), state_stride);
        else if (dp_mode == 1) hipLaunchKernelGGL((k_dp<7, 32>), dim3((unsigned)nC), dim3(64 * 8), lds_dp, c->sB, v, sv, cbuf, da, c->dpstate.as<double>(), state_stride);
        else                   hipLaunchKernelGGL((k_dp<15, 32>), dim3((unsigned)nC), dim3(64 * 16), lds_dp, c->sB, v, sv, cbuf, da, c->dpstate.as<double>(), state_stride);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev_dp1[stg], c->sB));
    }
    // ---- traceback, compaction, copy out ---------------------------------------------------------------------
    HIP_TRY(hipEventRecord(c->ev[4], c->sB));
    hipLaunchKernelGGL(k_trace, dim3((unsigned)nC), dim3(WG_BLOCK), 0, c->sB, v, c->tmp_borders.as<int32_t>(), c->nb.as<int32_t>());
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_border_offsets, dim3(1), dim3(WG_BLOCK), 0, c->sB, c->nb.as<int32_t>(), nC, d_boff);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_gather_borders, dim3((unsigned)nC), dim3(WG_BLOCK), 0, c->sB, v, c->tmp_borders.as<int32_t>(), c->nb.as<int32_t>(),
                       (const int64_t*)d_boff, d_bord);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev[5], c->sB));
    // A chunk has at most len + 1 borders: a batch whose upper bound is small comes home in one copy (offsets + lists, through a page-locked
    // landing area); a large one sends the offsets first and then exactly the lists — or, when the caller can use them (EarlyOut), the offsets
    // with the edges of the leading items' lists, then the other items' lists, and the leading lists by k_copy_out behind the caller's back.
    const size_t small_bytes = out_head + (size_t)(J + nC) * 4;
    const bool small = small_bytes <= (512u << 10);
    const int64_t n_lead = (early && !small && early->n_lead > 0 && early->n_lead <= nC) ? early->n_lead : 0;
    // early delivery: the edges of the leading lists and, behind them, the lists of the other items (their size bounded by len + 1 each) ride
    // home with the offsets in one copy
    size_t rest_cap = 0;
    if (n_lead) for (int64_t i = n_lead; i < nC; i++) rest_cap += (size_t)job.h[(size_t)i].len + 1;
    const size_t edge_bytes = (size_t)n_lead * 2 * WG_EDGE * 4, rest_bytes = rest_cap * 4;
    if (n_lead) {
        HIP_TRY(c->edges.ensure(edge_bytes + rest_bytes + 16));
        hipLaunchKernelGGL(k_gather_edges, dim3((unsigned)nC), dim3(WG_BLOCK), 0, c->sB, (const int64_t*)d_boff, (const int32_t*)d_bord, (int)n_lead, c->edges.as<int32_t>(),
                           c->edges.as<int32_t>() + edge_bytes / 4);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev[5], c->sB));
    }
    if (!c->h_out.ensure((small ? small_bytes : out_head) + edge_bytes + rest_bytes)) { set_err(err, errlen, "out of page-locked host memory"); return WGBSSEG_E_NOMEM; }
    HIP_TRY(hipMemcpyAsync(c->h_out.p, outb.p, small ? small_bytes : (size_t)(nC + 1) * 8, hipMemcpyDeviceToHost, c->sB));
    if (n_lead) HIP_TRY(hipMemcpyAsync(reinterpret_cast<char*>(c->h_out.p) + out_head, c->edges.p, edge_bytes + rest_bytes, hipMemcpyDeviceToHost, c->sB));
    if (small || n_lead) HIP_TRY(hipEventRecord(c->ev[6], c->sB));
    if (host_marks) hm[5] = wall_s();
    if (J >= (1 << 20)) {
        // a large batch: when its recurrence is done, ~0.2 ms of traceback and copies remain — just the time the host threads
        // of the junction stitching need to wake up (stitch.h)
        HIP_TRY(hipEventSynchronize(c->ev_dp1[n_stages - 1]));
        wgstitch::Pool::get().heat();
    }
    HIP_TRY(hipStreamSynchronize(c->sB));
    memcpy(borders_off, c->h_out.p, (size_t)(nC + 1) * 8);
    const int64_t total_b = borders_off[nC];
    int32_t* borders_out = alloc(total_b);
    if (!borders_out) { set_err(err, errlen, "borders_out too small: need %lld ints", (long long)total_b); return WGBSSEG_E_CAPACITY; }
    if (small) memcpy(borders_out, reinterpret_cast<const char*>(c->h_out.p) + out_head, (size_t)total_b * 4);
    else if (n_lead && early->dest_device_visible && (reinterpret_cast<uintptr_t>(borders_out) & 15) == 0) {
        const int64_t lead_b = borders_off[n_lead];              // the leading items' borders: [0, lead_b) of the lists
        // on the scan stream: idle by now, the lowest priority (the follow-up batch's kernels go first) and — as the one stream of its priority — a hardware
        // queue of its own (on a stream of the scoring streams' priority k_copy_out shared a queue with them: the follow-up batch's first kernel waited for it)
        HIP_TRY(hipStreamWaitEvent(c->sC, c->ev[5], 0));
        hipLaunchKernelGGL(k_copy_out, dim3(512), dim3(WG_BLOCK), 0, c->sC, (const int32_t*)d_bord, borders_out, lead_b);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->evD, c->sC));
        c->out_pending = true;
        memcpy(borders_out + lead_b, reinterpret_cast<const char*>(c->h_out.p) + out_head + edge_bytes, (size_t)(total_b - lead_b) * 4);
        early->edges = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(c->h_out.p) + out_head);
        early->pending = true;
    } else {
        HIP_TRY(hipMemcpyAsync(borders_out, d_bord, (size_t)total_b * 4, hipMemcpyDeviceToHost, c->sB));
        HIP_TRY(hipEventRecord(c->ev[6], c->sB));
        HIP_TRY(hipStreamSynchronize(c->sB));
    }
    if (host_marks) hm[6] = wall_s();
    // (events, not stream synchronisations: the scoring stream's work precedes the recurrence's by its events, and the scan stream ends in evS — a stream
    // synchronisation queues a marker of its own, which waited ~0.2 ms behind k_copy_out when the two streams shared a hardware queue)
    HIP_TRY(hipEventSynchronize(c->evS));
    // the scan's verdict (segmentor.cpp:186-189).  With the scan beside the scoring kernel an invalid file is found out at the end
    // of the batch; every kernel downstream of the counts is safe on such data (LDS indices out of a table's range read zeros,
    // the traceback bounds its steps) and what it produced is dropped here.
    if (hst[2].first_bad != ~0ULL) return report_bad_site(c, hst[2], err, errlen);

    // ---- timings ---------------------------------------------------------------------------------------------
    static const bool timeline = getenv("WGBSSEG_PROFILE") && atoi(getenv("WGBSSEG_PROFILE")) >= 2;
    if (timeline) {       // device time line of the batch, ms after its first event (scoring / recurrence: first and last stage)
        auto at = [&](hipEvent_t e) { float x = 0; (void)hipEventElapsedTime(&x, c->ev[0], e); return (double)x; };
        if (n_stages > 1) {
            fprintf(stderr, "[wgbsseg] %d stages%s, bounds", n_stages, gated ? " (gated: alternating scoring streams)" : "");
            for (int q = 0; q <= n_stages; q++) fprintf(stderr, " %d", (int)sb[(size_t)q]);
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "[wgbsseg] batch of %d chunks, %lld sites: windows done %.3f | stats copied %.3f | scan pass (its own stream) %.3f .. %.3f | plan + tiles done %.3f | "
                "scoring %.3f .. %.3f | recurrence %.3f .. %.3f | trace %.3f .. %.3f | borders on the host %.3f\n", nC, (long long)J,
                at(c->ev[1]), at(c->ev[7]), at(c->ev[8]), at(c->ev[2]), at(c->ev[3]), at(c->ev_cost0[0]), at(c->ev_cost1[n_stages - 1]),
                at(c->ev_dp0[0]), at(c->ev_dp1[n_stages - 1]), at(c->ev[4]), at(c->ev[5]), at(c->ev[6]));
    }
    wgbsseg_timings& T = c->tim;
    if (!c->accumulate) memset(&T, 0, sizeof(T));
    float ms = 0;
    // the scan pass = k_scan for a job with wide tiles (it reads every chunk row of the batch, after k_validate has read the same bytes), k_validate otherwise
    if (st.wide_units) HIP_TRY(hipEventElapsedTime(&ms, c->ev[10], c->ev[11])); else HIP_TRY(hipEventElapsedTime(&ms, c->ev[8], c->ev[9]));
    T.scan_ms += ms;
    // algorithmic bytes of the pass: with wide units k_scan reads every chunk row of the batch; without, k_validate reads the
    // batch's not-yet-validated sites once
    const int64_t scan_bytes = 2 * (st.wide_units ? J : job.val_sites) * c->n_samples;
    if (scan_bytes > T.scan_main_bytes) { T.scan_main_bytes = scan_bytes; T.scan_main_ms = ms; }
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); T.window_ms += ms;
    float cost_end = 0;      // scoring time = the union of the stages' intervals (neighbouring stages overlap on the two scoring streams)
    for (int stg = 0; stg < n_stages; stg++) {
        float t0 = 0, t1 = 0;
        HIP_TRY(hipEventElapsedTime(&t0, c->ev[0], c->ev_cost0[stg])); HIP_TRY(hipEventElapsedTime(&t1, c->ev[0], c->ev_cost1[stg]));
        if (t1 > std::max(t0, cost_end)) T.cost_ms += t1 - std::max(t0, cost_end);
        cost_end = std::max(cost_end, t1);
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_dp0[stg], c->ev_dp1[stg])); T.dp_ms += ms;
    }
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[4], c->ev[5])); T.trace_ms += ms;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[6])); T.total_ms += ms;
    T.sites += J; T.pairs += total_pairs; T.evals += total_pairs * c->n_samples;
    T.scan_bytes += scan_bytes; T.max_window = std::max<int32_t>(T.max_window, Wmax);
    T.n_stages = std::max<int32_t>(T.n_stages, n_stages); T.scan_launches += 1;
    if (term_mode == 2 && c->divs_enabled && c->divs_ok && c->divs_pc == P->pseudo_count) T.div_short = 1;
    c->last_sites = J; c->last_pairs = total_pairs; c->last_stages = n_stages; c->last_valid = true;
    c->batch_open = false;
    if (host_marks) {
        hm[7] = wall_s();
        fprintf(stderr, "[wgbsseg]   host clock of the batch, us after entry: tables up %.0f | windows queued %.0f | statistics here %.0f | scoring's tiles queued %.0f | everything queued %.0f | results here %.0f | return %.0f\n",
                (hm[1] - hm[0]) * 1e6, (hm[2] - hm[0]) * 1e6, (hm[3] - hm[0]) * 1e6, (hm[4] - hm[0]) * 1e6, (hm[5] - hm[0]) * 1e6, (hm[6] - hm[0]) * 1e6, (hm[7] - hm[0]) * 1e6);
    }
    return WGBSSEG_OK;
}
}  // namespace

extern "C" {

int wgbsseg_segment_chunks(wgbsseg_ctx* c, const int64_t* chunk_start0, const int32_t* chunk_len, int64_t n_chunks,
                           const wgbsseg_params* P, int32_t* borders_out, int64_t borders_cap, int64_t* borders_off,
                           char* err, size_t errlen)
{
    if (!borders_out) { set_err(err, errlen, "NULL borders pointer"); return WGBSSEG_E_ARG; }
    BorderAlloc alloc = [&](int64_t total) -> int32_t* { return total <= borders_cap ? borders_out : nullptr; };
    return segment_chunks_impl(c, chunk_start0, chunk_len, n_chunks, P, alloc, borders_off, err, errlen);
}

int wgbsseg_segment_chunks_host(const uint8_t* betas, int64_t n_samples, int64_t sample_pitch_bytes, int64_t n_sites_total,
                                const uint32_t* loci, const int64_t* chunk_start0, const int32_t* chunk_len, int64_t n_chunks,
                                const wgbsseg_params* params, int device, int32_t* borders_out, int64_t borders_cap,
                                int64_t* borders_off, char* err, size_t errlen)
{
    if (!betas || n_samples < 1 || sample_pitch_bytes < 2 * n_sites_total) { set_err(err, errlen, "bad beta buffer"); return WGBSSEG_E_ARG; }
    wgbsseg_ctx* c = nullptr;
    int rc = wgbsseg_create(device, &c, err, errlen);
    if (rc != WGBSSEG_OK) return rc;
    std::vector<const uint8_t*> ptrs((size_t)n_samples);
    for (int64_t s = 0; s < n_samples; s++) ptrs[(size_t)s] = betas + s * sample_pitch_bytes;
    rc = wgbsseg_set_betas_host(c, ptrs.data(), n_samples, n_sites_total, err, errlen);
    if (rc == WGBSSEG_OK) rc = wgbsseg_set_loci_host(c, loci, n_sites_total, err, errlen);
    if (rc == WGBSSEG_OK) rc = wgbsseg_segment_chunks(c, chunk_start0, chunk_len, n_chunks, params, borders_out, borders_cap, borders_off, err, errlen);
    wgbsseg_destroy(c);
    return rc;
}

}  // extern "C"

namespace {
// One GPU batch on one context: 0-based resident-relative site ranges -> CSR of relative borders in page-locked memory
// that stays alive in the context (buffer `slot` of c->pinned) while ropes point into it.
int run_ctx_batch(wgbsseg_ctx* c, const std::vector<int64_t>& st0, const std::vector<int32_t>& ln, const wgbsseg_params* P, int64_t slot,
                  bool accumulate, const int32_t*& flat, std::vector<int64_t>& off, std::unique_ptr<int32_t[]>& owned, std::string& msg, EarlyOut* early = nullptr)
{
    off.resize(st0.size() + 1);
    int32_t* dst = nullptr;
    BorderAlloc alloc = [&](int64_t total) -> int32_t* {
        if (c->pinned.size() <= (size_t)slot) c->pinned.resize((size_t)slot + 1);
        PinnedBuf& pb = c->pinned[(size_t)slot];
        dst = pb.ensure((size_t)std::max<int64_t>(total, 1) * 4) ? reinterpret_cast<int32_t*>(pb.p) : nullptr;
        if (early) early->dest_device_visible = dst != nullptr;
        if (!dst) { owned.reset(new int32_t[(size_t)std::max<int64_t>(total, 1)]); dst = owned.get(); }   // pageable fallback
        return dst;
    };
    char ebuf[512] = {0};
    const bool acc_before = c->accumulate;
    c->accumulate = accumulate;
    const int rc = segment_chunks_impl(c, st0.data(), ln.data(), (int64_t)st0.size(), P, alloc, off.data(), ebuf, sizeof(ebuf), true, early);
    c->accumulate = acc_before;
    if (rc != WGBSSEG_OK) { msg = ebuf; return rc; }
    flat = dst;
    return WGBSSEG_OK;
}

int map_stitch_rc(int rc, const std::string& msg, char* err, size_t errlen)
{
    if (rc == 0) return WGBSSEG_OK;
    set_err(err, errlen, "%s", msg.c_str());
    return rc == wgstitch::E_CAPACITY ? WGBSSEG_E_CAPACITY : (rc < -1 ? rc : WGBSSEG_E_ARG);
}
bool speculation_on()
{
    static const bool on = !(getenv("WGBSSEG_NO_SPECULATION") && atoi(getenv("WGBSSEG_NO_SPECULATION")));
    return on;
}
}  // namespace

extern "C" {

int wgbsseg_segment_regions(wgbsseg_ctx* c, const int64_t* region_start, const int64_t* region_end, int64_t n_regions,
                            int64_t chunk_size, const wgbsseg_params* P, int32_t* borders_out, int64_t borders_cap,
                            int64_t* borders_off, int64_t* stats, char* err, size_t errlen)
{
    if (!c || !P) { set_err(err, errlen, "bad arguments to segment_regions"); return WGBSSEG_E_ARG; }
    int64_t n_batches = 0;
    wgstitch::BatchFn run_batch = [&](const std::vector<wgstitch::Sites>& todo, wgstitch::BatchResult& res, std::string& msg) -> int {
        std::vector<int64_t> st0(todo.size()), off;
        std::vector<int32_t> ln(todo.size());
        for (size_t i = 0; i < todo.size(); i++) {
            st0[i] = todo[i].first - 1;
            if (todo[i].second - todo[i].first > 0x7fffffff) { msg = "chunk too long"; return WGBSSEG_E_ARG; }
            ln[i] = (int32_t)(todo[i].second - todo[i].first);
        }
        const int32_t* flat = nullptr;
        std::unique_ptr<int32_t[]> owned;
        // the stitcher can start on the edges of the chunks' lists (res.n_lead = its number of chunks): ask for early delivery
        static const bool early_on = !(getenv("WGBSSEG_NO_EARLY") && atoi(getenv("WGBSSEG_NO_EARLY")));
        EarlyOut eo;
        eo.n_lead = early_on ? res.n_lead : 0;
        const int rc = run_ctx_batch(c, st0, ln, P, n_batches, n_batches > 0, flat, off, owned, msg, eo.n_lead > 0 ? &eo : nullptr);
        if (rc != WGBSSEG_OK) return rc;
        if (owned) res.owned.push_back(std::move(owned));
        res.set_csr(flat, off.data(), todo.size());
        if (eo.pending) {
            res.edges = eo.edges; res.edge_n = WG_EDGE;
            res.finish = [c](std::string& m) -> int {
                char e[256] = {0};
                const int r = wait_pending_output(c, e, sizeof e);
                if (r != WGBSSEG_OK) m = e;
                return r;
            };
        }
        n_batches++;
        return WGBSSEG_OK;
    };
    std::string msg;
    int rc = wgstitch::segment_regions(region_start, region_end, n_regions, chunk_size, run_batch, borders_out, borders_cap,
                                       borders_off, stats, msg, speculation_on());
    {   // (an error path of the stitcher may have left the first batch's lists on their way: nothing may write into the context's buffers after this call)
        char e[256] = {0};
        const int r2 = wait_pending_output(c, e, sizeof e);
        if (rc == 0 && r2 != WGBSSEG_OK) { rc = r2; msg = e; }
    }
    if (profiling()) {
        fprintf(stderr, "[wgbsseg] segment_regions: %lld batches; allocations since the last report: %lld calls, %.1f MB, %.1f ms; "
                "device ms: scan %.2f window %.2f cost %.2f dp %.2f trace %.2f, time line %.2f\n",
                (long long)n_batches, g_alloc_calls.exchange(0), (double)g_alloc_bytes.exchange(0) * 1e-6, (double)g_alloc_us.exchange(0) * 1e-3,
                c->tim.scan_ms, c->tim.window_ms, c->tim.cost_ms, c->tim.dp_ms, c->tim.trace_ms, c->tim.total_ms);
    }
    return map_stitch_rc(rc, msg, err, errlen);
}

int wgbsseg_set_site_base(wgbsseg_ctx* c, int64_t site_base)
{
    if (!c || site_base < 0) return WGBSSEG_E_ARG;
    c->site_base = site_base;
    return WGBSSEG_OK;
}

// The native chunk grid + stitching around a caller-supplied chunk engine (see include/wgbsseg.h).
int wgbsseg_stitch_regions(const int64_t* region_start, const int64_t* region_end, int64_t n_regions, int64_t chunk_size,
                           wgbsseg_batch_fn fn, void* user, int32_t speculate, int32_t* borders_out, int64_t borders_cap,
                           int64_t* borders_off, int64_t* stats, char* err, size_t errlen)
{
    if (!fn) { set_err(err, errlen, "stitch_regions: no chunk engine"); return WGBSSEG_E_ARG; }
    wgstitch::BatchFn run_batch = [&](const std::vector<wgstitch::Sites>& todo, wgstitch::BatchResult& res, std::string& msg) -> int {
        std::vector<int64_t> s(todo.size()), e(todo.size());
        for (size_t i = 0; i < todo.size(); i++) { s[i] = todo[i].first; e[i] = todo[i].second; }
        res.ptr.assign(todo.size(), nullptr);
        res.cnt.assign(todo.size(), 0);
        const int rc = fn(user, s.data(), e.data(), (int64_t)todo.size(), res.ptr.data(), res.cnt.data());
        if (rc != 0) { msg = "the chunk engine failed (code " + std::to_string(rc) + ")"; return rc < -1 ? rc : WGBSSEG_E_ARG; }
        for (size_t i = 0; i < todo.size(); i++)
            if (!res.ptr[i] || res.cnt[i] < 2 || res.ptr[i][0] != 0 || (int64_t)res.ptr[i][res.cnt[i] - 1] != e[i] - s[i]) {
                msg = "the chunk engine returned a malformed border list for sites [" + std::to_string(s[i]) + ", " + std::to_string(e[i]) + ")";
                return WGBSSEG_E_ARG;
            }
        return 0;
    };
    std::string msg;
    const int rc = wgstitch::segment_regions(region_start, region_end, n_regions, chunk_size, run_batch, borders_out, borders_cap,
                                             borders_off, stats, msg, speculate != 0);
    return map_stitch_rc(rc, msg, err, errlen);
}

// The items of the first batch of wgbsseg_segment_regions / wgbsseg_stitch_regions over these regions (see include/wgbsseg.h).
int wgbsseg_first_batch_items(const int64_t* region_start, const int64_t* region_end, int64_t n_regions, int64_t chunk_size,
                              int32_t speculate, int64_t* starts, int64_t* ends, int64_t cap, int64_t* n_items, int64_t* n_chunks,
                              char* err, size_t errlen)
{
    if (!region_start || !region_end || n_regions < 1 || chunk_size < 1 || !n_items) { set_err(err, errlen, "bad arguments to first_batch_items"); return WGBSSEG_E_ARG; }
    wgstitch::FirstBatch fb;
    std::string msg;
    const int rc = wgstitch::first_batch(region_start, region_end, n_regions, chunk_size, speculate != 0, fb, msg);
    if (rc != 0) return map_stitch_rc(rc, msg, err, errlen);
    *n_items = (int64_t)fb.items.size();
    if (n_chunks) *n_chunks = fb.n_chunks;
    if (starts && ends) {
        if (cap < (int64_t)fb.items.size()) { set_err(err, errlen, "first_batch_items: %lld items, room for %lld", (long long)fb.items.size(), (long long)cap); return WGBSSEG_E_CAPACITY; }
        for (size_t i = 0; i < fb.items.size(); i++) { starts[i] = fb.items[i].first; ends[i] = fb.items[i].second; }
    }
    return WGBSSEG_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------
// Share groups: one process, several GPUs (or several contexts on one GPU).  The chunk grid of a region list is cut
// into contiguous runs of chunks, one per share, balanced by the work the chunks hold (number of scored blocks,
// from the loci); a share keeps only its own window of the beta bytes.  Every batch of the stitching loop is routed
// item by item to the share that holds it and runs on one host thread per share; the reference's pairwise tree
// then runs ONCE, on the host, over all shares' results — the answer does not depend on the number of shares.
// ------------------------------------------------------------------------------------------------------------
struct wgbsseg_group {
    std::vector<wgbsseg_ctx*> shares;
    std::vector<int64_t> own_lo, own_hi;     // 0-based sites [lo, hi) of the chunks a share owns (hi == lo: none)
    std::vector<int64_t> win_lo, win_hi;     // resident window of the share: owned sites +- halo, inside [0, n_sites)
    std::vector<int64_t> rs, re;             // the planned regions (1-based half-open)
    std::vector<int64_t> share_chunks, share_work;
    int64_t chunk_size = 0, n_sites = 0, halo = 0;
    wgbsseg_params P = {};
    bool planned = false;
    std::vector<char> loaded;
    // streaming upload (wgbsseg_group_load_host_async): one uploader per share; `ready` = sites of the share's window that are
    // resident for EVERY sample, counted from the window's first site
    struct Loader {
        std::thread th;
        std::atomic<int64_t> ready{0};
        std::atomic<int> finished{0};
        int rc = WGBSSEG_OK;
        std::string msg;
    };
    std::vector<std::unique_ptr<Loader>> loaders;
    bool streaming = false;
};

namespace {

// number of scored blocks of a chunk: sum over its sites k of F_k (segmentor.cpp:111-117), by two pointers
int64_t chunk_work(const uint32_t* loci, int64_t lo, int64_t hi, uint32_t max_cpg, uint32_t max_bp)
{
    int64_t w = 0, e = lo;
    for (int64_t k = lo; k < hi; k++) {
        if (e < k + 1) e = k + 1;
        while (e < hi && e - k < (int64_t)max_cpg && loci[e] >= loci[k] && (uint64_t)loci[e] - loci[k] <= max_bp) e++;
        w += e - k;
    }
    return w;
}

template <class F>
void parallel_for(int64_t n, int max_threads, F f)
{
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(max_threads, (int64_t)std::thread::hardware_concurrency()), n));
    if (T <= 1) { for (int64_t i = 0; i < n; i++) f(i); return; }
    std::atomic<int64_t> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([&]() { for (int64_t i; (i = next.fetch_add(1)) < n;) f(i); });
    for (auto& x : th) x.join();
}

}  // namespace

namespace {

// The CPUs next to a device: /sys/bus/pci/devices/<bus id>/local_cpulist (e.g. "0-63,128-191"), cut to the CPUs this process may use.
// An upload thread that runs on the OTHER socket of a two-socket host fills page-locked pieces over there and the DMA then crosses the
// socket link: measured on an MI355X box (round 5, x200 = 11.3 GB): the same upload at 9 .. 35 GB/s from run to run with free-running
// threads.  false: unknown (no sysfs, every CPU listed, WGBSSEG_UPLOAD_PIN=0) — the threads then run where the scheduler puts them.
// The CPUs next to a device (/sys/bus/pci/devices/<bus id>/local_cpulist) that this process may run on, as an affinity mask of up to WG_MAX_CPUS
// CPUs (a fixed cpu_set_t stops at 1024 and sched_getaffinity then FAILS on a larger host: ADVICE r05).  Worked out once per context (create).
bool device_local_cpus(int device, NearCpus* out)
{
    out->valid = false;
    { const char* e = getenv("WGBSSEG_UPLOAD_PIN"); if (e && atoi(e) == 0) return false; }
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return false; }
    for (char* q = bus; *q; q++) if (*q >= 'A' && *q <= 'F') *q = (char)(*q - 'A' + 'a');       // sysfs names are lower case
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bus);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    char line[8192] = {0};
    const bool got = fgets(line, (int)sizeof line, f) != nullptr;
    fclose(f);
    if (!got) return false;
    const size_t bytes = CPU_ALLOC_SIZE(WG_MAX_CPUS);
    std::vector<unsigned long> allowed(bytes / sizeof(unsigned long), 0ul);
    out->bits.assign(bytes / sizeof(unsigned long), 0ul);
    cpu_set_t* A = reinterpret_cast<cpu_set_t*>(allowed.data());
    cpu_set_t* L = reinterpret_cast<cpu_set_t*>(out->bits.data());
    if (sched_getaffinity(0, bytes, A) != 0) return false;
    int n_local = 0;
    for (const char* q = line; *q;) {
        while (*q == ',' || *q == ' ' || *q == '\n') q++;
        if (*q < '0' || *q > '9') break;
        char* end = nullptr;
        long a = strtol(q, &end, 10), b = a;
        if (*end == '-') b = strtol(end + 1, &end, 10);
        for (long x = a; x <= b && x < WG_MAX_CPUS; x++) if (CPU_ISSET_S((int)x, bytes, A)) { CPU_SET_S((int)x, bytes, L); n_local++; }
        q = end;
    }
    if (n_local == 0 || n_local == CPU_COUNT_S(bytes, A)) return false;      // nothing to choose from
    out->valid = true;
    return true;
}
inline void pin_to(const NearCpus& n)
{
    if (n.valid) (void)pthread_setaffinity_np(pthread_self(), n.bits.size() * sizeof(unsigned long), reinterpret_cast<const cpu_set_t*>(n.bits.data()));
}
// Page-cached file bytes -> a page-locked staging piece (round 6, profiles/r06_upload_ab.txt):
//   * the piece's pages are mapped in ONE call (madvise MADV_POPULATE_READ, Linux >= 5.14) instead of a minor fault per 4 KB page as the copy touches them;
//   * non-temporal stores: the staging piece is written once and read by the DMA engine — no read-for-ownership, no cache pollution.
// WGBSSEG_UPLOAD_POPULATE=0 / WGBSSEG_UPLOAD_NT=0 switch either off (A/B).  x200 (11.3 GB) end to end on one box: 0.38-0.45 s -> 0.32 s with both and
// 8 x 2 MB pieces; x32 0.107-0.114 -> 0.098-0.110.  The rate of ONE setting still swings between 30 and 47 GB/s from run to run on the same box.
struct FillMode { bool populate = true, nt = true; };
inline FillMode fill_mode()
{
    FillMode m;
    { const char* e = getenv("WGBSSEG_UPLOAD_POPULATE"); if (e) m.populate = atoi(e) != 0; }
    { const char* e = getenv("WGBSSEG_UPLOAD_NT"); if (e) m.nt = atoi(e) != 0; }
    return m;
}
inline void fill_piece(void* dst, const uint8_t* src, size_t n, const FillMode& m)
{
#ifdef MADV_POPULATE_READ
    if (m.populate) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(src) & ~(uintptr_t)4095, b = (reinterpret_cast<uintptr_t>(src) + n + 4095) & ~(uintptr_t)4095;
        (void)madvise(reinterpret_cast<void*>(a), b - a, MADV_POPULATE_READ);      // (an error — an old kernel, not a mapping — just leaves the faults to the copy)
    }
#endif
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    if (m.nt && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        __m128i* d = reinterpret_cast<__m128i*>(dst);
        const __m128i* s = reinterpret_cast<const __m128i*>(src);
        size_t v = n >> 6;
        for (; v > 0; v--, s += 4, d += 4) {
            const __m128i x0 = _mm_loadu_si128(s), x1 = _mm_loadu_si128(s + 1), x2 = _mm_loadu_si128(s + 2), x3 = _mm_loadu_si128(s + 3);
            _mm_stream_si128(d, x0); _mm_stream_si128(d + 1, x1); _mm_stream_si128(d + 2, x2); _mm_stream_si128(d + 3, x3);
        }
        _mm_sfence();
        const size_t done = n & ~(size_t)63;
        if (n > done) memcpy(static_cast<char*>(dst) + done, src + done, n - done);
        return;
    }
#endif
    memcpy(dst, src, n);
}

int upload_rows_streaming(wgbsseg_ctx* c, uint8_t* dst, int64_t dst_pitch, const uint8_t* const* rows, int64_t n_rows, int64_t row_bytes,
                          std::atomic<int64_t>& ready_sites, std::string& msg)
{
    const double t0 = wall_s();
    // (measured, free-running threads: 4 threads x 1 MB 33.5 GB/s, x 2 MB 26-29, 8-16 threads 17-30: profiles/r02_upload_sweep.txt.  Round 5, threads on the
    // device's CPUs: 8 threads x 4 MB 37-42 GB/s on 11.3 GB (x200: 0.35-0.38 s end to end against 0.43-0.50 with 4 x 1 MB); x32's 1.8 GB the same either
    // way: profiles/r05_upload_pinned_ab.txt.  Large cohorts take the wide form.  Round 6: 8 x 2 MB with the pieces' pages mapped per call and a non-temporal
    // fill: x200 0.32 s end to end on both runs of the A/B against 0.33-0.45 for 8 x 4 MB, profiles/r06_upload_ab.txt.)
    const bool big = row_bytes * n_rows >= (4LL << 30);
    int64_t piece = big ? (2 << 20) : (1 << 20);
    { const char* e = getenv("WGBSSEG_UPLOAD_PIECE_KB"); if (e && atoi(e) >= 64) piece = (int64_t)atoi(e) << 10; }
    const int64_t ppr = (row_bytes + piece - 1) / piece, n_tasks = ppr * n_rows;
    int T = big ? 8 : 4;
    { const char* e = getenv("WGBSSEG_UPLOAD_THREADS"); if (e && atoi(e) > 0) T = atoi(e); }
    T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(T, 64), n_tasks));
    std::vector<std::atomic<int>> rows_done((size_t)ppr);
    for (auto& x : rows_done) x.store(0);
    std::atomic<int64_t> next(0), pieces_done(0);
    std::mutex mu;
    std::vector<hipError_t> terr((size_t)T, hipSuccess);
    const bool pin = c->near_cpus.valid;                         // upload threads (and the pieces they allocate and fill) on the device's socket
    auto complete = [&](int64_t task) {                          // task = p * n_rows + r: row r's piece p is on the device
        const int64_t p = task / n_rows;
        if (rows_done[(size_t)p].fetch_add(1) + 1 == (int)n_rows) {
            std::lock_guard<std::mutex> lk(mu);
            int64_t d = pieces_done.load();
            while (d < ppr && rows_done[(size_t)d].load() == (int)n_rows) d++;
            pieces_done.store(d);
            ready_sites.store(std::min<int64_t>(d * piece, row_bytes) / 2);
        }
    };
    const FillMode fm = fill_mode();
    int depth = 2;                                               // staging pieces (copies in flight) per upload thread
    { const char* e = getenv("WGBSSEG_UPLOAD_DEPTH"); if (e && atoi(e) >= 2 && atoi(e) <= 8) depth = atoi(e); }
    auto worker = [&](int t) {
        pin_to(c->near_cpus);
        hipError_t e = hipSetDevice(c->device);
        hipStream_t st = nullptr;
        hipEvent_t ev[8] = {};
        void* stg[8] = {};
        int64_t task_of[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        for (int k = 0; k < depth && e == hipSuccess; k++) e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
        // (allocated — and so first touched — by the thread that fills them, which runs on the device's CPUs: the pieces sit on that socket)
        for (int k = 0; k < depth && e == hipSuccess; k++) e = hipHostMalloc(&stg[k], (size_t)piece, hipHostMallocDefault);
        int k = 0;
        while (e == hipSuccess) {
            const int64_t it = next.fetch_add(1);
            if (it >= n_tasks) break;
            const int64_t p = it / n_rows, r = it % n_rows, o = p * piece, b = std::min<int64_t>(piece, row_bytes - o);
            if (task_of[k] >= 0) { e = hipEventSynchronize(ev[k]); if (e != hipSuccess) break; complete(task_of[k]); task_of[k] = -1; }
            fill_piece(stg[k], rows[r] + o, (size_t)b, fm);
            e = hipMemcpyAsync(dst + r * dst_pitch + o, stg[k], (size_t)b, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipEventRecord(ev[k], st);
            task_of[k] = it;
            k = (k + 1) % depth;
        }
        for (int q = 0; q < depth && e == hipSuccess; q++, k = (k + 1) % depth)
            if (task_of[k] >= 0) { e = hipEventSynchronize(ev[k]); if (e == hipSuccess) complete(task_of[k]); task_of[k] = -1; }
        if (st) (void)hipStreamDestroy(st);
        for (auto& x : ev) if (x) (void)hipEventDestroy(x);
        for (auto& x : stg) if (x) (void)hipHostFree(x);
        terr[(size_t)t] = e;
    };
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back(worker, t);
    for (auto& x : th) x.join();
    for (int t = 0; t < T; t++)
        if (terr[(size_t)t] != hipSuccess) { msg = std::string("HIP error during the upload: ") + hipGetErrorString(terr[(size_t)t]); return WGBSSEG_E_HIP; }
    ready_sites.store(row_bytes / 2);
    if (profiling()) fprintf(stderr, "[wgbsseg] betas to the device (streaming): %.1f ms, %.1f GB/s (%d upload threads x %d pieces of %lld KB%s%s%s)\n", (wall_s() - t0) * 1e3,
                             (double)row_bytes * n_rows / (wall_s() - t0) * 1e-9, T, depth, (long long)(piece >> 10), pin ? ", on the device's CPUs" : "",
                             fm.populate ? ", pages mapped per piece" : "", fm.nt ? ", non-temporal fill" : "");
    return WGBSSEG_OK;
}

}  // namespace

extern "C" {

int wgbsseg_group_create(const int32_t* devices, int32_t n_shares, wgbsseg_group** out, char* err, size_t errlen)
{
    if (!out) { set_err(err, errlen, "out is NULL"); return WGBSSEG_E_ARG; }
    *out = nullptr;
    if (!devices || n_shares < 1 || n_shares > 1024) { set_err(err, errlen, "group_create: need 1..1024 shares"); return WGBSSEG_E_ARG; }
    std::unique_ptr<wgbsseg_group> g(new (std::nothrow) wgbsseg_group());
    if (!g) { set_err(err, errlen, "out of host memory"); return WGBSSEG_E_NOMEM; }
    // Shares that double up on a device (dry runs of the N > 1 form on fewer GPUs): their scan streams run at the scoring streams' priority.  The lowest priority
    // is right where k_validate competes with the kernels of its OWN batch only; beside seven other shares' scoring kernels it would wait for the end of THEIR batches
    // (measured: a group of eight on one MI355X 34.4 -> 38.7-40.6 ms per step).
    bool doubled = false;
    for (int32_t a = 0; a < n_shares && !doubled; a++) for (int32_t b = a + 1; b < n_shares; b++) if (devices[a] == devices[b]) { doubled = true; break; }
    for (int32_t d = 0; d < n_shares; d++) {
        wgbsseg_ctx* c = nullptr;
        const int rc = create_ctx(devices[d], !doubled, &c, err, errlen);
        if (rc != WGBSSEG_OK) { for (auto* x : g->shares) wgbsseg_destroy(x); return rc; }
        g->shares.push_back(c);
    }
    g->loaded.assign((size_t)n_shares, 0);
    *out = g.release();
    return WGBSSEG_OK;
}

void group_join_loaders(wgbsseg_group* g);

void wgbsseg_group_destroy(wgbsseg_group* g)
{
    if (!g) return;
    group_join_loaders(g);
    // releasing gigabytes of device buffers takes milliseconds per context: do the shares side by side
    parallel_for((int64_t)g->shares.size(), 64, [&](int64_t d) { wgbsseg_destroy(g->shares[(size_t)d]); });
    delete g;
}

int32_t wgbsseg_group_size(const wgbsseg_group* g) { return g ? (int32_t)g->shares.size() : 0; }

int wgbsseg_plan_shares(const uint32_t* loci, int64_t n_sites, const int64_t* region_start, const int64_t* region_end, int64_t n_regions,
                        int64_t chunk_size, const wgbsseg_params* P, int32_t n_shares, int64_t halo, int64_t* own_lo, int64_t* own_hi,
                        int64_t* win_lo, int64_t* win_hi, int64_t* share_chunks, int64_t* share_work, char* err, size_t errlen)
{
    return wgbsseg_plan_shares_weighted(loci, n_sites, region_start, region_end, n_regions, chunk_size, P, n_shares, nullptr, halo, own_lo, own_hi,
                                        win_lo, win_hi, share_chunks, share_work, err, errlen);
}

int wgbsseg_plan_shares_weighted(const uint32_t* loci, int64_t n_sites, const int64_t* region_start, const int64_t* region_end, int64_t n_regions,
                                 int64_t chunk_size, const wgbsseg_params* P, int32_t n_shares, const double* weights, int64_t halo, int64_t* own_lo,
                                 int64_t* own_hi, int64_t* win_lo, int64_t* win_hi, int64_t* share_chunks, int64_t* share_work, char* err, size_t errlen)
{
    if (!loci || n_sites < 1 || !region_start || !region_end || n_regions < 1 || chunk_size < 1 || !P || n_shares < 1 || !own_lo || !own_hi) {
        set_err(err, errlen, "bad arguments to plan_shares"); return WGBSSEG_E_ARG;
    }
    if (P->max_bp == 0 || P->max_cpg < 1) { set_err(err, errlen, "max_bp and max_cpg must be >= 1"); return WGBSSEG_E_ARG; }
    const int G = n_shares;
    struct Ck { int64_t lo, hi, w; };
    std::vector<Ck> cks;
    for (int64_t r = 0; r < n_regions; r++) {
        const int64_t a = region_start[r], b = region_end[r];
        if (a < 1 || b <= a || b - 1 > n_sites) { set_err(err, errlen, "region %lld = [%lld, %lld) is empty or outside the %lld sites", (long long)r, (long long)a, (long long)b, (long long)n_sites); return WGBSSEG_E_ARG; }
        if (r && a < region_end[r - 1]) { set_err(err, errlen, "plan_shares: regions must be ascending and disjoint"); return WGBSSEG_E_ARG; }
        for (int64_t s0 = a; s0 < b; s0 += chunk_size) cks.push_back({s0 - 1, std::min(s0 + chunk_size, b) - 1, 0});
    }
    // share d's target: weights[d] / sum(weights) of the work (NULL: equal shares)
    std::vector<double> upto((size_t)G);
    {
        double sum = 0;
        for (int d = 0; d < G; d++) {
            const double w = weights ? weights[d] : 1.0;
            if (!(w >= 0.0)) { set_err(err, errlen, "plan_shares: weights must be >= 0"); return WGBSSEG_E_ARG; }
            sum += w; upto[(size_t)d] = sum;
        }
        if (!(sum > 0.0)) { set_err(err, errlen, "plan_shares: all weights are zero"); return WGBSSEG_E_ARG; }
        for (auto& u : upto) u /= sum;
    }
    if (G == 1) {
        for (auto& c : cks) c.w = c.hi - c.lo;                  // nothing to balance: do not walk the loci
    } else {
        parallel_for((int64_t)cks.size(), 32, [&](int64_t i) {
            Ck& c = cks[(size_t)i];
            c.w = chunk_work(loci, c.lo, c.hi, P->max_cpg, P->max_bp) + 4 * (c.hi - c.lo);     // + the per-site passes (scan, windows, recurrence)
        });
    }
    int64_t total = 0;
    for (auto& c : cks) total += c.w;
    if (halo < 0) halo = std::max<int64_t>(chunk_size, 4096);
    std::vector<int64_t> nch((size_t)G, 0), wk((size_t)G, 0);
    std::vector<char> any((size_t)G, 0);
    {   // contiguous runs of chunks: share d ends where the cumulative work passes (d+1)/G of the total
        int d = 0;
        int64_t acc = 0;
        for (auto& c : cks) {
            while (d < G - 1 && (double)acc >= (double)total * (weights ? upto[(size_t)d] : (double)(d + 1) / G)) d++;
            if (!any[(size_t)d]) { own_lo[d] = c.lo; any[(size_t)d] = 1; }
            own_hi[d] = c.hi;
            nch[(size_t)d]++; wk[(size_t)d] += c.w;
            acc += c.w;
        }
    }
    for (int q = 0; q < G; q++) {
        if (!any[(size_t)q]) own_lo[q] = own_hi[q] = q ? own_hi[q - 1] : cks.front().lo;
        // window: owned sites +- halo, the lower edge on a multiple of 128 sites (views into one device buffer stay 256-byte aligned)
        const bool has = own_hi[q] > own_lo[q];
        if (win_lo) win_lo[q] = has ? (std::max<int64_t>(0, own_lo[q] - halo) & ~127LL) : 0;
        if (win_hi) win_hi[q] = has ? std::min<int64_t>(n_sites, own_hi[q] + halo) : 0;
        if (share_chunks) share_chunks[q] = nch[(size_t)q];
        if (share_work) share_work[q] = wk[(size_t)q];
    }
    return WGBSSEG_OK;
}

int wgbsseg_group_plan(wgbsseg_group* g, const uint32_t* loci, int64_t n_sites, const int64_t* region_start, const int64_t* region_end,
                       int64_t n_regions, int64_t chunk_size, const wgbsseg_params* P, int64_t halo, int64_t* win_lo, int64_t* win_hi,
                       int64_t* share_chunks, int64_t* share_work, char* err, size_t errlen)
{
    if (!g) { set_err(err, errlen, "group is NULL"); return WGBSSEG_E_ARG; }
    const int G = (int)g->shares.size();
    g->planned = false;
    std::fill(g->loaded.begin(), g->loaded.end(), 0);
    g->own_lo.assign((size_t)G, 0); g->own_hi.assign((size_t)G, 0);
    g->win_lo.assign((size_t)G, 0); g->win_hi.assign((size_t)G, 0);
    g->share_chunks.assign((size_t)G, 0); g->share_work.assign((size_t)G, 0);
    if (halo < 0) halo = std::max<int64_t>(chunk_size, 4096);
    int rc = wgbsseg_plan_shares(loci, n_sites, region_start, region_end, n_regions, chunk_size, P, G, halo, g->own_lo.data(), g->own_hi.data(),
                                 g->win_lo.data(), g->win_hi.data(), g->share_chunks.data(), g->share_work.data(), err, errlen);
    if (rc != WGBSSEG_OK) return rc;
    g->rs.assign(region_start, region_start + n_regions);
    g->re.assign(region_end, region_end + n_regions);
    g->chunk_size = chunk_size; g->n_sites = n_sites; g->halo = halo; g->P = *P;
    // every share's window of the loci goes to its device now (4 bytes per site)
    std::vector<int> rcs((size_t)G, WGBSSEG_OK);
    std::vector<std::string> msgs((size_t)G);
    parallel_for(G, 64, [&](int64_t d) {
        if (g->win_hi[(size_t)d] <= g->win_lo[(size_t)d]) return;
        char eb[512] = {0};
        rcs[(size_t)d] = wgbsseg_set_loci_host(g->shares[(size_t)d], loci + g->win_lo[(size_t)d], g->win_hi[(size_t)d] - g->win_lo[(size_t)d], eb, sizeof(eb));
        g->shares[(size_t)d]->site_base = g->win_lo[(size_t)d];
        msgs[(size_t)d] = eb;
    });
    for (int d = 0; d < G; d++) if (rcs[(size_t)d] != WGBSSEG_OK) { set_err(err, errlen, "share %d: %s", d, msgs[(size_t)d].c_str()); return rcs[(size_t)d]; }
    for (int d = 0; d < G; d++) {
        if (win_lo) win_lo[d] = g->win_lo[(size_t)d];
        if (win_hi) win_hi[d] = g->win_hi[(size_t)d];
        if (share_chunks) share_chunks[d] = g->share_chunks[(size_t)d];
        if (share_work) share_work[d] = g->share_work[(size_t)d];
    }
    g->planned = true;
    return WGBSSEG_OK;
}

int wgbsseg_group_load_host(wgbsseg_group* g, const uint8_t* const* samples, int64_t n_samples, int64_t n_sites, char* err, size_t errlen)
{
    if (!g || !g->planned) { set_err(err, errlen, "group_load_host: call wgbsseg_group_plan first"); return WGBSSEG_E_STATE; }
    if (!samples || n_samples < 1 || n_sites != g->n_sites) { set_err(err, errlen, "group_load_host: bad arguments (the plan is for %lld sites)", (long long)g->n_sites); return WGBSSEG_E_ARG; }
    const int G = (int)g->shares.size();
    std::vector<int> rcs((size_t)G, WGBSSEG_OK);
    std::vector<std::string> msgs((size_t)G);
    parallel_for(G, 64, [&](int64_t d) {
        const int64_t lo = g->win_lo[(size_t)d], hi = g->win_hi[(size_t)d];
        if (hi <= lo) return;
        std::vector<const uint8_t*> ptrs((size_t)n_samples);
        for (int64_t s = 0; s < n_samples; s++) ptrs[(size_t)s] = samples[s] + 2 * lo;
        char eb[512] = {0};
        rcs[(size_t)d] = wgbsseg_set_betas_host(g->shares[(size_t)d], ptrs.data(), n_samples, hi - lo, eb, sizeof(eb));
        msgs[(size_t)d] = eb;
        if (rcs[(size_t)d] == WGBSSEG_OK) g->loaded[(size_t)d] = 1;
    });
    for (int d = 0; d < G; d++) if (rcs[(size_t)d] != WGBSSEG_OK) { set_err(err, errlen, "share %d: %s", d, msgs[(size_t)d].c_str()); return rcs[(size_t)d]; }
    return WGBSSEG_OK;
}

void group_join_loaders(wgbsseg_group* g)
{
    for (auto& l : g->loaders) if (l && l->th.joinable()) l->th.join();
}

int wgbsseg_group_load_host_async(wgbsseg_group* g, const uint8_t* const* samples, int64_t n_samples, int64_t n_sites, char* err, size_t errlen)
{
    if (!g || !g->planned) { set_err(err, errlen, "group_load_host_async: call wgbsseg_group_plan first"); return WGBSSEG_E_STATE; }
    if (!samples || n_samples < 1 || n_sites != g->n_sites) { set_err(err, errlen, "group_load_host_async: bad arguments (the plan is for %lld sites)", (long long)g->n_sites); return WGBSSEG_E_ARG; }
    for (int64_t s = 0; s < n_samples; s++) if (!samples[s]) { set_err(err, errlen, "samples[%lld] is NULL", (long long)s); return WGBSSEG_E_ARG; }
    group_join_loaders(g);
    const int G = (int)g->shares.size();
    g->loaders.clear();
    g->loaders.resize((size_t)G);
    // the device rows exist (and the contexts point at them) before any byte moves; the uploaders then fill them front to back.
    // A failure here leaves no share half-way: nothing counts as loaded, no loader runs, the group is not streaming.
    auto fail = [&](hipError_t e, const char* what) {
        for (int q = 0; q < G; q++) { g->loaded[(size_t)q] = 0; g->shares[(size_t)q]->last_valid = false; }
        g->loaders.clear();
        g->streaming = false;
        set_err(err, errlen, "group_load_host_async: %s: %s", what, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? WGBSSEG_E_NOMEM : WGBSSEG_E_HIP;
    };
    for (int d = 0; d < G; d++) {
        const int64_t lo = g->win_lo[(size_t)d], hi = g->win_hi[(size_t)d];
        if (hi <= lo) continue;
        wgbsseg_ctx* c = g->shares[(size_t)d];
        hipError_t e = hipSetDevice(c->device);
        if (e != hipSuccess) return fail(e, "hipSetDevice");
        const int64_t n = hi - lo, pitch = round_up(2 * n, 256) + 256;
        e = c->betas_own.ensure((size_t)pitch * (size_t)n_samples);
        if (e != hipSuccess) return fail(e, "device rows of a share");
        c->betas = c->betas_own.as<uint8_t>();
        c->pitch = pitch; c->n_total = n; c->n_samples = (int32_t)n_samples; c->elem = 1;
        c->last_valid = false;
    }
    std::vector<const uint8_t*> base(samples, samples + n_samples);
    for (int d = 0; d < G; d++) {
        const int64_t lo = g->win_lo[(size_t)d], hi = g->win_hi[(size_t)d];
        if (hi <= lo) continue;
        g->loaders[(size_t)d].reset(new wgbsseg_group::Loader());
        wgbsseg_group::Loader* L = g->loaders[(size_t)d].get();
        wgbsseg_ctx* c = g->shares[(size_t)d];
        L->th = std::thread([L, c, base, lo, hi, n_samples]() {
            std::vector<const uint8_t*> ptrs((size_t)n_samples);
            for (int64_t s = 0; s < n_samples; s++) ptrs[(size_t)s] = base[(size_t)s] + 2 * lo;
            L->rc = upload_rows_streaming(c, c->betas_own.as<uint8_t>(), c->pitch, ptrs.data(), n_samples, 2 * (hi - lo), L->ready, L->msg);
            L->ready.store(L->rc == WGBSSEG_OK ? hi - lo : L->ready.load());
            L->finished.store(1);
        });
        g->loaded[(size_t)d] = 1;
    }
    g->streaming = true;
    return WGBSSEG_OK;
}

int wgbsseg_group_load_wait(wgbsseg_group* g, char* err, size_t errlen)
{
    if (!g) { set_err(err, errlen, "group is NULL"); return WGBSSEG_E_ARG; }
    group_join_loaders(g);
    g->streaming = false;
    for (size_t d = 0; d < g->loaders.size(); d++)
        if (g->loaders[d] && g->loaders[d]->rc != WGBSSEG_OK) { set_err(err, errlen, "share %d: %s", (int)d, g->loaders[d]->msg.c_str()); return g->loaders[d]->rc; }
    return WGBSSEG_OK;
}

int wgbsseg_group_share_set_device(wgbsseg_group* g, int32_t share, const void* base, int64_t n_samples, int64_t pitch_bytes, char* err, size_t errlen)
{
    if (!g || !g->planned) { set_err(err, errlen, "group_share_set_device: call wgbsseg_group_plan first"); return WGBSSEG_E_STATE; }
    if (share < 0 || share >= (int32_t)g->shares.size()) { set_err(err, errlen, "no share %d", (int)share); return WGBSSEG_E_ARG; }
    const int64_t n = g->win_hi[(size_t)share] - g->win_lo[(size_t)share];
    if (n <= 0) return WGBSSEG_OK;                          // a share without chunks needs no data
    const int rc = wgbsseg_set_betas_device(g->shares[(size_t)share], base, n_samples, pitch_bytes, n, err, errlen);
    if (rc == WGBSSEG_OK) g->loaded[(size_t)share] = 1;
    return rc;
}

int wgbsseg_group_segment_regions(wgbsseg_group* g, int32_t* borders_out, int64_t borders_cap, int64_t* borders_off, int64_t* stats,
                                  char* err, size_t errlen)
{
    if (!g || !g->planned) { set_err(err, errlen, "group_segment_regions: call wgbsseg_group_plan first"); return WGBSSEG_E_STATE; }
    return wgbsseg_group_segment_region_range(g, 0, (int64_t)g->rs.size(), borders_out, borders_cap, borders_off, stats, err, errlen);
}

int wgbsseg_group_segment_region_range(wgbsseg_group* g, int64_t first_region, int64_t end_region, int32_t* borders_out, int64_t borders_cap, int64_t* borders_off,
                                       int64_t* stats, char* err, size_t errlen)
{
    if (!g || !g->planned) { set_err(err, errlen, "group_segment_regions: call wgbsseg_group_plan first"); return WGBSSEG_E_STATE; }
    if (first_region < 0 || end_region <= first_region || end_region > (int64_t)g->rs.size()) { set_err(err, errlen, "group_segment_region_range: regions [%lld, %lld) of %lld planned", (long long)first_region, (long long)end_region, (long long)g->rs.size()); return WGBSSEG_E_ARG; }
    const int G = (int)g->shares.size();
    for (int d = 0; d < G; d++)
        if (g->win_hi[(size_t)d] > g->win_lo[(size_t)d] && !g->loaded[(size_t)d]) { set_err(err, errlen, "share %d has no beta data yet", d); return WGBSSEG_E_STATE; }
    int64_t n_batches = 0;
    std::vector<char> ran((size_t)G, first_region > 0 ? 1 : 0);     // the share's timings: reset on its first batch of the FIRST slice of the regions, summed after
    std::vector<int64_t> slot_next((size_t)G, 0);            // page-locked result buffers of a share used by this call so far
    wgstitch::BatchFn run_batch = [&](const std::vector<wgstitch::Sites>& todo, wgstitch::BatchResult& res, std::string& msg) -> int {
        // route: the share that owns the first site of the range; a junction patch reaches into the next share's first chunk,
        // which the halo of the window covers
        std::vector<std::vector<size_t>> items((size_t)G);
        for (size_t i = 0; i < todo.size(); i++) {
            const int64_t lo = todo[i].first - 1, hi = todo[i].second - 1;
            if (hi - lo > 0x7fffffff) { msg = "chunk too long"; return WGBSSEG_E_ARG; }
            int d = (int)(std::upper_bound(g->own_lo.begin(), g->own_lo.end(), lo) - g->own_lo.begin()) - 1;
            d = std::max(d, 0);
            while (d > 0 && g->own_hi[(size_t)d] <= g->own_lo[(size_t)d]) d--;              // shares without chunks own nothing
            int pick = -1;
            for (int q : {d, d + 1, d - 1})
                if (q >= 0 && q < G && g->win_lo[(size_t)q] <= lo && hi <= g->win_hi[(size_t)q] && g->win_hi[(size_t)q] > g->win_lo[(size_t)q]) { pick = q; break; }
            if (pick < 0) {
                msg = "sites [" + std::to_string(lo + 1) + ", " + std::to_string(hi + 1) + ") are not resident on any single share (halo " +
                      std::to_string(g->halo) + " sites): a junction patch outgrew it; rerun on one share";
                return WGBSSEG_E_STATE;
            }
            items[(size_t)pick].push_back(i);
        }
        res.ptr.assign(todo.size(), nullptr);
        res.cnt.assign(todo.size(), 0);
        std::vector<int> rcs((size_t)G, WGBSSEG_OK);
        std::vector<std::string> msgs((size_t)G);
        std::vector<std::vector<std::unique_ptr<int32_t[]>>> owned((size_t)G);
        auto run_items = [&](int d, const std::vector<size_t>& it) -> bool {
            std::vector<int64_t> st0(it.size()), off;
            std::vector<int32_t> ln(it.size());
            for (size_t k = 0; k < it.size(); k++) {
                st0[k] = todo[it[k]].first - 1 - g->win_lo[(size_t)d];
                ln[k] = (int32_t)(todo[it[k]].second - todo[it[k]].first);
            }
            const int32_t* flat = nullptr;
            std::unique_ptr<int32_t[]> own;
            rcs[(size_t)d] = run_ctx_batch(g->shares[(size_t)d], st0, ln, &g->P, slot_next[(size_t)d]++, ran[(size_t)d] != 0, flat, off, own, msgs[(size_t)d]);
            if (rcs[(size_t)d] != WGBSSEG_OK) return false;
            if (own) owned[(size_t)d].push_back(std::move(own));
            ran[(size_t)d] = 1;
            for (size_t k = 0; k < it.size(); k++) { res.ptr[it[k]] = flat + off[k]; res.cnt[it[k]] = off[k + 1] - off[k]; }
            return true;
        };
        auto work = [&](int d) {
            const std::vector<size_t>& it = items[(size_t)d];
            if (it.empty()) return;
            wgbsseg_group::Loader* L = (g->streaming && (size_t)d < g->loaders.size()) ? g->loaders[(size_t)d].get() : nullptr;
            if (!L || L->finished.load()) {
                if (L && L->rc != WGBSSEG_OK) { rcs[(size_t)d] = L->rc; msgs[(size_t)d] = L->msg; return; }
                run_items(d, it);
                return;
            }
            // the share's bytes are still arriving (front to back): segment what is resident while the rest is on its way —
            // items in order of their last site, a sub-batch whenever a fair part of the share has landed
            std::vector<size_t> order(it);
            std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return todo[a].second < todo[b].second; });
            const int64_t wlo = g->win_lo[(size_t)d];
            const int64_t min_take = std::max<int64_t>(4 * g->chunk_size, (g->win_hi[(size_t)d] - wlo) / 5);
            size_t pos = 0;
            while (pos < order.size()) {
                const int64_t first_end = todo[order[pos]].second - 1 - wlo;          // resident sites the next item needs
                int64_t r = 0;
                for (;;) {
                    const bool fin = L->finished.load() != 0;
                    r = L->ready.load();
                    if (fin && L->rc != WGBSSEG_OK) { rcs[(size_t)d] = L->rc; msgs[(size_t)d] = L->msg; return; }
                    if (fin) { r = g->win_hi[(size_t)d] - wlo; break; }
                    if (r >= first_end && r - (todo[order[pos]].first - 1 - wlo) >= min_take) break;
                    std::this_thread::sleep_for(std::chrono::microseconds(200));
                }
                std::vector<size_t> take;
                while (pos < order.size() && todo[order[pos]].second - 1 - wlo <= r) take.push_back(order[pos++]);
                if (!run_items(d, take)) return;
            }
        };
        int busy = 0, only = -1;
        for (int d = 0; d < G; d++) if (!items[(size_t)d].empty()) { busy++; only = d; }
        if (busy == 1) work(only);
        else {
            std::vector<std::thread> th;
            for (int d = 0; d < G; d++) if (!items[(size_t)d].empty()) th.emplace_back(work, d);
            for (auto& x : th) x.join();
        }
        for (int d = 0; d < G; d++) {
            if (rcs[(size_t)d] != WGBSSEG_OK) { msg = "share " + std::to_string(d) + ": " + msgs[(size_t)d]; return rcs[(size_t)d]; }
            for (auto& o : owned[(size_t)d]) res.owned.push_back(std::move(o));
        }
        n_batches++;
        return WGBSSEG_OK;
    };
    std::string msg;
    const int rc = wgstitch::segment_regions(g->rs.data() + first_region, g->re.data() + first_region, end_region - first_region, g->chunk_size, run_batch, borders_out,
                                             borders_cap, borders_off, stats, msg, speculation_on());
    if (g->streaming && (end_region == (int64_t)g->rs.size() || rc != 0)) {      // every byte has been consumed by now; collect the uploaders
        char eb[512] = {0};
        const int lrc = wgbsseg_group_load_wait(g, eb, sizeof(eb));
        if (rc == 0 && lrc != WGBSSEG_OK) { set_err(err, errlen, "%s", eb); return lrc; }
    }
    return map_stitch_rc(rc, msg, err, errlen);
}

int wgbsseg_group_get_timings(const wgbsseg_group* g, int32_t share, wgbsseg_timings* out)
{
    if (!g || share < 0 || share >= (int32_t)g->shares.size()) return WGBSSEG_E_ARG;
    return wgbsseg_get_timings(g->shares[(size_t)share], out);
}

int wgbsseg_scan_only(wgbsseg_ctx* c, const int64_t* chunk_start0, const int32_t* chunk_len, int64_t n_chunks, int repeat, int want_carry,
                      double* ms_per_launch, int64_t* bytes_per_launch, int64_t* carry_bytes_per_launch, char* err, size_t errlen)
{
    Job job;
    int rc = build_job(c, chunk_start0, chunk_len, n_chunks, job, false, err, errlen);
    if (rc != WGBSSEG_OK) return rc;
    rc = plan_validation(c, job, true, err, errlen);
    if (rc != WGBSSEG_OK) return rc;
    c->validated.clear();
    if (repeat < 1) repeat = 1;
    want_carry = want_carry ? 1 : 0;
    // want_carry 0: k_validate, the read-only pass of a job without wide tiles;
    // 1: k_scan — per-sample prefix sums of (meth, cov), a carry per 128 sites, the validation — as a job with wide tiles runs it
    auto one = [&]() { return want_carry ? launch_scan(c, job, c->sA, err, errlen) : launch_validate(c, job, c->sA, err, errlen); };
    rc = one();           // warm-up
    if (rc != WGBSSEG_OK) return rc;
    HIP_TRY(hipEventRecord(c->ev[0], c->sA));
    for (int r = 0; r < repeat; r++) { rc = one(); if (rc != WGBSSEG_OK) return rc; }
    HIP_TRY(hipEventRecord(c->ev[1], c->sA));
    HIP_TRY(hipStreamSynchronize(c->sA));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    if (ms_per_launch) *ms_per_launch = (double)ms / repeat;
    if (bytes_per_launch) *bytes_per_launch = 2 * (want_carry ? job.sites : job.val_sites) * c->n_samples;
    if (carry_bytes_per_launch) *carry_bytes_per_launch = want_carry ? (int64_t)sizeof(uint2) * job.carry_entries : 0;
    JobStatus st;
    HIP_TRY(hipMemcpyAsync(&st, c->status.p, sizeof(st), hipMemcpyDeviceToHost, c->sA));
    HIP_TRY(hipStreamSynchronize(c->sA));
    if (st.first_bad != ~0ULL) return report_bad_site(c, st, err, errlen);
    return WGBSSEG_OK;
}

int wgbsseg_prefix_sums(wgbsseg_ctx* c, int64_t start0, int64_t len, uint32_t* out, char* err, size_t errlen)
{
    if (!out || len < 1 || len > 0x7fffffff) { set_err(err, errlen, "bad arguments to prefix_sums"); return WGBSSEG_E_ARG; }
    const int32_t l32 = (int32_t)len;
    Job job;
    int rc = build_job(c, &start0, &l32, 1, job, false, err, errlen);
    if (rc != WGBSSEG_OK) return rc;
    rc = launch_scan(c, job, c->sA, err, errlen);           // the carries are what k_prefix_materialise builds on
    if (rc != WGBSSEG_OK) return rc;
    const size_t bytes = (size_t)c->n_samples * (size_t)(len + 1) * 8;
    HIP_TRY(c->dbg_a.ensure(bytes));
    const int nG = job.h[0].nG;
    hipLaunchKernelGGL(k_prefix_materialise, dim3((unsigned)nG, (unsigned)c->n_samples), dim3(64), 0, c->sA, job.v, nG, c->dbg_a.as<uint32_t>(), l32);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, c->dbg_a.p, bytes, hipMemcpyDeviceToHost, c->sA));
    JobStatus st;
    HIP_TRY(hipMemcpyAsync(&st, c->status.p, sizeof(st), hipMemcpyDeviceToHost, c->sA));
    HIP_TRY(hipStreamSynchronize(c->sA));
    if (st.first_bad != ~0ULL) return report_bad_site(c, st, err, errlen);
    c->last_valid = false;
    return WGBSSEG_OK;
}

int wgbsseg_block_sums(wgbsseg_ctx* c, const int64_t* start0, const int64_t* end0, int64_t n_blocks, int32_t mode,
                       uint32_t min_cov, void* out, char* err, size_t errlen)
{
    if (!c) { set_err(err, errlen, "ctx is NULL"); return WGBSSEG_E_ARG; }
    if (!c->betas) { set_err(err, errlen, "betas not set"); return WGBSSEG_E_STATE; }
    if (n_blocks < 0 || mode < 0 || mode > 3 || (n_blocks && (!start0 || !end0 || !out))) { set_err(err, errlen, "bad arguments to block_sums"); return WGBSSEG_E_ARG; }
    if (n_blocks == 0) return WGBSSEG_OK;
    if (n_blocks > 0x7fffffff || c->n_total > 0x7fffffff) { set_err(err, errlen, "too many blocks / sites for one block_sums call"); return WGBSSEG_E_ARG; }
    bool sorted = true;
    for (int64_t i = 0; i < n_blocks; i++) {
        if (start0[i] < 0 || end0[i] < start0[i] || end0[i] > c->n_total) {
            set_err(err, errlen, "block %lld = sites [%lld, %lld) is outside the %lld sites of the beta files or reversed",
                    (long long)i, (long long)start0[i], (long long)end0[i], (long long)c->n_total);
            return WGBSSEG_E_ARG;
        }
        if (mode == 0 && c->elem == 2 && end0[i] - start0[i] > 65536) { set_err(err, errlen, "block %lld: uint32 sums of uint16 counts are only exact up to 65536 sites per block", (long long)i); return WGBSSEG_E_ARG; }
        if (i && start0[i] < start0[i - 1]) sorted = false;
    }
    HIP_TRY(hipSetDevice(c->device));
    // the kernel wants the blocks in order of their first site (a table a segmentation wrote already is): sort a copy
    const int64_t n_tiles = (c->n_total + WG_BS_TILE - 1) / WG_BS_TILE;
    std::vector<int32_t> h((size_t)n_blocks * (sorted ? 2 : 3) + (size_t)n_tiles + 1);
    int32_t* hx0 = h.data();
    int32_t* hx1 = hx0 + n_blocks;
    int32_t* hperm = sorted ? nullptr : hx1 + n_blocks;
    int32_t* htf = hx1 + n_blocks + (sorted ? 0 : n_blocks);
    if (sorted) {
        for (int64_t i = 0; i < n_blocks; i++) { hx0[i] = (int32_t)start0[i]; hx1[i] = (int32_t)end0[i]; }
    } else {
        for (int64_t i = 0; i < n_blocks; i++) hperm[i] = (int32_t)i;
        std::stable_sort(hperm, hperm + n_blocks, [&](int32_t a, int32_t b) { return start0[a] < start0[b]; });
        for (int64_t i = 0; i < n_blocks; i++) { hx0[i] = (int32_t)start0[hperm[i]]; hx1[i] = (int32_t)end0[hperm[i]]; }
    }
    {   // first block that starts at or after every tile's first site (empty blocks ride along with their start site)
        int64_t b = 0;
        for (int64_t t = 0; t <= n_tiles; t++) {
            const int64_t lo = t * WG_BS_TILE;
            while (b < n_blocks && hx0[b] < lo) b++;
            htf[t] = (int32_t)b;
        }
        htf[n_tiles] = (int32_t)n_blocks;                          // blocks that start at n_total (empty) belong to the last tile
    }
    const size_t esz = mode == 0 ? 8 : (mode == 1 ? 2 : (mode == 2 ? 4 : 8));
    const size_t obytes = (size_t)c->n_samples * (size_t)n_blocks * esz;
    // uint8 rows and a table ordered by first AND last site (what a segmentation writes; beta_to_blocks' "nice" tables): the
    // streaming kernel.  Its tile table: the first block whose last site lies at or behind every 1024-site tile's first site.
    std::vector<int32_t> direct;
    bool monotone = c->elem == 1 && (uint64_t)n_blocks * 8 < (1ull << 32);      // (32-bit output offsets in the streaming kernel)
    for (int64_t i = 1; i < n_blocks && monotone; i++) monotone = hx1[i] >= hx1[i - 1];
    if (c->bs_general) monotone = false;                         // WGBSSEG_BLOCK_SUMS_GENERAL=1 (tests): the general kernel for every table
    const int64_t n_rtiles = (c->n_total + WG_BSR_TILE - 1) / WG_BSR_TILE;
    if (monotone) {
        h.resize((size_t)n_blocks * (sorted ? 2 : 3) + (size_t)n_rtiles + 1);
        hx0 = h.data(); hx1 = hx0 + n_blocks; hperm = sorted ? nullptr : hx1 + n_blocks;
        htf = hx1 + n_blocks + (sorted ? 0 : n_blocks);
        int64_t b = 0;
        for (int64_t t = 0; t <= n_rtiles; t++) {                  // tile of a block: the one holding its last site (empty blocks: their position)
            while (b < n_blocks && (hx1[b] == 0 ? 0 : ((int64_t)std::max(hx1[b] - 1, hx0[b])) / WG_BSR_TILE) < t) b++;
            htf[t] = (int32_t)b;
        }
        htf[n_rtiles] = (int32_t)n_blocks;
        // blocks the streaming kernel cannot resolve from its two-tile ring (they begin before their run, or more than a tile
        // before the tile they end in): a matter of the table alone; they get a wavefront per (block, sample) afterwards
        for (int64_t i = 0; i < n_blocks; i++) {
            if (hx1[i] <= hx0[i]) continue;
            const int64_t t = (hx1[i] - 1) / WG_BSR_TILE, lo = t * WG_BSR_TILE;                 // (k_block_sums_prep applies the same rule)
            if (t % WG_BSR_RUN == 0 ? hx0[i] < lo : hx0[i] < lo - (WG_BSR_TILE - 1)) direct.push_back((int32_t)i);
        }
        h.insert(h.end(), direct.begin(), direct.end());
        hx0 = h.data(); hx1 = hx0 + n_blocks; hperm = sorted ? nullptr : hx1 + n_blocks;
        htf = hx1 + n_blocks + (sorted ? 0 : n_blocks);
    }
    HIP_TRY(c->dbg_a.ensure(h.size() * 4));
    HIP_TRY(c->dbg_b.ensure(obytes));
    HIP_TRY(hipMemcpyAsync(c->dbg_a.p, h.data(), h.size() * 4, hipMemcpyHostToDevice, c->sA));
    const int32_t* dx0 = c->dbg_a.as<int32_t>();
    const int32_t* dx1 = dx0 + n_blocks;
    const int32_t* dperm = sorted ? nullptr : dx1 + n_blocks;
    const int32_t* dtf = dx1 + n_blocks + (sorted ? 0 : n_blocks);
    // samples per wavefront: enough workgroups to fill the chip, few enough that the block list is re-read rarely
    // samples per wavefront (the kernel keeps two tiles of each in registers, one being reduced, one in flight)
    const int spw = (c->elem == 1 && c->n_samples > 4) ? 2 : 1;
    const int64_t gx = (n_tiles + WG_BS_RUN - 1) / WG_BS_RUN;
    const unsigned gy = (unsigned)((c->n_samples + 4 * spw - 1) / (4 * spw));
    if (gy > 65535) { set_err(err, errlen, "too many samples for one block_sums call"); return WGBSSEG_E_ARG; }
    HIP_TRY(hipEventRecord(c->ev[0], c->sA));
    if (monotone) {
        HIP_TRY(c->bs_desc.ensure((size_t)n_blocks * 12));
        int32_t* dd1 = c->bs_desc.as<int32_t>();
        int32_t* dd0 = dd1 + n_blocks;
        int32_t* drr = dd0 + n_blocks;
        hipLaunchKernelGGL(k_block_sums_prep, dim3((unsigned)((n_blocks + WG_BLOCK - 1) / WG_BLOCK)), dim3(WG_BLOCK), 0, c->sA, dx0, dx1, dperm, n_blocks, n_rtiles, dd1, dd0, drr);
        HIP_TRY(hipGetLastError());
        const dim3 grid((unsigned)((n_rtiles + WG_BSR_RUN - 1) / WG_BSR_RUN), (unsigned)((c->n_samples + 3) / 4));
#define WG_LAUNCH_BSR(M) hipLaunchKernelGGL(k_block_sums_run<M>, grid, dim3(WG_BLOCK), 0, c->sA, c->betas, c->pitch, c->n_total, dd1, dd0, drr, dtf, \
                                            n_rtiles, n_blocks, (int)c->n_samples, min_cov, c->dbg_b.p)
        if (mode == 0) WG_LAUNCH_BSR(0); else if (mode == 1) WG_LAUNCH_BSR(1); else if (mode == 2) WG_LAUNCH_BSR(2); else WG_LAUNCH_BSR(3);
#undef WG_LAUNCH_BSR
        if (!direct.empty()) {
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(k_block_sums_direct, dim3((unsigned)direct.size(), (unsigned)((c->n_samples + 3) / 4)), dim3(WG_BLOCK), 0, c->sA,
                               c->betas, c->pitch, c->n_total, dx0, dx1, dperm, dtf + n_rtiles + 1, (int64_t)direct.size(), n_blocks,
                               (int)c->n_samples, (int)mode, min_cov, c->dbg_b.p);
        }
    } else if (c->elem == 1)
        hipLaunchKernelGGL(k_block_sums<1>, dim3((unsigned)gx, gy), dim3(WG_BLOCK), 0, c->sA, c->betas, c->pitch, c->n_total,
                           dx0, dx1, dperm, dtf, n_tiles, n_blocks, (int)c->n_samples, spw, (int)mode, min_cov, c->dbg_b.p);
    else
        hipLaunchKernelGGL(k_block_sums<2>, dim3((unsigned)gx, gy), dim3(WG_BLOCK), 0, c->sA, c->betas, c->pitch, c->n_total,
                           dx0, dx1, dperm, dtf, n_tiles, n_blocks, (int)c->n_samples, spw, (int)mode, min_cov, c->dbg_b.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev[1], c->sA));
    HIP_TRY(hipMemcpyAsync(out, c->dbg_b.p, obytes, hipMemcpyDeviceToHost, c->sA));
    HIP_TRY(hipStreamSynchronize(c->sA));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->last_block_sums_ms = ms;
    c->last_valid = false;
    c->table_blocks = mode == 3 ? n_blocks : 0;
    return WGBSSEG_OK;
}

int wgbsseg_marker_stats(wgbsseg_ctx* c, const int32_t* tg, int32_t n_tg, const int32_t* bg, int32_t n_bg, int64_t n_blocks, double* out,
                         char* err, size_t errlen)
{
    if (!c || !tg || !bg || n_tg < 1 || n_bg < 1 || !out || n_blocks < 1) { set_err(err, errlen, "bad arguments to marker_stats"); return WGBSSEG_E_ARG; }
    if (c->table_blocks != n_blocks) { set_err(err, errlen, "marker_stats: the last wgbsseg_block_sums call was not a mode-3 reduction over these %lld blocks", (long long)n_blocks); return WGBSSEG_E_STATE; }
    for (int i = 0; i < n_tg; i++) if (tg[i] < 0 || tg[i] >= c->n_samples) { set_err(err, errlen, "marker_stats: no sample %d", (int)tg[i]); return WGBSSEG_E_ARG; }
    for (int i = 0; i < n_bg; i++) if (bg[i] < 0 || bg[i] >= c->n_samples) { set_err(err, errlen, "marker_stats: no sample %d", (int)bg[i]); return WGBSSEG_E_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->dbg_c.ensure((size_t)n_blocks * 64 + (size_t)(n_tg + n_bg) * 4));
    double* dout = c->dbg_c.as<double>();
    int32_t* dtg = reinterpret_cast<int32_t*>(dout + n_blocks * 8);
    int32_t* dbg = dtg + n_tg;
    HIP_TRY(hipMemcpyAsync(dtg, tg, (size_t)n_tg * 4, hipMemcpyHostToDevice, c->sA));
    HIP_TRY(hipMemcpyAsync(dbg, bg, (size_t)n_bg * 4, hipMemcpyHostToDevice, c->sA));
    HIP_TRY(hipEventRecord(c->ev[0], c->sA));
    hipLaunchKernelGGL(k_marker_stats, dim3((unsigned)(((n_blocks + 1) / 2 + WG_BLOCK - 1) / WG_BLOCK)), dim3(WG_BLOCK), 0, c->sA, c->dbg_b.as<double>(), n_blocks,      // (two blocks per thread)
                       dtg, (int)n_tg, dbg, (int)n_bg, dout);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev[1], c->sA));
    HIP_TRY(hipMemcpyAsync(out, dout, (size_t)n_blocks * 64, hipMemcpyDeviceToHost, c->sA));
    HIP_TRY(hipStreamSynchronize(c->sA));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->last_block_sums_ms = ms;
    return WGBSSEG_OK;
}

double wgbsseg_last_block_sums_ms(const wgbsseg_ctx* c) { return c ? c->last_block_sums_ms : 0.0; }

int wgbsseg_add_loci(const uint32_t* loci, int64_t n_sites, const int64_t* chrom_cum, const char* const* chrom_names,
                     int32_t n_chroms, const int64_t* start_cpg, const int64_t* end_cpg, int64_t n_blocks,
                     const char* path, int32_t append, int32_t threads, char* err, size_t errlen)
{
    if (!loci || !chrom_cum || !chrom_names || n_chroms < 1 || n_sites < 1 || n_blocks < 0 || (n_blocks && (!start_cpg || !end_cpg))) {
        set_err(err, errlen, "add_loci: bad argument"); return WGBSSEG_E_ARG;
    }
    if (chrom_cum[n_chroms - 1] != n_sites) { set_err(err, errlen, "add_loci: chromosome sizes sum to %lld, loci has %lld sites", (long long)chrom_cum[n_chroms - 1], (long long)n_sites); return WGBSSEG_E_ARG; }
    wgadd::Genome g = {loci, n_sites, chrom_cum, chrom_names, n_chroms};
    std::string msg;
    int rc = 0;
    if (path) {
        // a regular file: format on all cores, then write the shards side by side at their offsets
        const int fd = open(path, O_WRONLY | O_CREAT | (append ? 0 : O_TRUNC), 0666);
        if (fd < 0) { set_err(err, errlen, "add_loci: cannot open %s", path); return WGBSSEG_E_ARG; }
        const off_t base = append ? lseek(fd, 0, SEEK_END) : 0;
        rc = base < 0 ? 3 : wgadd::add_loci_fd(g, start_cpg, end_cpg, n_blocks, fd, (int64_t)base, threads, msg);
        if (close(fd) != 0 && rc == 0) { set_err(err, errlen, "add_loci: write to %s failed", path); return WGBSSEG_E_ARG; }
        if (rc == 3 && msg.empty()) msg = "write failed";
    } else {
        rc = wgadd::add_loci(g, start_cpg, end_cpg, n_blocks, stdout, threads, msg);
    }
    if (rc) { set_err(err, errlen, "%s", msg.c_str()); return WGBSSEG_E_ARG; }
    return WGBSSEG_OK;
}

int wgbsseg_add_loci_borders(const uint32_t* loci, int64_t n_sites, const int64_t* chrom_cum, const char* const* chrom_names, int32_t n_chroms,
                             const int32_t* borders, const int64_t* borders_off, int64_t n_regions, int64_t min_cpg,
                             const char* path, int32_t append, int32_t threads, int64_t* n_written, int64_t* n_dropped, char* err, size_t errlen)
{
    if (n_written) *n_written = 0;
    if (n_dropped) *n_dropped = 0;
    if (!loci || !chrom_cum || !chrom_names || n_chroms < 1 || n_sites < 1 || n_regions < 0 || (n_regions && (!borders || !borders_off))) {
        set_err(err, errlen, "add_loci_borders: bad argument"); return WGBSSEG_E_ARG;
    }
    if (chrom_cum[n_chroms - 1] != n_sites) { set_err(err, errlen, "add_loci: chromosome sizes sum to %lld, loci has %lld sites", (long long)chrom_cum[n_chroms - 1], (long long)n_sites); return WGBSSEG_E_ARG; }
    // the rows must come out sorted by startCpG (segment.py:169 sorts them): region lists in ascending order, each ascending
    {
        int64_t last_r = -1;                                      // the last region that held a border, and that border
        int32_t last_b = 0;
        for (int64_t r = 0; r < n_regions; r++) {
            if (borders_off[r + 1] < borders_off[r] || (r == 0 && borders_off[0] < 0)) { set_err(err, errlen, "add_loci_borders: offsets not ascending"); return WGBSSEG_E_ARG; }
            if (borders_off[r + 1] == borders_off[r]) continue;
            if (last_r >= 0 && borders[borders_off[r]] < last_b) {
                set_err(err, errlen, "add_loci_borders: region %lld begins before region %lld ends (sort the blocks first)", (long long)r, (long long)last_r); return WGBSSEG_E_ARG;
            }
            // ... and ascending INSIDE the region: a descending pair would be counted as a dropped short block (b - a < min_cpg) where the
            // array form's check_row refuses "endCpG < startCpG" — refused here too, so that n_dropped counts short blocks only
            const int32_t* p = borders + borders_off[r];
            const int64_t nb = borders_off[r + 1] - borders_off[r];
            int32_t desc = 0;
            for (int64_t j = 0; j + 1 < nb; j++) desc |= (int32_t)(p[j + 1] < p[j]);
            if (desc) {
                int64_t j = 0;
                while (p[j + 1] >= p[j]) j++;
                set_err(err, errlen, "add_loci_borders: region %lld: border %lld (%d) follows %d (endCpG < startCpG; each region's borders must ascend)",
                        (long long)r, (long long)(j + 1), (int)p[j + 1], (int)p[j]);
                return WGBSSEG_E_ARG;
            }
            last_r = r; last_b = borders[borders_off[r + 1] - 1];
        }
    }
    wgadd::Genome g = {loci, n_sites, chrom_cum, chrom_names, n_chroms};
    wgadd::BorderRows rows{borders, borders_off, n_regions, min_cpg, {}};
    rows.index();
    const int64_t total = rows.total();
    std::string msg;
    int rc = 0;
    int64_t written = 0;
    if (path) {
        const int fd = open(path, O_WRONLY | O_CREAT | (append ? 0 : O_TRUNC), 0666);
        if (fd < 0) { set_err(err, errlen, "add_loci: cannot open %s", path); return WGBSSEG_E_ARG; }
        const off_t base = append ? lseek(fd, 0, SEEK_END) : 0;
        rc = base < 0 ? 3 : wgadd::add_loci_fd_rows(g, rows, total, fd, (int64_t)base, threads, msg, &written);
        if (close(fd) != 0 && rc == 0) { set_err(err, errlen, "add_loci: write to %s failed", path); return WGBSSEG_E_ARG; }
        if (rc == 3 && msg.empty()) msg = "write failed";
    } else {
        rc = wgadd::add_loci_rows(g, rows, total, stdout, threads, msg, &written);
    }
    if (n_written) *n_written = written;
    if (n_dropped) *n_dropped = rc ? 0 : total - written;
    if (rc) { set_err(err, errlen, "%s", msg.c_str()); return WGBSSEG_E_ARG; }
    return WGBSSEG_OK;
}

int wgbsseg_blocks_parse(const char* text, int64_t len, int64_t max_rows, int64_t cap, int64_t* line_off, int32_t* len3,
                         int64_t* start_cpg, int64_t* end_cpg, uint8_t* na, int64_t* n_rows,
                         int64_t* bp_start, int64_t* bp_end, int32_t* bp_ok, int32_t* first_fields)
{
    if (!n_rows) return WGBSSEG_E_ARG;
    *n_rows = 0;
    if (!text || len < 0 || cap < 0 || (cap && (!line_off || !len3 || !start_cpg || !end_cpg || !na))) return WGBSSEG_E_ARG;
    const int rc = wgtab::parse_blocks(text, len, max_rows, cap, line_off, len3, start_cpg, end_cpg, na, n_rows, bp_start, bp_end, bp_ok, first_fields);
    return rc == 0 ? WGBSSEG_OK : (rc == 1 ? 1 : WGBSSEG_E_ARG);
}

namespace {
int open_for_rows(const char* path, int append, int64_t* base, char* err, size_t errlen)
{
    const int fd = open(path, O_WRONLY | O_CREAT | (append ? 0 : O_TRUNC), 0666);
    if (fd < 0) { set_err(err, errlen, "cannot open %s", path); return -1; }
    const off_t b = append ? lseek(fd, 0, SEEK_END) : 0;
    if (b < 0) { close(fd); set_err(err, errlen, "cannot seek in %s", path); return -1; }
    *base = (int64_t)b;
    return fd;
}
}  // namespace

int wgbsseg_blocks_write_table(const char* path, int32_t append, const char* text, const int64_t* line_off, const int32_t* len3,
                               const int64_t* start_cpg, const int64_t* end_cpg, const uint8_t* na, int64_t n_rows,
                               const double* values, int64_t n_cols, int64_t stride, int32_t digits, int32_t threads, char* err, size_t errlen)
{
    if (n_rows < 0 || n_cols < 0 || stride < n_cols || (n_rows && (!text || !line_off || !len3 || !start_cpg || !end_cpg || !na || (n_cols && !values)))) {
        set_err(err, errlen, "write_table: bad argument"); return WGBSSEG_E_ARG;
    }
    int64_t base = -1;                                   // path NULL: standard output, whatever it is — shards in order
    const int fd = path ? open_for_rows(path, append, &base, err, errlen) : 1;
    if (fd < 0) return WGBSSEG_E_ARG;
    const wgtab::Rows R = {text, line_off, len3, start_cpg, end_cpg, na};
    std::string msg;
    const int rc = wgtab::write_table(fd, base, R, n_rows, values, n_cols, stride, digits, threads, msg);
    if (path && close(fd) != 0 && rc == 0) { set_err(err, errlen, "write to %s failed", path); return WGBSSEG_E_ARG; }
    if (rc) { set_err(err, errlen, "%s", msg.c_str()); return WGBSSEG_E_ARG; }
    return WGBSSEG_OK;
}

int wgbsseg_blocks_write_bedgraph(const char* path, const char* text, const int64_t* line_off, const int32_t* len3, int64_t n_rows,
                                  const void* rows, int32_t wide, int32_t threads, char* err, size_t errlen)
{
    if (!path || n_rows < 0 || (n_rows && (!text || !line_off || !len3 || !rows))) { set_err(err, errlen, "write_bedgraph: bad argument"); return WGBSSEG_E_ARG; }
    int64_t base = 0;
    const int fd = open_for_rows(path, 0, &base, err, errlen);
    if (fd < 0) return WGBSSEG_E_ARG;
    const wgtab::Rows R = {text, line_off, len3, nullptr, nullptr, nullptr};
    std::string msg;
    const int rc = wide ? wgtab::write_bedgraph<uint16_t>(fd, base, R, n_rows, static_cast<const uint16_t*>(rows), threads, msg)
                        : wgtab::write_bedgraph<uint8_t>(fd, base, R, n_rows, static_cast<const uint8_t*>(rows), threads, msg);
    if (close(fd) != 0 && rc == 0) { set_err(err, errlen, "write to %s failed", path); return WGBSSEG_E_ARG; }
    if (rc) { set_err(err, errlen, "%s", msg.c_str()); return WGBSSEG_E_ARG; }
    return WGBSSEG_OK;
}

int wgbsseg_bed_parse(const char* text, int64_t len, int64_t cap, const char* const* chrom_names, int32_t n_chroms, int64_t* line_off,
                      int32_t* len3, int32_t* row_len, int32_t* chrom, int64_t* start, int64_t* end, int64_t* n_rows, int32_t* width, int32_t* header)
{
    if (!n_rows || !width || !header) return WGBSSEG_E_ARG;
    *n_rows = 0; *width = 0; *header = 0;
    if (!text || len < 0 || cap < 0 || n_chroms < 0 || (n_chroms && !chrom_names) || (cap && (!line_off || !len3 || !row_len || !chrom || !start || !end))) return WGBSSEG_E_ARG;
    const int rc = wgtab::parse_bed(text, len, cap, chrom_names, n_chroms, line_off, len3, row_len, chrom, start, end, n_rows, width, header);
    return rc == 0 ? WGBSSEG_OK : (rc == 1 ? 1 : WGBSSEG_E_ARG);
}

int64_t wgbsseg_debug_canonical_float(const char* tokens, int64_t len, uint8_t* out, int64_t out_cap)
{
    if (!tokens || len < 0 || !out) return -1;
    int64_t n = 0;
    for (int64_t a = 0; a < len;) {
        const char* nl = static_cast<const char*>(memchr(tokens + a, '\n', (size_t)(len - a)));
        const int64_t b = nl ? (int64_t)(nl - tokens) : len;
        if (n >= out_cap) return -1;
        out[n++] = wgtab::canonical_float(tokens + a, (size_t)(b - a)) ? 1 : 0;
        a = b + 1;
    }
    return n;
}

int wgbsseg_bed_write_annotated(const char* path, const char* text, const int64_t* line_off, const int32_t* len3, const int32_t* row_len,
                                const int64_t* start_cpg, const int64_t* end_cpg, int64_t n_rows, int32_t threads, char* err, size_t errlen)
{
    if (n_rows < 0 || (n_rows && (!text || !line_off || !len3 || !row_len || !start_cpg || !end_cpg))) { set_err(err, errlen, "bed_write_annotated: bad argument"); return WGBSSEG_E_ARG; }
    int64_t base = -1;                                   // path NULL: standard output
    const int fd = path ? open_for_rows(path, 0, &base, err, errlen) : 1;
    if (fd < 0) return WGBSSEG_E_ARG;
    std::string msg;
    const int rc = wgtab::write_annotated_bed(fd, base, text, line_off, len3, row_len, start_cpg, end_cpg, n_rows, threads, msg);
    if (path && close(fd) != 0 && rc == 0) { set_err(err, errlen, "write to %s failed", path); return WGBSSEG_E_ARG; }
    if (rc) { set_err(err, errlen, "%s", msg.c_str()); return WGBSSEG_E_ARG; }
    return WGBSSEG_OK;
}

int64_t wgbsseg_format_fixed(const double* v, int64_t n, int32_t digits, char* out, int64_t out_cap)
{
    if (n < 0 || (n && (!v || !out))) return -1;
    char* p = out;
    for (int64_t i = 0; i < n; i++) {
        if (out_cap - (p - out) < 420) return -1;
        p = wgtab::put_fixed(p, v[i], digits);
        *p++ = '\n';
    }
    return (int64_t)(p - out);
}

int wgbsseg_convert_regions(wgbsseg_ctx* c, const int64_t* chrom_lo, const int64_t* chrom_hi, const int64_t* chrom_bp, const int64_t* start,
                            const int64_t* end, const uint8_t* slow, int64_t n, int64_t* start_cpg, int64_t* end_cpg, char* err, size_t errlen)
{
    if (!c) { set_err(err, errlen, "ctx is NULL"); return WGBSSEG_E_ARG; }
    if (!c->loci) { set_err(err, errlen, "loci not set"); return WGBSSEG_E_STATE; }
    if (n < 0 || (n && (!chrom_lo || !chrom_hi || !chrom_bp || !start || !end || !slow || !start_cpg || !end_cpg))) { set_err(err, errlen, "bad arguments to convert_regions"); return WGBSSEG_E_ARG; }
    if (n == 0) return WGBSSEG_OK;
    for (int64_t i = 0; i < n; i++)
        if (chrom_lo[i] < 0 || chrom_hi[i] < chrom_lo[i] || chrom_hi[i] > c->n_loci) {
            set_err(err, errlen, "region %lld: chromosome slice [%lld, %lld) is outside the %lld resident loci", (long long)i, (long long)chrom_lo[i], (long long)chrom_hi[i], (long long)c->n_loci);
            return WGBSSEG_E_ARG;
        }
    HIP_TRY(hipSetDevice(c->device));
    const size_t nb = (size_t)n * 8;
    HIP_TRY(c->dbg_a.ensure(5 * nb + (size_t)n));
    HIP_TRY(c->dbg_b.ensure(2 * nb));
    c->table_blocks = 0;
    char* d = c->dbg_a.as<char>();
    const int64_t* srcs[5] = {chrom_lo, chrom_hi, chrom_bp, start, end};
    for (int k = 0; k < 5; k++) HIP_TRY(hipMemcpyAsync(d + k * nb, srcs[k], nb, hipMemcpyHostToDevice, c->sA));
    HIP_TRY(hipMemcpyAsync(d + 5 * nb, slow, (size_t)n, hipMemcpyHostToDevice, c->sA));
    int64_t* o = c->dbg_b.as<int64_t>();
    const int64_t gx = (n + WG_BLOCK - 1) / WG_BLOCK;
    if (gx > 0x7fffffff) { set_err(err, errlen, "too many regions for one convert call"); return WGBSSEG_E_ARG; }
    HIP_TRY(hipEventRecord(c->ev[0], c->sA));
    hipLaunchKernelGGL(k_convert, dim3((unsigned)gx), dim3(WG_BLOCK), 0, c->sA, c->loci, reinterpret_cast<const int64_t*>(d),
                       reinterpret_cast<const int64_t*>(d + nb), reinterpret_cast<const int64_t*>(d + 2 * nb), reinterpret_cast<const int64_t*>(d + 3 * nb),
                       reinterpret_cast<const int64_t*>(d + 4 * nb), reinterpret_cast<const uint8_t*>(d + 5 * nb), n, o, o + n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev[1], c->sA));
    HIP_TRY(hipMemcpyAsync(start_cpg, o, nb, hipMemcpyDeviceToHost, c->sA));
    HIP_TRY(hipMemcpyAsync(end_cpg, o + n, nb, hipMemcpyDeviceToHost, c->sA));
    HIP_TRY(hipStreamSynchronize(c->sA));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->last_block_sums_ms = ms;                            // (shared "last auxiliary kernel" clock: wgbsseg_last_block_sums_ms)
    return WGBSSEG_OK;
}

}  // extern "C"

// pat -> beta accumulator: counts on one device, text chunks through two page-locked staging buffers so that the caller's
// decompression of chunk k+1 overlaps the copy and the kernel of chunk k.
struct wgbsseg_patbeta {
    int device = 0;
    hipStream_t st = nullptr;
    int64_t start = 1, end = 1;
    DevBuf meth, cov, text[2], outb, bad;
    PinnedBuf stage[2];
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    int k = 0;
    unsigned long long fed = 0;
    double kernel_ms = 0.0;                                       // k_pat_count launches whose events have been read
    hipEvent_t k0[2] = {nullptr, nullptr}, k1[2] = {nullptr, nullptr};   // around the counting kernel of the chunk in slot k
    bool timed[2] = {false, false};                               // slot k holds a pair not yet added to kernel_ms
    hipEvent_t t0 = nullptr, t1 = nullptr;
    void collect(int k)                                           // (after the slot's work is known to be complete)
    {
        float ms = 0.f;
        if (timed[k] && hipEventElapsedTime(&ms, k0[k], k1[k]) == hipSuccess) kernel_ms += (double)ms;
        timed[k] = false;
    }
};

extern "C" {

void wgbsseg_patbeta_destroy(wgbsseg_patbeta* p);

int wgbsseg_patbeta_create(int device, int64_t start_cpg, int64_t end_cpg, wgbsseg_patbeta** out, char* err, size_t errlen)
{
    if (!out) { set_err(err, errlen, "out is NULL"); return WGBSSEG_E_ARG; }
    *out = nullptr;
    if (start_cpg < 1 || end_cpg <= start_cpg || end_cpg - start_cpg > 0x7fffffff) { set_err(err, errlen, "patbeta: bad CpG range [%lld, %lld)", (long long)start_cpg, (long long)end_cpg); return WGBSSEG_E_ARG; }
    wgbsseg_ctx* probe = nullptr;                                 // device checks (gfx950, index) as for a segment context
    int rc = wgbsseg_create(device, &probe, err, errlen);
    if (rc != WGBSSEG_OK) return rc;
    wgbsseg_destroy(probe);
    // (the struct's members are plain handles: whatever has been created when a HIP call fails is released by the destroy
    // function, which is what the owner calls on every early return)
    struct Free { void operator()(wgbsseg_patbeta* q) const { wgbsseg_patbeta_destroy(q); } };
    std::unique_ptr<wgbsseg_patbeta, Free> p(new (std::nothrow) wgbsseg_patbeta());
    if (!p) { set_err(err, errlen, "out of host memory"); return WGBSSEG_E_NOMEM; }
    p->device = device; p->start = start_cpg; p->end = end_cpg;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&p->st, hipStreamNonBlocking));
    for (auto& e : p->ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(hipEventCreate(&p->t0)); HIP_TRY(hipEventCreate(&p->t1));
    for (int k = 0; k < 2; k++) { HIP_TRY(hipEventCreate(&p->k0[k])); HIP_TRY(hipEventCreate(&p->k1[k])); }
    const size_t nb = (size_t)(end_cpg - start_cpg) * 4;
    HIP_TRY(p->meth.ensure(nb)); HIP_TRY(p->cov.ensure(nb)); HIP_TRY(p->bad.ensure(8));
    HIP_TRY(hipMemsetAsync(p->meth.p, 0, nb, p->st));            // (the reference leaves its arrays uninitialised: stdin2beta.cpp:48-49)
    HIP_TRY(hipMemsetAsync(p->cov.p, 0, nb, p->st));
    HIP_TRY(hipMemsetAsync(p->bad.p, 0xff, 8, p->st));
    *out = p.release();
    return WGBSSEG_OK;
}

void wgbsseg_patbeta_destroy(wgbsseg_patbeta* p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->st) { (void)hipStreamSynchronize(p->st); }
    for (DevBuf* b : {&p->meth, &p->cov, &p->text[0], &p->text[1], &p->outb, &p->bad}) b->release();
    for (auto& s : p->stage) s.release();
    for (auto& e : p->ev) if (e) (void)hipEventDestroy(e);
    if (p->t0) (void)hipEventDestroy(p->t0);
    if (p->t1) (void)hipEventDestroy(p->t1);
    for (int k = 0; k < 2; k++) { if (p->k0[k]) (void)hipEventDestroy(p->k0[k]); if (p->k1[k]) (void)hipEventDestroy(p->k1[k]); }
    if (p->st) (void)hipStreamDestroy(p->st);
    delete p;
}

int wgbsseg_patbeta_feed(wgbsseg_patbeta* p, const char* text, int64_t n_bytes, char* err, size_t errlen)
{
    if (!p || (n_bytes && !text) || n_bytes < 0) { set_err(err, errlen, "bad arguments to patbeta_feed"); return WGBSSEG_E_ARG; }
    if (n_bytes == 0) return WGBSSEG_OK;
    if (text[n_bytes - 1] != '\n') { set_err(err, errlen, "patbeta_feed: a chunk must end with a complete line"); return WGBSSEG_E_ARG; }
    HIP_TRY(hipSetDevice(p->device));
    const int k = p->k;
    if (p->busy[k]) { HIP_TRY(hipEventSynchronize(p->ev[k])); p->collect(k); }      // its previous chunk has been consumed
    if (!p->stage[k].ensure((size_t)n_bytes)) { set_err(err, errlen, "out of page-locked host memory"); return WGBSSEG_E_NOMEM; }
    HIP_TRY(p->text[k].ensure((size_t)n_bytes));
    memcpy(p->stage[k].p, text, (size_t)n_bytes);
    HIP_TRY(hipMemcpyAsync(p->text[k].p, p->stage[k].p, (size_t)n_bytes, hipMemcpyHostToDevice, p->st));
    const int64_t gx = (n_bytes + WG_PAT_TILE - 1) / WG_PAT_TILE;     // one workgroup per tile of text
    if (gx > 0x7fffffff) { set_err(err, errlen, "patbeta_feed: chunk too large"); return WGBSSEG_E_ARG; }
    HIP_TRY(hipEventRecord(p->k0[k], p->st));
    hipLaunchKernelGGL(k_pat_count, dim3((unsigned)gx), dim3(WG_BLOCK), 0, p->st, p->text[k].as<char>(), n_bytes, p->start, p->end,
                       p->meth.as<int32_t>(), p->cov.as<int32_t>(), p->bad.as<unsigned long long>(), p->fed);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(p->k1[k], p->st));
    p->timed[k] = true;
    HIP_TRY(hipEventRecord(p->ev[k], p->st));
    p->busy[k] = true;
    p->k ^= 1;
    p->fed += (unsigned long long)n_bytes;
    return WGBSSEG_OK;
}

int wgbsseg_patbeta_finish(wgbsseg_patbeta* p, int32_t lbeta, void* out, char* err, size_t errlen)
{
    if (!p || !out) { set_err(err, errlen, "bad arguments to patbeta_finish"); return WGBSSEG_E_ARG; }
    HIP_TRY(hipSetDevice(p->device));
    const int64_t n = p->end - p->start;
    const size_t ob = (size_t)n * (lbeta ? 4 : 2);
    HIP_TRY(p->outb.ensure(ob));
    unsigned long long bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, p->bad.p, 8, hipMemcpyDeviceToHost, p->st));
    HIP_TRY(hipStreamSynchronize(p->st));
    if (bad != ~0ULL) {
        set_err(err, errlen, "failed calculating beta: invalid line at byte offset %llu of the input (too few columns, or a site / count that is not a number)", bad);
        return WGBSSEG_E_ARG;
    }
    hipLaunchKernelGGL(k_pat_trim, dim3((unsigned)((n + WG_BLOCK - 1) / WG_BLOCK)), dim3(WG_BLOCK), 0, p->st, p->meth.as<int32_t>(), p->cov.as<int32_t>(), n,
                       (int)lbeta, p->outb.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, p->outb.p, ob, hipMemcpyDeviceToHost, p->st));
    HIP_TRY(hipStreamSynchronize(p->st));
    return WGBSSEG_OK;
}

double wgbsseg_patbeta_kernel_ms(wgbsseg_patbeta* p)
{
    if (!p) return -1.0;
    if (hipSetDevice(p->device) != hipSuccess || hipStreamSynchronize(p->st) != hipSuccess) return -1.0;
    p->collect(0); p->collect(1);
    return p->kernel_ms;
}

int wgbsseg_get_timings(const wgbsseg_ctx* c, wgbsseg_timings* out)
{
    if (!c || !out) return WGBSSEG_E_ARG;
    *out = c->tim;
    return WGBSSEG_OK;
}

int64_t wgbsseg_debug_fetch(wgbsseg_ctx* c, const char* what, void* out, int64_t cap_bytes)
{
    if (!c || !what || !out || !c->last_valid) return WGBSSEG_E_STATE;
    const void* src = nullptr;
    int64_t bytes = 0;
    if (!strcmp(what, "window")) { src = c->W16.p; bytes = c->last_sites * 2; }
    else if (!strcmp(what, "cum")) { src = c->cum32.p; bytes = c->last_sites * 4; }
    else if (!strcmp(what, "back")) { src = c->back16.p; bytes = c->last_sites * 2; }
    else if (!strcmp(what, "dpstate")) { src = c->dpstate.p; bytes = c->last_dp_chunks * c->last_dp_stride * 8; }
    else if (!strcmp(what, "cost")) { if (c->last_stages != 1) return WGBSSEG_E_STATE; src = c->cost[0].p; bytes = c->last_pairs * 8; }
    else return WGBSSEG_E_ARG;
    if (bytes > cap_bytes) return WGBSSEG_E_CAPACITY;
    if (hipSetDevice(c->device) != hipSuccess) return WGBSSEG_E_HIP;
    if (hipMemcpy(out, src, (size_t)bytes, hipMemcpyDeviceToHost) != hipSuccess) return WGBSSEG_E_HIP;
    return bytes;
}

int wgbsseg_debug_sample_terms(wgbsseg_ctx* c, const float* nmeth, const float* ntotal, int64_t count, float pseudo_count, float* out)
{
    char* err = nullptr; size_t errlen = 0;
    if (!c || !nmeth || !ntotal || !out || count < 1) return WGBSSEG_E_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->dbg_a.ensure((size_t)count * 4)); HIP_TRY(c->dbg_b.ensure((size_t)count * 4)); HIP_TRY(c->dbg_c.ensure((size_t)count * 4));
    HIP_TRY(hipMemcpyAsync(c->dbg_a.p, nmeth, (size_t)count * 4, hipMemcpyHostToDevice, c->sA));
    HIP_TRY(hipMemcpyAsync(c->dbg_b.p, ntotal, (size_t)count * 4, hipMemcpyHostToDevice, c->sA));
    const unsigned blocks = (unsigned)std::min<int64_t>((count + 255) / 256, 8192);
    const int fast = wg_term_mode(pseudo_count);                                         // the library's own dispatch rule
    const int rows = fast == 2 ? wg_lookup_rows(pseudo_count, 255.0 * 8000) : 0;       // block totals up to 255 * 8000 < 2^21: the longest the guard-free form serves
    if (rows > WG_KY_KMIN + 1) return WGBSSEG_E_ARG;
    hipLaunchKernelGGL(k_debug_terms, dim3(blocks), dim3(256), 0, c->sA, c->dbg_a.as<float>(), c->dbg_b.as<float>(), count, pseudo_count, c->dbg_c.as<float>(), fast, rows);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, c->dbg_c.p, (size_t)count * 4, hipMemcpyDeviceToHost, c->sA));
    HIP_TRY(hipStreamSynchronize(c->sA));
    return WGBSSEG_OK;
}

int wgbsseg_debug_div(wgbsseg_ctx* c, const float* a, const float* b, int64_t count, uint32_t* out_fast, uint32_t* out_ieee)
{
    char* err = nullptr; size_t errlen = 0;
    if (!c || !a || !b || !out_fast || !out_ieee || count < 1) return WGBSSEG_E_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->dbg_a.ensure((size_t)count * 8)); HIP_TRY(c->dbg_b.ensure((size_t)count * 8));
    float* da = c->dbg_a.as<float>(); float* db = da + count;
    uint32_t* o1 = c->dbg_b.as<uint32_t>(); uint32_t* o2 = o1 + count;
    HIP_TRY(hipMemcpyAsync(da, a, (size_t)count * 4, hipMemcpyHostToDevice, c->sA));
    HIP_TRY(hipMemcpyAsync(db, b, (size_t)count * 4, hipMemcpyHostToDevice, c->sA));
    const unsigned blocks = (unsigned)std::min<int64_t>((count + 255) / 256, 16384);
    hipLaunchKernelGGL(k_debug_div, dim3(blocks), dim3(256), 0, c->sA, da, db, count, o1, o2);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_fast, o1, (size_t)count * 4, hipMemcpyDeviceToHost, c->sA));
    HIP_TRY(hipMemcpyAsync(out_ieee, o2, (size_t)count * 4, hipMemcpyDeviceToHost, c->sA));
    HIP_TRY(hipStreamSynchronize(c->sA));
    return WGBSSEG_OK;
}

int wgbsseg_debug_div_short(wgbsseg_ctx* c, const float* a, const float* b, int64_t count, uint32_t* out)
{
    char* err = nullptr; size_t errlen = 0;
    if (!c || !a || !b || !out || count < 1) return WGBSSEG_E_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->dbg_a.ensure((size_t)count * 8)); HIP_TRY(c->dbg_b.ensure((size_t)count * 4));
    float* da = c->dbg_a.as<float>(); float* db = da + count;
    HIP_TRY(hipMemcpyAsync(da, a, (size_t)count * 4, hipMemcpyHostToDevice, c->sA));
    HIP_TRY(hipMemcpyAsync(db, b, (size_t)count * 4, hipMemcpyHostToDevice, c->sA));
    const unsigned blocks = (unsigned)std::min<int64_t>((count + 255) / 256, 16384);
    hipLaunchKernelGGL(k_debug_div_short, dim3(blocks), dim3(256), 0, c->sA, da, db, count, c->dbg_b.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, c->dbg_b.p, (size_t)count * 4, hipMemcpyDeviceToHost, c->sA));
    HIP_TRY(hipStreamSynchronize(c->sA));
    return WGBSSEG_OK;
}

int wgbsseg_debug_check_div(wgbsseg_ctx* c, float pseudo_count, int32_t max_total, int64_t* mismatches)
{
    char* err = nullptr; size_t errlen = 0;
    if (!c || !mismatches || max_total < 0 || max_total > (1 << 21)) return WGBSSEG_E_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->divcheck.ensure(4));
    HIP_TRY(hipMemsetAsync(c->divcheck.p, 0, 4, c->sA));
    hipLaunchKernelGGL(k_check_div, dim3((unsigned)max_total + 1), dim3(WG_BLOCK), 0, c->sA, pseudo_count, pseudo_count + pseudo_count,
                       (int)max_total, c->divcheck.as<unsigned int>());
    HIP_TRY(hipGetLastError());
    unsigned int n = 0;
    HIP_TRY(hipMemcpyAsync(&n, c->divcheck.p, 4, hipMemcpyDeviceToHost, c->sA));
    HIP_TRY(hipStreamSynchronize(c->sA));
    *mismatches = (int64_t)n;
    return WGBSSEG_OK;
}

int wgbsseg_debug_log2(wgbsseg_ctx* c, uint32_t first_bits, int64_t count, uint32_t* out_f, uint64_t* out_d, uint64_t* out_fast)
{
    char* err = nullptr; size_t errlen = 0;
    if (!c || count < 1 || (!out_f && !out_d && !out_fast)) return WGBSSEG_E_ARG;
    HIP_TRY(hipSetDevice(c->device));
    if (out_f) HIP_TRY(c->dbg_a.ensure((size_t)count * 4));
    if (out_d) HIP_TRY(c->dbg_b.ensure((size_t)count * 8));
    if (out_fast) HIP_TRY(c->dbg_c.ensure((size_t)count * 8));
    const unsigned blocks = (unsigned)std::min<int64_t>((count + 255) / 256, 16384);
    hipLaunchKernelGGL(k_debug_log2, dim3(blocks), dim3(256), 0, c->sA, first_bits, count, out_f ? c->dbg_a.as<uint32_t>() : nullptr,
                       out_d ? c->dbg_b.as<uint64_t>() : nullptr, out_fast ? c->dbg_c.as<uint64_t>() : nullptr);
    HIP_TRY(hipGetLastError());
    if (out_f) HIP_TRY(hipMemcpyAsync(out_f, c->dbg_a.p, (size_t)count * 4, hipMemcpyDeviceToHost, c->sA));
    if (out_d) HIP_TRY(hipMemcpyAsync(out_d, c->dbg_b.p, (size_t)count * 8, hipMemcpyDeviceToHost, c->sA));
    if (out_fast) HIP_TRY(hipMemcpyAsync(out_fast, c->dbg_c.p, (size_t)count * 8, hipMemcpyDeviceToHost, c->sA));
    HIP_TRY(hipStreamSynchronize(c->sA));
    return WGBSSEG_OK;
}

}  // extern "C"
