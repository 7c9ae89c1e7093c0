// stitch.h — host-side chunk grid and junction stitching of `wgbstools segment`, native.
//
// Restates, on sorted int64 border lists, the reference driver's
//   break_to_chunks   segment.py:124-135   bords = range(start, end, step) + [end]
//   merge_df_list     segment.py:157-165   pairwise tree: (0,1),(2,3).. then again on the merged list
//   stitch_2_dfs      segment.py:199-232   patch = DP over [b1[-1]-p1, b1[-1]+p2), p = min(50, span), doubling on failure
//   is_2_overlap / find_dups / merge2 / increase_patch   segment.py:235-252
// A merged list is kept as a rope of runs into stable storage (chunk results, patch results), because a stitch only
// edits the neighbourhood of its junction; the rope is flattened once at the end.  The patch DP is a pure function
// of (start, end), so every junction's first-attempt patch is known before any DP has run and goes to the GPU in
// the same batch as the chunks; only failed attempts (doubling) need a follow-up batch.
#pragma once
#include <algorithm>
#include <atomic>
#include <cassert>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>
#include <unistd.h>

namespace wgstitch {

// A run of borders: value x = p[x] + add (chunk DPs return int32 borders relative to their start; `add` = that start).
struct Run { const int32_t* p; int64_t n; int64_t add; };   // n >= 1, values strictly ascending, also across runs

struct Rope {
    std::vector<Run> runs;
    // A rope over the EDGE of a list that has not arrived in full (BatchResult::edges): the walks below may run off the part that is there —
    // then the answer is unknown, `undecided` says so and the caller sets the junction aside.
    bool partial = false;
    mutable bool undecided = false;
    int64_t front() const { return runs.front().p[0] + runs.front().add; }
    int64_t back() const { return runs.back().p[runs.back().n - 1] + runs.back().add; }
    int64_t span() const { return back() - front(); }
    // is value x present?  (binary search over runs, then inside the run)
    bool contains(int64_t x, size_t* run_idx = nullptr, int64_t* pos = nullptr) const
    {
        size_t lo = 0, hi = runs.size();                 // last run with first value <= x
        while (hi - lo > 1) { size_t mid = (lo + hi) / 2; if (runs[mid].p[0] + runs[mid].add <= x) lo = mid; else hi = mid; }
        const Run& r = runs[lo];
        const int64_t xr = x - r.add;
        if (r.p[0] > xr) return false;
        const int32_t* e = std::lower_bound(r.p, r.p + r.n, xr, [](int32_t a, int64_t b) { return (int64_t)a < b; });
        if (e == r.p + r.n || (int64_t)*e != xr) return false;
        if (run_idx) *run_idx = lo;
        if (pos) *pos = e - r.p;
        return true;
    }
    // The same question for a value expected within a few dozen entries of the rope's END / START (a junction's
    // neighbourhood): walk in from that end, so that only the cache lines next to the junction are touched — the
    // border arrays have just arrived by DMA and a binary search would take a cold miss at almost every probe.
    bool contains_near_back(int64_t x, size_t* run_idx = nullptr, int64_t* pos = nullptr) const
    {
        int budget = 96;
        for (size_t ri = runs.size(); ri-- > 0 && budget > 0;) {
            const Run& r = runs[ri];
            const int64_t xr = x - r.add;
            for (int64_t q = r.n - 1; q >= 0 && budget > 0; q--, budget--) {
                if ((int64_t)r.p[q] == xr) { if (run_idx) *run_idx = ri; if (pos) *pos = q; return true; }
                if ((int64_t)r.p[q] < xr) return false;
            }
        }
        if (partial) { undecided = true; return false; }
        return budget > 0 ? false : contains(x, run_idx, pos);
    }
    bool contains_near_front(int64_t x, size_t* run_idx = nullptr, int64_t* pos = nullptr) const
    {
        int budget = 96;
        for (size_t ri = 0; ri < runs.size() && budget > 0; ri++) {
            const Run& r = runs[ri];
            const int64_t xr = x - r.add;
            for (int64_t q = 0; q < r.n && budget > 0; q++, budget--) {
                if ((int64_t)r.p[q] == xr) { if (run_idx) *run_idx = ri; if (pos) *pos = q; return true; }
                if ((int64_t)r.p[q] > xr) return false;
            }
        }
        if (partial) { undecided = true; return false; }
        return budget > 0 ? false : contains(x, run_idx, pos);
    }
    // keep everything up to and including value x (x must be present; it lies near the end)
    void truncate_after(int64_t x)
    {
        size_t ri = 0; int64_t pos = 0;
        contains_near_back(x, &ri, &pos);
        runs.resize(ri + 1);
        runs[ri].n = pos + 1;
    }
    // append the elements of `o` that are > x (x must be present in o; it lies near o's start)
    void append_after(const Rope& o, int64_t x)
    {
        size_t ri = 0; int64_t pos = 0;
        o.contains_near_front(x, &ri, &pos);
        if (pos + 1 < o.runs[ri].n) runs.push_back(Run{o.runs[ri].p + pos + 1, o.runs[ri].n - pos - 1, o.runs[ri].add});
        for (size_t q = ri + 1; q < o.runs.size(); q++) runs.push_back(o.runs[q]);
    }
    int64_t size() const { int64_t s = 0; for (auto& r : runs) s += r.n; return s; }
    void flatten(int32_t* out) const { for (auto& r : runs) { const int32_t a = (int32_t)r.add; for (int64_t q = 0; q < r.n; q++) out[q] = r.p[q] + a; out += r.n; } }
};

inline int64_t increase_patch(int64_t pre, int64_t maxval)    // segment.py:249-252
{
    if (pre == maxval) return maxval + 1;
    return std::min(pre * 2, maxval);
}

// One stitch_2_dfs in progress.
struct Stitch {
    Rope b1, b2;
    int64_t n1, n2, p1, p2;
    bool done = false;
    Rope result;
    bool init(Rope&& a, Rope&& b, std::string& err)
    {
        b1 = std::move(a); b2 = std::move(b);
        if (b1.back() != b2.front()) {                          // segment.py:202-205
            err = "[wt segment] Patch stitching Failed!              patches are not supposed to be merged";
            return false;
        }
        n1 = b1.span(); n2 = b2.span();
        p1 = std::min<int64_t>(50, n1); p2 = std::min<int64_t>(50, n2);
        return true;
    }
    // the patch this junction needs next; false + err when the reference would give up (segment.py:229-232)
    bool want(std::pair<int64_t, int64_t>& sites, std::string& err) const
    {
        if (!(p1 <= n1 && p2 <= n2)) {
            err = "[wt segment] Patch stitching Failed!              Try increasing chunk size (--chunk_size flag)";
            return false;
        }
        sites = {b1.back() - p1, b1.back() + p2};
        return true;
    }
    void feed(const int32_t* patch, int64_t np, int64_t add)      // patch value q = patch[q] + add
    {
        // is_2_overlap(b1, patch) / (patch, b2): a common value exists (segment.py:235-240)
        int64_t x1 = 0, x2 = 0;
        bool o1 = false, o2 = false;
        for (int64_t q = 0; q < np && !o1; q++) if (b1.contains_near_back(patch[q] + add)) { o1 = true; x1 = patch[q] + add; }   // smallest common value
        for (int64_t q = 0; q < np && !o2; q++) if (b2.contains_near_front(patch[q] + add)) { o2 = true; x2 = patch[q] + add; }
        if (o1 && o2) {
            // merge2(merge2(b1, patch), b2) (segment.py:221,243-246):
            //   m = b1[.. x1] + patch[> x1];   the first element of m that occurs in b2 is the junction value itself
            //   when x1 is the junction (then nothing of the patch survives), else the smallest patch value > x1 in b2.
            result = std::move(b1);
            result.truncate_after(x1);
            const int64_t junction = b2.front();
            if (x1 == junction) {
                result.append_after(b2, junction);
            } else {
                const int32_t* a = std::upper_bound(patch, patch + np, x1 - add, [](int64_t v, int32_t e) { return v < (int64_t)e; });
                const int32_t* b = std::lower_bound(patch, patch + np, x2 - add, [](int32_t e, int64_t v) { return (int64_t)e < v; });   // x2 >= junction > x1
                if (b + 1 > a) result.runs.push_back(Run{a, (b + 1) - a, add});
                result.append_after(b2, x2);
            }
            done = true;
        } else {
            if (!o1) p1 = increase_patch(p1, n1);
            if (!o2) p2 = increase_patch(p2, n2);
        }
    }
};

// First-attempt patch of every junction of a region cut into chunks of lengths `lens` starting at 1-based `start`:
// the pairwise tree fixes the operand spans and with them p1, p2.
inline void upfront_patches(int64_t start, const std::vector<int64_t>& lens, bool speculate, std::vector<std::pair<int64_t, int64_t>>& out)
{
    struct Seg { int64_t a, b; };                              // site range [a, b)
    std::vector<Seg> segs;
    int64_t pos = start;
    for (auto l : lens) { segs.push_back({pos, pos + l}); pos += l; }
    while (segs.size() > 1) {
        std::vector<Seg> nxt;
        for (size_t i = 1; i < segs.size(); i += 2) {
            const Seg &L = segs[i - 1], &R = segs[i];
            const int64_t n1 = L.b - L.a, n2 = R.b - R.a;
            const int64_t p1 = std::min<int64_t>(50, n1), p2 = std::min<int64_t>(50, n2);
            out.push_back({L.b - p1, L.b + p2});
            if (speculate) {                                   // the three possible second attempts (segment.py:222-227)
                const int64_t q1 = increase_patch(p1, n1), q2 = increase_patch(p2, n2);
                if (q1 <= n1) out.push_back({L.b - q1, L.b + p2});
                if (q2 <= n2) out.push_back({L.b - p1, L.b + q2});
                if (q1 <= n1 && q2 <= n2) out.push_back({L.b - q1, L.b + q2});
            }
            nxt.push_back({L.a, R.b});
        }
        if (segs.size() % 2) nxt.push_back(segs.back());
        segs.swap(nxt);
    }
}

// The junctions of the same tree, each with the two chunks that touch it and the spans of its operands.
struct Junction { int64_t left_item, right_item, n1, n2; };
inline void junctions_of_region(const std::vector<int64_t>& lens, int64_t first_item, std::vector<Junction>& out)
{
    struct Seg { int64_t len, lo, hi; };                       // items [lo, hi]
    std::vector<Seg> segs;
    for (size_t i = 0; i < lens.size(); i++) segs.push_back({lens[i], first_item + (int64_t)i, first_item + (int64_t)i});
    while (segs.size() > 1) {
        std::vector<Seg> nxt;
        for (size_t i = 1; i < segs.size(); i += 2) {
            out.push_back({segs[i - 1].hi, segs[i].lo, segs[i - 1].len, segs[i].len});
            nxt.push_back({segs[i - 1].len + segs[i].len, segs[i - 1].lo, segs[i].hi});
        }
        if (segs.size() % 2) nxt.push_back(segs.back());
        segs.swap(nxt);
    }
}

// A few persistent host threads for the junction-local work of a call (rehearsal of ~500 junctions, the trees of 25
// chromosomes, flattening 2.8 M borders): a call lasts a millisecond, so threads spawned per call would cost what they save.
// Created on first use, never torn down (the process exit takes them); WGBSSEG_STITCH_THREADS=1 keeps everything on the caller.
class Pool {
public:
    static Pool& get() { static Pool* p = new Pool(); return *p; }
    int threads() const { return (int)workers_.size() + 1; }
    // f(k) for k = 0 .. n-1 on the pool's threads and the calling one; returns when all are done
    void run(int64_t n, const std::function<void(int64_t)>& f)
    {
        if (n <= 0) return;
        // one parallel section at a time: a second caller (another context on another thread) does its own work itself, and so
        // does a forked child, which has inherited the object but not the threads
        // (and a section started from INSIDE a section — by the caller's own f(k) or by a worker's — runs on its thread: try_lock
        // on a mutex the thread already holds is undefined, so nesting is told apart by a thread-local flag first)
        static thread_local bool inside = false;
        if (inside || workers_.empty() || n == 1 || getpid() != pid_) { for (int64_t k = 0; k < n; k++) f(k); return; }
        std::unique_lock<std::mutex> turn(run_m_, std::try_to_lock);
        if (!turn.owns_lock()) { for (int64_t k = 0; k < n; k++) f(k); return; }
        struct Mark { bool& b; explicit Mark(bool& x) : b(x) { b = true; } ~Mark() { b = false; } } mark(inside);
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = &f; n_ = n; next_.store(0); pending_.store((int)workers_.size());
            grab_ = std::max<int64_t>(1, n / (4 * ((int64_t)workers_.size() + 1)));      // items per visit to the shared counter
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        for (int64_t k; (k = next_.fetch_add(grab_)) < n;) for (int64_t q = k; q < std::min(n, k + grab_); q++) f(q);
        while (pending_.load(std::memory_order_acquire) != 0) cpu_relax();       // (the stragglers are on their last item)
        job_ = nullptr;
    }
    // The threads sleep between calls, and a sleeping core takes ~100 us to come back — as long as the work they are wanted for.
    // heat() wakes them ahead of time (the library calls it when the recurrence of a batch is done: traceback and the copy of
    // the borders, ~0.4 ms, are still to come); awake, they poll for work for `us` microseconds before they go back to sleep.
    void heat(int us = 3000)
    {
        if (workers_.empty() || getpid() != pid_) return;
        hot_until_.store(now_us() + us, std::memory_order_relaxed);
        { std::lock_guard<std::mutex> g(m_); }
        cv_.notify_all();
    }
private:
    Pool()
    {
        int t = (int)std::min<unsigned>(8, std::max<unsigned>(1, std::thread::hardware_concurrency() / 2));
        if (const char* e = getenv("WGBSSEG_STITCH_THREADS")) t = std::max(1, atoi(e));
        for (int i = 1; i < t; i++) { workers_.emplace_back([this] { loop(); }); workers_.back().detach(); }
    }
    static int64_t now_us() { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    static void cpu_relax()
    {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            // wait for a new job: polling while hot, on the condition variable otherwise
            while (gen_.load(std::memory_order_acquire) == seen) {
                if (now_us() < hot_until_.load(std::memory_order_relaxed)) { cpu_relax(); continue; }
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return gen_.load(std::memory_order_acquire) != seen || now_us() < hot_until_.load(std::memory_order_relaxed); });
            }
            const std::function<void(int64_t)>* f;
            int64_t n, grab;
            {
                std::lock_guard<std::mutex> g(m_);
                seen = gen_.load(std::memory_order_acquire); f = job_; n = n_; grab = grab_;
            }
            for (int64_t k; (k = next_.fetch_add(grab)) < n;) for (int64_t q = k; q < std::min(n, k + grab); q++) (*f)(q);
            pending_.fetch_sub(1, std::memory_order_release);
            hot_until_.store(std::max(hot_until_.load(std::memory_order_relaxed), now_us() + 300), std::memory_order_relaxed);   // the next section usually follows at once
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_, run_m_;
    pid_t pid_ = getpid();
    std::condition_variable cv_;
    const std::function<void(int64_t)>* job_ = nullptr;
    int64_t n_ = 0, grab_ = 1;
    std::atomic<int64_t> next_{0};
    std::atomic<int> pending_{0};
    std::atomic<uint64_t> gen_{0};
    std::atomic<int64_t> hot_until_{0};
};

typedef std::pair<int64_t, int64_t> Sites;                     // 1-based [start, end)
// Result of one batch of chunk DPs: per item, int32 borders RELATIVE to the item's start (first 0, last end-start).
// The items of a batch may live in several buffers (one per GPU share); all must stay valid until segment_regions returns.
struct BatchResult {
    std::vector<const int32_t*> ptr;         // [items]
    std::vector<int64_t> cnt;                // [items]
    std::vector<std::unique_ptr<int32_t[]>> owned;   // optional owners of what `ptr` points into
    // Early delivery (optional).  The caller sets n_lead = the number of leading items (the chunks of the grid) whose lists it can do without
    // for a while; a batch function that can deliver early then returns with cnt[] of every item, ptr[] of the others (the junction patches)
    // and `edges` valid — [n_lead][front | back][edge_n] borders: the first and the last min(edge_n, cnt) entries of every leading list, the
    // back right-aligned — while the leading lists themselves are still in flight: ptr[i < n_lead] may be read after finish() has returned 0.
    int64_t n_lead = 0;
    const int32_t* edges = nullptr;
    int32_t edge_n = 0;
    std::function<int(std::string&)> finish;
    // CSR form: item i = flat[off[i] .. off[i+1])
    void set_csr(const int32_t* flat, const int64_t* off, size_t n)
    {
        ptr.resize(n); cnt.resize(n);
        for (size_t i = 0; i < n; i++) { ptr[i] = flat + off[i]; cnt[i] = off[i + 1] - off[i]; }
    }
};
typedef std::function<int(const std::vector<Sites>&, BatchResult&, std::string&)> BatchFn;
enum { E_ARG = -1, E_CAPACITY = -6 };

// Sites -> V, open addressing (round 5: two std::map<Sites, .> of ~2,000 patches each cost 0.2 ms of every whole-genome call in node allocations, on the host,
// before the first kernel could start).  Grows by doubling; nothing is ever erased; find() results are invalidated by an insertion, so readers that run beside each
// other (the rehearsal's pool threads) only ever find().
template <class V>
class SiteTable {
    struct Slot { Sites k; V v; };
    std::vector<Slot> t_;
    size_t n_ = 0;
    static constexpr int64_t EMPTY = INT64_MIN;
    static size_t hash(const Sites& s)
    {
        uint64_t x = (uint64_t)s.first * 0x9E3779B97F4A7C15ull ^ ((uint64_t)s.second + 0x632BE59BD9B4E019ull) * 0xC2B2AE3D27D4EB4Full;
        return (size_t)(x ^ (x >> 31));
    }
    size_t probe(const Sites& k) const
    {
        const size_t m = t_.size() - 1;
        size_t i = hash(k) & m;
        while (t_[i].k.first != EMPTY && t_[i].k != k) i = (i + 1) & m;
        return i;
    }
    void grow()
    {
        std::vector<Slot> old;
        old.swap(t_);
        t_.assign(old.size() * 2, Slot{Sites(EMPTY, 0), V()});
        for (Slot& s : old) if (s.k.first != EMPTY) t_[probe(s.k)] = s;
    }
public:
    explicit SiteTable(size_t expect = 8)
    {
        size_t c = 16;
        while (c < 2 * expect) c <<= 1;
        t_.assign(c, Slot{Sites(EMPTY, 0), V()});
    }
    const V* find(const Sites& k) const { const Slot& s = t_[probe(k)]; return s.k.first == EMPTY ? nullptr : &s.v; }
    bool count(const Sites& k) const { return find(k) != nullptr; }
    V& operator[](const Sites& k)                               // (inserts a default value when the key is new)
    {
        assert(k.first != EMPTY && "SiteTable: INT64_MIN is the empty slot's mark, never a site");      // (sites are 1-based: region_start >= 1 is checked by first_batch)
        size_t i = probe(k);
        if (t_[i].k.first == EMPTY) {
            if (2 * (n_ + 1) > t_.size()) { grow(); i = probe(k); }
            t_[i].k = k;
            n_++;
        }
        return t_[i].v;
    }
};

// The items of the FIRST batch of segment_regions — every chunk of the grid (segment.py:124-135), then every junction's
// first-attempt patch and (speculate) its three possible second attempts, without repeats — a pure function of the regions and
// the chunk size: a multi-process run lets every rank work out the same list and compute the items it holds.
struct FirstBatch {
    std::vector<Sites> items;                                  // chunks first, then patches
    std::vector<int64_t> region_first_chunk;                   // [n_regions + 1]
    std::vector<Sites> patches;                                // as planned (with repeats)
    std::vector<Junction> junctions;
    int64_t n_chunks = 0;
};
inline int first_batch(const int64_t* region_start, const int64_t* region_end, int64_t n_regions, int64_t chunk_size, bool speculate,
                       FirstBatch& fb, std::string& err)
{
    fb.region_first_chunk.assign((size_t)n_regions + 1, 0);
    for (int64_t r = 0; r < n_regions; r++) {
        const int64_t a = region_start[r], b = region_end[r];
        if (a < 1 || b <= a || b > 0x7fffffff) { err = "region " + std::to_string(r) + " is empty, starts before site 1 or ends beyond 2^31"; return E_ARG; }
        fb.region_first_chunk[(size_t)r] = (int64_t)fb.items.size();
        std::vector<int64_t> lens;
        for (int64_t s = a; s < b; s += chunk_size) { const int64_t e = std::min(s + chunk_size, b); fb.items.push_back({s, e}); lens.push_back(e - s); }
        upfront_patches(a, lens, speculate, fb.patches);
        junctions_of_region(lens, fb.region_first_chunk[(size_t)r], fb.junctions);
    }
    fb.region_first_chunk[(size_t)n_regions] = (int64_t)fb.items.size();
    fb.n_chunks = (int64_t)fb.items.size();
    SiteTable<char> seen(fb.patches.size());
    for (auto& p : fb.patches) if (!seen.count(p)) { seen[p] = 1; fb.items.push_back(p); }
    return 0;
}

// The whole driver loop of segment.py:137-165 over `n_regions` regions; see include/wgbsseg.h wgbsseg_segment_regions.
inline int segment_regions(const int64_t* region_start, const int64_t* region_end, int64_t n_regions, int64_t chunk_size,
                           const BatchFn& run_batch, int32_t* borders_out, int64_t borders_cap, int64_t* borders_off,
                           int64_t* stats, std::string& err, bool speculate = true)
{
    if (!region_start || !region_end || n_regions < 1 || chunk_size < 1 || !borders_out || !borders_off) { err = "bad arguments to segment_regions"; return E_ARG; }
    typedef std::chrono::steady_clock Clock;
    const Clock::time_point t_begin = Clock::now();
    int64_t us_batches = 0;
    const bool prof = getenv("WGBSSEG_PROFILE_STITCH") != nullptr;
    std::vector<std::pair<const char*, Clock::time_point>> marks;
    auto mark = [&](const char* what) { if (prof) marks.emplace_back(what, Clock::now()); };
    mark("begin");
    auto timed_batch = [&](const std::vector<Sites>& todo, BatchResult& res) -> int {
        const Clock::time_point t0 = Clock::now();
        const int r = run_batch(todo, res, err);
        us_batches += std::chrono::duration_cast<std::chrono::microseconds>(Clock::now() - t0).count();
        return r;
    };
    // ---- chunk grid (segment.py:124-135) and first-attempt patches ---------------------------------------------
    FirstBatch fb;
    const int grid_rc = first_batch(region_start, region_end, n_regions, chunk_size, speculate, fb, err);
    if (grid_rc != 0) return grid_rc;
    std::vector<Sites>& items = fb.items;                      // chunks first, then patches
    std::vector<int64_t>& region_first_chunk = fb.region_first_chunk;
    std::vector<Sites>& patches = fb.patches;
    std::vector<Junction>& junctions = fb.junctions;
    const int64_t n_chunks = fb.n_chunks;
    struct Patch { const int32_t* p; int64_t n; };             // into a BatchResult kept alive in `keep`
    SiteTable<Patch> cache(2 * (items.size() - (size_t)n_chunks) + 64);
    for (size_t i = (size_t)n_chunks; i < items.size(); i++) cache[items[i]] = Patch{nullptr, 0};
    int64_t n_batches = 0, n_patch_dp = 0;
    std::vector<std::unique_ptr<BatchResult>> keep;

    mark("grid + patch list");
    keep.emplace_back(new BatchResult());
    BatchResult& first = *keep.back();
    first.n_lead = speculate ? n_chunks : 0;                   // (early delivery, if the batch function offers it: see BatchResult)
    int rc = timed_batch(items, first);
    mark("first batch");
    const int64_t us_first = us_batches;
    if (rc != 0) return rc;
    n_batches++;
    for (size_t i = (size_t)n_chunks; i < items.size(); i++) { cache[items[i]] = Patch{first.ptr[i], first.cnt[i]}; n_patch_dp++; }

    // ---- rehearsal ---------------------------------------------------------------------------------------------------
    // The tree below meets its junctions level by level, and a junction whose cached attempts all fail costs a
    // follow-up batch at ITS level — up to one small, latency-bound batch per level.  But a junction's fate depends only
    // on the borders next to it, i.e. on the two chunks that touch it (patches are far smaller than chunks): so play
    // every junction now against those two chunks, collect ALL missing patches, and fetch them in one batch (repeat while
    // something is missing).  Only the cache is filled here; the tree then does the real work and, where the rehearsal
    // could not foresee a request (a patch outgrowing its chunk), still asks for it.
    // Round 6: with early delivery the first pass plays against the EDGES of the chunks' lists while the lists themselves
    // (11 MB for a genome) are still on their way, and the follow-up batch runs beside their transfer; a junction whose
    // walk runs off an edge is set aside for the passes on the full lists.
    Pool& pool = Pool::get();
    struct Sim { std::vector<Sites> want; bool still = false, unknown = false; };
    auto play = [&](const std::vector<Junction>& pend, bool on_edges, std::vector<Sim>& sims) {
        sims.assign(pend.size(), Sim());
        const int64_t E = first.edge_n;
        pool.run((int64_t)pend.size(), [&](int64_t k) {
            const Junction& jn = pend[(size_t)k];
            Sim& out = sims[(size_t)k];
            const size_t li = (size_t)jn.left_item, ri = (size_t)jn.right_item;
            const int64_t llen = items[li].second - items[li].first, rlen = items[ri].second - items[ri].first;
            Stitch t;
            Rope a, b;
            if (on_edges) {
                const int64_t ml = std::min<int64_t>(E, first.cnt[li]), mr = std::min<int64_t>(E, first.cnt[ri]);
                a.runs.push_back(Run{first.edges + ((int64_t)li * 2 + 1) * E + (E - ml), ml, items[li].first});      // the last ml borders of the left chunk
                b.runs.push_back(Run{first.edges + ((int64_t)ri * 2) * E, mr, items[ri].first});                      // the first mr of the right one
                a.partial = ml < first.cnt[li]; b.partial = mr < first.cnt[ri];
            } else {
                a.runs.push_back(Run{first.ptr[li], first.cnt[li], items[li].first});
                b.runs.push_back(Run{first.ptr[ri], first.cnt[ri], items[ri].first});
            }
            std::string e2;
            if (!t.init(std::move(a), std::move(b), e2)) return;
            t.n1 = jn.n1; t.n2 = jn.n2;
            t.p1 = std::min<int64_t>(50, t.n1); t.p2 = std::min<int64_t>(50, t.n2);
            while (!t.done) {
                Sites w;
                if (!t.want(w, e2) || t.p1 > llen || t.p2 > rlen) break;       // the tree will deal with it
                const Patch* it = cache.find(w);
                if (it == nullptr || it->p == nullptr) {
                    out.want.push_back(w);
                    const int64_t j = t.b1.back();
                    const int64_t q1 = increase_patch(t.p1, t.n1), q2 = increase_patch(t.p2, t.n2);
                    const Sites alt[3] = {{j - q1, j + t.p2}, {j - t.p1, j + q2}, {j - q1, j + q2}};
                    const bool ok[3] = {q1 <= t.n1, q2 <= t.n2, q1 <= t.n1 && q2 <= t.n2};
                    for (int q = 0; q < 3; q++) if (ok[q]) out.want.push_back(alt[q]);
                    out.still = true;
                    break;
                }
                t.feed(it->p, it->n, w.first);
                if (t.b1.undecided || t.b2.undecided || t.result.undecided) { out.unknown = true; out.want.clear(); out.still = false; break; }
            }
        });
    };
    // what the junctions `pend` still miss (in junction order), fetched in one batch; -> the junctions to play again
    auto fetch = [&](const std::vector<Junction>& pend, const std::vector<Sim>& sims, std::vector<Junction>& again, bool& fetched) -> int {
        std::vector<Sites> need;
        for (size_t k = 0; k < pend.size(); k++) {
            for (const Sites& w : sims[k].want) if (!cache.count(w)) { cache[w] = Patch{nullptr, 0}; need.push_back(w); }
            if (sims[k].still || sims[k].unknown) again.push_back(pend[k]);
        }
        mark("  rehearsal: list of missing patches");
        fetched = !need.empty();
        if (need.empty()) return 0;
        keep.emplace_back(new BatchResult());
        BatchResult& res = *keep.back();
        const int r = timed_batch(need, res);
        if (r != 0) return r;
        n_batches++;
        for (size_t i = 0; i < need.size(); i++) { cache[need[i]] = Patch{res.ptr[i], res.cnt[i]}; n_patch_dp++; }
        mark("  rehearsal: follow-up batch");
        return 0;
    };
    std::vector<Junction> pend;
    if (speculate) pend = junctions;
    if (first.finish) {
        if (speculate && first.edges && first.edge_n > 0) {
            std::vector<Sim> sims;
            play(pend, true, sims);
            mark("  rehearsal on the edges: junctions against the cache");
            std::vector<Junction> again;
            bool fetched = false;
            rc = fetch(pend, sims, again, fetched);
            if (rc != 0) { std::string e2; (void)first.finish(e2); return rc; }
            pend.swap(again);
        }
        rc = first.finish(err);
        mark("  the chunks' lists are home");
        if (rc != 0) return rc;
    }
    if (speculate) {
        for (int pass = 0; pass < 4 && !pend.empty(); pass++) {
            // every junction against the cache as it stands (read-only: the junctions run on the pool's threads) ...
            std::vector<Sim> sims;
            play(pend, false, sims);
            mark("  rehearsal: junctions against the cache");
            // ... then, in junction order, what is missing
            std::vector<Junction> again;
            bool fetched = false;
            rc = fetch(pend, sims, again, fetched);
            if (rc != 0) return rc;
            if (!fetched) break;
            pend.swap(again);
        }
    }
    mark("rehearsal (incl. its batches)");
    // ---- pairwise-tree stitching (segment.py:157-165), all regions advancing round by round ------------------------
    // Regions never interact: each region's tree on a pool thread, against the cache as the rehearsal left it (read-only).
    // A region whose tree asks for a patch that is not there (or fails) is set aside and goes through the batching loop below,
    // which fetches what is missing round by round and reports errors in the reference's order.
    std::vector<std::vector<Rope>> lists((size_t)n_regions);
    auto chunk_ropes = [&](int64_t r, std::vector<Rope>& l) {
        l.clear();
        for (int64_t q = region_first_chunk[(size_t)r]; q < region_first_chunk[(size_t)r + 1]; q++) {
            Rope rp;
            rp.runs.push_back(Run{first.ptr[(size_t)q], first.cnt[(size_t)q], items[(size_t)q].first});
            l.push_back(std::move(rp));
        }
    };
    pool.run(n_regions, [&](int64_t r) {
        std::vector<Rope> l;
        chunk_ropes(r, l);
        std::string e2;
        bool ok = true;
        while (ok && l.size() > 1) {
            std::vector<Rope> nxt;
            for (size_t i = 1; ok && i < l.size(); i += 2) {
                Stitch st1;
                if (!st1.init(std::move(l[i - 1]), std::move(l[i]), e2)) { ok = false; break; }
                while (!st1.done) {
                    Sites w;
                    if (!st1.want(w, e2)) { ok = false; break; }
                    const Patch* it = cache.find(w);
                    if (it == nullptr || it->p == nullptr) { ok = false; break; }
                    st1.feed(it->p, it->n, w.first);
                }
                if (ok) nxt.push_back(std::move(st1.result));
            }
            if (!ok) break;
            if (l.size() % 2) nxt.push_back(std::move(l.back()));
            l.swap(nxt);
        }
        if (ok) lists[(size_t)r] = std::move(l);                 // one rope: done
        else chunk_ropes(r, lists[(size_t)r]);                   // from scratch in the loop below
    });
    for (;;) {
        bool any = false;
        for (auto& l : lists) any = any || l.size() > 1;
        if (!any) break;
        std::vector<Stitch> st;
        std::vector<int64_t> owner;
        std::vector<Rope> leftover((size_t)n_regions);
        std::vector<char> has_left((size_t)n_regions, 0);
        for (int64_t r = 0; r < n_regions; r++) {
            auto& l = lists[(size_t)r];
            if (l.size() <= 1) continue;
            for (size_t i = 1; i < l.size(); i += 2) {
                st.emplace_back();
                if (!st.back().init(std::move(l[i - 1]), std::move(l[i]), err)) return E_ARG;
                owner.push_back(r);
            }
            if (l.size() % 2) { leftover[(size_t)r] = std::move(l.back()); has_left[(size_t)r] = 1; }
            l.clear();
        }
        for (;;) {
            std::vector<Sites> need;
            bool pending = false;
            for (auto& s : st) {
                if (s.done) continue;
                Sites w;
                if (!s.want(w, err)) return E_ARG;
                if (cache.count(w)) continue;
                cache[w] = Patch{nullptr, 0}; need.push_back(w);
                if (speculate) {                               // a follow-up batch is due anyway: add what the junction will ask for if this attempt fails too
                    const int64_t j = s.b1.back();
                    const int64_t q1 = increase_patch(s.p1, s.n1), q2 = increase_patch(s.p2, s.n2);
                    const Sites alt[3] = {{j - q1, j + s.p2}, {j - s.p1, j + q2}, {j - q1, j + q2}};
                    const bool ok[3] = {q1 <= s.n1, q2 <= s.n2, q1 <= s.n1 && q2 <= s.n2};
                    for (int a = 0; a < 3; a++)
                        if (ok[a] && !cache.count(alt[a])) { cache[alt[a]] = Patch{nullptr, 0}; need.push_back(alt[a]); }
                }
            }
            if (!need.empty()) {
                keep.emplace_back(new BatchResult());
                BatchResult& res = *keep.back();
                rc = timed_batch(need, res);
                if (rc != 0) return rc;
                n_batches++;
                for (size_t i = 0; i < need.size(); i++) { cache[need[i]] = Patch{res.ptr[i], res.cnt[i]}; n_patch_dp++; }
            }
            for (auto& s : st) {
                if (s.done) continue;
                Sites w;
                s.want(w, err);
                const Patch& pv = cache[w];
                s.feed(pv.p, pv.n, w.first);
                pending = pending || !s.done;
            }
            if (!pending) break;
        }
        for (size_t i = 0; i < st.size(); i++) lists[(size_t)owner[i]].push_back(std::move(st[i].result));
        for (int64_t r = 0; r < n_regions; r++) if (has_left[(size_t)r]) lists[(size_t)r].push_back(std::move(leftover[(size_t)r]));
    }
    mark("stitching (incl. follow-up batches)");
    // ---- flatten --------------------------------------------------------------------------------------------------
    int64_t total = 0;
    for (int64_t r = 0; r < n_regions; r++) { borders_off[r] = total; total += lists[(size_t)r][0].size(); }
    borders_off[n_regions] = total;
    if (total > borders_cap) { err = "borders_out too small: need " + std::to_string(total); return E_CAPACITY; }
    {   // pieces of ~64 k borders (whole runs) rather than whole regions: chr1 holds 9 % of a genome's borders, and the threads that got the small
        // chromosomes would wait for the one that got it
        struct Piece { const Run* r0; const Run* r1; int32_t* out; };
        std::vector<Piece> pieces;
        for (int64_t r = 0; r < n_regions; r++) {
            const std::vector<Run>& rr = lists[(size_t)r][0].runs;
            int32_t* out = borders_out + borders_off[r];
            int64_t acc = 0;
            size_t start = 0;
            for (size_t i = 0; i < rr.size(); i++) {
                acc += rr[i].n;
                if (acc >= 65536 || i + 1 == rr.size()) { pieces.push_back(Piece{rr.data() + start, rr.data() + i + 1, out}); out += acc; acc = 0; start = i + 1; }
            }
        }
        pool.run((int64_t)pieces.size(), [&](int64_t k) {
            const Piece& pc = pieces[(size_t)k];
            int32_t* out = pc.out;
            for (const Run* q = pc.r0; q != pc.r1; q++) { const int32_t a = (int32_t)q->add; for (int64_t x = 0; x < q->n; x++) out[x] = q->p[x] + a; out += q->n; }
        });
    }
    mark("flatten");
    if (prof) {
        for (size_t i = 1; i < marks.size(); i++)
            fprintf(stderr, "[stitch] %-38s %8.1f us\n", marks[i].first, std::chrono::duration<double, std::micro>(marks[i].second - marks[i - 1].second).count());
        fprintf(stderr, "[stitch] of which device batches %lld us (first %lld)\n", (long long)us_batches, (long long)us_first);
    }
    if (stats) {
        stats[0] = n_chunks; stats[1] = n_patch_dp; stats[2] = n_batches; stats[3] = (int64_t)patches.size();
        stats[4] = std::chrono::duration_cast<std::chrono::microseconds>(Clock::now() - t_begin).count();   // whole call, host wall
        stats[5] = us_first;                                                                                // first (main) batch
        stats[6] = us_batches - us_first;                                                                   // follow-up batches
        stats[7] = total;
    }
    return 0;
}

}  // namespace wgstitch
