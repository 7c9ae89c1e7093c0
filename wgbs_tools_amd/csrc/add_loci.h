// add_loci.h — host side of the path's last step: blocks (startCpG, endCpG) -> BED rows
//     chrom \t start \t end \t startCpG \t endCpG \n
// with start = loci[startCpG-1], end = loci[endCpG-2]+1 (start+2 for an empty block), the chromosome looked up in the
// cumulative CpG counts, and the reference's validations in the reference's order.  Replaces the `add_loci` binary the
// reference pipes its blocks through (src/cpg2bed/add_loci.cpp:22-57, cpg_dict.cpp:118-131), without the 25 tabix
// processes that binary needs to load the loci.  Formatting is sharded over host threads; rows are written in order
// while later shards are still being formatted.
#pragma once
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <unistd.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace wgadd {

struct Genome {
    const uint32_t* loci;            // bp position of CpG i+1
    int64_t n_sites;
    const int64_t* cum;              // cumulative CpG counts per chromosome (cum[n_chroms-1] == n_sites)
    const char* const* names;
    int n_chroms;
};

// Decimal text of v, two digits per division (a row carries ~34 digits: the divisions are what formatting costs).
inline char* put_u64(char* p, uint64_t v)
{
    static const char D2[201] =
        "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
    char tmp[24];
    int n = 0;
    if (v <= 0xffffffffu) {                                       // loci and CpG indices: 32-bit divisions
        uint32_t w = (uint32_t)v;
        while (w >= 100) { const uint32_t q = w / 100, r = w - q * 100; tmp[n++] = D2[2 * r + 1]; tmp[n++] = D2[2 * r]; w = q; }
        if (w >= 10) { tmp[n++] = D2[2 * w + 1]; tmp[n++] = D2[2 * w]; } else tmp[n++] = (char)('0' + w);
    } else {
        while (v >= 100) { const uint64_t q = v / 100, r = v - q * 100; tmp[n++] = D2[2 * r + 1]; tmp[n++] = D2[2 * r]; v = q; }
        if (v >= 10) { tmp[n++] = D2[2 * v + 1]; tmp[n++] = D2[2 * v]; } else tmp[n++] = (char)('0' + v);
    }
    while (n) *p++ = tmp[--n];
    return p;
}

// chromosome index of a 1-based CpG index (cpg_dict.cpp:118-131): the first chromosome whose cumulative count is >= loc;
// nr_sites+1 (a non-inclusive end) belongs to the last chromosome; anything else is an error (-1)
inline int loc2chrom(const Genome& g, int64_t loc, int hint = -1)
{
    // (tables come sorted: the chromosome of the previous site is almost always the answer)
    if (hint >= 0 && hint < g.n_chroms && loc <= g.cum[hint] && (hint == 0 || loc > g.cum[hint - 1]) && loc >= 1) return hint;
    const int64_t* e = g.cum + g.n_chroms;
    const int64_t* it = std::lower_bound(g.cum, e, loc);
    if (it != e) return (int)(it - g.cum);
    if (loc == g.n_sites + 1) return g.n_chroms - 1;
    return -1;
}

// Validates like add_loci.cpp:38-49.  Returns 0, or the 1-based position of the failing check with `line` / `msg` set.
inline int check_row(const Genome& g, int64_t s, int64_t e, int& c1, std::string& msg, int hint = -1)
{
    if (e < s) { msg = "endCpG < startCpG"; return 1; }
    if (s < 1) { msg = "startCpG < 1"; return 1; }
    if (e < 1) { msg = "endCpG < 1"; return 1; }
    // (loc2chrom admits nr_sites + 1 because an END may sit there; a block cannot START there: the reference reads one
    // element past its loci vector in that case — refused here, with the message of an unknown site)
    c1 = s > g.n_sites ? -1 : loc2chrom(g, s, hint);
    if (c1 < 0) { msg = "[ cpg_dict ] Could not find chromosome for site: " + std::to_string(s); return 2; }
    const int c2 = loc2chrom(g, e, c1);
    if (c2 < 0) { msg = "[ cpg_dict ] Could not find chromosome for site: " + std::to_string(e); return 2; }
    if (c1 != c2 && e - 1 != g.cum[c1]) { msg = "Cross chromosomes"; return 1; }
    return 0;
}

// Text of one shard of rows.  Raw storage (not std::string): resize() would zero-fill ~100 bytes per row first.
struct Shard {
    char* buf = nullptr; size_t len = 0;
    int64_t rows = 0;                                             // rows formatted (a BorderRows shard drops the blocks shorter than min_cpg)
    int64_t bad_line = -1; int bad_kind = 0; std::string msg;    // bad_line: position of the failing row among the shard's rows
    Shard() {}
    Shard(const Shard&) = delete;
    Shard& operator=(const Shard&) = delete;
    ~Shard() { free(buf); }
    void release() { free(buf); buf = nullptr; len = 0; }
};

// Where the rows come from.  ArrayRows: two arrays (startCpG, endCpG), every row is written.  BorderRows (round 4): the merged border lists
// of the regions as the segmentation leaves them (CSR: region r's ascending 1-based borders are flat[off[r] .. off[r+1])); row = a pair of
// consecutive borders (segment.py:154), kept when endCpG - startCpG >= min_cpg (segment.py:172-175 dump_result) — no (start, end) arrays
// are ever built (for hg19: 2 x 22 MB of numpy concatenations and a filter pass, ~25 ms of one Python thread around 15 ms of writing).
struct ArrayRows {
    const int64_t* s; const int64_t* e;
    struct Cursor { int64_t r; };
    Cursor at(int64_t r) const { return Cursor{r}; }
    bool get(Cursor& c, int64_t& a, int64_t& b) const { a = s[c.r]; b = e[c.r]; c.r++; return true; }
};
struct BorderRows {
    const int32_t* flat; const int64_t* off; int64_t n_regions; int64_t min_cpg;
    std::vector<int64_t> bcum;                                   // blocks before region r (unfiltered): [n_regions + 1]
    void index() { bcum.assign((size_t)n_regions + 1, 0); for (int64_t r = 0; r < n_regions; r++) bcum[(size_t)r + 1] = bcum[(size_t)r] + std::max<int64_t>(0, off[r + 1] - off[r] - 1); }
    int64_t total() const { return bcum.back(); }
    struct Cursor { int64_t reg, j; };
    Cursor at(int64_t r) const
    {
        const int64_t reg = (int64_t)(std::upper_bound(bcum.begin(), bcum.end(), r) - bcum.begin()) - 1;
        return Cursor{reg, r - bcum[(size_t)reg]};
    }
    bool get(Cursor& c, int64_t& a, int64_t& b) const
    {
        while (c.j >= off[c.reg + 1] - off[c.reg] - 1) { c.reg++; c.j = 0; }             // (regions without a block)
        const int32_t* p = flat + off[c.reg] + c.j;
        a = p[0]; b = p[1];
        c.j++;
        return b - a >= min_cpg;
    }
};

template <class Rows>
inline void format_range(const Genome& g, const Rows& rows, int64_t lo, int64_t hi, size_t max_name, Shard& out)
{
    const size_t row_cap = max_name + 4 * 20 + 5;              // name, four numbers of <= 20 digits, 4 tabs + newline
    out.buf = static_cast<char*>(malloc((size_t)(hi - lo) * row_cap + 1));
    if (!out.buf) { out.bad_line = lo; out.bad_kind = 3; out.msg = "out of memory"; return; }
    char* p = out.buf;
    int hint = -1;
    typename Rows::Cursor cur = rows.at(lo);
    for (int64_t r = lo; r < hi; r++) {
        int64_t sr, er;
        if (!rows.get(cur, sr, er)) continue;                   // (a block shorter than min_cpg: not a row)
        int c1 = 0;
        const int kind = check_row(g, sr, er, c1, out.msg, hint);
        hint = c1;
        if (kind) { out.bad_line = out.rows; out.bad_kind = kind; break; }      // (position among the shard's rows; the callers add the rows before it)
        const char* nm = g.names[c1];
        const size_t nl = strlen(nm);
        const uint64_t start = g.loci[sr - 1];
        const uint64_t end = (er == sr) ? start + 2 : (uint64_t)g.loci[er - 2] + 1;
        memcpy(p, nm, nl); p += nl;
        *p++ = '\t'; p = put_u64(p, start);
        *p++ = '\t'; p = put_u64(p, end);
        *p++ = '\t'; p = put_u64(p, (uint64_t)sr);
        *p++ = '\t'; p = put_u64(p, (uint64_t)er);
        *p++ = '\n';
        out.rows++;
    }
    out.len = (size_t)(p - out.buf);
}

// Writes the rows to `fp`.  Returns 0; 1 with err = "[wt add_loci] line N: ..." ; 2 with err = the cpg_dict message; 3 on I/O error.
// Rows before a failing row are written, as the reference's streaming loop would have.
// Shards of WG_ADD_SHARD rows are formatted by a pool of threads (next shard from a shared counter) while the calling
// thread writes finished shards in order: the wall time is max(formatting, writing), not their sum.
#define WG_ADD_SHARD 32768
template <class Rows>
inline int add_loci_rows(const Genome& g, const Rows& rows, int64_t n, FILE* fp, int threads, std::string& err, int64_t* n_written = nullptr)
{
    if (n_written) *n_written = 0;
    if (n <= 0) return 0;
    int T = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    const int64_t n_shards = (n + WG_ADD_SHARD - 1) / WG_ADD_SHARD;
    T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(T, 32), n_shards));
    size_t max_name = 0;
    for (int i = 0; i < g.n_chroms; i++) max_name = std::max(max_name, strlen(g.names[i]));
    std::vector<Shard> sh((size_t)n_shards);
    std::vector<std::atomic<int>> done((size_t)n_shards);
    for (auto& d : done) d.store(0, std::memory_order_relaxed);
    std::atomic<int64_t> next(0);
    std::atomic<int64_t> stop_at(n_shards);                       // no shard >= this needs formatting (a row before it failed)
    std::mutex mu;
    std::condition_variable cv;
    auto worker = [&]() {
        for (;;) {
            const int64_t k = next.fetch_add(1);
            if (k >= n_shards || k > stop_at.load()) break;
            format_range(g, rows, k * WG_ADD_SHARD, std::min<int64_t>(n, (k + 1) * WG_ADD_SHARD), max_name, sh[(size_t)k]);
            if (sh[(size_t)k].bad_line >= 0) {
                int64_t cur = stop_at.load();
                while (k < cur && !stop_at.compare_exchange_weak(cur, k)) {}
            }
            { std::lock_guard<std::mutex> lk(mu); done[(size_t)k].store(1, std::memory_order_release); }
            cv.notify_one();
        }
    };
    std::vector<std::thread> th;
    if (T > 1) for (int t = 0; t < T; t++) th.emplace_back(worker);
    int rc = 0;
    int64_t written = 0;
    for (int64_t k = 0; k < n_shards && rc == 0; k++) {
        if (T == 1) {
            format_range(g, rows, k * WG_ADD_SHARD, std::min<int64_t>(n, (k + 1) * WG_ADD_SHARD), max_name, sh[(size_t)k]);
        } else {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return done[(size_t)k].load(std::memory_order_acquire) != 0; });
        }
        Shard& x = sh[(size_t)k];
        if (x.bad_kind == 3) { err = x.msg; rc = 3; break; }
        if (x.len && fwrite(x.buf, 1, x.len, fp) != x.len) { err = "write failed"; rc = 3; break; }
        if (x.bad_line >= 0) {
            if (x.bad_kind == 1) err = "[wt add_loci] line " + std::to_string(written + x.bad_line) + ": " + x.msg;
            else err = x.msg;
            rc = x.bad_kind;
        }
        written += x.rows;
        x.release();
    }
    if (n_written) *n_written = written;
    if (rc != 0) { int64_t cur = stop_at.load(); while (cur > -1 && !stop_at.compare_exchange_weak(cur, -1)) {} }   // let the pool drain
    for (auto& t : th) t.join();
    if (fflush(fp) != 0 && rc == 0) { err = "write failed"; rc = 3; }
    return rc;
}
inline int add_loci(const Genome& g, const int64_t* s, const int64_t* e, int64_t n, FILE* fp, int threads, std::string& err)
{
    return add_loci_rows(g, ArrayRows{s, e}, n, fp, threads, err);
}

// The same rows into a regular FILE at byte offset `base` (descriptor `fd`): every shard is formatted by the pool first, the
// shard lengths give every shard its place in the file, and the pool then writes the shards side by side with pwrite — a
// single writer into the page cache tops out near 2 GB/s, a few of them do not.  Same return codes and messages as add_loci();
// rows before a failing row are written.
template <class Rows>
inline int add_loci_fd_rows(const Genome& g, const Rows& rows, int64_t n, int fd, int64_t base, int threads, std::string& err, int64_t* n_written = nullptr)
{
    if (n_written) *n_written = 0;
    if (n <= 0) return 0;
    int T = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    const int64_t n_shards = (n + WG_ADD_SHARD - 1) / WG_ADD_SHARD;
    T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(T, 32), n_shards));
    size_t max_name = 0;
    for (int i = 0; i < g.n_chroms; i++) max_name = std::max(max_name, strlen(g.names[i]));
    std::vector<Shard> sh((size_t)n_shards);
    std::atomic<int64_t> stop_at(n_shards);
    auto pool = [&](const std::function<void(int64_t)>& f) {
        std::atomic<int64_t> next(0);
        auto w = [&]() { for (int64_t k; (k = next.fetch_add(1)) < n_shards;) f(k); };
        if (T == 1) { w(); return; }
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back(w);
        for (auto& x : th) x.join();
    };
    pool([&](int64_t k) {
        if (k > stop_at.load()) return;
        format_range(g, rows, k * WG_ADD_SHARD, std::min<int64_t>(n, (k + 1) * WG_ADD_SHARD), max_name, sh[(size_t)k]);
        if (sh[(size_t)k].bad_line >= 0) {
            int64_t cur = stop_at.load();
            while (k < cur && !stop_at.compare_exchange_weak(cur, k)) {}
        }
    });
    const int64_t last = std::min<int64_t>(stop_at.load(), n_shards - 1);       // shards 0 .. last hold rows to write
    int rc = 0;
    std::vector<int64_t> off((size_t)last + 2, base);
    int64_t written = 0, before_last = 0;
    for (int64_t k = 0; k <= last; k++) {
        if (sh[(size_t)k].bad_kind == 3) { err = sh[(size_t)k].msg; return 3; }
        off[(size_t)k + 1] = off[(size_t)k] + (int64_t)sh[(size_t)k].len;
        before_last = written;
        written += sh[(size_t)k].rows;
    }
    if (n_written) *n_written = written;
    if (ftruncate(fd, off[(size_t)last + 1]) != 0) { err = "write failed"; return 3; }
    std::atomic<int> io_bad(0);
    pool([&](int64_t k) {
        if (k > last) return;
        const Shard& x = sh[(size_t)k];
        size_t done = 0;
        while (done < x.len) {
            const ssize_t w = pwrite(fd, x.buf + done, x.len - done, (off_t)(off[(size_t)k] + (int64_t)done));
            if (w <= 0) { io_bad.store(1); return; }
            done += (size_t)w;
        }
    });
    if (io_bad.load()) { err = "write failed"; return 3; }
    const Shard& b = sh[(size_t)last];
    if (b.bad_line >= 0) {
        if (b.bad_kind == 1) err = "[wt add_loci] line " + std::to_string(before_last + b.bad_line) + ": " + b.msg;
        else err = b.msg;
        rc = b.bad_kind;
    }
    return rc;
}
inline int add_loci_fd(const Genome& g, const int64_t* s, const int64_t* e, int64_t n, int fd, int64_t base, int threads, std::string& err)
{
    return add_loci_fd_rows(g, ArrayRows{s, e}, n, fd, base, threads, err);
}

}  // namespace wgadd
