// add_loci.h — host side of the path's last step: blocks (startCpG, endCpG) -> BED rows
//     chrom \t start \t end \t startCpG \t endCpG \n
// with start = loci[startCpG-1], end = loci[endCpG-2]+1 (start+2 for an empty block), the chromosome looked up in the
// cumulative CpG counts, and the reference's validations in the reference's order.  Replaces the `add_loci` binary the
// reference pipes its blocks through (src/cpg2bed/add_loci.cpp:22-57, cpg_dict.cpp:118-131), without the 25 tabix
// processes that binary needs to load the loci.  Formatting is sharded over host threads; rows are written in order.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
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

inline char* put_u64(char* p, uint64_t v)
{
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

// chromosome index of a 1-based CpG index (cpg_dict.cpp:118-131): the first chromosome whose cumulative count is >= loc;
// nr_sites+1 (a non-inclusive end) belongs to the last chromosome; anything else is an error (-1)
inline int loc2chrom(const Genome& g, int64_t loc)
{
    const int64_t* e = g.cum + g.n_chroms;
    const int64_t* it = std::lower_bound(g.cum, e, loc);
    if (it != e) return (int)(it - g.cum);
    if (loc == g.n_sites + 1) return g.n_chroms - 1;
    return -1;
}

// Validates like add_loci.cpp:38-49.  Returns 0, or the 1-based position of the failing check with `line` / `msg` set.
inline int check_row(const Genome& g, int64_t s, int64_t e, int& c1, std::string& msg)
{
    if (e < s) { msg = "endCpG < startCpG"; return 1; }
    if (s < 1) { msg = "startCpG < 1"; return 1; }
    if (e < 1) { msg = "endCpG < 1"; return 1; }
    c1 = loc2chrom(g, s);
    if (c1 < 0) { msg = "[ cpg_dict ] Could not find chromosome for site: " + std::to_string(s); return 2; }
    const int c2 = loc2chrom(g, e);
    if (c2 < 0) { msg = "[ cpg_dict ] Could not find chromosome for site: " + std::to_string(e); return 2; }
    if (c1 != c2 && e - 1 != g.cum[c1]) { msg = "Cross chromosomes"; return 1; }
    return 0;
}

struct Shard { std::string text; int64_t bad_line = -1; int bad_kind = 0; std::string msg; };

inline void format_range(const Genome& g, const int64_t* s, const int64_t* e, int64_t lo, int64_t hi, Shard& out)
{
    out.text.resize((size_t)(hi - lo) * 96 + 16);
    char* base = &out.text[0];
    char* p = base;
    size_t cap = out.text.size();
    for (int64_t r = lo; r < hi; r++) {
        int c1 = 0;
        const int kind = check_row(g, s[r], e[r], c1, out.msg);
        if (kind) { out.bad_line = r; out.bad_kind = kind; break; }
        const char* nm = g.names[c1];
        const size_t nl = strlen(nm);
        if ((size_t)(p - base) + nl + 90 > cap) {               // long chromosome names: grow
            const size_t used = (size_t)(p - base);
            out.text.resize(cap * 2 + nl + 128);
            base = &out.text[0]; p = base + used; cap = out.text.size();
        }
        const uint64_t start = g.loci[s[r] - 1];
        const uint64_t end = (e[r] == s[r]) ? start + 2 : (uint64_t)g.loci[e[r] - 2] + 1;
        memcpy(p, nm, nl); p += nl;
        *p++ = '\t'; p = put_u64(p, start);
        *p++ = '\t'; p = put_u64(p, end);
        *p++ = '\t'; p = put_u64(p, (uint64_t)s[r]);
        *p++ = '\t'; p = put_u64(p, (uint64_t)e[r]);
        *p++ = '\n';
    }
    out.text.resize((size_t)(p - base));
}

// Writes the rows to `fp`.  Returns 0; 1 with err = "[wt add_loci] line N: ..." ; 2 with err = the cpg_dict message; 3 on I/O error.
// Rows before a failing row are written, as the reference's streaming loop would have.
inline int add_loci(const Genome& g, const int64_t* s, const int64_t* e, int64_t n, FILE* fp, int threads, std::string& err)
{
    if (n <= 0) return 0;
    int T = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(T, 64), n / 20000 + 1));
    std::vector<Shard> sh((size_t)T);
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) {
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        if (T == 1) format_range(g, s, e, lo, hi, sh[0]);
        else th.emplace_back(format_range, std::cref(g), s, e, lo, hi, std::ref(sh[(size_t)t]));
    }
    for (auto& x : th) x.join();
    for (int t = 0; t < T; t++) {
        if (!sh[(size_t)t].text.empty() && fwrite(sh[(size_t)t].text.data(), 1, sh[(size_t)t].text.size(), fp) != sh[(size_t)t].text.size()) {
            err = "write failed"; return 3;
        }
        if (sh[(size_t)t].bad_line >= 0) {
            if (sh[(size_t)t].bad_kind == 1) err = "[wt add_loci] line " + std::to_string(sh[(size_t)t].bad_line) + ": " + sh[(size_t)t].msg;
            else err = sh[(size_t)t].msg;
            fflush(fp);
            return sh[(size_t)t].bad_kind;
        }
    }
    if (fflush(fp) != 0) { err = "write failed"; return 3; }
    return 0;
}

}  // namespace wgadd
