// Sanitizer harness for the host-side C++ of the library (TEST INFRASTRUCTURE): csrc/stitch.h (chunk grid, junction rehearsal,
// pairwise trees on the thread pool, flattening) and csrc/add_loci.h (BED rows), driven by a toy chunk engine that is a pure
// function of the site range — as the real DP is — so that junction patches share borders with their chunks or, where the toy
// makes them disagree, force the patch to double.  Built by tests/test_sanitizers_cpu.py three times (plain, ASan + UBSan,
// TSan); every build must print the same checksum lines.
//     san_host <threads> <out.bed>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "stitch.h"
#include "add_loci.h"

static uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

// borders of the "DP" over sites [a, b): a, b, and every x in between that the toy rule likes.  The rule looks at x alone, except
// near the START of a range (first `fuzzy` sites), where it also depends on where the range began — like the real DP, whose
// first borders depend on its left edge — so first-attempt patches do not always agree with the left chunk.
static void toy_borders(int64_t a, int64_t b, int fuzzy, std::vector<int32_t>& out)
{
    out.clear();
    out.push_back(0);
    for (int64_t x = a + 1; x < b; x++) {
        const bool near = x - a <= fuzzy;
        const uint64_t h = near ? mix((uint64_t)x * 31 + (uint64_t)(a % 97)) : mix((uint64_t)x);
        if (h % 9 == 0) out.push_back((int32_t)(x - a));
    }
    out.push_back((int32_t)(b - a));
}

static uint64_t run_world(int n_regions, int64_t region_len, int64_t chunk, int fuzzy, bool speculate, std::vector<int64_t>& starts_out,
                          std::vector<int64_t>& ends_out)
{
    std::vector<int64_t> rs((size_t)n_regions), re((size_t)n_regions);
    int64_t pos = 1;
    for (int r = 0; r < n_regions; r++) { rs[(size_t)r] = pos; pos += region_len + 37 * r; re[(size_t)r] = pos; }
    int64_t n_batches = 0;
    wgstitch::BatchFn fn = [&](const std::vector<wgstitch::Sites>& todo, wgstitch::BatchResult& res, std::string&) -> int {
        res.ptr.resize(todo.size()); res.cnt.resize(todo.size());
        std::vector<int32_t> tmp;
        for (size_t i = 0; i < todo.size(); i++) {
            toy_borders(todo[i].first, todo[i].second, fuzzy, tmp);
            std::unique_ptr<int32_t[]> own(new int32_t[tmp.size()]);
            memcpy(own.get(), tmp.data(), tmp.size() * 4);
            res.ptr[i] = own.get(); res.cnt[i] = (int64_t)tmp.size();
            res.owned.push_back(std::move(own));
        }
        n_batches++;
        return 0;
    };
    const int64_t cap = pos + n_regions;
    std::vector<int32_t> borders((size_t)cap);
    std::vector<int64_t> off((size_t)n_regions + 1);
    int64_t stats[8] = {0};
    std::string err;
    const int rc = wgstitch::segment_regions(rs.data(), re.data(), n_regions, chunk, fn, borders.data(), cap, off.data(), stats, err, speculate);
    if (rc != 0) { printf("world regions=%d len=%lld chunk=%lld fuzzy=%d spec=%d: rc %d (%s)\n", n_regions, (long long)region_len, (long long)chunk, fuzzy, (int)speculate, rc, err.c_str()); return 0; }
    uint64_t h = 1469598103934665603ULL;
    for (int r = 0; r < n_regions; r++) {
        for (int64_t q = off[(size_t)r]; q < off[(size_t)r + 1]; q++) { h = (h ^ (uint64_t)(uint32_t)borders[(size_t)q]) * 1099511628211ULL; }
        for (int64_t q = off[(size_t)r]; q + 1 < off[(size_t)r + 1]; q++) { starts_out.push_back(borders[(size_t)q]); ends_out.push_back(borders[(size_t)q + 1]); }
    }
    printf("world regions=%d len=%lld chunk=%lld fuzzy=%d spec=%d: %lld borders, %lld chunks, %lld patch DPs, checksum %016llx\n", n_regions,
           (long long)region_len, (long long)chunk, fuzzy, (int)speculate, (long long)off[(size_t)n_regions], (long long)stats[0], (long long)stats[1],
           (unsigned long long)h);
    return h;
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: san_host <threads> <out.bed>\n"); return 2; }
    setenv("WGBSSEG_STITCH_THREADS", argv[1], 1);
    std::vector<int64_t> s, e;
    for (int rep = 0; rep < 3; rep++) {                                      // the pool is reused across calls
        std::vector<int64_t> s1, e1;
        run_world(7, 40000, 5000, 0, true, s1, e1);                          // every first attempt agrees
        run_world(5, 30000, 3000, 40, true, s1, e1);                         // left edges disagree: doubling, follow-up batches
        run_world(5, 30000, 3000, 40, false, s1, e1);                        // the same without speculation
        run_world(1, 9000, 700, 200, true, s1, e1);                          // one region, patches that grow past a chunk
        run_world(3, 500, 60000, 0, true, s, e);                             // single-chunk regions (s, e: the blocks written below)
    }
    // BED rows of the last world's blocks on a toy genome of 3 chromosomes
    const int64_t n_sites = 2000;
    std::vector<uint32_t> loci((size_t)n_sites);
    for (int64_t i = 0; i < n_sites; i++) loci[(size_t)i] = (uint32_t)(100 + 13 * i + (mix((uint64_t)i) % 7));
    const int64_t cum[3] = {700, 1500, 2000};
    const char* names[3] = {"chr1", "chr2", "chrX"};
    wgadd::Genome g = {loci.data(), n_sites, cum, names, 3};
    std::vector<int64_t> bs, be;
    for (size_t i = 0; i < s.size(); i++) if (e[i] <= n_sites + 1 && wgadd::loc2chrom(g, s[i]) == wgadd::loc2chrom(g, e[i] - 1)) { bs.push_back(s[i]); be.push_back(e[i]); }
    FILE* fp = fopen(argv[2], "wb");
    if (!fp) return 3;
    std::string err;
    const int rc = wgadd::add_loci(g, bs.data(), be.data(), (int64_t)bs.size(), fp, atoi(argv[1]), err);
    fclose(fp);
    printf("add_loci: %zu rows, rc %d %s\n", bs.size(), rc, err.c_str());
    // and its refusals
    const int64_t bad_s[3] = {5, 0, 1999}, bad_e[3] = {3, 4, 2003};
    for (int i = 0; i < 3; i++) {
        FILE* nul = fopen("/dev/null", "wb");
        std::string m;
        const int r2 = wgadd::add_loci(g, bad_s + i, bad_e + i, 1, nul, 1, m);
        fclose(nul);
        printf("add_loci bad row %d: rc %d %s\n", i, r2, m.c_str());
    }
    return 0;
}
