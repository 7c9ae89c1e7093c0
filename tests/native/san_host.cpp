// Sanitizer harness for the host-side C++ of the library (TEST INFRASTRUCTURE): csrc/stitch.h (chunk grid, junction rehearsal,
// pairwise trees on the thread pool, flattening) and csrc/add_loci.h (BED rows), driven by a toy chunk engine that is a pure
// function of the site range — as the real DP is — so that junction patches share borders with their chunks or, where the toy
// makes them disagree, force the patch to double.  Built by tests/test_sanitizers_cpu.py three times (plain, ASan + UBSan,
// TSan); every build must print the same checksum lines.
//     san_host <threads> <out.bed>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <unistd.h>
#include <cmath>
#include "stitch.h"
#include "add_loci.h"
#include "table_io.h"

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
    {   // the stitcher's open-addressing table (round 5) against std::map: inserts that force it to grow, updates, hits and misses, then concurrent readers
        wgstitch::SiteTable<int64_t> tab(4);
        std::map<wgstitch::Sites, int64_t> ref;
        uint64_t x = 12345;
        long bad = 0;
        for (int i = 0; i < 60000; i++) {
            x = mix(x + (uint64_t)i);
            const wgstitch::Sites k((int64_t)(x % 5000) + 1, (int64_t)(x % 5000) + 1 + (int64_t)((x >> 20) % 300));
            if ((x >> 40) % 3 == 0) { tab[k] = (int64_t)i; ref[k] = (int64_t)i; }
            else {
                const int64_t* f = tab.find(k);
                auto it = ref.find(k);
                bad += (f == nullptr) != (it == ref.end()) || (f && *f != it->second) || tab.count(k) != (ref.count(k) != 0);
            }
        }
        std::vector<std::thread> th;
        std::vector<long> miss((size_t)4, 0);
        for (int t = 0; t < 4; t++) th.emplace_back([&, t] { for (auto& kv : ref) { const int64_t* f = tab.find(kv.first); miss[(size_t)t] += !f || *f != kv.second; } });
        for (auto& t : th) t.join();
        printf("sitetable: %zu keys, mismatches %ld, concurrent misses %ld\n", ref.size(), bad, miss[0] + miss[1] + miss[2] + miss[3]);
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
    {   // the same genome's rows straight from border lists (three regions = the chromosomes, one of them without a block), min_cpg 1 and 3
        std::vector<int32_t> flat;
        std::vector<int64_t> off{0};
        const int64_t lo[3] = {1, 701, 1501}, hi[3] = {701, 1501, 2001};
        for (int r = 0; r < 3; r++) {
            if (r != 1) for (int64_t b = lo[r]; b < hi[r]; b += 1 + (int64_t)(mix((uint64_t)b) % 9)) flat.push_back((int32_t)b);
            if (r != 1) flat.push_back((int32_t)hi[r]);
            off.push_back((int64_t)flat.size());
        }
        for (int64_t mc : {(int64_t)1, (int64_t)3}) {
            wgadd::BorderRows rows{flat.data(), off.data(), 3, mc, {}};
            rows.index();
            FILE* nul = fopen("/dev/null", "wb");
            std::string m;
            int64_t written = 0;
            const int r3 = wgadd::add_loci_rows(g, rows, rows.total(), nul, atoi(argv[1]), m, &written);
            fclose(nul);
            printf("add_loci from borders, min_cpg %lld: %lld of %lld blocks written, rc %d %s\n", (long long)mc, (long long)written, (long long)rows.total(), r3, m.c_str());
        }
    }
    // and its refusals
    const int64_t bad_s[3] = {5, 0, 1999}, bad_e[3] = {3, 4, 2003};
    for (int i = 0; i < 3; i++) {
        FILE* nul = fopen("/dev/null", "wb");
        std::string m;
        const int r2 = wgadd::add_loci(g, bad_s + i, bad_e + i, 1, nul, 1, m);
        fclose(nul);
        printf("add_loci bad row %d: rc %d %s\n", i, r2, m.c_str());
    }
    // the block tools' text paths (table_io.h): a table with a header, comments, NA fields and no last newline through the parser,
    // the sharded writers (file: pwrite side by side; a pipe-like descriptor: in order) and the number formatter
    {
        std::string text = "chr\tstart\tend\tstartCpG\tendCpG\n# c\n\n";
        const int64_t n_rows = 40000;
        for (int64_t i = 0; i < n_rows; i++) {
            char row[96];
            if (i % 97 == 5) snprintf(row, sizeof row, "chr%d\t%lld\t%lld\tNA\t\n", (int)(1 + i % 22), (long long)(10 * i), (long long)(10 * i + 7));
            else snprintf(row, sizeof row, "chr%d\t%lld\t%lld\t%lld\t%lld\textra\n", (int)(1 + i % 22), (long long)(10 * i), (long long)(10 * i + 7), (long long)(1 + 2 * i), (long long)(3 + 2 * i));
            text += row;
        }
        text.pop_back();
        std::vector<int64_t> lo((size_t)n_rows + 8), sc((size_t)n_rows + 8), ec((size_t)n_rows + 8);
        std::vector<int32_t> l3((size_t)n_rows + 8);
        std::vector<uint8_t> na((size_t)n_rows + 8);
        int64_t got = 0;
        const int prc = wgtab::parse_blocks(text.data(), (int64_t)text.size(), -1, n_rows + 8, lo.data(), l3.data(), sc.data(), ec.data(), na.data(), &got);
        int64_t n_na = 0;
        for (int64_t i = 0; i < got; i++) n_na += na[(size_t)i];
        printf("parse_blocks: rc %d rows %lld na %lld, row 1 = [%lld, %lld)\n", prc, (long long)got, (long long)n_na, (long long)sc[1], (long long)ec[1]);
        const std::string bad = "chr1\t1\t2\t3.5\t4\n";
        printf("parse_blocks on a float field: rc %d\n", wgtab::parse_blocks(bad.data(), (int64_t)bad.size(), -1, 4, lo.data(), l3.data(), sc.data(), ec.data(), na.data(), &got));
        wgtab::parse_blocks(text.data(), (int64_t)text.size(), -1, n_rows + 8, lo.data(), l3.data(), sc.data(), ec.data(), na.data(), &got);
        const int64_t n_cols = 5;
        std::vector<double> vals((size_t)(got * n_cols));
        for (size_t i = 0; i < vals.size(); i++) vals[i] = (mix(i) % 13 == 0) ? std::nan("") : (mix(i) % 11 == 0 ? 1.5e300 : (double)(mix(i) % 100001) / 100000.0);
        const wgtab::Rows R = {text.data(), lo.data(), l3.data(), sc.data(), ec.data(), na.data()};
        std::string tpath = std::string(argv[2]) + ".table";
        uint64_t h = 1469598103934665603ULL;
        for (int pass = 0; pass < 2; pass++) {                                // 0: regular file, 1: in order (the way a pipe is written)
            const int fd = open(tpath.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
            std::string m;
            const int wrc = wgtab::write_table(fd, pass ? -1 : 0, R, got, vals.data(), n_cols, n_cols, 3, atoi(argv[1]), m);
            close(fd);
            FILE* f = fopen(tpath.c_str(), "rb");
            uint64_t hh = 1469598103934665603ULL;
            int ch;
            long bytes = 0;
            while ((ch = fgetc(f)) != EOF) { hh = (hh ^ (uint64_t)ch) * 1099511628211ULL; bytes++; }
            fclose(f);
            printf("write_table pass %d: rc %d %s, %ld bytes, checksum %016llx%s\n", pass, wrc, m.c_str(), bytes, (unsigned long long)hh, pass && hh != h ? " DIFFERS" : "");
            h = hh;
        }
        std::vector<uint16_t> mc((size_t)got * 2);
        for (int64_t i = 0; i < got; i++) { mc[(size_t)(2 * i + 1)] = (uint16_t)(mix((uint64_t)i) % 65536); mc[(size_t)(2 * i)] = (uint16_t)(mc[(size_t)(2 * i + 1)] * (mix((uint64_t)i + 7) % 101) / 100); }
        const int fd = open(tpath.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
        std::string m;
        const int brc = wgtab::write_bedgraph<uint16_t>(fd, 0, R, got, mc.data(), atoi(argv[1]), m);
        close(fd);
        printf("write_bedgraph: rc %d %s\n", brc, m.c_str());
        // `convert -L`: a BED table with a text, an integer and a strand column, unknown chromosomes, blank lines
        std::string bed = "\n";
        const char* cn[3] = {"chr1", "chr2", "chrX"};
        const int64_t nb = 50000;
        for (int64_t i = 0; i < nb; i++) {
            char row[128];
            snprintf(row, sizeof row, "%s\t%lld\t%lld\t%s%lld\t%lld\t%c\n", i % 211 == 3 ? "chrUn" : cn[i % 3], (long long)(100 + 7 * i), (long long)(300 + 7 * i),
                     i % 5 ? "n" : "NA", (long long)(i % 5 ? i : 0), (long long)(i % 1000), i % 2 ? '+' : '-');
            if (i % 5 == 0) { std::string r2 = row; const size_t k = r2.find("NA0"); r2.replace(k, 3, "NA"); bed += r2; } else bed += row;
        }
        std::vector<int64_t> blo((size_t)nb + 4), bs((size_t)nb + 4), be((size_t)nb + 4);
        std::vector<int32_t> bl3((size_t)nb + 4), brl((size_t)nb + 4), bci((size_t)nb + 4);
        int64_t bn = 0;
        int32_t bw = 0;
        const int brc2 = wgtab::parse_bed(bed.data(), (int64_t)bed.size(), nb + 4, cn, 3, blo.data(), bl3.data(), brl.data(), bci.data(), bs.data(), be.data(), &bn, &bw);
        int64_t unknown = 0;
        for (int64_t i = 0; i < bn; i++) unknown += bci[(size_t)i] < 0;
        printf("parse_bed: rc %d rows %lld width %d unknown %lld\n", brc2, (long long)bn, (int)bw, (long long)unknown);
        const std::string fl = "chr1\t1\t2\t0.50\n";          // comes back as 0.5: not this path's
        printf("parse_bed on a float column: rc %d\n", wgtab::parse_bed(fl.data(), (int64_t)fl.size(), 4, cn, 3, blo.data(), bl3.data(), brl.data(), bci.data(), bs.data(), be.data(), &bn, &bw));
        wgtab::parse_bed(bed.data(), (int64_t)bed.size(), nb + 4, cn, 3, blo.data(), bl3.data(), brl.data(), bci.data(), bs.data(), be.data(), &bn, &bw);
        std::vector<int64_t> cs((size_t)bn), ce((size_t)bn);
        for (int64_t i = 0; i < bn; i++) { cs[(size_t)i] = bci[(size_t)i] < 0 ? 0 : 1 + i; ce[(size_t)i] = bci[(size_t)i] < 0 ? 0 : 4 + i; }
        const int fd2 = open(tpath.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
        const int arc = wgtab::write_annotated_bed(fd2, 0, bed.data(), blo.data(), bl3.data(), brl.data(), cs.data(), ce.data(), bn, atoi(argv[1]), m);
        close(fd2);
        FILE* f2 = fopen(tpath.c_str(), "rb");
        uint64_t h2 = 1469598103934665603ULL;
        int ch2;
        long bytes2 = 0;
        while ((ch2 = fgetc(f2)) != EOF) { h2 = (h2 ^ (uint64_t)ch2) * 1099511628211ULL; bytes2++; }
        fclose(f2);
        printf("write_annotated_bed: rc %d %s, %ld bytes, checksum %016llx\n", arc, m.c_str(), bytes2, (unsigned long long)h2);
        unlink(tpath.c_str());
    }
    return 0;
}
