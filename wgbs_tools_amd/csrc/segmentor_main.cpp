// `segmentor` on an MI355X: the executable the reference's driver starts per chunk
//     tabix rev.CpG.bed.gz chr:s-e | cut -f2 | segmentor a.beta b.beta ... -s START -n NR_SITES -max_cpg M -ps P -max_bp B
// (src/python/segment.py:41-59; src/segment_betas/main.cpp:40-112) with the same command line, the same standard input (one locus
// per line) and the same standard output (the block borders of the chunk, "%d " each, one line: segmentor.cpp:30-34), over the C ABI
// of libwgbsseg.so.  Putting this file's binary where the reference expects `src/segment_betas/segmentor` runs an unmodified
// wgbs_tools on the GPU — one process, one context and one upload per chunk, as the reference does it: the compatibility path.
// The fast path is the library itself behind `segment_process`.
//
// Behaviour kept from the reference's main / read_beta_file / load_dists:
//   * fewer than five arguments: the usage line on stderr, exit status 255 (`return -1`);
//   * -s and -n are mandatory, -max_cpg defaults to 1000, -ps to 1, -max_bp to 0; an option's value is the token after its FIRST
//     occurrence; numbers are read the way std::stoul / std::stof read them (leading digits count, the rest is ignored);
//   * every argument of six or more characters that ends in ".beta" is a beta file, in command-line order;
//   * bytes [2 START, 2 START + 2 NR_SITES) of every file; #meth > #cov anywhere: "invalid data, i = ..." and a failure status;
//   * NR_SITES lines of loci on stdin, any other count: "Error: nr_sites != number of loci: ..." and a failure status.
// Deliberately different: a failure is an exit status of 1 with the message on stderr (the reference throws an int or a string
// literal that nothing catches: SIGABRT); -max_bp 0 (never passed by segment.py, which always sets it: the reference then compares
// loci it never read, segmentor.cpp:38,114) and a file shorter than the requested range (the reference then computes on
// uninitialised memory) are refused with a message.
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/wgbsseg.h"

namespace {

struct Options {
    bool has_start = false, has_sites = false;
    unsigned long start = 0, sites = 0, max_cpg = 1000, max_bp = 0;
    float pseudo_count = 1.0f;
    std::vector<std::string> betas;
};

[[noreturn]] void fail(const std::string& msg)
{
    fprintf(stderr, "%s\n", msg.c_str());
    exit(1);
}

// the value std::stoul / std::stof would return for `text`, or failure when they would throw
unsigned long as_ulong(const char* opt, const std::string& text)
{
    errno = 0;
    char* end = nullptr;
    const unsigned long v = strtoul(text.c_str(), &end, 10);
    if (end == text.c_str()) fail(std::string("invalid value for ") + opt + ": " + text);
    if (errno == ERANGE) fail(std::string("value out of range for ") + opt + ": " + text);
    return v;
}

float as_float(const char* opt, const std::string& text)
{
    errno = 0;
    char* end = nullptr;
    const float v = strtof(text.c_str(), &end);
    if (end == text.c_str()) fail(std::string("invalid value for ") + opt + ": " + text);
    if (errno == ERANGE) fail(std::string("value out of range for ") + opt + ": " + text);
    return v;
}

Options read_command_line(int argc, char** argv)
{
    Options o;
    std::vector<std::string> tok(argv + 1, argv + argc);
    // value of an option = the token behind its first occurrence ("" when there is none: the option counts as absent)
    auto value = [&](const char* name) -> const std::string* {
        for (size_t i = 0; i + 1 < tok.size(); i++) if (tok[i] == name) return &tok[i + 1];
        return nullptr;
    };
    struct { const char* name; unsigned long* dst; bool* seen; } ints[] = {
        {"-s", &o.start, &o.has_start}, {"-n", &o.sites, &o.has_sites}, {"-max_cpg", &o.max_cpg, nullptr}, {"-max_bp", &o.max_bp, nullptr}};
    for (auto& it : ints) {
        const std::string* v = value(it.name);
        if (!v || v->empty()) continue;
        *it.dst = as_ulong(it.name, *v);
        if (it.seen) *it.seen = true;
    }
    if (const std::string* v = value("-ps")) if (!v->empty()) o.pseudo_count = as_float("-ps", *v);
    if (!o.has_start) fail("start sites (-s) must be provided");
    if (!o.has_sites) fail("number of sites (-n) must be provided");
    for (const auto& t : tok)
        if (t.size() >= 6 && t.compare(t.size() - 5, 5, ".beta") == 0) o.betas.push_back(t);
    return o;
}

// loci of the chunk: one integer per line of stdin, as std::stoi reads a line (leading blanks, sign, digits; the rest ignored)
std::vector<uint32_t> read_loci(unsigned long want)
{
    std::vector<uint32_t> loci;
    loci.reserve(want);
    char* line = nullptr;
    size_t cap = 0;
    while (getline(&line, &cap, stdin) >= 0) {
        errno = 0;
        char* end = nullptr;
        const long v = strtol(line, &end, 10);
        if (end == line) fail("invalid locus on standard input: " + std::string(line));
        if (errno == ERANGE || v > 0x7fffffffL || v < -0x7fffffffL - 1) fail("locus out of range on standard input: " + std::string(line));
        loci.push_back((uint32_t)(int32_t)v);
    }
    free(line);
    if (loci.size() != want)
        fail("Error: nr_sites != number of loci: " + std::to_string(want) + " != " + std::to_string(loci.size()) + ". Try different chunck size!");
    return loci;
}

void read_rows(const Options& o, std::vector<uint8_t>& rows, size_t pitch)
{
    for (size_t s = 0; s < o.betas.size(); s++) {
        FILE* f = fopen(o.betas[s].c_str(), "rb");
        if (!f) fail("cannot open " + o.betas[s] + ": " + strerror(errno));
        uint8_t* row = rows.data() + s * pitch;
        const bool sought = fseeko(f, (off_t)(2 * o.start), SEEK_SET) == 0;
        const size_t got = sought ? fread(row, 1, 2 * o.sites, f) : 0;
        fclose(f);
        if (got != 2 * o.sites)
            fail("beta path: " + o.betas[s] + ": the file ends before site " + std::to_string(o.start + o.sites) + " (" + std::to_string(got / 2) + " of " +
                 std::to_string(o.sites) + " sites read)");
        for (unsigned long i = 0; i < o.sites; i++)            // (segmentor.cpp:179-188; the library would find it too, this is the reference's wording)
            if (row[2 * i] > row[2 * i + 1]) {
                fprintf(stderr, "invalid data, i = %lu. data: %d, %d\nbeta path: %s\n", i, (int)row[2 * i], (int)row[2 * i + 1], o.betas[s].c_str());
                exit(1);
            }
    }
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc < 6) {
        fprintf(stderr, "Usage: segment BETA_PATH [BETA_PATH...] -s START -n NR_SITES  [-m max_cpg] [-ps PSEUDO_COUNT]\n");
        return -1;
    }
    const Options o = read_command_line(argc, argv);
    if (o.sites == 0) {                 // the traceback of an empty chunk is its start alone
        printf("0 \n");
        return 0;
    }
    if (o.max_bp == 0) fail("-max_bp must be at least 1 (wgbstools segment always passes it: --max_bp, default 2000)");
    if (o.betas.empty()) fail("no beta file on the command line");
    if (o.sites > 0x7fffffffUL / 2 || o.max_cpg > 0xffffffffUL || o.max_bp > 0xffffffffUL) fail("-n, -max_cpg or -max_bp out of range");

    const size_t pitch = (2 * o.sites + 255) / 256 * 256 + 256;
    std::vector<uint8_t> rows(o.betas.size() * pitch, 0);
    read_rows(o, rows, pitch);
    const std::vector<uint32_t> loci = read_loci(o.sites);

    wgbsseg_params p;
    memset(&p, 0, sizeof(p));
    p.pseudo_count = o.pseudo_count;
    p.max_cpg = (uint32_t)o.max_cpg;
    p.max_bp = (uint32_t)o.max_bp;
    const int64_t start0 = 0;
    const int32_t len = (int32_t)o.sites;
    std::vector<int32_t> borders(o.sites + 2);
    int64_t off[2] = {0, 0};
    char err[512] = "";
    const char* dev = getenv("WGBSSEG_DEVICE");
    const int rc = wgbsseg_segment_chunks_host(rows.data(), (int64_t)o.betas.size(), (int64_t)pitch, (int64_t)o.sites, loci.data(), &start0, &len, 1, &p,
                                               dev ? atoi(dev) : 0, borders.data(), (int64_t)borders.size(), off, err, sizeof(err));
    if (rc != 0) fail(std::string("segmentor: ") + (err[0] ? err : "the library failed") + " (status " + std::to_string(rc) + ")");
    for (int64_t i = off[0]; i < off[1]; i++) printf("%d ", borders[(size_t)i]);
    printf("\n");
    return 0;
}
