// Wall time of wgadd::add_loci (formatting by a thread pool overlapped with the in-order write of ~115 MB of BED text).
//   g++ -O2 -pthread -o _build/add_loci_prof add_loci_prof.cpp && _build/add_loci_prof /dev/shm/x.bed
#include "../../wgbs_tools_amd/csrc/add_loci.h"
#include <chrono>
#include <random>
#include <unistd.h>
#include <fcntl.h>
#include <sys/mman.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const char* path = argc > 1 ? argv[1] : "/tmp/x.bed";
    const int64_t n = 28217448, nb = 2800000;
    std::mt19937_64 rng(1);
    std::vector<uint32_t> loci((size_t)n);
    uint32_t p = 0;
    for (auto& x : loci) { p += 2 + rng() % 198; x = p; }
    std::vector<int64_t> cum(25);
    for (int i = 0; i < 25; i++) cum[i] = n * (i + 1) / 25;
    std::vector<std::string> nm; std::vector<const char*> names;
    for (int i = 0; i < 25; i++) nm.push_back("chr" + std::to_string(i + 1));
    for (auto& s : nm) names.push_back(s.c_str());
    std::vector<int64_t> b{1};
    for (int64_t i = 0; i < nb; i++) b.push_back(1 + rng() % n);
    for (auto c : cum) b.push_back(c + 1);
    std::sort(b.begin(), b.end()); b.erase(std::unique(b.begin(), b.end()), b.end());
    std::vector<int64_t> s(b.begin(), b.end() - 1), e(b.begin() + 1, b.end());
    wgadd::Genome g{loci.data(), n, cum.data(), names.data(), 25};
    for (int rep = 0; rep < 6; rep++) {
        unlink(path);
        FILE* fp = fopen(path, "wb");
        std::string err;
        const double t0 = now();
        const int rc = wgadd::add_loci(g, s.data(), e.data(), (int64_t)s.size(), fp, 0, err);
        const double t1 = now();
        fclose(fp);
        printf("add_loci rc %d: %.1f ms\n", rc, (t1 - t0) * 1e3);
    }
    return 0;
}
