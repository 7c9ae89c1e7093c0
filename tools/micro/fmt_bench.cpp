// The %.Nf formatter of csrc/table_io.h alone:  g++ -O3 -std=c++17 -pthread tools/micro/fmt_bench.cpp -o tools/micro/_build/fmt_bench && tools/micro/_build/fmt_bench
#include "../../wgbs_tools_amd/csrc/table_io.h"
#include <chrono>
#include <random>
int main() {
    const int n = 10000000;
    std::vector<double> v(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(0, 1);
    for (auto& x : v) x = u(g);
    std::vector<char> out((size_t)n * 8 + 1000);
    for (int rep = 0; rep < 3; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        char* p = out.data();
        for (int i = 0; i < n; i++) { p = wgtab::put_fixed(p, v[i], 2); *p++ = '\n'; }
        auto t1 = std::chrono::steady_clock::now();
        printf("%.1f ns per value (%ld bytes)\n", std::chrono::duration<double>(t1 - t0).count() / n * 1e9, (long)(p - out.data()));
    }
}
