#include "../../wgbs_tools_amd/csrc/stitch.h"
#include <cstdio>
using namespace wgstitch;
int main()
{
    // hg19-like: 25 regions
    const int64_t sizes[25] = {2266159,2211181,1800406,1737962,1644878,1555772,1446868,1330727,1283901,1232276,1227473,1216978,1047119,976022,932212,821501,738228,709875,537601,573024,437595,466461,1411719,539826,151};
    std::vector<int64_t> rs, re; int64_t pos = 1;
    for (int i = 0; i < 25; i++) { rs.push_back(pos); pos += sizes[i]; re.push_back(pos); }
    std::vector<int32_t> out(30000000); std::vector<int64_t> off(26); int64_t stats[8]; std::string err;
    // a stand-in chunk engine: a border every 16 sites (aligned to absolute multiples of 16, so neighbouring results agree and every junction stitches at once)
    std::vector<std::vector<int32_t>> flats; std::vector<std::vector<int64_t>> offs;
    BatchFn fn = [&](const std::vector<Sites>& items, BatchResult& res, std::string&) -> int {
        size_t total = 0;
        for (auto& it : items) total += (size_t)((it.second - it.first) / 16 + 3);
        flats.emplace_back(total); offs.emplace_back(items.size() + 1);
        std::vector<int32_t>& flat = flats.back(); std::vector<int64_t>& off = offs.back();
        int64_t w = 0;
        for (size_t i = 0; i < items.size(); i++) {
            off[i] = w;
            const int64_t s = items[i].first, e = items[i].second;
            flat[w++] = 0;
            for (int64_t x = (s / 16 + 1) * 16; x < e; x += 16) flat[w++] = (int32_t)(x - s);
            flat[w++] = (int32_t)(e - s);
        }
        off[items.size()] = w;
        res.set_csr(flat.data(), off.data(), items.size());
        return 0;
    };
    for (int rep = 0; rep < 5; rep++) {
        flats.clear(); offs.clear();
        auto t0 = std::chrono::steady_clock::now();
        int rc = segment_regions(rs.data(), re.data(), 25, 60000, fn, out.data(), (int64_t)out.size(), off.data(), stats, err);
        auto t1 = std::chrono::steady_clock::now();
        printf("rc %d wall %.0f us, batches(us) first %lld later %lld, total borders %lld\n", rc, std::chrono::duration<double, std::micro>(t1 - t0).count(), (long long)stats[5], (long long)stats[6], (long long)stats[7]);
    }
}
