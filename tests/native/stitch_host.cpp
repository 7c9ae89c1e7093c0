// tests/native/stitch_host.cpp — test shim: wgbs_tools_amd/csrc/stitch.h (the native chunk grid + junction stitching)
// driven by a Python callback as the chunk engine, so the CPU suite can check it against the reference driver's
// golden vectors without a GPU.  Build: g++ -O2 -std=c++17 -shared -fPIC.
#include "../../wgbs_tools_amd/csrc/stitch.h"
#include <cstdio>

extern "C" {
// callback: for `n` 1-based site ranges (starts/ends) fill CSR (out, off[n+1]) with ABSOLUTE borders; returns 0
typedef int (*engine_cb)(const int64_t* starts, const int64_t* ends, int64_t n, int64_t* out, int64_t cap, int64_t* off);

int stitch_segment_regions(const int64_t* rs, const int64_t* re, int64_t n_regions, int64_t chunk_size, engine_cb cb,
                           int32_t* borders_out, int64_t cap, int64_t* borders_off, int64_t* stats, char* err, size_t errlen, int speculate)
{
    // speculate: bit 0 = speculation on; bits 8.. = E > 0: EARLY DELIVERY with edges of E borders, as the GPU path offers it for the first
    // batch (BatchResult::n_lead / edges / finish) — the leading lists hold POISON until finish() is called, so a stitcher that reads a list
    // before it is "home" produces garbage and the comparison with the reference tree fails
    const int E = speculate >> 8;
    speculate &= 1;
    std::vector<int32_t> edges, truth;
    wgstitch::BatchFn fn = [&](const std::vector<wgstitch::Sites>& todo, wgstitch::BatchResult& res, std::string& msg) -> int {
        std::vector<int64_t> s(todo.size()), e(todo.size()), off(todo.size() + 1);
        int64_t c = 0;
        for (size_t i = 0; i < todo.size(); i++) { s[i] = todo[i].first; e[i] = todo[i].second; c += e[i] - s[i] + 1; }
        std::vector<int64_t> out((size_t)c);
        if (cb(s.data(), e.data(), (int64_t)todo.size(), out.data(), c, off.data()) != 0) { msg = "engine callback failed"; return -1; }
        res.owned.emplace_back(new int32_t[(size_t)c]);
        int32_t* flat = res.owned.back().get();
        for (size_t i = 0; i < todo.size(); i++)
            for (int64_t q = off[i]; q < off[i + 1]; q++) flat[(size_t)q] = (int32_t)(out[(size_t)q] - s[i]);   // relative, as the GPU path returns
        res.set_csr(flat, off.data(), todo.size());
        if (E > 0 && res.n_lead > 0 && res.n_lead <= (int64_t)todo.size()) {
            const int64_t nl = res.n_lead, lead_b = off[(size_t)nl];
            edges.assign((size_t)nl * 2 * E, -77777);
            for (int64_t i = 0; i < nl; i++) {
                const int64_t n = off[(size_t)i + 1] - off[(size_t)i], m = n < E ? n : E;
                for (int64_t q = 0; q < m; q++) {
                    edges[(size_t)((i * 2) * E + q)] = flat[(size_t)(off[(size_t)i] + q)];
                    edges[(size_t)((i * 2 + 1) * E + (E - m) + q)] = flat[(size_t)(off[(size_t)i] + n - m + q)];
                }
            }
            truth.assign(flat, flat + lead_b);
            for (int64_t q = 0; q < lead_b; q++) flat[(size_t)q] = -123456789 + (int32_t)(q % 1000);      // poison
            res.edges = edges.data(); res.edge_n = E;
            res.finish = [flat, lead_b, &truth](std::string&) -> int { for (int64_t q = 0; q < lead_b; q++) flat[(size_t)q] = truth[(size_t)q]; return 0; };
        }
        return 0;
    };
    std::string msg;
    int rc = wgstitch::segment_regions(rs, re, n_regions, chunk_size, fn, borders_out, cap, borders_off, stats, msg, speculate != 0);
    if (err && errlen) { snprintf(err, errlen, "%s", msg.c_str()); }
    return rc;
}
}
