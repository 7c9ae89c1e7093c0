// tests/native/exact_host.cpp — test shim: the HOST build of wgbs_tools_amd/csrc/exact_log2.h, exported so that
// tests can compare it with the live libm (oracle/libm_probe.c) and with the oracle's per-sample term.
// Build: g++ -O2 -ffp-contract=off -shared -fPIC (see __graft_entry__.build / tests/conftest.py).
#include "../../wgbs_tools_amd/csrc/exact_log2.h"
#include <thread>
#include <vector>

static wg_log_tables make_tab() { wg_log_tables t = WG_LOG_TABLES_INIT; wg_tables_finish(&t); return t; }
static const wg_log_tables g_tab = make_tab();
static wg_fast_tables make_fast()
{
    wg_fast_tables f;
    memcpy(f.f_tab, g_tab.f_tab, sizeof(f.f_tab)); memcpy(f.d_fast, g_tab.d_fast, sizeof(f.d_fast));
    return f;
}
static const wg_fast_tables g_fast = make_fast();
// k-scaled lookup tables of the narrow scoring kernel, all WG_KY_KMIN + 1 rows (the kernel builds wg_lookup_rows() of them)
struct ks_tables { wg_d2 iy[(WG_KY_KMIN + 1) * 16], ky[(WG_KY_KMIN + 1) * 64]; };
static ks_tables make_ks()
{
    ks_tables t;
    wg_log_tables raw = WG_LOG_TABLES_INIT;                      // (the builders re-centre the fast-log2 interval themselves)
    for (int x = 0; x < (WG_KY_KMIN + 1) * 16; x++) t.iy[x] = wg_ks_iy_entry(&raw, WG_KY_KMIN + 1, x);
    for (int x = 0; x < (WG_KY_KMIN + 1) * 64; x++) t.ky[x] = wg_ks_ky_entry(&raw, WG_KY_KMIN + 1, x);
    return t;
}
static const ks_tables g_ks = make_ks();
static const wg_d2* const g_iys0 = g_ks.iy + WG_KY_KMIN * 16;        // rows k = 0
static const wg_d2* const g_kys0 = g_ks.ky + WG_KY_KMIN * 64;

template <class F> static void par_for(uint64_t count, int threads, F f)
{
    if (threads < 1) threads = 1;
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([=]() { f(count * t / threads, count * (t + 1) / threads); });
    for (auto& x : th) x.join();
}

extern "C" {
void exact_log2f_fill(uint32_t first, uint64_t count, uint32_t* out, int threads)
{
    par_for(count, threads, [=](uint64_t a, uint64_t b) {
        for (uint64_t q = a; q < b; q++) out[q] = wg_f2u(wg_log2f(wg_u2f(first + (uint32_t)q), g_tab.f_tab));
    });
}
void exact_log2f_nofma_fill(uint32_t first, uint64_t count, uint32_t* out, int threads)
{
    par_for(count, threads, [=](uint64_t a, uint64_t b) {
        for (uint64_t q = a; q < b; q++) out[q] = wg_f2u(wg_log2f_nofma(wg_u2f(first + (uint32_t)q), g_tab.f_tab));
    });
}
void fast_log2_1mp_fill(uint32_t first, uint64_t count, uint64_t* out, int threads)
{
    par_for(count, threads, [=](uint64_t a, uint64_t b) {
        for (uint64_t q = a; q < b; q++) out[q] = wg_d2u(wg_fast_log2(1.0 - (double)wg_u2f(first + (uint32_t)q), g_tab.d_fast));
    });
}
void exact_sample_terms_plain(const float* nmeth, const float* ntotal, int64_t count, float pc, float* out)
{
    float pc2 = pc + pc;
    for (int64_t q = 0; q < count; q++) out[q] = wg_sample_term_plain(nmeth[q], ntotal[q], pc, pc2, &g_tab);
}
// largest distance, in ulps of the double sum, between the fast-path sum s' and the exact sum s of wg_sample_term's
// second term (the quantity its 6-ulp bound / 16-ulp guard band is about)
uint64_t sum_ulp_gap(const float* nmeth, const float* ntotal, int64_t count, float pc)
{
    const float pc2 = pc + pc;
    uint64_t mx = 0;
    for (int64_t q = 0; q < count; q++) {
        const float m = nmeth[q], t = ntotal[q];
        if (t == 0.0f) continue;
        const float p = (m + pc) / (t + pc2);
        float ll = 0.0f;
        if (!(p > 0.0f)) continue;                                 // the fast form returns +0 here without any log
        ll += m * wg_log2f_normal(p, g_tab.f_tab);
        const float df = t - m;
        if (!(p < 1.0f) || df == 0.0f) continue;
        const double xx = 1.0 - (double)p;
        const double s1 = (double)ll + (double)df * wg_fast_log2(xx, g_tab.d_fast);
        const double s0 = (double)ll + (double)df * wg_log2(xx, g_tab.d_tab, g_tab.d_tab2);
        const uint64_t a = wg_d2u(s1), b = wg_d2u(s0);
        const uint64_t d = a > b ? a - b : b - a;
        if (d > mx) mx = d;
    }
    return mx;
}
void exact_log2_1mp_fill(uint32_t first, uint64_t count, uint64_t* out, int threads)
{
    par_for(count, threads, [=](uint64_t a, uint64_t b) {
        for (uint64_t q = a; q < b; q++)
            out[q] = wg_d2u(wg_log2(1.0 - (double)wg_u2f(first + (uint32_t)q), g_tab.d_tab, g_tab.d_tab2));
    });
}
// The k-scaled forms against the originals, for `count` consecutive floats p from `first`: number of mismatches of
// wg_fast_log2_ks(1 - p) vs wg_fast_log2(1 - p) plus those of wg_log2f_ks(p) vs wg_log2f(p) (the libm restatement).
// Arguments whose exponent lies below the tables' first row are skipped and counted in *skipped.
uint64_t exact_ks_mismatches(uint32_t first, uint64_t count, int threads, uint64_t* skipped)
{
    std::vector<uint64_t> bad((size_t)(threads < 1 ? 1 : threads), 0), skip(bad.size(), 0);
    std::vector<std::thread> th;
    const int T = (int)bad.size();
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t]() {
            for (uint64_t q = count * t / T; q < count * (t + 1) / T; q++) {
                const float p = wg_u2f(first + (uint32_t)q);
                const double x = 1.0 - (double)p;
                if (x > 0.6875 * 0x1p-23) { if (wg_d2u(wg_fast_log2_ks<true>(x, g_kys0)) != wg_d2u(wg_fast_log2(x, g_tab.d_fast))) bad[(size_t)t]++; }
                else skip[(size_t)t]++;
                if (p > 0.69921875f * 0x1p-23f) { if (wg_f2u(wg_log2f_ks(p, (double)p, g_iys0)) != wg_f2u(wg_log2f(p, g_tab.f_tab))) bad[(size_t)t]++; }
                else skip[(size_t)t]++;
            }
        });
    for (auto& x : th) x.join();
    uint64_t b = 0, sk = 0;
    for (int t = 0; t < T; t++) { b += bad[(size_t)t]; sk += skip[(size_t)t]; }
    if (skipped) *skipped = sk;
    return b;
}
// Largest distance in ulps between the shortened polynomial wg_fast_log2_ks<false>(1 - p) and the libm restatement
// wg_log2(1 - p), over `count` consecutive floats p from `first` (arguments below the tables' first row skipped).
uint64_t ks_log2_max_ulp(uint32_t first, uint64_t count, int threads)
{
    std::vector<uint64_t> mx((size_t)(threads < 1 ? 1 : threads), 0);
    std::vector<std::thread> th;
    const int T = (int)mx.size();
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t]() {
            for (uint64_t q = count * t / T; q < count * (t + 1) / T; q++) {
                const double x = 1.0 - (double)wg_u2f(first + (uint32_t)q);
                if (!(x > 0.6875 * 0x1p-23) || !(x < 1.0)) continue;
                const uint64_t a = wg_d2u(wg_fast_log2_ks<false>(x, g_kys0)), b = wg_d2u(wg_log2(x, g_tab.d_tab, g_tab.d_tab2));
                const uint64_t d = a > b ? a - b : b - a;
                if (d > mx[(size_t)t]) mx[(size_t)t] = d;
            }
        });
    for (auto& x : th) x.join();
    uint64_t m = 0;
    for (auto v : mx) m = v > m ? v : m;
    return m;
}
// Largest distance, in ulps of the double sum, between the sum wg_sample_term_pcpos_ks rounds (fused, shortened
// polynomial) and the reference's fl(ll + fl(df * log2(1 - p))).
uint64_t ks_sum_ulp_gap(const float* nmeth, const float* ntotal, int64_t count, float pc)
{
    const float pc2 = pc + pc;
    uint64_t mx = 0;
    for (int64_t q = 0; q < count; q++) {
        const float m = nmeth[q], t = ntotal[q];
        const float p = (m + pc) / (t + pc2);
        const float ll = m * wg_log2f_ks(p, (double)p, g_iys0);
        const float df = t - m;
        const double xx = 1.0 - (double)p;
        const double s1 = WG_FMA((double)df, wg_fast_log2_ks<false>(xx, g_kys0), (double)ll);
        const double s0 = (double)ll + (double)df * wg_log2(xx, g_tab.d_tab, g_tab.d_tab2);
        const uint64_t a = wg_d2u(s1), b = wg_d2u(s0);
        const uint64_t d = a > b ? a - b : b - a;
        if (d > mx) mx = d;
    }
    return mx;
}
// wg_lookup_rows against the arguments the kernels really compute: for every block total T in [t_lo, t_hi] the smallest p
// (nmeth = 0) and the smallest 1 - p (nmeth = T) — and their neighbours nmeth = 1, T - 1 — must index inside the rows
// granted for max_total = T.  Returns the number of violations.
uint64_t lookup_rows_violations(float pc, uint32_t t_lo, uint32_t t_hi)
{
    const float pc2 = pc + pc;
    uint64_t bad = 0;
    for (uint32_t T = t_lo; T <= t_hi; T++) {
        const int rows = wg_lookup_rows(pc, (double)T);
        const float t = (float)T;
        const float ms[4] = {0.0f, 1.0f, t - 1.0f, t};
        for (int j = 0; j < 4; j++) {
            const float m = ms[j];
            if (m < 0.0f || m > t) continue;
            const float p = wg_div_f32(m + pc, t + pc2);
            const double x = 1.0 - (double)p;
            const int kf = (int32_t)(wg_f2u(p) - 0x3f330000u) >> 23, kd = (int32_t)((uint32_t)(wg_d2u(x) >> 32) - 0x3fe60000u) >> 20;
            if (kf > 0 || kd > 0 || kf < -(rows - 1) || kd < -(rows - 1) || !(p > 0.0f) || !(p < 1.0f)) bad++;
        }
    }
    return bad;
}
void exact_sample_terms(const float* nmeth, const float* ntotal, int64_t count, float pc, float* out)
{
    float pc2 = pc + pc;
    const int mode = wg_term_mode(pc);                              // same dispatch rule as the library
    for (int64_t q = 0; q < count; q++) {
        if (mode == 2) {
            // the guard-free form with the zero-coverage exception against what the scoring kernels run (k-scaled tables, no
            // exception): they may differ only in the sign of a zero when ntotal == 0; anything else comes back as NaN, and so
            // does any table index outside the rows wg_lookup_rows() grants this pseudo count and longest block
            const float a = wg_sample_term_pcpos(nmeth[q], ntotal[q], pc, pc2, &g_fast, &g_tab);
            out[q] = a;
            for (int cls = 0; cls < 2; cls++) {                       // narrow tiles (blocks <= 60 sites), wide tiles (the ABI's longest)
                const double max_total = cls ? 255.0 * 8000.0 : 255.0 * 60.0;
                if ((double)ntotal[q] > max_total) continue;
                const int rows = wg_lookup_rows(pc, max_total);
                const float p = (nmeth[q] + pc) / (ntotal[q] + pc2);
                const double x = 1.0 - (double)p;
                const int kf = (int32_t)(wg_f2u(p) - 0x3f330000u) >> 23, kd = (int32_t)((uint32_t)(wg_d2u(x) >> 32) - 0x3fe60000u) >> 20;
                const float c = wg_sample_term_pcpos_ks(nmeth[q], ntotal[q], pc, pc2, g_iys0, g_kys0, &g_tab);
                const bool same = wg_f2u(c) == wg_f2u(a) || (a == 0.0f && c == 0.0f && ntotal[q] == 0.0f);
                if (!same || rows > WG_KY_KMIN + 1 || kf > 0 || kd > 0 || kf < -(rows - 1) || kd < -(rows - 1)) out[q] = __builtin_nanf("");
            }
        } else {
            out[q] = mode == 1 ? wg_sample_term(nmeth[q], ntotal[q], pc, pc2, &g_fast, &g_tab) : wg_sample_term_plain(nmeth[q], ntotal[q], pc, pc2, &g_tab);
        }
    }
}
}
