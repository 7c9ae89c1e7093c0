// table_io.h — host side of the block tools: the TEXT either side of the block reduction.
//
// `wgbstools beta_to_table` / `beta_to_blocks` (reference: src/python/beta_to_table.py:59-106, beta_to_blocks.py:50-126) read a
// blocks table (tab-separated chr, start, end, startCpG, endCpG [, more]) and print one row per block with a %.Nf value per
// sample or group.  With the reduction itself at a fraction of a millisecond on the device, reading and printing the table
// IS the tool's run time (2.8 M blocks x 32 samples: 3.9 s + 26 s in Python); the reference spends it in pandas' C parser and
// to_csv.  Here: one pass over the file's bytes for the columns the device needs, and a sharded multi-threaded writer (the
// scheme of add_loci.h) whose number formatting is exact.
//
// The parser is a FAST PATH, not a second definition of the format: anything it does not recognise as a plain row (a row with
// fewer than five fields, a CpG field that is neither digits nor one of the NA spellings, carriage returns, non-ASCII bytes, an
// empty table …) makes it answer "irregular", and the caller runs its line-by-line Python parser, which owns the error
// messages and the odd cases.  tests/test_blocks_cpu.py compares the two on every fixture and on adversarial files.
#pragma once
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>

namespace wgtab {

// ---------------------------------------------------------------------------------------------------------------------------
// numbers
// ---------------------------------------------------------------------------------------------------------------------------
inline char* put_u64(char* p, uint64_t v)
{
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

// printf("%.*f", digits, v) — what Python's '%.Nf' % v prints as well: the decimal of the EXACT binary value, rounded half to
// even on that exact value.  Values of a methylation table are averages in [0, 1]: for those (and digits <= 9) the rounding is
// done in 128-bit integers — v = M 2^-s exactly, K = round_half_even(M 10^d / 2^s) — an order of magnitude cheaper than
// snprintf; everything else (negative, > 1, infinities, more digits) goes through snprintf.  NaN is the table's "NA".
inline char* put_fixed(char* p, double v, int digits)
{
    if (v != v) { *p++ = 'N'; *p++ = 'A'; return p; }
    if (!(v >= 0.0 && v <= 1.0) || std::signbit(v) || digits > 9 || digits < 0) {
        const int n = snprintf(p, 400, "%.*f", digits < 0 ? 6 : digits, v);
        return p + (n > 0 ? n : 0);
    }
    static const uint64_t P10[10] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull};
    const uint64_t P = P10[digits];
    uint64_t K = 0;
    // v 10^d in double arithmetic is off by at most 10^9 2^-53 ~ 1.1e-7: unless it lands that close to a rounding boundary
    // (k + 1/2) it already names K; only then the exact integers decide
    const double t = v * (double)P, fl = std::floor(t), fr = t - fl;
    if (std::fabs(fr - 0.5) > 1e-6) {
        K = (uint64_t)fl + (fr > 0.5 ? 1u : 0u);
    } else if (v > 0.0) {
        int x;
        const double f = std::frexp(v, &x);                        // v = f 2^x, f in [0.5, 1), x <= 1
        const uint64_t M = (uint64_t)std::ldexp(f, 53);            // 53-bit integer, exact
        const int s = 53 - x;                                      // v = M 2^-s, s >= 52
        if (s < 120) {                                             // (else v 10^d < 2^(53+30-120): rounds to 0)
            const unsigned __int128 N = (unsigned __int128)M * P;  // < 2^83
            unsigned __int128 q = N >> s;
            const unsigned __int128 rem = N - (q << s), half = (unsigned __int128)1 << (s - 1);
            if (rem > half || (rem == half && (q & 1))) q++;
            K = (uint64_t)q;
        }
    }
    const uint32_t K32 = (uint32_t)K, P32 = (uint32_t)P;            // K <= 10^9
    const uint32_t ip = K32 >= P32 ? 1u : 0u;
    uint32_t w = K32 - ip * P32;
    *p++ = (char)('0' + ip);
    if (digits > 0) {
        *p++ = '.';
        for (int i = digits - 1; i >= 0; i--) { const uint32_t q = w / 10u; p[i] = (char)('0' + (w - q * 10u)); w = q; }
        p += digits;
    }
    return p;
}

// ---------------------------------------------------------------------------------------------------------------------------
// the blocks table: bytes -> (row offsets, CpG columns)
// ---------------------------------------------------------------------------------------------------------------------------
inline bool is_na_token(const char* a, size_t n)
{
    // beta_to_blocks.py NA_TOKENS: '', 'NA', 'NaN', 'nan', 'N/A', 'NULL', 'null', '<NA>', 'n/a', '#N/A', 'None'
    static const char* const T[] = {"", "NA", "NaN", "nan", "N/A", "NULL", "null", "<NA>", "n/a", "#N/A", "None"};
    for (const char* t : T) if (strlen(t) == n && memcmp(t, a, n) == 0) return true;
    return false;
}

inline bool all_digits(const char* a, size_t n)
{
    if (n == 0) return false;
    for (size_t i = 0; i < n; i++) if (a[i] < '0' || a[i] > '9') return false;
    return true;
}

inline bool blank_line(const char* a, size_t n)                    // Python: not line.strip()  (ASCII whitespace)
{
    for (size_t i = 0; i < n; i++) {
        const unsigned char c = (unsigned char)a[i];
        if (!(c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31))) return false;
    }
    return true;
}

// Returns 0 with *n_rows rows filled; 1: irregular — use the line-by-line parser; 2: more than `cap` rows.
// Row i: line_off[i] = offset of its first byte, len3[i] = bytes of "chr \t start \t end", start_cpg / end_cpg (0 where na[i]).
// Optional (NULL: not wanted): bp_start / bp_end = the row's second and third field as integers, *bp_ok = 1 when every row's are
// plain digits (else the arrays are not to be used: find_markers then converts the text itself, as int() would);
// *first_fields = fields of the first line that is neither a comment nor blank (the header, when there is one), at most 7 —
// what decides whether the table has the two annotation columns.
inline int parse_blocks(const char* t, int64_t len, int64_t max_rows, int64_t cap, int64_t* line_off, int32_t* len3,
                        int64_t* start_cpg, int64_t* end_cpg, uint8_t* na, int64_t* n_rows,
                        int64_t* bp_start = nullptr, int64_t* bp_end = nullptr, int32_t* bp_ok = nullptr, int32_t* first_fields = nullptr)
{
    *n_rows = 0;
    bool bp_good = bp_start && bp_end;
    if (bp_ok) *bp_ok = 0;
    if (first_fields) *first_fields = 0;
    if (len < 0 || !t) return 1;
    for (int64_t i = 0; i < len; i++) if ((unsigned char)t[i] >= 0x80 || t[i] == '\r' || t[i] == '\0') return 1;
    bool first = true;
    int64_t n = 0;
    int64_t a = 0;
    while (a < len) {
        const char* nl = static_cast<const char*>(memchr(t + a, '\n', (size_t)(len - a)));
        const int64_t b = nl ? (int64_t)(nl - t) : len;           // line = [a, b)
        const char* L = t + a;
        const size_t ln = (size_t)(b - a);
        const int64_t next = b + 1;
        if ((ln && L[0] == '#') || blank_line(L, ln)) { a = next; continue; }
        // the first five fields
        size_t tab[5];
        int nt = 0;
        for (size_t i = 0; i < ln && nt < 5; i++) if (L[i] == '\t') tab[nt++] = i;
        if (nt < 4) return 1;                                      // fewer than five fields: an error or an "invalid input" note
        const size_t f1a = tab[0] + 1, f1b = tab[1];
        if (first) {
            first = false;
            if (first_fields) {
                int nf = 1;
                for (size_t i = 0; i < ln && nf < 7; i++) if (L[i] == '\t') nf++;
                *first_fields = nf;
            }
            if (!all_digits(L + f1a, f1b - f1a)) { a = next; continue; }     // a header line
        }
        const size_t f3a = tab[2] + 1, f3b = tab[3], f4a = tab[3] + 1, f4b = nt == 5 ? tab[4] : ln;
        const bool miss = is_na_token(L + f3a, f3b - f3a) || is_na_token(L + f4a, f4b - f4a);
        int64_t s = 0, e = 0;
        if (!miss) {
            // int(float(tok)) of plain digits; beyond 15 digits a double no longer holds the integer: not this path's business
            if (!all_digits(L + f3a, f3b - f3a) || !all_digits(L + f4a, f4b - f4a) || f3b - f3a > 15 || f4b - f4a > 15) return 1;
            for (size_t i = f3a; i < f3b; i++) s = s * 10 + (L[i] - '0');
            for (size_t i = f4a; i < f4b; i++) e = e * 10 + (L[i] - '0');
        }
        if (n >= cap) return 2;
        if (tab[2] > 0x7fffffffu) return 1;
        line_off[n] = a; len3[n] = (int32_t)tab[2];
        start_cpg[n] = s; end_cpg[n] = e; na[n] = miss ? 1 : 0;
        if (bp_good) {
            const size_t f2a = tab[1] + 1, f2b = tab[2];
            if (all_digits(L + f1a, f1b - f1a) && all_digits(L + f2a, f2b - f2a) && f1b - f1a <= 15 && f2b - f2a <= 15) {
                int64_t x = 0, y = 0;
                for (size_t i = f1a; i < f1b; i++) x = x * 10 + (L[i] - '0');
                for (size_t i = f2a; i < f2b; i++) y = y * 10 + (L[i] - '0');
                bp_start[n] = x; bp_end[n] = y;
            } else bp_good = false;
        }
        n++;
        if (max_rows >= 0 && n >= max_rows) break;
        a = next;
    }
    *n_rows = n;
    if (bp_ok) *bp_ok = bp_good ? 1 : 0;
    return n == 0 ? 1 : 0;                                         // an empty table: the slow path prints the reference's note
}

// ---------------------------------------------------------------------------------------------------------------------------
// sharded writer: shards of rows formatted by a pool, placed by their lengths, written side by side with pwrite
// ---------------------------------------------------------------------------------------------------------------------------
struct Shard {
    char* buf = nullptr; size_t len = 0; bool oom = false;
    Shard() {}
    Shard(const Shard&) = delete;
    Shard& operator=(const Shard&) = delete;
    ~Shard() { free(buf); }
};

// fmt(lo, hi, out): the text of rows [lo, hi) into a malloc'ed out.buf.  Returns 0, or 3 with err set.
// base < 0: `fd` is not a regular file (a pipe, a terminal: standard output) — the shards are written one after the other.
inline int sharded_write(int fd, int64_t base, int64_t n_rows, int64_t shard_rows, int threads,
                         const std::function<void(int64_t, int64_t, Shard&)>& fmt, std::string& err)
{
    if (n_rows <= 0) return 0;
    int T = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    const int64_t n_shards = (n_rows + shard_rows - 1) / shard_rows;
    T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(T, 32), n_shards));
    std::vector<Shard> sh((size_t)n_shards);
    auto pool = [&](const std::function<void(int64_t)>& f) {
        std::atomic<int64_t> next(0);
        auto w = [&]() { for (int64_t k; (k = next.fetch_add(1)) < n_shards;) f(k); };
        if (T == 1) { w(); return; }
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back(w);
        for (auto& x : th) x.join();
    };
    pool([&](int64_t k) { fmt(k * shard_rows, std::min<int64_t>(n_rows, (k + 1) * shard_rows), sh[(size_t)k]); });
    std::vector<int64_t> off((size_t)n_shards + 1, base < 0 ? 0 : base);
    for (int64_t k = 0; k < n_shards; k++) {
        if (sh[(size_t)k].oom) { err = "out of memory"; return 3; }
        off[(size_t)k + 1] = off[(size_t)k] + (int64_t)sh[(size_t)k].len;
    }
    if (base < 0) {
        for (int64_t k = 0; k < n_shards; k++) {
            const Shard& x = sh[(size_t)k];
            size_t done = 0;
            while (done < x.len) {
                const ssize_t w = write(fd, x.buf + done, x.len - done);
                if (w <= 0) { err = "write failed"; return 3; }
                done += (size_t)w;
            }
        }
        return 0;
    }
    if (ftruncate(fd, off[(size_t)n_shards]) != 0) { err = "write failed"; return 3; }
    std::atomic<int> io_bad(0);
    pool([&](int64_t k) {
        const Shard& x = sh[(size_t)k];
        size_t done = 0;
        while (done < x.len) {
            const ssize_t w = pwrite(fd, x.buf + done, x.len - done, (off_t)(off[(size_t)k] + (int64_t)done));
            if (w <= 0) { io_bad.store(1); return; }
            done += (size_t)w;
        }
    });
    if (io_bad.load()) { err = "write failed"; return 3; }
    return 0;
}

struct Rows {                                                       // a parsed blocks table (parse_blocks) and its text
    const char* text; const int64_t* line_off; const int32_t* len3;
    const int64_t* start_cpg; const int64_t* end_cpg; const uint8_t* na;
};

inline char* put_coords(char* p, const Rows& R, int64_t r)
{
    memcpy(p, R.text + R.line_off[r], (size_t)R.len3[r]); p += R.len3[r];          // chr \t start \t end, as the file has them
    *p++ = '\t';
    if (R.na[r]) { *p++ = 'N'; *p++ = 'A'; } else p = put_u64(p, (uint64_t)R.start_cpg[r]);
    *p++ = '\t';
    if (R.na[r]) { *p++ = 'N'; *p++ = 'A'; } else p = put_u64(p, (uint64_t)R.end_cpg[r]);
    return p;
}

#define WG_TAB_SHARD 16384

// beta_to_table.py:108-127 (dump of a chunk): coords, then one %.<digits>f (NA for NaN) per column; vals[r * stride + c].
inline int write_table(int fd, int64_t base, const Rows& R, int64_t n_rows, const double* vals, int64_t n_cols, int64_t stride,
                       int digits, int threads, std::string& err)
{
    auto fmt = [&](int64_t lo, int64_t hi, Shard& out) {
        size_t cap = 1;
        const size_t cell = (size_t)(digits > 9 || digits < 0 ? 400 : digits + 3) + 1;
        for (int64_t r = lo; r < hi; r++) cap += (size_t)R.len3[r] + 2 * 21 + 1 + (size_t)n_cols * cell;
        // (a value outside [0, 1] prints through snprintf: up to ~320 characters for a huge double — room for those as well)
        size_t big = 0;
        for (int64_t r = lo; r < hi; r++) for (int64_t c = 0; c < n_cols; c++) { const double v = vals[r * stride + c]; if (v == v && !(v >= 0.0 && v <= 1.0)) big++; }
        cap += big * 400;
        out.buf = static_cast<char*>(malloc(cap));
        if (!out.buf) { out.oom = true; return; }
        char* p = out.buf;
        for (int64_t r = lo; r < hi; r++) {
            p = put_coords(p, R, r);
            const double* v = vals + r * stride;
            for (int64_t c = 0; c < n_cols; c++) { *p++ = '\t'; p = put_fixed(p, v[c], digits); }
            *p++ = '\n';
        }
        out.len = (size_t)(p - out.buf);
    };
    return sharded_write(fd, base, n_rows, WG_TAB_SHARD, threads, fmt, err);
}

// beta_to_blocks.py:112-126 (--bedGraph): chr, start, end, beta (%.2f; -1 for 0 / 0), coverage — from the trimmed rows.
// WIDE: uint16 pairs (.lbeta rows) instead of uint8 (.bin).
template <typename T>
inline int write_bedgraph(int fd, int64_t base, const Rows& R, int64_t n_rows, const T* mc, int threads, std::string& err)
{
    auto fmt = [&](int64_t lo, int64_t hi, Shard& out) {
        size_t cap = 1;
        for (int64_t r = lo; r < hi; r++) cap += (size_t)R.len3[r] + 32;
        out.buf = static_cast<char*>(malloc(cap));
        if (!out.buf) { out.oom = true; return; }
        char* p = out.buf;
        for (int64_t r = lo; r < hi; r++) {
            memcpy(p, R.text + R.line_off[r], (size_t)R.len3[r]); p += R.len3[r];
            const uint64_t m = mc[2 * r], v = mc[2 * r + 1];
            *p++ = '\t';
            if (v) p = put_fixed(p, (double)m / (double)v, 2); else { *p++ = '-'; *p++ = '1'; }
            *p++ = '\t';
            p = put_u64(p, v);
            *p++ = '\n';
        }
        out.len = (size_t)(p - out.buf);
    };
    return sharded_write(fd, base, n_rows, WG_TAB_SHARD, threads, fmt, err);
}

// ---------------------------------------------------------------------------------------------------------------------------
// `convert -L`: a BED table -> (chromosome, start, end) per row, and the annotated rows back out
// ---------------------------------------------------------------------------------------------------------------------------
// What pandas.read_csv treats as missing by default (convert.py NA_TOKENS).
inline bool is_pandas_na(const char* a, size_t n)
{
    static const char* const T[] = {"", "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN", "<NA>",
                                    "N/A", "NA", "NULL", "NaN", "None", "n/a", "nan", "null"};
    for (const char* t : T) if (strlen(t) == n && memcmp(t, a, n) == 0) return true;
    return false;
}

inline bool canonical_uint(const char* a, size_t n, size_t max_digits)     // digits, no sign, no leading zero: prints as it reads
{
    if (!all_digits(a, n) || n > max_digits) return false;
    return n == 1 || a[0] != '0';
}

// A decimal number that prints as it reads after a round trip through a float (pandas: read_csv types the column float64, to_csv
// prints repr(v)): digits '.' digits with an optional '-', no needless zeros, between 1e-4 and 1e16 in magnitude (or zero) — the
// range where Python's repr uses positional notation — and already the SHORTEST text that names its double (std::to_chars and
// repr both print that one).  "0.50", "7", "+1.5", "1e-3", ".5" are not: they come back as 0.5, 7.0, 1.5, 0.001, 0.5.
inline bool canonical_float(const char* a, size_t n)
{
    if (n < 3 || n > 26) return false;
    size_t i = a[0] == '-' ? 1 : 0;
    const size_t i0 = i;
    while (i < n && a[i] >= '0' && a[i] <= '9') i++;
    if (i == i0 || i >= n || a[i] != '.') return false;
    if (i - i0 > 1 && a[i0] == '0') return false;                  // 007.5
    const size_t f0 = ++i;
    while (i < n && a[i] >= '0' && a[i] <= '9') i++;
    if (i != n || i == f0) return false;
    char tmp[32];
    memcpy(tmp, a, n); tmp[n] = 0;
    const double v = strtod(tmp, nullptr);
    const double m = std::fabs(v);
    if (!(m == 0.0 || (m >= 1e-4 && m < 1e16))) return false;
    char out[48];
    const auto r = std::to_chars(out, out + 40, v, std::chars_format::fixed);
    if (r.ec != std::errc()) return false;
    size_t len = (size_t)(r.ptr - out);
    if (!memchr(out, '.', len)) { out[len++] = '.'; out[len++] = '0'; }
    return len == n && memcmp(out, a, n) == 0;
}

// A token that float() surely cannot parse: it has a character no float literal has (digits, sign, point, underscore, exponent,
// white space, the letters of "infinity" / "nan"), or it mixes digits with those letters, or it has no digit and is not one of
// those words.  (False for everything else, numbers or not: the caller only needs certainty in one direction.)
inline bool surely_text(const char* a, size_t n)
{
    bool digit = false, word = false;
    for (size_t i = 0; i < n; i++) {
        const char c = a[i];
        if (c >= '0' && c <= '9') { digit = true; continue; }
        if (c == '+' || c == '-' || c == '.' || c == '_' || c == 'e' || c == 'E' || c == ' ' || (c >= 9 && c <= 13)) continue;
        if (c == 'i' || c == 'I' || c == 'n' || c == 'N' || c == 'f' || c == 'F' || c == 't' || c == 'T' || c == 'y' || c == 'Y' || c == 'a' || c == 'A') { word = true; continue; }
        return true;
    }
    if (digit) return word;                                        // digits and a letter of inf / nan: neither a number nor a word; digits alone: maybe a number
    size_t lo = 0, hi = n;
    while (lo < hi && (a[lo] == ' ' || (a[lo] >= 9 && a[lo] <= 13))) lo++;
    while (hi > lo && (a[hi - 1] == ' ' || (a[hi - 1] >= 9 && a[hi - 1] <= 13))) hi--;
    if (lo < hi && (a[lo] == '+' || a[lo] == '-')) lo++;
    char w[9];
    const size_t m = hi - lo;
    if (m != 3 && m != 8) return true;
    for (size_t i = 0; i < m; i++) w[i] = (char)(a[lo + i] | 0x20);
    return !((m == 3 && (memcmp(w, "inf", 3) == 0 || memcmp(w, "nan", 3) == 0)) || (m == 8 && memcmp(w, "infinity", 8) == 0));
}

// The rows of a BED table whose text can go back out VERBATIM around the two new columns (wgbs_tools_amd/convert.py restates
// what a round trip through pandas does to a column: integers are re-printed, numeric columns become floats, missing values NA):
// that is the case when every row has the same number (>= 3) of fields, start and end are plain integers as they print, and every
// other column either holds a token no number parser accepts (a text column: written back as it came), or nothing but plain
// integers, or nothing but decimal numbers that print as they read (canonical_float; gaps spelled NA).  A header line (first line whose second and third fields are not numbers) is skipped and reported in *header_out when
// that is given: every column is text then.  Returns 0 with the rows; 1: not such a table (comments, ragged rows, numeric or
// missing-value columns, carriage returns, non-ASCII bytes ...) — the caller's Python handles it; 2: more than cap rows.
// chrom[i] = index of the row's first field in names[0..n_names), -1 when it is none of them.
inline int parse_bed(const char* t, int64_t len, int64_t cap, const char* const* names, int n_names, int64_t* line_off, int32_t* len3,
                     int32_t* row_len, int32_t* chrom, int64_t* start, int64_t* end, int64_t* n_rows, int32_t* width_out, int32_t* header_out = nullptr)
{
    *n_rows = 0; *width_out = 0;
    if (header_out) *header_out = 0;
    bool raw = false;                                              // a header line was skipped: pandas then reads every column as text
    if (len < 0 || !t) return 1;
    for (int64_t i = 0; i < len; i++) {
        const unsigned char c = (unsigned char)t[i];
        if (c >= 0x80 || c == '#' || c == '\r' || c == 0 || c == 11 || c == 12 || (c >= 28 && c <= 30)) return 1;   // (str.splitlines breaks lines at 11, 12, 28-30 too)
    }
    std::vector<size_t> nlen((size_t)n_names);
    for (int i = 0; i < n_names; i++) nlen[(size_t)i] = strlen(names[i]);
    int width = 0;
    std::vector<uint8_t> has_text, all_int, all_float, other_na;      // per column
    std::vector<size_t> tabs;
    int64_t n = 0;
    int last_chrom = -1;
    for (int64_t a = 0; a < len;) {
        const char* nl = static_cast<const char*>(memchr(t + a, '\n', (size_t)(len - a)));
        const int64_t b = nl ? (int64_t)(nl - t) : len;
        const char* L = t + a;
        const size_t ln = (size_t)(b - a);
        const int64_t next = b + 1;
        if (blank_line(L, ln)) { a = next; continue; }
        tabs.clear();
        for (size_t i = 0; i < ln; i++) if (L[i] == '\t') tabs.push_back(i);
        const int w = (int)tabs.size() + 1;
        const bool first_row = width == 0;
        if (first_row) {
            width = w;
            if (width < 3) return 1;
            has_text.assign((size_t)width, 0); all_int.assign((size_t)width, 1); all_float.assign((size_t)width, 1); other_na.assign((size_t)width, 0);
        } else if (w != width) return 1;
        tabs.push_back(ln);
        if (first_row && header_out) {
            // convert.py:77-89: a first line whose second and third fields are not numbers (after strip()) is a header: ignored, and
            // every column of the table is then text — written back as it came
            auto stripped_digits = [&](size_t a0, size_t b0) {
                while (a0 < b0 && (L[a0] == ' ' || (L[a0] >= 9 && L[a0] <= 13) || (L[a0] >= 28 && L[a0] <= 31))) a0++;
                while (b0 > a0 && (L[b0 - 1] == ' ' || (L[b0 - 1] >= 9 && L[b0 - 1] <= 13) || (L[b0 - 1] >= 28 && L[b0 - 1] <= 31))) b0--;
                return all_digits(L + a0, b0 - a0);
            };
            if (!(stripped_digits(tabs[0] + 1, tabs[1]) && stripped_digits(tabs[1] + 1, tabs[2]))) { raw = true; *header_out = 1; a = next; continue; }
        }
        // start, end: plain integers that print as they read (<= 15 digits)
        const size_t s_a = tabs[0] + 1, s_b = tabs[1], e_a = tabs[1] + 1, e_b = tabs[2];
        if (!canonical_uint(L + s_a, s_b - s_a, 15) || !canonical_uint(L + e_a, e_b - e_a, 15)) return 1;
        int64_t sv = 0, ev = 0;
        for (size_t i = s_a; i < s_b; i++) sv = sv * 10 + (L[i] - '0');
        for (size_t i = e_a; i < e_b; i++) ev = ev * 10 + (L[i] - '0');
        // the other columns
        for (int c = 0; c < width; c++) {
            if (c == 1 || c == 2) continue;
            const size_t fa = c == 0 ? 0 : tabs[(size_t)c - 1] + 1, fb = tabs[(size_t)c];
            const char* f = L + fa;
            const size_t fn = fb - fa;
            if (is_pandas_na(f, fn)) {                            // a missing value: no evidence about the column's type;
                all_int[(size_t)c] = 0;                           // an integer column with gaps comes back as floats
                if (!(fn == 2 && f[0] == 'N' && f[1] == 'A')) other_na[(size_t)c] = 1;       // and only "NA" prints as it reads
                continue;
            }
            if (surely_text(f, fn)) has_text[(size_t)c] = 1;
            if (!canonical_uint(f, fn, 18)) all_int[(size_t)c] = 0;
            if (all_float[(size_t)c] && !canonical_float(f, fn)) all_float[(size_t)c] = 0;
        }
        if (n >= cap) return 2;
        if (ln > 0x7fffffffu) return 1;
        int ci = -1;
        const size_t cl = tabs[0];
        if (last_chrom >= 0 && nlen[(size_t)last_chrom] == cl && memcmp(names[last_chrom], L, cl) == 0) ci = last_chrom;
        else for (int i = 0; i < n_names; i++) if (nlen[(size_t)i] == cl && memcmp(names[i], L, cl) == 0) { ci = i; break; }
        if (ci >= 0) last_chrom = ci;
        line_off[n] = a; len3[n] = (int32_t)tabs[2]; row_len[n] = (int32_t)ln; chrom[n] = ci; start[n] = sv; end[n] = ev;
        n++;
        a = next;
    }
    if (n == 0) return 1;
    for (int c = 0; c < width; c++) {
        if (c == 1 || c == 2) continue;
        const bool text_col = (raw || has_text[(size_t)c]) && !other_na[(size_t)c];
        // (a float column may have gaps: they print NA)
        const bool float_col = !raw && all_float[(size_t)c] && !all_int[(size_t)c] && !other_na[(size_t)c] && !has_text[(size_t)c];
        if (!text_col && !(all_int[(size_t)c] && !raw) && !float_col) return 1;
    }
    *n_rows = n; *width_out = width;
    return 0;
}

// convert.py:44-74 (the output of `convert -L`): chr, start, end as they came, startCpG, endCpG (NA for 0), the rest of the row.
inline int write_annotated_bed(int fd, int64_t base, const char* text, const int64_t* line_off, const int32_t* len3, const int32_t* row_len,
                               const int64_t* start_cpg, const int64_t* end_cpg, int64_t n_rows, int threads, std::string& err)
{
    auto fmt = [&](int64_t lo, int64_t hi, Shard& out) {
        size_t cap = 1;
        for (int64_t r = lo; r < hi; r++) cap += (size_t)row_len[r] + 2 * 21 + 2;
        out.buf = static_cast<char*>(malloc(cap));
        if (!out.buf) { out.oom = true; return; }
        char* p = out.buf;
        for (int64_t r = lo; r < hi; r++) {
            const char* L = text + line_off[r];
            memcpy(p, L, (size_t)len3[r]); p += len3[r];
            *p++ = '\t';
            if (start_cpg[r] == 0) { *p++ = 'N'; *p++ = 'A'; } else p = put_u64(p, (uint64_t)start_cpg[r]);
            *p++ = '\t';
            if (end_cpg[r] == 0) { *p++ = 'N'; *p++ = 'A'; } else p = put_u64(p, (uint64_t)end_cpg[r]);
            memcpy(p, L + len3[r], (size_t)(row_len[r] - len3[r])); p += row_len[r] - len3[r];
            *p++ = '\n';
        }
        out.len = (size_t)(p - out.buf);
    };
    return sharded_write(fd, base, n_rows, WG_TAB_SHARD, threads, fmt, err);
}

}  // namespace wgtab
