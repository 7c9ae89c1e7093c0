/* Sanitizer harness for the oracle's C restatement of the chunk DP (TEST INFRASTRUCTURE): oracle/segment_oracle.c on synthetic
 * chunks — the single-chunk entry point, the threaded one (which must agree with it), the chunk pool, and the edge parameters the
 * reference has trouble with (a one-site chunk, max_cpg 1, pseudo count 0, max_bp 1).  Built by tests/test_sanitizers_cpu.py with
 * and without -fsanitize=address,undefined (the threaded entry points also with -fsanitize=thread); all builds print the same lines. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int oracle_segment_chunk(const uint8_t *const *slices, int n_samples, int n_sites, const uint32_t *loci, float pseudo_count, int max_cpg,
                         uint32_t max_bp, int32_t *borders, int *n_borders, int *bad_sample, int *bad_site, double *M_out, int32_t *T_out,
                         double *band_out);
int oracle_segment_chunk_mt(const uint8_t *const *slices, int n_samples, int n_sites, const uint32_t *loci, float pseudo_count, int max_cpg,
                            uint32_t max_bp, int threads, int32_t *borders, int *n_borders);
int oracle_segment_chunks(const uint8_t *const *samples, int n_samples, const uint32_t *loci, const int64_t *start0, const int32_t *len,
                          int64_t n_chunks, float pseudo_count, int max_cpg, uint32_t max_bp, int threads, int32_t *borders_out,
                          int64_t borders_cap, int64_t *borders_off);

static uint64_t rng_state = 88172645463325252ULL;
static uint32_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 16); }

int main(void)
{
    enum { N = 5, S = 4000 };
    static uint8_t data[N][2 * S];
    static uint32_t loci[S];
    const uint8_t *rows[N];
    uint32_t pos = 1000;
    for (int i = 0; i < S; i++) { pos += 2 + rnd() % ((i / 400) % 2 ? 30 : 400); loci[i] = pos; }
    for (int s = 0; s < N; s++) {
        rows[s] = data[s];
        for (int i = 0; i < S; i++) {
            const int cov = (int)(rnd() % 40), level = ((i / 250 + s) % 3) * 45 + 5;     /* piecewise methylation levels */
            int m = 0;
            for (int q = 0; q < cov; q++) m += (int)(rnd() % 100) < level;
            data[s][2 * i] = (uint8_t)m; data[s][2 * i + 1] = (uint8_t)cov;
        }
    }
    static int32_t b1[S + 1], b2[S + 1];
    const struct { int n; float pc; int max_cpg; uint32_t max_bp; } cases[] = {
        {S, 15.0f, 1000, 2000}, {S, 0.0f, 60, 5000}, {S, 0.5f, 1, 100}, {1, 15.0f, 1000, 2000}, {2, 3.25f, 7, 1}, {777, 1e-6f, 300, 100000}, {S, 15.0f, 17, 700}};
    for (unsigned c = 0; c < sizeof(cases) / sizeof(cases[0]); c++) {
        int n1 = 0, n2 = 0, bs = -1, bsite = -1;
        const int r1 = oracle_segment_chunk(rows, N, cases[c].n, loci, cases[c].pc, cases[c].max_cpg, cases[c].max_bp, b1, &n1, &bs, &bsite, NULL, NULL, NULL);
        const int r2 = oracle_segment_chunk_mt(rows, N, cases[c].n, loci, cases[c].pc, cases[c].max_cpg, cases[c].max_bp, 3, b2, &n2);
        uint64_t h = 1469598103934665603ULL;
        for (int i = 0; i < n1; i++) h = (h ^ (uint64_t)(uint32_t)b1[i]) * 1099511628211ULL;
        printf("case %u: rc %d / %d, %d borders, checksum %016llx, threaded %s\n", c, r1, r2, n1, (unsigned long long)h,
               (r1 == r2 && n1 == n2 && memcmp(b1, b2, (size_t)n1 * 4) == 0) ? "identical" : "DIFFERENT");
    }
    {   /* chunk pool: ragged chunks, more chunks than threads */
        const int64_t st[6] = {0, 1000, 1000, 2500, 3999, 10};
        const int32_t ln[6] = {1000, 1500, 1, 1499, 1, 3000};
        static int32_t out[6 * (S + 1)];
        int64_t off[7];
        const int rc = oracle_segment_chunks(rows, N, loci, st, ln, 6, 15.0f, 1000, 2000, 4, out, 6 * (S + 1), off);
        uint64_t h = 1469598103934665603ULL;
        for (int64_t i = 0; rc == 0 && i < off[6]; i++) h = (h ^ (uint64_t)(uint32_t)out[i]) * 1099511628211ULL;
        printf("pool: rc %d, %lld borders, checksum %016llx\n", rc, rc == 0 ? (long long)off[6] : -1LL, (unsigned long long)h);
    }
    {   /* meth > cov is reported, not read past */
        data[2][2 * 123] = 9; data[2][2 * 123 + 1] = 3;
        int n1 = 0, bs = -1, bsite = -1;
        const int r = oracle_segment_chunk(rows, N, 500, loci, 15.0f, 1000, 2000, b1, &n1, &bs, &bsite, NULL, NULL, NULL);
        printf("bad data: rc %d sample %d site %d\n", r, bs, bsite);
    }
    return 0;
}
