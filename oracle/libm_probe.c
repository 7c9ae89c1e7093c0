/*
 * oracle/libm_probe.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The reference's only third-party arithmetic is the platform libm: `log2f` at segmentor.cpp:130 and
 * `log2` at segmentor.cpp:133 (glibc 2.35 in this image).  The HIP path carries bit-exact restatements of
 * both (wgbs_tools_amd/csrc/exact_log2.h).  These helpers evaluate the LIVE host libm over ranges of float
 * bit patterns so that tests can compare the restatement (host build and device build) against it,
 * exhaustively over the path's whole input domain:
 *     log2f(p)              for every float p in (0, 1]
 *     log2(1.0 - (double)p) for every float p in (0, 1)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <pthread.h>

static inline float f_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t bits_of_f(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint64_t bits_of_d(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

typedef struct {
    uint32_t first; uint64_t count; int which;       /* 0: log2f(p) ; 1: log2(1-(double)p) */
    const void *cand;                                /* candidate bits: uint32[count] or uint64[count]; NULL = fill `out` */
    void *out;
    uint64_t mismatches; uint32_t first_bad;
    int tid, nthreads;
} probe_job;

static void *probe_worker(void *arg)
{
    probe_job *jb = (probe_job *)arg;
    uint64_t lo = jb->count * (uint64_t)jb->tid / (uint64_t)jb->nthreads;
    uint64_t hi = jb->count * (uint64_t)(jb->tid + 1) / (uint64_t)jb->nthreads;
    uint64_t bad = 0; uint32_t fb = 0xffffffffu;
    for (uint64_t q = lo; q < hi; q++) {
        float p = f_from_bits(jb->first + (uint32_t)q);
        if (jb->which == 0) {
            uint32_t r = bits_of_f(log2f(p));
            if (jb->cand) { if (((const uint32_t *)jb->cand)[q] != r) { if (!bad) fb = jb->first + (uint32_t)q; bad++; } }
            else ((uint32_t *)jb->out)[q] = r;
        } else {
            uint64_t r = bits_of_d(log2(1.0 - (double)p));
            if (jb->cand) { if (((const uint64_t *)jb->cand)[q] != r) { if (!bad) fb = jb->first + (uint32_t)q; bad++; } }
            else ((uint64_t *)jb->out)[q] = r;
        }
    }
    jb->mismatches = bad; jb->first_bad = fb;
    return NULL;
}

static uint64_t run_probe(uint32_t first, uint64_t count, int which, const void *cand, void *out,
                          int threads, uint32_t *first_bad)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256]; probe_job jb[256];
    for (int t = 0; t < threads; t++) {
        jb[t].first = first; jb[t].count = count; jb[t].which = which; jb[t].cand = cand; jb[t].out = out;
        jb[t].mismatches = 0; jb[t].first_bad = 0xffffffffu; jb[t].tid = t; jb[t].nthreads = threads;
        pthread_create(&th[t], NULL, probe_worker, &jb[t]);
    }
    uint64_t bad = 0; uint32_t fb = 0xffffffffu;
    for (int t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        bad += jb[t].mismatches;
        if (jb[t].first_bad < fb) fb = jb[t].first_bad;
    }
    if (first_bad) *first_bad = fb;
    return bad;
}

/* out[q] = bits(log2f(float_from_bits(first+q))) */
void probe_log2f_fill(uint32_t first, uint64_t count, uint32_t *out, int threads)
{ run_probe(first, count, 0, NULL, out, threads, NULL); }

/* out[q] = bits(log2(1.0 - (double)float_from_bits(first+q))) */
void probe_log2_1mp_fill(uint32_t first, uint64_t count, uint64_t *out, int threads)
{ run_probe(first, count, 1, NULL, out, threads, NULL); }

/* number of q with cand[q] != host libm; *first_bad = lowest offending float bit pattern */
uint64_t probe_log2f_compare(uint32_t first, uint64_t count, const uint32_t *cand, int threads, uint32_t *first_bad)
{ return run_probe(first, count, 0, cand, NULL, threads, first_bad); }

uint64_t probe_log2_1mp_compare(uint32_t first, uint64_t count, const uint64_t *cand, int threads, uint32_t *first_bad)
{ return run_probe(first, count, 1, cand, NULL, threads, first_bad); }

/* largest |cand[q] - libm| in units of the last place (bit patterns compared as integers; same sign assumed),
 * for log2(1.0 - (double)p): the error bound of the HIP path's fast log2 is established with this. */
typedef struct { uint32_t first; uint64_t a, b; const uint64_t *cand; uint64_t maxulp; } ulp_job;
static void *ulp_worker(void *arg)
{
    ulp_job *jb = (ulp_job *)arg;
    uint64_t mx = 0;
    for (uint64_t q = jb->a; q < jb->b; q++) {
        float p = f_from_bits(jb->first + (uint32_t)q);
        uint64_t r = bits_of_d(log2(1.0 - (double)p)), c = jb->cand[q];
        uint64_t d = r > c ? r - c : c - r;
        if (d > mx) mx = d;
    }
    jb->maxulp = mx;
    return NULL;
}
uint64_t probe_log2_1mp_maxulp(uint32_t first, uint64_t count, const uint64_t *cand, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256]; ulp_job jb[256];
    for (int t = 0; t < threads; t++) {
        jb[t].first = first; jb[t].cand = cand; jb[t].maxulp = 0;
        jb[t].a = count * (uint64_t)t / (uint64_t)threads; jb[t].b = count * (uint64_t)(t + 1) / (uint64_t)threads;
        pthread_create(&th[t], NULL, ulp_worker, &jb[t]);
    }
    uint64_t mx = 0;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); if (jb[t].maxulp > mx) mx = jb[t].maxulp; }
    return mx;
}
