/*
 * oracle/segment_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C CPU restatement of the reference chunk segmenter (nloyfer/wgbs_tools
 * src/segment_betas/segmentor.cpp).  It exists so that tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg have an independent checker for the HIP path.  Nothing under
 * wgbs_tools_amd/ links, imports or executes it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py compares this restatement with border
 * lists produced by the reference's own sources compiled where they lie (oracle/Makefile
 * target `ref` -> oracle/_ref/segmentor) on seeded inputs; the captured vectors are
 * tests/golden/chunk_cases.json (generator: tests/golden/make_golden.py).
 *
 * Every function cites the reference lines it follows.  The arithmetic order is the
 * reference's: per-sample running float sums along the block extension, float p, float
 * nmeth*log2f(p), double (ntotal-nmeth)*log2(1-p) folded back to float, double sum over
 * samples in argv order, strict '>' arg-max scan in ascending k.  log2f/log2 are the host
 * libm's (the reference links the same two symbols and nothing else from libm).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define ORACLE_OK            0
#define ORACLE_E_ARG        -1
#define ORACLE_E_METH_GT_COV -2   /* segmentor.cpp:181-188: "invalid data" -> throw 0 */
#define ORACLE_E_NOMEM      -3

/* Per-(block, sample) log-likelihood term.  segmentor.cpp:120-135. */
static inline float sample_term(float nmeth, float ntotal, float pseudo_count)
{
    /* :127  float division; `2 * pseudo_count` is an exact doubling in float */
    float p = (nmeth + pseudo_count) / (ntotal + (2 * pseudo_count));
    float ll = 0;
    if (p > 0.0) {                                   /* :129-131  float * log2f(float) */
        ll += (nmeth * log2f(p));
    }
    if (p < 1.0) {                                   /* :132-134  double product folded into the float */
        ll = (float)((double)ll + (double)(ntotal - nmeth) * log2(1.0 - (double)p));
    }
    return ll;
}

typedef struct {
    int n_samples;
    int n_sites;
    float pseudo_count;
    int max_cpg;
    uint32_t max_bp;
} oracle_params;

/*
 * The DP of segmentor.cpp:60-159 on already-widened float data.
 *   data[s][2*i] = #meth, data[s][2*i+1] = #cov of site i (segmentor.cpp:179 layout)
 *   loci[i]      = bp position of site i (segmentor.cpp:36-48 `dists`)
 * Outputs: M[n+1], T[n+1] (caller-allocated).  Optional band: band[i*max_cpg + j] = cost of the
 * block starting at site i spanning j+1 sites (the reference's row i, entry j), -inf when barred.
 */
static int dp_core(const oracle_params *pr, float *const *data, const uint32_t *loci,
                   double *M, int32_t *T, double *band)
{
    const int n = pr->n_sites, N = pr->n_samples, max_cpg = pr->max_cpg;
    const double NEG_INF = -INFINITY;

    /* ring of the last `max_cpg` cost rows (segmentor.cpp:92-95) — pow2 slots, row k -> slot k & mask */
    int ring = 1;
    while (ring < max_cpg) ring <<= 1;
    const int mask = ring - 1;
    double *rows = (double *)calloc((size_t)ring * (size_t)max_cpg, sizeof(double));
    float *run_m = (float *)malloc(sizeof(float) * (size_t)N);
    float *run_t = (float *)malloc(sizeof(float) * (size_t)N);
    if (!rows || !run_m || !run_t) { free(rows); free(run_m); free(run_t); return ORACLE_E_NOMEM; }

    M[0] = 0.0;                                         /* :97 value-initialised arrays */
    T[0] = 0;

    for (int i = 0; i < n; i++) {
        double *row = rows + (size_t)(i & mask) * (size_t)max_cpg;
        for (int j = 0; j < max_cpg; j++) row[j] = 0.0;             /* :106 */
        memset(run_m, 0, sizeof(float) * (size_t)N);                /* :108-109 */
        memset(run_t, 0, sizeof(float) * (size_t)N);

        int window = n - i < max_cpg ? n - i : max_cpg;             /* :111 */
        for (int j = 0; j < window; j++) {
            /* :114-117  unsigned distance test; a barred extension is skipped WITHOUT accumulating */
            if ((uint32_t)(loci[i + j] - loci[i]) > pr->max_bp || loci[i + j] < loci[i]) {
                row[j] = NEG_INF;
                continue;
            }
            double ll_sum = 0;
            for (int s = 0; s < N; s++) {                           /* :120-136, argv order */
                run_m[s] += data[s][(size_t)(i + j) * 2];
                run_t[s] += data[s][(size_t)(i + j) * 2 + 1];
                float nt = run_t[s], nm = run_m[s];
                if (!nt) continue;                                  /* :125 */
                ll_sum += sample_term(nm, nt, pr->pseudo_count);    /* :135 double += float */
            }
            if (ll_sum) row[j] = ll_sum;                            /* :137 */
        }
        if (band) memcpy(band + (size_t)i * (size_t)max_cpg, row, sizeof(double) * (size_t)max_cpg);

        /* :142-154  M[i+1] = max_k M[k] + row_k[i-k], first maximum wins */
        double best = NEG_INF;
        int best_k = -1;
        int k0 = i + 1 - max_cpg > 0 ? i + 1 - max_cpg : 0;
        for (int k = k0; k <= i; k++) {
            double v = M[k] + rows[(size_t)(k & mask) * (size_t)max_cpg + (size_t)(i - k)];
            if (v > best) { best = v; best_k = k; }
        }
        M[i + 1] = best;
        T[i + 1] = best_k;
    }
    free(rows); free(run_m); free(run_t);
    return ORACLE_OK;
}

/* segmentor.cpp:50-58 traceback + :30-34 print order (ascending, includes 0 and n). */
static int traceback(const int32_t *T, int n, int32_t *borders)
{
    int cnt = 0, i = n;
    borders[cnt++] = i;
    while (i > 0) { i = T[i] > 0 ? T[i] : 0; borders[cnt++] = i; }
    for (int a = 0, b = cnt - 1; a < b; a++, b--) { int32_t t = borders[a]; borders[a] = borders[b]; borders[b] = t; }
    return cnt;
}

/*
 * One chunk, the unit `segmentor` runs (segmentor.cpp:193-214 dp_wrapper).
 *   slices[s]  -> the 2*n bytes of sample s for this chunk (what read_beta_file seeks to and reads, :164-177)
 *   borders    -> capacity n+1; *n_borders receives the count
 *   bad_sample/bad_site -> filled on ORACLE_E_METH_GT_COV (:181-188)
 *   M_out/T_out (n+1 each) and band_out (n*max_cpg) are optional debugging outputs.
 */
int oracle_segment_chunk(const uint8_t *const *slices, int n_samples, int n_sites,
                         const uint32_t *loci, float pseudo_count, int max_cpg, uint32_t max_bp,
                         int32_t *borders, int *n_borders,
                         int *bad_sample, int *bad_site,
                         double *M_out, int32_t *T_out, double *band_out)
{
    if (n_samples < 1 || n_sites < 1 || max_cpg < 1 || !slices || !loci || !borders || !n_borders)
        return ORACLE_E_ARG;
    if (max_bp == 0) return ORACLE_E_ARG;   /* the reference leaves `dists` uninitialised here (:38): UB, rejected */

    int rc = ORACLE_OK;
    float **data = (float **)calloc((size_t)n_samples, sizeof(float *));
    double *M = M_out ? M_out : (double *)malloc(sizeof(double) * ((size_t)n_sites + 1));
    int32_t *T = T_out ? T_out : (int32_t *)malloc(sizeof(int32_t) * ((size_t)n_sites + 1));
    if (!data || !M || !T) { rc = ORACLE_E_NOMEM; goto done; }

    for (int s = 0; s < n_samples; s++) {
        data[s] = (float *)malloc(sizeof(float) * 2 * (size_t)n_sites);
        if (!data[s]) { rc = ORACLE_E_NOMEM; goto done; }
        for (size_t b = 0; b < 2 * (size_t)n_sites; b++) data[s][b] = (float)slices[s][b];   /* :179 */
        for (int i = 0; i < n_sites; i++) {                                                     /* :181-188 */
            if (data[s][2 * (size_t)i] > data[s][2 * (size_t)i + 1]) {
                if (bad_sample) *bad_sample = s;
                if (bad_site) *bad_site = i;
                rc = ORACLE_E_METH_GT_COV;
                goto done;
            }
        }
    }
    {
        oracle_params pr = { n_samples, n_sites, pseudo_count, max_cpg, max_bp };
        rc = dp_core(&pr, data, loci, M, T, band_out);
        if (rc == ORACLE_OK) *n_borders = traceback(T, n_sites, borders);
    }
done:
    if (data) { for (int s = 0; s < n_samples; s++) free(data[s]); free(data); }
    if (!M_out) free(M);
    if (!T_out) free(T);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Many chunks over whole-file sample arrays, `threads` worker threads: the shape of the reference's
 * Pool(threads).starmap(segment_process) (segment.py:144-146), minus fork/sh/tabix.  Used by tests for
 * bulk comparison and by bench.py's cpu_baseline ("port") leg.
 *   samples[s] -> whole beta array of sample s ([n_total][2] uint8); loci -> whole-genome loci
 *   chunk c covers sites [start0[c], start0[c]+len[c])
 *   borders_out/borders_off: CSR, chunk c's borders (relative to its start, incl. 0 and len) at
 *   borders_out[borders_off[c] .. borders_off[c+1]); capacity needed: sum(len)+n_chunks.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const uint8_t *const *samples; int n_samples; const uint32_t *loci;
    const int64_t *start0; const int32_t *len; int64_t n_chunks;
    float pc; int max_cpg; uint32_t max_bp;
    int32_t **chunk_borders; int *chunk_nb; int *chunk_rc;
    int64_t next; pthread_mutex_t mu;
} job_t;

static void *worker(void *arg)
{
    job_t *jb = (job_t *)arg;
    const uint8_t **sl = (const uint8_t **)malloc(sizeof(uint8_t *) * (size_t)jb->n_samples);
    for (;;) {
        pthread_mutex_lock(&jb->mu);
        int64_t c = jb->next++;
        pthread_mutex_unlock(&jb->mu);
        if (c >= jb->n_chunks) break;
        int n = jb->len[c];
        for (int s = 0; s < jb->n_samples; s++) sl[s] = jb->samples[s] + 2 * jb->start0[c];
        jb->chunk_borders[c] = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
        int bs = -1, bi = -1;
        jb->chunk_rc[c] = oracle_segment_chunk(sl, jb->n_samples, n, jb->loci + jb->start0[c], jb->pc,
                                               jb->max_cpg, jb->max_bp, jb->chunk_borders[c],
                                               &jb->chunk_nb[c], &bs, &bi, NULL, NULL, NULL);
    }
    free(sl);
    return NULL;
}

int oracle_segment_chunks(const uint8_t *const *samples, int n_samples, const uint32_t *loci,
                          const int64_t *start0, const int32_t *len, int64_t n_chunks,
                          float pseudo_count, int max_cpg, uint32_t max_bp, int threads,
                          int32_t *borders_out, int64_t borders_cap, int64_t *borders_off)
{
    if (n_chunks < 1 || threads < 1) return ORACLE_E_ARG;
    job_t jb;
    memset(&jb, 0, sizeof(jb));
    jb.samples = samples; jb.n_samples = n_samples; jb.loci = loci;
    jb.start0 = start0; jb.len = len; jb.n_chunks = n_chunks;
    jb.pc = pseudo_count; jb.max_cpg = max_cpg; jb.max_bp = max_bp;
    jb.chunk_borders = (int32_t **)calloc((size_t)n_chunks, sizeof(int32_t *));
    jb.chunk_nb = (int *)calloc((size_t)n_chunks, sizeof(int));
    jb.chunk_rc = (int *)calloc((size_t)n_chunks, sizeof(int));
    pthread_mutex_init(&jb.mu, NULL);
    if (threads > 256) threads = 256;
    pthread_t th[256];
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, worker, &jb);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    pthread_mutex_destroy(&jb.mu);

    int rc = ORACLE_OK;
    int64_t off = 0;
    for (int64_t c = 0; c < n_chunks; c++) {
        if (jb.chunk_rc[c] != ORACLE_OK && rc == ORACLE_OK) rc = jb.chunk_rc[c];
        borders_off[c] = off;
        if (rc == ORACLE_OK) {
            if (off + jb.chunk_nb[c] > borders_cap) { rc = ORACLE_E_ARG; }
            else { memcpy(borders_out + off, jb.chunk_borders[c], sizeof(int32_t) * (size_t)jb.chunk_nb[c]); off += jb.chunk_nb[c]; }
        }
    }
    borders_off[n_chunks] = off;
    for (int64_t c = 0; c < n_chunks; c++) free(jb.chunk_borders[c]);
    free(jb.chunk_borders); free(jb.chunk_nb); free(jb.chunk_rc);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * One BIG chunk on many threads (full-size parity tests: a 50,000-site chunk x 512 samples at max_cpg 5000 is
 * 1.3e11 likelihood evaluations).  Same arithmetic as dp_core, statement for statement: a cost row depends only on
 * its start site (segmentor.cpp:103-138), so rows are filled in parallel, slab by slab, into the reference's ring of
 * rows (segmentor.cpp:92-95); the recurrence (segmentor.cpp:142-154) then runs over the slab sequentially, exactly as
 * in dp_core.  tests/test_oracle_golden.py pins this variant to the single-threaded one and to the goldens.
 * ---------------------------------------------------------------------------------------------- */
static void cost_row(const oracle_params *pr, float *const *data, const uint32_t *loci, int i, double *row,
                     float *run_m, float *run_t)
{
    const int n = pr->n_sites, N = pr->n_samples, max_cpg = pr->max_cpg;
    for (int j = 0; j < max_cpg; j++) row[j] = 0.0;                 /* :106 */
    memset(run_m, 0, sizeof(float) * (size_t)N);                    /* :108-109 */
    memset(run_t, 0, sizeof(float) * (size_t)N);
    int window = n - i < max_cpg ? n - i : max_cpg;                 /* :111 */
    for (int j = 0; j < window; j++) {
        if ((uint32_t)(loci[i + j] - loci[i]) > pr->max_bp || loci[i + j] < loci[i]) {   /* :114-117 */
            row[j] = -INFINITY;
            continue;
        }
        double ll_sum = 0;
        for (int s = 0; s < N; s++) {                               /* :120-136, argv order */
            run_m[s] += data[s][(size_t)(i + j) * 2];
            run_t[s] += data[s][(size_t)(i + j) * 2 + 1];
            float nt = run_t[s], nm = run_m[s];
            if (!nt) continue;                                      /* :125 */
            ll_sum += sample_term(nm, nt, pr->pseudo_count);        /* :135 */
        }
        if (ll_sum) row[j] = ll_sum;                                /* :137 */
    }
}

typedef struct {
    const oracle_params *pr; float *const *data; const uint32_t *loci;
    double *rows; int mask; int i0, i1; int next; pthread_mutex_t *mu;
} slab_t;

static void *slab_worker(void *arg)
{
    slab_t *sb = (slab_t *)arg;
    const int N = sb->pr->n_samples, max_cpg = sb->pr->max_cpg;
    float *run_m = (float *)malloc(sizeof(float) * (size_t)N), *run_t = (float *)malloc(sizeof(float) * (size_t)N);
    for (;;) {
        pthread_mutex_lock(sb->mu);
        int i = sb->next; sb->next += 8;
        pthread_mutex_unlock(sb->mu);
        if (i >= sb->i1) break;
        for (int q = i; q < i + 8 && q < sb->i1; q++)
            cost_row(sb->pr, sb->data, sb->loci, q, sb->rows + (size_t)(q & sb->mask) * (size_t)max_cpg, run_m, run_t);
    }
    free(run_m); free(run_t);
    return NULL;
}

int oracle_segment_chunk_mt(const uint8_t *const *slices, int n_samples, int n_sites,
                            const uint32_t *loci, float pseudo_count, int max_cpg, uint32_t max_bp, int threads,
                            int32_t *borders, int *n_borders)
{
    if (n_samples < 1 || n_sites < 1 || max_cpg < 1 || !slices || !loci || !borders || !n_borders || max_bp == 0 || threads < 1)
        return ORACLE_E_ARG;
    if (threads > 512) threads = 512;
    const int n = n_sites;
    int rc = ORACLE_OK;
    const int SLAB = 2048;
    int ring = 1;
    while (ring < max_cpg + SLAB) ring <<= 1;
    const int mask = ring - 1;
    float **data = (float **)calloc((size_t)n_samples, sizeof(float *));
    double *M = (double *)malloc(sizeof(double) * ((size_t)n + 1));
    int32_t *T = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    double *rows = (double *)malloc(sizeof(double) * (size_t)ring * (size_t)max_cpg);
    if (!data || !M || !T || !rows) { rc = ORACLE_E_NOMEM; goto done; }
    for (int s = 0; s < n_samples; s++) {
        data[s] = (float *)malloc(sizeof(float) * 2 * (size_t)n);
        if (!data[s]) { rc = ORACLE_E_NOMEM; goto done; }
        for (size_t b = 0; b < 2 * (size_t)n; b++) data[s][b] = (float)slices[s][b];          /* :179 */
        for (int i = 0; i < n; i++)
            if (data[s][2 * (size_t)i] > data[s][2 * (size_t)i + 1]) { rc = ORACLE_E_METH_GT_COV; goto done; }   /* :181-188 */
    }
    {
        oracle_params pr = { n_samples, n, pseudo_count, max_cpg, max_bp };
        pthread_mutex_t mu;
        pthread_mutex_init(&mu, NULL);
        M[0] = 0.0; T[0] = 0;                                         /* :97 */
        for (int i0 = 0; i0 < n; i0 += SLAB) {
            const int i1 = i0 + SLAB < n ? i0 + SLAB : n;
            slab_t sb = { &pr, data, loci, rows, mask, i0, i1, i0, &mu };
            pthread_t th[512];
            for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, slab_worker, &sb);
            for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
            for (int i = i0; i < i1; i++) {                           /* :142-154, first maximum wins */
                double best = -INFINITY;
                int best_k = -1;
                int k0 = i + 1 - max_cpg > 0 ? i + 1 - max_cpg : 0;
                for (int k = k0; k <= i; k++) {
                    double v = M[k] + rows[(size_t)(k & mask) * (size_t)max_cpg + (size_t)(i - k)];
                    if (v > best) { best = v; best_k = k; }
                }
                M[i + 1] = best;
                T[i + 1] = best_k;
            }
        }
        pthread_mutex_destroy(&mu);
        *n_borders = traceback(T, n, borders);
    }
done:
    if (data) { for (int s = 0; s < n_samples; s++) free(data[s]); free(data); }
    free(M); free(T); free(rows);
    return rc;
}

/* The per-(block,sample) term on its own: lets tests compare the device evaluation point-wise. */
float oracle_sample_term(float nmeth, float ntotal, float pseudo_count)
{
    if (!ntotal) return 0.0f;
    return sample_term(nmeth, ntotal, pseudo_count);
}

void oracle_sample_terms(const float *nmeth, const float *ntotal, int64_t count, float pseudo_count, float *out)
{
    for (int64_t q = 0; q < count; q++) out[q] = oracle_sample_term(nmeth[q], ntotal[q], pseudo_count);
}
