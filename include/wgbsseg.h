/*
 * wgbsseg.h — C ABI of the MI355X-native `wgbstools segment` hot path (libwgbsseg.so, built by hipcc for gfx950).
 *
 * What it replaces.  The reference has no FFI on this path: its boundary is a PROCESS boundary.
 * src/python/segment.py:41-59 (`segment_process`) runs, per chunk of CpG sites,
 *     tabix rev.CpG.bed.gz chr:start-(end-1) | cut -f2 | segmentor b1.beta ... -s start-1 -n end-start
 *                                                         -max_cpg M -ps P -max_bp B
 * and parses the integers `segmentor` prints (src/segment_betas/main.cpp:89-112 ->
 * segmentor::dp_wrapper segmentor.cpp:193-214 -> read_beta_file :164-190, load_dists :36-48, dp :60-159,
 * traceback :50-58, print_borders :30-34).  One call of wgbsseg_segment_chunks() stands for a whole
 * Pool.starmap(segment_process, chunks) (segment.py:144-146): same inputs (beta bytes, loci, -s/-n per chunk,
 * -max_cpg/-ps/-max_bp), same outputs (per chunk the ascending border list including 0 and n, relative to the
 * chunk start), bit-exact.  INTEGRATION.md shows the ctypes stub a maintainer would put in segment.py.
 *
 * Conventions: plain pointers and sizes, no C++/torch types.  The caller owns every host buffer for the
 * duration of a call; the library never keeps a host pointer after returning.  Device buffers passed with the
 * *_device setters are borrowed (the caller keeps them alive until they are replaced or the context is
 * destroyed).  All calls on one context must come from one thread at a time; different contexts (one per GPU)
 * are independent.  Every entry point returns WGBSSEG_OK or a negative code and, when `err` is non-NULL,
 * a NUL-terminated message.  There is NO CPU fallback: without a gfx950 device wgbsseg_create() fails.
 */
#ifndef WGBSSEG_H
#define WGBSSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WGBSSEG_VERSION 220            /* 0.2.1: round 4 changed wgbsseg_scan_only's argument list and added wgbsseg_add_loci_borders without a bump; a binding compares wgbsseg_version() with the header it was written against */
#define WGBSSEG_MAX_CPG 65535          /* longest block, in sites (min(max_cpg, longest chunk)): see wgbsseg_segment_chunks */

#define WGBSSEG_OK              0
#define WGBSSEG_E_ARG          -1      /* bad argument (message says which) */
#define WGBSSEG_E_METH_GT_COV  -2      /* a requested site has #meth > #cov: segmentor.cpp:181-188 "invalid data" */
#define WGBSSEG_E_NOMEM        -3
#define WGBSSEG_E_HIP          -4      /* HIP runtime error / no usable device */
#define WGBSSEG_E_LOCI_ORDER   -5      /* (internal since round 4: chunks whose loci do not ascend take the plain path — the reference's loops as written, segmentor.cpp:114-117 — and are no error) */
#define WGBSSEG_E_CAPACITY     -6      /* borders_out too small */
#define WGBSSEG_E_STATE        -7      /* betas / loci not set */

typedef struct wgbsseg_ctx wgbsseg_ctx;

/* struct Params of segmentor.h:16-23, minus start/nr_sites (those are per chunk). */
typedef struct wgbsseg_params {
    float    pseudo_count;   /* -ps      (segment.py passes --pcount, default 15: segment.py:269)            */
    uint32_t max_cpg;        /* -max_cpg (segment.py passes min(--max_cpg, --max_bp/2): segment.py:65)       */
    uint32_t max_bp;         /* -max_bp  (must be >= 1: with 0 the reference reads uninitialised memory, segmentor.cpp:38,114) */
} wgbsseg_params;

/* HIP-event timings of the last wgbsseg_segment_chunks() call, milliseconds, and its work counters. */
typedef struct wgbsseg_timings {
    double scan_ms;          /* scan pass over the beta bytes: validation (+ carries when the job has wide tiles)  */
    double window_ms;        /* window extents from loci + CSR offsets                                 */
    double cost_ms;          /* block log-likelihood evaluation, summed over stages                    */
    double dp_ms;            /* changepoint recurrence, summed over stages                             */
    double trace_ms;         /* traceback + compaction                                                 */
    double total_ms;         /* first kernel start -> borders on the host (device time line)           */
    int64_t sites;           /* sum of chunk lengths                                                   */
    int64_t pairs;           /* sum over sites of the window F_k = candidate blocks scored             */
    int64_t evals;           /* pairs * n_samples = per-(block, sample) likelihood evaluations         */
    int64_t scan_bytes;      /* algorithmic bytes of the scan pass: 2 * n_samples * sites              */
    int32_t max_window;      /* largest F_k                                                            */
    int32_t n_stages;
    int32_t scan_launches;   /* kernel launches behind scan_ms (1 per batch)                           */
    int32_t div_short;       /* 1: the narrow scoring tiles of the call ran the verified 4-instruction division core */
    double  scan_main_ms;    /* duration of the largest scan launch of the call (the batch holding the chunks) */
    int64_t scan_main_bytes; /* its algorithmic bytes                                                  */
} wgbsseg_timings;

int wgbsseg_version(void);

/* Number of visible gfx950 devices (0 if none / no HIP runtime). */
int wgbsseg_device_count(void);

/* Create a context bound to HIP device `device` (its own streams and scratch memory). */
int wgbsseg_create(int device, wgbsseg_ctx** out, char* err, size_t errlen);
void wgbsseg_destroy(wgbsseg_ctx* ctx);

/*
 * Beta data: n_samples arrays of n_sites x 2 uint8 (#meth, #cov) — the `.beta` file format
 * (docs/beta_format.md:3-8; what read_beta_file segmentor.cpp:164-177 reads).  Sample order = argv order of the
 * reference (it fixes the order of the double accumulation, segmentor.cpp:120-136).
 * _host: copies every sample into one device allocation [n_samples][pitch] (pitch = 2*n_sites rounded up to 256 B);
 *        the pointers may be pageable memory (e.g. memory-mapped .beta files): above 32 MB the copy runs on a few host
 *        threads through page-locked staging pieces (WGBSSEG_UPLOAD_THREADS, default 4).  Blocking.
 * _device: borrows a device buffer with that layout; `base` and `pitch_bytes` must be multiples of 16.
 */
int wgbsseg_set_betas_host(wgbsseg_ctx* ctx, const uint8_t* const* samples, int64_t n_samples, int64_t n_sites,
                           char* err, size_t errlen);
/* .lbeta rows (uint16 pairs, docs/beta_format.md:41-44; utils_wgbs.py:311-319 reads them as np.uint16) for the block
 * reduction only: wgbsseg_block_sums accepts them, the segment calls refuse (the reference's segmentor reads uint8 only). */
int wgbsseg_set_lbetas_host(wgbsseg_ctx* ctx, const uint16_t* const* samples, int64_t n_samples, int64_t n_sites,
                            char* err, size_t errlen);
int wgbsseg_set_betas_device(wgbsseg_ctx* ctx, const void* base, int64_t n_samples, int64_t pitch_bytes,
                             int64_t n_sites, char* err, size_t errlen);

/* loci[i] = bp position of CpG i+1 (column 2 of CpG.bed.gz; what `tabix | cut -f2` feeds load_dists, segmentor.cpp:36-48). */
int wgbsseg_set_loci_host(wgbsseg_ctx* ctx, const uint32_t* loci, int64_t n_sites, char* err, size_t errlen);
int wgbsseg_set_loci_device(wgbsseg_ctx* ctx, const void* loci, int64_t n_sites, char* err, size_t errlen);

/*
 * Segment n_chunks chunks of the resident data.  Chunk c covers 0-based sites
 * [chunk_start0[c], chunk_start0[c] + chunk_len[c])  (== segmentor's -s / -n).
 * Output (CSR): chunk c's borders — ascending, relative to its start, first 0, last chunk_len[c], exactly the
 * integers the reference prints (segmentor.cpp:30-34) — are borders_out[borders_off[c] .. borders_off[c+1]).
 * borders_off has n_chunks+1 entries; borders_cap >= sum(chunk_len) + n_chunks always suffices.
 * Errors the reference also raises: #meth > #cov inside a requested chunk (message names sample and site).
 * Rejected up front (WGBSSEG_E_ARG), every one of them:
 *   - max_bp == 0: the reference then reads loci it never loaded (segmentor.cpp:38,114: undefined behaviour); unreachable from its CLI (segment.py:65-66);
 *   - max_cpg < 1;
 *   - pseudo_count < 0 or NaN: the reference computes SOMETHING there (p = (nmeth + pc) / (ntotal + 2 pc) may leave [0, 1]: log2 of a negative number is
 *     NaN, the double sum goes NaN and its comparisons decide the borders), but nothing a user can mean — `-ps` is a pseudo COUNT, segment.py:276 defaults
 *     it to 15 — and the exactness proofs of csrc/exact_log2.h cover pseudo counts >= 0 only.  A deliberate divergence: an error here, garbage there;
 *   - a chunk that is empty, longer than 2^30 sites or outside the resident sites; NULL pointers; n_chunks < 1;
 *   - blocks of more than WGBSSEG_MAX_CPG =
 * 65535 sites — i.e. min(max_cpg, longest chunk of the call) > 65535 (a window never exceeds its chunk, segmentor.cpp:110).
 * The reference sizes its ring to any max_cpg (segmentor.cpp:92-95); but with 255 * block sites >= 2^24 (65,794 sites) its own
 * float sums of the counts (segmentor.cpp:122-123) stop being exact, so nothing above that is reproducible from prefix sums
 * (SURVEY.md 8b grants this rejection), and windows are stored in 16 bits here, which moves the line from 65,793 to 65,535.
 * Up to round 2 the limit was 8000 (block totals < 2^21, which the guard-free form of the likelihood term rests on,
 * csrc/exact_log2.h); windows beyond that now score with the general guarded form (same bits, ~45 % more instructions).
 */
int wgbsseg_segment_chunks(wgbsseg_ctx* ctx, const int64_t* chunk_start0, const int32_t* chunk_len,
                           int64_t n_chunks, const wgbsseg_params* params,
                           int32_t* borders_out, int64_t borders_cap, int64_t* borders_off,
                           char* err, size_t errlen);

/*
 * Region-level form of the whole driver loop of segment.py:137-165 (SegmentByChunks.run + merge_df_list +
 * stitch_2_dfs), native: regions (1-based half-open CpG ranges [region_start, region_end), e.g. one per chromosome)
 * are cut into chunks `range(start, end, chunk_size) + [end]` (segment.py:124-135), every chunk and every
 * junction's first-attempt patch [b-p1, b+p2), p = min(50, operand span) (segment.py:209-216) go to the GPU as one
 * batch, then the junctions are stitched in the reference's pairwise order with its overlap / merge2 /
 * patch-doubling rules (segment.py:219-252); failed attempts are re-batched.  Output (CSR over regions): the
 * merged ABSOLUTE 1-based border list of each region as int32 (first region_start, last region_end); consecutive pairs
 * are the blocks (startCpG, endCpG) the reference writes (segment.py:154).  borders_cap >= sum(region lengths) +
 * n_regions always suffices.  stats (optional, 8 x int64): chunks, patch DPs run, GPU batches, patches planned up
 * front, host wall microseconds of the whole call / of the first batch / of the follow-up batches, total borders.
 * wgbsseg_get_timings() afterwards returns the sums over all batches of the call.
 */
int wgbsseg_segment_regions(wgbsseg_ctx* ctx, const int64_t* region_start, const int64_t* region_end, int64_t n_regions,
                            int64_t chunk_size, const wgbsseg_params* params,
                            int32_t* borders_out, int64_t borders_cap, int64_t* borders_off, int64_t* stats,
                            char* err, size_t errlen);

/*
 * The native chunk grid + junction stitching of wgbsseg_segment_regions around a CALLER-SUPPLIED chunk engine: `fn` is
 * called with batches of 1-based half-open site ranges (chunks and junction patches) and must set, for every range i,
 * out_ptr[i] / out_cnt[i] to the range's border list RELATIVE to its start (int32, ascending, first 0, last end-start —
 * what `segmentor` prints, segmentor.cpp:30-34); the lists must stay valid until wgbsseg_stitch_regions returns.
 * Non-zero return of `fn` aborts.  Used by the multi-process driver (rank 0 stitches what the ranks' GPUs produced) and
 * by the CPU test-suite (engine = the oracle).  speculate != 0: ask for a junction's possible second attempts together
 * with the first (fewer, larger batches).  No device is touched by this call itself.
 */
typedef int (*wgbsseg_batch_fn)(void* user, const int64_t* starts, const int64_t* ends, int64_t n,
                                const int32_t** out_ptr, int64_t* out_cnt);
int wgbsseg_stitch_regions(const int64_t* region_start, const int64_t* region_end, int64_t n_regions, int64_t chunk_size,
                           wgbsseg_batch_fn fn, void* user, int32_t speculate, int32_t* borders_out, int64_t borders_cap,
                           int64_t* borders_off, int64_t* stats, char* err, size_t errlen);

/*
 * The site ranges of the FIRST batch wgbsseg_segment_regions / wgbsseg_stitch_regions hand to their chunk engine for these
 * regions: the chunks of the grid (segment.py:124-135; *n_chunks of them, region by region), then the junction patches
 * planned up front (first attempts, and with speculate != 0 the three possible second attempts: segment.py:209-227), 1-based
 * half-open, no repeats.  A pure function of the arguments: the ranks of a multi-process run (one process per GPU) each
 * work out this list, compute the items whose first site they hold, and rank 0 hands the gathered lists to
 * wgbsseg_stitch_regions as its first batch.  starts / ends may be NULL to ask for *n_items only.
 */
int wgbsseg_first_batch_items(const int64_t* region_start, const int64_t* region_end, int64_t n_regions, int64_t chunk_size,
                              int32_t speculate, int64_t* starts, int64_t* ends, int64_t cap, int64_t* n_items, int64_t* n_chunks,
                              char* err, size_t errlen);

/* Absolute 0-based index of resident site 0 (default 0).  Coordinates of every call stay relative to the resident data;
 * the base only makes error messages ("invalid data ... site N") name absolute sites when a context holds a slice. */
int wgbsseg_set_site_base(wgbsseg_ctx* ctx, int64_t site_base);

/*
 * Share groups: the whole `Pool(threads)` of segment.py:144-146 spread over several GPUs from ONE process (the
 * reference's `-@`).  A group owns one context per share (devices[d]: a device may appear more than once).
 *   wgbsseg_group_plan     cuts the chunk grid of the regions (segment.py:124-135) into n_shares contiguous runs of chunks
 *                          balanced by WORK (the number of candidate blocks the chunks hold, counted from the host loci
 *                          with the window rule of segmentor.cpp:111-117, plus a per-site term), uploads every share's
 *                          window of the loci, and reports the 0-based site window [win_lo, win_hi) each share must hold:
 *                          its chunks +- `halo` sites (halo < 0: max(chunk_size, 4096)), lower edge on a multiple of 128.
 *   wgbsseg_group_load_host        every share uploads ITS window of the beta bytes (samples[s] = whole-genome array of
 *                          sample s, e.g. a memory-mapped .beta file), all shares side by side.
 *   wgbsseg_group_share_set_device lends share `share` a device buffer that holds exactly its window: row s at
 *                          base + s * pitch_bytes, n = win_hi - win_lo sites (base, pitch multiples of 16).
 *   wgbsseg_group_segment_regions  = wgbsseg_segment_regions over the planned regions: every batch of the stitching loop
 *                          (chunks + junction patches, then the follow-ups) is routed item by item to the share holding
 *                          it and runs on one host thread per share; the reference's pairwise tree (segment.py:157-165)
 *                          runs ONCE on the host over all results, so the borders are those of a one-GPU run whatever
 *                          the number of shares.  No device-to-device traffic.  A junction patch that outgrows the halo
 *                          (patch doubling past `halo` sites at a share boundary) fails with WGBSSEG_E_STATE.
 * share_chunks / share_work (optional, n_shares each): chunks and work units given to each share.
 */
typedef struct wgbsseg_group wgbsseg_group;
/* The planner on its own (host only, no device): own_lo/own_hi = 0-based sites [lo, hi) of the chunks given to each of the
 * n_shares shares (hi == lo: none), win_* = those +- halo.  The multi-process driver (one rank per GPU) cuts the genome
 * with it so that every rank computes the same shares. */
int wgbsseg_plan_shares(const uint32_t* loci, int64_t n_sites, const int64_t* region_start, const int64_t* region_end,
                        int64_t n_regions, int64_t chunk_size, const wgbsseg_params* params, int32_t n_shares, int64_t halo,
                        int64_t* own_lo, int64_t* own_hi, int64_t* win_lo, int64_t* win_hi, int64_t* share_chunks,
                        int64_t* share_work, char* err, size_t errlen);
/* The same with unequal targets: share d takes weights[d] / sum(weights) of the work (NULL: equal shares).  The multi-process driver
 * gives rank 0 — which also runs the stitching tree of every step — a smaller share than the other ranks. */
int wgbsseg_plan_shares_weighted(const uint32_t* loci, int64_t n_sites, const int64_t* region_start, const int64_t* region_end,
                                 int64_t n_regions, int64_t chunk_size, const wgbsseg_params* params, int32_t n_shares, const double* weights,
                                 int64_t halo, int64_t* own_lo, int64_t* own_hi, int64_t* win_lo, int64_t* win_hi,
                                 int64_t* share_chunks, int64_t* share_work, char* err, size_t errlen);
int wgbsseg_group_create(const int32_t* devices, int32_t n_shares, wgbsseg_group** out, char* err, size_t errlen);
void wgbsseg_group_destroy(wgbsseg_group* g);
int32_t wgbsseg_group_size(const wgbsseg_group* g);
int wgbsseg_group_plan(wgbsseg_group* g, const uint32_t* loci, int64_t n_sites, const int64_t* region_start,
                       const int64_t* region_end, int64_t n_regions, int64_t chunk_size, const wgbsseg_params* params,
                       int64_t halo, int64_t* win_lo, int64_t* win_hi, int64_t* share_chunks, int64_t* share_work,
                       char* err, size_t errlen);
int wgbsseg_group_load_host(wgbsseg_group* g, const uint8_t* const* samples, int64_t n_samples, int64_t n_sites,
                            char* err, size_t errlen);
/* The same upload without waiting for it: returns once the device rows exist; every share's bytes then arrive front to back
 * on background threads (site-major: piece k of every sample before piece k+1 of any), and wgbsseg_group_segment_regions
 * segments what is resident while the rest is on its way (the first batch of a share runs as a few sub-batches).  The
 * sample buffers must stay valid until wgbsseg_group_segment_regions or wgbsseg_group_load_wait has returned. */
int wgbsseg_group_load_host_async(wgbsseg_group* g, const uint8_t* const* samples, int64_t n_samples, int64_t n_sites,
                                  char* err, size_t errlen);
int wgbsseg_group_load_wait(wgbsseg_group* g, char* err, size_t errlen);
int wgbsseg_group_share_set_device(wgbsseg_group* g, int32_t share, const void* base, int64_t n_samples,
                                   int64_t pitch_bytes, char* err, size_t errlen);
int wgbsseg_group_segment_regions(wgbsseg_group* g, int32_t* borders_out, int64_t borders_cap, int64_t* borders_off,
                                  int64_t* stats, char* err, size_t errlen);

/*
 * The same over a SLICE of the planned regions, [first_region, end_region): regions never interact (segment.py:84-86,129-134), so a caller may
 * take them in slices and do something with one slice's blocks — write their BED rows — while the next slice is segmented and the beta bytes of
 * the later regions are still on their way up (wgbsseg_group_load_host_async streams site-major): `wgbstools segment` does (round 6).  borders_off has
 * end_region - first_region + 1 entries.  The slices of one pass must be taken in ascending order; the uploaders are collected with the last one.
 */
int wgbsseg_group_segment_region_range(wgbsseg_group* g, int64_t first_region, int64_t end_region, int32_t* borders_out, int64_t borders_cap,
                                       int64_t* borders_off, int64_t* stats, char* err, size_t errlen);
int wgbsseg_group_get_timings(const wgbsseg_group* g, int32_t share, wgbsseg_timings* out);

/*
 * One-shot form (host buffers in, host borders out): creates a context on `device`, uploads, segments, destroys.
 * betas = [n_samples][sample_pitch_bytes] host bytes, each row holding n_sites_total x 2 uint8.
 */
int wgbsseg_segment_chunks_host(const uint8_t* betas, int64_t n_samples, int64_t sample_pitch_bytes,
                                int64_t n_sites_total, const uint32_t* loci,
                                const int64_t* chunk_start0, const int32_t* chunk_len, int64_t n_chunks,
                                const wgbsseg_params* params, int device,
                                int32_t* borders_out, int64_t borders_cap, int64_t* borders_off,
                                char* err, size_t errlen);

/*
 * Materialised per-sample prefix sums of one site range (the quantity the scan pass and the LDS tiles of the
 * scoring kernel are built from; also what a block-sum reduction such as beta_to_blocks.py:101-105 needs):
 * out[s][t][0] = sum_{u<t} meth_s[start0+u], out[s][t][1] = sum_{u<t} cov_s[start0+u], t = 0..len  (uint32).
 * out is a HOST buffer of n_samples*(len+1)*2 uint32.
 */
int wgbsseg_prefix_sums(wgbsseg_ctx* ctx, int64_t start0, int64_t len, uint32_t* out, char* err, size_t errlen);

/* The scan/validation pass alone over the given chunks (bandwidth benchmark): runs it `repeat` times and
 * returns the mean HIP-event time per launch in *ms_per_launch and the algorithmic bytes per launch (2 bytes per
 * sample and site read).  want_carry 0: the read-only pass of a job without wide scoring tiles (k_validate: what is left
 * of segmentor.cpp:164-190 there is the read and its `meth > cov` abort); 1: the prefix-sum pass proper (k_scan: per-sample
 * prefix sums of (meth, cov), one carry per 128 sites written, the same validation) as jobs with windows beyond 252 sites
 * run it; *carry_bytes_per_launch = the carries it writes (implementation traffic, not credited as algorithmic). */
int wgbsseg_scan_only(wgbsseg_ctx* ctx, const int64_t* chunk_start0, const int32_t* chunk_len, int64_t n_chunks,
                      int repeat, int want_carry, double* ms_per_launch, int64_t* bytes_per_launch,
                      int64_t* carry_bytes_per_launch, char* err, size_t errlen);

/*
 * Block sums: (#meth, #cov) of every block in every resident sample — the reduction of the reference's
 * beta_to_blocks / beta_to_table (beta_to_blocks.py:101-126 reduce_data: np.add.reduceat over the (meth, cov) rows,
 * or the per-row slice sums of its slow_method for tables that are not "nice"), the immediate consumer of the BED this
 * library produces.  Blocks are 0-based half-open site ranges [start0, end0) = [startCpG-1, endCpG-1); any order,
 * overlaps and empty blocks (sum 0: the reference's NA rows) are allowed.  out is a HOST buffer [n_samples][n_blocks]:
 *   mode 0  uint32[2]   the sums
 *   mode 1  uint8[2]    .bin  rows: cov > 255   -> meth = trunc(meth / cov * 255),   cov = 255   (utils_wgbs.py:277-290)
 *   mode 2  uint16[2]   .lbeta rows: the same with 65535
 *   mode 3  double      meth / cov, NaN where cov < min_cov                                    (utils_wgbs.py:270-274)
 */
int wgbsseg_block_sums(wgbsseg_ctx* ctx, const int64_t* start0, const int64_t* end0, int64_t n_blocks, int32_t mode,
                       uint32_t min_cov, void* out, char* err, size_t errlen);
/*
 * find_markers' per-block group statistics (find_markers.py:188-196, :318-335) over the ratio table the LAST wgbsseg_block_sums
 * call left on the device (it must have been a mode-3 call over n_blocks blocks): for a target set and a background set of
 * sample indexes (argument order of the setter; visited in the given order) out[b] = {n_tg, sum_tg, min_tg, max_tg, n_bg, sum_bg,
 * min_bg, max_bg} as doubles — the number of samples with a value (coverage >= min_cov), their sequential sum (nanmean =
 * sum / n), smallest and largest value (NaN when none).  out is a HOST buffer of n_blocks * 8 doubles.
 */
int wgbsseg_marker_stats(wgbsseg_ctx* ctx, const int32_t* tg, int32_t n_tg, const int32_t* bg, int32_t n_bg, int64_t n_blocks,
                         double* out, char* err, size_t errlen);
/* HIP-event time of the kernel of the last wgbsseg_block_sums call (ms) */
double wgbsseg_last_block_sums_ms(const wgbsseg_ctx* ctx);

/*
 * BED regions -> CpG index ranges: the join of `wgbstools convert -L` (convert.py:147-185 chr_thread, :133-145
 * slow_conversion + genomic_region.py:126-161), the step users run right before `segment -L` and `beta_to_blocks`; the
 * inverse direction is wgbsseg_add_loci below.  Against the loci resident in `ctx` (wgbsseg_set_loci_*; no betas
 * needed).  Region i lies on the chromosome whose CpGs are the 0-based sites [chrom_lo[i], chrom_hi[i]) (equal: unknown
 * chromosome -> NA) and whose length is chrom_bp[i] base pairs; (start[i], end[i]) is its bp interval.  slow[i] selects the
 * reference's rule set for the row's chromosome — it uses the as-of joins when the chromosome's regions do not overlap
 * (slow 0): startCpG = first CpG with locus >= start, endCpG = first CpG with locus >= end (+1 when exactly on `end`; last
 * CpG of the chromosome + 1 when none); and one GenomicRegion per row when they do (slow 1): CpGs with start <= locus <= end,
 * endCpG = last + 1 (last itself when its locus == end); end <= start, start < 1, end > chrom_bp: no answer.
 * Output: 1-based global indexes; (0, 0) where the reference writes NA (no CpG inside).
 */
int wgbsseg_convert_regions(wgbsseg_ctx* ctx, const int64_t* chrom_lo, const int64_t* chrom_hi, const int64_t* chrom_bp,
                            const int64_t* start, const int64_t* end, const uint8_t* slow, int64_t n,
                            int64_t* start_cpg, int64_t* end_cpg, char* err, size_t errlen);

/*
 * Blocks -> BED rows: the path's last step (segment.py:186-190 -> convert.py:242-248 add_bed_to_cpgs, which pipes the
 * blocks through the reference's `add_loci` binary: src/cpg2bed/add_loci.cpp:22-57, cpg_dict.cpp:40-131).  Host-side,
 * no device involved, no ctx needed.  For every block i (1-based half-open CpG interval) one row
 *     chrom \t loci[startCpG-1] \t loci[endCpG-2]+1 \t startCpG \t endCpG \n        (start+2 when endCpG == startCpG)
 * appended to `path` (NULL: stdout), in input order.  chrom_cum = cumulative CpG counts of the chromosomes in
 * CpG.chrome.size order (last = n_sites).  Validations and messages are the reference's (add_loci.cpp:38-49,
 * cpg_dict.cpp:130): on the first offending row the rows before it have been written and the call returns
 * WGBSSEG_E_ARG with err = "[wt add_loci] line N: endCpG < startCpG | startCpG < 1 | endCpG < 1 | Cross chromosomes"
 * or "[ cpg_dict ] Could not find chromosome for site: N".  threads <= 0: all host cores.
 */
int wgbsseg_add_loci(const uint32_t* loci, int64_t n_sites, const int64_t* chrom_cum, const char* const* chrom_names,
                     int32_t n_chroms, const int64_t* start_cpg, const int64_t* end_cpg, int64_t n_blocks,
                     const char* path, int32_t append, int32_t threads, char* err, size_t errlen);

/* The same rows straight from the merged border lists of a segmentation (round 4) — what wgbsseg_segment_regions /
 * wgbsseg_group_segment_regions leave: region r's ascending 1-based borders are borders[borders_off[r] .. borders_off[r+1]) — without
 * building (start, end) arrays first: a row is a pair of consecutive borders of a region (segment.py:154), written when
 * endCpG - startCpG >= min_cpg (the filter of segment.py:172-175 dump_result); *n_written / *n_dropped count both kinds (the numbers of
 * dump_result's stderr summary).  The regions must be given in ascending order (the rows come out sorted by startCpG as segment.py:169
 * sorts them; refused otherwise).  Same validations, messages, path / append / threads as wgbsseg_add_loci. */
int wgbsseg_add_loci_borders(const uint32_t* loci, int64_t n_sites, const int64_t* chrom_cum, const char* const* chrom_names,
                             int32_t n_chroms, const int32_t* borders, const int64_t* borders_off, int64_t n_regions,
                             int64_t min_cpg, const char* path, int32_t append, int32_t threads,
                             int64_t* n_written, int64_t* n_dropped, char* err, size_t errlen);

/*
 * The text either side of the block reduction (`wgbstools beta_to_table`: beta_to_table.py:59-127, `beta_to_blocks --bedGraph`:
 * beta_to_blocks.py:112-126; the reference reads the blocks table with pandas.read_csv and prints with DataFrame.to_csv).  Host
 * side, no device, no ctx.
 *
 * wgbsseg_blocks_parse: the bytes of a blocks table (tab-separated chr, start, end, startCpG, endCpG [, more]; '#' comments,
 * blank lines and a header line are skipped) -> per row the offset of its first byte (line_off), the length of its
 * "chr \t start \t end" text (len3), the two CpG columns and na = 1 where one of them is missing (NA, empty, ...).  A FAST PATH:
 * returns WGBSSEG_OK with *n_rows rows (at most max_rows when max_rows >= 0), or 1 when the text is not a plain table — a row
 * with fewer than five fields, a CpG field that is neither digits nor a missing-value spelling, carriage returns, non-ASCII
 * bytes, no row at all — and the caller then parses it line by line (the Python host owns those cases and their messages).
 * WGBSSEG_E_ARG: more than `cap` rows (cap = number of '\n' + 1 always suffices).
 * Optional outputs (NULL: not wanted): bp_start / bp_end = the rows' second and third fields as integers, valid when *bp_ok = 1
 * (every row's are plain digits) — `find_markers` filters blocks by their length in base pairs; *first_fields = fields (at most
 * 7) of the first line that is neither a comment nor blank: a table has the two annotation columns when that line has 7.
 *
 * wgbsseg_blocks_write_table: one output row per parsed row r: its "chr \t start \t end" bytes, startCpG, endCpG (NA where
 * na[r]), then values[r * stride + c], c < n_cols, as printf("%.<digits>f") (NaN: NA), tab-separated, '\n'; appended to `path`
 * (append 0: the file is truncated first; path NULL: written to the process's standard output).  The digits are those of the exact binary value rounded half to even, as glibc's
 * printf and Python's % operator print them.  wgbsseg_blocks_write_bedgraph: chr, start, end, meth / cov as %.2f (-1 for
 * 0 / 0), cov — from rows of uint8 (wide 0) or uint16 (wide 1) pairs.  threads <= 0: all host cores (at most 32).
 * wgbsseg_format_fixed: the number formatter alone, one value per line into `out` (test hook); returns the bytes written, -1
 * when out_cap is too small.
 */
int wgbsseg_blocks_parse(const char* text, int64_t len, int64_t max_rows, int64_t cap, int64_t* line_off, int32_t* len3,
                         int64_t* start_cpg, int64_t* end_cpg, uint8_t* na, int64_t* n_rows,
                         int64_t* bp_start, int64_t* bp_end, int32_t* bp_ok, int32_t* first_fields);
int wgbsseg_blocks_write_table(const char* path, int32_t append, const char* text, const int64_t* line_off, const int32_t* len3,
                               const int64_t* start_cpg, const int64_t* end_cpg, const uint8_t* na, int64_t n_rows,
                               const double* values, int64_t n_cols, int64_t stride, int32_t digits, int32_t threads, char* err, size_t errlen);
int wgbsseg_blocks_write_bedgraph(const char* path, const char* text, const int64_t* line_off, const int32_t* len3, int64_t n_rows,
                                  const void* rows, int32_t wide, int32_t threads, char* err, size_t errlen);
int64_t wgbsseg_format_fixed(const double* v, int64_t n, int32_t digits, char* out, int64_t out_cap);

/*
 * The text of `wgbstools convert -L` (convert.py:44-89: pandas.read_csv of a BED table, the two CpG columns inserted after the
 * third, DataFrame.to_csv).  Host side.  wgbsseg_bed_parse: rows of a BED table whose text can go back out VERBATIM around the
 * new columns -> per row its offset, the length of "chr \t start \t end" (len3) and of the whole row (row_len), the index of
 * its chromosome in chrom_names (-1: none of them), start and end.  A FAST PATH like wgbsseg_blocks_parse: returns 1 — and the
 * caller's own parser takes over — for anything a round trip through pandas would re-print: '#' comments, rows
 * of different widths, starts / ends that are not plain integers, columns that read as numbers and would print differently
 * (integers with gaps come back as floats, 0.50 as 0.5, +5 as 5; plain integers and decimals in their shortest form pass) or
 * hold missing-value spellings other than NA, carriage returns, non-ASCII bytes, an empty table.  *width = fields per row;
 * *header = 1 when the first line was a header (second and third fields not numbers): it is skipped, the caller prints the
 * reference's note, and every column of such a table is text.
 * wgbsseg_bed_write_annotated: row r as its first len3 bytes, \t startCpG \t endCpG (NA for 0), the rest of the row, \n — to
 * `path` (truncated first; NULL: standard output).
 */
int wgbsseg_bed_parse(const char* text, int64_t len, int64_t cap, const char* const* chrom_names, int32_t n_chroms, int64_t* line_off,
                      int32_t* len3, int32_t* row_len, int32_t* chrom, int64_t* start, int64_t* end, int64_t* n_rows, int32_t* width, int32_t* header);
/* test hook: out[i] = 1 when the i-th newline-separated token is a decimal number that prints as it reads after a round trip
 * through a float (csrc/table_io.h: canonical_float); returns the number of tokens, -1 when out_cap is too small */
int64_t wgbsseg_debug_canonical_float(const char* tokens, int64_t len, uint8_t* out, int64_t out_cap);
int wgbsseg_bed_write_annotated(const char* path, const char* text, const int64_t* line_off, const int32_t* len3, const int32_t* row_len,
                                const int64_t* start_cpg, const int64_t* end_cpg, int64_t n_rows, int32_t threads, char* err, size_t errlen);

/*
 * pat -> beta: the producer of the path's input (`wgbstools pat2beta`: pat2beta.py:17-44 pipes `gunzip -c x.pat.gz` into
 * the reference's stdin2beta binary, src/pat2beta/stdin2beta.cpp:59-123, and trims the counts with trim_to_uint8,
 * utils_wgbs.py:277-290).  An accumulator holds (#meth, #cov) of the CpGs [start_cpg, end_cpg) (1-based, half-open; the whole
 * genome: [1, nr_sites + 1)) on one device.  feed(): a chunk of pat TEXT made of whole lines
 * "chr \t first CpG index \t pattern over {C,T,H,.} \t count [\t ...]": every site under a C / T / H gains `count`
 * coverage, under C / H also `count` methylated; reads outside the range are skipped, empty lines too.  Asynchronous (the
 * text is copied): the caller decompresses the next chunk meanwhile.  finish(): `.beta` rows (uint8 pairs; lbeta != 0:
 * `.lbeta`, uint16 pairs) into out[2 * (end - start)], coverage above the type's maximum M scaled to (trunc(meth / cov * M), M).
 * A line with fewer than four fields, or whose site / count is not a number, is an error (the reference prints "failed
 * calculating beta" and writes nothing): finish() returns WGBSSEG_E_ARG naming the byte offset of the line.
 */
typedef struct wgbsseg_patbeta wgbsseg_patbeta;
int wgbsseg_patbeta_create(int device, int64_t start_cpg, int64_t end_cpg, wgbsseg_patbeta** out, char* err, size_t errlen);
int wgbsseg_patbeta_feed(wgbsseg_patbeta* pb, const char* text, int64_t n_bytes, char* err, size_t errlen);
int wgbsseg_patbeta_finish(wgbsseg_patbeta* pb, int32_t lbeta, void* out, char* err, size_t errlen);
void wgbsseg_patbeta_destroy(wgbsseg_patbeta* pb);
/* device time (ms, HIP events) of the counting-kernel launches of all chunks fed so far; waits for them.  < 0: error */
double wgbsseg_patbeta_kernel_ms(wgbsseg_patbeta* pb);

int wgbsseg_get_timings(const wgbsseg_ctx* ctx, wgbsseg_timings* out);

/*
 * Test hooks (used by tests/ to compare device intermediates with the oracle; not part of the drop-in surface).
 * wgbsseg_debug_fetch copies an intermediate of the LAST wgbsseg_segment_chunks() call to a host buffer:
 *   "window"  uint16[sites]   F_k (number of admissible ends of a block starting at k)
 *   "cum"     uint32[sites]   exclusive prefix of F inside each chunk (row offset of start site k)
 *   "back"    uint16[sites]   i+1-argmax_k of M[i+1]
 *   "cost"    double[pairs]   scored blocks, start-major CSR (row k = ends k .. k+F_k-1), only if the call ran as a
 *                             single stage
 * Returns the number of bytes written, or a negative code.
 * wgbsseg_debug_sample_terms evaluates the per-(block,sample) term on the device for arrays of (nmeth, ntotal).
 * wgbsseg_debug_log2 evaluates the device arithmetic for `count` consecutive float bit patterns p starting at
 * `first_bits`: out_f = uint32 bits of log2f(p); out_d = uint64 bits of the exact log2(1.0-(double)p) restatement;
 * out_fast = uint64 bits of the bounded-error fast log2 the scoring kernel tries first (any of them may be NULL).
 */
int64_t wgbsseg_debug_fetch(wgbsseg_ctx* ctx, const char* what, void* out, int64_t cap_bytes);
int wgbsseg_debug_sample_terms(wgbsseg_ctx* ctx, const float* nmeth, const float* ntotal, int64_t count,
                               float pseudo_count, float* out);
int wgbsseg_debug_log2(wgbsseg_ctx* ctx, uint32_t first_bits, int64_t count, uint32_t* out_f, uint64_t* out_d,
                       uint64_t* out_fast);
/* wgbsseg_debug_div: the scoring kernel's 8-instruction fp32 division core and the compiler's IEEE `/`, both evaluated
 * on the device for `count` operand pairs (uint32 bit patterns out). */
int wgbsseg_debug_div(wgbsseg_ctx* ctx, const float* a, const float* b, int64_t count, uint32_t* out_fast, uint32_t* out_ieee);
/* wgbsseg_debug_check_div: the verdict the library takes once per context and pseudo count before it lets the narrow scoring
 * tiles use the 4-instruction division core (v_rcp_f32, quotient, one residual correction): the number of operand pairs
 * a = fl(nmeth + pc), b = fl(ntotal + 2 pc), 0 <= nmeth <= ntotal <= max_total, whose quotient differs from IEEE `/`
 * (all evaluated on the device).  The library uses the short core only when this is 0 for max_total = 255 * 60. */
int wgbsseg_debug_check_div(wgbsseg_ctx* ctx, float pseudo_count, int32_t max_total, int64_t* mismatches);
/* wgbsseg_debug_div_short: the short core itself on arrays of operands (bit patterns of the quotients). */
int wgbsseg_debug_div_short(wgbsseg_ctx* ctx, const float* a, const float* b, int64_t count, uint32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* WGBSSEG_H */
