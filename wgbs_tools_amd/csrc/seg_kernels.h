// seg_kernels.h — gfx950 kernels of the `wgbstools segment` hot path.
//
// Pipeline for one batch of chunks (a chunk = what one reference `segmentor` process handles, segmentor.cpp:193-214):
//   k_scan    per-sample prefix scan of (#meth,#cov) with 128-site carries + the #meth<=#cov validation of
//             read_beta_file (segmentor.cpp:179-188).  HBM-bound: reads every beta byte once, writes 1/16 of that.
//   k_window  forward window F_k of every site from the loci (the bp/CpG limits of segmentor.cpp:111-117: the
//             extensions of a block starting at k that are admissible); k_window_scan + k_window_cum: the CSR row offsets
//             of the scored-block matrix, which is START-major like the reference's own rows (segmentor.cpp:103-138).
//   k_cost    block log-likelihoods (segmentor.cpp:119-137) for every (start k, end i) inside the window, scored
//             from LDS-staged prefix tiles; samples are visited in file order inside each lane (the double
//             accumulation order is part of the bit-exactness contract).  fp64-VALU bound.
//   k_dp      the changepoint recurrence (segmentor.cpp:142-154) in PUSH form: one wavefront owns one chunk; lane l
//             holds the running maximum of the pending step i == l (mod 64); when M[k] is final, ONE vector add +
//             compare folds candidate k into the 64 steps it can start (ascending k and strict '>' = the reference's
//             first-maximum rule); M[k+1] is then read from lane k mod 64.  No cross-lane reduction on the chain.
//             Blocks of 65..128 sites use a second pending register; longer ones are pushed by the worker waves.
//   k_trace   traceback (segmentor.cpp:50-58) out of an LDS-resident window of back-pointers, cut into segments that
//             are walked speculatively in parallel and joined by one short sequential pass.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "exact_log2.h"
#include "wave_prims.h"

#ifndef WG_CARRY_SHIFT
#define WG_CARRY_SHIFT  7           // (-DWG_CARRY_SHIFT=8 .. 10: A/B builds, tools/build_carry_libs.sh)
#endif
#define WG_SCAN_STAGED  512         // carry groups of a chunk that k_scan keeps in LDS until the row is done
#define WG_CARRY_G      (1 << WG_CARRY_SHIFT)   // a carry (chunk-relative exclusive prefix) is stored at every absolute site index
                                    // that is a multiple of WG_CARRY_G inside the chunk, plus (group 0) at the chunk start itself
static_assert(WG_CARRY_SHIFT >= 4 && WG_CARRY_SHIFT <= 10, "k_scan: a lane vector is 16 sites, an iteration 1024");
#define WG_BLOCK        256
#define WG_WIN_TILE     1024        // sites per k_window workgroup
#define WG_PAIR_CAP     4096        // candidate blocks per k_cost tile (bounds the LDS partial-sum array)
#define WG_TRACE_WIN    32768       // back-pointers staged in LDS by k_trace (64 KiB)
#define WG_NARROW_WMAX  60          // widest window of a narrow scoring tile: TI + 60 + 1 <= 125 entries, + 3 of alignment <= 128 = 32 lanes x 4 sites
#define WG_MEDIUM_WMAX  252         // widest window of a medium scoring tile (16 starts, every end): block counts <= 255 * 252 < 2^16 still fit the packed
                                    // tile-local prefixes of the narrow tiles (a block's counts are ONE difference of two dwords), no carries of k_scan
#define WG_MEDIUM_TS    16          // start sites of a medium tile = one 16-site unit

// Exact-restatement tables: global (constant) memory, read only by the rare guard-band fallback and by the plain kernel.
__device__ const wg_log_tables g_wg_tables = WG_LOG_TABLES_INIT;

// Fill the LDS-resident fast tables (all threads of the workgroup).  The caller must __syncthreads() before use.
__device__ __forceinline__ void wg_fast_tables_to_lds(wg_fast_tables* ft, int tid, int nthreads)
{
    const double* f = reinterpret_cast<const double*>(g_wg_tables.f_tab);
    const double* d = reinterpret_cast<const double*>(g_wg_tables.d_tab);
    double* of = reinterpret_cast<double*>(ft->f_tab);
    double* od = reinterpret_cast<double*>(ft->d_fast);
    for (int x = tid; x < 32; x += nthreads) of[x] = f[x];
    for (int x = tid; x < 128; x += nthreads) {
        double v = d[x];
        if (x == 2 * WG_FAST_CENTRE_ENTRY) v = 1.0;              // wg_tables_finish(): interval just below 1 centred on 1
        if (x == 2 * WG_FAST_CENTRE_ENTRY + 1) v = 0.0;
        od[x] = v;
    }
}

// The two k-scaled lookup tables of the guard-free scoring kernels (wg_log2f_ks / wg_fast_log2_ks; entries:
// wg_ks_iy_entry / wg_ks_ky_entry), `rows` exponents each, iy then ky back to back.  The host builds them once per
// call (`src`, global memory): a workgroup just copies 16-byte entries.  src == NULL: build them here (test hook).
__device__ __forceinline__ void wg_lookup_tables_to_lds(wg_d2* iy, wg_d2* ky, int rows, const wg_d2* __restrict__ src, int tid, int nthreads)
{
    if (src) {
        wg_d2* dst = iy;                                           // ky == iy + rows * 16
        for (int x = tid; x < rows * (16 + 64); x += nthreads) dst[x] = src[x];
        return;
    }
    for (int x = tid; x < rows * 16; x += nthreads) iy[x] = wg_ks_iy_entry(&g_wg_tables, rows, x);
    for (int x = tid; x < rows * 64; x += nthreads) ky[x] = wg_ks_ky_entry(&g_wg_tables, rows, x);
}

struct ChunkDesc {
    int64_t start0;      // first site (0-based, absolute)
    int64_t site_off;    // offset of this chunk in the job-site arrays (W16, cum32, back16)
    int64_t carry_off;   // offset (in uint2) of this chunk's carries: [n_samples][nG]
    int64_t unit_off;    // offset of this chunk in umax16: one entry per 16 sites of the chunk
    int32_t len;
    int32_t nG;          // carry groups (WG_CARRY_G sites of ABSOLUTE site index) the chunk touches: ((start0+len-1)>>S) - (start0>>S) + 1
};

struct JobView {
    const uint8_t* betas;     // [n_samples][pitch] bytes, row s = n_total x (meth, cov)
    int64_t pitch;
    int64_t n_total;
    const uint32_t* loci;     // [n_total]
    const ChunkDesc* chunks;
    uint2* carry;
    uint16_t* W16;            // [job sites] forward window F_k: blocks starting at k may end at k .. k+F_k-1
    uint32_t* cum32;          // [job sites] exclusive prefix of F inside the chunk = row offset of start site k
    uint16_t* back16;         // [job sites] i + 1 - argmax_k for M[i+1]  (length of the best block ending at i)
    uint16_t* umax16;         // [job 16-site units] largest F_k of the unit: decides the unit's k_cost tile class
    int64_t* chunk_pairs;     // [n_chunks] sum of W over the chunk
    int32_t n_samples;
    int32_t n_chunks;
};

struct JobStatus {            // zeroed (first_bad = ~0) before every call
    unsigned long long first_bad;   // min over bad sites of (sample << 40 | absolute site)
    unsigned long long total_pairs;
    unsigned int max_window;
    unsigned int loci_disorder;     // 1 + chunk index of a chunk whose loci are not ascending
    unsigned int overflow;          // a chunk's pair count does not fit 32 bits
    unsigned int wide_units;        // 16-site units with a window beyond the medium tiles' (> WG_MEDIUM_WMAX sites by default): they are scored in wide tiles, from the carries of k_scan
};

struct StageView {            // tables produced by k_stage_plan, row `stage` of each
    const int64_t* cbase;     // [n_stages][n_chunks] element offset of the chunk's rows in the stage's cost buffer
    const uint32_t* cum0;     // [n_stages][n_chunks] cum32 at the chunk's first site of the stage
    const int64_t* tbaseA;    // [n_stages][n_chunks+1] exclusive prefix of the narrow k_cost tiles
    const int64_t* tbaseB;    // [n_stages][n_chunks+1] exclusive prefix of the wide k_cost tiles
    int32_t stage;
    const int32_t* sb;        // [n_stages + 1] stage s covers the chunk-relative sites [sb[s], sb[s+1]) of every chunk (multiples of 64; the last bound >= the longest chunk)
};

// ------------------------------------------------------------------------------------------------------------
// k_scan: one wavefront streams one (chunk, sample) row: 64 lanes x 16 B = 512 sites per iteration.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 wg_load16_guarded(const uint8_t* row, int64_t abs_site, int64_t n_total)
{
    // 16-byte aligned load of 8 sites starting at abs_site (multiple of 8); bytes past the row end read as 0
    if (abs_site + 8 <= n_total) return *reinterpret_cast<const uint4*>(row + 2 * abs_site);
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    for (int j = 0; j < 8; j++) {
        int64_t a = abs_site + j;
        if (a < n_total) {
            uint32_t m = row[2 * a], c = row[2 * a + 1];
            w[j >> 1] |= (m | (c << 8)) << (16 * (j & 1));
        }
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// Sums of the 8 (#meth, #cov) byte pairs of a 16-byte vector, and whether any pair has #meth > #cov.
// A dword holds two sites: bytes (m0, c0, m1, c1).  Even bytes / odd bytes are summed in two 16-bit lanes.
__device__ __forceinline__ void wg_sum8(const uint4 v, uint32_t& tm, uint32_t& tt, bool& anybad)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t sm = 0, sc = 0, ok = 0x01000100u;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t m = w[d] & 0x00ff00ffu, c = (w[d] >> 8) & 0x00ff00ffu;
        sm += m; sc += c;                                  // <= 4*255 per 16-bit lane
        ok &= (c | 0x01000100u) - m;                       // bit 8 / 24 stays set iff c >= m in that lane
    }
    tm = (sm & 0xffffu) + (sm >> 16);
    tt = (sc & 0xffffu) + (sc >> 16);
    anybad = (ok & 0x01000100u) != 0x01000100u;
}

// One wavefront streams one (chunk, sample) row, 64 lanes x 32 B = 1024 sites per iteration, the next iteration in
// flight.  Lane vectors are 32-byte aligned in the sample row, so every 8th lane starts on an absolute site index that
// is a multiple of 128: that lane holds the carry of its group, no intra-vector partial sums needed; the two wave scans
// are amortised over 2 KB.  Register-lean on purpose (32-bit chunk-relative indices, the rare paths out of line):
// 8 wavefronts per SIMD, so that the 15,456 equally long rows of an hg19 x 32 job run in two rounds, not three.
__device__ __noinline__ uint4 wg_blank_outside(uint4 v, int rel0, int len)
{
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int rel = rel0 + j;
        if (rel < 0 || rel >= len) w[j >> 1] &= ~(0xffffu << (16 * (j & 1)));
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __noinline__ int wg_first_bad_site(uint4 v0, uint4 v1)      // index (0..15) of the first site with #meth > #cov
{
    const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    int bad = 16;
    for (int j = 15; j >= 0; j--) {
        const uint32_t h = w[j >> 1] >> (16 * (j & 1));
        if ((h & 0xffu) > ((h >> 8) & 0xffu)) bad = j;
    }
    return bad;
}

__global__ __launch_bounds__(WG_BLOCK) void k_scan(JobView J, JobStatus* st)
{
    const int lane = threadIdx.x & 63;
    // The carries have ONE consumer: the wide scoring tiles (windows > WG_MEDIUM_WMAX sites: deep mode, the densest islands).  The host
    // launches this kernel only for a job that has any (it knows from the windows' statistics), or for a caller that asks for the carries;
    // every other job's scan pass is the read-only k_validate, which starts with the batch.
    const int64_t rowid = (int64_t)blockIdx.x * (WG_BLOCK / 64) + (threadIdx.x >> 6);      // a wave task = one (chunk, sample) row
    const int64_t nrows = (int64_t)J.n_chunks * J.n_samples;
    if (rowid >= nrows) return;                       // whole wave leaves together
    const int c = (int)(rowid / J.n_samples);
    const int s = (int)(rowid - (int64_t)c * J.n_samples);
    const ChunkDesc cd = J.chunks[c];
    const int64_t a_abs = cd.start0 & ~15LL;          // first 32-byte aligned site of the stream
    const int head = (int)(cd.start0 - a_abs);        // sites of the first lane vector that precede the chunk
    const int len = cd.len;
    const int span = head + len;                      // sites from a_abs to the chunk end (< 2^31: checked by the host)
    // 16-byte vectors of the row from a_abs on; every vector that STARTS inside the row is readable (pitch is a
    // multiple of 16 bytes >= 2 n_total), later ones are clamped onto the last readable one and blanked below
    const uint4* rv = reinterpret_cast<const uint4*>(J.betas + (int64_t)s * J.pitch) + (a_abs >> 3);
    const int64_t vl = ((J.n_total - 1) >> 3) - (a_abs >> 3);
    const int vlast = vl > 0x7ffffff0 ? 0x7ffffff0 : (int)vl;
    // a_abs and start0 share their carry group (rounding down to 16 never crosses a multiple of 128), so the carry of
    // the group that begins at stream offset `off` is entry (A6 + off) >> WG_CARRY_SHIFT, A6 = a_abs mod the group size
    uint2* carry = J.carry + cd.carry_off + (int64_t)s * cd.nG;
    const int A6 = (int)(a_abs & (WG_CARRY_G - 1));
    uint32_t run_m = 0, run_t = 0;
    int bad_rel = 0x7fffffff;
    if (lane == 0) carry[0] = make_uint2(0u, 0u);     // group 0: the chunk start
    const int nit = (span + 1023) >> 10;              // 1024-site iterations of the row
    // The row's carries stay in LDS (up to WG_SCAN_STAGED groups = 4 KB per wavefront, 16 KB per workgroup: eight workgroups per CU
    // still fit) and leave as full-wavefront stores when the row is done: a store in every iteration of the read stream is what
    // held this pass at 0.54-0.58 of the HBM peak (round 3, hg19 x 32 with islands: 0.375-0.408 -> 0.357-0.359 ms = 0.65; the
    // volume of the stores, the instruction count and the loads in flight measured irrelevant: DESIGN.md 8.5).  A chunk of more
    // groups than that (> 65,000 sites) stores directly.
    __shared__ uint2 cstage[WG_BLOCK / 64][WG_SCAN_STAGED];
    uint2* cs = cstage[threadIdx.x >> 6];
    const bool staged = cd.nG <= WG_SCAN_STAGED;

    int vi = 2 * lane;                                // this lane's first vector of the current iteration
    uint4 c0 = rv[vi < vlast ? vi : vlast], c1 = rv[vi + 1 < vlast ? vi + 1 : vlast];
    for (int base = 0; base < (nit << 10); base += 1024) {
        const int vn = vi + 128;                      // one iteration ahead (clamped: never past the row)
        const uint4 m0 = rv[vn < vlast ? vn : vlast], m1 = rv[vn + 1 < vlast ? vn + 1 : vlast];
        uint4 v0 = c0, v1 = c1;
        const int off = base + lane * 16;             // site offset of this lane's 16 sites from a_abs
        const int rel0 = off - head;                  // chunk-relative index of the lane's first site
        if (rel0 < 0 || rel0 + 16 > len) {            // edge lane: blank the sites outside the chunk
            v0 = wg_blank_outside(v0, rel0, len);
            v1 = wg_blank_outside(v1, rel0 + 8, len);
        }
        uint32_t tm0, tt0, tm1, tt1;
        bool bad0, bad1;
        wg_sum8(v0, tm0, tt0, bad0);
        wg_sum8(v1, tm1, tt1, bad1);
        if ((bad0 || bad1) && bad_rel == 0x7fffffff) bad_rel = rel0 + wg_first_bad_site(v0, v1);   // rare
        const uint32_t tm = tm0 + tm1, tt = tt0 + tt1;
        // wave-wide inclusive prefix of the lane totals (two 32-bit DPP scans)
        const uint32_t im = wg_wave_incl_scan_dpp_u32(tm);
        const uint32_t it = wg_wave_incl_scan_dpp_u32(tt);
        // (one 8-byte store from every 8th lane.  Round 3: the same store with a non-temporal hint measured 0.432 against 0.434 ms for
        // hg19 x 32 with islands — no difference, the hint is not used)
        if (((A6 + off) & (WG_CARRY_G - 1)) == 0 && rel0 > 0 && rel0 < len) {
            const uint2 cv = make_uint2(run_m + (im - tm), run_t + (it - tt));
            if (staged) cs[(A6 + off) >> WG_CARRY_SHIFT] = cv; else carry[(A6 + off) >> WG_CARRY_SHIFT] = cv;
        }
        run_m += (uint32_t)__builtin_amdgcn_readlane((int)im, 63);
        run_t += (uint32_t)__builtin_amdgcn_readlane((int)it, 63);
        c0 = m0; c1 = m1; vi = vn;
    }
    if (staged) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int g = 1 + lane; g < cd.nG; g += 64) {
            const int rel0 = (g << WG_CARRY_SHIFT) - A6 - head;          // chunk-relative first site of group g: stored above iff inside the chunk
            if (rel0 > 0 && rel0 < len) carry[g] = cs[g];
        }
    }
    if (bad_rel != 0x7fffffff)
        atomicMin(&st->first_bad, ((unsigned long long)s << 40) | (unsigned long long)(cd.start0 + bad_rel));
}

// k_validate: the scan pass of a job WITHOUT wide tiles — nobody reads carries, what is left of segmentor.cpp:164-190 is the
// read itself and its `meth > cov` abort.  The host hands over the sites of the batch as disjoint pieces (the chunks of a
// genome tile their regions; junction patches lie inside chunks the same call has already checked and add nothing), one
// wave task = one (piece, sample): 64 lanes x 32 B per iteration, the next iteration in flight, loads clamped to the
// piece so that no byte outside it is fetched; a packed compare per dword, no sums, no scans, no stores.
// Round 6: it needs nothing from the windows pass, so a batch launches it first of all, on the scan stream beside k_window (a job that
// turns out to have wide tiles runs k_scan afterwards, which checks the same bytes again while it builds the carries).
struct ScanPiece {
    int64_t lo;          // first site (0-based, resident-relative)
    int32_t n;           // sites
    int32_t pad;
};

__device__ __forceinline__ uint32_t wg_ok8(const uint4 v, uint32_t ok)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int d = 0; d < 4; d++) {
        // a dword = two sites, bytes (m0, c0, m1, c1).  16-bit lanes (c + 0xff00) - m: bit 8 stays set iff cov >= meth (a lane
        // never borrows from its neighbour: it is >= 0xff00 - 255)
        const uint32_t c = __builtin_amdgcn_perm(0u, w[d], 0x0d030d01u);           // (c0, 0xff, c1, 0xff)
        ok &= c - (w[d] & 0x00ff00ffu);
    }
    return ok;
}

__global__ __launch_bounds__(WG_BLOCK) void k_validate(JobView J, JobStatus* st, const ScanPiece* __restrict__ pieces, int64_t n_pieces)
{
    const int lane = threadIdx.x & 63;
    const int64_t task = (int64_t)blockIdx.x * (WG_BLOCK / 64) + (threadIdx.x >> 6);
    const int64_t pi = task / J.n_samples;
    if (pi >= n_pieces) return;
    const int s = (int)(task - pi * J.n_samples);
    const ScanPiece P = pieces[pi];
    const int64_t a_abs = P.lo & ~15LL;
    const int head = (int)(P.lo - a_abs);
    const int len = P.n;
    const int span = head + len;
    const uint4* rv = reinterpret_cast<const uint4*>(J.betas + (int64_t)s * J.pitch) + (a_abs >> 3);
    const int vlast = (span - 1) >> 3;                // last vector that holds a site of the piece (it starts inside the row)
    const int nit = (span + 1023) >> 10;
    int bad_rel = 0x7fffffff;
    int vi = 2 * lane;
    uint4 c0 = rv[vi < vlast ? vi : vlast], c1 = rv[vi + 1 < vlast ? vi + 1 : vlast];
    for (int base = 0; base < (nit << 10); base += 1024) {
        const int vn = vi + 128;
        const uint4 m0 = rv[vn < vlast ? vn : vlast], m1 = rv[vn + 1 < vlast ? vn + 1 : vlast];
        uint4 v0 = c0, v1 = c1;
        const int rel0 = base + lane * 16 - head;
        if (rel0 < 0 || rel0 + 16 > len) {
            v0 = wg_blank_outside(v0, rel0, len);
            v1 = wg_blank_outside(v1, rel0 + 8, len);
        }
        const uint32_t ok = wg_ok8(v1, wg_ok8(v0, 0x01000100u));
        if ((ok & 0x01000100u) != 0x01000100u && bad_rel == 0x7fffffff) bad_rel = rel0 + wg_first_bad_site(v0, v1);   // rare
        c0 = m0; c1 = m1; vi = vn;
    }
    if (bad_rel != 0x7fffffff)
        atomicMin(&st->first_bad, ((unsigned long long)s << 40) | (unsigned long long)(P.lo + bad_rel));
}

// ------------------------------------------------------------------------------------------------------------
// k_window: F_k = number of admissible ends of a block starting at site k:
//   i admissible  <=>  k <= i < len,  i-k < max_cpg  and  loci[i]-loci[k] <= max_bp   (segmentor.cpp:111-117, loci ascending)
// Grid: 1024-site tiles (`wtile_off` = exclusive prefix of tiles per chunk), four sites per thread.  Round 6: the windows pass is
// three launches — k_window (F, per-unit maxima, the tile's total), k_window_scan (per chunk: exclusive prefix of its tiles'
// totals: 59 numbers for a 60,000-site chunk) and k_window_cum (per tile: F -> CSR row offsets) — where rounds 1-5 had one
// workgroup per chunk walk its 60,000 sites in eight dependent steps (0.10 ms of latency in front of every batch).
//
// The search.  F_k - 1 = the largest d <= hi - k with loci[k + d] - loci[k] <= max_bp, found bit by bit from the top (d |= step
// when the probe at d + step is admissible): a fixed number of probes, no data-dependent loop, uint32 differences (ascending
// loci: a chunk whose loci are not is flagged and takes the plain path, whatever this kernel made of it).  A wavefront none of
// whose sites reaches 64 sites ahead — every wavefront of a default-parameter genome outside CpG islands — starts at step 32:
// 1 + 6 probes instead of the 10 a binary search over [k, k + max_cpg) takes.
// ------------------------------------------------------------------------------------------------------------
// Staged form: the tile's loci sit in LDS from the site before the tile's first one (the order check of that site) to the last one a
// search can touch; the LDS area is as long as the farthest PROBE (tile + 2 * top_step entries), so a probe beyond a site's reach needs
// no branch: it reads whatever is there and the reach test discards it.
template <bool STAGED>
__device__ __forceinline__ void wg_window_body(const JobView& J, JobStatus* st, const ChunkDesc& cd, int c, int k0, const uint32_t* __restrict__ sloc, int kb,
                                               uint32_t max_cpg, uint32_t max_bp, int top_step, int wide_from, uint32_t& tile_total)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t* __restrict__ loc = J.loci + cd.start0;
    int reach[4], d[4];
    uint32_t lk[4];
    const uint32_t* at[4];                                       // locus of site k[j] + x: at[j][x]
    bool disorder = false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int k = k0 + j * WG_BLOCK + tid;
        at[j] = STAGED ? sloc + (k - kb) : loc + k;
        reach[j] = -1; lk[j] = 0u; d[j] = 0;
        if (k < cd.len) {
            lk[j] = at[j][0];
            if (k > 0 && at[j][-1] > lk[j]) disorder = true;
            const int64_t far_end = (int64_t)k + max_cpg - 1 < cd.len - 1 ? (int64_t)k + max_cpg - 1 : cd.len - 1;
            reach[j] = (int)(far_end - k);                       // 0 .. max_cpg - 1
        }
    }
    // does any site of the wavefront reach 64 sites ahead?
    bool far = false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (STAGED) { const uint32_t l64 = at[j][64]; far = far | ((reach[j] >= 64) & (l64 - lk[j] <= max_bp)); }      // (bitwise: the read is unconditional, four of them in flight)
        else if (reach[j] >= 64) far = far || (at[j][64] - lk[j] <= max_bp);
    }
    int step = __any(far) ? top_step : (top_step < 32 ? top_step : 32);
    for (; step > 0; step >>= 1) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int cand = d[j] + step;
            if (STAGED) {
                const uint32_t lm = at[j][cand];
                d[j] = ((cand <= reach[j]) & (lm - lk[j] <= max_bp)) ? cand : d[j];      // (bitwise: no branch around the read)
            } else if (cand <= reach[j]) {
                if (at[j][cand] - lk[j] <= max_bp) d[j] = cand;
            }
        }
    }
    uint32_t wmax_t = 0, wsum_t = 0, nwide = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int k = k0 + j * WG_BLOCK + tid;
        const bool valid = k < cd.len;
        const uint32_t w = valid ? (uint32_t)(d[j] + 1) : 0u;
        if (valid) J.W16[cd.site_off + k] = (uint16_t)w;
        // largest window of each 16-site unit = of each row of 16 lanes (DPP butterflies inside the row)
        uint32_t um = w;
        um = max(um, wg_dpp_u32<WG_DPP_QUAD_1032>(um)); um = max(um, wg_dpp_u32<WG_DPP_QUAD_2301>(um));
        um = max(um, wg_dpp_u32<WG_DPP_ROW_HMIRROR>(um)); um = max(um, wg_dpp_u32<WG_DPP_ROW_MIRROR>(um));
        const bool unit_head = (lane & 15) == 0 && valid;
        if (unit_head) J.umax16[cd.unit_off + (k >> 4)] = (uint16_t)um;
        nwide += (uint32_t)__popcll(__ballot(unit_head && um > (uint32_t)wide_from));      // (wave-uniform)
        wmax_t = max(wmax_t, w);
        wsum_t += w;
    }
    const uint32_t wmax = wg_wave_max_u32(wmax_t);
    const uint32_t wsum = wg_wave_incl_scan_dpp_u32(wsum_t);     // lane 63: the wavefront's total
    const bool any_dis = __any(disorder);
    if (lane == 0) {
        // tens of thousands of wavefronts: touch the shared words only when they would change
        if (wmax > __hip_atomic_load(&st->max_window, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&st->max_window, wmax);
        if (any_dis) atomicMax(&st->loci_disorder, (unsigned int)(c + 1));
        if (nwide) atomicAdd(&st->wide_units, nwide);
    }
    tile_total = wsum;
}

__global__ __launch_bounds__(WG_BLOCK) void k_window(JobView J, JobStatus* st, const int64_t* __restrict__ wtile_off,
                                                     const int32_t* __restrict__ wtile_hint, uint32_t max_cpg, uint32_t max_bp, int lds_cap,
                                                     int top_step, int wide_from, uint32_t* __restrict__ tile_tot, int4* __restrict__ tile_info)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t t = blockIdx.x;
    // chunk of this tile = the last c with wtile_off[c] <= t.  The host hands over the answer for every 256th tile
    // (wtile_hint), so the search runs over the chunks 256 tiles can span: one or two for 60,000-site chunks — a
    // full binary search would put nine DEPENDENT L2 loads in front of every one of the 27 k tiles of hg19
    int clo = wtile_hint[t >> 8], chi = wtile_hint[(t >> 8) + 1] + 1;
    while (chi - clo > 1) { const int mid = (clo + chi) >> 1; if (wtile_off[mid] <= t) clo = mid; else chi = mid; }
    const int c = clo;
    const ChunkDesc cd = J.chunks[c];
    const int k0 = (int)(t - wtile_off[c]) * WG_WIN_TILE;
    // the loci the tile's searches can touch, [kb, k0 + tile + max_cpg - 1) with kb = the site before the tile, staged in LDS when the
    // probes fit (lds_cap entries; the host sizes the area to 1 + tile + 2 * top_step + 8): the dependent probes of a search then cost
    // LDS latency instead of L2 latency.  16-byte loads from the aligned address below the first locus (`shift` entries earlier);
    // a vector that crosses an end of the loci array is read entry by entry.
    extern __shared__ __attribute__((aligned(16))) uint32_t sloc[];
    __shared__ uint32_t wtot[WG_BLOCK / 64];
    const int kb = k0 > 0 ? k0 - 1 : 0;
    const int64_t want = (int64_t)(k0 - kb) + WG_WIN_TILE + max_cpg - 1;
    const int span = (int)((int64_t)cd.len - kb < want ? (int64_t)cd.len - kb : want);
    const bool staged = lds_cap > 0;
    uint32_t total = 0;
    if (staged) {
        const int64_t g0 = cd.start0 + kb;                      // absolute index of the first staged locus
        const int shift = (int)((reinterpret_cast<uintptr_t>(J.loci + g0) >> 2) & 3);      // entries between the 16-byte boundary below it and the locus
        const int64_t a0 = g0 - shift;
        const int nv = (shift + span + 3) >> 2;
        for (int x = tid; x < nv; x += WG_BLOCK) {
            const int64_t a = a0 + 4 * (int64_t)x;
            uint4 v;
            if (a >= 0 && a + 4 <= J.n_total) v = *reinterpret_cast<const uint4*>(J.loci + a);
            else {
                v.x = a >= 0 ? J.loci[a] : 0u;
                v.y = (a + 1 >= 0 && a + 1 < J.n_total) ? J.loci[a + 1] : 0u;
                v.z = (a + 2 >= 0 && a + 2 < J.n_total) ? J.loci[a + 2] : 0u;
                v.w = (a + 3 >= 0 && a + 3 < J.n_total) ? J.loci[a + 3] : 0u;
            }
            *reinterpret_cast<uint4*>(sloc + 4 * x) = v;
        }
        __syncthreads();
        wg_window_body<true>(J, st, cd, c, k0, sloc + shift, kb, max_cpg, max_bp, top_step, wide_from, total);
    } else {
        wg_window_body<false>(J, st, cd, c, k0, nullptr, kb, max_cpg, max_bp, top_step, wide_from, total);
    }
    if (lane == 63) wtot[tid >> 6] = total;
    __syncthreads();
    if (tid == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (int q = 0; q < WG_BLOCK / 64; q++) tot += wtot[q];
        tile_tot[t] = tot;                                       // <= 1024 * 65,535
        // what k_window_cum needs of this tile, in one 16-byte record: where its sites sit in the job-site arrays, how many there are
        const int64_t at = cd.site_off + k0;
        const int nv = cd.len - k0 < WG_WIN_TILE ? cd.len - k0 : WG_WIN_TILE;
        tile_info[t] = make_int4((int)(uint32_t)at, (int)(at >> 32), nv, c);
    }
}

// k_window_scan: one workgroup per chunk turns its tiles' totals into their exclusive prefix (chunk-relative, the row offset of
// the tile's first site), and the chunk's total into chunk_pairs / the job's statistics.
__global__ __launch_bounds__(WG_BLOCK) void k_window_scan(JobView J, JobStatus* st, const int64_t* __restrict__ wtile_off,
                                                          const uint32_t* __restrict__ tile_tot, uint32_t* __restrict__ tile_base)
{
    __shared__ uint64_t ws[WG_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c = blockIdx.x;
    const int64_t t0 = wtile_off[c], t1 = wtile_off[c + 1];
    uint64_t run = 0;
    for (int64_t base = t0; base < t1; base += WG_BLOCK) {
        const int64_t t = base + tid;
        const uint64_t v = t < t1 ? (uint64_t)tile_tot[t] : 0;
        const uint64_t incl = wg_wave_incl_scan_u64(v, lane);
        if (lane == 63) ws[wv] = incl;
        __syncthreads();
        uint64_t o = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < WG_BLOCK / 64; q++) { if (q < wv) o += ws[q]; tot += ws[q]; }
        __syncthreads();
        if (t < t1) tile_base[t] = (uint32_t)(run + o + incl - v);
        run += tot;
    }
    if (tid == 0) {
        J.chunk_pairs[c] = (int64_t)run;
        atomicAdd(&st->total_pairs, (unsigned long long)run);
        if (run >> 32) atomicMax(&st->overflow, 1u);
    }
}

// k_window_cum: cum32[k] = the tile's base + the exclusive prefix of F inside the tile.  One workgroup takes WG_CUM_TILES consecutive
// 1024-site tiles, four consecutive sites per thread and tile; site_off and the tile starts are multiples of 8, so F arrives as one
// 8-byte load and the offsets leave as one 16-byte store per thread.  The pass is a chain of dependent loads (the tile's record, then
// its F) in front of a trivial scan: every tile's record comes from k_window in one 16-byte entry, and the loads of a workgroup's
// tiles are all in flight before the first scan (one tile per workgroup and four dependent header loads: 0.10 ms for hg19, latency).
#define WG_CUM_TILES 4
__global__ __launch_bounds__(WG_BLOCK) void k_window_cum(const uint16_t* __restrict__ W16, uint32_t* __restrict__ cum32, const int4* __restrict__ tile_info,
                                                         const uint32_t* __restrict__ tile_base, int64_t n_tiles)
{
    __shared__ uint32_t ws[WG_CUM_TILES][WG_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * WG_CUM_TILES;
    int4 info[WG_CUM_TILES];
    uint32_t base[WG_CUM_TILES];
#pragma unroll
    for (int q = 0; q < WG_CUM_TILES; q++) {
        const int64_t t = t0 + q < n_tiles ? t0 + q : n_tiles - 1;
        info[q] = tile_info[t];
        base[q] = tile_base[t];
        if (t0 + q >= n_tiles) info[q].z = 0;
    }
    uint2 v[WG_CUM_TILES];
#pragma unroll
    for (int q = 0; q < WG_CUM_TILES; q++) {
        const int64_t at = (int64_t)(((uint64_t)(uint32_t)info[q].y << 32) | (uint32_t)info[q].x) + 4 * tid;
        v[q] = make_uint2(0u, 0u);
        if (4 * tid < info[q].z) v[q] = *reinterpret_cast<const uint2*>(W16 + at);       // (entries beyond the chunk's end: inside the padded array, masked below)
    }
    uint32_t w[WG_CUM_TILES][4], tot[WG_CUM_TILES], incl[WG_CUM_TILES];
#pragma unroll
    for (int q = 0; q < WG_CUM_TILES; q++) {
        const int k = 4 * tid, n = info[q].z;
        w[q][0] = k < n ? v[q].x & 0xffffu : 0u; w[q][1] = k + 1 < n ? v[q].x >> 16 : 0u;
        w[q][2] = k + 2 < n ? v[q].y & 0xffffu : 0u; w[q][3] = k + 3 < n ? v[q].y >> 16 : 0u;
        tot[q] = w[q][0] + w[q][1] + w[q][2] + w[q][3];
        incl[q] = wg_wave_incl_scan_dpp_u32(tot[q]);
        if (lane == 63) ws[q][wv] = incl[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < WG_CUM_TILES; q++) {
        const int k = 4 * tid, n = info[q].z;
        if (k >= n) continue;
        uint32_t e = base[q] + (incl[q] - tot[q]);
#pragma unroll
        for (int x = 0; x < WG_BLOCK / 64; x++) if (x < wv) e += ws[q][x];
        const int64_t at = (int64_t)(((uint64_t)(uint32_t)info[q].y << 32) | (uint32_t)info[q].x) + k;
        uint32_t* C = cum32 + at;
        const uint4 o = make_uint4(e, e + w[q][0], e + w[q][0] + w[q][1], e + w[q][0] + w[q][1] + w[q][2]);
        if (k + 4 <= n) *reinterpret_cast<uint4*>(C) = o;
        else { C[0] = o.x; if (k + 1 < n) C[1] = o.y; if (k + 2 < n) C[2] = o.z; }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Tile plan of the scoring kernel.  Windows differ by an order of magnitude along a genome (a few CpGs per 2 kb in
// open sea, hundreds inside a CpG island), so the start sites are scored in two classes of tiles:
//   narrow (A): TI consecutive start sites (aligned), all with F_k <= WA: starts and ends share one LDS prefix row;
//   wide   (B): 16 consecutive start sites x one tile of TK end sites, as many end tiles as the unit's widest block needs.
// An aligned group of TI starts is narrow when every 16-site unit in it has umax16 <= WA, else all its units are wide.
// k_tile_count counts the tiles of each class per (stage, chunk); k_stage_plan turns the counts into per-stage
// prefixes over the chunks; k_tile_emit writes the tile descriptors in site order.
// ------------------------------------------------------------------------------------------------------------
struct PlanArgs { const int32_t* sb; int32_t TI, WA, TK, n_stages, WM; };     // sb: stage bounds, as in StageView; WM: widest window of a medium tile (0: no medium class)

struct TileDesc { int32_t chunk, ka, nk, et_lo; };     // start sites [ka, ka+nk); wide tiles: end sites [et_lo, et_lo+TK)

// Tiles of the aligned group of TI start sites at ka: ONE narrow tile when every 16-site unit of it has umax16 <= WA; otherwise
// unit by unit a medium tile (umax16 <= WM: the unit's 16 starts with all their ends) or as many wide tiles as its widest block
// needs end tiles of TK sites.
__device__ __forceinline__ void wg_group_tiles(const JobView& J, const ChunkDesc& cd, const PlanArgs& P, int ka, int s1,
                                               uint32_t& nA, uint32_t& nB, uint32_t& nM)
{
    const int kb = (ka + P.TI < s1) ? ka + P.TI : s1;
    uint32_t m = 0, tb = 0, tm = 0;
    for (int k0 = ka; k0 < kb; k0 += 16) {
        const uint32_t um = J.umax16[cd.unit_off + (k0 >> 4)];
        const int nk = (kb - k0 < 16) ? kb - k0 : 16;
        m = um > m ? um : m;
        if (um <= (uint32_t)P.WM) tm += 1u;
        else tb += ((uint32_t)(nk - 1) + um + (uint32_t)P.TK - 1u) / (uint32_t)P.TK;     // ends k0 .. k0+nk-1+um-1
    }
    const bool narrow = m <= (uint32_t)P.WA;
    nA = narrow ? 1u : 0u;
    nB = narrow ? 0u : tb;
    nM = narrow ? 0u : tm;
}

__global__ __launch_bounds__(WG_BLOCK) void k_tile_count(JobView J, PlanArgs P, uint32_t* __restrict__ cntA, uint32_t* __restrict__ cntB,
                                                         uint32_t* __restrict__ cntM)
{
    __shared__ uint32_t wa[WG_BLOCK / 64], wb[WG_BLOCK / 64], wm[WG_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c = blockIdx.x, stg = blockIdx.y;
    const ChunkDesc cd = J.chunks[c];
    const int s0 = P.sb[stg];
    uint32_t a = 0, b = 0, m = 0;
    if (s0 < cd.len) {
        const int s1 = (P.sb[stg + 1] < cd.len) ? P.sb[stg + 1] : cd.len;
        const int nU = (s1 - s0 + P.TI - 1) / P.TI;
        for (int u = tid; u < nU; u += WG_BLOCK) {
            uint32_t na, nb, nm;
            wg_group_tiles(J, cd, P, s0 + u * P.TI, s1, na, nb, nm);
            a += na; b += nb; m += nm;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { a += (uint32_t)__shfl_down((int)a, o); b += (uint32_t)__shfl_down((int)b, o); m += (uint32_t)__shfl_down((int)m, o); }
    if (lane == 0) { wa[wv] = a; wb[wv] = b; wm[wv] = m; }
    __syncthreads();
    if (tid == 0) {
        uint32_t ta = 0, tb = 0, tm = 0;
        for (int q = 0; q < WG_BLOCK / 64; q++) { ta += wa[q]; tb += wb[q]; tm += wm[q]; }
        cntA[(int64_t)stg * J.n_chunks + c] = ta;
        cntB[(int64_t)stg * J.n_chunks + c] = tb;
        cntM[(int64_t)stg * J.n_chunks + c] = tm;
    }
}

// one workgroup per stage; per chunk the number of scored blocks in the stage, and the exclusive prefixes over the
// chunks of the blocks and of the tiles of both classes
#define WG_PLAN_BLOCK 256       // (round 6: 1024 threads — three rounds over the 2,315 items of a whole-genome batch instead of ten — took 0.24 ms instead of 0.03:
                                // a workgroup of sixteen wavefronts waits for a whole CU while k_validate holds the wavefront slots)
__global__ __launch_bounds__(WG_PLAN_BLOCK) void k_stage_plan(JobView J, PlanArgs P, const uint32_t* __restrict__ cntA,
                                                         const uint32_t* __restrict__ cntB, const uint32_t* __restrict__ cntM, int64_t* cbase, uint32_t* cum0,
                                                         int64_t* tbaseA, int64_t* tbaseB, int64_t* tbaseM, int64_t* stage_pairs, int64_t* stage_tiles)
{
    __shared__ uint64_t wa[WG_PLAN_BLOCK / 64], wb[WG_PLAN_BLOCK / 64], wc[WG_PLAN_BLOCK / 64], wd[WG_PLAN_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int stg = blockIdx.x;
    const int nC = J.n_chunks;
    uint64_t runA = 0, runB = 0, runC = 0, runD = 0;
    for (int base = 0; base < nC; base += WG_PLAN_BLOCK) {
        const int c = base + tid;
        uint64_t sz = 0, nt = 0, nu = 0, nm = 0;
        uint32_t c0 = 0;
        if (c < nC) {
            const ChunkDesc cd = J.chunks[c];
            const int64_t s0 = P.sb[stg];
            if (s0 < cd.len) {
                const int64_t s1 = (P.sb[stg + 1] < cd.len) ? P.sb[stg + 1] : cd.len;
                c0 = J.cum32[cd.site_off + s0];
                const uint64_t cend = (s1 < cd.len) ? (uint64_t)J.cum32[cd.site_off + s1] : (uint64_t)J.chunk_pairs[c];
                sz = cend - c0;
                // (cntA == NULL: a job without a window beyond the narrow tiles' — every aligned group of TI starts is one narrow tile)
                nt = cntA ? (uint64_t)cntA[(int64_t)stg * nC + c] : (uint64_t)((s1 - s0 + P.TI - 1) / P.TI);
                nu = cntA ? (uint64_t)cntB[(int64_t)stg * nC + c] : 0;
                nm = cntA ? (uint64_t)cntM[(int64_t)stg * nC + c] : 0;
            }
        }
        const uint64_t ia = wg_wave_incl_scan_u64(sz, lane), ib = wg_wave_incl_scan_u64(nt, lane), ic = wg_wave_incl_scan_u64(nu, lane),
                       id = wg_wave_incl_scan_u64(nm, lane);
        if (lane == 63) { wa[wv] = ia; wb[wv] = ib; wc[wv] = ic; wd[wv] = id; }
        __syncthreads();
        uint64_t oa = 0, ob = 0, oc = 0, od = 0, ta = 0, tb2 = 0, tc = 0, td = 0;
#pragma unroll
        for (int q = 0; q < WG_PLAN_BLOCK / 64; q++) {
            if (q < wv) { oa += wa[q]; ob += wb[q]; oc += wc[q]; od += wd[q]; }
            ta += wa[q]; tb2 += wb[q]; tc += wc[q]; td += wd[q];
        }
        __syncthreads();
        if (c < nC) {
            cbase[(int64_t)stg * nC + c] = (int64_t)(runA + oa + ia - sz);
            cum0[(int64_t)stg * nC + c] = c0;
            tbaseA[(int64_t)stg * (nC + 1) + c] = (int64_t)(runB + ob + ib - nt);
            tbaseB[(int64_t)stg * (nC + 1) + c] = (int64_t)(runC + oc + ic - nu);
            tbaseM[(int64_t)stg * (nC + 1) + c] = (int64_t)(runD + od + id - nm);
        }
        runA += ta; runB += tb2; runC += tc; runD += td;
    }
    if (tid == 0) {
        tbaseA[(int64_t)stg * (nC + 1) + nC] = (int64_t)runB;
        tbaseB[(int64_t)stg * (nC + 1) + nC] = (int64_t)runC;
        tbaseM[(int64_t)stg * (nC + 1) + nC] = (int64_t)runD;
        stage_pairs[stg] = (int64_t)runA;
        stage_tiles[3 * stg] = (int64_t)runB;
        stage_tiles[3 * stg + 1] = (int64_t)runC;
        stage_tiles[3 * stg + 2] = (int64_t)runD;
    }
}

// one workgroup per chunk (of one stage): the tile descriptors of both classes, in site order
__global__ __launch_bounds__(WG_BLOCK) void k_tile_emit(JobView J, PlanArgs P, int stg, const int64_t* __restrict__ tbaseA,
                                                        const int64_t* __restrict__ tbaseB, const int64_t* __restrict__ tbaseM,
                                                        TileDesc* __restrict__ tilesA, TileDesc* __restrict__ tilesB, TileDesc* __restrict__ tilesM)
{
    __shared__ uint32_t wa[WG_BLOCK / 64], wb[WG_BLOCK / 64], wm[WG_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c = blockIdx.x;
    const int nC = J.n_chunks;
    const ChunkDesc cd = J.chunks[c];
    const int s0 = P.sb[stg];
    if (s0 >= cd.len) return;
    const int s1 = (P.sb[stg + 1] < cd.len) ? P.sb[stg + 1] : cd.len;
    const int nU = (s1 - s0 + P.TI - 1) / P.TI;
    int64_t runA = tbaseA[(int64_t)stg * (nC + 1) + c], runB = tbaseB[(int64_t)stg * (nC + 1) + c], runM = tbaseM[(int64_t)stg * (nC + 1) + c];
    for (int base = 0; base < nU; base += WG_BLOCK) {
        const int u = base + tid;
        const int ka = s0 + u * P.TI;
        uint32_t na = 0, nb = 0, nm = 0;
        if (u < nU) wg_group_tiles(J, cd, P, ka, s1, na, nb, nm);
        const uint32_t ia = wg_wave_incl_scan_dpp_u32(na), ib = wg_wave_incl_scan_dpp_u32(nb), im = wg_wave_incl_scan_dpp_u32(nm);
        if (lane == 63) { wa[wv] = ia; wb[wv] = ib; wm[wv] = im; }
        __syncthreads();
        uint32_t oa = 0, ob = 0, om = 0, ta = 0, tb = 0, tm = 0;
#pragma unroll
        for (int q = 0; q < WG_BLOCK / 64; q++) { if (q < wv) { oa += wa[q]; ob += wb[q]; om += wm[q]; } ta += wa[q]; tb += wb[q]; tm += wm[q]; }
        __syncthreads();
        if (u < nU) {
            const int kb = (ka + P.TI < s1) ? ka + P.TI : s1;
            if (na) {
                TileDesc d = {c, ka, kb - ka, 0};
                tilesA[runA + oa + ia - na] = d;
            } else {
                int64_t o = runB + ob + ib - nb, om2 = runM + om + im - nm;
                for (int k0 = ka; k0 < kb; k0 += 16) {
                    const uint32_t um = J.umax16[cd.unit_off + (k0 >> 4)];
                    const int nk = (kb - k0 < 16) ? kb - k0 : 16;
                    if (um <= (uint32_t)P.WM) { TileDesc d = {c, k0, nk, 0}; tilesM[om2++] = d; continue; }
                    const int kt = (int)(((uint32_t)(nk - 1) + um + (uint32_t)P.TK - 1u) / (uint32_t)P.TK);
                    for (int e = 0; e < kt; e++) { TileDesc d = {c, k0, nk, k0 + e * P.TK}; tilesB[o++] = d; }
                }
            }
        }
        runA += ta; runB += tb; runM += tm;
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_cost: one workgroup scores the blocks of one tile: up to TI consecutive START sites (times one tile of end sites
// in the wide class).  LDS holds, per sample of the current group, the exclusive prefixes P[x] of (#meth, #cov) over the sites the
// tile touches: a block (k, i) is then P[i+1] - P[k].
// ------------------------------------------------------------------------------------------------------------
struct CostArgs {
    float pc, pc2;
    int32_t NS;        // samples per LDS group
    int32_t rows;      // guard-free kernels: exponents held by the two lookup tables (wg_lookup_rows for the longest block of the tile class)
    const wg_d2* tab;  // those tables, built by the host: rows * 16 log2f entries, then rows * 64 fast-log2 entries
    int32_t xcd_group; // consecutive tiles that go to one XCD before the next XCD's group begins (see k_cost)
    int32_t cmap;      // log2 of the blocks per entry of the block -> start map of the narrow / medium tiles: 0 = a byte per block (small cohorts: LDS to spare), 3 = a byte per eight blocks + forward steps
    uint32_t* finished; // staged jobs (or NULL): counts the tiles of this launch that are DONE — what k_stage_gate watches (a scheduling hint, no data hangs on it)
};

// Gate between the scoring launches of two consecutive stages.  A kernel packet always waits for the kernel ahead of it in its hardware queue
// to END (hipExtAnyOrderLaunch is not honoured on gfx9: tools/micro/any_order.hip), so a stage's last tiles leave the CUs partly idle for
// most of a tile's duration (~40 of ~60 us) before the next stage's first tile.  The stages of a staged job therefore alternate between two
// streams, and stage s + 1 is held back by this one-wavefront kernel until all but `slack` tiles of stage s are done — about as many as the chip
// holds at a time, so the rest is in flight and every workgroup slot that falls free from now on would stay empty: it goes to stage s + 1, while
// stage s still ends, and releases its recurrence, as early as before (two ungated streams were measured in round 3: the stages interleave, every
// recurrence starts late).  Nothing but timing depends on the gate (the stages write disjoint buffers), so the wait is bounded: `max_ticks` of the
// 100 MHz clock.
__global__ __launch_bounds__(64) void k_stage_gate(const uint32_t* __restrict__ finished, uint32_t need, long long max_ticks)
{
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need && wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(8);
}

// Stage the exclusive prefixes P[A+x], x = x0 .. x0+cnt-1, of sample row `row` into dst[0..cnt) (one wavefront).
// A is chunk-relative; start0+A is either the chunk start or a multiple of WG_CARRY_G (wg_group_start), so a carry of
// k_scan seeds the scan; the up to WG_CARRY_G sites between A and A+x0 are summed but not stored.
__device__ __forceinline__ void wg_stage_prefix_row(uint2* __restrict__ dst, const uint8_t* __restrict__ row,
                                                    const uint2* __restrict__ carry, const ChunkDesc& cd,
                                                    int64_t n_total, int A, int x0, int cnt0, int lane)
{
    const int cnt = x0 + cnt0;
    const int64_t abs0 = cd.start0 + A;
    const uint2 c0 = carry[(abs0 >> WG_CARRY_SHIFT) - (cd.start0 >> WG_CARRY_SHIFT)];
    uint32_t run_m = c0.x, run_t = c0.y;
    const int64_t al = abs0 & ~3LL;                    // 8-byte aligned
    const int hs = (int)(abs0 - al);
    for (int p0 = 0; p0 < cnt + hs; p0 += 256) {
        const int sidx = p0 + lane * 4;                // site offset from `al`
        const int64_t a = al + sidx;
        uint32_t w0 = 0, w1 = 0;
        if (sidx < cnt + hs) {
            if (a + 4 <= n_total) {
                const uint2 v = *reinterpret_cast<const uint2*>(row + 2 * a);
                w0 = v.x; w1 = v.y;
            } else {
                for (int j = 0; j < 4; j++) if (a + j < n_total) {
                    const uint32_t h = (uint32_t)row[2 * (a + j)] | ((uint32_t)row[2 * (a + j) + 1] << 8);
                    if (j < 2) w0 |= h << (16 * j); else w1 |= h << (16 * (j - 2));
                }
            }
        }
        uint32_t m[4], t[4];
        uint32_t tot = 0;                              // packed lane total: meth | cov << 16 (<= 1020 each)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t h = ((j < 2) ? w0 : w1) >> (16 * (j & 1));
            const int x = sidx + j - hs;               // entry index == site index relative to A
            const bool in = x >= 0 && (A + x) < cd.len;
            m[j] = in ? (h & 0xffu) : 0u;
            t[j] = in ? ((h >> 8) & 0xffu) : 0u;
            tot += m[j] | (t[j] << 16);
        }
        const uint32_t incl = wg_wave_incl_scan_dpp_u32(tot);   // <= 64*1020 = 65280 per half: no carry between halves
        const uint32_t excl = incl - tot;
        uint32_t em = run_m + (excl & 0xffffu), et = run_t + (excl >> 16);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int x = sidx + j - hs;
            if (x >= x0 && x < cnt) dst[x - x0] = make_uint2(em, et);
            em += m[j]; et += t[j];
        }
        const uint32_t wt = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        run_m += wt & 0xffffu; run_t += wt >> 16;
    }
}

// Largest carry position <= chunk-relative site k: the 64-aligned absolute index below it, or the chunk start.
__device__ __forceinline__ int wg_group_start(const ChunkDesc& cd, int k)
{
    const int64_t a = (cd.start0 + k) & ~(int64_t)(WG_CARRY_G - 1);
    return a <= cd.start0 ? 0 : (int)(a - cd.start0);
}

// Narrow tiles: TILE-LOCAL exclusive prefixes L[x] = sum of the sites ka .. ka+x-1, x = 0 .. cnt-1, of `ns` sample rows,
// (meth | cov << 16) packed in one dword each — a narrow tile spans <= WG_NARROW_ROW - 1 <= 124 sites, so both sums
// stay below 2^15, and a block's counts are a difference of two entries of the same row: no carry of k_scan needed.
// Two sample rows per wavefront pass (one per 32-lane half, 4 sites per lane, 8-byte loads), wave `wv` of 4.
template <int ROW>
__device__ __forceinline__ void wg_stage_local_rows(uint32_t* __restrict__ Et, const uint8_t* __restrict__ betas, int64_t pitch,
                                                    int s_first, int ns, const ChunkDesc& cd, int64_t n_total, int ka, int cnt,
                                                    int lane, int wv)
{
    // rows of more than 128 entries (TI = 128: up to 189 + 3 of alignment): one sample row per wavefront pass, all 64 lanes
    constexpr bool WHOLE = ROW > 128;
    const int half = WHOLE ? 0 : lane >> 5, l5 = WHOLE ? lane : lane & 31;
    const int64_t abs0 = cd.start0 + ka;
    const int64_t al = abs0 & ~3LL;                    // 8-byte aligned
    const int hs = (int)(abs0 - al);
    const int sidx = l5 * 4;                           // site offset of this lane from `al`
    const int64_t a = al + sidx;
    const int nsite = cnt - 1;                         // sites ka .. ka+cnt-2 are summed
    for (int r0 = WHOLE ? wv : wv * 2; r0 < ns; r0 += (WHOLE ? 1 : 2) * (WG_BLOCK / 64)) {
        const int rr = r0 + half;
        const bool act = rr < ns;
        uint32_t w0 = 0, w1 = 0;
        if (act && sidx < nsite + hs) {
            const uint8_t* row = betas + (int64_t)(s_first + rr) * pitch;
            if (a + 4 <= n_total) {
                const uint2 v = *reinterpret_cast<const uint2*>(row + 2 * a);
                w0 = v.x; w1 = v.y;
            } else {
                for (int j = 0; j < 4; j++) if (a + j < n_total) {
                    const uint32_t h = (uint32_t)row[2 * (a + j)] | ((uint32_t)row[2 * (a + j) + 1] << 8);
                    if (j < 2) w0 |= h << (16 * j); else w1 |= h << (16 * (j - 2));
                }
            }
        }
        uint32_t mt[4];
        uint32_t tot = 0;                              // packed lane total: meth | cov << 16 (<= 1020 each)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t h = ((j < 2) ? w0 : w1) >> (16 * (j & 1));
            const int x = sidx + j - hs;               // site index relative to ka
            const bool in = x >= 0 && x < nsite;
            mt[j] = in ? ((h & 0xffu) | ((h & 0xff00u) << 8)) : 0u;
            tot += mt[j];
        }
        // <= 32*1020 (whole wave: 64*1020 = 65280) per field: no carry between the fields
        const uint32_t incl = WHOLE ? wg_wave_incl_scan_dpp_u32(tot) : wg_half_incl_scan_dpp_u32(tot);
        uint32_t e = incl - tot;
        uint32_t* dst = Et + (size_t)rr * ROW;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int x = sidx + j - hs;
            if (act && x >= 0 && x < cnt) dst[x] = e;
            e += mt[j];
        }
    }
}

// The same for the rows of a MEDIUM tile (16 starts, windows <= WG_MEDIUM_WMAX: up to 269 entries): one sample row per wavefront,
// 4 sites (one 8-byte load) per lane and pass, the running total carried from pass to pass (two passes per row).  The packed sums may now carry from the #meth field into the #cov field
// (35 lanes x 8 x 255 > 2^16) and wrap at 2^32: harmless, because an entry is the plain integer  sum(meth) + 2^16 sum(cov)  mod 2^32,
// so the difference of two entries is  d(meth) + 2^16 d(cov)  mod 2^32, and both differences of a block of <= 252 sites are below
// 2^16: the two halves of the difference ARE the block's counts.
template <int ROW>
__device__ __forceinline__ void wg_stage_local_rows_long(uint32_t* __restrict__ Et, const uint8_t* __restrict__ betas, int64_t pitch,
                                                     int s_first, int ns, const ChunkDesc& cd, int64_t n_total, int ka, int cnt,
                                                     int lane, int wv)
{
    // one sample row per wavefront, 256 sites (64 lanes x 4 sites, one 8-byte load each) per pass with the running total carried
    // over: a row of 269 entries takes two passes (the second one a few lanes wide).  (A single pass of 8 sites per lane cost six
    // more registers than the kernel's main loop needs: 101 VGPRs, one wavefront per SIMD fewer.)
    const int64_t abs0 = cd.start0 + ka;
    const int64_t al = abs0 & ~3LL;                    // 8-byte aligned
    const int hs = (int)(abs0 - al);
    const int nsite = cnt - 1;                         // sites ka .. ka+cnt-2 are summed
    for (int rr = wv; rr < ns; rr += WG_BLOCK / 64) {
        const uint8_t* row = betas + (int64_t)(s_first + rr) * pitch;
        uint32_t* dst = Et + (size_t)rr * ROW;
        uint32_t run = 0;
        for (int p0 = 0; p0 < nsite + hs; p0 += 256) {
            const int sidx = p0 + lane * 4;            // site offset of this lane from `al`
            const int64_t a = al + sidx;
            uint32_t w0 = 0, w1 = 0;
            if (sidx < nsite + hs) {
                if (a + 4 <= n_total) {
                    const uint2 v = *reinterpret_cast<const uint2*>(row + 2 * a);
                    w0 = v.x; w1 = v.y;
                } else {
                    for (int j = 0; j < 4; j++) if (a + j < n_total) {
                        const uint32_t h = (uint32_t)row[2 * (a + j)] | ((uint32_t)row[2 * (a + j) + 1] << 8);
                        if (j < 2) w0 |= h << (16 * j); else w1 |= h << (16 * (j - 2));
                    }
                }
            }
            uint32_t mt[4];
            uint32_t tot = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t h = ((j < 2) ? w0 : w1) >> (16 * (j & 1));
                const int x = sidx + j - hs;           // site index relative to ka
                const bool in = x >= 0 && x < nsite;
                mt[j] = in ? ((h & 0xffu) | ((h & 0xff00u) << 8)) : 0u;
                tot += mt[j];
            }
            const uint32_t incl = wg_wave_incl_scan_dpp_u32(tot);
            uint32_t e = run + (incl - tot);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int x = sidx + j - hs;
                if (x >= 0 && x < cnt) dst[x] = e;
                e += mt[j];
            }
            run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        // the entry behind the last site when it falls exactly on a pass boundary (x == nsite, sidx + j - hs beyond the loop's reach)
        if (lane == 0 && ((nsite + hs) & 255) == 0 && nsite < cnt) dst[nsite] = run;
    }
}

#define WG_WIDE_TK      128         // end sites per wide tile
#define WG_WIDE_TS      16          // start sites per wide tile

// Scored blocks of one tile.  SPLIT 0: narrow tile, TI start sites whose windows are all <= WG_NARROW_WMAX, prefixes
// tile-local and packed (wg_stage_local_rows).  SPLIT 1: wide tile, WG_WIDE_TS start sites x WG_WIDE_TK end sites,
// prefixes of starts and ends staged separately from the carries of k_scan.  SPLIT 2 (round 3): medium tile, the 16 start sites
// of one unit whose windows are all <= WG_MEDIUM_WMAX with ALL their ends — CpG islands: windows of 60 .. 250 sites — scored like
// a narrow tile (tile-local packed prefixes from the raw bytes, one subtraction per block and sample, the short division core)
// where the wide tiles spent two carry-seeded scans per sample and end tile and ran at half the narrow tiles' rate.  The LDS row
// strides are compile-time constants: the sample loop is unrolled by four with the row offsets in the instructions' offset fields.
template <int TI, int FAST, int SPLIT, bool ONEG>      // FAST = wg_term_mode(pseudo count); ONEG: every sample of the job in LDS at once (one sample group)
__global__ __launch_bounds__(WG_BLOCK) void k_cost(JobView J, StageView SV, CostArgs A, const TileDesc* __restrict__ tiles,
                                                   int64_t n_tiles, double* __restrict__ cost, int64_t n_tiles_padded)
{
    constexpr bool WIDE = SPLIT == 1;
    constexpr int WM = SPLIT == 2 ? WG_MEDIUM_WMAX : WG_NARROW_WMAX;       // widest window of a tile with tile-local prefixes
    static_assert(SPLIT != 2 || TI == WG_MEDIUM_TS, "a medium tile is one 16-site unit");
    constexpr int KS = WIDE ? WG_WIDE_TK + 1 : TI + WM + 1;                // entries per sample row of the E array
    constexpr int IS = WIDE ? WG_WIDE_TS + 1 : 0;                          // entries per sample row of the S array
    // pseudo count >= 1: both logs on their k-scaled lookup tables, A.rows exponents each (sized by the host to the
    // longest block of the tile class)
    constexpr bool KY = (FAST >= 2);             // 2: k-scaled tables; 3: the same with the short division core (narrow tiles, verified per call)
    constexpr bool DIVS = (FAST == 3);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(sizeof(wg_fast_tables) % 16 == 0 && sizeof(wg_d2) == 16, "the arrays behind the tables are 16-byte aligned");
    // KY: only the two lookup tables, A.rows exponents each; otherwise the general fast tables
    const size_t TB = KY ? (size_t)A.rows * (16 + 64) * sizeof(wg_d2) : sizeof(wg_fast_tables);
    wg_fast_tables* tb = reinterpret_cast<wg_fast_tables*>(smem);
    wg_d2* iyt = reinterpret_cast<wg_d2*>(smem);
    wg_d2* kyt = iyt + (size_t)A.rows * 16;
    const wg_d2* iy0 = iyt + (A.rows - 1) * 16;                                  // (KY) the rows of k = 0
    const wg_d2* ky0 = kyt + (A.rows - 1) * 64;
    uint2* Et = reinterpret_cast<uint2*>(smem + TB);                             // wide: [NS][KS] P[i+1] of the ends
    uint2* St = Et + (size_t)A.NS * KS;                                          // wide: [NS][IS] P[k] of the starts
    uint32_t* Lt = reinterpret_cast<uint32_t*>(smem + TB);                       // narrow: [NS][KS] packed local prefixes
    char* after = WIDE ? reinterpret_cast<char*>(St + (((size_t)A.NS * (KS + IS) + 1) & ~(size_t)1) - (size_t)A.NS * KS)
                       : reinterpret_cast<char*>(Lt + (((size_t)A.NS * KS + 3) & ~(size_t)3));              // 16-byte aligned
    // One 16-byte record per start site kl of the tile (round 4; rounds 1-3 kept offs[], ist[] and radj[] apart: three LDS round trips and
    // ~45 VALU instructions per block outside the sample loop, which a small cohort amortises over few evaluations):
    //   offs  first flattened block of the start (rec[nk].offs = Q closes the table)
    //   ea    block q of the start reads the prefix entry  q + ea  (= i + 1 - first staged site: i = ist + (q - offs))
    //   sb    ... and is stored at cost element  q + sb        (= row offset of the start - k + i)
    struct Rec { int32_t offs, ea; int64_t sb; };
    Rec* rec = reinterpret_cast<Rec*>(after);                                    // [TI + 1]
    int32_t* misc = reinterpret_cast<int32_t*>(rec + (TI + 1));                  // [8]
    // The start site of a block (narrow / medium tiles): a byte per block where LDS is to spare (A.cmap = 0: the 128-start tiles of small
    // cohorts), else a byte per EIGHT blocks — the start of block 8 g — and a step forward from there when needed (a start has ~20
    // blocks).  Why not a byte per block everywhere: LDS is handed out in granules of 1280 bytes and a 64-start tile at 32 samples sits
    // at 31.6 KB = 25 granules = five workgroups per CU; one granule more and it is four (-5 %).
    uint8_t* cmap = reinterpret_cast<uint8_t*>(misc + 8);                        // [(TI * WM >> A.cmap) + 1]
    constexpr bool CMAP = !WIDE;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nC = J.n_chunks;
    // XCD-aware remap.  Workgroup i runs on XCD i mod 8, each with its own L2: consecutive tiles (which share halo sites) go to
    // the same XCD in groups of G; the groups are dealt round-robin, so that every XCD gets an even sample of the genome (tile
    // costs follow the CpG density: eight contiguous eighths — round 1's mapping — left the densest eighth's XCD working alone
    // at the end of a small job)
    const int64_t G = A.xcd_group, jx = (int64_t)(blockIdx.x >> 3);
    const int64_t t = (jx / G) * (8 * G) + (int64_t)(blockIdx.x & 7) * G + (jx % G);
    if (t >= n_tiles) return;
    const TileDesc td = tiles[t];
    const int c = td.chunk;
    const ChunkDesc cd = J.chunks[c];
    const int ka = td.ka, nk = td.nk, kb = ka + nk;      // start sites [ka, kb)
    // end-site tile [et_lo, et_hi) of a wide tile; narrow tiles: no restriction
    const int et_lo = WIDE ? td.et_lo : 0;
    const int et_hi = WIDE ? et_lo + WG_WIDE_TK : (1 << 30);
    const uint32_t cum0 = SV.cum0[(int64_t)SV.stage * nC + c];

    if (KY) wg_lookup_tables_to_lds(iyt, kyt, A.rows, A.tab, tid, WG_BLOCK);
    else wg_fast_tables_to_lds(tb, tid, WG_BLOCK);
    constexpr int PW = (TI + 63) / 64;                   // wavefronts that share the tile's start sites (two for TI = 128)
    if (wv < PW) {
        const int kl = wv * 64 + lane;
        const int k = ka + kl;
        const bool valid = kl < nk;
        int cnt = 0, is = 0, ie = -1;
        if (valid) {
            const int f = J.W16[cd.site_off + k];
            is = k > et_lo ? k : et_lo;                  // ends i in [k, k+f) cut to the tile
            ie = (k + f < et_hi ? k + f : et_hi) - 1;
            cnt = ie - is + 1;
            if (cnt < 0) cnt = 0;
        }
        const uint32_t incl = wg_wave_incl_scan_dpp_u32((uint32_t)cnt);
        // the record of the start, relative to the FIRST staged entry being that of site ka (narrow / medium) — the wide tiles' first entry
        // (site imin + 1) is known only after the barrier and is subtracted there; second wavefront: still without the first one's total
        if (kl < TI) {
            const int32_t o = (int32_t)(incl - (uint32_t)cnt);
            Rec r;
            r.offs = o;
            r.ea = is - o + 1 - (WIDE ? 0 : ka);
            r.sb = (valid ? (int64_t)(J.cum32[cd.site_off + k] - cum0) - k : 0) + (is - o);      // row offset of k, minus k: + i addresses (k, i)
            rec[kl] = r;
        }
        const uint32_t imin = wg_wave_min_u32(cnt > 0 ? (uint32_t)is : 0x7fffffffu);
        const uint32_t imax = wg_wave_max_u32(cnt > 0 ? (uint32_t)ie : 0u);
        if (lane == 0) { misc[2 * wv] = (int32_t)imin; misc[2 * wv + 1] = (int32_t)imax; }
        if (lane == 63) misc[4 + wv] = (int32_t)incl;
    }
    __syncthreads();
    const int imin = PW > 1 ? (misc[0] < misc[2] ? misc[0] : misc[2]) : misc[0];
    const int imax = PW > 1 ? (misc[1] > misc[3] ? misc[1] : misc[3]) : misc[1];
    const int Q = PW > 1 ? misc[4] + misc[5] : misc[4];
    if (Q == 0) { if (A.finished != nullptr && tid == 0) __hip_atomic_fetch_add(A.finished, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    // wide: E array = P[x] for x = eA .. imax+1 (ends use P[i+1]), S array = P[k] for k = ka .. kb-1.
    // narrow: one array L[x - ka], x = ka .. imax+1, serves both.
    const int eA = WIDE ? imin + 1 : ka;
    if (PW > 1 || WIDE) {
        if (wv < PW && wv * 64 + lane < TI) {
            const int t0 = wv == 1 ? misc[4] : 0;
            Rec& r = rec[wv * 64 + lane];
            if (PW > 1 && wv == 1) { r.offs += t0; r.sb -= t0; }
            if (WIDE || (PW > 1 && wv == 1)) r.ea -= t0 + (WIDE ? eA : 0);
        }
    }
    if (PW > 1 || WIDE) __syncthreads();
    if (tid == 0) rec[nk].offs = Q;                                              // closes the table: no start beyond nk - 1 is ever stepped to (read after the next barrier)
    if (CMAP && tid < nk) {                                                      // (read after the barrier that opens the sample group)
        const int o0 = rec[tid].offs, o1 = tid + 1 < nk ? rec[tid + 1].offs : Q;
        const int sh = A.cmap;
        for (int g = (o0 + (1 << sh) - 1) >> sh; (g << sh) < o1; g++) cmap[g] = (uint8_t)tid;
    }
    // carry position the scan of a wide row starts from.  eA == cd.len when the tile's first end is the chunk's last site (P[len]
    // alone is wanted): when start0 + len is a multiple of WG_CARRY_G that position is a group of its own, one past the cd.nG
    // carries k_scan wrote for the chunk — the scan starts from the last group INSIDE the chunk instead (up to WG_CARRY_G sites summed).
    const int eG = wg_group_start(cd, eA < cd.len ? eA : cd.len - 1);
    const int Ecnt = imax + 2 - eA;
    const int sG = wg_group_start(cd, ka);
    const int Scnt = kb - ka;

    // Blocks of the tile, flattened: q -> (kl, i).  Every thread walks its blocks q = tid, tid+256, ...; for each it
    // runs the samples of the LDS-resident group IN FILE ORDER, carrying the double sum in a register (and, when the
    // samples do not fit LDS at once, across groups in a register array indexed by the block's round) — the accumulation
    // order of segmentor.cpp:120-136.
    const float pc = A.pc, pc2 = A.pc2;
    double* cb = cost + SV.cbase[(int64_t)SV.stage * nC + c];
    // One sample group (ONEG: the host launches this form when NS >= the job's samples; TI = 128 tiles — up to 7680 blocks — are only
    // planned then): no partial sums across groups, and the 32 registers of their array are free (x 32: 95 -> 63 VGPRs)
    static_assert(TI <= 64 || ONEG, "128-start tiles hold every sample at once");
    constexpr bool ONEGROUP = ONEG;
    double accR[ONEGROUP ? 1 : WG_PAIR_CAP / WG_BLOCK];  // partial sums of this thread's blocks across sample groups
    for (int g0 = 0; g0 < J.n_samples; g0 += A.NS) {
        const int ns = (J.n_samples - g0 < A.NS) ? J.n_samples - g0 : A.NS;
        const bool firstg = ONEGROUP || g0 == 0, lastg = ONEGROUP || g0 + ns >= J.n_samples;
        __syncthreads();
        if (WIDE) {
            for (int rr = wv; rr < ns; rr += WG_BLOCK / 64) {
                const int s = g0 + rr;
                const uint8_t* row = J.betas + (int64_t)s * J.pitch;
                const uint2* carry = J.carry + cd.carry_off + (int64_t)s * cd.nG;
                wg_stage_prefix_row(Et + (size_t)rr * KS, row, carry, cd, J.n_total, eG, eA - eG, Ecnt, lane);
                wg_stage_prefix_row(St + (size_t)rr * IS, row, carry, cd, J.n_total, sG, ka - sG, Scnt, lane);
            }
        } else if (SPLIT == 2) {
            wg_stage_local_rows_long<KS>(Lt, J.betas, J.pitch, g0, ns, cd, J.n_total, ka, Ecnt, lane, wv);
        } else {
            wg_stage_local_rows<KS>(Lt, J.betas, J.pitch, g0, ns, cd, J.n_total, ka, Ecnt, lane, wv);
        }
        __syncthreads();
        int qi = 0;
        for (int q = tid; q < Q; q += WG_BLOCK, qi++) {
            int lo = 0, hi = nk;                           // largest kl with rec[kl].offs <= q
            if (CMAP) { lo = (int)cmap[q >> A.cmap]; if (A.cmap) while (rec[lo + 1].offs <= q) lo++; }
            else while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (rec[mid].offs <= q) lo = mid; else hi = mid; }
            const Rec r = rec[lo];
            double acc = firstg ? 0.0 : accR[qi];
            // sample loop, unrolled by four by hand (the optimiser leaves a loop with the rare exact path inside alone):
            // one address update per four evaluations, the row offsets sit in the instructions' offset fields
            auto term = [&](float nm, float nt) -> double {
                const float ll = FAST == 1 ? wg_sample_term(nm, nt, pc, pc2, tb, &g_wg_tables) : wg_sample_term_plain(nm, nt, pc, pc2, &g_wg_tables);
                return (double)ll;                                               // segmentor.cpp:135 adds the float term to the double sum
            };
            if (WIDE) {
                const uint2* Ep = Et + (q + r.ea);         // P[i+1] of sample sl at Ep[sl * KS]
                const uint2* Sp = St + lo;                 // P[k]   of sample sl at Sp[sl * IS]
                auto one = [&](int sl) {
                    const uint2 pi = Ep[sl * KS];
                    const uint2 pk = Sp[sl * IS];
                    if (KY) acc += (double)wg_sample_term_pcpos_ks<DIVS>((float)(pi.x - pk.x), (float)(pi.y - pk.y), pc, pc2, iy0, ky0, &g_wg_tables);
                    else acc += term((float)(pi.x - pk.x), (float)(pi.y - pk.y));
                };
                int sl = 0;
                for (; sl + 4 <= ns; sl += 4) { one(sl); one(sl + 1); one(sl + 2); one(sl + 3); }
                for (; sl < ns; sl++) one(sl);
            } else {
                const uint32_t* Ep = Lt + (q + r.ea);      // L[i+1-ka] of sample sl at Ep[sl * KS]
                const uint32_t* Sp = Lt + lo;              // L[k-ka]
                auto one = [&](int sl) {
                    const uint32_t d = Ep[sl * KS] - Sp[sl * KS];                // both fields at once: no borrow, L is monotone per field
                    if (KY) acc += (double)wg_sample_term_pcpos_ks<DIVS>((float)(d & 0xffffu), (float)(d >> 16), pc, pc2, iy0, ky0, &g_wg_tables);
                    else acc += term((float)(d & 0xffffu), (float)(d >> 16));
                };
                int sl = 0;
                for (; sl + 4 <= ns; sl += 4) { one(sl); one(sl + 1); one(sl + 2); one(sl + 3); }
                for (; sl < ns; sl++) one(sl);
            }
            // segmentor.cpp:106,137 `if (ll_sum) row[j] = ll_sum` over the 0.0 fill: a sum that began at +0.0 is never -0.0 (IEEE: +0 + -0 = +0,
            // an exact cancellation of non-zero terms gives +0), so the sum itself is what the reference stores
            if (lastg) cb[r.sb + q] = acc;
            else accR[qi] = acc;
        }
    }
    // (counted here, behind the loop: the same line at the tile's entry costs every form of the kernel two VGPRs — 95 -> 97 takes the forms with sample groups from five to four workgroups per CU)
    if (A.finished != nullptr && tid == 0) __hip_atomic_fetch_add(A.finished, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------------------
// k_dp<NW, BL>: one workgroup of 1 + NW wavefronts per chunk.  Wave 0 owns the chunk's recurrence; the others are its
// workers: while wave 0 sweeps the BL steps of batch b out of LDS, they bring the scored-block rows of batch b+1 from
// HBM into the other LDS slot.  One s_barrier per batch.
//
// Push form of segmentor.cpp:142-154.  Lane l of wave 0 holds (best, arg) of the pending step i == l (mod 64),
// i.e. the running  max_k M[k] + cost(k, i)  over the candidates k seen so far.  Iteration k: M[k] is final;
//   every lane with j = (l - k) mod 64 < F_k folds  M[k] + cost(k, k+j)  into its pending step (strict '>' in
//   ascending k keeps the FIRST maximum, as the reference's scan does);  step i = k has now seen its last candidate,
//   so M[k+1] = best of lane k mod 64, and that lane moves on to step k+64.
//
// BL = 64: every window of the job is <= 64 sites, the above is everything.
// BL = 32 (some F_k > 64 somewhere in the job): blocks of 65..128 sites are folded the same way into a SECOND pending
//   register per lane (bestB: step i+64), which becomes the first when the lane moves on.  Blocks longer than 128 sites
//   are pushed by the WORKER waves, one batch behind the recurrence, target-major (one thread per end site, sources
//   in ascending k: no races, first maximum kept), into a ring of pending maxima in global memory (L2).  The
//   recurrence merges a step's ring entry when the batch that finishes the step begins: every ring candidate has a
//   smaller k than any candidate still to come, so '>=' on the merge followed by '>' keeps the reference's rule.
//   Timeline, batch b = steps [base, base+32): sources of batch b are pushed during batch b+1 (targets >= base+128),
//   the ring entries of batch b+2 are fetched (and reset) during batch b+1 — always disjoint from the pushes.
// ------------------------------------------------------------------------------------------------------------
#ifdef WGBSSEG_DP_TIMING
#define WG_DP_T(...) __VA_ARGS__
#else
#define WG_DP_T(...)
#endif
struct DpArgs { int32_t ringN; int32_t pad[3]; };      // ringN: pending-step ring (pow2 >= max window + 128), 0 if BL == 64

#define WG_DP_STATE_HDR 257   // doubles of per-chunk state ahead of the ring: M[k], bestA[64], argA[64], bestB[64], argB[64]

__device__ __forceinline__ double wg_ld_l2_f64(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int32_t wg_ld_l2_i32(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Worker waves: the scored-block rows of the BL steps starting at `base` into an LDS slot, ARRANGED for the push form:
//   slotA[s*64 + l] = cost(k, k+j),      k = base+s, j = (l - k) mod 64, for j < F_k,        else -inf
//   slotB[s*64 + l] = cost(k, k+64+j)                                     for 64+j < F_k,     else -inf   (wide batch only)
// so the recurrence needs no predicate at all: M[k] + (-inf) can never beat a pending maximum.  Worker `lw` arranges
// the steps s == lw (mod NW).  Two phases, one batch apart: the row loads of batch b+2 are issued (into registers) at
// the end of batch b and stored to the free LDS slot at the end of batch b+1 — a full batch to land.
template <int NW, int BL>
struct DpRows {                                          // one worker's share of a batch, in registers
    static constexpr int PER = (BL + NW - 1) / NW;
    double va[PER], vb[PER];
    uint32_t fmax;                                       // widest window of the batch
    bool wideb;
};

struct DpMeta { uint32_t w, rel; };                      // lane l: window and row offset of step base + l of a batch

// Windows and row offsets reach the workers through an LDS ring of 1024 steps, refilled 512 steps at a time (all
// workers, a handful of loads each, every 512 steps): the per-batch work of a worker then has no global load whose
// latency it must sit out — what it loads (rows, ring entries) it consumes a full batch later.
#define WG_DP_META_RING   1024
#define WG_DP_META_REGION 512

template <int NW>
struct DpRefill {
    static constexpr int R = (WG_DP_META_REGION + 64 * NW - 1) / (64 * NW);
    uint32_t w[R], c[R];
};

template <int NW>
__device__ __forceinline__ void wg_dp_refill_issue(DpRefill<NW>& F, const uint16_t* __restrict__ Wp, const uint32_t* __restrict__ Cp,
                                                   uint32_t cum0, int first, int s1, int lane, int lw)
{
#pragma unroll
    for (int q = 0; q < DpRefill<NW>::R; q++) {
        const int e = (q * NW + lw) * 64 + lane;
        const int k = first + e;
        const bool in = e < WG_DP_META_REGION && k < s1;
        F.w[q] = in ? (uint32_t)Wp[k] : 0u;              // steps past the stage's end: no candidates
        F.c[q] = in ? Cp[k] - cum0 : 0u;
    }
}

template <int NW>
__device__ __forceinline__ void wg_dp_refill_commit(const DpRefill<NW>& F, uint16_t* __restrict__ metaW, uint32_t* __restrict__ metaC,
                                                    int first_rel, int lane, int lw)
{
#pragma unroll
    for (int q = 0; q < DpRefill<NW>::R; q++) {
        const int e = (q * NW + lw) * 64 + lane;
        if (e < WG_DP_META_REGION) {
            const int x = (first_rel + e) & (WG_DP_META_RING - 1);
            metaW[x] = (uint16_t)F.w[q]; metaC[x] = F.c[q];
        }
    }
}

template <int BL>
__device__ __forceinline__ DpMeta wg_dp_meta_lds(const uint16_t* __restrict__ metaW, const uint32_t* __restrict__ metaC, int base_rel, int lane)
{
    const int x = (base_rel + lane) & (WG_DP_META_RING - 1);
    DpMeta m;
    m.w = lane < BL ? (uint32_t)metaW[x] : 0u;
    m.rel = metaC[x];
    return m;
}

template <int NW, int BL>
__device__ __forceinline__ void wg_dp_rows_issue(DpRows<NW, BL>& R, const double* __restrict__ cb, const DpMeta M, int base, int lane, int lw)
{
    constexpr bool WIDEJOB = BL < 64;
    const double NEG_INF = -__builtin_inf();
    const uint32_t w = M.w, rel = M.rel;
    R.fmax = wg_wave_max_u32(w);
    R.wideb = WIDEJOB && R.fmax > 64u;
    const int stp0 = base & 63;
#pragma unroll
    for (int q = 0; q < DpRows<NW, BL>::PER; q++) {
        const int sidx = lw + q * NW;
        R.va[q] = NEG_INF; R.vb[q] = NEG_INF;
        if (sidx < BL) {
            const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)w, sidx);
            const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)rel, sidx);
            const uint32_t j = (uint32_t)(lane - stp0 - sidx) & 63u;
            if (j < f) R.va[q] = cb[(int64_t)r + j];
            if (WIDEJOB && R.wideb && j + 64u < f) R.vb[q] = cb[(int64_t)r + j + 64u];
        }
    }
}

template <int NW, int BL>
__device__ __forceinline__ void wg_dp_rows_commit(const DpRows<NW, BL>& R, double* __restrict__ slotA, double* __restrict__ slotB,
                                                  uint32_t* __restrict__ kind, int lane, int lw)
{
    constexpr bool WIDEJOB = BL < 64;
    if (lw == 0 && lane == 0) *kind = R.wideb ? 1u : (R.fmax > (uint32_t)WG_NARROW_WMAX ? 2u : 0u);      // 0: every window of the batch <= 60 sites, 2: <= 64, 1: wide (slot B in use)
#pragma unroll
    for (int q = 0; q < DpRows<NW, BL>::PER; q++) {
        const int sidx = lw + q * NW;
        if (sidx < BL) {
            slotA[sidx * 64 + lane] = R.va[q];
            if (WIDEJOB && R.wideb) slotB[sidx * 64 + lane] = R.vb[q];
        }
    }
}

// Worker waves: blocks longer than 128 sites that START in the batch at `base` (whose M[k] the recurrence has left in
// Mring), folded target-major into the ring.  A ring entry is always updated by the same wave (64-target tile t>>6
// belongs to worker (t>>6) mod NW), so its read-modify-write sequence is one wave's program order.
template <int NW, int BL>
__device__ __forceinline__ void wg_dp_far(const double* __restrict__ cb, const uint16_t* __restrict__ Wp, const uint32_t* __restrict__ Cp,
                                          uint32_t cum0, int base, int s1, const double* __restrict__ Mring,
                                          double* __restrict__ pendB, int32_t* __restrict__ pendA, int rmask, int lane, int lw)
{
    const double NEG_INF = -__builtin_inf();
    const int il = base + lane;
    const bool inb = lane < BL && il < s1;
    const uint32_t w = inb ? (uint32_t)Wp[il] : 0u;
    const uint32_t rel = inb ? Cp[il] - cum0 : 0u;
    const uint32_t fmax = wg_wave_max_u32(w);
    const double m = inb ? Mring[il & 127] : 0.0;
    const int thi = base + BL - 1 + (int)fmax;            // targets are < thi
    int tile = (base + 128) >> 6;
    tile += (lw - tile % NW + NW) % NW;                   // first tile at or after it that this worker owns
    for (; (tile << 6) < thi; tile += NW) {
        const int t = (tile << 6) + lane;
        const int sl = t & rmask;
        double best = wg_ld_l2_f64(pendB + sl);
        int32_t arg = wg_ld_l2_i32(pendA + sl);
        bool any = false;
#pragma unroll 1
        for (int g = 0; g < BL; g += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)w, g + u);
                const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)rel, g + u);
                const uint32_t j = (uint32_t)(t - (base + g + u));          // > 0
                v[u] = NEG_INF;
                if (f > 128u && j >= 128u && j < f) { v[u] = cb[(int64_t)r + j]; any = true; }
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const double cand = wg_readlane_f64(m, g + u) + v[u];
                const bool upd = cand > best;                                // ascending k, strict: first maximum
                best = upd ? cand : best;
                arg = upd ? base + g + u : arg;
            }
        }
        if (any) { pendB[sl] = best; pendA[sl] = arg; }
    }
}

// One step of a batch without blocks longer than 64 sites, in a job that has wider windows elsewhere (32-step
// batches; STP = lane of the step, known at compile time): fold M[k] into the 64 pending steps, read M[k+1] off lane
// STP, and let that lane move on to step k+64 with what earlier 65..128-site blocks left for it in (bestB, argB).
// On the chain: add -> max -> readlane (-> next add).  The comparison that maintains `arg` reads the OLD best and runs
// beside it (strict: the first maximum wins, segmentor.cpp:148).  No NaN can occur (finite scores, -inf fill), so
// the bare v_max_f64 stands for fmax.
template <int STP>
__device__ __forceinline__ void wg_dp_step32(double& best, int32_t& arg, uint32_t& tbk, double& Mk, const double cv,
                                             const int k, const int lane, const double bestB, const int32_t argB)
{
    const double cand = Mk + cv;
    const bool upd = cand > best;
    double nbest;
    asm("v_max_f64 %0, %1, %2" : "=v"(nbest) : "v"(best), "v"(cand));
    Mk = wg_readlane_f64(nbest, STP);                  // M[k+1]
    arg = upd ? k : arg;
    const int ak = __builtin_amdgcn_readlane(arg, STP);
    asm("v_writelane_b32 %0, %1, %2" : "+v"(tbk) : "s"(ak), "n"(STP));             // start of the best block ending at k
    const bool mine = lane == STP;                     // this lane moves on to step k+64
    best = mine ? bestB : nbest;
    arg = mine ? argB : arg;
}

// One step of a batch with blocks of up to 128 sites in the registers (longer ones are the workers' business): the
// same, plus the fold into the second pending register, which is re-armed when the lane moves on; M[k+1] stays with
// the lane (Mfin) for the workers.
template <int STP>
__device__ __forceinline__ void wg_dp_wide_step32(double& best, int32_t& arg, double& bestB, int32_t& argB, uint32_t& tbk,
                                                  uint32_t& Mfin_lo, uint32_t& Mfin_hi, double& Mk, const double ca, const double cb2,
                                                  const int k, const int lane, const uint32_t ninf_hi)
{
    const double candA = Mk + ca;
    const double candB = Mk + cb2;
    const bool updA = candA > best;
    double nbest;
    asm("v_max_f64 %0, %1, %2" : "=v"(nbest) : "v"(best), "v"(candA));
    Mk = wg_readlane_f64(nbest, STP);                  // M[k+1]
    arg = updA ? k : arg;
    const bool updB = candB > bestB;
    double nbestB;
    asm("v_max_f64 %0, %1, %2" : "=v"(nbestB) : "v"(bestB), "v"(candB));
    argB = updB ? k : argB;
    const int ak = __builtin_amdgcn_readlane(arg, STP);
    asm("v_writelane_b32 %0, %1, %2" : "+v"(tbk) : "s"(ak), "n"(STP));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(Mfin_lo) : "s"((uint32_t)__double_as_longlong(Mk)), "n"(STP));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(Mfin_hi) : "s"((uint32_t)(__double_as_longlong(Mk) >> 32)), "n"(STP));
    const bool mine = lane == STP;
    best = mine ? nbestB : nbest;
    arg = mine ? argB : arg;
    uint32_t lo = (uint32_t)__double_as_longlong(nbestB), hi = (uint32_t)(__double_as_longlong(nbestB) >> 32);
    asm("v_writelane_b32 %0, 0, %1" : "+v"(lo) : "n"(STP));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(hi) : "s"(ninf_hi), "n"(STP));
    bestB = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// 32 steps of a batch whose first step sits on lane STP0 (0 or 32), out of the arranged slot: 8 (4) rows at a time,
// the next rows already in flight.
template <int STP0>
__device__ __forceinline__ void wg_dp_batch32(double& best, int32_t& arg, uint32_t& tbk, double& Mk, const double* __restrict__ my,
                                              const int base, const int lane, const double bestB, const int32_t argB)
{
    double cur[8], nxt[8];
#pragma unroll
    for (int u = 0; u < 8; u++) cur[u] = my[u * 64];
#define WG_DP_GROUP32(G)                                                                                   \
    if ((G) + 8 < 32) {                                                                                    \
        _Pragma("unroll") for (int u = 0; u < 8; u++) nxt[u] = my[((G) + 8 + u) * 64];                     \
    }                                                                                                      \
    wg_dp_step32<STP0 + (G) + 0>(best, arg, tbk, Mk, cur[0], base + (G) + 0, lane, bestB, argB);           \
    wg_dp_step32<STP0 + (G) + 1>(best, arg, tbk, Mk, cur[1], base + (G) + 1, lane, bestB, argB);           \
    wg_dp_step32<STP0 + (G) + 2>(best, arg, tbk, Mk, cur[2], base + (G) + 2, lane, bestB, argB);           \
    wg_dp_step32<STP0 + (G) + 3>(best, arg, tbk, Mk, cur[3], base + (G) + 3, lane, bestB, argB);           \
    wg_dp_step32<STP0 + (G) + 4>(best, arg, tbk, Mk, cur[4], base + (G) + 4, lane, bestB, argB);           \
    wg_dp_step32<STP0 + (G) + 5>(best, arg, tbk, Mk, cur[5], base + (G) + 5, lane, bestB, argB);           \
    wg_dp_step32<STP0 + (G) + 6>(best, arg, tbk, Mk, cur[6], base + (G) + 6, lane, bestB, argB);           \
    wg_dp_step32<STP0 + (G) + 7>(best, arg, tbk, Mk, cur[7], base + (G) + 7, lane, bestB, argB);           \
    _Pragma("unroll") for (int u = 0; u < 8; u++) cur[u] = nxt[u];
    WG_DP_GROUP32(0) WG_DP_GROUP32(8) WG_DP_GROUP32(16) WG_DP_GROUP32(24)
#undef WG_DP_GROUP32
}

template <int STP0>
__device__ __forceinline__ void wg_dp_wide_batch32(double& best, int32_t& arg, double& bestB, int32_t& argB, uint32_t& tbk, double& Mfin,
                                                   double& Mk, const double* __restrict__ my, const double* __restrict__ myB,
                                                   const int base, const int lane)
{
    const uint32_t ninf_hi = 0xfff00000u;
    uint32_t mlo = 0, mhi = 0;
    double cur[4], curB[4], nxt[4], nxtB[4];          // 4 steps (~0.3 us) cover the LDS latency; 8 would spill at 16 waves
#pragma unroll
    for (int u = 0; u < 4; u++) { cur[u] = my[u * 64]; curB[u] = myB[u * 64]; }
#define WG_DP_WGROUP32(G)                                                                                                          \
    if ((G) + 4 < 32) {                                                                                                            \
        _Pragma("unroll") for (int u = 0; u < 4; u++) { nxt[u] = my[((G) + 4 + u) * 64]; nxtB[u] = myB[((G) + 4 + u) * 64]; }      \
    }                                                                                                                              \
    wg_dp_wide_step32<STP0 + (G) + 0>(best, arg, bestB, argB, tbk, mlo, mhi, Mk, cur[0], curB[0], base + (G) + 0, lane, ninf_hi);  \
    wg_dp_wide_step32<STP0 + (G) + 1>(best, arg, bestB, argB, tbk, mlo, mhi, Mk, cur[1], curB[1], base + (G) + 1, lane, ninf_hi);  \
    wg_dp_wide_step32<STP0 + (G) + 2>(best, arg, bestB, argB, tbk, mlo, mhi, Mk, cur[2], curB[2], base + (G) + 2, lane, ninf_hi);  \
    wg_dp_wide_step32<STP0 + (G) + 3>(best, arg, bestB, argB, tbk, mlo, mhi, Mk, cur[3], curB[3], base + (G) + 3, lane, ninf_hi);  \
    _Pragma("unroll") for (int u = 0; u < 4; u++) { cur[u] = nxt[u]; curB[u] = nxtB[u]; }
    WG_DP_WGROUP32(0) WG_DP_WGROUP32(4) WG_DP_WGROUP32(8) WG_DP_WGROUP32(12)
    WG_DP_WGROUP32(16) WG_DP_WGROUP32(20) WG_DP_WGROUP32(24) WG_DP_WGROUP32(28)
#undef WG_DP_WGROUP32
    Mfin = __longlong_as_double((long long)(((unsigned long long)mhi << 32) | mlo));
}

// The same step for a job whose windows are all <= 64 sites, with the lane of the step known at compile time (64-step
// batches start at lane 0): lane selects and the source mark are inline constants.  `arg` holds the LANE of the best
// source (a source is at most 63 steps back, so (STP - arg) mod 64 is the distance), the finished lane is re-armed
// with -inf by two v_writelane, and the source's lane goes to its lane of tbk by a third (turned into the block
// length once per batch).  10 VALU and no scalar arithmetic per step:
//     A  v_add_f64   cand = M[k] + row            C  v_cmp_gt_f64  upd = cand > best  (strict: first maximum, :148)
//     X  v_max_f64   best = max(best, cand)       D  v_cndmask     arg  = upd ? STP : arg
//     R  2 x v_readlane  M[k+1] = best[STP]       W  2 x v_writelane  best[STP] = -inf
//     K  v_readlane  ak = arg[STP]                T  v_writelane   tbk[STP] = ak
// Eight steps are one hand-scheduled asm block on fixed registers (best v[2:3], arg v4, tbk v5, cand v[6:7], M s[20:21],
// ak s22, the high word of -inf s23; the eight rows in v[8:23] or v[24:39], alternating between groups so that the
// next group's LDS reads land while this one runs).  Order per step: A C T(prev) X D Rlo Rhi Wlo Whi K — the chain
// A -> X -> R -> A(next) has independent instructions between its links, and the gfx940-family wait-state rules hold
// without a single s_nop inside: VALU-written VGPR -> v_readlane of it >= 1 instruction apart (X,D,R / D..K), VALU-written
// SGPR -> VALU reading it >= 2 apart (R..W,W,K..A / K,A,C,T).  A stand-alone wave issues this sequence in 18 ns per step
// (tools/micro/dp_chain.hip).
#define WG_DP64_STEP(CV, P, TPREV)                      \
    "v_add_f64 v[6:7], s[20:21], " CV "\n"              \
    "v_cmp_gt_f64 vcc, v[6:7], v[2:3]\n"                \
    TPREV                                               \
    "v_max_f64 v[2:3], v[2:3], v[6:7]\n"                \
    "v_cndmask_b32_e64 v4, v4, %[" P "], vcc\n"         \
    "v_readlane_b32 s20, v2, %[" P "]\n"                \
    "v_readlane_b32 s21, v3, %[" P "]\n"                \
    "v_writelane_b32 v2, 0, %[" P "]\n"                 \
    "v_writelane_b32 v3, s23, %[" P "]\n"               \
    "v_readlane_b32 s22, v4, %[" P "]\n"
#define WG_DP64_T(P) "v_writelane_b32 v5, s22, %[" P "]\n"

template <int G, int SET>
__device__ __forceinline__ void wg_dp_group64(double& best, int32_t& arg, uint32_t& tbk, double& Mk, const double (&cur)[8], const uint32_t ninf_hi)
{
    if (SET == 0)
        asm volatile(
            WG_DP64_STEP("v[8:9]", "p0", "")             WG_DP64_STEP("v[10:11]", "p1", WG_DP64_T("p0"))
            WG_DP64_STEP("v[12:13]", "p2", WG_DP64_T("p1")) WG_DP64_STEP("v[14:15]", "p3", WG_DP64_T("p2"))
            WG_DP64_STEP("v[16:17]", "p4", WG_DP64_T("p3")) WG_DP64_STEP("v[18:19]", "p5", WG_DP64_T("p4"))
            WG_DP64_STEP("v[20:21]", "p6", WG_DP64_T("p5")) WG_DP64_STEP("v[22:23]", "p7", WG_DP64_T("p6"))
            "s_nop 1\n" WG_DP64_T("p7")
            : "+{v[2:3]}"(best), "+{v4}"(arg), "+{v5}"(tbk), "+{s[20:21]}"(Mk)
            : "{v[8:9]}"(cur[0]), "{v[10:11]}"(cur[1]), "{v[12:13]}"(cur[2]), "{v[14:15]}"(cur[3]),
              "{v[16:17]}"(cur[4]), "{v[18:19]}"(cur[5]), "{v[20:21]}"(cur[6]), "{v[22:23]}"(cur[7]), "{s23}"(ninf_hi),
              [p0] "n"(G + 0), [p1] "n"(G + 1), [p2] "n"(G + 2), [p3] "n"(G + 3), [p4] "n"(G + 4), [p5] "n"(G + 5), [p6] "n"(G + 6), [p7] "n"(G + 7)
            : "v6", "v7", "s22", "vcc");
    else
        asm volatile(
            WG_DP64_STEP("v[24:25]", "p0", "")             WG_DP64_STEP("v[26:27]", "p1", WG_DP64_T("p0"))
            WG_DP64_STEP("v[28:29]", "p2", WG_DP64_T("p1")) WG_DP64_STEP("v[30:31]", "p3", WG_DP64_T("p2"))
            WG_DP64_STEP("v[32:33]", "p4", WG_DP64_T("p3")) WG_DP64_STEP("v[34:35]", "p5", WG_DP64_T("p4"))
            WG_DP64_STEP("v[36:37]", "p6", WG_DP64_T("p5")) WG_DP64_STEP("v[38:39]", "p7", WG_DP64_T("p6"))
            "s_nop 1\n" WG_DP64_T("p7")
            : "+{v[2:3]}"(best), "+{v4}"(arg), "+{v5}"(tbk), "+{s[20:21]}"(Mk)
            : "{v[24:25]}"(cur[0]), "{v[26:27]}"(cur[1]), "{v[28:29]}"(cur[2]), "{v[30:31]}"(cur[3]),
              "{v[32:33]}"(cur[4]), "{v[34:35]}"(cur[5]), "{v[36:37]}"(cur[6]), "{v[38:39]}"(cur[7]), "{s23}"(ninf_hi),
              [p0] "n"(G + 0), [p1] "n"(G + 1), [p2] "n"(G + 2), [p3] "n"(G + 3), [p4] "n"(G + 4), [p5] "n"(G + 5), [p6] "n"(G + 6), [p7] "n"(G + 7)
            : "v6", "v7", "s22", "vcc");
}

// The same eight steps for a job whose windows are all <= 60 sites (WG_NARROW_WMAX: every default-parameter genome): 6.75
// instead of 10 VALU instructions per step.  A row k then reaches at most step k + 59, so the lane a step has finished in
// sees no candidate for the next FIVE steps, and the bookkeeping of a finished lane need not follow it step by step: the four
// lanes of steps g .. g+3 hand their arg to tbk with ONE v_cndmask after step g+3 and are re-armed with -inf by two more during
// step g+4 (their next candidate can arrive with step g+5 at the earliest; until then their rows hold -inf, which neither
// max nor the strict compare lets through).  Per step: A C X D Rlo Rhi; the two instruction slots the wait-state rule puts
// between Rhi and the next A (VALU-written SGPR -> VALU reading it: 2 apart) carry the batched work, the scalar moves of the
// lane masks, or an s_nop.  Registers as above, plus the lane mask of the last four finished lanes in s[24:25] (carried from
// block to block: the re-arm of a block's last four lanes is the first thing the next block does; the batch's last four are
// re-armed by the caller) and -inf's high word in every lane of v40.
#define WG_DP64L_CORE(CV, P)                            \
    "v_add_f64 v[6:7], s[20:21], " CV "\n"              \
    "v_cmp_gt_f64 vcc, v[6:7], v[2:3]\n"                \
    "v_max_f64 v[2:3], v[2:3], v[6:7]\n"                \
    "v_cndmask_b32_e64 v4, v4, %[" P "], vcc\n"         \
    "v_readlane_b32 s20, v2, %[" P "]\n"                \
    "v_readlane_b32 s21, v3, %[" P "]\n"
#define WG_DP64L_REARM "v_cndmask_b32_e64 v2, v2, 0, s[24:25]\n" "v_cndmask_b32_e64 v3, v3, v40, s[24:25]\n"
#define WG_DP64L_CAPT  "v_cndmask_b32_e64 v5, v5, v4, s[24:25]\n" "s_nop 0\n"
#define WG_DP64L_MASK(LO, HI) "s_mov_b32 s24, %[" LO "]\n" "s_mov_b32 s25, %[" HI "]\n"
#define WG_DP64L_BLOCK(R0, R1, R2, R3, R4, R5, R6, R7, FIRST)                                             \
            WG_DP64L_CORE(R0, "p0") FIRST                                                                  \
            WG_DP64L_CORE(R1, "p1") "s_nop 1\n"                                                            \
            WG_DP64L_CORE(R2, "p2") WG_DP64L_MASK("ml0", "mh0")                                            \
            WG_DP64L_CORE(R3, "p3") WG_DP64L_CAPT                                                          \
            WG_DP64L_CORE(R4, "p4") WG_DP64L_REARM                                                         \
            WG_DP64L_CORE(R5, "p5") "s_nop 1\n"                                                            \
            WG_DP64L_CORE(R6, "p6") WG_DP64L_MASK("ml1", "mh1")                                            \
            WG_DP64L_CORE(R7, "p7") WG_DP64L_CAPT
#define WG_DP64L_OPERANDS(V0, V1, V2, V3, V4, V5, V6, V7)                                                  \
            : "+{v[2:3]}"(best), "+{v4}"(arg), "+{v5}"(tbk), "+{s[20:21]}"(Mk), "+{s[24:25]}"(mask)        \
            : V0(cur[0]), V1(cur[1]), V2(cur[2]), V3(cur[3]), V4(cur[4]), V5(cur[5]), V6(cur[6]), V7(cur[7]), "{v40}"(ninf_splat), \
              [p0] "n"(G + 0), [p1] "n"(G + 1), [p2] "n"(G + 2), [p3] "n"(G + 3), [p4] "n"(G + 4), [p5] "n"(G + 5), [p6] "n"(G + 6), [p7] "n"(G + 7), \
              [ml0] "n"((int)(uint32_t)(0xfull << G)), [mh0] "n"((int)(uint32_t)((0xfull << G) >> 32)),   \
              [ml1] "n"((int)(uint32_t)(0xfull << (G + 4))), [mh1] "n"((int)(uint32_t)((0xfull << (G + 4)) >> 32)) \
            : "v6", "v7", "vcc"

template <int G, int SET>
__device__ __forceinline__ void wg_dp_group64_lean(double& best, int32_t& arg, uint32_t& tbk, double& Mk, uint64_t& mask, const double (&cur)[8],
                                                   const uint32_t ninf_splat)
{
    if (SET == 0) {
        if (G == 0) asm volatile(WG_DP64L_BLOCK("v[8:9]", "v[10:11]", "v[12:13]", "v[14:15]", "v[16:17]", "v[18:19]", "v[20:21]", "v[22:23]", "s_nop 1\n")
                                 WG_DP64L_OPERANDS("{v[8:9]}", "{v[10:11]}", "{v[12:13]}", "{v[14:15]}", "{v[16:17]}", "{v[18:19]}", "{v[20:21]}", "{v[22:23]}"));
        else        asm volatile(WG_DP64L_BLOCK("v[8:9]", "v[10:11]", "v[12:13]", "v[14:15]", "v[16:17]", "v[18:19]", "v[20:21]", "v[22:23]", WG_DP64L_REARM)
                                 WG_DP64L_OPERANDS("{v[8:9]}", "{v[10:11]}", "{v[12:13]}", "{v[14:15]}", "{v[16:17]}", "{v[18:19]}", "{v[20:21]}", "{v[22:23]}"));
    } else {
        asm volatile(WG_DP64L_BLOCK("v[24:25]", "v[26:27]", "v[28:29]", "v[30:31]", "v[32:33]", "v[34:35]", "v[36:37]", "v[38:39]", WG_DP64L_REARM)
                     WG_DP64L_OPERANDS("{v[24:25]}", "{v[26:27]}", "{v[28:29]}", "{v[30:31]}", "{v[32:33]}", "{v[34:35]}", "{v[36:37]}", "{v[38:39]}"));
    }
}

// Round 3: the same lean block for the NARROW batches (every window <= 60 sites) of a job that has wider windows elsewhere (CpG
// islands: ~6 % of the 16-site units of an hg19-like genome) — 94 % of such a job's steps used to run the compiler-scheduled
// 14-instruction step of wg_dp_step32.  What differs from the block above: `arg` holds the absolute source site (a pending maximum
// may come from the second register or the ring: sources up to thousands of sites back), kept in v45; and a finished lane does not restart from -inf but MOVES ON to what earlier 65..128-site blocks left for its next step in
// (bestB v[42:43], argB v44), which is then re-armed: per four steps one capture, three moves and two re-arms (v_cndmask under the lane
// mask of the four finished lanes), and the source site advances by one v_add per step — 7 + 6/4 = 8.5 VALU per step (14 before).  The deferral is safe for the same reason as above:
// rows of a narrow batch reach at most 59 steps ahead, so neither a finished lane's next step (64 ahead) nor any second register is
// touched by the rows of the four steps in between.  32-step batches (the slot holds both planes for the wide ones), first lane G0.
#define WG_DP32W_CORE(CV, P)                            \
    "v_add_f64 v[6:7], s[20:21], " CV "\n"              \
    "v_cmp_gt_f64 vcc, v[6:7], v[2:3]\n"                \
    "v_max_f64 v[2:3], v[2:3], v[6:7]\n"                \
    "v_cndmask_b32_e32 v4, v4, v45, vcc\n"              \
    "v_readlane_b32 s20, v2, %[" P "]\n"                \
    "v_readlane_b32 s21, v3, %[" P "]\n"                \
    "v_add_u32_e32 v45, 1, v45\n"
// (the source site k of the step lives in every lane of v45: a v_cndmask cannot take it from a scalar register beside vcc)
#define WG_DP32W_MOVE  "v_cndmask_b32_e64 v2, v2, v42, s[24:25]\n" "v_cndmask_b32_e64 v3, v3, v43, s[24:25]\n" "v_cndmask_b32_e64 v4, v4, v44, s[24:25]\n"
#define WG_DP32W_REARM "v_cndmask_b32_e64 v42, v42, 0, s[24:25]\n" "v_cndmask_b32_e64 v43, v43, v40, s[24:25]\n"
#define WG_DP32W_CAPT  "v_cndmask_b32_e64 v5, v5, v4, s[24:25]\n"
// After four steps g .. g+3 (mask = their lanes): the capture right behind step g+3, the three moves right behind step g+4 (a
// finished lane's next candidate can arrive with step g+5), the re-arm of the second register behind step g+5; the mask of the
// next four lanes is set a step ahead of its first use.  A block starts with the moves of the PREVIOUS block's last four lanes.
#define WG_DP32W_BLOCK(R0, R1, R2, R3, R4, R5, R6, R7)                                                     \
            WG_DP32W_CORE(R0, "p0") WG_DP32W_MOVE                                                          \
            WG_DP32W_CORE(R1, "p1") WG_DP32W_REARM                                                         \
            WG_DP32W_CORE(R2, "p2") WG_DP64L_MASK("ml0", "mh0")                                            \
            WG_DP32W_CORE(R3, "p3") WG_DP32W_CAPT                                                          \
            WG_DP32W_CORE(R4, "p4") WG_DP32W_MOVE                                                          \
            WG_DP32W_CORE(R5, "p5") WG_DP32W_REARM                                                         \
            WG_DP32W_CORE(R6, "p6") WG_DP64L_MASK("ml1", "mh1")                                            \
            WG_DP32W_CORE(R7, "p7") WG_DP32W_CAPT
#define WG_DP32W_OPERANDS(V0, V1, V2, V3, V4, V5, V6, V7)                                                  \
            : "+{v[2:3]}"(best), "+{v4}"(arg), "+{v5}"(tbk), "+{s[20:21]}"(Mk), "+{s[24:25]}"(mask), "+{v45}"(kk), "+{v[42:43]}"(bestB), "+{v44}"(argB) \
            : V0(cur[0]), V1(cur[1]), V2(cur[2]), V3(cur[3]), V4(cur[4]), V5(cur[5]), V6(cur[6]), V7(cur[7]), "{v40}"(ninf_splat), \
              [p0] "n"(G + 0), [p1] "n"(G + 1), [p2] "n"(G + 2), [p3] "n"(G + 3), [p4] "n"(G + 4), [p5] "n"(G + 5), [p6] "n"(G + 6), [p7] "n"(G + 7), \
              [ml0] "n"((int)(uint32_t)(0xfull << G)), [mh0] "n"((int)(uint32_t)((0xfull << G) >> 32)),   \
              [ml1] "n"((int)(uint32_t)(0xfull << (G + 4))), [mh1] "n"((int)(uint32_t)((0xfull << (G + 4)) >> 32)) \
            : "v6", "v7", "vcc"

// Eight steps on lanes G .. G+7.  `mask` enters as the lane mask of the PREVIOUS four finished lanes (0: none — the first block of a
// batch), whose moves this block begins with, and leaves as the mask of lanes G+4 .. G+7 (their capture done, their moves pending).
template <int G, int SET>
__device__ __forceinline__ void wg_dp_group32w(double& best, int32_t& arg, double& bestB, int32_t& argB, uint32_t& tbk, double& Mk, uint64_t& mask,
                                               uint32_t& kk, const double (&cur)[8], const uint32_t ninf_splat)
{
    if (SET == 0) asm volatile(WG_DP32W_BLOCK("v[8:9]", "v[10:11]", "v[12:13]", "v[14:15]", "v[16:17]", "v[18:19]", "v[20:21]", "v[22:23]")
                               WG_DP32W_OPERANDS("{v[8:9]}", "{v[10:11]}", "{v[12:13]}", "{v[14:15]}", "{v[16:17]}", "{v[18:19]}", "{v[20:21]}", "{v[22:23]}"));
    else          asm volatile(WG_DP32W_BLOCK("v[24:25]", "v[26:27]", "v[28:29]", "v[30:31]", "v[32:33]", "v[34:35]", "v[36:37]", "v[38:39]")
                               WG_DP32W_OPERANDS("{v[24:25]}", "{v[26:27]}", "{v[28:29]}", "{v[30:31]}", "{v[32:33]}", "{v[34:35]}", "{v[36:37]}", "{v[38:39]}"));
}

// 32 steps of a narrow batch (every window <= 60) of a wide job, first step on lane STP0 (0 or 32), first site `base`
template <int STP0>
__device__ __forceinline__ void wg_dp_batch32w(double& best, int32_t& arg, double& bestB, int32_t& argB, uint32_t& tbk, double& Mk,
                                               const double* __restrict__ my, const int base, const int lane)
{
    const uint32_t ninf_hi = 0xfff00000u;
    const double NEG_INF = -__builtin_inf();
    double Ms = wg_readlane_f64(Mk, 0);
    uint32_t kk = (uint32_t)base;                      // source site of the step, in every lane
    uint64_t mask = 0;
    double ra[8], rb[8];
#pragma unroll
    for (int u = 0; u < 8; u++) ra[u] = my[u * 64];
#pragma unroll
    for (int u = 0; u < 8; u++) rb[u] = my[(8 + u) * 64];
    wg_dp_group32w<STP0 + 0, 0>(best, arg, bestB, argB, tbk, Ms, mask, kk, ra, ninf_hi);
#pragma unroll
    for (int u = 0; u < 8; u++) ra[u] = my[(16 + u) * 64];
    wg_dp_group32w<STP0 + 8, 1>(best, arg, bestB, argB, tbk, Ms, mask, kk, rb, ninf_hi);
#pragma unroll
    for (int u = 0; u < 8; u++) rb[u] = my[(24 + u) * 64];
    wg_dp_group32w<STP0 + 16, 0>(best, arg, bestB, argB, tbk, Ms, mask, kk, ra, ninf_hi);
    wg_dp_group32w<STP0 + 24, 1>(best, arg, bestB, argB, tbk, Ms, mask, kk, rb, ninf_hi);
    // the batch's last four finished lanes: their capture is done, their move to the second register is not
    const bool last4 = lane >= STP0 + 28 && lane < STP0 + 32;
    best = last4 ? bestB : best;
    arg = last4 ? argB : arg;
    bestB = last4 ? NEG_INF : bestB;
    Mk = Ms;
}

// state kept per chunk in global memory: [0] M[k] of the next step, [1..64] best, [65..128] arg, [129..192] bestB,
// [193..256] argB (args as doubles' bits) — written only between stages — then the ring: ringN doubles, ringN int32
// (Tried in round 3: the wavefront that shares the recurrence wavefront's SIMD — a workgroup's wavefronts go to the four SIMDs in
// cyclic order, so wavefront 4 — leaving at once, six or five workers on the other three SIMDs: 1.63 -> 1.67 / 1.76 ms for the 483
// chunks of hg19, 3.00 -> 3.05 / 3.04 for a 61-chunk share: the workers' issue slots on that SIMD are not what the step waits for.)
template <int NW, int BL, bool LEAN = false>      // LEAN (BL == 64 only): every window of the job is <= WG_NARROW_WMAX sites
__global__ __launch_bounds__(64 * (1 + NW)) void k_dp(JobView J, StageView SV, const double* __restrict__ cost, DpArgs A,
                                                      double* __restrict__ state, int64_t state_stride)
{
    constexpr bool WIDEJOB = BL < 64;
    constexpr int SLOT = BL * 64 * (WIDEJOB ? 2 : 1);                         // doubles per LDS slot: A (+ B)
    extern __shared__ __attribute__((aligned(16))) char smem_dp[];
    double* slots = reinterpret_cast<double*>(smem_dp);                       // [2][SLOT]
    double* Mring = slots + 2 * SLOT;                                         // [128] M[k] of the last batches' steps
    double* pendLB = Mring + 128;                                             // [2][32] ring entries of the next batches' steps
    int32_t* pendLA = reinterpret_cast<int32_t*>(pendLB + 64);                // [2][32]
    uint32_t* kinds = reinterpret_cast<uint32_t*>(pendLA + 64);               // [2] 1: the slot holds a wide batch (A and B)
    uint32_t* metaC = kinds + 4;                                              // [1024] row offsets of the coming steps
    uint16_t* metaW = reinterpret_cast<uint16_t*>(metaC + WG_DP_META_RING);   // [1024] their windows
    const int lane = threadIdx.x & 63;
    // (Tried: rotating which wavefront of the workgroup runs the recurrence by dispatch round, so that two workgroups sharing
    // a CU do not both put theirs on wavefront 0's SIMD — 1.75 -> 2.35 ms for 483 chunks: worse; wavefront 0 it stays.)
    // (Round 3, measured: the wavefront index in a SCALAR register (readfirstlane) turns a worker's row indices into scalar arithmetic and
    // removes a v_readfirstlane + four wait states per row — and the recurrence gets SLOWER: 1.62 -> 1.71 ms for hg19, 4.29 -> 5.04 ms
    // with CpG islands (profiles/r03_dp_experiments.txt).  The workers' idle slots are slots the recurrence wavefront gets.)
    // (Round 5, measured with the HW_ID probe of the timing build: the two workgroups of a CU already have their recurrence wavefronts on
    // DIFFERENT SIMDs — the dispatcher rotates the SIMD of a workgroup's first wavefront — so there is nothing to select: profiles/r05_dp_placement.txt.)
    const int wvl = (int)(threadIdx.x >> 6);          // 0 = recurrence
    const bool worker = wvl != 0;
    const int lw = wvl - 1;                           // worker index (0..NW-1)
    const int c = blockIdx.x;
    const int nC = J.n_chunks;
    const ChunkDesc cd = J.chunks[c];
    const int s0 = SV.sb[SV.stage];
    if (s0 >= cd.len) return;
    const int s1 = (SV.sb[SV.stage + 1] < cd.len) ? SV.sb[SV.stage + 1] : cd.len;
    const int rmask = A.ringN - 1;
    double* gs = state + (int64_t)c * state_stride;
    double* pendB = gs + WG_DP_STATE_HDR;                                     // [ringN]
    int32_t* pendA = reinterpret_cast<int32_t*>(pendB + A.ringN);             // [ringN]
    const double* cb = cost + SV.cbase[(int64_t)SV.stage * nC + c];
    const uint32_t cum0 = SV.cum0[(int64_t)SV.stage * nC + c];
    const uint16_t* Wp = J.W16 + cd.site_off;
    const uint32_t* Cp = J.cum32 + cd.site_off;
    const double NEG_INF = -__builtin_inf();
    const int nb = (s1 - s0 + BL - 1) / BL;

    if (!worker) __builtin_amdgcn_s_setprio(3);        // the recurrence is one dependent chain: let it win issue arbitration
    double best = NEG_INF, bestB = NEG_INF;             // pending steps of this lane (wave 0)
    int32_t arg = 0, argB = 0;
    double Mk = 0.0;                                    // M[k] of the step about to run; M[0] = 0 (segmentor.cpp:97)
    DpRows<NW, BL> rows;                                // (workers) rows of the batch after the next, in flight
    DpRefill<NW> refill;                                // (workers) a region of windows / row offsets, in flight
    uint32_t fm_prev = 0, fm_cur = 0, fm_n1 = 0;        // (workers) widest window of batches b-1, b, b+1
    constexpr int RB = WG_DP_META_REGION / BL;          // batches per region of the meta ring
    if (worker) {
        if (WIDEJOB && s0 == 0)
            for (int x = lw * 64 + lane; x < A.ringN; x += 64 * NW) pendB[x] = NEG_INF;
        wg_dp_refill_issue<NW>(refill, Wp, Cp, cum0, s0, s1, lane, lw);
        wg_dp_refill_commit<NW>(refill, metaW, metaC, 0, lane, lw);
        wg_dp_refill_issue<NW>(refill, Wp, Cp, cum0, s0 + WG_DP_META_REGION, s1, lane, lw);
        wg_dp_refill_commit<NW>(refill, metaW, metaC, WG_DP_META_REGION, lane, lw);
        if (WIDEJOB && lw == 0 && lane < BL) {
            double v = NEG_INF;
            int32_t a = 0;
            if (s0 != 0) {
                const int sl = (s0 + lane) & rmask;
                v = wg_ld_l2_f64(pendB + sl); a = wg_ld_l2_i32(pendA + sl);
                pendB[sl] = NEG_INF;
            }
            pendLB[lane] = v; pendLA[lane] = a;
        }
    } else if (s0 != 0) {
        Mk = gs[0];
        best = gs[1 + lane];
        arg = (int32_t)__double_as_longlong(gs[65 + lane]);
        if (WIDEJOB) {
            bestB = gs[129 + lane];
            argB = (int32_t)__double_as_longlong(gs[193 + lane]);
        }
        // have the loaded state in registers HERE: otherwise the wait for it is placed inside the loop, where it would
        // also sit out the latency of the loop's own global stores, every batch
        asm volatile("" : "+v"(Mk), "+v"(best), "+v"(arg), "+v"(bestB), "+v"(argB));
    }
    __syncthreads();
    if (worker) {
        wg_dp_rows_issue<NW, BL>(rows, cb, wg_dp_meta_lds<BL>(metaW, metaC, 0, lane), s0, lane, lw);
        wg_dp_rows_commit<NW, BL>(rows, slots, slots + BL * 64, kinds, lane, lw);
        fm_cur = rows.fmax;
        if (nb > 1) { wg_dp_rows_issue<NW, BL>(rows, cb, wg_dp_meta_lds<BL>(metaW, metaC, BL, lane), s0 + BL, lane, lw); fm_n1 = rows.fmax; }
    }
    __syncthreads();

    // Two loops, one per role, with the same barriers: inside ONE loop with a branch the register allocator has to keep the workers'
    // rows and the recurrence's state alive side by side (a wavefront only ever runs one side, but that is not visible to it).
    // LDS is all that must be settled at a barrier.  Global memory: nobody in the workgroup reads what the recurrence wave stores; a
    // ring entry is only ever updated by its owner wave, and fetched (by worker 0) no sooner than two barriers after its last update
    // — by then the updating wave has waited for loads it issued after that store (the row loads that end every batch), and vector
    // memory operations of a wave complete in order.  Waiting for store latency there, every 32 steps, would cost more than the steps.
#define WG_DP_BARRIER asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    WG_DP_T(uint64_t dbg_wait = 0; uint64_t dbg_vm = 0; uint64_t dbg_commit = 0; uint64_t dbg_issue = 0; const uint64_t dbg_t0 = __builtin_amdgcn_s_memtime();)
    WG_DP_T(const int dbg_mode = A.pad[1];      /* timing builds only (WGBSSEG_DP_DEBUG): 1 workers leave, 2 no commits, 4 no row loads — WRONG results, timing only */
            if (worker && (dbg_mode & 1)) return;)
    if (worker) {
      for (int b = 0; b < nb; b++) {
        const int base = s0 + b * BL;
        {
            // Everything a worker loads it consumes a full batch later.  Order of the batch: (1) ring entries of the next
            // batch's steps: loads, and the reset store right behind them (the memory pipeline keeps a wave's accesses
            // to one address in order); (2) meta ring refill; (3) pushes of the blocks > 128 sites of batch b-1;
            // (4) the rows of batch b+1, loaded a batch ago, into the free slot; (5) the loads of the rows of batch b+2.
            // The row loads come LAST on purpose: the next batch waits for them before its barrier, and vector memory
            // operations of a wave complete in order — so every ring store of this batch is in L2 one barrier from now.
            const bool fetch = WIDEJOB && b + 1 < nb && lw == 0 && lane < BL;
            const int fsl = (base + BL + lane) & rmask;
            double fv = NEG_INF;
            int32_t fa = 0;
            if (fetch) {                                          // complete: their sources lie >= 128 sites back
                fv = wg_ld_l2_f64(pendB + fsl); fa = wg_ld_l2_i32(pendA + fsl);
                pendB[fsl] = NEG_INF;
            }
            // regions 0 and 1 are in the ring from the start; region r+1 replaces region r-1 when the row loads have
            // moved on to region r (they run two batches ahead of b), and is first read RB-3 batches later
            if (b >= RB && b % RB == 0)      wg_dp_refill_issue<NW>(refill, Wp, Cp, cum0, s0 + (b / RB + 1) * WG_DP_META_REGION, s1, lane, lw);
            else if (b > RB && b % RB == 1)  wg_dp_refill_commit<NW>(refill, metaW, metaC, (b / RB + 1) * WG_DP_META_REGION, lane, lw);
            if (WIDEJOB && b >= 1 && fm_prev > 128u)
                wg_dp_far<NW, BL>(cb, Wp, Cp, cum0, base - BL, s1, Mring, pendB, pendA, rmask, lane, lw);
            // (Tried three times for the 64-step batches: a second register set so that rows are loaded THREE batches ahead.  Before the
            // loop was split by role it did not fit the registers (137 VGPRs with seven workers; nine wavefronts per workgroup with
            // eight: one workgroup per CU, 1.75 -> 2.9-3.0 ms); after the split it fits (84-114 VGPRs) and, with every load out of its
            // branch so that the waits leave the younger set in flight, measures 1.79 against 1.61 ms: the rows are not what the
            // recurrence waits for.)
            WG_DP_T(const uint64_t tc0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg_vm += __builtin_amdgcn_s_memtime() - tc0;)
            if (b + 1 < nb WG_DP_T(&& !(dbg_mode & 2)))
                wg_dp_rows_commit<NW, BL>(rows, slots + (size_t)((b + 1) & 1) * SLOT, slots + (size_t)((b + 1) & 1) * SLOT + BL * 64,
                                          kinds + ((b + 1) & 1), lane, lw);
            WG_DP_T(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dbg_commit += __builtin_amdgcn_s_memtime() - tc0;)
            if (fetch) {
                pendLB[((b + 1) & 1) * 32 + lane] = fv;
                pendLA[((b + 1) & 1) * 32 + lane] = fa;
            }
            uint32_t fm_n2 = 0;
            WG_DP_T(const uint64_t ti0 = __builtin_amdgcn_s_memtime();)
            if (b + 2 < nb WG_DP_T(&& !(dbg_mode & 4))) {
                wg_dp_rows_issue<NW, BL>(rows, cb, wg_dp_meta_lds<BL>(metaW, metaC, (b + 2) * BL, lane), base + 2 * BL, lane, lw);
                fm_n2 = rows.fmax;
            }
            WG_DP_T(dbg_issue += __builtin_amdgcn_s_memtime() - ti0;)
            fm_prev = fm_cur; fm_cur = fm_n1; fm_n1 = fm_n2;
        }
        WG_DP_T(const uint64_t tb0 = __builtin_amdgcn_s_memtime();)
        WG_DP_BARRIER;
        WG_DP_T(dbg_wait += __builtin_amdgcn_s_memtime() - tb0;)
      }
    } else {
      for (int b = 0; b < nb; b++) {
        const int base = s0 + b * BL;
        {
            const double* slot = slots + (size_t)(b & 1) * SLOT;
            const bool wideb = WIDEJOB && kinds[b & 1] == 1u;
            const int stp0 = WIDEJOB ? (base & 63) : 0;          // lane of the batch's first step
            const int d = (lane - stp0) & 63;                    // this lane finishes step base + d (in this batch iff d < BL)
            const bool fin = d < BL;
            if (WIDEJOB && fin) {
                // ring entry of the step this lane is about to finish: all its candidates precede the ones to come
                const double pb = pendLB[(b & 1) * 32 + d];
                const int32_t pa = pendLA[(b & 1) * 32 + d];
                const bool mrg = pb >= best;
                best = mrg ? pb : best;
                arg = mrg ? pa : arg;
            }
            uint32_t tbk = 0;
            const double* my = slot + lane;
            if (!wideb) {
                // ---- no block of the batch is longer than 64 sites: BL steps, 8 at a time, next 8 rows in flight ----
                if (!WIDEJOB) {
                    const uint32_t ninf_hi = 0xfff00000u;
                    double Ms = wg_readlane_f64(Mk, 0);   // M[k] is wave-uniform: the asm blocks want it in scalar registers
                    double ra[8], rb[8];                  // the rows of two consecutive groups of 8 steps (register sets 0 and 1)
#pragma unroll
                    for (int u = 0; u < 8; u++) ra[u] = my[u * 64];
                    uint64_t lmask = 0;                   // (LEAN) lane mask of the last four finished lanes, carried between the blocks
#define WG_DP_PAIR64(G)                                                                                   \
                    _Pragma("unroll") for (int u = 0; u < 8; u++) rb[u] = my[((G) + 8 + u) * 64];         \
                    if (LEAN) wg_dp_group64_lean<(G), 0>(best, arg, tbk, Ms, lmask, ra, ninf_hi);         \
                    else wg_dp_group64<(G), 0>(best, arg, tbk, Ms, ra, ninf_hi);                          \
                    if ((G) + 16 < 64) {                                                                  \
                        _Pragma("unroll") for (int u = 0; u < 8; u++) ra[u] = my[((G) + 16 + u) * 64];    \
                    }                                                                                     \
                    if (LEAN) wg_dp_group64_lean<(G) + 8, 1>(best, arg, tbk, Ms, lmask, rb, ninf_hi);     \
                    else wg_dp_group64<(G) + 8, 1>(best, arg, tbk, Ms, rb, ninf_hi);
                    WG_DP_PAIR64(0) WG_DP_PAIR64(16) WG_DP_PAIR64(32) WG_DP_PAIR64(48)
#undef WG_DP_PAIR64
                    if (LEAN) best = lane >= 60 ? NEG_INF : best;     // the batch's last four finished lanes (the blocks re-arm the others)
                    Mk = Ms;
                    tbk = (((uint32_t)lane - tbk) & 63u) + 1u;        // source lane -> length of the best block ending here
                } else if (kinds[b & 1] == 0u && A.pad[0] == 0) {     // every window of the batch <= 60: the hand-scheduled lean step (pad[0]: off, A/B)
                    if (stp0 == 0) wg_dp_batch32w<0>(best, arg, bestB, argB, tbk, Mk, my, base, lane);
                    else           wg_dp_batch32w<32>(best, arg, bestB, argB, tbk, Mk, my, base, lane);
                } else if (stp0 == 0) {
                    wg_dp_batch32<0>(best, arg, tbk, Mk, my, base, lane, bestB, argB);
                    bestB = fin ? NEG_INF : bestB;
                } else {
                    wg_dp_batch32<32>(best, arg, tbk, Mk, my, base, lane, bestB, argB);
                    bestB = fin ? NEG_INF : bestB;
                }
                if (WIDEJOB) tbk = (uint32_t)(base + d + 1) - tbk;    // start of the best block -> its length
            } else {
                const double* myB = my + BL * 64;
                double Mfin = 0.0;
                if (lane == 0) Mring[base & 127] = Mk;
                if (stp0 == 0) wg_dp_wide_batch32<0>(best, arg, bestB, argB, tbk, Mfin, Mk, my, myB, base, lane);
                else           wg_dp_wide_batch32<32>(best, arg, bestB, argB, tbk, Mfin, Mk, my, myB, base, lane);
                tbk = (uint32_t)(base + d + 1) - tbk;
                if (fin) Mring[(base + d + 1) & 127] = Mfin;      // M[k+1] of the batch's steps, for the workers
            }
            if (fin && base + d < s1) J.back16[cd.site_off + base + d] = (uint16_t)tbk;
        }
        WG_DP_T(const uint64_t tb0 = __builtin_amdgcn_s_memtime();)
        WG_DP_BARRIER;
        WG_DP_T(dbg_wait += __builtin_amdgcn_s_memtime() - tb0;)
      }
    }
#undef WG_DP_BARRIER
    // timing diagnostics (-DWGBSSEG_DP_TIMING builds only; tools/dp_timing.py): [wavefront 0 | worker 0] x [loop cycles, of which at the
    // barrier], then worker 0's wait for its rows, its arranging and its load-issue phase — left in the chunk's state slots
    WG_DP_T(if (s1 >= cd.len && lane == 0 && wvl <= 1) {
        gs[2 * wvl] = (double)(__builtin_amdgcn_s_memtime() - dbg_t0);
        gs[2 * wvl + 1] = (double)dbg_wait;
        if (wvl == 1) { gs[4] = (double)dbg_vm; gs[5] = (double)dbg_commit; gs[6] = (double)dbg_issue; }
    })
    // ... and where the workgroup ran: HW_ID of every wavefront by ROLE (slot 8 = the recurrence), XCC_ID, and when (s_memtime)
    WG_DP_T(if (s1 >= cd.len && lane == 0) {
        gs[8 + wvl] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        if (wvl == 0) { gs[16] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 20); gs[17] = (double)dbg_t0; gs[18] = (double)__builtin_amdgcn_s_memtime(); }
        gs[20 + wvl] = (double)dbg_wait;                     // barrier wait of every role (slot 20 = the recurrence), load-issue phase of every worker
        gs[40 + wvl] = (double)dbg_issue;
    })
    if (WIDEJOB && worker && fm_prev > 128u)
        wg_dp_far<NW, BL>(cb, Wp, Cp, cum0, s0 + (nb - 1) * BL, s1, Mring, pendB, pendA, rmask, lane, lw);
    if (!worker && s1 < cd.len) {
        if (lane == 0) gs[0] = Mk;
        gs[1 + lane] = best;
        gs[65 + lane] = __longlong_as_double((long long)arg);
        if (WIDEJOB) {
            gs[129 + lane] = bestB;
            gs[193 + lane] = __longlong_as_double((long long)argB);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_dp16<NW>: the recurrence of a job whose windows are all <= 64 sites (k_dp<.,64>'s case) in 16-step batches, built to
// run BESIDE the scoring kernel: 24 KB of LDS instead of 73 KB and 1 + NW = 4 wavefronts, i.e. the footprint of ONE scoring
// workgroup — when the stages of a many-chunk job are pipelined (scoring of stage s+1 on stream A, recurrence of stage s
// on stream B) a recurrence workgroup takes the place of a single scoring workgroup on its CU instead of waiting for three
// of them to retire.  Same push form, same 8-step hand-scheduled groups (wg_dp_group64), same arranged rows
// (-inf where a lane holds no candidate).  What differs: a barrier every 16 steps, so the workers keep the rows of FOUR
// batches in flight in registers (a 16-step batch lasts ~0.4 us, a row load under scoring traffic 1-2 us); the batch
// loop is unrolled by four, which makes the lane of a batch's first step (0, 16, 32, 48) and the register set of the
// rows in flight compile-time constants; back-pointers are stored once per 64 steps.
// ------------------------------------------------------------------------------------------------------------
template <int NW>
struct Dp16Rows { static constexpr int PER = (16 + NW - 1) / NW; double va[PER]; };

template <int NW>
__device__ __forceinline__ void wg_dp16_issue(Dp16Rows<NW>& R, const double* __restrict__ cb, const uint16_t* __restrict__ metaW,
                                              const uint32_t* __restrict__ metaC, int base_rel, int base, int lane, int lw)
{
    const int x = (base_rel + lane) & (WG_DP_META_RING - 1);
    const uint32_t w = lane < 16 ? (uint32_t)metaW[x] : 0u;
    const uint32_t rel = metaC[x];
    const int stp0 = base & 63;
#pragma unroll
    for (int q = 0; q < Dp16Rows<NW>::PER; q++) {
        const int sidx = lw + q * NW;
        R.va[q] = -__builtin_inf();
        if (sidx < 16) {
            const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)w, sidx);
            const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)rel, sidx);
            const uint32_t j = (uint32_t)(lane - stp0 - sidx) & 63u;
            if (j < f) R.va[q] = cb[(int64_t)r + j];
        }
    }
}

template <int NW>
__device__ __forceinline__ void wg_dp16_commit(const Dp16Rows<NW>& R, double* __restrict__ slot, int lane, int lw)
{
#pragma unroll
    for (int q = 0; q < Dp16Rows<NW>::PER; q++) {
        const int sidx = lw + q * NW;
        if (sidx < 16) slot[sidx * 64 + lane] = R.va[q];
    }
}

template <int NW>
__global__ __launch_bounds__(64 * (1 + NW)) void k_dp16(JobView J, StageView SV, const double* __restrict__ cost,
                                                         double* __restrict__ state, int64_t state_stride)
{
    constexpr int BL = 16, D = 4;
    constexpr int RB = WG_DP_META_REGION / BL;                                // batches per region of the meta ring
    extern __shared__ __attribute__((aligned(16))) char smem_dp[];
    double* slots = reinterpret_cast<double*>(smem_dp);                       // [2][16 * 64]
    uint32_t* metaC = reinterpret_cast<uint32_t*>(slots + 2 * BL * 64);       // [1024] row offsets of the coming steps
    uint16_t* metaW = reinterpret_cast<uint16_t*>(metaC + WG_DP_META_RING);   // [1024] their windows
    const int lane = threadIdx.x & 63;
    const bool worker = threadIdx.x >= 64;
    const int lw = (int)(threadIdx.x >> 6) - 1;
    const int c = blockIdx.x;
    const int nC = J.n_chunks;
    const ChunkDesc cd = J.chunks[c];
    const int s0 = SV.sb[SV.stage];
    if (s0 >= cd.len) return;
    const int s1 = (SV.sb[SV.stage + 1] < cd.len) ? SV.sb[SV.stage + 1] : cd.len;
    double* gs = state + (int64_t)c * state_stride;
    const double* cb = cost + SV.cbase[(int64_t)SV.stage * nC + c];
    const uint32_t cum0 = SV.cum0[(int64_t)SV.stage * nC + c];
    const uint16_t* Wp = J.W16 + cd.site_off;
    const uint32_t* Cp = J.cum32 + cd.site_off;
    const int nb = (s1 - s0 + BL - 1) / BL;

    if (!worker) __builtin_amdgcn_s_setprio(3);
    double best = -__builtin_inf();
    int32_t arg = 0;
    double Mk = 0.0;                                    // M[0] = 0 (segmentor.cpp:97)
    Dp16Rows<NW> r0, r1, r2, r3;                        // (workers) the rows of four batches in flight: batch q in set q mod 4
    DpRefill<NW> refill;
    if (worker) {
        wg_dp_refill_issue<NW>(refill, Wp, Cp, cum0, s0, s1, lane, lw);
        wg_dp_refill_commit<NW>(refill, metaW, metaC, 0, lane, lw);
        wg_dp_refill_issue<NW>(refill, Wp, Cp, cum0, s0 + WG_DP_META_REGION, s1, lane, lw);
        wg_dp_refill_commit<NW>(refill, metaW, metaC, WG_DP_META_REGION, lane, lw);
    } else if (s0 != 0) {
        Mk = gs[0];
        best = gs[1 + lane];
        arg = (int32_t)__double_as_longlong(gs[65 + lane]);
        asm volatile("" : "+v"(Mk), "+v"(best), "+v"(arg));
    }
    __syncthreads();
    if (worker) {
        wg_dp16_issue<NW>(r0, cb, metaW, metaC, 0, s0, lane, lw);
        wg_dp16_commit<NW>(r0, slots, lane, lw);
        if (1 < nb) wg_dp16_issue<NW>(r1, cb, metaW, metaC, 1 * BL, s0 + 1 * BL, lane, lw);
        if (2 < nb) wg_dp16_issue<NW>(r2, cb, metaW, metaC, 2 * BL, s0 + 2 * BL, lane, lw);
        if (3 < nb) wg_dp16_issue<NW>(r3, cb, metaW, metaC, 3 * BL, s0 + 3 * BL, lane, lw);
        if (4 < nb) wg_dp16_issue<NW>(r0, cb, metaW, metaC, 4 * BL, s0 + 4 * BL, lane, lw);
    }
    __syncthreads();

    const uint32_t ninf_hi = 0xfff00000u;
    uint32_t tbk = 0;
    // one batch: Q = b mod 4 (compile time): the batch's first step sits on lane 16 Q; the set holding batch b+1 is (Q+1) mod 4
    // (two loops, one per role, with the same barriers: see k_dp)
#define WG_DP16_WORKER(Q, RNEXT)                                                                                           \
    if (b4 + (Q) < nb) {                                                                                                   \
        const int b = b4 + (Q);                                                                                            \
        if (b >= RB && b % RB == 0)      wg_dp_refill_issue<NW>(refill, Wp, Cp, cum0, s0 + (b / RB + 1) * WG_DP_META_REGION, s1, lane, lw); \
        else if (b > RB && b % RB == 1)  wg_dp_refill_commit<NW>(refill, metaW, metaC, (b / RB + 1) * WG_DP_META_REGION, lane, lw); \
        if (b + 1 < nb) wg_dp16_commit<NW>(RNEXT, slots + (size_t)((b + 1) & 1) * (BL * 64), lane, lw);                       \
        if (b + 1 + D < nb) wg_dp16_issue<NW>(RNEXT, cb, metaW, metaC, (b + 1 + D) * BL, s0 + (b + 1 + D) * BL, lane, lw);     \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                                    \
    }
#define WG_DP16_REC(Q)                                                                                                     \
    if (b4 + (Q) < nb) {                                                                                                   \
        const int b = b4 + (Q);                                                                                            \
        const double* my = slots + (size_t)(b & 1) * (BL * 64) + lane;                                                     \
        double ra[8], rb[8];                                                                                               \
        _Pragma("unroll") for (int u = 0; u < 8; u++) { ra[u] = my[u * 64]; rb[u] = my[(8 + u) * 64]; }                    \
        double Ms = wg_readlane_f64(Mk, 0);   /* M[k] is wave-uniform: the asm blocks want it in scalar registers */       \
        wg_dp_group64<16 * (Q), 0>(best, arg, tbk, Ms, ra, ninf_hi);                                                       \
        wg_dp_group64<16 * (Q) + 8, 1>(best, arg, tbk, Ms, rb, ninf_hi);                                                   \
        Mk = Ms;                                                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                                    \
    }
    if (worker) {
        for (int b4 = 0; b4 < nb; b4 += 4) { WG_DP16_WORKER(0, r1) WG_DP16_WORKER(1, r2) WG_DP16_WORKER(2, r3) WG_DP16_WORKER(3, r0) }
    } else {
        for (int b4 = 0; b4 < nb; b4 += 4) {
            WG_DP16_REC(0) WG_DP16_REC(1) WG_DP16_REC(2) WG_DP16_REC(3)
            // the 64 lanes of tbk now hold the source lanes of the steps s0 + 16 b4 + lane (those that ran)
            const int i = s0 + b4 * BL + lane;
            const uint32_t len = (((uint32_t)lane - tbk) & 63u) + 1u;
            if (i < s1) J.back16[cd.site_off + i] = (uint16_t)len;
            tbk = 0;
        }
    }
#undef WG_DP16_WORKER
#undef WG_DP16_REC
    if (!worker && s1 < cd.len) {
        if (lane == 0) gs[0] = Mk;
        gs[1 + lane] = best;
        gs[65 + lane] = __longlong_as_double((long long)arg);
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_trace / k_border_offsets / k_gather_borders
// ------------------------------------------------------------------------------------------------------------
// k_trace: one workgroup per chunk walks T[] back from the chunk's end (segmentor.cpp:50-58) and leaves the borders in
// DESCENDING order in tmp.  The walk is a dependent chain of LDS reads, so every LDS window of back-pointers is cut into
// one segment per thread (segment j = the nodes (lim_j, p_j], p_j = hi - j L):
//   1. thread j walks, speculatively, from the top node of its segment down to the segment's end, marking the nodes it
//      visits and remembering where it left (e_j);
//   2. where does the TRUE path enter segment j?  At the node where it leaves segment j - 1 — and a walk that starts anywhere
//      in a segment joins the speculative walk of that segment within a block or two (each step depends only on the node:
//      from a common node on, two walks are one), after which it leaves at e_j.  So every thread assumes the entry
//      a_j = e_(j-1), follows it until it meets a marked node (exit e_j) or the segment's end (exit = where it left), and
//      the assumptions are checked against the exits of the neighbours above: a thread whose assumption was wrong takes the
//      right entry and walks again.  a_0 = hi is exact, so the first wrong assumption is corrected in every round: the
//      fixed point — one or two rounds, each a few steps long — is the true path.  (Rounds 1-5 had one thread follow the
//      true path through all 128 segments, a chain of ~6 dependent LDS reads per segment: 0.08 of the kernel's 0.18 ms.)
//   3. thread j marks the walk from its true entry as path; the marked nodes are emitted by a bit scan.
// The window arrives as aligned 16-byte loads (the chunks' slices of back16 begin at multiples of 8 entries).
__device__ __forceinline__ int wg_trace_step(const uint16_t* win, int i, int wlo)
{
    int stepb = (int)win[i - 1 - wlo];                       // i = T[i] (segmentor.cpp:55)
    if (stepb < 1 || stepb > i) stepb = i;                   // never loops on a corrupt back-pointer (would show up as a parity failure)
    return i - stepb;
}

__global__ __launch_bounds__(WG_BLOCK) void k_trace(JobView J, int32_t* __restrict__ tmp, int32_t* __restrict__ nb)
{
    __shared__ __attribute__((aligned(16))) uint16_t win_raw[WG_TRACE_WIN + 8];      // back-pointer of node i (wlo < i <= hi) at win[i-1-wlo], win = win_raw + (wlo mod 8)
    __shared__ uint32_t vis[WG_TRACE_WIN / 32 + 1];          // bit i-wlo: node visited by the speculative walk of its own segment
    __shared__ uint32_t tp[WG_TRACE_WIN / 32 + 1];           // bit i-wlo: node is on the path
    __shared__ int seg_exit[WG_BLOCK], seg_r[WG_BLOCK];
    __shared__ uint32_t wsum[WG_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c = blockIdx.x;
    const ChunkDesc cd = J.chunks[c];
    int32_t* out = tmp + cd.site_off + c;                    // len+1 slots
    const uint16_t* bk = J.back16 + cd.site_off;             // (16-byte aligned: site_off is a multiple of 8)
    int hi = cd.len, cnt = 0;
    while (hi > 0) {
        const int wlo = (hi > WG_TRACE_WIN) ? hi - WG_TRACE_WIN : 0;
        const int nw = (hi - wlo) / 32 + 1;                   // bitmap words in use
        // nodes per segment: a thread each for a full window (128 nodes), never fewer than 64 nodes — a block longer than a segment makes the
        // path jump over segments, and every jumped segment costs the fixed point a round (a 400-site junction patch cut into 2-node
        // segments took 200 rounds)
        const int L0 = (hi - wlo + WG_BLOCK - 1) / WG_BLOCK;
        const int L = L0 < 64 ? 64 : L0;
        const int S = (hi - wlo + L - 1) / L;                 // segments in use (<= WG_BLOCK)
        const int shift = wlo & 7;
        {
            const uint4* src = reinterpret_cast<const uint4*>(bk + (wlo - shift));
            const int nv = (shift + hi - wlo + 7) >> 3;       // (the last vector may reach into the padding behind the chunk's slice)
            for (int x = tid; x < nv; x += WG_BLOCK) reinterpret_cast<uint4*>(win_raw)[x] = src[x];
        }
        const uint16_t* win = win_raw + shift;
        for (int x = tid; x < nw; x += WG_BLOCK) { vis[x] = 0u; tp[x] = 0u; }
        __syncthreads();
        const int p = hi - tid * L;                           // top node of this thread's segment
        const int lim = (p - L > wlo) ? p - L : wlo;
        const bool mine = tid < S;
        if (mine) {
            int i = p;
            while (i > lim) { atomicOr(&vis[(i - wlo) >> 5], 1u << ((i - wlo) & 31)); i = wg_trace_step(win, i, wlo); }
            seg_exit[tid] = i;
        }
        __syncthreads();
        // the fixed point of (entry of segment j) = (exit of segment j - 1 for ITS entry)
        int a = mine ? (tid == 0 ? hi : seg_exit[tid - 1]) : 0;
        bool need = mine;
        for (;;) {
            if (need) {
                int i = a;
                while (i > lim) {
                    if ((vis[(i - wlo) >> 5] >> ((i - wlo) & 31)) & 1u) { i = seg_exit[tid]; break; }
                    i = wg_trace_step(win, i, wlo);
                }
                seg_r[tid] = i;
            }
            __syncthreads();
            need = false;
            if (mine && tid > 0) { const int ra = seg_r[tid - 1]; if (ra != a) { a = ra; need = true; } }
            if (!__syncthreads_or(need ? 1 : 0)) break;
        }
        if (mine) {
            int i = a;
            while (i > lim) { atomicOr(&tp[(i - wlo) >> 5], 1u << ((i - wlo) & 31)); i = wg_trace_step(win, i, wlo); }
        }
        const int next_hi = seg_r[S - 1];                     // where the path leaves the window
        __syncthreads();
        // emit the window's path nodes, largest first: thread t owns the words nw-1-t*wpt .. (descending)
        const int wpt = (nw + WG_BLOCK - 1) / WG_BLOCK;
        uint32_t mine_n = 0;
        for (int q = 0; q < wpt; q++) { const int w = nw - 1 - (tid * wpt + q); if (w >= 0) mine_n += (uint32_t)__popc(tp[w]); }
        const uint32_t incl = wg_wave_incl_scan_dpp_u32(mine_n);
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        uint32_t off = incl - mine_n, tot = 0;
#pragma unroll
        for (int q = 0; q < WG_BLOCK / 64; q++) { if (q < wv) off += wsum[q]; tot += wsum[q]; }
        int pos = cnt + (int)off;
        for (int q = 0; q < wpt; q++) {
            const int w = nw - 1 - (tid * wpt + q);
            if (w < 0) break;
            uint32_t bits = tp[w];
            while (bits) { const int b = 31 - __clz((int)bits); out[pos++] = wlo + w * 32 + b; bits &= ~(1u << b); }
        }
        cnt += (int)tot;
        hi = next_hi;
        __syncthreads();
    }
    if (tid == 0) { out[cnt] = 0; nb[c] = cnt + 1; }          // the walk ends at border 0 (segmentor.cpp:54-57)
}

__global__ __launch_bounds__(WG_BLOCK) void k_border_offsets(const int32_t* __restrict__ nb, int n_chunks, int64_t* __restrict__ boff)
{
    __shared__ uint64_t ws[WG_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint64_t run = 0;
    for (int base = 0; base < n_chunks; base += WG_BLOCK) {
        const int c = base + tid;
        const uint64_t v = (c < n_chunks) ? (uint64_t)nb[c] : 0;
        const uint64_t incl = wg_wave_incl_scan_u64(v, lane);
        if (lane == 63) ws[wv] = incl;
        __syncthreads();
        uint64_t o = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < WG_BLOCK / 64; q++) { if (q < wv) o += ws[q]; tot += ws[q]; }
        __syncthreads();
        if (c < n_chunks) boff[c] = (int64_t)(run + o + incl - v);
        run += tot;
    }
    if (tid == 0) boff[n_chunks] = (int64_t)run;
}

__global__ __launch_bounds__(WG_BLOCK) void k_gather_borders(JobView J, const int32_t* __restrict__ tmp, const int32_t* __restrict__ nb,
                                                             const int64_t* __restrict__ boff, int32_t* __restrict__ out)
{
    const int c = blockIdx.x;
    const ChunkDesc cd = J.chunks[c];
    const int32_t* src = tmp + cd.site_off + c;
    const int n = nb[c];
    int32_t* dst = out + boff[c];
    for (int x = threadIdx.x; x < n; x += WG_BLOCK) dst[x] = src[n - 1 - x];     // ascending (segmentor.cpp:30-34)
}

// k_gather_edges / k_copy_out (round 6): a whole-genome batch brings 11 MB of borders home (0.24 ms of PCIe) while what the host
// needs FIRST — to find out which junctions still lack a patch — lies within a few dozen borders of every chunk's two ends.
// k_gather_edges leaves the first and the last WG_EDGE borders of the leading `n_lead` items (the chunks) in a compact array
// [item][front | back][WG_EDGE], the back right-aligned; it goes home ahead of the lists themselves, which k_copy_out then writes
// straight into the page-locked result buffer (stores over the link, no copy engine: the small copies of the follow-up batch that
// runs meanwhile do not queue behind it).
#define WG_EDGE 128
__global__ __launch_bounds__(WG_BLOCK) void k_gather_edges(const int64_t* __restrict__ boff, const int32_t* __restrict__ bord, int n_lead, int32_t* __restrict__ edges,
                                                           int32_t* __restrict__ rest)      // rest: the lists of the items from n_lead on, back to back (they ride home with the edges)
{
    const int c = blockIdx.x;
    const int64_t b0 = boff[c];
    const int n = (int)(boff[c + 1] - b0);
    if (c >= n_lead) {
        int32_t* dst = rest + (b0 - boff[n_lead]);
        for (int x = threadIdx.x; x < n; x += WG_BLOCK) dst[x] = bord[b0 + x];
        return;
    }
    const int m = n < WG_EDGE ? n : WG_EDGE;
    const int x = threadIdx.x;                                  // 0 .. 255: front | back
    const int side = x >> 7, q = x & (WG_EDGE - 1);
    int32_t* dst = edges + ((int64_t)c * 2 + side) * WG_EDGE;
    if (side == 0) { if (q < m) dst[q] = bord[b0 + q]; }
    else if (q >= WG_EDGE - m) dst[q] = bord[b0 + n - WG_EDGE + q];
}

__global__ __launch_bounds__(WG_BLOCK) void k_copy_out(const int32_t* __restrict__ src, int32_t* __restrict__ dst, int64_t n)      // both 16-byte aligned
{
    const int64_t nv = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * WG_BLOCK;
    for (int64_t x = (int64_t)blockIdx.x * WG_BLOCK + threadIdx.x; x < nv; x += stride)
        reinterpret_cast<int4*>(dst)[x] = reinterpret_cast<const int4*>(src)[x];
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[4 * nv + threadIdx.x] = src[4 * nv + threadIdx.x];
}

// ------------------------------------------------------------------------------------------------------------
// k_block_sums: (#meth, #cov) of every block of a blocks table in every sample — the reduction of
// beta_to_blocks.py:101-126 (np.add.reduceat over the sample's (meth, cov) rows / the per-row slice sums of its
// slow_method), with its output conversions:
//   mode 0  uint32 pairs (the sums)
//   mode 1  uint8  pairs (.bin):   utils_wgbs.py:277-290 trim_to_uint8: cov > 255   -> meth = trunc(meth / cov * 255), cov = 255
//   mode 2  uint16 pairs (.lbeta): the same with 65535
//   mode 3  double meth/cov, NaN where cov < min_cov (utils_wgbs.py:270-274 beta2vec)
// HBM-bound: a blocks table that tiles the genome reads every beta byte once (2 N n bytes; 4 N n for uint16 .lbeta input).
//
// The blocks arrive SORTED by their first site (the host sorts when the table is not; `perm` then names the row a block
// came from).  One workgroup takes one tile of WG_BS_TILE consecutive sites and 4 x spw samples: every wavefront streams
// its sample's bytes of the tile ONCE with 16-byte loads (coalesced: 1 KB per wave instruction), leaves the tile's
// exclusive prefix sums of (meth, cov) in its own LDS row (two DPP scans per 512 sites), and then every lane takes blocks
// that START in the tile: a block inside the tile is one subtraction of two LDS entries; the part of a block beyond
// the tile's end (a few per cent of the blocks of a segmentation) is summed from memory by the lane.  No byte is fetched
// twice for a table that tiles the genome, whatever the block lengths; a thread per block (the first version of this
// kernel) fetched 2-3 aligned vectors per 20-byte block.
// ------------------------------------------------------------------------------------------------------------
#define WG_BS_TILE 896         // tile stride; WG_BS_EXT = 1024 sites are staged per tile (one wavefront pass of 16 sites per lane for uint8 rows)

// sums of sites [a, b) of one sample row straight from memory (tails of blocks that leave their tile): 16-byte vectors
template <int ELEM>
__device__ __forceinline__ void wg_direct_sum(const uint8_t* __restrict__ row, int64_t a, int64_t b, int64_t n_total, uint64_t& m, uint64_t& c)
{
    constexpr int SPV = ELEM == 1 ? 8 : 4;                         // sites per 16-byte vector
    for (int64_t v0 = a & ~(int64_t)(SPV - 1); v0 < b; v0 += SPV) {
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        if (v0 + SPV <= n_total) {
            const uint4 v = *reinterpret_cast<const uint4*>(row + (size_t)v0 * 2 * ELEM);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
            for (int j = 0; j < SPV; j++) if (v0 + j < n_total) {
                if (ELEM == 1) w[j >> 1] |= ((uint32_t)row[2 * (v0 + j)] | ((uint32_t)row[2 * (v0 + j) + 1] << 8)) << (16 * (j & 1));
                else w[j] = (uint32_t)reinterpret_cast<const uint16_t*>(row)[2 * (v0 + j)] | ((uint32_t)reinterpret_cast<const uint16_t*>(row)[2 * (v0 + j) + 1] << 16);
            }
        }
#pragma unroll
        for (int j = 0; j < SPV; j++) {
            const int64_t x = v0 + j;
            if (x < a || x >= b) continue;
            if (ELEM == 1) { const uint32_t h = w[j >> 1] >> (16 * (j & 1)); m += h & 0xffu; c += (h >> 8) & 0xffu; }
            else { m += w[j] & 0xffffu; c += w[j] >> 16; }
        }
    }
}

__device__ __forceinline__ void wg_block_sum_store(void* __restrict__ out, int64_t o, int mode, uint32_t min_cov, uint64_t m, uint64_t c)
{
    if (mode == 0) {
        reinterpret_cast<uint2*>(out)[o] = make_uint2((uint32_t)m, (uint32_t)c);
    } else if (mode == 1 || mode == 2) {
        const uint64_t maxv = mode == 1 ? 255u : 65535u;
        if (c > maxv) { m = (uint64_t)((double)m / (double)c * (double)maxv); c = maxv; }
        if (mode == 1) reinterpret_cast<uchar2*>(out)[o] = make_uchar2((unsigned char)m, (unsigned char)c);
        else           reinterpret_cast<ushort2*>(out)[o] = make_ushort2((unsigned short)m, (unsigned short)c);
    } else {
        reinterpret_cast<double*>(out)[o] = (c >= (uint64_t)min_cov) ? (double)m / (double)c : __builtin_nan("");
    }
}

template <int ELEM>
__device__ __noinline__ uint4 wg_bs_load_tail(const uint8_t* __restrict__ row, int64_t site, int64_t n_total)
{
    // the last vector of a row: sites beyond n_total read as zero (rare: once per row)
    constexpr int SPL = ELEM == 1 ? 8 : 4;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    for (int j = 0; j < SPL; j++) if (site + j < n_total) {
        if (ELEM == 1) w[j >> 1] |= ((uint32_t)row[2 * (site + j)] | ((uint32_t)row[2 * (site + j) + 1] << 8)) << (16 * (j & 1));
        else w[j] = (uint32_t)reinterpret_cast<const uint16_t*>(row)[2 * (site + j)] | ((uint32_t)reinterpret_cast<const uint16_t*>(row)[2 * (site + j) + 1] << 16);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

#define WG_BS_RUN 8            // consecutive tiles streamed by one workgroup
#define WG_BS_EXT 1024         // sites staged per tile = WG_BS_TILE + 128: a block that starts in the tile and ends within 128 sites
                               // of its end (almost every tile's last block) is still served from LDS; no global load sits in the
                               // reduction of the common case, where it would have to wait for the NEXT tile's prefetch as well

template <int ELEM>
struct BsTile {                // one tile's inputs in registers: the sample bytes (32 per lane and pass), and the first 128 block descriptors
    static constexpr int SPL = ELEM == 1 ? 16 : 8;             // sites per lane and pass (two 16-byte vectors)
    static constexpr int SPP = 64 * SPL;                       // sites per pass of the wavefront
    static constexpr int NPASS = WG_BS_EXT / SPP;              // uint8 rows: ONE pass stages the tile; uint16 rows: two
    static constexpr int MAXQ = ELEM == 1 ? 2 : 1;             // samples per wave
    uint4 v[MAXQ][NPASS][2];
    int32_t b0, b1, xa0, xa1, xb0, xb1, ra, rb;
};

template <int ELEM>           // bytes per count: 1 = .beta / .bin (uint8 pairs), 2 = .lbeta (uint16 pairs)
__global__ __launch_bounds__(WG_BLOCK) void k_block_sums(const uint8_t* __restrict__ betas, int64_t pitch, int64_t n_total,
                                                         const int32_t* __restrict__ x0s, const int32_t* __restrict__ x1s,
                                                         const int32_t* __restrict__ perm, const int32_t* __restrict__ tile_first,
                                                         int64_t n_tiles, int64_t n_blocks, int n_samples, int spw, int mode, uint32_t min_cov,
                                                         void* __restrict__ out)
{
    // per wave: exclusive prefixes of the staged sites of the sample it is working on.  Waves never touch each other's row and
    // a wave's LDS instructions execute in program order, so no workgroup barrier is needed anywhere in this kernel.
    // uint8 rows: a lane's 16 sites are summed IN the lane as packed pairs (meth | cov << 16: 16 x 255 fits 16 bits, one add
    // per site for both counts) and stored as such (PK), next to the lane's own base (BASE, from two wave scans per tile): a
    // prefix is BASE[x >> 4] + unpack(PK[x]).  uint16 rows (.lbeta) keep full 32-bit pairs per site (E).
    constexpr int ROW_BYTES = ELEM == 1 ? (WG_BS_EXT + 16) * 4 + (WG_BS_EXT / 16 + 2) * 8 : (WG_BS_EXT + 8) * 8;
    __shared__ __attribute__((aligned(16))) char lds[WG_BLOCK / 64][ROW_BYTES];
    typedef BsTile<ELEM> T;
    constexpr int SPL = T::SPL, SPP = T::SPP, NPASS = T::NPASS, MAXQ = T::MAXQ;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint2* Ew = reinterpret_cast<uint2*>(lds[wv]);                                        // (uint16 rows) [EXT + 1]
    uint32_t* PK = reinterpret_cast<uint32_t*>(lds[wv]);                                  // (uint8 rows) [EXT + 1]
    uint2* BASE = reinterpret_cast<uint2*>(lds[wv] + (WG_BS_EXT + 16) * 4);               // (uint8 rows) [EXT / 16 + 1]
    const int s_first = ((int)blockIdx.y * (WG_BLOCK / 64) + wv) * spw;
    if (s_first >= n_samples) return;
    const int64_t t_first = (int64_t)blockIdx.x * WG_BS_RUN;
    const size_t esz = mode == 0 ? 8 : (mode == 1 ? 2 : (mode == 2 ? 4 : 8));

    // everything a tile needs, requested in one go (nothing is waited for here)
    auto issue = [&](T& R, int64_t tile) {
        R.b0 = R.b1 = 0;
        if (tile >= n_tiles) return;
        R.b0 = tile_first[tile]; R.b1 = tile_first[tile + 1];
        if (R.b0 == R.b1) return;                                  // no block starts in this tile: nothing to read
        const int64_t lo = tile * WG_BS_TILE;
        const int64_t hi = lo + WG_BS_EXT < n_total ? lo + WG_BS_EXT : n_total;          // staged sites [lo, hi)
#pragma unroll
        for (int q = 0; q < MAXQ; q++) {
            const int s = s_first + q;
            const uint8_t* row = betas + (int64_t)(s < n_samples ? s : 0) * pitch;
#pragma unroll
            for (int p = 0; p < NPASS; p++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int64_t site = lo + p * SPP + (int64_t)lane * SPL + h * (SPL / 2);
                    R.v[q][p][h] = make_uint4(0u, 0u, 0u, 0u);
                    if (q < spw && s < n_samples && site < hi) {
                        if (site + SPL / 2 <= n_total) R.v[q][p][h] = *reinterpret_cast<const uint4*>(row + (size_t)site * 2 * ELEM);
                        else R.v[q][p][h] = wg_bs_load_tail<ELEM>(row, site, n_total);
                    }
                }
        }
        const int ba = R.b0 + lane, bb = R.b0 + 64 + lane;
        R.xa0 = ba < R.b1 ? x0s[ba] : 0; R.xa1 = ba < R.b1 ? x1s[ba] : 0;
        R.xb0 = bb < R.b1 ? x0s[bb] : 0; R.xb1 = bb < R.b1 ? x1s[bb] : 0;
        R.ra = ba < R.b1 ? (perm ? perm[ba] : ba) : 0; R.rb = bb < R.b1 ? (perm ? perm[bb] : bb) : 0;
    };
    auto compute = [&](const T& R, int64_t tile) {
        if (R.b0 == R.b1) return;                                  // wave-uniform
        const int lo = (int)(tile * WG_BS_TILE);
        const int hi = (int64_t)lo + WG_BS_EXT < n_total ? lo + WG_BS_EXT : (int)n_total;   // staged sites [lo, hi)
        const int b0 = R.b0, b1 = R.b1;
#pragma unroll
        for (int q = 0; q < MAXQ; q++) {
            const int s = s_first + q;
            if (q >= spw || s >= n_samples) break;                 // wave-uniform
            const uint8_t* row = betas + (int64_t)s * pitch;
            char* orow = reinterpret_cast<char*>(out) + (size_t)s * (size_t)n_blocks * esz;
            uint32_t run_m = 0, run_c = 0;
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                const uint32_t w[8] = {R.v[q][p][0].x, R.v[q][p][0].y, R.v[q][p][0].z, R.v[q][p][0].w,
                                       R.v[q][p][1].x, R.v[q][p][1].y, R.v[q][p][1].z, R.v[q][p][1].w};
                if (ELEM == 1) {
                    uint32_t e[16], acc = 0;                       // packed exclusive prefixes inside the lane
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        e[j] = acc;
                        // site j as (meth | cov << 16): bytes (m, 0, c, 0) picked out of the dword that holds two sites
                        acc += __builtin_amdgcn_perm(0u, w[j >> 1], (j & 1) ? 0x0c030c02u : 0x0c010c00u);
                    }
                    const uint32_t tm = acc & 0xffffu, tc = acc >> 16;
                    const uint32_t im = wg_wave_incl_scan_dpp_u32(tm), ic = wg_wave_incl_scan_dpp_u32(tc);
                    BASE[lane] = make_uint2(im - tm, ic - tc);
                    uint4* dst = reinterpret_cast<uint4*>(PK + lane * 16);
#pragma unroll
                    for (int j = 0; j < 16; j += 4) dst[j >> 2] = make_uint4(e[j], e[j + 1], e[j + 2], e[j + 3]);
                    run_m = (uint32_t)__builtin_amdgcn_readlane((int)im, 63);
                    run_c = (uint32_t)__builtin_amdgcn_readlane((int)ic, 63);
                } else {
                    uint32_t m[SPL], c[SPL], tm = 0, tc = 0;
#pragma unroll
                    for (int j = 0; j < SPL; j++) { m[j] = w[j] & 0xffffu; c[j] = w[j] >> 16; tm += m[j]; tc += c[j]; }
                    const uint32_t im = wg_wave_incl_scan_dpp_u32(tm), ic = wg_wave_incl_scan_dpp_u32(tc);
                    uint32_t em = run_m + (im - tm), ec = run_c + (ic - tc);
                    uint2 e[SPL];
#pragma unroll
                    for (int j = 0; j < SPL; j++) { e[j] = make_uint2(em, ec); em += m[j]; ec += c[j]; }
                    uint4* dst = reinterpret_cast<uint4*>(Ew + p * SPP + lane * SPL);          // 16-byte stores, lane-contiguous
#pragma unroll
                    for (int j = 0; j < SPL; j += 2) dst[j >> 1] = make_uint4(e[j].x, e[j].y, e[j + 1].x, e[j + 1].y);
                    run_m += (uint32_t)__builtin_amdgcn_readlane((int)im, 63);
                    run_c += (uint32_t)__builtin_amdgcn_readlane((int)ic, 63);
                }
            }
            if (lane == 0) {                                       // the entry behind the last staged site (sites past `hi` were read as zeros)
                if (ELEM == 1) { BASE[WG_BS_EXT / 16] = make_uint2(run_m, run_c); PK[WG_BS_EXT] = 0u; }
                else Ew[WG_BS_EXT] = make_uint2(run_m, run_c);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            auto prefix = [&](int x) -> uint2 {                    // sums of the staged sites before x
                if (ELEM == 1) { const uint2 b = BASE[x >> 4]; const uint32_t k = PK[x]; return make_uint2(b.x + (k & 0xffffu), b.y + (k >> 16)); }
                return Ew[x];
            };
            auto one = [&](int x0, int x1, int r) {
                uint32_t m32 = 0, c32 = 0;
                if (x1 > x0) {
                    const int e = x1 < hi ? x1 : hi;
                    const uint2 pe = prefix(e - lo), ps = prefix(x0 - lo);
                    m32 = pe.x - ps.x; c32 = pe.y - ps.y;
                }
                if (x1 > hi) {                                     // rare: a block reaching beyond the staged sites
                    uint64_t m = m32, c = c32;
                    wg_direct_sum<ELEM>(row, hi, x1, n_total, m, c);
                    wg_block_sum_store(orow, r, mode, min_cov, m, c);
                } else {
                    wg_block_sum_store(orow, r, mode, min_cov, (uint64_t)m32, (uint64_t)c32);
                }
            };
            if (b0 + lane < b1) one(R.xa0, R.xa1, R.ra);
            if (b0 + 64 + lane < b1) one(R.xb0, R.xb1, R.rb);
            for (int b = b0 + 128 + lane; b < b1; b += 64) one(x0s[b], x1s[b], perm ? perm[b] : b);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();                                                       // reads done before the next row overwrites
        }
    };
    // stream the run of tiles: the next tile's bytes are in flight while this one is reduced
    T A, B;
    issue(A, t_first);
#pragma unroll 1
    for (int k = 0; k < WG_BS_RUN; k += 2) {
        issue(B, t_first + k + 1);
        compute(A, t_first + k);
        issue(A, t_first + k + 2 < t_first + WG_BS_RUN ? t_first + k + 2 : n_tiles);
        compute(B, t_first + k + 1);
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_block_sums_run: the block reduction for uint8 rows and a table whose blocks are ordered by first AND by last site (every
// table a segmentation wrote; every "nice" table of beta_to_blocks.py:45-60) — the common case, streamed:
// one wavefront takes one sample and a RUN of consecutive 1024-site tiles and walks them in order, keeping running totals,
// so that the prefixes it leaves in LDS (two tiles: a ring) are prefixes of the whole run; after staging tile t it resolves
// the blocks whose LAST site lies in tile t: their first site is in tile t or t-1 (blocks up to 1024 sites), so a block is
// one subtraction of two prefixes whatever tile boundary it crosses.  Every byte is read once (no halo), two tiles are in
// flight while one is reduced, one sample per wavefront keeps it at 60-odd VGPRs.  Blocks that begin before the run or more
// than a tile back are summed from memory by their lane (one or two per run; blocks longer than 1024 sites).
// The general kernel above stays for .lbeta rows and for tables with nested / overlapping blocks.
// ------------------------------------------------------------------------------------------------------------
#define WG_BSR_SPL 16                          // sites per lane of a tile: two 16-byte vectors.  (8 = 512-site tiles, runs of 16: 68 VGPRs and 20 KB of
                                               // LDS per workgroup, i.e. 28 instead of 16 resident wavefronts per CU — measured 0.467 vs 0.434-0.455 ms: the
                                               // pass is instruction-bound, and the per-tile work is spread over half the sites)
#define WG_BSR_TILE (64 * WG_BSR_SPL)
#define WG_BSR_RUN 8
#define WG_BSR_PK (WG_BSR_TILE + 16)          // packed in-lane prefixes of a tile (+ the entry behind its last site)

// trunc(fl(fl(m / c) * 255)) — utils_wgbs.py:277-290 trim_to_uint8, float64 division, product, truncation — for integers 0 <= m <= c, 255 < c <= 65535 * 255.
// Round 5: it IS floor(255 m / c), always, so it is computed in integers (10 VALU instructions, none of them fp64; rounds 2-4: the reference's own three
// float64 operations, 16 instructions with v_rcp_f64 and the v_div_* family, which nearly every wavefront of the block reduction executes).  Why:
//   * 255 m / c not an integer: it lies at least 1 / c >= 6e-8 from the nearest integer, the two roundings move the product by less than 255 * 2^-52 = 6e-14;
//   * 255 m / c = K an integer: then m / c = K / 255 as a real number, so fl(m / c) = fl(K / 255) whatever m and c are, and trunc(fl(fl(K / 255) * 255)) = K for
//     every K in 0 .. 255 (256 cases: tests/test_blocks_cpu.py::test_trim_rescale_is_an_integer_division walks them, and 65535 likewise).
// The division: q = floor of a float estimate of N / c, N = 255 m < 2^32, pushed DOWN by 1e-4 — Nf and the product carry 2^-24 each, v_rcp_f32 one ulp
// (2^-23): the estimate is within 6.1e-5 of N / c <= 255, so the biased one lies in (N / c - 1.7e-4, N / c) and its floor is K or K - 1 (0 when negative:
// v_cvt_u32_f32 clamps) — then one exact correction from the remainder.  c < 2^24 is exact in float; q * c <= N needs no wider type.
__device__ __forceinline__ uint32_t wg_rescale_255(uint32_t m, uint32_t c)
{
    const uint32_t N = __umul24(m, 255u);            // m < 2^24: the low 32 bits of the 48-bit product are the product
    const float est = __builtin_fmaf((float)N, __builtin_amdgcn_rcpf((float)c), -1.0e-4f);
    uint32_t q = (uint32_t)est;                                           // (negative -> 0)
    const uint32_t r = N - __umul24(q, c);
    return q + (r >= c ? 1u : 0u);
}

// k_block_sums_direct: the blocks the streaming kernel leaves out — those that begin before their run or more than a tile
// before the tile they end in (blocks longer than 1024 sites, and one or two per run boundary): one wavefront per (block,
// sample) reads the block's bytes with 16-byte loads, 512 sites per iteration, and reduces across lanes.  Which blocks these are
// depends on the table alone: the host lists them.
__global__ __launch_bounds__(WG_BLOCK) void k_block_sums_direct(const uint8_t* __restrict__ betas, int64_t pitch, int64_t n_total,
                                                                const int32_t* __restrict__ x0s, const int32_t* __restrict__ x1s,
                                                                const int32_t* __restrict__ perm, const int32_t* __restrict__ list, int64_t n_list,
                                                                int64_t n_blocks, int n_samples, int mode, uint32_t min_cov, void* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int s = (int)blockIdx.y * (WG_BLOCK / 64) + wv;
    if (s >= n_samples || (int64_t)blockIdx.x >= n_list) return;
    const int b = list[blockIdx.x];
    const int x0 = x0s[b], x1 = x1s[b];
    const uint8_t* row = betas + (int64_t)s * pitch;
    const int64_t last_vec = ((n_total + 7) >> 3) - 1;
    unsigned long long m = 0, c = 0;
    for (int64_t v0 = (int64_t)(x0 & ~7) + 8 * lane; v0 < x1; v0 += 512) {
        const int64_t vi = v0 >> 3;
        const uint4 v = reinterpret_cast<const uint4*>(row)[vi < last_vec ? vi : last_vec];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int64_t x = v0 + j;
            if (x < x0 || x >= x1) continue;
            const uint32_t hh = w[j >> 1] >> (16 * (j & 1));
            m += hh & 0xffu; c += (hh >> 8) & 0xffu;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { m += __shfl_down(m, o); c += __shfl_down(c, o); }
    if (lane == 0) {
        const size_t esz = mode == 0 ? 8 : (mode == 1 ? 2 : (mode == 2 ? 4 : 8));
        char* orow = reinterpret_cast<char*>(out) + (size_t)s * (size_t)n_blocks * esz;
        wg_block_sum_store(orow, perm ? perm[b] : b, mode, min_cov, (uint64_t)m, (uint64_t)c);
    }
}

#ifndef WG_BSR_PRE
#define WG_BSR_PRE 3                           // rounds of 64 block descriptors fetched a tile ahead (a tile of a segmentation ends ~100 blocks)
#endif
#define WG_BSR_RING (2 * WG_BSR_TILE)
#ifndef WG_BSR_AHEAD
#define WG_BSR_AHEAD 4                         // tiles of sample bytes in flight per wavefront (2 KB each)
#endif

// LDS layout of the prefix ring: ring position q = (half, site x of the tile); the four sites 4 g .. 4 g + 3 of lane L (x = 16 L +
// 4 g + k) sit at dwords (g * 64 + L) * 4 + k of their half: consecutive lanes write consecutive 16-byte slots (no bank
// conflicts; lane-major rows of 16 dwords are 16-way conflicts).
__host__ __device__ __forceinline__ uint32_t wg_bsr_pk_at(uint32_t q)
{
    // x = SPL L + 4 g + k  ->  (g * 64 + L) * 4 + k
    constexpr uint32_t GM = (WG_BSR_SPL / 4 - 1) << 2;           // the bits of g in x
    constexpr int LS = WG_BSR_SPL == 16 ? 4 : 3;                  // log2(SPL)
    return (q & ~(uint32_t)(WG_BSR_TILE - 1)) | ((q & GM) << 6) | (((q & (WG_BSR_TILE - 1)) >> LS) << 2) | (q & 3u);
}

// k_block_sums_prep: what the streaming kernel needs to know about a block depends on the table alone, so it is worked out once
// per call, not once per sample: the LDS addresses of the two prefixes whose difference is the block's sum — INCLUSIVE
// prefixes I(x1 - 1) - I(x0 - 1): the last site x1 - 1 lies in the block's tile, x0 - 1 in it or in the previous one, an
// empty block reads one entry twice, and the position before the run's first site is an entry kept at zero — and the row
// the result goes to (-1: a block the ring cannot serve, k_block_sums_direct's).
//   d1[b] / d0[b] = byte offset of the PK entry | byte offset of the BASE entry << 16
__global__ __launch_bounds__(WG_BLOCK) void k_block_sums_prep(const int32_t* __restrict__ x0s, const int32_t* __restrict__ x1s, const int32_t* __restrict__ perm,
                                                              int64_t n_blocks, int64_t n_tiles, int32_t* __restrict__ d1, int32_t* __restrict__ d0, int32_t* __restrict__ rr)
{
    const int64_t b = (int64_t)blockIdx.x * WG_BLOCK + threadIdx.x;
    if (b >= n_blocks) return;
    const int x0 = x0s[b], x1 = x1s[b];
    int64_t t = (int64_t)(x1 - 1 > x0 ? x1 - 1 : x0) / WG_BSR_TILE;        // the tile the block is resolved in (k_block_sums_run's table uses the same rule)
    if (t > n_tiles - 1) t = n_tiles - 1;                                   // (an empty block at the very end of the row)
    const int i = (int)(t % WG_BSR_RUN), h = i & 1;
    const int lo = (int)(t * WG_BSR_TILE);
    const uint32_t q1 = (uint32_t)(h * WG_BSR_TILE + (x1 - 1 - lo)) & (WG_BSR_RING - 1), q0 = (uint32_t)(h * WG_BSR_TILE + (x0 - 1 - lo)) & (WG_BSR_RING - 1);
    const bool reach = x1 <= x0 || (i == 0 ? x0 >= lo : x0 >= lo - (WG_BSR_TILE - 1));
    d1[b] = (int32_t)((wg_bsr_pk_at(q1) * 4u) | (((q1 / WG_BSR_SPL) * 8u) << 16));
    d0[b] = (int32_t)((wg_bsr_pk_at(q0) * 4u) | (((q0 / WG_BSR_SPL) * 8u) << 16));
    rr[b] = reach ? (perm ? perm[b] : (int32_t)b) : -1;
}

// Straight-line code on purpose: the tile loop is unrolled (register sets rotate by renaming, not by moves), everything a
// wavefront shares is forced into scalar registers (its sample's row, its LDS rows, the run's tile table), loads that may
// fall outside are clamped instead of predicated, no calls and no per-block address arithmetic (k_block_sums_prep), and the
// output mode is a template parameter.
template <int MODE>
__global__ __launch_bounds__(WG_BLOCK) void k_block_sums_run(const uint8_t* __restrict__ betas, int64_t pitch, int64_t n_total,
                                                             const int32_t* __restrict__ d1s, const int32_t* __restrict__ d0s,
                                                             const int32_t* __restrict__ rrs, const int32_t* __restrict__ end_first,
                                                             int64_t n_tiles, int64_t n_blocks, int n_samples, uint32_t min_cov,
                                                             void* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) uint32_t PKs[WG_BLOCK / 64][WG_BSR_RING];      // packed (meth | cov << 16) INCLUSIVE prefixes inside a lane's 16 sites
    __shared__ uint2 BASEs[WG_BLOCK / 64][WG_BSR_RING / WG_BSR_SPL];                        // the run's totals before each lane's sites
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int s = (int)blockIdx.y * (WG_BLOCK / 64) + wv;
    if (s >= n_samples) return;
    const int64_t t0 = (int64_t)blockIdx.x * WG_BSR_RUN;
    const int nt = (int)((t0 + WG_BSR_RUN < n_tiles ? t0 + WG_BSR_RUN : n_tiles) - t0);       // tiles of this run
    // the run's slice of the tile table in one register (lane i: first block resolved in tile t0 + i or later)
    const int efv = end_first[t0 + (lane <= nt ? lane : nt)];
    if (__builtin_amdgcn_readlane(efv, 0) == __builtin_amdgcn_readlane(efv, nt)) return;     // no block ends in this run: nothing to read
    const uint8_t* row = betas + (int64_t)s * pitch;
    constexpr uint32_t ESZ = MODE == 0 ? 8u : (MODE == 1 ? 2u : (MODE == 2 ? 4u : 8u));
    char* orow = reinterpret_cast<char*>(out) + (size_t)s * (size_t)n_blocks * ESZ;      // (the host keeps n_blocks * 8 below 2^32: 32-bit offsets)
    uint32_t* PK = PKs[wv];
    uint2* BASE = BASEs[wv];
    const char* PKc = reinterpret_cast<const char*>(PK);
    const char* BASEc = reinterpret_cast<const char*>(BASE);
    const uint32_t site0 = (uint32_t)(t0 * WG_BSR_TILE);           // first site of the run (n_total < 2^31)
    const uint32_t last_vec = (uint32_t)(((n_total + 7) >> 3) - 1);                       // last 16-byte vector that holds a site of the row
    const int nb_all = (int)n_blocks;

    // this lane's 16 sites of the run's tile i: two 16-byte vectors.  A vector beyond the row is clamped onto the row's last
    // one (readable: the pitch is a multiple of 16 bytes) — it only feeds prefixes behind every block's last site.  The last
    // vector itself may hold sites beyond n_total: bytes of the row's padding, which no block reaches either.
    auto load = [&](int i, uint4& a, uint4& b) {                   // (a tile behind the run's last: that one again — no branch, nobody uses it)
        constexpr uint32_t VPL = WG_BSR_SPL / 8;                   // 16-byte vectors per lane
        const uint32_t v = ((site0 + (uint32_t)(i < nt ? i : nt - 1) * WG_BSR_TILE) >> 3) + VPL * (uint32_t)lane;
        const uint4* rv = reinterpret_cast<const uint4*>(row);
        a = rv[v < last_vec ? v : last_vec];
        if (VPL > 1) b = rv[v + 1u < last_vec ? v + 1u : last_vec]; else b = a;
    };
    struct Desc { int32_t d1[WG_BSR_PRE], d0[WG_BSR_PRE], r[WG_BSR_PRE]; };
    auto descriptors = [&](int i, Desc& D) {                       // the first 64 x WG_BSR_PRE blocks resolved in tile i of the run
        const int b0 = __builtin_amdgcn_readlane(efv, i), b1 = __builtin_amdgcn_readlane(efv, i + 1);
#pragma unroll
        for (int k = 0; k < WG_BSR_PRE; k++) {
            const int b = b0 + 64 * k + lane;
            const int bq = b < nb_all ? b : nb_all - 1;            // (clamped, not predicated)
            D.d1[k] = d1s[bq];
            D.d0[k] = d0s[bq];
            const int r = rrs[bq];
            D.r[k] = b < b1 ? r : -1;
        }
    };
    uint32_t run_m = 0, run_c = 0;                                 // totals of the run's sites before the current tile
    auto stage = [&](int h, const uint4& c0, const uint4& c1) {
        const uint32_t w[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        uint32_t e[WG_BSR_SPL], acc = 0;                          // packed inclusive prefixes inside the lane (16 x 255 fits 16 bits)
#pragma unroll
        for (int j = 0; j < WG_BSR_SPL; j++) {
            acc += __builtin_amdgcn_perm(0u, w[j >> 1], (j & 1) ? 0x0c030c02u : 0x0c010c00u);     // site j as (meth | cov << 16)
            e[j] = acc;
        }
        const uint32_t tm = acc & 0xffffu, tc = acc >> 16;
        const uint32_t im = wg_wave_incl_scan_dpp_u32(tm), ic = wg_wave_incl_scan_dpp_u32(tc);
        BASE[h * 64 + lane] = make_uint2(run_m + (im - tm), run_c + (ic - tc));
        uint4* dst = reinterpret_cast<uint4*>(PK + h * WG_BSR_TILE) + lane;
#pragma unroll
        for (int j = 0; j < WG_BSR_SPL; j += 4) dst[(j >> 2) * 64] = make_uint4(e[j], e[j + 1], e[j + 2], e[j + 3]);
        run_m += (uint32_t)__builtin_amdgcn_readlane((int)im, 63);
        run_c += (uint32_t)__builtin_amdgcn_readlane((int)ic, 63);
    };
    auto one = [&](int d1, int d0, int r) {
        const uint2 be = *reinterpret_cast<const uint2*>(BASEc + ((uint32_t)d1 >> 16)), bs = *reinterpret_cast<const uint2*>(BASEc + ((uint32_t)d0 >> 16));
        const uint32_t ke = *reinterpret_cast<const uint32_t*>(PKc + ((uint32_t)d1 & 0xffffu)), ks = *reinterpret_cast<const uint32_t*>(PKc + ((uint32_t)d0 & 0xffffu));
        uint32_t m32 = (be.x + (ke & 0xffffu)) - (bs.x + (ks & 0xffffu));
        uint32_t c32 = (be.y + (ke >> 16)) - (bs.y + (ks >> 16));
        const uint32_t o = (uint32_t)r * ESZ;
        if (MODE == 0) {
            *reinterpret_cast<uint2*>(orow + o) = make_uint2(m32, c32);
        } else if (MODE == 3) {
            *reinterpret_cast<double*>(orow + o) = (c32 >= min_cov) ? (double)m32 / (double)c32 : __builtin_nan("");
        } else if (MODE == 1) {
            if (c32 > 255u) {
                // the integer form is proved for 0 <= m <= c (tests/test_blocks_cpu.py); a corrupt file with meth > cov — the block reduction, like
                // the reference's (beta_to_blocks.py:101-126), does not check — takes the float64 form every other kernel of the reduction uses
                // (wg_block_sum_store), so that a table's rows do not depend on which kernel produced them
                m32 = m32 <= c32 ? wg_rescale_255(m32, c32) : (uint32_t)(uint64_t)((double)m32 / (double)c32 * 255.0);
                c32 = 255u;
            }
            *reinterpret_cast<uchar2*>(orow + o) = make_uchar2((unsigned char)m32, (unsigned char)c32);
        } else {
            if (c32 > 65535u) { m32 = (uint32_t)(uint64_t)((double)m32 / (double)c32 * 65535.0); c32 = 65535u; }     // (> 257 saturated sites: rare)
            *reinterpret_cast<ushort2*>(orow + o) = make_ushort2((unsigned short)m32, (unsigned short)c32);
        }
    };
    auto resolve = [&](int i, const Desc& D) {
#pragma unroll
        for (int k = 0; k < WG_BSR_PRE; k++)
            if (D.r[k] >= 0) one(D.d1[k], D.d0[k], D.r[k]);
        const int b1 = __builtin_amdgcn_readlane(efv, i + 1);
        for (int b = __builtin_amdgcn_readlane(efv, i) + 64 * WG_BSR_PRE + lane; b < b1; b += 64) { const int r = rrs[b]; if (r >= 0) one(d1s[b], d0s[b], r); }
    };

    if (lane == 0) { PK[WG_BSR_RING - 1] = 0u; BASE[WG_BSR_RING / WG_BSR_SPL - 1] = make_uint2(0u, 0u); }      // I(-1) of the run: the entry "before" tile 0
    constexpr int AHEAD = WG_BSR_AHEAD;                            // tiles in flight behind the one being staged
    uint4 va[AHEAD + 1], vb[AHEAD + 1];                            // tile i in set i mod (AHEAD + 1)
    Desc D[2];                                                     // descriptors of tile i in set i mod 2
    load(0, va[0], vb[0]);
    descriptors(0, D[0]);
#pragma unroll
    for (int a = 1; a < AHEAD; a++) load(a, va[a], vb[a]);
#pragma unroll
    for (int i = 0; i < WG_BSR_RUN; i++) {
        if (i < nt) {                                              // (wave-uniform)
            // (vector memory operations of a wave complete in order: the descriptors, needed one tile from now, go first, so that
            // waiting for them leaves the bytes of the tiles ahead in flight)
            descriptors(i + 1, D[(i + 1) & 1]);                    // (behind the run's last tile: an empty range)
            load(i + AHEAD, va[(i + AHEAD) % (AHEAD + 1)], vb[(i + AHEAD) % (AHEAD + 1)]);
            stage(i & 1, va[i % (AHEAD + 1)], vb[i % (AHEAD + 1)]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            resolve(i, D[i & 1]);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();                       // (the half written next is the one last read a tile ago)
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_convert: BED regions -> CpG index ranges against the resident loci (the join of `wgbstools convert -L`,
// convert.py:147-185 chr_thread / :133-145 slow_conversion + genomic_region.py:126-161).  One thread per region: two
// binary searches in its chromosome's slice [clo, chi) of the loci.  Region r: bp interval (start, end), chromosome
// slice, chromosome length in bp, and which of the reference's two rule sets applies (it picks per chromosome: the
// as-of joins when the chromosome's regions do not overlap, one GenomicRegion per row when they do).
//   fast: startCpG = first CpG with locus >= start; endCpG = first CpG with locus >= end (+1 when it sits exactly on
//         `end`; the chromosome's last CpG + 1 when there is none)
//   slow: CpGs with start <= locus <= end are rows first..last; endCpG = last + 1, or last when the last locus == end;
//         end <= start, start < 1, end > chromosome length: no answer
// No CpG inside / start beyond the last CpG: (0, 0) = NA.  Indexes are 1-based and global (init_genome.py:151-157).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t wg_lower_bound_loci(const uint32_t* __restrict__ L, int64_t lo, int64_t hi, int64_t x, bool upper)
{
    while (lo < hi) {                                      // first index with L[i] >= x (upper: > x)
        const int64_t mid = (lo + hi) >> 1;
        const int64_t v = (int64_t)L[mid];
        if (upper ? v <= x : v < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(WG_BLOCK) void k_convert(const uint32_t* __restrict__ loci, const int64_t* __restrict__ clo, const int64_t* __restrict__ chi,
                                                      const int64_t* __restrict__ cbp, const int64_t* __restrict__ start, const int64_t* __restrict__ end,
                                                      const uint8_t* __restrict__ slow, int64_t n, int64_t* __restrict__ s_cpg, int64_t* __restrict__ e_cpg)
{
    const int64_t r = (int64_t)blockIdx.x * WG_BLOCK + threadIdx.x;
    if (r >= n) return;
    const int64_t lo = clo[r], hi = chi[r], a = start[r], b = end[r];
    int64_t sc = 0, ec = 0;
    if (hi > lo) {
        if (!slow[r]) {
            const int64_t s = wg_lower_bound_loci(loci, lo, hi, a, false);
            const int64_t j = wg_lower_bound_loci(loci, lo, hi, b, false);
            const int64_t hit = (j < hi && (int64_t)loci[j] == b) ? 1 : 0;
            if (s < hi && j + hit > s) { sc = s + 1; ec = j + 1 + hit; }
        } else if (b > a && a >= 1 && b <= cbp[r]) {
            const int64_t i0 = wg_lower_bound_loci(loci, lo, hi, a, false);
            const int64_t i1 = wg_lower_bound_loci(loci, lo, hi, b, true);
            if (i1 > i0) {
                const int64_t e = i1 + ((int64_t)loci[i1 - 1] < b ? 1 : 0);
                if (e != i0 + 1) { sc = i0 + 1; ec = e; }
            }
        }
    }
    s_cpg[r] = sc; e_cpg[r] = ec;
}

// ------------------------------------------------------------------------------------------------------------
// k_pat_count / k_pat_trim: a pat file -> (#meth, #cov) per CpG, the producer of the path's input (src/pat2beta/
// stdin2beta.cpp:59-93 proc_line, :95-123 parse; utils_wgbs.py:277-290 trim_to_uint8).  A pat line is
//     chr \t first CpG index \t pattern over {C, T, H, .} \t number of reads with that pattern [\t ...]
// Every line adds `count` to the coverage of every site under a C / T / H and to the methylated count under C / H
// (atomics on int32: reads overlap).  Reads that end before `start` or begin at or after `end` are skipped, sites outside
// are ignored, empty lines are skipped (stdin2beta.cpp:75-78,:100).  A line with fewer than four fields or a non-numeric
// site / count makes the reference give up ("failed calculating beta"): its offset is reported through `bad`.
// ------------------------------------------------------------------------------------------------------------
// Round 5: the text is parsed out of LDS, one LINE per thread.  (Rounds 3-4: one thread per BYTE, the thread on a line's first byte parsing
// it alone with byte loads from global memory — 1 lane in ~25 at work, each a chain of dependent loads.)  A workgroup takes WG_PAT_TILE
// bytes of the chunk + WG_PAT_OVER bytes behind them (a line that begins in the tile may end there) into LDS with 16-byte loads, finds
// the line starts of its tile (16 bytes per thread, a workgroup-wide prefix count), and thread l then parses line l: ~160 lines of ~25
// bytes per tile.  A line that runs past the staged bytes (a read of hundreds of CpGs) reads the rest from global memory.
#define WG_PAT_TILE 4096
#define WG_PAT_OVER 1024
struct PatText {
    const char* lds; const char* __restrict__ g; int64_t base, n;          // staged bytes [base, base + WG_PAT_TILE + WG_PAT_OVER) of g[0, n)
    __device__ __forceinline__ char at(int64_t i) const { const int64_t r = i - base; return r < WG_PAT_TILE + WG_PAT_OVER ? lds[r] : g[i]; }
};

__device__ __forceinline__ bool wg_parse_int(const PatText& t, int64_t& i, int64_t n, int64_t& val)
{
    // std::stoi: leading white space, an optional sign, at least one digit; anything after the digits is ignored
    char ch;
    while (i < n && ((ch = t.at(i)) == ' ' || (ch >= 9 && ch <= 13 && ch != '\n' && ch != '\t'))) i++;
    bool neg = false;
    if (i < n && ((ch = t.at(i)) == '-' || ch == '+')) { neg = ch == '-'; i++; }
    if (!(i < n && (ch = t.at(i)) >= '0' && ch <= '9')) return false;
    int64_t v = 0;
    while (i < n && (ch = t.at(i)) >= '0' && ch <= '9') { v = v * 10 + (ch - '0'); if (v > 0x7fffffffLL) return false; i++; }
    val = neg ? -v : v;
    return true;
}

__global__ __launch_bounds__(WG_BLOCK) void k_pat_count(const char* __restrict__ text, int64_t n, int64_t start, int64_t end,
                                                        int32_t* __restrict__ meth, int32_t* __restrict__ cov, unsigned long long* bad,
                                                        unsigned long long chunk_off)
{
    static_assert(WG_PAT_TILE == 16 * WG_BLOCK && WG_PAT_OVER % 16 == 0 && WG_PAT_OVER / 16 <= WG_BLOCK, "16 bytes per thread");
    __shared__ __attribute__((aligned(16))) char tx[16 + WG_PAT_TILE + WG_PAT_OVER];     // tx[15] = the byte before the tile; the tile from tx[16]
    __shared__ uint16_t lstart[WG_PAT_TILE / 2 + 1];         // tile-relative first bytes of the lines that begin in the tile (at most every other byte)
    __shared__ uint32_t wtot[WG_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * WG_PAT_TILE;
    // ---- stage: 16 bytes per thread (the chunk's buffer is 16-byte aligned and so is base), bytes at or past n as '\n'
    auto stage16 = [&](int64_t off) {                        // off: tile-relative, multiple of 16
        const int64_t a = base + off;
        uint4 v;
        if (a + 16 <= n) v = *reinterpret_cast<const uint4*>(text + a);
        else {
            uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int j = 0; j < 16; j++) w[j >> 2] |= (uint32_t)(unsigned char)(a + j < n ? text[a + j] : '\n') << (8 * (j & 3));
            v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
        }
        *reinterpret_cast<uint4*>(tx + 16 + off) = v;
    };
    stage16((int64_t)tid * 16);
    if (tid < WG_PAT_OVER / 16) stage16(WG_PAT_TILE + (int64_t)tid * 16);
    if (tid == 0) tx[15] = base > 0 ? text[base - 1] : '\n';
    __syncthreads();
    // ---- line starts of the tile: byte x begins a line when the byte before it is a newline and it is not one itself (empty lines: skipped,
    // stdin2beta.cpp:100)
    uint32_t mask = 0;
    {
        const char* q = tx + 16 + tid * 16;
        char prev = q[-1];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const char c = q[j];
            if (prev == '\n' && c != '\n' && base + tid * 16 + j < n) mask |= 1u << j;
            prev = c;
        }
    }
    const uint32_t cnt = (uint32_t)__popc(mask);
    const uint32_t incl = wg_wave_incl_scan_dpp_u32(cnt);
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    uint32_t before = incl - cnt, total = 0;
#pragma unroll
    for (int w = 0; w < WG_BLOCK / 64; w++) { if (w < wv) before += wtot[w]; total += wtot[w]; }
    while (mask) {
        const int j = __ffs((int)mask) - 1;
        mask &= mask - 1;
        lstart[before++] = (uint16_t)(tid * 16 + j);
    }
    __syncthreads();
    // ---- one line per thread
    const PatText T = {tx + 16, text, base, n};
    const int64_t nr = end - start;
    for (uint32_t l = (uint32_t)tid; l < total; l += WG_BLOCK) {
        const int64_t p = base + lstart[l];
        int64_t i = p;
        char ch = 0;
        bool ok = true;
        while (i < n && (ch = T.at(i)) != '\t' && ch != '\n') i++;   // field 1: chromosome
        ok = i < n && ch == '\t';
        int64_t site = 0, count = 0, ps = 0, plen = 0;
        if (ok) {
            i++;
            ok = wg_parse_int(T, i, n, site);                           // field 2: index of the read's first CpG
        }
        if (ok) {
            while (i < n && (ch = T.at(i)) != '\t' && ch != '\n') i++;
            ok = i < n && ch == '\t';
        }
        if (ok) {
            i++;
            ps = i;                                                     // field 3: the pattern
            while (i < n && (ch = T.at(i)) != '\t' && ch != '\n') i++;
            ok = i < n && ch == '\t';
            plen = i - ps;
        }
        if (ok) {
            i++;
            ok = i < n && T.at(i) != '\n' && wg_parse_int(T, i, n, count);   // field 4: how many reads (an empty one: stoi throws)
        }
        if (!ok) { atomicMin(bad, chunk_off + (unsigned long long)p); continue; }
        if (site + plen - 1 < start || site >= end) continue;           // stdin2beta.cpp:75-78
        for (int64_t k = 0; k < plen; k++) {
            const int64_t x = site - start + k;
            if (x < 0 || x >= nr) continue;
            const char c = T.at(ps + k);
            if (!(c == 'T' || c == 'C' || c == 'H')) continue;
            atomicAdd(&cov[x], (int32_t)count);
            if (c != 'T') atomicAdd(&meth[x], (int32_t)count);
        }
    }
}

// counts -> .beta (uint8 pairs) or .lbeta (uint16 pairs): rows whose coverage exceeds the type's maximum M become
// (trunc(meth / cov * M), M) (utils_wgbs.py:277-290; the same rule as modes 1 / 2 of the block reduction)
__global__ __launch_bounds__(WG_BLOCK) void k_pat_trim(const int32_t* __restrict__ meth, const int32_t* __restrict__ cov, int64_t n, int lbeta, void* __restrict__ out)
{
    const int64_t x = (int64_t)blockIdx.x * WG_BLOCK + threadIdx.x;
    if (x >= n) return;
    wg_block_sum_store(out, x, lbeta ? 2 : 1, 0u, (uint64_t)(uint32_t)meth[x], (uint64_t)(uint32_t)cov[x]);
}

// ------------------------------------------------------------------------------------------------------------
// k_marker_stats: per block, the statistics of a target and a background set of samples that `wgbstools find_markers` filters
// on (find_markers.py:188-196 coverage filter, :318-335 find_X_markers: nanmean / min / max per group), from the
// device-resident table of meth/cov ratios (mode 3 of the block reduction: NaN = below min_cov).  One thread per block;
// the samples of a set are visited in the order given (the column order of the reference's DataFrame): the sum is the
// sequential one numpy takes over that axis.  out[b] = {n_tg, sum_tg, min_tg, max_tg, n_bg, sum_bg, min_bg, max_bg}
// (min / max NaN when the set has no value).
// ------------------------------------------------------------------------------------------------------------
// (round 6: two blocks per thread, a group of four samples' loads in flight before the first of them is used, selects instead of branches —
//  the same operations on the same values in the same order: a thread's one dependent 8-byte load at a time held the pass at 0.50 of the HBM peak)
__device__ __forceinline__ void wg_marker_fold(double v, double& cnt, double& sum, double& mn, double& mx)
{
    const bool num = v == v;
    cnt += num ? 1.0 : 0.0;
    sum += num ? v : 0.0;                                       // numpy's nanmean adds the zero it put in place of the NaN
    const double lo = (mn == mn) ? (v < mn ? v : mn) : v, hi = (mx == mx) ? (v > mx ? v : mx) : v;
    mn = num ? lo : mn;
    mx = num ? hi : mx;
}

// A wavefront's 64 result rows (8 doubles each) are contiguous in `out`: staged in LDS and written as 16-byte vectors, a wavefront store = 1 KB in one
// piece (the direct form — eight 8-byte stores per thread, 64 bytes apart across the lanes — touched 64 cache lines per store instruction).
__device__ __forceinline__ void wg_marker_store(double (*st)[9], double* __restrict__ out, int64_t b, int64_t n_blocks, int lane, const double (&r)[8])
{
    const int64_t wave_b = b - lane;                             // first block of the wavefront
    if (wave_b + 64 <= n_blocks) {
#pragma unroll
        for (int q = 0; q < 8; q++) st[lane][q] = r[q];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double2* o2 = reinterpret_cast<double2*>(out + wave_b * 8);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int e = 128 * j + 2 * lane;                    // element of the wavefront's 512 doubles
            o2[64 * j + lane] = make_double2(st[e >> 3][e & 7], st[e >> 3][(e & 7) + 1]);
        }
        __builtin_amdgcn_wave_barrier();
    } else if (b < n_blocks) {
        double* o = out + b * 8;
#pragma unroll
        for (int q = 0; q < 8; q++) o[q] = r[q];
    }
}

__global__ __launch_bounds__(WG_BLOCK) void k_marker_stats(const double* __restrict__ V, int64_t n_blocks, const int32_t* __restrict__ tg, int n_tg,
                                                           const int32_t* __restrict__ bg, int n_bg, double* __restrict__ out)
{
    __shared__ double stage[WG_BLOCK / 64][64][9];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t half = (int64_t)gridDim.x * WG_BLOCK;         // the grid covers half of the blocks: thread t takes blocks t and t + half
    const int64_t b0 = (int64_t)blockIdx.x * WG_BLOCK + threadIdx.x, b1 = b0 + half;
    // (no early return: the wavefront stores below want all 64 lanes; a lane without a block reads block 0 and drops the result)
    const int64_t c0 = b0 < n_blocks ? b0 : 0, c1 = b1 < n_blocks ? b1 : c0;
    const double nan = __builtin_nan("");
    double r0[8], r1[8];
#pragma unroll
    for (int set = 0; set < 2; set++) {
        const int32_t* idx = set ? bg : tg;
        const int n = set ? n_bg : n_tg;
        double cnt0 = 0.0, sum0 = 0.0, mn0 = nan, mx0 = nan, cnt1 = 0.0, sum1 = 0.0, mn1 = nan, mx1 = nan;
        int k = 0;
        for (; k + 4 <= n; k += 4) {
            double u[4], w[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { const int64_t row = (int64_t)idx[k + q] * n_blocks; u[q] = V[row + c0]; w[q] = V[row + c1]; }
#pragma unroll
            for (int q = 0; q < 4; q++) { wg_marker_fold(u[q], cnt0, sum0, mn0, mx0); wg_marker_fold(w[q], cnt1, sum1, mn1, mx1); }
        }
        for (; k < n; k++) {
            const int64_t row = (int64_t)idx[k] * n_blocks;
            const double u = V[row + c0], w = V[row + c1];
            wg_marker_fold(u, cnt0, sum0, mn0, mx0); wg_marker_fold(w, cnt1, sum1, mn1, mx1);
        }
        r0[4 * set] = cnt0; r0[4 * set + 1] = sum0; r0[4 * set + 2] = mn0; r0[4 * set + 3] = mx0;
        r1[4 * set] = cnt1; r1[4 * set + 1] = sum1; r1[4 * set + 2] = mn1; r1[4 * set + 3] = mx1;
    }
    wg_marker_store(stage[wv], out, b0, n_blocks, lane, r0);
    wg_marker_store(stage[wv], out, b1, n_blocks, lane, r1);
}

// ------------------------------------------------------------------------------------------------------------
// test hooks
// ------------------------------------------------------------------------------------------------------------
// fast == 2: `rows` = wg_lookup_rows(pc, the ABI's largest block total), the k-scaled tables the scoring kernels would build
__global__ void k_debug_terms(const float* nm, const float* nt, int64_t count, float pc, float* out, int fast, int rows)
{
    __shared__ wg_fast_tables tb;
    __shared__ wg_d2 iyt[(WG_KY_KMIN + 1) * 16], kyt[(WG_KY_KMIN + 1) * 64];
    wg_fast_tables_to_lds(&tb, threadIdx.x, blockDim.x);
    if (fast == 2) wg_lookup_tables_to_lds(iyt, kyt, rows, nullptr, threadIdx.x, blockDim.x);
    __syncthreads();
    const float pc2 = pc + pc;
    if (fast == 2) {
        // the guard-free form with the zero-coverage exception against what the scoring kernels run (k-scaled tables, no
        // exception): the two may differ only in the sign of a zero (ntotal == 0: +0 vs -0, the same contribution to a
        // sum); anything else comes back as NaN
        const wg_d2* iy0 = iyt + (rows - 1) * 16;
        const wg_d2* ky0 = kyt + (rows - 1) * 64;
        for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < count; q += (int64_t)gridDim.x * blockDim.x) {
            const float a = wg_sample_term_pcpos(nm[q], nt[q], pc, pc2, &tb, &g_wg_tables);
            const float b = wg_sample_term_pcpos_ks(nm[q], nt[q], pc, pc2, iy0, ky0, &g_wg_tables);
            out[q] = (wg_f2u(a) == wg_f2u(b) || (a == 0.0f && b == 0.0f && nt[q] == 0.0f)) ? a : __builtin_nanf("");
        }
        return;
    }
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < count; q += (int64_t)gridDim.x * blockDim.x)
        out[q] = fast == 1 ? wg_sample_term(nm[q], nt[q], pc, pc2, &tb, &g_wg_tables) : wg_sample_term_plain(nm[q], nt[q], pc, pc2, &g_wg_tables);
}

// k_check_div: does the short division core give the IEEE quotient for EVERY operand pair a narrow scoring tile can form with
// this pseudo count?  a = fl(nmeth + pc), b = fl(ntotal + pc2), 0 <= nmeth <= ntotal <= max_total.  One workgroup per ntotal.
// *mismatches counts the pairs that differ (0 = the short core may be used).
__global__ __launch_bounds__(WG_BLOCK) void k_check_div(float pc, float pc2, int max_total, unsigned int* __restrict__ mismatches)
{
    const int nt_i = (int)blockIdx.x;
    if (nt_i > max_total) return;
    const float b = (float)nt_i + pc2;
    unsigned int bad = 0;
    for (int nm_i = (int)threadIdx.x; nm_i <= nt_i; nm_i += WG_BLOCK) {
        const float a = (float)nm_i + pc;
        const float want = wg_opaque_f32(a) / wg_opaque_f32(b);                  // the compiler's full IEEE sequence
        bad += wg_f2u(wg_div_f32_short(a, b)) != wg_f2u(want) ? 1u : 0u;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// test hook: the short division core on arrays of operands
__global__ void k_debug_div_short(const float* a, const float* b, int64_t count, uint32_t* out)
{
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < count; q += (int64_t)gridDim.x * blockDim.x)
        out[q] = wg_f2u(wg_div_f32_short(a[q], b[q]));
}

// test hook: wg_div_f32 vs the compiler's IEEE division, both on the device
__global__ void k_debug_div(const float* a, const float* b, int64_t count, uint32_t* out_fast, uint32_t* out_ieee)
{
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < count; q += (int64_t)gridDim.x * blockDim.x) {
        out_fast[q] = wg_f2u(wg_div_f32(a[q], b[q]));
        out_ieee[q] = wg_f2u(a[q] / b[q]);
    }
}

__global__ void k_debug_log2(uint32_t first, int64_t count, uint32_t* out_f, uint64_t* out_d, uint64_t* out_fast)
{
    __shared__ wg_fast_tables tb;
    wg_fast_tables_to_lds(&tb, threadIdx.x, blockDim.x);
    __syncthreads();
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < count; q += (int64_t)gridDim.x * blockDim.x) {
        const float p = wg_u2f(first + (uint32_t)q);
        if (out_f) out_f[q] = wg_f2u(wg_log2f(p, tb.f_tab));
        if (out_d) out_d[q] = wg_d2u(wg_log2(1.0 - (double)p, g_wg_tables.d_tab, g_wg_tables.d_tab2));
        if (out_fast) out_fast[q] = wg_d2u(wg_fast_log2(1.0 - (double)p, tb.d_fast));
    }
}

// Materialise P[t], t = 0..len, for one range of one sample from the carries (test / block-sum helper):
// one wavefront per (sample, carry group of absolute site index).
__global__ __launch_bounds__(64) void k_prefix_materialise(JobView J, int nG, uint32_t* __restrict__ out, int len)
{
    const int lane = threadIdx.x;
    const int g = blockIdx.x, s = blockIdx.y;
    const ChunkDesc cd = J.chunks[0];
    const uint8_t* row = J.betas + (int64_t)s * J.pitch;
    const uint2 c0 = J.carry[cd.carry_off + (int64_t)s * cd.nG + g];
    const int64_t gabs = (((cd.start0 >> WG_CARRY_SHIFT) + g) << WG_CARRY_SHIFT);
    uint32_t* o = out + ((int64_t)s * (len + 1)) * 2;
    uint32_t run_m = c0.x, run_t = c0.y;
    for (int q = 0; q < WG_CARRY_G; q += 64) {
        const int64_t x = gabs + q + lane - cd.start0;      // chunk-relative site of this lane
        uint32_t m = 0, t = 0;
        if (x >= 0 && x < len) { m = row[2 * (cd.start0 + x)]; t = row[2 * (cd.start0 + x) + 1]; }
        const uint32_t im = wg_wave_incl_scan_dpp_u32(m), it = wg_wave_incl_scan_dpp_u32(t);
        if (x >= 0 && x < len) { o[2 * (x + 1)] = run_m + im; o[2 * (x + 1) + 1] = run_t + it; }
        run_m += (uint32_t)__builtin_amdgcn_readlane((int)im, 63);
        run_t += (uint32_t)__builtin_amdgcn_readlane((int)it, 63);
    }
    if (g == 0 && lane == 0) { o[0] = 0; o[1] = 0; }
    (void)nG;
}
