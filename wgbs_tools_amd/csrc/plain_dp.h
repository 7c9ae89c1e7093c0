// The chunk DP of the reference, statement for statement, for chunks whose loci are NOT ascending (hand-made genomes; a real
// CpG.bed.gz never has them).  segmentor.cpp:103-155 bars an extension whose locus lies more than max_bp ahead of the start's
// OR behind it, writes -inf for it and `continue`s WITHOUT adding the site to its running sums (:114-117) — so the counts of a
// block (k, i) are the sums over the sites of [k, i] that start k does not bar, a set that depends on k: no prefix sums, no
// windows, none of the fast path's structure survives.  These kernels follow the loops as written, at whatever speed:
//
//   k_plain_rows   one thread per start site k of a band: row k of the cost matrix, samples in file order, running float sums
//                  along j per sample (segmentor.cpp:119-137), the exact term (wg_sample_term_plain: libm restatements);
//                  also the `meth > cov` check of segmentor.cpp:181-188 for the chunk's bytes
//   k_plain_dp     one workgroup per chunk, the steps of the band in order: M[i+1] = max_k M[k] + row_k[i - k], first maximum
//                  (segmentor.cpp:142-154); T on the way
// The rows live in a ring of R >= band + W - 1 slots, j-major ([j][k mod R]) so that the threads of k_plain_rows write side by side.
#pragma once

struct PlainArgs {
    const uint8_t* betas; int64_t pitch, n_total; int32_t n_samples;
    const uint32_t* loci;        // of the whole resident range
    int64_t start0; int32_t n;   // the chunk
    int32_t W;                   // min(max_cpg, n)
    uint32_t max_bp; float pc, pc2;
    int32_t R;                   // ring slots
    double* buf;                 // [W][R]
    double* M; int32_t* T;       // [n + 1]
    unsigned long long* first_bad;
};

__global__ __launch_bounds__(WG_BLOCK) void k_find_disorder(JobView J, uint32_t* __restrict__ flags)
{
    const int c = blockIdx.x;
    const ChunkDesc cd = J.chunks[c];
    const uint32_t* loc = J.loci + cd.start0;
    bool dis = false;
    for (int k = threadIdx.x + 1; k < cd.len; k += WG_BLOCK) dis = dis || loc[k - 1] > loc[k];
    if (dis) flags[c] = 1u;
}

__global__ __launch_bounds__(WG_BLOCK) void k_plain_rows(PlainArgs A, int b0, int b1)
{
    const int k = b0 + (int)(blockIdx.x * WG_BLOCK + threadIdx.x);
    if (k >= b1) return;
    const uint32_t* loc = A.loci + A.start0;
    const uint32_t lk = loc[k];
    const int window = (A.n - k < A.W) ? A.n - k : A.W;                       // segmentor.cpp:110
    double* row = A.buf + (k % A.R);                                            // element j at row[j * R]
    for (int j = 0; j < A.W; j++) row[(size_t)j * A.R] = 0.0;                   // :106 std::fill
    for (int s = 0; s < A.n_samples; s++) {
        const uint8_t* b = A.betas + (int64_t)s * A.pitch + 2 * (A.start0 + k);
        float nmeth = 0.0f, ntotal = 0.0f;                                      // :108-109 (one sample's pair of the arrays)
        for (int j = 0; j < window; j++) {
            const uint32_t lj = loc[k + j];
            if ((uint32_t)(lj - lk) > A.max_bp || lj < lk) continue;            // :114-117 (the -inf is written below, once)
            const uint32_t m = b[2 * j], t = b[2 * j + 1];
            if (m > t) atomicMin(A.first_bad, ((unsigned long long)s << 40) | (unsigned long long)(A.start0 + k + j));      // :181-188
            nmeth += (float)m; ntotal += (float)t;                              // :122-123
            if (ntotal == 0.0f) continue;                                       // :125
            const float ll = wg_sample_term_plain(nmeth, ntotal, A.pc, A.pc2, &g_wg_tables);      // :127-134
            row[(size_t)j * A.R] += (double)ll;                                 // :135, samples in file order
        }
    }
    for (int j = 0; j < window; j++) {
        const uint32_t lj = loc[k + j];
        double& e = row[(size_t)j * A.R];
        if ((uint32_t)(lj - lk) > A.max_bp || lj < lk) e = -__builtin_inf();    // :115
        else if (!(e != 0.0)) e = 0.0;                                          // :137 `if (ll_sum) row[j] = ll_sum` over the 0.0 fill
    }
}

__global__ __launch_bounds__(WG_BLOCK) void k_plain_dp(PlainArgs A, int b0, int b1)
{
    __shared__ double sv[WG_BLOCK / 64];
    __shared__ int sk[WG_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (b0 == 0 && tid == 0) { __hip_atomic_store(&A.M[0], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); A.T[0] = 0; }      // :97-98 (zero-initialised arrays)
    __syncthreads();
    for (int i = b0; i < b1; i++) {
        const int start_k = (i + 1 - A.W > 0) ? i + 1 - A.W : 0;                // :145
        double best = -__builtin_inf();                                         // :143 (a float -inf widened)
        int arg = -1;
        for (int k = start_k + tid; k <= i; k += WG_BLOCK) {                    // ascending k per thread: strict '>' keeps its first maximum
            const double v = __hip_atomic_load(&A.M[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + A.buf[(size_t)(i - k) * A.R + (k % A.R)];
            if (v > best) { best = v; arg = k; }
        }
        // the first maximum over all threads = the largest value, the smallest k among equals (arg -1 = no candidate beat -inf)
        for (int d = 1; d < 64; d <<= 1) {
            const double ov = __shfl_xor(best, d);
            const int ok = __shfl_xor(arg, d);
            if (ok >= 0 && (arg < 0 || ov > best || (ov == best && ok < arg))) { best = ov; arg = ok; }
        }
        if (lane == 0) { sv[wv] = best; sk[wv] = arg; }
        __syncthreads();
        if (tid == 0) {
            for (int q = 1; q < WG_BLOCK / 64; q++)
                if (sk[q] >= 0 && (arg < 0 || sv[q] > best || (sv[q] == best && sk[q] < arg))) { best = sv[q]; arg = sk[q]; }
            __hip_atomic_store(&A.M[i + 1], best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            A.T[i + 1] = arg;
        }
        __syncthreads();
    }
}
