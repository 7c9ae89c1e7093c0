// Micro-benchmark: cycles per step of the recurrence's dependent chain on gfx950, in several instruction orders.
// Build: hipcc --offload-arch=gfx950 -O3 -o dp_chain dp_chain.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define STEP_CHAIN(STP) \
    "v_add_f64 v[10:11], s[20:21], v[12:13]\n" \
    "v_max_f64 v[2:3], v[2:3], v[10:11]\n" \
    "v_readlane_b32 s20, v2, " #STP "\n" \
    "v_readlane_b32 s21, v3, " #STP "\n"

// compiler-like order (all 10 instructions)
#define STEP_FULL(STP) \
    "v_add_f64 v[10:11], s[20:21], v[12:13]\n" \
    "v_cmp_gt_f64 vcc, v[10:11], v[2:3]\n" \
    "v_max_f64 v[2:3], v[2:3], v[10:11]\n" \
    "v_readlane_b32 s21, v3, " #STP "\n" \
    "v_cndmask_b32_e64 v4, v4, " #STP ", vcc\n" \
    "v_readlane_b32 s20, v2, " #STP "\n" \
    "v_readlane_b32 s22, v4, " #STP "\n" \
    "v_writelane_b32 v5, s22, " #STP "\n" \
    "v_writelane_b32 v2, 0, " #STP "\n" \
    "v_writelane_b32 v3, s23, " #STP "\n"

// interleaved: tail of the step after the next add is not expressible inside one step macro; approximate by ordering
#define STEP_ILV(STP) \
    "v_add_f64 v[10:11], s[20:21], v[12:13]\n" \
    "v_writelane_b32 v5, s22, " #STP "\n" \
    "v_cmp_gt_f64 vcc, v[10:11], v[2:3]\n" \
    "v_max_f64 v[2:3], v[2:3], v[10:11]\n" \
    "v_cndmask_b32_e64 v4, v4, " #STP ", vcc\n" \
    "v_readlane_b32 s20, v2, " #STP "\n" \
    "v_readlane_b32 s21, v3, " #STP "\n" \
    "v_writelane_b32 v2, 0, " #STP "\n" \
    "v_writelane_b32 v3, s23, " #STP "\n" \
    "v_readlane_b32 s22, v4, " #STP "\n"

#define R8(M, B) M(B) M(B+1) M(B+2) M(B+3) M(B+4) M(B+5) M(B+6) M(B+7)
#define S8(M,a,b,c,d,e,f,g,h) M(a) M(b) M(c) M(d) M(e) M(f) M(g) M(h)
#define ALL64(M) S8(M,0,1,2,3,4,5,6,7) S8(M,8,9,10,11,12,13,14,15) S8(M,16,17,18,19,20,21,22,23) S8(M,24,25,26,27,28,29,30,31) \
                 S8(M,32,33,34,35,36,37,38,39) S8(M,40,41,42,43,44,45,46,47) S8(M,48,49,50,51,52,53,54,55) S8(M,56,57,58,59,60,61,62,63)

template <int V>
__global__ void k(double* out, long long* cyc, int iters)
{
    double best = -1.0 * threadIdx.x, cv = -0.25;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (V == 0)
            asm volatile("v_mov_b32 v2, %0\n v_mov_b32 v3, %1\n v_mov_b32 v12, %2\n v_mov_b32 v13, %3\n s_mov_b32 s20, 0\n s_mov_b32 s21, 0\n s_mov_b32 s23, 0xfff00000\n"
                         ALL64(STEP_CHAIN) "v_mov_b32 %0, v2\n v_mov_b32 %1, v3\n"
                         : "+v"(((uint32_t*)&best)[0]), "+v"(((uint32_t*)&best)[1]) : "v"(((uint32_t*)&cv)[0]), "v"(((uint32_t*)&cv)[1])
                         : "v2","v3","v4","v5","v10","v11","v12","v13","s20","s21","s22","s23","vcc");
        if (V == 1)
            asm volatile("v_mov_b32 v2, %0\n v_mov_b32 v3, %1\n v_mov_b32 v12, %2\n v_mov_b32 v13, %3\n s_mov_b32 s20, 0\n s_mov_b32 s21, 0\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0xfff00000\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n"
                         ALL64(STEP_FULL) "v_mov_b32 %0, v2\n v_mov_b32 %1, v3\n"
                         : "+v"(((uint32_t*)&best)[0]), "+v"(((uint32_t*)&best)[1]) : "v"(((uint32_t*)&cv)[0]), "v"(((uint32_t*)&cv)[1])
                         : "v2","v3","v4","v5","v10","v11","v12","v13","s20","s21","s22","s23","vcc");
        if (V == 2)
            asm volatile("v_mov_b32 v2, %0\n v_mov_b32 v3, %1\n v_mov_b32 v12, %2\n v_mov_b32 v13, %3\n s_mov_b32 s20, 0\n s_mov_b32 s21, 0\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0xfff00000\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n"
                         ALL64(STEP_ILV) "v_mov_b32 %0, v2\n v_mov_b32 %1, v3\n"
                         : "+v"(((uint32_t*)&best)[0]), "+v"(((uint32_t*)&best)[1]) : "v"(((uint32_t*)&cv)[0]), "v"(((uint32_t*)&cv)[1])
                         : "v2","v3","v4","v5","v10","v11","v12","v13","s20","s21","s22","s23","vcc");
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = best;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    double* out; long long* cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int v = 0; v < 3; v++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(k<0>, 1, 64, 0, 0, out, cyc, iters);
            if (v == 1) hipLaunchKernelGGL(k<1>, 1, 64, 0, 0, out, cyc, iters);
            if (v == 2) hipLaunchKernelGGL(k<2>, 1, 64, 0, 0, out, cyc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("variant %d: %.1f ns/step (event), %lld counter ticks/step*100\n", v, ms * 1e6 / (64.0 * iters), c * 100 / (64LL * iters));
        }
    }
    return 0;
}
