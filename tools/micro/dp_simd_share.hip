// Micro-benchmark (round 5): what does a recurrence wavefront lose when ANOTHER recurrence wavefront — or a busy worker — shares its SIMD?
// One workgroup of 8 wavefronts on one CU (wavefront w lands on SIMD w mod 4: checked by reading HW_ID); a mask says which wavefronts run the
// recurrence's dependent chain, another which run a "worker-like" independent VALU stream; the rest leave.  Reported: ns per step of wavefront 0's
// chain (s_memtime around the loop), for the product's lean step (A C X D Rlo Rhi + two wait-state slots) and for the chain without the arg-max
// bookkeeping (A X Rlo Rhi + two slots).
// Build: hipcc --offload-arch=gfx950 -O3 -o dp_simd_share dp_simd_share.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define STEP_LEAN(STP) \
    "v_add_f64 v[10:11], s[20:21], v[12:13]\n" \
    "v_cmp_gt_f64 vcc, v[10:11], v[2:3]\n" \
    "v_max_f64 v[2:3], v[2:3], v[10:11]\n" \
    "v_cndmask_b32_e64 v4, v4, " #STP ", vcc\n" \
    "v_readlane_b32 s20, v2, " #STP "\n" \
    "v_readlane_b32 s21, v3, " #STP "\n" \
    "s_nop 1\n"

#define STEP_CHAIN(STP) \
    "v_add_f64 v[10:11], s[20:21], v[12:13]\n" \
    "v_max_f64 v[2:3], v[2:3], v[10:11]\n" \
    "v_readlane_b32 s20, v2, " #STP "\n" \
    "v_readlane_b32 s21, v3, " #STP "\n" \
    "s_nop 1\n"

#define S8(M,a,b,c,d,e,f,g,h) M(a) M(b) M(c) M(d) M(e) M(f) M(g) M(h)
#define ALL64(M) S8(M,0,1,2,3,4,5,6,7) S8(M,8,9,10,11,12,13,14,15) S8(M,16,17,18,19,20,21,22,23) S8(M,24,25,26,27,28,29,30,31) \
                 S8(M,32,33,34,35,36,37,38,39) S8(M,40,41,42,43,44,45,46,47) S8(M,48,49,50,51,52,53,54,55) S8(M,56,57,58,59,60,61,62,63)

template <int V>
__global__ __launch_bounds__(512) void k(double* out, long long* cyc, uint32_t* hw, int iters, uint32_t chain_mask, uint32_t busy_mask, int prio)
{
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) hw[w] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    double best = -1.0 * (threadIdx.x & 63), cv = -0.25;
    if ((chain_mask >> w) & 1u) {
        if (prio) __builtin_amdgcn_s_setprio(3);
        const long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; it++) {
            if (V == 0)
                asm volatile("v_mov_b32 v2, %0\n v_mov_b32 v3, %1\n v_mov_b32 v12, %2\n v_mov_b32 v13, %3\n s_mov_b32 s20, 0\n s_mov_b32 s21, 0\n v_mov_b32 v4, 0\n"
                             ALL64(STEP_LEAN) "v_mov_b32 %0, v2\n v_mov_b32 %1, v3\n"
                             : "+v"(((uint32_t*)&best)[0]), "+v"(((uint32_t*)&best)[1]) : "v"(((uint32_t*)&cv)[0]), "v"(((uint32_t*)&cv)[1])
                             : "v2","v3","v4","v5","v10","v11","v12","v13","s20","s21","s22","s23","vcc");
            else
                asm volatile("v_mov_b32 v2, %0\n v_mov_b32 v3, %1\n v_mov_b32 v12, %2\n v_mov_b32 v13, %3\n s_mov_b32 s20, 0\n s_mov_b32 s21, 0\n"
                             ALL64(STEP_CHAIN) "v_mov_b32 %0, v2\n v_mov_b32 %1, v3\n"
                             : "+v"(((uint32_t*)&best)[0]), "+v"(((uint32_t*)&best)[1]) : "v"(((uint32_t*)&cv)[0]), "v"(((uint32_t*)&cv)[1])
                             : "v2","v3","v4","v5","v10","v11","v12","v13","s20","s21","s22","s23","vcc");
        }
        const long long t1 = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) cyc[w] = t1 - t0;
        out[threadIdx.x] = best;
    } else if ((busy_mask >> w) & 1u) {
        // a worker-like stream: independent integer / fp32 / fp64 work, no dependence on the chain; runs about as long as the chains do
        float a = (float)threadIdx.x, b = 1.0001f;
        double d = 0.5 * threadIdx.x;
        uint32_t u = threadIdx.x;
        for (int it = 0; it < iters * 12; it++) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                a = a * b + 0.5f; u = u * 1664525u + 1013904223u; d = d * 1.0000001 + 0.25; a += (float)(u >> 20);
            }
        }
        out[threadIdx.x] = (double)a + d + (double)u;
    }
}

int main()
{
    double* out; long long* cyc; uint32_t* hw;
    hipMalloc(&out, 512 * 8); hipMalloc(&cyc, 8 * 8); hipMalloc(&hw, 8 * 4);
    const int iters = 4000;
    struct Cfg { const char* what; uint32_t chain, busy; int prio; };
    const Cfg cfgs[] = {
        {"wavefront 0 alone", 0x01, 0x00, 1},
        {"chains on wavefronts 0 and 4 (the SAME SIMD)", 0x11, 0x00, 1},
        {"chains on wavefronts 0 and 1 (different SIMDs)", 0x03, 0x00, 1},
        {"chains on wavefronts 0 and 2 (different SIMDs)", 0x05, 0x00, 1},
        {"chains on wavefronts 0,1,2,3 (one per SIMD)", 0x0f, 0x00, 1},
        {"chain on 0, busy worker on 4 (same SIMD), chain has priority", 0x01, 0x10, 1},
        {"chain on 0, busy worker on 4 (same SIMD), no priority", 0x01, 0x10, 0},
        {"chain on 0, busy workers on 1,2,3,5,6,7 (other SIMDs)", 0x01, 0xee, 1},
        {"chain on 0, busy workers on all of 1..7", 0x01, 0xfe, 1},
        {"chains on 0 and 4, busy workers on 1,2,3,5,6,7", 0x11, 0xee, 1},
        {"chains on 0 and 1, busy workers on 2,3,6,7 (SIMDs 2,3 only)", 0x03, 0xcc, 1},
    };
    // s_memtime ticks per ns: calibrate against an event-timed run
    for (int v = 0; v < 2; v++) {
        printf("== %s\n", v == 0 ? "lean step: A C X D Rlo Rhi + 2 wait-state slots (the product's k_dp<7,64,LEAN> chain, 6 VALU)" : "chain only: A X Rlo Rhi + 2 wait-state slots (4 VALU)");
        for (const Cfg& c : cfgs) {
            hipMemset(cyc, 0, 64);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (v == 0) hipLaunchKernelGGL(k<0>, 1, 512, 0, 0, out, cyc, hw, iters, c.chain, c.busy, c.prio);
                else        hipLaunchKernelGGL(k<1>, 1, 512, 0, 0, out, cyc, hw, iters, c.chain, c.busy, c.prio);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[8]; uint32_t hh[8];
            hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost); hipMemcpy(hh, hw, 32, hipMemcpyDeviceToHost);
            printf("  %-62s wavefront 0: %7.2f memtime ticks/step", c.what, (double)h[0] / (64.0 * iters));
            for (int w = 1; w < 8; w++) if ((c.chain >> w) & 1u) printf(", w%d %7.2f", w, (double)h[w] / (64.0 * iters));
            printf("  | kernel %.2f ms (%.1f ns/step if the chain is the kernel) | SIMD of w0..7:", ms, ms * 1e6 / (64.0 * iters));
            for (int w = 0; w < 8; w++) printf(" %u", (hh[w] >> 4) & 3);
            printf("\n");
        }
    }
    return 0;
}
