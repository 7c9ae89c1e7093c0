// What does one sparse vector load cost the CU?  k_dp's workers bring every row of scored blocks with ONE global_load_dwordx2 whose
// lanes j < F (about 20 of 64) are active; 7 workers x 2 workgroups per CU issue ~140 of them per batch of 64 steps, and the workers'
// time goes into issuing them (tools/dp_timing.py).  This measures wavefront-level load instructions per CU and microsecond for
// 64 / 32 / 16 / 8 active lanes (contiguous, 8 or 16 bytes per lane), data resident in L2, 8 wavefronts per CU issuing back to back.
//   hipcc --offload-arch=gfx950 -O3 -o _build/vmem_rate vmem_rate.hip && _build/vmem_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int BYTES>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, int active, int iters, int stride_bytes, double* out)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const char* p = src + ((size_t)blockIdx.x * 8 + wv) * 65536 + (size_t)lane * BYTES;
    double acc = 0;
    if (lane < active) {
        for (int it = 0; it < iters; it++) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const char* q = p + (size_t)((it * 8 + u) & 63) * stride_bytes;
                if (BYTES == 8) v[u] = *reinterpret_cast<const double*>(q);
                else { const double2 w = *reinterpret_cast<const double2*>(q); v[u] = w.x + w.y; }
            }
#pragma unroll
            for (int u = 0; u < 8; u++) acc += v[u];
            asm volatile("" : "+v"(acc));
        }
    }
    if (acc == 123.456) out[0] = acc;
}

template <int BYTES>
void run(const char* src, int active, int stride, double* out)
{
    const int iters = 2000, wgs = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<BYTES>, wgs, 512, 0, 0, src, active, 50, stride, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<BYTES>, wgs, 512, 0, 0, src, active, iters, stride, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_cu = (double)iters * 8 * 8;                 // wavefront load instructions per CU
    printf("%2d bytes/lane, %2d active lanes, row stride %4d B: %7.1f ns per wavefront load per CU (%.1f loads/us/CU), %.0f GB/s of useful bytes on the chip\n",
           BYTES, active, stride, ms * 1e6 / per_cu, per_cu / (ms * 1e3), per_cu * 256 * active * BYTES / (ms * 1e-3) / 1e9);
}

int main()
{
    char* src; double* out;
    hipMalloc(&src, (size_t)256 * 8 * 65536 + 65536); hipMemset(src, 0, (size_t)256 * 8 * 65536 + 65536); hipMalloc(&out, 8);
    for (int active : {64, 32, 20, 16, 8, 4}) run<8>(src, active, 512, out);
    for (int active : {64, 32, 16, 8}) run<16>(src, active, 1024, out);
    for (int active : {64, 20}) run<8>(src, active, 168, out);       // rows packed back to back (20 doubles + a gap), as a CSR row is
    return 0;
}
