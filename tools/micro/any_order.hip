// Does hipExtAnyOrderLaunch clear the barrier bit of a kernel packet on gfx950 (hip_ext.h says "not supported on GFX9xx" for the
// module form)?  Kernel A: a launch whose LAST workgroups run long (a tail); kernel B behind it on the SAME stream, once as an ordinary
// launch, once with hipExtAnyOrderLaunch.  If the flag works, B's workgroups fill A's tail: B's stop event comes ~one B-duration after
// A's start of tail, not after A's end.
//   hipcc --offload-arch=gfx950 -O2 -o _build/any_order any_order.hip && _build/any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin(long long ticks_short, long long ticks_long, int n_long, unsigned* sink)
{
    const long long want = (int)blockIdx.x >= (int)gridDim.x - n_long ? ticks_long : ticks_short;
    const long long t0 = wall_clock64();
    unsigned x = 0;
    while (wall_clock64() - t0 < want) x += 1;
    if (x == 0xffffffffu) *sink = x;
}

int main()
{
    hipStream_t s;
    CK(hipStreamCreate(&s));
    unsigned* sink;
    CK(hipMalloc(&sink, 4));
    hipEvent_t a0, a1, b0, b1;
    CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    // wall_clock64: 100 MHz.  A: 4096 workgroups of 0.1 ms, the last 64 of 3 ms.  B: 4096 workgroups of 0.1 ms.
    const long long t_short = 10000, t_long = 300000;
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            CK(hipDeviceSynchronize());
            hipExtLaunchKernelGGL(spin, dim3(4096), dim3(256), 0, s, a0, a1, 0, t_short, t_long, 64, sink);
            CK(hipGetLastError());
            if (mode == 0) hipExtLaunchKernelGGL(spin, dim3(4096), dim3(256), 0, s, b0, b1, 0, t_short, t_short, 0, sink);
            else if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(4096), dim3(256), 0, s, b0, b1, hipExtAnyOrderLaunch, t_short, t_short, 0, sink);
            else {      // ordinary launches + event records (what the library does today)
                CK(hipEventRecord(b0, s));
                hipLaunchKernelGGL(spin, dim3(4096), dim3(256), 0, s, t_short, t_short, 0, sink);
                CK(hipEventRecord(b1, s));
            }
            CK(hipGetLastError());
            CK(hipStreamSynchronize(s));
            float A = 0, B = 0, AB = 0;
            CK(hipEventElapsedTime(&A, a0, a1));
            CK(hipEventElapsedTime(&AB, a0, b1));
            if (mode != 2) CK(hipEventElapsedTime(&B, b0, b1));
            printf("mode %d (%s) rep %d: A %.3f ms, B %.3f ms, A.start -> B.stop %.3f ms\n", mode,
                   mode == 0 ? "ext launch, flags 0" : mode == 1 ? "ext launch, hipExtAnyOrderLaunch" : "plain launch", rep, A, B, AB);
        }
    }
    return 0;
}
