// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths k_cost uses (the guide calibrates only 16 B per lane reads):
// each kernel reads (or writes) a 1 GiB buffer exactly once with W bytes per lane, wavefront-contiguous; tools/pmc_cost_traffic.py runs this binary
// under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` and divides the known byte count by the counter.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/_build/fetch_calib tools/micro/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T>
__global__ __launch_bounds__(256) void calib_read(const T* __restrict__ p, size_t n, uint32_t* out)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (; i < n; i += stride) {
        const T v = p[i];
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
        for (unsigned j = 0; j < sizeof(T) / 4; j++) acc ^= w[j];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <typename T>
__global__ __launch_bounds__(256) void calib_write(T* __restrict__ p, size_t n, uint32_t seed)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        T v;
        uint32_t* w = reinterpret_cast<uint32_t*>(&v);
        for (unsigned j = 0; j < sizeof(T) / 4; j++) w[j] = seed + (uint32_t)i;
        p[i] = v;
    }
}
int main()
{
    const size_t bytes = 1ull << 30;
    void* buf; uint32_t* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, bytes);
    (void)hipDeviceSynchronize();
    const int wgs = 256 * 16;
    hipLaunchKernelGGL(calib_read<uint32_t>, wgs, 256, 0, 0, (const uint32_t*)buf, bytes / 4, out);
    hipLaunchKernelGGL(calib_read<uint2>, wgs, 256, 0, 0, (const uint2*)buf, bytes / 8, out);
    hipLaunchKernelGGL(calib_read<uint4>, wgs, 256, 0, 0, (const uint4*)buf, bytes / 16, out);
    hipLaunchKernelGGL(calib_write<uint2>, wgs, 256, 0, 0, (uint2*)buf, bytes / 8, 7u);
    hipLaunchKernelGGL(calib_write<uint4>, wgs, 256, 0, 0, (uint4*)buf, bytes / 16, 9u);
    (void)hipDeviceSynchronize();
    printf("fetch_calib: %zu bytes per kernel\n", bytes);
    return 0;
}
