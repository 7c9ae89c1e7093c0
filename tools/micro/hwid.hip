// Where do the waves of co-resident workgroups land?  Prints HW_ID fields of every wave of a 512-workgroup launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t* xcc)
{
    extern __shared__ char sm[];
    sm[threadIdx.x] = 1;
    const uint32_t h = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const uint32_t x = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    if ((threadIdx.x & 63) == 0) { out[blockIdx.x * 4 + (threadIdx.x >> 6)] = h; xcc[blockIdx.x * 4 + (threadIdx.x >> 6)] = x; }
    // stay resident for a while so that the launch fills the machine
    long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 2000000) { }
}
int main()
{
    const int nb = 483;
    uint32_t *d, *dx; hipMalloc(&d, nb * 16); hipMalloc(&dx, nb * 16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, nb, 256, 67000, 0, d, dx);
    std::vector<uint32_t> h(nb * 4), x(nb * 4);
    hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost); hipMemcpy(x.data(), dx, nb * 16, hipMemcpyDeviceToHost);
    for (int b = 0; b < 12; b++) {
        printf("wg %3d:", b);
        for (int w = 0; w < 4; w++) { uint32_t v = h[b * 4 + w]; printf("  [w%d slot %u simd %u pipe %u cu %u sh %u se %u xcc %u raw %08x]", w, v & 15, (v >> 4) & 3, (v >> 6) & 3, (v >> 8) & 15, (v >> 12) & 1, (v >> 13) & 7, x[b * 4 + w] & 15, v); }
        printf("\n");
    }
    // how many workgroup pairs share (xcc, se, sh, cu), and do their wave-0s share a SIMD?
    std::map<uint32_t, std::vector<int>> cu;
    for (int b = 0; b < nb; b++) { uint32_t v = h[b * 4]; cu[((x[b * 4] & 15) << 16) | (v & 0xff00)].push_back(b); }
    int shared = 0, same_simd = 0, key_clash = 0;
    for (auto& kv : cu) if (kv.second.size() >= 2) {
        shared++;
        int a = kv.second[0], b = kv.second[1];
        if (((h[a * 4] >> 4) & 3) == ((h[b * 4] >> 4) & 3)) same_simd++;
        auto dp = [&](int g) { for (int w = 0; w < 4; w++) { uint32_t v = h[g * 4 + w]; if (((((v >> 4) & 3) - (v & 15)) & 3) == 0) return (int)((v >> 4) & 3); } return (int)((h[g * 4] >> 4) & 3); };
        if (dp(a) == dp(b)) key_clash++;
    }
    printf("CUs in use %zu, CUs with >= 2 workgroups %d, of those wave0 on the same SIMD %d, rule picks the same SIMD %d\n", cu.size(), shared, same_simd, key_clash);
    return 0;
}
