// synth.hip — device-side generator of synthetic `.beta` bytes (bench / tests only; libwgbssynth.so).
// Bit-identical to wgbs_tools_amd/synth.py::synth_betas (integer-only, counter-based splitmix64), so a test can
// regenerate on the CPU exactly what the GPU holds.  Not part of the drop-in surface.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FORCE_BLOCK 4096
#define BLOCK_ODDS 40ull
#define ZERO_COV_THRESH 3277ull
#define S_BLOCK 1
#define S_LEVEL 2
#define S_SAMPLE0 16

__host__ __device__ inline uint64_t splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline uint64_t stream_key(uint64_t seed, uint64_t stream) { return splitmix64(seed ^ (stream * 0xD1B54A32D192ED03ull)); }
__device__ inline uint64_t hash_at(uint64_t key, uint64_t idx) { return splitmix64(key + idx); }

__device__ inline int level_map(int u)
{
    if (u < 64) return u >> 2;
    if (u < 208) return 200 + ((u - 64) * 55) / 144;
    return 16 + ((u - 208) * 184) / 48;
}

__global__ void k_block_starts(uint64_t key_block, int64_t lo, int64_t n, int32_t* bs)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int64_t i = lo + t;                                  // absolute site
    int64_t j = i;
    for (;;) {
        if ((j % FORCE_BLOCK) == 0) break;
        const uint64_t h = hash_at(key_block, (uint64_t)j);
        if ((((h >> 32) * BLOCK_ODDS) >> 32) == 0) break;
        j--;
    }
    bs[t] = (int32_t)j;
}

__global__ void k_fill_sample(uint64_t key_level, uint64_t key_jit, uint64_t key_cov, uint64_t key_bern, int64_t lo, int64_t n,
                              const int32_t* __restrict__ bs, uint8_t* __restrict__ row)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int64_t i = lo + t;                                  // absolute site; row[] starts at site lo
    const uint64_t b = (uint64_t)bs[t];
    const int base = level_map((int)(hash_at(key_level, b) & 0xFF));
    const uint64_t hs = hash_at(key_jit, b);
    int level;
    if ((hs & 15) == 0) level = level_map((int)((hs >> 8) & 0xFF));
    else {
        level = base + (int)((hs >> 16) & 31) - 16;
        level = level < 0 ? 0 : (level > 255 ? 255 : level);
    }
    const uint64_t hc = hash_at(key_cov, (uint64_t)i);
    int cov = __popcll((hc >> 16) & ((1ull << 48) - 1)) + 6;
    if ((hc & 0xFFFF) < ZERO_COV_THRESH) cov = 0;
    int meth = 0;
    for (int g = 0; g < 7; g++) {
        if (g * 8 >= cov) break;
        const uint64_t hb = hash_at(key_bern, (uint64_t)i * 8 + g);
        for (int byte = 0; byte < 8; byte++) {
            const int t = g * 8 + byte;
            const int trial = (int)((hb >> (8 * byte)) & 0xFF);
            meth += (trial < level && t < cov) ? 1 : 0;
        }
    }
    reinterpret_cast<uint16_t*>(row)[t] = (uint16_t)(meth | (cov << 8));
}

extern "C" {

// Fill rows [sample_first, sample_first+n_samples) of a device buffer [*][pitch] on HIP device `device` (-1: the current
// one) with sites [site_lo, site_hi) of synthetic samples number sample_first.. of seed `seed`: row byte 0 = site_lo.
// Returns 0 or a hipError_t.
int wgbssynth_fill_betas_range(void* d_base, int64_t pitch, int64_t site_lo, int64_t site_hi, int sample_first, int n_samples,
                               uint64_t seed, int device)
{
    hipError_t e;
    if (device >= 0) { e = hipSetDevice(device); if (e != hipSuccess) return (int)e; }
    const int64_t n = site_hi - site_lo;
    if (n <= 0) return 0;
    int32_t* bs = nullptr;
    e = hipMalloc(&bs, (size_t)n * 4);
    if (e != hipSuccess) return (int)e;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_block_starts, dim3(blocks), dim3(256), 0, 0, stream_key(seed, S_BLOCK), site_lo, n, bs);
    for (int s = 0; s < n_samples; s++) {
        const uint64_t st = S_SAMPLE0 + 4ull * (uint64_t)(sample_first + s);
        uint8_t* row = reinterpret_cast<uint8_t*>(d_base) + (int64_t)(sample_first + s) * pitch;
        hipLaunchKernelGGL(k_fill_sample, dim3(blocks), dim3(256), 0, 0, stream_key(seed, S_LEVEL), stream_key(seed, st + 0),
                           stream_key(seed, st + 1), stream_key(seed, st + 2), site_lo, n, bs, row);
    }
    e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipGetLastError();
    (void)hipFree(bs);
    return (int)e;
}

// Sites 0..n_sites-1 on the current device (`scratch` is ignored; kept for the callers of the first version).
int wgbssynth_fill_betas(void* d_base, int64_t pitch, int64_t n_sites, int sample_first, int n_samples, uint64_t seed, void* scratch)
{
    (void)scratch;
    return wgbssynth_fill_betas_range(d_base, pitch, 0, n_sites, sample_first, n_samples, seed, -1);
}

}  // extern "C"
