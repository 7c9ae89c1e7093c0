// How fast can 1.8 GB be read once on this GPU, in the launch geometries the scan pass could use?
//   v0  grid-stride, 16 B per lane, 4 loads in flight per lane (the "float4 copy" style reference point)
//   v1  one wavefront per 117 KB row (k_scan's geometry: 483 chunks x 32 samples), 3 loads in flight per lane
//   v2  the same rows, 6 loads in flight per lane
//   v3  one wavefront per row, 32 B per lane per iteration (two adjacent vectors), 3 iterations in flight
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

__global__ __launch_bounds__(256) void v0(const uint4* __restrict__ p, size_t n16, uint32_t* out)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc ^= fold(a) ^ fold(b) ^ fold(c) ^ fold(d);
    }
    for (; i < n16; i += stride) acc ^= fold(p[i]);
    if (acc == 0x12345678u) out[0] = acc;
}

template <int DEPTH>
__global__ __launch_bounds__(256) void v12(const uint8_t* __restrict__ base, size_t pitch, int rows_per_sample, int row_bytes, uint32_t* out)
{
    const int lane = threadIdx.x & 63;
    const long rowid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int c = (int)(rowid / 32), s = (int)(rowid % 32);
    if (c >= rows_per_sample) return;
    const uint4* p = reinterpret_cast<const uint4*>(base + (size_t)s * pitch + (size_t)c * row_bytes);
    const int n = row_bytes / 16;
    uint32_t acc = 0;
    uint4 q[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) q[d] = (lane + 64 * d < n) ? p[lane + 64 * d] : make_uint4(0, 0, 0, 0);
    for (int i = lane; i < n; i += 64 * DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            acc ^= fold(q[d]);
            const int j = i + 64 * (d + DEPTH);
            q[d] = (j < n) ? p[j] : make_uint4(0, 0, 0, 0);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void v3(const uint8_t* __restrict__ base, size_t pitch, int rows_per_sample, int row_bytes, uint32_t* out)
{
    const int lane = threadIdx.x & 63;
    const long rowid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int c = (int)(rowid / 32), s = (int)(rowid % 32);
    if (c >= rows_per_sample) return;
    const uint4* p = reinterpret_cast<const uint4*>(base + (size_t)s * pitch + (size_t)c * row_bytes);
    const int n = row_bytes / 16;
    uint32_t acc = 0;
    uint4 q[3][2];
#pragma unroll
    for (int d = 0; d < 3; d++)
        for (int h = 0; h < 2; h++) { const int j = 2 * lane + h + 128 * d; q[d][h] = (j < n) ? p[j] : make_uint4(0, 0, 0, 0); }
    for (int i = 2 * lane; i < n; i += 128 * 3) {
#pragma unroll
        for (int d = 0; d < 3; d++) {
            acc ^= fold(q[d][0]) ^ fold(q[d][1]);
            for (int h = 0; h < 2; h++) { const int j = i + h + 128 * (d + 3); q[d][h] = (j < n) ? p[j] : make_uint4(0, 0, 0, 0); }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

#include "../../wgbs_tools_amd/csrc/wave_prims.h"
//   v4  v3's loads (one iteration ahead, 2 x 16 B per lane) + SWAR sums + two DPP wave scans + readlane running totals
//   v5  v4 + the carry store of every 4th lane (8 B per 64 sites)
template <int STORE>
__global__ __launch_bounds__(256) void v45(const uint8_t* __restrict__ base, size_t pitch, int rows_per_sample, int row_bytes, uint2* carry, uint32_t* out)
{
    const int lane = threadIdx.x & 63;
    const long rowid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int c = (int)(rowid / 32), s = (int)(rowid % 32);
    if (c >= rows_per_sample) return;
    const uint4* p = reinterpret_cast<const uint4*>(base + (size_t)s * pitch + (size_t)c * row_bytes);
    uint2* cr = carry + rowid * (row_bytes / 128 + 1);
    const int n = row_bytes / 16;
    uint32_t run_m = 0, run_t = 0;
    int vi = 2 * lane;
    uint4 c0 = p[vi < n - 1 ? vi : n - 1], c1 = p[vi + 1 < n - 1 ? vi + 1 : n - 1];
    for (int b = 0; b < n; b += 128) {
        const int vn = vi + 128;
        const uint4 m0 = p[vn < n - 1 ? vn : n - 1], m1 = p[vn + 1 < n - 1 ? vn + 1 : n - 1];
        const uint32_t w[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        uint32_t sm = 0, sc = 0;
#pragma unroll
        for (int d = 0; d < 8; d++) { sm += w[d] & 0x00ff00ffu; sc += (w[d] >> 8) & 0x00ff00ffu; }
        const uint32_t tm = (sm & 0xffffu) + (sm >> 16), tt = (sc & 0xffffu) + (sc >> 16);
        const uint32_t im = wg_wave_incl_scan_dpp_u32(tm), it = wg_wave_incl_scan_dpp_u32(tt);
        if (STORE == 1 && (lane & 3) == 0) cr[(b >> 3) + (lane >> 2)] = make_uint2(run_m + im - tm, run_t + it - tt);
        if (STORE == 2 && (lane & 3) == 0) __builtin_nontemporal_store((unsigned long long)(run_m + im - tm) | ((unsigned long long)(run_t + it - tt) << 32), reinterpret_cast<unsigned long long*>(cr + (b >> 3) + (lane >> 2)));
        if (STORE == 3 && (lane & 3) == 0) carry[((rowid * 64 + (b >> 3) + (lane >> 2)) & 0x1ffff)] = make_uint2(run_m + im - tm, run_t + it - tt);
        run_m += (uint32_t)__builtin_amdgcn_readlane((int)im, 63);
        run_t += (uint32_t)__builtin_amdgcn_readlane((int)it, 63);
        c0 = m0; c1 = m1; vi = vn;
    }
    if (run_m + run_t == 0x12345678u) out[0] = run_m;
}

//   v6  v4 + carries staged in LDS and flushed as full-wave 8-byte stores every 4 iterations
__global__ __launch_bounds__(256) void v6(const uint8_t* __restrict__ base, size_t pitch, int rows_per_sample, int row_bytes, uint2* carry, uint32_t* out)
{
    __shared__ uint2 stage[4][128];
    const int lane = threadIdx.x & 63;
    uint2* stg = stage[threadIdx.x >> 6];
    const long rowid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int c = (int)(rowid / 32), s = (int)(rowid % 32);
    if (c >= rows_per_sample) return;
    const uint4* p = reinterpret_cast<const uint4*>(base + (size_t)s * pitch + (size_t)c * row_bytes);
    uint2* cr = carry + rowid * (row_bytes / 128 + 1);
    const int n = row_bytes / 16, nG = row_bytes / 128;
    uint32_t run_m = 0, run_t = 0;
    int vi = 2 * lane, gfl = -1;
    uint4 c0 = p[vi < n - 1 ? vi : n - 1], c1 = p[vi + 1 < n - 1 ? vi + 1 : n - 1];
    for (int b = 0; b < n; b += 128) {
        const int vn = vi + 128;
        const uint4 m0 = p[vn < n - 1 ? vn : n - 1], m1 = p[vn + 1 < n - 1 ? vn + 1 : n - 1];
        const uint32_t w[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        uint32_t sm = 0, sc = 0;
#pragma unroll
        for (int d = 0; d < 8; d++) { sm += w[d] & 0x00ff00ffu; sc += (w[d] >> 8) & 0x00ff00ffu; }
        const uint32_t tm = (sm & 0xffffu) + (sm >> 16), tt = (sc & 0xffffu) + (sc >> 16);
        const uint32_t im = wg_wave_incl_scan_dpp_u32(tm), it = wg_wave_incl_scan_dpp_u32(tt);
        const int g = (b >> 3) + (lane >> 2);
        if ((lane & 3) == 0) stg[g & 127] = make_uint2(run_m + im - tm, run_t + it - tt);
        const int gtop = (b >> 3) + 15;
        if (gtop - gfl >= 64 || b + 128 >= n) {
            for (int q = gfl + 1 + lane; q <= gtop && q < nG; q += 64) cr[q] = stg[q & 127];
            gfl = gtop;
        }
        run_m += (uint32_t)__builtin_amdgcn_readlane((int)im, 63);
        run_t += (uint32_t)__builtin_amdgcn_readlane((int)it, 63);
        c0 = m0; c1 = m1; vi = vn;
    }
    if (run_m + run_t == 0x12345678u) out[0] = run_m;
}

int main()
{
    const size_t n_sites = 28217448, pitch = ((2 * n_sites + 255) / 256) * 256 + 256;
    const int N = 32, chunks = 483, row_bytes = 60000 * 2 / 16 * 16;
    uint8_t* d; uint32_t* out;
    hipMalloc(&d, pitch * N); hipMalloc(&out, 4);
    hipMemset(d, 1, pitch * N);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes_rows = (double)chunks * N * row_bytes;
    uint2* carry; hipMalloc(&carry, (size_t)chunks * N * (row_bytes / 128 + 1) * 8);
    for (int v = 0; v < 9; v++) {
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0);
            double bytes = bytes_rows;
            if (v == 0) { bytes = (double)pitch * N; hipLaunchKernelGGL(v0, 256 * 8, 256, 0, 0, reinterpret_cast<const uint4*>(d), pitch * N / 16, out); }
            if (v == 1) hipLaunchKernelGGL(v12<3>, (chunks * N + 3) / 4, 256, 0, 0, d, pitch, chunks, row_bytes, out);
            if (v == 2) hipLaunchKernelGGL(v12<6>, (chunks * N + 3) / 4, 256, 0, 0, d, pitch, chunks, row_bytes, out);
            if (v == 3) hipLaunchKernelGGL(v3, (chunks * N + 3) / 4, 256, 0, 0, d, pitch, chunks, row_bytes, out);
            if (v == 4) hipLaunchKernelGGL(v45<0>, (chunks * N + 3) / 4, 256, 0, 0, d, pitch, chunks, row_bytes, carry, out);
            if (v == 7) hipLaunchKernelGGL(v45<2>, (chunks * N + 3) / 4, 256, 0, 0, d, pitch, chunks, row_bytes, carry, out);
            if (v == 8) hipLaunchKernelGGL(v45<3>, (chunks * N + 3) / 4, 256, 0, 0, d, pitch, chunks, row_bytes, carry, out);
            if (v == 6) hipLaunchKernelGGL(v6, (chunks * N + 3) / 4, 256, 0, 0, d, pitch, chunks, row_bytes, carry, out);
            if (v == 5) hipLaunchKernelGGL(v45<1>, (chunks * N + 3) / 4, 256, 0, 0, d, pitch, chunks, row_bytes, carry, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("v%d: %.3f ms  %.2f TB/s\n", v, ms, bytes / ms / 1e9);
        }
    }
    return 0;
}
