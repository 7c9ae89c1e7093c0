// wave_prims.h — wave64 scan/reduce primitives for gfx950 built on DPP (data-parallel primitives:
// cross-lane operand routing inside the VALU, no LDS round trip).  Every function must be called with all
// 64 lanes of the wave active.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WG_WAVE 64

// dpp_ctrl encodings (gfx9): quad_perm = 0x00..0xFF, row_shr:n = 0x110+n, row_mirror = 0x140,
// row_half_mirror = 0x141, row_bcast:15 = 0x142, row_bcast:31 = 0x143
#define WG_DPP_QUAD_1032   0xB1
#define WG_DPP_QUAD_2301   0x4E
#define WG_DPP_ROW_HMIRROR 0x141
#define WG_DPP_ROW_MIRROR  0x140
#define WG_DPP_ROW_SHR(n)  (0x110 + (n))
#define WG_DPP_BCAST15     0x142
#define WG_DPP_BCAST31     0x143

template <int CTRL>
__device__ __forceinline__ uint32_t wg_dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}

template <int CTRL>
__device__ __forceinline__ double wg_dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wg_readlane_f64(double v, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wg_sel_max(double a, double b) { return b > a ? b : a; }

// Maximum of a double over the wave, returned in every lane (wave-uniform).  No NaNs expected.
// 4 butterfly steps inside each row of 16 lanes, then the 4 row results are combined through SGPRs.
__device__ __forceinline__ double wg_wave_max_f64(double v)
{
    v = wg_sel_max(v, wg_dpp_f64<WG_DPP_QUAD_1032>(v));
    v = wg_sel_max(v, wg_dpp_f64<WG_DPP_QUAD_2301>(v));
    v = wg_sel_max(v, wg_dpp_f64<WG_DPP_ROW_HMIRROR>(v));
    v = wg_sel_max(v, wg_dpp_f64<WG_DPP_ROW_MIRROR>(v));
    double r0 = wg_readlane_f64(v, 0), r1 = wg_readlane_f64(v, 16);
    double r2 = wg_readlane_f64(v, 32), r3 = wg_readlane_f64(v, 48);
    return wg_sel_max(wg_sel_max(r0, r1), wg_sel_max(r2, r3));
}

__device__ __forceinline__ uint32_t wg_umin(uint32_t a, uint32_t b) { return b < a ? b : a; }

__device__ __forceinline__ uint32_t wg_wave_min_u32(uint32_t v)
{
    v = wg_umin(v, wg_dpp_u32<WG_DPP_QUAD_1032>(v));
    v = wg_umin(v, wg_dpp_u32<WG_DPP_QUAD_2301>(v));
    v = wg_umin(v, wg_dpp_u32<WG_DPP_ROW_HMIRROR>(v));
    v = wg_umin(v, wg_dpp_u32<WG_DPP_ROW_MIRROR>(v));
    uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return wg_umin(wg_umin(r0, r1), wg_umin(r2, r3));
}

__device__ __forceinline__ uint32_t wg_wave_max_u32(uint32_t v)
{
    return ~wg_wave_min_u32(~v);
}

// Inclusive prefix sum over the 64 lanes in 6 DPP adds: Kogge-Stone inside each row of 16 lanes
// (row_shr 1,2,4,8; lanes whose source falls outside the row add 0), then row_bcast:15 adds the total of
// rows 0/2 to rows 1/3 and row_bcast:31 adds lane 31 (= rows 0+1) to rows 2 and 3.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t wg_dpp_or0_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}

__device__ __forceinline__ uint32_t wg_wave_incl_scan_dpp_u32(uint32_t v)
{
    v += wg_dpp_or0_u32<WG_DPP_ROW_SHR(1), 0xf>(v);
    v += wg_dpp_or0_u32<WG_DPP_ROW_SHR(2), 0xf>(v);
    v += wg_dpp_or0_u32<WG_DPP_ROW_SHR(4), 0xf>(v);
    v += wg_dpp_or0_u32<WG_DPP_ROW_SHR(8), 0xf>(v);
    v += wg_dpp_or0_u32<WG_DPP_BCAST15, 0xa>(v);
    v += wg_dpp_or0_u32<WG_DPP_BCAST31, 0xc>(v);
    return v;
}

// The same within each 32-lane half of the wavefront (two independent scans): the last broadcast step is left out.
__device__ __forceinline__ uint32_t wg_half_incl_scan_dpp_u32(uint32_t v)
{
    v += wg_dpp_or0_u32<WG_DPP_ROW_SHR(1), 0xf>(v);
    v += wg_dpp_or0_u32<WG_DPP_ROW_SHR(2), 0xf>(v);
    v += wg_dpp_or0_u32<WG_DPP_ROW_SHR(4), 0xf>(v);
    v += wg_dpp_or0_u32<WG_DPP_ROW_SHR(8), 0xf>(v);
    v += wg_dpp_or0_u32<WG_DPP_BCAST15, 0xa>(v);
    return v;
}

// Inclusive prefix sum over the 64 lanes (uint32, wrap-around).  Kogge-Stone with ds_bpermute-free shuffles:
// __shfl_up is used here (LDS-crossbar permute); this primitive sits in bandwidth/latency-tolerant kernels.
__device__ __forceinline__ uint32_t wg_wave_incl_scan_u32(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = (uint32_t)__shfl_up((int)v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

__device__ __forceinline__ uint64_t wg_wave_incl_scan_u64(uint64_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, d, 64);
        uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d, 64);
        if (lane >= d) v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

__device__ __forceinline__ uint32_t wg_wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
    return v;
}

__device__ __forceinline__ uint64_t wg_wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, d, 64);
        uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}
