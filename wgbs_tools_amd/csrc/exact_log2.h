// exact_log2.h — bit-exact restatements of the two libm functions on the reference's hot path.
//
// The reference's per-(block,sample) term calls the platform libm twice (segmentor.cpp:130 `log2f`,
// segmentor.cpp:133 `log2`); block boundaries are a byte-identity contract (segmentor.cpp:76-79), so the
// device must reproduce glibc 2.35's results bit for bit.  Both functions are table-driven polynomial
// evaluations in IEEE binary64; evaluated with the same operations in the same order (NO fused
// multiply-add: this file must be compiled with -ffp-contract=off) they give identical bits on any IEEE
// machine.  Domain used by the path: log2f(p) for float p in (0,1];  log2(1.0-(double)p) for float p in (0,1).
// tests/test_exact_log2.py compares these restatements (host build AND gfx950 build) with the live libm over
// that whole domain (oracle/libm_probe.c).
//
// Table provenance: glibc 2.35 sysdeps/ieee754/flt-32/e_log2f_data.c (__log2f_data, N=16, poly order 4) and
// sysdeps/ieee754/dbl-64/e_log2_data.c (__log2_data, N=64, non-FMA variant with tab2), as read out of this
// image's /lib/x86_64-linux-gnu/libm.so.6.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define WG_HD __host__ __device__ __forceinline__
#else
#define WG_HD static inline
#endif

// ---- tables as initialiser lists (instantiated once for host, once in device constant/LDS memory) ----
#define WG_LOG2F_TAB { \
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2}, \
    {0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2}, {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2}, \
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3}, \
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4}, \
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1.0000000000000p+0,  0x0.0p+0}, \
    {0x1.e608cfd9a47acp-1,  0x1.338ca9f24f53dp-4}, {0x1.ca4b31f026aa0p-1,  0x1.476a9543891bap-3}, \
    {0x1.b2036576afce6p-1,  0x1.e840b4ac4e4d2p-3}, {0x1.9c2d163a1aa2dp-1,  0x1.40645f0c6651cp-2}, \
    {0x1.886e6037841edp-1,  0x1.88e9c2c1b9ff8p-2}, {0x1.767dcf5534862p-1,  0x1.ce0a44eb17bccp-2} }
#define WG_LOG2F_A0 (-0x1.712b6f70a7e4dp-2)
#define WG_LOG2F_A1 ( 0x1.ecabf496832e0p-2)
#define WG_LOG2F_A2 (-0x1.715479ffae3dep-1)
#define WG_LOG2F_A3 ( 0x1.715475f35c8b8p+0)

#define WG_LOG2_INVLN2HI 0x1.7154765200000p+0
#define WG_LOG2_INVLN2LO 0x1.705fc2eefa200p-33
#define WG_LOG2_A0 (-0x1.71547652b8339p-1)
#define WG_LOG2_A1 ( 0x1.ec709dc3a04bep-2)
#define WG_LOG2_A2 (-0x1.7154764702ffbp-2)
#define WG_LOG2_A3 ( 0x1.2776c50034c48p-2)
#define WG_LOG2_A4 (-0x1.ec7b328ea92bcp-3)
#define WG_LOG2_A5 ( 0x1.a6225e117f92ep-3)
#define WG_LOG2_B0 (-0x1.71547652b82fep-1)
#define WG_LOG2_B1 ( 0x1.ec709dc3a03f7p-2)
#define WG_LOG2_B2 (-0x1.71547652b7c3fp-2)
#define WG_LOG2_B3 ( 0x1.2776c50f05be4p-2)
#define WG_LOG2_B4 (-0x1.ec709dd768fe5p-3)
#define WG_LOG2_B5 ( 0x1.a61761ec4e736p-3)
#define WG_LOG2_B6 (-0x1.7153fbc64a79bp-3)
#define WG_LOG2_B7 ( 0x1.484d154f01b4ap-3)
#define WG_LOG2_B8 (-0x1.289e4a72c383cp-3)
#define WG_LOG2_B9 ( 0x1.0b32f285aee66p-3)

// {invc, logc}
#define WG_LOG2_TAB { \
    {0x1.724286bb1acf8p+0,-0x1.1095feecdb000p-1}, {0x1.6e1f766d2cca1p+0,-0x1.08494bd76d000p-1}, \
    {0x1.6a13d0e30d48ap+0,-0x1.00143aee8f800p-1}, {0x1.661ec32d06c85p+0,-0x1.efec5360b4000p-2}, \
    {0x1.623fa951198f8p+0,-0x1.dfdd91ab7e000p-2}, {0x1.5e75ba4cf026cp+0,-0x1.cffae0cc79000p-2}, \
    {0x1.5ac055a214fb8p+0,-0x1.c043811fda000p-2}, {0x1.571ed0f166e1ep+0,-0x1.b0b67323ae000p-2}, \
    {0x1.53909590bf835p+0,-0x1.a152f5a2db000p-2}, {0x1.5014fed61adddp+0,-0x1.9217f5af86000p-2}, \
    {0x1.4cab88e487bd0p+0,-0x1.8304db0719000p-2}, {0x1.49539b4334feep+0,-0x1.74189f9a9e000p-2}, \
    {0x1.460cbdfafd569p+0,-0x1.6552bb5199000p-2}, {0x1.42d664ee4b953p+0,-0x1.56b23a29b1000p-2}, \
    {0x1.3fb01111dd8a6p+0,-0x1.483650f5fa000p-2}, {0x1.3c995b70c5836p+0,-0x1.39de937f6a000p-2}, \
    {0x1.3991c4ab6fd4ap+0,-0x1.2baa1538d6000p-2}, {0x1.3698e0ce099b5p+0,-0x1.1d98340ca4000p-2}, \
    {0x1.33ae48213e7b2p+0,-0x1.0fa853a40e000p-2}, {0x1.30d191985bdb1p+0,-0x1.01d9c32e73000p-2}, \
    {0x1.2e025cab271d7p+0,-0x1.e857da2fa6000p-3}, {0x1.2b404cf13cd82p+0,-0x1.cd3c8633d8000p-3}, \
    {0x1.288b02c7ccb50p+0,-0x1.b26034c14a000p-3}, {0x1.25e2263944de5p+0,-0x1.97c1c2f4fe000p-3}, \
    {0x1.234563d8615b1p+0,-0x1.7d6023f800000p-3}, {0x1.20b46e33eaf38p+0,-0x1.633a71a05e000p-3}, \
    {0x1.1e2eefdcda3ddp+0,-0x1.494f5e9570000p-3}, {0x1.1bb4a580b3930p+0,-0x1.2f9e424e0a000p-3}, \
    {0x1.19453847f2200p+0,-0x1.162595afdc000p-3}, {0x1.16e06c0d5d73cp+0,-0x1.f9c9a75bd8000p-4}, \
    {0x1.1485f47b7e4c2p+0,-0x1.c7b575bf9c000p-4}, {0x1.12358ad0085d1p+0,-0x1.960c60ff48000p-4}, \
    {0x1.0fef00f532227p+0,-0x1.64ce247b60000p-4}, {0x1.0db2077d03a8fp+0,-0x1.33f78b2014000p-4}, \
    {0x1.0b7e6d65980d9p+0,-0x1.0387d1a42c000p-4}, {0x1.0953efe7b408dp+0,-0x1.a6f9208b50000p-5}, \
    {0x1.07325cac53b83p+0,-0x1.47a954f770000p-5}, {0x1.05197e40d1b5cp+0,-0x1.d23a8c50c0000p-6}, \
    {0x1.03091c1208ea2p+0,-0x1.16a2629780000p-6}, {0x1.0101025b37e21p+0,-0x1.720f8d8e80000p-8}, \
    {0x1.fc07ef9caa76bp-1, 0x1.6fe53b1500000p-7}, {0x1.f4465d3f6f184p-1, 0x1.11ccce10f8000p-5}, \
    {0x1.ecc079f84107fp-1, 0x1.c4dfc8c8b8000p-5}, {0x1.e573a99975ae8p-1, 0x1.3aa321e574000p-4}, \
    {0x1.de5d6f0bd3de6p-1, 0x1.918a0d08b8000p-4}, {0x1.d77b681ff38b3p-1, 0x1.e72e9da044000p-4}, \
    {0x1.d0cb5724de943p-1, 0x1.1dcd2507f6000p-3}, {0x1.ca4b2dc0e7563p-1, 0x1.476ab03dea000p-3}, \
    {0x1.c3f8ee8d6cb51p-1, 0x1.7074377e22000p-3}, {0x1.bdd2b4f020c4cp-1, 0x1.98ede8ba94000p-3}, \
    {0x1.b7d6c006015cap-1, 0x1.c0db86ad2e000p-3}, {0x1.b20366e2e338fp-1, 0x1.e840aafcee000p-3}, \
    {0x1.ac57026295039p-1, 0x1.0790ab4678000p-2}, {0x1.a6d01bc2731ddp-1, 0x1.1ac056801c000p-2}, \
    {0x1.a16d3bc3ff18bp-1, 0x1.2db11d4fee000p-2}, {0x1.9c2d14967feadp-1, 0x1.406464ec58000p-2}, \
    {0x1.970e4f47c9902p-1, 0x1.52dbe093af000p-2}, {0x1.920fb3982bcf2p-1, 0x1.651902050d000p-2}, \
    {0x1.8d30187f759f1p-1, 0x1.771d2cdeaf000p-2}, {0x1.886e5ebb9f66dp-1, 0x1.88e9c857d9000p-2}, \
    {0x1.83c97b658b994p-1, 0x1.9a80155e16000p-2}, {0x1.7f405ffc61022p-1, 0x1.abe186ed3d000p-2}, \
    {0x1.7ad22181415cap-1, 0x1.bd0f2aea0e000p-2}, {0x1.767dcf99eff8cp-1, 0x1.ce0a43dbf4000p-2} }

// {chi, clo}
#define WG_LOG2_TAB2 { \
    {0x1.6200012b90a8ep-1, 0x1.904ab0644b605p-55}, {0x1.66000045734a6p-1, 0x1.1ff9bea62f7a9p-57}, \
    {0x1.69fffc325f2c5p-1, 0x1.27ecfcb3c90bap-55}, {0x1.6e00038b95a04p-1, 0x1.8ff8856739326p-55}, \
    {0x1.71fffe09994e3p-1, 0x1.afd40275f82b1p-55}, {0x1.7600015590e10p-1,-0x1.2fd75b4238341p-56}, \
    {0x1.7a00012655bd5p-1, 0x1.808e67c242b76p-56}, {0x1.7e0003259e9a6p-1,-0x1.208e426f622b7p-57}, \
    {0x1.81fffedb4b2d2p-1,-0x1.402461ea5c92fp-55}, {0x1.860002dfafcc3p-1, 0x1.df7f4a2f29a1fp-57}, \
    {0x1.89ffff78c6b50p-1,-0x1.e0453094995fdp-55}, {0x1.8e00039671566p-1,-0x1.a04f3bec77b45p-55}, \
    {0x1.91fffe2bf1745p-1,-0x1.7fa34400e203cp-56}, {0x1.95fffcc5c9fd1p-1,-0x1.6ff8005a0695dp-56}, \
    {0x1.9a0003bba4767p-1, 0x1.0f8c4c4ec7e03p-56}, {0x1.9dfffe7b92da5p-1, 0x1.e7fd9478c4602p-55}, \
    {0x1.a1fffd72efdafp-1,-0x1.a0c554dcdae7ep-57}, {0x1.a5fffde04ff95p-1, 0x1.67da98ce9b26bp-55}, \
    {0x1.a9fffca5e8d2bp-1,-0x1.284c9b54c13dep-55}, {0x1.adfffddad03eap-1, 0x1.812c8ea602e3cp-58}, \
    {0x1.b1ffff10d3d4dp-1,-0x1.efaddad27789cp-55}, {0x1.b5fffce21165ap-1, 0x1.3cb1719c61237p-58}, \
    {0x1.b9fffd950e674p-1, 0x1.3f7d94194ce00p-56}, {0x1.be000139ca8afp-1, 0x1.50ac4215d9bc0p-56}, \
    {0x1.c20005b46df99p-1, 0x1.beea653e9c1c9p-57}, {0x1.c600040b9f7aep-1,-0x1.c079f274a70d6p-56}, \
    {0x1.ca0006255fd8ap-1,-0x1.a0b4076e84c1fp-56}, {0x1.cdfffd94c095dp-1, 0x1.8f933f99ab5d7p-55}, \
    {0x1.d1ffff975d6cfp-1,-0x1.82c08665fe1bep-58}, {0x1.d5fffa2561c93p-1,-0x1.b04289bd295f3p-56}, \
    {0x1.d9fff9d228b0cp-1, 0x1.70251340fa236p-55}, {0x1.de00065bc7e16p-1,-0x1.5011e16a4d80cp-56}, \
    {0x1.e200002f64791p-1, 0x1.9802f09ef62e0p-55}, {0x1.e600057d7a6d8p-1,-0x1.e0b75580cf7fap-56}, \
    {0x1.ea00027edc00cp-1,-0x1.c848309459811p-55}, {0x1.ee0006cf5cb7cp-1,-0x1.f8027951576f4p-55}, \
    {0x1.f2000782b7dccp-1,-0x1.f81d97274538fp-55}, {0x1.f6000260c450ap-1,-0x1.071002727ffdcp-59}, \
    {0x1.f9fffe88cd533p-1,-0x1.81bdce1fda8b0p-58}, {0x1.fdfffd50f8689p-1, 0x1.7f91acb918e6ep-55}, \
    {0x1.0200004292367p+0, 0x1.b7ff365324681p-54}, {0x1.05fffe3e3d668p+0, 0x1.6fa08ddae957bp-55}, \
    {0x1.0a0000a85a757p+0,-0x1.7e2de80d3fb91p-58}, {0x1.0e0001a5f3fccp+0,-0x1.1823305c5f014p-54}, \
    {0x1.11ffff8afbaf5p+0,-0x1.bfabb6680bac2p-55}, {0x1.15fffe54d91adp+0,-0x1.d7f121737e7efp-54}, \
    {0x1.1a00011ac36e1p+0, 0x1.c000a0516f5ffp-54}, {0x1.1e00019c84248p+0,-0x1.082fbe4da5da0p-54}, \
    {0x1.220000ffe5e6ep+0,-0x1.8fdd04c9cfb43p-55}, {0x1.26000269fd891p+0, 0x1.cfe2a7994d182p-55}, \
    {0x1.2a00029a6e6dap+0,-0x1.00273715e8bc5p-56}, {0x1.2dfffe0293e39p+0, 0x1.b7c39dab2a6f9p-54}, \
    {0x1.31ffff7dcf082p+0, 0x1.df1336edc5254p-56}, {0x1.35ffff05a8b60p+0,-0x1.e03564ccd31ebp-54}, \
    {0x1.3a0002e0eaeccp+0, 0x1.5f0e74bd3a477p-56}, {0x1.3e000043bb236p+0, 0x1.c7dcb149d8833p-54}, \
    {0x1.4200002d187ffp+0, 0x1.e08afcf2d3d28p-56}, {0x1.460000d387cb1p+0, 0x1.20837856599a6p-55}, \
    {0x1.4a00004569f89p+0,-0x1.9fa5c904fbcd2p-55}, {0x1.4e000043543f3p+0,-0x1.81125ed175329p-56}, \
    {0x1.51fffcc027f0fp+0, 0x1.883d8847754dcp-54}, {0x1.55ffffd87b36fp+0,-0x1.709e731d02807p-55}, \
    {0x1.59ffff21df7bap+0, 0x1.7f79f68727b02p-55}, {0x1.5dfffebfc3481p+0,-0x1.180902e30e93ep-54} }

struct wg_d2 { double a, b; };

// All tables in one POD so that a kernel can copy it into LDS with one loop (2304 bytes).
struct wg_log_tables {
    wg_d2 f_tab[16];     // log2f {invc, logc}
    wg_d2 d_tab[64];     // log2  {invc, logc}
    wg_d2 d_tab2[64];    // log2  {chi, clo}
    wg_d2 d_fast[64];    // d_tab with entry WG_FAST_CENTRE_ENTRY replaced by {1, 0}: filled by wg_tables_finish()
};
#define WG_LOG_TABLES_INIT { WG_LOG2F_TAB, WG_LOG2_TAB, WG_LOG2_TAB2, WG_LOG2_TAB }
// What the scoring kernels of the general fast form (pseudo count 0 or in [2^-20, 4)) keep in LDS: the log2f table and the
// fast-log2 table (1.3 KB).  The exact-log2 tables are only needed by the rare fallback and stay in global/constant memory.
struct wg_fast_tables {
    wg_d2 f_tab[16];     // log2f {invc, logc}
    wg_d2 d_fast[64];    // d_tab with entry WG_FAST_CENTRE_ENTRY replaced by {1, 0}
};
// With a pseudo count >= 1 (guard-free form) the scoring kernels use per-(k, i) tables for both logs instead
// (wg_log2f_ks / wg_fast_log2_ks below): p and 1 - p are >= pc / (ntotal + 2 pc), so with blocks of at most 60 sites
// (narrow tiles) the exponent k lies in [-14, 0], with the ABI's longest blocks (255 * 8000) in [-21, 0]: at most
// WG_KY_KMIN + 1 rows; how many a pseudo count and a longest block really need is wg_lookup_rows() (11 for the default
// 15 and narrow tiles).  The kernels size both tables to that.
#define WG_KY_KMIN 23
// The fast log2 uses d_tab with ONE entry changed: interval 39 = [0.9921875, 1) gets the centre exactly 1
// (invc = 1, logc = 0), so that arguments just below 1 need no separate cancellation-free branch.
// (Interval 40 = [1, 1.015625) cannot be treated the same way: it also serves z = 2^-k x for x in [0.5, 0.5078) etc.)
#define WG_FAST_CENTRE_ENTRY 39

WG_HD uint32_t wg_f2u(float f)   { uint32_t u; memcpy(&u, &f, 4); return u; }
WG_HD float    wg_u2f(uint32_t u) { float f;    memcpy(&f, &u, 4); return f; }
WG_HD uint64_t wg_d2u(double d)  { uint64_t u; memcpy(&u, &d, 8); return u; }
WG_HD double   wg_u2d(uint64_t u) { double d;   memcpy(&d, &u, 8); return d; }

#define WG_FMA(a, b, c) __builtin_fma((a), (b), (c))     // one IEEE fused multiply-add (v_fma_f64 on the device)

// a*b + K with K a compile-time constant (a polynomial coefficient).  Same IEEE operation as WG_FMA; on the device it
// is spelled as the 3-operand VOP3 form with K in a scalar register pair, because hipcc otherwise emits
// v_mov_b64 tmp, K ; v_fmac_f64 tmp, a, b  for every Horner step (the fmac form overwrites its addend).
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ double wg_fma_k(double a, double b, double k)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(k));
    return d;
}
#define WG_FMA_K(a, b, k) wg_fma_k((a), (b), (k))
#else
#define WG_FMA_K(a, b, k) __builtin_fma((a), (b), (k))
#endif

// glibc 2.35 __log2f (sysdeps/ieee754/flt-32/e_log2f.c), positive finite inputs.  x == 1 -> +0.
// Evaluated without any fused multiply-add, operation by operation as the generic C source reads.
WG_HD float wg_log2f_nofma(float x, const wg_d2* __restrict__ ftab)
{
    uint32_t ix = wg_f2u(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix < 0x00800000u) {                       // subnormal: normalise (x * 2^23, exponent - 23)
        ix = wg_f2u(x * 0x1p23f);
        ix -= 23u << 23;
    }
    uint32_t tmp = ix - 0x3f330000u;
    uint32_t i = (tmp >> 19) & 15u;
    uint32_t top = tmp & 0xff800000u;
    uint32_t iz = ix - top;
    int32_t k = (int32_t)tmp >> 23;
    double invc = ftab[i].a, logc = ftab[i].b;
    double z = (double)wg_u2f(iz);
    double r = z * invc - 1;
    double y0 = logc + (double)k;
    double r2 = r * r;
    double y = WG_LOG2F_A1 * r + WG_LOG2F_A2;
    y = WG_LOG2F_A0 * r2 + y;
    double p = WG_LOG2F_A3 * r + y0;
    y = y * r2 + p;
    return (float)y;
}

// The same evaluation with every multiply-add fused (what glibc's FMA ifunc variant computes).  The double result
// differs from the unfused one in its last bits, the FLOAT result never does: both forms are compared with the live
// libm over every float in (0, 1] (and every subnormal) by tests/test_exact_log2_cpu.py — 0 mismatches each.
// 7 fp64 operations instead of 12: this is the form the kernels use.
WG_HD float wg_log2f(float x, const wg_d2* __restrict__ ftab)
{
    // x == 1 needs no special case here: its table entry is {1, 0}, r = 0, and the polynomial returns +0 exactly.
    uint32_t ix = wg_f2u(x);
    if (ix < 0x00800000u) return wg_log2f_nofma(x, ftab);       // subnormal p: never on real data, kept exact
    uint32_t tmp = ix - 0x3f330000u;
    uint32_t i = (tmp >> 19) & 15u;
    uint32_t top = tmp & 0xff800000u;
    uint32_t iz = ix - top;
    int32_t k = (int32_t)tmp >> 23;
    double invc = ftab[i].a, logc = ftab[i].b;
    double z = (double)wg_u2f(iz);
    double r = WG_FMA(z, invc, -1.0);
    double y0 = logc + (double)k;
    double r2 = r * r;
    double y = WG_FMA(WG_LOG2F_A1, r, WG_LOG2F_A2);
    y = WG_FMA(WG_LOG2F_A0, r2, y);
    double p = WG_FMA(WG_LOG2F_A3, r, y0);
    y = WG_FMA(y, r2, p);
    return (float)y;
}

// Same, for callers that guarantee a NORMAL x (the scoring kernel in fast mode: p >= 2^-85, see WG_FAST_MIN_PC).
WG_HD float wg_log2f_normal(float x, const wg_d2* __restrict__ ftab)
{
    const uint32_t ix = wg_f2u(x);
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const int32_t k = (int32_t)tmp >> 23;
    const double invc = ftab[i].a, logc = ftab[i].b;
    const double r = WG_FMA_K((double)wg_u2f(iz), invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double y = WG_FMA_K(r, WG_LOG2F_A1, WG_LOG2F_A2);          // (same products as A1*r + A2 etc.: multiplication commutes exactly)
    y = WG_FMA(WG_LOG2F_A0, r2, y);
    const double p = WG_FMA(WG_LOG2F_A3, r, y0);
    return (float)WG_FMA(y, r2, p);
}

// glibc 2.35 __log2 (sysdeps/ieee754/dbl-64/e_log2.c, !__FP_FAST_FMA build), positive normal inputs.
WG_HD double wg_log2(double x, const wg_d2* __restrict__ dtab, const wg_d2* __restrict__ dtab2)
{
    const uint64_t ix = wg_d2u(x);
    const uint64_t LO = 0x3feea4af00000000ull;   // asuint64(1.0 - 0x1.5b51p-5)
    const uint64_t HI = 0x3ff0b55900000000ull;   // asuint64(1.0 + 0x1.6ab2p-5)
    if (ix - LO < HI - LO) {
        if (ix == 0x3ff0000000000000ull) return 0.0;
        double r = x - 1.0;
        double rhi = wg_u2d(wg_d2u(r) & 0xffffffff00000000ull);
        double rlo = r - rhi;
        double hi = rhi * WG_LOG2_INVLN2HI;
        double lo = rlo * WG_LOG2_INVLN2HI + r * WG_LOG2_INVLN2LO;
        double r2 = r * r;
        double r4 = r2 * r2;
        double p = r2 * (WG_LOG2_B0 + r * WG_LOG2_B1);
        double y = hi + p;
        lo += hi - y + p;
        lo += r4 * (WG_LOG2_B2 + r * WG_LOG2_B3 + r2 * (WG_LOG2_B4 + r * WG_LOG2_B5)
                    + r4 * (WG_LOG2_B6 + r * WG_LOG2_B7 + r2 * (WG_LOG2_B8 + r * WG_LOG2_B9)));
        y += lo;
        return y;
    }
    uint64_t tmp = ix - 0x3fe6000000000000ull;
    uint32_t i = (uint32_t)(tmp >> 46) & 63u;
    int32_t k = (int32_t)((int64_t)tmp >> 52);        // |k| < 2^11: one cvt_f64_i32 on the device
    uint64_t iz = ix - (tmp & (0xfffull << 52));
    double invc = dtab[i].a, logc = dtab[i].b;
    double z = wg_u2d(iz);
    double kd = (double)k;
    double r = (z - dtab2[i].a - dtab2[i].b) * invc;
    double rhi = wg_u2d(wg_d2u(r) & 0xffffffff00000000ull);
    double rlo = r - rhi;
    double t1 = rhi * WG_LOG2_INVLN2HI;
    double t2 = rlo * WG_LOG2_INVLN2HI + r * WG_LOG2_INVLN2LO;
    double t3 = kd + logc;
    double hi = t3 + t1;
    double lo = t3 - hi + t1 + t2;
    double r2 = r * r;
    double r4 = r2 * r2;
    double p = WG_LOG2_A0 + r * WG_LOG2_A1 + r2 * (WG_LOG2_A2 + r * WG_LOG2_A3) + r4 * (WG_LOG2_A4 + r * WG_LOG2_A5);
    double y = lo + r2 * p + hi;
    return y;
}

// Call once on a freshly initialised table set (host or, by one thread, in LDS).
WG_HD void wg_tables_finish(wg_log_tables* tb)
{
    tb->d_fast[WG_FAST_CENTRE_ENTRY].a = 1.0;
    tb->d_fast[WG_FAST_CENTRE_ENTRY].b = 0.0;
}

// A cheap log2 for x = 1.0 - (double)p STRICTLY below 1 (p a float in [2^-53, 1), so x in [2^-24, 1 - 2^-53]): glibc's table
// and main polynomial, evaluated by plain Horner with fused multiply-adds, without the hi/lo compensation, without
// tab2, and without glibc's separate near-1 branch — instead the table interval just below 1 is centred exactly on 1
// (WG_FAST_CENTRE_ENTRY), which keeps the relative error flat up to x -> 1.
// It is NOT bit-identical to libm, but its distance from libm's result is bounded: tests/test_exact_log2_cpu.py
// measures |wg_fast_log2 - log2| <= WG_FAST_LOG2_MAX_ULP over that WHOLE domain (all 612,368,383 floats p),
// exhaustively, for the host build, and the gfx950 build is compared bit for bit with the host build (IEEE fma is
// deterministic).  9 fp64 operations instead of 30-34, no divergent branch.
#define WG_LOG2_INVLN2 0x1.71547652b82fep+0
#define WG_FAST_LOG2_MAX_ULP 1
// `dfast` = wg_log_tables::d_fast after wg_tables_finish().
WG_HD double wg_fast_log2(double x, const wg_d2* __restrict__ dfast)
{
    const uint64_t ix = wg_d2u(x);
    const uint32_t xhi = (uint32_t)(ix >> 32);
    const uint32_t hi = xhi - 0x3fe60000u;                  // high word of ix - 0x3fe6000000000000 (the low word of the constant is 0)
    const int32_t k = (int32_t)hi >> 20;
    const uint64_t iz = ((uint64_t)(xhi - (hi & 0xfff00000u)) << 32) | (uint32_t)ix;      // only the high word changes
#if defined(__HIP_DEVICE_COMPILE__)
    // bit-field extract, then shift-add onto the table base: two instructions where shift / mask / add would be three
    // (inline asm: the optimiser rewrites the builtin back into shift + mask)
    uint32_t i6;
    asm("v_bfe_u32 %0, %1, 14, 6" : "=v"(i6) : "v"(hi));
    const wg_d2* ent = reinterpret_cast<const wg_d2*>(reinterpret_cast<const char*>(dfast) + (i6 << 4));
    const double invc = ent->a, logc = ent->b;
#else
    const uint32_t i = (hi >> 14) & 63u;
    const double invc = dfast[i].a, logc = dfast[i].b;
#endif
    const double r = WG_FMA_K(wg_u2d(iz), invc, -1.0);
    double q = WG_LOG2_A5;
    q = WG_FMA_K(q, r, WG_LOG2_A4); q = WG_FMA_K(q, r, WG_LOG2_A3); q = WG_FMA_K(q, r, WG_LOG2_A2);
    q = WG_FMA_K(q, r, WG_LOG2_A1); q = WG_FMA_K(q, r, WG_LOG2_A0); q = WG_FMA_K(q, r, WG_LOG2_INVLN2);
    return WG_FMA(q, r, (double)k + logc);
}

// The two logs with BOTH the argument's normalisation and its exponent term folded into per-(k, i) tables (the narrow
// scoring kernel; wg_lookup_rows() rows each, pointers at the entries of k = 0, i = 0):
//   log2f:      entry (k, i) = {invc_f[i] * 2^-k, logc_f[i] + k}   index (tmp >> 19) = k * 16 + i
//   fast log2:  entry (k, i) = {invc_d[i] * 2^-k, k + logc_d[i]}   index (hi >> 14)  = k * 64 + i
// The reduced argument of either log is r = z * invc - 1 with z = v * 2^-k (exact); v * (invc * 2^-k) is the same real
// number (both scalings are by a power of two, exact), so the ONE rounding of the fused multiply-add gives the same r bit
// for bit — without extracting k, rebuilding z and (log2f) converting it to double.  The second table entry is the
// addition of the exponent, done once per entry with the operation the original performs per call.  Everything after r
// is the original code.  `xd` is (double)x, which the caller has anyway.
WG_HD float wg_log2f_ks(float x, double xd, const wg_d2* __restrict__ iys0)
{
    const uint32_t tmp = wg_f2u(x) - 0x3f330000u;
    const int32_t ki = (int32_t)tmp >> 19;                       // k * 16 + i
    const wg_d2 e = iys0[ki];
    const double r = WG_FMA_K(xd, e.a, -1.0);
    const double r2 = r * r;
    double y = WG_FMA_K(r, WG_LOG2F_A1, WG_LOG2F_A2);
    y = WG_FMA(WG_LOG2F_A0, r2, y);
    const double p = WG_FMA(WG_LOG2F_A3, r, e.b);
    return (float)WG_FMA(y, r2, p);
}
// FULL = true: every polynomial term of wg_fast_log2 (bit-identical to it).  FULL = false: the highest term (A5 r^7,
// |r| <= 2^-7) left out — one multiply-add fewer, up to WG_KS_LOG2_MAX_ULP ulp from libm instead of 1 (measured
// exhaustively, tests/test_exact_log2_cpu.py); the caller's guard band is widened to match (WG_GUARD_ULPS_KS).
#define WG_KS_LOG2_MAX_ULP 256
#define WG_GUARD_ULPS_KS 1024u      // >= 2 * WG_KS_LOG2_MAX_ULP + 4 (see wg_sample_term_pcpos_ks); band hit: 2^-18 per evaluation
template <bool FULL>
WG_HD double wg_fast_log2_ks(double x, const wg_d2* __restrict__ kys0)
{
    // index k * 64 + i = (xhi - 0x3fe60000) >> 14 = (xhi >> 14) - (0x3fe60000 >> 14), the constant having no bits below 14:
    // its subtraction moves into the table pointer (x > 0: the shift is a plain bit-field extract)
    const uint32_t xhi = (uint32_t)(wg_d2u(x) >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t kiu;
    asm("v_bfe_u32 %0, %1, 14, 18" : "=v"(kiu) : "v"(xhi));      // (inline asm: the optimiser would rebuild shift + add)
#else
    const uint32_t kiu = xhi >> 14;
#endif
    const wg_d2 e = (kys0 - (0x3fe60000u >> 14))[kiu];
    const double r = WG_FMA_K(x, e.a, -1.0);
    double q = WG_LOG2_A4;
    if (FULL) { q = WG_LOG2_A5; q = WG_FMA_K(q, r, WG_LOG2_A4); }
    q = WG_FMA_K(q, r, WG_LOG2_A3); q = WG_FMA_K(q, r, WG_LOG2_A2);
    q = WG_FMA_K(q, r, WG_LOG2_A1); q = WG_FMA_K(q, r, WG_LOG2_A0); q = WG_FMA_K(q, r, WG_LOG2_INVLN2);
    return WG_FMA(q, r, e.b);
}

// The reference's per-(block, sample) log-likelihood term, segmentor.cpp:125-135, on exact integer counts.
//   nmeth, ntotal : block sums of the sample (exact in float: 255*max_cpg < 2^24 is enforced by the ABI)
//   pc, pc2       : pseudo_count and pseudo_count+pseudo_count (== the reference's float `2 * pseudo_count`)
// Returns the float ll_k that the reference adds into its double ll_sum; 0 when the block has no coverage.
// Straightforward form: every operation as the reference performs it (exact libm restatements).
WG_HD float wg_sample_term_plain(float nmeth, float ntotal, float pc, float pc2, const wg_log_tables* __restrict__ tb)
{
    if (ntotal == 0.0f) return 0.0f;                               // :125
    float p = (nmeth + pc) / (ntotal + pc2);                       // :127 IEEE binary32 add, add, divide
    float ll = 0.0f;
    if (p > 0.0f) ll += nmeth * wg_log2f_nofma(p, tb->f_tab);      // :129-131 (kept as 0 + x: -0 becomes +0)
    if (p < 1.0f) {                                                // :132-134
        double t = (double)(ntotal - nmeth) * wg_log2(1.0 - (double)p, tb->d_tab, tb->d_tab2);
        ll = (float)((double)ll + t);
    }
    return ll;
}

// Production form, identical results (Ziv's strategy).  The second term is
//     ll = (float)( (double)ll + (double)(ntotal-nmeth) * L ),   L = libm log2(1 - p)
// and only its FLOAT rounding is observable.  With L' = wg_fast_log2 (|L'-L| <= 1 ulp, measured exhaustively) the
// double sum s' differs from the true s by at most 6 ulp(s): the products differ by <= (2^-52 + 2*2^-53)|prod|, both
// addends are <= 0 so |prod| <= |s|, and each sum adds one rounding (2^-53|s|).  So (float)s' == (float)s unless s'
// lies within 6 ulp of a float rounding midpoint (low 29 mantissa bits == 2^28); we use a 16-ulp guard band, and in
// that band (probability 2^-24 per evaluation) the exact restatement decides.  When ntotal == nmeth the reference adds -0.0 and ll is unchanged.
#define WG_GUARD_ULPS 16u
// "low 29 bits of s within WG_GUARD_ULPS of 2^28", as one shift-add and one compare: with tail = lo & (2^29-1) and
// C = 2^28 - G, (tail - C) mod 2^32 <= 2G  <=>  (8*tail - 8*C) mod 2^32 <= 16G, and 8*tail mod 2^32 is just lo << 3.
WG_HD bool wg_in_guard_band(double s, const uint32_t guard_ulps = WG_GUARD_ULPS)
{
    const uint32_t t3 = ((uint32_t)wg_d2u(s) << 3) + (0u - ((0x10000000u - guard_ulps) << 3));
    return t3 <= 16u * guard_ulps;
}
// IEEE-754 binary32 division a / b for operands in a "comfortable" range (0 <= a <= b, 2^-20 <= b < 2^26, or a == 0).
// On the device this is the core of the sequence hipcc emits for `/` (v_rcp_f32, one Newton step on the reciprocal,
// two residual corrections of the quotient) WITHOUT v_div_scale (a no-op unless an operand or the quotient sits near
// the exponent limits) and v_div_fixup (special values only): 8 instructions instead of 11, same intermediate values,
// hence the same correctly rounded quotient.  tests/test_gpu_parity.py::test_02b compares it with `/` on the device.
// A copy of v the optimiser cannot see through (keeps a value from being held in registers for a rare path).
WG_HD float wg_opaque_f32(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}
WG_HD float wg_div_f32(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float r = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = a * r;
    float res = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(res, r, q);
    res = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(res, r, q);
#else
    return a / b;
#endif
}

// The short core: v_rcp_f32 (1 ulp), the quotient, ONE residual correction — 4 instructions.  The correction term carries a
// relative error of about 2^-22 ulp of the quotient, so the result is the correctly rounded quotient unless a / b lies that
// close to a rounding midpoint.  That cannot be excluded for arbitrary floats, but the operands of a narrow scoring tile are
// few: a = fl(nmeth + pc), b = fl(ntotal + 2 pc), 0 <= nmeth <= ntotal <= 255 * 60.  The library checks ALL of them on the
// device against the compiler's IEEE division for the pseudo count of a call (k_check_div: 1.2e8 pairs, ~0.1 ms, once per
// context and pseudo count) and uses this form only when not one quotient differs; otherwise, and for wide tiles, the
// 8-instruction core above.  (Integer pseudo counts: a / b of integers below 2^14 stays >= 2^-15 ulp away from every
// midpoint, far outside the error; the check covers fractional ones.)
WG_HD float wg_div_f32_short(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_rcpf(b);
    const float q = a * r;
    const float res = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(res, r, q);
#else
    return a / b;
#endif
}

// Preconditions of the fast form (the library dispatches on them; otherwise wg_sample_term_plain is used):
// pc == 0 or pc >= WG_FAST_MIN_PC.  Then (a) p == 0 exactly or p >= 2^-45: p is a normal float and, when p > 0,
// x = 1 - p < 1 strictly (wg_fast_log2's domain); (b) every non-zero sum is >= 2^-60 in magnitude, so the float
// result is never subnormal and the guard-band test needs no exponent check.  For p == 0 (pc == 0 and nmeth == 0)
// the reference computes 0 + (ntotal-nmeth)*log2(1.0) = +0: ll stays +0.
#define WG_FAST_MIN_PC 0x1p-20f
// With a pseudo count >= 1 the guards of segmentor.cpp:129,132 are always true: 0 < p, and p < 1 even after all three
// float roundings: the ABI keeps ntotal <= 255 * 8000 < 2^21, where a float sum is off by at most 2^-4, so
// fl(nmeth+pc) / fl(ntotal+2pc) <= 1 - (pc - 2^-3) / (ntotal + 2pc) <= 1 - 0.875 / (2^21 + 2) — more than 2^-22 below 1,
// while the quotient's own rounding moves it by at most 2^-25.  The
// `0 +` of :131 and the df == 0 exception fall away too: nmeth*log2f(p) is non-zero unless nmeth == 0, and then
// df = ntotal > 0 makes the second term non-zero, so no sign of zero survives; df == 0 adds -0.0 to a non-zero ll.
// Three compares, a move and two branch levels fewer per evaluation, same bits (tests: test_02, host twin).
#define WG_POS_MIN_PC 1.0f
WG_HD float wg_sample_term_pcpos(float nmeth, float ntotal, float pc, float pc2, const wg_fast_tables* __restrict__ ft,
                                 const wg_log_tables* __restrict__ xt)
{
    if (ntotal == 0.0f) return 0.0f;                               // :125
    const float p = wg_div_f32(nmeth + pc, ntotal + pc2);          // :127
    const float ll = nmeth * wg_log2f_normal(p, ft->f_tab);        // :129-131
    const float df = ntotal - nmeth;
    const double x = 1.0 - (double)p;                              // :132-134
    const double s = (double)ll + (double)df * wg_fast_log2(x, ft->d_fast);
    const uint32_t tail = (uint32_t)wg_d2u(s) & 0x1fffffffu;
    float res = (float)s;
    if ((uint32_t)(tail - (0x10000000u - WG_GUARD_ULPS)) <= 2u * WG_GUARD_ULPS)
        res = (float)((double)ll + (double)df * wg_log2(x, xt->d_tab, xt->d_tab2));
    return res;
}

// What the scoring kernels run for pseudo counts >= 1: the guard-free form WITHOUT the zero-coverage exception — the
// callers ADD the term to a running sum, and with ntotal == 0: p = 1/2, ll = 0 * log2f = -0.0, df = 0, s = -0.0 + 0 * L
// = -0.0, and adding -0.0 to the running double sum leaves it unchanged, bit for bit: the reference's `continue` (:125)
// without a branch — on the k-scaled tables (which must hold the rows wg_lookup_rows() names for the longest block
// scored), with a cheaper approximate sum: L' = wg_fast_log2_ks<false> is within E = WG_KS_LOG2_MAX_ULP ulp of libm's L, and the sum is ONE
// fused multiply-add.  Against the reference's s = fl(ll + fl(df L)):  |df L' - fl(df L)| <= (E 2^-52 + 2^-53) |df L|,
// |df L| <= |s| (both addends <= 0), one more rounding of at most half an ulp, and ulp(s) >= 2^-53 |s|: |s' - s| <=
// (2 E + 2) ulp(s) < WG_GUARD_ULPS_KS.  Inside that band around a float rounding midpoint the exact form decides, as ever.
template <bool DIVS = false>       // DIVS: the short division core (only where k_check_div has verified it for the call's operands)
WG_HD float wg_sample_term_pcpos_ks(float nmeth, float ntotal, float pc, float pc2, const wg_d2* __restrict__ iys0,
                                    const wg_d2* __restrict__ kys0, const wg_log_tables* __restrict__ xt)
{
    const float p = DIVS ? wg_div_f32_short(nmeth + pc, ntotal + pc2) : wg_div_f32(nmeth + pc, ntotal + pc2);          // :127
    const double pd = (double)p;
    const float ll = nmeth * wg_log2f_ks(p, pd, iys0);             // :129-131
    const float df = wg_opaque_f32(ntotal) - nmeth;                // (opaque: or the subtraction moves to the integers the counts came from, one conversion more)
    const double x = 1.0 - pd;                                     // :132-134
    const double s = WG_FMA((double)df, wg_fast_log2_ks<false>(x, kys0), (double)ll);
    float res = (float)s;
    if (wg_in_guard_band(s, WG_GUARD_ULPS_KS))
        res = (float)((double)ll + (double)df * wg_log2(1.0 - (double)wg_opaque_f32(p), xt->d_tab, xt->d_tab2));
    return res;
}
// Rows (exponents k = -(rows-1) .. 0) the two lookup tables need when every block has ntotal <= max_total < 2^21 and the
// pseudo count is pc >= 1.  In exact arithmetic p and 1 - p are both >= v = pc / (max_total + 2 pc) >= 2^-21.01.  The
// computed p = fl(fl(nmeth+pc) / fl(ntotal+2pc)) carries three float roundings, a relative error below 3 * 2^-24: the
// smallest computed p is >= v (1 - 2^-22), and the smallest computed 1 - p (the subtraction is exact in double) is
// >= v - 3 * 2^-24 > v - 2^-22 — an ABSOLUTE error, a quarter of v at the extreme.  An argument w has exponent
// k = floor(log2(w / 0.6875)) in wg_fast_log2 and floor(log2(w / 0.69921875)) in wg_log2f: the LARGER base gives the lower
// exponent, so the rows must reach down to 0.69921875 * 2^-(rows-1) <= w (tests/test_exact_log2_cpu.py walks every block
// total the ABI admits).
static inline int wg_lookup_rows(float pc, double max_total)
{
    const double v = (double)pc / (max_total + 2.0 * (double)pc);
    const double wmin = v - 0x1p-22;
    int rows = 1;
    for (double lo = 0.69921875; !(lo <= wmin) && rows <= 64; lo *= 0.5) rows++;
    return rows;
}

// Entry x of the two k-scaled tables with `rows` exponents (k = -(rows-1) .. 0), from the constant tables:
//   iy[(k + rows-1) * 16 + i] = {invc_f[i] * 2^-k, logc_f[i] + k}     ky[(k + rows-1) * 64 + i] = {invc_d[i] * 2^-k, k + logc_d[i]}
// (interval WG_FAST_CENTRE_ENTRY of the fast log2 centred on 1, as in wg_tables_finish()).  The scaling is exact (a
// power of two), the second member is the one IEEE addition the unscaled functions perform per call: host and device
// builds produce the same bits.
WG_HD wg_d2 wg_ks_iy_entry(const wg_log_tables* __restrict__ tb, int rows, int x)
{
    const int k = (x >> 4) - (rows - 1);
    wg_d2 e;
    e.a = tb->f_tab[x & 15].a * (double)(1u << -k);
    e.b = tb->f_tab[x & 15].b + (double)k;                        // logc[i] + k, exactly as wg_log2f_normal adds them
    return e;
}
WG_HD wg_d2 wg_ks_ky_entry(const wg_log_tables* __restrict__ tb, int rows, int x)
{
    const int i = x & 63, k = (x >> 6) - (rows - 1);
    const bool centre = i == WG_FAST_CENTRE_ENTRY;
    wg_d2 e;
    e.a = (centre ? 1.0 : tb->d_tab[i].a) * (double)(1u << -k);
    e.b = (double)k + (centre ? 0.0 : tb->d_tab[i].b);            // (double)k + logc, exactly as wg_fast_log2 adds them
    return e;
}

// term mode of a pseudo count: 0 plain exact form, 1 fast form with the guards, 2 fast form without them
// Above WG_FAST_MAX_PC the plain exact form is used as well: the division core is validated for divisors below 2^26
// (ntotal + 2 pc with ntotal < 2^24), and a pseudo count whose double overflows makes p == 0, which the guard-free form excludes.
#define WG_FAST_MAX_PC 0x1p24f
WG_HD int wg_term_mode(float pc)
{
    if (!(pc <= WG_FAST_MAX_PC)) return 0;
    return pc >= WG_POS_MIN_PC ? 2 : ((pc == 0.0f || pc >= WG_FAST_MIN_PC) ? 1 : 0);
}
WG_HD float wg_sample_term(float nmeth, float ntotal, float pc, float pc2, const wg_fast_tables* __restrict__ ft,
                           const wg_log_tables* __restrict__ xt)
{
    if (ntotal == 0.0f) return 0.0f;                               // :125
    const float p = wg_div_f32(nmeth + pc, ntotal + pc2);          // :127 (pc == 0 or pc >= 2^-20: operands in range)
    float ll = 0.0f;
    if (p > 0.0f) {
        ll += nmeth * wg_log2f_normal(p, ft->f_tab);               // :129-131
        const float df = ntotal - nmeth;
        if (p < 1.0f && df != 0.0f) {                              // :132-134 (df == 0 adds -0.0: ll unchanged)
            const double x = 1.0 - (double)p;
            const double s = (double)ll + (double)df * wg_fast_log2(x, ft->d_fast);
            const uint32_t tail = (uint32_t)wg_d2u(s) & 0x1fffffffu;
            float res = (float)s;
            if ((uint32_t)(tail - (0x10000000u - WG_GUARD_ULPS)) <= 2u * WG_GUARD_ULPS)
                res = (float)((double)ll + (double)df * wg_log2(x, xt->d_tab, xt->d_tab2));
            ll = res;
        }
    }
    return ll;
}
