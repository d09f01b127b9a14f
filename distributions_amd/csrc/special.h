// Table-driven special functions of the hot path, for gfx950 device code and
// for the host-side scalar Group API.
//
// These restate distributions/special.hpp and vendor/fmath.hpp of the
// reference (file:line cited per function, relative to /root/reference) in
// the operation order its release build executes (observed in the compiled
// reference, see DESIGN.md "Numeric contract"); the tables are the reference's
// tables as built (ref_tables.h, generated).  The library is compiled with
// -ffp-contract=off and -fgpu-flush-denormals-to-zero: no fused multiply-add
// may be formed from these expressions and f32 denormals flush, as they do
// under the reference's -ffast-math (crtfastmath FTZ/DAZ).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <string.h>

#include "ref_tables.h"

#define DIST_HD __host__ __device__ __forceinline__

namespace dist {

// Tables live in device global memory (L2-resident: 64 KiB + 4 KiB + small)
// and in host memory; kernels that gather per lane stage the hot ones in LDS.
struct Tables {
    uint32_t log_table[16384];   // special.cc:35-44, FastLog(14)
    uint32_t exp_table[1024];    // vendor/fmath.hpp:165-172
    uint32_t lgamma_coeff5[192]; // special.cc:144-211
    uint32_t lgamma_nu_coeff3[80];   // special.cc:232-269
    uint32_t log_factorial[64];  // special.cc:213-230
    uint32_t exp_ab[2];          // vendor/fmath.hpp:154-160: 1024/logf(2), logf(2)/1024
};

extern __device__ Tables g_tables_dev;   // defined in kernels.hip
extern const Tables * g_tables_host;     // host copy

// glibc lgammaf values for the few arguments below 2.5 that a model can
// reach (special.hpp:121-123 calls libm there): lgammaf is a pure function,
// so one table of (y, lgammaf(y)) pairs, sorted by y, serves every feature
// on the device.  The host registers a model's reachable arguments when the
// feature is created (register_small_lgamma, dist_hip.hip).
constexpr int kLgammaLutCap = 16384;
struct LgammaLut {
    int n;
    float y[kLgammaLutCap];
    float v[kLgammaLutCap];
};
// Two copies: the host fills the one no kernel reads and then flips
// g_lgamma_cur, so a table is never rewritten under a running kernel (another
// host thread may launch while this one registers a feature).
extern __device__ LgammaLut g_lgamma_lut[2];
extern __device__ int g_lgamma_cur;

DIST_HD float u2f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}
DIST_HD uint32_t f2u(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u; memcpy(&u, &f, 4); return u;
#endif
}

DIST_HD const Tables & tables() {
#if defined(__HIP_DEVICE_COMPILE__)
    return g_tables_dev;
#else
    return *g_tables_host;
#endif
}

// round-to-nearest-even float -> int32 with the x86 cvtss2si out-of-range
// result (0x80000000), which fmath::exp's range test relies on
DIST_HD int32_t cvtss_si32(float x) {
    if (!(x < 2147483648.0f) || x < -2147483648.0f) return (int32_t)0x80000000;
#if defined(__HIP_DEVICE_COMPILE__)
    return __float2int_rn(x);
#else
    return (int32_t)rintf(x);   // default rounding mode: nearest-even
#endif
}

// special.hpp:57-67 FastLog::log (N = 14); `table` may point at an LDS copy
DIST_HD float fast_log_t(float x, const uint32_t * table) {
    const int32_t intx = (int32_t)f2u(x);
    const int e = ((intx >> 23) & 255) - 127;
    const int man = (intx & 0x7FFFFF) >> 9;
    return ((float)e + u2f(table[man])) * 0.69314718055994529f;
}
DIST_HD float fast_log(float x) { return fast_log_t(x, tables().log_table); }

// vendor/fmath.hpp:438-459 fmath::exp (SSE branch) == special.hpp:87-89.
// Release-build operation order: ((x + 1) - float(r)*b) * 2^u*tbl[v].
DIST_HD float fast_exp_core(float x, const uint32_t * exp_table, float a,
                            float b) {
    const float xa = x * a;
    const int32_t r = cvtss_si32(xa);
    const uint32_t v = (uint32_t)r & 1023u;
    const int32_t u = r >> 10;
    const uint32_t bits = ((uint32_t)(u + 127) << 23) | exp_table[v];
    const float rb = (float)r * b;
    return ((x + 1.0f) - rb) * u2f(bits);
}
DIST_HD float fast_exp(float x) {
    const Tables & t = tables();
    const int32_t limit = cvtss_si32(x) & 0x7fffffff;
    if (limit > 0x42b00000) {
        x = fminf(x, 88.0f);   // minss/maxss: second operand wins on NaN
        x = fmaxf(x, -88.0f);
    }
    return fast_exp_core(x, t.exp_table, u2f(t.exp_ab[0]), u2f(t.exp_ab[1]));
}
// the sampler only ever passes score - max <= 0, for which the range test of
// fmath::exp reduces to max(x, -88) (it fires for every x < -0.5 and is a
// no-op on [-0.5, 0])
DIST_HD float fast_exp_nonpos(float x, const uint32_t * exp_table, float a,
                              float b) {
    x = fmaxf(x, -88.0f);
    return fast_exp_core(x, exp_table, a, b);
}

// libm lgammaf stand-in for the y < 2.5 branch of fast_lgamma.  The reference
// calls glibc's lgammaf there (special.hpp:121-123); this evaluates lgamma in
// binary64 and rounds once, which is exact for y = 1, 2 and within 1 ulp of
// glibc elsewhere (DESIGN.md "Known numeric gaps").
#if defined(__HIP_DEVICE_COMPILE__)
// Out of line on purpose: the branch is rare (groups of at most two members)
// and its table search plus binary64 lgamma would otherwise be inlined into
// every scoring loop and cost them their registers.
static __device__ __noinline__ float libm_lgammaf_device(float y) {
    // registered arguments: glibc's value, bit for bit
    const LgammaLut & lut = g_lgamma_lut[g_lgamma_cur & 1];
    int lo = 0, hi = lut.n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (lut.y[mid] < y) lo = mid + 1; else hi = mid;
    }
    if (lo < lut.n && lut.y[lo] == y) return lut.v[lo];
    return (float)::lgamma((double)y);
}
#endif
DIST_HD float libm_lgammaf(float y) {
#if defined(__HIP_DEVICE_COMPILE__)
    return libm_lgammaf_device(y);
#else
    return ::lgammaf(y);   // host side: the very libm call of the reference
#endif
}

// special.hpp:114-171
DIST_HD float fast_lgamma(float y) {
    if (y < 2.5f || 4294967295.0f <= y) {
        return libm_lgammaf(y);
    }
    const int32_t x = (int32_t)f2u(y);
    const int c = (x >> 23) - 127;
    const uint32_t * co = tables().lgamma_coeff5 + c * 6;
    const float a5 = u2f(co[0]), a4 = u2f(co[1]), a3 = u2f(co[2]);
    const float a2 = u2f(co[3]), a1 = u2f(co[4]), a0 = u2f(co[5]);
    double yprod = y;
    double sum = a0;
    sum += a1 * yprod;
    yprod *= y;
    sum += a2 * yprod;
    yprod *= y;
    sum += a3 * yprod;
    yprod *= y;
    sum += a4 * yprod;
    yprod *= y;
    sum += a5 * yprod;
    return (float)sum;
}

// special.hpp:208-214
DIST_HD float fast_log_factorial(uint32_t n) {
    if (n < 64) return u2f(tables().log_factorial[n]);
    return fast_lgamma((float)(n + 1u));
}

// special.hpp:224-273; release-build order (a0 + xx*a2) + x*(a1 + xx*a3)
DIST_HD float fast_lgamma_nu(float nu) {
    if (nu < 0.0625f || 4294967295.0f <= nu) {
        return libm_lgammaf((nu + 1.0f) * 0.5f) - libm_lgammaf(nu * 0.5f);
    }
    const int32_t x = (int32_t)f2u(nu);
    const int c = (x >> 23) - 127;
    const uint32_t * co = tables().lgamma_nu_coeff3 + ((c + 4) / 2) * 4;
    const float a3 = u2f(co[0]), a2 = u2f(co[1]), a1 = u2f(co[2]),
                a0 = u2f(co[3]);
    const float xx = nu * nu;
    const float p = xx * a3 + a1;
    const float q = xx * a2 + a0;
    return q + p * nu;
}

// ---------------------------------------------------------------------------
// rng_t = std::default_random_engine = minstd_rand0 (random_fwd.hpp:34):
// x <- 16807 x mod (2^31 - 1).  Position j of a stream is seed * 16807^j.

DIST_HD uint32_t lcg_mulmod(uint32_t a, uint32_t b) {
    // (a * b) mod (2^31 - 1) for a, b < 2^31 - 1, by Mersenne folding
    uint64_t p = (uint64_t)a * (uint64_t)b;
    uint64_t r = (p & 0x7fffffffull) + (p >> 31);
    r = (r & 0x7fffffffull) + (r >> 31);
    return (uint32_t)(r == 0x7fffffffull ? 0 : r);
}
DIST_HD uint32_t lcg_jump(uint32_t state, uint64_t steps) {
    uint32_t result = state, base = 16807u;
    while (steps) {
        if (steps & 1) result = lcg_mulmod(result, base);
        base = lcg_mulmod(base, base);
        steps >>= 1;
    }
    return result;
}
// random.hpp:47-50 over libstdc++'s generate_canonical<float,24>: one engine
// step, u = float(x - 1) / 2^31, clamped below 1
DIST_HD float lcg_unif01(uint32_t x) {
    float ret = (float)(x - 1u) * 4.656612873077392578125e-10f;  // 2^-31, exact
    return ret >= 1.0f ? u2f(0x3f7fffffu) : ret;
}

}  // namespace dist
