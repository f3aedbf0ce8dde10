// ims_math.h -- device statement of the numerics spec (DESIGN.md "Numerics spec") for gfx950.
//
// Philox4x32-10 and binary64 elementary functions written only with + - * / fma sqrt and integer
// ops, evaluated in a fixed order, so every photon gets the same bits on any launch geometry and
// on any GPU.  Build with -ffp-contract=off: the only fused operations are the explicit fma()s.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define IMS_DEV __device__ __forceinline__



namespace ims {

struct Draw { uint64_t a, b; };

// fma(a, b, K) for a literal K (a Horner step).  gfx950 cannot encode 64-bit literals; left to itself the compiler
// materialises K with two v_mov_b32 into the accumulator of a two-address v_fmac_f64 -- three vector instructions
// per step (a fifth of the photon kernels' vector issue slots).  Stated explicitly, K sits in a scalar register pair
// (two s_mov_b32 on the scalar unit, which issues beside the vector work of the other waves) and the step is ONE
// v_fma_f64.  Same operation, same rounding.
IMS_DEV double fma_k(double a, double b, double k)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}

IMS_DEV void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]);
        const uint32_t lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]);
        const uint32_t lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0;
        const uint32_t n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// draw addressed by (seed, object id, photon index, slot)
IMS_DEV Draw draw(uint64_t seed, int64_t obj_id, int64_t photon, uint32_t slot)
{
    uint32_t c[4];
    c[0] = (uint32_t)((uint64_t)photon);
    c[1] = (uint32_t)((uint64_t)photon >> 32);
    c[2] = slot;
    c[3] = (uint32_t)((uint64_t)obj_id);
    const uint32_t k0 = (uint32_t)seed;
    const uint32_t k1 = (uint32_t)(seed >> 32) ^ (uint32_t)((uint64_t)obj_id >> 32);
    philox4x32_10(c, k0, k1);
    Draw d;
    d.a = (((uint64_t)c[0] << 32) | c[1]) >> 11;
    d.b = (((uint64_t)c[2] << 32) | c[3]) >> 11;
    return d;
}
// The photon pipeline (spec v4) consumes the raw block: four 32-bit words per (object, photon,
// slot), each mapped to the open interval (0,1).  A photon keeps its last block in registers, so
// consecutive consumers of one slot (two PSF components, two ops of the chain, the four sensor
// draws) cost ONE Philox evaluation.
struct Rng { uint32_t slot; uint32_t w[4]; };
IMS_DEV void rng_reset(Rng& r) { r.slot = 0xFFFFFFFFu; r.w[0] = 0u; r.w[1] = 0u; r.w[2] = 0u; r.w[3] = 0u; }
IMS_DEV void rng_block(Rng& r, uint64_t seed, int64_t obj_id, int64_t photon, uint32_t slot)
{
    if (r.slot == slot) return;
    uint32_t c[4];
    c[0] = (uint32_t)((uint64_t)photon);
    c[1] = (uint32_t)((uint64_t)photon >> 32);
    c[2] = slot;
    c[3] = (uint32_t)((uint64_t)obj_id);
    const uint32_t k0 = (uint32_t)seed;
    const uint32_t k1 = (uint32_t)(seed >> 32) ^ (uint32_t)((uint64_t)obj_id >> 32);
    philox4x32_10(c, k0, k1);
    r.w[0] = c[0]; r.w[1] = c[1]; r.w[2] = c[2]; r.w[3] = c[3];
    r.slot = slot;
}
// (w + 1/2) / 2^32, exact
IMS_DEV double w01(uint32_t w) { return fma((double)w, 0x1.0p-32, 0x1.0p-33); }

IMS_DEV double u01(uint64_t k) { return (double)k * 0x1.0p-53; }
IMS_DEV double u01_open(uint64_t k) { return (double)(k + 1) * 0x1.0p-53; }

// IEEE-correct division and square root for operands in the normal range.  The compiler's expansions wrap the same
// Newton / residual-correction cores in range scaling (v_div_scale x2, v_ldexp x2, compares) and special-value
// fix-ups (v_div_fixup, v_cmp_class + selects) for zeros, infinities and results near the ends of the exponent
// range: 11 and 18 instructions.  The photon arithmetic never gets there (positions, slopes, indices and
// wavelengths of order 1e-12 .. 1e12), so the hot path states the cores alone -- 8 and 10 instructions with the
// same correctly rounded result (tests/test_parity_gpu.py::test_lean_div_sqrt compares 2^26 random operand pairs
// per decade range bit for bit with the full expansions).
//   ddiv(a, b):   b finite, non-zero, |b| and |a/b| within 2^+-500
//   dsqrt_n(x):   x > 0 finite within 2^+-500;   dsqrt0(x): additionally exact for x == 0
IMS_DEV double ddiv(double a, double b)
{
    double y = __builtin_amdgcn_rcp(b);
    double e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    const double q = a * y;
    const double r = fma(-b, q, a);
    return fma(r, y, q);
}
IMS_DEV double dsqrt_n(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double g0 = x * y;
    const double h0 = y * 0.5;
    const double r0 = fma(-h0, g0, 0.5);
    const double h1 = fma(h0, r0, h0);
    const double g1 = fma(g0, r0, g0);
    const double d0 = fma(-g1, g1, x);
    const double g2 = fma(d0, h1, g1);
    const double d1 = fma(-g2, g2, x);
    return fma(d1, h1, g2);
}
IMS_DEV double dsqrt0(double x) { const double r = dsqrt_n(x); return x == 0.0 ? 0.0 : r; }

constexpr double LN2_HI = 6.93147180369123816490e-01;
constexpr double LN2_LO = 1.90821492927058770002e-10;
constexpr double INV_LN2 = 1.44269504088896338700e+00;
constexpr double TWO_PI = 6.283185307179586476925;
constexpr double INV_TWO_PI = 0.15915494309189533577;
constexpr double PI_2 = 1.57079632679489661923;
constexpr double PI_4 = 0.78539816339744830962;

IMS_DEV double dlog(double x)
{
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    int64_t e = (int64_t)((b >> 52) & 0x7FF) - 1023;
    double m = __longlong_as_double((long long)((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double s = ddiv(m - 1.0, m + 1.0);
    const double z = s * s;
    double p = 1.0 / 25.0;
    p = fma_k(p, z, 1.0 / 23.0);
    p = fma_k(p, z, 1.0 / 21.0);
    p = fma_k(p, z, 1.0 / 19.0);
    p = fma_k(p, z, 1.0 / 17.0);
    p = fma_k(p, z, 1.0 / 15.0);
    p = fma_k(p, z, 1.0 / 13.0);
    p = fma_k(p, z, 1.0 / 11.0);
    p = fma_k(p, z, 1.0 / 9.0);
    p = fma_k(p, z, 1.0 / 7.0);
    p = fma_k(p, z, 1.0 / 5.0);
    p = fma_k(p, z, 1.0 / 3.0);
    p = fma(p, z, 1.0);
    const double lm = 2.0 * s * p;
    const double de = (double)e;
    return fma(de, LN2_HI, fma(de, LN2_LO, lm));
}

IMS_DEV double dexp(double x)
{
    const double kf = floor(fma(x, INV_LN2, 0.5));
    double r = fma(-kf, LN2_HI, x);
    r = fma(-kf, LN2_LO, r);
    double p = 1.0 / 6227020800.0;
    p = fma_k(p, r, 1.0 / 479001600.0);
    p = fma_k(p, r, 1.0 / 39916800.0);
    p = fma_k(p, r, 1.0 / 3628800.0);
    p = fma_k(p, r, 1.0 / 362880.0);
    p = fma_k(p, r, 1.0 / 40320.0);
    p = fma_k(p, r, 1.0 / 5040.0);
    p = fma_k(p, r, 1.0 / 720.0);
    p = fma_k(p, r, 1.0 / 120.0);
    p = fma_k(p, r, 1.0 / 24.0);
    p = fma_k(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int64_t k = (int64_t)kf;
    if (k < -1000) return 0.0;
    if (k > 1000) k = 1000;
    const double scale = __longlong_as_double((long long)((uint64_t)(k + 1023) << 52));
    return p * scale;
}

IMS_DEV double sin_kernel(double t)
{
    const double z = t * t;
    double p = 1.0 / 355687428096000.0;
    p = fma_k(p, z, -1.0 / 1307674368000.0);
    p = fma_k(p, z, 1.0 / 6227020800.0);
    p = fma_k(p, z, -1.0 / 39916800.0);
    p = fma_k(p, z, 1.0 / 362880.0);
    p = fma_k(p, z, -1.0 / 5040.0);
    p = fma_k(p, z, 1.0 / 120.0);
    p = fma_k(p, z, -1.0 / 6.0);
    return fma(t * z, p, t);
}
IMS_DEV double cos_kernel(double t)
{
    const double z = t * t;
    double p = -1.0 / 6402373705728000.0;
    p = fma_k(p, z, 1.0 / 20922789888000.0);
    p = fma_k(p, z, -1.0 / 87178291200.0);
    p = fma_k(p, z, 1.0 / 479001600.0);
    p = fma_k(p, z, -1.0 / 3628800.0);
    p = fma_k(p, z, 1.0 / 40320.0);
    p = fma_k(p, z, -1.0 / 720.0);
    p = fma_k(p, z, 1.0 / 24.0);
    p = fma(p, z, -0.5);
    return fma(z, p, 1.0);
}
// sin, cos of 2*pi*u, u in [0,1]: the quadrant reduction is exact in binary
IMS_DEV void sincos2pi(double u, double& s, double& c)
{
    const double qf = floor(fma(4.0, u, 0.5));
    const double r = fma(-0.25, qf, u);
    const double t = r * TWO_PI;
    const double sk = sin_kernel(t), ck = cos_kernel(t);
    // quadrant q: (s, c) = (sk, ck), (ck, -sk), (-sk, -ck), (-ck, sk) -- as two selects and two sign flips instead of
    // a four-way branch (negation is exact, so the bits are those of the branches)
    const int q = (int)qf & 3;
    const bool odd = (q & 1) != 0;
    const double s0 = odd ? ck : sk, c0 = odd ? sk : ck;
    const unsigned long long sbit = (unsigned long long)(q & 2) << 62;               // q = 2, 3: sin negative
    const unsigned long long cbit = (unsigned long long)((q + 1) & 2) << 62;         // q = 1, 2: cos negative
    s = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(s0) ^ sbit));
    c = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(c0) ^ cbit));
}
IMS_DEV void dsincos(double x, double& s, double& c)
{
    double u = x * INV_TWO_PI;
    u = u - floor(u);
    sincos2pi(u, s, c);
}

IMS_DEV double datan(double x)
{
    const double ax = x < 0.0 ? -x : x;
    double base = 0.0, sign = 1.0, a = ax;
    if (ax > 1.0) { a = 1.0 / ax; base = PI_2; sign = -1.0; }
    double off = 0.0, b = a;
    if (a > 0.41421356237309503) { b = ddiv(a - 1.0, a + 1.0); off = PI_4; }
    const double cc = ddiv(b, 1.0 + dsqrt_n(fma(b, b, 1.0)));
    const double z = cc * cc;
    double p = 1.0 / 27.0;
    p = fma_k(p, z, -1.0 / 25.0);
    p = fma_k(p, z, 1.0 / 23.0);
    p = fma_k(p, z, -1.0 / 21.0);
    p = fma_k(p, z, 1.0 / 19.0);
    p = fma_k(p, z, -1.0 / 17.0);
    p = fma_k(p, z, 1.0 / 15.0);
    p = fma_k(p, z, -1.0 / 13.0);
    p = fma_k(p, z, 1.0 / 11.0);
    p = fma_k(p, z, -1.0 / 9.0);
    p = fma_k(p, z, 1.0 / 7.0);
    p = fma_k(p, z, -1.0 / 5.0);
    p = fma_k(p, z, 1.0 / 3.0);
    p = fma(p, z, -1.0);
    const double at = -(cc * p);
    double res = fma(2.0, at, off);
    res = fma(sign, res, base);
    return x < 0.0 ? -res : res;
}

// atan2 with the usual quadrant conventions (finite arguments, not both zero -> 0 for (0,0))
IMS_DEV double datan2(double y, double x)
{
    constexpr double PI = 3.14159265358979323846;
    if (x > 0.0) return datan(y / x);
    if (x < 0.0) return y >= 0.0 ? datan(y / x) + PI : datan(y / x) - PI;
    if (y > 0.0) return PI_2;
    if (y < 0.0) return -PI_2;
    return 0.0;
}

IMS_DEV double dtanh_pos(double x)
{
    if (x > 20.0) return 1.0;
    const double e = dexp(-2.0 * x);
    return ddiv(1.0 - e, 1.0 + e);
}
IMS_DEV double dpow(double x, double y) { return dexp(y * dlog(x)); }

IMS_DEV void gauss_pair(Draw d, double& g0, double& g1)
{
    const double u1 = u01_open(d.a);
    const double u2 = u01(d.b);
    const double r = sqrt(-2.0 * dlog(u1));
    double s, c;
    sincos2pi(u2, s, c);
    g0 = r * c; g1 = r * s;
}

// Deviates from 32-bit words (spec v6).  The argument (w + 1/2) / 2^32 has a granularity of 2^-32, so the series are cut
// where the truncation error falls below 2^-36 of the result (sine to t^11, cosine to t^12, the logarithm's atanh series to
// s^13), and the quadrant reduction of the angle is integer work on the word itself; t is the same product as in sincos2pi.
IMS_DEV double sin_kernel_w(double t, double z)
{
    double p = -1.0 / 39916800.0;
    p = fma_k(p, z, 1.0 / 362880.0);
    p = fma_k(p, z, -1.0 / 5040.0);
    p = fma_k(p, z, 1.0 / 120.0);
    p = fma_k(p, z, -1.0 / 6.0);
    return fma(t * z, p, t);
}
IMS_DEV double cos_kernel_w(double z)
{
    double p = 1.0 / 479001600.0;
    p = fma_k(p, z, -1.0 / 3628800.0);
    p = fma_k(p, z, 1.0 / 40320.0);
    p = fma_k(p, z, -1.0 / 720.0);
    p = fma_k(p, z, 1.0 / 24.0);
    p = fma(p, z, -0.5);
    return fma(z, p, 1.0);
}
// sin, cos of 2 pi (w + 1/2) / 2^32
IMS_DEV void sincos2pi_w(uint32_t w, double& s, double& c)
{
    const uint32_t q = (w + 0x20000000u) >> 30;              // nearest quarter turn, modulo 4
    const int32_t ri = (int32_t)(w - (q << 30));             // offset from it in units of 2^-32 turns, [-2^29, 2^29)
    const double t = fma((double)ri, 0x1.0p-32, 0x1.0p-33) * TWO_PI;
    const double z = t * t;
    const double sk = sin_kernel_w(t, z), ck = cos_kernel_w(z);
    const bool odd = (q & 1u) != 0u;
    const double s0 = odd ? ck : sk, c0 = odd ? sk : ck;
    const unsigned long long sbit = (unsigned long long)(q & 2u) << 62;               // q = 2, 3: sin negative
    const unsigned long long cbit = (unsigned long long)((q + 1u) & 2u) << 62;        // q = 1, 2: cos negative
    s = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(s0) ^ sbit));
    c = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(c0) ^ cbit));
}
// log((w + 1/2) / 2^32)
IMS_DEV double dlog_w(uint32_t w)
{
    const double x = w01(w);
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    int64_t e = (int64_t)((b >> 52) & 0x7FF) - 1023;
    double m = __longlong_as_double((long long)((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double s = ddiv(m - 1.0, m + 1.0);
    const double z = s * s;
    double p = 1.0 / 13.0;
    p = fma_k(p, z, 1.0 / 11.0);
    p = fma_k(p, z, 1.0 / 9.0);
    p = fma_k(p, z, 1.0 / 7.0);
    p = fma_k(p, z, 1.0 / 5.0);
    p = fma_k(p, z, 1.0 / 3.0);
    p = fma(p, z, 1.0);
    const double lm = 2.0 * s * p;
    const double de = (double)e;
    return fma(de, LN2_HI, fma(de, LN2_LO, lm));
}

IMS_DEV void gauss_words(uint32_t w0, uint32_t w1, double& g0, double& g1)
{
    const double r = dsqrt_n(-2.0 * dlog_w(w0));      // w01 < 1: the argument is > 0
    double s, c;
    sincos2pi_w(w1, s, c);
    g0 = r * c; g1 = r * s;
}

// one Gaussian deviate from two words: the cosine half of the pair above, bit for bit
IMS_DEV double gauss_word_cos(uint32_t w0, uint32_t w1)
{
    const double r = dsqrt_n(-2.0 * dlog_w(w0));
    const uint32_t q = (w1 + 0x20000000u) >> 30;
    const int32_t ri = (int32_t)(w1 - (q << 30));
    const double t = fma((double)ri, 0x1.0p-32, 0x1.0p-33) * TWO_PI;
    const double z = t * t;
    const double c0 = (q & 1u) ? sin_kernel_w(t, z) : cos_kernel_w(z);
    const unsigned long long cbit = (unsigned long long)((q + 1u) & 2u) << 62;
    return r * __longlong_as_double((long long)((unsigned long long)__double_as_longlong(c0) ^ cbit));
}

// RNG slots and word assignment (spec v6)
constexpr uint32_t SLOT_SHOOT = 0;        // w0 wavelength, w1 profile radius, w2 profile angle
constexpr uint32_t SLOT_KNOT = 1;         // photon index = knot index: w0,w1 Gaussian position of a RandomKnots point
constexpr uint32_t SLOT_PSF = 2;          // + (component >> 1); component c owns words 2(c&1), 2(c&1)+1
constexpr uint32_t SLOT_OP = 8;           // + (op index >> 1); op k owns words 2(k&1), 2(k&1)+1
constexpr uint32_t SLOT_PSF_TIME = 20;    // + component: w0 arrival time drawn by a phase-screen PSF
constexpr uint32_t SLOT_SENSOR = 24;      // w0,w1 diffusion pair, w2 conversion depth, w3 pixel-not-found coin

}  // namespace ims
