/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * orc_math.h: CPU statement of the numerics spec (DESIGN.md "Numerics spec"): Philox4x32-10 and the
 * binary64 elementary functions built only from + - * / fma sqrt and integer operations, so that
 * a CPU and a GPU evaluation give identical bits.  Compile with -ffp-contract=off -mfma.
 *
 * Parity status: the reference's RNG is GalSim's BaseDeviate (boost mt19937; stamp.py:166,
 * photon_ops.py:264-272) which is not reproducible here -> "parity unpinned" at the bit level for
 * random streams; distributions are pinned statistically in tests/.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H
#include <stdint.h>
#include <string.h>
#include <math.h>

static inline double orc_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
static inline double orc_sqrt(double a) { return __builtin_sqrt(a); }
static inline uint64_t orc_bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double orc_from_bits(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

/* ---------------- Philox4x32-10 (Salmon et al. 2011) ---------------- */
static inline void orc_philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

/* Random draw addressed by (seed, object id, photon index, slot): two 53-bit integers. */
typedef struct { uint64_t a, b; } orc_draw_t;
static inline orc_draw_t orc_draw(uint64_t seed, int64_t obj_id, int64_t photon, uint32_t slot)
{
    uint32_t c[4];
    c[0] = (uint32_t)((uint64_t)photon);
    c[1] = (uint32_t)((uint64_t)photon >> 32);
    c[2] = slot;
    c[3] = (uint32_t)((uint64_t)obj_id);
    uint32_t k0 = (uint32_t)seed;
    uint32_t k1 = (uint32_t)(seed >> 32) ^ (uint32_t)((uint64_t)obj_id >> 32);
    orc_philox4x32_10(c, k0, k1);
    orc_draw_t d;
    d.a = (((uint64_t)c[0] << 32) | c[1]) >> 11;
    d.b = (((uint64_t)c[2] << 32) | c[3]) >> 11;
    return d;
}
/* The photon pipeline (since spec v4) consumes the raw block: four 32-bit words per (object, photon, slot),
 * each mapped to the open interval (0,1) -- the granularity of GalSim's own UniformDeviate. */
typedef struct { uint32_t w[4]; } orc_words_t;
static inline orc_words_t orc_words(uint64_t seed, int64_t obj_id, int64_t photon, uint32_t slot)
{
    uint32_t c[4];
    c[0] = (uint32_t)((uint64_t)photon);
    c[1] = (uint32_t)((uint64_t)photon >> 32);
    c[2] = slot;
    c[3] = (uint32_t)((uint64_t)obj_id);
    uint32_t k0 = (uint32_t)seed;
    uint32_t k1 = (uint32_t)(seed >> 32) ^ (uint32_t)((uint64_t)obj_id >> 32);
    orc_philox4x32_10(c, k0, k1);
    orc_words_t r;
    r.w[0] = c[0]; r.w[1] = c[1]; r.w[2] = c[2]; r.w[3] = c[3];
    return r;
}
/* (w + 1/2) / 2^32, exact */
static inline double orc_w01(uint32_t w) { return orc_fma((double)w, 0x1.0p-32, 0x1.0p-33); }

/* [0,1) */
static inline double orc_u01(uint64_t k) { return (double)k * 0x1.0p-53; }
/* (0,1] for logarithms */
static inline double orc_u01_open(uint64_t k) { return (double)(k + 1) * 0x1.0p-53; }

/* ---------------- elementary functions ---------------- */
#define ORC_LN2_HI 6.93147180369123816490e-01
#define ORC_LN2_LO 1.90821492927058770002e-10
#define ORC_INV_LN2 1.44269504088896338700e+00
#define ORC_TWO_PI 6.283185307179586476925
#define ORC_INV_TWO_PI 0.15915494309189533577
#define ORC_PI 3.14159265358979323846
#define ORC_PI_2 1.57079632679489661923
#define ORC_PI_4 0.78539816339744830962

/* natural log, x > 0 normal */
static inline double orc_log(double x)
{
    uint64_t b = orc_bits(x);
    int64_t e = (int64_t)((b >> 52) & 0x7FF) - 1023;
    double m = orc_from_bits((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    double p = 1.0 / 25.0;
    p = orc_fma(p, z, 1.0 / 23.0);
    p = orc_fma(p, z, 1.0 / 21.0);
    p = orc_fma(p, z, 1.0 / 19.0);
    p = orc_fma(p, z, 1.0 / 17.0);
    p = orc_fma(p, z, 1.0 / 15.0);
    p = orc_fma(p, z, 1.0 / 13.0);
    p = orc_fma(p, z, 1.0 / 11.0);
    p = orc_fma(p, z, 1.0 / 9.0);
    p = orc_fma(p, z, 1.0 / 7.0);
    p = orc_fma(p, z, 1.0 / 5.0);
    p = orc_fma(p, z, 1.0 / 3.0);
    p = orc_fma(p, z, 1.0);
    double lm = 2.0 * s * p;
    double de = (double)e;
    return orc_fma(de, ORC_LN2_HI, orc_fma(de, ORC_LN2_LO, lm));
}

/* exp, |x| < 700 */
static inline double orc_exp(double x)
{
    double kf = floor(orc_fma(x, ORC_INV_LN2, 0.5));
    double r = orc_fma(-kf, ORC_LN2_HI, x);
    r = orc_fma(-kf, ORC_LN2_LO, r);
    double p = 1.0 / 6227020800.0;            /* 1/13! */
    p = orc_fma(p, r, 1.0 / 479001600.0);     /* 1/12! */
    p = orc_fma(p, r, 1.0 / 39916800.0);
    p = orc_fma(p, r, 1.0 / 3628800.0);
    p = orc_fma(p, r, 1.0 / 362880.0);
    p = orc_fma(p, r, 1.0 / 40320.0);
    p = orc_fma(p, r, 1.0 / 5040.0);
    p = orc_fma(p, r, 1.0 / 720.0);
    p = orc_fma(p, r, 1.0 / 120.0);
    p = orc_fma(p, r, 1.0 / 24.0);
    p = orc_fma(p, r, 1.0 / 6.0);
    p = orc_fma(p, r, 0.5);
    p = orc_fma(p, r, 1.0);
    p = orc_fma(p, r, 1.0);
    int64_t k = (int64_t)kf;
    if (k < -1000) return 0.0;
    if (k > 1000) k = 1000;
    double scale = orc_from_bits((uint64_t)(k + 1023) << 52);
    return p * scale;
}

/* kernels on |t| <= pi/4 */
static inline double orc_sin_kernel(double t)
{
    double z = t * t;
    double p = 1.0 / 355687428096000.0;           /* 1/17! */
    p = orc_fma(p, z, -1.0 / 1307674368000.0);    /* -1/15! */
    p = orc_fma(p, z, 1.0 / 6227020800.0);        /* 1/13! */
    p = orc_fma(p, z, -1.0 / 39916800.0);         /* -1/11! */
    p = orc_fma(p, z, 1.0 / 362880.0);            /* 1/9! */
    p = orc_fma(p, z, -1.0 / 5040.0);
    p = orc_fma(p, z, 1.0 / 120.0);
    p = orc_fma(p, z, -1.0 / 6.0);
    return orc_fma(t * z, p, t);
}
static inline double orc_cos_kernel(double t)
{
    double z = t * t;
    double p = -1.0 / 6402373705728000.0;         /* -1/18! */
    p = orc_fma(p, z, 1.0 / 20922789888000.0);    /* 1/16! */
    p = orc_fma(p, z, -1.0 / 87178291200.0);      /* -1/14! */
    p = orc_fma(p, z, 1.0 / 479001600.0);         /* 1/12! */
    p = orc_fma(p, z, -1.0 / 3628800.0);          /* -1/10! */
    p = orc_fma(p, z, 1.0 / 40320.0);
    p = orc_fma(p, z, -1.0 / 720.0);
    p = orc_fma(p, z, 1.0 / 24.0);
    p = orc_fma(p, z, -0.5);
    return orc_fma(z, p, 1.0);
}
/* sin and cos of 2*pi*u for u in [0,1] */
static inline void orc_sincos2pi(double u, double* s, double* c)
{
    double qf = floor(orc_fma(4.0, u, 0.5));
    double r = orc_fma(-0.25, qf, u);
    double t = r * ORC_TWO_PI;
    double sk = orc_sin_kernel(t), ck = orc_cos_kernel(t);
    int q = (int)qf & 3;
    if (q == 0) { *s = sk; *c = ck; }
    else if (q == 1) { *s = ck; *c = -sk; }
    else if (q == 2) { *s = -sk; *c = -ck; }
    else { *s = -ck; *c = sk; }
}
static inline void orc_sincos(double x, double* s, double* c)
{
    double u = x * ORC_INV_TWO_PI;
    u = u - floor(u);
    orc_sincos2pi(u, s, c);
}

static inline double orc_atan(double x)
{
    double ax = x < 0.0 ? -x : x;
    double base = 0.0, sign = 1.0;
    double a = ax;
    if (ax > 1.0) { a = 1.0 / ax; base = ORC_PI_2; sign = -1.0; }
    double off = 0.0;
    double b = a;
    if (a > 0.41421356237309503) { b = (a - 1.0) / (a + 1.0); off = ORC_PI_4; }
    double cc = b / (1.0 + orc_sqrt(orc_fma(b, b, 1.0)));
    double z = cc * cc;
    double p = 1.0 / 27.0;
    p = orc_fma(p, z, -1.0 / 25.0);
    p = orc_fma(p, z, 1.0 / 23.0);
    p = orc_fma(p, z, -1.0 / 21.0);
    p = orc_fma(p, z, 1.0 / 19.0);
    p = orc_fma(p, z, -1.0 / 17.0);
    p = orc_fma(p, z, 1.0 / 15.0);
    p = orc_fma(p, z, -1.0 / 13.0);
    p = orc_fma(p, z, 1.0 / 11.0);
    p = orc_fma(p, z, -1.0 / 9.0);
    p = orc_fma(p, z, 1.0 / 7.0);
    p = orc_fma(p, z, -1.0 / 5.0);
    p = orc_fma(p, z, 1.0 / 3.0);
    p = orc_fma(p, z, -1.0);
    /* atan(cc) = cc - cc^3/3 + ... = -cc * p  */
    double at = -(cc * p);
    double res = orc_fma(2.0, at, off);          /* atan(a) */
    res = orc_fma(sign, res, base);              /* atan(ax) */
    return x < 0.0 ? -res : res;
}

static inline double orc_atan2(double y, double x)
{
    if (x > 0.0) return orc_atan(y / x);
    if (x < 0.0) return y >= 0.0 ? orc_atan(y / x) + ORC_PI : orc_atan(y / x) - ORC_PI;
    if (y > 0.0) return ORC_PI_2;
    if (y < 0.0) return -ORC_PI_2;
    return 0.0;
}

static inline double orc_tanh_pos(double x)   /* x >= 0 */
{
    if (x > 20.0) return 1.0;
    double e = orc_exp(-2.0 * x);
    return (1.0 - e) / (1.0 + e);
}
static inline double orc_pow(double x, double y) { return orc_exp(y * orc_log(x)); }

/* Box-Muller pair from one draw */
static inline void orc_gauss_pair(orc_draw_t d, double* g0, double* g1)
{
    double u1 = orc_u01_open(d.a);
    double u2 = orc_u01(d.b);
    double r = orc_sqrt(-2.0 * orc_log(u1));
    double s, c;
    orc_sincos2pi(u2, &s, &c);
    *g0 = r * c; *g1 = r * s;
}

/* Deviates from 32-bit words (spec v6): the argument (w + 1/2) / 2^32 has a granularity of 2^-32, so the series stop where
 * the truncation error is below 2^-36 of the result, and the quadrant of the angle comes from the word's top bits. */
static inline double orc_sin_kernel_w(double t, double z)
{
    double p = -1.0 / 39916800.0;                 /* -1/11! */
    p = orc_fma(p, z, 1.0 / 362880.0);
    p = orc_fma(p, z, -1.0 / 5040.0);
    p = orc_fma(p, z, 1.0 / 120.0);
    p = orc_fma(p, z, -1.0 / 6.0);
    return orc_fma(t * z, p, t);
}
static inline double orc_cos_kernel_w(double z)
{
    double p = 1.0 / 479001600.0;                 /* 1/12! */
    p = orc_fma(p, z, -1.0 / 3628800.0);
    p = orc_fma(p, z, 1.0 / 40320.0);
    p = orc_fma(p, z, -1.0 / 720.0);
    p = orc_fma(p, z, 1.0 / 24.0);
    p = orc_fma(p, z, -0.5);
    return orc_fma(z, p, 1.0);
}
/* quarter turn nearest to w / 2^32 turns (mod 4) and the angle left over, |t| <= pi/4 */
static inline uint32_t orc_reduce_w(uint32_t w, double* t)
{
    uint32_t q = (uint32_t)(w + 0x20000000u) >> 30;
    int32_t ri = (int32_t)(uint32_t)(w - (q << 30));
    *t = orc_fma((double)ri, 0x1.0p-32, 0x1.0p-33) * ORC_TWO_PI;
    return q;
}
static inline void orc_sincos2pi_w(uint32_t w, double* s, double* c)
{
    double t;
    uint32_t q = orc_reduce_w(w, &t);
    double z = t * t;
    double sk = orc_sin_kernel_w(t, z), ck = orc_cos_kernel_w(z);
    if (q == 0) { *s = sk; *c = ck; }
    else if (q == 1) { *s = ck; *c = -sk; }
    else if (q == 2) { *s = -sk; *c = -ck; }
    else { *s = -ck; *c = sk; }
}
/* log((w + 1/2) / 2^32) */
static inline double orc_log_w(uint32_t w)
{
    double x = orc_w01(w);
    uint64_t b = orc_bits(x);
    int64_t e = (int64_t)((b >> 52) & 0x7FF) - 1023;
    double m = orc_from_bits((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    double p = 1.0 / 13.0;
    p = orc_fma(p, z, 1.0 / 11.0);
    p = orc_fma(p, z, 1.0 / 9.0);
    p = orc_fma(p, z, 1.0 / 7.0);
    p = orc_fma(p, z, 1.0 / 5.0);
    p = orc_fma(p, z, 1.0 / 3.0);
    p = orc_fma(p, z, 1.0);
    double lm = 2.0 * s * p;
    double de = (double)e;
    return orc_fma(de, ORC_LN2_HI, orc_fma(de, ORC_LN2_LO, lm));
}

/* Box-Muller pair from two words */
static inline void orc_gauss_words(uint32_t w0, uint32_t w1, double* g0, double* g1)
{
    double r = orc_sqrt(-2.0 * orc_log_w(w0));
    double s, c;
    orc_sincos2pi_w(w1, &s, &c);
    *g0 = r * c; *g1 = r * s;
}
/* its cosine half alone (one deviate from two words) */
static inline double orc_gauss_word_cos(uint32_t w0, uint32_t w1)
{
    double s, c;
    orc_sincos2pi_w(w1, &s, &c);
    return orc_sqrt(-2.0 * orc_log_w(w0)) * c;
}

/* RNG slots and word assignment (spec v6) */
#define ORC_SLOT_SHOOT     0   /* w0 wavelength, w1 profile radius, w2 profile angle */
#define ORC_SLOT_KNOT      1   /* photon index = knot index: w0,w1 Gaussian position of a RandomKnots point */
#define ORC_SLOT_PSF       2   /* + (component >> 1); component c owns words 2(c&1), 2(c&1)+1 */
#define ORC_SLOT_OP        8   /* + (op index >> 1); op k owns words 2(k&1), 2(k&1)+1 */
#define ORC_SLOT_PSF_TIME 20   /* + component: w0 arrival time drawn by a phase-screen PSF */
#define ORC_SLOT_SENSOR   24   /* w0,w1 diffusion pair, w2 conversion depth, w3 pixel-not-found coin */

#endif
