// ims_fft.h -- device functions of the FFT branch (LSST_SiliconBuilder.draw, method 'fft',
// imsim/stamp.py:482-525): analytic k-space values and the Poisson deviate of the noise step.
#pragma once
#include "ims_math.h"
#include "../../include/imsim_hip.h"

namespace ims {

// radial k-table: linear interpolation, 0 beyond the tabulated range
IMS_DEV double ktable_lookup(const ims_lin_tables_t& t, int table, double arg)
{
    const double* v = t.val + (int64_t)table * t.n_pts;
    const double f = (arg - t.arg_min) / t.arg_step;
    if (!(f > 0.0)) return v[0];
    if (f >= (double)(t.n_pts - 1)) return 0.0;
    const int i = (int)f;
    const double a = f - (double)i;
    return v[i] + a * (v[i + 1] - v[i]);
}

// spectrum of object o (profile x PSF MTFs x pixel response x centring phase) at the k-vector (kx, ky) [rad/arcsec]
IMS_DEV void kspace_at(const ims_fft_params_t& P, const ims_fft_object_t& o, double kx, double ky, double& re, double& im)
{
    // profile: transformed k-vector J^T k
    double amp = o.flux;
    if (o.prof_ktable >= 0) {
        const double qx = o.jac[0] * kx + o.jac[2] * ky;
        const double qy = o.jac[1] * kx + o.jac[3] * ky;
        amp = amp * ktable_lookup(P.ktables, o.prof_ktable, sqrt(qx * qx + qy * qy) * o.prof_scale);
    }
    const double k2 = kx * kx + ky * ky;
    for (int c = 0; c < P.n_kpsf; ++c) {
        const ims_kpsf_t& p = P.kpsf[c];
        if (p.kind == IMS_KPSF_GAUSSIAN) amp = amp * dexp(-0.5 * p.p0 * p.p0 * k2);
        else if (p.kind == IMS_KPSF_KOLMOGOROV) {
            if (k2 > 0.0) amp = amp * dexp(-dpow(sqrt(k2) / p.p0, 5.0 / 3.0));
        } else amp = amp * ktable_lookup(P.ktables, p.table, sqrt(k2) * p.p0);
    }
    // pixel response: sinc(kx s / 2) sinc(ky s / 2)
    const double hx = 0.5 * kx * P.pixel_scale, hy = 0.5 * ky * P.pixel_scale;
    double s, c;
    if (hx != 0.0) { dsincos(hx, s, c); amp = amp * (s / hx); }
    if (hy != 0.0) { dsincos(hy, s, c); amp = amp * (s / hy); }
    // centre at (cx, cy) pixels: phase exp(-i (kx cx + ky cy) s)
    const double ph = (kx * o.cx + ky * o.cy) * P.pixel_scale;
    dsincos(ph, s, c);
    re = amp * c; im = -amp * s;
}

// The spectrum at (kx, ky) AND at (kx, -ky) -- rows i and n - i of a half spectrum -- in one go.  The PSF factors depend on
// kx^2 + ky^2 and the pixel response along x on kx alone: the same values for both rows ((-ky) (-ky) is ky ky exactly), formed
// once.  What differs -- the sheared profile's radius, the pixel response along y (dsincos is not odd bit for bit), the
// centring phase -- is formed per row by kspace_at's operations in kspace_at's order: (re, im) and (re2, im2) are the bits
// kspace_at(kx, ky) and kspace_at(kx, -ky) return.  (The k-space fill was three exponentials, a logarithm and three sine / cosine
// pairs per point: 27 % of the FFT branch at throughput, all arithmetic; round 6.)
// tab != nullptr: the three pixel-response factors (s / h, or 1 where h is 0 -- a product with 1 is the value itself) come from the
// caller's tables {along x; along y for ky; along y for -ky}, formed by pixel_response below: the same function of the same argument.
IMS_DEV double pixel_response(double k, double pixel_scale)
{
    const double h = 0.5 * k * pixel_scale;
    if (h == 0.0) return 1.0;
    double s, c;
    dsincos(h, s, c);
    return s / h;
}

IMS_DEV void kspace_pair(const ims_fft_params_t& P, const ims_fft_object_t& o, double kx, double ky, double& re, double& im,
                         double& re2, double& im2, const double* tab = nullptr)
{
    const double kym = -ky;
    double amp = o.flux, amp2 = o.flux;
    if (o.prof_ktable >= 0) {
        const double qx = o.jac[0] * kx + o.jac[2] * ky;
        const double qy = o.jac[1] * kx + o.jac[3] * ky;
        amp = amp * ktable_lookup(P.ktables, o.prof_ktable, sqrt(qx * qx + qy * qy) * o.prof_scale);
        const double qx2 = o.jac[0] * kx + o.jac[2] * kym;
        const double qy2 = o.jac[1] * kx + o.jac[3] * kym;
        amp2 = amp2 * ktable_lookup(P.ktables, o.prof_ktable, sqrt(qx2 * qx2 + qy2 * qy2) * o.prof_scale);
    }
    const double k2 = kx * kx + ky * ky;
    for (int c = 0; c < P.n_kpsf; ++c) {
        const ims_kpsf_t& p = P.kpsf[c];
        if (p.kind == IMS_KPSF_GAUSSIAN) {
            const double g = dexp(-0.5 * p.p0 * p.p0 * k2);
            amp = amp * g; amp2 = amp2 * g;
        } else if (p.kind == IMS_KPSF_KOLMOGOROV) {
            if (k2 > 0.0) {
                const double g = dexp(-dpow(sqrt(k2) / p.p0, 5.0 / 3.0));
                amp = amp * g; amp2 = amp2 * g;
            }
        } else {
            const double g = ktable_lookup(P.ktables, p.table, sqrt(k2) * p.p0);
            amp = amp * g; amp2 = amp2 * g;
        }
    }
    double s, c;
    if (tab != nullptr) {
        amp = amp * tab[0]; amp2 = amp2 * tab[0];
        amp = amp * tab[1]; amp2 = amp2 * tab[2];
    } else {
        const double hx = 0.5 * kx * P.pixel_scale, hy = 0.5 * ky * P.pixel_scale, hy2 = 0.5 * kym * P.pixel_scale;
        if (hx != 0.0) { dsincos(hx, s, c); const double sx = s / hx; amp = amp * sx; amp2 = amp2 * sx; }
        if (hy != 0.0) { dsincos(hy, s, c); amp = amp * (s / hy); }
        if (hy2 != 0.0) { dsincos(hy2, s, c); amp2 = amp2 * (s / hy2); }
    }
    const double ph = (kx * o.cx + ky * o.cy) * P.pixel_scale;
    dsincos(ph, s, c);
    re = amp * c; im = -amp * s;
    const double ph2 = (kx * o.cx + kym * o.cy) * P.pixel_scale;
    dsincos(ph2, s, c);
    re2 = amp2 * c; im2 = -amp2 * s;
}

// value of the half spectrum of object o at grid index (i = ky index, j = kx index).  The image is the profile
// convolved with the pixel, SAMPLED at the pixel centres: its discrete spectrum is the continuous one folded at the
// sampling frequency 2 pi / pixel_scale.  P.n_alias = m adds the (2m+1)^2 - 1 nearest aliases (rows of b ascending, a
// ascending inside); 0 = base band only, right whenever the PSF's MTF is negligible at the Nyquist frequency.
IMS_DEV void kspace_value(const ims_fft_params_t& P, const ims_fft_object_t& o, int i, int j, double& re, double& im)
{
    const int n = o.nfft;
    const double dk = TWO_PI / ((double)n * P.pixel_scale);          // rad / arcsec
    const double kx = (double)j * dk;
    const double ky = (double)(i < n / 2 ? i : i - n) * dk;
    if (P.n_alias <= 0) { kspace_at(P, o, kx, ky, re, im); return; }
    const double ks = TWO_PI / P.pixel_scale;
    double sr = 0.0, si = 0.0;
    for (int b = -P.n_alias; b <= P.n_alias; ++b)
        for (int a = -P.n_alias; a <= P.n_alias; ++a) {
            double r1, i1;
            kspace_at(P, o, kx + (double)a * ks, ky + (double)b * ks, r1, i1);
            sr = sr + r1; si = si + i1;
        }
    re = sr; im = si;
}

// log Gamma(x) for x > 0 by upward recurrence to x >= 8 and the Stirling series
IMS_DEV double dlgamma(double x)
{
    double shift = 0.0;
    while (x < 8.0) { shift = shift + dlog(x); x = x + 1.0; }
    const double ix = 1.0 / x, ix2 = ix * ix;
    double ser = -1.0 / 1680.0;
    ser = fma(ser, ix2, 1.0 / 1260.0);
    ser = fma(ser, ix2, -1.0 / 360.0);
    ser = fma(ser, ix2, 1.0 / 12.0);
    return (x - 0.5) * dlog(x) - x + 0.91893853320467274178 + ser * ix - shift;
}

// Poisson deviate, counter-addressed by (seed, object id, pixel): multiplication method below
// mean 10, Hormann's PTRS transformed rejection above
IMS_DEV double poisson(double mean, uint64_t seed, int64_t obj_id, int64_t pixel)
{
    if (!(mean > 0.0)) return 0.0;
    uint32_t slot = 32;
    if (mean < 10.0) {
        const double L = dexp(-mean);
        double p = 1.0;
        double k = 0.0;
        for (int it = 0; it < 64; ++it) {
            const Draw d = draw(seed, obj_id, pixel, slot++);
            p = p * u01_open(d.a);
            if (p <= L) return k;
            k = k + 1.0;
            p = p * u01_open(d.b);
            if (p <= L) return k;
            k = k + 1.0;
        }
        return k;
    }
    const double slam = sqrt(mean), loglam = dlog(mean);
    const double b = 0.931 + 2.53 * slam;
    const double a = -0.059 + 0.02483 * b;
    const double invalpha = 1.1239 + 1.1328 / (b - 3.4);
    const double vr = 0.9277 - 3.6224 / (b - 2.0);
    for (int it = 0; it < 64; ++it) {
        const Draw d = draw(seed, obj_id, pixel, slot++);
        const double U = u01(d.a) - 0.5;
        const double V = u01_open(d.b);
        const double us = 0.5 - fabs(U);
        const double k = floor((2.0 * a / us + b) * U + mean + 0.43);
        if (us >= 0.07 && V <= vr) return k;
        if (k < 0.0 || (us < 0.013 && V > us)) continue;
        if (dlog(V) + dlog(invalpha) - dlog(a / (us * us) + b) <= -mean + k * loglam - dlgamma(k + 1.0)) return k;
    }
    return floor(mean + 0.5);
}

// un-normalised spike stencil at integer offset (a rows, b columns)
// (prepare_psf_field_rotation, imsim/diffraction_fft.py:78-123, evaluated analytically)
IMS_DEV double spike_stencil(const ims_spikes_t& k, int a, int b)
{
    const double x = (double)a, y = (double)b;
    const double xr = k.cos0 * x + k.sin0 * y;
    const double yr = -k.sin0 * x + k.cos0 * y;
    const double m = fabs(xr) < fabs(yr) ? fabs(xr) : fabs(yr);
    // Almost every offset of the (2 cutoff + 1)^2 stencil is an exact zero: more than a pixel from both arms of the cross
    // (val = 0 below) AND outside the wedge the field rotation sweeps, whose points lie within r sin(d_alpha / 2) of an arm.
    // Those return before the three arctangents -- same value (+0.0), so the sums that skip them are the same sums.
    if (m > 1.0) {
        const double t = m - 1.0e-6, lim = 0.5 * fabs(k.d_alpha) + 1.0e-6;
        if (t * t > lim * lim * (x * x + y * y)) return 0.0;
    }
    double val = 1.0 - m;
    if (val < 0.0) val = 0.0;
    const double half_pi = PI_2;
    double dth = datan2(y, x) - k.a_lo;
    dth = dth - floor(dth / half_pi) * half_pi;
    if (dth <= k.d_alpha) val = 1.0;
    const double r = sqrt(x * x + y * y);
    const double prof = 0.63661977236758134308 * (datan((r + 0.5) * k.scale / k.r0) - datan((r - 0.5) * k.scale / k.r0));
    const double arc = r * k.d_alpha;
    val = val * prof / (arc > 1.0 ? arc : 1.0);
    if (a == 0 && b == 0) val = 2.0 * val;
    return val;
}

}  // namespace ims
