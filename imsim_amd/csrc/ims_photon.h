// ims_photon.h -- per-photon device functions of the stamp-rendering hot path (gfx950).
//
// One photon lives in registers from profile sampling to the pixel it lands in; nothing in here
// touches memory except read-only tables, the pixel-boundary state and the final atomics.
// Reference behaviour restated (imSim tree): stamp.py:527-573 (phot branch), photon_ops.py
// (RubinOptics/RubinDiffraction/RubinDiffractionOptics, XyToV, ray_vector_to_photon_array),
// diffraction.py (spider kick), config/imsim-config.yaml:281-320 (op chain), and GalSim's
// SiliconSensor as described in doc/validation/{brighter-fatter,diffusion,tree-ring}.rst.
#pragma once
#include "ims_math.h"
#include "../../include/imsim_hip.h"

namespace ims {

struct Photon {
    double x, y, flux, dxdz, dydz, wl, pu, pv, t;
};

// ---------------- tables ----------------
IMS_DEV double lin_lookup(const ims_lin_tables_t& t, int table, double arg)
{
    const double* v = t.val + (int64_t)table * t.n_pts;
    const double f = ddiv(arg - t.arg_min, t.arg_step);
    if (!(f > 0.0)) return v[0];
    const int n = t.n_pts;
    if (f >= (double)(n - 1)) return v[n - 1];
    const int i = (int)f;
    const double a = f - (double)i;
    return v[i] + a * (v[i + 1] - v[i]);
}

IMS_DEV double radial_r2(const ims_radial_tables_t& t, int table, double u)
{
    const int nb = t.n_bins;
    const double* cdf = t.cdf + (int64_t)table * (nb + 1);
    const double* r2 = t.r2 + (int64_t)table * (nb + 1);
    int lo = 0, hi = nb;
    if (t.guide != nullptr) {
        // cdf[guide[g]] <= g/n_guide <= u < (g+1)/n_guide < cdf[guide[g+1] + 1]: the bisection invariant holds on the
        // narrowed range, so it ends in the same bin (usually after 0-2 steps instead of 9)
        const int32_t* gd = t.guide + (int64_t)table * (t.n_guide + 1);
        const int g = (int)(u * (double)t.n_guide);
        lo = gd[g];
        const int h = gd[g + 1] + 1;
        hi = h < nb ? h : nb;
    }
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] <= u) lo = mid; else hi = mid;
    }
    const double c0 = cdf[lo];
    const double w = cdf[lo + 1] - c0;
    const double f = (w > 0.0) ? ddiv(u - c0, w) : 0.0;
    const double a = r2[lo];
    return a + f * (r2[lo + 1] - a);
}

// pixel of an image profile by inverse CDF; position in image pixels relative to the image centre.  (In line: a real
// call in a photon kernel costs the whole kernel its register allocation -- measured 2.9 -> 0.7 M objects/s.)
// offset drawn from |K| of the image's interpolant (ims_image_tables_t.kx, kcdf); neg: K is negative there
IMS_DEV double interp_offset(const ims_image_tables_t& T, double u, bool& neg)
{
    int lo = 0, hi = T.n_k;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (T.kcdf[mid] <= u) lo = mid; else hi = mid;
    }
    const double c0 = T.kcdf[lo], wd = T.kcdf[lo + 1] - c0;
    const double f = (wd > 0.0) ? (u - c0) / wd : 0.0;
    const double x0 = T.kx[lo];
    const double d = fma(f, T.kx[lo + 1] - x0, x0);
    const double ad = fabs(d);
    neg = (ad > T.neg[0] && ad < T.neg[1]) || (ad > T.neg[2] && ad < T.neg[3]);
    return d;
}

// fscale: the factor on the photon's flux (1 without an interpolant)
IMS_DEV void image_sample(const ims_image_tables_t& T, int k, double u, double u2, double& gx, double& gy, double& fscale)
{
    const int w = T.size[2 * k], h = T.size[2 * k + 1];
    const double* cdf = T.cdf + T.offset[k];
    int lo = 0, hi = w * h;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] <= u) lo = mid; else hi = mid;
    }
    const double c0 = cdf[lo], wd = cdf[lo + 1] - c0;
    const double f = (wd > 0.0) ? (u - c0) / wd : 0.0;
    const int px = lo % w, py = lo / w;
    if (T.interp != 0) {
        bool nx, ny;
        const double dx = interp_offset(T, f, nx), dy = interp_offset(T, u2, ny);
        gx = (((double)px + 0.5) + dx) - 0.5 * (double)w;
        gy = (((double)py + 0.5) + dy) - 0.5 * (double)h;
        fscale = (nx != ny) ? -T.norm : T.norm;
    } else {
        gx = ((double)px + f) - 0.5 * (double)w;
        gy = ((double)py + u2) - 0.5 * (double)h;
        fscale = 1.0;
    }
}

// ---------------- shooting ----------------
// photon k of object `o`: wavelength + profile sample, relative to image_pos, in pixels
IMS_DEV void shoot(const ims_render_params_t& P, const ims_object_t& o, int64_t k, Rng& rng, Photon& ph)
{
    rng_block(rng, P.seed, o.obj_id, k, SLOT_SHOOT);
    double wl = o.sed_wave;
    if (o.sed_table >= 0) wl = lin_lookup(P.sed, o.sed_table, w01(rng.w[0]));
    double pu = 0.0, pv = 0.0, fscale = 1.0;
    if (o.prof_table != IMS_PROF_POINT) {
        double gu, gv;
        if (o.prof_table >= 0) {
            const double r2 = radial_r2(P.radial, o.prof_table, w01(rng.w[1]));
            const double r = dsqrt0(r2) * o.prof_scale;
            double s, c;
            sincos2pi_w(rng.w[2], s, c);
            gu = r * c; gv = r * s;
        } else if (o.prof_table == IMS_PROF_BOX) {
            gu = (w01(rng.w[1]) - 0.5) * o.prof_scale;
            gv = (w01(rng.w[2]) - 0.5) * o.prof_aux;
        } else if (o.prof_table == IMS_PROF_IMAGE) {
            image_sample(P.images, (int)o.prof_aux, w01(rng.w[1]), w01(rng.w[2]), gu, gv, fscale);
            gu = gu * o.prof_scale; gv = gv * o.prof_scale;
        } else {
            // RandomKnots: the photon picks one of the knots; knot m sits at a Gaussian deviate addressed by
            // (object, m) in the knot slot, the same for every photon of the object
            const uint32_t m = __umulhi(rng.w[1], (uint32_t)o.prof_aux);
            Rng kr;
            rng_reset(kr);
            rng_block(kr, P.seed, o.obj_id, (int64_t)m, SLOT_KNOT);
            double g0, g1;
            gauss_words(kr.w[0], kr.w[1], g0, g1);
            gu = o.prof_scale * g0; gv = o.prof_scale * g1;
        }
        pu = o.jac[0] * gu + o.jac[1] * gv;
        pv = o.jac[2] * gu + o.jac[3] * gv;
    }
    ph.x = o.winv[0] * pu + o.winv[1] * pv;
    ph.y = o.winv[2] * pu + o.winv[3] * pv;
    ph.flux = (o.prof_table == IMS_PROF_IMAGE) ? o.flux_per_photon * fscale : o.flux_per_photon;
    ph.dxdz = 0.0; ph.dydz = 0.0;
    ph.wl = wl;
    ph.pu = 0.0; ph.pv = 0.0; ph.t = 0.0;
}

IMS_DEV int wrap_index(double fl, double dn, double inv_n)
{
    // fl mod n for an integer-valued double (exact: |fl| < 2^53); the estimate of the quotient may be off by one
    double r = fl - floor(fl * inv_n) * dn;
    if (r < 0.0) r = r + dn;
    if (r >= dn) r = r - dn;
    return (int)r;
}

typedef double dvec2q __attribute__((ext_vector_type(2)));
constexpr int SCREEN_BATCH = 6;   // layers of imSim's atmosphere (atmPSF.py:179: six altitudes)
// sum over layers of the gradient of the bilinear interpolant of the periodic phase screens [nm/m]
// PLAIN: always the four samples of the screens themselves (the pre-pass of ims_screen_prepass keeps an XCD's windows in its
// L2, where the four-fold table of 2 x 2 cells would only quadruple the footprint)
// BATCH: the kernels specialised for imSim's default PSF (run_psf<2>) ask for it; the generic kernels keep the rolled loop and
// their register count
template <bool PLAIN = false, bool BATCH = false>
IMS_DEV void screen_gradient(const ims_atmosphere_t& A, double pu, double pv, double t, double tanx, double tany,
                             double& gx, double& gy)
{
    double sx = 0.0, sy = 0.0;
    const int n = A.npix;
    const double dn = A.dn, inv_n = A.inv_n, inv_scale = A.inv_scale;      // ims_fill_derived_atmosphere
    const bool pow2 = (n & (n - 1)) == 0;      // the screens of the reference are 8192 wide: the wrap is a mask
    const bool quads = !PLAIN && A.screen_quads != nullptr;
    if (BATCH && !PLAIN && quads && pow2 && A.n_layers == SCREEN_BATCH) {
        // the reference's atmosphere (six layers, 8192-wide screens, atmPSF.py:164-205): the layers' gathers do not depend on each
        // other, so all six are requested before the first is used -- ONE trip through the 6.4-GB table per photon instead of
        // six in a row (the rolled loop below waits for every load before it forms the next address).  Same operations on the
        // same values, summed in the same order.
        typedef float fvec4 __attribute__((ext_vector_type(4)));
        fvec4 q[SCREEN_BATCH];
        double ax[SCREEN_BATCH], ay[SCREEN_BATCH];
#pragma unroll
        for (int l = 0; l < SCREEN_BATCH; ++l) {
            const double x = pu - t * A.vx[l] + A.alt[l] * tanx;
            const double y = pv - t * A.vy[l] + A.alt[l] * tany;
            const double fx = (x - A.x0) * inv_scale, fy = (y - A.x0) * inv_scale;
            const double flx = floor(fx), fly = floor(fy);
            ax[l] = fx - flx; ay[l] = fy - fly;
            const int ix = (int)flx & (n - 1), iy = (int)fly & (n - 1);
            q[l] = *(const IMS_G fvec4*)(A.screen_quads + (((int64_t)l * n + iy) * n + ix) * 4);
        }
#pragma unroll
        for (int l = 0; l < SCREEN_BATCH; ++l) {
            const double f00 = (double)q[l].x, f10 = (double)q[l].y, f01 = (double)q[l].z, f11 = (double)q[l].w;
            sx = sx + ((f10 - f00) * (1.0 - ay[l]) + (f11 - f01) * ay[l]);
            sy = sy + ((f01 - f00) * (1.0 - ax[l]) + (f11 - f10) * ax[l]);
        }
        gx = sx * inv_scale; gy = sy * inv_scale;
        return;
    }
    for (int l = 0; l < A.n_layers; ++l) {
        const double x = pu - t * A.vx[l] + A.alt[l] * tanx;
        const double y = pv - t * A.vy[l] + A.alt[l] * tany;
        const double fx = (x - A.x0) * inv_scale, fy = (y - A.x0) * inv_scale;
        const double flx = floor(fx), fly = floor(fy);
        const double ax = fx - flx, ay = fy - fly;
        int ix, iy, ix1, iy1;
        if (pow2) {
            // two's complement: the low bits of the (possibly negative) integer ARE the non-negative remainder
            ix = (int)flx & (n - 1); iy = (int)fly & (n - 1);
            ix1 = (ix + 1) & (n - 1); iy1 = (iy + 1) & (n - 1);
        } else {
            ix = wrap_index(flx, dn, inv_n); iy = wrap_index(fly, dn, inv_n);
            ix1 = ix + 1 == n ? 0 : ix + 1; iy1 = iy + 1 == n ? 0 : iy + 1;
        }
        double f00, f10, f01, f11;
        if (quads) {
            // the 2 x 2 cell of sample (iy, ix) as one 16-byte item (ims_atmosphere_t.screen_quads): one load per layer
            typedef float fvec4 __attribute__((ext_vector_type(4)));
            const fvec4 q = *(const IMS_G fvec4*)(A.screen_quads + (((int64_t)l * n + iy) * n + ix) * 4);
            f00 = (double)q.x; f10 = (double)q.y; f01 = (double)q.z; f11 = (double)q.w;
        } else if (PLAIN) {
            // the two samples of a row are neighbours in memory unless the cell straddles the periodic wrap: one 8-byte load per
            // row (4-byte aligned is all the hardware asks) instead of two 4-byte ones -- half the requests of the pre-pass,
            // whose gathers hit in L2 and are bound by the number of requests
            const float* S = A.screens + (int64_t)l * n * n;
            const uint32_t r0 = (uint32_t)iy * (uint32_t)n, r1 = (uint32_t)iy1 * (uint32_t)n;
            if (ix1 != 0) {
                typedef float fvec2 __attribute__((ext_vector_type(2), aligned(4)));
                const fvec2 a = *(const IMS_G fvec2*)(S + r0 + (uint32_t)ix), b = *(const IMS_G fvec2*)(S + r1 + (uint32_t)ix);
                f00 = (double)a.x; f10 = (double)a.y; f01 = (double)b.x; f11 = (double)b.y;
            } else {
                f00 = (double)S[r0 + (uint32_t)ix]; f10 = (double)S[r0];
                f01 = (double)S[r1 + (uint32_t)ix]; f11 = (double)S[r1];
            }
        } else {
            const float* S = A.screens + (int64_t)l * n * n;
            const uint32_t r0 = (uint32_t)iy * (uint32_t)n, r1 = (uint32_t)iy1 * (uint32_t)n;     // a screen has < 2^31 samples
            f00 = (double)S[r0 + (uint32_t)ix]; f10 = (double)S[r0 + (uint32_t)ix1];
            f01 = (double)S[r1 + (uint32_t)ix]; f11 = (double)S[r1 + (uint32_t)ix1];
        }
        sx = sx + ((f10 - f00) * (1.0 - ay) + (f11 - f01) * ay);
        sy = sy + ((f01 - f00) * (1.0 - ax) + (f11 - f10) * ax);
    }
    gx = sx * inv_scale; gy = sy * inv_scale;
}

// KIND >= 0: the component's kind as a compile-time constant (kernels specialised for a PSF, run_psf<>)
template <int KIND = -1>
IMS_DEV void apply_psf(const ims_render_params_t& P, const ims_object_t& o, int comp, int64_t k, Rng& rng, Photon& ph)
{
    const ims_psf_component_t& c = P.psf[comp];
    const int kind = (KIND >= 0) ? KIND : c.kind;
    rng_block(rng, P.seed, o.obj_id, k, SLOT_PSF + ((uint32_t)comp >> 1));
    const uint32_t wa = (comp & 1) ? rng.w[2] : rng.w[0], wb = (comp & 1) ? rng.w[3] : rng.w[1];
    double scale = c.p0;
    if (c.chrom_alpha != 0.0) scale = scale * dpow(ddiv(ph.wl, c.chrom_base), c.chrom_alpha);
    double ku, kv;
    if (kind == IMS_PSF_GAUSSIAN) {
        double g0, g1;
        gauss_words(wa, wb, g0, g1);
        ku = scale * g0; kv = scale * g1;
    } else if (kind == IMS_PSF_DOUBLE_GAUSSIAN) {
        // sum of two Gaussians: the photon belongs to the first with probability p2
        rng_block(rng, P.seed, o.obj_id, k, SLOT_PSF_TIME + (uint32_t)comp);
        const double sigma = (w01(rng.w[0]) < c.p2) ? scale : c.p1;
        double g0, g1;
        gauss_words(wa, wb, g0, g1);
        ku = sigma * g0; kv = sigma * g1;
    } else if (kind == IMS_PSF_SCREENS) {
        const ims_atmosphere_t& A = *P.atm;
        const double r = dsqrt0(A.aper_ri2 + w01(wa) * A.aper_dr2);
        double s, cc;
        sincos2pi_w(wb, s, cc);
        const double pu = r * cc, pv = r * s;
        rng_block(rng, P.seed, o.obj_id, k, SLOT_PSF_TIME + (uint32_t)comp);
        const double t = A.t0 + w01(rng.w[0]) * A.exptime;
        double gx, gy;
        if (P.screen_kick != nullptr) {              // gathered ahead, in cache-friendly order (ims_screen_prepass): the same sums
            const dvec2q g = *(const IMS_G dvec2q*)(P.screen_kick + 2 * (o.screen_base + k));
            gx = g.x; gy = g.y;
        } else {
            screen_gradient<false, (KIND >= 0)>(A, pu, pv, t, o.atm_tan_x, o.atm_tan_y, gx, gy);
        }
        ku = scale * gx; kv = scale * gy;
        ph.pu = pu; ph.pv = pv; ph.t = t;
    } else {
        const double r2 = radial_r2(P.radial, c.table, w01(wa));
        const double r = dsqrt0(r2) * scale;
        double s, cc;
        sincos2pi_w(wb, s, cc);
        ku = r * cc; kv = r * s;
    }
    ph.x = ph.x + (o.winv[0] * ku + o.winv[1] * kv);
    ph.y = ph.y + (o.winv[2] * ku + o.winv[3] * kv);
}

// The PSF components of a launch.  PSF 0: whatever the descriptor lists; 1: a radial-table profile then a Gaussian (imSim's
// analytic atmosphere: Kolmogorov (+) Gaussian) as straight-line code.  The host checks the descriptor.
template <int PSF>
IMS_DEV void run_psf(const ims_render_params_t& P, const ims_object_t& o, int64_t k, Rng& rng, Photon& ph)
{
    if (PSF == 1) {
        apply_psf<IMS_PSF_RADIAL>(P, o, 0, k, rng, ph);
        apply_psf<IMS_PSF_GAUSSIAN>(P, o, 1, k, rng, ph);
    } else if (PSF == 2) {
        // imSim's default AtmosphericPSF: phase screens, second kick (a radial table), Gaussian
        apply_psf<IMS_PSF_SCREENS>(P, o, 0, k, rng, ph);
        apply_psf<IMS_PSF_RADIAL>(P, o, 1, k, rng, ph);
        apply_psf<IMS_PSF_GAUSSIAN>(P, o, 2, k, rng, ph);
    } else {
        for (int c = 0; c < P.n_psf; ++c) apply_psf(P, o, c, k, rng, ph);
    }
}

// ---------------- media / air ----------------
// Air index (Filippenko 1982, as GalSim's dcr module and batoid.Air use it), spec v4: the
// pressure/temperature and water-vapour factors are uniform and come precomputed from the host
// (ims_air_factors), and the dispersion formula is put over ONE denominator:
//   n-1 = air_p (64.328 + 29498.1/(146 - s2) + 255.4/(41 - s2)) - air_w (0.0624 - 0.00068 s2),  s2 = 1/w2
IMS_DEV double air_n_minus_one(double wave_nm, double air_p, double air_w)
{
    const double wm = wave_nm * 1.0e-3;
    const double w2 = wm * wm;
    const double d1 = fma(146.0, w2, -1.0), d2 = fma(41.0, w2, -1.0);
    const double den = d1 * d2;
    const double num = fma(29498.1, d2, 255.4 * d1);
    const double disp = fma(64.328, den, w2 * num);             // dispersion * den
    const double wat = fma(0.0624, w2, -0.000680) * den;         // water term * den * w2
    return ddiv(air_p * (disp * w2) - air_w * wat, den * w2);
}
IMS_DEV double refraction_r0(double nm1) { return ddiv(nm1 * (nm1 + 2.0) / 2.0, nm1 * nm1 + 2.0 * nm1 + 1.0); }

IMS_DEV double medium_n(int kind, const double* c, double wave_nm)
{
    if (kind == IMS_MEDIUM_CONST) return c[0];
    if (kind == IMS_MEDIUM_SELLMEIER) {
        const double l = wave_nm * 1.0e-3;
        const double l2 = l * l;
        const double n2 = 1.0 + ddiv(c[0] * l2, l2 - c[3]) + ddiv(c[1] * l2, l2 - c[4]) + ddiv(c[2] * l2, l2 - c[5]);
        return dsqrt_n(n2);
    }
    return 1.0 + air_n_minus_one(wave_nm, c[3], c[4]);
}

// ---------------- TAN-SIP, trig-free ----------------
// SIP polynomials over the triangle p + q <= 4 (coefficients beyond `order` are zero in the table),
// nested Horner: inner in v for every power of u, outer in u.  14 fma per polynomial.
IMS_DEV double sip_row(const double* a, int p, double v)
{
    double r = a[p * 5 + (4 - p)];
#pragma unroll
    for (int q = 3 - p; q >= 0; --q) r = fma(r, v, a[p * 5 + q]);
    return r;
}
IMS_DEV double sip_value(const double* a, double u, double v)
{
    double f = a[4 * 5 + 0];
#pragma unroll
    for (int p = 3; p >= 0; --p) f = fma(f, u, sip_row(a, p, v));
    return f;
}
// value and both partial derivatives by the simultaneous Horner recurrences (no coefficient scaling)
IMS_DEV void sip_value_grad(const double* a, double u, double v, double& f, double& fu, double& fv)
{
    double F = a[4 * 5 + 0], Fu = 0.0, Fv = 0.0;
#pragma unroll
    for (int p = 3; p >= 0; --p) {
        double r = a[p * 5 + (4 - p)], rv = 0.0;
#pragma unroll
        for (int q = 3 - p; q >= 0; --q) { rv = fma(rv, v, r); r = fma(r, v, a[p * 5 + q]); }
        Fu = fma(Fu, u, F);
        F = fma(F, u, r);
        Fv = fma(Fv, u, rv);
    }
    f = F; fu = Fu; fv = Fv;
}

IMS_DEV void wcs_pix_to_vec(const ims_tansip_t& w, double x, double y, double (&p)[3])
{
    double u = x - w.crpix[0], v = y - w.crpix[1];
    if (w.order > 0) {
        const double f = sip_value(w.a, u, v), g = sip_value(w.b, u, v);
        u = u + f; v = v + g;
    }
    const double xi = w.cd[0] * u + w.cd[1] * v;
    const double eta = w.cd[2] * u + w.cd[3] * v;
    // not normalised: every consumer projects the direction onto a tangent plane, where the norm cancels
    p[0] = w.rot[0] + w.rot[3] * xi + w.rot[6] * eta;
    p[1] = w.rot[1] + w.rot[4] * xi + w.rot[7] * eta;
    p[2] = w.rot[2] + w.rot[5] * xi + w.rot[8] * eta;
}

// direction -> pixel: Newton inversion of the SIP polynomial from the undistorted position, until
// the step falls below 1e-10 of the position (the NEXT error is then its square) or 6 steps
IMS_DEV void wcs_vec_to_pix(const ims_tansip_t& w, const double (&p)[3], double& x, double& y)
{
    const double t0 = w.rot[0] * p[0] + w.rot[1] * p[1] + w.rot[2] * p[2];
    const double t1 = w.rot[3] * p[0] + w.rot[4] * p[1] + w.rot[5] * p[2];
    const double t2 = w.rot[6] * p[0] + w.rot[7] * p[1] + w.rot[8] * p[2];
    const double it0 = ddiv(1.0, t0);
    const double xi = t1 * it0, eta = t2 * it0;
    const double U = w.cdinv[0] * xi + w.cdinv[1] * eta;
    const double V = w.cdinv[2] * xi + w.cdinv[3] * eta;
    double u = U, v = V;
    if (w.order > 0) {
        for (int it = 0; it < 6; ++it) {
            double f, g, fu, fv, gu, gv;
            sip_value_grad(w.a, u, v, f, fu, fv);
            sip_value_grad(w.b, u, v, g, gu, gv);
            const double r0 = u + f - U, r1 = v + g - V;
            const double j00 = 1.0 + fu, j11 = 1.0 + gv;
            const double idet = ddiv(1.0, j00 * j11 - fv * gu);
            const double du = (j11 * r0 - fv * r1) * idet;
            const double dv = (j00 * r1 - gu * r0) * idet;
            u = u - du; v = v - dv;
            if (fabs(du) + fabs(dv) <= 1.0e-10 * (fabs(u) + fabs(v))) break;
        }
    }
    x = u + w.crpix[0]; y = v + w.crpix[1];
}

IMS_DEV void xy_to_v(const ims_optics_t& o, double x, double y, double wave_nm, double (&v)[3])
{
    // spec v6: the direction (tan theta_x, tan theta_y, -1) as it is -- nothing downstream needs |v| = 1/n (the kick of
    // diffract() is proportional to v_z, reflection and refraction are homogeneous in v, the slopes are ratios)
    double p[3], thx, thy;
    wcs_pix_to_vec(o.img_wcs, x, y, p);
    wcs_vec_to_pix(o.icrf_to_field, p, thx, thy);
    v[0] = thx; v[1] = thy; v[2] = -1.0;
}
IMS_DEV void v_to_xy(const ims_optics_t& o, const double (&v)[3], double& x, double& y)
{
    const double ivz = ddiv(1.0, v[2]);
    const double thx = -v[0] * ivz, thy = -v[1] * ivz;
    double p[3];
    wcs_pix_to_vec(o.icrf_to_field, thx, thy, p);
    wcs_vec_to_pix(o.img_wcs, p, x, y);
}

// ---------------- spider diffraction ----------------
IMS_DEV void field_rotation(const ims_optics_t& o, double t, double& c, double& s)
{
    double sn, cs;
    dsincos(o.omega * t, sn, cs);
    const double ez0 = o.cos_lat * cs, ez1 = o.cos_lat * sn, ez2 = o.sin_lat;
    const double* ef = o.e_focal;
    const double eh0_ = ef[1] * ez2 - ef[2] * ez1, eh1_ = ef[2] * ez0 - ef[0] * ez2, eh2_ = ef[0] * ez1 - ef[1] * ez0;
    const double g0 = o.rot_g[0], g1 = o.rot_g[1], g2 = o.rot_g[2];        // e_focal x e_z0 (ims_fill_derived_optics)
    const double nrm = dsqrt_n(eh0_ * eh0_ + eh1_ * eh1_ + eh2_ * eh2_) * o.rot_gnorm;
    c = ddiv(eh0_ * g0 + eh1_ * g1 + eh2_ * g2, nrm);
    s = ddiv(ez0 * g0 + ez1 * g1 + ez2 * g2, nrm);
}

IMS_DEV void directed_dist(const ims_optics_t& o, double px, double py, double& dist, double& nx, double& ny)
{
    double dl = 0.0; int il = -1;
    for (int l = 0; l < o.n_lines; ++l) {
        const double* L = o.lines[l];
        const double d = fabs(fabs(L[0] * px + L[1] * py - L[2]) - L[3]);
        if (il < 0 || d < dl) { dl = d; il = l; }
    }
    double dc = 0.0; int ic = -1;
    for (int c = 0; c < o.n_circles; ++c) {
        const double* C = o.circles[c];
        const double ex = px - C[0], ey = py - C[1];
        const double d = fabs(dsqrt0(ex * ex + ey * ey) - C[2]);
        if (ic < 0 || d < dc) { dc = d; ic = c; }
    }
    if (il >= 0 && (ic < 0 || dl < dc)) {
        dist = dl; nx = o.lines[il][0]; ny = o.lines[il][1];
    } else {
        const double ex = o.circles[ic][0] - px, ey = o.circles[ic][1] - py;
        const double nr = dsqrt_n(ex * ex + ey * ey);
        dist = dc; nx = ddiv(ex, nr); ny = ddiv(ey, nr);
    }
}

IMS_DEV void diffract(const ims_optics_t& o, bool field_rot, double pu, double pv, double t,
                      double wavelength_m, double gauss, double (&v)[3])
{
    double c = 1.0, s = 0.0;
    if (field_rot) field_rotation(o, t, c, s);
    const double qx = c * pu - s * pv;
    const double qy = s * pu + c * pv;
    double d, nx, ny;
    directed_dist(o, qx, qy, d, nx, ny);
    const double k = ddiv(TWO_PI, wavelength_m);
    const double dtp = gauss * datan(1.0 / (2.0 * k * d));
    const double vz = -v[2];
    const double sx = dtp * vz * nx, sy = dtp * vz * ny;
    const double rx = c * sx + s * sy;
    const double ry = -s * sx + c * sy;
    // spec v6: the reference rescales v to its old length here (imsim/diffraction.py:45-66); every consumer of v is
    // homogeneous in it, so the direction alone is kept
    v[0] = v[0] + rx; v[1] = v[1] + ry;
}

// ---------------- sequential ray trace ----------------
IMS_DEV bool obscured(const ims_surface_t& S, double r2)
{
    if (S.obsc_kind == IMS_OBSC_NONE) return false;
    const double i2 = S.obsc_i2, o2 = S.obsc_o2;
    switch (S.obsc_kind) {
    case IMS_OBSC_CLEAR_ANNULUS: return !(r2 >= i2 && r2 <= o2);
    case IMS_OBSC_CLEAR_CIRCLE:  return !(r2 <= o2);
    case IMS_OBSC_OBSC_CIRCLE:   return r2 < o2;
    case IMS_OBSC_OBSC_ANNULUS:  return (r2 >= i2 && r2 < o2);
    }
    return false;
}

// Propagate to surface S (spec v5): plane exact; conic by the closed-form root of smaller
// |t|, with the un-normalised normal (-c x, -c y, 1-(1+k) c z) that needs no sqrt; even-asphere terms
// by Newton on the implicit conic form from the conic root until |G| <= 1e-8 (spec v6: 10 nm of sag, < 5e-4 pixel at the
// detector; one step from the conic root for the Rubin mirrors, whose residual is then <= 8e-9 on M2 and <= 2e-10 elsewhere).
// SHAPE: the surface's form as a compile-time constant (trace_seq: kernels specialised for an optics layout) -- 0 plane,
// 1 conic with R != 0 and no asphere terms, 2 conic with R != 0 and asphere terms; -1 = read it from the descriptor.
constexpr int SH_PLANE = 0, SH_CONIC = 1, SH_ASPHERE = 2;
template <int SHAPE = -1>
IMS_DEV bool surf_hit(const ims_surface_t& S, double (&pos)[3], const double (&vel)[3], double (&N)[3], double& nn, double& r2_out)
{
    const bool is_plane = (SHAPE >= 0) ? (SHAPE == SH_PLANE) : (S.R == 0.0 && S.n_asphere == 0);
    const bool is_curved = (SHAPE >= 0) ? (SHAPE != SH_PLANE) : (S.R != 0.0);
    const bool is_asphere = (SHAPE >= 0) ? (SHAPE == SH_ASPHERE) : (S.n_asphere > 0);
    const double pz = pos[2] - S.z0;
    double t;
    if (is_plane) {
        t = ddiv(-pz, vel[2]);
        pos[0] = fma(vel[0], t, pos[0]); pos[1] = fma(vel[1], t, pos[1]); pos[2] = S.z0;
        N[0] = 0.0; N[1] = 0.0; N[2] = 1.0; nn = 1.0;
        r2_out = fma(pos[0], pos[0], pos[1] * pos[1]);
        return true;
    }
    const double c = S.inv_R, k1 = S.k1;       // k1, k1c, m2R, cc, asph_d: ims_fill_derived_optics
    if (is_curved) {
        // A t^2 + 2 hb t + C = 0 (spec v5: the dot products are fma chains, hb is half the linear coefficient)
        const double k1vz = k1 * vel[2];
        const double A = fma(vel[0], vel[0], fma(vel[1], vel[1], k1vz * vel[2]));
        const double hb = fma(pos[0], vel[0], fma(pos[1], vel[1], fma(k1vz, pz, -(S.R * vel[2]))));
        const double C = fma(pos[0], pos[0], fma(pos[1], pos[1], pz * fma(k1, pz, S.m2R)));
        const double dq = fma(hb, hb, -(A * C));
        if (dq < 0.0) return false;
        const double sq = dsqrt0(dq);
        const double q = -(hb + (hb < 0.0 ? -sq : sq));
        t = ddiv(C, q);                              // the root of smaller |t| (q carries the larger magnitude)
    } else {
        t = ddiv(-pz, vel[2]);
    }
    // even asphere z = conic(r2) + p(r2), p = sum a_k r^(2k+4): Newton on the implicit conic form
    // G = c (r2 + k1 w^2) - 2 w with w = z - p(r2), which needs neither sqrt nor a second division.  A surface without
    // asphere terms is the same code with no iteration: w = z, dp = 0, and the tail below gives g = fma(2 m, 0, c) = c,
    // nn = fma(c c, r2, m m) -- the bits of the closed conic form (cc = c c is the same IEEE product), one path to merge.
    double x = fma(vel[0], t, pos[0]), y = fma(vel[1], t, pos[1]), z = fma(vel[2], t, pz);
    double r2 = fma(x, x, y * y), w = z, dp = 0.0;
    if (is_asphere) {
        for (int it = 0; it < 6; ++it) {
            if (it > 0) {
                x = fma(vel[0], t, pos[0]); y = fma(vel[1], t, pos[1]); z = fma(vel[2], t, pz);
                r2 = fma(x, x, y * y);
            }
            double p = 0.0, rp = r2;
            dp = 0.0;
            for (int k = 0; k < S.n_asphere; ++k) {
                dp = fma(S.asph_d[k], rp, dp);
                rp = rp * r2;
                p = fma(S.asph[k], rp, p);
            }
            w = z - p;
            const double k1w = k1 * w;
            const double G = fma(c, fma(k1w, w, r2), -2.0 * w);
            if (fabs(G) <= 1.0e-8 || it == 5) break;
            const double s = fma(x, vel[0], y * vel[1]);
            const double wp = fma(-2.0 * dp, s, vel[2]);
            const double Gp = 2.0 * fma(c, fma(k1w, wp, s), -wp);
            t = t - ddiv(G, Gp);
        }
    }
    const double m = fma(-S.k1c, w, 1.0);
    if (!(m > 0.0)) return false;
    const double g = fma(2.0 * m, dp, c);
    pos[0] = x; pos[1] = y; pos[2] = S.z0 + z;
    N[0] = -g * x; N[1] = -g * y; N[2] = m;
    r2_out = r2;
    nn = fma(g * g, r2, m * m);
    return true;
}

// the state a ray carries from surface to surface besides position and direction
struct TraceState {
    int vignetted;
    double n_cur;
    // the index of a medium is a pure function of (medium, wavelength): computed once per photon and
    // reused when the same medium recurs (all Rubin lenses and filters are fused silica)
    int glass_id;
    double glass_n, glass_in;
};

// One surface of the sequential trace: intersect, vignette, reflect or refract.  KIND / SHAPE >= 0: compile-time constants
// (trace_seq), -1: read from the descriptor.  Returns false when the ray is lost.
template <int KIND = -1, int SHAPE = -1>
IMS_DEV bool trace_step(const ims_surface_t& S, TraceState& st, double (&pos)[3], double (&vel)[3], double wave_nm)
{
    const int kind = (KIND >= 0) ? KIND : S.kind;
    double N[3], nn, r2;
    if (!surf_hit<SHAPE>(S, pos, vel, N, nn, r2)) return false;
    if (obscured(S, r2)) st.vignetted = 1;
    if (kind == IMS_SURF_BAFFLE || kind == IMS_SURF_DETECTOR) return true;
    if (kind == IMS_SURF_MIRROR) {
        // spec v6: (N.N) v - 2 (v.N) N, the reflected direction times |N|^2 -- no division; the length of the velocity
        // carries no information (every later use is homogeneous in it)
        const double vn = fma(vel[0], N[0], fma(vel[1], N[1], vel[2] * N[2]));
        const double d = 2.0 * vn;
        vel[0] = fma(nn, vel[0], -(d * N[0])); vel[1] = fma(nn, vel[1], -(d * N[1])); vel[2] = fma(nn, vel[2], -(d * N[2]));
    } else {
        double n2, in2;
        if (S.medium_kind == IMS_MEDIUM_CONST) { n2 = S.medium_c[0]; in2 = S.medium_c[1]; }
        else if (S.medium_id == st.glass_id) { n2 = st.glass_n; in2 = st.glass_in; }
        else {
            n2 = medium_n(S.medium_kind, S.medium_c, wave_nm);
            in2 = ddiv(1.0, n2);
            st.glass_id = S.medium_id; st.glass_n = n2; st.glass_in = in2;
        }
        // Snell with un-normalised velocity and normal (spec v6).  With eta = n1 / n2, c = v.N / (|v| |N|) the refracted unit
        // vector is eta v^ - (eta c - sign(c) sqrt(1 - eta^2 (1 - c^2))) N^; times |v| |N|^2 that is
        //   eta (N.N) v - (eta (v.N) - sign(v.N) sqrt(D)) N,   D = (v.v)(N.N) + eta^2 ((v.N)^2 - (v.v)(N.N)),
        // one square root and no division; D < 0 is total internal reflection
        const double vn = fma(vel[0], N[0], fma(vel[1], N[1], vel[2] * N[2]));
        const double v2 = fma(vel[0], vel[0], fma(vel[1], vel[1], vel[2] * vel[2]));
        const double eta = st.n_cur * in2;
        const double e2 = eta * eta;
        const double v2nn = v2 * nn;
        const double D = fma(e2, fma(vn, vn, -v2nn), v2nn);
        if (D < 0.0) return false;
        const double sq = dsqrt0(D);
        const double q = fma(eta, vn, vn < 0.0 ? sq : -sq);
        const double en = eta * nn;
        vel[0] = fma(en, vel[0], -(q * N[0]));
        vel[1] = fma(en, vel[1], -(q * N[1]));
        vel[2] = fma(en, vel[2], -(q * N[2]));
        st.n_cur = n2;
    }
    return true;
}

// An optics LAYOUT is the sequence of (kind, shape) of the surfaces as 4-bit codes, first surface in the lowest nibble:
// code = 1 + 3 kind_class + shape with kind_class 0 mirror, 1 refracting, 2 detector / baffle; a zero nibble ends the list
// (at most 15 surfaces).  engine.optics_layout() computes it from the descriptor (ims_render_params_t.optics_layout), and the
// library holds kernels whose trace is unrolled for the layouts it knows (IMS_LAYOUT_RUBIN_LIKE); any other layout, and 0,
// run the surface loop.
template <unsigned long long CODE, int K>
struct TraceSeq {
    static IMS_DEV bool run(const ims_optics_t& o, TraceState& st, double (&pos)[3], double (&vel)[3], double wave_nm)
    {
        constexpr int code = (int)((CODE >> (4 * K)) & 15ull);
        if constexpr (code == 0) {
            return true;
        } else {
            constexpr int kc = (code - 1) / 3, shape = (code - 1) % 3;
            constexpr int kind = (kc == 0) ? IMS_SURF_MIRROR : (kc == 1) ? IMS_SURF_REFRACT : IMS_SURF_DETECTOR;
            if (!trace_step<kind, shape>(o.surf[K], st, pos, vel, wave_nm)) return false;
            return TraceSeq<CODE, K + 1>::run(o, st, pos, vel, wave_nm);
        }
    }
};
template <unsigned long long CODE>
struct TraceSeq<CODE, 16> {
    static IMS_DEV bool run(const ims_optics_t&, TraceState&, double (&)[3], double (&)[3], double) { return true; }
};

// the approximate Rubin prescription of optics.rubin_like_telescope: three aspheric mirrors, L1 (two conics), L2 (plane,
// asphere), filter (two conics), L3 (two conics), detector plane
constexpr unsigned long long IMS_LAYOUT_RUBIN_LIKE =
    (3ull << 0) | (3ull << 4) | (3ull << 8) | (5ull << 12) | (5ull << 16) | (4ull << 20) | (6ull << 24) | (5ull << 28) |
    (5ull << 32) | (5ull << 36) | (5ull << 40) | (7ull << 44);

// returns 0 ok, 1 vignetted, 2 failed.  LAYOUT != 0: the surfaces unrolled for that layout (the host has checked that the
// descriptor has it), 0: the loop over whatever the descriptor lists.
template <unsigned long long LAYOUT = 0ull>
IMS_DEV int trace(const ims_optics_t& o, double (&pos)[3], double (&vel)[3], double wave_nm)
{
    TraceState st;
    st.vignetted = 0;
    if (o.in_medium_kind == IMS_MEDIUM_CONST) st.n_cur = o.in_medium_c[0];
    else st.n_cur = medium_n(o.in_medium_kind, o.in_medium_c, wave_nm);
    st.glass_id = -1; st.glass_n = 0.0; st.glass_in = 0.0;
    if (LAYOUT != 0ull) {
        if (!TraceSeq<LAYOUT, 0>::run(o, st, pos, vel, wave_nm)) return 2;
    } else {
        for (int k = 0; k < o.n_surfaces; ++k)
            if (!trace_step(o.surf[k], st, pos, vel, wave_nm)) return 2;
    }
    return st.vignetted;
}

template <unsigned long long LAYOUT = 0ull>
IMS_DEV void rubin_op(const ims_render_params_t& P, const ims_op_t& op, int kind, int op_index,
                      const ims_object_t& o, int64_t k, Rng& rng, Photon& ph)
{
    const ims_optics_t& opt = *P.optics;
    const bool do_diff = (kind != IMS_OP_RUBIN_OPTICS);
    const bool do_trace = (kind != IMS_OP_RUBIN_DIFFRACTION);
    const bool frot = !(op.p[1] != 0.0);
    double v[3];
    xy_to_v(opt, ph.x, ph.y, ph.wl, v);
    if (do_diff) {
        double g0, g1;
        rng_block(rng, P.seed, o.obj_id, k, SLOT_OP + ((uint32_t)op_index >> 1));
        g0 = gauss_word_cos((op_index & 1) ? rng.w[2] : rng.w[0], (op_index & 1) ? rng.w[3] : rng.w[1]); g1 = 0.0;
        diffract(opt, frot, ph.pu, ph.pv, ph.t, ph.wl * 1.0e-9, g0, v);
    }
    if (!do_trace) { v_to_xy(opt, v, ph.x, ph.y); return; }
    double pos[3] = { ph.pu, ph.pv, opt.stop_z };
    const int st = trace<LAYOUT>(opt, pos, v, ph.wl);
    if (st == 2) { ph.x = 0.0; ph.y = 0.0; ph.dxdz = 0.0; ph.dydz = 0.0; ph.flux = 0.0; return; }
    const double c = opt.cam_rot[0], s = opt.cam_rot[1];
    const double rx = c * pos[0] + s * pos[1], ry = -s * pos[0] + c * pos[1];
    const double rvx = c * v[0] + s * v[1], rvy = -s * v[0] + c * v[1];
    const double fpx = ry * 1.0e3, fpy = rx * 1.0e3;
    ph.x = opt.fp_to_pix[0] * fpx + opt.fp_to_pix[1] * fpy + opt.fp_to_pix[2];
    ph.y = opt.fp_to_pix[3] * fpx + opt.fp_to_pix[4] * fpy + opt.fp_to_pix[5];
    const double ivz = ddiv(1.0, v[2]);
    ph.dxdz = (opt.slope_jac[0] * rvx + opt.slope_jac[1] * rvy) * ivz;
    ph.dydz = (opt.slope_jac[2] * rvx + opt.slope_jac[3] * rvy) * ivz;
    if (st == 1) ph.flux = 0.0;
}

// one configured photon operator (config/imsim-config.yaml:281-320).  KIND >= 0: the operator's kind as a compile-time
// constant (the kernels specialised for the default chain, run_ops<1>): the switch folds away.
template <int KIND = -1, unsigned long long LAYOUT = 0ull>
IMS_DEV void apply_op(const ims_render_params_t& P, int op_index, const ims_object_t& o, int64_t k, Rng& rng, Photon& ph)
{
    const ims_op_t& op = P.ops[op_index];
    const int kind = (KIND >= 0) ? KIND : op.kind;
    const uint32_t slot = SLOT_OP + ((uint32_t)op_index >> 1);
    const int wsel = op_index & 1;
    if (kind == IMS_OP_BANDPASS_RATIO) {
        ph.flux = ph.flux * lin_lookup(P.ratio, op.table, ph.wl);
        return;
    }
    if (o.flags & IMS_OBJ_FAINT) return;
    switch (kind) {
    case IMS_OP_TIME_SAMPLER: {
        rng_block(rng, P.seed, o.obj_id, k, slot);
        ph.t = op.p[0] + w01(wsel ? rng.w[2] : rng.w[0]) * op.p[1];
        break; }
    case IMS_OP_PUPIL_ANNULUS_SAMPLER: {
        rng_block(rng, P.seed, o.obj_id, k, slot);
        const double r = dsqrt0(op.p[2] + w01(wsel ? rng.w[2] : rng.w[0]) * op.p[3]);      // p2 = R_inner^2, p3 = R_outer^2 - R_inner^2
        double s, c;
        sincos2pi_w(wsel ? rng.w[3] : rng.w[1], s, c);
        ph.pu = r * c; ph.pv = r * s;
        break; }
    case IMS_OP_PHOTON_DCR: {
        // p[5], p[6]: air factors; p[7]: refraction constant at the base wavelength (ims_fill_derived_op)
        const double r0 = refraction_r0(air_n_minus_one(ph.wl, op.p[5], op.p[6]));
        const double shift = (r0 - op.p[7]) * o.dcr_tanz * op.p[4];
        const double du = -shift * o.dcr_sinp;
        const double dv = shift * o.dcr_cosp;
        ph.x = ph.x + (o.winv[0] * du + o.winv[1] * dv);
        ph.y = ph.y + (o.winv[2] * du + o.winv[3] * dv);
        break; }
    case IMS_OP_FOCUS_DEPTH:
        ph.x = ph.x + ph.dxdz * op.p[0];
        ph.y = ph.y + ph.dydz * op.p[0];
        break;
    case IMS_OP_REFRACTION: {
        const double a = ph.dxdz, b = ph.dydz;
        const double rho2 = a * a + b * b;
        const double f = ddiv(1.0, dsqrt_n(op.p[1] + op.p[2] * rho2));                    // p1 = n^2, p2 = n^2 - 1
        ph.dxdz = a * f; ph.dydz = b * f;
        break; }
    case IMS_OP_RUBIN_OPTICS:
    case IMS_OP_RUBIN_DIFFRACTION:
    case IMS_OP_RUBIN_DIFFRACTION_OPTICS:
        rubin_op<LAYOUT>(P, op, kind, op_index, o, k, rng, ph);
        break;
    default: break;
    }
}

// -DIMS_PROBE (measurement builds only, tools/dbg/round_probe.py): thread 0 of the middle workgroup of a kernel (PROBE) or of
// each of the first 64 workgroups (PROBE_WG) stamps the 100 MHz real-time counter at marked places of the round kernels.
#ifdef IMS_PROBE
__device__ unsigned long long g_probe[32];
__device__ unsigned long long g_probe_wg[64][8];
#define PROBE(k) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) ims::g_probe[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define PROBE_WG(k, tid) do { if (blockIdx.x < 64 && threadIdx.x == (tid)) ims::g_probe_wg[blockIdx.x][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PROBE(k) do { } while (0)
#define PROBE_WG(k, tid) do { } while (0)
#endif

// ---------------- Silicon sensor ----------------
// the slot of an object's pixel-boundary region in this launch (ims_render_params_t.bf_slot_shift)
IMS_DEV int slot_index(const ims_render_params_t& P, const ims_object_t& o)
{
    return o.bf_state > 0 ? o.bf_state + (int)P.bf_slot_shift : o.bf_state;
}

struct SlotView {
    int xmin, ymin, nx, ny;
    int64_t offset;
};
IMS_DEV int64_t cell_index(const SlotView& sl, int i, int j) { return sl.offset + (int64_t)j * (sl.nx + 1) + i; }

// vertex k (0..nv-1) of pixel (i,j), counter-clockwise from the lower-left corner.  A cell owns
// its bottom row (LL corner, bottom points, LR corner) and its left-edge points.
IMS_DEV void polygon_vertex(const ims_sensor_t& s, const SlotView& sl, int i, int j, int k, double zfactor,
                            double& vx, double& vy)
{
    const int nV = s.num_vertices, npo = 2 * nV + 2;
    int ci = i, cj = j, q;
    double ax = 0.0, ay = 0.0;
    if (k <= nV + 1) { q = k; }                                                     // own bottom row
    else if (k <= 2 * nV + 1) { ci = i + 1; ax = 1.0; q = nV + 2 + (k - nV - 2); }  // right edge = left edge of the cell to the right
    else if (k <= 3 * nV + 3) { cj = j + 1; ay = 1.0; q = nV + 1 - (k - 2 * nV - 2); }  // top row = bottom row of the cell above, reversed
    else { q = nV + 2 + (nV - 1 - (k - 3 * nV - 4)); }                              // own left edge, top->bottom
    const double* pt = s.bf_boundary + (cell_index(sl, ci, cj) * npo + q) * 2;
    vx = pt[0] + ax; vy = pt[1] + ay;
    if (zfactor != 1.0) {
        const double ex = s.emptypoly[2 * k], ey = s.emptypoly[2 * k + 1];
        vx = ex + (vx - ex) * zfactor;
        vy = ey + (vy - ey) * zfactor;
    }
}

typedef double dvec2 __attribute__((ext_vector_type(2)));
typedef double dvec4 __attribute__((ext_vector_type(4)));

// Crossing-number test of (x, y) against the polygon of the pixel whose owner cell is `own` (right neighbour cell `rgt`,
// upper neighbour cell `upp`), every vertex pulled towards the undistorted polygon by zfactor.  NV > 0: the sensor model's
// vertices per edge as a compile-time constant (4 or 8): the polygon unrolls into straight-line code with static vertex
// addresses, every batch of four vertex loads is in flight together and neither the crossing tests nor the zfactor scaling
// carry a branch.  (With NV read from the descriptor the compiler emitted one load, one wait and a branch ladder per vertex:
// 20 dependent L2 latencies per polygon -- the bulk of a brighter-fatter round.)  NV = 0 is the generic loop.  Same
// arithmetic, same decisions.
template <int NV>
IMS_DEV bool polygon_test(const ims_sensor_t& s, const IMS_G double* own, const IMS_G double* rgt, const IMS_G double* upp,
                          double x, double y, double zfactor)
{
    const int nV = (NV > 0) ? NV : s.num_vertices, nv = 4 * nV + 4;
    const bool scaled = (zfactor != 1.0);
    bool inside = false;
    if (NV > 0) {
        // The vertex loads of the polygon are stated before the first use (in halves for 8 vertices per edge), so that the
        // compiler keeps as many in flight as the register budget allows: every dependent memory round trip of a round's
        // kernels costs ~1.4 us (the lines were written by the previous kernel on other XCDs), tools/dbg/round_probe.py.
        constexpr int NVX = (NV > 0) ? NV : 1, NVT = 4 * NVX + 4, CH = (NVT <= 20) ? NVT : NVT / 2;
        double lx, ly;
        {
            const double2 p = *(const IMS_G double2*)(own + 2 * (NVX + 2));               // closing vertex NVT - 1: own left edge, first point
            const double ex = s.emptypoly[2 * (NVT - 1)], ey = s.emptypoly[2 * (NVT - 1) + 1];
            const double sx = ex + (p.x - ex) * zfactor, sy = ey + (p.y - ey) * zfactor;
            lx = scaled ? sx : p.x; ly = scaled ? sy : p.y;
        }
#pragma unroll
        for (int k0 = 0; k0 < NVT; k0 += CH) {
            dvec2 v[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int k = k0 + u;
                const IMS_G double* bb = own; int q = k;
                if (k > NVX + 1) {
                    if (k <= 2 * NVX + 1) { bb = rgt; q = k; }
                    else if (k <= 3 * NVX + 3) { bb = upp; q = 3 * NVX + 3 - k; }
                    else { q = 5 * NVX + 5 - k; }
                }
                v[u] = *(const IMS_G dvec2*)(bb + 2 * q);
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int k = k0 + u;
                const double ax = (k > NVX + 1 && k <= 2 * NVX + 1) ? 1.0 : 0.0;
                const double ay = (k > 2 * NVX + 1 && k <= 3 * NVX + 3) ? 1.0 : 0.0;
                double kx = v[u].x + ax, ky = v[u].y + ay;
                const double ex = s.emptypoly[2 * k], ey = s.emptypoly[2 * k + 1];
                const double sx = ex + (kx - ex) * zfactor, sy = ey + (ky - ey) * zfactor;
                kx = scaled ? sx : kx; ky = scaled ? sy : ky;
                // x < (lx - kx) (y - ky) / (ly - ky) + kx, cross-multiplied (ly != ky when the edge crosses): no division
                const double dy = ly - ky;
                const double lhs = (x - kx) * dy, rhs = (lx - kx) * (y - ky);
                const bool crosses = (ky > y) != (ly > y);
                const bool left = (dy > 0.0) ? (lhs < rhs) : (lhs > rhs);
                inside = inside != (crosses && left);
                lx = kx; ly = ky;
            }
        }
    } else {
        // the vertex addresses depend only on the loop counter: unrolled four-fold to keep four independent loads in flight
        double lx, ly;
        {
            const int k = nv - 1;                                                         // closing vertex: own left edge, first point
            const double2 p = *(const IMS_G double2*)(own + 2 * (5 * nV + 5 - k));
            lx = p.x; ly = p.y;
            if (scaled) {
                const double ex = s.emptypoly[2 * k], ey = s.emptypoly[2 * k + 1];
                lx = ex + (lx - ex) * zfactor;
                ly = ey + (ly - ey) * zfactor;
            }
        }
#pragma unroll 4
        for (int k = 0; k < nv; ++k) {
            const IMS_G double* bb = own; int q = k; double ax = 0.0, ay = 0.0;
            if (k > nV + 1) {
                if (k <= 2 * nV + 1) { bb = rgt; ax = 1.0; q = k; }                       // nV + 2 + (k - nV - 2)
                else if (k <= 3 * nV + 3) { bb = upp; ay = 1.0; q = 3 * nV + 3 - k; }     // nV + 1 - (k - 2 nV - 2)
                else { q = 5 * nV + 5 - k; }                                             // nV + 2 + (nV - 1 - (k - 3 nV - 4))
            }
            const double2 p = *(const IMS_G double2*)(bb + 2 * q);
            double kx = p.x + ax, ky = p.y + ay;
            if (scaled) {
                const double ex = s.emptypoly[2 * k], ey = s.emptypoly[2 * k + 1];
                kx = ex + (kx - ex) * zfactor;
                ky = ey + (ky - ey) * zfactor;
            }
            if ((ky > y) != (ly > y)) {
                // x < (lx - kx) (y - ky) / (ly - ky) + kx, cross-multiplied (ly != ky here): no division
                const double dy = ly - ky;
                const double lhs = (x - kx) * dy, rhs = (lx - kx) * (y - ky);
                if ((dy > 0.0) ? (lhs < rhs) : (lhs > rhs)) inside = !inside;
            }
            lx = kx; ly = ky;
        }
    }
    return inside;
}

// GalSim's Silicon::insidePixel.  ZF: `z` is already the polygon shrink factor tanh(zconv / 12) (converted pools), otherwise
// the conversion depth zconv it is computed from when the polygon is needed.
template <int NV = 0, bool ZF = false>
IMS_DEV bool inside_pixel(const ims_sensor_t& s, const SlotView& sl, int ix, int iy, double x, double y,
                          double z, bool want_edge, bool& off_edge)
{
    const int i = ix - sl.xmin, j = iy - sl.ymin;
    if (i < 0 || i >= sl.nx || j < 0 || j >= sl.ny) {
        if (want_edge) off_edge = true;
        return false;
    }
    const IMS_G double* b = s.bf_bounds + cell_index(sl, i, j) * 8;
    // the whole 64-byte line at once (the outer bounds used to be four dependent loads behind short-circuit tests)
    const dvec4 bi = *(const IMS_G dvec4*)b, bo = *(const IMS_G dvec4*)(b + 4);
    const double b0 = bi.x, b1 = bi.y, b2 = bi.z, b3 = bi.w, b4 = bo.x, b5 = bo.y, b6 = bo.z, b7 = bo.w;
    bool inside;
    if (x > b0 && x < b1 && y > b2 && y < b3) inside = true;
    else if (!(x >= b4 && x <= b5 && y >= b6 && y <= b7)) inside = false;
    else {
        // a wave takes this path whenever ONE of its 64 photons misses the inner bounds, i.e. almost always
        const double zfactor = ZF ? z : dtanh_pos(ddiv(z, 12.0));
        const int nV = (NV > 0) ? NV : s.num_vertices, npo = 2 * nV + 2;
        const IMS_G double* bnd = s.bf_boundary;
        inside = polygon_test<NV>(s, bnd + cell_index(sl, i, j) * npo * 2, bnd + cell_index(sl, i + 1, j) * npo * 2,
                                  bnd + cell_index(sl, i, j + 1) * npo * 2, x, y, zfactor);
    }
    if (!inside && want_edge) {
        off_edge = false;
        if (i == 0 && x < b0) off_edge = true;
        if (i == sl.nx - 1 && x > b1) off_edge = true;
        if (j == 0 && y < b2) off_edge = true;
        if (j == sl.ny - 1 && y > b3) off_edge = true;
    }
    return inside;
}

// GalSim's neighbour table of the pixel search, n = 0 (the pixel itself), 1 .. 8 counter-clockwise from the right:
// xoff = {0, 1, 1, 0, -1, -1, -1, 0, 1}, yoff = {0, 0, 1, 1, 1, 0, -1, -1, -1}, as 2-bit codes (offset + 1) so that a lookup
// is two ALU instructions instead of a dependent load
IMS_DEV int xoff(int n) { return (int)((0x24069u >> (2 * n)) & 3u) - 1; }
IMS_DEV int yoff(int n) { return (int)((0x006a5u >> (2 * n)) & 3u) - 1; }

// first neighbour GalSim's search tries for a point (x, y) of the unit pixel that is not inside its own polygon
IMS_DEV int search_step(double x, double y)
{
    if ((x > y) && (x > 1.0 - y)) return 1;
    if ((x > y) && (x < 1.0 - y)) return 7;
    if ((x < y) && (x > 1.0 - y)) return 3;
    return 5;
}

// SiliconSensor.accumulate, first half: everything that does not look at the pixel boundaries -- conversion depth from the
// absorption length, lateral walk of an inclined photon down to that depth, diffusion.  Returns false when the photon
// converts beyond the back of the sensor (lost).  coin: the photon's "pixel not found" coin (stay in the nominal pixel).
// has_angles: the chain contains a ray-tracing op, so dxdz/dydz are meaningful.
IMS_DEV bool land_convert(const ims_render_params_t& P, const ims_object_t& o, int64_t k, Rng& rng, const Photon& ph,
                          bool has_angles, double& x0, double& y0, double& zconv, bool& coin)
{
    const ims_sensor_t& s = *P.sensor;
    x0 = ph.x; y0 = ph.y;
    rng_block(rng, P.seed, o.obj_id, k, SLOT_SENSOR);
    double g0, g1;
    gauss_words(rng.w[0], rng.w[1], g0, g1);
    coin = (rng.w[3] & 0x80000000u) != 0u;
    const double f = ddiv(ph.wl - s.abs_wl_min, s.abs_wl_step);
    double abs_len;
    if (!(f > 0.0)) abs_len = s.abs_len[0];
    else if (f >= (double)(s.n_abs - 1)) abs_len = s.abs_len[s.n_abs - 1];
    else { const int t = (int)f; const double a = f - (double)t; const double v0 = s.abs_len[t]; abs_len = v0 + a * (s.abs_len[t + 1] - v0); }
    const double si_length = -abs_len * dlog_w(rng.w[2]);
    double dz = si_length;
    if (has_angles) {
        dz = ddiv(si_length, dsqrt_n(1.0 + ph.dxdz * ph.dxdz + ph.dydz * ph.dydz));
        if (dz > s.thick_m1) dz = s.thick_m1;
        const double dzp = ddiv(dz, s.pixel_size);
        x0 = x0 + ph.dxdz * dzp;
        y0 = y0 + ph.dydz * dzp;
    }
    zconv = s.thickness - dz;
    if (zconv < 0.0) return false;
    if (s.diff_step != 0.0) {
        double ds = s.diff_coef * dsqrt0(zconv * s.thickness);
        if (ds < 0.0) ds = 0.0;
        x0 = x0 + ds * g0;
        y0 = y0 + ds * g1;
    }
    return true;
}

// SiliconSensor.accumulate, second half: the pixel the converted photon (x0, y0) is collected in, by GalSim's search over
// the distorted pixel polygons.  ZF: z is the shrink factor (converted pool), else zconv.  Returns false when the photon
// is lost.  (Testing the nominal pixel and the first neighbour of the search together, both bounds lines and both polygons
// in one batch of loads, was measured: 41.2 against 41.5 us per round of the brightest star, +1 % on the C3 step -- the
// search is not where a round's latency goes; not kept.)
// ims_render_params_t.lazy_static: a photon that would look at the (absent) state of slot 0 is set aside for the second launch
// (k_margin_photons): position at the conversion depth, the depth with the coin in its sign, flux, object row.  No atomic in the
// common case: wavefront `wave_slot` of the launch owns an octet of records and a count byte (2 % of 64 photons: 1.3 records
// expected); a ninth record goes to the overflow region behind one counter.  (One counter for all -- an addition with return
// per wavefront -- made the fused launch of C3's 98 k objects 14 % SLOWER than reading the state: 600 k additions to one address.)
IMS_DEV void margin_append(const ims_render_params_t& P, const ims_object_t& o, double x0, double y0, double zconv, bool coin, double flux,
                           int64_t wave_slot)
{
    const unsigned long long active = __builtin_amdgcn_ballot_w64(true);
    const int lane = (int)(threadIdx.x & 63);
    const int rank = (int)__popcll(active & ((1ull << lane) - 1ull)), n = (int)__popcll(active);
    double* r;
    if (wave_slot >= 0 && rank < 8) {
        r = P.margin_list + 5 * (size_t)(wave_slot * 8 + rank);
        if (rank == 0) P.margin_wave_count[wave_slot] = (unsigned char)(n < 8 ? n : 8);
    } else {
        const int idx = atomicAdd(P.margin_count, 1);
        if ((unsigned int)idx >= P.margin_cap) { atomicAdd(P.margin_count + 1, 1); return; }        // (sized for every photon of the launch)
        r = P.margin_list + 5 * ((size_t)P.margin_waves * 8 + (size_t)idx);
    }
    r[0] = x0; r[1] = y0; r[2] = coin ? -zconv : zconv; r[3] = flux;
    r[4] = __longlong_as_double((long long)(&o - P.objects));
}

template <int NV = 0, bool ZF = false>
IMS_DEV bool land_search(const ims_render_params_t& P, const ims_object_t& o, double x0, double y0, double z, bool coin,
                         int& ix, int& iy, double flux = 0.0, int64_t wave_slot = -1)
{
    const ims_sensor_t& s = *P.sensor;
    const ims_bf_slot_t bs = s.bf_slots[slot_index(P, o)];
    const SlotView sl = { bs.xmin, bs.ymin, bs.nx, bs.ny, bs.offset };
    ix = (int)floor(x0 + 0.5); iy = (int)floor(y0 + 0.5);
    if (ix < o.stamp_xmin || ix > o.stamp_xmax || iy < o.stamp_ymin || iy > o.stamp_ymax) return false;
    const double x = x0 - (double)ix + 0.5, y = y0 - (double)iy + 0.5;
    bool off_edge = false;
    bool found;
    {
        // pristine slot 0 (ims_sensor_t.pristine_margin): no vertex of the pixel polygon is further than m from its
        // nominal place, so a point more than m from every edge passes the inner-bounds test -- known without the load
        const double m = s.pristine_margin;
        const int pi = ix - sl.xmin, pj = iy - sl.ymin;
        if (o.bf_state == 0 && m >= 0.0 && pi >= 0 && pi < sl.nx && pj >= 0 && pj < sl.ny && x > m && x < 1.0 - m && y > m && y < 1.0 - m)
            found = true;
        else if (!ZF && P.lazy_static != 0u && o.bf_state == 0 && m >= 0.0 && pi >= 0 && pi < sl.nx && pj >= 0 && pj < sl.ny) {
            margin_append(P, o, x0, y0, z, coin, flux, wave_slot);   // slot 0 holds no state: the second launch finishes this photon
            return false;
        } else
            found = inside_pixel<NV, ZF>(s, sl, ix, iy, x, y, z, true, off_edge);
    }
    PROBE_WG(3, 0);
    if (!found && off_edge) return false;
    int step = 0;
    if (!found) {
        // The search visits the eight neighbours in the order n_m = ((m step - 1) mod 8) + 1, m = 1 .. 8, and stops at the
        // first whose polygon holds the point.  A photon that no polygon holds (the polygons pulled in by the depth factor
        // leave slivers between them) walks all eight, and among the 10 000 photons of a round there always is one: as
        // eight dependent bounds-then-polygon tests that walk was most of the duration of a round's kernel.  So first the
        // OUTER bounds of all eight are fetched in one batch; a neighbour whose outer bounds do not contain the point
        // cannot hold it (inside_pixel would say no), and the ordered walk only tests the others -- the same answers.
        step = search_step(x, y);
        unsigned cand = 0u;
#pragma unroll
        for (int m = 1; m < 9; ++m) {
            const int n = ((m * step - 1) & 7) + 1;
            const int i = ix + xoff(n) - sl.xmin, j = iy + yoff(n) - sl.ymin;
            const bool valid = !(i < 0 || i >= sl.nx || j < 0 || j >= sl.ny);
            const IMS_G double* b = s.bf_bounds + (valid ? cell_index(sl, i, j) : sl.offset) * 8;
            const double o4 = b[4], o5 = b[5], o6 = b[6], o7 = b[7];
            const double xb = x - (double)xoff(n), yb = y - (double)yoff(n);
            if (valid && xb >= o4 && xb <= o5 && yb >= o6 && yb <= o7) cand |= 1u << m;
        }
        PROBE_WG(4, 0);
        for (int m = 1; m < 9 && cand != 0u; ++m) {
            if (!(cand & (1u << m))) continue;
            cand &= ~(1u << m);
            const int n = ((m * step - 1) & 7) + 1;
            const int jx = ix + xoff(n), jy = iy + yoff(n);
            bool dummy;
            if (inside_pixel<NV, ZF>(s, sl, jx, jy, x - (double)xoff(n), y - (double)yoff(n), z, false, dummy)) {
                ix = jx; iy = jy; found = true; break;
            }
        }
    }
    if (!found) {
        const int n = coin ? 0 : step;
        ix = ix + xoff(n); iy = iy + yoff(n);
    }
    // the caller deposits the charge (CCD image and, for tracked regions, the delta-charge image)
    return !(ix < o.stamp_xmin || ix > o.stamp_xmax || iy < o.stamp_ymin || iy > o.stamp_ymax);
}

// Decide the landing pixel.  Returns false when the photon is lost.
template <int NV = 0>
IMS_DEV bool land(const ims_render_params_t& P, const ims_object_t& o, int64_t k, Rng& rng, const Photon& ph,
                  bool silicon, bool has_angles, int& ix, int& iy, int64_t wave_slot = -1)
{
    if (!silicon || (o.flags & IMS_OBJ_FAINT)) {
        ix = (int)floor(ph.x + 0.5); iy = (int)floor(ph.y + 0.5);
        return !(ix < o.stamp_xmin || ix > o.stamp_xmax || iy < o.stamp_ymin || iy > o.stamp_ymax);
    }
    double x0, y0, zconv;
    bool coin;
    if (!land_convert(P, o, k, rng, ph, has_angles, x0, y0, zconv, coin)) return false;
    return land_search<NV, false>(P, o, x0, y0, zconv, coin, ix, iy, ph.flux, wave_slot);
}

IMS_DEV bool chain_has_angles(const ims_render_params_t& P)
{
    bool r = false;
    for (int k = 0; k < P.n_ops; ++k)
        if (P.ops[k].kind == IMS_OP_RUBIN_OPTICS || P.ops[k].kind == IMS_OP_RUBIN_DIFFRACTION_OPTICS) r = true;
    return r;
}

// The photon-op chain of a launch.  CHAIN 0: whatever the descriptor lists, one switch per operator.  CHAIN 1: imSim's
// default chain (config/imsim-config.yaml:281-320: TimeSampler, PupilAnnulusSampler, PhotonDCR, RubinDiffractionOptics,
// FocusDepth, Refraction, in that order -- the host checks the descriptor before it picks the kernel) as straight-line code:
// no operator loop, no switch, nothing of the photon copied where the branches of the switch would meet.
constexpr int IMS_DEFAULT_CHAIN_LEN = 6;
template <int CHAIN, unsigned long long LAYOUT = 0ull>
IMS_DEV void run_ops(const ims_render_params_t& P, const ims_object_t& o, int64_t k, Rng& rng, Photon& ph)
{
    if (CHAIN == 1) {
        apply_op<IMS_OP_TIME_SAMPLER>(P, 0, o, k, rng, ph);
        apply_op<IMS_OP_PUPIL_ANNULUS_SAMPLER>(P, 1, o, k, rng, ph);
        apply_op<IMS_OP_PHOTON_DCR>(P, 2, o, k, rng, ph);
        apply_op<IMS_OP_RUBIN_DIFFRACTION_OPTICS, LAYOUT>(P, 3, o, k, rng, ph);
        apply_op<IMS_OP_FOCUS_DEPTH>(P, 4, o, k, rng, ph);
        apply_op<IMS_OP_REFRACTION>(P, 5, o, k, rng, ph);
    } else {
        for (int q = 0; q < P.n_ops; ++q) apply_op(P, q, o, k, rng, ph);
    }
}

}  // namespace ims
