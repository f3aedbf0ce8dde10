/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h).
 *
 * orc_fft.c: CPU restatement of the FFT branch of LSST_SiliconBuilder.draw (imsim/stamp.py:482-525):
 * analytic half-spectrum of Convolve([gal] + psfs) with the pixel response (GalSim drawImage
 * method='fft', third-party, unpinned), clip of negatives (:519), Poisson noise (:522) and the
 * stamp -> CCD add (:524).  The inverse transform itself is done by the test with numpy.fft.
 * Parity unpinned at the bit level (GalSim's FFT renderer and boost's Poisson deviate are not
 * reproducible here); pinned by the reference's FFT-vs-phot criteria (tests/test_psf.py:341-438).
 */
#include "orc.h"

static double ktable_lookup(const ims_lin_tables_t* t, int table, double arg)
{
    const double* v = t->val + (int64_t)table * t->n_pts;
    double f = (arg - t->arg_min) / t->arg_step;
    if (!(f > 0.0)) return v[0];
    if (f >= (double)(t->n_pts - 1)) return 0.0;
    int i = (int)f;
    double a = f - (double)i;
    return v[i] + a * (v[i + 1] - v[i]);
}

/* spectrum at one k-vector (csrc/ims_fft.h::kspace_at restated) */
static void kspace_at(const ims_fft_params_t* P, const ims_fft_object_t* o, double kx, double ky, double* re, double* im)
{
    double amp = o->flux;
    if (o->prof_ktable >= 0) {
        double qx = o->jac[0] * kx + o->jac[2] * ky;
        double qy = o->jac[1] * kx + o->jac[3] * ky;
        amp = amp * ktable_lookup(&P->ktables, o->prof_ktable, orc_sqrt(qx * qx + qy * qy) * o->prof_scale);
    }
    double k2 = kx * kx + ky * ky;
    for (int c = 0; c < P->n_kpsf; ++c) {
        const ims_kpsf_t* p = &P->kpsf[c];
        if (p->kind == IMS_KPSF_GAUSSIAN) amp = amp * orc_exp(-0.5 * p->p0 * p->p0 * k2);
        else if (p->kind == IMS_KPSF_KOLMOGOROV) {
            if (k2 > 0.0) amp = amp * orc_exp(-orc_pow(orc_sqrt(k2) / p->p0, 5.0 / 3.0));
        } else amp = amp * ktable_lookup(&P->ktables, p->table, orc_sqrt(k2) * p->p0);
    }
    double hx = 0.5 * kx * P->pixel_scale, hy = 0.5 * ky * P->pixel_scale;
    double s, c;
    if (hx != 0.0) { orc_sincos(hx, &s, &c); amp = amp * (s / hx); }
    if (hy != 0.0) { orc_sincos(hy, &s, &c); amp = amp * (s / hy); }
    double ph = (kx * o->cx + ky * o->cy) * P->pixel_scale;
    orc_sincos(ph, &s, &c);
    *re = amp * c;
    *im = -amp * s;
}

void orc_fft_kspace_fill(const ims_fft_params_t* P, const ims_fft_object_t* objs, int64_t n_objects, double* kbuf)
{
    for (int64_t oi = 0; oi < n_objects; ++oi) {
        const ims_fft_object_t* o = &objs[oi];
        int n = o->nfft, nh = n / 2 + 1;
        double dk = ORC_TWO_PI / ((double)n * P->pixel_scale);
        double ks = ORC_TWO_PI / P->pixel_scale;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < nh; ++j) {
                double kx = (double)j * dk;
                double ky = (double)(i < n / 2 ? i : i - n) * dk;
                double re, im;
                if (P->n_alias <= 0) kspace_at(P, o, kx, ky, &re, &im);
                else {
                    re = 0.0; im = 0.0;
                    for (int b = -P->n_alias; b <= P->n_alias; ++b)
                        for (int a = -P->n_alias; a <= P->n_alias; ++a) {
                            double r1, i1;
                            kspace_at(P, o, kx + (double)a * ks, ky + (double)b * ks, &r1, &i1);
                            re = re + r1; im = im + i1;
                        }
                }
                int64_t e = o->k_offset + (int64_t)i * nh + j;
                kbuf[2 * e] = re;
                kbuf[2 * e + 1] = im;
            }
    }
}

static double orc_lgamma(double x)
{
    double shift = 0.0;
    while (x < 8.0) { shift = shift + orc_log(x); x = x + 1.0; }
    double ix = 1.0 / x, ix2 = ix * ix;
    double ser = -1.0 / 1680.0;
    ser = orc_fma(ser, ix2, 1.0 / 1260.0);
    ser = orc_fma(ser, ix2, -1.0 / 360.0);
    ser = orc_fma(ser, ix2, 1.0 / 12.0);
    return (x - 0.5) * orc_log(x) - x + 0.91893853320467274178 + ser * ix - shift;
}

double orc_poisson(double mean, uint64_t seed, int64_t obj_id, int64_t pixel)
{
    if (!(mean > 0.0)) return 0.0;
    uint32_t slot = 32;
    if (mean < 10.0) {
        double L = orc_exp(-mean), p = 1.0, k = 0.0;
        for (int it = 0; it < 64; ++it) {
            orc_draw_t d = orc_draw(seed, obj_id, pixel, slot++);
            p = p * orc_u01_open(d.a);
            if (p <= L) return k;
            k = k + 1.0;
            p = p * orc_u01_open(d.b);
            if (p <= L) return k;
            k = k + 1.0;
        }
        return k;
    }
    double slam = orc_sqrt(mean), loglam = orc_log(mean);
    double b = 0.931 + 2.53 * slam;
    double a = -0.059 + 0.02483 * b;
    double invalpha = 1.1239 + 1.1328 / (b - 3.4);
    double vr = 0.9277 - 3.6224 / (b - 2.0);
    for (int it = 0; it < 64; ++it) {
        orc_draw_t d = orc_draw(seed, obj_id, pixel, slot++);
        double U = orc_u01(d.a) - 0.5;
        double V = orc_u01_open(d.b);
        double us = 0.5 - fabs(U);
        double k = floor((2.0 * a / us + b) * U + mean + 0.43);
        if (us >= 0.07 && V <= vr) return k;
        if (k < 0.0 || (us < 0.013 && V > us)) continue;
        if (orc_log(V) + orc_log(invalpha) - orc_log(a / (us * us) + b) <= -mean + k * loglam - orc_lgamma(k + 1.0)) return k;
    }
    return floor(mean + 0.5);
}

/* clip, noise, add; image is a double accumulation buffer [ny][nx] */
void orc_fft_finish(const ims_fft_params_t* P, const ims_fft_object_t* objs, int64_t n_objects, const double* rbuf,
                    double* image, double* realized)
{
    for (int64_t oi = 0; oi < n_objects; ++oi) {
        const ims_fft_object_t* o = &objs[oi];
        for (int iy = 0; iy < o->nfft; ++iy)
            for (int ix = 0; ix < o->nfft; ++ix) {
                int px = o->x0 + ix, py = o->y0 + iy;
                if (px < o->stamp_xmin || px > o->stamp_xmax || py < o->stamp_ymin || py > o->stamp_ymax) continue;
                int64_t local = (int64_t)iy * o->nfft + ix;
                double v = rbuf[o->r_offset + local];
                if (v < 0.0) v = 0.0;
                if (realized) realized[oi] += v;
                if (P->add_noise) v = orc_poisson(v, P->seed, o->obj_id, local);
                int cx = px - P->xmin, cy = py - P->ymin;
                if (cx < 0 || cx >= P->nx || cy < 0 || cy >= P->ny) continue;
                image[(int64_t)cy * P->nx + cx] += v;
            }
    }
}

double orc_spike_stencil(const ims_spikes_t* k, int a, int b)
{
    double x = (double)a, y = (double)b;
    double xr = k->cos0 * x + k->sin0 * y;
    double yr = -k->sin0 * x + k->cos0 * y;
    double m = fabs(xr) < fabs(yr) ? fabs(xr) : fabs(yr);
    /* exact zeros of the stencil (further than a pixel from both arms and outside the swept wedge, whose points lie within
       r sin(d_alpha / 2) of an arm) return before the arctangents: the same +0.0 the full expression gives */
    if (m > 1.0) {
        double t = m - 1.0e-6, lim = 0.5 * fabs(k->d_alpha) + 1.0e-6;
        if (t * t > lim * lim * (x * x + y * y)) return 0.0;
    }
    double val = 1.0 - m;
    if (val < 0.0) val = 0.0;
    double dth = orc_atan2(y, x) - k->a_lo;
    dth = dth - floor(dth / ORC_PI_2) * ORC_PI_2;
    if (dth <= k->d_alpha) val = 1.0;
    double r = orc_sqrt(x * x + y * y);
    double prof = 0.63661977236758134308 * (orc_atan((r + 0.5) * k->scale / k->r0) - orc_atan((r - 0.5) * k->scale / k->r0));
    double arc = r * k->d_alpha;
    val = val * prof / (arc > 1.0 ? arc : 1.0);
    if (a == 0 && b == 0) val = 2.0 * val;
    return val;
}

/* apply_diffraction_psf on every object's stamp (imsim/diffraction_fft.py:126-208), after the clip */
void orc_fft_spikes(const ims_fft_params_t* P, const ims_fft_object_t* objs, int64_t n_objects, const double* rin, double* rout)
{
    const ims_spikes_t* k = &P->spikes;
    for (int64_t oi = 0; oi < n_objects; ++oi) {
        const ims_fft_object_t* o = &objs[oi];
        int n = o->nfft;
        int r0 = 0x7fffffff, r1 = -1, c0 = 0x7fffffff, c1 = -1;
        for (int iy = 0; iy < n; ++iy)
            for (int ix = 0; ix < n; ++ix) {
                int px = o->x0 + ix, py = o->y0 + iy;
                if (px < o->stamp_xmin || px > o->stamp_xmax || py < o->stamp_ymin || py > o->stamp_ymax) continue;
                if (k->enabled && rin[o->r_offset + (int64_t)iy * n + ix] > k->threshold) {
                    if (iy < r0) r0 = iy;
                    if (iy > r1) r1 = iy;
                    if (ix < c0) c0 = ix;
                    if (ix > c1) c1 = ix;
                }
            }
        for (int iy = 0; iy < n; ++iy)
            for (int ix = 0; ix < n; ++ix) {
                int64_t local = (int64_t)iy * n + ix;
                double v = rin[o->r_offset + local];
                if (v < 0.0) v = 0.0;
                int px = o->x0 + ix, py = o->y0 + iy;
                int in_stamp = !(px < o->stamp_xmin || px > o->stamp_xmax || py < o->stamp_ymin || py > o->stamp_ymax);
                if (k->enabled && in_stamp && r1 >= r0) {
                    if (iy >= r0 && iy <= r1 && ix >= c0 && ix <= c1) v = 0.0;
                    double acc = 0.0;
                    /* further from the arms of every source pixel's cross than any non-zero stencil value reaches: the whole sum is
                       exact zeros (the offsets to the sources differ from the offset to the box centre by at most half the box) */
                    int none = 0;
                    {
                        double ac = (double)iy - 0.5 * (double)(r0 + r1), bc = (double)ix - 0.5 * (double)(c0 + c1);
                        double ha = 0.5 * (double)(r1 - r0), hb = 0.5 * (double)(c1 - c0);
                        double e = fabs(k->cos0) * ha + fabs(k->sin0) * hb + fabs(k->sin0) * ha + fabs(k->cos0) * hb;
                        double xc = k->cos0 * ac + k->sin0 * bc, yc = -k->sin0 * ac + k->cos0 * bc;
                        double mc = (fabs(xc) < fabs(yc) ? fabs(xc) : fabs(yc)) - e;
                        double rmax = sqrt(ac * ac + bc * bc) + sqrt(ha * ha + hb * hb) + 1.0;
                        double lim = 0.5 * fabs(k->d_alpha) + 1.0e-6;
                        none = mc > 1.0 + 1.0e-3 && mc - 1.0e-3 > lim * rmax;
                    }
                    if (!none) {
                        /* of a source row only the columns near the two arms through this pixel can be non-zero (|xr| or |yr| within
                           the stencil's reach: one interval of columns each), visited in ascending order as the full row was */
                        double lim = 0.5 * fabs(k->d_alpha) + 1.0e-6;
                        double bfar = fabs((double)(ix - c0)) > fabs((double)(ix - c1)) ? fabs((double)(ix - c0)) : fabs((double)(ix - c1));
                        int has_s = fabs(k->sin0) > 1.0e-12, has_c = fabs(k->cos0) > 1.0e-12;
                        double inv_s = has_s ? 1.0 / k->sin0 : 0.0, inv_c = has_c ? 1.0 / k->cos0 : 0.0;
                        for (int ry = r0; ry <= r1; ++ry) {
                            int a = iy - ry;
                            if (a < -k->cutoff || a > k->cutoff) continue;
                            double da = (double)a;
                            double T = 1.0 + 1.0e-3 + lim * sqrt(da * da + bfar * bfar);
                            int lo[2], hi[2];
                            if (has_s) {
                                double b1 = (-T - k->cos0 * da) * inv_s, b2 = (T - k->cos0 * da) * inv_s;
                                double bl = b1 < b2 ? b1 : b2, bh = b1 < b2 ? b2 : b1;
                                lo[0] = (int)floor((double)ix - bh) - 1; hi[0] = (int)ceil((double)ix - bl) + 1;
                            } else if (fabs(k->cos0 * da) <= T) { lo[0] = c0; hi[0] = c1; }
                            else { lo[0] = 1; hi[0] = 0; }
                            if (has_c) {
                                double b1 = (-T + k->sin0 * da) * inv_c, b2 = (T + k->sin0 * da) * inv_c;
                                double bl = b1 < b2 ? b1 : b2, bh = b1 < b2 ? b2 : b1;
                                lo[1] = (int)floor((double)ix - bh) - 1; hi[1] = (int)ceil((double)ix - bl) + 1;
                            } else if (fabs(k->sin0 * da) <= T) { lo[1] = c0; hi[1] = c1; }
                            else { lo[1] = 1; hi[1] = 0; }
                            for (int q = 0; q < 2; ++q) { if (lo[q] < c0) lo[q] = c0; if (hi[q] > c1) hi[q] = c1; }
                            if (lo[1] < lo[0]) { int tl = lo[0], th = hi[0]; lo[0] = lo[1]; hi[0] = hi[1]; lo[1] = tl; hi[1] = th; }
                            if (hi[0] >= lo[0] && hi[1] >= lo[1] && lo[1] <= hi[0] + 1) { if (hi[1] > hi[0]) hi[0] = hi[1]; lo[1] = 1; hi[1] = 0; }
                            for (int q = 0; q < 2; ++q)
                                for (int rx = lo[q]; rx <= hi[q]; ++rx) {
                                    int b = ix - rx;
                                    if (b < -k->cutoff || b > k->cutoff) continue;
                                    double src = rin[o->r_offset + (int64_t)ry * n + rx];
                                    if (src < 0.0) src = 0.0;
                                    acc = acc + orc_spike_stencil(k, a, b) / k->norm * src;
                                }
                        }
                    }
                    v = v + acc;
                }
                rout[o->r_offset + local] = v;
            }
    }
}

void orc_test_stencil(const ims_spikes_t* k, int half, double* out)
{
    int w = 2 * half + 1;
    for (int a = -half; a <= half; ++a)
        for (int b = -half; b <= half; ++b) out[(a + half) * w + (b + half)] = orc_spike_stencil(k, a, b);
}

void orc_test_poisson(const double* mean, double* out, int64_t n, uint64_t seed, int64_t obj_id)
{
    for (int64_t i = 0; i < n; ++i) out[i] = orc_poisson(mean[i], seed, obj_id, i);
}
