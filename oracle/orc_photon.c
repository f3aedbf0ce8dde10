/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h).
 *
 * orc_photon.c: CPU restatement of photon shooting and the simple photon operators, written the
 * way the reference runs them: one operator at a time over a whole photon array
 * (GalSim drawImage(method='phot') as driven by imsim/stamp.py:527-573 -- recalled call order in
 * shoot -> each photon_op.applyTo -> sensor.accumulate).
 *
 * Third-party algorithms restated here (GalSim is unpinned in the reference, setup.py:21; absent
 * from /root/reference): WavelengthSampler, GSObject.shoot for DeltaFunction/Sersic/Gaussian,
 * PSF-as-PhotonOp (PhotonArray.convolve), TimeSampler, PupilAnnulusSampler, PhotonDCR,
 * FocusDepth, Refraction.  Parity unpinned at the bit level (no golden photon arrays exist in the
 * reference); distributions are pinned by the reference tests' statistical criteria in tests/.
 */
#include <stdlib.h>
#include "orc.h"

/* ---------- table helpers ---------- */
double orc_lin_lookup(const ims_lin_tables_t* t, int table, double arg)
{
    const double* v = t->val + (int64_t)table * t->n_pts;
    double f = (arg - t->arg_min) / t->arg_step;
    if (!(f > 0.0)) return v[0];
    int n = t->n_pts;
    if (f >= (double)(n - 1)) return v[n - 1];
    int i = (int)f;
    double a = f - (double)i;
    return v[i] + a * (v[i + 1] - v[i]);
}

/* inverse-CDF sample of a radial table: returns r^2 in table units */
double orc_radial_r2(const ims_radial_tables_t* t, int table, double u)
{
    int nb = t->n_bins;
    const double* cdf = t->cdf + (int64_t)table * (nb + 1);
    const double* r2 = t->r2 + (int64_t)table * (nb + 1);
    /* largest i in [0, nb-1] with cdf[i] <= u */
    int lo = 0, hi = nb;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (cdf[mid] <= u) lo = mid; else hi = mid;
    }
    double w = cdf[lo + 1] - cdf[lo];
    double f = (w > 0.0) ? (u - cdf[lo]) / w : 0.0;
    return r2[lo] + f * (r2[lo + 1] - r2[lo]);
}

/* ---------- photon array ---------- */
int orc_photons_alloc(ims_photons_t* p, int64_t n)
{
    p->n = n;
    size_t b = (size_t)(n > 0 ? n : 1) * sizeof(double);
    p->x = malloc(b); p->y = malloc(b); p->flux = malloc(b); p->dxdz = malloc(b); p->dydz = malloc(b);
    p->wavelength = malloc(b); p->pupil_u = malloc(b); p->pupil_v = malloc(b); p->time = malloc(b);
    p->obj_index = malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
    return (p->x && p->y && p->flux && p->dxdz && p->dydz && p->wavelength && p->pupil_u && p->pupil_v
            && p->time && p->obj_index) ? 0 : -1;
}
void orc_photons_free(ims_photons_t* p)
{
    free(p->x); free(p->y); free(p->flux); free(p->dxdz); free(p->dydz);
    free(p->wavelength); free(p->pupil_u); free(p->pupil_v); free(p->time); free(p->obj_index);
    memset(p, 0, sizeof(*p));
}

/* offset drawn from |K| of an image profile's interpolant: bisection in the tabulated cumulative distribution of |K|
 * (ims_image_tables_t.kx, kcdf), linear inside the interval; *neg = K is negative at the offset */
static double interp_offset(const ims_image_tables_t* T, double u, int* neg)
{
    int lo = 0, hi = T->n_k;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (T->kcdf[mid] <= u) lo = mid; else hi = mid; }
    double c0 = T->kcdf[lo], wd = T->kcdf[lo + 1] - c0;
    double f = (wd > 0.0) ? (u - c0) / wd : 0.0;
    double x0 = T->kx[lo];
    double d = fma(f, T->kx[lo + 1] - x0, x0);
    double ad = fabs(d);
    *neg = (ad > T->neg[0] && ad < T->neg[1]) || (ad > T->neg[2] && ad < T->neg[3]);
    return d;
}

/* ---------- shooting (stamp.py:562-572 -> GalSim drawImage phot) ---------- */
/* Photon j (j = 0..n_phot-1) of `obj` is written at pool index base+j; its stream index is
 * obj->phot_first + j.  Positions are left RELATIVE to the object's image_pos, in pixels. */
void orc_shoot_object(const ims_render_params_t* P, const ims_object_t* obj, int32_t obj_index,
                      ims_photons_t* ph, int64_t base)
{
    const double j0 = obj->jac[0], j1 = obj->jac[1], j2 = obj->jac[2], j3 = obj->jac[3];
    const double w0 = obj->winv[0], w1 = obj->winv[1], w2 = obj->winv[2], w3 = obj->winv[3];
    for (int64_t j = 0; j < obj->n_phot; ++j) {
        int64_t i = base + j, k = obj->phot_first + j;
        orc_words_t d0 = orc_words(P->seed, obj->obj_id, k, ORC_SLOT_SHOOT);
        /* WavelengthSampler */
        double wl = obj->sed_wave;
        if (obj->sed_table >= 0) wl = orc_lin_lookup(&P->sed, obj->sed_table, orc_w01(d0.w[0]));
        /* profile */
        double pu = 0.0, pv = 0.0, fscale = 1.0;
        if (obj->prof_table != IMS_PROF_POINT) {
            double gu, gv;
            if (obj->prof_table >= 0) {
                double r2 = orc_radial_r2(&P->radial, obj->prof_table, orc_w01(d0.w[1]));
                double r = orc_sqrt(r2) * obj->prof_scale;
                double s, c;
                orc_sincos2pi_w(d0.w[2], &s, &c);
                gu = r * c; gv = r * s;
            } else if (obj->prof_table == IMS_PROF_BOX) {          /* galsim.Box: uniform over length x width */
                gu = (orc_w01(d0.w[1]) - 0.5) * obj->prof_scale;
                gv = (orc_w01(d0.w[2]) - 0.5) * obj->prof_aux;
            } else if (obj->prof_table == IMS_PROF_IMAGE) {        /* galsim.InterpolatedImage */
                const ims_image_tables_t* T = &P->images;
                int kimg = (int)obj->prof_aux;
                int w = T->size[2 * kimg], h = T->size[2 * kimg + 1];
                const double* cdf = T->cdf + T->offset[kimg];
                double u = orc_w01(d0.w[1]);
                int lo = 0, hi = w * h;
                while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (cdf[mid] <= u) lo = mid; else hi = mid; }
                double c0 = cdf[lo], wd = cdf[lo + 1] - c0;
                double f = (wd > 0.0) ? (u - c0) / wd : 0.0;
                double u2 = orc_w01(d0.w[2]);
                if (T->interp != 0) {
                    /* SBInterpolatedImage::shoot + Interpolant::shoot: pixel centre plus an offset per axis drawn from |K| of the
                     * x-interpolant by its tabulated inverse CDF, flux signed like K(dx) K(dy) */
                    int lx, ly;
                    double dx = interp_offset(T, f, &lx), dy = interp_offset(T, u2, &ly);
                    gu = ((((double)(lo % w) + 0.5) + dx) - 0.5 * (double)w) * obj->prof_scale;
                    gv = ((((double)(lo / w) + 0.5) + dy) - 0.5 * (double)h) * obj->prof_scale;
                    fscale = (lx != ly) ? -T->norm : T->norm;
                } else {
                    gu = ((((double)(lo % w) + f) - 0.5 * (double)w)) * obj->prof_scale;
                    gv = ((((double)(lo / w) + u2) - 0.5 * (double)h)) * obj->prof_scale;
                }
            } else {                                                /* galsim.RandomKnots */
                uint32_t m = (uint32_t)(((uint64_t)d0.w[1] * (uint64_t)(uint32_t)obj->prof_aux) >> 32);
                orc_words_t kd = orc_words(P->seed, obj->obj_id, (int64_t)m, ORC_SLOT_KNOT);
                double g0, g1;
                orc_gauss_words(kd.w[0], kd.w[1], &g0, &g1);
                gu = obj->prof_scale * g0; gv = obj->prof_scale * g1;
            }
            pu = j0 * gu + j1 * gv;
            pv = j2 * gu + j3 * gv;
        }
        ph->x[i] = w0 * pu + w1 * pv;
        ph->y[i] = w2 * pu + w3 * pv;
        ph->flux[i] = (obj->prof_table == IMS_PROF_IMAGE) ? obj->flux_per_photon * fscale : obj->flux_per_photon;
        ph->dxdz[i] = 0.0; ph->dydz[i] = 0.0;
        ph->wavelength[i] = wl;
        ph->pupil_u[i] = 0.0; ph->pupil_v[i] = 0.0; ph->time[i] = 0.0;
        ph->obj_index[i] = obj_index;
    }
}

/* Sum over layers of the gradient of the bilinear interpolant of the (periodic) screen
 * (galsim AtmosphericScreen._wavefront_gradient over LookupTable2D(..., edge_mode='wrap'), recalled;
 * call site imsim/atmPSF.py:306-315).  Returns d(OPD)/du, d(OPD)/dv in nm/m. */
static int wrap_index(double fl, double dn, double inv_n)
{
    /* fl mod n for an integer-valued double; the estimate of the quotient may be off by one */
    double r = fl - floor(fl * inv_n) * dn;
    if (r < 0.0) r = r + dn;
    if (r >= dn) r = r - dn;
    return (int)r;
}

void orc_screen_gradient(const ims_atmosphere_t* A, double pu, double pv, double t, double tanx, double tany,
                         double* gx, double* gy)
{
    double sx = 0.0, sy = 0.0;
    const int n = A->npix;
    const double dn = (double)n, inv_n = 1.0 / dn, inv_scale = 1.0 / A->scale;
    for (int l = 0; l < A->n_layers; ++l) {
        double x = pu - t * A->vx[l] + A->alt[l] * tanx;
        double y = pv - t * A->vy[l] + A->alt[l] * tany;
        double fx = (x - A->x0) * inv_scale, fy = (y - A->x0) * inv_scale;
        double flx = floor(fx), fly = floor(fy);
        double ax = fx - flx, ay = fy - fly;
        int ix = wrap_index(flx, dn, inv_n), iy = wrap_index(fly, dn, inv_n);
        int ix1 = ix + 1 == n ? 0 : ix + 1, iy1 = iy + 1 == n ? 0 : iy + 1;
        const float* S = A->screens + (int64_t)l * n * n;
        int64_t r0 = (int64_t)iy * n, r1 = (int64_t)iy1 * n;
        double f00 = (double)S[r0 + ix], f10 = (double)S[r0 + ix1];
        double f01 = (double)S[r1 + ix], f11 = (double)S[r1 + ix1];
        sx = sx + ((f10 - f00) * (1.0 - ay) + (f11 - f01) * ay);
        sy = sy + ((f01 - f00) * (1.0 - ax) + (f11 - f10) * ax);
    }
    *gx = sx * inv_scale; *gy = sy * inv_scale;
}

/* PSF components act as photon ops: shoot the same number of photons and add positions
 * (GSObject.applyTo -> PhotonArray.convolve; stamp.py:553 `photon_ops = psfs + photon_ops`). */
void orc_apply_psf(const ims_render_params_t* P, const ims_object_t* obj, int comp,
                   ims_photons_t* ph, int64_t base)
{
    const ims_psf_component_t* c = &P->psf[comp];
    const double w0 = obj->winv[0], w1 = obj->winv[1], w2 = obj->winv[2], w3 = obj->winv[3];
    for (int64_t j = 0; j < obj->n_phot; ++j) {
        int64_t i = base + j, k = obj->phot_first + j;
        orc_words_t d = orc_words(P->seed, obj->obj_id, k, ORC_SLOT_PSF + ((uint32_t)comp >> 1));
        uint32_t wa = d.w[2 * (comp & 1)], wb = d.w[2 * (comp & 1) + 1];
        double scale = c->p0;
        if (c->chrom_alpha != 0.0) scale = scale * orc_pow(ph->wavelength[i] / c->chrom_base, c->chrom_alpha);
        double ku, kv;
        if (c->kind == IMS_PSF_GAUSSIAN) {
            double g0, g1;
            orc_gauss_words(wa, wb, &g0, &g1);
            ku = scale * g0; kv = scale * g1;
        } else if (c->kind == IMS_PSF_DOUBLE_GAUSSIAN) {
            /* DoubleGaussianPSF = sum of two Gaussians (atmPSF.py:478-484): Sum.shoot gives a photon to the
             * first component with probability p2 */
            orc_words_t ds = orc_words(P->seed, obj->obj_id, k, ORC_SLOT_PSF_TIME + (uint32_t)comp);
            double sigma = (orc_w01(ds.w[0]) < c->p2) ? scale : c->p1;
            double g0, g1;
            orc_gauss_words(wa, wb, &g0, &g1);
            ku = sigma * g0; kv = sigma * g1;
        } else if (c->kind == IMS_PSF_SCREENS) {
            /* PhaseScreenPSF geometric shooting: random pupil position and arrival time, kick =
             * wavefront gradient; the photon keeps (pupil_u, pupil_v, time) for later operators */
            const ims_atmosphere_t* A = P->atm;
            double ro2 = A->aper_r_outer * A->aper_r_outer, ri2 = A->aper_r_inner * A->aper_r_inner;
            double r = orc_sqrt(ri2 + orc_w01(wa) * (ro2 - ri2));
            double s, cc;
            orc_sincos2pi_w(wb, &s, &cc);
            double pu = r * cc, pv = r * s;
            orc_words_t dt = orc_words(P->seed, obj->obj_id, k, ORC_SLOT_PSF_TIME + (uint32_t)comp);
            double t = A->t0 + orc_w01(dt.w[0]) * A->exptime;
            double gx, gy;
            orc_screen_gradient(A, pu, pv, t, obj->atm_tan_x, obj->atm_tan_y, &gx, &gy);
            ku = scale * gx; kv = scale * gy;
            ph->pupil_u[i] = pu; ph->pupil_v[i] = pv; ph->time[i] = t;
        } else {
            double r2 = orc_radial_r2(&P->radial, c->table, orc_w01(wa));
            double r = orc_sqrt(r2) * scale;
            double s, cc;
            orc_sincos2pi_w(wb, &s, &cc);
            ku = r * cc; kv = r * s;
        }
        ph->x[i] = ph->x[i] + (w0 * ku + w1 * kv);
        ph->y[i] = ph->y[i] + (w2 * ku + w3 * kv);
    }
}

/* move to CCD coordinates: x += image_pos (the `shift_photons` of photon_ops.py:100-102 and the
 * stamp_center shift of stamp.py:740-742, done once here because everything downstream works in
 * full-image coordinates) */
void orc_shift_to_image(const ims_object_t* obj, ims_photons_t* ph, int64_t base)
{
    for (int64_t j = 0; j < obj->n_phot; ++j) {
        ph->x[base + j] = obj->x0 + ph->x[base + j];
        ph->y[base + j] = obj->y0 + ph->y[base + j];
    }
}

/* ---------- simple photon ops; `k` is the photon's stream index, looked up through obj_index ---------- */
static inline int64_t stream_index(const ims_render_params_t* P, const ims_photons_t* ph,
                                   const int64_t* photon_offset, int64_t i, const ims_object_t** obj)
{
    int32_t oi = ph->obj_index[i];
    *obj = &P->objects[oi];
    return (*obj)->phot_first + (i - photon_offset[oi]);
}

/* Filippenko (1982) air index as used by GalSim's dcr module (recalled), in
 * the spec-v4 form: uniform pressure/temperature/water factors (orc_air_factors) and the dispersion
 * formula over one denominator:
 *   n-1 = air_p (64.328 + 29498.1/(146 - s2) + 255.4/(41 - s2)) - air_w (0.0624 - 0.00068 s2), s2 = 1/w2 */
void orc_air_factors(double p_kpa, double t_k, double h2o_kpa, double* air_p, double* air_w)
{
    double Pm = p_kpa * 7.50061683;
    double T = t_k - 273.15;
    double W = h2o_kpa * 7.50061683;
    double tf = 1.0 + 0.003661 * T;
    *air_p = 1.0e-6 * (Pm * (1.0 + (1.049 - 0.0157 * T) * 1.0e-6 * Pm) / (720.883 * tf));
    *air_w = W * 1.0e-6 / tf;
}
double orc_air_n_minus_one(double wave_nm, double air_p, double air_w)
{
    double wm = wave_nm * 1.0e-3;
    double w2 = wm * wm;
    double d1 = orc_fma(146.0, w2, -1.0), d2 = orc_fma(41.0, w2, -1.0);
    double den = d1 * d2;
    double num = orc_fma(29498.1, d2, 255.4 * d1);
    double disp = orc_fma(64.328, den, w2 * num);
    double wat = orc_fma(0.0624, w2, -0.000680) * den;
    return (air_p * (disp * w2) - air_w * wat) / (den * w2);
}
static double refraction_r0(double nm1)
{
    return nm1 * (nm1 + 2.0) / 2.0 / (nm1 * nm1 + 2.0 * nm1 + 1.0);
}

/* launch-wide derived fields (include/imsim_hip.h: ims_fill_derived_op / ims_fill_derived_medium) */
int orc_fill_derived_op(ims_op_t* op)
{
    if (op->kind == IMS_OP_PHOTON_DCR) {
        orc_air_factors(op->p[1], op->p[2], op->p[3], &op->p[5], &op->p[6]);
        op->p[7] = refraction_r0(orc_air_n_minus_one(op->p[0], op->p[5], op->p[6]));
    }
    return 0;
}
int orc_fill_derived_medium(int32_t kind, double* c6)
{
    if (kind == IMS_MEDIUM_AIR) orc_air_factors(c6[0], c6[1], c6[2], &c6[3], &c6[4]);
    return 0;
}

void orc_apply_op(const ims_render_params_t* P, int op_index, ims_photons_t* ph,
                  const int64_t* photon_offset)
{
    const ims_op_t* op = &P->ops[op_index];
    const uint32_t slot = ORC_SLOT_OP + ((uint32_t)op_index >> 1);
    const int wb = 2 * (op_index & 1);          /* this op's two words of the block */
    const int64_t n = ph->n;
    switch (op->kind) {
    case IMS_OP_TIME_SAMPLER:
        for (int64_t i = 0; i < n; ++i) {
            const ims_object_t* obj; int64_t k = stream_index(P, ph, photon_offset, i, &obj);
            if (obj->flags & IMS_OBJ_FAINT) continue;
            orc_words_t d = orc_words(P->seed, obj->obj_id, k, slot);
            ph->time[i] = op->p[0] + orc_w01(d.w[wb]) * op->p[1];
        }
        break;
    case IMS_OP_PUPIL_ANNULUS_SAMPLER: {
        double ro2 = op->p[0] * op->p[0], ri2 = op->p[1] * op->p[1];
        for (int64_t i = 0; i < n; ++i) {
            const ims_object_t* obj; int64_t k = stream_index(P, ph, photon_offset, i, &obj);
            if (obj->flags & IMS_OBJ_FAINT) continue;
            orc_words_t d = orc_words(P->seed, obj->obj_id, k, slot);
            double r = orc_sqrt(ri2 + orc_w01(d.w[wb]) * (ro2 - ri2));
            double s, c;
            orc_sincos2pi_w(d.w[wb + 1], &s, &c);
            ph->pupil_u[i] = r * c;
            ph->pupil_v[i] = r * s;
        }
        break; }
    case IMS_OP_PHOTON_DCR: {
        double base_r0 = op->p[7];               /* derived fields p5..p7: orc_fill_derived_op */
        for (int64_t i = 0; i < n; ++i) {
            const ims_object_t* obj; (void)stream_index(P, ph, photon_offset, i, &obj);
            if (obj->flags & IMS_OBJ_FAINT) continue;
            double r0 = refraction_r0(orc_air_n_minus_one(ph->wavelength[i], op->p[5], op->p[6]));
            double shift = (r0 - base_r0) * obj->dcr_tanz * op->p[4];   /* arcsec */
            double du = -shift * obj->dcr_sinp;
            double dv = shift * obj->dcr_cosp;
            ph->x[i] = ph->x[i] + (obj->winv[0] * du + obj->winv[1] * dv);
            ph->y[i] = ph->y[i] + (obj->winv[2] * du + obj->winv[3] * dv);
        }
        break; }
    case IMS_OP_FOCUS_DEPTH:
        for (int64_t i = 0; i < n; ++i) {
            const ims_object_t* obj; (void)stream_index(P, ph, photon_offset, i, &obj);
            if (obj->flags & IMS_OBJ_FAINT) continue;
            ph->x[i] = ph->x[i] + ph->dxdz[i] * op->p[0];
            ph->y[i] = ph->y[i] + ph->dydz[i] * op->p[0];
        }
        break;
    case IMS_OP_REFRACTION: {
        double nn = op->p[0] * op->p[0];
        for (int64_t i = 0; i < n; ++i) {
            const ims_object_t* obj; (void)stream_index(P, ph, photon_offset, i, &obj);
            if (obj->flags & IMS_OBJ_FAINT) continue;
            double a = ph->dxdz[i], b = ph->dydz[i];
            double rho2 = a * a + b * b;
            double f = 1.0 / orc_sqrt(nn + (nn - 1.0) * rho2);
            ph->dxdz[i] = a * f;
            ph->dydz[i] = b * f;
        }
        break; }
    case IMS_OP_BANDPASS_RATIO:
        for (int64_t i = 0; i < n; ++i)
            ph->flux[i] = ph->flux[i] * orc_lin_lookup(&P->ratio, op->table, ph->wavelength[i]);
        break;
    case IMS_OP_RUBIN_OPTICS:
    case IMS_OP_RUBIN_DIFFRACTION:
    case IMS_OP_RUBIN_DIFFRACTION_OPTICS:
        orc_apply_rubin_op(P, op_index, ph, photon_offset);
        break;
    default:
        break;
    }
}
