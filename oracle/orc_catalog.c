/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * LSST_SiliconBuilder.setup for a whole catalog, restated on the CPU: the checker of ims_build_object_table
 * (include/imsim_hip.h, "object table on the device").  Follows imsim/stamp.py:109-249 (flux realisation :190, skip and
 * tiny-flux rules :199-210, stamp size :212-232), imsim/stamp_utils.py:79-155 (stars: folding threshold rounded down to
 * e-folds), :158-189 (galaxies: GoodImageSize with the DoubleGaussian proxy), imsim/instcat.py:498-527 (Sersic affine: shear
 * from the axis ratio with the flipped position angle, then the lens), and GalSim's wcs.local / dcr.zenith_parallactic_angles
 * in the trig-free vector form the header states.  Parity unpinned at the bit level against GalSim (absent here); the numpy
 * builder imsim_amd/catalog.py, which reproduces the reference's stored stamp sizes, is the value-level check
 * (tests/test_device_table.py). */
#include <math.h>
#include <string.h>
#include "orc.h"
#include "orc_math.h"

double orc_poisson(double mean, uint64_t seed, int64_t obj_id, int64_t pixel);
void orc_sip_value_grad(const double* a, double u, double v, double* f, double* fu, double* fv);
void orc_wcs_vec_to_pix(const ims_tansip_t* w, const double p[3], double* x, double* y);

static int good_image_size(double stepk, double pixel_scale)
{
    double nn = ceil(6.283185307179586476925286766559 / (stepk * pixel_scale));
    long long n = (long long)nn;
    return (int)(2 * ((n + 1) / 2));
}

static void stamp_bounds(ims_object_t* o, int size)
{
    long long icx = (long long)floor(o->x0 + 0.5), icy = (long long)floor(o->y0 + 0.5);
    o->stamp_xmin = (int)(icx - size / 2); o->stamp_xmax = (int)(icx - size / 2 + size - 1);
    o->stamp_ymin = (int)(icy - size / 2); o->stamp_ymax = (int)(icy - size / 2 + size - 1);
}

/* get_good_phot_stamp_size1 (imsim/stamp_utils.py:293-354) for a transformed Sersic profile: the square grown by 10 % until the
 * profile's xValue on its edge midpoints and corners is below `keep`, capped at nmax, shrunk again while the next smaller one
 * still is (not below 64) */
static double sersic_edge_max(double h, const double* j, double det, double hlr, double amp, double b, double inv_n)
{
    const double px[4] = { h, 0.0, h, h }, py[4] = { 0.0, h, h, -h };        /* point symmetry: four of the eight points */
    double best = 0.0;
    for (int k = 0; k < 4; ++k) {
        double u = (j[3] * px[k] - j[1] * py[k]) / det, v = (-j[2] * px[k] + j[0] * py[k]) / det;
        double r = orc_sqrt(u * u + v * v) / hlr;
        double val = amp * orc_exp(-b * orc_pow(r, inv_n));
        if (val > best) best = val;
    }
    return best;
}

static long long phot_stamp_size1(int own, double keep, int nmax, double pixel_scale, const double* j, double hlr, double flux,
                                  double norm, double b, double inv_n)
{
    double det = j[0] * j[3] - j[1] * j[2];
    double amp = flux * norm / (hlr * hlr * fabs(det));
    double N = (double)own;
    int active = N < (double)nmax;
    for (int it = 0; it < 200 && active; ++it) {
        double mv = sersic_edge_max(N * 0.5 * pixel_scale, j, det, hlr, amp, b, inv_n);
        if (mv < keep) break;
        N = N * 1.1;
        active = N < (double)nmax;
    }
    if (N > (double)nmax) N = (double)nmax;
    active = N >= 64.0 * 1.1;
    for (int it = 0; it < 200 && active; ++it) {
        double mv = sersic_edge_max(N / (2.0 * 1.1) * pixel_scale, j, det, hlr, amp, b, inv_n);
        if (mv > keep) break;
        N = N / 1.1;
        active = N >= 64.0 * 1.1;
    }
    return (long long)N;
}

int orc_build_object_table(const ims_catalog_t* C, const ims_optics_t* optics, ims_object_t* rows, ims_object_meta_t* meta)
{
    const double pi = 3.14159265358979323846;
    for (int64_t i = 0; i < C->n; ++i) {
        ims_object_t o;
        ims_object_meta_t m = { 0, 0, 0 };
        memset(&o, 0, sizeof(o));
        int kind = C->kind[i];
        double nominal = C->nominal_flux[i];
        int64_t id = C->obj_id ? C->obj_id[i] : i;
        int64_t phot = C->phot_flux ? C->phot_flux[i] : (int64_t)orc_poisson(nominal, C->seed, id, (int64_t)IMS_FLUX_PIXEL);
        if (kind < 0 || kind > 2) { m.flags = IMS_META_HOST_ROW; m.n_phot = phot; rows[i] = o; meta[i] = m; continue; }
        double x = C->x[i], y = C->y[i];
        o.obj_id = id; o.n_phot = phot; o.x0 = x; o.y0 = y; o.flux_per_photon = 1.0;
        double j0 = 1.0, j1 = 0.0, j2 = 0.0, j3 = 1.0;
        if (kind == 0) { o.prof_table = IMS_PROF_POINT; o.prof_scale = 0.0; }
        else {
            o.prof_table = C->prof_table[i]; o.prof_scale = C->hlr[i];
            double q = C->q[i], g = (1.0 - q) / (1.0 + q), s2, c2;
            orc_sincos(2.0 * ((90.0 - C->pa[i]) * 0.017453292519943295), &s2, &c2);
            double f = 1.0 / orc_sqrt(1.0 - g * g), sg1 = g * c2, sg2 = g * s2;
            j0 = f * (1.0 + sg1); j1 = f * sg2; j2 = f * sg2; j3 = f * (1.0 - sg1);
            if (C->g1) {
                double l1 = C->g1[i], l2 = C->g2[i], lg2 = l1 * l1 + l2 * l2;
                double lf = orc_sqrt(C->mu[i]) / orc_sqrt(1.0 - lg2);
                double a0 = lf * (1.0 + l1), a1 = lf * l2, a2 = lf * l2, a3 = lf * (1.0 - l1);
                double b0 = j0, b1 = j1, b2 = j2, b3 = j3;
                j0 = a0 * b0 + a1 * b2; j1 = a0 * b1 + a1 * b3; j2 = a2 * b0 + a3 * b2; j3 = a2 * b1 + a3 * b3;
            }
        }
        o.jac[0] = j0; o.jac[1] = j1; o.jac[2] = j2; o.jac[3] = j3;
        const ims_tansip_t* w = &optics->img_wcs;
        double p[3], px[3], py[3];
        {
            double u = x - w->crpix[0], v = y - w->crpix[1], fu = 0.0, fv = 0.0, gu = 0.0, gv = 0.0;
            if (w->order > 0) {
                double f, g;
                orc_sip_value_grad(w->a, u, v, &f, &fu, &fv);
                orc_sip_value_grad(w->b, u, v, &g, &gu, &gv);
                u = u + f; v = v + g;
            }
            double Ux = 1.0 + fu, Uy = fv, Vx = gu, Vy = 1.0 + gv;
            double xi = w->cd[0] * u + w->cd[1] * v, eta = w->cd[2] * u + w->cd[3] * v;
            double xi_x = w->cd[0] * Ux + w->cd[1] * Vx, xi_y = w->cd[0] * Uy + w->cd[1] * Vy;
            double et_x = w->cd[2] * Ux + w->cd[3] * Vx, et_y = w->cd[2] * Uy + w->cd[3] * Vy;
            for (int k = 0; k < 3; ++k) {
                p[k] = w->rot[k] + w->rot[3 + k] * xi + w->rot[6 + k] * eta;
                px[k] = w->rot[3 + k] * xi_x + w->rot[6 + k] * et_x;
                py[k] = w->rot[3 + k] * xi_y + w->rot[6 + k] * et_y;
            }
        }
        double inv = 1.0 / orc_sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
        p[0] = p[0] * inv; p[1] = p[1] * inv; p[2] = p[2] * inv;
        double cd = orc_sqrt(p[0] * p[0] + p[1] * p[1]), icd = 1.0 / cd;
        double e0 = -p[1] * icd, e1 = p[0] * icd;
        double n0 = -p[2] * e1, n1 = p[2] * e0, n2 = p[0] * e1 - p[1] * e0;
        double k = 206264.80624709636 * inv;
        double dudx = -(px[0] * e0 + px[1] * e1) * k, dudy = -(py[0] * e0 + py[1] * e1) * k;
        double dvdx = (px[0] * n0 + px[1] * n1 + px[2] * n2) * k, dvdy = (py[0] * n0 + py[1] * n1 + py[2] * n2) * k;
        double idet = 1.0 / (dudx * dvdy - dudy * dvdx);
        o.winv[0] = dvdy * idet; o.winv[1] = -dudy * idet; o.winv[2] = -dvdx * idet; o.winv[3] = dudx * idet;
        double cz = p[0] * C->zenith[0] + p[1] * C->zenith[1] + p[2] * C->zenith[2];
        double ez = e0 * C->zenith[0] + e1 * C->zenith[1];
        double nz = n0 * C->zenith[0] + n1 * C->zenith[1] + n2 * C->zenith[2];
        double hz = orc_sqrt(ez * ez + nz * nz);
        o.dcr_tanz = hz / cz;
        o.dcr_sinp = hz > 0.0 ? ez / hz : 0.0;
        o.dcr_cosp = hz > 0.0 ? nz / hz : 1.0;
        if (C->has_field) orc_wcs_vec_to_pix(&optics->icrf_to_field, p, &o.atm_tan_x, &o.atm_tan_y);
        o.sed_table = C->sed_table ? C->sed_table[i] : C->sed_table_all;
        o.flags = nominal < C->max_flux_simple ? IMS_OBJ_FAINT : 0;
        int size = C->stamp_size ? C->stamp_size[i] : 0;
        if (size <= 0) {
            if (nominal < C->tiny_flux) size = 32;
            else if (kind == 0) {
                double ft = C->noise_var / nominal;
                int kk = 0;
                if (ft < 5.0e-3 && ft != 0.0) kk = (int)(-floor(orc_log(ft)));
                if (kk >= C->n_star_size) kk = C->n_star_size - 1;
                size = C->star_size[kk];
            } else {
                double s1 = j0 * j0 + j1 * j1 + j2 * j2 + j3 * j3;
                double dd = j0 * j0 + j1 * j1 - j2 * j2 - j3 * j3, od = j0 * j2 + j1 * j3;
                double s2 = orc_sqrt(fmax(dd * dd + 4.0 * od * od, 0.0));
                double smax = orc_sqrt(0.5 * (s1 + s2));
                int t = o.prof_table;
                if (t < 0) t = 0;
                if (t >= C->n_gal_radius) t = C->n_gal_radius - 1;
                double rr = C->gal_radius[t] * C->hlr[i] * smax;
                double stepk = 1.0 / orc_sqrt(rr * rr / (pi * pi) + 1.0 / (C->dg_stepk * C->dg_stepk));
                size = good_image_size(stepk, C->pixel_scale);
                if (nominal > 10.0 * (double)size * (double)size || size > C->nmax) {
                    if (C->sb_tables) {
                        const double jj[4] = { j0, j1, j2, j3 };
                        int own = good_image_size(pi / rr, C->pixel_scale);
                        double flux = C->sb_flux ? C->sb_flux[i] : nominal;
                        long long g1 = phot_stamp_size1(own, C->keep_sb, C->nmax, C->pixel_scale, jj, C->hlr[i], flux, C->sersic_norm[t],
                                                        C->sersic_b[t], C->sersic_inv_n[t]);
                        long long sz = (long long)orc_sqrt((double)g1 * (double)g1 + (double)C->psf_size_keep * (double)C->psf_size_keep);
                        if (sz > C->nmax) {
                            long long g3 = phot_stamp_size1(own, 3.0 * C->keep_sb, C->nmax, C->pixel_scale, jj, C->hlr[i], flux,
                                                            C->sersic_norm[t], C->sersic_b[t], C->sersic_inv_n[t]);
                            sz = (long long)orc_sqrt((double)g3 * (double)g3 + (double)C->psf_size_keep3 * (double)C->psf_size_keep3);
                        }
                        size = (int)(sz > C->nmax ? C->nmax : sz);
                    } else {
                        m.flags |= IMS_META_SIZE_PENDING;
                    }
                }
                if (size > C->nmax) size = C->nmax;
            }
        }
        stamp_bounds(&o, size);
        m.n_phot = phot; m.size = size;
        rows[i] = o; meta[i] = m;
    }
    return 0;
}
