/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h).
 *
 * orc_silicon.c: CPU restatement of galsim.SiliconSensor.accumulate as imSim configures and calls
 * it (config/imsim-config.yaml:230-235, imsim/lsst_image.py:93-103, imsim/stamp.py:558-569,
 * imsim/photon_pooling.py:195-225).  GalSim (unpinned, setup.py:21) is absent from
 * /root/reference; the algorithm below follows the published description
 * (doc/validation/brighter-fatter.rst:29-60, doc/validation/diffusion.rst:66-99,
 * doc/validation/tree-ring.rst:105-113; Lage, Bradshaw & Tyson 2017) and SURVEY.md Appendix A.
 * Pinned statistically by tests/test_sensor_models.py and tests/test_flats.py criteria of the
 * reference (SURVEY.md 8c i, vii).  Bit level: parity unpinned.
 *
 * Pixel-boundary storage: every owner cell (i,j) of a region owns its bottom row (lower-left
 * corner, the num_vertices interior points of the bottom edge left to right, lower-right corner)
 * and the num_vertices interior points of its left edge (bottom to top), in pixel-local
 * coordinates (undistorted lower-left corner = (0,0)).  A pixel's top row is the bottom row of the
 * cell above and its right edge the left edge of the cell to the right, so the two copies of a
 * corner that the Poisson model tabulates for horizontally adjacent pixels are both kept, as in
 * GalSim's horizontal/vertical boundary arrays.  A region of nx*ny pixels has (nx+1)*(ny+1) cells.
 */
#include <stdlib.h>
#include "orc.h"

int orc_owned_points(const ims_sensor_t* s) { return 2 * s->num_vertices + 2; }

static inline int64_t cell_index(const ims_bf_slot_t* sl, int i, int j)
{
    return sl->offset + (int64_t)j * (sl->nx + 1) + i;
}

/* undistorted position of owned point n */
static void empty_owned(const ims_sensor_t* s, int n, double* x, double* y)
{
    int nV = s->num_vertices;
    if (n == 0) { *x = 0.0; *y = 0.0; return; }
    if (n <= nV) { *x = s->emptypoly[2 * n]; *y = 0.0; return; }           /* bottom point m = n-1 */
    if (n == nV + 1) { *x = 1.0; *y = 0.0; return; }                        /* lower-right corner */
    int m = n - nV - 2;                                                     /* left point, bottom->top */
    *x = 0.0; *y = s->emptypoly[2 * (1 + m)];                               /* same abscissa set as the bottom edge */
}

static double treering_shift(const ims_sensor_t* s, double r)
{
    if (s->n_tr <= 0) return 0.0;
    double f = r / s->tr_dr;
    if (!(f > 0.0) || f >= (double)(s->n_tr - 1)) return 0.0;
    int i = (int)f;
    double b = f - (double)i;
    if (s->tr_table2 == NULL) return s->tr_table[i] + b * (s->tr_table[i + 1] - s->tr_table[i]);
    /* natural cubic spline (the LookupTable 'spline' interpolant) */
    double a = 1.0 - b;
    double h2 = s->tr_dr * s->tr_dr / 6.0;
    return a * s->tr_table[i] + b * s->tr_table[i + 1]
         + ((a * a * a - a) * s->tr_table2[i] + (b * b * b - b) * s->tr_table2[i + 1]) * h2;
}

/* assemble the nv-vertex polygon of pixel (i,j) of a slot, scaled by zfactor towards the
 * undistorted polygon (Silicon::insidePixel, recalled) */
static void assemble_polygon(const ims_sensor_t* s, const ims_bf_slot_t* sl, int i, int j,
                             double zfactor, double* vx, double* vy)
{
    const int nV = s->num_vertices, npo = 2 * nV + 2;
    const double* own = s->bf_boundary + cell_index(sl, i, j) * npo * 2;
    const double* rgt = s->bf_boundary + cell_index(sl, i + 1, j) * npo * 2;
    const double* up  = s->bf_boundary + cell_index(sl, i, j + 1) * npo * 2;
    int n = 0;
    for (int m = 0; m <= nV + 1; ++m, ++n) { vx[n] = own[2 * m]; vy[n] = own[2 * m + 1]; }              /* LL, bottom, LR */
    for (int m = 0; m < nV; ++m, ++n) { vx[n] = rgt[2 * (nV + 2 + m)] + 1.0; vy[n] = rgt[2 * (nV + 2 + m) + 1]; }
    for (int m = 0; m <= nV + 1; ++m, ++n) { int q = nV + 1 - m; vx[n] = up[2 * q]; vy[n] = up[2 * q + 1] + 1.0; }  /* UR, top, UL */
    for (int m = 0; m < nV; ++m, ++n) { int q = nV + 2 + (nV - 1 - m); vx[n] = own[2 * q]; vy[n] = own[2 * q + 1]; }
    if (zfactor != 1.0) {
        const int nv = 4 * nV + 4;
        for (int k = 0; k < nv; ++k) {
            double ex = s->emptypoly[2 * k], ey = s->emptypoly[2 * k + 1];
            vx[k] = ex + (vx[k] - ex) * zfactor;
            vy[k] = ey + (vy[k] - ey) * zfactor;
        }
    }
}

/* ---------- LSST_Flat, area branch (imsim/flat.py:209-236) ---------- */
#define ORC_FLAT_ID_BASE 0x7F00000000ll
double orc_poisson(double mean, uint64_t seed, int64_t obj_id, int64_t pixel);

/* Silicon::fillWithPixelAreas: shoelace area of every pixel polygon of a slot; the exact sum of the
 * areas quantised to 2^-32 goes to *sum_q32 */
void orc_sensor_pixel_areas(const ims_sensor_t* s, int slot, double* area, long long* sum_q32)
{
    const ims_bf_slot_t* sl = &s->bf_slots[slot];
    const int nv = 4 * s->num_vertices + 4;
    double vx[4 * 32 + 4], vy[4 * 32 + 4];
    for (int j = 0; j < sl->ny; ++j)
        for (int i = 0; i < sl->nx; ++i) {
            assemble_polygon(s, sl, i, j, 1.0, vx, vy);
            double a2 = 0.0;
            for (int k = 1; k <= nv; ++k) {
                int kk = (k < nv) ? k : 0;
                a2 = a2 + (vx[k - 1] * vy[kk] - vx[kk] * vy[k - 1]);
            }
            double a = 0.5 * a2;
            area[(int64_t)j * sl->nx + i] = a;
            *sum_q32 += (long long)floor(a * 0x1.0p32 + 0.5);
        }
}

void orc_flat_add(const double* area, const double* base, double level, double inv_mean_area, uint64_t seed,
                  int64_t iteration, int32_t nx, int32_t ny, double* image, double* delta)
{
    for (int64_t p = 0; p < (int64_t)nx * ny; ++p) {
        double mean = level;
        if (base) mean = mean * base[p];
        if (area) mean = mean * (area[p] * inv_mean_area);
        double v = orc_poisson(mean, seed, ORC_FLAT_ID_BASE + iteration, p);
        image[p] = image[p] + v;
        if (delta) {
            int i = (int)(p % nx), j = (int)(p / nx);
            delta[(int64_t)j * (nx + 1) + i] = delta[(int64_t)j * (nx + 1) + i] + v;
        }
    }
}

static void refresh_bounds(const ims_sensor_t* s, const ims_bf_slot_t* sl, int i, int j)
{
    const int nV = s->num_vertices, nv = 4 * nV + 4;
    double vx[4 * 32 + 4], vy[4 * 32 + 4];
    assemble_polygon(s, sl, i, j, 1.0, vx, vy);
    double ixmin = 0.0, ixmax = 1.0, iymin = 0.0, iymax = 1.0;
    double oxmin = 0.0, oxmax = 1.0, oymin = 0.0, oymax = 1.0;
    for (int k = 0; k < nv; ++k) {
        if (vx[k] < oxmin) oxmin = vx[k];
        if (vx[k] > oxmax) oxmax = vx[k];
        if (vy[k] < oymin) oymin = vy[k];
        if (vy[k] > oymax) oymax = vy[k];
    }
    for (int k = 0; k <= nV + 1; ++k) if (vy[k] > iymin) iymin = vy[k];                 /* bottom side */
    for (int k = nV + 1; k <= 2 * nV + 2; ++k) if (vx[k] < ixmax) ixmax = vx[k];         /* right side */
    for (int k = 2 * nV + 2; k <= 3 * nV + 3; ++k) if (vy[k] < iymax) iymax = vy[k];     /* top side */
    for (int k = 3 * nV + 3; k < nv; ++k) if (vx[k] > ixmin) ixmin = vx[k];              /* left side */
    if (vx[0] > ixmin) ixmin = vx[0];
    double* b = s->bf_bounds + cell_index(sl, i, j) * 8;
    b[0] = ixmin; b[1] = ixmax; b[2] = iymin; b[3] = iymax;
    b[4] = oxmin; b[5] = oxmax; b[6] = oymin; b[7] = oymax;
}

/* Silicon::initialize + addTreeRingDistortions: undistorted boundaries shifted radially by the
 * tree-ring function f(r) about the tree-ring centre (imsim/treerings.py:169-195). */
void orc_sensor_init_boundaries(const ims_sensor_t* s, int first_slot, int n_slots)
{
    const int npo = orc_owned_points(s);
    for (int k = first_slot; k < first_slot + n_slots; ++k) {
        const ims_bf_slot_t* sl = &s->bf_slots[k];
        for (int j = 0; j <= sl->ny; ++j)
            for (int i = 0; i <= sl->nx; ++i) {
                double* pts = s->bf_boundary + cell_index(sl, i, j) * npo * 2;
                for (int n = 0; n < npo; ++n) {
                    double ex, ey;
                    empty_owned(s, n, &ex, &ey);
                    double tx = ((double)(sl->xmin + i) - 0.5 + ex) - s->tr_cx;
                    double ty = ((double)(sl->ymin + j) - 0.5 + ey) - s->tr_cy;
                    double r = orc_sqrt(tx * tx + ty * ty);
                    double sh = treering_shift(s, r);
                    double px = ex, py = ey;
                    if (r > 0.0 && sh != 0.0) { px = ex + sh * tx / r; py = ey + sh * ty / r; }
                    pts[2 * n] = px; pts[2 * n + 1] = py;
                }
                s->bf_delta[cell_index(sl, i, j)] = 0.0;
            }
        for (int j = 0; j < sl->ny; ++j)
            for (int i = 0; i < sl->nx; ++i) refresh_bounds(s, sl, i, j);
    }
}

/* index of owned point n of a model pixel in the nv-vertex distortion table */
static int owned_to_vertex(const ims_sensor_t* s, int n)
{
    int nV = s->num_vertices;
    if (n <= nV + 1) return n;                   /* LL corner, bottom points, LR corner */
    int m = n - nV - 2;                          /* left point, bottom->top */
    return 3 * nV + 4 + (nV - 1 - m);            /* left edge is stored top->bottom in the polygon */
}

/* Silicon::updatePixelDistortions: linear superposition of the tabulated vertex displacements,
 * scaled by charge / num_elec, over the qdist neighbourhood; uses (and then clears) the delta
 * charge accumulated since the previous update. */
void orc_sensor_update_distortions(const ims_sensor_t* s, int first_slot, int n_slots)
{
    const int nV = s->num_vertices, npo = 2 * nV + 2, nv = 4 * nV + 4, q = s->qdist;
    const int cx = (s->nx - 1) / 2, cy = (s->ny - 1) / 2;
    for (int k = first_slot; k < first_slot + n_slots; ++k) {
        const ims_bf_slot_t* sl = &s->bf_slots[k];
        for (int j = 0; j <= sl->ny; ++j)
            for (int i = 0; i <= sl->nx; ++i) {
                double* pts = s->bf_boundary + cell_index(sl, i, j) * npo * 2;
                /* (di,dj) = owner cell minus charged pixel; one extra row/column on the shared side */
                for (int dj = -q; dj <= q + 1; ++dj) {
                    int sj = j - dj;
                    if (sj < 0 || sj >= sl->ny) continue;
                    for (int di = -q; di <= q + 1; ++di) {
                        int si = i - di;
                        if (si < 0 || si >= sl->nx) continue;
                        double charge = s->bf_delta[cell_index(sl, si, sj)];
                        if (charge == 0.0) continue;
                        double w = charge / s->num_elec;
                        const double* dist = s->distortions + ((int64_t)(di + cx) * s->ny + (dj + cy)) * nv * 2;
                        for (int n = 0; n < npo; ++n) {
                            /* the bottom row (shared with the pixel below) takes the extra row but not the
                               extra column; the left edge (shared with the pixel to the left) the reverse */
                            if (n <= nV + 1 && di == q + 1) continue;
                            if (n > nV + 1 && dj == q + 1) continue;
                            int vtx = owned_to_vertex(s, n);
                            pts[2 * n] = orc_fma(dist[2 * vtx], w, pts[2 * n]);
                            pts[2 * n + 1] = orc_fma(dist[2 * vtx + 1], w, pts[2 * n + 1]);
                        }
                    }
                }
            }
        for (int j = 0; j <= sl->ny; ++j)
            for (int i = 0; i <= sl->nx; ++i) s->bf_delta[cell_index(sl, i, j)] = 0.0;
        for (int j = 0; j < sl->ny; ++j)
            for (int i = 0; i < sl->nx; ++i) refresh_bounds(s, sl, i, j);
    }
}

/* Silicon::insidePixel (recalled): is the point (x,y) (pixel-local) inside pixel (ix,iy)? */
static int inside_pixel(const ims_sensor_t* s, const ims_bf_slot_t* sl, int ix, int iy,
                        double x, double y, double zconv, int* off_edge)
{
    int i = ix - sl->xmin, j = iy - sl->ymin;
    if (i < 0 || i >= sl->nx || j < 0 || j >= sl->ny) {
        if (off_edge) *off_edge = 1;
        return 0;
    }
    const double* b = s->bf_bounds + cell_index(sl, i, j) * 8;
    int inside;
    if (x > b[0] && x < b[1] && y > b[2] && y < b[3]) inside = 1;
    else if (!(x >= b[4] && x <= b[5] && y >= b[6] && y <= b[7])) inside = 0;
    else {
        const double zfit = 12.0;
        double zfactor = orc_tanh_pos(zconv / zfit);
        const int nv = 4 * s->num_vertices + 4;
        double vx[4 * 32 + 4], vy[4 * 32 + 4];
        assemble_polygon(s, sl, i, j, zfactor, vx, vy);
        inside = 0;
        for (int k = 0, l = nv - 1; k < nv; l = k++) {
            if ((vy[k] > y) != (vy[l] > y)) {
                /* x < (lx - kx) (y - ky) / (ly - ky) + kx, cross-multiplied (ly != ky here): no division */
                double dy = vy[l] - vy[k];
                double lhs = (x - vx[k]) * dy, rhs = (vx[l] - vx[k]) * (y - vy[k]);
                if ((dy > 0.0) ? (lhs < rhs) : (lhs > rhs)) inside = !inside;
            }
        }
    }
    if (!inside && off_edge) {
        *off_edge = 0;
        if (i == 0 && x < b[0]) *off_edge = 1;
        if (i == sl->nx - 1 && x > b[1]) *off_edge = 1;
        if (j == 0 && y < b[2]) *off_edge = 1;
        if (j == sl->ny - 1 && y > b[3]) *off_edge = 1;
    }
    return inside;
}

static const int XOFF[9] = {0, 1, 1, 0, -1, -1, -1, 0, 1};
static const int YOFF[9] = {0, 0, 1, 1, 1, 0, -1, -1, -1};

static int chain_has_angles(const ims_render_params_t* P)
{
    for (int k = 0; k < P->n_ops; ++k)
        if (P->ops[k].kind == IMS_OP_RUBIN_OPTICS || P->ops[k].kind == IMS_OP_RUBIN_DIFFRACTION_OPTICS) return 1;
    return 0;
}

void orc_accumulate_range(const ims_render_params_t* P, const ims_photons_t* ph,
                          const int64_t* photon_offset, int64_t i0, int64_t i1,
                          double* image, double* realized_flux, int32_t* pixel_index_out)
{
    const ims_sensor_t* s = P->sensor;
    const int silicon = (s != NULL && s->kind == IMS_SENSOR_SILICON);
    const int has_angles = chain_has_angles(P);
    for (int64_t i = i0; i < i1; ++i) {
        int32_t oi = ph->obj_index[i];
        const ims_object_t* obj = &P->objects[oi];
        int64_t k = obj->phot_first + (i - photon_offset[oi]);
        if (pixel_index_out) pixel_index_out[i] = -1;
        double flux = ph->flux[i];
        if (flux == 0.0) continue;
        double x0 = ph->x[i], y0 = ph->y[i];
        int ix, iy;
        if (!silicon || (obj->flags & IMS_OBJ_FAINT)) {
            ix = (int)floor(x0 + 0.5); iy = (int)floor(y0 + 0.5);
            if (ix < obj->stamp_xmin || ix > obj->stamp_xmax || iy < obj->stamp_ymin || iy > obj->stamp_ymax) continue;
        } else {
            const ims_bf_slot_t* sl = &s->bf_slots[obj->bf_state];
            orc_words_t dc = orc_words(P->seed, obj->obj_id, k, ORC_SLOT_SENSOR);
            double g0, g1;
            orc_gauss_words(dc.w[0], dc.w[1], &g0, &g1);
            /* conversion depth (Silicon::calculateConversionDepth, recalled) */
            double wl = ph->wavelength[i];
            double f = (wl - s->abs_wl_min) / s->abs_wl_step;
            double abs_len;
            if (!(f > 0.0)) abs_len = s->abs_len[0];
            else if (f >= (double)(s->n_abs - 1)) abs_len = s->abs_len[s->n_abs - 1];
            else { int t = (int)f; double a = f - (double)t; abs_len = s->abs_len[t] + a * (s->abs_len[t + 1] - s->abs_len[t]); }
            double si_length = -abs_len * orc_log_w(dc.w[2]);
            double dz = si_length;
            if (has_angles) {
                double dxdz = ph->dxdz[i], dydz = ph->dydz[i];
                dz = si_length / orc_sqrt(1.0 + dxdz * dxdz + dydz * dydz);
                if (dz > s->thickness - 1.0) dz = s->thickness - 1.0;
                double dzp = dz / s->pixel_size;
                x0 = x0 + dxdz * dzp;
                y0 = y0 + dydz * dzp;
            }
            double zconv = s->thickness - dz;
            if (zconv < 0.0) continue;
            if (s->diff_step != 0.0) {
                double ds = s->diff_step / (s->thickness * s->pixel_size) * orc_sqrt(zconv * s->thickness);
                if (ds < 0.0) ds = 0.0;
                x0 = x0 + ds * g0;
                y0 = y0 + ds * g1;
            }
            ix = (int)floor(x0 + 0.5); iy = (int)floor(y0 + 0.5);
            /* the reference's target is the object's stamp: a nominal pixel off the stamp is lost */
            if (ix < obj->stamp_xmin || ix > obj->stamp_xmax || iy < obj->stamp_ymin || iy > obj->stamp_ymax) continue;
            double x = x0 - (double)ix + 0.5, y = y0 - (double)iy + 0.5;
            int off_edge = 0;
            int found = inside_pixel(s, sl, ix, iy, x, y, zconv, &off_edge);
            if (!found && off_edge) continue;
            int step = 0;
            if (!found) {
                /* Silicon searchNeighbors (recalled): nearest neighbour first, then cycle */
                if ((x > y) && (x > 1.0 - y)) step = 1;
                else if ((x > y) && (x < 1.0 - y)) step = 7;
                else if ((x < y) && (x > 1.0 - y)) step = 3;
                else step = 5;
                int n = step;
                for (int m = 1; m < 9; ++m) {
                    int jx = ix + XOFF[n], jy = iy + YOFF[n];
                    if (inside_pixel(s, sl, jx, jy, x - (double)XOFF[n], y - (double)YOFF[n], zconv, NULL)) {
                        ix = jx; iy = jy; found = 1; break;
                    }
                    n = ((n - 1) + step) % 8 + 1;
                }
            }
            if (!found) {
                int n = (dc.w[3] & 0x80000000u) ? 0 : step;
                ix = ix + XOFF[n]; iy = iy + YOFF[n];
            }
            if (ix < obj->stamp_xmin || ix > obj->stamp_xmax || iy < obj->stamp_ymin || iy > obj->stamp_ymax) continue;
            /* charge bookkeeping for the next distortion update (Silicon's _delta image) */
            if (obj->bf_state > 0 || P->track_static_delta) {
                int di = ix - sl->xmin, dj = iy - sl->ymin;
                if (di >= 0 && di < sl->nx && dj >= 0 && dj < sl->ny)
                    s->bf_delta[cell_index(sl, di, dj)] += flux;
            }
        }
        if (realized_flux) realized_flux[oi] += flux;
        int px = ix - P->xmin, py = iy - P->ymin;
        if (px < 0 || px >= P->nx || py < 0 || py >= P->ny) continue;
        int64_t pidx = (int64_t)py * P->nx + px;
        image[pidx] += flux;
        if (pixel_index_out) pixel_index_out[i] = (int32_t)pidx;
    }
}
