/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h).
 *
 * orc_render.c: the draw loop itself, in the reference's order.
 *   LSST_Image mode  (imsim/lsst_image.py:342-368 + imsim/stamp.py:527-573): for each object:
 *     photons in chunks of maxN = 1e6 (stamp.py:478); within the Silicon sensor the pixel
 *     boundaries are re-superposed every `nrecalc` electrons of that object's own charge.
 *   Pooling mode (imsim/photon_pooling.py:141-160): orc_shoot_pool + orc_apply_op per op +
 *     orc_accumulate_range, driven by the test/bench host code.
 */
#include <stdlib.h>
#include "orc.h"

static void run_chunk(const ims_render_params_t* P, const ims_object_t* obj, int64_t first, int64_t count,
                      double* image, double* realized)
{
    ims_object_t o = *obj;
    o.phot_first = first; o.n_phot = count;
    ims_render_params_t Q = *P;
    Q.objects = &o; Q.n_objects = 1;
    ims_photons_t ph;
    if (orc_photons_alloc(&ph, count) != 0) return;
    int64_t off[2] = {0, count};
    orc_shoot_object(&Q, &o, 0, &ph, 0);
    for (int c = 0; c < Q.n_psf; ++c) orc_apply_psf(&Q, &o, c, &ph, 0);
    orc_shift_to_image(&o, &ph, 0);
    for (int k = 0; k < Q.n_ops; ++k) orc_apply_op(&Q, k, &ph, off);
    orc_accumulate_range(&Q, &ph, off, 0, count, image, realized, NULL);
    orc_photons_free(&ph);
}

int orc_render_objects(const ims_render_params_t* P, int64_t nrecalc, double* image_out, double* realized_flux)
{
    const int64_t npix = (int64_t)P->nx * P->ny;
    double* img = calloc((size_t)npix, sizeof(double));
    if (!img) return -1;
    const ims_sensor_t* s = P->sensor;
    const int silicon = (s != NULL && s->kind == IMS_SENSOR_SILICON);
    const int64_t maxN = 1000000;
    for (int64_t oi = 0; oi < P->n_objects; ++oi) {
        const ims_object_t* obj = &P->objects[oi];
        double realized = 0.0;
        int bf = silicon && !(obj->flags & IMS_OBJ_FAINT) && obj->bf_state > 0 && nrecalc > 0;
        int64_t chunk = bf ? nrecalc : maxN;
        int64_t done = 0;
        while (done < obj->n_phot) {
            int64_t c = obj->n_phot - done;
            if (c > chunk) c = chunk;
            run_chunk(P, obj, obj->phot_first + done, c, img, &realized);
            done += c;
            if (bf && done < obj->n_phot) orc_sensor_update_distortions(s, obj->bf_state, 1);
        }
        if (realized_flux) realized_flux[oi] += realized;
    }
    for (int64_t i = 0; i < npix; ++i) image_out[i] += img[i];
    free(img);
    return 0;
}

/* LSST_PhotonsBuilder.draw for every object of a sub-batch + merge_photon_arrays
 * (imsim/stamp.py:708-743, imsim/photon_pooling.py:177-192) */
int orc_shoot_pool(const ims_render_params_t* P, const int64_t* photon_offset, ims_photons_t* pool)
{
    for (int64_t oi = 0; oi < P->n_objects; ++oi) {
        const ims_object_t* obj = &P->objects[oi];
        int64_t base = photon_offset[oi];
        orc_shoot_object(P, obj, (int32_t)oi, pool, base);
        for (int c = 0; c < P->n_psf; ++c) orc_apply_psf(P, obj, c, pool, base);
        orc_shift_to_image(obj, pool, base);
    }
    return 0;
}

/* test hooks for the elementary functions */
void orc_test_math(int which, const double* in, double* out, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        switch (which) {
        case 0: out[i] = orc_log(in[i]); break;
        case 1: out[i] = orc_exp(in[i]); break;
        case 2: { double s, c; orc_sincos2pi(in[i], &s, &c); out[2 * i] = s; out[2 * i + 1] = c; break; }
        case 3: out[i] = orc_atan(in[i]); break;
        case 4: { double s, c; orc_sincos(in[i], &s, &c); out[2 * i] = s; out[2 * i + 1] = c; break; }
        case 5: out[i] = orc_tanh_pos(in[i]); break;
        case 10: { double s, c; orc_sincos2pi_w((uint32_t)in[i], &s, &c); out[2 * i] = s; out[2 * i + 1] = c; break; }
        case 11: out[i] = orc_log_w((uint32_t)in[i]); break;
        case 12: out[i] = orc_gauss_word_cos((uint32_t)in[2 * i], (uint32_t)in[2 * i + 1]); break;
        }
    }
}
void orc_test_philox(uint32_t c[4], uint32_t k0, uint32_t k1) { orc_philox4x32_10(c, k0, k1); }
void orc_test_draw(uint64_t seed, int64_t obj, int64_t photon, uint32_t slot, uint64_t out[2])
{
    orc_draw_t d = orc_draw(seed, obj, photon, slot); out[0] = d.a; out[1] = d.b;
}
void orc_test_gauss(uint64_t seed, int64_t obj, int64_t first, int64_t n, uint32_t slot, double* out)
{
    for (int64_t i = 0; i < n; ++i) {
        orc_words_t d = orc_words(seed, obj, first + i, slot);
        orc_gauss_words(d.w[0], d.w[1], &out[2 * i], &out[2 * i + 1]);
    }
}
int orc_struct_size(int which)
{
    switch (which) {
    case 0: return (int)sizeof(ims_object_t);
    case 1: return (int)sizeof(ims_radial_tables_t);
    case 2: return (int)sizeof(ims_lin_tables_t);
    case 3: return (int)sizeof(ims_psf_component_t);
    case 4: return (int)sizeof(ims_op_t);
    case 5: return (int)sizeof(ims_surface_t);
    case 6: return (int)sizeof(ims_tansip_t);
    case 7: return (int)sizeof(ims_optics_t);
    case 8: return (int)sizeof(ims_bf_slot_t);
    case 9: return (int)sizeof(ims_sensor_t);
    case 10: return (int)sizeof(ims_photons_t);
    case 11: return (int)sizeof(ims_render_params_t);
    case 12: return (int)sizeof(ims_plan_item_t);
    case 13: return (int)sizeof(ims_atmosphere_t);
    case 14: return (int)sizeof(ims_fft_object_t);
    case 15: return (int)sizeof(ims_fft_params_t);
    case 16: return (int)sizeof(ims_readout_t);
    case 17: return (int)sizeof(ims_chain_t);
    case 18: return (int)sizeof(ims_catalog_t);
    case 19: return (int)sizeof(ims_object_meta_t);
    case 20: return (int)sizeof(ims_plan_input_t);
    case 21: return (int)sizeof(ims_plan_sizes_t);
    case 22: return (int)sizeof(ims_tuning_t);
    }
    return -1;
}
