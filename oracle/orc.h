/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h).  Shared declarations of the CPU oracle.
 * The oracle reads the same POD structs as the C-ABI (include/imsim_hip.h) but with HOST pointers.
 */
#ifndef ORC_H
#define ORC_H
#include "../include/imsim_hip.h"
#include "orc_math.h"

double orc_lin_lookup(const ims_lin_tables_t* t, int table, double arg);
double orc_radial_r2(const ims_radial_tables_t* t, int table, double u);
int    orc_photons_alloc(ims_photons_t* p, int64_t n);
void   orc_photons_free(ims_photons_t* p);
void   orc_shoot_object(const ims_render_params_t* P, const ims_object_t* obj, int32_t obj_index,
                        ims_photons_t* ph, int64_t base);
void   orc_apply_psf(const ims_render_params_t* P, const ims_object_t* obj, int comp,
                     ims_photons_t* ph, int64_t base);
void   orc_shift_to_image(const ims_object_t* obj, ims_photons_t* ph, int64_t base);
void   orc_apply_op(const ims_render_params_t* P, int op_index, ims_photons_t* ph,
                    const int64_t* photon_offset);
void   orc_apply_rubin_op(const ims_render_params_t* P, int op_index, ims_photons_t* ph,
                          const int64_t* photon_offset);
void   orc_screen_gradient(const ims_atmosphere_t* A, double pu, double pv, double t, double tanx, double tany,
                           double* gx, double* gy);
double orc_air_n_minus_one(double wave_nm, double air_p, double air_w);
void   orc_air_factors(double p_kpa, double t_k, double h2o_kpa, double* air_p, double* air_w);
int    orc_fill_derived_op(ims_op_t* op);
int    orc_fill_derived_medium(int32_t kind, double* c6);

/* sensor */
int    orc_owned_points(const ims_sensor_t* s);        /* 2*num_vertices + 1 */
void   orc_sensor_init_boundaries(const ims_sensor_t* s, int first_slot, int n_slots);
void   orc_sensor_update_distortions(const ims_sensor_t* s, int first_slot, int n_slots);
/* accumulate photons [i0,i1) of the pool into `image` (double accumulation buffer) */
void   orc_accumulate_range(const ims_render_params_t* P, const ims_photons_t* ph,
                            const int64_t* photon_offset, int64_t i0, int64_t i1,
                            double* image, double* realized_flux, int32_t* pixel_index_out);

/* drivers */
int    orc_render_objects(const ims_render_params_t* P, int64_t nrecalc, double* image_out,
                          double* realized_flux);
int    orc_shoot_pool(const ims_render_params_t* P, const int64_t* photon_offset, ims_photons_t* pool);
#endif
