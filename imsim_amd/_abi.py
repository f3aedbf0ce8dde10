"""ctypes mirror of include/imsim_hip.h and loader of libimsim_hip.so.

The product path has no CPU fallback: if the HIP library is missing, or a compute entry point is
called without a GPU, this raises.  Struct layouts are self-checked against ims_struct_size().
"""
import ctypes as C
import os

import numpy as np

from . import tuning

IMS_OBJ_FAINT = 1
IMS_PSF_GAUSSIAN, IMS_PSF_RADIAL, IMS_PSF_SCREENS, IMS_PSF_DOUBLE_GAUSSIAN = 1, 2, 3, 4
IMS_MAX_LAYERS = 8
IMS_MAX_PSF = 4
(IMS_OP_TIME_SAMPLER, IMS_OP_PUPIL_ANNULUS_SAMPLER, IMS_OP_PHOTON_DCR, IMS_OP_RUBIN_OPTICS,
 IMS_OP_RUBIN_DIFFRACTION, IMS_OP_RUBIN_DIFFRACTION_OPTICS, IMS_OP_FOCUS_DEPTH, IMS_OP_REFRACTION,
 IMS_OP_BANDPASS_RATIO) = range(1, 10)
IMS_MAX_OPS = 12
IMS_SENSOR_NONE, IMS_SENSOR_SILICON = 0, 1
IMS_SURF_MIRROR, IMS_SURF_REFRACT, IMS_SURF_DETECTOR, IMS_SURF_BAFFLE = 1, 2, 3, 4
IMS_MEDIUM_CONST, IMS_MEDIUM_SELLMEIER, IMS_MEDIUM_AIR = 0, 1, 2
IMS_PROF_POINT, IMS_PROF_BOX, IMS_PROF_KNOTS, IMS_PROF_IMAGE = -1, -2, -3, -4
(IMS_OBSC_NONE, IMS_OBSC_CLEAR_ANNULUS, IMS_OBSC_CLEAR_CIRCLE, IMS_OBSC_OBSC_CIRCLE,
 IMS_OBSC_OBSC_ANNULUS) = range(5)
IMS_MAX_SURFACES = 24

c_d, c_i32, c_i64, c_u64, c_vp = C.c_double, C.c_int32, C.c_int64, C.c_uint64, C.c_void_p


class Object(C.Structure):
    _fields_ = [("obj_id", c_i64), ("phot_first", c_i64), ("n_phot", c_i64),
                ("x0", c_d), ("y0", c_d), ("flux_per_photon", c_d), ("prof_scale", c_d),
                ("jac", c_d * 4), ("winv", c_d * 4),
                ("dcr_tanz", c_d), ("dcr_sinp", c_d), ("dcr_cosp", c_d),
                ("prof_table", c_i32), ("sed_table", c_i32), ("flags", c_i32),
                ("stamp_xmin", c_i32), ("stamp_xmax", c_i32), ("stamp_ymin", c_i32), ("stamp_ymax", c_i32),
                ("bf_state", c_i32), ("sed_wave", c_d), ("atm_tan_x", c_d), ("atm_tan_y", c_d), ("prof_aux", c_d), ("screen_base", c_i64), ("reserved", c_d * 5)]


# numpy view of the same 256-byte row, for vectorised object-table construction
OBJECT_DTYPE = np.dtype([
    ("obj_id", "<i8"), ("phot_first", "<i8"), ("n_phot", "<i8"),
    ("x0", "<f8"), ("y0", "<f8"), ("flux_per_photon", "<f8"), ("prof_scale", "<f8"),
    ("jac", "<f8", (4,)), ("winv", "<f8", (4,)),
    ("dcr_tanz", "<f8"), ("dcr_sinp", "<f8"), ("dcr_cosp", "<f8"),
    ("prof_table", "<i4"), ("sed_table", "<i4"), ("flags", "<i4"),
    ("stamp_xmin", "<i4"), ("stamp_xmax", "<i4"), ("stamp_ymin", "<i4"), ("stamp_ymax", "<i4"),
    ("bf_state", "<i4"), ("sed_wave", "<f8"), ("atm_tan_x", "<f8"), ("atm_tan_y", "<f8"),
    ("prof_aux", "<f8"), ("screen_base", "<i8"), ("reserved", "<f8", (5,))], align=True)
assert OBJECT_DTYPE.itemsize == 256


class RadialTables(C.Structure):
    _fields_ = [("n_tables", c_i32), ("n_bins", c_i32), ("r2", c_vp), ("cdf", c_vp), ("guide", c_vp), ("n_guide", c_i32),
                ("pad", c_i32)]


class ImageTables(C.Structure):
    _fields_ = [("n_images", c_i32), ("interp", c_i32), ("size", c_vp), ("offset", c_vp), ("cdf", c_vp),
                ("kx", c_vp), ("kcdf", c_vp), ("n_k", c_i32), ("pad", c_i32), ("norm", C.c_double), ("neg", C.c_double * 4)]


class LinTables(C.Structure):
    _fields_ = [("n_tables", c_i32), ("n_pts", c_i32), ("arg_min", c_d), ("arg_step", c_d), ("val", c_vp)]


class PsfComponent(C.Structure):
    _fields_ = [("kind", c_i32), ("table", c_i32), ("p0", c_d), ("chrom_alpha", c_d), ("chrom_base", c_d),
                ("p1", c_d), ("p2", c_d)]


class Atmosphere(C.Structure):
    _fields_ = [("n_layers", c_i32), ("npix", c_i32), ("scale", c_d), ("x0", c_d), ("t0", c_d), ("exptime", c_d),
                ("aper_r_outer", c_d), ("aper_r_inner", c_d), ("vx", c_d * 8), ("vy", c_d * 8), ("alt", c_d * 8),
                ("screens", c_vp),
                ("dn", c_d), ("inv_n", c_d), ("inv_scale", c_d), ("aper_ri2", c_d), ("aper_dr2", c_d),
                ("screen_quads", c_vp)]


class KPsf(C.Structure):
    _fields_ = [("kind", c_i32), ("table", c_i32), ("p0", c_d)]


IMS_KPSF_GAUSSIAN, IMS_KPSF_KOLMOGOROV, IMS_KPSF_TABLE = 1, 2, 3

FFT_OBJECT_DTYPE = np.dtype([
    ("obj_id", "<i8"), ("k_offset", "<i8"), ("r_offset", "<i8"), ("flux", "<f8"), ("cx", "<f8"), ("cy", "<f8"),
    ("prof_scale", "<f8"), ("jac", "<f8", (4,)), ("nfft", "<i4"), ("prof_ktable", "<i4"), ("x0", "<i4"), ("y0", "<i4"),
    ("stamp_xmin", "<i4"), ("stamp_xmax", "<i4"), ("stamp_ymin", "<i4"), ("stamp_ymax", "<i4"), ("pad", "<i4", (2,))],
    align=True)


class FftObject(C.Structure):
    _fields_ = [("obj_id", c_i64), ("k_offset", c_i64), ("r_offset", c_i64), ("flux", c_d), ("cx", c_d), ("cy", c_d),
                ("prof_scale", c_d), ("jac", c_d * 4), ("nfft", c_i32), ("prof_ktable", c_i32), ("x0", c_i32), ("y0", c_i32),
                ("stamp_xmin", c_i32), ("stamp_xmax", c_i32), ("stamp_ymin", c_i32), ("stamp_ymax", c_i32), ("pad", c_i32 * 2)]


class Spikes(C.Structure):
    _fields_ = [("enabled", c_i32), ("cutoff", c_i32), ("threshold", c_d), ("cos0", c_d), ("sin0", c_d), ("a_lo", c_d),
                ("d_alpha", c_d), ("scale", c_d), ("r0", c_d), ("norm", c_d), ("tab_row", c_vp), ("tab_col", c_vp), ("tab_val", c_vp)]


class FftParams(C.Structure):
    _fields_ = [("seed", c_u64), ("pixel_scale", c_d), ("n_kpsf", c_i32), ("add_noise", c_i32),
                ("kpsf", KPsf * IMS_MAX_PSF), ("ktables", LinTables), ("image", c_vp),
                ("nx", c_i32), ("ny", c_i32), ("xmin", c_i32), ("ymin", c_i32), ("realized_flux", c_vp),
                ("spikes", Spikes), ("n_alias", c_i32), ("rbuf_raw", c_i32)]


class Op(C.Structure):
    _fields_ = [("kind", c_i32), ("table", c_i32), ("p", c_d * 8)]


class Surface(C.Structure):
    _fields_ = [("kind", c_i32), ("obsc_kind", c_i32), ("medium_kind", c_i32), ("n_asphere", c_i32), ("medium_id", c_i32), ("pad", c_i32),
                ("z0", c_d), ("R", c_d), ("inv_R", c_d), ("conic", c_d), ("asph", c_d * 4),
                ("obsc_inner", c_d), ("obsc_outer", c_d), ("medium_c", c_d * 6),
                ("k1", c_d), ("k1c", c_d), ("m2R", c_d), ("cc", c_d), ("obsc_i2", c_d), ("obsc_o2", c_d), ("asph_d", c_d * 4)]


class TanSip(C.Structure):
    _fields_ = [("crpix", c_d * 2), ("cd", c_d * 4), ("cdinv", c_d * 4), ("rot", c_d * 9),
                ("order", c_i32), ("pad", c_i32), ("a", c_d * 25), ("b", c_d * 25)]


class Optics(C.Structure):
    _fields_ = [("img_wcs", TanSip), ("icrf_to_field", TanSip),
                ("in_medium_kind", c_i32), ("n_surfaces", c_i32), ("in_medium_c", c_d * 6),
                ("stop_z", c_d), ("surf", Surface * IMS_MAX_SURFACES),
                ("cam_rot", c_d * 2), ("fp_to_pix", c_d * 6), ("slope_jac", c_d * 4),
                ("n_lines", c_i32), ("n_circles", c_i32), ("lines", (c_d * 4) * 8), ("circles", (c_d * 3) * 4),
                ("e_z0", c_d * 3), ("e_focal", c_d * 3), ("cos_lat", c_d), ("sin_lat", c_d), ("omega", c_d),
                ("rot_g", c_d * 3), ("rot_gnorm", c_d)]


class BfSlot(C.Structure):
    _fields_ = [("xmin", c_i32), ("ymin", c_i32), ("nx", c_i32), ("ny", c_i32), ("offset", c_i64)]


BFSLOT_DTYPE = np.dtype([("xmin", "<i4"), ("ymin", "<i4"), ("nx", "<i4"), ("ny", "<i4"), ("offset", "<i8")], align=True)


class Sensor(C.Structure):
    _fields_ = [("kind", c_i32), ("num_vertices", c_i32), ("nx", c_i32), ("ny", c_i32), ("qdist", c_i32),
                ("n_abs", c_i32), ("n_tr", c_i32), ("pad", c_i32),
                ("num_elec", c_d), ("pixel_size", c_d), ("thickness", c_d), ("diff_step", c_d),
                ("abs_wl_min", c_d), ("abs_wl_step", c_d), ("tr_dr", c_d), ("tr_cx", c_d), ("tr_cy", c_d),
                ("abs_len", c_vp), ("tr_table", c_vp), ("tr_table2", c_vp), ("distortions", c_vp), ("emptypoly", c_vp),
                ("n_bf_slots", c_i32), ("pad2", c_i32), ("bf_slots", c_vp),
                ("bf_boundary", c_vp), ("bf_bounds", c_vp), ("bf_delta", c_vp),
                ("bf_tile_charge", c_vp), ("bf_tile_changed", c_vp), ("pristine_margin", c_d),
                ("diff_coef", c_d), ("thick_m1", c_d), ("bf_dl", c_vp)]


class Photons(C.Structure):
    _fields_ = [("n", c_i64)] + [(f, c_vp) for f in
                                 ("x", "y", "flux", "dxdz", "dydz", "wavelength", "pupil_u", "pupil_v", "time", "obj_index")] + [
                                     ("converted", c_i32), ("pad", c_i32)]


class RenderParams(C.Structure):
    _fields_ = [("seed", c_u64), ("objects", c_vp), ("n_objects", c_i64), ("seg_prefix", c_vp),
                ("n_segments", c_i64), ("seg_size", c_i32), ("n_psf", c_i32),
                ("psf", PsfComponent * IMS_MAX_PSF), ("n_ops", c_i32), ("track_static_delta", c_i32),
                ("ops", Op * IMS_MAX_OPS), ("radial", RadialTables), ("sed", LinTables), ("ratio", LinTables),
                ("atm", c_vp), ("optics", c_vp), ("sensor", c_vp), ("image", c_vp),
                ("nx", c_i32), ("ny", c_i32), ("xmin", c_i32), ("ymin", c_i32), ("realized_flux", c_vp),
                ("bf_tag", C.c_uint32), ("bf_slot_shift", C.c_uint32), ("seg_object", c_vp), ("images", ImageTables),
                ("optics_layout", c_u64), ("screen_kick", c_vp),
                ("lazy_static", C.c_uint32), ("margin_cap", C.c_uint32), ("margin_list", c_vp), ("margin_count", c_vp),
                ("margin_wave_count", c_vp), ("margin_waves", c_i64)]


IMS_PLAN_ROUNDS = 9
IMS_MAX_CHAINS = 4
IMS_MAX_CHAIN_EDGES = 8


class Chain(C.Structure):
    _fields_ = [("params", c_vp), ("pool", c_vp), ("pool_start", c_vp), ("n_phot", c_vp), ("tile_prefix", c_vp),
                ("tile_prefix_host", c_vp), ("n_objects", c_i32), ("first_slot", c_i32), ("stream", c_i32), ("nrecalc", c_i32),
                ("n_rounds", c_i32), ("use_tags", c_i32), ("ev_base", c_i32), ("n_edges", c_i32),
                ("edges", c_i32 * IMS_MAX_CHAIN_EDGES)]


class Catalog(C.Structure):
    """ims_catalog_t: the columns of a catalog on the device + the launch-wide constants of the table builder"""
    _fields_ = [("n", c_i64)] + [(f, c_vp) for f in ("x", "y", "nominal_flux", "hlr", "q", "pa", "g1", "g2", "mu", "kind",
                                                      "prof_table", "sed_table", "stamp_size", "obj_id", "phot_flux")] + [
        ("seed", c_u64), ("sed_table_all", c_i32), ("n_star_size", c_i32), ("n_gal_radius", c_i32), ("nmax", c_i32),
        ("noise_var", c_d), ("max_flux_simple", c_d), ("tiny_flux", c_d), ("pixel_scale", c_d), ("dg_stepk", c_d),
        ("star_size", c_vp), ("gal_radius", c_vp), ("zenith", c_d * 3), ("has_field", c_i32), ("pad", c_i32),
        ("sb_flux", c_vp), ("sersic_b", c_vp), ("sersic_norm", c_vp), ("sersic_inv_n", c_vp), ("keep_sb", c_d),
        ("psf_size_keep", c_i32), ("psf_size_keep3", c_i32), ("sb_tables", c_i32), ("pad2", c_i32)]


class ObjectMeta(C.Structure):
    _fields_ = [("n_phot", c_i64), ("size", c_i32), ("flags", c_i32)]


META_DTYPE = np.dtype([("n_phot", "<i8"), ("size", "<i4"), ("flags", "<i4")])
IMS_META_SIZE_PENDING, IMS_META_HOST_ROW = 1, 2
IMS_FLUX_PIXEL = -7


class PlanItem(C.Structure):
    _fields_ = [("kind", c_i32), ("stream", c_i32), ("params", c_vp), ("pool", c_vp), ("aux", c_vp),
                ("first_slot", c_i32), ("n_slots", c_i32), ("n_tiles", c_i64), ("tag", C.c_uint32), ("pad", C.c_uint32),
                ("aux2", c_vp)]


(IMS_PLAN_RENDER, IMS_PLAN_SHOOT_POOL, IMS_PLAN_ACC_POOL, IMS_PLAN_UPDATE, IMS_PLAN_INIT, IMS_PLAN_RECORD,
 IMS_PLAN_WAIT) = range(1, 8)

IMS_MAX_AMPS = 16


class Amp(C.Structure):
    _fields_ = [("x0", c_i32), ("y0", c_i32), ("flip_x", c_i32), ("flip_y", c_i32), ("gain", C.c_float),
                ("bias_level", C.c_float), ("read_noise", C.c_float), ("pad", c_i32)]


class Readout(C.Structure):
    _fields_ = [("n_amps", c_i32), ("seg_w", c_i32), ("seg_h", c_i32), ("raw_w", c_i32), ("raw_h", c_i32),
                ("data_x0", c_i32), ("data_y0", c_i32), ("has_xtalk", c_i32), ("amps", Amp * IMS_MAX_AMPS),
                ("xtalk", C.c_float * (IMS_MAX_AMPS * IMS_MAX_AMPS))]


class PlanInput(C.Structure):
    """ims_plan_input_t: what the native LSST_Image planner reads of a catalog (host arrays)"""
    _fields_ = [("n", c_i64), ("row", c_vp), ("n_phot", c_vp), ("stamp", c_vp), ("faint", c_vp), ("nrecalc", c_i32),
                ("n_class_rounds", c_i32), ("class_rounds", c_i32 * 4), ("n_static_slots", c_i32), ("slot_capacity", c_i32),
                ("static_cells", c_i64), ("scratch_cells", c_i64), ("max_pool_photons", c_i64), ("seg_size", c_i32),
                ("want_realized", c_i32), ("event_base", c_i32), ("use_tags", c_i32), ("coarse_slices", c_i32), ("pad", c_i32)]


class PlanSizes(C.Structure):
    _fields_ = [("arena_bytes", c_i64), ("rows_bytes", c_i64), ("pool_photons", c_i64), ("realized_count", c_i64),
                ("n_groups", c_i32), ("n_events", c_i32), ("n_render_launches", c_i64), ("render_photons", c_i64),
                ("render_rows", c_i64), ("render_segments", c_i64), ("n_shoot_launches", c_i64), ("shoot_photons", c_i64),
                ("shoot_rows", c_i64), ("shoot_segments", c_i64), ("chain_rows", c_i64), ("n_objects", c_i64), ("n_round_launches", c_i64)]


Tuning = tuning.Tuning          # ims_tuning_t lives with the one module that fills it

STRUCTS = [Object, RadialTables, LinTables, PsfComponent, Op, Surface, TanSip, Optics, BfSlot, Sensor, Photons,
           RenderParams, PlanItem, Atmosphere, FftObject, FftParams, Readout, Chain, Catalog, ObjectMeta, PlanInput, PlanSizes, Tuning]

# every symbol include/imsim_hip.h declares
EXPORTS = ["ims_abi_version", "ims_last_error", "ims_device_count", "ims_device_info", "ims_known_optics_layout",
           "ims_tuning_defaults", "ims_get_tuning", "ims_set_tuning", "ims_shoot_accumulate",
           "ims_shoot_photons", "ims_shoot_ops_photons", "ims_accumulate_segments", "ims_accumulate_small", "ims_accumulate_round", "ims_run_plan",
           "ims_fft_kspace_fill", "ims_fft_finish", "ims_fft_spikes", "ims_fft_spike_table", "ims_apply_ops", "ims_accumulate", "ims_sensor_init_boundaries",
           "ims_sensor_update_distortions", "ims_sensor_update_distortions_fold", "ims_sensor_fold_delta", "ims_image_add", "ims_image_to_float", "ims_fill_derived_op", "ims_fill_derived_medium", "ims_fill_derived_optics", "ims_fill_derived_atmosphere", "ims_fill_derived_sensor", "ims_sensor_pixel_areas", "ims_flat_add", "ims_last_kernel_ms", "ims_enable_timing",
           "ims_readout_bleed", "ims_readout_segments", "ims_readout_cte", "ims_readout_finish",
           "ims_build_object_table", "ims_patch_stamp_sizes", "ims_gather_rows", "ims_parse_instcat_objects", "ims_screen_prepass",
           "ims_plan_lsst_image", "ims_plan_bind", "ims_plan_upload", "ims_plan_run", "ims_plan_run_deferred", "ims_plans_run_joint", "ims_plan_join", "ims_plan_add_realized", "ims_plan_destroy",
           "ims_fft_inverse", "ims_fft_inverse_raw", "ims_fft_spikes_listed", "ims_fft_warm", "ims_comm_unique_id", "ims_comm_init", "ims_comm_destroy", "ims_reduce_image", "ims_allreduce_delta",
           "ims_count_inexact", "ims_struct_size", "ims_test_math"]

_LIB_PATH = tuning.env("IMSIM_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libimsim_hip.so")
_lib = None


class ImsimHipError(RuntimeError):
    pass


def lib_path():
    return _LIB_PATH


def load():
    """Load libimsim_hip.so (built in-tree by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImsimHipError(f"{_LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback)")
    # Device pointers and stream handles are exchanged with torch (which only does memory /
    # stream / RCCL plumbing), so both must sit on ONE HIP runtime instance: torch ships its own
    # libamdhip64 and must be loaded first, so that this library binds to the same copy.
    import torch  # noqa: F401
    if torch.cuda.device_count() > 0:          # (counting devices does not initialise the GPU; a host without one makes no transforms)
        tuning.fft_kernel_cache(os.path.join(os.path.dirname(_LIB_PATH), "rocfft_kernels.db"))
    lib = C.CDLL(_LIB_PATH)
    lib.ims_last_error.restype = C.c_char_p
    for k, st in enumerate(STRUCTS):
        got = lib.ims_struct_size(k)
        if got != C.sizeof(st):
            raise ImsimHipError(f"ABI mismatch for {st.__name__}: library {got} bytes, binding {C.sizeof(st)}")
    lib.ims_shoot_accumulate.argtypes = [C.POINTER(RenderParams), c_vp]
    lib.ims_shoot_photons.argtypes = [C.POINTER(RenderParams), c_vp, C.POINTER(Photons), c_vp]
    lib.ims_apply_ops.argtypes = [C.POINTER(RenderParams), c_vp, C.POINTER(Photons), c_vp]
    lib.ims_shoot_ops_photons.argtypes = [C.POINTER(RenderParams), c_vp, C.POINTER(Photons), c_vp]
    lib.ims_accumulate_segments.argtypes = [C.POINTER(RenderParams), C.POINTER(Photons), c_vp, c_i32, c_vp]
    lib.ims_accumulate_small.argtypes = [C.POINTER(RenderParams), C.POINTER(Photons), c_vp, c_i32, c_vp]
    lib.ims_accumulate_round.argtypes = [C.POINTER(RenderParams), C.POINTER(Photons), c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]
    lib.ims_accumulate.argtypes = [C.POINTER(RenderParams), c_vp, C.POINTER(Photons), c_vp, c_vp]
    lib.ims_sensor_init_boundaries.argtypes = [c_vp, C.POINTER(Sensor), c_i32, c_i32, c_vp, c_i64, c_vp]
    lib.ims_sensor_update_distortions.argtypes = [c_vp, C.POINTER(Sensor), c_i32, c_i32, c_vp, c_i64, c_vp, C.c_uint32, c_vp]
    lib.ims_sensor_update_distortions_fold.argtypes = [c_vp, C.POINTER(Sensor), c_vp, c_i64, c_vp, C.c_uint32, c_vp, c_i32, c_i32, c_vp]
    lib.ims_sensor_fold_delta.argtypes = [c_vp, C.POINTER(Sensor), c_i32, c_vp, c_i32, c_i32, c_vp]
    lib.ims_image_add.argtypes = [c_vp, c_vp, c_i64, c_vp]
    lib.ims_image_to_float.argtypes = [c_vp, c_vp, c_i64, c_vp]
    lib.ims_sensor_pixel_areas.argtypes = [c_vp, C.POINTER(Sensor), c_i32, c_vp, c_vp, c_vp]
    lib.ims_flat_add.argtypes = [c_vp, c_vp, c_d, c_d, c_u64, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp]
    lib.ims_fill_derived_op.argtypes = [c_vp]
    lib.ims_fill_derived_medium.argtypes = [c_i32, C.POINTER(c_d)]
    lib.ims_fill_derived_optics.argtypes = [c_vp]
    lib.ims_fill_derived_atmosphere.argtypes = [c_vp]
    lib.ims_fill_derived_sensor.argtypes = [c_vp]
    lib.ims_fft_kspace_fill.argtypes = [C.POINTER(FftParams), c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]
    lib.ims_fft_finish.argtypes = [C.POINTER(FftParams), c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]
    lib.ims_fft_spikes.argtypes = [C.POINTER(FftParams), c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]
    lib.ims_fft_spike_table.argtypes = [C.POINTER(Spikes), c_vp, c_vp, c_vp, c_vp, c_vp]
    lib.ims_run_plan.argtypes = [C.POINTER(PlanItem), c_i64, c_vp, C.POINTER(Sensor), c_vp, C.POINTER(c_vp), c_i32]
    lib.ims_last_kernel_ms.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int)]
    lib.ims_enable_timing.argtypes = [C.c_int]
    lib.ims_known_optics_layout.argtypes = [c_u64]
    lib.ims_build_object_table.argtypes = [C.POINTER(Catalog), c_vp, c_vp, c_vp, c_vp]
    lib.ims_patch_stamp_sizes.argtypes = [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]
    lib.ims_gather_rows.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i64, c_vp]
    lib.ims_screen_prepass.argtypes = [C.POINTER(RenderParams), c_i32, c_i32, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]
    lib.ims_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.ims_device_info.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(c_i64), C.POINTER(c_i64)]
    lib.ims_readout_bleed.argtypes = [c_vp, c_vp, c_i32, c_i32, c_d, c_i32, c_vp]
    lib.ims_readout_segments.argtypes = [c_vp, c_i32, c_i32, C.POINTER(Readout), c_vp, c_vp]
    lib.ims_readout_cte.argtypes = [c_vp, c_vp, C.POINTER(Readout), c_vp, c_i32, c_i32, c_vp]
    lib.ims_readout_finish.argtypes = [c_vp, C.POINTER(Readout), c_u64, c_vp, c_vp]
    lib.ims_test_math.argtypes = [C.c_int, c_vp, c_vp, c_i64, c_u64, c_i64, C.c_uint32, c_vp]
    lib.ims_plan_lsst_image.argtypes = [C.POINTER(PlanInput), C.POINTER(c_vp), C.POINTER(PlanSizes)]
    lib.ims_plan_bind.argtypes = [c_vp, C.POINTER(RenderParams), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]
    lib.ims_plan_upload.argtypes = [c_vp, c_vp]
    lib.ims_plan_run.argtypes = [c_vp, c_vp, C.POINTER(Sensor), c_vp, c_vp, c_vp, C.POINTER(c_vp), c_i32, c_i32]
    lib.ims_plan_run_deferred.argtypes = [c_vp, c_vp, C.POINTER(Sensor), c_vp, c_vp, c_vp, C.POINTER(c_vp), c_i32, c_i32, C.POINTER(c_i32)]
    lib.ims_plans_run_joint.argtypes = [C.POINTER(c_vp), c_i32, c_vp, c_i32, c_i32]
    lib.ims_plan_join.argtypes = [c_vp, c_vp]
    lib.ims_plan_add_realized.argtypes = [c_vp, c_vp, c_vp]
    lib.ims_plan_destroy.argtypes = [c_vp]
    lib.ims_fft_spikes_listed.argtypes = [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]
    lib.ims_fft_inverse.argtypes = [c_vp, c_vp, c_i32, c_i64, c_vp]
    lib.ims_fft_inverse_raw.argtypes = [c_vp, c_vp, c_i32, c_i64, c_vp]
    lib.ims_fft_warm.argtypes = [c_i32, c_vp]
    lib.ims_comm_unique_id.argtypes = [c_vp]
    lib.ims_comm_init.argtypes = [c_vp, c_i32, c_i32, C.POINTER(c_vp)]
    lib.ims_comm_destroy.argtypes = [c_vp]
    lib.ims_reduce_image.argtypes = [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]
    lib.ims_allreduce_delta.argtypes = [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp]
    lib.ims_count_inexact.argtypes = [c_vp, c_i64, c_i32, c_vp, c_vp]
    lib.ims_tuning_defaults.argtypes = [C.POINTER(Tuning)]
    lib.ims_get_tuning.argtypes = [C.POINTER(Tuning)]
    lib.ims_set_tuning.argtypes = [C.POINTER(Tuning)]
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().ims_last_error()
        raise ImsimHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
