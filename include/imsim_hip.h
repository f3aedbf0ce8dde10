/*
 * imsim_hip.h -- C-ABI of libimsim_hip.so, the MI355X (gfx950) stamp-rendering hot path.
 *
 * This is the drop-in boundary for imSim's per-object draw loop.  Every entry point takes plain
 * pointers, sizes and POD structs (no torch / C++ types).  Device pointers are HIP device
 * addresses; `stream` is a hipStream_t passed as void*.  All functions return 0 (IMS_OK) or a
 * negative error code; ims_last_error() returns a thread-local message.
 *
 * Reference interfaces replaced (paths relative to the imSim reference tree):
 *   ims_shoot_accumulate   <- LSST_SiliconBuilder.draw, phot branch: gal.drawImage(method='phot',
 *                             n_photons=phot_flux, sensor=..., photon_ops=psfs+ops)
 *                             imsim/stamp.py:527-573, and the stamp->CCD add imsim/lsst_image.py:353-368
 *   ims_shoot_photons      <- LSST_PhotonsBuilder.draw (save_photons=True, NullSensor)
 *                             imsim/stamp.py:708-743
 *   ims_apply_ops          <- `for op in photon_ops: op.applyTo(photons, local_wcs, rng)`
 *                             imsim/photon_pooling.py:154-155 ; PhotonOp.applyTo imsim/photon_ops.py:81,304,520
 *   ims_accumulate         <- accumulate_photons -> sensor.accumulate(photons, image, resume, recalc)
 *                             imsim/photon_pooling.py:195-225
 *   ims_sensor_*           <- galsim SiliconSensor as configured by imsim/lsst_image.py:93-103 and
 *                             config/imsim-config.yaml:230-235 (tree rings imsim/treerings.py:169-195)
 *   ims_fft_*              <- LSST_SiliconBuilder.draw, fft branch imsim/stamp.py:482-525
 *
 * Numerics contract: all photon arithmetic is IEEE binary64 built only from + - * / fma sqrt and
 * integer ops (DESIGN.md "numerics spec"), so the HIP kernels and the CPU oracle produce the same
 * bits.  Random numbers are Philox4x32-10, counter-addressed by (object id, photon index, slot):
 * results do not depend on batching, launch geometry or the number of GPUs.
 *
 * Process model: ONE process per GPU.  The library keeps process-global state -- the hipEvent pairs of
 * ims_enable_timing / ims_last_kernel_ms, the events of ims_run_plan's record / wait items, the table of streams of
 * ims_plan_* -- in file-static containers without locks: calls into it must come from one thread at a time (the
 * message of ims_last_error is thread-local, the state is not).  Launches are asynchronous: an entry point returns
 * after hipGetLastError() of its own launches, so a fault inside a kernel surfaces as IMS_ERR_HIP at the next call
 * that synchronises (ims_last_kernel_ms, a copy, the caller's own stream synchronisation), not at the call that
 * launched it.
 */
#ifndef IMSIM_HIP_H
#define IMSIM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMS_OK              0
#define IMS_ERR_ARG        -1
#define IMS_ERR_HIP        -2
#define IMS_ERR_NO_DEVICE  -3
#define IMS_ERR_UNSUPPORTED -4

#define IMS_ABI_VERSION 22

/* Pointers stored INSIDE descriptors that live in device memory (ims_sensor_t, ims_atmosphere_t) always point to global
 * device memory.  Device code is told so (address space 1): a pointer read out of memory is otherwise a generic pointer and
 * every access through it a slow flat_load / flat_store.  Host compilers and bindings see plain pointers; the layout is the same. */
#if defined(__HIP_DEVICE_COMPILE__)
#define IMS_G __attribute__((address_space(1)))
#else
#define IMS_G
#endif

/* ---- object flags ---- */
#define IMS_OBJ_FAINT   1   /* nominal_flux < max_flux_simple: no photon ops, no sensor (stamp.py:435-465,555-556) */

/* ---- PSF component kinds (psfs are applied as the first photon ops, stamp.py:553) ---- */
#define IMS_PSF_GAUSSIAN 1  /* p0 = sigma [arcsec] */
#define IMS_PSF_RADIAL   2  /* table = radial table id, p0 = scale [arcsec per table unit] */
#define IMS_PSF_SCREENS  3  /* PhaseScreenPSF, geometric photon shooting through ims_atmosphere (imsim/atmPSF.py:298-320);
                               p0 = arcsec per (nm/m) of wavefront gradient = 1e-9 * 206265; also samples pupil_u/v and time */
#define IMS_PSF_DOUBLE_GAUSSIAN 4  /* imsim DoubleGaussianPSF (atmPSF.py:448-486): Gaussian of sigma p0 with probability p2,
                                    * else of sigma p1 [arcsec] */
#define IMS_MAX_PSF 4

/* ---- photon-op kinds (names follow the registered PhotonOp types) ---- */
#define IMS_OP_TIME_SAMPLER              1  /* p0=t0, p1=exptime                      (config/imsim-config.yaml:282-285) */
#define IMS_OP_PUPIL_ANNULUS_SAMPLER     2  /* p0=R_outer, p1=R_inner                 (:286-289); p2 = p1^2, p3 = p0^2 - p1^2 derived by ims_fill_derived_op */
#define IMS_OP_PHOTON_DCR                3  /* p0=base_wavelength [nm], p1=pressure kPa, p2=temperature K, p3=H2O kPa, p4=scale (rad->arcsec) (:290-296);
                                             * p5..p7 derived by ims_fill_derived_op: air factors, refraction constant at p0 */
#define IMS_OP_RUBIN_OPTICS              4  /* p0=shift_photons (0/1)                 (imsim/photon_ops.py:24-127) */
#define IMS_OP_RUBIN_DIFFRACTION         5  /* p0=shift_photons, p1=disable_field_rotation (imsim/photon_ops.py:211-358) */
#define IMS_OP_RUBIN_DIFFRACTION_OPTICS  6  /* p0=shift_photons, p1=disable_field_rotation (imsim/photon_ops.py:151-208) */
#define IMS_OP_FOCUS_DEPTH               7  /* p0=depth [pixels]                      (:309-315) */
#define IMS_OP_REFRACTION                8  /* p0=index_ratio                         (:317-318); p1 = p0^2, p2 = p0^2 - 1 derived by ims_fill_derived_op */
#define IMS_OP_BANDPASS_RATIO            9  /* table = ratio table id                 (imsim/photon_ops.py:506-521) */
#define IMS_MAX_OPS 12

/* ---- sensor kinds ---- */
#define IMS_SENSOR_NONE    0  /* photons.addTo(image): pixel = floor(x+0.5) */
#define IMS_SENSOR_SILICON 1  /* galsim SiliconSensor restated */

/* ---- surface kinds for the sequential ray trace (batoid Optic.trace restated) ---- */
#define IMS_SURF_MIRROR   1
#define IMS_SURF_REFRACT  2
#define IMS_SURF_DETECTOR 3
#define IMS_SURF_BAFFLE   4   /* plane that only applies its obscuration */
#define IMS_MEDIUM_CONST     0   /* n = c0; c1 must hold 1/c0 */
#define IMS_MEDIUM_SELLMEIER 1   /* n^2 = 1 + sum B_i l^2/(l^2 - C_i), l in micron; c0..c5 = B1,B2,B3,C1,C2,C3 */
#define IMS_MEDIUM_AIR       2   /* Filippenko/Edlen air; c0=pressure kPa, c1=temperature K, c2=H2O kPa;
                                  * c3,c4 derived by ims_fill_derived_medium (pressure/temperature and water factors) */
#define IMS_OBSC_NONE          0
#define IMS_OBSC_CLEAR_ANNULUS 1  /* vignetted unless inner <= r <= outer */
#define IMS_OBSC_CLEAR_CIRCLE  2  /* vignetted unless r <= outer */
#define IMS_OBSC_OBSC_CIRCLE   3  /* vignetted if r < outer */
#define IMS_OBSC_OBSC_ANNULUS  4  /* vignetted if inner <= r < outer */
#define IMS_MAX_SURFACES 24

/* One catalog source for one call.  256 bytes, 64-byte aligned rows. */
#define IMS_PROF_POINT (-1)
#define IMS_PROF_BOX   (-2)
#define IMS_PROF_KNOTS (-3)
#define IMS_PROF_IMAGE (-4)
typedef struct ims_object {
    int64_t obj_id;        /* RNG stream id = catalog object number (per-object rng, stamp.py:166) */
    int64_t phot_first;    /* index of the first photon of this call within the object's stream */
    int64_t n_phot;        /* photons to shoot in this call (phot_flux, or one pooling-batch share) */
    double  x0, y0;        /* image_pos: CCD pixel coordinates of the object (GalSim convention) */
    double  flux_per_photon; /* +1 for n_photons=phot_flux, poisson_flux=False (stamp.py:562-572) */
    double  prof_scale;    /* arcsec per radial-table unit (e.g. half-light radius) */
    double  jac[4];        /* profile affine a,b,c,d: (u,v)' = (a u + b v, c u + d v) [arcsec]; shear/lens/rot (instcat.py:498-527) */
    double  winv[4];       /* local WCS inverse jacobian: (dx,dy) = winv * (du,dv), arcsec -> pixels */
    double  dcr_tanz;      /* PhotonDCR: tan(zenith angle) of this object */
    double  dcr_sinp;      /* PhotonDCR: sin(parallactic angle) */
    double  dcr_cosp;      /* PhotonDCR: cos(parallactic angle) */
    int32_t prof_table;    /* >= 0: radial profile table id (Sersic ...), scaled by prof_scale;
                            * IMS_PROF_POINT: DeltaFunction; IMS_PROF_BOX: galsim.Box(prof_scale, prof_aux) [arcsec] (streaks,
                            * imsim/instcat.py:487-496); IMS_PROF_KNOTS: galsim.RandomKnots of prof_aux points drawn from a
                            * Gaussian of sigma prof_scale (instcat.py:529-546) */
    int32_t sed_table;     /* wavelength inverse-CDF table id; -1 = monochromatic at sed_wave */
    int32_t flags;         /* IMS_OBJ_* */
    int32_t stamp_xmin, stamp_xmax, stamp_ymin, stamp_ymax; /* stamp bounds, inclusive; photons outside are lost */
    int32_t bf_state;      /* >=0: index into ims_sensor.bf_slots (private pixel boundaries); -1: static CCD boundaries */
    double  sed_wave;      /* wavelength [nm] when sed_table < 0 */
    double  atm_tan_x, atm_tan_y; /* tan of the field angle of the object from the boresight (theta of atm.makePSF, atmPSF.py:304) */
    double  prof_aux;      /* IMS_PROF_BOX: width [arcsec]; IMS_PROF_KNOTS: number of knots; IMS_PROF_IMAGE: image index
                            * (prof_scale = arcsec per image pixel) */
    int64_t screen_base;   /* launches with params->screen_kick: the kick of photon k (absolute index in the object's stream) of
                            * the phase-screen PSF component is screen_kick[2 (screen_base + k)], [.. + 1] (ims_screen_prepass) */
    double  reserved[5];   /* pads the row to 256 bytes */
} ims_object_t;

/* Pixel images sampled as profiles (galsim.InterpolatedImage of a FITS stamp, imsim/instcat.py:552-561): image k has
 * size[2k] x size[2k+1] pixels (row-major, first axis x) and a cumulative distribution of its non-negative pixel values
 * cdf[offset[k] .. offset[k] + w*h] (first entry 0, last 1).  A photon picks pixel p by inverse CDF of one deviate u; the
 * remainder f of that deviate inside the pixel's CDF bin and a second deviate u2 place it
 *   interp == 0: uniformly inside the pixel, (p_x + f, p_y + u2) (GalSim's "nearest" interpolant);
 *   interp == 1: at the pixel centre plus one offset per axis drawn from |K| of the image's x-interpolant K, as GalSim's
 *                SBInterpolatedImage::shoot followed by Interpolant::shoot does: the cumulative distribution of |K| is
 *                tabulated at n_k + 1 knots kx (offsets in pixels, increasing, every sign change of K among them) with
 *                values kcdf (0 .. 1); a deviate finds its interval by bisection and is placed linearly inside it.  The
 *                photon's flux is multiplied by norm = (integral |K|)^2 and by the sign of K(dx) K(dy): K < 0 where
 *                neg[0] < |d| < neg[1] or neg[2] < |d| < neg[3] (Quintic, GalSim's default: 1 .. 2 and
 *                (25 + sqrt 31) / 11 .. 3). */
typedef struct ims_image_tables {
    int32_t n_images;
    int32_t interp;
    const int32_t* size;     /* [n_images][2] */
    const int64_t* offset;   /* [n_images] */
    const double*  cdf;
    const double*  kx;       /* interp == 1: [n_k + 1] */
    const double*  kcdf;     /* interp == 1: [n_k + 1] */
    int32_t n_k;
    int32_t pad;
    double  norm;            /* interp == 1: (integral |K|)^2 */
    double  neg[4];          /* interp == 1: the two intervals of |d| on which K is negative (empty: lower >= upper) */
} ims_image_tables_t;

/* Tabulated circular profiles sampled by inverse CDF with uniform density inside each annulus:
 * r^2 = r2[i] + (u - cdf[i])/(cdf[i+1]-cdf[i]) * (r2[i+1]-r2[i]).  Table units are scaled per use. */
typedef struct ims_radial_tables {
    int32_t n_tables;
    int32_t n_bins;          /* each table has n_bins+1 knots */
    const double* r2;        /* [n_tables][n_bins+1] squared radii, increasing */
    const double* cdf;       /* [n_tables][n_bins+1] enclosed flux fraction, cdf[0]=0, cdf[n_bins]=1 */
    const int32_t* guide;    /* optional [n_tables][n_guide+1]: guide[g] = last knot with cdf <= g/n_guide, so that the bin
                              * search of a deviate u starts inside [guide[g], guide[g+1]], g = floor(u n_guide); same bin
                              * as the full bisection.  NULL: bisect over the whole table */
    int32_t n_guide;         /* a power of two */
    int32_t pad;
} ims_radial_tables_t;

/* Generic 1-D tables uniform in their argument, linear interpolation, clamped at the ends.
 * Used for: wavelength inverse CDFs (argument u in [0,1]), bandpass ratios (argument wavelength). */
typedef struct ims_lin_tables {
    int32_t n_tables;
    int32_t n_pts;
    double  arg_min;
    double  arg_step;
    const double* val;       /* [n_tables][n_pts] */
} ims_lin_tables_t;

typedef struct ims_psf_component {
    int32_t kind;            /* IMS_PSF_* */
    int32_t table;           /* radial table id for IMS_PSF_RADIAL */
    double  p0;              /* sigma or scale [arcsec] */
    double  chrom_alpha;     /* size scales as (wavelength/chrom_base)^alpha; 0 = achromatic */
    double  chrom_base;      /* nm */
    double  p1, p2;          /* kind-specific (IMS_PSF_DOUBLE_GAUSSIAN) */
} ims_psf_component_t;

/* The frozen-flow atmosphere of one visit (galsim.Atmosphere as built by imsim/atmPSF.py:164-205):
 * n_layers periodic von Karman phase screens of npix x npix samples (optical path difference, nm),
 * each drifting with its wind.  A photon entering the pupil at (u,v) [m] at time t sees layer l at
 * (u - t vx_l + h_l tan(theta_x), v - t vy_l + h_l tan(theta_y)); its position kick is the sum of the
 * bilinear-interpolant gradients there. */
#define IMS_MAX_LAYERS 8
typedef struct ims_atmosphere {
    int32_t n_layers;
    int32_t npix;
    double  scale;                 /* m per screen sample */
    double  x0;                    /* coordinate of sample 0 [m] (= -0.5 * npix * scale) */
    double  t0, exptime;           /* s */
    double  aper_r_outer, aper_r_inner;   /* pupil sampling annulus [m] (diam 8.36, obscuration 0.61: atmPSF.py:168) */
    double  vx[IMS_MAX_LAYERS], vy[IMS_MAX_LAYERS];   /* m/s */
    double  alt[IMS_MAX_LAYERS];   /* m */
    const float IMS_G*  screens;         /* [n_layers][npix][npix], fp32 samples (the arithmetic on them is f64) */
    /* launch-wide constants derived by ims_fill_derived_atmosphere (the same IEEE operations the kernel would otherwise repeat
     * per photon): (double)npix, 1/npix, 1/scale, aper_r_inner^2, aper_r_outer^2 - aper_r_inner^2 */
    double  dn, inv_n, inv_scale, aper_ri2, aper_dr2;
    /* optional (NULL = gather the four samples from `screens`): [n_layers][npix][npix][4] fp32, for every sample (iy, ix) the
     * 2 x 2 cell the bilinear gradient needs, {s[iy][ix], s[iy][ix+1], s[iy+1][ix], s[iy+1][ix+1]} with the periodic wrap
     * applied -- ONE 16-byte load per photon and layer instead of four scattered 4-byte loads on two rows (the fused
     * kernel of C3b moved 22 GB per launch that way).  Same sample values, same arithmetic, 4 x the memory. */
    const float IMS_G* screen_quads;
} ims_atmosphere_t;

typedef struct ims_op {
    int32_t kind;            /* IMS_OP_* */
    int32_t table;
    double  p[8];
} ims_op_t;

typedef struct ims_surface {
    int32_t kind;            /* IMS_SURF_* */
    int32_t obsc_kind;       /* IMS_OBSC_* */
    int32_t medium_kind;     /* medium AFTER this surface (refractive surfaces) */
    int32_t n_asphere;       /* number of even asphere coefficients used (0..4): r^4, r^6, r^8, r^10 */
    int32_t medium_id;       /* host-assigned: equal ids <=> identical medium (lets the kernel reuse n(lambda)) */
    int32_t pad;
    double  z0;              /* vertex z in telescope coordinates [m] */
    double  R;               /* radius of curvature [m]; 0 = plane */
    double  inv_R;           /* 1/R (0 for a plane), precomputed on the host */
    double  conic;
    double  asph[4];
    double  obsc_inner, obsc_outer;   /* [m] */
    double  medium_c[6];
    /* derived by ims_fill_derived_optics: 1 + conic, (1 + conic) * inv_R, -2 R, inv_R^2, obsc_inner^2, obsc_outer^2,
     * asph[k] * (k + 2) -- wave-uniform products the ray trace would otherwise form per photon and surface */
    double  k1, k1c, m2R, cc, obsc_i2, obsc_o2, asph_d[4];
} ims_surface_t;

/* TAN-SIP world coordinate system, trig-free vector form:
 * pixel -> (u,v)=pix-crpix -> +SIP(A,B) -> CD -> tangent plane (xi,eta) [rad] -> unit vector
 *   p = rot^T (1, xi, eta)/norm.  Inverse solves the SIP by Newton iteration. */
typedef struct ims_tansip {
    double crpix[2];
    double cd[4];            /* rad per pixel */
    double cdinv[4];
    double rot[9];           /* rows: unit vectors of the tangent point (e0), +xi (e1), +eta (e2) in ICRF */
    int32_t order;           /* SIP order, 0 = pure TAN; <= 4 */
    int32_t pad;
    double a[25];            /* a[p*5+q] u^p v^q */
    double b[25];
} ims_tansip_t;

/* Everything RubinOptics / RubinDiffraction / RubinDiffractionOptics need (imsim/photon_ops.py). */
typedef struct ims_optics {
    ims_tansip_t img_wcs;        /* base['current_image'].wcs: pixel <-> ICRF */
    ims_tansip_t icrf_to_field;  /* base['_icrf_to_field']: field angle [rad] <-> ICRF */
    int32_t in_medium_kind;      /* telescope.inMedium */
    int32_t n_surfaces;
    double  in_medium_c[6];
    double  stop_z;              /* z of the stop surface plane (rays start at (pupil_u, pupil_v, stop_z)) */
    ims_surface_t surf[IMS_MAX_SURFACES];
    double  cam_rot[2];          /* cos, sin of the camera rotator angle applied to the detector-plane hit (telescope_loader.py:242-246) */
    double  fp_to_pix[6];        /* focal_to_pixel affine (imsim/utils.py:42-59): x = m0*fpx + m1*fpy + m2 ; y = m3*fpx + m4*fpy + m5, fp in mm;
                                    called as focal_to_pixel(ray.y*1e3, ray.x*1e3) (imsim/photon_ops.py:495) */
    double  slope_jac[4];        /* normalised M @ jac_focal_to_pixel (imsim/photon_ops.py:497-500): dxdz = (s0 vx + s1 vy)/vz, dydz = (s2 vx + s3 vy)/vz */
    /* spider diffraction (imsim/diffraction.py:32-42) */
    int32_t n_lines, n_circles;
    double  lines[8][4];         /* nx, ny, d, thickness */
    double  circles[4][3];       /* cx, cy, r */
    double  e_z0[3];             /* zenith at t=0, equatorial frame (diffraction.py:284-304) */
    double  e_focal[3];          /* pointing in equatorial frame (diffraction.py:387-415) */
    double  cos_lat, sin_lat;
    double  omega;               /* Earth rotation rate [rad/s] (diffraction.py:280) */
    /* derived by ims_fill_derived_optics: e_focal x e_z0 and its norm (the time-independent half of the field-rotation angle) */
    double  rot_g[3], rot_gnorm;
} ims_optics_t;

/* Private pixel-boundary state of one brighter-fatter active region (a bright object's stamp in
 * LSST_Image mode, or the whole CCD in photon-pooling mode). */
typedef struct ims_bf_slot {
    int32_t xmin, ymin, nx, ny;   /* region in CCD pixel coordinates */
    int64_t offset;               /* owner-cell offset of this region; a region has (nx+1)*(ny+1) owner cells, row-major, x fastest */
} ims_bf_slot_t;

typedef struct ims_sensor {
    int32_t kind;                /* IMS_SENSOR_* */
    int32_t num_vertices;        /* NumVertices per edge (cfg); nv = 4*num_vertices+4 */
    int32_t nx, ny;              /* PixelBoundaryNx/Ny of the model (9) */
    int32_t qdist;               /* neighbour range for superposition (GalSim default 3) */
    int32_t n_abs;               /* absorption-length table points */
    int32_t n_tr;                /* tree-ring table points (0 = no tree rings) */
    int32_t pad;
    double  num_elec;            /* CollectedCharge_0_0 / strength */
    double  pixel_size;          /* micron */
    double  thickness;           /* micron */
    double  diff_step;           /* micron; 0 = no diffusion */
    double  abs_wl_min, abs_wl_step;   /* nm */
    double  tr_dr;               /* tree-ring table step [pixels] (imsim/treerings.py:100-103) */
    double  tr_cx, tr_cy;        /* tree-ring centre in CCD pixel coordinates (imsim/treerings.py:174-189) */
    const double IMS_G* abs_len;       /* [n_abs] micron */
    const double IMS_G* tr_table;      /* [n_tr] radial shift f(r) [pixels] */
    const double IMS_G* tr_table2;     /* [n_tr] second derivatives of the natural cubic spline through tr_table (galsim.LookupTable.from_func
                                    default interpolant, imsim/treerings.py:192-194); NULL = linear interpolation */
    const double IMS_G* distortions;   /* [nx][ny][nv][2] vertex displacement (pixel units) per num_elec of charge in the centre pixel */
    const double IMS_G* emptypoly;     /* [nv][2] undistorted polygon, counter-clockwise */
    /* brighter-fatter state */
    int32_t n_bf_slots;
    int32_t pad2;
    const ims_bf_slot_t IMS_G* bf_slots;
    double IMS_G* bf_boundary;         /* per owner cell of every slot: [2*num_vertices+2][2] owned boundary points (LL corner, bottom pts, LR corner, left pts) */
    double IMS_G* bf_bounds;           /* per owner cell (one 64-byte line): inner xmin,xmax,ymin,ymax, outer xmin,xmax,ymin,ymax */
    double IMS_G* bf_delta;            /* per owner cell: charge accumulated since the last recalc (Silicon's double _delta) */
    /* optional (NULL = off): one byte per owner cell, used at the first cell of every 16x16 tile of a region.
     * A launch that deposits charge with params->bf_tag != 0 stores the tag in bf_tile_charge; the update with
     * the same tag then skips tiles with no charge in reach and stores the tag in bf_tile_changed for tiles
     * whose boundary points moved.  Stale tags only cost work, never correctness. */
    unsigned char IMS_G* bf_tile_charge;
    unsigned char IMS_G* bf_tile_changed;
    /* >= 0: slot 0 is in its pristine (tree-ring only) state and no boundary point of it is displaced by more than this
     * [pixels]: a photon of a slot-0 object that converts further than the margin from every pixel edge is inside its
     * nominal pixel without looking at the boundary state.  The host derives it from the tree-ring table
     * (rigorous bound of the spline); ims_sensor_update_distortions on slot 0 sets it to -1 (= off) on the device. */
    double pristine_margin;
    /* derived by ims_fill_derived_sensor: diff_step / (thickness * pixel_size), thickness - 1 */
    double diff_coef, thick_m1;
    /* optional (qdist 3): the displacement table re-ordered for the update kernel, [dj + 3][di + 3][owned point][x, y] for
     * dj, di = -3 .. 4 (8 x 8 x (2 num_vertices + 2) x 2 doubles, device memory): distortions[di + cx][dj + cy][vertex of the owned
     * point].  With it the kernel reads a neighbour's row through the scalar cache; NULL = staged through LDS per tile. */
    const double IMS_G* bf_dl;
} ims_sensor_t;

/* A photon pool in device memory, SoA, the fields of galsim.PhotonArray (imsim/photon_ops.py:81). */
typedef struct ims_photons {
    int64_t n;
    double *x, *y, *flux, *dxdz, *dydz, *wavelength, *pupil_u, *pupil_v, *time;
    int32_t *obj_index;          /* row in the object table that produced the photon (for stamp clipping / truth) */
    /* converted != 0 (LSST_Image chains, ims_shoot_ops_photons -> ims_accumulate_round / ims_accumulate_segments): the pool holds
     * photons that have ALREADY been through the half of SiliconSensor.accumulate that does not look at the pixel boundaries
     * (conversion depth from the absorption length, lateral walk of an inclined photon, diffusion): x, y = position at the
     * conversion depth [pixels], flux (0 = lost in the silicon), dxdz = the polygon shrink factor tanh(zconv / 12) of GalSim's
     * insidePixel with the "pixel not found" coin of the photon in its sign bit (set = stay in the nominal pixel).  dydz,
     * wavelength, pupil_u, pupil_v, time are not used (may be NULL).  Objects without the Silicon model (or IMS_OBJ_FAINT) keep
     * x, y as they are.  Everything is the same arithmetic as the unconverted path, only done in the producing kernel. */
    int32_t converted;
    int32_t pad;
} ims_photons_t;

typedef struct ims_render_params {
    uint64_t seed;
    const ims_object_t* objects;     /* device, n_objects rows */
    int64_t  n_objects;
    const int64_t* seg_prefix;       /* device, [n_objects+1]: prefix sum of ceil(n_phot/seg_size) */
    int64_t  n_segments;
    int32_t  seg_size;               /* photons per workgroup segment; must be 256 (= workgroup size) */
    int32_t  n_psf;
    ims_psf_component_t psf[IMS_MAX_PSF];
    int32_t  n_ops;
    int32_t  track_static_delta;     /* 1 in photon-pooling mode: charge landing in slot 0 (whole CCD) is kept for the next recalc;
                                      * 2: ... and kept THERE ONLY -- the image takes it from the delta image at the recalculation
                                      * (ims_sensor_update_distortions_fold / ims_sensor_fold_delta) */
    ims_op_t ops[IMS_MAX_OPS];
    ims_radial_tables_t radial;
    ims_lin_tables_t    sed;         /* wavelength inverse CDFs */
    ims_lin_tables_t    ratio;       /* BandpassRatio tables */
    const ims_atmosphere_t* atm;     /* device pointer or NULL (needed by IMS_PSF_SCREENS) */
    const ims_optics_t* optics;      /* device pointer or NULL */
    const ims_sensor_t* sensor;      /* device pointer (struct itself lives in device memory) or NULL = IMS_SENSOR_NONE */
    /* target image */
    double*  image;                  /* device f64 accumulation image, row-major [ny][nx], pixel (ix,iy) at
                                      * image[(iy-ymin)*nx + (ix-xmin)]; ims_image_to_float makes the ImageF */
    int32_t  nx, ny, xmin, ymin;
    double*  realized_flux;          /* device [n_objects] or NULL: flux added per object (base['realized_flux'], stamp.py:573) */
    uint32_t bf_tag;                 /* 1..255: mark the tiles that receive delta charge (ims_sensor_t.bf_tile_charge); 0 = off */
    uint32_t bf_slot_shift;          /* added to the bf_state of objects with a private region (bf_state > 0) when the launch looks the
                                      * slot up; 0 (regions held as slot pairs used it: removed in round 5) */
    const int32_t* seg_object;       /* device [n_segments] or NULL: object index of every segment; when given, a
                                      * workgroup finds its object with one load instead of a search in seg_prefix */
    ims_image_tables_t images;       /* IMS_PROF_IMAGE profiles */
    /* optional (0 = none): the LAYOUT of *optics -- the (kind, shape) of its surfaces as 4-bit codes, first surface in the
     * lowest nibble: code = 1 + 3 kc + shape, kc 0 mirror / 1 refracting / 2 detector or baffle, shape 0 plane / 1 conic
     * (R != 0, no asphere terms) / 2 conic with asphere terms; a zero nibble ends the list.  It MUST describe the descriptor
     * `optics` points to (the kernels cannot check it): for the layouts the library holds an unrolled ray trace for
     * (ims_known_optics_layout) the launch takes that kernel, any other value runs the loop over the surfaces. */
    uint64_t optics_layout;
    /* optional (NULL = every photon gathers the phase screens itself): the sum over the layers of the screen gradient [nm/m] of
     * every photon, x and y interleaved, filled by ims_screen_prepass and indexed through ims_object_t.screen_base */
    const double* screen_kick;
    /* optional (0 = off).  An LSST_Image render whose slot 0 is the pristine CCD (tree rings only; track_static_delta 0) and is read
     * by nothing but ims_shoot_accumulate need not HOLD that state -- 235 B per pixel, written per CCD: photons further than
     * ims_sensor_t.pristine_margin from every pixel edge never look at it, and the ~2 % that do are set aside by the launch
     * (40 B each) and finished by a second launch that evaluates the polygons it needs from the tree-ring closed form -- the same
     * function the stored state is made by, so the same bits (round 5).  Needs 4 vertices per edge and a pristine_margin >= 0.
     * The caller sets lazy_static; the library fills the three fields behind it for its launches. */
    uint32_t lazy_static;
    uint32_t margin_cap;             /* (library) records of the overflow region */
    double*  margin_list;            /* (library) margin_waves x 8 records of 40 B, one octet per wavefront of the launch, then the overflow region */
    int32_t* margin_count;           /* (library) [0] records in the overflow region, [1] records that found no room (never: it is sized for the launch) */
    unsigned char* margin_wave_count;/* (library) per wavefront of the launch: records in its octet */
    int64_t  margin_waves;           /* (library) wavefronts of the launch: 4 x n_segments */
} ims_render_params_t;

/* ---- library ---- */
int  ims_abi_version(void);
const char* ims_last_error(void);
int  ims_device_count(int* count);
/* device properties used by the host scheduler: cus, xcds, lds bytes */
int  ims_device_info(int device, int* n_cu, int* n_xcd, int64_t* lds_bytes, int64_t* hbm_bytes);

/* 1 when the library holds kernels specialised for this optics layout (ims_render_params_t.optics_layout) */
int  ims_known_optics_layout(uint64_t layout);

/* ---- tuning: which of its equivalent forms the library launches ----
 * Every alternative below computes the same bits by another route (the parity tests run under each of them); the defaults are
 * the measured-fastest forms on MI355X.  The library reads NO environment variable for this (its behaviour does not depend on the
 * caller's environment): a caller that wants another form says so with ims_set_tuning, process-wide, before the launches it is to
 * apply to.  (The Python host keeps its own IMS_* environment parsing in ONE module, imsim_amd/tuning.py, which fills this block;
 * a replacement of imSim's GalSim calls that never touches it gets the defaults.)  Replaces nothing in the reference: GalSim has no
 * such switches; listed here because SURVEY 8(b) asks for one explicit boundary. */
typedef struct ims_tuning {
    int32_t chain_kernels;       /* 1: kernels with imSim's default photon-op chain / analytic PSF as straight-line code (run_ops<1>,
                                    run_psf<1>); 0: the loops over the descriptors */
    int32_t layout_kernels;      /* 1: ray trace unrolled for a known optics layout (ims_known_optics_layout); 0: loop over the surfaces */
    int32_t psf_screens_kernel;  /* 1: straight-line PSF code for imSim's default AtmosphericPSF (screens, second kick, Gaussian) */
    int32_t photon_lds;          /* dynamic LDS bytes a photon-kernel launch asks for without using it (caps the workgroups per CU);
                                    -1: automatic -- 41 984 B for launches that gather phase screens, 0 otherwise */
    int32_t round_compact;       /* 1: the pixel search of a round takes a 120-byte argument block; 0: the 1.4-KB launch parameters */
    int32_t init_tiles;          /* 1: tiled initial pixel-boundary state (4 vertices per edge); 0: one thread per owner cell */
    int32_t upd_dpp;             /* 1: updatePixelDistortions with its table delivered by DPP broadcasts for launches of at most
                                    upd_dpp_max tiles; 0: always the scalar-register form */
    int32_t joint_lists;         /* 1: joint rounds update / refresh over lists of the tiles with charge in reach (rounds of more than
                                    joint_list_min tiles); 0: sweeps over every tile */
    int64_t upd_dpp_max;         /* 128 */
    int64_t joint_list_min;      /* 1024 */
    double  active_fraction;     /* 0.25: workgroups launched per tile of a round for the list walkers (0 < f <= 1) */
    int32_t round_two_segments;  /* 0; 1: the pixel search of a round takes two 256-photon segments per workgroup, both pool records
                                    requested before the first search (measured: EXPERIMENTS.md, round 5) */
    int32_t joint_fine_marks;    /* 1: the lists of a joint round are built from charge marks per 4 x 4 pixels (a tile is listed when
                                    charge lies within the update's reach of it, not when one of its 3 x 3 tile neighbours holds some) */
    int32_t joint_search_lists;  /* 1: ... and the lists are appended to by the pixel search itself, where the charge lands (no launch that
                                    scans the marks of every tile afterwards); 0 = k_build_active_j */
    int32_t pad;
} ims_tuning_t;
int  ims_tuning_defaults(ims_tuning_t* out);
int  ims_get_tuning(ims_tuning_t* out);
int  ims_set_tuning(const ims_tuning_t* tuning);

/* ---- fused path: shoot -> PSF -> ops -> sensor -> CCD image (LSST_Silicon / LSST_Image) ---- */
int  ims_shoot_accumulate(const ims_render_params_t* params, void* stream);

/* ---- pooled path (LSST_Photons / LSST_PhotonPoolingImage) ---- */
/* photon_offset[n_objects+1] (device): where each object's photons live in the pool */
int  ims_shoot_photons(const ims_render_params_t* params, const int64_t* photon_offset,
                       const ims_photons_t* pool, void* stream);
/* shoot + PSF + the whole photon-op chain, stored to the pool (everything of a photon that does not
 * depend on the sensor state); pool->pupil_u/pupil_v/time/obj_index may be NULL */
int  ims_shoot_ops_photons(const ims_render_params_t* params, const int64_t* photon_offset,
                           const ims_photons_t* pool, void* stream);
/* sensor.accumulate for the objects of `params` (segment-mapped like ims_shoot_accumulate): photon j of
 * object row i is read from the (converted) pool at pool_start[i] + j.  num_vertices as for ims_accumulate_round.
 * Besides the brighter-fatter chains this is the per-batch step of photon-pooling mode when the photons of ALL batches
 * have been shot into an HBM-resident pool up front (photon_pooling.prepared_image): the row of an object holds the
 * photon count of its share of the batch, pool_start[i] the place of that share in the object's photons. */
int  ims_accumulate_segments(const ims_render_params_t* params, const ims_photons_t* pool,
                             const int64_t* pool_start, int32_t num_vertices, void* stream);
/* The same result for tables of SMALL shares (a few dozen photons per row): one wavefront per object row, no workgroup
 * charge tile; params->seg_prefix / seg_object / n_segments are not used.  Any photon count is accepted (a wavefront
 * loops over its row), the segment-mapped entry point above is the faster one above ~128 photons per row. */
int  ims_accumulate_small(const ims_render_params_t* params, const ims_photons_t* pool,
                          const int64_t* pool_start, int32_t num_vertices, void* stream);
int  ims_apply_ops(const ims_render_params_t* params, const int64_t* photon_offset,
                   const ims_photons_t* pool, void* stream);
/* pixel_index_out (device, [pool->n], may be NULL): flat image index each photon landed in, -1 = lost */
int  ims_accumulate(const ims_render_params_t* params, const int64_t* photon_offset,
                    const ims_photons_t* pool, int32_t* pixel_index_out, void* stream);

/* ---- Silicon sensor state ---- */
/* `sensor_dev` = the struct in device memory (all its pointers are device pointers);
 * `sensor_host` = a host copy whose bf_slots points to a HOST copy of the slot table (sizes the launches). */
/* boundary points = undistorted + tree rings, bounds refreshed, delta = 0 (Silicon::initialize).  tile_prefix_dev / n_tiles
 * (optional, as for ims_sensor_update_distortions: device prefix sum of the 16x16-cell tiles of the slots first_slot ..., its
 * last entry): with it the tiled kernel runs over exactly the tiles of the range; NULL / 0 = the library sizes its launches from
 * the host slot table. */
int  ims_sensor_init_boundaries(const ims_sensor_t* sensor_dev, const ims_sensor_t* sensor_host,
                                int32_t first_slot, int32_t n_slots, const int64_t* tile_prefix_dev, int64_t n_tiles, void* stream);
/* boundaries += distortions (x) delta / num_elec over the qdist neighbourhood; refresh bounds; delta = 0
 * (Silicon::updatePixelDistortions; the `recalc` of imsim/photon_pooling.py:159).
 * tile_prefix_dev[n_slots+1] (device): prefix sum of ceil((nx+1)/16)*ceil((ny+1)/16) over the slots of the
 * range (16x16 owner-cell tiles); n_tiles = its last entry.  changed_dev (device): one byte per owner
 * cell of the sensor, scratch. */
int  ims_sensor_update_distortions(const ims_sensor_t* sensor_dev, const ims_sensor_t* sensor_host,
                                   int32_t first_slot, int32_t n_slots, const int64_t* tile_prefix_dev,
                                   int64_t n_tiles, unsigned char* changed_dev, uint32_t tag, void* stream);
/* The same recalculation of slot 0 where slot 0 is the pixel grid of the CCD image (photon pooling), with Silicon's `target += delta`:
 * GalSim's SiliconSensor accumulates a batch into its delta image and adds the delta to the target when updatePixelDistortions
 * consumes it (the recalc of imsim/photon_pooling.py:159,195-225).  Launches whose
 * ims_render_params_t.track_static_delta is 2 deposit into the delta image ONLY -- one atomic add per photon instead of two -- and
 * this entry point adds every consumed delta cell to image_dev[j * nx + i] before it zeroes it.  Exact for integer charge (unit
 * photon fluxes), whatever the order.  ims_sensor_fold_delta does the adding alone (no recalculation): after the LAST batch, and
 * wherever the image is wanted in between. */
int  ims_sensor_update_distortions_fold(const ims_sensor_t* sensor_dev, const ims_sensor_t* sensor_host, const int64_t* tile_prefix_dev,
                                        int64_t n_tiles, unsigned char* changed_dev, uint32_t tag, double* image_dev, int32_t nx, int32_t ny,
                                        void* stream);
int  ims_sensor_fold_delta(const ims_sensor_t* sensor_dev, const ims_sensor_t* sensor_host, int32_t slot, double* image_dev, int32_t nx,
                           int32_t ny, void* stream);

/* ---- FFT branch: LSST_SiliconBuilder.draw, method == 'fft' (imsim/stamp.py:482-525) ----
 * For very bright objects (nominal_flux >= 1e6 and max_sb > fft_sb_thresh, imsim/stamp.py:275-277,
 * imsim/psf_utils.py:152-239) the reference draws Convolve([gal] + psfs) with GalSim's FFT renderer,
 * clips negatives, adds Poisson noise and adds the stamp to the CCD.  Here:
 *   ims_fft_kspace_fill : half-spectrum of every object of a batch = profile k-value x PSF MTFs x
 *                         pixel response x sub-pixel phase, on an nfft x (nfft/2+1) grid per object
 *   (inverse real 2-D FFT of each grid, unnormalised "backward" 1/N^2 convention, by the caller:
 *    rocFFT / hipFFT / torch.fft.irfft2 -- a plain library transform)
 *   ims_fft_finish      : clip < 0, Poisson noise (PoissonNoise, stamp.py:522), add stamp to the CCD */
#define IMS_KPSF_GAUSSIAN   1   /* p0 = sigma [arcsec]:            exp(-sigma^2 k^2 / 2) */
#define IMS_KPSF_KOLMOGOROV 2   /* p0 = k0 [rad/arcsec]:           exp(-(k/k0)^(5/3)) */
#define IMS_KPSF_TABLE      3   /* radial k-table `table` at k*p0 (VonKarman, Airy: imsim/psf_utils.py:109-126) */
typedef struct ims_kpsf {
    int32_t kind;
    int32_t table;
    double  p0;
} ims_kpsf_t;

typedef struct ims_fft_object {
    int64_t obj_id;
    int64_t k_offset;        /* complex-element offset of this object's half spectrum in the k buffer */
    int64_t r_offset;        /* real-element offset of its nfft x nfft image in the real buffer */
    double  flux;            /* fft_flux (nominal_flux x vignetting, imsim/psf_utils.py:220-233) */
    double  cx, cy;          /* object centre in FFT-grid index coordinates [pixels] */
    double  prof_scale;      /* arcsec per k-table unit (half-light radius) */
    double  jac[4];          /* real-space profile affine; the k-vector is transformed by its transpose */
    int32_t nfft;            /* FFT size (even) */
    int32_t prof_ktable;     /* k-table id of the profile; -1 = DeltaFunction */
    int32_t x0, y0;          /* CCD pixel coordinates of FFT-grid index (0,0) */
    int32_t stamp_xmin, stamp_xmax, stamp_ymin, stamp_ymax;
    int32_t pad[2];
} ims_fft_object_t;

/* Diffraction spikes of FFT-drawn objects (stamp.diffraction_fft, imsim/stamp.py:36-68, :520-521;
 * imsim/diffraction_fft.py:78-208): the bounding box of the pixels above `threshold` is convolved
 * with an analytic Lorentzian-cross stencil. */
typedef struct ims_spikes {
    int32_t enabled;
    int32_t cutoff;          /* spike_length_cutoff [pixels] */
    double  threshold;       /* brightness_threshold */
    double  cos0, sin0;      /* cos / sin of (alpha - d_alpha/2), alpha = pi/4 - rotTelPos */
    double  a_lo;            /* alpha - d_alpha */
    double  d_alpha;         /* field rotation over the exposure */
    double  scale;           /* 577.6 nm / wavelength */
    double  r0;              /* Lorentzian scale R_0 */
    double  norm;            /* sum of the stencil over (2 cutoff + 1)^2 offsets (host-computed) */
    /* optional (NULL = every stencil value evaluated in place): the non-zero entries of the NORMALISED stencil, stencil(a, b) / norm, as
     * a compressed table made once per visit by ims_fft_spike_table (the reference builds the whole (2 cutoff + 1)^2 array once per
     * visit, prepare_psf_field_rotation, imsim/diffraction_fft.py:78-123: 512 MB of which ~1e5 entries are not zero).  Row a + cutoff
     * holds the entries tab_row[a + cutoff] .. tab_row[a + cutoff + 1] - 1, column offsets b DESCENDING (= source columns ascending,
     * the order of the sum). */
    const int32_t* tab_row;  /* device, [2 cutoff + 2] */
    const int32_t* tab_col;  /* device, b of every entry */
    const double*  tab_val;  /* device, stencil(a, b) / norm of every entry */
} ims_spikes_t;

typedef struct ims_fft_params {
    uint64_t seed;
    double   pixel_scale;            /* arcsec / pixel */
    int32_t  n_kpsf;
    int32_t  add_noise;              /* 1: replace every pixel by a Poisson variate of its value */
    ims_kpsf_t kpsf[IMS_MAX_PSF];
    ims_lin_tables_t ktables;        /* radial k-space tables, uniform in their argument, 0 beyond the end */
    double*  image;                  /* CCD accumulation image, as in ims_render_params_t */
    int32_t  nx, ny, xmin, ymin;
    double*  realized_flux;          /* [n_objects] or NULL */
    ims_spikes_t spikes;
    int32_t  n_alias;                /* k-space fill: fold the aliases -n_alias..n_alias of the sampling frequency per axis (0 = base band) */
    int32_t  rbuf_raw;               /* ims_fft_spikes / ims_fft_finish: 1 = the real-space buffer they READ is ims_fft_inverse_raw's (no
                                      * 1 / N^2 yet): every value read is multiplied by 1 / (nfft * nfft) of its object first -- the
                                      * multiplication ims_fft_inverse does in a pass of its own, the same product, the same bits.
                                      * ims_fft_spikes WRITES the image itself (its output is never raw). */
} ims_fft_params_t;

/* elem_prefix[n_objects+1] (device): prefix sum of nfft*(nfft/2+1); kbuf: interleaved (re, im) doubles */
int  ims_fft_kspace_fill(const ims_fft_params_t* params, const ims_fft_object_t* objects_dev, int64_t n_objects,
                         const int64_t* elem_prefix_dev, int64_t n_elems, double* kbuf, void* stream);
/* Optional spike step between the inverse transform and ims_fft_finish: clip rbuf_in, find each object's
 * saturated bounding box (bbox_dev: int32 [n_objects][4] scratch = rowmin,rowmax,colmin,colmax), write
 * the spiked image to rbuf_out.  No-op copy when params->spikes.enabled == 0. */
/* The table of ims_spikes_t.tab_*: called twice.  First with row_ptr_dev = NULL: the number of non-zero entries of every row a = -cutoff
 * .. cutoff goes to row_count_dev[2 cutoff + 1]; the caller forms the prefix sum row_ptr (2 cutoff + 2 entries) and calls again with it
 * and with col_dev / val_dev of row_ptr[2 cutoff + 1] entries.  The values are spike_stencil(a, b) / norm as the kernel would have
 * formed them in place: the spikes are the same bits with and without the table. */
int  ims_fft_spike_table(const ims_spikes_t* spikes, const int32_t* row_ptr_dev, int32_t* row_count_dev, int32_t* col_dev, double* val_dev,
                         void* stream);
int  ims_fft_spikes(const ims_fft_params_t* params, const ims_fft_object_t* objects_dev, int64_t n_objects,
                    const int64_t* pix_prefix_dev, int64_t n_pix, const double* rbuf_in, double* rbuf_out,
                    int32_t* bbox_dev, void* stream);
/* ims_fft_spikes in two launches: the image streamed (clip, saturated box to zero) with the pixels near an arm of the cross -- the only
 * ones whose spike sum is not exactly zero, ~3 % of a bright star's stamp -- appended to list_dev (uint64 [list_cap]; count_dev: one
 * uint32, zeroed here), then their sums formed with every lane at work and added.  The same values in rbuf_out, bit for bit.  A list
 * that runs over (more such pixels than list_cap) costs one more launch that does the whole step as ims_fft_spikes does; n_pix / 8
 * entries are plenty. */
int  ims_fft_spikes_listed(const ims_fft_params_t* params, const ims_fft_object_t* objects_dev, int64_t n_objects,
                           const int64_t* pix_prefix_dev, int64_t n_pix, const double* rbuf_in, double* rbuf_out,
                           int32_t* bbox_dev, uint64_t* list_dev, int64_t list_cap, uint32_t* count_dev, void* stream);
/* pix_prefix[n_objects+1] (device): prefix sum of nfft*nfft; rbuf: the inverse-transformed images */
int  ims_fft_finish(const ims_fft_params_t* params, const ims_fft_object_t* objects_dev, int64_t n_objects,
                    const int64_t* pix_prefix_dev, int64_t n_pix, const double* rbuf, void* stream);

/* ---- phase-screen gather with managed locality (imsim/atmPSF.py:298-336; N1 of the survey's scope) ----
 * A photon of an AtmosphericPSF reads a 2 x 2 cell of each of the six 8192^2 screens at (pupil position + altitude x field
 * angle - wind x arrival time): a random place on a strip of 84 x (wind x exposure) samples per layer.  Gathered where the
 * photon is made, one workgroup per object, every access is a fresh line from a 1.6 GB table (5 x the bytes the gradient
 * needs).  This pre-pass does the gathers of ALL photons of a render in an order that keeps them in cache:
 *   1. every photon's arrival-time bucket (its time draw, slot 20 + component, is addressed by object and photon index, so it
 *      can be evaluated ahead) is counted per (slice, bucket): the objects of the spatially sorted table are cut into EIGHT
 *      contiguous slices of about equal photon counts, one per XCD;
 *   2. the (object, photon index) pairs are scattered into slice-major, time-minor order;
 *   3. slice x is walked by the workgroups that run on XCD x (block b -> XCD b mod 8) in time order: at any moment an XCD's L2
 *      holds the windows its eighth of the CCD looks through at one instant (~1 - 2 MB over the six layers), and every line of
 *      a screen is fetched about once per slice;
 *   4. the gradient sums go to screen_kick, which the shooting kernels read (coalesced) in place of the gathers.
 * Same arithmetic per photon (pupil position, time, bilinear gradient, layer order): bit-identical photons.
 * params: the FULL object table of the render (objects with phot_first / n_phot of the whole render and screen_base set,
 * seg_prefix / seg_object / n_segments over it).  comp: index of the IMS_PSF_SCREENS component.  n_buckets: a power of two,
 * at most 256.  slice_first_host[18]: first object of every slice, then first SEGMENT of every slice, each with the end as ninth entry
 * (HOST); max_slice_photons: photons of the largest slice.
 * scratch_dev: 8 n_buckets + 16 + 4 n_objects int64.  entries_dev:
 * one int64 per photon.  kick_dev: two doubles per photon. */
int  ims_screen_prepass(const ims_render_params_t* params, int32_t comp, int32_t n_buckets, const int64_t* slice_first_host,
                        int64_t max_slice_photons, int64_t* scratch_dev, int64_t* entries_dev, double* kick_dev, void* stream);

/* ---- object table on the device: LSST_SiliconBuilder.setup for a whole catalog in one launch ----
 * imsim/stamp.py:109-249 run per object: Poisson realisation of the flux (:190), skip / tiny-flux / stamp-size decisions
 * (:199-232, imsim/stamp_utils.py:79-155 for stars, :158-189 for galaxies), faint classification (:435-436); the profile
 * affine of a Sersic (imsim/instcat.py:498-527), the local WCS jacobian at image_pos (wcs.local), the zenith and
 * parallactic angle of PhotonDCR, the field angle of the atmospheric PSF (imsim/atmPSF.py:304).  One thread per catalog
 * source writes one ims_object_t row (phot_first 0, bf_state 0, flux_per_photon 1) and one ims_object_meta_t.
 * Everything is sqrt / division / fma arithmetic plus the spec's log, exp and sincos, so the oracle's
 * restatement (oracle/orc_catalog.c) gives the same bits:
 *   flux      phot_flux[i] when given, else the Poisson deviate of ims_fft_finish's generator keyed (seed, obj_id,
 *             pixel IMS_FLUX_PIXEL);
 *   affine    kind 0: identity; else Shear(q, beta = 90 deg - pa) [then Lens(g1, g2, mu)] (area preserving forms);
 *   local WCS the ANALYTIC jacobian of img_wcs at (x, y) in GalSim's (u west, v north) [arcsec], inverted -> winv;
 *   DCR       with Z the unit vector of the zenith and p the object's: cos z = p.Z, and (sin q, cos q) the normalised
 *             components of Z along local east and north -- trig-free;
 *   size      stamp_size[i] > 0 when given; nominal flux < tiny_flux: 32; stars: star_size[min(-floor(ln(noise_var /
 *             flux)), n_star_size - 1)] (the folding threshold rounded down to e-folds; index 0 = the default threshold);
 *             galaxies: GoodImageSize of the profile convolved with the proxy PSF, 1 / stepk^2 = (r hlr s)^2 / pi^2 +
 *             1 / dg_stepk^2 with r = gal_radius[prof_table] and s the largest singular value of the affine.  A galaxy
 *             with more than 10 photons per stamp pixel, or a stamp beyond nmax, needs the surface-brightness loop of
 *             get_good_phot_stamp_size (the bright few: host, ims_patch_stamp_sizes): meta.flags gets IMS_META_SIZE_PENDING.
 * Kinds other than 0, 1, 2 (knots, streaks, FITS stamps) are left to the host: their rows are zeroed and flagged
 * IMS_META_HOST_ROW. */
#define IMS_FLUX_PIXEL        (-7)
#define IMS_META_SIZE_PENDING 1
#define IMS_META_HOST_ROW     2
typedef struct ims_catalog {
    int64_t n;
    const double* x;              /* device arrays [n] */
    const double* y;
    const double* nominal_flux;
    const double* hlr;            /* arcsec */
    const double* q;              /* axis ratio */
    const double* pa;             /* position angle [deg] */
    const double* g1;             /* lensing: reduced shear and magnification; all three NULL = none */
    const double* g2;
    const double* mu;
    const int32_t* kind;          /* 0 point, 1 / 2 Sersic */
    const int32_t* prof_table;    /* radial table of a Sersic row (ignored for points) */
    const int32_t* sed_table;     /* NULL: sed_table_all */
    const int32_t* stamp_size;    /* NULL or entries <= 0: computed here */
    const int64_t* obj_id;        /* NULL: object i has id i */
    const int64_t* phot_flux;     /* NULL: realised here */
    uint64_t seed;
    int32_t sed_table_all, n_star_size, n_gal_radius, nmax;
    double noise_var, max_flux_simple, tiny_flux, pixel_scale, dg_stepk;
    const int32_t* star_size;     /* device [n_star_size] */
    const double* gal_radius;     /* device [n_gal_radius]: radius enclosing 1 - folding_threshold of the flux, in half-light radii */
    double zenith[3];             /* unit vector of the zenith in the frame of img_wcs (ICRS at the visit) */
    int32_t has_field, pad;       /* != 0: atm_tan_x / atm_tan_y from icrf_to_field */
    /* The surface-brightness loop of bright or oversized galaxies (get_good_phot_stamp_size, imsim/stamp_utils.py:196-220,
     * :300-354) on the device when sb_tables != 0: the galaxy's own GoodImageSize grown by 10 % until the profile's xValue on
     * the four edge midpoints and the four corners of the square is below keep_sb, capped at nmax, shrunk again while the next
     * smaller square still is (not below 64), added in quadrature to the same size of the DoubleGaussian proxy PSF
     * (psf_size_keep, an object-independent integer the host supplies), relaxed to 3 keep_sb (psf_size_keep3) when that exceeds
     * nmax.  Rows then come back WITHOUT IMS_META_SIZE_PENDING.  sb_tables == 0: the kernel flags those rows and the host
     * patches them (ims_patch_stamp_sizes). */
    const double* sb_flux;        /* [n] flux of obj_achrom (the object at the effective wavelength), or NULL: nominal_flux */
    const double* sersic_b;       /* [n_gal_radius] b_n of the Sersic index of radial table t */
    const double* sersic_norm;    /* [n_gal_radius] I(0) hlr^2 / flux = b^(2n) / (2 pi n Gamma(2n)) */
    const double* sersic_inv_n;   /* [n_gal_radius] 1 / n */
    double keep_sb;               /* sqrt(noise_var) / 8 */
    int32_t psf_size_keep, psf_size_keep3, sb_tables, pad2;
} ims_catalog_t;
typedef struct ims_object_meta {
    int64_t n_phot;
    int32_t size, flags;
} ims_object_meta_t;
/* rows_dev[n], meta_dev[n]: device outputs.  optics_dev: device ims_optics_t (img_wcs, icrf_to_field). */
int  ims_build_object_table(const ims_catalog_t* cat, const ims_optics_t* optics_dev, ims_object_t* rows_dev,
                            ims_object_meta_t* meta_dev, void* stream);
/* stamp sizes decided on the host for the rows index[k] (device arrays): bounds re-centred as in the builder */
int  ims_patch_stamp_sizes(ims_object_t* rows_dev, ims_object_meta_t* meta_dev, const int64_t* index_dev, const int32_t* size_dev,
                           int64_t n, void* stream);
/* Launch tables of a plan from the device-resident master table: dst[k] = rows[index[k]] with phot_first += first[k] (NULL: 0),
 * n_phot = count[k] (NULL: unchanged), bf_state = bf_state[k] (NULL: 0) and flags &= ~clear_flags (photon pooling clears
 * IMS_OBJ_FAINT: there the operators and the sensor see every photon, imsim/photon_pooling.py:154-159). */
int  ims_gather_rows(const ims_object_t* rows_dev, const int64_t* index_dev, const int64_t* first_dev, const int64_t* count_dev,
                     const int32_t* bf_state_dev, int32_t clear_flags, ims_object_t* dst_dev, int64_t n, void* stream);

/* ---- instance-catalog tokenizer (host code) ----
 * The `object` lines of a phosim instance catalog (grammar: imsim/instcat.py:231-297) from a text buffer: lines that do not
 * start with "object" or contain " inf " are skipped (:233), invalid objects too (magnorm >= 50, sersic2d / knots with
 * a < b, knots with npoints <= 0; :276-286).  Per kept object: num[16] = ra [deg], dec [deg], magnorm, redshift, gamma1, gamma2,
 * kappa, a, b, position angle, Sersic index rounded to 0.05 (knots: the number of points), internal Av, Rv, galactic Av, Rv,
 * 0; kind = 0 point, 1 sersic2d, 2 knots, 3 streak, 4 FITS stamp, 5 unknown type; span[6] = byte offset and length in `text`
 * of the id, the SED file name and the type token.  Returns the number of objects written (at most max_objects), or -(k + 1)
 * when object line k (counted from 0 over the lines that start with "object") has too few fields or a field that is not a
 * number -- the caller's line-by-line reader then raises what the reference raises. */
int64_t ims_parse_instcat_objects(const char* text, int64_t n_bytes, int64_t max_objects, double* num, int32_t* kind,
                                  int64_t* span);

/* ---- the inverse transforms of the FFT branch and the exchanges between GPUs, behind the C-ABI ----
 * (ims_fft_draw_batch / ims_reduce / ims_allreduce_delta.)  hipFFT and RCCL are looked up at first use -- the copy
 * already loaded in the process if there is one (a Python host brings torch's), else the ordinary library search path -- so
 * the library has no link-time dependency on either and never brings a second HIP runtime into the process.
 *
 * ims_fft_inverse: `batch` inverse real 2-D transforms of size nfft x nfft (imsim/stamp.py:502-504, GalSim's FFT draw): kbuf
 * holds the half spectra [batch][nfft][nfft / 2 + 1] complex128 (destroyed), rbuf receives [batch][nfft][nfft] float64 with
 * numpy's "backward" normalisation.  Plans are cached per (nfft, batch) for the life of the process.
 * ims_fft_inverse_raw: the same without the normalisation (hipFFT's own output: N^2 times the image), for callers whose next step
 * reads the buffer with ims_fft_params_t.rbuf_raw = 1 -- one pass less over the largest buffer of the branch (a read and a write
 * of 8 bytes per pixel).
 *
 * Communicator: rank 0 asks ims_comm_unique_id for the 128-byte id, the host hands it to every rank by its own means (a file,
 * a socket, torch.distributed's store), every rank calls ims_comm_init.  ims_reduce_image sums the ranks' f64 CCD images onto
 * `root` (imsim/lsst_image.py:353-368 summed over the shards); ims_allreduce_delta sums the delta-charge image of photon-pooling
 * mode onto every rank before a recalculation (imsim/photon_pooling.py:159).  integer_counts != 0: the values are exchanged as
 * int32 (half the bytes; exact when every value is an integer count and the sum stays below 2^31 -- ims_count_inexact adds the
 * number of values of a rank's image that are not integer counts below 2^31 / world to *bad_dev). */
int  ims_fft_inverse(double* kbuf_dev, double* rbuf_dev, int32_t nfft, int64_t batch, void* stream);
int  ims_fft_inverse_raw(double* kbuf_dev, double* rbuf_dev, int32_t nfft, int64_t batch, void* stream);
/* Make the plan of the nfft x nfft transforms of `stream` ahead of time (a process's first hipFFT plan costs seconds: the
 * library starts up and compiles its kernels at run time); may be called from another host thread while the caller goes on. */
int  ims_fft_warm(int32_t nfft, void* stream);
int  ims_comm_unique_id(void* id128);
int  ims_comm_init(const void* id128, int32_t rank, int32_t world, void** comm_out);
int  ims_comm_destroy(void* comm);
int  ims_reduce_image(void* comm, double* image_dev, int32_t* scratch_i32_dev, int64_t n, int32_t root, int32_t integer_counts, void* stream);
int  ims_allreduce_delta(void* comm, double* delta_dev, int32_t* scratch_i32_dev, int64_t n, int32_t integer_counts, void* stream);
int  ims_count_inexact(const double* image_dev, int64_t n, int32_t world, unsigned long long* bad_dev, void* stream);

/* ---- launch plans ----
 * The brighter-fatter chain of LSST_Image mode is hundreds of short dependent launches; ims_run_plan
 * issues a whole prepared list from C so the host cost per launch is one hipLaunchKernel.
 * An item goes to streams[item.stream], in list order per stream; RECORD / WAIT items order the
 * streams among each other (one wide "bulk" stream plus several concurrent chains). */
#define IMS_PLAN_RENDER     1   /* ims_shoot_accumulate(params) */
#define IMS_PLAN_SHOOT_POOL 2   /* ims_shoot_ops_photons(params, aux = photon_offset, pool) */
#define IMS_PLAN_ACC_POOL   3   /* ims_accumulate_segments(params, pool, aux = pool_start) */
#define IMS_PLAN_UPDATE     4   /* ims_sensor_update_distortions(first_slot, n_slots, aux = tile_prefix, n_tiles) */
#define IMS_PLAN_INIT       5   /* ims_sensor_init_boundaries(first_slot, n_slots, aux = tile_prefix or NULL, n_tiles) */
#define IMS_PLAN_RECORD     6   /* record library event number n_slots on the item's stream */
#define IMS_PLAN_WAIT       7   /* make the item's stream wait for library event number n_slots */
#define IMS_PLAN_ROUNDS     9   /* the per-round launches of up to IMS_MAX_CHAINS brighter-fatter chains, interleaved round by round:
                                   aux2 = HOST array of ims_chain_t, n_slots = its length */
#define IMS_MAX_CHAINS      4
#define IMS_MAX_CHAIN_EDGES 8
/* One chain class of LSST_Image mode (Renderer.plan_lsst_image): objects whose photons are already in the pool and whose
 * private pixel-boundary regions are the slots first_slot, first_slot + 1, ... in table order.  Round r lands the photons
 * [r nrecalc, (r + 1) nrecalc) of every object that has them (ims_accumulate_round), then re-superposes the distortions of the
 * regions of the objects that go on to round r + 1 (ims_sensor_update_distortions).  The table is sorted by photon count,
 * brightest first, so the objects of a round are a prefix of it.  Rounds edges[j] (j >= 1) first make the chain's stream wait
 * for library event ev_base + j: the pool slice holding their photons (IMS_PLAN_RECORD of the producer). */
typedef struct ims_chain {
    const ims_render_params_t* params;   /* host pointer; objects = the class table with the FULL photon counts */
    const ims_photons_t* pool;           /* host pointer */
    const int64_t* pool_start;           /* device: pool index of photon 0 of every object */
    const int64_t* n_phot;               /* HOST: photon count of every object (descending) */
    const int64_t* tile_prefix;          /* device: prefix sum of the 16x16-cell tiles of the slots first_slot.. (as IMS_PLAN_UPDATE aux) */
    const int64_t* tile_prefix_host;     /* HOST copy of tile_prefix */
    int32_t n_objects, first_slot, stream, nrecalc;
    int32_t n_rounds, use_tags, ev_base, n_edges;
    int32_t edges[IMS_MAX_CHAIN_EDGES];
} ims_chain_t;

typedef struct ims_plan_item {
    int32_t kind;
    int32_t stream;
    const ims_render_params_t* params;   /* host pointer */
    const ims_photons_t* pool;           /* host pointer */
    const int64_t* aux;                  /* device pointer, see kinds */
    int32_t first_slot, n_slots;
    int64_t n_tiles;
    uint32_t tag;                        /* IMS_PLAN_UPDATE: the bf_tag of the launches that deposited the charge (0 = no tile skipping) */
    uint32_t pad;
    void*    aux2;                       /* device pointer, see kinds */
} ims_plan_item_t;
/* One round of a chain class (see ims_chain_t): sensor.accumulate of the photons [round nrecalc, (round + 1) nrecalc) of the first
 * n_active objects of params->objects, photon j of object o read from the pool at pool_start[o] + j.  params->seg_prefix /
 * seg_object / n_segments are not used: every object gets ceil(nrecalc / 256) workgroups.  num_vertices: the sensor model's
 * NumVertices when the caller knows it (must equal the descriptor's; 4 and 8 select kernels with the polygon test unrolled at
 * compile time), 0 = read it from the descriptor. */
int  ims_accumulate_round(const ims_render_params_t* params, const ims_photons_t* pool, const int64_t* pool_start,
                          int32_t round, int32_t nrecalc, int32_t n_active, int32_t num_vertices, void* stream);
int  ims_run_plan(const ims_plan_item_t* items, int64_t n_items, const ims_sensor_t* sensor_dev,
                  const ims_sensor_t* sensor_host, unsigned char* changed_dev, void* const* streams, int32_t n_streams);

/* ---- LSST_Image launch planner (the object loop of LSST_ImageBuilder.buildImage, imsim/lsst_image.py:341-368, over the per-object
 *      decisions of LSST_SiliconBuilder, imsim/stamp.py:109-249, :527-573) ----
 * From the 16 bytes per object the device-built table hands back (photon count, stamp bounds, faint flag) the planner derives
 * everything a CCD's render launches need: which objects trigger pixel-boundary recalculations of their own (n_phot > nrecalc:
 * a private boundary region each, in groups that fit the scratch capacity), their chain classes by round count, the slices of
 * rounds in which their photons are shot into the converted pool, pool offsets, segment tables, slot tables, tile prefixes and
 * the item lists ims_run_plan replays -- what Renderer.plan_lsst_image builds in numpy, in one native pass.  The planner
 * allocates nothing on the device: ims_plan_lsst_image reports the sizes, the caller provides the memory (ims_plan_bind), one
 * copy moves the tables (ims_plan_upload, which also gathers the launch tables from the master table), ims_plan_run enqueues
 * the CCD.  A plan may be run any number of times.
 *
 * Objects are the entries with n_phot > 0 of the input arrays, in input order; master row of entry k = row[k] (NULL: k).
 * Streams by role, as Renderer.STREAMS: 0 top chain, 1 bulk, 2 .. 4 further chain classes. */
typedef struct ims_plan_input {
    int64_t n;                           /* entries of the arrays below */
    const int64_t* row;                  /* [n] index into the master table, or NULL (identity) */
    const int64_t* n_phot;               /* [n] photons to shoot (0: the object is skipped) */
    const int32_t* stamp;                /* [n][4] xmin, xmax, ymin, ymax of the stamp (1-based pixel indices, inclusive) */
    const uint8_t* faint;                /* [n] or NULL: non-zero = faint object (no operators, no sensor: never a chain) */
    int32_t nrecalc;                     /* photons between recalculations; 0 = no chains (no Silicon sensor) */
    int32_t n_class_rounds;              /* thresholds that cut the bright objects into chain classes (at most 3) */
    int32_t class_rounds[4];
    int32_t n_static_slots, slot_capacity;
    int64_t static_cells, scratch_cells, max_pool_photons;
    int32_t seg_size, want_realized, event_base, use_tags;
    int32_t coarse_slices;               /* 1: the photons of a chain class are shot in two launches (round 0, then the rest) instead of
                                            up to six slices of rounds -- for plans whose rounds are deferred to a joint run */
    int32_t pad;
} ims_plan_input_t;
typedef struct ims_plan_sizes {
    int64_t arena_bytes;                 /* tables of the plan (host image, page-locked by the caller, and its device copy) */
    int64_t rows_bytes;                  /* launch tables gathered on the device */
    int64_t pool_photons;                /* converted pool: 4 f64 arrays of this many photons (largest group) */
    int64_t realized_count;              /* f64 per-launch realized fluxes (want_realized) */
    int32_t n_groups, n_events;
    int64_t n_render_launches, render_photons, render_rows, render_segments;
    int64_t n_shoot_launches, shoot_photons, shoot_rows, shoot_segments;
    int64_t chain_rows, n_objects;
    int64_t n_round_launches;            /* ims_accumulate_round launches of one run (one per round and chain class) */
} ims_plan_sizes_t;
int  ims_plan_lsst_image(const ims_plan_input_t* in, void** plan_out, ims_plan_sizes_t* sizes);
/* base: the scene's launch parameters (tables, PSF, operators, optics, sensor, image, nx .. ymin; objects / seg_* are ignored).
 * arena_host: page-locked, arena_bytes; must stay untouched while the plan lives (the slot tables are copied out of it at every
 * run).  pool_dev: 4 * pool_photons doubles.  realized_dev: realized_count doubles or NULL. */
int  ims_plan_bind(void* plan, const ims_render_params_t* base, void* arena_host, void* arena_dev, void* rows_dev,
                   const ims_object_t* master_dev, double* pool_dev, double* realized_dev);
int  ims_plan_upload(void* plan, void* stream);
/* sensor_host: the caller's host copy of the sensor descriptor, whose bf_slots (host table) and n_bf_slots the run UPDATES as it
 * goes through the groups; slots_dev: the device slot table sensor_dev->bf_slots points to.  own_work_queued: 0 when nothing of
 * this renderer is queued on the streams yet (the first slot table then does not wait for the streams' earlier work). */
int  ims_plan_run(void* plan, ims_sensor_t* sensor_dev, ims_sensor_t* sensor_host, ims_bf_slot_t* slots_dev, unsigned char* changed_dev,
                  void* main_stream, void* const* streams, int32_t n_streams, int32_t own_work_queued);
/* The CCDs of a focal plane side by side (the fan-out of imsim/ccd.py:72-89: every CCD an independent LSST_Image build): with a
 * bright tail each CCD is bound by the dependent rounds of its brightest star, and chains of different CCDs on different streams
 * barely overlap on the device -- but ONE launch holding the same round of several chains costs little more than the round of
 * one.  ims_plan_run_deferred enqueues a plan like ims_plan_run but (a) does not join the plan's streams into main_stream --
 * it records where the plan's work ends on each of them, so that streams shared by role may go on with the next CCD -- and (b)
 * leaves the rounds of its chain classes out when they can run jointly: *deferred = the number of chains left (0: none -- no
 * bright object, several region groups, 8 vertices per edge: the plan ran whole).  ims_plans_run_joint runs the chains
 * first_chain .. first_chain + n_chains - 1 (0 = the top class) of the given plans -- at most 64 chains -- in lockstep on
 * joint_stream: three launches per round for all of them (pixel search, updatePixelDistortions, bounds refresh: the kernels of
 * ims_accumulate_round / ims_sensor_update_distortions with the argument blocks of all chains in a device table) and marks the
 * end of every plan's last round; plans with nothing left in that range are skipped.  Every chain left must be run by some call.
 * ims_plan_join makes `stream` wait for everything of one plan (its streams' recorded ends and its joint rounds): what
 * ims_plan_run does at its end.  The pointers handed to ims_plan_run_deferred must stay valid until the last joint run of the
 * plan has returned.  Same images as ims_plan_run, bit for bit. */
int  ims_plan_run_deferred(void* plan, ims_sensor_t* sensor_dev, ims_sensor_t* sensor_host, ims_bf_slot_t* slots_dev,
                           unsigned char* changed_dev, void* main_stream, void* const* streams, int32_t n_streams,
                           int32_t own_work_queued, int32_t* deferred);
int  ims_plans_run_joint(void* const* plans, int32_t n_plans, void* joint_stream, int32_t first_chain, int32_t n_chains);
int  ims_plan_join(void* plan, void* stream);
/* out[master row] += realized flux of every object (after ims_plan_run, on the same main stream) */
int  ims_plan_add_realized(void* plan, double* out_dev, void* stream);
int  ims_plan_destroy(void* plan);

/* ---- LSST_Flat (imsim/flat.py:133-283, area branch) ----
 * One iteration of the flat builder is: area = sensor.calculate_pixel_areas(section); temp = base * area / mean(area);
 * Poisson(temp); section += temp.  ims_sensor_pixel_areas writes the polygon area of every pixel of a slot's
 * region (row-major [ny][nx]) and adds round(area * 2^32) of every pixel to *sum_q32 (an exact, order-independent
 * sum: mean area = sum_q32 / (nx ny 2^32)).  ims_flat_add draws the Poisson deviate of
 * level * base[p] * area[p] * inv_mean_area for every pixel (stream keyed by (seed, iteration, pixel)) and adds it
 * to image[p] and, when delta is given, to delta[j*(nx+1)+i] (the slot's delta-charge image, consumed by the next
 * ims_sensor_update_distortions).  area / base may be NULL (= 1). */
int  ims_sensor_pixel_areas(const ims_sensor_t* sensor_dev, const ims_sensor_t* sensor_host, int32_t slot,
                            double* area_dev, long long* sum_q32_dev, void* stream);
int  ims_flat_add(const double* area_dev, const double* base_dev, double level, double inv_mean_area, uint64_t seed,
                  int64_t iteration, int32_t nx, int32_t ny, double* image_dev, double* delta_dev, void* stream);

/* ---- CCD readout: e-image -> amplifier segments (imsim/readout.py:413-478 CcdReadout.build_amp_images,
 *      imsim/bleed_trails.py:28-152) ----
 * Steps, in the reference's order (the host mirror is imsim_amd/readout.py):
 *   ims_readout_bleed     bleed_eimage: per column (per half column with the e2v midline stop) every run of pixels
 *                         above full well is clipped and its excess spread alternately down and up the column;
 *                         charge leaving through the bottom is lost, the top is closed.  In place on the f64
 *                         e-image (integer electron counts: every sum is exact).  `flags` is scratch of
 *                         IMS_READOUT_SCRATCH_BYTES(nx, ny) bytes (saturation flags + first/last saturated row per channel).
 *   ims_flat_add          dark current: Poisson(dark_current * dark_time) per pixel (area, base, delta = NULL).
 *   ims_readout_segments  amp_data = float32(e-image)[amp.bounds] / gain in readout order (raw_flip_x/y), plus
 *                         intra-CCD crosstalk a_i + sum_j x_ij a_j (float32, j ascending), placed at the data
 *                         section of the zero-filled raw segment (prescan / overscan), [n_amps][raw_h][raw_w].
 *   ims_readout_cte       q_i = sum_j M_ij q0_j along the parallel (axis 0: y) or serial (axis 1: x) direction with
 *                         the banded CTE matrix of readout.py:163-205 (band[i*n_band + d] = M[i][i-d], d < n_band);
 *                         binary64 dot product, j ascending, rounded to float32; src and dst must differ.
 *   ims_readout_finish    + bias level, + Gaussian read noise (stream keyed by (seed, amp, pixel)), truncation to
 *                         int32 ADU (numpy astype(int32), readout.py:504). */
#define IMS_MAX_AMPS 16
#define IMS_READOUT_SCRATCH_BYTES(nx, ny) (((int64_t)(nx) * (ny) + 15) / 16 * 16 + 16 * (int64_t)(nx))
typedef struct {
    int32_t x0, y0;            /* 0-based lower-left pixel of the amp's imaging section in the e-image */
    int32_t flip_x, flip_y;    /* raw_flip_x, raw_flip_y */
    float   gain;              /* e-/ADU */
    float   bias_level;        /* ADU */
    float   read_noise;        /* ADU, sigma */
    int32_t pad;
} ims_amp_t;
typedef struct {
    int32_t n_amps;
    int32_t seg_w, seg_h;      /* imaging section of one amp */
    int32_t raw_w, raw_h;      /* raw segment including prescan and overscan */
    int32_t data_x0, data_y0;  /* offset of the imaging section inside the raw segment */
    int32_t has_xtalk;
    ims_amp_t amps[IMS_MAX_AMPS];
    float   xtalk[IMS_MAX_AMPS * IMS_MAX_AMPS];      /* row i: x_ij */
} ims_readout_t;
int  ims_readout_bleed(double* image_dev, unsigned char* flags_dev, int32_t nx, int32_t ny, double full_well,
                       int32_t midline_stop, void* stream);
int  ims_readout_segments(const double* image_dev, int32_t nx, int32_t ny, const ims_readout_t* ro, float* seg_dev,
                          void* stream);
int  ims_readout_cte(const float* src_dev, float* dst_dev, const ims_readout_t* ro, const double* band_dev,
                     int32_t n_band, int32_t axis, void* stream);
int  ims_readout_finish(const float* seg_dev, const ims_readout_t* ro, uint64_t seed, int32_t* out_dev, void* stream);

/* ---- image helpers ---- */
int  ims_image_add(double* dst, const double* src, int64_t n, void* stream);
/* round the f64 accumulation image to the float32 CCD image the reference hands on (galsim.ImageF) */
int  ims_image_to_float(const double* src, float* dst, int64_t n, void* stream);

/* ---- timing of the dominant kernel ----
 * After ims_enable_timing(which) every launch of the selected kernel is bracketed by a hipEvent pair on its
 * stream: which = 1 k_shoot_accumulate (ims_shoot_accumulate), 2 k_shoot_photons<true> (ims_shoot_ops_photons),
 * 4 k_accumulate_round (the pixel search of a brighter-fatter round), 0 = off.  ims_last_kernel_ms returns the SUM of their durations and their count since the last query
 * (and resets the accumulation). */
int  ims_last_kernel_ms(float* ms, int* n_launches);
int  ims_enable_timing(int which);

/* ---- numerics probe used by the parity tests: evaluates the spec's elementary functions on device ----
 * which: 0 log, 1 exp, 2 sincos2pi (2 outputs), 3 atan, 4 sincos (2), 5 tanh, 6 gaussian pair of draw(seed,obj,i,slot) (2) */
/* sizeof() of the ABI structs as compiled, for binding self-checks:
 * 0 object, 1 radial_tables, 2 lin_tables, 3 psf_component, 4 op, 5 surface, 6 tansip, 7 optics, 8 bf_slot,
 * 9 sensor, 10 photons, 11 render_params, 12 plan_item, 13 atmosphere, 14 fft_object, 15 fft_params, 16 readout */
/* Host helpers: fill the derived (uniform) fields of an op / a medium from its primary parameters, so
 * that the kernels do not recompute launch-wide constants per photon.  Call them once when the op
 * chain / the optics descriptor is built; ops and media without derived fields are left untouched. */
int  ims_fill_derived_op(ims_op_t* op);
int  ims_fill_derived_medium(int32_t kind, double* c6);
/* Launch-wide constants of the optics / atmosphere / sensor descriptors (fields marked "derived"): products, squares and
 * reciprocals of wave-uniform parameters, formed once on the host with the IEEE operations the kernels would otherwise repeat
 * per photon.  Call on the HOST copy of the struct before it is uploaded. */
int  ims_fill_derived_optics(ims_optics_t* optics);
int  ims_fill_derived_atmosphere(ims_atmosphere_t* atm);
int  ims_fill_derived_sensor(ims_sensor_t* sensor);
int  ims_struct_size(int which);
int  ims_test_math(int which, const double* in_dev, double* out_dev, int64_t n, uint64_t seed, int64_t obj,
                   uint32_t slot, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IMSIM_HIP_H */
