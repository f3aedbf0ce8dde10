"""Scene builders for the BASELINE.json measurement configs."""
import numpy as np

import time

from . import _abi, tables, catalog, parallel, tuning
from .engine import Scene


def cold_lsst_image(scene, objects, device):
    """Cold render of one CCD on a fresh renderer that already holds the scene: plan (host) + uploads + run, the part
    of the reference's draw loop that bench.py's replayed step leaves out (per-object setup, stamp.py:109-249)."""
    import torch
    from .engine import Renderer
    r = Renderer(scene, device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if r.native_plan_ok(objects):
        plan = r.native_plan(objects)                # ims_plan_lsst_image + bind + ONE upload + gathers (enqueued)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        plan.run()
    else:
        plan, _ = r.plan_lsst_image(objects)
        compiled = r._compile_plan(plan)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        r.execute_plan(plan, compiled)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    del plan
    del r
    return {"plan_ms": 1e3 * (t1 - t0), "cold_render_ms": 1e3 * (t2 - t0),
            "note": "fresh renderer, scene tables resident: launch-plan construction + object-table uploads + one run"}


def end_to_end_lsst_image(scene, cat, device, repeats=3):
    """Catalog columns on the HOST -> float32 CCD image on the device, everything in between timed: columns up, object
    table built on the device (flux realisation, local WCS, DCR angles, stamp sizes, classification:
    device_table.DeviceTable), 16 bytes per object back, the bright few's stamp sizes on the host, launch plan, ONE upload of
    its tables, launch tables gathered on the device, render, image rounded to float32.  A fresh renderer holds the scene
    (tables, 3.8 GB of static pixel-boundary state) as for a CCD of a visit; the first pass also pays for page-locked
    buffers and the allocator's first blocks and is reported separately."""
    import torch
    from .engine import Renderer
    from .device_table import DeviceTable
    r = Renderer(scene, device)
    torch.cuda.synchronize()
    times, parts = [], {}
    for _ in range(repeats):
        r.image.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        table = DeviceTable(r, cat, dict(VISIT))
        t1 = time.perf_counter()
        if r.native_plan_ok(table):
            plan = compiled = r.native_plan(table)
            t2 = time.perf_counter()
            plan.run()
        else:
            plan, _ = r.plan_lsst_image(table)
            compiled = r._compile_plan(plan)
            t2 = time.perf_counter()
            r.execute_plan(plan, compiled)
        img = r.image_float()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        times.append(1e3 * (t3 - t0))
        parts = {"table_ms": 1e3 * (t1 - t0), "plan_ms": 1e3 * (t2 - t1), "render_ms": 1e3 * (t3 - t2)}
        del plan, compiled, table, img
    del r
    return {"end_to_end_ms": min(times[1:]) if len(times) > 1 else times[0], "end_to_end_first_ms": times[0],
            "end_to_end_parts_last": parts,
            "end_to_end_note": "catalog columns on the host -> float32 CCD image on the device: device-built object table "
                               "(Poisson fluxes realised by the kernel), plan, render; fresh renderer with resident scene"}


def end_to_end_pooling(scene, cat, device, nbatch=10, repeats=2):
    """end_to_end_lsst_image for photon-pooling semantics (C4): catalog columns on the host -> device table -> batch shares by
    index arithmetic -> shoot and batch tables gathered on the device -> ONE shoot of all photons + the batches' pixel
    searches with the whole-CCD recalculation between them -> float32 image."""
    import torch
    from .engine import Renderer
    from .device_table import DeviceTable
    from . import photon_pooling, stamp
    r = Renderer(scene, device)
    torch.cuda.synchronize()
    times, parts = [], {}
    for _ in range(repeats):
        r.image.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        table = DeviceTable(r, cat, dict(VISIT))
        t1 = time.perf_counter()
        modes = np.where(table.n_phot.astype(np.float64) < 100.0, stamp.ProcessingMode.FAINT.value, stamp.ProcessingMode.PHOT.value)
        step = photon_pooling.prepared_image(r, table, modes, nbatch=nbatch, seed=scene.seed)
        t2 = time.perf_counter()
        step()
        img = r.image_float()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        times.append(1e3 * (t3 - t0))
        parts = {"table_ms": 1e3 * (t1 - t0), "plan_ms": 1e3 * (t2 - t1), "render_ms": 1e3 * (t3 - t2)}
        del step, table, img
    del r
    return {"end_to_end_ms": min(times[1:]) if len(times) > 1 else times[0], "end_to_end_first_ms": times[0],
            "end_to_end_parts_last": parts,
            "end_to_end_note": "catalog columns on the host -> float32 CCD image on the device, photon-pooling semantics: "
                               "device-built object table, batch shares, gathered launch tables, shoot, batches"}


def standard_tables():
    """Radial tables: 0 = Sersic n=1, 1 = Sersic n=4, 2 = Kolmogorov (units of FWHM)."""
    tabs = [tables.sersic_table(1.0), tables.sersic_table(4.0), tables.kolmogorov_table()]
    return tables.stack_radial(tabs)


def quantise_sersic_n(n):
    """`n = round(n * 20.) / 20.` (imsim/instcat.py:511-517): the Sersic index grid of the catalog reader"""
    return np.round(np.asarray(n, dtype=np.float64) * 20.0) / 20.0


def add_sersic_tables(scene, n_values):
    """Make sure the scene holds a radial (photon-shooting) table for every Sersic index in n_values (quantised to
    0.05 and limited to GalSim's range 0.3 <= n <= 6.2).  Tables 0 / 1 are always n = 1 / 4; further indices are
    appended behind whatever the scene already holds (PSF tables).  Sets scene.sersic_index {n: radial table id} and
    scene.sersic_extra_n (their order defines the k-table ids of the FFT branch, FftDrawer)."""
    n_values = quantise_sersic_n(n_values)
    n_values = n_values[n_values > 0.0]
    if np.any((n_values < 0.3) | (n_values > 6.2)):
        raise ValueError("Sersic index outside GalSim's range 0.3 <= n <= 6.2")
    index = dict(getattr(scene, "sersic_index", None) or {1.0: 0, 4.0: 1})
    extra = list(getattr(scene, "sersic_extra_n", ()))
    new = [float(n) for n in sorted(set(np.round(n_values, 2).tolist())) if float(n) not in index]
    if new:
        base = len(np.atleast_2d(scene.radial_r2))
        tabs = [tables.sersic_table(n) for n in new]
        scene.radial_r2 = np.concatenate([np.atleast_2d(scene.radial_r2), np.stack([t[0] for t in tabs])])
        scene.radial_cdf = np.concatenate([np.atleast_2d(scene.radial_cdf), np.stack([t[1] for t in tabs])])
        for j, n in enumerate(new):
            index[n] = base + j
        extra += new
    scene.sersic_index, scene.sersic_extra_n = index, tuple(extra)
    return index


def r_band_sed_table():
    wl, thr = tables.synthetic_r_band()
    return tables.inverse_cdf_table(wl, thr)[None, :], tables.effective_wavelength(wl, thr)


def scene_c2(nx=4096, ny=4096, seed=398414, airmass=1.2, raw_seeing=0.75):
    """C2: photon shooting, Kolmogorov-equivalent Gaussian atmospheric PSF (+ Gaussian system
    term), no photon ops, no sensor (image.sensor: "")."""
    r2, cdf = standard_tables()
    sed, _ = r_band_sed_table()
    fwhm_atm, fwhm_sys = catalog.kolmogorov_gaussian_fwhm(airmass, raw_seeing, "r")
    s = 1.0 / 2.3548200450309493
    psf = [(_abi.IMS_PSF_GAUSSIAN, 0, fwhm_atm * s, 0.0, 1.0),
           (_abi.IMS_PSF_GAUSSIAN, 0, fwhm_sys * s, 0.0, 1.0)]
    return Scene(nx=nx, ny=ny, seed=seed, psf=psf, ops=[], radial_r2=r2, radial_cdf=cdf, sed_tables=sed)


def _c2_objects(cat, phot, scene=None):
    return catalog.build_object_table(cat, phot)


def _c3_scene_bench():
    sc = scene_c3()
    sc.sensor.scratch_cells = 24_000_000      # private brighter-fatter regions of the bright objects
    sc.sensor.max_slots = 8192
    return sc


def _c3_cpu_scene(scene):
    """Same physics on a detector-sized image for the CPU sample (the oracle allocates like the GPU)."""
    return scene


BENCH_CONFIGS = {
    "c3": dict(
        n_objects=100000,
        workload="C3: 100k-source synthetic instcat, photon_shooting + TimeSampler/PupilAnnulusSampler/PhotonDCR/"
                 "RubinDiffractionOptics/FocusDepth/Refraction + Silicon (lsst_e2v_50_4) brighter-fatter + tree rings, "
                 "LSST_Image semantics (nrecalc=10000), Kolmogorov+Gaussian PSF, 4096x4096 CCD",
        scene=_c3_scene_bench,
        objects=None,
        make_step=lambda renderer, objects, rank=0, world=1: renderer.prepared_lsst_image(
            parallel.shard_objects(objects, rank, world)),
        cold=lambda scene, objects, device: cold_lsst_image(scene, objects, device),
        end_to_end=lambda scene, cat, device: end_to_end_lsst_image(scene, cat, device),
        timed_kernel=2,
        kernel="k_shoot_photons<2>",
        cpu_sample=20000,
        cpu_scene=_c3_cpu_scene,
        cpu_step=lambda orc, sample: orc.render_lsst_image(sample),
    ),
    "c2": dict(
        n_objects=10000,
        workload="C2: 10k-source synthetic instcat, photon_shooting, Gaussian atmPSF, Silicon sensor off, 4096x4096 CCD",
        scene=scene_c2,
        objects=_c2_objects,
        make_step=lambda renderer, objects, rank=0, world=1: renderer.prepared(parallel.shard_objects(objects, rank, world)),
        timed_kernel=1,
        kernel="k_shoot_accumulate",
        cpu_sample=10000,
        cpu_scene=lambda scene: scene,
        cpu_step=lambda orc, sample: orc.render(sample),
    ),
}


# ---------------------------------------------------------------------------------------------
# C3: the full photon chain of config/imsim-config.yaml:281-320 + Silicon sensor
# ---------------------------------------------------------------------------------------------
import functools
import math
import os

from . import optics as opticsmod, sensor as sensormod, treerings, diffraction
from .engine import SensorSetup, make_slots

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

# visit metadata = header of examples/example_instance_catalog.txt:1-20
VISIT = dict(ra=60.4927, dec=-38.1626, altitude=53.1637, azimuth=114.3933, rottelpos=40.04, rotskypos=200.0,
             seed=398414, raw_seeing=0.75, airmass=1.2489, exptime=30.0, band="r", latitude=-30.24463,
             hour_angle=-20.0)


@functools.lru_cache(maxsize=8)
def rubin_optics_struct(nx=4096, ny=4096, rottelpos=None, altitude=None, azimuth=None, ra=None, dec=None, rotskypos=None):
    """Approximate Rubin telescope + WCS pair fitted to it + spider geometry -> _abi.Optics.  Angles in degrees; the
    visit of the bench configs (VISIT) supplies whatever is not given."""
    v = dict(VISIT)
    for k, val in dict(rottelpos=rottelpos, altitude=altitude, azimuth=azimuth, ra=ra, dec=dec, rotskypos=rotskypos).items():
        if val is not None:
            v[k] = val
    tel = opticsmod.rubin_like_telescope(v["band"])
    fp = (100.0, 0.0, (nx - 1) / 2.0 + 1.0 - 0.5, 0.0, 100.0, (ny - 1) / 2.0 + 1.0 - 0.5)
    o = _abi.Optics()
    rot_tel = math.radians(v["rottelpos"])
    opticsmod.fill_optics(o, tel, fp, rot_tel)
    img_wcs, i2f, _ = opticsmod.build_wcs_pair(tel, fp, math.radians(v["ra"]), math.radians(v["dec"]),
                                               rot_sky=math.radians(v["rotskypos"]), rot_tel_pos=rot_tel,
                                               nx=nx, ny=ny)
    o.img_wcs, o.icrf_to_field = img_wcs, i2f
    diffraction.fill_optics(o, math.radians(v["latitude"]), math.radians(v["azimuth"]),
                            math.radians(v["altitude"]))
    return o


def local_wcs_inverse(img_wcs, x, y):
    """Per-object inverse local-WCS jacobian [pixels/arcsec] in GalSim's (u = +west, v = +north)
    convention, by central differences of the TAN-SIP (what `wcs.local(image_pos)` provides)."""
    from . import wcs as wcsmod
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    h = 0.5
    p0 = wcsmod.tansip_pix_to_vec(img_wcs, x, y)
    z = np.array([0.0, 0.0, 1.0])
    east = np.cross(z, p0)
    east /= np.linalg.norm(east, axis=1)[:, None]
    north = np.cross(p0, east)
    rad2as = 180.0 / math.pi * 3600.0

    def uv(xx, yy):
        p = wcsmod.tansip_pix_to_vec(img_wcs, xx, yy)
        t0 = np.sum(p * p0, axis=1)
        return -np.sum(p * east, axis=1) / t0 * rad2as, np.sum(p * north, axis=1) / t0 * rad2as
    ux1, vx1 = uv(x + h, y)
    ux0, vx0 = uv(x - h, y)
    uy1, vy1 = uv(x, y + h)
    uy0, vy0 = uv(x, y - h)
    dudx, dvdx = (ux1 - ux0) / (2 * h), (vx1 - vx0) / (2 * h)
    dudy, dvdy = (uy1 - uy0) / (2 * h), (vy1 - vy0) / (2 * h)
    det = dudx * dvdy - dudy * dvdx
    return np.stack([dvdy / det, -dudy / det, -dvdx / det, dudx / det], axis=1), p0


def dcr_angles(p0, latitude, hour_angle_center, ra_center):
    """Per-object zenith and parallactic angles (galsim.dcr.zenith_parallactic_angles restated)."""
    ra = np.arctan2(p0[:, 1], p0[:, 0])
    dec = np.arcsin(np.clip(p0[:, 2], -1, 1))
    ha = hour_angle_center + (ra_center - ra)
    cosz = np.sin(latitude) * np.sin(dec) + np.cos(latitude) * np.cos(dec) * np.cos(ha)
    zen = np.arccos(np.clip(cosz, -1, 1))
    q = np.arctan2(np.sin(ha), np.tan(latitude) * np.cos(dec) - np.sin(dec) * np.cos(ha))
    return np.tan(zen), np.sin(q), np.cos(q)


def silicon_setup(nx, ny, xmin=1, ymin=1, model_name="lsst_e2v_50_4", tree_rings=True, extra_regions=(),
                  det_name="R22_S11", nrecalc=10000, strength=1.0):
    model = sensormod.load_silicon_model(os.path.join(DATA_DIR, "sensor_models", model_name),
                                         strength=strength, nrecalc=nrecalc)
    wl, al = tables.silicon_abs_length_table()
    kw = {}
    if tree_rings:
        tr = treerings.TreeRings(os.path.join(DATA_DIR, "tree_ring_data", "tree_ring_parameters_2026-04-02_R22_S11.txt"))
        func = tr.get_func(det_name)
        kw = dict(tr_table=func.f, tr_table2=func.f2, tr_dr=func.dr, tr_center=tr.get_center(det_name))
    slots = make_slots([(xmin, ymin, nx, ny)] + list(extra_regions))
    return SensorSetup(model=model, abs_wl=wl, abs_len=al, slots=slots, **kw)


def scene_flat(nx=256, ny=256, seed=1234, sensor=True, treering=None, treering_center=(0.0, 0.0), strength=1.0,
               buffer_size=5):
    """Scene of an `LSST_Flat` image (tests/test_flats.py): a bare CCD, optionally with the default Silicon
    model (no tree rings unless `treering`, a treerings.TreeRingTable, is given).  The working image is the
    CCD plus `buffer_size` pixels on every side (image.bounds.withBorder(buffer_size), imsim/flat.py:166), so
    that the pixels that are kept see charge on all sides; flat.crop() cuts the border off again."""
    b = int(buffer_size)
    nx, ny = nx + 2 * b, ny + 2 * b
    sc = Scene(nx=nx, ny=ny, xmin=1 - b, ymin=1 - b, seed=seed)
    sc.flat_buffer = b
    sc.sed_tables, _ = r_band_sed_table()              # for the sed (photon) branch: flat-in-photons SED over the r band
    if sensor:
        model = sensormod.load_silicon_model(os.path.join(DATA_DIR, "sensor_models", "lsst_itl_50_4"), strength=strength)
        wl, al = tables.silicon_abs_length_table()
        kw = {}
        if treering is not None:
            kw = dict(tr_table=treering.f, tr_table2=treering.f2, tr_dr=treering.dr, tr_center=tuple(treering_center))
        sc.sensor = SensorSetup(model=model, abs_wl=wl, abs_len=al, slots=make_slots([(1 - b, 1 - b, nx, ny)]), **kw)
    return sc


def c3_ops(exptime=30.0, base_wavelength=620.0, shift_photons=1.0):
    """The default photon_ops chain, config/imsim-config.yaml:281-320 (r band: FocusDepth depth 0)."""
    rad2as = 180.0 / math.pi * 3600.0
    return [
        (_abi.IMS_OP_TIME_SAMPLER, 0, [0.0, exptime]),
        (_abi.IMS_OP_PUPIL_ANNULUS_SAMPLER, 0, [4.18, 2.55]),
        (_abi.IMS_OP_PHOTON_DCR, 0, [base_wavelength, 69.328, 293.15, 1.067, rad2as]),
        (_abi.IMS_OP_RUBIN_DIFFRACTION_OPTICS, 0, [shift_photons, 0.0]),
        (_abi.IMS_OP_FOCUS_DEPTH, 0, [0.0]),
        (_abi.IMS_OP_REFRACTION, 0, [3.9]),
    ]


def scene_c3(nx=4096, ny=4096, seed=398414, sensor=True, tree_rings=True, extra_regions=()):
    """C3: phot + TimeSampler/PupilAnnulusSampler/PhotonDCR/RubinDiffractionOptics/FocusDepth/
    Refraction, Kolmogorov (+) Gaussian analytic PSF (atmPSF.py:490-538), Silicon lsst_e2v_50_4 with
    brighter-fatter and the R22_S11 tree rings."""
    r2, cdf = standard_tables()
    sed, wl_eff = r_band_sed_table()
    fwhm_atm, fwhm_sys = catalog.kolmogorov_gaussian_fwhm(VISIT["airmass"], VISIT["raw_seeing"], VISIT["band"])
    psf = [(_abi.IMS_PSF_RADIAL, 2, fwhm_atm, 0.0, 1.0),
           (_abi.IMS_PSF_GAUSSIAN, 0, fwhm_sys / 2.3548200450309493, 0.0, 1.0)]
    sc = Scene(nx=nx, ny=ny, seed=seed, psf=psf, ops=c3_ops(VISIT["exptime"], wl_eff), radial_r2=r2, radial_cdf=cdf,
               sed_tables=sed, optics=rubin_optics_struct(nx, ny))
    if sensor:
        sc.sensor = silicon_setup(nx, ny, tree_rings=tree_rings, extra_regions=extra_regions)
    return sc


def c3_objects(cat, phot, scene, nrecalc=10000, bf_private=True):
    """Object table for C3: per-object local WCS, DCR angles, and a private brighter-fatter region
    for every object whose own charge will trigger a pixel-boundary recalculation (> nrecalc)."""
    winv, p0 = local_wcs_inverse(scene.optics.img_wcs, cat["x"], cat["y"])
    tanz, sinp, cosp = dcr_angles(p0, math.radians(VISIT["latitude"]), math.radians(VISIT["hour_angle"]),
                                  math.radians(VISIT["ra"]))
    objects, sizes = catalog.build_object_table(cat, phot, airmass=VISIT["airmass"], raw_seeing=VISIT["raw_seeing"],
                                                sersic_index=getattr(scene, "sersic_index", None))
    keep = phot > 0
    objects["winv"] = winv[keep]
    objects["dcr_tanz"], objects["dcr_sinp"], objects["dcr_cosp"] = tanz[keep], sinp[keep], cosp[keep]
    return objects, sizes

BENCH_CONFIGS["c3"]["objects"] = lambda cat, phot, scene: c3_objects(cat, phot, scene)


def field_angles(scene, x, y):
    """Field angle (tangent-plane coordinates, rad) of pixel positions relative to the boresight:
    the `theta` handed to atm.makePSF (imsim/atmPSF.py:304, :435)."""
    from . import wcs as wcsmod
    p = wcsmod.tansip_pix_to_vec(scene.optics.img_wcs, x, y)
    thx, thy = wcsmod.tansip_vec_to_pix(scene.optics.icrf_to_field, p)
    return thx, thy


def scene_c3b(nx=4096, ny=4096, seed=398414, sensor=True, screen_size=819.2, screen_scale=0.1, device=None, **kw):
    """C3b: C3 with the default config's 6-screen AtmosphericPSF (config/imsim-config.yaml:239-256):
    Convolve[AtmosphericPSF (phase screens + second kick), Gaussian fwhm 0.3]."""
    from . import atm_psf
    sc = scene_c3(nx=nx, ny=ny, seed=seed, sensor=sensor, **kw)
    atm = atm_psf.AtmosphericPSF(VISIT["airmass"], VISIT["raw_seeing"], VISIT["band"], seed=seed,
                                 exptime=VISIT["exptime"], screen_size=screen_size, screen_scale=screen_scale,
                                 device=device)
    r2, cdf = sc.radial_r2, sc.radial_cdf
    sk = atm.second_kick
    sc.radial_r2 = np.concatenate([r2, sk[0][None, :]])
    sc.radial_cdf = np.concatenate([cdf, sk[1][None, :]])
    sc.atm = atm
    sc.psf = atm.psf_components(second_kick_table_id=len(r2)) + [(_abi.IMS_PSF_GAUSSIAN, 0, 0.3 / 2.3548200450309493, 0.0, 1.0)]
    return sc


def c3b_objects(cat, phot, scene, **kw):
    objects, sizes = c3_objects(cat, phot, scene, **kw)
    keep = phot > 0
    thx, thy = field_angles(scene, cat["x"][keep], cat["y"][keep])
    objects["atm_tan_x"], objects["atm_tan_y"] = thx, thy
    return objects, sizes


def _c3b_scene_bench():
    import torch
    sc = scene_c3b(device=torch.device("cuda", torch.cuda.current_device()))
    sc.sensor.scratch_cells = 24_000_000
    sc.sensor.max_slots = 8192
    return sc


BENCH_CONFIGS["c3b"] = dict(BENCH_CONFIGS["c3"])
BENCH_CONFIGS["c3b"].update(
    workload=BENCH_CONFIGS["c3"]["workload"].replace("C3:", "C3b:").replace(
        "Kolmogorov+Gaussian PSF", "6-screen AtmosphericPSF (8192^2 von Karman screens) + second kick + Gaussian(0.3) PSF"),
    scene=_c3b_scene_bench,
    scene_needs_gpu=True,       # the phase screens are generated on the GPU: the CPU legs run after the GPU is up (no fork)
    objects=lambda cat, phot, scene: c3b_objects(cat, phot, scene),
    cpu_sample=8000,
)


def _c4_scene_bench():
    sc = scene_c3()
    sc.track_static_delta = 1                     # pooling mode: the whole CCD is one brighter-fatter region
    return sc


def _c4_step(renderer, objects, rank=0, world=1):
    from . import photon_pooling, stamp
    modes = stamp.classify(objects["n_phot"].astype(float), 100.0)
    return photon_pooling.prepared_image(renderer, objects, modes, nbatch=10, seed=renderer.scene.seed, rank=rank, world=world)


def _c4_cpu_step(orc, sample):
    from . import photon_pooling, stamp
    modes = stamp.classify(sample["n_phot"].astype(float), 100.0)
    photon_pooling.build_image(orc, sample, modes, nbatch=10, nsubbatch=50, seed=orc.scene.seed)


BENCH_CONFIGS["c4"] = dict(
    n_objects=1000000,
    workload="C4: 1M-source synthetic instcat, LSST_PhotonPoolingImage semantics (nbatch 10, one pixel-boundary "
             "recalculation of the whole CCD per batch, nrecalc 0), full photon-op chain, Silicon (lsst_e2v_50_4) "
             "brighter-fatter + tree rings, Kolmogorov+Gaussian PSF, 4096x4096 CCD",
    scene=_c4_scene_bench,
    objects=lambda cat, phot, scene: c3_objects(cat, phot, scene),
    make_step=_c4_step,
    end_to_end=lambda scene, cat, device: end_to_end_pooling(scene, cat, device),
    timed_kernel=2,                 # the one launch that shoots the photons of all batches into the HBM-resident pool
    kernel="k_shoot_photons<2>",
    cpu_sample=10000,
    cpu_scene=lambda scene: scene,
    cpu_step=_c4_cpu_step,
    cpu_allcore=False,          # pooling mode shares ONE sensor state between all objects: no per-object process parallelism
)


# ---------------------------------------------------------------------------------------------
# FFT branch (C1 semantics: draw_method fft, no sensor): the measurement config of the profile x PSF convolution path
# ---------------------------------------------------------------------------------------------
def stratified_catalog(n_objects, nx, ny, bright=0):
    """The C1 mix of SURVEY 8(d): 50 % point / 30 % Sersic n = 1 / 20 % n = 4 taken in catalog order from each class
    of the synthetic catalog; `bright` further objects of >= 1e6 electrons stand for the objects imSim actually sends
    down the FFT branch (stamp.py:275-277)."""
    base = catalog.synthetic_catalog(max(20 * n_objects, 2000), nx=nx, ny=ny)
    want = {0: n_objects // 2, 1: (3 * n_objects) // 10, 2: n_objects - n_objects // 2 - (3 * n_objects) // 10}
    sel = np.sort(np.concatenate([np.flatnonzero(base["kind"] == k)[:m] for k, m in want.items()]))
    cat = {k: (v[sel].copy() if isinstance(v, np.ndarray) else v) for k, v in base.items()}
    if bright:
        rng = np.random.default_rng(77)
        idx = rng.choice(len(sel), size=bright, replace=False)
        cat["nominal_flux"][idx] = 10.0 ** rng.uniform(6.0, 7.3, bright)
        cat["sb_flux"][idx] = cat["nominal_flux"][idx] / 80.0
    cat["obj_id"] = np.arange(len(sel), dtype=np.int64)
    return cat


def _fft_scene():
    return scene_c2()


def _fft_objects(cat, phot, scene):
    return catalog.build_object_table(cat, np.maximum(phot, 1))


def _fft_rows(objects):
    from . import fft_draw
    flux = objects["n_phot"].astype(np.float64)
    rows, _ = fft_draw.build_fft_objects(objects, flux, fft_draw.profile_ktable_ids(None, objects["prof_table"]))
    return rows


def _fft_kpsf():
    from . import fft_draw
    return fft_draw.kolmogorov_gaussian_kpsf(*catalog.kolmogorov_gaussian_fwhm(1.2, 0.75, "r"))


def _fft_step(renderer, objects, rank=0, world=1):
    from . import fft_draw
    mine = objects[rank::world] if world > 1 else objects
    drawer = fft_draw.FftDrawer(renderer, _fft_kpsf(), add_noise=True)
    launch = drawer.prepared(_fft_rows(mine))
    launch.photons = int(mine["n_phot"].sum())
    launch.object_rows = len(mine)
    # k-space fill: one complex128 per half-spectrum element written (16 B); the whole branch moves SURVEY 8(d)'s 24 N^2 B
    # per object of FFT size N (at the binary32 widths the survey assumed; this build computes the branch in binary64)
    launch.timed = {3: (1, 16 * launch.kspace_elements), 1: (0, 0), 2: (0, 0)}
    launch.branch_bytes = 24 * launch.pixels
    launch.keep = drawer
    return launch


BENCH_CONFIGS["fft"] = dict(
    n_objects=100,
    metric="objects/sec into one 4k x 4k LSST CCD (FFT branch)",
    catalog=lambda n, scene: stratified_catalog(n, scene.nx, scene.ny, bright=max(n // 5, 1)),
    workload="C1 mix forced down the FFT branch: 100-source stratified synthetic instcat (50 % point / 30 % Sersic n=1 / 20 % n=4, "
             "a fifth of them at 1e6..2e7 electrons), draw_method fft (k-space profile x Kolmogorov x Gaussian x pixel, inverse "
             "real 2-D FFT, clip, Poisson noise, stamp -> CCD), no sensor, 4096x4096 CCD",
    scene=_fft_scene,
    objects=_fft_objects,
    make_step=_fft_step,
    timed_kernel=3,
    kernel="k_fft_kspace_fill",
    cpu_sample=100,
    cpu_scene=lambda scene: scene,
    cpu_step=None,                      # bench.py supplies the checker's FFT branch (nothing in this package touches the oracle)
    fft_rows=_fft_rows,
    fft_kpsf=_fft_kpsf,
    cpu_allcore=False,
    parity_mode="close",
)


# FFT branch at THROUGHPUT: the objects imSim actually sends down it -- stars above fft_sb_thresh (>= 1.1e7 electrons here: the
# FFT-drawn part of the C5 visit's bright tail, log-uniform up to 1e8), drawn with the spike stencil (stamp.py:482-525) on the
# 1024^2 / 2048^2 / 4096^2 grids their stamps need, thousands of them, in chunks that share one set of buffers.  The 100-object
# `fft` config above is a launch-latency measurement (0.3 ms per step); this one moves ~1 TB per step.
def _fftx_catalog(n_objects, scene):
    cat = catalog.synthetic_catalog(n_objects, seed=20261001 + 7, nx=scene.nx, ny=scene.ny)
    rng = np.random.default_rng([20261001, 0xFF7])
    flux = 10.0 ** rng.uniform(math.log10(1.2e7), 8.0, n_objects)
    cat["nominal_flux"][:] = flux
    cat["mag"][:] = 28.13 - 2.5 * np.log10(flux / 30.0)
    cat["kind"][:] = 0
    wl, thr = tables.synthetic_r_band()
    cat["sb_flux"][:] = flux / float(np.trapezoid(thr, wl))
    return cat


def _fftx_objects(cat, phot, scene):
    return c3_objects(cat, np.maximum(phot, 1), scene)


def _fftx_rows(objects):
    from . import fft_draw
    rows, _ = fft_draw.build_fft_objects(objects, objects["n_phot"].astype(np.float64), fft_draw.profile_ktable_ids(None, objects["prof_table"]))
    return rows


def _fftx_step(renderer, objects, rank=0, world=1, spikes=False):
    from . import fft_draw
    mine = objects[rank::world] if world > 1 else objects
    v = c5_visit_fft()
    kw = dict(diffraction_fft=v["diffraction_fft"], wavelength=v["wavelength"]) if spikes else {}
    drawer = fft_draw.FftDrawer(renderer, v["kpsf"], add_noise=True, **kw)
    launch = drawer.prepared_chunked(_fftx_rows(mine))
    launch.photons = 0
    launch.object_rows = len(mine)
    launch.timed = {3: (launch.chunks, 16 * launch.kspace_elements), 1: (0, 0), 2: (0, 0)}
    launch.branch_bytes = 24 * launch.pixels
    grids = {int(k): int(n) for k, n in zip(*np.unique(_fftx_rows(mine)["nfft"], return_counts=True))}
    launch.workload_note = f"FFT grid: count {grids}, {launch.chunks} chunks of at most 2^30 grid points sharing one set of buffers"
    launch.keep2 = drawer
    return launch


BENCH_CONFIGS["fftx"] = dict(
    n_objects=5000,
    metric="objects/sec into one 4k x 4k LSST CCD (FFT branch, throughput)",
    catalog=_fftx_catalog,
    workload="FFT branch at throughput: 5 000 stars of 1.2e7 .. 1e8 electrons (what crosses fft_sb_thresh in the C5 visit), draw_method "
             "fft (k-space delta x Kolmogorov x Gaussian x pixel, inverse real 2-D FFT on the 1024^2 / 2048^2 / 4096^2 grids their stamps "
             "need, clip, Poisson noise, stamp -> CCD: the steps SURVEY 8(d)'s 24 N^2 B per object count), no sensor, 4096x4096 CCD",
    scene=lambda: scene_c3(sensor=False),
    objects=_fftx_objects,
    make_step=_fftx_step,
    timed_kernel=3,
    kernel="k_fft_kspace_fill",
    cpu_sample=2,
    cpu_scene=lambda scene: scene,
    cpu_step=None,
    fft_rows=_fftx_rows,
    fft_kpsf=lambda: c5_visit_fft()["kpsf"],
    cpu_allcore=False,
    parity_mode="close",
)
# ... and with stamp.diffraction_fft on, as the C5 visit draws them (DiffractionFFT.apply between the clip and the noise,
# stamp.py:519-522): the spike stencil of a saturated star is arithmetic, not bytes -- 60 % of this variant's step
BENCH_CONFIGS["fftxs"] = dict(BENCH_CONFIGS["fftx"])
BENCH_CONFIGS["fftxs"].update(
    metric="objects/sec into one 4k x 4k LSST CCD (FFT branch with the diffraction-spike stencil, throughput)",
    workload=BENCH_CONFIGS["fftx"]["workload"].replace("clip, Poisson noise", "clip, diffraction-spike stencil (stamp.diffraction_fft), Poisson noise"),
    make_step=lambda renderer, objects, rank=0, world=1: _fftx_step(renderer, objects, rank, world, spikes=True),
    fft_spikes=True,
)


# ---------------------------------------------------------------------------------------------
# C5: a whole focal plane -- 189 CCDs x 10 k sources, every CCD an independent LSST_Image build on its own stream
# ---------------------------------------------------------------------------------------------
N_CCD_FOCAL_PLANE = 189


class _FocalPlaneCatalog(dict):
    """the catalogs of all CCDs back to back; `ccd_offsets[k] : ccd_offsets[k + 1]` are the rows of CCD k"""
    ccd_offsets = None


class _CcdTable(np.ndarray):
    """object rows of all CCDs back to back; a slice (bench.py's CPU sample) is a plain table of one CCD"""
    ccd_offsets = None


def _c5_scene_bench():
    sc = scene_c3()
    sc.sensor.scratch_cells = 6_000_000       # private brighter-fatter regions of one CCD's ~160 bright objects
    sc.sensor.max_slots = 2048
    return sc


C5_BRIGHT_PER_CCD = 3            # stars of 1e6 .. 1e8 electrons on every CCD (the bright tail of a real pointing)
C5_FFT_SB_THRESH = 2.0e5         # stamp.fft_sb_thresh of config/imsim-config.yaml


def c5_bright_tail(cat, det, k=C5_BRIGHT_PER_CCD):
    """The first k objects of a CCD's catalog become stars with fluxes log-uniform in [1e6, 1e8] electrons (r = 16.8 .. 11.8):
    the survey's flux law stops at 2.1e6 e- (8d), a real pointing does not -- several stars brighter than r = 16 fall on
    every 13' x 13' CCD.  Above ~1.1e7 e- a star crosses fft_sb_thresh and takes the FFT branch with the spike stencil
    (imsim/stamp.py:275-308, :482-525); between 1e6 and that it is photon-shot through hundreds of brighter-fatter rounds."""
    k = min(int(k), len(cat["x"]))
    if k <= 0:
        return cat
    rng = np.random.default_rng([20261001, int(det), 0xB817])
    flux = 10.0 ** rng.uniform(6.0, 8.0, k)
    cat["nominal_flux"][:k] = flux
    cat["mag"][:k] = 28.13 - 2.5 * np.log10(flux / 30.0)
    cat["kind"][:k] = 0
    wl, thr = tables.synthetic_r_band()
    cat["sb_flux"][:k] = flux / float(np.trapezoid(thr, wl))
    return cat


def _c5_catalog(n_objects, scene, n_ccd=None, bright=C5_BRIGHT_PER_CCD):
    if n_ccd is None and tuning.env("IMS_C5_CCDS"):
        # a part of the focal plane at the full per-CCD workload (kernel traces): the first IMS_C5_CCDS CCDs
        n_ccd = int(tuning.env("IMS_C5_CCDS"))
        n_objects = n_ccd * (n_objects // N_CCD_FOCAL_PLANE)
    n_ccd = N_CCD_FOCAL_PLANE if n_ccd is None else int(n_ccd)
    per = max(n_objects // n_ccd, 1)
    parts = [c5_bright_tail(catalog.synthetic_catalog(per, seed=20261001 + det, nx=scene.nx, ny=scene.ny), det, bright)
             for det in range(n_ccd)]
    cat = _FocalPlaneCatalog({k: np.concatenate([p[k] for p in parts]) for k in parts[0]})
    cat["obj_id"] = np.concatenate([p["obj_id"] for p in parts])           # ids restart per CCD, like object numbers per file
    cat.ccd_offsets = np.arange(n_ccd + 1) * per
    return cat


def _c5_objects(cat, phot, scene):
    """LSST_SiliconBuilder.setup for every CCD of the visit (one vectorised pass; the CCDs share the analytic WCS of the
    bench scene, their catalogs, seeds and photon streams differ)."""
    objects, sizes = c3_objects(cat, phot, scene)
    kept = np.concatenate([[0], np.cumsum(phot > 0)])
    t = objects.view(_CcdTable)
    t.ccd_offsets = kept[cat.ccd_offsets]
    t.cat, t.phot, t.cat_offsets = cat, phot, np.asarray(cat.ccd_offsets)
    # which rows the per-CCD build will FFT-draw (their Poisson fluxes are not photons shot): bench.py's photon count
    from . import lsst_image
    v = c5_visit_fft()
    is_fft = lsst_image.LSST_ImageBuilderBase._use_fft(cat, cat["nominal_flux"], v["fwhm_total"], C5_FFT_SB_THRESH, v["kpsf"], ())
    t.fft_mask = (is_fft & (np.asarray(cat["kind"]) < 3))[phot > 0]
    return t, sizes


def c5_visit_fft():
    """What the FFT branch of every CCD of the bench visit shares: k-space PSF of the Kolmogorov (+) Gaussian atmosphere, total
    FWHM, the `stamp.diffraction_fft` block of config/imsim-config.yaml:269-279 (its stencil normalisation is computed once)."""
    from . import fft_draw
    from .diffraction_fft import DiffractionFFT
    fwhm_atm, fwhm_sys = catalog.kolmogorov_gaussian_fwhm(VISIT["airmass"], VISIT["raw_seeing"], VISIT["band"])
    dfft = _C5_DFFT.get("dfft")
    if dfft is None:
        dfft = _C5_DFFT["dfft"] = DiffractionFFT(exptime=VISIT["exptime"], azimuth=math.radians(VISIT["azimuth"]),
                                                 altitude=math.radians(VISIT["altitude"]), rotTelPos=math.radians(VISIT["rottelpos"]))
    return dict(kpsf=fft_draw.kolmogorov_gaussian_kpsf(fwhm_atm, fwhm_sys), fwhm_total=math.hypot(fwhm_atm, fwhm_sys),
                diffraction_fft=dfft, wavelength=r_band_sed_table()[1])


_C5_DFFT = {}


def c5_job(scene, cat, phot, rows, nrecalc=10000, fft_sb_thresh=C5_FFT_SB_THRESH, max_flux_simple=100.0, visit=None):
    """The per-CCD build of the reference for one CCD of the bench visit (imsim/lsst_image.py:276-395): every object classified
    FFT / photons / faint by stamp.py:275-336 with the default config's fft_sb_thresh -> lsst_image.CcdJob.  rows: the CCD's
    prebuilt OBJECT_DTYPE rows (objects with photons, catalog order)."""
    from . import lsst_image
    v = visit or c5_visit_fft()
    b = lsst_image.LSST_ImageBuilder()
    b.setup({"det_name": "R22_S11", "xsize": scene.nx, "ysize": scene.ny})
    return b.prepare(scene, cat, phot, lambda sub, ph: (np.array(rows, dtype=_abi.OBJECT_DTYPE), None), fft_sb_thresh=fft_sb_thresh,
                     max_flux_simple=max_flux_simple, kpsf=v["kpsf"], fwhm_total=v["fwhm_total"], diffraction_fft=v["diffraction_fft"],
                     wavelength=v["wavelength"], nrecalc=nrecalc)


def c5_cpu_sample(objects, scene, limit=0):
    """bench.py's CPU sample of a focal plane: CCD 0 whole, as the job the GPU runs (FFT objects, photon-shot objects).
    limit > 0: a bounded part of that CCD for the short runs (bench.py's extra.configs) -- its first `limit` catalog rows of
    at most 3e5 photons plus its FFT-drawn objects on grids of at most 2048^2, again as ONE job of the per-CCD build."""
    a, b = int(objects.ccd_offsets[0]), int(objects.ccd_offsets[1])
    ca, cb = int(objects.cat_offsets[0]), int(objects.cat_offsets[1])
    sub = {k: v[ca:cb] for k, v in objects.cat.items() if isinstance(v, np.ndarray)}
    rows = np.asarray(objects[a:b])
    phot = objects.phot[ca:cb]
    job = c5_job(scene, sub, phot, rows)
    if limit and limit > 0:
        kept = np.flatnonzero(phot > 0)                              # catalog rows that have an object row
        take = np.zeros(len(phot), dtype=bool)
        small = np.flatnonzero((phot > 0) & (phot <= 3.0e5) & ~np.isin(np.arange(len(phot)), kept[job.host["fft_rows"]]))
        take[small[:int(limit)]] = True
        if job.n_fft:
            take[kept[job.fft_index[np.asarray(job.fft_rows["nfft"]) <= 2048]]] = True
        sub = {k: v[take] for k, v in sub.items()}
        rows = rows[take[kept]]
        phot = phot[take]
        job = c5_job(scene, sub, phot, rows)
    sample = rows.view(_CcdTable)
    sample.job = job
    return sample


def _c5_step(renderer, objects, rank=0, world=1, concurrent=None):
    """One step = every CCD this rank owns (CCD i -> rank i mod world, no exchange), each through a FRESH renderer: scene
    tables, the CCD's static pixel-boundary state, then the per-CCD build -- FFT-drawn objects first, launch plan of the
    photon-shot ones, ONE arena upload, the run -- and the float32 image back on the host: what
    `focal_plane.render_focal_plane` does per CCD.  Up to `concurrent` CCDs are in flight on the device's streams."""
    import copy
    import zlib
    from . import focal_plane, lsst_image
    from .config import ccd_seed
    offs = getattr(objects, "ccd_offsets", None)
    if offs is None:                       # one CCD (bench.py's parity leg): the CCD of the given renderer
        job = getattr(objects, "job", None)
        if job is None:
            return renderer.prepared_lsst_image(objects)
        return lambda: lsst_image.draw_job(renderer, job)
    # four top-chain streams, four CCDs in flight: with the bright tail a CCD is bound by the rounds of its brightest star
    # (hundreds of dependent rounds of ~50 us), so more chains side by side pay (tools/dbg/r4_c5_sweep.sh: 25.1 -> 22.5 ms per CCD)
    tuning.setdefault("IMS_FOCAL_TOPS", "4")
    # the hipFFT plans of the FFT-drawn objects are made in the background while the jobs of the CCDs are built below (a fresh
    # process's first plan costs seconds: focal_plane.warm_fft)
    launch_warm = focal_plane.warm_fft(str(renderer.device))
    if concurrent is None:
        concurrent = int(tuning.env("IMS_FOCAL_CONCURRENT", "4"))
    base = renderer.scene
    cat, phot, coffs = objects.cat, objects.phot, objects.cat_offsets
    mine = parallel.shard_ccds(list(range(len(offs) - 1)), rank, world)
    nrecalc = 10000
    visit = c5_visit_fft()
    visit["diffraction_fft"].constants(visit["wavelength"])          # once per visit, outside the timed region
    jobs = {}
    for det in mine:
        sub = {k: v[coffs[det]:coffs[det + 1]] for k, v in cat.items() if isinstance(v, np.ndarray)}
        jobs[det] = c5_job(base, sub, phot[coffs[det]:coffs[det + 1]], np.asarray(objects[offs[det]:offs[det + 1]]), nrecalc, visit=visit)

    def build(det):
        sc = copy.copy(base)
        sc.seed = base.seed if det == 0 else ccd_seed(base.seed, det)
        return sc, jobs[det]

    def sink(det, image):
        # stands for the FITS writer: the image is on the host; a coarse checksum proves it arrived
        launch.checksums[det] = float(image[::64, ::64].sum())
        if want_hashes:
            launch.hashes[det] = zlib.crc32(np.ascontiguousarray(image).view(np.uint8).reshape(-1))

    want_hashes = bool(tuning.env("IMS_BENCH_DUMP"))      # bench.py's test hook: the CRC of every CCD's float32 image

    def launch():
        if launch_warm is not None:
            launch_warm.join()
        launch.checksums = {}
        launch.hashes = {}
        focal_plane.render_focal_plane(list(range(len(offs) - 1)), build, device=str(renderer.device), rank=rank, world=world,
                                       concurrent=concurrent, nrecalc=nrecalc, sink=sink,
                                       chain_hint=lambda det: int(jobs[det].objects["n_phot"].max()) if len(jobs[det].objects) else 0)
    n_phot = np.concatenate([jobs[d].objects["n_phot"] for d in mine]) if mine else np.zeros(0, dtype=np.int64)
    ordinary = n_phot[n_phot <= nrecalc]
    launch.photons = int(n_phot.sum())
    launch.object_rows = len(n_phot)
    # 1, the fused launch of the ordinary objects (one per CCD): f64 image RMW 16 B per photon + one 256-B row per object;
    # 2, the pool launches of the bright objects (the kernel with the largest summed time of a step, profiles/round4_c5_kernel_stats.txt):
    # every pooled photon's 32-byte converted record written + one 256-B row per object and launch
    from .engine import plan_sizes
    z = [plan_sizes(renderer, jobs[d].objects, nrecalc) for d in mine]
    launch.timed = {1: (len(mine), int(ordinary.sum()) * 16 + len(ordinary) * 256),
                    2: (int(sum(v.n_shoot_launches for v in z)), int(sum(v.shoot_photons * 32 + v.shoot_rows * 256 for v in z)))}
    launch.timed_waves = {1: 4 * int(sum(v.render_segments for v in z)), 2: 4 * int(sum(v.shoot_segments for v in z))}
    launch.n_ccds = len(mine)
    launch.hashes = {}
    n_fft = sum(jobs[d].n_fft for d in mine)
    nfft = np.concatenate([jobs[d].fft_rows["nfft"] for d in mine if jobs[d].n_fft] or [np.zeros(0, dtype=np.int64)]).astype(np.int64)
    bright_phot = int(np.count_nonzero(n_phot >= 1_000_000))
    launch.n_fft = n_fft
    grids = {int(k): int(v) for k, v in zip(*np.unique(nfft, return_counts=True))}
    flags = np.concatenate([jobs[d].objects["flags"] for d in mine]) if mine else np.zeros(0, dtype=np.int32)
    launch.workload_note = (f"{n_fft} objects FFT-drawn with the spike stencil (FFT grid: count {grids}), {len(n_phot)} photon-shot "
                            f"({bright_phot} of them stars of >= 1e6 photons through the brighter-fatter rounds, "
                            f"{int(np.count_nonzero(flags & _abi.IMS_OBJ_FAINT))} faint) on this rank's {len(mine)} CCDs")
    return launch


BENCH_CONFIGS["c5"] = dict(
    n_objects=N_CCD_FOCAL_PLANE * 10000,
    workload="C5: 189-CCD focal plane, 10k-source synthetic catalog per CCD (own catalog, seed and photon streams; three stars of "
             "1e6 .. 1e8 e- per CCD as the bright tail of a real pointing), every CCD the reference's per-CCD build "
             "(imsim/lsst_image.py:276-395) through a fresh renderer on the device's streams: draw_method by the reference's rule "
             "(FFT above 1e6 e- AND a peak surface brightness above fft_sb_thresh = 2e5, config/imsim-config.yaml) -- FFT-drawn "
             "objects first (k-space fill, inverse FFT, diffraction-spike stencil, Poisson noise), then the photon-shot ones (C3 "
             "physics: full photon-op chain, Silicon brighter-fatter + tree rings), CCD i -> GPU i mod N, image back on the host; "
             "host object tables are inputs",
    scene=_c5_scene_bench,
    catalog=_c5_catalog,
    objects=_c5_objects,
    make_step=_c5_step,
    reduce=False,                         # CCDs are independent: no exchange between the ranks
    focal=True,                           # every CCD through a fresh renderer on the device's four streams by role
    timed_kernel=2,                       # the pool launches of the bright objects: the largest summed kernel time of a C5 step
    kernel="k_shoot_photons<2>",
    cpu_sample=10000,
    cpu_sample_of=c5_cpu_sample,          # CCD 0 whole, as the job the GPU runs
    cpu_scene=_c3_cpu_scene,
    cpu_step=None,                        # bench.py runs the job on the checker (FFT objects, then the photon-shot ones)
    cpu_allcore=False,
    parity_mode="close",                  # FFT stamps: library transforms agree to ~1e-11 of the peak (see the fft config)
    metric="objects/sec over a 189-CCD focal plane (photon-shooting path, one CCD per stream)",
    sharding="CCD i -> rank i mod N, no exchange",
)
