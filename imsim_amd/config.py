"""A driver for imSim-style YAML configs restricted to the stamp-rendering hot path.

imSim is a plugin pack for GalSim's config system: a YAML `type:` string selects a registered
builder.  GalSim is not available here, so this module provides the part of that
machinery the four imSim templates exercise for THIS path: template inheritance, dotted-key
overrides, `$` (Python eval) and `@` (reference) values, the value types OpsimData / Degrees /
RADec / Eval / FormattedStr / Sequence / TreeRingCenter / TreeRingFunc, and registries holding the
same type names (LSST_Silicon, LSST_Photons, LSST_Image, LSST_PhotonPoolingImage, the PhotonOp
names, AtmosphericPSF / KolmogorovPSF / DoubleGaussianPSF, InstCatObj, ...).  Everything outside
the path (sky_model, checkpoint, vignetting, readout, truth files, FITS output) is accepted and
ignored with a note in `Process(...).ignored`.
"""
import copy
import math
import os
import types

import numpy as np
import yaml

from . import _abi, atm_psf, catalog, configs, diffraction, fft_draw, instcat, lsst_image, optics as opticsmod
from . import flat, parallel, readout, sensor as sensormod, tables, treerings, tuning
from .engine import Scene, SensorSetup, make_slots
from .lsst_image import GalSimConfigError

# ---------------- registries (names only matter: they are what YAML files refer to) ----------------
valid_stamp_types, valid_image_types, valid_photon_op_types = {}, {}, {}
valid_psf_types, valid_input_types, valid_output_types, valid_value_types = {}, {}, {}, {}
templates = {}


def RegisterStampType(name, builder): valid_stamp_types[name] = builder
def RegisterImageType(name, builder): valid_image_types[name] = builder
def RegisterPhotonOpType(name, builder): valid_photon_op_types[name] = builder
def RegisterObjectType(name, builder): valid_psf_types[name] = builder
def RegisterInputType(name, builder): valid_input_types[name] = builder
def RegisterOutputType(name, builder): valid_output_types[name] = builder
def RegisterValueType(name, builder): valid_value_types[name] = builder
def RegisterTemplate(name, path): templates[name] = path


from .camera import RAFTS, SENSORS, ITL_RAFTS, det_type_of      # noqa: E402  (one table of the focal-plane layout)


def det_name_of(det_num):
    """LsstCamSim detector number -> name (imsim/ccd.py:72-89; 94 = R22_S11)."""
    return f"{RAFTS[det_num // 9]}_{SENSORS[det_num % 9]}"


# ---------------- config loading ----------------
def _set_dotted(cfg, key, value):
    parts = key.split(".")
    d = cfg
    for p in parts[:-1]:
        if isinstance(d, list):                    # `psf.items.0.fwhm`: an index into a list
            d = d[int(p)]
            continue
        if p not in d or not isinstance(d[p], (dict, list)):
            d[p] = {}
        d = d[p]
    if isinstance(d, list):                        # `psf.items.0: {...}` replaces (or appends) a list element
        k = int(parts[-1])
        if k < len(d):
            d[k] = value
        else:
            d.append(value)
    else:
        d[parts[-1]] = value


def _merge(base, over):
    for k, v in over.items():
        if k == "template":
            continue
        if "." in k:
            _set_dotted(base, k, copy.deepcopy(v))
        elif isinstance(v, dict) and isinstance(base.get(k), dict):
            _merge(base[k], v)
        else:
            base[k] = copy.deepcopy(v)
    return base


def load_config(path_or_dict, template_dirs=(), overrides=None):
    """Read a YAML file (or dict), resolve its `template:` chain (GalSim template semantics: the
    template is the base, the file's own keys -- including dotted ones -- override it) and apply
    command-line style overrides ("a.b.c": value)."""
    if isinstance(path_or_dict, dict):
        cfg = copy.deepcopy(path_or_dict)
        here = None
    else:
        with open(path_or_dict) as f:
            cfg = yaml.safe_load(f)
        here = os.path.dirname(os.path.abspath(path_or_dict))
    if "template" in cfg:
        name = cfg["template"]
        cand = [templates.get(name)] + [os.path.join(d, name + ext) for d in ([here] if here else []) + list(template_dirs)
                                        for ext in ("", ".yaml")]
        tpath = next((c for c in cand if c and os.path.isfile(c)), None)
        if tpath is None:
            raise GalSimConfigError(f"template {name} not found (searched {template_dirs})")
        base = load_config(tpath, template_dirs)
        cfg = _merge(base, cfg)
    else:
        cfg = _merge({}, cfg)
    for k, v in (overrides or {}).items():
        _set_dotted(cfg, k, v)
    return cfg


# ---------------- value evaluation ----------------
class Evaluator:
    def __init__(self, base):
        self.base = base
        self.vars = {}

    def namespace(self):
        ns = {"np": np, "math": math, "os": os}
        ns.update(self.vars)
        return ns

    def lookup(self, path):
        d = self.base
        for p in path.split("."):
            d = d[int(p)] if isinstance(d, list) else d[p]
        return self.value(d)

    def value(self, v):
        if isinstance(v, str):
            if v.startswith("$"):
                return eval(v[1:], self.namespace())          # noqa: S307 -- config files are trusted input, as in GalSim
            if v.startswith("@"):
                return self.lookup(v[1:])
            parts = v.split()
            if len(parts) == 2 and parts[1] in ("degrees", "deg", "radians", "rad"):
                x = float(parts[0])
                return math.radians(x) if parts[1].startswith("deg") else x
            return v
        if isinstance(v, dict) and "type" in v:
            t = v["type"]
            if t in valid_value_types:
                return valid_value_types[t](v, self)
            return v
        return v

    def load_eval_variables(self, ev):
        pending = dict(ev)
        for _ in range(4):                         # variables may refer to each other
            for k in list(pending):
                try:
                    self.vars[k[1:]] = self.value(pending[k])
                    del pending[k]
                except (NameError, KeyError):
                    pass
        if pending:
            raise GalSimConfigError(f"cannot evaluate eval_variables {sorted(pending)}")


def _opsim(v, ev):
    data = ev.base.get("_opsim_data")
    if data is None:
        raise GalSimConfigError("No input opsim_data available for type OpsimData")
    if v["field"] not in data or data[v["field"]] is None:
        raise GalSimConfigError(f"OpsimData field {v['field']} not found")
    return data[v["field"]]


def _eval_type(v, ev):
    ns = ev.namespace()
    for k, val in v.items():
        if k not in ("type", "str") and len(k) > 1:
            ns[k[1:]] = ev.value(val)
    return eval(v["str"], ns)                       # noqa: S307


RegisterValueType("OpsimData", _opsim)
RegisterValueType("Degrees", lambda v, ev: math.radians(float(ev.value(v["theta"]))))
RegisterValueType("Radians", lambda v, ev: float(ev.value(v["theta"])))
RegisterValueType("RADec", lambda v, ev: (ev.value(v["ra"]), ev.value(v["dec"])))
RegisterValueType("Eval", _eval_type)
RegisterValueType("FormattedStr", lambda v, ev: str(ev.value(v["format"])) % tuple(ev.value(i) for i in v["items"]))
RegisterValueType("Sequence", lambda v, ev: int(ev.vars["_sequence_index"]) if ev.vars.get("_sequence_index") is not None
                  else int(ev.value(v.get("first", 0))))
RegisterValueType("TreeRingCenter", lambda v, ev: ev.base["_tree_rings"].get_center(ev.value(v["det_name"])))
RegisterValueType("TreeRingFunc", lambda v, ev: ev.base["_tree_rings"].get_func(ev.value(v["det_name"])))
for _name in ("InstCatWorldPos", "SkyCatWorldPos", "SkyLevel", "RowData", "Random", "XY"):
    RegisterValueType(_name, lambda v, ev: v)

for _name in ("LSST_Silicon", "LSST_Photons"):
    RegisterStampType(_name, _name)
RegisterImageType("LSST_Image", lsst_image.LSST_ImageBuilder)
RegisterImageType("LSST_PhotonPoolingImage", lsst_image.LSST_PhotonPoolingImageBuilder)
RegisterImageType("LSST_Flat", flat.LSST_FlatBuilder)
RegisterOutputType("LSST_CCD", "LSST_CCD")
for _name in ("atm_psf", "tree_rings", "instance_catalog", "opsim_data", "telescope", "sky_model", "sky_catalog", "checkpoint",
              "vignetting", "table_row"):
    RegisterInputType(_name, _name)
OUT_OF_SCOPE_INPUTS = {"sky_model", "sky_catalog", "table_row"}

# photon ops: (kind, required keys, optional keys) -- the reference's _req_params/_opt_params
PHOTON_OPS = {
    "TimeSampler": (_abi.IMS_OP_TIME_SAMPLER, {"exptime"}, {"t0"}),
    "PupilAnnulusSampler": (_abi.IMS_OP_PUPIL_ANNULUS_SAMPLER, {"R_outer"}, {"R_inner"}),
    "PhotonDCR": (_abi.IMS_OP_PHOTON_DCR, {"base_wavelength"}, {"latitude", "HA", "zenith_angle", "parallactic_angle",
                                                                  "pressure", "temperature", "H2O_pressure", "alpha", "scale_unit",
                                                                  "obj_coord", "zenith_coord"}),
    "RubinOptics": (_abi.IMS_OP_RUBIN_OPTICS, {"boresight", "camera", "det_name"}, {"shift_photons"}),
    "RubinDiffraction": (_abi.IMS_OP_RUBIN_DIFFRACTION, {"altitude", "azimuth", "latitude"},
                         {"disable_field_rotation", "stamp_center", "shift_photons"}),
    "RubinDiffractionOptics": (_abi.IMS_OP_RUBIN_DIFFRACTION_OPTICS, {"boresight", "camera", "det_name", "altitude", "azimuth"},
                               {"latitude", "disable_field_rotation", "shift_photons"}),
    "FocusDepth": (_abi.IMS_OP_FOCUS_DEPTH, {"depth"}, set()),
    "Refraction": (_abi.IMS_OP_REFRACTION, {"index_ratio"}, set()),
    "BandpassRatio": (_abi.IMS_OP_BANDPASS_RATIO, {"target_bandpass", "initial_bandpass"}, set()),
}
for _name in PHOTON_OPS:
    RegisterPhotonOpType(_name, PHOTON_OPS[_name])
for _name in ("AtmosphericPSF", "KolmogorovPSF", "DoubleGaussianPSF", "Gaussian", "Kolmogorov", "Convolve", "InstCatObj", "SkyCatObj"):
    RegisterObjectType(_name, _name)


def build_photon_ops(op_cfgs, ev, base_wavelength):
    """BuildPhotonOps: YAML list -> Scene.ops tuples, with the reference's parameter checking."""
    ops, meta = [], {}
    rad2as = 180.0 / math.pi * 3600.0
    for oc in op_cfgs:
        t = oc.get("type")
        if t not in valid_photon_op_types:
            raise GalSimConfigError(f"Invalid photon_op type {t}")
        kind, req, opt = PHOTON_OPS[t]
        for k in req:
            if k not in oc:
                raise GalSimConfigError(f"Attribute {k} is required for photon_op {t}")
        for k in oc:
            if k not in req and k not in opt and k != "type":
                raise GalSimConfigError(f"Unexpected attribute {k} found for photon_op {t}")
        p = {k: ev.value(v) for k, v in oc.items() if k != "type"}
        if t == "TimeSampler":
            ops.append((kind, 0, [float(p.get("t0", 0.0)), float(p["exptime"])]))
        elif t == "PupilAnnulusSampler":
            ops.append((kind, 0, [float(p["R_outer"]), float(p.get("R_inner", 0.0))]))
        elif t == "PhotonDCR":
            ops.append((kind, 0, [float(p["base_wavelength"]), float(p.get("pressure", 69.328)), float(p.get("temperature", 293.15)),
                                  float(p.get("H2O_pressure", 1.067)), rad2as]))
            meta["dcr"] = dict(latitude=p.get("latitude"), HA=p.get("HA"))
        elif t in ("RubinOptics", "RubinDiffraction", "RubinDiffractionOptics"):
            ops.append((kind, 0, [1.0 if p.get("shift_photons", True) else 0.0, 1.0 if p.get("disable_field_rotation", False) else 0.0]))
            if "altitude" in p:
                meta["pointing"] = dict(altitude=p["altitude"], azimuth=p["azimuth"], latitude=p.get("latitude", math.radians(-30.244633)))
        elif t == "FocusDepth":
            ops.append((kind, 0, [float(p["depth"])]))
        elif t == "Refraction":
            ops.append((kind, 0, [float(p["index_ratio"])]))
        else:                                                   # BandpassRatio (imsim/photon_ops.py:506-533)
            target, initial = p["target_bandpass"], p["initial_bandpass"]
            if not isinstance(target, tables.Bandpass) or not isinstance(initial, tables.Bandpass):
                raise GalSimConfigError("BandpassRatio: target_bandpass and initial_bandpass must be Bandpass objects")
            if "ratio" in meta:
                raise GalSimConfigError("only one BandpassRatio op per chain is supported")
            meta["ratio"] = target.ratio_table(initial)
            ops.append((kind, 0, []))
    return ops, meta


def double_gaussian_psf(fwhm, pixel_scale=0.2):
    """BuildDoubleGaussianPSF (imsim/atmPSF.py:448-486): 0.909 (G(sigma1) + 0.1 G(sigma2)) with alpha = fwhm / 2.3835,
    sigma1^2 = alpha^2 - s^2/12, sigma2^2 = 4 alpha^2 - s^2/12.  Returns the photon-op component, its k-table on the
    common q grid and the table's p0."""
    alpha = fwhm / 2.3835
    eff = pixel_scale * pixel_scale / 12.0
    s1, s2 = math.sqrt(alpha * alpha - eff), math.sqrt(4.0 * alpha * alpha - eff)
    f1 = 1.0 / 1.1
    comp = (_abi.IMS_PSF_DOUBLE_GAUSSIAN, 0, s1, 0.0, 1.0, s2, f1)
    q = np.linspace(0.0, tables.KTABLE_QMAX, tables.KTABLE_NPTS)
    p0 = tables.KTABLE_QMAX / (9.0 / s1)                       # the table spans k up to 9 / sigma1 [rad/arcsec]
    k = q / p0
    tab = f1 * np.exp(-0.5 * (k * s1) ** 2) + (1.0 - f1) * np.exp(-0.5 * (k * s2) ** 2)
    return comp, tab, p0


def build_psf(psf_cfg, ev, scene_tables):
    """psf field -> (Scene.psf list, k-space list for FFT mode, total FWHM, AtmosphericPSF or None, extra k-tables
    of the FFT-mode PSF)."""
    items = psf_cfg["items"] if psf_cfg.get("type") == "Convolve" else [psf_cfg]
    psf, kpsf, fw2, atm, extra = [], [], 0.0, None, []
    s = 1.0 / 2.3548200450309493
    for it in items:
        t = it.get("type")
        if t not in valid_psf_types:
            raise GalSimConfigError(f"Invalid psf type {t}")
        if t == "Gaussian":
            fwhm = float(ev.value(it["fwhm"])) if "fwhm" in it else float(ev.value(it["sigma"])) / s
            psf.append((_abi.IMS_PSF_GAUSSIAN, 0, fwhm * s, 0.0, 1.0))
            kpsf.append((_abi.IMS_KPSF_GAUSSIAN, 0, fwhm * s))
            fw2 += fwhm ** 2
        elif t == "Kolmogorov":                                          # plain galsim.Kolmogorov(fwhm=...)
            fa = float(ev.value(it["fwhm"]))
            psf.append((_abi.IMS_PSF_RADIAL, scene_tables["kolmogorov"], fa, 0.0, 1.0))
            kpsf.append((_abi.IMS_KPSF_KOLMOGOROV, 0, fft_draw.KOLMOGOROV_K0 / fa))
            fw2 += fa ** 2
        elif t == "KolmogorovPSF":
            p = lsst_image.get_all_params({k: ev.value(v) for k, v in it.items() if k != "type"},
                                          {"airmass": float, "rawSeeing": float, "band": str}, {})
            fa, fs = catalog.kolmogorov_gaussian_fwhm(float(p["airmass"]), float(p["rawSeeing"]), p["band"])
            psf += [(_abi.IMS_PSF_RADIAL, scene_tables["kolmogorov"], fa, 0.0, 1.0), (_abi.IMS_PSF_GAUSSIAN, 0, fs * s, 0.0, 1.0)]
            kpsf += fft_draw.kolmogorov_gaussian_kpsf(fa, fs)
            fw2 += fa ** 2 + fs ** 2
        elif t == "DoubleGaussianPSF":
            p = lsst_image.get_all_params({k: ev.value(v) for k, v in it.items() if k != "type"}, {"fwhm": float},
                                          {"pixel_scale": float})
            comp, tab, p0 = double_gaussian_psf(float(p["fwhm"]), float(p.get("pixel_scale", 0.2)))
            psf.append(comp)
            kpsf.append((_abi.IMS_KPSF_TABLE, 2 + len(extra), p0))
            extra.append(tab)
            fw2 += float(p["fwhm"]) ** 2
        elif t == "AtmosphericPSF":
            atm = ev.base["_atm_psf"]
            fw2 += atm.targetFWHM ** 2
            psf.append("ATM")
            # FFT mode swaps PhaseScreenPSF -> VonKarman and SecondKick -> Airy (make_fft_psf, psf_utils.py:94-149);
            # their k-tables follow the two Sersic profile tables of the FftDrawer
            kk, tabs = fft_draw.atmospheric_fft_kpsf(atm, atm.wlen_eff, first_table=2 + len(extra))
            kpsf += kk
            extra += tabs
        else:
            raise GalSimConfigError(f"psf type {t} is not supported on this path")
    return psf, kpsf, math.sqrt(fw2), atm, extra


class ProcessResult:
    def __init__(self):
        self.images, self.truth, self.ignored, self.det_names = [], [], [], []
        self.eimages, self.raw, self.files = [], [], []


def _process_flat(cfg, ev, image, res, device, data_dir):
    """`image.type: LSST_Flat` (imsim/flat.py): counts_per_pixel electrons per pixel through the pixel-area
    feedback of the (optional) Silicon sensor; with an `sed` item the photon branch (flat.py:237-262) runs instead,
    with the flat-in-photons SED over the tabulated band standing in for the configured SED object."""
    from .engine import Renderer
    photon_branch = "sed" in image
    if photon_branch:
        res.ignored.append("image.sed")
    builder = flat.LSST_FlatBuilder()
    img_cfg = {k: ev.value(v) for k, v in image.items() if k in flat.FLAT_REQ or k in flat.FLAT_OPT}
    if not (img_cfg.get("xsize") or img_cfg.get("size")):
        det_name = ev.value(image.get("det_name", "R22_S11"))
        img_cfg["xsize"], img_cfg["ysize"] = lsst_image.DETECTOR_SIZE[det_type_of(det_name)]
    nx, ny = builder.setup(img_cfg)
    seed = int(ev.value(image.get("random_seed", 0)))
    sens = image.get("sensor", "")
    tr, center, strength, on = None, (0.0, 0.0), 1.0, False
    if isinstance(sens, dict) and sens.get("type", "Silicon") == "Silicon":
        on = True
        strength = float(sens.get("strength", 1.0))
        func = ev.value(sens["treering_func"]) if "treering_func" in sens else None
        if func is not None:
            tr, center = func, tuple(ev.value(sens.get("treering_center", (0.0, 0.0))))
    scene = configs.scene_flat(nx, ny, seed=seed, sensor=on, treering=tr, treering_center=center, strength=strength,
                               buffer_size=builder.buffer_size)
    if photon_branch:
        scene.track_static_delta = 1
    renderer = Renderer(scene, device)
    img = builder.build_image_photons(renderer, seed=seed) if photon_branch else builder.build_image(renderer, seed=seed)
    res.images.append(img.to(renderer.torch.float32).cpu().numpy())
    res.det_names.append(str(image.get("det_name", "flat")))
    res.truth.append({"counts_per_pixel": builder.counts_per_pixel, "niter": builder.iterations()[0]})
    out = cfg.get("output", {})
    if "file_name" in out or "readout" in out:
        # flats go through the same e-image / readout outputs, without opsim data (tests/test_readout.py:124-160 of the
        # reference): the header falls back to the defaults of imsim/ccd.py:138-204
        header_vals = dict(out.get("header") or {})
        header_vals.setdefault("image_type", "FLAT")
        out = dict(out, header=header_vals)
        det_name = str(ev.value(image.get("det_name", "R22_S11")))
        _process_outputs(out, ev, res, img.to(renderer.torch.float64).contiguous(), det_name, cfg.get("_opsim_data", {}), seed)
    return res


# `eval_variables.dcamera_info` of the reference's templates loads data/<camera>_info.yaml (config/imsim-config.yaml:56-58):
# the names of the per-camera data files and the number of detectors
CAMERA_INFO = {
    "LsstCamSim": {"tree_rings_file_name": "tree_ring_parameters_2026-04-02.txt", "vignetting_file_name": "LSSTCam_vignetting_data.json",
                   "telescope_format": "LSST_%s.yaml", "bias_levels_file": "LSSTCam_bias_levels_run_13421.json",
                   "camera_name": "LsstCamSim", "ndets": 189},
    "LsstComCamSim": {"tree_rings_file_name": "tree_ring_parameters_2026-04-02.txt", "vignetting_file_name": "LSSTComCamSim_vignetting_data.json",
                      "telescope_format": "ComCam_%s.yaml", "bias_levels_file": "LSSTComCamSim_bias_levels.json",
                      "camera_name": "LsstComCamSim", "ndets": 9},
}


def camera_info(camera, data_dir=None):
    """<data_dir>/<camera>_info.yaml when it is there (an imSim data directory), else the built-in table."""
    if data_dir:
        path = os.path.join(data_dir, camera + "_info.yaml")
        if os.path.isfile(path):
            with open(path) as fobj:
                return yaml.safe_load(fobj)
    if camera not in CAMERA_INFO:
        raise GalSimConfigError(f"no camera info for {camera}")
    return dict(CAMERA_INFO[camera])


READOUT_OPT = {"camera": str, "readout_time": float, "dark_current": float, "bias_level": float, "scti": float, "pcti": float,
               "full_well": float, "read_noise": float, "bias_levels_file": str}
READOUT_IGNORE = ("file_name", "dir", "hdu", "filter", "added_keywords", "compression")


def ccd_seed(seed, det):
    """Seed of one CCD of a visit: the visit seed mixed with the detector number (splitmix64 finaliser), so that sky
    noise, dark current, read noise and the photon streams of objects in the edge overlap differ between the CCDs of a
    focal plane, on every rank, as GalSim's per-file RNG offsets make them differ in the reference."""
    m = (1 << 64) - 1
    x = (int(seed) + 0x9E3779B97F4A7C15 * (int(det) + 1)) & m
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & m
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & m
    return (x ^ (x >> 31)) >> 2


def _process_outputs(out, ev, res, image_dev, det_name, meta, seed):
    """`output` of type LSST_CCD (imsim/ccd.py:92-204) and its `readout` extra output (imsim/readout.py:535-602):
    the e-image gets the header the raw file is built from; with `output.file_name` it is written as FITS; with
    `output.readout` the CCD is read out on the GPU into 16 raw segments (kept in res.raw, written with
    `readout.file_name`)."""
    exptime = float(ev.value(out.get("exptime", meta.get("exptime") or 30.0)))
    camera_name = ev.value(out.get("camera", "LsstCamSim"))
    header_vals = {k: ev.value(v) for k, v in (out.get("header") or {}).items()}
    opsim = {k: v for k, v in meta.items() if v is not None}
    opsim.setdefault("rotSkyPos", meta.get("rotSkyPos") or 0.0)
    if opsim.get("mjd") is not None and opsim.get("fieldRA") is not None:       # HASTART / HAEND (ccd.py:186-187)
        mjd_obs = float(opsim.get("observationStartMJD", opsim["mjd"]))
        opsim.setdefault("HASTART", instcat.hour_angle(mjd_obs, opsim["fieldRA"]))
        opsim.setdefault("HAEND", instcat.hour_angle(mjd_obs + exptime / 86400.0, opsim["fieldRA"]))
    hdr = readout.eimage_header(det_name, exptime, opsim_data=opsim, header_vals=header_vals, camera=camera_name)
    eimg = readout.EImage(image_dev, hdr)
    res.eimages.append(eimg)
    out_dir = ev.value(out["dir"]) if "dir" in out else ""
    if "file_name" in out:
        fn = os.path.join(out_dir, str(ev.value(out["file_name"])))
        os.makedirs(os.path.dirname(fn) or ".", exist_ok=True)
        eimg.write(fn)
        res.files.append(fn)
    ro_cfg = out.get("readout")
    if ro_cfg is None:
        return
    for k in ro_cfg:
        if k not in READOUT_OPT and k not in READOUT_IGNORE:
            raise GalSimConfigError(f"Unexpected parameter {k} in output.readout")
    kwargs = {k: t(ev.value(ro_cfg[k])) for k, t in READOUT_OPT.items() if k in ro_cfg}
    if "added_keywords" in ro_cfg:
        kwargs["added_keywords"] = {k: str(ev.value(v)) for k, v in ro_cfg["added_keywords"].items()}
    if "bias_levels_file" in kwargs:
        cand = [kwargs["bias_levels_file"], os.path.join(configs.DATA_DIR, kwargs["bias_levels_file"])]
        if not any(os.path.isfile(c) for c in cand):
            res.ignored.append(f"output.readout.bias_levels_file ({kwargs['bias_levels_file']} not present: bias_level is used)")
            del kwargs["bias_levels_file"]
    ccd_readout = readout.CcdReadout(eimg, None, **kwargs)
    hdus = ccd_readout.prepare_hdus(seed)
    res.raw.append(hdus)
    if "file_name" in ro_cfg:
        fn = os.path.join(ev.value(ro_cfg.get("dir", out_dir)), str(ev.value(ro_cfg["file_name"])))
        os.makedirs(os.path.dirname(fn) or ".", exist_ok=True)
        # `compression: RICE_1` writes the segments tile-compressed as the reference always does (readout.py:500-510);
        # the default here is plain IMAGE extensions (same pixels and keywords, ~20 s of host time saved per CCD)
        readout.CcdReadout.write_raw_file(hdus, fn, compression=ev.value(ro_cfg.get("compression")))
        res.files.append(fn)


def Process(config, template_dirs=(), overrides=None, device="cuda:0", data_dir=None, logger=None, rank=0, world=1):
    """galsim.config.Process restricted to this path: reads inputs, then for every requested CCD
    builds the scene and runs the image builder on the GPU.  Returns a ProcessResult.

    rank / world: one process per GPU; the CCDs of the visit (output.det_num.first, output.nfiles) are dealt
    round-robin to the ranks (parallel.shard_ccds, the per-CCD fan-out of imsim/ccd.py:72-89) and need no
    exchange at all."""
    from .engine import Renderer
    cfg = load_config(config, template_dirs, overrides)
    res = ProcessResult()
    data_dir = data_dir or tuning.env("IMSIM_DATA_DIR") or configs.DATA_DIR
    ev = Evaluator(cfg)
    # an input (or any section) set to "" is switched off, the way the reference's tests and users disable template items
    inp = {k: v for k, v in cfg.get("input", {}).items() if v not in ("", None)}
    for k in inp:
        if k not in valid_input_types:
            raise GalSimConfigError(f"Invalid input type {k}")
        if k in OUT_OF_SCOPE_INPUTS:
            res.ignored.append(f"input.{k}")
    # opsim_data first: other inputs refer to it (atmPSF.py:374-383)
    if "opsim_data" in inp or "instance_catalog" in inp:
        od = inp.get("opsim_data") or {}
        fn = str(ev.value(od.get("file_name", "@input.instance_catalog.file_name")))
        if fn.endswith(".db"):                     # an OpSim database: needs the visit (opsim_data.py:60-76)
            if "visit" not in od:
                raise GalSimConfigError("input.opsim_data: an OpSim db file needs `visit`")
            cfg["_opsim_data"] = instcat.read_opsim_db(fn, int(ev.value(od["visit"])), int(ev.value(od.get("snap", 0))))
        else:
            cfg["_opsim_data"] = instcat.read_header(fn)
        for k in ("image_type", "reason"):         # passed through to the header (opsim_data.py:78-93)
            if k in od:
                cfg["_opsim_data"][k] = ev.value(od[k])
    out = cfg.get("output", {})
    if out.get("type", "LSST_CCD") not in valid_output_types:
        raise GalSimConfigError(f"Invalid output type {out.get('type')}")
    for k in ("truth", "photon_pooling_truth", "opd", "sag", "process_info", "cosmic_ray_rate"):
        if k in out:
            res.ignored.append(f"output.{k}")
    ev.vars["det_name"] = None
    ev.vars["camera_info"] = camera_info(str(ev.value(out.get("camera", "LsstCamSim"))), data_dir)
    ev.load_eval_variables({k: v for k, v in cfg.get("eval_variables", {}).items() if k not in ("dcamera_info",)})
    if "tree_rings" in inp:
        fn = ev.value(inp["tree_rings"].get("file_name", "tree_ring_parameters_2026-04-02.txt"))
        cand = [fn, os.path.join(data_dir, "tree_ring_data", fn)]
        path = next((c for c in cand if os.path.isfile(c)), None)
        if path is None:
            raise OSError("TreeRing file %s not found" % fn)
        cfg["_tree_rings"] = treerings.TreeRings(path, only_dets=inp["tree_rings"].get("only_dets"))
    if "sky_catalog" in inp:
        raise GalSimConfigError("input.sky_catalog (skyCatalogs) is outside this path: use an instance catalog (input.instance_catalog)")
    image = cfg["image"]
    itype = image.get("type", "LSST_Image")
    if itype not in valid_image_types:
        raise GalSimConfigError(f"Invalid image type {itype}")
    if itype == "LSST_Flat":
        return _process_flat(cfg, ev, image, res, device, data_dir)
    stamp_cfg = cfg.get("stamp", {})
    stype = stamp_cfg.get("type", "LSST_Silicon")
    if stype not in valid_stamp_types:
        raise GalSimConfigError(f"Invalid stamp type {stype}")
    det_num = out.get("det_num", {})
    first = int(ev.value(det_num.get("first", 0))) if isinstance(det_num, dict) else int(det_num)
    nfiles = int(ev.value(out.get("nfiles", 1)))
    meta = cfg.get("_opsim_data", {})
    band = meta.get("band", "r")
    seed = int(ev.value(image.get("random_seed", meta.get("seed", 0))))
    dets = parallel.shard_ccds(range(first, first + nfiles), rank, world)

    def prepare(det):
        """Host half of one CCD: scene, catalog, object classification -- everything up to the first GPU call of the CCD
        (the phase screens of input.atm_psf, generated on the GPU once per visit, aside)."""
        det_name = det_name_of(det)
        ev.vars["det_name"] = det_name
        ev.vars["_sequence_index"] = det                   # what `@output.det_num` (type Sequence) yields for this CCD
        builder = valid_image_types[itype]()
        img_cfg = {k: ev.value(v) if k in ("det_name", "nbatch", "nsubbatch", "nbatch_fft", "size", "xsize", "ysize", "nobjects") else v
                   for k, v in image.items()}
        if itype == "LSST_PhotonPoolingImage":
            nx, ny = builder.setup(img_cfg, stype, det_type_of(det_name))
        else:
            nx, ny = builder.setup(img_cfg, det_type_of(det_name))
        # telescope + WCS (input.telescope, image.wcs type Batoid: built by ray tracing, batoid_wcs.py:429-453)
        tel_cfg = inp.get("telescope", {})
        tel_file = ev.value(tel_cfg.get("file_name", "")) if tel_cfg else ""
        tel = opticsmod.load_batoid_yaml(tel_file) if tel_file and os.path.isfile(tel_file) else opticsmod.rubin_like_telescope(band)
        if not (tel_file and os.path.isfile(tel_file)):
            res.ignored.append("input.telescope.file_name (batoid data not present: approximate Rubin prescription)")
        rot_tel = math.radians(meta.get("rotTelPos") or 0.0)
        fp = (100.0, 0.0, (nx - 1) / 2.0 + 0.5, 0.0, 100.0, (ny - 1) / 2.0 + 0.5)
        optics = _abi.Optics()
        opticsmod.fill_optics(optics, tel, fp, rot_tel)
        ra0, dec0 = math.radians(meta.get("fieldRA") or 0.0), math.radians(meta.get("fieldDec") or 0.0)
        optics.img_wcs, optics.icrf_to_field, _ = opticsmod.build_wcs_pair(
            tel, fp, ra0, dec0, rot_sky=math.radians(meta.get("rotSkyPos") or 0.0), rot_tel_pos=rot_tel, nx=nx, ny=ny)
        # bandpass: the real tables are external data (imsim/bandpass.py); synthetic stand-in
        wl, thr = tables.synthetic_r_band()
        res.ignored.append("image.bandpass (rubin_sim throughputs not present: synthetic r-band table)")
        wl_eff = tables.effective_wavelength(wl, thr)
        ev.vars["bandpass"] = tables.Bandpass(wl, thr)
        # inputs that depend on the exposure
        if "atm_psf" in inp:
            a = {k: ev.value(v) for k, v in inp["atm_psf"].items()}
            lsst_image.get_all_params(a, {"airmass": float, "rawSeeing": float, "band": str, "boresight": None},
                                      {"t0": float, "exptime": float, "kcrit": float, "screen_size": float, "screen_scale": float,
                                       "doOpt": bool, "exponent": float, "nproc": int, "save_file": str, "_no2k": bool})
            import torch
            cfg["_atm_psf"] = atm_psf.AtmosphericPSF(float(a["airmass"]), float(a["rawSeeing"]), a["band"], seed=seed,
                                                     t0=float(a.get("t0", 0.0)), exptime=float(a.get("exptime", 30.0)),
                                                     kcrit=float(a.get("kcrit", 0.2)), screen_size=float(a.get("screen_size", 819.2)),
                                                     screen_scale=float(a.get("screen_scale", 0.1)), exponent=float(a.get("exponent", -0.3)),
                                                     device=torch.device(device))
        r2, cdf = configs.standard_tables()
        psf, kpsf, fwhm_total, atm, extra_ktables = build_psf(cfg["psf"], ev, {"kolmogorov": 2})
        if atm is not None:
            sk = atm.second_kick
            r2 = np.concatenate([r2, sk[0][None, :]])
            cdf = np.concatenate([cdf, sk[1][None, :]])
            k = psf.index("ATM")
            psf[k:k + 1] = atm.psf_components(second_kick_table_id=len(r2) - 1)
        ops, opmeta = build_photon_ops(stamp_cfg.get("photon_ops", []), ev, wl_eff)
        if "pointing" in opmeta:
            pt = opmeta["pointing"]
            diffraction.fill_optics(optics, pt["latitude"], pt["azimuth"], pt["altitude"])
        sed = tables.inverse_cdf_table(wl, thr)[None, :]
        # every CCD of the visit gets its own random streams (GalSim offsets the image-level RNG per file); the
        # atmosphere above keeps the visit seed -- all CCDs look through the same screens
        seed_ccd = ccd_seed(seed, det)
        scene = Scene(nx=nx, ny=ny, seed=seed_ccd, psf=psf, ops=ops, radial_r2=r2, radial_cdf=cdf, sed_tables=sed, optics=optics, atm=atm)
        if "ratio" in opmeta:
            scene.ratio_tables, scene.ratio_wl_min, scene.ratio_wl_step = opmeta["ratio"]
        sens = image.get("sensor", "")
        nrecalc = None
        if isinstance(sens, dict) and sens.get("type", "Silicon") == "Silicon":
            lsst_image.get_all_params(sens, {}, {"type": str, "strength": float, "index_key": str, "treering_center": None,
                                                 "treering_func": None, "name": str, "nrecalc": int, "diffusion_factor": float,
                                                 "qdist": int, "transpose": bool})
            name = sens.get("name") or sensormod.sensor_model_path(data_dir, det_type_of(det_name))
            model = sensormod.load_silicon_model(ev.value(name), strength=float(sens.get("strength", 1.0)),
                                                 nrecalc=int(sens.get("nrecalc", 10000)))
            nrecalc = model.nrecalc
            awl, al = tables.silicon_abs_length_table()
            kw = {}
            if "treering_func" in sens and "_tree_rings" in cfg:
                func = ev.value(sens["treering_func"])
                if func is not None:
                    kw = dict(tr_table=func.f, tr_table2=func.f2, tr_dr=func.dr, tr_center=ev.value(sens["treering_center"]))
            scene.sensor = SensorSetup(model=model, abs_wl=awl, abs_len=al, slots=make_slots([(1, 1, nx, ny)]),
                                       scratch_cells=int(cfg.get("_bf_scratch_cells", 24_000_000)), max_slots=8192, **kw)
            scene.track_static_delta = 1 if itype == "LSST_PhotonPoolingImage" else 0
        # catalog
        ic = inp.get("instance_catalog")
        if ic is None:
            raise GalSimConfigError("only input.instance_catalog object sources are supported on this path")
        parsed = instcat.parse_objects(ev.value(ic["file_name"]))
        sed_dir = ev.value(ic["sed_dir"]) if "sed_dir" in ic else tuning.env("SIMS_SED_LIBRARY_DIR")
        cat_file = ev.value(ic["file_name"])
        cat = instcat.to_catalog(parsed, optics.img_wcs, nx, ny, float(np.trapezoid(thr, wl)), float(meta.get("exptime") or 30.0),
                                 sort_mag=bool(ic.get("sort_mag", True)), edge_pix=int(ic.get("edge_pix", 100)),
                                 sed_dir=sed_dir, inst_dir=os.path.dirname(os.path.abspath(cat_file)), bandpass=(wl, thr))
        if cat["missing_seds"]:
            res.ignored.append(f"{len(cat['missing_seds'])} SED file(s) not found in sed_dir or beside the catalog: flat-in-photons "
                               f"SED without redshift / dust for their objects (first: {cat['missing_seds'][0]})")
        if np.any(parsed["dust"][:, 0] != 0.0):
            res.ignored.append("internal dust (Av, Rv): not applied -- nor does the reference (imsim/instcat.py:402-403)")
        if cat["sed_tables"] is not None:                      # per-object wavelength distributions behind the flat fallback table
            flat = tables.inverse_cdf_table(wl, thr, n_pts=cat["sed_tables"].shape[1])[None, :]
            scene.sed_tables = np.concatenate([flat, cat["sed_tables"]])
        configs.add_sersic_tables(scene, cat["sersic_n"])       # radial tables for every Sersic index of the catalog (:511-517)
        phot = catalog.realize_fluxes(cat["nominal_flux"], seed_ccd)
        scene.image_profiles = cat.get("images") or None          # FITS-stamp objects (instcat.py:552-561)

        def make_objects(sub, ph, scene=scene):
            objs, sizes = configs.c3b_objects(sub, ph, scene) if scene.atm is not None else configs.c3_objects(sub, ph, scene)
            return objs, sizes
        truth = {}
        vig = None
        if "vignetting" in inp:
            from .vignetting import Vignetting
            vig = Vignetting(str(ev.value(inp["vignetting"]["file_name"])), data_dir)
        max_simple = float(ev.value(stamp_cfg.get("max_flux_simple", 100)))
        chk = None
        if "checkpoint" in inp and itype == "LSST_PhotonPoolingImage":     # input.checkpoint: {file_name, dir} (checkpoint.py:19-20)
            from .checkpoint import Checkpointer
            c = {k: ev.value(v) for k, v in inp["checkpoint"].items()}
            if "file_name" not in c:
                raise GalSimConfigError("Attribute file_name is required for input.checkpoint")
            chk = Checkpointer(str(c["file_name"]), dir=c.get("dir"))
        elif "checkpoint" in inp:
            # LSST_Image: a CCD is ONE launch plan of some tens of milliseconds here, there is no state between batches to save
            note = ("input.checkpoint (LSST_Image renders a CCD in one launch plan; checkpoints are kept per photon batch "
                    "in LSST_PhotonPoolingImage mode)")
            if note not in res.ignored:
                res.ignored.append(note)
        dfft = cfg.get("_diffraction_fft")
        if dfft is None and "diffraction_fft" in stamp_cfg:
            from .diffraction_fft import DiffractionFFT
            d = {k: ev.value(v) for k, v in stamp_cfg["diffraction_fft"].items()}
            dfft = cfg["_diffraction_fft"] = DiffractionFFT(**d)        # one per visit: its stencil normalisation is shared by the CCDs
        fft_photon_ops = stamp_cfg.get("fft_photon_ops")
        if fft_photon_ops:
            # stamp.fft_photon_ops (imsim/stamp.py:493-499, :675-678): photon ops applied to FFT-drawn objects by shooting the
            # FFT image; the path draws FFT stamps without a photon stage, as the reference does whenever the list is empty
            # (its default: config/imsim-config.yaml has no fft_photon_ops)
            build_photon_ops(fft_photon_ops, ev, wl_eff)                 # parsed and validated like stamp.photon_ops
            if str(ev.value(stamp_cfg.get("draw_method", "auto"))) != "phot":
                # an object CAN take the FFT branch, and the reference would then turn its FFT image into photons, run these
                # operators over them and accumulate them again (GalSim's drawImage(method='fft', photon_ops=...)): that stage is not
                # built here, and a render that silently left it out would not be the reference's image (no
                # parsed-but-ignored key on the path)
                raise GalSimConfigError("stamp.fft_photon_ops is not supported: FFT-drawn stamps are not re-shot through photon operators "
                                        "on this path (imsim/stamp.py:493-499).  Remove the key (as config/imsim-config.yaml does) or force "
                                        "stamp.draw_method: phot")
        ctx = types.SimpleNamespace(det=det, det_name=det_name, scene=scene, builder=builder, truth=truth, seed_ccd=seed_ccd,
                                    nx=nx, ny=ny, job=None, pooling=None)
        if itype == "LSST_PhotonPoolingImage":
            ctx.pooling = dict(cat=cat, phot_flux=phot, make_objects=make_objects, max_flux_simple=max_simple, seed=seed_ccd, truth=truth,
                               fft_sb_thresh=float(ev.value(stamp_cfg.get("fft_sb_thresh", 0.0))), kpsf=kpsf, fwhm_total=fwhm_total,
                               diffraction_fft=dfft, wavelength=wl_eff, extra_ktables=extra_ktables, vignetting=vig, checkpoint=chk)
        else:
            ctx.job = builder.prepare(scene, cat, phot, make_objects,
                                      fft_sb_thresh=float(ev.value(stamp_cfg.get("fft_sb_thresh", 0.0))), max_flux_simple=max_simple,
                                      draw_method=ev.value(stamp_cfg.get("draw_method", "auto")), kpsf=kpsf, fwhm_total=fwhm_total,
                                      diffraction_fft=dfft, wavelength=wl_eff, nrecalc=nrecalc, extra_ktables=extra_ktables,
                                      vignetting=vig)
            ctx.job.want_realized = True
        # sky + noise (imsim/lsst_image.py:128-200) when a numeric sky level is configured; the Rubin sky model,
        # vignetting and fringing inputs are out of scope and reported as ignored
        sky = image.get("sky_level")
        ctx.sky = None
        if isinstance(sky, (int, float)) and not isinstance(sky, bool) and image.get("noise"):
            # sky x radial vignetting per pixel (lsst_image.py:172-176) when input.vignetting is configured, x the
            # fringing map of E2V sensors when image.apply_fringing is set (lsst_image.py:178-197)
            mult = vig(det_name, nx, ny) if vig is not None else None
            if ev.value(image.get("apply_fringing", False)) and det_type_of(det_name) == "E2V":
                from . import sky_model, wcs as wcsmod
                from .camera import make_ccd
                cx, cy = (nx + 1) / 2.0, (ny + 1) / 2.0
                pc = wcsmod.tansip_pix_to_vec(optics.img_wcs, np.array([cx]), np.array([cy]))[0]
                centre = (math.atan2(pc[1], pc[0]), math.asin(max(-1.0, min(1.0, pc[2]))))
                fr = sky_model.CCD_Fringing(true_center=centre, boresight=(ra0, dec0),
                                            seed=sky_model.sensor_seed(str(make_ccd(det_name).getSerial())), spatial_vary=True,
                                            data_dir=data_dir if os.path.isdir(os.path.join(data_dir, "fringing_data")) else None)
                xarr, yarr = np.meshgrid(np.arange(nx), np.arange(ny))
                fmap = fr.calculate_fringe_amplitude(xarr, yarr)
                mult = fmap if mult is None else mult * fmap
            # a Silicon sensor's pixels collect sky in proportion to their (tree-ring distorted) area; image.use_flux_sky_areas
            # adds the one-step brighter-fatter distortion from the flux already drawn (config/imsim-config.yaml:222-228)
            ctx.sky = dict(sky_level=float(sky), seed=seed_ccd, multiplier=mult,
                           use_flux_sky_areas=bool(ev.value(image.get("use_flux_sky_areas", False))))
        elif "sky_level" in image and "image.sky_level" not in res.ignored:
            res.ignored.append("image.sky_level")
        if ctx.job is not None:
            ctx.job.sky = ctx.sky
        return ctx

    done = {}

    def finish(ctx, renderer):
        """One CCD's image is through: truth record, host copy, `output` (e-image, readout) from the device image."""
        ev.vars["det_name"] = ctx.det_name                   # `output.file_name` and friends are evaluated per CCD
        ev.vars["_sequence_index"] = ctx.det
        renderer.synchronize()
        if ctx.job is not None:
            lsst_image.fill_truth(ctx.truth, ctx.job, ctx.job.realized.cpu().numpy())
        sub = ProcessResult()
        img = renderer.image_numpy()                         # the e-image as rendered: the readout chain bleeds the device image in place
        _process_outputs(out, ev, sub, renderer.image, ctx.det_name, meta, ctx.seed_ccd)
        done[ctx.det] = (img, ctx.truth, ctx.det_name, sub)

    # Several CCDs of LSST_Image type go through the overlapped focal-plane path (focal_plane.render_focal_plane, the per-CCD
    # fan-out of imsim/ccd.py:72-89): the host prepares the next CCD while the GPU works through the launch plans of the
    # previous ones; IMS_PROCESS_FOCAL=0 renders them one after the other (same images: every photon's stream is addressed by
    # CCD seed, object and photon index)
    overlapped = itype != "LSST_PhotonPoolingImage" and len(dets) > 1 and tuning.env("IMS_PROCESS_FOCAL", "1") != "0"
    if overlapped:
        from . import focal_plane
        ctxs = {}
        if stamp_cfg.get("draw_method", "auto") != "phot":
            focal_plane.warm_fft(device)          # hipFFT's start-up (seconds in a fresh process) beside the first CCD's host work

        def build(det):
            ctxs[det] = prepare(det)
            return ctxs[det].scene, ctxs[det].job

        focal_plane.render_focal_plane(dets, build, device=device, rank=0, world=1,
                                       concurrent=int(tuning.env("IMS_PROCESS_CONCURRENT", "3")),
                                       sink=lambda det, img: None, post=lambda det, r: finish(ctxs.pop(det), r))
    else:
        for det in dets:
            ctx = prepare(det)
            renderer = Renderer(ctx.scene, device)
            if ctx.pooling is not None:
                ctx.builder.build_image(renderer, **ctx.pooling)
                if ctx.sky is not None:
                    kw = dict(ctx.sky)
                    areas = ctx.builder.sky_pixel_areas(renderer, use_flux=kw.pop("use_flux_sky_areas"))
                    ctx.builder.add_noise(renderer, kw.pop("sky_level"), pixel_areas=areas, **kw)
            else:
                lsst_image.draw_job(renderer, ctx.job)
            finish(ctx, renderer)
    for det in dets:
        img, truth, det_name, sub = done[det]
        res.images.append(img)
        res.truth.append(truth)
        res.det_names.append(det_name)
        res.eimages += sub.eimages
        res.raw += sub.raw
        res.files += sub.files
    return res
