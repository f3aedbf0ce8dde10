"""GalSim-config adapter: `modules: [imsim_amd.galsim_plugin]` in place of `modules: [imsim]`.

Registers the type names imSim registers (imsim/stamp.py:586,747; imsim/lsst_image.py:398; imsim/photon_pooling.py;
imsim/photon_ops.py:361-451; imsim/flat.py) with galsim.config, so that `galsim config.yaml` resolves `image.type:
LSST_Image` / `LSST_PhotonPoolingImage` / `LSST_Flat`, `stamp.type: LSST_Silicon` / `LSST_Photons` and the photon-op names
to this package.  The granularity differs from the reference on purpose: GalSim's loop would build one Python GSObject
and call `draw` once per catalog source; here the IMAGE builder takes the whole CCD -- it hands the (already
template-merged) config dict to imsim_amd.config.Process, which reads the catalog, builds the object table in bulk
and renders the CCD on the GPU -- and returns the finished galsim.ImageF to GalSim's output stage.  The stamp and
photon-op builders therefore only validate and carry their parameters.

GalSim is not installed in the build container: this module imports without it (nothing is registered then) and its
wiring is exercised in tests/test_galsim_plugin.py against a stand-in of the galsim.config registration interface.
"""
import copy

try:                                            # pragma: no cover -- GalSim absent in the build image
    import galsim
    import galsim.config as gsconfig
except ImportError:
    galsim = None
    gsconfig = None

IMAGE_TYPES = ("LSST_Image", "LSST_PhotonPoolingImage", "LSST_Flat")
STAMP_TYPES = ("LSST_Silicon", "LSST_Photons")
PHOTON_OP_TYPES = ("RubinOptics", "RubinDiffractionOptics", "RubinDiffraction", "BandpassRatio")
USER_SECTIONS = ("modules", "template", "eval_variables", "input", "image", "stamp", "psf", "gal", "output")


def plain_config(base):
    """the user-level sections of GalSim's `base` dict (its bookkeeping keys start with '_' or are not sections)"""
    out = {}
    for k in USER_SECTIONS:
        if k in base:
            out[k] = copy.deepcopy({kk: vv for kk, vv in base[k].items() if not str(kk).startswith("_")}
                                   if isinstance(base[k], dict) else base[k])
    out.pop("modules", None)
    return out


def make_builders(gs, gscfg, process=None, device="cuda:0"):
    """Build the adapter classes against a galsim / galsim.config pair (the real ones, or the tests' stand-in)."""
    from . import config as our_config
    process = process or our_config.Process

    class CcdImageBuilder(gscfg.ImageBuilder):
        """image.type LSST_Image / LSST_PhotonPoolingImage / LSST_Flat: one call renders the CCD"""

        def setup(self, config, base, image_num, obj_num, ignore, logger):
            from . import lsst_image
            self.det_name = gscfg.ParseValue(config, "det_name", base, str)[0] if "det_name" in config else base.get("det_name", "R22_S11")
            xsize, ysize = lsst_image.DETECTOR_SIZE[our_config.det_type_of(self.det_name)]
            base["det_xsize"], base["det_ysize"] = xsize, ysize                 # imsim/lsst_image.py:73-74
            return int(config.get("xsize", config.get("size", xsize))), int(config.get("ysize", config.get("size", ysize)))

        def buildImage(self, config, base, image_num, obj_num, logger):
            cfg = plain_config(base)
            if "det_num" in base:                                              # set by LSST_CCD's setup for this file (imsim/ccd.py:55)
                det = int(base["det_num"])
            else:
                det = int(base.get("file_num", 0)) + int(cfg.get("output", {}).get("det_num", {}).get("first", 0)
                                                           if isinstance(cfg.get("output", {}).get("det_num"), dict) else 0)
            res = process(cfg, overrides={"output.nfiles": 1, "output.det_num": {"type": "Sequence", "first": det, "nitems": 1},
                                          "output.file_name": "", "output.readout": ""}, device=device)
            base["_imsim_amd_truth"] = res.truth[0] if res.truth else {}
            return gs.ImageF(res.images[0], xmin=1, ymin=1), 0.0

        def addNoise(self, image, config, base, image_num, obj_num, current_var, logger):
            return                                                             # sky + noise already added by Process (image.noise / sky_level)

    class CarryStampBuilder(gscfg.StampBuilder):
        """stamp.type LSST_Silicon / LSST_Photons: parameters are read by the image builder from the config"""

        def setup(self, config, base, xsize, ysize, ignore, logger):
            raise gscfg.SkipThisObject("imsim_amd renders whole CCDs in the image builder")

    class CarryPhotonOp:
        """A registered photon operator as GalSim's `photon_ops` list holds it.  The image builder reads the operators from the
        CONFIG (whole-CCD granularity); a caller that applies one itself -- op.applyTo(photon_array, local_wcs, rng), the
        PhotonOp contract of imsim/photon_ops.py:81, :304, :520 -- gets the device operator of imsim_amd.photon_ops, built on
        first use from the parameters as configured (`optics`: an _abi.Optics, else the approximate Rubin telescope of the
        bench visit)."""

        def __init__(self, name, **kw):
            self.name, self.kwargs, self._op = name, kw, None

        def device_op(self):
            if self._op is None:
                from . import photon_ops, configs
                kw = self.kwargs
                if self.name == "BandpassRatio":
                    self._op = photon_ops.BandpassRatio(kw["target_bandpass"], kw["initial_bandpass"])
                else:
                    optics = kw.get("optics") or configs.rubin_optics_struct()
                    self._op = getattr(photon_ops, self.name)(optics, shift_photons=bool(kw.get("shift_photons", False)),
                                                              stamp_center=kw.get("stamp_center"),
                                                              disable_field_rotation=bool(kw.get("disable_field_rotation", False)))
            return self._op

        def applyTo(self, photon_array, local_wcs=None, rng=None):
            return self.device_op().applyTo(photon_array, local_wcs, rng)

    class CarryPhotonOpBuilder(gscfg.PhotonOpBuilder):
        def __init__(self, name=None):
            self.name = name

        def buildPhotonOp(self, config, base, logger):
            return CarryPhotonOp(self.name or config.get("type"), **{k: v for k, v in config.items() if k != "type"})

    return CcdImageBuilder, CarryStampBuilder, CarryPhotonOpBuilder


INPUT_TYPES = ("atm_psf", "tree_rings", "instance_catalog", "opsim_data", "telescope", "sky_model", "sky_catalog", "checkpoint",
               "vignetting", "table_row")
OBJECT_TYPES = ("AtmosphericPSF", "DoubleGaussianPSF", "KolmogorovPSF", "InstCatObj", "SkyCatObj")
VALUE_TYPES = ("TreeRingCenter", "TreeRingFunc", "InstCatWorldPos", "SkyCatWorldPos", "OpsimData", "SkyLevel", "RowData")
EXTRA_OUTPUTS = ("readout", "photon_pooling_truth", "opd", "sag", "process_info")
WCS_TYPES = ("Batoid", "Dict")
TEMPLATES = ("imsim-config", "imsim-config-instcat", "imsim-config-skycat", "imsim-config-photon-pooling",
             "imsim-config-instcat-comcam", "imsim-config-skycat-comcam")


class Carrier:
    """What a registered builder of this adapter hands back where the reference hands back a live object (an input catalog,
    a GSObject, a WCS ...): the type name and the parameters as configured.  The image builder reads the CONFIG, not these
    objects -- they exist so that GalSim's input / output / value machinery finds every name a reference YAML uses."""

    def __init__(self, kind, name, **kwargs):
        self.kind, self.name, self.kwargs = kind, name, kwargs

    def __repr__(self):
        return f"imsim_amd.{self.kind}.{self.name}({self.kwargs})"


def _user_items(config):
    return {k: v for k, v in config.items() if k != "type" and not str(k).startswith("_")}


def opsim_meta(base):
    """The visit's OpSim record, read once per config from input.opsim_data (file_name: an instance catalog header or an
    OpSim db with `visit`; imsim/opsim_data.py:60-93) or, failing that, from the instance catalog itself"""
    if "_imsim_amd_opsim" not in base:
        from . import instcat
        inp = base.get("input", {})
        od = inp.get("opsim_data") or {}
        fn = od.get("file_name") if isinstance(od, dict) else None
        if isinstance(fn, str) and fn.startswith("@"):
            fn = None
        if fn is None:
            fn = (inp.get("instance_catalog") or {}).get("file_name")
        if fn is None:
            raise ValueError("OpsimData needs input.opsim_data or input.instance_catalog")
        if str(fn).endswith(".db"):
            base["_imsim_amd_opsim"] = instcat.read_opsim_db(str(fn), int(od["visit"]), int(od.get("snap", 0)))
        else:
            base["_imsim_amd_opsim"] = instcat.read_header(str(fn))
    return base["_imsim_amd_opsim"]


def make_carriers(gs, gscfg):
    """Loaders / builders for everything imSim registers beside the image, stamp and photon-op types (imsim/
    instcat.py:667-671, atmPSF.py:540-543, treerings.py:241-243, telescope_loader.py:466, opsim_data.py:377-378, sky_model.py:
    271-272, checkpoint.py:123, vignetting.py:125, table_row.py:139-140, skycat.py:302-305, ccd.py:208, readout.py, batoid_wcs.py:
    643, dict_wcs.py:55, bandpass.py:227).  Returns a dict of the classes by role."""
    from . import config as our_config

    class CarryInputLoader(gscfg.InputLoader):
        """an input item: its parameters are kept as configured (the image builder evaluates them with the whole config)"""

        def __init__(self, name):
            self.name = name
            try:
                gscfg.InputLoader.__init__(self, lambda **kw: Carrier("input", name, **kw), file_scope=True)
            except TypeError:
                pass

        def getKwargs(self, config, base, logger):
            return _user_items(config), True

        def setupImage(self, input_obj, config, base, logger=None):
            return

    def build_object(name):
        def build(config, base, ignore, gsparams, logger):
            return Carrier("object", name, **_user_items(config)), False
        return build

    def gen_value(name):
        def gen(config, base, value_type):
            if name == "OpsimData":                                  # imsim/opsim_data.py:339-375
                field = config["field"]
                meta = opsim_meta(base)
                if field not in meta or meta[field] is None:
                    raise ValueError(f"OpsimData field {field} not present in metadata")
                return value_type(meta[field]), True
            if name in ("TreeRingCenter", "TreeRingFunc"):           # imsim/treerings.py:198-239
                tr = base.get("_imsim_amd_tree_rings")
                if tr is None:
                    raise ValueError(f"{name} needs input.tree_rings")
                det = config["det_name"] if "det_name" in config else base.get("det_name")
                return (tr.get_center(det) if name == "TreeRingCenter" else tr.get_func(det)), False
            return Carrier("value", name, **_user_items(config)), False
        return gen

    class CcdOutputBuilder(gscfg.OutputBuilder):
        """output.type LSST_CCD (imsim/ccd.py:13-204): 189 files unless nfiles says otherwise, detector name from det_num,
        exposure time and detector size into `base`, one image per file from the image builder, e-image FITS writer"""

        def setup(self, config, base, file_num, logger):
            from . import lsst_image
            if "det_num" not in config:
                config["det_num"] = {"type": "Sequence", "nitems": 189}
            det_num = gscfg.ParseValue(config, "det_num", base, int)[0]
            det_name = config["only_dets"][det_num] if "only_dets" in config else our_config.det_name_of(det_num)
            base["det_num"], base["det_name"] = det_num, det_name
            self.det_name = det_name
            base["exptime"] = gscfg.ParseValue(config, "exptime", base, float)[0] if "exptime" in config else 30.0
            base["det_xsize"], base["det_ysize"] = lsst_image.DETECTOR_SIZE[our_config.det_type_of(det_name)]

        def getNFiles(self, config, base, logger=None):
            return gscfg.ParseValue(config, "nfiles", base, int)[0] if "nfiles" in config else 189

        def buildImages(self, config, base, file_num, image_num, obj_num, ignore, logger):
            image = gscfg.BuildImage(base, image_num, obj_num, logger=logger)
            return [image]

        def writeFile(self, data, file_name, config, base, logger):
            from . import readout
            hdr = readout.eimage_header(base.get("det_name", "R22_S11"), float(base.get("exptime", 30.0)),
                                        opsim_data=base.get("_imsim_amd_opsim") or {}, header_vals={})
            readout.EImage(data[0].array, hdr).write(file_name)

    class CarryExtraOutput(gscfg.ExtraOutputBuilder):
        """readout / photon_pooling_truth / opd / sag / process_info: accepted; the e-image -> raw-file chain of this package
        runs from imsim_amd.config.Process (`output.readout`), not from GalSim's extra-output hooks"""

        def initialize(self, data, scratch, config, base, logger):
            self.data, self.scratch = data, scratch

        def finalize(self, config, base, main_data, logger):
            return None

    class CarryWCSBuilder(gscfg.WCSBuilder):
        def buildWCS(self, config, base, logger):
            scale = getattr(gs, "PixelScale", None)
            return scale(0.2) if scale is not None else Carrier("wcs", config.get("type", "Batoid"), **_user_items(config))

    class CarryBandpassBuilder(gscfg.BandpassBuilder):
        def buildBandpass(self, config, base, logger):
            from . import tables
            wl, thr = tables.synthetic_r_band()
            if hasattr(gs, "Bandpass") and hasattr(gs, "LookupTable"):          # pragma: no cover -- the real GalSim
                return gs.Bandpass(gs.LookupTable(wl, thr), wave_type="nm"), False
            return tables.Bandpass(wl, thr), False

    class CarrySEDBuilder(gscfg.SEDBuilder):
        def buildSED(self, config, base, logger):
            return Carrier("sed", "InstCatSED", **_user_items(config)), False

    return dict(input=CarryInputLoader, object=build_object, value=gen_value, output=CcdOutputBuilder, extra=CarryExtraOutput,
                wcs=CarryWCSBuilder, bandpass=CarryBandpassBuilder, sed=CarrySEDBuilder)


def register(gs=None, gscfg=None, **kw):
    """Register every name; returns the list of (kind, name) registered."""
    gs = gs or galsim
    gscfg = gscfg or gsconfig
    if gs is None or gscfg is None:
        raise ImportError("galsim is not importable: imsim_amd.galsim_plugin has nothing to register with")
    image_b, stamp_b, op_b = make_builders(gs, gscfg, **kw)
    done = []
    for name in IMAGE_TYPES:
        gscfg.RegisterImageType(name, image_b())
        done.append(("image", name))
    for name in STAMP_TYPES:
        gscfg.RegisterStampType(name, stamp_b())
        done.append(("stamp", name))
    for name in PHOTON_OP_TYPES:
        gscfg.RegisterPhotonOpType(name, op_b(name))
        done.append(("photon_op", name))
    if not hasattr(gscfg, "RegisterInputType"):
        return done                                  # a stand-in of the image / stamp / photon-op interface only
    c = make_carriers(gs, gscfg)
    for name in INPUT_TYPES:
        gscfg.RegisterInputType(name, c["input"](name))
        done.append(("input", name))
    for name in OBJECT_TYPES:
        gscfg.RegisterObjectType(name, c["object"](name))
        done.append(("object", name))
    any_type = [float, int, bool, str, object, None]
    for name in VALUE_TYPES:
        gscfg.RegisterValueType(name, c["value"](name), any_type)
        done.append(("value", name))
    gscfg.RegisterOutputType("LSST_CCD", c["output"]())
    done.append(("output", "LSST_CCD"))
    for name in EXTRA_OUTPUTS:
        gscfg.RegisterExtraOutput(name, c["extra"]())
        done.append(("extra_output", name))
    for name in WCS_TYPES:
        gscfg.RegisterWCSType(name, c["wcs"]())
        done.append(("wcs", name))
    gscfg.RegisterBandpassType("RubinBandpass", c["bandpass"]())
    done.append(("bandpass", "RubinBandpass"))
    gscfg.RegisterSEDType("InstCatSED", c["sed"]())
    done.append(("sed", "InstCatSED"))
    # template aliases (imsim/templates.py:12-17) when a directory of imSim config files is given: IMSIM_CONFIG_DIR
    import os
    from . import tuning
    cfg_dir = tuning.env("IMSIM_CONFIG_DIR")
    if cfg_dir and hasattr(gscfg, "RegisterTemplate"):
        for name in TEMPLATES:
            path = os.path.join(cfg_dir, name + ".yaml")
            if os.path.isfile(path):
                gscfg.RegisterTemplate(name, path)
                done.append(("template", name))
    return done


if galsim is not None:                          # pragma: no cover
    register()
