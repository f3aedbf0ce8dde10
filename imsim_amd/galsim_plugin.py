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
        def __init__(self, **kw):
            self.kwargs = kw

    class CarryPhotonOpBuilder(gscfg.PhotonOpBuilder):
        def buildPhotonOp(self, config, base, logger):
            return CarryPhotonOp(**{k: v for k, v in config.items() if k != "type"})

    return CcdImageBuilder, CarryStampBuilder, CarryPhotonOpBuilder


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
        gscfg.RegisterPhotonOpType(name, op_b())
        done.append(("photon_op", name))
    return done


if galsim is not None:                          # pragma: no cover
    register()
