"""The GalSim-config adapter against a stand-in of the galsim.config registration interface (GalSim itself is not
installed here): every imSim type name of the hot path is registered, and the image builder hands the merged config to
imsim_amd.config.Process for exactly the CCD GalSim asks for."""
import types

import numpy as np
import pytest

from imsim_amd import galsim_plugin


def _fake_galsim(full=True):
    """a stand-in of galsim.config's registration interface: base classes, Register* functions, ParseValue (with registered
    value types and the Sequence of output.det_num) and a minimal ProcessInput (every input item through its loader)"""
    reg = {k: {} for k in ("image", "stamp", "photon_op", "input", "object", "value", "output", "extra", "wcs", "bandpass", "sed", "template")}

    def parse_value(config, key, base, typ):
        v = config[key]
        if isinstance(v, dict) and "type" in v:
            if v["type"] == "Sequence":
                return typ(int(v.get("first", 0)) + int(base.get("file_num", 0))), False
            gen, _ = reg["value"][v["type"]]
            return gen(v, base, typ)
        return typ(v), True

    def process_input(config, logger=None):
        objs = config.setdefault("_input_objs", {})
        for key, item in config.get("input", {}).items():
            if key not in reg["input"]:
                raise ValueError(f"Invalid input type {key}")
            loader = reg["input"][key]
            kwargs, safe = loader.getKwargs(item, config, logger)
            objs[key] = [loader.init_func(**kwargs)]
        return objs

    class InputLoader:
        def __init__(self, init_func, has_nobj=False, file_scope=False, takes_logger=False, use_proxy=True):
            self.init_func, self.file_scope = init_func, file_scope

    cfg = types.SimpleNamespace(
        ImageBuilder=type("ImageBuilder", (), {}), StampBuilder=type("StampBuilder", (), {}), PhotonOpBuilder=type("PhotonOpBuilder", (), {}),
        SkipThisObject=type("SkipThisObject", (Exception,), {}),
        RegisterImageType=lambda name, b: reg["image"].__setitem__(name, b),
        RegisterStampType=lambda name, b: reg["stamp"].__setitem__(name, b),
        RegisterPhotonOpType=lambda name, b: reg["photon_op"].__setitem__(name, b),
        ParseValue=parse_value)
    if full:
        cfg.InputLoader = InputLoader
        for base_name in ("OutputBuilder", "ExtraOutputBuilder", "WCSBuilder", "BandpassBuilder", "SEDBuilder"):
            setattr(cfg, base_name, type(base_name, (), {}))
        cfg.RegisterInputType = lambda name, loader: reg["input"].__setitem__(name, loader)
        cfg.RegisterObjectType = lambda name, fn, input_type=None: reg["object"].__setitem__(name, fn)
        cfg.RegisterValueType = lambda name, fn, valid, input_type=None: reg["value"].__setitem__(name, (fn, valid))
        cfg.RegisterOutputType = lambda name, b: reg["output"].__setitem__(name, b)
        cfg.RegisterExtraOutput = lambda name, b: reg["extra"].__setitem__(name, b)
        cfg.RegisterWCSType = lambda name, b, input_type=None: reg["wcs"].__setitem__(name, b)
        cfg.RegisterBandpassType = lambda name, b, input_type=None: reg["bandpass"].__setitem__(name, b)
        cfg.RegisterSEDType = lambda name, b, input_type=None: reg["sed"].__setitem__(name, b)
        cfg.RegisterTemplate = lambda name, path: reg["template"].__setitem__(name, path)
        cfg.ProcessInput = process_input
        cfg.BuildImage = lambda base, image_num, obj_num, logger=None: reg["image"][base["image"]["type"]].buildImage(
            base["image"], base, image_num, obj_num, logger)[0]

    class ImageF:
        def __init__(self, array, xmin=1, ymin=1):
            self.array, self.xmin, self.ymin = array, xmin, ymin
    return types.SimpleNamespace(ImageF=ImageF), cfg, reg


def test_names_are_registered_and_the_image_builder_renders_the_requested_ccd():
    gs, cfg, reg = _fake_galsim()
    calls = []

    def fake_process(config, overrides=None, device=None, **kw):
        calls.append((config, overrides))
        return types.SimpleNamespace(images=[np.ones((4004, 4096), dtype=np.float32)], truth=[{"mode": ["phot"]}])
    done = galsim_plugin.register(gs, cfg, process=fake_process)
    assert set(reg["image"]) == {"LSST_Image", "LSST_PhotonPoolingImage", "LSST_Flat"}
    assert set(reg["stamp"]) == {"LSST_Silicon", "LSST_Photons"}
    assert set(reg["photon_op"]) == {"RubinOptics", "RubinDiffractionOptics", "RubinDiffraction", "BandpassRatio"}
    assert len([d for d in done if d[0] in ("image", "stamp", "photon_op")]) == 9
    base = {"image": {"type": "LSST_Image", "det_name": "R22_S11", "_internal": 1}, "stamp": {"type": "LSST_Silicon"},
            "output": {"det_num": {"type": "Sequence", "first": 94, "nitems": 189}, "nfiles": 3}, "file_num": 2,
            "_objects": "galsim bookkeeping", "modules": ["imsim_amd.galsim_plugin"]}
    b = reg["image"]["LSST_Image"]
    assert b.setup(base["image"], base, 0, 0, [], None) == (4096, 4004) and base["det_xsize"] == 4096
    image, var = b.buildImage(base["image"], base, 0, 0, None)
    assert image.array.shape == (4004, 4096) and image.xmin == 1 and var == 0.0
    (config, overrides), = calls
    assert "_objects" not in config and "modules" not in config and "_internal" not in config["image"]
    assert overrides["output.det_num"]["first"] == 96 and overrides["output.nfiles"] == 1        # file 2 of a run that starts at CCD 94
    assert base["_imsim_amd_truth"] == {"mode": ["phot"]}
    op = reg["photon_op"]["RubinOptics"].buildPhotonOp({"type": "RubinOptics", "det_name": "R22_S11", "camera": "LsstCamSim"}, base, None)
    assert op.kwargs == {"det_name": "R22_S11", "camera": "LsstCamSim"} and op.name == "RubinOptics" and callable(op.applyTo)


def test_every_name_of_the_reference_is_registered_and_its_default_config_passes_the_input_and_output_stages():
    """input, object, value, output, extra-output, WCS, bandpass and SED types of imSim are all known to GalSim's
    registries once the module is imported, so that a reference YAML (`modules: [imsim_amd.galsim_plugin]` in place of
    `modules: [imsim]`) gets through ProcessInput and the output stage to the image builder: a walk over the reference's own
    config/imsim-config-instcat.yaml (template chain resolved by imsim_amd.config.load_config) with the stand-in."""
    import os
    from imsim_amd import config as our_config
    gs, cfg, reg = _fake_galsim()
    rendered = []

    def fake_process(config, overrides=None, device=None, **kw):
        rendered.append(overrides["output.det_num"]["first"])
        return types.SimpleNamespace(images=[np.zeros((4004, 4096), dtype=np.float32)], truth=[{}])
    done = galsim_plugin.register(gs, cfg, process=fake_process)
    assert set(reg["input"]) == set(galsim_plugin.INPUT_TYPES) and len(galsim_plugin.INPUT_TYPES) == 10
    assert set(reg["object"]) == {"AtmosphericPSF", "DoubleGaussianPSF", "KolmogorovPSF", "InstCatObj", "SkyCatObj"}
    assert set(reg["value"]) == {"TreeRingCenter", "TreeRingFunc", "InstCatWorldPos", "SkyCatWorldPos", "OpsimData", "SkyLevel", "RowData"}
    assert set(reg["output"]) == {"LSST_CCD"} and set(reg["wcs"]) == {"Batoid", "Dict"}
    assert set(reg["extra"]) == {"readout", "photon_pooling_truth", "opd", "sag", "process_info"}
    assert set(reg["bandpass"]) == {"RubinBandpass"} and set(reg["sed"]) == {"InstCatSED"}
    assert ("output", "LSST_CCD") in done
    here = os.path.dirname(os.path.abspath(__file__))
    ref_cfg = "/root/reference/config/imsim-config-instcat.yaml"
    if os.path.isfile(ref_cfg):                                 # the reference's own file where it is present (not on the GPU box)
        base = our_config.load_config(ref_cfg, template_dirs=["/root/reference/config"],
                                      overrides={"input.instance_catalog.file_name": os.path.join(here, "golden", "example_instcat_subset.txt")})
    else:
        base = our_config.load_config(os.path.join(here, "data", "test-config-instcat.yaml"), template_dirs=[os.path.join(here, "data")],
                                      overrides={"input.instance_catalog.file_name": os.path.join(here, "golden", "example_instcat_subset.txt")})
    base["input"] = {k: v for k, v in base["input"].items() if v not in ("", None)}
    for key in base["input"]:
        assert key in reg["input"], key
    objs = cfg.ProcessInput(base)
    assert set(objs) == set(base["input"]) and objs["instance_catalog"][0].kwargs["file_name"].endswith("example_instcat_subset.txt")
    # value types: OpsimData reads the visit's record from the instance catalog header
    alt, safe = cfg.ParseValue({"a": {"type": "OpsimData", "field": "altitude"}}, "a", base, float)
    assert abs(alt - 53.16185928082866) < 1e-9 and safe
    with pytest.raises(ValueError):
        cfg.ParseValue({"a": {"type": "OpsimData", "field": "no_such_field"}}, "a", base, float)
    # the types the psf / gal / stamp sections name
    for section in ("psf", "gal"):
        items = base[section].get("items", [base[section]]) if isinstance(base.get(section), dict) else []
        for it in items:
            if it.get("type") in galsim_plugin.OBJECT_TYPES:
                obj, _ = reg["object"][it["type"]](it, base, [], None, None)
                assert obj.name == it["type"]
    # the output stage: LSST_CCD counts 189 files unless nfiles says otherwise, names the detector, calls the image builder
    out = reg["output"]["LSST_CCD"]
    assert out.getNFiles({}, base) == 189 and out.getNFiles({"nfiles": 3}, base) == 3
    base["file_num"] = 2
    ocfg = dict(base["output"])
    ocfg["det_num"] = {"type": "Sequence", "first": 94, "nitems": 189}
    ocfg.pop("exptime", None)                                   # `$exptime`: GalSim's own eval strings, not this stand-in's business
    out.setup(ocfg, base, 2, None)
    assert base["det_num"] == 96 and base["det_name"] == our_config.det_name_of(96) and base["exptime"] == 30.0
    assert (base["det_xsize"], base["det_ysize"]) in ((4096, 4004), (4072, 4000))
    base["image"].setdefault("type", "LSST_Image")
    images = out.buildImages(ocfg, base, 2, 0, 0, [], None)
    assert len(images) == 1 and images[0].array.shape == (4004, 4096) and rendered == [96]
    wcs = reg["wcs"]["Batoid"].buildWCS({"type": "Batoid"}, base, None)
    assert wcs is not None
    bp, _ = reg["bandpass"]["RubinBandpass"].buildBandpass({"type": "RubinBandpass", "band": "r"}, base, None)
    assert bp is not None


def test_without_galsim_the_module_imports_and_refuses_to_register():
    import pytest
    if galsim_plugin.galsim is None:
        with pytest.raises(ImportError):
            galsim_plugin.register()
