"""The GalSim-config adapter against a stand-in of the galsim.config registration interface (GalSim itself is not
installed here): every imSim type name of the hot path is registered, and the image builder hands the merged config to
imsim_amd.config.Process for exactly the CCD GalSim asks for."""
import types

import numpy as np

from imsim_amd import galsim_plugin


def _fake_galsim():
    reg = {"image": {}, "stamp": {}, "photon_op": {}}
    cfg = types.SimpleNamespace(
        ImageBuilder=type("ImageBuilder", (), {}), StampBuilder=type("StampBuilder", (), {}), PhotonOpBuilder=type("PhotonOpBuilder", (), {}),
        SkipThisObject=type("SkipThisObject", (Exception,), {}),
        RegisterImageType=lambda name, b: reg["image"].__setitem__(name, b),
        RegisterStampType=lambda name, b: reg["stamp"].__setitem__(name, b),
        RegisterPhotonOpType=lambda name, b: reg["photon_op"].__setitem__(name, b),
        ParseValue=lambda config, key, base, typ: (typ(config[key]), True))

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
    assert len(done) == 9
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
    assert op.kwargs == {"det_name": "R22_S11", "camera": "LsstCamSim"}


def test_without_galsim_the_module_imports_and_refuses_to_register():
    import pytest
    if galsim_plugin.galsim is None:
        with pytest.raises(ImportError):
            galsim_plugin.register()
