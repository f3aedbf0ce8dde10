"""The YAML config surface (template inheritance, dotted overrides, $/@ values, registered type
names, parameter checking) -- no GPU needed up to the point of rendering."""
import math
import os

import pytest

from imsim_amd import config, lsst_image, photon_pooling

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "data")
INSTCAT = os.path.join(HERE, "golden", "example_instcat_subset.txt")


def load(**over):
    o = {"input.instance_catalog.file_name": INSTCAT}
    o.update(over)
    return config.load_config(os.path.join(DATA, "test-config-instcat.yaml"), template_dirs=[DATA], overrides=o)


def test_template_chain_and_dotted_overrides():
    cfg = load(**{"image.nobjects": 5, "stamp.draw_method": "fft", "image.sensor": ""})
    assert cfg["image"]["type"] == "LSST_Image" and cfg["image"]["nbatch"] == 100          # from the base template
    assert cfg["input"]["instance_catalog"]["sort_mag"] is False                            # dotted key in the file
    assert cfg["input"]["opsim_data"]["file_name"] == "@input.instance_catalog.file_name"
    assert cfg["image"]["nobjects"] == 5 and cfg["stamp"]["draw_method"] == "fft" and cfg["image"]["sensor"] == ""
    assert cfg["output"]["det_num"]["first"] == 94 and cfg["output"]["det_num"]["nitems"] == "$camera_info['ndets']"


def test_registered_plugin_names():
    """the names existing YAML files refer to."""
    assert {"LSST_Silicon", "LSST_Photons"} <= set(config.valid_stamp_types)
    assert {"LSST_Image", "LSST_PhotonPoolingImage"} <= set(config.valid_image_types)
    assert {"RubinOptics", "RubinDiffractionOptics", "RubinDiffraction", "BandpassRatio", "TimeSampler", "PupilAnnulusSampler",
            "PhotonDCR", "FocusDepth", "Refraction"} <= set(config.valid_photon_op_types)
    assert {"AtmosphericPSF", "KolmogorovPSF", "DoubleGaussianPSF", "InstCatObj"} <= set(config.valid_psf_types)
    assert {"atm_psf", "tree_rings", "instance_catalog", "opsim_data", "telescope", "checkpoint"} <= set(config.valid_input_types)
    assert {"OpsimData", "TreeRingCenter", "TreeRingFunc", "Degrees", "Eval"} <= set(config.valid_value_types)
    assert config.det_name_of(94) == "R22_S11" and config.det_type_of("R22_S11") == "E2V" and config.det_type_of("R01_S00") == "ITL"


def test_value_evaluation():
    from imsim_amd import instcat
    cfg = load()
    cfg["_opsim_data"] = instcat.read_header(INSTCAT)
    ev = config.Evaluator(cfg)
    ev.load_eval_variables(cfg["eval_variables"])
    assert ev.vars["band"] == "r" and ev.vars["exptime"] == 30.0
    assert abs(ev.vars["altitude"] - math.radians(53.16185928082866)) < 1e-15
    assert abs(ev.vars["pupil_area"] - math.pi * (4.18 ** 2 - 2.55 ** 2) * 1e4) < 1e-6
    assert ev.value("@input.instance_catalog.sort_mag") is False
    assert ev.value("-30.24463 degrees") == math.radians(-30.24463)
    assert ev.value({"type": "FormattedStr", "format": "eimage_%08d-%s.fits", "items": [{"type": "OpsimData", "field": "observationId"}, "$band"]}) \
        == "eimage_00398414-r.fits"
    with pytest.raises(config.GalSimConfigError):
        ev.value({"type": "OpsimData", "field": "nonexistent"})


def test_photon_op_parameter_checking():
    ev = config.Evaluator({})
    ev.vars.update(exptime=30.0)
    ops, _ = config.build_photon_ops([{"type": "TimeSampler", "t0": 0.0, "exptime": "$exptime"},
                                      {"type": "Refraction", "index_ratio": 3.9}], ev, 620.0)
    assert ops[0][2] == [0.0, 30.0] and ops[1][2] == [3.9]
    with pytest.raises(config.GalSimConfigError):
        config.build_photon_ops([{"type": "RubinOptics", "camera": "LsstCamSim"}], ev, 620.0)           # missing required keys
    with pytest.raises(config.GalSimConfigError):
        config.build_photon_ops([{"type": "Refraction", "index_ratio": 3.9, "bogus": 1}], ev, 620.0)  # unexpected key
    with pytest.raises(config.GalSimConfigError):
        config.build_photon_ops([{"type": "NoSuchOp"}], ev, 620.0)


def test_image_setup_parameter_surface():
    b = lsst_image.LSST_ImageBuilder()
    assert b.setup({"type": "LSST_Image", "det_name": "R22_S11"}) == (4096, 4004)
    assert (b.nbatch, b.nsubbatch, b.nbatch_fft, b.nbatch_per_checkpoint) == (10, 50, 1, 1)          # lsst_image.py:112-120
    assert b.setup({"det_name": "R01_S00", "xsize": 100, "ysize": 120}, "ITL") == (100, 120)
    assert lsst_image.LSST_ImageBuilder().setup({"det_name": "R01_S00"}, "ITL") == (4072, 4000)
    with pytest.raises(lsst_image.GalSimConfigError):
        b.setup({"type": "LSST_Image"})                                  # det_name is required
    with pytest.raises(lsst_image.GalSimConfigError):
        b.setup({"det_name": "R22_S11", "apply_fringing": True})         # fringing needs boresight (:84-87)
    with pytest.raises(lsst_image.GalSimConfigError):
        b.setup({"det_name": "R22_S11", "no_such_key": 1})
    with pytest.raises(photon_pooling.GalSimConfigValueError):
        lsst_image.LSST_PhotonPoolingImageBuilder().setup({"det_name": "R22_S11"}, "LSST_Silicon")


def test_camera_info_and_sequence_values(tmp_path):
    """`eval_variables.dcamera_info` (config/imsim-config.yaml:56-58) and the per-CCD value of `output.det_num`
    (type Sequence) inside file-name formats (config/imsim-config.yaml:335-352)."""
    from imsim_amd import config
    info = config.camera_info("LsstCamSim")
    assert info["ndets"] == 189 and info["camera_name"] == "LsstCamSim" and info["telescope_format"] % "r" == "LSST_r.yaml"
    assert config.camera_info("LsstComCamSim")["ndets"] == 9
    (tmp_path / "LsstCamSim_info.yaml").write_text("ndets: 3\ncamera_name: LsstCamSim\ntree_rings_file_name: mine.txt\n")
    assert config.camera_info("LsstCamSim", str(tmp_path))["tree_rings_file_name"] == "mine.txt"       # an imSim data dir wins
    cfg = {"output": {"camera": "LsstCamSim", "det_num": {"type": "Sequence", "nitems": "$camera_info['ndets']"},
                      "file_name": {"type": "FormattedStr", "format": "eimage_%s-det%03d.fits",
                                    "items": ["$det_name", "@output.det_num"]}}}
    ev = config.Evaluator(cfg)
    ev.vars["camera_info"] = info
    assert ev.value(cfg["output"]["det_num"]["nitems"]) == 189
    ev.vars["det_name"], ev.vars["_sequence_index"] = "R22_S11", 94
    assert ev.value(cfg["output"]["file_name"]) == "eimage_R22_S11-det094.fits"
    ev.vars["_sequence_index"] = None
    assert ev.value(cfg["output"]["det_num"]) == 0


def test_list_index_overrides_and_disabled_sections():
    """`psf.items.0: {...}` replaces a list element and `input.atm_psf: ""` switches an input off, as the reference's
    tests do (tests/test_stamp.py:200-218)."""
    from imsim_amd import config
    base = {"psf": {"type": "Convolve", "items": [{"type": "AtmosphericPSF"}, {"type": "Gaussian", "fwhm": 0.3}]},
            "input": {"atm_psf": {"airmass": 1.1}}}
    cfg = config.load_config(base, overrides={"psf.items.0": {"type": "Kolmogorov", "fwhm": 0.7}, "psf.items.1.fwhm": 0.25,
                                              "input.atm_psf": ""})
    assert cfg["psf"]["items"][0] == {"type": "Kolmogorov", "fwhm": 0.7} and cfg["psf"]["items"][1]["fwhm"] == 0.25
    assert cfg["input"]["atm_psf"] == ""
    psf, kpsf, fwhm, atm, extra = config.build_psf(cfg["psf"], config.Evaluator(cfg), {"kolmogorov": 2})
    assert atm is None and len(psf) == 2 and psf[0][2] == 0.7 and abs(fwhm - (0.7 ** 2 + 0.25 ** 2) ** 0.5) < 1e-12


def test_bandpass_ratio_op_from_config():
    """`BandpassRatio` with `$bandpass` / `$bandpass*0.8` (tests/test_photon_ops.py:768-790 of the reference): the
    builder tabulates target / initial over the common wavelength range."""
    import numpy as np
    from imsim_amd import tables, _abi
    wl, thr = tables.synthetic_r_band()
    ev = config.Evaluator({})
    ev.vars["bandpass"] = tables.Bandpass(wl, thr)
    ops, meta = config.build_photon_ops([{"type": "BandpassRatio", "initial_bandpass": "$bandpass",
                                          "target_bandpass": "$bandpass*0.8"}], ev, 620.0)
    assert ops == [(_abi.IMS_OP_BANDPASS_RATIO, 0, [])]
    table, wl_min, wl_step = meta["ratio"]
    inside = thr[np.searchsorted(wl, wl_min + wl_step * np.arange(len(table))).clip(0, len(wl) - 1)] > 0
    assert np.allclose(table[inside], 0.8, rtol=1e-12) and wl_min == wl[0] and abs(wl_min + wl_step * (len(table) - 1) - wl[-1]) < 1e-9
    with pytest.raises(config.GalSimConfigError):
        config.build_photon_ops([{"type": "BandpassRatio", "initial_bandpass": 1.0, "target_bandpass": 2.0}], ev, 620.0)
    with pytest.raises(config.GalSimConfigError):
        config.build_photon_ops([{"type": "BandpassRatio", "initial_bandpass": "$bandpass"}], ev, 620.0)
    # a realistic ratio: the same band seen through more atmosphere (grey extinction here) -- the effective wavelength survives
    b2 = tables.Bandpass(wl, thr * np.exp(-0.1 * (wl / 600.0) ** -4))
    r, _, _ = b2.ratio_table(ev.vars["bandpass"])
    assert np.all(r <= 1.0) and r[-1] > r[1]


def test_every_ccd_of_a_visit_gets_its_own_seed():
    """sky noise, dark current and read noise are addressed by (seed, stream, pixel): the seed must differ
    between the CCDs of one Process() call."""
    seeds = [config.ccd_seed(398414, det) for det in range(189)]
    assert len(set(seeds)) == 189 and all(0 <= s < 2 ** 62 for s in seeds)
    assert config.ccd_seed(398414, 94) == config.ccd_seed(398414, 94) != config.ccd_seed(398415, 94)
