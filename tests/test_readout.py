"""CCD readout chain: oracle against the golden vectors generated from the reference's
bleed_trails.py / cte_matrix, the camera geometry against the known answers of the reference's tests, the FITS
writer, and the host logic.  CPU only; the GPU parity tests are in test_parity_gpu.py."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from imsim_amd import _abi, camera, fits_io, readout
from oracle import orc_loader

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "readout_golden.npz"))


def small_ccd(seg=(12, 20), raw=(20, 26), prescan=3, vendor="E2V", xtalk=True):
    """a CCD with the LSSTCam segment topology at toy size"""
    old = camera.SEGMENT[vendor]
    camera.SEGMENT[vendor] = dict(seg=seg, raw=raw, prescan=prescan)
    try:
        return camera.make_ccd("R22_S11" if vendor == "E2V" else "R01_S00", xtalk=xtalk)
    finally:
        camera.SEGMENT[vendor] = old


def test_bleed_channel_matches_reference():
    fw = float(GOLD["full_well"])
    out = orc_loader.bleed_eimage(GOLD["chan_in"][:, None], fw, midline_stop=False)[:, 0]
    assert np.array_equal(out, GOLD["chan_out"])
    # the reference's own assertions (tests/test_bleed_trails.py:41-64): charge conserved, clipped at full well,
    # one uninterrupted trail centred on the star
    assert out.sum() == GOLD["chan_in"].sum() and out.max() == fw
    idx = np.where(out == fw)[0]
    assert np.array_equal(idx, np.arange(idx[0], idx[-1] + 1)) and abs(1000 - (idx[0] - 1 + idx[-1]) / 2) <= 1


def test_bleed_eimage_matches_reference_with_and_without_midline_stop():
    fw = float(GOLD["full_well"])
    for key, mid in (("img_midline", True), ("img_nomidline", False)):
        out = orc_loader.bleed_eimage(GOLD["img_in"], fw, midline_stop=mid)
        assert np.array_equal(out, GOLD[key]), key
        assert out.max() <= fw
    # charge only ever leaves through the bottom of a channel
    assert GOLD["img_nomidline"].sum() < GOLD["img_in"].sum()
    assert not np.array_equal(GOLD["img_midline"], GOLD["img_nomidline"])


def test_bleed_regression_channel_of_the_reference_has_no_negative_pixels():
    """tests/test_bleed_trails.py:66-75 (data file neg_pixel_bleed.pickle); in float64 the restatement follows the
    reference operation by operation, the reference's float32 run agrees to float32 rounding."""
    out = orc_loader.bleed_eimage(GOLD["neg_in"][:, None], float(GOLD["neg_fw"]), midline_stop=False)[:, 0]
    assert np.array_equal(out, GOLD["neg_out"])
    assert (out > 0).all()
    assert np.allclose(out, GOLD["neg_out_native"], rtol=3e-6)


def test_cte_matrix_matches_reference():
    for key, (n, cti, nt) in {"cte_64_1e-6": (64, 1e-6, 20), "cte_64_1e-3": (64, 1e-3, 20), "cte_40_1e-2_nt5": (40, 1e-2, 5)}.items():
        m = readout.cte_matrix(n, cti, nt)
        assert np.allclose(m, GOLD[key], rtol=1e-13, atol=0)
        band = readout.cte_band(n, cti, nt)
        assert band.shape == (n, nt + 1)
        # no charge is created; what is lost trails behind by more than ntransfers pixels
        q = m @ np.ones(n)
        assert (q <= 1 + 1e-15).all() and q[0] == 1 - cti


def test_camera_geometry_known_answers():
    """tests/test_readout.py:63-91 of the reference: an E2V CCD is 4096 x 4004 with 16 raw segments of 2048 x 576,
    Segment10 / Segment17 DATASEC and DETSEC."""
    cam = camera.Camera("LsstCamSim")
    assert len(cam) == 189
    ccd = cam["R22_S11"]
    assert ccd.bounds.numpyShape() == (4004, 4096) and list(ccd) == camera.CHANNELS
    for amp in ccd.values():
        assert amp.raw_bounds.numpyShape() == (2048, 576)
    a10, a17 = ccd["C10"], ccd["C17"]
    assert readout.section_keyword(a10.raw_data_bounds) == "[11:522,1:2002]"
    assert readout.section_keyword(a10.bounds, flipx=a10.raw_flip_x, flipy=a10.raw_flip_y) == "[512:1,4004:2003]"
    assert readout.section_keyword(a17.bounds, flipx=a17.raw_flip_x, flipy=a17.raw_flip_y) == "[4096:3585,4004:2003]"
    itl = cam["R01_S00"]
    assert itl.bounds.numpyShape() == (4000, 4072) and itl["C00"].raw_data_bounds.xmin == 4
    # the segments tile the CCD exactly once
    cover = np.zeros(ccd.bounds.numpyShape(), dtype=int)
    for amp in ccd.values():
        b = amp.bounds
        cover[b.ymin - 1:b.ymax, b.xmin - 1:b.xmax] += 1
    assert (cover == 1).all()
    with pytest.raises(ValueError):
        camera.Camera("NoSuchCam")


def test_camera_bias_levels(tmp_path):
    """tests/test_camera.py:16-45 of the reference: per-amp levels from a json file (by name or full path), or one
    level for all amps; CcdReadout.bias_level is None when a file is given (tests/test_readout.py:71-78)."""
    levels = {f"{r}_{s}": {a: 20000.0 + 7 * k + i for i, a in enumerate(camera.CHANNELS)}
              for k, (r, s) in enumerate((r, s) for r in camera.RAFTS for s in camera.SENSORS)}
    path = tmp_path / "bias_levels.json"
    path.write_text(json.dumps(levels))
    for arg, kw in ((str(path), {}), ("bias_levels.json", {"data_dir": str(tmp_path)})):
        cam = camera.Camera("LsstCamSim", bias_levels_file=arg, **kw)
        for det, ccd in cam.items():
            for name, amp in ccd.items():
                assert amp.bias_level == levels[det][name]
    cam = camera.Camera("LsstCamSim", bias_level=1234.0)
    assert all(amp.bias_level == 1234.0 for ccd in cam.values() for amp in ccd.values())
    with pytest.raises(FileNotFoundError):
        camera.Camera("LsstCamSim", bias_levels_file="nope.json")
    eimg = readout.EImage(None, readout.eimage_header("R22_S11", 30.0))
    ro = readout.CcdReadout(eimg, bias_level=1234.0, bias_levels_file=str(path))
    assert ro.bias_level is None and ro.descriptor().amps[3].bias_level == levels["R22_S11"]["C13"]
    assert readout.CcdReadout(eimg, bias_level=1234.0).descriptor().amps[3].bias_level == 1234.0
    # json round trip of a complete camera description
    cam.to_json(tmp_path / "cam.json")
    back = camera.Camera.from_json(tmp_path / "cam.json")
    assert back["R10_S02"]["C05"].bounds == cam["R10_S02"]["C05"].bounds and back["R10_S02"].xtalk == cam["R10_S02"].xtalk


def _toy_readout(xtalk=True, vendor="E2V", **kw):
    ccd = small_ccd(vendor=vendor, xtalk=xtalk)
    ny, nx = ccd.bounds.numpyShape()
    eimg = readout.EImage(None, readout.eimage_header("R22_S11" if vendor == "E2V" else "R01_S00", 30.0))
    ro = readout.CcdReadout(eimg, camera_obj={eimg.header["DET_NAME"]: ccd}, **kw)
    return ccd, ro, nx, ny


def test_segments_gain_flips_and_scan_regions():
    """Every e-image pixel appears once, divided by its amp's gain, at the readout position the flips imply; prescan
    and overscan stay empty (readout.py:440-460)."""
    for vendor in ("E2V", "ITL"):
        ccd, ro, nx, ny = _toy_readout(xtalk=False, vendor=vendor, scti=0, pcti=0, read_noise=0.0, dark_current=0.0, bias_level=0.0)
        e = np.arange(nx * ny, dtype=np.float64).reshape(ny, nx)
        d = ro.descriptor()
        stages = {}
        out = orc_loader.readout_chain(e, d, 1e9, ro.midline_stop(), 0.0, readout.DARK_STREAM, 5, None, None, stages)
        seg = stages["segments"]
        for k, amp in enumerate(ccd.values()):
            b = amp.bounds
            ref = e[b.ymin - 1:b.ymax, b.xmin - 1:b.xmax].astype(np.float32) / np.float32(amp.gain)
            if amp.raw_flip_x:
                ref = ref[:, ::-1]
            if amp.raw_flip_y:
                ref = ref[::-1, :]
            r = amp.raw_data_bounds
            assert np.array_equal(seg[k, r.ymin - 1:r.ymax, r.xmin - 1:r.xmax], ref)
            mask = np.ones(seg[k].shape, bool)
            mask[r.ymin - 1:r.ymax, r.xmin - 1:r.xmax] = False
            assert (seg[k][mask] == 0).all()
            assert np.array_equal(out[k], seg[k].astype(np.int32))       # no bias, no noise: plain truncation


def test_crosstalk_cte_bias_and_noise():
    ccd, ro, nx, ny = _toy_readout(xtalk=True, scti=1e-3, pcti=2e-3, dark_current=0.0, bias_level=1000.0, read_noise=4.0)
    rng = np.random.default_rng(3)
    e = rng.poisson(500.0, size=(ny, nx)).astype(np.float64)
    e[5:9, 30:34] += 40000.0
    d = ro.descriptor()
    st = {}
    out = orc_loader.readout_chain(e, d, 1e9, True, 0.0, readout.DARK_STREAM, 11, ro.pcte_band, ro.scte_band, st)
    # crosstalk: amp i sees a_i + sum_j x_ij a_j of the flipped arrays (readout.py:403-411), float32
    plain = {}
    ro2 = readout.CcdReadout(ro.eimage, camera_obj={ro.det_name: small_ccd(xtalk=False)}, scti=0, pcti=0)
    orc_loader.readout_chain(e, ro2.descriptor(), 1e9, True, 0.0, readout.DARK_STREAM, 11, None, None, plain)
    a = plain["segments"]
    want = a.copy()
    for i, row in enumerate(ccd.xtalk):
        s = np.zeros_like(a[0])
        for j, x in enumerate(row):
            s = s + np.float32(x) * a[j]
        want[i] = a[i] + s
    r = list(ccd.values())[0].raw_data_bounds
    sl = (slice(None), slice(r.ymin - 1, r.ymax), slice(r.xmin - 1, r.xmax))
    assert np.array_equal(st["segments"][sl], want[sl])
    # CTE: dense matrices applied along columns, then rows (readout.py:391-401)
    pm, sm = readout.cte_matrix(d.raw_h, 2e-3), readout.cte_matrix(d.raw_w, 1e-3)
    x = st["segments"].astype(np.float64)
    x = np.einsum("ij,ajk->aik", pm, x).astype(np.float32).astype(np.float64)
    x = np.einsum("ij,akj->aki", sm, x).astype(np.float32)
    assert np.allclose(st["cte"], x, rtol=2e-6, atol=1e-4)
    # deferred charge shows up in the overscan, the image loses it
    assert st["cte"][:, :, r.xmax:].sum() > 0 and st["cte"][:, r.ymax:, :].sum() > 0
    # bias + read noise: mean and sigma of the overscan corner, where no signal arrives
    resid = (out.astype(np.float64) - st["cte"] - 1000.0)
    assert abs(resid.mean() + 0.5) < 0.1                      # truncation towards zero of positive values: -0.5 on average
    assert abs(resid.std() - np.sqrt(16.0 + 1.0 / 12.0)) < 0.1
    again = orc_loader.readout_chain(e, d, 1e9, True, 0.0, readout.DARK_STREAM, 11, ro.pcte_band, ro.scte_band)
    other = orc_loader.readout_chain(e, d, 1e9, True, 0.0, readout.DARK_STREAM, 12, ro.pcte_band, ro.scte_band)
    assert np.array_equal(again, out) and not np.array_equal(other, out)


def test_dark_current_is_poisson():
    ccd, ro, nx, ny = _toy_readout(xtalk=False, scti=0, pcti=0, dark_current=0.5, readout_time=2.0)
    assert ro.dark_level() == 0.5 * 32.0
    st = {}
    orc_loader.readout_chain(np.zeros((ny, nx)), ro.descriptor(), 1e9, True, ro.dark_level(), readout.DARK_STREAM, 1, None, None, st)
    d = st["dark"]
    assert abs(d.mean() - 16.0) < 0.2 and abs(d.var() - 16.0) < 1.0 and (d == np.round(d)).all()


def test_fits_writer_round_trip(tmp_path):
    hdr = readout.eimage_header("R22_S11", 30.0, opsim_data={"mjd": 60000.25, "fieldRA": 60.49, "fieldDec": -38.16, "band": "r",
                                                             "rotTelPos": 12.5, "observationId": 398414, "airmass": 1.1},
                                header_vals={"TESTKEY1": "TESTVAL1"})
    assert hdr["DAYOBS"] == "20230224" and hdr["RUNNUM"] == 398414 and hdr["TESTKEY1"] == "TESTVAL1"
    assert readout.mjd_to_isot(51444.0) == "1999-09-23T00:00:00.000" and readout.mjd_to_isot(60000.25) == "2023-02-25T06:00:00.000"
    img = np.arange(12, dtype=np.float32).reshape(3, 4) * 1.5
    seg = (np.arange(35, dtype=np.int32).reshape(5, 7) - 10) * 70000
    f = tmp_path / "t.fits"
    fits_io.write_fits(str(f), [(hdr, img), ({"EXTNAME": "Segment10", "DATASEC": "[11:522,1:2002]"}, seg)])
    raw = f.read_bytes()
    assert len(raw) % 2880 == 0 and raw[:30] == b"SIMPLE  =                    T"
    assert b"HIERARCH ROTTELPOS = " in raw and b"XTENSION= 'IMAGE   '" in raw
    (h0, d0), (h1, d1) = fits_io.read_fits(str(f))
    assert np.array_equal(d0, img) and d0.dtype == np.float32 and np.array_equal(d1, seg) and d1.dtype == np.int32
    assert h0["DET_NAME"] == "R22_S11" and h0["EXPTIME"] == 30.0 and h0["ROTTELPOS"] == 12.5 and h0["MJD-OBS"] == 60000.25
    assert h1["EXTNAME"] == "Segment10" and h1["DATASEC"] == "[11:522,1:2002]" and h1["NAXIS1"] == 7 and h1["NAXIS2"] == 5


def test_primary_header_has_the_keywords_the_stack_needs():
    """tests/test_readout.py:93-122 of the reference"""
    hdr = readout.eimage_header("R22_S11", 30.0, opsim_data={"mjd": 60000.25, "band": "i", "altitude": 70.0, "azimuth": 10.0,
                                                             "airmass": 1.06, "HASTART": -0.3, "HAEND": -0.29})
    ro = readout.CcdReadout(readout.EImage(None, hdr), added_keywords={"TESTKEY1": "TESTVAL1", "SOMEMATH": "3"})
    ph = readout.get_primary_hdu(ro.eimage, ro.ccd.getSerial(), camera_name=ro.camera_name, added_keywords=ro.added_keywords)
    for key in ("RA", "DEC", "RASTART", "DECSTART", "ROTPA", "ROTCOORD", "HASTART", "ELSTART", "AZSTART", "AMSTART", "TRACKSYS",
                "RADESYS", "ORIGIN", "TELCODE", "IMSIMVER", "TESTKEY1", "SOMEMATH", "LSST_NUM", "CHIPID", "OBSID", "DATE-OBS"):
        assert key in ph, key
    assert ph["EXPTIME"] == 30.0 and ph["FILTER"] == "i_39" and ph["OBSID"] == "MC_S_20230224_000000" and ph["RAFTBAY"] == "R22"
    flat = readout.eimage_header("R22_S11", 30.0, header_vals={"image_type": "FLAT"})
    assert readout.get_primary_hdu(readout.EImage(None, flat), "x")["TRACKSYS"] == "LOCAL"


def test_compute_rotSkyPos_from_pointing_rotator_and_time():
    """imsim/readout.py:95-149: ROTANGLE = 270 - rotTelPos + pseudo parallactic angle.  The reference gets the angle from
    astropy / ERFA through its batoid WCS factory; the geometric restatement is pinned by the header of the reference's own
    example instance catalog (tests/golden/example_instcat_subset.txt): its altitude and azimuth are reproduced from
    (ra, dec, mjd) alone, and OpSim's rotSkyPos there is rotTelPos minus the parallactic angle."""
    hdr = {}
    for line in open(os.path.join(os.path.dirname(__file__), "golden", "example_instcat_subset.txt")):
        k, _, v = line.partition(" ")
        if k in ("rightascension", "declination", "mjd", "altitude", "azimuth", "rotskypos", "rottelpos"):
            hdr[k] = float(v)
    alt, az, pq = readout.pointing_geometry(hdr["rightascension"], hdr["declination"], hdr["mjd"])
    assert abs(alt - hdr["altitude"]) < 0.1 and abs(az - hdr["azimuth"]) < 0.1
    assert abs((hdr["rottelpos"] - pq) % 360.0 - hdr["rotskypos"]) < 0.2
    theta = readout.compute_rotSkyPos(hdr["rightascension"], hdr["declination"], hdr["rottelpos"], hdr["mjd"], "r")
    assert 0.0 <= theta < 360.0 and abs(theta - (270.0 - hdr["rottelpos"] + pq) % 360.0) < 1e-12
    # the parallactic angle of a source on the meridian south of the zenith is 0 (zenith due north of it): the hour angle
    # follows the right ascension, so stepping ra through the local sidereal time must cross pq = 0 with d(pq)/d(ha) > 0
    ra = np.linspace(0.0, 360.0, 7201)
    pqs = np.array([readout.pointing_geometry(r, -60.0, 60143.4)[2] for r in ra])
    alts = np.array([readout.pointing_geometry(r, -60.0, 60143.4)[0] for r in ra])
    k = int(np.argmax(alts))                                   # upper culmination
    assert abs((pqs[k] + 180.0) % 360.0 - 180.0) < 0.2 and abs(alts[k] - (90.0 - abs(-60.0 - readout.RUBIN_LATITUDE_DEG))) < 0.2   # precession of 23 years moves the declination by 0.13 deg
    # and the raw file's header carries the recomputed angle, not the catalog's
    eh = readout.eimage_header("R22_S11", 30.0, opsim_data={"mjd": hdr["mjd"], "band": "r", "fieldRA": hdr["rightascension"],
                                                            "fieldDec": hdr["declination"], "rotTelPos": hdr["rottelpos"],
                                                            "rotSkyPos": hdr["rotskypos"]})
    ph = readout.get_primary_hdu(readout.EImage(None, eh), "x")
    assert ph["ROTANGLE"] == theta and ph["ROTPA"] == theta


def test_readout_needs_the_gpu():
    ccd, ro, nx, ny = _toy_readout()
    import torch
    ro.eimage.array = torch.zeros((ny, nx), dtype=torch.float64)
    with pytest.raises(_abi.ImsimHipError):
        ro.build_amp_images(1)


def test_treering_displacement_bound_is_rigorous():
    """engine.treering_displacement_bound (ims_sensor_t.pristine_margin): no point of the spline the kernels evaluate
    (chord + cubic term, imsim_hip.hip treering_shift) exceeds it, and it is not wastefully loose."""
    from imsim_amd import configs
    from imsim_amd.engine import treering_displacement_bound
    ss = configs.silicon_setup(512, 512)
    bound = treering_displacement_bound(ss)
    v, m, h = np.asarray(ss.tr_table), np.asarray(ss.tr_table2), float(ss.tr_dr)
    b = np.linspace(0.0, 1.0, 41)[None, :]
    a = 1.0 - b
    f = a * v[:-1, None] + b * v[1:, None] + ((a ** 3 - a) * m[:-1, None] + (b ** 3 - b) * m[1:, None]) * h * h / 6.0
    assert np.abs(f).max() <= bound <= 1.2 * np.abs(f).max() + 1e-9
    assert 0 < bound < 0.1                                   # tree rings move boundaries by a few per cent of a pixel
    none = configs.silicon_setup(512, 512, tree_rings=False)
    assert treering_displacement_bound(none) == 0.0


def test_rice_tile_compression_round_trip(tmp_path):
    """The reference writes the raw segments RICE_1 tile-compressed (imsim/readout.py:500-510).  Rice coder (CFITSIO's
    fits_rcomp, 32-bit, block 32): hand-checked bit patterns, lossless round trips over quiet, noisy, extreme and ragged
    rows, and a file written with write_fits_compressed read back through the binary-table reader."""
    from imsim_amd import fits_io
    const = fits_io.rice_compress_rows(np.full((1, 20), 7, dtype=np.int32))[0]
    assert list(const) == [0, 0, 0, 7, 0]                        # 32 bits of the first pixel + the 5-bit header of an all-zero block
    ramp = ''.join(str(b) for b in np.unpackbits(fits_io.rice_compress_rows(np.array([[1, 2, 3, 4]], dtype=np.int32))[0]))
    assert ramp.startswith('0' * 31 + '1' + '00001' + '1' + '001' * 3)        # FS = 0: zig-zag differences 0, 2, 2, 2 in unary
    rng = np.random.default_rng(0)
    for shape, scale in (((5, 37), 3), ((8, 576), 50), ((3, 64), 1e5), ((4, 100), 2e9)):
        img = rng.normal(1000, scale, shape).clip(-2 ** 31, 2 ** 31 - 1).astype(np.int32)
        img[0, 5], img[0, 6] = 2 ** 31 - 1, -2 ** 31
        rows = fits_io.rice_compress_rows(img)
        back = np.stack([fits_io.rice_decompress_row(b, shape[1]) for b in rows])
        assert np.array_equal(back, img)
    seg = rng.normal(25000, 8, (64, 576)).astype(np.int32)
    assert sum(len(b) for b in fits_io.rice_compress_rows(seg)) < 0.3 * seg.nbytes     # read noise of 8 ADU: ~0.75 bytes per pixel
    f = str(tmp_path / "raw.fits")
    fits_io.write_fits_compressed(f, [({"OBSID": "x"}, None), ({"EXTNAME": "Segment10", "DATASEC": "[4:512,1:2000]"}, seg)])
    hdus = fits_io.read_fits(f)
    assert hdus[0][0]["OBSID"] == "x" and hdus[1][0]["ZCMPTYPE"] == "RICE_1" and hdus[1][0]["EXTNAME"] == "Segment10"
    assert hdus[1][0]["ZNAXIS1"] == 576 and hdus[1][0]["ZTILE2"] == 1 and hdus[1][0]["DATASEC"] == "[4:512,1:2000]"
    assert np.array_equal(fits_io.read_compressed_image(*hdus[1]), seg)
