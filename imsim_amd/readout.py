"""e-image -> raw amplifier file: host mirror of imsim/readout.py (CcdReadout :323-533, cte_matrix :163-205,
section_keyword :152-160, get_primary_hdu :208-299) and of the e-image header of imsim/ccd.py:138-204.

The pixel work runs on the GPU through the C-ABI (`ims_readout_*`, `ims_flat_add`): bleed trails, dark current,
gain / flips / crosstalk into raw segments with prescan and overscan, parallel and serial charge-transfer
inefficiency, bias and read noise, truncation to int32 ADU.  Camera geometry comes from `camera.Camera`
(stand-in for lsst.obs.lsst, see camera.py); FITS files are written by `fits_io` (no astropy here)."""
import ctypes as C
import datetime
import math
import os

import numpy as np

from . import _abi, fits_io
from .camera import Camera

DARK_STREAM = (1 << 20) + 7          # iteration id of the dark-current Poisson stream (flat iterations: 0..niter, sky noise: 1<<20)
NTRANSFERS = 20

LSSTCam_filter_map = {"u": "u_24", "g": "g_6", "r": "r_57", "i": "i_39", "z": "z_20", "y": "y_10"}
ComCam_filter_map = {"u": "u_02", "g": "g_01", "r": "r_03", "i": "i_06", "z": "z_03", "y": "y_04"}
SIMONYI_TELESCOPE = "Simonyi Survey Telescope"


def section_keyword(bounds, flipx=False, flipy=False):
    """Image bounds as a NOAO image section keyword value (readout.py:152-160)."""
    xmin, xmax, ymin, ymax = bounds.xmin, bounds.xmax, bounds.ymin, bounds.ymax
    if flipx:
        xmin, xmax = xmax, xmin
    if flipy:
        ymin, ymax = ymax, ymin
    return "[%i:%i,%i:%i]" % (xmin, xmax, ymin, ymax)


def cte_band(npix, cti, ntransfers=NTRANSFERS):
    """The CTE matrix of readout.py:163-205 in banded form: band[i, d] = M[i, i-d], d = 0..ntransfers.
    Row i (1-based i+1 transfers): diagonal (1-cti)^i, and binom(i-1, i-j) (1-cti)^j cti^(i-j) for the up to
    `ntransfers` source pixels j before it."""
    from scipy.special import binom
    band = np.zeros((npix, ntransfers + 1))
    for i in range(1, npix + 1):
        band[i - 1, 0] = (1. - cti) ** i
        jmin = max(1, i - ntransfers)
        j = np.arange(jmin, i)
        band[i - 1, (i - j)] = binom(i - 1, i - j) * (1. - cti) ** j * cti ** (i - j)
    return band


def cte_matrix(npix, cti, ntransfers=NTRANSFERS):
    """Dense npix x npix matrix with q_i = sum_j M_ij q0_j (same signature as the reference's)."""
    band = cte_band(npix, cti, ntransfers)
    m = np.zeros((npix, npix))
    for d in range(ntransfers + 1):
        idx = np.arange(d, npix)
        m[idx, idx - d] = band[idx, d]
    return m


def mjd_to_isot(mjd):
    """astropy Time(mjd, format='mjd').isot for the header dates (calendar conversion only, millisecond digits)."""
    t = datetime.datetime(1858, 11, 17) + datetime.timedelta(days=float(mjd))
    us = int(round(t.microsecond / 1000.0)) * 1000
    if us >= 1000000:
        t, us = t + datetime.timedelta(seconds=1), 0
    return t.replace(microsecond=us).strftime("%Y-%m-%dT%H:%M:%S.") + "%03d" % (us // 1000)


def eimage_header(det_name, exptime, opsim_data=None, header_vals=None, camera="LsstCamSim"):
    """Keywords the image builder attaches to the e-image so that the raw file can be written from it alone
    (imsim/ccd.py:138-204).  Priority: header_vals, then opsim_data, then the reference's defaults."""
    vals = dict(header_vals or {})
    opsim = opsim_data or {}

    def parse(item, default):
        if item in vals:
            return vals.pop(item)
        v = opsim.get(item, default)
        return default if v is None else v
    mjd = parse("mjd", 51444.0)
    mjd_obs = parse("observationStartMJD", mjd)
    h = {"EXPTIME": exptime, "DET_NAME": det_name, "MJD": mjd, "MJD-OBS": (mjd_obs, "Start of exposure")}
    dayobs = (datetime.datetime(1858, 11, 17) + datetime.timedelta(days=float(mjd_obs) - 0.5)).strftime("%Y%m%d")
    ratel, dectel = parse("fieldRA", 0.0), parse("fieldDec", 0.0)
    airmass = parse("airmass", "N/A")
    h.update({"DAYOBS": dayobs, "SEQNUM": parse("seqnum", 0), "CONTRLLR": ("S", "simulated data"),
              "RUNNUM": parse("observationId", -999), "IMGTYPE": parse("image_type", "SKYEXP"),
              "REASON": parse("reason", "survey"), "RATEL": ratel, "DECTEL": dectel,
              "ROTTELPOS": parse("rotTelPos", 0.0), "FILTER": parse("band", "N/A/"), "CAMERA": camera,
              "HASTART": parse("HASTART", "N/A"), "HAEND": parse("HAEND", "N/A"), "AMSTART": airmass, "AMEND": airmass,
              "FOCUSZ": parse("focusZ", 0.0), "ALTITUDE": parse("altitude", "N/A"), "AZIMUTH": parse("azimuth", "N/A"),
              "ROTANGLE": parse("rotSkyPos", 0.0)})
    h.update(vals)                       # anything left in header_vals goes in as is
    return h


RUBIN_LONGITUDE_DEG, RUBIN_LATITUDE_DEG = -70.7494, -30.2446        # Simonyi telescope, east longitude / latitude
TAI_MINUS_UTC_S = 37.0                                               # since 2017-01-01


def _precession_matrix(mjd_tt):
    """Equatorial rotation J2000 -> mean equator and equinox of date (IAU 1976 angles zeta, z, theta)."""
    t = (mjd_tt - 51544.5) / 36525.0
    arc = math.pi / (180.0 * 3600.0)
    zeta = (2306.2181 * t + 0.30188 * t * t + 0.017998 * t ** 3) * arc
    z = (2306.2181 * t + 1.09468 * t * t + 0.018203 * t ** 3) * arc
    theta = (2004.3109 * t - 0.42665 * t * t - 0.041833 * t ** 3) * arc
    cz, sz, cZ, sZ, ct, st = math.cos(zeta), math.sin(zeta), math.cos(z), math.sin(z), math.cos(theta), math.sin(theta)
    return ((cz * ct * cZ - sz * sZ, -sz * ct * cZ - cz * sZ, -st * cZ),
            (cz * ct * sZ + sz * cZ, -sz * ct * sZ + cz * cZ, -st * sZ),
            (cz * st, -sz * st, ct))


def _to_date(P, ra, dec):
    v = (math.cos(dec) * math.cos(ra), math.cos(dec) * math.sin(ra), math.sin(dec))
    w = [sum(P[i][j] * v[j] for j in range(3)) for i in range(3)]
    return math.atan2(w[1], w[0]), math.asin(max(-1.0, min(1.0, w[2])))


def _position_angle(ra, dec, ra2, dec2):
    """Position angle of point 2 seen from point 1, from north through east (erfa.pas)."""
    dl = ra2 - ra
    return math.atan2(math.cos(dec2) * math.sin(dl), math.sin(dec2) * math.cos(dec) - math.cos(dec2) * math.sin(dec) * math.cos(dl))


def pointing_geometry(ra0, dec0, obsmjd, longitude=RUBIN_LONGITUDE_DEG, latitude=RUBIN_LATITUDE_DEG):
    """(altitude, azimuth, pseudo parallactic angle) in degrees of the ICRS direction (ra0, dec0) at TAI MJD obsmjd, geometric
    (no refraction, nutation or polar motion: a few 0.01 deg).  The pseudo parallactic angle is the position angle of the
    zenith at the boresight measured from ICRS north through east, which is what BatoidWCSFactory.pq of the reference
    returns (imsim/batoid_wcs.py:270-308): here the true parallactic angle of date minus the position angle that ICRS
    north has in the frame of date."""
    ra, dec, lat = math.radians(ra0), math.radians(dec0), math.radians(latitude)
    mjd_utc = obsmjd - TAI_MINUS_UTC_S / 86400.0
    t = (mjd_utc - 51544.5) / 36525.0                    # UT1 ~ UTC (|DUT1| < 0.9 s = 0.004 deg of sidereal time)
    gmst = (67310.54841 + (876600.0 * 3600.0 + 8640184.812866) * t + 0.093104 * t * t - 6.2e-6 * t ** 3) / 240.0   # degrees
    lst = math.radians((gmst + longitude) % 360.0)
    P = _precession_matrix(obsmjd + 32.184 / 86400.0)
    ra_d, dec_d = _to_date(P, ra, dec)
    ha = lst - ra_d
    sin_alt = math.sin(lat) * math.sin(dec_d) + math.cos(lat) * math.cos(dec_d) * math.cos(ha)
    alt = math.asin(max(-1.0, min(1.0, sin_alt)))
    az = math.atan2(-math.cos(dec_d) * math.sin(ha), math.sin(dec_d) * math.cos(lat) - math.cos(dec_d) * math.cos(ha) * math.sin(lat))
    q = math.atan2(math.sin(ha), math.tan(lat) * math.cos(dec_d) - math.sin(dec_d) * math.cos(ha))
    small = math.radians(10.0 / 3600.0)
    if math.pi / 2 - dec >= small:
        ra_n, dec_n = ra, dec + small
    else:
        ra_n, dec_n = ra + math.pi, math.pi / 2 - (small - (math.pi / 2 - dec))
    north_in_date = _position_angle(ra_d, dec_d, *_to_date(P, ra_n, dec_n))
    return math.degrees(alt), math.degrees(az) % 360.0, math.degrees(q - north_in_date)


_rotSkyPos_cache = {}


def compute_rotSkyPos(ra0, dec0, rottelpos, obsmjd, band="r", camera_name="LsstCamSim", **_):
    """The nominal rotation angle of the focal plane with respect to celestial north, with +y of the pixel coordinates as the
    reference direction: 270 - rotTelPos + pseudo parallactic angle, in [0, 360) degrees (imsim/readout.py:95-149).  The
    reference takes the angle from its batoid WCS factory (astropy / ERFA, with refraction); here it is the geometric value of
    `pointing_geometry`, good to a few hundredths of a degree."""
    key = (ra0, dec0, rottelpos, obsmjd, band, camera_name)
    if key not in _rotSkyPos_cache:
        theta = 270.0 - rottelpos + pointing_geometry(ra0, dec0, obsmjd)[2]
        _rotSkyPos_cache[key] = theta % 360.0
    return _rotSkyPos_cache[key]


class EImage:
    """The rendered CCD in electrons: a float64 device tensor [ny][nx] of integer counts (Renderer.image) with the
    header of `eimage_header` (the reference's galsim.ImageF + FitsHeader)."""

    def __init__(self, array, header):
        self.array = array
        self.header = header

    def write(self, file_name):
        """The e-image file (float32, like the reference's ImageF)."""
        import torch
        data = self.array.to(torch.float32).cpu().numpy()
        fits_io.write_fits(file_name, [(self.header, data)])


def get_primary_hdu(eimage, lsst_num, camera_name=None, added_keywords=None):
    """Primary header of the raw file with the keywords the LSST stack needs (readout.py:208-299).  ROTANGLE / ROTPA are
    recomputed from the pointing, the rotator angle and the time, as the reference does "instead of using likely
    inconsistent values from the instance catalog or opsim db" (readout.py:238-246, compute_rotSkyPos)."""
    eh = {k: (v[0] if isinstance(v, tuple) else v) for k, v in eimage.header.items()}
    exptime = eh["EXPTIME"]
    det_name = eh["DET_NAME"]
    raft, sensor = det_name.split("_")
    camera_name = camera_name or eh["CAMERA"]
    ratel, dectel, band = eh["RATEL"], eh["DECTEL"], eh["FILTER"]
    mjd_obs = eh["MJD-OBS"]
    mjd_end = mjd_obs + exptime / 86400.
    rotang = compute_rotSkyPos(ratel, dectel, eh.get("ROTTELPOS", 0.0), mjd_obs, band, camera_name=camera_name)
    comcam = camera_name == "LsstComCamSim"
    telcode = "CC" if comcam else "MC"
    h = {"RUNNUM": eh["RUNNUM"], "MJD": eh["MJD"], "DATE": mjd_to_isot(eh["MJD"]), "DAYOBS": eh["DAYOBS"],
         "SEQNUM": eh["SEQNUM"], "CONTRLLR": eh["CONTRLLR"], "EXPTIME": exptime, "DARKTIME": exptime, "TIMESYS": "TAI",
         "LSST_NUM": lsst_num, "IMGTYPE": eh["IMGTYPE"], "OBSTYPE": eh["IMGTYPE"], "REASON": eh["REASON"], "MONOWL": -1,
         "ROTANGLE": rotang,
         "FILTER": (ComCam_filter_map if comcam else LSSTCam_filter_map).get(band, "NONE"),
         "INSTRUME": "ComCamSim" if comcam else "LSSTCamSim", "RAFTBAY": raft, "CCDSLOT": sensor, "RA": ratel,
         "DEC": dectel, "ROTCOORD": "sky", "ROTPA": rotang, "TELESCOP": SIMONYI_TELESCOPE, "TELCODE": telcode,
         "RASTART": ratel, "DECSTART": dectel, "ELSTART": eh["ALTITUDE"], "AZSTART": eh["AZIMUTH"]}
    if eh["IMGTYPE"] == "SKYEXP":
        h["RADESYS"] = "ICRS"
        h["TRACKSYS"] = "RADEC"
    else:
        h["TRACKSYS"] = "LOCAL"
    h["OBSID"] = f"{telcode}_{eh['CONTRLLR']}_{eh['DAYOBS']}_{int(eh['SEQNUM']):06d}"
    h.update({"MJD-OBS": mjd_obs, "HASTART": eh["HASTART"], "HAEND": eh["HAEND"], "DATE-OBS": mjd_to_isot(mjd_obs),
              "DATE-END": mjd_to_isot(mjd_end), "AMSTART": eh["AMSTART"], "AMEND": eh["AMEND"], "ORIGIN": "imSim",
              "IMSIMVER": "imsim_amd", "CHIPID": det_name, "FOCUSZ": eh["FOCUSZ"]})
    h.update(added_keywords or {})
    return h


class CcdReadout:
    """eimage -> "raw" file with one image HDU per amplifier segment, with the electronics readout effects
    (readout.py:323-533; same constructor parameters, `rng` becomes an integer seed)."""

    def __init__(self, eimage, logger=None, camera=None, readout_time=2.0, dark_current=0.02, bias_level=1000.0,
                 scti=1.0e-6, pcti=1.0e-6, full_well=None, read_noise=None, bias_levels_file=None, added_keywords=None,
                 camera_obj=None):
        self.eimage = eimage
        hdr = {k: (v[0] if isinstance(v, tuple) else v) for k, v in eimage.header.items()}
        self.det_name = hdr["DET_NAME"]
        self.camera_name = camera if camera is not None else hdr["CAMERA"]
        self.logger = logger
        if camera_obj is None:
            camera_obj = Camera(self.camera_name, bias_levels_file=bias_levels_file)
        self.ccd = camera_obj[self.det_name]
        self.exptime = hdr["EXPTIME"]
        self.readout_time = readout_time
        self.dark_current = dark_current
        self.bias_level = bias_level if bias_levels_file is None else None
        self.full_well = self.ccd.full_well if full_well is None else full_well
        self.read_noise = read_noise
        amp_bounds = list(self.ccd.values())[0].raw_bounds
        self.scte_band = None if scti == 0 else cte_band(amp_bounds.xmax, scti)
        self.pcte_band = None if pcti == 0 else cte_band(amp_bounds.ymax, pcti)
        self.added_keywords = added_keywords
        self.amp_images = None

    # ---- descriptor shared with the library (and with the oracle in the tests) ----
    def descriptor(self):
        amps = list(self.ccd.values())
        a0 = amps[0]
        ro = _abi.Readout()
        ro.n_amps = len(amps)
        ro.seg_h, ro.seg_w = a0.bounds.numpyShape()
        ro.raw_h, ro.raw_w = a0.raw_bounds.numpyShape()
        ro.data_x0 = a0.raw_data_bounds.xmin - a0.raw_bounds.xmin
        ro.data_y0 = a0.raw_data_bounds.ymin - a0.raw_bounds.ymin
        ro.has_xtalk = 0 if self.ccd.xtalk is None else 1
        for k, amp in enumerate(amps):
            if amp.bounds.numpyShape() != (ro.seg_h, ro.seg_w):
                raise ValueError("amplifier segments of one CCD must have one shape")
            ro.amps[k].x0 = amp.bounds.xmin - self.ccd.bounds.xmin
            ro.amps[k].y0 = amp.bounds.ymin - self.ccd.bounds.ymin
            ro.amps[k].flip_x, ro.amps[k].flip_y = int(bool(amp.raw_flip_x)), int(bool(amp.raw_flip_y))
            ro.amps[k].gain = amp.gain
            ro.amps[k].bias_level = amp.bias_level if self.bias_level is None else self.bias_level
            ro.amps[k].read_noise = amp.read_noise if self.read_noise is None else self.read_noise
        if self.ccd.xtalk is not None:
            for i, row in enumerate(self.ccd.xtalk):
                for j, x in enumerate(row):
                    ro.xtalk[i * _abi.IMS_MAX_AMPS + j] = x
        return ro

    def dark_level(self):
        return self.dark_current * (self.exptime + self.readout_time)

    def midline_stop(self):
        return str(self.ccd.getSerial()).startswith("E2V")     # only e2v CCDs have the midline bleed stop

    def build_amp_images(self, seed):
        """bleed trails, dark current, amp segments in ADU and readout order, crosstalk, prescan / overscan, CTI,
        bias and read noise (readout.py:413-478), all on the device that holds the e-image.  Returns (and keeps
        in `amp_images`) an int32 device tensor [n_amps][raw_h][raw_w]."""
        import torch
        lib = _abi.load()
        img = self.eimage.array
        if not img.is_cuda:
            raise _abi.ImsimHipError("CcdReadout needs the e-image on the GPU (there is no CPU fallback)")
        if img.dtype != torch.float64 or not img.is_contiguous():
            raise ValueError("the e-image must be a contiguous float64 device tensor [ny][nx]")
        ny, nx = img.shape
        st = C.c_void_p(torch.cuda.current_stream(img.device).cuda_stream)
        ro = self.descriptor()
        flags = torch.empty((nx * ny + 15) // 16 * 16 + 16 * nx, dtype=torch.uint8, device=img.device)   # IMS_READOUT_SCRATCH_BYTES
        _abi.check(lib.ims_readout_bleed(img.data_ptr(), flags.data_ptr(), nx, ny, float(self.full_well),
                                         int(self.midline_stop()), st), "ims_readout_bleed")
        _abi.check(lib.ims_flat_add(None, None, float(self.dark_level()), 1.0, int(seed), DARK_STREAM, nx, ny,
                                    img.data_ptr(), None, st), "ims_flat_add (dark current)")
        shape = (ro.n_amps, ro.raw_h, ro.raw_w)
        a = torch.empty(shape, dtype=torch.float32, device=img.device)
        b = torch.empty(shape, dtype=torch.float32, device=img.device)
        _abi.check(lib.ims_readout_segments(img.data_ptr(), nx, ny, C.byref(ro), a.data_ptr(), st), "ims_readout_segments")
        for band, axis in ((self.pcte_band, 0), (self.scte_band, 1)):
            if band is None:
                continue
            bd = torch.from_numpy(np.ascontiguousarray(band)).to(img.device)
            _abi.check(lib.ims_readout_cte(a.data_ptr(), b.data_ptr(), C.byref(ro), bd.data_ptr(), band.shape[1], axis, st),
                       "ims_readout_cte")
            a, b = b, a
        out = torch.empty(shape, dtype=torch.int32, device=img.device)
        _abi.check(lib.ims_readout_finish(a.data_ptr(), C.byref(ro), int(seed), out.data_ptr(), st), "ims_readout_finish")
        self.amp_images = out
        return out

    def prepare_hdus(self, seed):
        """[(header, data)]: the primary HDU and one int32 image HDU per segment with EXTNAME, DATASEC, DETSEC
        (readout.py:480-527)."""
        amp_images = self.build_amp_images(seed).cpu().numpy()
        channels = "10 11 12 13 14 15 16 17 07 06 05 04 03 02 01 00".split()
        hdus = [(get_primary_hdu(self.eimage, self.ccd.getSerial(), camera_name=self.camera_name,
                                 added_keywords=self.added_keywords), None)]
        for amp_num in range(amp_images.shape[0]):
            amp_info = self.ccd["C" + channels[amp_num]]
            h = {"EXTNAME": "Segment" + channels[amp_num], "DATASEC": section_keyword(amp_info.raw_data_bounds),
                 "DETSEC": section_keyword(amp_info.bounds, flipx=amp_info.raw_flip_x, flipy=amp_info.raw_flip_y)}
            hdus.append((h, amp_images[amp_num]))
        return hdus

    @staticmethod
    def write_raw_file(hdus, file_name, compression=None):
        """compression="RICE_1": the segments as tile-compressed images, as the reference writes them
        (fits.CompImageHDU(..., compression_type='RICE_1'), imsim/readout.py:500-510); None: plain IMAGE extensions."""
        hdus[0][0]["OUTFILE"] = os.path.basename(file_name)
        if compression in (None, "", "none"):
            fits_io.write_fits(file_name, hdus)
        elif compression == "RICE_1":
            fits_io.write_fits_compressed(file_name, hdus)
        else:
            raise ValueError(f"unsupported compression {compression}")
