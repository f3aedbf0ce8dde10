"""phosim instance catalogs: host-side mirror of imsim/instcat.py (InstCatalog :162-338, getObj
:467-576) and of the header reader in imsim/opsim_data.py:158-206.

This is the step immediately before the hot path: it turns catalog lines into the
per-object rows the kernels consume.  Parsing is line-based like the reference; everything after
the parse (cull, flux, geometry) is vectorised.

Fluxes: the reference integrates each object's SED through the bandpass (SED files come from the
external `seds_170124` library, absent here); when an SED file cannot be found the documented
fallback is a flat-in-photons SED with the catalog normalisation at 500 nm for magnorm = 0
(imsim/instcat.py:167-170, :395-398), i.e. flux = FLUX_DENSITY_500 * integral(T) * 10^(-0.4 magnorm)
* pupil_area * exptime.
"""
import gzip
import math
import os

import numpy as np

RUBIN_AREA = math.pi * (418.0 ** 2 - 255.0 ** 2)          # cm^2 (imsim/utils.py:30)
# 0 ABmag at 500 nm in photons / nm / s / cm^2 (InstCatalog._flux_density, instcat.py:167-170)
FLUX_DENSITY_500 = 3.631e-20 * 2.99792458e17 / 500.0 ** 2 / (6.62607015e-27 * 2.99792458e10 / 500.0e-7)

DUST_INDEX = {"point": 13, "sersic2d": 17, "knots": 17, "streak": 16}     # instcat.py:210-216
HEADER_FLOATS = ("rightascension", "declination", "mjd", "altitude", "azimuth", "rotskypos", "rottelpos", "seeing",
                 "vistime", "moonalt", "moonra", "moondec", "moonphase", "dist2moon", "sunalt")
HEADER_INTS = ("filter", "nsnap", "obshistid", "seed", "seqnum")
BANDS = "ugrizy"


def fopen(path):
    return gzip.open(path, "rt") if path.endswith(".gz") else open(path, "rt")


def catalog_lines(path, _depth=0):
    """Lines of an instance catalog with every `includeobj <file>` directive replaced by the lines of that file, found
    relative to the including file's directory, plain or gzipped, recursively (fopen_generator, imsim/instcat.py:146-160:
    this is how real catalogs carry their objects).  A missing file raises OSError like the reference (:131-132)."""
    if _depth > 16:
        raise OSError(f"includeobj nesting too deep at {path}")
    if not os.path.isfile(path):
        raise OSError("File not found: %s" % path)
    here = os.path.split(os.path.abspath(path))[0]
    with fopen(path) as f:
        for line in f:
            if line.startswith("includeobj"):
                yield from catalog_lines(os.path.join(here, line.strip().split()[-1]), _depth + 1)
            else:
                yield line


def read_header(path):
    """The phosim commands at the top of an instance catalog -> dict with the OpsimData field
    names the configs use (imsim/opsim_data.py:158-206): fieldRA, fieldDec, altitude, azimuth,
    rotTelPos, rotSkyPos, band, seed, rawSeeing, exptime, mjd, observationId."""
    raw = {}
    with fopen(path) as f:
        for line in f:
            if line.startswith("object") or line.startswith("includeobj"):
                break                                  # the phosim commands precede the objects (opsim_data.py:158-206)
            tok = line.split()
            if len(tok) >= 2:
                raw[tok[0].lower()] = tok[1]
    out = {k: float(raw[k]) for k in HEADER_FLOATS if k in raw}
    out.update({k: int(float(raw[k])) for k in HEADER_INTS if k in raw})
    meta = dict(fieldRA=out.get("rightascension"), fieldDec=out.get("declination"), altitude=out.get("altitude"),
                azimuth=out.get("azimuth"), rotTelPos=out.get("rottelpos"), rotSkyPos=out.get("rotskypos"),
                mjd=out.get("mjd"), seed=out.get("seed"), rawSeeing=out.get("seeing"), exptime=out.get("vistime"),
                observationId=out.get("obshistid"), snap=out.get("nsnap", 1) - 1 if "nsnap" in out else 0)
    if "filter" in out:
        meta["band"] = BANDS[out["filter"]]
    if meta.get("mjd") is not None and meta.get("fieldRA") is not None:
        meta["HA"] = hour_angle(meta["mjd"], meta["fieldRA"])
    if meta.get("altitude") is not None:
        meta["airmass"] = get_airmass(meta["altitude"])
    return meta


BAND_WAVELENGTH = dict(u=365.49, g=480.03, r=622.20, i=754.06, z=868.21, y=991.66)


RUBIN_LONGITUDE = -70.749417          # deg east (lsst.obs.lsst SIMONYI_LOCATION)


def hour_angle(mjd, ra):
    """Local hour angle [deg] of right ascension `ra` [deg] at Rubin for the given MJD: local apparent sidereal time
    minus RA, not wrapped (OpsimDataLoader.getHourAngle, imsim/opsim_data.py:335-361, which asks astropy).  Here: GMST
    of IAU 1982 plus the two leading terms of the equation of the equinoxes -- good to a few arcseconds, far below what
    differential chromatic refraction or the header keywords can tell."""
    d = float(mjd) - 51544.5
    t = d / 36525.0
    gmst = 280.46061837 + 360.98564736629 * d + 0.000387933 * t * t - t ** 3 / 38710000.0
    omega = math.radians(125.04452 - 1934.136261 * t)
    lsun = math.radians(280.4665 + 36000.7698 * t)
    dpsi = (-17.20 * math.sin(omega) - 1.32 * math.sin(2.0 * lsun)) / 3600.0          # nutation in longitude [deg]
    last = (gmst + dpsi * math.cos(math.radians(23.4393)) + RUBIN_LONGITUDE) % 360.0
    return last - float(ra)


def read_opsim_db(file_name, visit, snap=0):
    """Visit metadata from an OpSim database (OpsimDataLoader._read_opsim_db, imsim/opsim_data.py:95-151): the row of
    `observations` for `visit`, plus the derived items the configs use (band, exptime per snap, mjd at the middle of
    the snap, HA, seed, rawSeeing, FWHMeff / FWHMgeom, seqnum counted from Rubin's day boundary)."""
    import sqlite3
    with sqlite3.connect(file_name) as con:
        columns = [r[0] for r in con.execute("select name from pragma_table_info('observations')")]
        rows = list(con.execute(f"select {','.join(columns)} from observations where observationId={int(visit)}"))
        if not rows:
            raise ValueError(f"visit {visit} not found in {file_name}")
        meta = dict(zip(columns, rows[0]))
        t0 = int(meta["observationStartMJD"] - 0.5) + 0.5
        earlier = con.execute(f"select numExposures from observations where {t0} <= observationStartMJD and "
                              f"observationId < {int(visit)}").fetchall()
    meta["snap"] = int(snap)
    meta["seqnum"] = sum(r[0] for r in earlier) + meta["snap"]
    if meta["snap"] >= meta["numExposures"]:
        raise ValueError("Invalid snap value: %d. For this visit, must have snap < %d" % (meta["snap"], meta["numExposures"]))
    meta["band"] = meta["filter"]
    meta["exptime"] = meta["visitExposureTime"] / meta["numExposures"]
    readout_time = (meta["visitTime"] - meta["visitExposureTime"]) / meta["numExposures"]
    meta["mjd"] = meta["observationStartMJD"] + (meta["snap"] * (meta["exptime"] + readout_time) + meta["exptime"] / 2) / 24. / 3600.
    meta["HA"] = hour_angle(meta["mjd"], meta["fieldRA"])
    meta["seed"] = meta["observationId"]
    meta["rawSeeing"] = meta["seeingFwhm500"]
    meta["FWHMeff"], meta["FWHMgeom"] = meta["seeingFwhmEff"], meta["seeingFwhmGeom"]
    return meta


def get_airmass(altitude):
    """Airmass from the altitude [deg], equation 3 of Krisciunas & Schaefer 1991 (OpsimDataLoader.getAirmass,
    imsim/opsim_data.py:242-260)."""
    alt = math.radians(altitude)
    return 1.0 / math.sqrt(1.0 - 0.96 * math.sin(0.5 * math.pi - alt) ** 2)


def fwhm_eff(raw_seeing, band, altitude):
    """Effective FWHM of a single Gaussian describing the PSF [arcsec] (OpsimDataLoader.FWHMeff, opsim_data.py:262-300)."""
    x = get_airmass(altitude)
    fwhm_atm = raw_seeing * (BAND_WAVELENGTH[band] / 500.0) ** (-0.3) * x ** 0.6
    fwhm_sys = 0.4 * x ** 0.6
    return 1.16 * math.sqrt(fwhm_sys ** 2 + 1.04 * fwhm_atm ** 2)


def fwhm_geom(raw_seeing, band, altitude):
    """Geometric size of the PSF [arcsec] (OpsimDataLoader.FWHMgeom, opsim_data.py:303-330: 0.822 FWHMeff + 0.052)."""
    return 0.822 * fwhm_eff(raw_seeing, band, altitude) + 0.052


def parse_objects(path, max_objects=None):
    """Parse the `object` lines (grammar: imsim/instcat.py:231-297).  Lines containing ' inf ' are
    skipped (:233); invalid objects (magnorm >= 50, sersic/knots with a < b, knots with npoints <= 0)
    are skipped (:276-286).  Returns a dict of arrays in file order.

    The text (includeobj files spliced in) goes through the library's tokenizer in one call
    (ims_parse_instcat_objects: ~25 ms per 100 000 objects against ~0.6 s of per-line Python); a catalog it cannot
    read, or a missing library, falls back to `_parse_objects_loop`, the line-by-line form it is tested against."""
    import ctypes as C
    with fopen(path) as f:
        text = f.read()
    if "includeobj" in text:                            # splice the included files in (rare: one directive per file)
        text = "".join(catalog_lines(path))
    text = text.encode()
    try:
        from . import _abi
        lib = _abi.load()
    except Exception:                                   # no library: the reference's way
        return _parse_objects_loop(path, max_objects)
    cap = text.count(b"\nobject") + (1 if text.startswith(b"object") else 0)
    if max_objects is not None:
        cap = min(cap, int(max_objects))
    num = np.zeros((max(cap, 1), 16), dtype=np.float64)
    kind = np.zeros(max(cap, 1), dtype=np.int32)
    span = np.zeros((max(cap, 1), 6), dtype=np.int64)
    lib.ims_parse_instcat_objects.restype = C.c_int64
    lib.ims_parse_instcat_objects.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    n = lib.ims_parse_instcat_objects(text, len(text), cap, num.ctypes.data, kind.ctypes.data, span.ctypes.data)
    if n < 0:
        return _parse_objects_loop(path, max_objects)   # raises what the line-by-line reader raises on that line
    num, kind, span = num[:n], kind[:n], span[:n]
    inst_dir = os.path.dirname(os.path.abspath(path))
    def tokens(j, rows=None):
        """the j-th span of every (given) object as a numpy array of str"""
        a = span[:, j] if rows is None else span[rows, j]
        b = a + (span[:, j + 1] if rows is None else span[rows, j + 1])
        cut = [text[x:y] for x, y in zip(a.tolist(), b.tolist())]
        return np.array(cut, dtype=bytes).astype(str) if cut else np.zeros(0, dtype=str)
    ids = tokens(0)
    sed = list(zip(tokens(2).tolist(), num[:, 3].tolist()))
    fits = np.full(n, "", dtype=object)
    stamps = np.flatnonzero(kind == 4)
    if len(stamps):
        fits[stamps] = [os.path.join(inst_dir, t) for t in tokens(4, stamps).tolist()]
    return dict(id=ids, ra=np.radians(num[:, 0]), dec=np.radians(num[:, 1]), magnorm=num[:, 2].copy(), sed=sed,
                dust=num[:, 11:15].copy().reshape(-1, 4), gamma1=num[:, 4].copy(), gamma2=num[:, 5].copy(), kappa=num[:, 6].copy(),
                objtype=kind.copy(), a=num[:, 7].copy(), b=num[:, 8].copy(), pa=num[:, 9].copy(), n=num[:, 10].copy(), fits_file=fits)


def _parse_objects_loop(path, max_objects=None):
    """parse_objects line by line (the form the tokenizer is tested against, and the fallback)"""
    ids, ra, dec, mag, sed, lens, kind, a, b, pa, n_or_pts, fits = [], [], [], [], [], [], [], [], [], [], [], []
    dust = []
    inst_dir = os.path.dirname(os.path.abspath(path))
    for line in catalog_lines(path):
        if " inf " in line or not line.startswith("object"):
            continue
        t = line.split()
        typ = t[12].lower()
        magnorm = float(t[4])
        di = DUST_INDEX.get(typ, 15)                              # where the dust fields start (:210-216, :274)
        if typ == "point":
            k, aa, bb, pp, nn = 0, 0.0, 0.0, 0.0, 0.0
        elif typ == "sersic2d":
            k, aa, bb, pp, nn = 1, float(t[13]), float(t[14]), float(t[15]), round(float(t[16]) * 20.0) / 20.0
        elif typ == "knots":
            k, aa, bb, pp, nn = 2, float(t[13]), float(t[14]), float(t[15]), float(int(t[16]))
        elif typ == "streak":
            k, aa, bb, pp, nn = 3, float(t[13]), float(t[14]), float(t[15]), 0.0
        elif t[12].endswith(".fits") or t[12].endswith(".fits.gz"):
            # galsim.InterpolatedImage(file relative to the catalog, scale=pixel_scale).rotate(-theta), :552-561
            k, aa, bb, pp, nn = 4, float(t[13]), 0.0, float(t[14]), 0.0
        else:
            k, aa, bb, pp, nn = 5, 0.0, 0.0, 0.0, 0.0            # unknown object type
        valid = magnorm < 50.0 and not (k in (1, 2) and aa < bb) and not (k == 2 and nn <= 0)
        if not valid:
            continue
        ids.append(t[1]); ra.append(float(t[2])); dec.append(float(t[3])); mag.append(magnorm)
        sed.append((t[5], float(t[6])))
        dust.append(parse_dust(t[di:]))
        lens.append((float(t[7]), float(t[8]), float(t[9])))
        kind.append(k); a.append(aa); b.append(bb); pa.append(pp); n_or_pts.append(nn)
        fits.append(os.path.join(inst_dir, t[12]) if k == 4 else "")
        if max_objects is not None and len(ids) >= max_objects:
            break
    lens = np.array(lens, dtype=np.float64).reshape(-1, 3)
    dust = np.array(dust, dtype=np.float64).reshape(-1, 4)
    return dict(id=np.array(ids), ra=np.radians(ra), dec=np.radians(dec), magnorm=np.array(mag), sed=sed, dust=dust,
                gamma1=lens[:, 0], gamma2=lens[:, 1], kappa=lens[:, 2], objtype=np.array(kind, dtype=np.int32),
                a=np.array(a), b=np.array(b), pa=np.array(pa), n=np.array(n_or_pts), fits_file=np.array(fits, dtype=object))


def parse_dust(params):
    """(internal Av, internal Rv, galactic Av, galactic Rv) from the trailing fields `none | MODEL Av Rv` x 2
    (InstCatalog.getDust, imsim/instcat.py:446-465)."""
    params = list(params)
    iav, irv, gav, grv = 0.0, 3.1, 0.0, 3.1
    if params and params[0].lower() != "none":
        iav, irv = float(params[1]), float(params[2])
        params = params[3:]
    else:
        params = params[1:]
    if params and params[0].lower() != "none":
        gav, grv = float(params[1]), float(params[2])
    return iav, irv, gav, grv


def to_catalog(parsed, img_wcs, xsize, ysize, bandpass_integral, exptime, pupil_area=RUBIN_AREA, edge_pix=100,
               sort_mag=True, flip_g2=True, sed_dir=None, inst_dir=None, bandpass=None, sed_points=257):
    """Cull to the CCD (+- edge_pix, instcat.py:243-259), compute nominal fluxes and the profile
    geometry (instcat.py:498-527, :433-444, :569-573) -> the catalog dict build_object_table takes.
    point, sersic2d, knots, streak and FITS-image objects reach the kernels; a FITS-image object whose file is
    missing is dropped with the count in n_dropped_unsupported."""
    from . import wcs as wcsmod
    vec = wcsmod.unit_vector(parsed["ra"], parsed["dec"]).T
    x, y = wcsmod.tansip_vec_to_pix(img_wcs, vec)
    on = (x >= 1 - edge_pix) & (x <= xsize + edge_pix) & (y >= 1 - edge_pix) & (y <= ysize + edge_pix)
    supported = np.isin(parsed["objtype"], (0, 1, 2, 3))
    files = parsed.get("fits_file")
    if files is not None:
        supported = supported | ((parsed["objtype"] == 4) & np.array([bool(f) and os.path.isfile(f) for f in files]))
    keep = on & supported
    idx = np.flatnonzero(keep)
    if sort_mag:
        idx = idx[np.argsort(parsed["magnorm"][idx], kind="stable")]       # brightest first (:328-338)
    flux = FLUX_DENSITY_500 * bandpass_integral * np.exp(-0.9210340371976184 * parsed["magnorm"][idx]) * pupil_area * exptime
    # SED x bandpass (imsim/instcat.py:380-431, :563-573): objects whose SED file is found get their own flux and their
    # own wavelength distribution (redshift and Milky-Way extinction applied); the others keep the flat-in-photons
    # fallback above and are reported in cat["missing_seds"]
    sed_tables, sed_table, missing = None, np.zeros(len(idx), dtype=np.int32), []
    if bandpass is not None and (sed_dir or inst_dir) and len(idx):
        from . import sed as sedmod
        names = np.array([parsed["sed"][i][0] for i in idx], dtype=object)
        z = np.array([parsed["sed"][i][1] for i in idx], dtype=np.float64)
        d = parsed["dust"][idx] if "dust" in parsed else np.tile([0.0, 3.1, 0.0, 3.1], (len(idx), 1))
        f0, tabs, missing = sedmod.object_spectra(names, z, d[:, 2], d[:, 3], bandpass[0], bandpass[1],
                                                  sedmod.SedLibrary(sed_dir, inst_dir), n_pts=sed_points)
        found = f0 >= 0.0
        flux = np.where(found, f0 * np.exp(-0.9210340371976184 * parsed["magnorm"][idx]) * pupil_area * exptime, flux)
        if found.any():
            sed_tables = tabs[found]
            sed_table[found] = 1 + np.arange(int(found.sum()))          # table 0 stays the flat fallback
    objtype = parsed["objtype"][idx]
    n = parsed["n"][idx]
    a, b = parsed["a"][idx], parsed["b"][idx]
    hlr = np.sqrt(np.maximum(a * b, 0.0))                                   # instcat.py:517
    q = np.where((a > 0) & (objtype != 4), b / np.where(a > 0, a, 1.0), 1.0)      # FITS stamps carry (scale, -, theta) there
    g2_sign = -1.0 if flip_g2 else 1.0
    kappa = parsed["kappa"][idx]
    g1 = parsed["gamma1"][idx] / (1.0 - kappa)
    g2 = g2_sign * parsed["gamma2"][idx] / (1.0 - kappa)
    mu = 1.0 / ((1.0 - kappa) ** 2 - (parsed["gamma1"][idx] ** 2 + parsed["gamma2"][idx] ** 2))
    sersic_n = np.where(objtype == 1, n, 0.0)
    cat = dict(x=x[idx], y=y[idx], nominal_flux=flux, mag=parsed["magnorm"][idx], hlr=hlr, q=q,
               pa=parsed["pa"][idx] if flip_g2 else -parsed["pa"][idx], g1=g1, g2=g2, mu=mu,
               kind=np.where(objtype == 0, 0, np.where(objtype == 2, 3, np.where(objtype == 3, 4, np.where(objtype == 4, 5,
                             np.where(np.isclose(sersic_n, 1.0), 1, 2))))).astype(np.int32),
               n_knots=np.where(objtype == 2, n, 0.0), box_length=np.where(objtype == 3, a, 0.0),
               box_width=np.where(objtype == 3, b, 0.0),
               sersic_n=sersic_n, obj_id=idx.astype(np.int64), object_id=parsed["id"][idx],
               sed_table=sed_table, sed_tables=sed_tables, missing_seds=missing)
    # what obj.evaluateAtWavelength(effective wavelength) carries in the reference: photons per nm (stamp_utils.py:176-220)
    cat["sb_flux"] = flux / bandpass_integral
    # FITS-image objects: every distinct file becomes one image profile (Scene.image_profiles)
    is_img = objtype == 4
    cat["images"], cat["image_index"] = [], np.zeros(len(idx), dtype=np.int64)
    cat["image_scale"], cat["image_extent"] = np.where(is_img, a, 0.0), np.zeros(len(idx))
    if is_img.any():
        from . import fits_io
        seen = {}
        for k in np.flatnonzero(is_img):
            fn = files[idx[k]]
            if fn not in seen:
                data = next(d for _, d in fits_io.read_fits(fn) if d is not None and d.ndim == 2)
                seen[fn] = len(cat["images"])
                cat["images"].append(np.asarray(data, dtype=np.float64))
            cat["image_index"][k] = seen[fn]
            shp = cat["images"][seen[fn]].shape
            cat["image_extent"][k] = max(shp) * a[k]
    cat["n_dropped_unsupported"] = int(np.count_nonzero(on & ~supported))
    return cat
