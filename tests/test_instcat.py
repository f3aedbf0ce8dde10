"""Instance-catalog reader (imsim/instcat.py grammar) on a subset of the reference's example catalog."""
import os

import numpy as np

from imsim_amd import instcat

HERE = os.path.dirname(os.path.abspath(__file__))
INSTCAT = os.path.join(HERE, "golden", "example_instcat_subset.txt")


def test_header_fields():
    h = instcat.read_header(INSTCAT)
    assert h["band"] == "r" and h["seed"] == 398414 and h["observationId"] == 398414
    assert abs(h["fieldRA"] - 60.49045502638663) < 1e-12 and abs(h["rotTelPos"] - 40.0386345) < 1e-12
    assert h["exptime"] == 30.0 and abs(h["rawSeeing"] - 0.750181) < 1e-9


def test_object_grammar_and_validity_rules(tmp_path):
    p = instcat.parse_objects(INSTCAT)
    assert len(p["id"]) == 250 and set(np.unique(p["objtype"])) == {0, 1}
    g = p["objtype"] == 1
    assert np.all(p["a"][g] >= p["b"][g]) and np.all(np.isin(p["n"][g], (1.0, 4.0)))
    bad = tmp_path / "bad.txt"
    bad.write_text("object 1 60.4 -38.1 55.0 sed 0 0 0 0 0 0 point none none\n"                  # magnorm >= 50: skipped
                   "object 2 60.4 -38.1 20.0 sed 0 0 0 0 0 0 sersic2d 0.1 0.2 10 1 none none\n"    # a < b: skipped
                   "object 3 60.4 -38.1 20.0 sed 0 0 0 0 0 0 point inf none\n".replace("point inf", "point  inf ")
                   + "object 4 60.4 -38.1 20.0 sed 0 0 0 0 0 0 sersic2d 0.2 0.1 10 1.02 none none\n")
    q = instcat.parse_objects(str(bad))
    assert list(q["id"]) == ["4"] and q["n"][0] == 1.0          # n quantised to 0.05 (instcat.py:511-517)


def test_flux_and_geometry():
    p = instcat.parse_objects(INSTCAT)
    from imsim_amd import configs
    o = configs.rubin_optics_struct(4096, 4096)
    cat = instcat.to_catalog(p, o.img_wcs, 4096, 4096, 80.0, 30.0, sort_mag=True)
    assert np.all(np.diff(cat["mag"]) >= 0)                                       # brightest first
    k = 0
    expect = instcat.FLUX_DENSITY_500 * 80.0 * np.exp(-0.9210340371976184 * cat["mag"][k]) * instcat.RUBIN_AREA * 30.0
    assert abs(cat["nominal_flux"][k] / expect - 1) < 1e-12
    assert abs(instcat.FLUX_DENSITY_500 / 1.0959e4 - 1) < 1e-3
    gal = cat["kind"] > 0
    assert np.all(cat["hlr"][gal] > 0) and np.all((cat["q"][gal] > 0) & (cat["q"][gal] <= 1))
    assert np.all((cat["x"] > -100) & (cat["x"] < 4197))


def test_knots_and_streak_lines_reach_the_object_table(tmp_path):
    """instcat.py:487-496 (streak -> Box.rotate) and :529-546 (knots -> RandomKnots, sheared and lensed)"""
    from imsim_amd import catalog, configs
    f = tmp_path / "mixed.txt"
    f.write_text("object 11 60.49 -38.16 21.0 sed 0 0.02 0.01 0.03 0 0 knots 0.8 0.4 35.0 12 none none\n"
                 "object 12 60.50 -38.15 22.0 sed 0 0 0 0 0 0 streak 30.0 0.5 70.0 none none\n"
                 "object 13 60.49 -38.16 21.0 sed 0 0 0 0 0 0 knots 0.8 0.4 35.0 0 none none\n"         # npoints <= 0: skipped
                 "object 14 60.49 -38.16 21.0 sed 0 0 0 0 0 0 knots 0.3 0.4 35.0 5 none none\n")        # a < b: skipped
    p = instcat.parse_objects(str(f))
    assert list(p["id"]) == ["11", "12"] and list(p["objtype"]) == [2, 3]
    scene = configs.scene_c3(nx=4096, ny=4004, sensor=False)
    cat = instcat.to_catalog(p, scene.optics.img_wcs, 4096, 4004, 80.0, 30.0, sort_mag=False, edge_pix=10 ** 6)
    assert list(cat["kind"]) == [catalog.KIND_KNOTS, catalog.KIND_STREAK]
    assert cat["n_knots"][0] == 12 and cat["box_length"][1] == 30.0 and cat["box_width"][1] == 0.5
    objects, sizes = catalog.build_object_table(cat, np.array([1000, 1000]))
    assert list(objects["prof_table"]) == [-3, -2]
    np.testing.assert_allclose(objects["prof_scale"][0], np.sqrt(0.8 * 0.4) / 1.1774100225154747)     # hlr = sqrt(a b)
    assert objects["prof_aux"][0] == 12 and objects["prof_scale"][1] == 30.0 and objects["prof_aux"][1] == 0.5
    # streak: pure rotation by the position angle; knots: shear (q = b/a) times lens, not a rotation
    t = np.deg2rad(70.0)
    np.testing.assert_allclose(objects["jac"][1], [np.cos(t), -np.sin(t), np.sin(t), np.cos(t)])
    j = objects["jac"][0]
    assert abs(j[0] * j[3] - j[1] * j[2] - cat["mu"][0]) < 1e-12          # det = magnification
    assert sizes[1] >= 2 * 30.0 / 0.2


def test_airmass_and_fwhm_known_answers():
    """tests/test_FWHMgeom.py:20-60 of the reference: airmass(52.542 deg) = 1.24522984; FWHMeff / FWHMgeom of visit
    197356 (rawSeeing 0.5059960, r band, altitude 52.54199126195116065) within 0.03 of 0.8300650 / 0.7343130."""
    from imsim_amd import instcat
    assert abs(instcat.get_airmass(52.542) - 1.24522984) < 5e-8
    assert abs(instcat.fwhm_eff(0.5059960, "r", 52.54199126195116065) - 0.8300650) < 0.03
    assert abs(instcat.fwhm_geom(0.5059960, "r", 52.54199126195116065) - 0.7343130) < 0.03


def test_hour_angle():
    """OpsimDataLoader.getHourAngle (imsim/opsim_data.py:335-361): local apparent sidereal time minus RA.  At J2000.0
    Greenwich mean sidereal time is 280.4606 deg; the header's own altitude of the example catalog is recovered from
    HA, declination and Rubin's latitude."""
    import math
    from imsim_amd import instcat
    assert abs(instcat.hour_angle(51544.5, 280.46061837 + instcat.RUBIN_LONGITUDE)) < 0.01
    assert abs(instcat.hour_angle(51544.5 + 0.99726957, 10.0) - instcat.hour_angle(51544.5, 10.0)) < 0.01   # one sidereal day
    m = instcat.read_header(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "example_instcat_subset.txt"))
    lat, dec, ha = math.radians(-30.244639), math.radians(m["fieldDec"]), math.radians(m["HA"])
    alt = math.degrees(math.asin(math.sin(lat) * math.sin(dec) + math.cos(lat) * math.cos(dec) * math.cos(ha)))
    assert abs(alt - m["altitude"]) < 0.5


def test_metadata_from_opsim_db(tmp_path):
    """tests/test_instcat_parser.py:95-118 of the reference (visit 22184 of data/small_opsim.db): the same row, rebuilt
    here as a one-row database, gives the same derived metadata."""
    import sqlite3
    from imsim_amd import instcat
    row = {"observationId": 22184, "fieldRA": 65.00821243449612, "fieldDec": -33.20121826915378,
           "observationStartMJD": 60248.33830784654, "visitExposureTime": 30.0, "visitTime": 34.0, "numExposures": 2, "filter": "z",
           "altitude": 68.29298936358147, "azimuth": 255.57244075182493, "rotSkyPos": 287.86098563593913,
           "rotTelPos": 16.923971097101944, "airmass": 1.0763250938907971, "seeingFwhm500": 0.5833528497734382,
           "seeingFwhmEff": 0.7790170013788277, "seeingFwhmGeom": 0.6923519751333964}
    db = str(tmp_path / "opsim.db")
    with sqlite3.connect(db) as con:
        con.execute("create table observations (%s)" % ", ".join(f"{k} {'text' if isinstance(v, str) else 'real'}" for k, v in row.items()))
        con.execute("insert into observations values (%s)" % ",".join("?" * len(row)), list(row.values()))
        early = dict(row, observationId=22100, observationStartMJD=60248.2)        # an earlier visit of the same night
        con.execute("insert into observations values (%s)" % ",".join("?" * len(row)), list(early.values()))
    md = instcat.read_opsim_db(db, 22184, snap=0)
    assert md["observationId"] == 22184 and md["fieldRA"] == 65.00821243449612 and md["fieldDec"] == -33.20121826915378
    assert md["rawSeeing"] == 0.5833528497734382 and md["FWHMeff"] == 0.7790170013788277 and md["FWHMgeom"] == 0.6923519751333964
    assert abs(md["mjd"] - (60248.33830784654 + 7.5 / 86400)) < 1e-9 and md["airmass"] == 1.0763250938907971
    assert md["band"] == "z" and md["exptime"] == 15 and md["seed"] == 22184 and md["seqnum"] == 2
    assert instcat.read_opsim_db(db, 22184, snap=1)["mjd"] > md["mjd"] and instcat.read_opsim_db(db, 22184, snap=1)["seqnum"] == 3
    import pytest
    with pytest.raises(ValueError):
        instcat.read_opsim_db(db, 22184, snap=2)
    with pytest.raises(ValueError):
        instcat.read_opsim_db(db, 1)


# ---- includeobj / gzip recursion (imsim/instcat.py:115-160) and SED x bandpass fluxes (:380-431) ----
SED_DIR = os.path.join(HERE, "data", "sed_library")
# the two objects of the reference's tests/data/phosim_combined.txt whose SED files are kept as fixtures
GALAXY = ("object 61441544815642 52.96594129201541 -27.87740393606027 22.5915298 galaxySED/Exp.25E09.02Z.spec.gz 1.30053163 "
          "0.00372763043 0.00272741678 0.0108909156 0 0 sersic2d 0.586330473 0.584176481 65.9815598 4 CCM 0.2837847 2.7550051 "
          "CCM 0.0232418057 3.1\n")
STAR = ("object 1605472734212 53.62218479644643 -26.3123565130416 23.9253783 starSED/kurucz/km10_5250.fits_g15_5250.gz 0 0 0 0 0 0 "
        "point none CCM 0.03380581 3.1\n")


def test_includeobj_recursion_with_gzipped_children(tmp_path):
    import gzip
    import pytest
    sub = tmp_path / "sub"
    sub.mkdir()
    with gzip.open(sub / "stars.txt.gz", "wt") as f:
        f.write(STAR)
        f.write("includeobj deeper.txt\n")                       # relative to the including file's directory
    (sub / "deeper.txt").write_text(STAR.replace("1605472734212", "77"))
    parent = tmp_path / "parent.txt"
    parent.write_text("rightascension 53.0\ndeclination -27.5\nfilter 2\nseed 11\nobshistid 11\nvistime 30\n"
                      + GALAXY + "includeobj sub/stars.txt.gz\n")
    lines = list(instcat.catalog_lines(str(parent)))
    assert sum(ln.startswith("object") for ln in lines) == 3 and not any(ln.startswith("includeobj") for ln in lines)
    p = instcat.parse_objects(str(parent))
    assert list(p["id"]) == ["61441544815642", "1605472734212", "77"]
    assert instcat.read_header(str(parent))["band"] == "r"
    np.testing.assert_allclose(p["dust"][0], [0.2837847, 2.7550051, 0.0232418057, 3.1])
    np.testing.assert_allclose(p["dust"][1], [0.0, 3.1, 0.03380581, 3.1])
    (tmp_path / "broken.txt").write_text("includeobj nowhere.txt\n")
    with pytest.raises(OSError):
        instcat.parse_objects(str(tmp_path / "broken.txt"))


def test_sed_normalisation_flux_redshift_and_dust(tmp_path):
    """tests/test_instcat_parser.py:239-260 of the reference: the cached SED holds the magnorm = 0 flux density at 500 nm,
    and an object's flux is sed.calculateFlux(bandpass) * 10^(-0.4 magnorm) * area * exptime."""
    from imsim_amd import sed as sedmod, tables
    lib = sedmod.SedLibrary(SED_DIR, None)
    star = lib.get("starSED/kurucz/km10_5250.fits_g15_5250.gz")
    gal = lib.get("galaxySED/Exp.25E09.02Z.spec.gz")
    assert abs(float(star(500.0)) / instcat.FLUX_DENSITY_500 - 1) < 1e-12 and abs(float(gal(500.0)) / instcat.FLUX_DENSITY_500 - 1) < 1e-12
    assert lib.get("no/such/file.gz") is None
    wl, thr = tables.synthetic_r_band()
    names = ["starSED/kurucz/km10_5250.fits_g15_5250.gz", "galaxySED/Exp.25E09.02Z.spec.gz", "galaxySED/Exp.25E09.02Z.spec.gz",
             "galaxySED/Exp.25E09.02Z.spec.gz", "missing.gz"]
    z = np.array([0.0, 0.0, 1.30053163, 1.30053163, 0.0])
    av = np.array([0.0, 0.0, 0.0, 0.5, 0.0])
    flux, tabs, missing = sedmod.object_spectra(names, z, av, np.full(5, 3.1), wl, thr, lib, n_pts=129)
    assert missing == ["missing.gz"] and flux[4] == -1.0
    # independent quadrature of the same integrand
    grid = np.linspace(wl[0], wl[-1], 20001)
    T = np.interp(grid, wl, thr)
    want0 = np.trapezoid(star(grid) * T, grid)
    np.testing.assert_allclose(flux[0], want0, rtol=2e-4)
    want2 = np.trapezoid(gal.at_redshift(1.30053163)(grid) * T, grid)
    np.testing.assert_allclose(flux[2], want2, rtol=2e-4)
    # A(r) / A(V) of the CCM curve is ~ 0.86 in the r band: Av = 0.5 dims the object by 10^(-0.4 * 0.43)
    assert 0.64 < flux[3] / flux[2] < 0.70
    assert abs(sedmod.ccm89(np.array([549.5]), 3.1)[0] - 1.0) < 0.01             # the curve is normalised at V
    # wavelength tables: monotone, inside the band, and the reddened one is shifted to the red
    assert np.all(np.diff(tabs[:4], axis=1) >= 0) and tabs[:4].min() >= wl[0] and tabs[:4].max() <= wl[-1]
    assert tabs[3][64] >= tabs[2][64]
    # end to end through the catalog reader: found SEDs replace the flat fallback, missing ones keep it and are reported
    f = tmp_path / "cat.txt"
    f.write_text(GALAXY + STAR + STAR.replace("starSED/kurucz/km10_5250.fits_g15_5250.gz", "starSED/none.gz").replace("1605472734212", "5"))
    p = instcat.parse_objects(str(f))
    from imsim_amd import configs
    o = configs.rubin_optics_struct(4096, 4096)
    integral = float(np.trapezoid(thr, wl))
    kw = dict(sort_mag=False, edge_pix=10 ** 7)
    flat = instcat.to_catalog(p, o.img_wcs, 4096, 4096, integral, 30.0, **kw)
    cat = instcat.to_catalog(p, o.img_wcs, 4096, 4096, integral, 30.0, sed_dir=SED_DIR, bandpass=(wl, thr), sed_points=129, **kw)
    assert cat["missing_seds"] == ["starSED/none.gz"] and list(cat["sed_table"]) == [1, 2, 0] and cat["sed_tables"].shape == (2, 129)
    assert cat["nominal_flux"][2] == flat["nominal_flux"][2]
    mw = sedmod.extinction_factor(grid, [0.03380581], [3.1])[0]
    want = np.trapezoid(star(grid) * mw * T, grid) * np.exp(-0.9210340371976184 * 23.9253783) * instcat.RUBIN_AREA * 30.0
    np.testing.assert_allclose(cat["nominal_flux"][1], want, rtol=3e-4)
    assert cat["nominal_flux"][1] != flat["nominal_flux"][1]
