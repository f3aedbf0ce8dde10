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
