"""Host-side pieces of the bench configurations that need no GPU."""
import numpy as np

from imsim_amd import catalog, configs


def test_c5_catalog_and_table_keep_the_ccd_boundaries(monkeypatch):
    """C5 (189 CCDs x 10 k sources): the catalogs of the CCDs lie back to back, `ccd_offsets` survives the drop of
    zero-flux objects, and a slice (the CPU sample of bench.py) is a plain one-CCD table."""
    monkeypatch.setattr(configs, "N_CCD_FOCAL_PLANE", 3)
    scene = configs.scene_c2(nx=512, ny=512)
    scene.optics = configs.rubin_optics_struct(512, 512)
    cat = configs._c5_catalog(3 * 40, scene)
    assert list(cat.ccd_offsets) == [0, 40, 80, 120] and len(cat["x"]) == 120
    assert not np.array_equal(cat["x"][:40], cat["x"][40:80])            # every CCD has its own catalog
    phot = catalog.realize_fluxes(cat["nominal_flux"], 7)
    phot[[3, 50, 51]] = 0                                                   # SkipThisObject rows
    objects, _ = configs._c5_objects(cat, phot, scene)
    assert list(objects.ccd_offsets) == [0, 39, 77, 117] and len(objects) == 117
    assert objects[:10].ccd_offsets is None and np.asarray(objects[39:77]).dtype == objects.dtype
    for k in range(3):
        a, b = objects.ccd_offsets[k], objects.ccd_offsets[k + 1]
        assert np.array_equal(np.asarray(objects["obj_id"][a:b]), cat["obj_id"][40 * k:40 * (k + 1)][phot[40 * k:40 * (k + 1)] > 0])
