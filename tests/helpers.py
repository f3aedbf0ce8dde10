"""Shared helpers for the parity tests."""
import numpy as np

from imsim_amd import configs, catalog


def free_port():
    """a TCP port nobody listens on (asked of the kernel, not derived from the pid)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def small_case(n_obj=300, nx=512, ny=512, flux_seed=1, scene=None, **kw):
    scene = scene if scene is not None else configs.scene_c2(nx=nx, ny=ny)
    cat = catalog.synthetic_catalog(n_obj, nx=nx, ny=ny)
    phot = catalog.realize_fluxes(cat["nominal_flux"], flux_seed)
    objects, sizes = catalog.build_object_table(cat, phot, **kw)
    return scene, objects, sizes


def assert_bits_equal(a, b, what=""):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    same = a.view(np.uint8) == b.view(np.uint8)
    if not same.all():
        idx = np.flatnonzero(a.reshape(-1) != b.reshape(-1))
        raise AssertionError(f"{what}: {idx.size} of {a.size} values differ, first at {idx[:5]}: "
                             f"{a.reshape(-1)[idx[:5]]} vs {b.reshape(-1)[idx[:5]]}")


def c3_small_case(n_obj=150, n=512, flux_seed=1, scratch=2_000_000, **kw):
    """a small C3 scene (full op chain, Silicon sensor with tree rings) and its object table"""
    scene = configs.scene_c3(nx=n, ny=n, **kw)
    if scene.sensor is not None:
        scene.sensor.scratch_cells = scratch
    cat = catalog.synthetic_catalog(n_obj, nx=n, ny=n)
    phot = catalog.realize_fluxes(cat["nominal_flux"], flux_seed)
    objects, sizes = configs.c3_objects(cat, phot, scene)
    return scene, objects
