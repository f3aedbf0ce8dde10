"""Generate golden vectors for the spider-diffraction functions by importing the reference's
imsim/diffraction.py (numpy only) in THIS container.  Run once; the .npz is committed.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_diffraction_golden.py

Inputs follow tests/test_photon_ops.py:45-66 (`create_test_photon_array`: seed 42, r_uv in
U(2.5, 4.2), wavelength 577.6 nm); the random `distribution` callable is replaced by fixed normal
deviates passed in, so the outputs are deterministic functions of the stored inputs.
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/imsim"
pkg = types.ModuleType("imsim")
pkg.__path__ = [REF]
sys.modules["imsim"] = pkg
spec = importlib.util.spec_from_file_location("imsim.diffraction", os.path.join(REF, "diffraction.py"))
diffraction = importlib.util.module_from_spec(spec)
sys.modules["imsim.diffraction"] = diffraction
spec.loader.exec_module(diffraction)

n = 4000
rng = np.random.default_rng(seed=42)
r_uv = rng.uniform(2.5, 4.2, n)
phi_uv = rng.uniform(0.0, 2.0 * np.pi, n)
pos = np.c_[r_uv * np.cos(phi_uv), r_uv * np.sin(phi_uv)]
# a few points exactly on / very near the spider elements
pos[:8] = [[0.0, 2.558], [4.18, 0.0], [0.4 * np.sqrt(2), 0.0], [3.0, 3.0 - 0.4 * np.sqrt(2)],
           [2.0, -2.0], [-3.1, 0.2], [0.0, 3.3], [2.9, 2.9]]
wavelength = np.full(n, 577.6e-9)
wavelength[n // 2:] = rng.uniform(320e-9, 1050e-9, n - n // 2)
t = rng.uniform(0.0, 30.0, n)
gauss = rng.standard_normal(n)
th = rng.uniform(0, 0.03, n)
ph = rng.uniform(0, 2 * np.pi, n)
v = np.c_[np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), -np.cos(th)] / 1.000277

geom = diffraction.RUBIN_SPIDER_GEOMETRY
dist, nrm = diffraction.directed_dist(geom, pos.copy())
with np.errstate(divide="ignore"):
    phi = diffraction.phi_star(dist, wavelength)

lat, az, alt = np.deg2rad(-30.24463), np.deg2rad(45.0), np.deg2rad(89.9)
lat2, az2, alt2 = np.deg2rad(-30.24463), np.deg2rad(114.39), np.deg2rad(53.16)
out = dict(pos=pos, wavelength=wavelength, t=t, gauss=gauss, v=v, dist=dist, normal=nrm, phi_star=phi,
           thick_lines=geom.thick_lines, circles=geom.circles, omega=diffraction.OMEGA_EARTH)
for tag, (la, a, al) in dict(zenith=(lat, az, alt), visit=(lat2, az2, alt2)).items():
    rot = diffraction.prepare_field_rotation_matrix(latitude=la, azimuth=a, altitude=al)
    R = rot(t)
    distribution = lambda phi_s: phi_s * gauss      # noqa: E731  N(0, phi*^2) with fixed deviates
    with np.errstate(divide="ignore"):
        v_rot = diffraction.apply_diffraction_delta_field_rot(pos.copy(), v.copy(), t, wavelength, rot, geom, distribution)
        v_norot = diffraction.apply_diffraction_delta(pos.copy(), v.copy(), wavelength, geom, distribution)
    out[f"{tag}_lat_az_alt"] = np.array([la, a, al])
    out[f"{tag}_R"] = R
    out[f"{tag}_v_rot"] = v_rot
    out[f"{tag}_v_norot"] = v_norot
    out[f"{tag}_e_focal"] = diffraction.e_equatorial(latitude=la, azimuth=a, altitude=al)
    out[f"{tag}_e_z0"] = diffraction.prepare_e_z(la)[0]
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "diffraction_golden.npz"), **out)
print({k: np.shape(v_) for k, v_ in out.items()})
