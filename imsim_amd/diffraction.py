"""Spider diffraction: host-side geometry and pointing setup for the GPU kernel.

Mirrors the data definitions of imsim/diffraction.py (Geometry :15-29, RUBIN_SPIDER_GEOMETRY
:32-42, OMEGA_EARTH :280, e_equatorial :387-415, prepare_e_z :284-304); the per-photon arithmetic
(directed_dist, phi_star, field rotation, apply_delta_v) runs in csrc/ims_photon.h::diffract.
"""
import dataclasses
import math

import numpy as np

# Earth rotation rate 2 pi / (sidereal day) [rad/s] (imsim/diffraction.py:279-280)
OMEGA_EARTH = 7.292115826090781e-05


@dataclasses.dataclass
class Geometry:
    """2-D pupil-plane geometry: thick lines [nx, ny, d, thickness] and circles [x, y, r]."""
    thick_lines: np.ndarray
    circles: np.ndarray


_S = 1.0 / math.sqrt(2.0)
RUBIN_SPIDER_GEOMETRY = Geometry(
    thick_lines=np.array([[_S, _S, -0.4, 0.025], [-_S, _S, -0.4, 0.025],
                          [_S, _S, 0.4, 0.025], [-_S, _S, 0.4, 0.025]]),
    circles=np.array([[0.0, 0.0, 2.558], [0.0, 0.0, 4.18]]),
)


def zenith_direction(latitude):
    """Direction to the observer's zenith at t = 0 in the equatorial frame (x: observer's
    meridian projected on the equator, z: Earth axis)."""
    return np.array([math.cos(latitude), 0.0, math.sin(latitude)])


def e_equatorial(latitude, altitude, azimuth):
    """Unit vector of the pointing (altitude, azimuth) seen from `latitude`, equatorial frame."""
    zen = zenith_direction(latitude)
    east = np.array([0.0, 1.0, 0.0])
    north = np.array([-zen[2], 0.0, zen[0]])
    ca = math.cos(altitude)
    return east * ca * math.sin(azimuth) + north * ca * math.cos(azimuth) + zen * math.sin(altitude)


def fill_optics(optics, latitude, azimuth, altitude, geometry=RUBIN_SPIDER_GEOMETRY):
    """Write the diffraction block of an _abi.Optics struct."""
    lines, circles = np.atleast_2d(geometry.thick_lines), np.atleast_2d(geometry.circles)
    if len(lines) > 8 or len(circles) > 4:
        raise ValueError("spider geometry too large for ims_optics_t")
    optics.n_lines, optics.n_circles = len(lines), len(circles)
    for k, row in enumerate(lines):
        for m in range(4):
            optics.lines[k][m] = float(row[m])
    for k, row in enumerate(circles):
        for m in range(3):
            optics.circles[k][m] = float(row[m])
    ez0 = zenith_direction(latitude)
    ef = e_equatorial(latitude, altitude, azimuth)
    for m in range(3):
        optics.e_z0[m] = float(ez0[m])
        optics.e_focal[m] = float(ef[m])
    optics.cos_lat, optics.sin_lat = math.cos(latitude), math.sin(latitude)
    optics.omega = OMEGA_EARTH
    return optics
