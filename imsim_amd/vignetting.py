"""Vignetting: radial B-spline model of the throughput across the focal plane (imsim/vignetting.py:10-120).

Used in two places of the path's surroundings: the flux of FFT-drawn objects (for photon-shot objects vignetting emerges
from the ray trace; get_fft_psf_maybe, imsim/psf_utils.py:220-233) and the sky background per pixel
(LSST_ImageBuilder.addNoise, imsim/lsst_image.py:172-176).  The spline data (knots, coefficients, degree) are imSim's
`data/LSST*_vignetting_data.json`.  Detector positions: the reference asks lsst.afw.cameraGeom; here the nominal LSSTCam
layout stands in (rafts on a 127 mm pitch, CCDs on a 42.25 mm pitch inside a raft, 10 micron pixels, no rotations of the
science rafts), which is what the radius of a pixel from the focal-plane centre needs.
"""
import json
import os

import numpy as np
from scipy import interpolate

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
RAFT_PITCH_MM = 127.0
CCD_PITCH_MM = 42.25
PIXEL_MM = 0.01


def detector_center_mm(det_name):
    """Focal-plane position [mm] of the centre of science CCD 'Rxy_Sxy' in the nominal layout (R22_S11 at the origin)."""
    raft, sensor = det_name.split("_")
    rx, ry = int(raft[1]), int(raft[2])
    sx, sy = int(sensor[1]), int(sensor[2])
    return ((rx - 2) * RAFT_PITCH_MM + (sx - 1) * CCD_PITCH_MM, (ry - 2) * RAFT_PITCH_MM + (sy - 1) * CCD_PITCH_MM)


class Vignetting:
    _req_params = {"file_name": str}

    def __init__(self, file_name, data_dir=None):
        if not os.path.isfile(file_name):
            cand = os.path.join(data_dir or DATA_DIR, file_name)
            if not os.path.isfile(cand):
                raise OSError(f"Vignetting data file {file_name} not found.")
            file_name = cand
        with open(file_name) as fobj:
            t, c, k = json.load(fobj)
        self.spline_model = interpolate.BSpline(np.asarray(t, dtype=np.float64), np.asarray(c, dtype=np.float64), int(k))
        self.value_at_zero = float(self.spline_model(0))       # the profile is normalised to the centre of the focal plane

    def apply_to_radii(self, radii):
        return self.spline_model(radii) / self.value_at_zero

    @staticmethod
    def get_pixel_radii(det_name, nx, ny, n_quarter=0):
        """Distance [mm] of every pixel of a CCD from the focal-plane centre, [ny][nx] (imsim/vignetting.py:44-84)."""
        cx, cy = detector_center_mm(det_name)
        dx = PIXEL_MM * (np.arange(-nx / 2, nx / 2, dtype=float) + 0.5)
        dy = PIXEL_MM * (np.arange(-ny / 2, ny / 2, dtype=float) + 0.5)
        n_rot = n_quarter % 4
        if n_rot == 0:
            xarr, yarr = np.meshgrid(cx + dx, cy + dy)
        elif n_rot == 1:
            yarr, xarr = np.meshgrid(cy + dx, cx - dy)
        elif n_rot == 2:
            xarr, yarr = np.meshgrid(cx - dx, cy - dy)
        else:
            yarr, xarr = np.meshgrid(cy - dx, cx + dy)
        return np.sqrt(xarr ** 2 + yarr ** 2)

    def __call__(self, det_name, nx, ny, n_quarter=0):
        return self.apply_to_radii(self.get_pixel_radii(det_name, nx, ny, n_quarter))

    def at_pixel(self, det_name, x, y, nx, ny):
        """Vignetting factor at image positions (x, y) (1-based pixel coordinates) of a CCD: what at_sky_coord returns
        once the WCS has mapped the sky position to the image (imsim/vignetting.py:90-120)."""
        cx, cy = detector_center_mm(det_name)
        fx = cx + (np.asarray(x, dtype=np.float64) - 0.5 - nx / 2.0) * PIXEL_MM
        fy = cy + (np.asarray(y, dtype=np.float64) - 0.5 - ny / 2.0) * PIXEL_MM
        return self.apply_to_radii(np.hypot(fx, fy))
