"""Tree rings: host-side mirror of imsim/treerings.py (TreeRingRadialFunction :14-68, TreeRings
:71-218).  The radial function is tabulated on 2667 points (r in [0, 8000] px, step 3) exactly as
`galsim.LookupTable.from_func(..., x_min=0, x_max=8000, npoints=2667)` does (:100-103, :192-194);
the GPU applies it to the pixel boundary points (csrc k_init_boundaries)."""
import os
import warnings

import numpy as np


class TreeRingsError(Exception):
    pass


class TreeRingRadialFunction:
    """Radial function describing tree rings in a CCD (imsim/treerings.py:14-68)."""

    def __init__(self, info_block):
        items = info_block[1].split()
        self.A = float(items[6])
        self.B = float(items[7])
        self.cfreqs, self.cphases, self.sfreqs, self.sphases = np.genfromtxt(info_block[3:]).T

    def __call__(self, r):
        r = np.asarray(r, dtype=np.float64)
        shift = np.zeros_like(r)
        for f, p in zip(self.cfreqs, self.cphases):
            shift = shift + np.sin(2 * np.pi * (r / f) + p) * f / (2.0 * np.pi)
        for f, p in zip(self.sfreqs, self.sphases):
            shift = shift - np.cos(2 * np.pi * (r / f) + p) * f / (2.0 * np.pi)
        return shift * (self.A + self.B * r ** 4) * 0.01

    def dfdr(self, r):
        r = np.asarray(r, dtype=np.float64)
        val = np.zeros_like(r)
        for f, p in zip(self.cfreqs, self.cphases):
            val = val + np.cos(2 * np.pi * (r / f) + p)
        for f, p in zip(self.sfreqs, self.sphases):
            val = val + np.sin(2 * np.pi * (r / f) + p)
        val = val * (self.A + self.B * r ** 4) * 0.01
        return val + self(r) / (self.A + self.B * r ** 4) * self.B * r ** 3 / 4.0


class TreeRings:
    """Reads a tree_ring_parameters file; `get_center` / `get_func` per detector name."""
    numfreqs = 20
    r_max = 8000.0
    dr = 3.0

    def __init__(self, file_name, only_dets=None, data_dir=None, defer_load=True):
        self.file_name = file_name
        if not os.path.isfile(self.file_name) and data_dir is not None:
            self.file_name = os.path.join(data_dir, "tree_ring_data", file_name)
        if not os.path.isfile(self.file_name):
            raise OSError("TreeRing file %s not found" % file_name)
        self.npoints = int(self.r_max / self.dr) + 1
        self._read_info_blocks()
        self.info = {}
        if not defer_load:
            self.fill_dict(only_dets)

    def _read_info_blocks(self):
        with open(self.file_name) as f:
            lines = f.readlines()
        block_size = self.numfreqs + 3
        self.info_blocks = {}
        for iblock in range(len(lines) // block_size):
            block = lines[iblock * block_size:(iblock + 1) * block_size]
            items = block[1].split()
            self.info_blocks["R%s%s_S%s%s" % tuple(items[:4])] = block

    def fill_dict(self, only_dets=None):
        for det_name in (only_dets if only_dets is not None else self.info_blocks.keys()):
            if det_name not in self.info_blocks:
                continue
            block = self.info_blocks[det_name]
            items = block[1].split()
            center = (float(items[4]) + 2048.5, float(items[5]) + 2048.5)
            func = TreeRingRadialFunction(block)
            r = np.linspace(0.0, self.r_max, self.npoints)
            self.info[det_name] = (center, TreeRingTable(r, func(r)))

    def get_dfdr(self, det_name):
        return TreeRingRadialFunction(self.info_blocks[det_name]).dfdr

    def get_center(self, det_name):
        if det_name not in self.info:
            self.fill_dict((det_name,))
        if det_name in self.info:
            return self.info[det_name][0]
        warnings.warn("No treering information available for %s.  Setting treering_center to (0, 0)." % det_name)
        return (0.0, 0.0)

    def get_func(self, det_name):
        if det_name not in self.info:
            self.fill_dict((det_name,))
        if det_name in self.info:
            return self.info[det_name][1]
        warnings.warn("No treering information available for %s.  Setting treering_func to None." % det_name)
        return None


class TreeRingTable:
    """Uniform lookup table with natural-cubic-spline interpolation: what
    `galsim.LookupTable.from_func` builds by default (interpolant='spline') and the sensor consumes.
    `f2` holds the spline's second derivatives at the knots for the device-side evaluation."""

    def __init__(self, r, f):
        from scipy.interpolate import CubicSpline
        self.r = np.asarray(r, dtype=np.float64)
        self.f = np.ascontiguousarray(f, dtype=np.float64)
        self.dr = float(self.r[1] - self.r[0])
        self._spline = CubicSpline(self.r, self.f, bc_type="natural")
        self.f2 = np.ascontiguousarray(self._spline(self.r, 2), dtype=np.float64)

    def __call__(self, r):
        return self._spline(r)


def simple_treerings(amplitude=0.5, period=100.0, r_max=8000.0, dr=3.0):
    """f(r) = amplitude * cos(2 pi r / period) tabulated like TreeRings (tests/test_flats.py:119-134)."""
    r = np.linspace(0.0, r_max, int(r_max / dr) + 1)
    return TreeRingTable(r, amplitude * np.cos(r / period * 2.0 * np.pi))
