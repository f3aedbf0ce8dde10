"""Stamp layer: the per-object decisions of LSST_SiliconBuilder / LSST_PhotonsBuilder
(imsim/stamp.py) expressed over whole object tables.

The reference decides one object at a time inside GalSim's config machinery; here the same rules
are applied in bulk (numpy) to build the rows the GPU kernels consume:
  * phot_flux ~ Poisson(nominal_flux), skip if 0                      stamp.py:190-202
  * faint: nominal_flux < max_flux_simple -> no photon ops, no sensor stamp.py:435-465, 555-556
  * FFT only if nominal_flux >= 1e6 and fft_sb_thresh set             stamp.py:275-277
"""
import dataclasses
from enum import Enum, auto

import numpy as np


class ProcessingMode(Enum):
    FFT = auto()
    PHOT = auto()
    FAINT = auto()


@dataclasses.dataclass
class ObjectInfo:
    """Quantities of one object needed to pick its rendering mode (imsim/stamp.py:23-33)."""
    index: int
    phot_flux: float
    mode: ProcessingMode


def classify(nominal_flux, max_flux_simple=100.0, fft_sb_thresh=0.0, max_sb=None):
    """Vectorised build_obj mode decision (imsim/stamp.py:85-91, :275-277; psf_utils.py:212):
    FFT when the object is bright enough AND its peak surface brightness exceeds fft_sb_thresh,
    FAINT below max_flux_simple, PHOT otherwise.  Returns an array of ProcessingMode."""
    nominal_flux = np.asarray(nominal_flux, dtype=np.float64)
    mode = np.full(nominal_flux.shape, ProcessingMode.PHOT, dtype=object)
    mode[nominal_flux < max_flux_simple] = ProcessingMode.FAINT
    if fft_sb_thresh:
        cand = (nominal_flux >= 1.0e6) & (nominal_flux >= fft_sb_thresh)
        if max_sb is not None:
            cand &= np.asarray(max_sb) > fft_sb_thresh
        mode[cand] = ProcessingMode.FFT
    return mode


def object_infos(phot_flux, modes):
    return [ObjectInfo(i, f, m) for i, (f, m) in enumerate(zip(phot_flux, modes))]
