"""Cosmic rays on the e-image (imsim/cosmic_rays.py:16-185): hits harvested from real dark frames, stored as footprints of
serial-direction spans in a FITS binary table (`data/cosmic_rays_itl_2017.fits.gz`, imSim's catalog), painted at random
places of the CCD -- the number per exposure Poisson distributed around exptime x ccd_rate x (image pixels / catalog
sensor pixels).

The reference paints span by span into a numpy array; here the spans of ALL hits of an exposure are flattened on the
host into (pixel index, electrons) pairs and added to the device image in one scatter-add."""
import os
from collections import namedtuple, defaultdict

import numpy as np

from . import fits_io

CR_Span = namedtuple("CR_Span", "x0 y0 pixel_values".split())
DEFAULT_CATALOG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "cosmic_rays_itl_2017.fits.gz")


class CosmicRays(list):
    """Each element is one cosmic ray: a list of CR_Span(x0, y0, pixel_values).  Attributes num_pix (pixels of the
    sensors the hits were taken from), exptime (summed dark time [s]), ccd_rate (hits per second per CCD)."""

    def __init__(self, ccd_rate=None, catalog_file=DEFAULT_CATALOG):
        super().__init__()
        self._read_catalog(catalog_file, ccd_rate)

    @classmethod
    def read_catalog(cls, catalog_file, ccd_rate=None, extname="COSMIC_RAYS"):
        ret = cls.__new__(cls)
        list.__init__(ret)
        ret._read_catalog(catalog_file, ccd_rate, extname=extname)
        return ret

    def _read_catalog(self, catalog_file, ccd_rate, extname="COSMIC_RAYS"):
        table = None
        for hdr, data in fits_io.read_fits(catalog_file):
            if hdr.get("XTENSION") == "BINTABLE" and str(hdr.get("EXTNAME", "")).strip() == extname:
                table, header = data, hdr
        if table is None:
            raise OSError(f"no {extname} table in {catalog_file}")
        self.num_pix = header["NUM_PIX"]
        self.exptime = header["EXPTIME"]
        crs = defaultdict(list)
        for fp, x0, y0, vals in zip(table["fp_id"], table["x0"], table["y0"], table["pixel_values"]):
            crs[int(fp)].append(CR_Span(int(x0), int(y0), np.asarray(vals)))
        self.extend(crs.values())
        self.ccd_rate = float(len(self)) / self.exptime if ccd_rate is None else ccd_rate

    # -- the reference's array interface (used by its tests) --
    def paint_cr(self, image_array, rng, index=None, pixel=None):
        if index is None:
            index = int(rng.random() * len(self))
        cr = self[index]
        if pixel is None:
            pixel = (int(rng.random() * image_array.shape[1]), int(rng.random() * image_array.shape[0]))
        for span in cr:
            for dx, value in enumerate(span.pixel_values):
                row, col = pixel[1] + span.y0 - cr[0].y0, pixel[0] + span.x0 - cr[0].x0 + dx
                if 0 <= row < image_array.shape[0] and 0 <= col < image_array.shape[1]:
                    image_array[row, col] += value
        return image_array

    def draw_hits(self, shape, rng, exptime=30.0, num_crs=None):
        """(flat pixel index, electrons) of every span pixel of this exposure's hits that falls on an image of `shape`"""
        ny, nx = shape
        if num_crs is None:
            num_crs = int(rng.poisson(exptime * self.ccd_rate * float(nx * ny) / self.num_pix))
        idx, val = [], []
        for _ in range(num_crs):
            cr = self[int(rng.random() * len(self))]
            px, py = int(rng.random() * nx), int(rng.random() * ny)
            for span in cr:
                row = py + span.y0 - cr[0].y0
                cols = px + span.x0 - cr[0].x0 + np.arange(len(span.pixel_values))
                ok = (cols >= 0) & (cols < nx) & (0 <= row < ny)
                idx.append(row * nx + cols[ok])
                val.append(np.asarray(span.pixel_values, dtype=np.float64)[ok])
        if not idx:
            return np.zeros(0, dtype=np.int64), np.zeros(0)
        return np.concatenate(idx).astype(np.int64), np.concatenate(val)

    def paint(self, image_array, rng, exptime=30.0, num_crs=None):
        idx, val = self.draw_hits(image_array.shape, rng, exptime, num_crs)
        np.add.at(image_array.reshape(-1), idx, val)
        return image_array

    def paint_device(self, image_dev, rng, exptime=30.0, num_crs=None):
        """the same on a [ny][nx] float64 device tensor: one scatter-add"""
        import torch
        idx, val = self.draw_hits(tuple(image_dev.shape), rng, exptime, num_crs)
        if len(idx):
            image_dev.view(-1).index_add_(0, torch.from_numpy(idx).to(image_dev.device), torch.from_numpy(val).to(image_dev.device))
        return len(idx)


def write_cosmic_ray_catalog(fp_id, x0, y0, pixel_values, exptime, num_pix, outfile="cosmic_ray_catalog.fits", overwrite=True):
    """FITS binary table of footprint spans (imsim/cosmic_rays.py:147-185)"""
    if os.path.exists(outfile) and not overwrite:
        raise OSError(f"{outfile} exists")
    cols = [("fp_id", "J", np.asarray(fp_id)), ("x0", "I", np.asarray(x0)), ("y0", "I", np.asarray(y0)),
            ("pixel_values", "PJ()", list(pixel_values))]
    with open(outfile, "wb") as f:
        f.write(fits_io.hdu_bytes({}, None, primary=True))
        f.write(fits_io.bintable_hdu_bytes(cols, header=[("EXPTIME", exptime), ("NUM_PIX", num_pix)], extname="COSMIC_RAYS"))
