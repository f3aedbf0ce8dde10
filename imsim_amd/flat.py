"""`image.type: LSST_Flat` (imsim/flat.py:21-283).  Area branch (no sed): the flat is built in `niter` iterations of

    area = sensor.calculate_pixel_areas(image so far)        (tree rings + brighter-fatter)
    temp = base * area / mean(area);  Poisson(temp);  image += temp

on the GPU: `ims_sensor_pixel_areas` (polygon areas of the live boundary state), `ims_flat_add`
(Poisson realisation + add + delta charge) and `ims_sensor_update_distortions` between iterations.
With an sed (flat.py:237-262) every iteration instead shoots Poisson(counts_per_iter x area) photons uniformly
over the working image, samples their wavelengths and accumulates them through the sensor (conversion depth,
diffusion, current pixel boundaries), `build_image_photons`.
The reference builds the CCD in nx x ny sections with a buffer only to bound host memory
(flat.py:183-197); here the whole CCD is one section.  Parameter surface as in the reference:
counts_per_pixel (required), xsize, ysize, max_counts_per_iter, buffer_size, nx, ny (flat.py:45-58).
"""
import math

import numpy as np

from . import _abi

def crop(scene, image):
    """the part of the working image that is the CCD (drops configs.scene_flat's border)"""
    b = int(getattr(scene, "flat_buffer", 0))
    return image[b:image.shape[0] - b, b:image.shape[1] - b] if b else image


FLAT_REQ = {"counts_per_pixel": float}
FLAT_OPT = {"max_counts_per_iter": float, "buffer_size": int, "nx": int, "ny": int, "size": int, "xsize": int,
            "ysize": int, "det_name": str}


class LSST_FlatBuilder:
    def setup(self, config):
        """imsim/flat.py:31-118.  Returns (xsize, ysize)."""
        for k in FLAT_REQ:
            if k not in config:
                raise ValueError(f"Attribute {k} is required")
        self.counts_per_pixel = float(config["counts_per_pixel"])
        self.max_counts_per_iter = float(config.get("max_counts_per_iter", 1000.0))
        self.buffer_size = int(config.get("buffer_size", 5))
        self.nx, self.ny = int(config.get("nx", 8)), int(config.get("ny", 2))
        size = int(config.get("size", 0))
        self.xsize, self.ysize = int(config.get("xsize", size)), int(config.get("ysize", size))
        if self.xsize == 0 or self.ysize == 0:
            raise ValueError("LSST_Flat needs xsize/ysize (or a det_name resolved by the caller)")
        return self.xsize, self.ysize

    def iterations(self):
        """niter and counts per iteration (flat.py:159-161)"""
        niter = int(math.ceil(self.counts_per_pixel / self.max_counts_per_iter))
        return niter, self.counts_per_pixel / niter

    def build_image(self, renderer, seed=0, base=None):
        """addNoise of the reference (flat.py:133-268, `sed is None` branch) on `renderer`, whose scene is the
        flat's CCD (slot 0 = the whole image when it has a Silicon sensor).  base: optional [ny][nx] array of
        relative WCS pixel areas (makeSkyImage with sky_level=1, flat.py:171-176); None = uniform."""
        torch = renderer.torch
        sc = renderer.scene
        lib = renderer.lib
        niter, counts_per_iter = self.iterations()
        silicon = sc.sensor is not None
        n = sc.nx * sc.ny
        base_t = None
        level = counts_per_iter
        if base is not None:
            b = np.ascontiguousarray(base, dtype=np.float64)
            level = counts_per_iter / float(b.mean())             # mean sky level = counts_per_iter (flat.py:174-176)
            base_t = torch.from_numpy(b).to(renderer.device)
        st = renderer._stream()
        if silicon:
            renderer._need_static("LSST_Flat")
            area = torch.empty(n, dtype=torch.float64, device=renderer.device)
            acc = torch.zeros(1, dtype=torch.int64, device=renderer.device)
            delta_ptr = renderer.bound.sensor_struct.bf_delta       # slot 0 starts at offset 0
            if int(renderer.bound._slots_host[0]["offset"]) != 0:
                raise ValueError("LSST_Flat expects slot 0 at offset 0")
        for it in range(niter):
            if silicon:
                acc.zero_()
                _abi.check(lib.ims_sensor_pixel_areas(renderer.bound.sensor_dev_ptr, _abi.C.byref(renderer.bound.sensor_host), 0,
                                                      area.data_ptr(), acc.data_ptr(), st), "ims_sensor_pixel_areas")
                mean_area = float(int(acc.item())) / float(n) * 2.0 ** -32
                _abi.check(lib.ims_flat_add(area.data_ptr(), base_t.data_ptr() if base_t is not None else None, level,
                                            1.0 / mean_area, seed, it, sc.nx, sc.ny, renderer.image.data_ptr(), delta_ptr, st),
                           "ims_flat_add")
                if it + 1 < niter:
                    renderer.update_distortions(0, 1)
            else:
                _abi.check(lib.ims_flat_add(None, base_t.data_ptr() if base_t is not None else None, level, 1.0, seed, it,
                                            sc.nx, sc.ny, renderer.image.data_ptr(), None, st), "ims_flat_add")
        return crop(sc, renderer.image)


    def build_image_photons(self, renderer, seed=0, sed_table=0):
        """The sed branch (flat.py:237-262): per iteration nphotons ~ Poisson(counts_per_iter * area) photons at
        uniform positions over the working image, wavelengths from scene.sed_tables[sed_table], through
        sensor.accumulate(photons, section, resume=(it > 0)); the boundaries are updated between iterations from
        the charge of the last one.  One Box-profile object per iteration carries the photons."""
        from . import catalog
        from ._abi import OBJECT_DTYPE, IMS_PROF_BOX
        sc = renderer.scene
        niter, counts_per_iter = self.iterations()
        silicon = sc.sensor is not None
        if silicon and not sc.track_static_delta:
            raise ValueError("photon flats need scene.track_static_delta = 1 (the whole image is one brighter-fatter region)")
        rng = np.random.default_rng([int(seed), 0xF1A7])
        area = sc.nx * sc.ny
        for it in range(niter):
            n = int(rng.poisson(counts_per_iter * area))
            obj = np.zeros(1, dtype=OBJECT_DTYPE)
            obj["obj_id"] = catalog.FLAT_OBJECT_ID + it
            obj["n_phot"] = n
            obj["x0"], obj["y0"] = sc.xmin + (sc.nx - 1) / 2.0, sc.ymin + (sc.ny - 1) / 2.0      # centre of the working image
            obj["flux_per_photon"] = 1.0
            obj["prof_table"], obj["prof_scale"], obj["prof_aux"] = IMS_PROF_BOX, sc.nx * 0.2, sc.ny * 0.2
            obj["jac"] = (1.0, 0.0, 0.0, 1.0)
            obj["winv"] = (5.0, 0.0, 0.0, 5.0)
            obj["dcr_cosp"] = 1.0
            obj["sed_table"] = sed_table
            obj["stamp_xmin"], obj["stamp_xmax"] = sc.xmin, sc.xmin + sc.nx - 1
            obj["stamp_ymin"], obj["stamp_ymax"] = sc.ymin, sc.ymin + sc.ny - 1
            if it > 0 and silicon:
                renderer.update_distortions(0, 1)
            renderer.render(obj)
        return crop(sc, renderer.image)
