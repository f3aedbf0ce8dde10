"""LSST_SiliconBuilder.setup for a whole catalog ON THE DEVICE (include/imsim_hip.h "object table on the device").

`catalog.build_object_table` (numpy) writes 100 000 rows of 256 bytes in ~120 ms on the GPU box's host -- five times the
25 ms render of C3.  Here the catalog's columns go up once (a few MB), one launch of `ims_build_object_table` writes the
rows where the render reads them, and only 16 bytes per object come back (photon count, stamp size, two flag bits): what the
launch plan needs.  The host keeps the two things that are not per-object arithmetic:

* the surface-brightness loop of `get_good_phot_stamp_size` for the few bright or oversized galaxies the kernel flags
  (`catalog.gal_stamp_size` on that subset, then `ims_patch_stamp_sizes`);
* rows of the rare kinds (knots, streaks, FITS stamps), built by the numpy builder and copied over.

Reference: imsim/stamp.py:109-249, imsim/stamp_utils.py:79-189, imsim/instcat.py:498-527.
"""
import ctypes as C
import functools
import math

import numpy as np

from . import _abi, catalog, tables
from ._abi import OBJECT_DTYPE, META_DTYPE

N_STAR_SIZE = 48


def star_size_table(noise_var_unused=None, airmass=1.2, raw_seeing=0.7, band="r", nmax=catalog.NMAX):
    """star_size[k]: get_star_stamp_size for a folding threshold of exp(-k) (k >= 6), entries 0 .. 5 the default threshold
    (imsim/stamp_utils.py:126-155: the threshold noise_var / flux is rounded DOWN to e-folds, so the size is a function of
    the integer -floor(ln(noise_var / flux)) alone)."""
    k = np.arange(N_STAR_SIZE, dtype=np.float64)
    ft = np.where(k < 6, catalog.FT_DEFAULT, np.exp(-k))
    fwhm_atm, fwhm_sys = catalog.kolmogorov_gaussian_fwhm(airmass, raw_seeing, band)
    sk = catalog.kolmogorov_stepk(fwhm_atm, ft)
    sg = catalog.gaussian_stepk(fwhm_sys / 2.3548200450309493, ft)
    stepk = 1.0 / np.sqrt(1.0 / sk ** 2 + 1.0 / sg ** 2)
    return np.minimum(catalog._good_size(stepk), nmax).astype(np.int32)


def gal_radius_table(sersic_index=None):
    """gal_radius[t]: radius enclosing 1 - folding_threshold of a Sersic profile's flux [half-light radii], at least
    stepk_minimum_hlr, for radial table t (0: n = 1, 1: n = 4, further tables: scene.sersic_index)."""
    by_table = {0: 1.0, 1: 4.0}
    for n, t in (sersic_index or {}).items():
        by_table[int(t)] = float(n)
    out = np.full(max(by_table) + 1, catalog.STEPK_MIN_HLR)
    for t, n in by_table.items():
        out[t] = max(float(catalog._radius_enclosing(tables.sersic_table(n), 1.0 - catalog.FT_DEFAULT)), catalog.STEPK_MIN_HLR)
    return out


def sersic_sb_tables(sersic_index=None):
    """per radial table t: b_n, I(0) hlr^2 / flux and 1 / n of its Sersic index (what sersic_xvalue needs of it)"""
    by_table = {0: 1.0, 1: 4.0}
    for n, t in (sersic_index or {}).items():
        by_table[int(t)] = float(n)
    m = max(by_table) + 1
    b, norm, inv_n = np.ones(m), np.zeros(m), np.ones(m)
    for t, n in by_table.items():
        b[t], norm[t] = catalog._sersic_norm(n)
        inv_n[t] = 1.0 / n
    return b, norm, inv_n


@functools.lru_cache(maxsize=16)
def psf_phot_sizes(noise_var, nmax=catalog.NMAX):
    """get_good_phot_stamp_size1 of the DoubleGaussian proxy PSF at the surface-brightness limits sqrt(noise_var) / 8 and three
    times that (stamp_utils.py:196-220): object-independent, so the host hands the two integers to the table kernel"""
    keep = math.sqrt(noise_var) / 8.0
    dg_stepk = min(catalog.gaussian_stepk(0.6 / 2.355, catalog.FT_DEFAULT), catalog.gaussian_stepk(0.12 / 2.355, catalog.FT_DEFAULT))
    n0 = np.array([catalog._good_size(dg_stepk)])
    f = lambda h: catalog._edge_max(catalog.double_gaussian_xvalue, h)     # noqa: E731
    return (keep, int(catalog._phot_stamp_size1(n0, f, keep, nmax)[0]), int(catalog._phot_stamp_size1(n0, f, 3.0 * keep, nmax)[0]))


def zenith_vector(latitude, hour_angle_center, ra_center):
    """unit vector of the zenith in the frame of the image WCS: declination = latitude, right ascension = the local sidereal
    time ra_center + hour_angle_center [radians]"""
    lst = ra_center + hour_angle_center
    return (math.cos(latitude) * math.cos(lst), math.cos(latitude) * math.sin(lst), math.sin(latitude))


COLUMNS = (("x", np.float64), ("y", np.float64), ("nominal_flux", np.float64), ("hlr", np.float64), ("q", np.float64),
           ("pa", np.float64), ("g1", np.float64), ("g2", np.float64), ("mu", np.float64), ("kind", np.int32),
           ("prof_table", np.int32), ("sed_table", np.int32), ("stamp_size", np.int32), ("obj_id", np.int64),
           ("phot_flux", np.int64), ("sb_flux", np.float64))


def catalog_columns(cat, phot_flux=None, sersic_index=None, stamp_size=None):
    """The columns ims_catalog_t wants, from a catalog dict (catalog.synthetic_catalog, instcat.InstCatalog.columns)."""
    n = len(cat["x"])
    kind = np.asarray(cat["kind"], dtype=np.int32)
    n_obj = catalog._sersic_n_of(kind, cat.get("sersic_n"))
    table = np.where(kind == 2, 1, 0).astype(np.int32)
    odd = (n_obj > 0) & (n_obj != 1.0) & (n_obj != 4.0)
    if odd.any():
        index = sersic_index or {}
        missing = sorted(set(np.round(n_obj[odd], 2).tolist()) - set(index))
        if missing:
            raise ValueError(f"no radial table for Sersic index {missing}: call configs.add_sersic_tables(scene, cat['sersic_n'])")
        table[odd] = [index[round(float(v), 2)] for v in n_obj[odd]]
    cols = dict(x=cat["x"], y=cat["y"], nominal_flux=cat["nominal_flux"], hlr=cat["hlr"], q=cat["q"], pa=cat["pa"], kind=kind,
                prof_table=table, obj_id=cat.get("obj_id"))
    if "g1" in cat:
        cols.update(g1=cat["g1"], g2=cat["g2"], mu=cat["mu"])
    if cat.get("sb_flux") is not None:
        cols["sb_flux"] = cat["sb_flux"]
    if cat.get("sed_table") is not None:
        cols["sed_table"] = np.broadcast_to(np.asarray(cat["sed_table"], dtype=np.int32), (n,))
    if phot_flux is not None:
        cols["phot_flux"] = phot_flux
    if stamp_size is not None:
        cols["stamp_size"] = np.broadcast_to(np.asarray(stamp_size, dtype=np.int32), (n,))
    return cols


def fill_catalog_struct(cols, ptr_of, seed, visit, optics_has_field=True, noise_var=800.0, max_flux_simple=100.0, sed_table=0,
                        sersic_index=None, sizes_on_device=True):
    """ims_catalog_t over columns that already live where the kernel (or the oracle) reads them.  ptr_of(name, array, dtype)
    -> address.  visit: dict with airmass, raw_seeing, band, latitude, hour_angle, ra [degrees]."""
    st = _abi.Catalog()
    st.n = len(cols["x"])
    for name, dt in COLUMNS:
        a = cols.get(name)
        setattr(st, name, ptr_of(name, a, dt) if a is not None else None)
    st.seed = int(seed)
    st.sed_table_all = int(sed_table)
    star = star_size_table(airmass=visit["airmass"], raw_seeing=visit["raw_seeing"], band=visit["band"])
    gal = gal_radius_table(sersic_index)
    st.n_star_size, st.n_gal_radius, st.nmax = len(star), len(gal), catalog.NMAX
    st.star_size = ptr_of("star_size", star, np.int32)
    st.gal_radius = ptr_of("gal_radius", gal, np.float64)
    st.noise_var, st.max_flux_simple, st.tiny_flux, st.pixel_scale = noise_var, max_flux_simple, catalog.TINY_FLUX, catalog.PIXEL_SCALE
    st.dg_stepk = float(min(catalog.gaussian_stepk(0.6 / 2.355, catalog.FT_DEFAULT), catalog.gaussian_stepk(0.12 / 2.355, catalog.FT_DEFAULT)))
    z = zenith_vector(math.radians(visit["latitude"]), math.radians(visit["hour_angle"]), math.radians(visit["ra"]))
    st.zenith[0], st.zenith[1], st.zenith[2] = z
    st.has_field = 1 if optics_has_field else 0
    if sizes_on_device:
        # the surface-brightness loop of the bright / oversized galaxies in the kernel (ims_catalog_t.sb_tables)
        b, norm, inv_n = sersic_sb_tables(sersic_index)
        if len(b) != len(gal):
            raise ValueError("sersic tables and gal_radius disagree in length")
        st.sersic_b, st.sersic_norm, st.sersic_inv_n = ptr_of("sersic_b", b, np.float64), ptr_of("sersic_norm", norm, np.float64), ptr_of("sersic_inv_n", inv_n, np.float64)
        st.keep_sb, st.psf_size_keep, st.psf_size_keep3 = psf_phot_sizes(float(noise_var))
        st.sb_tables = 1
    return st


def host_fixups(cat, meta, build_kw):
    """What the kernel leaves to the host, from the catalog and the 16 bytes per object that came back: (index, size) of
    the galaxies whose stamp follows from the surface-brightness loop, and (index, rows) of the kinds it does not build."""
    flags = meta["flags"]
    pend = np.flatnonzero(flags & _abi.IMS_META_SIZE_PENDING)
    sizes = np.zeros(0, dtype=np.int32)
    if len(pend):
        kind = np.asarray(cat["kind"])[pend]
        sub = {k: np.asarray(v)[pend] for k, v in cat.items() if isinstance(v, np.ndarray) and len(v) == len(cat["x"])}
        jac = catalog.shear_matrix(sub["q"], 90.0 - sub["pa"])
        if "g1" in sub:
            jac = catalog._mat2(catalog.lens_matrix(sub["g1"], sub["g2"], sub["mu"]), jac)
        a, b, c, d = jac[:, 0], jac[:, 1], jac[:, 2], jac[:, 3]
        s1 = a * a + b * b + c * c + d * d
        s2 = np.sqrt(np.maximum((a * a + b * b - c * c - d * d) ** 2 + 4 * (a * c + b * d) ** 2, 0.0))
        n_obj = catalog._sersic_n_of(kind, sub.get("sersic_n"))
        sizes = catalog.gal_stamp_size(kind, sub["hlr"], np.sqrt(0.5 * (s1 + s2)), jac=jac, nominal_flux=sub["nominal_flux"],
                                       noise_var=build_kw.get("noise_var", 800.0), sb_flux=sub.get("sb_flux"),
                                       sersic_n=n_obj).astype(np.int32)
    host = np.flatnonzero(flags & _abi.IMS_META_HOST_ROW)
    return pend.astype(np.int64), sizes, host.astype(np.int64)


_PINNED = {}
_PINNED_LOCK = __import__("threading").Lock()       # tables of several CCDs may be built from several host threads (focal plane)


def _pinned(torch, nbytes, slot=0):
    """a page-locked staging buffer of at least nbytes, kept for the next table (allocating one costs milliseconds per 10 MB);
    the caller synchronises before the buffer is handed out again (DeviceTable waits for its meta download)"""
    import threading
    key = (slot, threading.get_ident())              # one staging buffer per thread: a table being staged is never overwritten
    with _PINNED_LOCK:
        buf = _PINNED.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = _PINNED[key] = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, pin_memory=True)
    return buf[:nbytes]


class DeviceTable:
    """The object table of one catalog, resident on the device: `rows` (torch uint8 tensor of n x 256 bytes) and the host's
    copy of what planning needs (`n_phot`, stamp bounds, flags as a slim structured array)."""

    def __init__(self, renderer, cat, visit, phot_flux=None, noise_var=800.0, max_flux_simple=100.0, sed_table=0,
                 stamp_size=None, sizes_on_device=True):
        """sizes_on_device: the surface-brightness loop of the bright / oversized galaxies runs in the table kernel (the default);
        False: the kernel flags those rows and the numpy loop of catalog.gal_stamp_size patches them (the checker)"""
        torch = renderer.torch
        dev = renderer.device
        self.renderer = renderer
        n = len(cat["x"])
        self.n = n
        sersic_index = getattr(renderer.scene, "sersic_index", None)
        cols = catalog_columns(cat, phot_flux, sersic_index, stamp_size)
        # ONE page-locked staging buffer for all columns, one copy up
        parts, off = [], 0
        for name, dt in COLUMNS:
            a = cols.get(name)
            if a is None:
                continue
            a = np.ascontiguousarray(a, dtype=dt)
            parts.append((name, off, a))
            off = (off + a.nbytes + 255) & ~255
        star = star_size_table(airmass=visit["airmass"], raw_seeing=visit["raw_seeing"], band=visit["band"])
        gal = gal_radius_table(sersic_index)
        sb_b, sb_norm, sb_inv_n = sersic_sb_tables(sersic_index)
        for name, a in (("star_size", star), ("gal_radius", gal), ("sersic_b", sb_b), ("sersic_norm", sb_norm), ("sersic_inv_n", sb_inv_n)):
            parts.append((name, off, a))
            off = (off + a.nbytes + 255) & ~255
        stage = _pinned(torch, max(off, 8))
        snp = stage.numpy()
        for name, o, a in parts:
            snp[o:o + a.nbytes] = a.view(np.uint8).reshape(-1)
        self._cols = stage.to(dev, non_blocking=True)
        base = self._cols.data_ptr()
        where = {name: base + o for name, o, a in parts}
        st = fill_catalog_struct(cols, lambda name, a, dt: where[name], renderer.scene.seed, visit,
                                 optics_has_field=True, noise_var=noise_var, max_flux_simple=max_flux_simple,
                                 sed_table=sed_table, sersic_index=sersic_index, sizes_on_device=sizes_on_device)
        self.rows = torch.empty(max(n, 1) * OBJECT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        self._meta_dev = torch.empty(max(n, 1) * META_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        optics_ptr = renderer.bound.base_params.optics
        if not optics_ptr:
            raise ValueError("DeviceTable needs a scene with optics (the image WCS lives in ims_optics_t)")
        stream = renderer._stream()
        _abi.check(renderer.lib.ims_build_object_table(C.byref(st), optics_ptr, self.rows.data_ptr(), self._meta_dev.data_ptr(),
                                                       stream), "ims_build_object_table")
        meta_pin = _pinned(torch, max(n, 1) * META_DTYPE.itemsize, slot=1)
        meta_pin.copy_(self._meta_dev, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        meta = meta_pin.numpy()[:n * META_DTYPE.itemsize].view(META_DTYPE).copy()
        pend, sizes, host = host_fixups(cat, meta, dict(noise_var=noise_var))
        if len(pend):
            idx_t = torch.from_numpy(pend).to(dev)
            sz_t = torch.from_numpy(sizes).to(dev)
            _abi.check(renderer.lib.ims_patch_stamp_sizes(self.rows.data_ptr(), self._meta_dev.data_ptr(), idx_t.data_ptr(),
                                                          sz_t.data_ptr(), len(pend), stream), "ims_patch_stamp_sizes")
            meta["size"][pend] = sizes
            meta["flags"][pend] &= ~_abi.IMS_META_SIZE_PENDING
            self._keep = (idx_t, sz_t)
        self.host_rows = host
        if len(host):
            # knots, streaks, FITS stamps: rows from the numpy builder (with the per-object local WCS and DCR angles of
            # configs.c3_objects), copied over the zeroed rows the kernel left
            from . import configs
            sub = {k: (np.asarray(v)[host] if isinstance(v, np.ndarray) and len(v) == n else v) for k, v in cat.items()}
            # the photon counts of these rows were realised by the kernel with the counter-based Poisson of every other object
            # (ims_object_meta_t.n_phot of a HOST_ROW entry): the same flux whichever rows are host rows, on every rank
            ph = (np.asarray(phot_flux)[host] if phot_flux is not None else meta["n_phot"][host]).astype(np.int64)
            keep = ph > 0
            rows_h = np.zeros(len(host), dtype=OBJECT_DTYPE)
            size_h = np.zeros(len(host), dtype=np.int32)
            if keep.any():
                built, sz = catalog.build_object_table(sub, ph, noise_var=noise_var, max_flux_simple=max_flux_simple,
                                                       sed_table=sed_table, airmass=visit["airmass"],
                                                       raw_seeing=visit["raw_seeing"], band=visit["band"],
                                                       sersic_index=sersic_index)
                winv, p0 = configs.local_wcs_inverse(renderer.scene.optics.img_wcs, sub["x"][keep], sub["y"][keep])
                tz, sp, cp = configs.dcr_angles(p0, math.radians(visit["latitude"]), math.radians(visit["hour_angle"]),
                                                math.radians(visit["ra"]))
                built["winv"], built["dcr_tanz"], built["dcr_sinp"], built["dcr_cosp"] = winv, tz, sp, cp
                rows_h[keep], size_h[keep] = built, sz
            rows2d = self.rows[:n * OBJECT_DTYPE.itemsize].view(n, OBJECT_DTYPE.itemsize)
            rows2d[torch.from_numpy(host).to(dev)] = torch.from_numpy(rows_h.view(np.uint8).reshape(len(host), -1)).to(dev)
            meta["n_phot"][host] = rows_h["n_phot"]
            meta["size"][host] = size_h
        # the slim host view the planner works on
        icx = np.floor(np.asarray(cat["x"]) + 0.5).astype(np.int64)
        icy = np.floor(np.asarray(cat["y"]) + 0.5).astype(np.int64)
        size = meta["size"].astype(np.int64)
        self.meta = meta
        self.n_phot = meta["n_phot"].copy()
        self.stamp = np.stack([icx - size // 2, icx - size // 2 + size - 1, icy - size // 2, icy - size // 2 + size - 1], axis=1)
        self.faint = np.asarray(cat["nominal_flux"]) < max_flux_simple
        self.x, self.y = np.asarray(cat["x"], dtype=np.float64), np.asarray(cat["y"], dtype=np.float64)

    def rows_numpy(self):
        """the table back on the host (tests)"""
        return self.rows.cpu().numpy()[:self.n * OBJECT_DTYPE.itemsize].view(OBJECT_DTYPE).copy()
