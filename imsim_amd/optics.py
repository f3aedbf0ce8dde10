"""Telescope description for the sequential ray-trace kernel (host side).

imSim traces photons with batoid (`telescope.trace(ray_vec)`, imsim/photon_ops.py:109-123) through
a telescope loaded from batoid's YAML files (imsim/telescope_loader.py:121-252).  Here the
telescope is a flat list of coaxial surfaces (`Surface`) that the HIP kernel walks; it can be read
from a batoid-format YAML (`load_batoid_yaml`, the subset of the format the Rubin files use:
coaxial CompoundOptic / Lens / Mirror / RefractiveInterface / Baffle / Detector, Plane / Sphere /
Paraboloid / Quadric / Asphere, annular/circular obscurations, constant / Sellmeier / Air media).

batoid's own `LSST_r.yaml` is not part of the reference tree (it ships with batoid), so
`rubin_like_telescope()` provides a clearly-labelled APPROXIMATE Rubin prescription (public LSST
optical design values, refocused numerically) so the benchmark configs have a realistic surface
list to trace.  Users with batoid's files point `load_batoid_yaml` at them.
"""
import dataclasses
import math
from typing import List, Optional

import numpy as np

from . import _abi, wcs as wcsmod

SILICA = (_abi.IMS_MEDIUM_SELLMEIER,
          (0.6961663, 0.4079426, 0.8974794, 0.00467914825849, 0.013512063073959999, 97.93400253792099))
VACUUM = (_abi.IMS_MEDIUM_CONST, (1.0, 0, 0, 0, 0, 0))
AIR = (_abi.IMS_MEDIUM_AIR, (69.328, 293.15, 1.067, 0, 0, 0))   # batoid.Air defaults [kPa, K, kPa]


@dataclasses.dataclass
class Surface:
    kind: int
    z0: float
    R: float = 0.0
    conic: float = 0.0
    asph: tuple = ()
    obsc_kind: int = _abi.IMS_OBSC_NONE
    obsc_inner: float = 0.0
    obsc_outer: float = 0.0
    medium: tuple = VACUUM          # medium after the surface (refractive only)
    name: str = ""


@dataclasses.dataclass
class Telescope:
    surfaces: List[Surface]
    stop_z: float = 0.4393899
    in_medium: tuple = VACUUM
    pupil_outer: float = 4.18
    pupil_inner: float = 2.558
    name: str = "telescope"

    def with_detector_z(self, z):
        surf = [dataclasses.replace(s) for s in self.surfaces]
        surf[-1] = dataclasses.replace(surf[-1], z0=z)
        return dataclasses.replace(self, surfaces=surf)


def medium_n(medium, wave_nm):
    kind, c = medium
    wave_nm = np.asarray(wave_nm, dtype=np.float64)
    if kind == _abi.IMS_MEDIUM_CONST:
        return np.full_like(wave_nm, c[0])
    if kind == _abi.IMS_MEDIUM_SELLMEIER:
        l2 = (wave_nm * 1e-3) ** 2
        return np.sqrt(1.0 + c[0] * l2 / (l2 - c[3]) + c[1] * l2 / (l2 - c[4]) + c[2] * l2 / (l2 - c[5]))
    P = c[0] * 7.50061683
    T = c[1] - 273.15
    W = c[2] * 7.50061683
    s2 = 1.0 / (wave_nm * 1e-3) ** 2
    n1 = (64.328 + 29498.1 / (146.0 - s2) + 255.4 / (41.0 - s2)) * 1e-6
    n1 = n1 * (P * (1.0 + (1.049 - 0.0157 * T) * 1e-6 * P) / (720.883 * (1.0 + 0.003661 * T)))
    n1 = n1 - (0.0624 - 0.000680 * s2) / (1.0 + 0.003661 * T) * W * 1e-6
    return 1.0 + n1


def _sag(S, r2):
    z = np.zeros_like(r2)
    dz = np.zeros_like(r2)
    ok = np.ones(r2.shape, dtype=bool)
    if S.R != 0.0:
        c = 1.0 / S.R
        arg = 1.0 - (1.0 + S.conic) * c * c * r2
        ok = arg >= 0
        sq = np.sqrt(np.where(ok, arg, 1.0))
        z = c * r2 / (1.0 + sq)
        dz = c / (2.0 * sq)
    rp = r2.copy()
    for k, a in enumerate(S.asph):
        dz = dz + a * (k + 2) * rp
        rp = rp * r2
        z = z + a * rp
    return z, dz, ok


def trace_numpy(tel: Telescope, pos, vel, wave_nm):
    """Vectorised host tracer with the same algorithm as the kernel (used to fit the WCS and to
    focus the approximate prescription; NOT part of the photon path).  pos, vel: (n,3)."""
    pos = np.array(pos, dtype=np.float64)
    vel = np.array(vel, dtype=np.float64)
    wave_nm = np.broadcast_to(np.asarray(wave_nm, dtype=np.float64), pos.shape[:1])
    vig = np.zeros(len(pos), dtype=bool)
    fail = np.zeros(len(pos), dtype=bool)
    n_cur = medium_n(tel.in_medium, wave_nm)
    for S in tel.surfaces:
        pz = pos[:, 2] - S.z0
        if S.R != 0.0:
            k1 = 1.0 + S.conic
            A = vel[:, 0] ** 2 + vel[:, 1] ** 2 + k1 * vel[:, 2] ** 2
            B = 2.0 * (pos[:, 0] * vel[:, 0] + pos[:, 1] * vel[:, 1] + k1 * pz * vel[:, 2] - S.R * vel[:, 2])
            Cq = pos[:, 0] ** 2 + pos[:, 1] ** 2 + k1 * pz * pz - 2.0 * S.R * pz
            disc = B * B - 4.0 * A * Cq
            fail |= disc < 0
            sq = np.sqrt(np.clip(disc, 0.0, None))
            q = -0.5 * (B + np.where(B < 0, -sq, sq))
            with np.errstate(divide="ignore", invalid="ignore"):
                t1, t2 = q / A, Cq / q
            z1, z2 = pz + vel[:, 2] * t1, pz + vel[:, 2] * t2
            t = np.where((np.abs(z2) <= np.abs(z1)) | ~(np.abs(z1) < 1e300), t2, t1)
        else:
            t = -pz / vel[:, 2]
        for _ in range(5 if S.asph else 0):
            x = pos[:, 0] + vel[:, 0] * t
            y = pos[:, 1] + vel[:, 1] * t
            z = pz + vel[:, 2] * t
            sag, ds, ok = _sag(S, x * x + y * y)
            fail |= ~ok
            f = z - sag
            fp = vel[:, 2] - 2.0 * ds * (x * vel[:, 0] + y * vel[:, 1])
            t = np.where(np.abs(f) <= 1e-14, t, t - f / fp)
        x = pos[:, 0] + vel[:, 0] * t
        y = pos[:, 1] + vel[:, 1] * t
        sag, ds, ok = _sag(S, x * x + y * y)
        fail |= ~ok
        pos = np.stack([x, y, S.z0 + sag], axis=1)
        r = np.hypot(x, y)
        if S.obsc_kind == _abi.IMS_OBSC_CLEAR_ANNULUS:
            vig |= ~((r >= S.obsc_inner) & (r <= S.obsc_outer))
        elif S.obsc_kind == _abi.IMS_OBSC_CLEAR_CIRCLE:
            vig |= ~(r <= S.obsc_outer)
        elif S.obsc_kind == _abi.IMS_OBSC_OBSC_CIRCLE:
            vig |= r < S.obsc_outer
        elif S.obsc_kind == _abi.IMS_OBSC_OBSC_ANNULUS:
            vig |= (r >= S.obsc_inner) & (r < S.obsc_outer)
        if S.kind in (_abi.IMS_SURF_BAFFLE, _abi.IMS_SURF_DETECTOR):
            continue
        nrm = np.stack([-2.0 * ds * x, -2.0 * ds * y, np.ones_like(x)], axis=1)
        nrm /= np.linalg.norm(nrm, axis=1)[:, None]
        if S.kind == _abi.IMS_SURF_MIRROR:
            d = np.sum(vel * nrm, axis=1)
            vel = vel - 2.0 * d[:, None] * nrm
        else:
            n2 = medium_n(S.medium, wave_nm)
            dvec = vel * n_cur[:, None]
            alpha = np.sum(dvec * nrm, axis=1)
            flip = alpha > 0
            nrm[flip] = -nrm[flip]
            alpha = np.where(flip, -alpha, alpha)
            eta = n_cur / n2
            sinsqr = eta * eta * (1.0 - alpha * alpha)
            fail |= sinsqr > 1.0
            nfac = eta * alpha + np.sqrt(np.clip(1.0 - sinsqr, 0.0, None))
            vel = (eta[:, None] * dvec - nfac[:, None] * nrm) / n2[:, None]
            n_cur = n2
    return pos, vel, vig, fail


def rubin_like_telescope(band="r", refocus=True):
    """APPROXIMATE Rubin/LSST prescription (public optical-design values recalled, not batoid's
    LSST_r.yaml): M1/M2/M3, three fused-silica lenses, filter, detector.  The detector position is
    refocused numerically so the system forms a sharp image.  For benchmarking and tests only."""
    M, RF, DET = _abi.IMS_SURF_MIRROR, _abi.IMS_SURF_REFRACT, _abi.IMS_SURF_DETECTOR
    CA, CC = _abi.IMS_OBSC_CLEAR_ANNULUS, _abi.IMS_OBSC_CLEAR_CIRCLE
    z_m3 = -0.2338
    z_l1 = z_m3 + 3.6305
    filt_t = dict(u=0.0265, g=0.0215, r=0.0179, i=0.0158, z=0.0144, y=0.0130).get(band, 0.0179)
    z_l1b = z_l1 + 0.08223
    z_l2 = z_l1b + 0.41264
    z_l2b = z_l2 + 0.030
    z_f = z_l2b + 0.34958
    z_fb = z_f + filt_t
    z_l3 = z_fb + 0.0511
    z_l3b = z_l3 + 0.060
    z_det = z_l3b + 0.0285
    surf = [
        Surface(M, 0.0, 19.835, -1.215, (0.0, -1.381e-9), CA, 2.558, 4.18, name="M1"),
        Surface(M, 6.1562006, 6.788, -0.222, (0.0, 1.274e-5, 9.68e-7), CA, 0.9, 1.71, name="M2"),
        Surface(M, z_m3, 8.3445, 0.155, (0.0, 4.5e-7, 8.15e-9), CA, 0.55, 2.508, name="M3"),
        Surface(RF, z_l1, 2.824, 0.0, (), CC, 0.0, 0.775, SILICA, "L1_entrance"),
        Surface(RF, z_l1b, 5.021, 0.0, (), CC, 0.0, 0.775, VACUUM, "L1_exit"),
        Surface(RF, z_l2, 0.0, 0.0, (), CC, 0.0, 0.551, SILICA, "L2_entrance"),
        Surface(RF, z_l2b, 2.529, -1.57, (0.0, -1.656e-3), CC, 0.0, 0.551, VACUUM, "L2_exit"),
        Surface(RF, z_f, 5.632, 0.0, (), CC, 0.0, 0.375, SILICA, "Filter_entrance"),
        Surface(RF, z_fb, 5.606, 0.0, (), CC, 0.0, 0.375, VACUUM, "Filter_exit"),
        Surface(RF, z_l3, 3.169, -0.962, (), CC, 0.0, 0.361, SILICA, "L3_entrance"),
        Surface(RF, z_l3b, -13.36, 0.0, (), CC, 0.0, 0.361, VACUUM, "L3_exit"),
        Surface(DET, z_det, 0.0, 0.0, (), CC, 0.0, 0.4, name="Detector"),
    ]
    tel = Telescope(surf, stop_z=0.4393899, in_medium=VACUUM, name=f"rubin_like_{band}")
    if refocus:
        tel = refocus_detector(tel)
    return tel


def pupil_rays(tel, thx, thy, n_ring=6, n_az=24, wave_nm=620.0):
    """Rays filling the annular pupil for field angle (thx, thy) [rad]."""
    rr = np.linspace(tel.pupil_inner + 0.05, tel.pupil_outer - 0.05, n_ring)
    aa = np.linspace(0.0, 2 * np.pi, n_az, endpoint=False)
    r, a = np.meshgrid(rr, aa)
    x, y = (r * np.cos(a)).ravel(), (r * np.sin(a)).ravel()
    g = 1.0 / math.sqrt(1.0 + thx * thx + thy * thy)
    n = medium_n(tel.in_medium, np.array([wave_nm]))[0]
    vel = np.tile(np.array([thx * g, thy * g, -g]) / n, (len(x), 1))
    pos = np.stack([x, y, np.full_like(x, tel.stop_z)], axis=1)
    return pos, vel


def spot_rms(tel, fields=((0.0, 0.0), (0.012, 0.0), (0.0, 0.02), (0.018, 0.018)), wave_nm=620.0):
    tot = 0.0
    for thx, thy in fields:
        pos, vel = pupil_rays(tel, thx, thy, wave_nm=wave_nm)
        p, _, vig, fail = trace_numpy(tel, pos, vel, wave_nm)
        good = ~(vig | fail)
        if good.sum() < 10:
            return 1e9
        tot += np.var(p[good, 0]) + np.var(p[good, 1])
    return math.sqrt(tot / len(fields))


def refocus_detector(tel, span=0.02, n=81):
    """Move the detector plane to the best-focus z (minimum mean spot rms over a few field points)."""
    z0 = tel.surfaces[-1].z0
    zs = np.linspace(z0 - span, z0 + span, n)
    best = min(zs, key=lambda z: spot_rms(tel.with_detector_z(z)))
    zs = np.linspace(best - 2 * span / n, best + 2 * span / n, 41)
    best = min(zs, key=lambda z: spot_rms(tel.with_detector_z(z)))
    return tel.with_detector_z(float(best))


# ---------------- batoid-format YAML ----------------
def _medium_from_yaml(m, default):
    if m is None:
        return default
    if isinstance(m, (int, float)):
        return (_abi.IMS_MEDIUM_CONST, (float(m), 0, 0, 0, 0, 0))
    t = m.get("type", "ConstMedium")
    if t == "ConstMedium":
        return (_abi.IMS_MEDIUM_CONST, (float(m["n"]), 0, 0, 0, 0, 0))
    if t == "SellmeierMedium":
        return (_abi.IMS_MEDIUM_SELLMEIER, tuple(float(m[k]) for k in ("B1", "B2", "B3", "C1", "C2", "C3")))
    if t == "Air":
        return (_abi.IMS_MEDIUM_AIR, (float(m.get("pressure", 69.328)), float(m.get("temperature", 293.15)),
                                      float(m.get("h2o_pressure", 1.067)), 0, 0, 0))
    raise ValueError(f"unsupported medium {t}")


def _surface_from_yaml(s):
    t = s.get("type", "Plane")
    if t == "Plane":
        return 0.0, 0.0, ()
    if t == "Sphere":
        return float(s["R"]), 0.0, ()
    if t == "Paraboloid":
        return float(s["R"]), -1.0, ()
    if t == "Quadric":
        return float(s["R"]), float(s["conic"]), ()
    if t == "Asphere":
        return float(s["R"]), float(s["conic"]), tuple(float(c) for c in s.get("coefs", []))
    raise ValueError(f"unsupported surface {t}")


def _obsc_from_yaml(o):
    if o is None:
        return _abi.IMS_OBSC_NONE, 0.0, 0.0
    t = o["type"]
    if t == "ClearAnnulus":
        return _abi.IMS_OBSC_CLEAR_ANNULUS, float(o["inner"]), float(o["outer"])
    if t == "ClearCircle":
        return _abi.IMS_OBSC_CLEAR_CIRCLE, 0.0, float(o["radius"])
    if t == "ObscCircle":
        return _abi.IMS_OBSC_OBSC_CIRCLE, 0.0, float(o["radius"])
    if t == "ObscAnnulus":
        return _abi.IMS_OBSC_OBSC_ANNULUS, float(o["inner"]), float(o["outer"])
    if t == "ObscNegation":
        inner = o["original"]
        k, a, b = _obsc_from_yaml(inner)
        flip = {_abi.IMS_OBSC_OBSC_CIRCLE: _abi.IMS_OBSC_CLEAR_CIRCLE, _abi.IMS_OBSC_OBSC_ANNULUS: _abi.IMS_OBSC_CLEAR_ANNULUS,
                _abi.IMS_OBSC_CLEAR_CIRCLE: _abi.IMS_OBSC_OBSC_CIRCLE, _abi.IMS_OBSC_CLEAR_ANNULUS: _abi.IMS_OBSC_OBSC_ANNULUS}
        return flip[k], a, b
    raise ValueError(f"unsupported obscuration {t}")


def _coord_z(node, z_parent):
    cs = node.get("coordSys") or {}
    for k in ("x", "y", "rotX", "rotY", "rotZ"):
        if cs.get(k, 0.0) not in (0, 0.0):
            raise ValueError(f"coordSys.{k} != 0 is not supported (coaxial systems only); use the camera rotator angle instead")
    return z_parent + float(cs.get("z", 0.0))


def _walk_yaml(node, z_parent, in_medium, out):
    z = _coord_z(node, z_parent)
    t = node["type"]
    if t in ("CompoundOptic", "Lens"):
        medium = _medium_from_yaml(node.get("medium"), in_medium)
        items = node.get("items", [])
        if t == "Lens":
            first, last = items[0], items[-1]
            _walk_item(first, z, in_medium, medium, out)
            _walk_item(last, z, medium, in_medium, out)
        else:
            for it in items:
                _walk_yaml(it, z, _medium_from_yaml(node.get("inMedium"), in_medium), out)
        return
    _walk_item(node, z_parent, in_medium, _medium_from_yaml(node.get("outMedium", node.get("medium")), in_medium), out)


def _walk_item(node, z_parent, in_medium, out_medium, out):
    z = _coord_z(node, z_parent)
    R, conic, asph = _surface_from_yaml(node.get("surface", {}))
    ok, oi, oo = _obsc_from_yaml(node.get("obscuration"))
    kind = {"Mirror": _abi.IMS_SURF_MIRROR, "RefractiveInterface": _abi.IMS_SURF_REFRACT,
            "Detector": _abi.IMS_SURF_DETECTOR, "Baffle": _abi.IMS_SURF_BAFFLE,
            "Interface": _abi.IMS_SURF_BAFFLE}.get(node["type"])
    if kind is None:
        raise ValueError(f"unsupported optic type {node['type']}")
    out.append(Surface(kind, z, R, conic, asph, ok, oi, oo, out_medium, node.get("name", "")))


def load_batoid_yaml(path):
    """Read a batoid optic YAML (coaxial subset) into a Telescope."""
    import yaml
    with open(path) as f:
        cfg = yaml.safe_load(f)["opticalSystem"]
    in_medium = _medium_from_yaml(cfg.get("inMedium", cfg.get("medium")), VACUUM)
    out: List[Surface] = []
    _walk_yaml(cfg, 0.0, in_medium, out)
    stop = cfg.get("stopSurface", {})
    stop_z = float((stop.get("coordSys") or {}).get("z", 0.0))
    pupil = float(cfg.get("pupilSize", 8.36)) / 2.0
    return Telescope(out, stop_z=stop_z, in_medium=in_medium, pupil_outer=pupil,
                     pupil_inner=pupil * float(cfg.get("pupilObscuration", 0.612)), name=cfg.get("name", "telescope"))


# ---------------- ABI struct ----------------
def fill_optics(o: "_abi.Optics", tel: Telescope, fp_to_pix, rot_tel_pos=0.0):
    """Write the telescope, the camera rotator and the focal-plane->pixel affine into an Optics struct.

    fp_to_pix = (m0, m1, m2, m3, m4, m5): x_pix = m0*fpx + m1*fpy + m2, y_pix = m3*fpx + m4*fpy + m5
    (imsim/utils.py:42-59; e.g. R22_S11: (100, 0, 2047.5, 0, 100, 2001.5), tests/test_photon_ops.py:668-691)."""
    if len(tel.surfaces) > _abi.IMS_MAX_SURFACES:
        raise ValueError("too many surfaces")
    def medium_coeffs(medium):
        c = [float(v) for v in medium[1]]
        if medium[0] == _abi.IMS_MEDIUM_CONST:
            c[1] = 1.0 / c[0]            # constant media carry 1/n for the kernel (include/imsim_hip.h)
        return c

    o.in_medium_kind = tel.in_medium[0]
    for k, v in enumerate(medium_coeffs(tel.in_medium)):
        o.in_medium_c[k] = v
    o.n_surfaces = len(tel.surfaces)
    o.stop_z = tel.stop_z
    media = {}
    for k, S in enumerate(tel.surfaces):
        s = o.surf[k]
        s.medium_id = media.setdefault((S.medium[0], tuple(float(c) for c in S.medium[1])), len(media))
        s.kind, s.obsc_kind, s.medium_kind = S.kind, S.obsc_kind, S.medium[0]
        if len(S.asph) > 4:
            raise ValueError("at most 4 asphere coefficients")
        s.n_asphere = len(S.asph)
        s.z0, s.R, s.conic = S.z0, S.R, S.conic
        s.inv_R = (1.0 / S.R) if S.R != 0.0 else 0.0
        for m in range(4):
            s.asph[m] = float(S.asph[m]) if m < len(S.asph) else 0.0
        s.obsc_inner, s.obsc_outer = S.obsc_inner, S.obsc_outer
        for m, v in enumerate(medium_coeffs(S.medium)):
            s.medium_c[m] = v
    o.cam_rot[0], o.cam_rot[1] = math.cos(rot_tel_pos), math.sin(rot_tel_pos)
    for k in range(6):
        o.fp_to_pix[k] = float(fp_to_pix[k])
    a, b, c, d = fp_to_pix[0], fp_to_pix[1], fp_to_pix[3], fp_to_pix[4]
    s = math.sqrt(abs(a * d - b * c))
    # normalised M @ J with M = [[0, 1e3], [1e3, 0]] (imsim/photon_ops.py:497-500)
    o.slope_jac[0], o.slope_jac[1], o.slope_jac[2], o.slope_jac[3] = c / s, d / s, a / s, b / s
    return o


def field_to_pixel(tel, thx, thy, fp_to_pix, rot_tel_pos=0.0, wave_nm=620.0):
    """Pixel position of the pupil-averaged image of field angle (thx, thy) (the focal-plane
    position batoid_wcs.py:352-373 computes, followed by focal_to_pixel)."""
    pos, vel = pupil_rays(tel, thx, thy, wave_nm=wave_nm)
    p, _, vig, fail = trace_numpy(tel, pos, vel, wave_nm)
    good = ~(vig | fail)
    x, y = p[good, 0].mean(), p[good, 1].mean()
    c, s = math.cos(rot_tel_pos), math.sin(rot_tel_pos)
    rx, ry = c * x + s * y, -s * x + c * y
    fpx, fpy = ry * 1e3, rx * 1e3
    return (fp_to_pix[0] * fpx + fp_to_pix[1] * fpy + fp_to_pix[2],
            fp_to_pix[3] * fpx + fp_to_pix[4] * fpy + fp_to_pix[5])


def build_wcs_pair(tel, fp_to_pix, boresight_ra, boresight_dec, rot_sky=0.0, rot_tel_pos=0.0,
                   nx=4096, ny=4004, wave_nm=620.0, order=3):
    """Build (img_wcs, icrf_to_field) consistent with the telescope by ray tracing, as
    imsim/batoid_wcs.py does: icrf_to_field is the TAN projection about the boresight rotated by
    `rot_sky`; img_wcs is an order-3 TAN-SIP fitted through traced field points over the detector."""
    basis = wcsmod.tangent_basis(boresight_ra, boresight_dec, rot_sky)
    icrf_to_field = wcsmod.make_tansip((0.0, 0.0), np.eye(2), basis)
    # field angle of the detector centre by Newton iteration on the traced mapping
    target = np.array([(nx + 1) / 2.0, (ny + 1) / 2.0])
    th = np.zeros(2)
    h = 1e-4
    for _ in range(8):
        p0 = np.array(field_to_pixel(tel, th[0], th[1], fp_to_pix, rot_tel_pos, wave_nm))
        px = np.array(field_to_pixel(tel, th[0] + h, th[1], fp_to_pix, rot_tel_pos, wave_nm))
        py = np.array(field_to_pixel(tel, th[0], th[1] + h, fp_to_pix, rot_tel_pos, wave_nm))
        J = np.stack([(px - p0) / h, (py - p0) / h], axis=1)
        th = th - np.linalg.solve(J, p0 - target)
    # hexapolar grid of field angles of radius 0.16 deg about the detector centre (batoid_wcs.py:408-427)
    rings = [(0, 1)] + [(k, 6 * k) for k in range(1, 7)]
    pts = []
    for k, m in rings:
        for j in range(m):
            r = math.radians(0.16) * k / 6.0
            a = 2 * math.pi * j / m
            pts.append((th[0] + r * math.cos(a), th[1] + r * math.sin(a)))
    pts = np.array(pts)
    pix = np.array([field_to_pixel(tel, a, b, fp_to_pix, rot_tel_pos, wave_nm) for a, b in pts])
    vec = wcsmod.tansip_pix_to_vec(icrf_to_field, pts[:, 0], pts[:, 1])
    img_wcs = wcsmod.fit_tansip(pix[:, 0], pix[:, 1], vec, crpix=target, order=order)
    return img_wcs, icrf_to_field, th
