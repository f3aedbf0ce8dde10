"""TAN-SIP world coordinate systems in the trig-free vector form the kernels evaluate.

The photon operators consume two of these (imsim/photon_ops.py:454-483): `img_wcs`
(base['current_image'].wcs, a galsim.FittedSIPWCS of order 3 built by imsim/batoid_wcs.py:429-453)
and `icrf_to_field` (imsim/batoid_wcs.py:499-506).  Building them by ray tracing is host work done
once per CCD (`fit_tansip`); evaluating them per photon is the kernels' job.
"""
import numpy as np

from . import _abi


def unit_vector(ra, dec):
    return np.array([np.cos(dec) * np.cos(ra), np.cos(dec) * np.sin(ra), np.sin(dec)])


def tangent_basis(ra, dec, rot=0.0):
    """Rows e0 (tangent point), e1 (+xi), e2 (+eta); `rot` rotates (xi, eta) about e0 [rad]."""
    e0 = unit_vector(ra, dec)
    east = np.array([-np.sin(ra), np.cos(ra), 0.0])
    north = np.cross(e0, east)
    c, s = np.cos(rot), np.sin(rot)
    e1 = c * east + s * north
    e2 = -s * east + c * north
    return np.stack([e0, e1, e2])


def make_tansip(crpix, cd, basis, a=None, b=None, order=0):
    w = _abi.TanSip()
    w.crpix[0], w.crpix[1] = float(crpix[0]), float(crpix[1])
    cd = np.asarray(cd, dtype=np.float64).reshape(2, 2)
    cdinv = np.linalg.inv(cd)
    for k in range(4):
        w.cd[k] = float(cd.reshape(-1)[k])
        w.cdinv[k] = float(cdinv.reshape(-1)[k])
    for k in range(9):
        w.rot[k] = float(np.asarray(basis).reshape(-1)[k])
    w.order = int(order)
    if a is not None:
        for p in range(5):
            for q in range(5):
                w.a[p * 5 + q] = float(a[p][q])
                w.b[p * 5 + q] = float(b[p][q])
    return w


def tansip_pix_to_vec(w, x, y):
    """numpy mirror of the kernel's pixel -> unit vector (host convenience)."""
    u = np.asarray(x, dtype=np.float64) - w.crpix[0]
    v = np.asarray(y, dtype=np.float64) - w.crpix[1]
    if w.order > 0:
        f = np.zeros_like(u)
        g = np.zeros_like(u)
        up = [u ** p for p in range(w.order + 1)]          # the powers once (same values as u ** p in every term)
        vq = [v ** q for q in range(w.order + 1)]
        for p in range(w.order + 1):
            for q in range(w.order + 1 - p):
                f = f + w.a[p * 5 + q] * up[p] * vq[q]
                g = g + w.b[p * 5 + q] * up[p] * vq[q]
        u, v = u + f, v + g
    xi = w.cd[0] * u + w.cd[1] * v
    eta = w.cd[2] * u + w.cd[3] * v
    inv = 1.0 / np.sqrt(1.0 + xi * xi + eta * eta)
    rot = np.array(list(w.rot)).reshape(3, 3)
    t = np.stack([inv, xi * inv, eta * inv], axis=-1)
    return t @ rot


def tansip_vec_to_pix(w, p):
    rot = np.array(list(w.rot)).reshape(3, 3)
    t = np.asarray(p) @ rot.T
    xi, eta = t[..., 1] / t[..., 0], t[..., 2] / t[..., 0]
    U = w.cdinv[0] * xi + w.cdinv[1] * eta
    V = w.cdinv[2] * xi + w.cdinv[3] * eta
    u, v = U.copy(), V.copy()
    if w.order > 0:
        for _ in range(6):
            f = np.zeros_like(u); g = np.zeros_like(u)
            fu = np.zeros_like(u); fv = np.zeros_like(u); gu = np.zeros_like(u); gv = np.zeros_like(u)
            for pp in range(w.order + 1):
                for q in range(w.order + 1 - pp):
                    a, b = w.a[pp * 5 + q], w.b[pp * 5 + q]
                    f = f + a * u ** pp * v ** q
                    g = g + b * u ** pp * v ** q
                    if pp > 0:
                        fu = fu + a * pp * u ** (pp - 1) * v ** q
                        gu = gu + b * pp * u ** (pp - 1) * v ** q
                    if q > 0:
                        fv = fv + a * q * u ** pp * v ** (q - 1)
                        gv = gv + b * q * u ** pp * v ** (q - 1)
            r0, r1 = u + f - U, v + g - V
            j00, j01, j10, j11 = 1 + fu, fv, gu, 1 + gv
            det = j00 * j11 - j01 * j10
            u = u - (j11 * r0 - j01 * r1) / det
            v = v - (j00 * r1 - j10 * r0) / det
    return u + w.crpix[0], v + w.crpix[1]


def fit_tansip(x, y, vec, crpix, order=3):
    """Least-squares TAN-SIP through pixel positions (x, y) and their sky unit vectors `vec`
    (the FittedSIPWCS step of imsim/batoid_wcs.py:429-453).  The tangent point is the sky
    direction of `crpix`, found from a first linear fit and refined once."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    vec = np.asarray(vec, dtype=np.float64)
    u, v = x - crpix[0], y - crpix[1]
    # initial tangent point: mean direction
    e0 = vec.mean(axis=0)
    e0 /= np.linalg.norm(e0)
    for _ in range(3):
        z = np.array([0.0, 0.0, 1.0])
        east = np.cross(z, e0)
        east /= np.linalg.norm(east)
        north = np.cross(e0, east)
        t0 = vec @ e0
        xi, eta = (vec @ east) / t0, (vec @ north) / t0
        terms = [(p, q) for p in range(order + 1) for q in range(order + 1 - p)]
        A = np.stack([u ** p * v ** q for p, q in terms], axis=1)
        cx, *_ = np.linalg.lstsq(A, xi, rcond=None)
        cy, *_ = np.linalg.lstsq(A, eta, rcond=None)
        # move the tangent point to the fitted sky position of crpix (constant terms -> 0)
        k0 = terms.index((0, 0))
        d = e0 + cx[k0] * east + cy[k0] * north
        e0 = d / np.linalg.norm(d)
    coef = {t: (cx[k], cy[k]) for k, t in enumerate(terms)}
    cd = np.array([[coef[(1, 0)][0], coef[(0, 1)][0]], [coef[(1, 0)][1], coef[(0, 1)][1]]])
    cdinv = np.linalg.inv(cd)
    a = np.zeros((5, 5))
    b = np.zeros((5, 5))
    for (p, q), (c0, c1) in coef.items():
        if p + q >= 2:
            fg = cdinv @ np.array([c0, c1])
            a[p, q], b[p, q] = fg
    basis = np.stack([e0, east, north])
    return make_tansip(crpix, cd, basis, a, b, order)
