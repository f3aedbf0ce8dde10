"""The oracle fed by parameter blocks that were NOT built by the product's host code.

Everywhere else the oracle and the HIP path receive the structs imsim_amd.engine.BoundScene fills (oracle/orc_loader.py),
so a wrong table order, slot layout or constant there would be common to both sides and no parity test would see it.  Here
the Silicon sensor block of the reference's sensor-model case (tests/test_sensor_models.py:42-59, plus tree rings) is
assembled from the model files by code written for this test alone -- its own .cfg / .dat reader, its own vertex
placement and vertex -> table order, its own diffusion step, its own memory layout of the boundary state -- and the oracle
must render the same image and end with the same pixel-boundary state as through BoundScene.  Only the ctypes layouts of
the structs are shared (tests/test_abi.py pins those to include/imsim_hip.h)."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from imsim_amd import _abi, sensor as sensormod
from imsim_amd._abi import OBJECT_DTYPE, BFSLOT_DTYPE
from imsim_amd.engine import Scene, SensorSetup, make_slots
from oracle import orc_loader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 17


def _read_model(stem):
    """Poisson_CCD22 files read independently of imsim_amd.sensor: returns what ims_sensor_t needs."""
    cfg = {}
    for line in open(stem + ".cfg"):
        body = line.partition("#")[0]
        if "=" in body:
            k, _, v = body.partition("=")
            cfg[k.strip()] = v.split()
    nV = int(cfg["NumVertices"][0])
    nx, ny = int(cfg["PixelBoundaryNx"][0]), int(cfg["PixelBoundaryNy"][0])
    pix = float(cfg["PixelSizeX"][0])
    thick = float(cfg["SensorThickness"][0])
    nv = 4 * nV + 4
    # nominal polygon: corners and, on every edge, NumVertices points at (1 + tan(angle)) / 2, the angles dividing the
    # quarter turn the edge subtends into NumVertices + 1 equal parts; counter-clockwise from the lower-left corner
    frac = [(1.0 + math.tan(-0.25 * math.pi + (m + 1) * 0.5 * math.pi / (nV + 1))) / 2.0 for m in range(nV)]
    poly = [(0.0, 0.0)] + [(f, 0.0) for f in frac] + [(1.0, 0.0)] + [(1.0, f) for f in frac]
    poly += [(1.0, 1.0)] + [(f, 1.0) for f in reversed(frac)] + [(0.0, 1.0)] + [(0.0, f) for f in reversed(frac)]
    poly = np.array(poly)
    rows = np.array([[float(t) for t in line.split()] for line in list(open(stem + ".dat"))[1:] if line.strip()])
    assert rows.shape == (nx * ny * nv, 5)
    table = np.zeros((nx, ny, nv, 2))
    # a pixel's nv rows are contiguous; which row is which vertex: by the nominal POSITION nearest to the listed one, decided
    # on the corner pixel (where the listed positions are the nominal ones to 1e-3 pixel) and applied to every pixel
    first = rows[:nv]
    rel = (first[:, 3:5] - first[:, 0:2]) / pix + 0.5
    vertex_of_row = np.array([int(np.argmin(((poly - p) ** 2).sum(axis=1))) for p in rel])
    assert sorted(vertex_of_row.tolist()) == list(range(nv))
    ll = float(cfg.get("PixelBoundaryLowerLeft", ["10.0", "10.0"])[0])
    for b in range(nx * ny):
        blk = rows[b * nv:(b + 1) * nv]
        i, j = int((blk[0, 0] - ll) // pix), int((blk[0, 1] - ll) // pix)
        for r in range(nv):
            v = vertex_of_row[r]
            table[i, j, v, 0] = (blk[r, 3] - blk[r, 0]) / pix + 0.5 - poly[v, 0]
            table[i, j, v, 1] = (blk[r, 4] - blk[r, 1]) / pix + 0.5 - poly[v, 1]
    # diffusion step at the entrance surface (doc/validation/diffusion.rst:66-93)
    f = lambda k, d=None: float(cfg[k][0]) if k in cfg else d
    phases, collecting = f("NumPhases"), f("CollectingPhases")
    cs = 2.0 * (f("ChannelStopWidth") / 2.0 + f("FieldOxideTaper", 0.0))
    a_cs, a_col = cs * pix, (pix - cs) * pix * collecting / phases
    a_bar = (pix - cs) * pix * (phases - collecting) / phases
    v_front = (a_cs * f("qfh", 0.0) + a_col * (f("Vparallel_hi") + 12.0) + a_bar * (f("Vparallel_lo") + 15.0)) / pix ** 2
    v_diff = max(v_front - f("Vbb"), 1.0)
    diff_step = math.sqrt(2 * 0.026 * f("CCDTemperature") / 298.0 / v_diff / 0.27) * thick
    return dict(nV=nV, nx=nx, ny=ny, pix=pix, thick=thick, table=table, poly=poly, diff_step=diff_step,
                num_elec=float(cfg["CollectedCharge_0_0"][0]))


@pytest.mark.parametrize("model_name", ["lsst_e2v_50_4", "lsst_itl_50_8"])
def test_oracle_with_independently_filled_sensor_block(model_name):
    stem = os.path.join(ROOT, "imsim_amd", "data", "sensor_models", model_name)
    n_phot = 300000
    tr_r = np.arange(0, 200) * 3.0
    tr = 0.01 * np.sin(tr_r / 7.0)                       # a tree-ring table f(r), linear interpolation
    obj = np.zeros(1, dtype=OBJECT_DTYPE)
    obj["obj_id"], obj["n_phot"], obj["x0"], obj["y0"], obj["flux_per_photon"] = 5, n_phot, 9.0, 9.0, 1.0
    obj["jac"], obj["winv"] = (1, 0, 0, 1), (1 / 0.3, 0, 0, 1 / 0.3)
    obj["prof_table"], obj["sed_table"], obj["sed_wave"] = -1, -1, 600.0
    obj["stamp_xmin"], obj["stamp_xmax"], obj["stamp_ymin"], obj["stamp_ymax"] = 1, N, 1, N
    obj["bf_state"] = 1

    # (a) through the product's host code
    sc = Scene(nx=N, ny=N, seed=77, psf=[(_abi.IMS_PSF_GAUSSIAN, 0, 0.3, 0.0, 1.0)], ops=[])
    sc.sensor = SensorSetup(model=sensormod.load_silicon_model(stem), abs_wl=np.array([300.0, 1100.0]),
                            abs_len=np.array([1.0, 1.0]), tr_table=tr, tr_dr=3.0, tr_center=(-40.0, 25.0),
                            slots=make_slots([(1, 1, N, N), (1, 1, N, N)]))
    ref = orc_loader.OracleScene(sc)
    ref.render(obj, nrecalc=10000)

    # (b) every block assembled here
    lib = orc_loader.load()
    m = _read_model(stem)
    keep = []

    def ptr(a, dt=np.float64):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a.ctypes.data

    S = _abi.Sensor()
    S.kind, S.num_vertices, S.nx, S.ny, S.qdist = _abi.IMS_SENSOR_SILICON, m["nV"], m["nx"], m["ny"], 3
    S.num_elec, S.pixel_size, S.thickness, S.diff_step = m["num_elec"], m["pix"], m["thick"], m["diff_step"]
    S.n_abs, S.abs_wl_min, S.abs_wl_step = 2, 300.0, 800.0
    S.abs_len = ptr([1.0, 1.0])
    S.n_tr, S.tr_dr, S.tr_cx, S.tr_cy = len(tr), 3.0, -40.0, 25.0
    S.tr_table = ptr(tr)
    S.distortions = ptr(m["table"])
    S.emptypoly = ptr(m["poly"])
    cells_per_slot = (N + 1) * (N + 1)
    slots = np.zeros(2, dtype=BFSLOT_DTYPE)
    for k in range(2):
        slots[k] = (1, 1, N, N, k * cells_per_slot)
    keep.append(slots)
    S.n_bf_slots, S.bf_slots = 2, slots.ctypes.data
    npo = 2 * m["nV"] + 2
    boundary = np.zeros(2 * cells_per_slot * npo * 2)
    bounds = np.zeros(2 * cells_per_slot * 8)
    delta = np.zeros(2 * cells_per_slot)
    S.bf_boundary, S.bf_bounds, S.bf_delta = boundary.ctypes.data, bounds.ctypes.data, delta.ctypes.data
    S.pristine_margin = -1.0
    P = _abi.RenderParams()
    P.seed, P.seg_size, P.n_psf = 77, 256, 1
    P.psf[0] = _abi.PsfComponent(_abi.IMS_PSF_GAUSSIAN, 0, 0.3, 0.0, 1.0)
    P.sensor = C.addressof(S)
    image = np.zeros((N, N))
    prefix = np.array([0, (n_phot + 255) // 256], dtype=np.int64)
    P.objects, P.n_objects, P.seg_prefix, P.n_segments = obj.ctypes.data, 1, prefix.ctypes.data, int(prefix[-1])
    P.image, P.nx, P.ny, P.xmin, P.ymin = image.ctypes.data, N, N, 1, 1
    lib.orc_sensor_init_boundaries(C.addressof(S), 0, 2)
    assert lib.orc_render_objects(C.byref(P), 10000, image.ctypes.data, None) == 0

    assert image.sum() > 0.99 * n_phot
    assert np.array_equal(image, ref.image64), "image: independent fill vs BoundScene"
    assert np.array_equal(boundary, ref.sensor_array("boundary")[:len(boundary)]), "boundary points"
    assert np.array_equal(bounds, ref.sensor_array("bounds")[:len(bounds)]), "bounds lines"
    # and the charge did move the boundaries of the private slot (the comparison is not vacuous)
    assert not np.array_equal(boundary[:cells_per_slot * npo * 2], boundary[cells_per_slot * npo * 2:])
