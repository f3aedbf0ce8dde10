"""Silicon sensor model: host-side setup mirroring galsim.SiliconSensor as imSim configures it
(config/imsim-config.yaml:230-235; model name chosen in imsim/lsst_image.py:93-103 from
data/sensor_models/lsst_{itl,e2v}_50_{4,8,32}.{cfg,dat}).

The numerics run on the GPU (imsim_amd/csrc); this module only parses the Poisson_CCD22 model
files into the vertex-displacement table the kernels consume.
"""
import dataclasses
import math
import os

import numpy as np


def read_sensor_cfg(path):
    """Parse a Poisson_CCD22 .cfg file into a dict (values: int, float, str or list)."""
    out = {}
    with open(path) as f:
        for line in f:
            line = line.split("#", 1)[0].strip()
            if "=" not in line:
                continue
            key, val = [s.strip() for s in line.split("=", 1)]
            parts = val.split()
            conv = []
            for p in parts:
                try:
                    conv.append(int(p))
                except ValueError:
                    try:
                        conv.append(float(p))
                    except ValueError:
                        conv.append(p)
            out[key] = conv[0] if len(conv) == 1 else conv
    return out


def calculate_diff_step(cfg):
    """Diffusion step [micron] for conversion at the entrance surface -- the formula documented in
    doc/validation/diffusion.rst:66-93 (effective-mass MobilityFactor and simulated front voltage)."""
    num_phases = cfg["NumPhases"]
    collecting = cfg["CollectingPhases"]
    pixel = cfg["PixelSizeX"]
    thickness = cfg["SensorThickness"]
    v_channel_stop = cfg.get("qfh", 0.0)
    v_collect = cfg["Vparallel_hi"] + 12.0
    v_barrier = cfg["Vparallel_lo"] + 15.0
    cs_width = 2.0 * (cfg["ChannelStopWidth"] / 2.0 + cfg.get("FieldOxideTaper", 0.0))
    cs_area = cs_width * pixel
    collect_area = (pixel - cs_width) * pixel * collecting / num_phases
    barrier_area = (pixel - cs_width) * pixel * (num_phases - collecting) / num_phases
    v_front = (cs_area * v_channel_stop + collect_area * v_collect + barrier_area * v_barrier) / pixel ** 2
    v_diff = max(v_front - cfg["Vbb"], 1.0)
    mobility_factor = 0.27
    return math.sqrt(2 * 0.026 * cfg["CCDTemperature"] / 298.0 / v_diff / mobility_factor) * thickness


def empty_polygon(num_vertices):
    """Undistorted pixel polygon, counter-clockwise from the lower-left corner: corner, then the
    NumVertices edge points at (tan(theta)+1)/2 with theta equally spaced in angle (the vertex
    placement of the Poisson_CCD22 files: cfg comment 'NumVertices', file column Theta)."""
    nV = num_vertices
    dtheta = math.pi / (2.0 * (nV + 1.0))
    t = np.array([(math.tan(-math.pi / 4.0 + (m + 1.0) * dtheta) + 1.0) / 2.0 for m in range(nV)])
    pts = [(0.0, 0.0)] + [(tm, 0.0) for tm in t]
    pts += [(1.0, 0.0)] + [(1.0, tm) for tm in t]
    pts += [(1.0, 1.0)] + [(tm, 1.0) for tm in t[::-1]]
    pts += [(0.0, 1.0)] + [(0.0, tm) for tm in t[::-1]]
    return np.array(pts, dtype=np.float64)


@dataclasses.dataclass
class SiliconModel:
    num_vertices: int
    nx: int
    ny: int
    num_elec: float
    pixel_size: float
    thickness: float
    diff_step: float
    distortions: np.ndarray     # [nx][ny][nv][2], pixel units, per num_elec of charge in the centre pixel
    emptypoly: np.ndarray       # [nv][2]
    qdist: int = 3              # galsim.SiliconSensor default
    nrecalc: int = 10000        # galsim.SiliconSensor default


def load_silicon_model(name, strength=1.0, diffusion_factor=1.0, qdist=3, nrecalc=10000):
    """Load `<name>.cfg` / `<name>.dat` (e.g. .../data/sensor_models/lsst_e2v_50_4)."""
    cfg = read_sensor_cfg(name + ".cfg")
    nV = int(cfg["NumVertices"])
    nx, ny = int(cfg["PixelBoundaryNx"]), int(cfg["PixelBoundaryNy"])
    pixel = float(cfg["PixelSizeX"])
    nv = 4 * nV + 4
    data = np.loadtxt(name + ".dat", skiprows=1)
    if data.shape != (nx * ny * nv, 5):
        raise ValueError(f"{name}.dat has shape {data.shape}, expected {(nx * ny * nv, 5)}")
    lower_left = cfg.get("PixelBoundaryLowerLeft", [10.0, 10.0])
    empty = empty_polygon(nV)
    empty_theta = np.arctan2(empty[:, 1] - 0.5, empty[:, 0] - 0.5)
    # The Theta column is the polar angle of the (distorted) vertex about the pixel centre.  In the pixels around the
    # charged one the vertices move by more than half their spacing once there are 32 of them per edge, so a vertex cannot
    # be recognised by its angle alone; what identifies it is its PLACE in the pixel's block of nv rows, which run
    # counter-clockwise from just past the middle of the left edge.  The place -> vertex map is read off a block whose
    # pixel is far from the charge (nearest nominal angle, every vertex hit exactly once) and used for all blocks.
    blocks = data.reshape(nx * ny, nv, 5)
    if not (np.all(blocks[:, :, 0] == blocks[:, :1, 0]) and np.all(blocks[:, :, 1] == blocks[:, :1, 1])):
        raise ValueError(f"{name}.dat: the rows of a pixel are not contiguous")
    bi = np.floor((blocks[:, 0, 0] - lower_left[0]) / pixel).astype(int)
    bj = np.floor((blocks[:, 0, 1] - lower_left[1]) / pixel).astype(int)
    far = int(np.argmax(np.hypot(bi - (nx - 1) / 2.0, bj - (ny - 1) / 2.0)))
    place = np.empty(nv, dtype=int)
    for k, th in enumerate(blocks[far, :, 2]):
        d = np.angle(np.exp(1j * (empty_theta - th)))
        place[k] = int(np.argmin(np.abs(d)))
        if abs(d[place[k]]) > 0.5 * math.pi / (2.0 * (nV + 1.0)):
            raise ValueError(f"{name}.dat: cannot match vertex theta={th} of pixel ({bi[far]},{bj[far]})")
    if len(set(place.tolist())) != nv:
        raise ValueError(f"{name}.dat: the vertex order of a block is not a permutation of the polygon")
    dist = np.zeros((nx, ny, nv, 2))
    seen = np.zeros((nx, ny), dtype=bool)
    for b in range(nx * ny):
        i, j = int(bi[b]), int(bj[b])
        if not (0 <= i < nx and 0 <= j < ny) or seen[i, j]:
            raise ValueError(f"{name}.dat: pixel ({i},{j}) out of range or listed twice")
        seen[i, j] = True
        dist[i, j, place, 0] = (blocks[b, :, 3] - blocks[b, :, 0]) / pixel + 0.5 - empty[place, 0]
        dist[i, j, place, 1] = (blocks[b, :, 4] - blocks[b, :, 1]) / pixel + 0.5 - empty[place, 1]
    if not seen.all():
        raise ValueError(f"{name}.dat: missing pixels")
    num_elec = float(cfg["CollectedCharge_0_0"]) / strength
    return SiliconModel(num_vertices=nV, nx=nx, ny=ny, num_elec=num_elec, pixel_size=pixel,
                        thickness=float(cfg["SensorThickness"]),
                        diff_step=calculate_diff_step(cfg) * diffusion_factor,
                        distortions=np.ascontiguousarray(dist), emptypoly=empty, qdist=qdist, nrecalc=nrecalc)


def sensor_model_path(data_dir, det_type):
    """imsim/lsst_image.py:93-103: ITL -> lsst_itl_50_4, E2V -> lsst_e2v_50_4 (the 8- and 32-vertex files of the same
    directory load by name: image.sensor.name)."""
    return os.path.join(data_dir, "sensor_models", {"ITL": "lsst_itl_50_4", "E2V": "lsst_e2v_50_4"}[det_type])
