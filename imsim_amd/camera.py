"""Amplifier geometry and electronics of the LSSTCam CCDs (mirror of imsim/camera.py:22-216: Amp, CCD, Camera).

The reference fills these classes from `lsst.obs.lsst` (`get_camera`, imsim/camera.py:148-171), which is neither in
/root/reference nor installed.  The geometry below is the published LSSTCam segment layout and reproduces the
known answers of the reference's tests/test_readout.py:80-91 (raw segment 576 x 2048, DATASEC [11:522,1:2002],
DETSEC [512:1,4004:2003] for Segment10 and [4096:3585,4004:2003] for Segment17 of an E2V CCD); gains, read noise,
full well and crosstalk are STAND-IN values (obs_lsst data).  `Camera.from_json` / `to_json` exchange a complete
description, so a site with the LSST stack can dump the real one.
"""
import json
import os
from collections import namedtuple


class Bounds(namedtuple("Bounds", "xmin xmax ymin ymax")):
    """galsim.BoundsI stand-in: 1-based, inclusive."""
    __slots__ = ()

    def numpyShape(self):
        return (self.ymax - self.ymin + 1, self.xmax - self.xmin + 1)


# channel order of the amplifier list (readout.py:489 `channels`) and the segment layout per vendor
CHANNELS = ["C10", "C11", "C12", "C13", "C14", "C15", "C16", "C17", "C07", "C06", "C05", "C04", "C03", "C02", "C01", "C00"]
SEGMENT = {"E2V": dict(seg=(512, 2002), raw=(576, 2048), prescan=10),
           "ITL": dict(seg=(509, 2000), raw=(576, 2048), prescan=3)}
# (raw_flip_x, raw_flip_y) of the upper (C1x) and lower (C0x) row
FLIPS = {"E2V": {"C1": (True, True), "C0": (False, False)},
         "ITL": {"C1": (True, True), "C0": (True, False)}}
# stand-ins for the obs_lsst electronics data
STANDIN_GAIN0, STANDIN_GAIN_STEP = 1.40, 0.02          # gain of channel k in CHANNELS order: 1.40 + 0.02 k  [e-/ADU]
STANDIN_READ_NOISE = 5.0                               # ADU
STANDIN_SATURATION = {"E2V": 90000.0, "ITL": 80000.0}  # ADU above zero: full_well = (saturation - bias) * gain
STANDIN_XTALK = 2.0e-4                                 # nearest neighbour in the same row; halves per further segment

RAFTS = ["R01", "R02", "R03", "R10", "R11", "R12", "R13", "R14", "R20", "R21", "R22", "R23", "R24", "R30", "R31",
         "R32", "R33", "R34", "R41", "R42", "R43"]
SENSORS = ["S00", "S01", "S02", "S10", "S11", "S12", "S20", "S21", "S22"]
ITL_RAFTS = {"R01", "R02", "R03", "R10", "R20", "R41", "R42", "R43"}


def det_type_of(det_name):
    return "ITL" if det_name[:3] in ITL_RAFTS else "E2V"


class Amp:
    """Pixel geometry and readout properties of one amplifier segment (imsim/camera.py:22-79)."""
    FIELDS = ("bounds", "raw_flip_x", "raw_flip_y", "gain", "full_well", "raw_bounds", "raw_data_bounds", "read_noise",
              "bias_level")

    def __init__(self):
        for f in self.FIELDS:
            setattr(self, f, None)

    def update(self, other):
        self.__dict__.update(other.__dict__)


class CCD(dict):
    """Amps keyed by amplifier name, plus CCD-level data (imsim/camera.py:81-145)."""

    def __init__(self):
        super().__init__()
        self.bounds = None
        self.xtalk = None
        self.full_well = None
        self.serial = None

    def getSerial(self):
        return self.serial

    def update(self, other):
        self.__dict__.update(other.__dict__)
        for key, value in other.items():
            if key not in self:
                self[key] = Amp()
            self[key].update(value)


def make_ccd(det_name, bias_level=1000.0, bias_levels_dict=None, xtalk=True, vendor=None):
    vendor = vendor or det_type_of(det_name)
    g = SEGMENT[vendor]
    w, h = g["seg"]
    rw, rh = g["raw"]
    ccd = CCD()
    ccd.bounds = Bounds(1, 8 * w, 1, 2 * h)
    ccd.serial = f"{vendor}-{det_name}"
    for k, name in enumerate(CHANNELS):
        amp = Amp()
        col = int(name[2])
        upper = name[1] == "1"
        amp.bounds = Bounds(col * w + 1, (col + 1) * w, (h if upper else 0) + 1, (2 * h if upper else h))
        amp.raw_flip_x, amp.raw_flip_y = FLIPS[vendor][name[:2]]
        amp.gain = STANDIN_GAIN0 + STANDIN_GAIN_STEP * k
        amp.bias_level = float(bias_levels_dict[name]) if bias_levels_dict is not None else float(bias_level)
        amp.full_well = (STANDIN_SATURATION[vendor] - 1000.0) * amp.gain
        amp.raw_bounds = Bounds(1, rw, 1, rh)
        amp.raw_data_bounds = Bounds(g["prescan"] + 1, g["prescan"] + w, 1, h)
        amp.read_noise = STANDIN_READ_NOISE
        ccd[name] = amp
    ccd.full_well = max(a.full_well for a in ccd.values())
    if xtalk:
        # victim i, aggressor j (readout.py:403-411): neighbours in the same row couple, falling off by halves
        ccd.xtalk = [[0.0] * 16 for _ in range(16)]
        for i in range(16):
            for j in range(16):
                if i != j and (i < 8) == (j < 8):
                    ccd.xtalk[i][j] = STANDIN_XTALK * 0.5 ** (abs(i - j) - 1)
    return ccd


class Camera(dict):
    """CCDs keyed by detector name, e.g. 'R22_S11' (imsim/camera.py:174-216)."""

    def __init__(self, camera_class="LsstCamSim", bias_levels_file=None, bias_level=1000.0, data_dir=None):
        super().__init__()
        if camera_class not in ("LsstCamSim", "LsstComCamSim"):
            raise ValueError("Invalid camera: %s" % camera_class)
        self.camera_class = camera_class
        levels = None
        if bias_levels_file is not None:
            if not os.path.isfile(bias_levels_file):
                cand = os.path.join(data_dir or os.path.join(os.path.dirname(os.path.abspath(__file__)), "data"), bias_levels_file)
                if not os.path.isfile(cand):
                    raise FileNotFoundError(f"{bias_levels_file} not found.")
                bias_levels_file = cand
            with open(bias_levels_file) as fobj:
                levels = json.load(fobj)
        names = ([f"R22_{s}" for s in SENSORS] if camera_class == "LsstComCamSim"
                 else [f"{r}_{s}" for r in RAFTS for s in SENSORS])
        for det_name in names:
            self[det_name] = make_ccd(det_name, bias_level=bias_level,
                                      bias_levels_dict=None if levels is None else levels[det_name],
                                      vendor="ITL" if camera_class == "LsstComCamSim" else None)    # ComCam: ITL sensors

    def to_json(self, path):
        out = {}
        for det, ccd in self.items():
            out[det] = dict(bounds=list(ccd.bounds), xtalk=ccd.xtalk, full_well=ccd.full_well, serial=ccd.serial,
                            amps={n: {f: (list(getattr(a, f)) if isinstance(getattr(a, f), tuple) else getattr(a, f))
                                      for f in Amp.FIELDS} for n, a in ccd.items()})
        with open(path, "w") as fobj:
            json.dump(out, fobj)

    @classmethod
    def from_json(cls, path, camera_class="LsstCamSim"):
        """A camera description dumped by `to_json` (e.g. from the real obs_lsst camera at a site with the stack)."""
        self = dict.__new__(cls)
        dict.__init__(self)
        self.camera_class = camera_class
        with open(path) as fobj:
            data = json.load(fobj)
        for det, d in data.items():
            ccd = CCD()
            ccd.bounds, ccd.xtalk, ccd.full_well, ccd.serial = Bounds(*d["bounds"]), d["xtalk"], d["full_well"], d["serial"]
            for n, fields in d["amps"].items():
                amp = Amp()
                for f, v in fields.items():
                    setattr(amp, f, Bounds(*v) if isinstance(v, list) else v)
                ccd[n] = amp
            self[det] = ccd
        return self
