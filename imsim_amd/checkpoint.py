"""Checkpointing of a CCD in progress (imsim/checkpoint.py:10-121): named records in one file per CCD, written with the
reference's crash-safe sequence (move the file to _bak, copy to _new, edit _new, move _new into place, delete _bak) and
recovered with its four-state logic (:43-64).

Container: the reference keeps pickles in an HDF5 file (h5py, absent from this image).  When h5py can be imported the
same layout is written (one uint8 dataset of pickle bytes per record); otherwise the records go into a zip archive
of .pkl members under the same file name -- same API, same recovery behaviour, different container, stated in the
file's first member `FORMAT`.
"""
import os
import pickle
import shutil
import zipfile

import numpy as np

try:                                          # pragma: no cover -- not installed here
    import h5py
except ImportError:
    h5py = None


class Checkpointer:
    _req_params = {"file_name": str}
    _opt_params = {"dir": str}

    def __init__(self, file_name, dir=None, logger=None):
        self.file_name = os.path.join(dir, file_name) if dir is not None else file_name
        d = os.path.dirname(self.file_name)
        if d:
            os.makedirs(d, exist_ok=True)
        self.file_name_bak = self.file_name + "_bak"
        self.file_name_new = self.file_name + "_new"
        self.log = []
        if os.path.isfile(self.file_name):                       # cases C or D
            self.log.append("exists")
            if os.path.isfile(self.file_name_bak):               # C: failed between steps 4 and 5
                os.remove(self.file_name_bak)
                self.log.append("deleted backup")
        elif os.path.isfile(self.file_name_bak):                 # B: failed between steps 1 and 4
            os.rename(self.file_name_bak, self.file_name)
            self.log.append("recovered from backup")
            if os.path.isfile(self.file_name_new):
                os.remove(self.file_name_new)
        else:
            self.log.append("none")                              # A: starting from scratch

    # -- container --
    def _write_record(self, path, name, blob):
        if h5py is not None:                                     # pragma: no cover
            with h5py.File(path, "a") as hdf:
                if name in hdf:
                    del hdf[name]
                hdf.create_dataset(name, data=np.frombuffer(blob, dtype=np.uint8))
            return
        records = {}
        if os.path.isfile(path):
            with zipfile.ZipFile(path, "r") as z:
                records = {n: z.read(n) for n in z.namelist()}
        records["FORMAT"] = b"imsim_amd checkpoint: zip of pickles (h5py unavailable); record <name>.pkl"
        records[name + ".pkl"] = blob
        with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as z:
            for n, b in records.items():
                z.writestr(n, b)

    def _read_record(self, path, name):
        if h5py is not None:                                     # pragma: no cover
            with h5py.File(path, "r") as hdf:
                return None if name not in hdf else hdf[name][:].tobytes()
        with zipfile.ZipFile(path, "r") as z:
            return z.read(name + ".pkl") if name + ".pkl" in z.namelist() else None

    # -- the reference's interface --
    def save(self, name, data):
        if os.path.isfile(self.file_name):
            os.rename(self.file_name, self.file_name_bak)                 # 1
            shutil.copy(self.file_name_bak, self.file_name_new)           # 2
        self._write_record(self.file_name_new, name, pickle.dumps(data))  # 3
        os.rename(self.file_name_new, self.file_name)                     # 4
        if os.path.isfile(self.file_name_bak):
            os.remove(self.file_name_bak)                                 # 5

    def load(self, name):
        """the record, or None when there is no file or no such record"""
        if not os.path.isfile(self.file_name):
            return None
        blob = self._read_record(self.file_name, name)
        return None if blob is None else pickle.loads(blob)
