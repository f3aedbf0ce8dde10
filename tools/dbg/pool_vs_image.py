import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from test_parity_gpu import _process
a = _process(**{"image.nobjects": 60, "stamp.draw_method": "phot"})
b = _process(**{"image.nobjects": 60, "image.type": "LSST_PhotonPoolingImage", "stamp.type": "LSST_Photons",
                "image.nbatch": 4, "image.nsubbatch": 3})
ta, tb = a.truth[0], b.truth[0]
np.set_printoptions(linewidth=200, suppress=True)
print("img sums", a.images[0].sum(dtype=np.float64), b.images[0].sum(dtype=np.float64))
print("realized sums", ta["realized_flux"].sum(), tb["incident_flux"].sum(), "phot", ta["phot_flux"].sum(), tb["phot_flux"].sum())
for i in range(len(ta["index"])):
    print(i, ta["index"][i], round(ta["x"][i]), round(ta["y"][i]), ta["mode"][i], ta["phot_flux"][i], ta["realized_flux"][i], tb["incident_flux"][i])
