"""Generate golden vectors for the CCD readout chain from the reference, in THIS container.  Run once; the .npz
is committed.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_readout_golden.py

* bleed trails: imsim/bleed_trails.py is numpy-only and is imported as it is.  Inputs are float64 arrays of
  integer electron counts (all sums exact), including the reference's own regression channel
  tests/data/neg_pixel_bleed.pickle (tests/test_bleed_trails.py:66-75).
* cte_matrix: imsim/readout.py imports galsim / astropy / lsst at module level and cannot be imported; the one
  numpy + scipy function is compiled from its own source text (located with `ast`, nothing is written to the repo).
"""
import ast
import importlib.util
import os
import pickle
import sys
import types

import numpy as np
import scipy.special

REF = "/root/reference"
pkg = types.ModuleType("imsim")
pkg.__path__ = [os.path.join(REF, "imsim")]
sys.modules["imsim"] = pkg
spec = importlib.util.spec_from_file_location("imsim.bleed_trails", os.path.join(REF, "imsim", "bleed_trails.py"))
bleed = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bleed)

src = open(os.path.join(REF, "imsim", "readout.py")).read()
node = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "cte_matrix")
ns = {"np": np, "scipy": scipy}
exec(compile(ast.Module(body=[node], type_ignores=[]), "readout.py:cte_matrix", "exec"), ns)
cte_matrix = ns["cte_matrix"]

out = {}
rng = np.random.default_rng(20261002)
full_well = 100000.0

# 1. one channel with a saturated star in the middle (tests/test_bleed_trails.py:41-64)
ch = np.full(2000, 800.0)
ch[980:1020] = 2 * full_well
out["chan_in"], out["chan_out"] = ch, bleed.bleed_channel(ch, full_well)

# 2. an image with several kinds of runs: near the bottom edge (charge leaves), near the top edge (closed),
#    two runs close enough to merge, a run across the midline, isolated hot pixels
img = rng.poisson(800.0, size=(96, 40)).astype(np.float64)
img[2:6, 3] += 5 * full_well            # bleeds off the bottom
img[90:95, 7] += 3 * full_well          # reaches the closed top
img[30:34, 11] += 4 * full_well         # two runs that meet
img[40:43, 11] += 6 * full_well
img[44:52, 15] += 2.5 * full_well       # straddles the midline (48)
img[60, 20] += 1.2 * full_well          # single hot pixel
img[10:80, 25] += 1.5 * full_well       # long run: more charge than the column can hold below the top
img[0:96, 30] += 2 * full_well          # a fully saturated column
img[47:49, 33] += 30 * full_well
out["img_in"] = img
out["img_midline"] = bleed.bleed_eimage(img.copy(), full_well, midline_stop=True)
out["img_nomidline"] = bleed.bleed_eimage(img.copy(), full_well, midline_stop=False)
out["full_well"] = np.array(full_well)

# 3. the reference's regression channel (a data file of its test suite)
with open(os.path.join(REF, "tests", "data", "neg_pixel_bleed.pickle"), "rb") as fobj:
    channel_data, fw = pickle.load(fobj)
cd = np.asarray(channel_data, dtype=np.float64)
out["neg_in"], out["neg_fw"] = cd, np.array(float(fw))
out["neg_out"] = bleed.bleed_channel(cd, float(fw))
out["neg_out_native"] = np.asarray(bleed.bleed_channel(np.asarray(channel_data), fw), dtype=np.float64)

# 4. CTE matrices
out["cte_64_1e-6"] = cte_matrix(64, 1.0e-6)
out["cte_64_1e-3"] = cte_matrix(64, 1.0e-3)
out["cte_40_1e-2_nt5"] = cte_matrix(40, 1.0e-2, ntransfers=5)

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "readout_golden.npz")
np.savez_compressed(path, **out)
print("wrote", path, {k: v.shape for k, v in out.items()})
print("neg channel dtype", np.asarray(channel_data).dtype, "fw", fw, "native == float64 path:",
      np.array_equal(out["neg_out"], out["neg_out_native"]))
