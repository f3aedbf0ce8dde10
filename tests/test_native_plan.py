"""The native LSST_Image planner (ims_plan_lsst_image, csrc/ims_plan.h) against the numpy planner's decisions
(engine.plan_bf_groups): host code only, no GPU."""
import ctypes as C

import numpy as np
import pytest

from imsim_amd import _abi
from imsim_amd.engine import plan_bf_groups
from helpers import c3_small_case


def _plan(objects, nrecalc, scratch_cells, static_cells=513 * 513, n_static=1, capacity=4096, classes=(40, 6), max_pool=1 << 40,
          want_realized=True):
    lib = _abi.load()
    n_phot = np.ascontiguousarray(objects["n_phot"], dtype=np.int64)
    stamp = np.stack([objects["stamp_xmin"], objects["stamp_xmax"], objects["stamp_ymin"], objects["stamp_ymax"]], axis=1).astype(np.int32)
    faint = ((objects["flags"] & _abi.IMS_OBJ_FAINT) != 0).astype(np.uint8)
    inp = _abi.PlanInput()
    inp.n, inp.n_phot, inp.stamp, inp.faint = len(n_phot), n_phot.ctypes.data, stamp.ctypes.data, faint.ctypes.data
    inp.nrecalc, inp.n_class_rounds = nrecalc, len(classes)
    for k, v in enumerate(classes):
        inp.class_rounds[k] = v
    inp.n_static_slots, inp.slot_capacity, inp.static_cells, inp.scratch_cells = n_static, capacity, static_cells, scratch_cells
    inp.max_pool_photons, inp.seg_size, inp.want_realized, inp.event_base = max_pool, 256, int(want_realized), 3000
    handle, sizes = C.c_void_p(), _abi.PlanSizes()
    rc = lib.ims_plan_lsst_image(C.byref(inp), C.byref(handle), C.byref(sizes))
    if rc == 0:
        lib.ims_plan_destroy(handle)
    return rc, sizes


@pytest.mark.parametrize("nrecalc,scratch", [(10000, 2_000_000), (1000, 2_000_000), (1000, 80_000), (500, 120_000)])
def test_native_planner_agrees_with_the_numpy_planner(nrecalc, scratch):
    scene, objects = c3_small_case(n_obj=300, scratch=scratch)
    objects = objects.copy()
    objects["n_phot"][5] = 0                                    # an object without photons is skipped
    rc, z = _plan(objects, nrecalc, scratch)
    assert rc == 0
    normal, groups = plan_bf_groups(objects, nrecalc, 1, 513 * 513, scratch, 4096)
    n = objects["n_phot"].astype(np.int64)
    bright = np.concatenate([g[0] for g in groups]) if groups else np.zeros(0, dtype=np.int64)
    assert z.n_groups == max(len(groups), 1)
    assert z.n_objects == np.count_nonzero(n > 0)
    assert z.render_photons == n[normal].sum() and z.render_rows == np.count_nonzero(n[normal] > 0) and z.n_render_launches == 1
    assert z.shoot_photons == n[bright].sum() and z.chain_rows == len(bright)
    assert z.pool_photons == max([int(n[g[0]].sum()) for g in groups] + [0])
    assert z.render_segments == int(((n[normal] + 255) // 256).sum())
    assert z.realized_count == z.render_rows + z.chain_rows
    # pool slices: every class is shot in slices that start at rounds 0, 1, 3, 8, 20, 60
    n_slices = 0
    for idx, _ in groups:
        rounds = (n[idx] + nrecalc - 1) // nrecalc
        cuts = sorted({0, len(idx)} | {int(np.count_nonzero(rounds >= t)) for t in (40, 6)})
        for a in cuts[:-1]:
            n_slices += 1 + sum(1 for e in (1, 3, 8, 20, 60) if e < rounds[a])
    assert z.n_shoot_launches == n_slices
    assert z.arena_bytes % 256 == 0 and z.rows_bytes >= 256 * (z.render_rows + z.chain_rows + z.shoot_rows)


def test_native_planner_argument_errors_and_limits():
    scene, objects = c3_small_case(n_obj=120)
    rc, _ = _plan(objects, 1000, 100)                            # scratch smaller than one stamp
    assert rc != 0 and b"scratch capacity" in _abi.load().ims_last_error()
    rc, z = _plan(objects, 0, 0)                                 # no sensor: one fused launch, nothing else
    assert rc == 0 and z.n_groups == 1 and z.n_shoot_launches == 0 and z.pool_photons == 0 and z.n_events == 0
    rc, z = _plan(objects[:0], 1000, 1000)                       # an empty table
    assert rc == 0 and z.n_groups == 0 and z.n_objects == 0 and z.realized_count == 0
    # a pool limit splits the bright objects into more groups, each holding at most the limit (or one object)
    n = objects["n_phot"].astype(np.int64)
    lim = int(n.max()) + 1
    rc, z = _plan(objects, 1000, 2_000_000, max_pool=lim)
    _, groups = plan_bf_groups(objects, 1000, 1, 513 * 513, 2_000_000, 4096, lim)
    assert rc == 0 and z.n_groups == len(groups) and z.pool_photons <= lim
    lib = _abi.load()
    assert lib.ims_plan_lsst_image(None, None, None) != 0
