"""The C-ABI library loads, exports every symbol include/imsim_hip.h declares, and its struct
layouts match the ctypes binding.  No compute calls (runs without a GPU)."""
import ctypes as C
import os
import re

import pytest

from imsim_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "imsim_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ims_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = _abi.load()
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/imsim_hip.h but not exported"
    assert sorted(_abi.EXPORTS) == names


def test_struct_sizes_match_binding():
    lib = _abi.load()
    for k, st in enumerate(_abi.STRUCTS):
        assert lib.ims_struct_size(k) == C.sizeof(st), st.__name__
    assert C.sizeof(_abi.Object) == 256 == _abi.OBJECT_DTYPE.itemsize
    assert C.sizeof(_abi.BfSlot) == _abi.BFSLOT_DTYPE.itemsize


def test_abi_version_and_error_string():
    lib = _abi.load()
    assert lib.ims_abi_version() == 6
    # argument checking happens before any HIP call, so it is testable without a GPU
    assert lib.ims_shoot_accumulate(None, None) == -1
    assert b"NULL" in lib.ims_last_error()


def test_no_cpu_fallback_without_gpu():
    """The product path must fail loudly when there is no GPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from imsim_amd import configs
    from imsim_amd.engine import Renderer
    with pytest.raises(_abi.ImsimHipError):
        Renderer(configs.scene_c2(nx=64, ny=64))


def test_product_does_not_import_oracle():
    """Nothing under imsim_amd/ may reference oracle/."""
    pkg = os.path.join(ROOT, "imsim_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dp, f)).read()
                assert "orc_" not in text and "liboracle" not in text and "import oracle" not in text \
                    and "from oracle" not in text, f


def test_argument_errors_are_reported_before_any_launch():
    """Error behaviour of the C-ABI: every entry point returns a negative code and sets ims_last_error; the checks
    run before the first HIP call, so they are testable without a GPU."""
    lib = _abi.load()
    P = _abi.RenderParams()
    P.seg_size, P.n_objects, P.n_segments = 128, 0, 0
    assert lib.ims_shoot_accumulate(C.byref(P), None) < 0 and b"seg_size" in lib.ims_last_error()
    P.seg_size, P.n_psf = 256, _abi.IMS_MAX_PSF + 1
    assert lib.ims_shoot_accumulate(C.byref(P), None) < 0 and b"n_psf" in lib.ims_last_error()
    P.n_psf, P.n_ops = 0, _abi.IMS_MAX_OPS + 1
    assert lib.ims_shoot_accumulate(C.byref(P), None) < 0 and b"n_ops" in lib.ims_last_error()
    P.n_ops = 1
    P.ops[0].kind = _abi.IMS_OP_RUBIN_OPTICS                     # ray tracing without an optics descriptor
    assert lib.ims_shoot_accumulate(C.byref(P), None) < 0 and b"optics" in lib.ims_last_error()
    P.n_ops, P.n_psf = 0, 1
    P.psf[0].kind = _abi.IMS_PSF_SCREENS                         # phase screens without an atmosphere
    assert lib.ims_shoot_accumulate(C.byref(P), None) < 0 and b"atmosphere" in lib.ims_last_error()
    P.n_psf, P.n_objects = 0, 5                                  # objects announced but no table
    assert lib.ims_shoot_accumulate(C.byref(P), None) < 0 and b"NULL" in lib.ims_last_error()
    P.n_objects = 0                                              # valid and empty, but no image
    assert lib.ims_shoot_accumulate(C.byref(P), None) < 0 and b"image" in lib.ims_last_error()
    # sensor entry points need the host copy of the slot table
    assert lib.ims_sensor_init_boundaries(None, None, 0, 1, None) < 0
    S = _abi.Sensor()
    assert lib.ims_sensor_init_boundaries(C.byref(S), C.byref(S), 0, 1, None) < 0 and b"slot" in lib.ims_last_error()
    assert lib.ims_flat_add(None, None, 1.0, 1.0, 0, 0, 8, 8, None, None, None) < 0 and b"image" in lib.ims_last_error()
    assert lib.ims_image_to_float(None, None, 8, None) < 0
    # a plan that names a stream it was not given
    it = _abi.PlanItem()
    it.kind, it.stream = _abi.IMS_PLAN_RECORD, 3
    streams = (C.c_void_p * 1)(None)
    assert lib.ims_run_plan(C.byref(it), 1, None, None, None, streams, 1) < 0 and b"stream" in lib.ims_last_error()
    # the derived-field helpers are pure host code
    op = _abi.Op()
    op.kind = _abi.IMS_OP_PHOTON_DCR
    op.p[0], op.p[1], op.p[2], op.p[3] = 620.0, 69.328, 293.15, 1.067
    assert lib.ims_fill_derived_op(C.byref(op)) == 0
    assert 1.0e-7 < op.p[5] < 1.0e-5 and 0.0 < op.p[6] < 1.0e-5 and 1.0e-4 < op.p[7] < 4.0e-4      # n - 1 ~ 1.9e-4 at 69 kPa
