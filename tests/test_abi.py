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
    assert lib.ims_abi_version() == 2
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
