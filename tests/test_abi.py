"""The C-ABI library loads, exports every symbol include/imsim_hip.h declares, and its struct
layouts match the ctypes binding.  No compute calls (runs without a GPU)."""
import ctypes as C
import os
import re

import numpy as np

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


def test_tuning_block_is_the_only_way_to_change_which_kernels_the_library_launches(monkeypatch):
    """ims_tuning_t (include/imsim_hip.h): its size matches the binding, the defaults are today's fast forms, a set block is
    what a get returns, out-of-range values are refused -- and the library itself reads no environment variable for any of it
    (only the file names of the libraries it looks up at run time): the one module of the package that does, tuning.py,
    translates the environment into the block."""
    from imsim_amd import tuning
    lib = _abi.load()
    assert lib.ims_struct_size(_abi.STRUCTS.index(_abi.Tuning)) == C.sizeof(tuning.Tuning) == 72
    d = tuning.Tuning()
    assert lib.ims_tuning_defaults(C.byref(d)) == 0
    assert (d.chain_kernels, d.layout_kernels, d.psf_screens_kernel, d.photon_lds, d.round_compact, d.init_tiles, d.upd_dpp,
            d.joint_lists, d.upd_dpp_max, d.joint_list_min, d.active_fraction, d.round_two_segments, d.joint_fine_marks, d.joint_search_lists) == (1, 1, 1, -1, 1, 1, 1, 1, 128, 1024, 0.25, 0, 1, 1)
    for name in ("IMS_CHAIN_KERNELS", "IMS_LAYOUT_KERNELS", "IMS_PSF_SCREENS_KERNEL", "IMS_PHOTON_LDS", "IMS_ROUND_COMPACT", "IMS_INIT_TILES",
                 "IMS_UPD_DPP", "IMS_UPD_DPP_MAX", "IMS_JOINT_LISTS", "IMS_JOINT_LIST_MIN", "IMS_ACTIVE_FRACTION", "IMS_ROUND_TWO_SEGMENTS", "IMS_JOINT_FINE_MARKS", "IMS_JOINT_SEARCH_LISTS"):
        monkeypatch.delenv(name, raising=False)
    assert bytes(tuning.library_tuning()) == bytes(d)              # an empty environment asks for the defaults
    monkeypatch.setenv("IMS_CHAIN_KERNELS", "0")
    monkeypatch.setenv("IMS_ACTIVE_FRACTION", "0.01")
    monkeypatch.setenv("IMS_UPD_DPP_MAX", "100000")
    tuning.sync_library(lib)
    got = tuning.Tuning()
    assert lib.ims_get_tuning(C.byref(got)) == 0
    assert (got.chain_kernels, got.active_fraction, got.upd_dpp_max, got.layout_kernels) == (0, 0.01, 100000, 1)
    bad = tuning.Tuning.from_buffer_copy(bytes(d))
    bad.active_fraction = 0.0
    assert lib.ims_set_tuning(C.byref(bad)) == -1 and b"tuning" in lib.ims_last_error()
    assert lib.ims_set_tuning(None) == -1
    assert lib.ims_get_tuning(C.byref(got)) == 0 and got.chain_kernels == 0          # a refused block changes nothing
    assert lib.ims_set_tuning(C.byref(d)) == 0
    tuning._LAST[0] = None
    with pytest.raises(KeyError):
        tuning.env("IMS_NO_SUCH_SWITCH")
    # the library's sources read the environment in one place only: the run-time lookup of hipFFT / RCCL by file name
    csrc = os.path.join(ROOT, "imsim_amd", "csrc")
    reads = {f: open(os.path.join(csrc, f)).read().count("getenv(") for f in os.listdir(csrc)}
    assert {f: n for f, n in reads.items() if n} == {"ims_libs.h": 1}
    # and the package's Python reads it in tuning.py only
    pkg = os.path.join(ROOT, "imsim_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py") and f != "tuning.py":
            assert "os.environ" not in open(os.path.join(pkg, f)).read(), f


def test_joint_run_entry_points_check_their_arguments_before_any_hip_call():
    """ims_plan_run_deferred / ims_plans_run_joint / ims_plan_join (the CCDs of a visit side by side): argument errors are
    reported without touching the GPU; an empty list of plans is a no-op."""
    lib = _abi.load()
    left = C.c_int32(7)
    assert lib.ims_plan_run_deferred(None, None, None, None, None, None, None, 0, 0, C.byref(left)) == -1
    assert b"plan" in lib.ims_last_error()
    assert lib.ims_plans_run_joint(None, 1, None, 0, 1) == -1 and b"plans" in lib.ims_last_error()
    one = (C.c_void_p * 1)(None)
    assert lib.ims_plans_run_joint(one, 1, None, 0, 1) == -1 and b"NULL entry" in lib.ims_last_error()
    assert lib.ims_plans_run_joint(one, 1, None, 0, 0) == -1 and b"chain range" in lib.ims_last_error()
    assert lib.ims_plans_run_joint(None, 0, None, 0, 1) == 0
    assert lib.ims_plan_join(None, None) == -1 and b"plan is NULL" in lib.ims_last_error()


def test_abi_version_and_error_string():
    lib = _abi.load()
    assert lib.ims_abi_version() == 22
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
    assert lib.ims_sensor_init_boundaries(None, None, 0, 1, None, 0, None) < 0
    S = _abi.Sensor()
    assert lib.ims_sensor_init_boundaries(C.byref(S), C.byref(S), 0, 1, None, 0, None) < 0 and b"slot" in lib.ims_last_error()
    assert lib.ims_flat_add(None, None, 1.0, 1.0, 0, 0, 8, 8, None, None, None) < 0 and b"image" in lib.ims_last_error()
    assert lib.ims_image_to_float(None, None, 8, None) < 0
    # a plan that names a stream it was not given
    it = _abi.PlanItem()
    it.kind, it.stream = _abi.IMS_PLAN_RECORD, 3
    streams = (C.c_void_p * 1)(None)
    assert lib.ims_run_plan(C.byref(it), 1, None, None, None, streams, 1) < 0 and b"stream" in lib.ims_last_error()
    # one round of a chain class / the rounds plan item
    P, ph = _abi.RenderParams(), _abi.Photons()
    assert lib.ims_accumulate_round(None, C.byref(ph), None, 0, 10000, 1, 4, None) < 0 and b"params" in lib.ims_last_error()
    assert lib.ims_accumulate_round(C.byref(P), C.byref(ph), None, 0, 10000, 1, 4, None) < 0
    it = _abi.PlanItem()
    it.kind, it.stream, it.n_slots, it.aux2 = _abi.IMS_PLAN_ROUNDS, 0, 1, None
    assert lib.ims_run_plan(C.byref(it), 1, None, None, None, streams, 1) < 0 and b"chains" in lib.ims_last_error()
    ch = (_abi.Chain * 1)()
    it.aux2 = C.addressof(ch)
    assert lib.ims_run_plan(C.byref(it), 1, None, None, None, streams, 1) < 0 and b"NULL pointer" in lib.ims_last_error()
    # lazy_static parameters (slot 0 without stored state) are for the fused render only: the pooled accumulates, which read
    # slot 0 for ordinary rows, refuse them before any launch
    P, ph = _abi.RenderParams(), _abi.Photons()
    P.seg_size, P.lazy_static, ph.converted, ph.n = 256, 1, 1, 4
    one = (C.c_double * 4)()
    P.image = C.cast(one, C.c_void_p)
    off = (C.c_int64 * 2)(0, 4)
    for entry, args in ((lib.ims_accumulate, (C.byref(P), off, C.byref(ph), None, None)),
                        (lib.ims_accumulate_segments, (C.byref(P), C.byref(ph), off, 4, None)),
                        (lib.ims_accumulate_small, (C.byref(P), C.byref(ph), off, 4, None))):
        P.n_objects, P.objects = (1, C.cast(one, C.c_void_p)) if entry is lib.ims_accumulate_small else (0, None)
        assert entry(*args) < 0 and b"lazy_static" in lib.ims_last_error(), entry
    # the derived-field helpers are pure host code
    op = _abi.Op()
    op.kind = _abi.IMS_OP_PHOTON_DCR
    op.p[0], op.p[1], op.p[2], op.p[3] = 620.0, 69.328, 293.15, 1.067
    assert lib.ims_fill_derived_op(C.byref(op)) == 0
    assert 1.0e-7 < op.p[5] < 1.0e-5 and 0.0 < op.p[6] < 1.0e-5 and 1.0e-4 < op.p[7] < 4.0e-4      # n - 1 ~ 1.9e-4 at 69 kPa


def test_derived_constants_are_the_plain_ieee_expressions():
    """ims_fill_derived_optics / _atmosphere / _sensor / _op (host code): every derived field equals the single IEEE
    operation the kernels would otherwise repeat per photon -- the same expression evaluated here in numpy float64."""
    lib = _abi.load()
    o = _abi.Optics()
    o.n_surfaces = 2
    s = o.surf[0]
    s.R, s.conic = 19.835, -1.215
    s.inv_R = 1.0 / s.R
    s.obsc_inner, s.obsc_outer = 2.558, 4.18
    s.asph[0], s.asph[1], s.asph[2] = 1.381e-9, -3.2e-13, 7.7e-17
    o.surf[1].R = 0.0
    for k, (a, b) in enumerate(zip((0.1, -0.7, 0.7), (0.6, 0.3, -0.74))):
        o.e_focal[k], o.e_z0[k] = a, b
    assert lib.ims_fill_derived_optics(C.byref(o)) == 0
    f = np.float64
    k1 = f(1.0) + f(s.conic)
    assert s.k1 == k1 and s.k1c == k1 * f(s.inv_R) and s.m2R == f(-2.0) * f(s.R) and s.cc == f(s.inv_R) * f(s.inv_R)
    assert s.obsc_i2 == f(2.558) * f(2.558) and s.obsc_o2 == f(4.18) * f(4.18)
    assert [s.asph_d[k] for k in range(4)] == [f(s.asph[k]) * f(k + 2) for k in range(4)]
    assert o.surf[1].k1 == 1.0 and o.surf[1].k1c == 0.0 and o.surf[1].m2R == 0.0
    ef, z0 = [f(v) for v in o.e_focal], [f(v) for v in o.e_z0]
    g = [ef[1] * z0[2] - ef[2] * z0[1], ef[2] * z0[0] - ef[0] * z0[2], ef[0] * z0[1] - ef[1] * z0[0]]
    assert [o.rot_g[k] for k in range(3)] == g and o.rot_gnorm == np.sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2])
    o.n_surfaces = 1000
    assert lib.ims_fill_derived_optics(C.byref(o)) < 0 and b"n_surfaces" in lib.ims_last_error()
    a = _abi.Atmosphere()
    a.npix, a.scale, a.aper_r_outer, a.aper_r_inner = 8192, 0.1, 4.18, 2.5498
    assert lib.ims_fill_derived_atmosphere(C.byref(a)) == 0
    assert a.dn == 8192.0 and a.inv_n == 1.0 / 8192.0 and a.inv_scale == f(1.0) / f(0.1)
    assert a.aper_ri2 == f(2.5498) * f(2.5498) and a.aper_dr2 == f(4.18) * f(4.18) - f(2.5498) * f(2.5498)
    a.npix = 0
    assert lib.ims_fill_derived_atmosphere(C.byref(a)) < 0
    sn = _abi.Sensor()
    sn.kind, sn.thickness, sn.pixel_size, sn.diff_step = _abi.IMS_SENSOR_SILICON, 100.0, 10.0, 4.9
    assert lib.ims_fill_derived_sensor(C.byref(sn)) == 0
    assert sn.diff_coef == f(4.9) / (f(100.0) * f(10.0)) and sn.thick_m1 == 99.0
    sn.thickness = 0.0
    assert lib.ims_fill_derived_sensor(C.byref(sn)) < 0
    op = _abi.Op()
    op.kind, op.p[0], op.p[1] = _abi.IMS_OP_PUPIL_ANNULUS_SAMPLER, 4.18, 2.55
    assert lib.ims_fill_derived_op(C.byref(op)) == 0
    assert op.p[2] == f(2.55) * f(2.55) and op.p[3] == f(4.18) * f(4.18) - f(2.55) * f(2.55)
    op = _abi.Op()
    op.kind, op.p[0] = _abi.IMS_OP_REFRACTION, 3.9
    assert lib.ims_fill_derived_op(C.byref(op)) == 0
    assert op.p[1] == f(3.9) * f(3.9) and op.p[2] == f(3.9) * f(3.9) - f(1.0)
    assert lib.ims_fill_derived_optics(None) < 0 and lib.ims_fill_derived_atmosphere(None) < 0 and lib.ims_fill_derived_sensor(None) < 0


def test_hand_written_dpp_instructions_have_no_hazard(tmp_path):
    """The v_fmac_f64_dpp broadcasts of the update kernel are inline asm, which the compiler's hazard recogniser does not
    look into: compile the device code to assembly and check the wait states before every DPP instruction."""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = tmp_path / "imsim_hip.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S",
                    os.path.join(root, "imsim_amd", "csrc", "imsim_hip.hip"), "-o", str(out)], check=True, capture_output=True)
    sys.path.insert(0, os.path.join(root, "tools"))
    import check_dpp_hazard
    n_dpp, bad = check_dpp_hazard.check(str(out))
    assert n_dpp >= 1000, "the DPP update kernel is gone?"
    assert not bad, bad[:5]


def test_the_ctypes_stub_of_integration_md_loads_the_library():
    """INTEGRATION.md section 2 is what a maintainer of the reference pastes: its version assertion and its mirror of the object
    row must hold against the library as built (the stub names the library by its bare file name; the test points it at the
    in-tree build)."""
    import re
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(.*?)```", text, re.S).group(1)
    code = block.split("# ... ims_render_params_t")[0]
    assert "ims_abi_version() ==" in code and "ims_struct_size(0)" in code
    code = code.replace('C.CDLL("libimsim_hip.so")', f'C.CDLL({_abi.lib_path()!r})')
    scope = {}
    exec(compile(code, "INTEGRATION.md", "exec"), scope)
    assert scope["lib"].ims_abi_version() == _abi.load().ims_abi_version()
    header = open(os.path.join(ROOT, "include", "imsim_hip.h")).read()
    assert f"#define IMS_ABI_VERSION {scope['lib'].ims_abi_version()}" in header
    assert C.sizeof(scope["ImsObject"]) == 256


def test_the_dpp_hazard_check_follows_labels_and_branches(tmp_path):
    """tools/check_dpp_hazard.py on hand-made assembly: a VALU write of the DPP source one instruction above a label is a
    hazard on the fall-through path, so is one in front of a branch that names the label; enough s_nop clears both."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_dpp_hazard

    def run(text):
        f = tmp_path / "k.s"
        f.write_text(text)
        return check_dpp_hazard.check(str(f))
    fall = "\tv_mov_b32 v2, v9\n.LBB0_1:\n\tv_fmac_f64_dpp v[4:5], v[2:3], v[6:7] row_newbcast:1\n"
    assert len(run(fall)[1]) == 1
    assert len(run("\tv_mov_b32 v2, v9\n\ts_nop 1\n.LBB0_1:\n\tv_fmac_f64_dpp v[4:5], v[2:3], v[6:7] row_newbcast:1\n")[1]) == 0
    jump = ("\tv_mov_b32 v3, v9\n\ts_cbranch_vccnz .LBB0_2\n\ts_nop 4\n\ts_branch .LBB0_3\n.LBB0_2:\n"
            "\tv_fmac_f64_dpp v[4:5], v[2:3], v[6:7] row_newbcast:1\n.LBB0_3:\n\ts_endpgm\n")
    n, bad = run(jump)
    assert n == 1 and len(bad) == 1 and "2 wait states" in bad[0][4]
    assert len(run(jump.replace("\ts_cbranch_vccnz", "\ts_nop 0\n\ts_cbranch_vccnz"))[1]) == 0
    assert len(run("\tv_cmpx_gt_u32 v1, v0\n\ts_nop 2\n.LBB0_1:\n\tv_fmac_f64_dpp v[4:5], v[2:3], v[6:7] row_newbcast:1\n")[1]) == 1
