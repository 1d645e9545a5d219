"""CPU: the C-ABI library loads and exports every symbol include/gldm.h declares
(no compute calls without a GPU), and the oracle's C library builds."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "gldm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gldm_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    names = _declared()
    assert "gldm_ball_query" in names and "gldm_sa_group" in names and len(names) >= 11


def test_library_exports_every_declared_symbol():
    from graspldm_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    h = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(h, n)]
    assert not missing, f"declared in gldm.h but not exported: {missing}"
    lib = _lib.lib()
    assert lib.gldm_abi_version() == _lib.ABI_VERSION
    assert lib.gldm_status_string(0) == b"ok"


def test_python_binding_covers_every_declared_symbol():
    from graspldm_amd import _lib
    bound = set(_lib._SIGNATURES) | {"gldm_abi_version", "gldm_status_string", "gldm_r1d_workspace_bytes", "gldm_r1d_tile_columns", "gldm_conv3d_partial_floats", "gldm_squeeze_parts"}
    assert set(_declared()) <= bound, sorted(set(_declared()) - bound)


def test_invalid_arguments_return_status_not_exit():
    from graspldm_amd import _lib
    h = _lib.lib()
    # null pointers / non-positive sizes are rejected before any HIP call
    assert h.gldm_ball_query(None, None, 1, 1, 1, 0.1, 1, None, None) == -1
    assert h.gldm_grouping_forward(None, None, 0, 0, 0, 0, 0, None, None) == -1
    with pytest.raises(_lib.GldmError):
        _lib.call("gldm_furthest_point_sampling", None, 1, 16, 4, None, None)


def test_backend_has_the_twelve_reference_names():
    from graspldm_amd.backend import _backend
    names = ["gather_features_forward", "gather_features_backward", "furthest_point_sampling", "ball_query",
             "grouping_forward", "grouping_backward", "three_nearest_neighbors_interpolate_forward",
             "three_nearest_neighbors_interpolate_backward", "trilinear_devoxelize_forward",
             "trilinear_devoxelize_backward", "avg_voxelize_forward", "avg_voxelize_backward"]
    for n in names:
        assert callable(getattr(_backend, n)), n
    with pytest.raises(NotImplementedError):
        _backend.grouping_backward(None, None, 0)


def test_backend_rejects_cpu_tensors_like_the_reference():
    import torch
    from graspldm_amd.backend import _backend
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        _backend.ball_query(torch.zeros(1, 3, 4), torch.zeros(1, 3, 8), 0.1, 2)


def test_dpp_hazard_scan_flags_a_close_write(tmp_path):
    """tools/isa/dpp_hazard_scan.py (part of tools/isa/lint.sh): a VALU write one wait state in front of a DPP read of the
    same register is reported, the same pair behind an `s_nop 1` is not."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = tmp_path / "bad.s"
    bad.write_text("0000000000001000 <my_kernel>:\n"
                   "\tv_add_f32_e32 v5, v1, v2                                 // 000000001000: 020A0501\n"
                   "\tv_max_f32_dpp v6, v5, v5 row_ror:4 row_mask:0xf bank_mask:0xf // 000000001004: 160C0AFA\n"
                   "\ts_endpgm\n")
    good = tmp_path / "good.s"
    good.write_text("0000000000001000 <my_kernel>:\n"
                    "\tv_add_f32_e32 v5, v1, v2\n\ts_nop 1\n"
                    "\tv_max_f32_dpp v6, v5, v5 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                    "\tv_mov_b32_e32 v9, v1\n\tv_mov_b32_e32 v10, v1\n\tv_mov_b32_e32 v11, v1\n"
                    "\tv_mov_b32_dpp v7, v9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_endpgm\n")
    scan = os.path.join(root, "tools", "isa", "dpp_hazard_scan.py")
    r = subprocess.run([sys.executable, scan, str(bad), "my_kernel"], capture_output=True, text=True)
    assert r.returncode == 1 and "1 VALU-write" in r.stdout, r.stdout
    r = subprocess.run([sys.executable, scan, str(good), "my_kernel"], capture_output=True, text=True)
    assert r.returncode == 0 and "2 DPP instructions, 0 VALU-write" in r.stdout, r.stdout


def test_hazard_scan_flags_a_read_of_a_fresh_matrix_result(tmp_path):
    """Second check of tools/isa/dpp_hazard_scan.py (round 6): a VALU instruction that reads a matrix instruction's result
    closer than the compiler ever leaves them -- only an asm statement taking accumulators straight out of the matrix pipe can
    do that (csrc/quad_narrow.h: pos_max8 did, 3 wait states behind a v_mfma_f32_16x16x4_f32) -- is reported; the same read
    behind enough other instructions, or of a register overwritten in between, is not."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scan = os.path.join(root, "tools", "isa", "dpp_hazard_scan.py")
    mfma = "\tv_mfma_f32_16x16x4_f32 v[60:63], v5, v69, 0\n"
    filler = "\tv_mov_b32_e32 v9, v1\n"
    read = "\tv_max_f32_dpp v3, v60, v60 row_ror:4 row_mask:0xf bank_mask:0xf\n"
    head, tail = "0000000000001000 <my_kernel>:\n", "\ts_endpgm\n"
    bad = tmp_path / "bad.s"
    bad.write_text(head + mfma + 3 * filler + read + tail)
    far = tmp_path / "far.s"
    far.write_text(head + mfma + 3 * filler + "\ts_nop 7\n" + read + tail)
    over = tmp_path / "over.s"
    over.write_text(head + mfma + "\tv_mov_b32_e32 v60, v1\n" + 2 * filler + read + tail)
    r = subprocess.run([sys.executable, scan, str(bad), "my_kernel"], capture_output=True, text=True)
    assert r.returncode == 1 and "1 VALU reads of a matrix" in r.stdout, r.stdout
    for f in (far, over):
        r = subprocess.run([sys.executable, scan, str(f), "my_kernel"], capture_output=True, text=True)
        assert r.returncode == 0 and "0 VALU reads of a matrix" in r.stdout, (f, r.stdout)
