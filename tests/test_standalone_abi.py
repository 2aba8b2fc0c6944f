"""The boundary is self-contained: a binding that imports NOTHING of this repository's Python -- only libwbc_amd.so through raw
ctypes, the header's struct layouts (it never needs them here: the structs stay opaque pointers) and the packaged asset file --
loads the asset (what gym.load_asset + the get_asset_* getters + the config resolution hand to the reference, widowGo1.py:268-294,
78-121), creates a sim, reads its tensors and steps it. INTEGRATION.md section 2 is this file in prose."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "deep-whole-body-control_amd", "wbc_amd", "libwbc_amd.so")
ASSET = os.path.join(ROOT, "deep-whole-body-control_amd", "wbc_amd", "assets", "widowgo1_default.wbcasset")


def _bind():
    L = C.CDLL(LIB)
    L.wbc_last_error.restype = C.c_char_p
    L.wbc_asset_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    L.wbc_asset_free.argtypes = [C.c_void_p]
    L.wbc_asset_free.restype = None
    for fn in ("wbc_asset_dof_count", "wbc_asset_rigid_body_count"):
        getattr(L, fn).argtypes = [C.c_void_p]
    for fn in ("wbc_asset_dof_name", "wbc_asset_rigid_body_name"):
        getattr(L, fn).argtypes = [C.c_void_p, C.c_int]
        getattr(L, fn).restype = C.c_char_p
    L.wbc_asset_dof_properties.argtypes = [C.c_void_p] * 5
    for fn in ("wbc_asset_model", "wbc_asset_task_cfg"):
        getattr(L, fn).argtypes = [C.c_void_p]
        getattr(L, fn).restype = C.c_void_p
    L.wbc_asset_curriculum.argtypes = [C.c_void_p, C.c_int]
    L.wbc_asset_curriculum.restype = C.c_void_p
    L.wbc_sim_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.wbc_sim_destroy.argtypes = [C.c_void_p]
    L.wbc_sim_set_curriculum.argtypes = [C.c_void_p, C.c_void_p]
    L.wbc_sim_reset_all.argtypes = [C.c_void_p, C.c_void_p]
    L.wbc_sim_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.wbc_sim_get_tensor.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return L


def test_asset_file_loads_through_raw_ctypes_and_is_current():
    L = _bind()
    a = C.c_void_p()
    assert L.wbc_asset_load(ASSET.encode(), C.byref(a)) == 0, L.wbc_last_error()
    assert L.wbc_asset_dof_count(a) == 20 and L.wbc_asset_rigid_body_count(a) == 27
    dofs = [L.wbc_asset_dof_name(a, i).decode() for i in range(20)]
    bodies = [L.wbc_asset_rigid_body_name(a, i).decode() for i in range(27)]
    assert dofs[:3] == ["FL_hip_joint", "FL_thigh_joint", "FL_calf_joint"] and dofs[12] == "widow_waist" and dofs[-1] == "widow_right_finger"   # quirk Q1
    assert bodies[0] == "base" and bodies[1] == "trunk" and bodies[-3] == "wx250s/ee_gripper_link"
    lo, hi, vel, eff = ((C.c_float * 20)() for _ in range(4))
    assert L.wbc_asset_dof_properties(a, lo, hi, vel, eff) == 0
    np.testing.assert_allclose(list(eff)[:12], 23.7)                                       # URDF effort limits (LR:294-299)
    np.testing.assert_allclose([lo[1], hi[1]], [-0.663, 2.967], atol=1e-3)
    assert L.wbc_asset_curriculum(a, 2) is None and L.wbc_asset_dof_name(a, 20) is None
    L.wbc_asset_free(a)
    bad = C.c_void_p()
    assert L.wbc_asset_load(b"/nonexistent.wbcasset", C.byref(bad)) == -2 and b"cannot open" in L.wbc_last_error()
    # the packaged file equals what the Python host path builds from the URDF tables and the shipped config (tools/make_asset.py)
    sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd"))
    from wbc_amd import abi
    from wbc_amd.config import WidowGo1RoughCfg
    assert open(ASSET, "rb").read() == abi.asset_bytes(abi.load_default_model(), WidowGo1RoughCfg()), "stale asset: run python tools/make_asset.py"


_DRIVER = r'''
import ctypes as C, sys, numpy as np, torch
assert not any(m == "wbc_amd" or m.startswith("wbc_amd.") for m in sys.modules)
sys.path.insert(0, sys.argv[3])
import test_standalone_abi as T
L = T._bind()
a = C.c_void_p(); assert L.wbc_asset_load(sys.argv[2].encode(), C.byref(a)) == 0
n = 256
sim = C.c_void_p()
assert L.wbc_sim_create(L.wbc_asset_model(a), L.wbc_asset_task_cfg(a), n, 0, 7, None, 0, C.byref(sim)) == 0, L.wbc_last_error()
assert L.wbc_sim_set_curriculum(sim, L.wbc_asset_curriculum(a, 1)) == 0
stream = torch.cuda.current_stream().cuda_stream
assert L.wbc_sim_reset_all(sim, stream) == 0
def tensor(tid):
    ptr, shape, nd, dt = C.c_void_p(), (C.c_int64 * 4)(), C.c_int(), C.c_int()
    assert L.wbc_sim_get_tensor(sim, tid, C.byref(ptr), shape, C.byref(nd), C.byref(dt)) == 0
    return ptr.value, tuple(shape[i] for i in range(nd.value)), dt.value
OBS, ROOT, RESET = 6, 0, 17                                  # enum wbc_tensor_id (include/wbc_sim.h)
ptr, shape, dt = tensor(OBS)
assert shape == (n, 860) and dt == 0
obs = torch.empty(n, 860, device="cuda")
actions = torch.zeros(n, 18, device="cuda")
import ctypes
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
z = []
for step in range(30):
    assert L.wbc_sim_step(sim, actions.data_ptr(), stream) == 0, L.wbc_last_error()
    torch.cuda.synchronize()
    assert hip.hipMemcpy(obs.data_ptr(), ptr, n * 860 * 4, 3) == 0
    rp, rs, _ = tensor(ROOT)
    root = torch.empty(n, 2, 13, device="cuda")
    assert hip.hipMemcpy(root.data_ptr(), rp, n * 26 * 4, 3) == 0
    z.append(float(root[:, 0, 2].mean()))
assert torch.isfinite(obs).all() and obs.abs().max() <= 100.0
# zero actions: the PD law holds the default stance; the robots drop from the 0.42 m spawn height and stand (resets re-drop some)
assert 0.25 < z[-1] < 0.45 and z[0] > z[5], z
assert L.wbc_sim_destroy(sim) == 0
L.wbc_asset_free(a)
print("STANDALONE_OK", round(z[-1], 3))
'''


@pytest.mark.gpu
def test_create_get_tensor_step_without_this_repositorys_python():
    """A fresh interpreter that never imports wbc_amd: asset -> wbc_sim_create -> wbc_sim_get_tensor -> wbc_sim_step x 30 with raw
    ctypes (torch only lends device memory and the stream)."""
    p = subprocess.run([sys.executable, "-c", _DRIVER, "x", ASSET, os.path.dirname(os.path.abspath(__file__))], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "STANDALONE_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
