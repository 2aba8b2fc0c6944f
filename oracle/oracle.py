"""TEST INFRASTRUCTURE -- ctypes front end of oracle/wbc_oracle.c (the scalar CPU restatement of
the widowGo1 rollout step). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; the product path (wbc_amd) never does.

`OracleSim(precision="f64"|"f32")` exposes get/set by tensor name with the same names, shapes and
semantics as the device tensors of include/wbc_sim.h, so a parity test reads
    ora.set("DOF_STATE", x); ora.step(a); np.testing.assert_allclose(sim.tensor("DOF_STATE"), ora.get("DOF_STATE"))
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "deep-whole-body-control_amd"))
from wbc_amd import abi  # noqa: E402  (struct definitions only)


def build(force: bool = False) -> None:
    """Compile both precisions of the oracle with gcc (oracle/Makefile)."""
    need = force or any(
        not os.path.exists(os.path.join(HERE, f"libwbc_oracle_{p}.so")) or
        os.path.getmtime(os.path.join(HERE, f"libwbc_oracle_{p}.so")) < os.path.getmtime(os.path.join(HERE, "wbc_oracle.c"))
        for p in ("f64", "f32"))
    if need:
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if force else []))


_LIBS = {}


def _lib(precision: str):
    if precision not in _LIBS:
        path = os.path.join(HERE, f"libwbc_oracle_{precision}.so")
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        lib.ora_create.restype = C.c_void_p
        lib.ora_create.argtypes = [C.POINTER(abi.WbcModel), C.POINTER(abi.WbcTaskCfg), C.c_int, C.c_uint64]
        lib.ora_destroy.argtypes = [C.c_void_p]
        lib.ora_set_curriculum.argtypes = [C.c_void_p, C.POINTER(abi.WbcCurriculum)]
        lib.ora_set_step_counter.argtypes = [C.c_void_p, C.c_int64]
        lib.ora_get_step_counter.argtypes = [C.c_void_p]
        lib.ora_get_step_counter.restype = C.c_int64
        lib.ora_set_heightfield.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_double] * 5
        lib.ora_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.ora_set.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.ora_field_size.argtypes = [C.c_int]
        lib.ora_step.argtypes = [C.c_void_p, C.c_void_p]
        for fn in ("ora_reset_all", "ora_simulate", "ora_refresh_rigid_body_state", "ora_compute_torques"):
            getattr(lib, fn).argtypes = [C.c_void_p]
        lib.ora_debug_aba.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ora_debug_contacts.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        _LIBS[precision] = lib
    return _LIBS[precision]


class OracleSim:
    def __init__(self, model: abi.WbcModel, cfg: abi.WbcTaskCfg, num_envs: int, seed: int = 1,
                 precision: str = "f64"):
        self.lib = _lib(precision)
        self.n = num_envs
        self.model, self.cfg = model, cfg
        self.h = self.lib.ora_create(C.byref(model), C.byref(cfg), num_envs, seed)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ora_destroy(self.h)
            self.h = None

    @property
    def threads(self) -> int:
        """OpenMP threads `step` spreads the envs over."""
        return int(self.lib.ora_threads())

    def set_curriculum(self, cur: abi.WbcCurriculum):
        self.lib.ora_set_curriculum(self.h, C.byref(cur))

    @property
    def step_counter(self) -> int:
        return self.lib.ora_get_step_counter(self.h)

    @step_counter.setter
    def step_counter(self, v: int):
        self.lib.ora_set_step_counter(self.h, int(v))

    def get(self, name: str) -> np.ndarray:
        shape = (self.n,) + abi.TENSOR_SHAPES[name]
        out = np.zeros(shape, dtype=np.float64)
        rc = self.lib.ora_get(self.h, abi.T[name], out.ctypes.data)
        assert rc == 0, name
        return out

    def set(self, name: str, value) -> None:
        shape = (self.n,) + abi.TENSOR_SHAPES[name]
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=np.float64), shape))
        rc = self.lib.ora_set(self.h, abi.T[name], v.ctypes.data)
        assert rc == 0, name

    def set_env_params(self, friction, base_dmass, base_dcom, gripper_dmass, motor_strength, env_origins,
                       box_delta_y, traj_timesteps, traj_total_timesteps, robot_model, box_dmass=None) -> None:
        """Same inputs as wbc_sim_set_env_params; the composite inertias are computed by the shared
        host helper abi.body_params_from_randomisation."""
        n = self.n
        self.set("FRICTION", np.asarray(friction).reshape(n))
        mp = np.concatenate([np.asarray(base_dmass).reshape(n, 1), np.asarray(base_dcom).reshape(n, 3),
                             np.asarray(gripper_dmass).reshape(n, 1)], axis=1)
        self.set("MASS_PARAMS", mp)
        self.set("MOTOR_STRENGTH", np.asarray(motor_strength).reshape(n, 18))
        self.set("ENV_ORIGINS", np.asarray(env_origins).reshape(n, 3))
        self.set("BOX_DELTA_Y", np.asarray(box_delta_y).reshape(n))
        self.set("BOX_MASS", float(self.model.box_mass) + (0.0 if box_dmass is None else np.asarray(box_dmass, dtype=np.float64).reshape(n)))
        self.set("BODY_PARAMS", abi.body_params_from_randomisation(robot_model, base_dmass, base_dcom, gripper_dmass))
        g = self.get("GOAL_STATE")
        g[:, 22] = np.asarray(traj_timesteps).reshape(n)
        g[:, 23] = np.asarray(traj_total_timesteps).reshape(n)
        self.set("GOAL_STATE", g)

    def set_heightfield(self, heights, hscale, vscale, tx, ty, tz):
        if heights is None:
            self.lib.ora_set_heightfield(self.h, None, 0, 0, 0, 0, 0, 0, 0)
            return
        h = np.ascontiguousarray(heights, dtype=np.int16)
        self.lib.ora_set_heightfield(self.h, h.ctypes.data, h.shape[0], h.shape[1], hscale, vscale, tx, ty, tz)

    def step(self, actions) -> None:
        a = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.n, 18)
        self.lib.ora_step(self.h, a.ctypes.data)

    def reset_all(self):
        self.lib.ora_reset_all(self.h)

    def simulate(self):
        self.lib.ora_simulate(self.h)

    def compute_torques(self):
        self.lib.ora_compute_torques(self.h)

    def refresh_rigid_body_state(self):
        self.lib.ora_refresh_rigid_body_state(self.h)

    def debug_contacts(self, env: int = 0):
        """Contact list of one substep run on a copy of env's state: active, nshare, impulses, normals, points (frame F)."""
        act = np.zeros(abi.NCP, dtype=np.int32); nsh = np.zeros(abi.NCP, dtype=np.int32)
        lam, n, xc = np.zeros((abi.NCP, 3)), np.zeros((abi.NCP, 3)), np.zeros((abi.NCP, 3))
        self.lib.ora_debug_contacts(self.h, env, act.ctypes.data, nsh.ctypes.data, lam.ctypes.data, n.ctypes.data, xc.ctypes.data)
        return dict(active=act.astype(bool), nshare=nsh, lam=lam, n=n, xc=xc)

    def debug_aba(self, env: int = 0):
        qdd = np.zeros(20)
        a0 = np.zeros(6)
        self.lib.ora_debug_aba(self.h, env, qdd.ctypes.data, a0.ctypes.data, None)
        return qdd, a0


def default_curriculum(cfg, update_counter: int = 1) -> abi.WbcCurriculum:
    """update_command_curriculum (WG:675-692) evaluated at `update_counter` (host-side table builder
    shared with the product: it only fills a struct, it computes nothing on the path)."""
    from wbc_amd.curriculum import make_curriculum
    return make_curriculum(cfg, update_counter)
