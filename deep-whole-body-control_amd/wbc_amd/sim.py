"""`WbcSim`: the torch-facing handle of one `wbc_sim` (include/wbc_sim.h). torch provides the
device arena, the stream and zero-copy tensor views; every computation is a kernel in libwbc_amd.so.

This is the layer that stands where `isaacgym.gymapi` + `gymtorch.wrap_tensor` stand in the
reference (legged_gym/envs/widowGo1/widowGo1.py:505-551)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import abi
from .native import check, lib

_TORCH_DT = {"f32": torch.float32, "i64": torch.int64, "u8": torch.uint8}


class SideJob:
    """A wbc_side_job plus the tensors it points into (kept alive until it has run)."""

    def __init__(self, L, keep):
        self.L, self.keep, self.c, self.done = L, keep, abi.WbcSideJob(), False

    def run(self, stream: int) -> None:
        """Stand-alone execution (nobody carried it)."""
        if not self.done:
            check(self.L.wbc_side_job_run(C.byref(self.c), stream), "wbc_side_job_run")
            self.done = True


class WbcSim:
    def __init__(self, model: abi.WbcModel, cfg: abi.WbcTaskCfg, num_envs: int, device: torch.device, seed: int = 1):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("WbcSim needs a ROCm device: the rollout step only exists as HIP kernels")
        if device.index is not None and device.index != torch.cuda.current_device():
            raise RuntimeError(f"WbcSim on {device} while torch's current device is cuda:{torch.cuda.current_device()}: call "
                               "torch.cuda.set_device(device) first (one process per GPU; the C-ABI launches on the caller's "
                               "current stream and does not switch devices behind torch's back)")
        self.L = lib()
        self.device = device
        self.num_envs = num_envs
        self.model, self.cfg = model, cfg
        nbytes = self.L.wbc_sim_arena_bytes(num_envs)
        self.arena = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        torch.cuda.synchronize(device)
        h = C.c_void_p()
        dev_index = device.index if device.index is not None else torch.cuda.current_device()
        check(self.L.wbc_sim_create(C.byref(model), C.byref(cfg), num_envs, dev_index, seed,
                                    self.arena.data_ptr(), nbytes, C.byref(h)), "wbc_sim_create")
        self.h = h
        self._views: Dict[str, torch.Tensor] = {}
        base = self.arena.data_ptr()
        for name in abi.TENSOR_IDS:
            p, shape, nd, dt = C.c_void_p(), (C.c_int64 * 4)(), C.c_int(), C.c_int()
            check(self.L.wbc_sim_get_tensor(self.h, abi.T[name], C.byref(p), shape, C.byref(nd), C.byref(dt)), "wbc_sim_get_tensor")
            dims = [int(shape[i]) for i in range(nd.value)]
            tdt = _TORCH_DT[["f32", "i64", "u8"][dt.value]]
            off = p.value - base
            nb = int(np.prod(dims)) * torch.empty((), dtype=tdt).element_size()
            self._views[name] = self.arena[off:off + nb].view(tdt).view(dims)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            torch.cuda.synchronize(self.device)
            self.L.wbc_sim_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- tensors ---------------------------------------------------------------------------
    def tensor(self, name: str) -> torch.Tensor:
        """Zero-copy view of a device tensor (names: abi.TENSOR_IDS)."""
        return self._views[name]

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    # ---- setup -----------------------------------------------------------------------------
    def set_env_params(self, friction=None, base_dmass=None, base_dcom=None, gripper_dmass=None, motor_strength=None,
                       env_origins=None, box_delta_y=None, traj_timesteps=None, traj_total_timesteps=None, box_dmass=None):
        n = self.num_envs
        keep = []

        def ptr(x, cols):
            if x is None:
                return None
            a = np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(n * cols))
            keep.append(a)
            return a.ctypes.data
        torch.cuda.synchronize(self.device)
        check(self.L.wbc_sim_set_env_params(self.h, ptr(friction, 1), ptr(base_dmass, 1), ptr(base_dcom, 3), ptr(gripper_dmass, 1),
                                            ptr(motor_strength, 18), ptr(env_origins, 3), ptr(box_delta_y, 1),
                                            ptr(traj_timesteps, 1), ptr(traj_total_timesteps, 1), ptr(box_dmass, 1)),
              "wbc_sim_set_env_params")

    def set_curriculum(self, cur: abi.WbcCurriculum):
        check(self.L.wbc_sim_set_curriculum(self.h, C.byref(cur)), "wbc_sim_set_curriculum")

    def set_heightfield(self, heights: Optional[np.ndarray], hscale=0.0, vscale=0.0, tx=0.0, ty=0.0, tz=0.0):
        if heights is None:
            check(self.L.wbc_sim_set_heightfield(self.h, None, 0, 0, 0, 0, 0, 0, 0), "wbc_sim_set_heightfield")
            return
        h = np.ascontiguousarray(heights, dtype=np.int16)
        check(self.L.wbc_sim_set_heightfield(self.h, h.ctypes.data, h.shape[0], h.shape[1], hscale, vscale, tx, ty, tz),
              "wbc_sim_set_heightfield")

    @property
    def step_counter(self) -> int:
        v = C.c_int64()
        check(self.L.wbc_sim_get_step_counter(self.h, C.byref(v)))
        return v.value

    @step_counter.setter
    def step_counter(self, v: int):
        check(self.L.wbc_sim_set_step_counter(self.h, int(v)))

    # ---- stepping --------------------------------------------------------------------------
    def step(self, actions: torch.Tensor, obs_out: torch.Tensor = None, store=None) -> None:
        """One env step. obs_out (f32 [N, 860], contiguous, same device): write the observations there instead of OBS_BUF.
        store = (values f32 [N,2], gamma, out_rewards f32 [N,2], out_dones u8 [N,1] or [N]): also write this transition's
        rollout-storage reward (with the time-out bootstrap) and done slots (wbc_sim_step_rollout)."""
        assert actions.is_cuda and actions.dtype == torch.float32 and actions.is_contiguous()
        assert actions.shape == (self.num_envs, abi.NACT)
        if obs_out is not None:
            assert (obs_out.is_cuda and obs_out.device == actions.device and obs_out.dtype == torch.float32 and obs_out.is_contiguous()
                    and obs_out.shape == (self.num_envs, abi.NOBS))
        if obs_out is None and store is None:
            check(self.L.wbc_sim_step(self.h, actions.data_ptr(), self._stream()), "wbc_sim_step")
        elif store is None:
            check(self.L.wbc_sim_step_to(self.h, actions.data_ptr(), obs_out.data_ptr(), self._stream()), "wbc_sim_step_to")
        else:
            values, gamma, rewards, dones = store
            n = self.num_envs
            assert values.is_cuda and values.dtype == torch.float32 and values.is_contiguous() and values.shape == (n, 2)
            assert rewards.is_cuda and rewards.dtype == torch.float32 and rewards.is_contiguous() and rewards.shape == (n, 2)
            assert dones.is_cuda and dones.dtype == torch.uint8 and dones.is_contiguous() and dones.numel() == n
            check(self.L.wbc_sim_step_rollout(self.h, actions.data_ptr(), obs_out.data_ptr() if obs_out is not None else None,
                                              values.data_ptr(), float(gamma), rewards.data_ptr(), dones.data_ptr(), self._stream()),
                  "wbc_sim_step_rollout")

    def reset_all(self) -> None:
        check(self.L.wbc_sim_reset_all(self.h, self._stream()), "wbc_sim_reset_all")

    def arm_dynamics(self, link_rb9, link_mass9):
        """(mm [N,6,6], ee Jacobian [N,6,6], gravity torques [N,6]) of the arm from the current state: what the
        reference reads from Isaac Gym's mass-matrix / Jacobian tensors (WG:550-558, 1201-1207)."""
        n = self.num_envs
        mm = torch.empty(n, 6, 6, dtype=torch.float32, device=self.device)
        jac = torch.empty(n, 6, 6, dtype=torch.float32, device=self.device)
        gt = torch.empty(n, 6, dtype=torch.float32, device=self.device)
        rb = (C.c_int * 9)(*[int(x) for x in link_rb9])
        ms = (C.c_float * 9)(*[float(x) for x in link_mass9])
        check(self.L.wbc_sim_arm_dynamics(self.h, rb, ms, mm.data_ptr(), jac.data_ptr(), gt.data_ptr(), self._stream()), "wbc_sim_arm_dynamics")
        return mm, jac, gt

    def episode_stats(self, scale: float, track_state: torch.Tensor = None, track_cap: int = 0) -> torch.Tensor:
        """Means over the envs that reset in the last step of their finished episode's reward sums [NREW] and metric
        sums [NMETRIC], times `scale`, as one fresh device tensor (WG:743-754 without a host sync). With `track_state`
        (wbc_runner_track_state_floats(n, cap) floats) the launch also advances the runner's episode deques (OPR:140-154)."""
        out = torch.empty(abi.NREW + abi.NMETRIC, dtype=torch.float32, device=self.device)
        prev = self.__dict__.get("_last_episode_stats")         # a step without resets re-publishes the previous values (WG:705-706)
        check(self.L.wbc_sim_episode_stats_track(self.h, float(scale), prev.data_ptr() if prev is not None else None, out.data_ptr(),
                                                 track_state.data_ptr() if track_state is not None else None, int(track_cap),
                                                 self._stream()), "wbc_sim_episode_stats")
        self._last_episode_stats = out
        return out

    def episode_stats_job(self, scale: float, track_state: torch.Tensor = None, track_cap: int = 0):
        """episode_stats without the launch: (out tensor, SideJob). Whoever takes the job executes it -- as extra workgroups of the
        policy inference that follows (ActorCritic.fused_act(side_job=...)) or stand-alone (SideJob.run) -- before the next step."""
        out = torch.empty(abi.NREW + abi.NMETRIC, dtype=torch.float32, device=self.device)
        prev = self.__dict__.get("_last_episode_stats")
        job = SideJob(self.L, (out, prev, track_state))
        check(self.L.wbc_sim_episode_stats_job(self.h, float(scale), prev.data_ptr() if prev is not None else None, out.data_ptr(),
                                               track_state.data_ptr() if track_state is not None else None, int(track_cap),
                                               C.byref(job.c)), "wbc_sim_episode_stats_job")
        self._last_episode_stats = out
        return out, job

    def set_dof_forces(self, torques: torch.Tensor) -> None:
        assert torques.is_cuda and torques.dtype == torch.float32 and torques.is_contiguous()
        check(self.L.wbc_sim_set_dof_forces(self.h, torques.data_ptr(), self._stream()), "wbc_sim_set_dof_forces")

    def simulate(self) -> None:
        check(self.L.wbc_sim_simulate(self.h, self._stream()), "wbc_sim_simulate")

    def set_root_state(self, root: torch.Tensor) -> None:
        check(self.L.wbc_sim_set_root_state(self.h, root.data_ptr(), self._stream()), "wbc_sim_set_root_state")

    def set_dof_state(self, dof: torch.Tensor) -> None:
        check(self.L.wbc_sim_set_dof_state(self.h, dof.data_ptr(), self._stream()), "wbc_sim_set_dof_state")

    def set_root_state_indexed(self, root: torch.Tensor, env_ids: torch.Tensor) -> None:
        ids = env_ids.to(torch.int32).contiguous()
        check(self.L.wbc_sim_set_root_state_indexed(self.h, root.data_ptr(), ids.data_ptr(), ids.numel(), self._stream()))

    def set_dof_state_indexed(self, dof: torch.Tensor, env_ids: torch.Tensor) -> None:
        ids = env_ids.to(torch.int32).contiguous()
        check(self.L.wbc_sim_set_dof_state_indexed(self.h, dof.data_ptr(), ids.data_ptr(), ids.numel(), self._stream()))

    def refresh_rigid_body_state(self) -> None:
        check(self.L.wbc_sim_refresh_rigid_body_state(self.h, self._stream()), "wbc_sim_refresh_rigid_body_state")
