"""Loader of libwbc_amd.so (the HIP extension, built in-tree by __graft_entry__.build()).
There is no CPU fallback: if the library is missing or a call fails, this raises."""
from __future__ import annotations

import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WBC_AMD_LIB") or os.path.join(_HERE, "libwbc_amd.so")   # override: kernel-variant A/B runs
_lib = None

c_void, c_int, c_f32p = C.c_void_p, C.c_int, C.c_void_p


class WbcError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        # torch ships its own libamdhip64; it must be in the process before this library resolves its HIP
        # dependency, or two HIP runtimes end up loaded and the second one finds no device
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise WbcError(f"{LIB_PATH} not found: build the HIP extension first "
                           f"(python -c 'import __graft_entry__ as g; g.build()')")
        L = C.CDLL(LIB_PATH)
        L.wbc_last_error.restype = C.c_char_p
        L.wbc_sim_arena_bytes.restype = C.c_size_t
        L.wbc_sim_arena_bytes.argtypes = [c_int]
        L.wbc_sim_create.argtypes = [C.POINTER(abi.WbcModel), C.POINTER(abi.WbcTaskCfg), c_int, c_int, C.c_uint64,
                                     c_void, C.c_size_t, C.POINTER(c_void)]
        L.wbc_sim_destroy.argtypes = [c_void]
        L.wbc_sim_get_tensor.argtypes = [c_void, c_int, C.POINTER(c_void), C.POINTER(C.c_int64), C.POINTER(c_int), C.POINTER(c_int)]
        L.wbc_sim_set_env_params.argtypes = [c_void] + [c_void] * 10
        L.wbc_sim_set_heightfield.argtypes = [c_void, c_void, c_int, c_int] + [C.c_float] * 5
        L.wbc_sim_set_curriculum.argtypes = [c_void, C.POINTER(abi.WbcCurriculum)]
        L.wbc_sim_step.argtypes = [c_void, c_void, c_void]
        L.wbc_sim_step_to.argtypes = [c_void, c_void, c_void, c_void]
        L.wbc_sim_step_rollout.argtypes = [c_void, c_void, c_void, c_void, C.c_float, c_void, c_void, c_void]
        L.wbc_get_heights.argtypes = [c_void, c_int, c_void, c_int, c_void, c_void, c_int, c_int, C.c_float, C.c_float, C.c_float, c_void, c_int, c_int, c_void]
        L.wbc_sim_reset_all.argtypes = [c_void, c_void]
        L.wbc_sim_set_dof_forces.argtypes = [c_void, c_void, c_void]
        L.wbc_sim_simulate.argtypes = [c_void, c_void]
        L.wbc_sim_set_root_state.argtypes = [c_void, c_void, c_void]
        L.wbc_sim_set_dof_state.argtypes = [c_void, c_void, c_void]
        L.wbc_sim_set_root_state_indexed.argtypes = [c_void, c_void, c_void, c_int, c_void]
        L.wbc_sim_set_dof_state_indexed.argtypes = [c_void, c_void, c_void, c_int, c_void]
        for fn in ("wbc_sim_refresh_dof_state", "wbc_sim_refresh_root_state", "wbc_sim_refresh_net_contact_force",
                   "wbc_sim_refresh_force_sensor"):
            getattr(L, fn).argtypes = [c_void]
        L.wbc_sim_refresh_rigid_body_state.argtypes = [c_void, c_void]
        L.wbc_sim_get_step_counter.argtypes = [c_void, C.POINTER(C.c_int64)]
        L.wbc_sim_set_step_counter.argtypes = [c_void, C.c_int64]
        L.wbc_gae_compute.argtypes = [c_void] * 7 + [c_int, c_int, C.c_float, C.c_float, c_void]
        L.wbc_gae_normalize.argtypes = [c_void, c_void, C.c_int64, c_void]
        L.wbc_gae_workspace_doubles.argtypes = [c_int]
        L.wbc_abi_sizes.argtypes = [C.POINTER(c_int)]
        L.wbc_tensor_spec.argtypes = [c_int, C.POINTER(C.c_int64), C.POINTER(c_int), C.POINTER(c_int)]
        L.wbc_hist_latent.argtypes = [c_void, c_void, c_void, c_int, c_void]
        L.wbc_sim_arm_dynamics.argtypes = [c_void, c_void, c_void, c_void, c_void, c_void, c_void]
        L.wbc_sim_episode_stats.argtypes = [c_void, C.c_float, c_void, c_void, c_void]
        L.wbc_sim_episode_stats_track.argtypes = [c_void, C.c_float, c_void, c_void, c_void, c_int, c_void]
        L.wbc_rollout_store.argtypes = [c_void] * 5 + [C.c_float, c_void, c_void, c_int, c_void]
        L.wbc_runner_track_episodes.argtypes = [c_void, c_void, c_void, c_int, c_int, c_void, c_void]
        L.wbc_runner_track_state_floats.argtypes = [c_int, c_int]
        L.wbc_runner_track_state_floats.restype = C.c_size_t
        L.wbc_policy_act.argtypes = [c_void] * 9 + [c_int, c_void]
        L.wbc_policy_act_job.argtypes = [c_void] * 9 + [c_int, C.POINTER(abi.WbcSideJob), c_void]
        L.wbc_sim_episode_stats_job.argtypes = [c_void, C.c_float, c_void, c_void, c_void, c_int, C.POINTER(abi.WbcSideJob)]
        L.wbc_side_job_run.argtypes = [C.POINTER(abi.WbcSideJob), c_void]
        L.wbc_policy_pack.argtypes = [c_void, c_void, c_void]
        L.wbc_ppo_minibatch_grad.argtypes = [c_void] * 9 + [c_int] + [C.c_float] * 4 + [c_int, c_void, c_void, c_void, c_void]
        L.wbc_ppo_minibatch_grad_packed.argtypes = L.wbc_ppo_minibatch_grad.argtypes
        L.wbc_ppo_pack_invalidate.argtypes = [c_void]
        L.wbc_asset_load.argtypes = [C.c_char_p, C.POINTER(c_void)]
        L.wbc_asset_free.argtypes = [c_void]
        L.wbc_asset_free.restype = None
        for fn in ("wbc_asset_dof_count", "wbc_asset_rigid_body_count"):
            getattr(L, fn).argtypes = [c_void]
        for fn in ("wbc_asset_dof_name", "wbc_asset_rigid_body_name"):
            getattr(L, fn).argtypes = [c_void, c_int]
            getattr(L, fn).restype = C.c_char_p
        L.wbc_asset_dof_properties.argtypes = [c_void] * 5
        L.wbc_asset_model.argtypes = [c_void]
        L.wbc_asset_model.restype = C.POINTER(abi.WbcModel)
        L.wbc_asset_task_cfg.argtypes = [c_void]
        L.wbc_asset_task_cfg.restype = C.POINTER(abi.WbcTaskCfg)
        L.wbc_asset_curriculum.argtypes = [c_void, c_int]
        L.wbc_asset_curriculum.restype = C.POINTER(abi.WbcCurriculum)
        L.wbc_ppo_sq_partials_offset.argtypes = [c_int]
        L.wbc_ppo_sq_partials_offset.restype = C.c_size_t
        L.wbc_ppo_clip_adam.argtypes = [c_void] * 4 + [C.c_float] * 7 + [c_void, c_void, c_void]
        L.wbc_ppo_clip_adam_packed.argtypes = [c_void] * 4 + [C.c_float] * 7 + [c_void, c_void, c_void, c_int, c_void]
        L.wbc_hist_train_grad.argtypes = [c_void, c_void, c_void, c_void, c_int, c_void, c_void, c_void]
        L.wbc_hist_train_workspace_floats.restype = C.c_size_t
        L.wbc_hist_clip_adam.argtypes = [c_void] * 4 + [C.c_float] * 7 + [c_int, c_void, c_void]
        L.wbc_priv_latent.argtypes = [c_void, c_void, c_void, c_int, c_void]
        L.wbc_ppo_workspace_floats.argtypes = [c_int]
        L.wbc_ppo_workspace_floats.restype = C.c_size_t
        _lib = L
    return _lib


EXPORTED_SYMBOLS = [
    "wbc_last_error", "wbc_sim_arena_bytes", "wbc_sim_create", "wbc_sim_destroy", "wbc_sim_get_tensor",
    "wbc_sim_set_env_params", "wbc_sim_set_heightfield", "wbc_sim_set_curriculum", "wbc_sim_step", "wbc_sim_step_to", "wbc_sim_step_rollout", "wbc_get_heights", "wbc_sim_reset_all",
    "wbc_sim_set_dof_forces", "wbc_sim_simulate", "wbc_sim_set_root_state", "wbc_sim_set_dof_state",
    "wbc_sim_set_root_state_indexed", "wbc_sim_set_dof_state_indexed", "wbc_sim_refresh_dof_state",
    "wbc_sim_refresh_root_state", "wbc_sim_refresh_net_contact_force", "wbc_sim_refresh_force_sensor",
    "wbc_sim_refresh_rigid_body_state", "wbc_sim_get_step_counter", "wbc_sim_set_step_counter", "wbc_gae_compute",
    "wbc_gae_normalize", "wbc_gae_workspace_doubles", "wbc_abi_sizes", "wbc_sim_episode_stats", "wbc_rollout_store", "wbc_hist_latent", "wbc_sim_arm_dynamics", "wbc_policy_act", "wbc_policy_pack", "wbc_policy_pack_floats", "wbc_ppo_minibatch_grad", "wbc_ppo_minibatch_grad_packed", "wbc_ppo_pack_invalidate", "wbc_ppo_clip_adam", "wbc_ppo_clip_adam_packed", "wbc_ppo_clip_adam_workspace_floats", "wbc_ppo_grad_floats",
    "wbc_ppo_num_splits", "wbc_ppo_workspace_floats", "wbc_ppo_sq_partials_offset", "wbc_hist_train_grad", "wbc_hist_train_grad_floats",
    "wbc_hist_train_workspace_floats", "wbc_hist_clip_adam", "wbc_priv_latent", "wbc_runner_track_episodes", "wbc_sim_episode_stats_track", "wbc_tensor_spec", "wbc_policy_act_job", "wbc_sim_episode_stats_job", "wbc_side_job_run",
    "wbc_runner_track_state_floats", "wbc_asset_load", "wbc_asset_free", "wbc_asset_dof_count", "wbc_asset_rigid_body_count", "wbc_asset_dof_name",
    "wbc_asset_rigid_body_name", "wbc_asset_dof_properties", "wbc_asset_model", "wbc_asset_task_cfg", "wbc_asset_curriculum"]


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise WbcError(f"{what} failed ({rc}): {lib().wbc_last_error().decode()}")
