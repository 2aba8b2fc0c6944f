"""`LeggedRobot` / `WidowGo1`: the task classes with the reference's public surface (constructor
arguments, attribute names, tensor views, `step` / `reset` / `update_command_curriculum`), backed by
the fused HIP step instead of Isaac Gym + ~150 eager torch ops.

Reference: legged_gym/envs/widowGo1/widowGo1.py:49 (WidowGo1), legged_gym/envs/base/legged_robot.py:52-77
and legged_gym/envs/base/base_task.py:41-131 (constructor, buffers, reset). Everything the reference
computes per step in Python lives in csrc/wbc_step_kernel.hip; this file only does what the reference
does on the host at construction time (domain randomisation draws, WG:207-228,402-408,431-496,574-575)
and exposes views.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from . import abi
from .curriculum import make_curriculum
from .rsl_rl.env import VecEnv
from .sim import WbcSim
from .urdf_model import RobotModel, build_model


# isaacgym.torch_utils helpers the operational-space controller uses (quaternions are xyzw)
def _quat_mul(a, b):
    x1, y1, z1, w1 = a.unbind(-1)
    x2, y2, z2, w2 = b.unbind(-1)
    return torch.stack([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2], dim=-1)


def _quat_apply(q, v):
    xyz = q[:, :3]
    t = torch.linalg.cross(xyz, v, dim=-1) * 2
    return v + q[:, 3:] * t + torch.linalg.cross(xyz, t, dim=-1)


def _orientation_error(desired, current):                                           # widowGo1.py orientation_error
    cc = torch.cat([-current[:, :3], current[:, 3:]], dim=-1)
    q_r = _quat_mul(desired, cc)
    return q_r[:, 0:3] * torch.sign(q_r[:, 3]).unsqueeze(-1)


def _yaw_quat(quat):
    """Quaternion of the yaw of `quat` alone (base_yaw_quat, WG:878-880)."""
    x, y, z, w = quat.unbind(-1)
    yaw = torch.atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))
    zero = torch.zeros_like(yaw)
    return torch.stack([zero, zero, torch.sin(yaw / 2), torch.cos(yaw / 2)], dim=-1)


def stale_time_outs(prev, time_out_buf, reset_buf):
    """extras['time_outs'] as the reference publishes it (quirk Q9). `check_termination` builds a NEW time_out_buf tensor every
    step (WG:943-944) but `extras["time_outs"]` is only re-bound inside reset_idx after its early return for an empty env_ids
    (WG:705-706, 753-754): on a step in which no env resets, the learner is handed the mask of the LAST step that had a reset and
    bootstraps those envs' rewards once more (PPO:133-134). Device-side select, no host synchronisation."""
    return torch.where((reset_buf != 0).any(), time_out_buf, prev)


class EpisodeInfo(dict):
    """extras['episode'] (WG:743-750): a dict of 0-dim device tensors and floats, as the reference's. `vector` is the one device
    tensor the tensor entries are views of and `vector_index` maps their keys to its rows, so a logger can average a rollout's
    worth of these with one stack + one host copy instead of one .item() per key (OnPolicyRunner.log)."""
    vector = None
    vector_index = None


def draw_env_params(cfg, n, seed, dt, strip_origins=True, levels=False, gen=None):
    """The per-env constants the reference draws on the host at construction, as numpy arrays keyed by WbcSim.set_env_params'
    arguments (+ 'env_origins' as a torch tensor and 'levels_gen', the generator positioned where the terrain-level draw takes
    it): env origins on the strip in front of the field (WG:207-224), the box's lateral offset (WG:226-227), per-env friction from
    1000 buckets (WG:480-490), base mass / centre of mass / gripper mass (WG:431-456), the box's added mass (WG:458-466), motor
    strengths (WG:402-408), EE-trajectory timing (WG:574-575). Pure: nothing but `cfg` decides the result for a given seed."""
    if gen is None:
        gen = torch.Generator(device="cpu")
        gen.manual_seed(int(seed))
    nprng = np.random.default_rng(int(seed))
    rand = lambda lo, hi, *shape: (hi - lo) * torch.rand(*shape, generator=gen) + lo   # noqa: E731
    t, dr = cfg.terrain, cfg.domain_rand
    origins = torch.zeros(n, 3)
    if strip_origins:                           # WG:207-224: a strip in front of the (Perlin) field
        half_col = t.tot_cols * t.horizontal_scale / 2
        half_row = t.tot_rows * t.horizontal_scale / 2
        origins[:, 0] = rand(-2.5 * half_col / 5, -2 * half_col / 5, n)
        origins[:, 1] = rand(-half_row + 10, half_row - 10, n)
    levels_gen = torch.Generator(device="cpu")
    levels_gen.set_state(gen.get_state())       # (the terrain-level draw of LR:717-731 comes next in the stream)
    if levels:                                  # advance past it exactly as _get_env_origins_levels will
        max_init_level = t.max_init_terrain_level if t.curriculum else t.num_rows - 1
        torch.randint(0, max_init_level + 1, (n,), generator=gen)
    sign = torch.randint(0, 2, (n,), generator=gen) * 2 - 1
    box_dy = sign * rand(cfg.box.box_env_origins_y_range[0], cfg.box.box_env_origins_y_range[1], n)
    if dr.randomize_friction:
        buckets = rand(dr.friction_range[0], dr.friction_range[1], 1000)
        friction = buckets[torch.randint(0, 1000, (n,), generator=gen)]
    else:
        friction = torch.ones(n)
    dmass = nprng.uniform(*dr.added_mass_range, size=n) if dr.randomize_base_mass else np.zeros(n)
    gmass = nprng.uniform(*dr.gripper_added_mass_range, size=n) if dr.randomize_gripper_mass else np.zeros(n)
    if dr.randomize_base_com:
        lo = [dr.added_com_range_x[0], dr.added_com_range_y[0], dr.added_com_range_z[0]]
        hi = [dr.added_com_range_x[1], dr.added_com_range_y[1], dr.added_com_range_z[1]]
        dcom = nprng.uniform(lo, hi, size=(n, 3))
    else:
        dcom = np.zeros((n, 3))
    if dr.randomize_motor:
        motor = torch.cat([rand(*dr.leg_motor_strength_range, n, 12), rand(*dr.arm_motor_strength_range, n, 6)], dim=1)
    else:
        motor = torch.ones(n, cfg.env.num_torques)
    box_dmass = nprng.uniform(*cfg.box.added_mass_range, size=n) if cfg.box.randomize_base_mass else np.zeros(n)   # WG:458-466
    traj = rand(cfg.goal_ee.traj_time[0], cfg.goal_ee.traj_time[1], n) / dt
    total = traj + rand(cfg.goal_ee.hold_time[0], cfg.goal_ee.hold_time[1], n) / dt
    return dict(friction=friction.numpy(), base_dmass=dmass, base_dcom=dcom, gripper_dmass=gmass, motor_strength=motor.numpy(),
                env_origins=origins, box_delta_y=box_dy.numpy(), traj_timesteps=traj.numpy(), traj_total_timesteps=total.numpy(),
                box_dmass=box_dmass, levels_gen=levels_gen)


class BaseTask(VecEnv):
    """Buffer/attribute contract of legged_gym/envs/base/base_task.py:41-131 (viewer omitted: headless)."""

    def __init__(self, cfg, sim_params=None, physics_engine=None, sim_device="cuda:0", headless=True):
        self.cfg = cfg
        self.sim_params = sim_params
        self.physics_engine = physics_engine
        self.sim_device = str(sim_device)
        self.device = torch.device(self.sim_device)
        self.headless = headless
        self.num_envs = cfg.env.num_envs
        self.num_obs = cfg.env.num_observations
        self.num_privileged_obs = cfg.env.num_privileged_obs
        self.num_actions = cfg.env.num_actions
        self.privileged_obs_buf = None
        self._obs_output = None
        self._store_output = None
        self.extras = {}
        self.viewer = None
        self.enable_viewer_sync = False

    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return self.privileged_obs_buf

    def reset_idx(self, env_ids):
        raise NotImplementedError

    def reset(self):
        """Reset all robots, then one zero-action step (BT:127-131)."""
        self.reset_idx(torch.arange(self.num_envs, device=self.device), start=True)
        obs, privileged_obs, _, _, _, _ = self.step(torch.zeros(self.num_envs, self.num_actions, device=self.device))
        return obs, privileged_obs

    def render(self, sync_frame_time=True):
        return None


# Methods of the reference's WidowGo1 / LeggedRobot whose work happens INSIDE the fused step (csrc/wbc_step_kernel.hip): a Python
# override in a subclass would never be called, so defining one is an error at class-creation time, not a silent no-op.
FUSED_METHODS = ("compute_observations", "_compute_torques", "check_termination", "compute_reward", "post_physics_step",
                 "_post_physics_step_callback", "_push_robots", "_reset_dofs", "_reset_root_states", "update_curr_ee_goal",
                 "collision_check", "get_body_orientation", "_prepare_reward_function", "_process_rigid_body_props",
                 "_process_rigid_shape_props", "_process_dof_props", "_get_noise_scale_vec")
FUSED_PREFIXES = ("_reward_", "_resample_")


def _fused_overrides(namespace):
    return sorted(n for n in namespace if n in FUSED_METHODS or n.startswith(FUSED_PREFIXES))


class LeggedRobot(BaseTask):
    """Shared pieces of LR:52-77,279-305: config parsing, DoF limits from the URDF, views."""

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        bad = _fused_overrides(vars(cls))
        if bad:
            raise TypeError(
                f"{cls.__name__} overrides {', '.join(bad)}: the fused step computes this on the device (csrc/wbc_step_kernel.hip, spec "
                "oracle/wbc_oracle.c) and never calls the Python method. Change a reward through cfg.rewards.scales / arm_scales (every "
                "_reward_* of the reference is a table entry of wbc_curriculum), thresholds and gains through the config (wbc_task_cfg); for "
                "new arithmetic either add it to the kernel and the oracle, or step your own torch code on the C-ABI tensors "
                "(wbc_sim_get_tensor: ROOT_STATES, DOF_STATE, TORQUES, ... are live views) after WidowGo1.step returns.")

    def __init__(self, cfg, sim_params=None, physics_engine=None, sim_device="cuda:0", headless=True,
                 robot_model: Optional[RobotModel] = None, seed: Optional[int] = None):
        super().__init__(cfg, sim_params, physics_engine, sim_device, headless)
        self.debug_viz = False
        self.init_done = False
        self.robot_model = robot_model if robot_model is not None else self._load_model(cfg)
        self._seed = int(seed if seed is not None else getattr(cfg, "seed", 1))
        self._parse_cfg(cfg)
        self.create_sim()
        self._init_buffers()
        self.init_done = True

    @staticmethod
    def _load_model(cfg) -> RobotModel:
        path = cfg.asset.file
        if "{LEGGED_GYM_ROOT_DIR}" not in path and path.endswith(".urdf"):
            return build_model(path)
        return abi.load_default_model()      # the packaged widowGo1 tables (tools/extract_model.py)


class WidowGo1(LeggedRobot):
    def _parse_cfg(self, cfg):                                                     # WG:78-121
        sim_dt = self.sim_params.dt if self.sim_params is not None and hasattr(self.sim_params, "dt") else cfg.sim.dt
        self.num_torques = cfg.env.num_torques
        self.dt = cfg.control.decimation * sim_dt
        self._sim_dt = sim_dt
        self.obs_scales = cfg.normalization.obs_scales
        self.update_counter = 0
        self.max_episode_length_s = cfg.env.episode_length_s
        self.max_episode_length = np.ceil(self.max_episode_length_s / self.dt)
        self.push_interval = np.ceil(cfg.domain_rand.push_interval_s / self.dt)
        self.clip_actions = cfg.normalization.clip_actions
        self.action_delay = cfg.env.action_delay
        if cfg.terrain.mesh_type not in ["heightfield", "trimesh"]:
            cfg.terrain.curriculum = False
        self.collect_episode_stats = True
        # opt-in (OnPolicyRunner.learn sets it): extras['episode'] is computed on a side stream, overlapping the next policy
        # inference; its consumer must synchronise the DEVICE before reading the values (the runner does, at the end of the rollout)
        self.async_episode_stats = False
        self._stats_stream = None
        self._stats_pending = False
        self._track = (None, 0)          # (state, cap) of attach_episode_tracker: advanced by the statistics launch
        # opt-in (OnPolicyRunner.learn sets it): step() does not launch the episode statistics itself but leaves them as a side job
        # (take_stats_job) that the policy inference following every rollout step carries as a few extra workgroups -- one launch
        # and one inter-kernel dependency stall less per env step. extras['episode'] is then complete once that launch has run;
        # a job nobody takes is executed by the next step() / flush_stats_job().
        self.defer_episode_stats = False
        self._stats_job = None
        # cfg.env.reference_stale_time_outs: publish extras['time_outs'] the way the reference does (quirk Q9). The fused step's
        # in-kernel reward bootstrap uses the CURRENT mask, so with the option on the reward / done slots are filled by
        # wbc_rollout_store from the published (possibly stale) mask instead.
        self._stale_time_outs_on = bool(getattr(cfg.env, "reference_stale_time_outs", False))
        self._stale_mask = None

    # ---- construction ----------------------------------------------------------------------
    def create_sim(self):                                                           # WG:230-237, 255-429
        cfg, m, n = self.cfg, self.robot_model, self.num_envs
        # asset.self_collisions is Isaac Gym's collision FILTER: 0 = self-collision enabled (widowGo1_config.py:180)
        self.wmodel = abi.fill_model(m, foot_name=cfg.asset.foot_name, self_collisions=int(getattr(cfg.asset, "self_collisions", 0)) == 0,
                                     box_size=float(cfg.box.box_size), rest_offset=float(abi._get(cfg, "sim.physx.rest_offset", 0.0)))
        self.tcfg = abi.fill_task_cfg(cfg, m, sim_dt=self._sim_dt)
        self.sim = WbcSim(self.wmodel, self.tcfg, n, self.device, seed=self._seed)
        self.num_dofs, self.num_bodies = m.num_dofs, m.num_rigid_bodies
        self.dof_names, self.body_names = list(m.dof_names), list(m.rb_names)
        self.dof_wo_gripper_names = self.dof_names[:-2]
        self.body_names_to_idx = {n_: i for i, n_ in enumerate(self.body_names)}
        self.dof_names_to_idx = {n_: i for i, n_ in enumerate(self.dof_names)}
        self.gripper_idx = self.body_names_to_idx["wx250s/ee_gripper_link"]
        dev = self.device
        feet = [i for i, s in enumerate(self.body_names) if cfg.asset.foot_name in s]
        self.feet_indices = torch.tensor(feet, dtype=torch.long, device=dev)
        pen = [i for name in cfg.asset.penalize_contacts_on for i, s in enumerate(self.body_names) if name in s]
        self.penalized_contact_indices = torch.tensor(pen, dtype=torch.long, device=dev)
        term = [i for name in cfg.asset.terminate_after_contacts_on for i, s in enumerate(self.body_names) if name in s]
        self.termination_contact_indices = torch.tensor(term, dtype=torch.long, device=dev)
        self._terrain_setup()
        self._randomise()

    def _terrain_setup(self):
        """create_sim's terrain part (WG:235-253): mesh_type 'plane' / None is flat ground at z = 0; 'trimesh' or
        'heightfield' with the widowGo1 Perlin parameters (zScale, tot_cols, tot_rows) generates the fractal
        terrain of utils/terrain.py:40-99 and attaches its height grid to the contact kernel."""
        t = self.cfg.terrain
        self.height_samples = None
        self.terrain = None
        if t.mesh_type in ("trimesh", "heightfield") and hasattr(t, "zScale"):
            from .terrain import TerrainPerlin
            self.terrain = TerrainPerlin(t, seed=self._seed)
            self.set_heightfield(self.terrain.heightsamples, self.terrain.horizontal_scale, self.terrain.vertical_scale,
                                 *self.terrain.transform)
        elif t.mesh_type in ("trimesh", "heightfield") and hasattr(t, "terrain_proportions"):
            # the base class's sub-terrain grid (LR:79-95 create_sim -> Terrain(cfg.terrain, num_envs), utils/terrain.py:101-227);
            # np.random is the reference's stream for it (seeded as helpers.set_seed does)
            from .terrain import Terrain
            saved = np.random.get_state()             # building an env must not reseed the process-global numpy stream
            try:
                np.random.seed(self._seed)
                self.terrain = Terrain(t, self.num_envs)
            finally:
                np.random.set_state(saved)
            self.set_heightfield(self.terrain.heightsamples, self.terrain.horizontal_scale, self.terrain.vertical_scale,
                                 *self.terrain.transform)

    def set_heightfield(self, heights_i16: np.ndarray, horizontal_scale, vertical_scale, tx, ty, tz):
        self.sim.set_heightfield(heights_i16, horizontal_scale, vertical_scale, tx, ty, tz)
        self.height_samples = torch.from_numpy(np.asarray(heights_i16)).to(self.device)

    def _randomise(self):
        """Host-side draws of _get_env_origins WG:207-228, _process_rigid_shape_props WG:468-496,
        _process_rigid_body_props WG:431-456, motor strengths WG:402-408, trajectory timing WG:574-575 (draw_env_params)."""
        cfg, n = self.cfg, self.num_envs
        t = cfg.terrain
        gen = torch.Generator(device="cpu")
        gen.manual_seed(self._seed)
        grid = self.terrain is not None and hasattr(self.terrain, "proportions")          # the base class's Terrain
        self._terrain_levels_on = bool(self.terrain is not None and (grid or t.curriculum))
        p = draw_env_params(cfg, n, self._seed, self.dt, strip_origins=not grid, levels=self._terrain_levels_on, gen=gen)
        origins = p.pop("env_origins")
        self.env_origins = origins.to(self.device)
        self.custom_origins = True
        if self._terrain_levels_on:                 # base-class placement on the terrain's (level, type) platforms, LR:717-731
            if not grid:
                self.terrain.level_grid(int(t.num_rows), int(t.num_cols))
            origins = self._get_env_origins_levels(p.pop("levels_gen")).cpu()
        p.pop("levels_gen", None)
        self.sim.set_env_params(env_origins=origins.numpy(), **p)
        self.sim.set_curriculum(make_curriculum(cfg, max(self.update_counter, 0)))

    def _init_buffers(self):
        """Zero-copy views with the attribute names of WG:498-672."""
        s, dev, tc = self.sim, self.device, self.tcfg
        T = s.tensor
        self._root_states = T("ROOT_STATES")
        self.root_states, self.box_root_state = self._root_states[:, 0, :], self._root_states[:, 1, :]
        self.dof_state = T("DOF_STATE").view(self.num_envs * self.num_dofs, 2)
        dv = T("DOF_STATE")
        self.dof_pos, self.dof_vel = dv[..., 0], dv[..., 1]
        self.dof_pos_wo_gripper, self.dof_vel_wo_gripper = self.dof_pos[:, :-2], self.dof_vel[:, :-2]
        self.base_quat = self.root_states[:, 3:7]
        self._contact_forces = T("NET_CONTACT_FORCE")
        self.contact_forces, self.box_contact_force = self._contact_forces[:, :-1, :], self._contact_forces[:, -1, :]
        self._rigid_body_state = T("RIGID_BODY_STATE")
        self.rigid_body_state, self.box_rigid_body_state = self._rigid_body_state[:, :-1, :], self._rigid_body_state[:, -1, :]
        self.force_sensor_tensor = T("FORCE_SENSOR")
        self.ee_pos = self.rigid_body_state[:, self.gripper_idx, :3]
        self.ee_orn = self.rigid_body_state[:, self.gripper_idx, 3:7]
        self.ee_vel = self.rigid_body_state[:, self.gripper_idx, 7:]
        self.box_pos = self.box_root_state[:, 0:3]
        self.obs_buf = T("OBS_BUF")
        self.obs_history_buf = T("OBS_HISTORY")
        self.action_history_buf = T("ACTION_HISTORY")
        self.rew_buf, self.arm_rew_buf = T("REW_BUF"), T("ARM_REW_BUF")
        self.reset_buf = T("RESET_BUF")
        self._episode_length_buf = T("EPISODE_LENGTH")
        self.time_out_buf = T("TIME_OUT_BUF").view(torch.bool)
        self.torques, self.actions, self.last_actions = T("TORQUES"), T("ACTIONS"), T("LAST_ACTIONS")
        self.last_dof_vel, self.last_root_vel = T("LAST_DOF_VEL"), T("LAST_ROOT_VEL")
        self.commands = T("COMMANDS")
        self.base_lin_vel, self.base_ang_vel = T("BASE_LIN_VEL"), T("BASE_ANG_VEL")
        g = T("GOAL_STATE")
        self.ee_start_sphere, self.ee_goal_sphere, self.ee_goal_cart = g[:, 0:3], g[:, 3:6], g[:, 6:9]
        self.curr_ee_goal_sphere, self.curr_ee_goal_cart = g[:, 9:12], g[:, 12:15]
        self.ee_goal_delta_orn_euler, self.ee_goal_orn_euler = g[:, 15:18], g[:, 18:21]
        self.goal_timer, self.traj_timesteps, self.traj_total_timesteps = g[:, 21], g[:, 22], g[:, 23]
        self.curr_ee_goal = self.curr_ee_goal_cart if self.cfg.goal_ee.command_mode == "cart" else self.curr_ee_goal_sphere
        self.mass_params_tensor = T("MASS_PARAMS")
        self.friction_coeffs_tensor = T("FRICTION")
        self.motor_strength = T("MOTOR_STRENGTH")
        self._episode_sums, self._metric_sums = T("EPISODE_SUMS"), T("METRIC_SUMS")
        self._episode_sums_done, self._metric_sums_done = T("EPISODE_SUMS_DONE"), T("METRIC_SUMS_DONE")
        self.episode_sums = {name: self._episode_sums[:, i] for i, name in enumerate(abi.REWARD_TERMS)}
        self.episode_metric_sums = {name: self._metric_sums[:, i] for i, name in enumerate(abi.METRIC_NAMES)}
        f = lambda arr: torch.tensor(list(arr), dtype=torch.float, device=dev)    # noqa: E731
        self.p_gains, self.d_gains = f(tc.p_gains), f(tc.d_gains)
        self.default_dof_pos = f(tc.default_dof_pos)
        self.default_dof_pos_wo_gripper = self.default_dof_pos[:-2]
        self.torque_limits = f(tc.torque_limits)
        self.action_scale = f(tc.action_scale)
        m = self.robot_model
        self.dof_pos_limits = torch.tensor(np.stack([m.dof_lower, m.dof_upper], 1), dtype=torch.float, device=dev)   # LR:294-296
        self.dof_vel_limits = torch.tensor(m.dof_velocity, dtype=torch.float, device=dev)
        self.commands_scale = f(tc.commands_scale)
        # operational-space controller of the torque-supervision path (WG:559-565, 664-670)
        self.arm_osc_kp = torch.tensor(np.asarray(self.cfg.arm.osc_kp, dtype=np.float64), dtype=torch.float, device=dev)
        self.arm_osc_kd = torch.tensor(np.asarray(self.cfg.arm.osc_kd, dtype=np.float64), dtype=torch.float, device=dev)
        self.ee_orn_des = torch.tensor([0, 0.7071068, 0, 0.7071068], device=dev).repeat((self.num_envs, 1))
        self.z_invariant_offset = torch.full((self.num_envs, 1), float(tc.z_invariant_offset), device=dev)
        self._arm_link_rb = list(range(m.num_rigid_bodies - 9, m.num_rigid_bodies))
        self.link_mass = torch.tensor(m.rb_mass[-9:], dtype=torch.float, device=dev).unsqueeze(0).repeat((self.num_envs, 1))
        self.common_step_counter = 0
        self.extras = {"episode": {}}
        self._active_terms = None
        self._reset_travel = T("RESET_TRAVEL")
        self._sim_env_origins = T("ENV_ORIGINS")                                     # what the kernel's resets read
        if self.cfg.terrain.measure_heights:                                         # WG:637-639
            self.height_points = self._init_height_points()
        self.measured_heights = 0
        self._refresh_ranges()

    @property
    def episode_length_buf(self):
        return self._episode_length_buf

    @episode_length_buf.setter
    def episode_length_buf(self, value):          # OnPolicyRunner.learn re-binds this attribute (OPR:107-108)
        self._episode_length_buf.copy_(value)

    # ---- curriculum ------------------------------------------------------------------------
    def _refresh_ranges(self):
        cur = make_curriculum(self.cfg, self.update_counter)
        self._cur = cur
        self.lin_vel_x_ranges = np.array(list(cur.lin_vel_x_range))
        self.ang_vel_yaw_ranges = np.array(list(cur.ang_vel_yaw_range))
        self.goal_ee_l_ranges = np.array(list(cur.goal_l_range))
        self.goal_ee_p_ranges = np.array(list(cur.goal_p_range))
        self.goal_ee_y_ranges = np.array(list(cur.goal_y_range))
        # keys = the config's non-zero scales, fixed at construction (WG:128-157); values = the current (scheduled) scales
        self.reward_scales = {n: cur.leg_reward_scale[i] for i, n in enumerate(abi.REWARD_TERMS) if (cur.leg_active_mask >> i) & 1}
        self.arm_reward_scales = {n: cur.arm_reward_scale[i] for i, n in enumerate(abi.REWARD_TERMS) if (cur.arm_active_mask >> i) & 1}
        if self._active_terms is None:     # episode_sums keys = the config's non-zero scales, fixed at construction (WG:128-163)
            from .curriculum import _scales
            cfg_leg, cfg_arm = _scales(self.cfg.rewards.scales), _scales(self.cfg.rewards.arm_scales)
            self._active_terms = [(i, n) for i, n in enumerate(abi.REWARD_TERMS) if cfg_leg.get(n, 0) != 0 or cfg_arm.get(n, 0) != 0]
            self._episode_vector_index = {"rew_" + n: i for i, n in self._active_terms}
            self._episode_vector_index.update({"metric_" + n: abi.NREW + i for i, n in enumerate(abi.METRIC_NAMES)})
        return cur

    def update_command_curriculum(self):                                            # WG:678-692
        self.update_counter += 1
        self.sim.set_curriculum(self._refresh_ranges())

    # ---- stepping --------------------------------------------------------------------------
    def reset_idx(self, env_ids, start=False):
        if len(env_ids) == 0:
            return
        if not start or len(env_ids) != self.num_envs:
            raise NotImplementedError("per-env resets happen inside the fused step; only reset_idx(all, start=True) is exposed")
        self.sim.reset_all()
        self._fill_extras(start=True)

    def _fill_extras(self, start=False):
        """extras['episode'] / extras['time_outs'] of WG:743-754, 902-906, without host syncs."""
        if self.collect_episode_stats:
            scale = 1.0 / self.max_episode_length_s
            self.flush_stats_job()                       # a deferred job nobody took: its inputs are about to be overwritten
            if self.defer_episode_stats and not start:   # the launch is left to whoever takes the job (the next policy inference)
                stv, self._stats_job = self.sim.episode_stats_job(scale, *self._track)
            elif self.async_episode_stats and self.device.type == "cuda":
                if self._stats_stream is None:
                    self._stats_stream = torch.cuda.Stream(self.device)
                side = self._stats_stream
                side.wait_stream(torch.cuda.current_stream(self.device))                # after the step kernel's writes
                with torch.cuda.stream(side):
                    stv = self.sim.episode_stats(scale, *self._track)
                self._stats_pending = True                                              # the next step() waits for it (it rewrites the inputs)
            else:
                stv = self.sim.episode_stats(scale, *self._track)
            st = stv.unbind(0)                                                           # one launch (or side job), 32 scalar views
            ep = EpisodeInfo()
            ep.vector, ep.vector_index = stv, self._episode_vector_index                 # the same numbers as ONE device tensor
            for i, name in self._active_terms:
                ep["rew_" + name] = st[i]
            for i, name in enumerate(abi.METRIC_NAMES):
                ep["metric_" + name] = st[abi.NREW + i]
            ep["coeff_lin_vel_x_upper_bound"] = self.lin_vel_x_ranges[1]
            ep["coeff_lin_vel_x_lower_bound"] = self.lin_vel_x_ranges[0]
            ep["coeff_ang_vel_yaw_upper_bound"] = self.ang_vel_yaw_ranges[1]
            ep["coeff_ang_vel_yaw_lower_bound"] = self.ang_vel_yaw_ranges[0]
            ep["coeff_tracking_ang_vel_yaw_exp"] = self.reward_scales.get("tracking_ang_vel_yaw_exp", 0.0)
            self.extras["episode"] = ep
        if self.cfg.env.send_timeouts:
            if self._stale_time_outs_on:                 # opt-in quirk Q9; at construction the reference binds the all-False initial mask
                prev = self._stale_mask if self._stale_mask is not None and not start else torch.zeros_like(self.time_out_buf)
                self._stale_mask = stale_time_outs(prev, self.time_out_buf, self.reset_buf) if not start else prev
                self.extras["time_outs"] = self._stale_mask
            else:
                self.extras["time_outs"] = self.time_out_buf

    # ---- torque supervision (WG:1178-1181, 1201-1242): default off (WGC:173) ------------------------------------
    def _refresh_arm_dynamics(self):
        self.mm, self.ee_j_eef, self._g_torque = self.sim.arm_dynamics(self._arm_link_rb, self.robot_model.rb_mass[-9:])

    def get_g_torques(self):                                                        # WG:1201-1207
        self._refresh_arm_dynamics()
        return self._g_torque

    def get_arm_mm(self):                                                           # WG:1209-1211
        self._refresh_arm_dynamics()
        return self.mm

    def get_ee_jac(self):                                                           # WG:1213-1215
        self._refresh_arm_dynamics()
        return self.ee_j_eef

    def get_arm_ee_control_torques(self):
        """Operational-space control torques of the arm, the supervision target (WG:1217-1242), line by line; the
        mass matrix, the Jacobian and the gravity torques come from wbc_sim_arm_dynamics instead of Isaac Gym."""
        self._refresh_arm_dynamics()
        m_inv = torch.linalg.pinv(self.mm)
        m_eef = torch.linalg.pinv(self.ee_j_eef @ m_inv @ torch.transpose(self.ee_j_eef, 1, 2))
        ee_orn_normalized = self.ee_orn / torch.norm(self.ee_orn, dim=-1).unsqueeze(-1)
        orn_err = _orientation_error(self.ee_orn_des, ee_orn_normalized)
        yaw = _yaw_quat(self.base_quat)
        pos_err = (torch.cat([self.root_states[:, :2], self.z_invariant_offset], dim=1) + _quat_apply(yaw, self.curr_ee_goal_cart) - self.ee_pos)
        dpose = torch.cat([pos_err, orn_err], -1)
        u = (torch.transpose(self.ee_j_eef, 1, 2) @ m_eef @ (self.arm_osc_kp * dpose - self.arm_osc_kd * self.ee_vel)[:, :6].unsqueeze(-1)).squeeze(-1)
        return u + self._g_torque

    def step(self, actions):
        """WG:1156-1199 as one kernel launch; returns the reference's 6-tuple (views, overwritten by the
        next step)."""
        a = actions.to(self.device, dtype=torch.float32)
        if not a.is_contiguous():
            a = a.contiguous()
        if self.cfg.control.torque_supervision:                                     # WG:1178-1181: state at t = 0 of the step
            self.arm_ee_control_torques = self.get_arm_ee_control_torques()
            self.extras["target_arm_torques"] = self.arm_ee_control_torques
            self.extras["current_arm_dof_pos"] = self.dof_pos[:, -8:-2].clone()
            self.extras["current_arm_dof_vel"] = self.dof_vel[:, -8:-2].clone()
        out, self._obs_output = self._obs_output, None
        store, self._store_output = self._store_output, None
        if self._stale_time_outs_on:
            store = None                              # the learner's process_env_step bootstraps from extras['time_outs']
        self.flush_stats_job()                        # deferred statistics of the previous step that nobody carried
        if self._stats_pending:                       # the side-stream statistics of the previous step read what this step overwrites
            torch.cuda.current_stream(self.device).wait_stream(self._stats_stream)
            self._stats_pending = False
        self.sim.step(a, out, store)
        self.extras["rollout_stored"] = store[2].data_ptr() if store is not None else None
        self.common_step_counter += 1
        if self._terrain_levels_on and self.cfg.terrain.curriculum:                  # WG:708-709 (reset_idx -> _update_terrain_curriculum)
            self._apply_terrain_curriculum()
        if self.cfg.terrain.measure_heights:                                         # WG:932-933
            self.measured_heights = self._get_heights()
        self._fill_extras()
        return (self.obs_buf if out is None else out), self.privileged_obs_buf, self.rew_buf, self.arm_rew_buf, self.reset_buf, self.extras

    # ---- the base class's terrain bookkeeping (legged_robot.py:421-441, 717-731, 777-829) --------------
    def _init_height_points(self):                                                   # LR:777-791
        """Points (base frame) at which the terrain height is sampled: [num_envs, num_height_points, 3]."""
        y = torch.tensor(self.cfg.terrain.measured_points_y, device=self.device, requires_grad=False)
        x = torch.tensor(self.cfg.terrain.measured_points_x, device=self.device, requires_grad=False)
        grid_x, grid_y = torch.meshgrid(x, y, indexing="ij")
        self.num_height_points = grid_x.numel()
        points = torch.zeros(self.num_envs, self.num_height_points, 3, device=self.device, requires_grad=False)
        points[:, :, 0] = grid_x.flatten()
        points[:, :, 1] = grid_y.flatten()
        return points

    def _get_heights(self, env_ids=None):                                            # LR:793-829
        """Terrain heights under the height points of each robot (rotated by the base yaw, offset by the base position):
        one launch of wbc_get_heights (csrc/wbc_terrain_kernel.hip), bit-exact against oracle/terrain_oracle.py.
        `height_samples` is the [x, y] grid the contact kernel uses (the reference views its array with the two dimensions
        swapped, quirk Q3, which only stays harmless while measure_heights is False)."""
        t = self.cfg.terrain
        if t.mesh_type == "plane":
            return torch.zeros(self.num_envs, self.num_height_points, device=self.device, requires_grad=False)
        elif t.mesh_type == "none":
            raise NameError("Can't measure height with terrain mesh type 'none'")
        if self.height_samples is None:
            raise RuntimeError("no height samples attached (set_heightfield)")
        quat, pos, pts = self.base_quat, self.root_states, self.height_points
        if env_ids is not None and len(env_ids) > 0:                                 # (the reference: `if env_ids:`)
            ids = torch.as_tensor(env_ids, dtype=torch.long, device=self.device)
            quat, pos, pts = quat[ids].contiguous(), pos[ids].contiguous(), pts[ids]
        pts = pts.contiguous()
        n, npts = pts.shape[0], pts.shape[1]
        hs = self.height_samples
        assert hs.dtype == torch.int16 and hs.is_contiguous() and quat.stride(1) == 1 and pos.stride(1) == 1
        out = torch.empty(n, npts, dtype=torch.float32, device=self.device)
        from .native import check, lib
        check(lib().wbc_get_heights(quat.data_ptr(), quat.stride(0), pos.data_ptr(), pos.stride(0), pts.data_ptr(), hs.data_ptr(),
                                    hs.shape[0], hs.shape[1], float(t.border_size), float(t.horizontal_scale), float(t.vertical_scale),
                                    out.data_ptr(), n, npts, torch.cuda.current_stream(self.device).cuda_stream), "wbc_get_heights")
        return out

    def _get_env_origins_levels(self, gen=None):                                     # LR:717-731, the terrain-level branch
        """terrain_levels / terrain_types / terrain_origins and env_origins on the terrain's platforms."""
        t, n, dev = self.cfg.terrain, self.num_envs, self.device
        max_init_level = t.max_init_terrain_level
        if not t.curriculum:
            max_init_level = t.num_rows - 1
        self.terrain_levels = torch.randint(0, max_init_level + 1, (n,), generator=gen).to(dev)
        self.terrain_types = torch.div(torch.arange(n, device=dev), (n / t.num_cols), rounding_mode="floor").to(torch.long)
        self.max_terrain_level = t.num_rows
        self.terrain_origins = torch.from_numpy(self.terrain.env_origins).to(dev).to(torch.float)
        self.env_origins = torch.zeros(n, 3, device=dev, requires_grad=False)
        self.env_origins[:] = self.terrain_origins[self.terrain_levels, self.terrain_types]
        return self.env_origins

    def _update_terrain_curriculum(self, env_ids):                                   # LR:421-441
        """The game-inspired curriculum for the envs being reset. `distance` and the command norm are the finished episode's
        (WBC_T_RESET_TRAVEL: the fused step has already re-placed these robots and may have resampled their commands)."""
        if not self.init_done:
            return                                                                   # don't change on the initial reset
        travel = self._reset_travel[env_ids]
        distance = travel[:, 0]
        move_up = distance > self.terrain.env_length / 2                             # walked far enough: harder terrain
        move_down = (distance < travel[:, 1] * self.max_episode_length_s * 0.5) * ~move_up   # less than half the commanded distance
        self.terrain_levels[env_ids] += 1 * move_up - 1 * move_down
        self.terrain_levels[env_ids] = torch.where(self.terrain_levels[env_ids] >= self.max_terrain_level,    # solved the last level:
                                                   torch.randint_like(self.terrain_levels[env_ids], self.max_terrain_level),   # a random one
                                                   torch.clip(self.terrain_levels[env_ids], 0))
        self.env_origins[env_ids] = self.terrain_origins[self.terrain_levels[env_ids], self.terrain_types[env_ids]]

    def _apply_terrain_curriculum(self):
        """reset_idx's call of _update_terrain_curriculum (WG:708-709) for the envs the fused step has just reset, without a
        host synchronisation: the rule of LR:431-441 over all envs under the reset mask (same arithmetic as
        _update_terrain_curriculum on reset_buf.nonzero()), then those robots (and their boxes, WG:769-771) move from the old
        platform to the new one -- the in-kernel reset placed them relative to the old origin."""
        if not self.init_done:
            return
        m = self.reset_buf.bool()
        distance, cmd_norm = self._reset_travel[:, 0], self._reset_travel[:, 1]
        move_up = m & (distance > self.terrain.env_length / 2)
        move_down = m & (distance < cmd_norm * self.max_episode_length_s * 0.5) & ~move_up
        lv = self.terrain_levels + (1 * move_up - 1 * move_down)
        lv = torch.where(m & (lv >= self.max_terrain_level), torch.randint_like(lv, self.max_terrain_level), torch.clip(lv, 0))
        self.terrain_levels.copy_(lv)
        new = self.terrain_origins[lv, self.terrain_types]
        delta = new - self._sim_env_origins                            # zero for the envs that did not reset
        self.env_origins.copy_(new)
        self._sim_env_origins.copy_(new)
        self.root_states[:, :3] += delta
        self.box_root_state[:, 1] += delta[:, 1]

    def take_stats_job(self):
        """The deferred statistics of the last step (sim.SideJob) or None; the taker executes it before the next step()."""
        job, self._stats_job = self._stats_job, None
        return job

    def flush_stats_job(self):
        job = self.take_stats_job()
        if job is not None:
            job.run(torch.cuda.current_stream(self.device).cuda_stream)

    def attach_episode_tracker(self, state, cap):
        """The training loop's episode deques (OnPolicyRunner.learn, OPR:140-154) ride on the per-step statistics launch as one
        extra workgroup (wbc_sim_episode_stats_track): `state` = wbc_runner_track_state_floats(num_envs, cap) zeroed floats on the
        sim device, or None to detach. Needs collect_episode_stats (the launch it rides on). Effective from the next step():
        returns the value common_step_counter will have after the first step this env accounts for (steps before it are the caller's)."""
        self._track = (state, int(cap)) if state is not None else (None, 0)
        return int(self.common_step_counter) + 1

    def restore_terrain_levels(self, levels, arena_restored=False):
        """Checkpoint resume (OnPolicyRunner.load): adopt saved terrain levels. env_origins and the sim's ENV_ORIGINS are
        re-derived from terrain_origins[level, type]; unless the whole sim arena came back with them (then every robot already
        stands where it stood), all robots are re-placed by a full reset on their restored platforms -- the next
        _apply_terrain_curriculum must see a zero origin delta for every env that did not just reset."""
        if not self._terrain_levels_on:
            return
        self.terrain_levels.copy_(levels.to(self.terrain_levels.device, self.terrain_levels.dtype))
        new = self.terrain_origins[self.terrain_levels, self.terrain_types]
        self.env_origins.copy_(new)
        if not arena_restored:
            self._sim_env_origins.copy_(new)
            self.sim.reset_all()                      # in-kernel reset: places robot + box relative to ENV_ORIGINS
            self._fill_extras(start=True)

    def set_rollout_output(self, values, gamma, rewards, dones):
        """The NEXT step() also writes this transition's rollout-storage slots (PPO.process_env_step's tensor work, ppo.py:129-141):
        rewards [N,2] = (rew, arm_rew) + gamma * values * time_outs, dones [N,1] u8 = reset_buf != 0; `values` = the critic's
        [N,2] output PPO.act stored for the acting observation. One-shot; extras['rollout_stored'] = rewards.data_ptr() tells
        the learner the slots are filled."""
        self._store_output = (values, float(gamma), rewards, dones)

    def set_obs_output(self, tensor):
        """The NEXT step() writes its observations into `tensor` (f32 [num_envs, 860], contiguous, on the sim device) and
        returns it instead of obs_buf -- the rollout loop passes the storage slot of the next transition, which saves the
        14 MB copy `observations[step].copy_(obs)` per step. One-shot; obs_buf keeps the last observation written to it."""
        self._obs_output = tensor

    # gym-tensor-API style helpers some callers of the reference use
    def get_foot_contacts(self):                                                    # WG:1090-1098
        return self.force_sensor_tensor.norm(dim=-1) > 1.5
