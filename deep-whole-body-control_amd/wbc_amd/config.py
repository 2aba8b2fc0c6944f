"""Config classes with the reference's names and attribute paths.

`BaseConfig` reproduces the behaviour of legged_gym/envs/base/base_config.py:33-55
(instantiating a config instantiates every nested class recursively, so
`cfg.env.num_envs` works on instances and nested classes can be subclassed).
`LeggedRobotCfg`, `LeggedRobotCfgPPO`, `WidowGo1RoughCfg` and `WidowGo1RoughCfgPPO`
carry the values of legged_gym/envs/base/legged_robot_config.py:33-243 and
legged_gym/envs/widowGo1/widowGo1_config.py:37-382, so an existing widowGo1 experiment
config (a subclass overriding a few fields) is drop-in.

The classes are built from nested dict specs by `make_cfg`, which keeps the whole
hyper-parameter table in one reviewable place (SURVEY.md Appendix C lists the
resolved values this table must equal).
"""
from __future__ import annotations

import inspect
import math

PI = math.pi


class BaseConfig:
    def __init__(self) -> None:
        self.init_member_classes(self)

    @staticmethod
    def init_member_classes(obj) -> None:
        for key in dir(obj):
            if key == "__class__":
                continue
            var = getattr(obj, key)
            if inspect.isclass(var):
                inst = var()
                setattr(obj, key, inst)
                BaseConfig.init_member_classes(inst)


def make_cfg(name: str, base: type, spec: dict) -> type:
    """Build a config class: dict values become nested classes (inheriting from the base's
    nested class of the same name when there is one), everything else a class attribute."""
    ns = {}
    for key, val in spec.items():
        if isinstance(val, dict) and not val.get("__leaf__", False):
            parent = getattr(base, key, None)
            nested_base = parent if inspect.isclass(parent) else object
            ns[key] = make_cfg(key, nested_base, val)
        elif isinstance(val, dict):
            ns[key] = {k: v for k, v in val.items() if k != "__leaf__"}
        else:
            ns[key] = val
    return type(name, (base,), ns)


def leaf(d: dict) -> dict:
    """Mark a dict as a plain value (e.g. default_joint_angles), not a nested class."""
    out = dict(d)
    out["__leaf__"] = True
    return out


# --------------------------------------------------------------------------------------
# LeggedRobotCfg / LeggedRobotCfgPPO: legged_robot_config.py:33-243
# --------------------------------------------------------------------------------------
_GRID_X = [round(-0.8 + 0.1 * i, 1) for i in range(17)]
_GRID_Y = [round(-0.5 + 0.1 * i, 1) for i in range(11)]

LeggedRobotCfg = make_cfg("LeggedRobotCfg", BaseConfig, {
    "env": dict(num_envs=4096, num_observations=235, num_privileged_obs=None, num_actions=12,
                env_spacing=3.0, send_timeouts=True, episode_length_s=20,
                # not in the reference: True reproduces its stale extras['time_outs'] (quirk Q9, envs.stale_time_outs)
                reference_stale_time_outs=False),
    "terrain": dict(mesh_type="trimesh", horizontal_scale=0.1, vertical_scale=0.005, border_size=25,
                    curriculum=True, static_friction=1.0, dynamic_friction=1.0, restitution=0.0,
                    measure_heights=True, measured_points_x=_GRID_X, measured_points_y=_GRID_Y,
                    selected=False, terrain_kwargs=None, max_init_terrain_level=5,
                    terrain_length=8.0, terrain_width=8.0, num_rows=10, num_cols=20,
                    terrain_proportions=[0.1, 0.1, 0.35, 0.25, 0.2], slope_treshold=0.75),
    "commands": dict(curriculum=False, max_curriculum=1.0, num_commands=4, resampling_time=10.0,
                     heading_command=True,
                     ranges=dict(lin_vel_x=[-1.0, 1.0], lin_vel_y=[-1.0, 1.0], ang_vel_yaw=[-1, 1],
                                 heading=[-3.14, 3.14])),
    "init_state": dict(pos=[0.0, 0.0, 1.0], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0],
                       ang_vel=[0.0, 0.0, 0.0],
                       default_joint_angles=leaf({"joint_a": 0.0, "joint_b": 0.0})),
    "control": dict(control_type="P", stiffness=leaf({"joint_a": 10.0, "joint_b": 15.0}),
                    damping=leaf({"joint_a": 1.0, "joint_b": 1.5}), action_scale=0.5, decimation=4),
    "asset": dict(file="", foot_name="None", penalize_contacts_on=[], terminate_after_contacts_on=[],
                  disable_gravity=False, collapse_fixed_joints=True, fix_base_link=False,
                  default_dof_drive_mode=3, self_collisions=0, replace_cylinder_with_capsule=True,
                  flip_visual_attachments=True, density=0.001, angular_damping=0.0, linear_damping=0.0,
                  max_angular_velocity=1000.0, max_linear_velocity=1000.0, armature=0.0, thickness=0.01),
    "domain_rand": dict(randomize_friction=True, friction_range=[0.5, 1.25], randomize_base_mass=False,
                        added_mass_range=[-1.0, 1.0], push_robots=True, push_interval_s=15,
                        max_push_vel_xy=1.0),
    "rewards": dict(
        scales=dict(termination=-0.0, tracking_lin_vel=1.0, tracking_ang_vel=0.5, lin_vel_z=-2.0,
                    ang_vel_xy=-0.05, orientation=-0.0, torques=-0.00001, dof_vel=-0.0, dof_acc=-2.5e-7,
                    base_height=-0.0, feet_air_time=1.0, collision=-1.0, feet_stumble=-0.0,
                    action_rate=-0.01, stand_still=-0.0),
        only_positive_rewards=True, tracking_sigma=0.25, soft_dof_pos_limit=1.0, soft_dof_vel_limit=1.0,
        soft_torque_limit=1.0, base_height_target=1.0, max_contact_force=100.0),
    "normalization": dict(obs_scales=dict(lin_vel=2.0, ang_vel=0.25, dof_pos=1.0, dof_vel=0.05,
                                          height_measurements=5.0),
                          clip_observations=100.0, clip_actions=100.0),
    "noise": dict(add_noise=True, noise_level=1.0,
                  noise_scales=dict(dof_pos=0.01, dof_vel=1.5, lin_vel=0.1, ang_vel=0.2, gravity=0.05,
                                    height_measurements=0.1)),
    "viewer": dict(ref_env=0, pos=[10, 0, 6], lookat=[11.0, 5, 3.0]),
    "sim": dict(dt=0.005, substeps=1, gravity=[0.0, 0.0, -9.81], up_axis=1,
                physx=dict(num_threads=10, solver_type=1, num_position_iterations=4,
                           num_velocity_iterations=0, contact_offset=0.01, rest_offset=0.0,
                           bounce_threshold_velocity=0.5, max_depenetration_velocity=1.0,
                           max_gpu_contact_pairs=2 ** 23, default_buffer_size_multiplier=5,
                           contact_collection=2)),
})

LeggedRobotCfgPPO = make_cfg("LeggedRobotCfgPPO", BaseConfig, {
    "seed": 1,
    "runner_class_name": "OnPolicyRunner",
    "policy": dict(init_noise_std=1.0, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128],
                   activation="elu"),
    "algorithm": dict(value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.01,
                      num_learning_epochs=5, num_mini_batches=4, learning_rate=5e-4, schedule="adaptive",
                      gamma=0.99, lam=0.95, desired_kl=0.01, max_grad_norm=1.0),
    "runner": dict(policy_class_name="ActorCritic", algorithm_class_name="PPO", num_steps_per_env=24,
                   max_iterations=20000, save_interval=500, experiment_name="test", run_name="",
                   resume=False, load_run=-1, checkpoint=-1, resume_path=None),
})

# --------------------------------------------------------------------------------------
# WidowGo1RoughCfg / WidowGo1RoughCfgPPO: widowGo1_config.py:35-382
# --------------------------------------------------------------------------------------
RESUME = True   # widowGo1_config.py:35 (hard-coded True upstream; rewrites two schedules below)

_FINAL_L, _FINAL_P, _FINAL_Y = [0.2, 0.7], [-2 * PI / 5, 1 * PI / 5], [-3 * PI / 5, 3 * PI / 5]
_BOX = 0.1
_LEG = [0.4, 0.45, 0.45]
_OSC_KP = [100.0, 100.0, 100.0, 30.0, 30.0, 30.0]
_TOT_COLS, _TOT_ROWS, _HSCALE = 600, 10000, 0.025

_STANCE = {}
for _leg, _hip in (("FL", 0.1), ("RL", 0.1), ("FR", -0.1), ("RR", -0.1)):
    _STANCE[f"{_leg}_hip_joint"] = _hip
    _STANCE[f"{_leg}_thigh_joint"] = 0.8
    _STANCE[f"{_leg}_calf_joint"] = -1.5
for _j in ("waist", "shoulder", "elbow", "wrist_angle", "forearm_roll", "wrist_rotate", "left_finger",
           "right_finger"):
    _STANCE[f"widow_{_j}"] = 0

WidowGo1RoughCfg = make_cfg("WidowGo1RoughCfg", LeggedRobotCfg, {
    "goal_ee": dict(
        num_commands=3, traj_time=[1, 3], hold_time=[0.5, 2],
        collision_upper_limits=[0.3, 0.15, 0.05 - 0.165], collision_lower_limits=[-0.2, -0.15, -0.35 - 0.165],
        underground_limit=-0.57, num_collision_check_samples=10, command_mode="sphere",
        l_schedule=[0, 1], p_schedule=[0, 1], y_schedule=[0, 1], tracking_ee_reward_schedule=[0, 1],
        ranges=dict(final_pos_l=_FINAL_L, final_pos_p=_FINAL_P, final_pos_y=_FINAL_Y,
                    init_pos_l=[0.6, 0.6], init_pos_p=[PI / 4, PI / 4], init_pos_y=[-PI / 6, PI / 6],
                    final_delta_orn=[[-0, 0], [-0, 0], [-0, 0]], final_tracking_ee_reward=0.55),
        sphere_error_scale=[1 / (_FINAL_L[1] - _FINAL_L[0]), 1 / (_FINAL_P[1] - _FINAL_P[0]),
                            1 / (_FINAL_Y[1] - _FINAL_Y[0])],
        orn_error_scale=[2 / PI, 2 / PI, 2 / PI],
        init_ranges=dict(pos_l=[0.3, 0.5], pos_p=[PI / 4, 3 * PI / 4], pos_y=[0, 0])),
    "commands": dict(
        curriculum=True, num_commands=3, resampling_time=3.0,
        lin_vel_x_schedule=[0, 1], ang_vel_yaw_schedule=[0, 1], tracking_ang_vel_yaw_schedule=[0, 1],
        ang_vel_yaw_clip=0.6, lin_vel_x_clip=0.3,
        ranges=dict(final_lin_vel_x=[0, 0.9], final_ang_vel_yaw=[-1.0, 1.0], init_lin_vel_x=[0, 0],
                    init_ang_vel_yaw=[0, 0], final_tracking_ang_vel_yaw_exp=0.15)),
    "normalization": dict(obs_scales=dict(lin_vel=1.0, ang_vel=1.0, dof_pos=1.0, dof_vel=0.05,
                                          height_measurements=5.0),
                          clip_observations=100.0, clip_actions=100.0),
    "env": dict(num_envs=5000, num_actions=12 + 6, num_torques=12 + 6, action_delay=2,
                num_proprio=2 + 3 + 20 + 20 + 18 + 4 + 3 + 3 + 3, num_priv=5 + 1 + 18, history_len=10,
                num_observations=76 * (10 + 1) + 24, num_privileged_obs=None, send_timeouts=True,
                episode_length_s=10, reorder_dofs=True),
    "init_state": dict(pos=[0.0, 0.0, 0.42], default_joint_angles=leaf(_STANCE)),
    "control": dict(stiffness=leaf({"joint": 50, "widow": 5}), damping=leaf({"joint": 1, "widow": 0.5}),
                    adaptive_arm_gains=False, action_scale=_LEG * 4 + [2.1, 0.6, 0.6, 0, 0, 0],
                    decimation=4, torque_supervision=False),
    "asset": dict(file="{LEGGED_GYM_ROOT_DIR}/resources/robots/widowGo1/urdf/widowGo1.urdf",
                  foot_name="foot", penalize_contacts_on=["thigh", "trunk"], terminate_after_contacts_on=[],
                  self_collisions=0, flip_visual_attachments=False, collapse_fixed_joints=True,
                  fix_base_link=False),
    "box": dict(box_size=_BOX, randomize_base_mass=True, added_mass_range=[-0.001, 0.050],
                box_env_origins_x=0, box_env_origins_y_range=[0.1, 0.3], box_env_origins_z=_BOX / 2 + 0.16,
                box_pos_obs_range=1.0),
    "arm": dict(init_target_ee_base=[0.2, 0.0, 0.2], grasp_offset=0.08, osc_kp=_OSC_KP,
                osc_kd=[2 * math.sqrt(k) for k in _OSC_KP]),
    "domain_rand": dict(observe_priv=True, randomize_friction=True, friction_range=[-0.5, 3.0],
                        randomize_base_mass=True, added_mass_range=[-0.5, 2.5], randomize_base_com=True,
                        added_com_range_x=[-0.15, 0.15], added_com_range_y=[-0.15, 0.15],
                        added_com_range_z=[-0.15, 0.15], randomize_motor=True,
                        leg_motor_strength_range=[0.7, 1.3], arm_motor_strength_range=[0.7, 1.3],
                        randomize_gripper_mass=True, gripper_added_mass_range=[0, 0.1],
                        push_robots=True, push_interval_s=3, max_push_vel_xy=0.5, cube_y_range=[0.2, 0.4]),
    "noise": dict(add_noise=False),
    "rewards": dict(
        scales=dict(termination=-0, tracking_lin_vel=0.0, tracking_ang_vel=0.0, lin_vel_z=-0.0, ang_vel_xy=-0.0,
                    orientation=-0.0, torques=0, energy_square=-6e-5, dof_vel=0, dof_acc=-0, base_height=0,
                    feet_air_time=0, collision=0, feet_stumble=-0, action_rate=-0, stand_still=0, survive=0.2,
                    leg_energy=-0, leg_energy_abs_sum=-0, tracking_lin_vel_x_l1=0.5, tracking_lin_vel_x_exp=0.0,
                    tracking_ang_vel_yaw_l1=0, tracking_ang_vel_yaw_exp=0.15, tracking_lin_vel_y_l2=0,
                    tracking_lin_vel_z_l2=-0.0, leg_action_l2=-0.0, hip_action_l2=-0.01, foot_contacts_z=-1e-4),
        arm_scales=dict(termination=-0.0, tracking_ee_sphere=0.55, tracking_ee_cart=0.0, arm_orientation=-0.0,
                        arm_energy_abs_sum=-0.0040, tracking_ee_orn=0.0, tracking_ee_orn_ry=0.0),
        only_positive_rewards=False, tracking_sigma=1, tracking_ee_sigma=1, soft_dof_pos_limit=1.0,
        soft_dof_vel_limit=1.0, soft_torque_limit=1.0, base_height_target=0.25, max_contact_force=100.0),
    "viewer": dict(pos=[-20, 0, 20], lookat=[0, 0, -2]),
    "termination": dict(r_threshold=0.78, p_threshold=0.60, z_threshold=0.325),
    "terrain": dict(mesh_type="trimesh", add_slopes=True, slope_incline=0.2, horizontal_scale=_HSCALE,
                    vertical_scale=1 / 100000, border_size=0, tot_cols=_TOT_COLS, tot_rows=_TOT_ROWS, zScale=0.15,
                    transform_x=-_TOT_COLS * _HSCALE / 2, transform_y=-_TOT_ROWS * _HSCALE / 2, transform_z=0.0,
                    curriculum=False, static_friction=1.0, dynamic_friction=1.0, restitution=0.0,
                    measure_heights=False, measured_points_x=_GRID_X, measured_points_y=_GRID_Y,
                    slope_treshold=100000000, origin_perturb_range=0.5, init_vel_perturb_range=0.1),
})

WidowGo1RoughCfgPPO = make_cfg("WidowGo1RoughCfgPPO", LeggedRobotCfgPPO, {
    "seed": 1,
    "runner_class_name": "OnPolicyRunner",
    "policy": dict(init_std=[[0.8, 1.0, 1.0] * 4 + [1.0] * 6], actor_hidden_dims=[128], critic_hidden_dims=[128],
                   activation="elu", leg_control_head_hidden_dims=[128, 128],
                   arm_control_head_hidden_dims=[128, 128], priv_encoder_dims=[64, 20], num_leg_actions=12,
                   num_arm_actions=6, adaptive_arm_gains=False, adaptive_arm_gains_scale=10.0),
    "algorithm": dict(value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.0,
                      num_learning_epochs=5, num_mini_batches=4, learning_rate=2e-4, schedule="fixed", gamma=0.99,
                      lam=0.95, desired_kl=None, max_grad_norm=1.0,
                      min_policy_std=[[0.15, 0.25, 0.25] * 4 + [0.2] * 3 + [0.05] * 3],
                      mixing_schedule=[1.0, 0, 3000] if not RESUME else [1.0, 0, 1],
                      torque_supervision=False, torque_supervision_schedule=[0.0, 1000, 1000],
                      adaptive_arm_gains=False, dagger_update_freq=20,
                      priv_reg_coef_schedual=[0, 0.1, 3000, 7000] if not RESUME else [0, 1, 1000, 1000]),
    "runner": dict(policy_class_name="ActorCritic", algorithm_class_name="PPO", num_steps_per_env=40,
                   max_iterations=40000, save_interval=500, experiment_name="rough_widowGo1", run_name="",
                   resume=RESUME, load_run=-1, checkpoint=-1, resume_path=None),
})


def class_to_dict(obj) -> dict:
    """legged_gym/utils/helpers.py:41-56 semantics: nested config object -> plain dict."""
    if not hasattr(obj, "__dict__") and not inspect.isclass(obj):
        return obj
    result = {}
    for key in dir(obj):
        if key.startswith("_") or key == "init_member_classes":
            continue
        val = getattr(obj, key)
        if callable(val) and not inspect.isclass(val):
            continue
        if isinstance(val, list):
            result[key] = [class_to_dict(item) for item in val]
        elif inspect.isclass(val) or hasattr(val, "__dict__"):
            result[key] = class_to_dict(val)
        else:
            result[key] = val
    return result


def use_grid_terrain(cfg, num_rows=10, num_cols=20):
    """Swap a widowGo1 config's terrain for the base class's sub-terrain grid, LeggedRobotCfg.terrain
    (legged_robot_config.py:43-66: trimesh, 0.1 m / 5 mm scales, 25 m border, num_rows difficulty levels x num_cols
    types of 8 m x 8 m tiles with proportions [.1, .1, .35, .25, .2], terrain-level curriculum on) -- BASELINE.json
    configs[2] / SURVEY.md section 8d config 3. The task-specific reset ranges of the widowGo1 terrain block are kept;
    measure_heights is off (the widowGo1 observation has no height scan, WG:966-1001)."""
    old = cfg.terrain
    new = LeggedRobotCfg().terrain
    new.num_rows, new.num_cols = num_rows, num_cols
    new.measure_heights = False
    for k in ("origin_perturb_range", "init_vel_perturb_range"):
        setattr(new, k, getattr(old, k))
    cfg.terrain = new
    return cfg
