"""What happens to every field of `WidowGo1RoughCfg` (legged_gym/envs/widowGo1/widowGo1_config.py:37-340 over
legged_gym/envs/base/legged_robot_config.py:33-199): each leaf of the config tree is in exactly ONE class --

  kernel     it changes the bytes handed to the kernels (wbc_task_cfg / wbc_model / wbc_curriculum: abi.fill_task_cfg,
             abi.fill_model, curriculum.make_curriculum);
  host       host code consumes it at construction or per step (the file is named: the domain-randomisation draws of
             envs.draw_env_params, the terrain generators, WidowGo1's own switches);
  constant   compiled into the kernels; any other value raises (abi.KERNEL_CONSTANTS);
  refused    the reference reads it and this framework does not implement the non-default value: NotImplementedError
             (abi.UNSUPPORTED_SWITCHES);
  no_effect  the reference's widowGo1 path drops it as well, or it sizes a PhysX buffer (abi.REFERENCE_NO_OPS, with the line).

tests/test_host_logic.py walks every leaf of the config (and of the golden flattening of the REFERENCE's own class) and checks the
class's behaviour by flipping the field: nothing the reference reads is silently ignored."""
from __future__ import annotations

from . import abi

KERNEL, HOST, CONSTANT, REFUSED, NO_EFFECT = "kernel", "host", "constant", "refused", "no_effect"

_KERNEL = """
asset.armature asset.foot_name asset.penalize_contacts_on asset.self_collisions asset.terminate_after_contacts_on
box.box_env_origins_x box.box_env_origins_z box.box_size
commands.ang_vel_yaw_clip commands.ang_vel_yaw_schedule commands.lin_vel_x_clip commands.lin_vel_x_schedule
commands.ranges.final_ang_vel_yaw commands.ranges.final_lin_vel_x commands.ranges.final_tracking_ang_vel_yaw_exp
commands.ranges.init_ang_vel_yaw commands.ranges.init_lin_vel_x commands.resampling_time commands.tracking_ang_vel_yaw_schedule
control.action_scale control.damping control.decimation control.stiffness
domain_rand.max_push_vel_xy domain_rand.push_interval_s domain_rand.push_robots
env.action_delay env.episode_length_s
goal_ee.collision_lower_limits goal_ee.collision_upper_limits goal_ee.command_mode goal_ee.l_schedule goal_ee.num_collision_check_samples
goal_ee.orn_error_scale goal_ee.p_schedule goal_ee.ranges.final_delta_orn goal_ee.ranges.final_pos_l goal_ee.ranges.final_pos_p
goal_ee.ranges.final_pos_y goal_ee.ranges.final_tracking_ee_reward goal_ee.ranges.init_pos_l goal_ee.ranges.init_pos_p
goal_ee.ranges.init_pos_y goal_ee.sphere_error_scale goal_ee.tracking_ee_reward_schedule goal_ee.underground_limit goal_ee.y_schedule
init_state.ang_vel init_state.default_joint_angles init_state.lin_vel init_state.pos init_state.rot
normalization.clip_actions normalization.clip_observations normalization.obs_scales.ang_vel normalization.obs_scales.dof_pos
normalization.obs_scales.dof_vel normalization.obs_scales.lin_vel
rewards.base_height_target rewards.max_contact_force rewards.only_positive_rewards rewards.soft_dof_pos_limit
rewards.soft_dof_vel_limit rewards.soft_torque_limit rewards.tracking_ee_sigma rewards.tracking_sigma
sim.dt sim.gravity sim.physx.contact_offset sim.physx.max_depenetration_velocity sim.physx.num_position_iterations sim.physx.rest_offset
termination.z_threshold terrain.init_vel_perturb_range terrain.origin_perturb_range terrain.static_friction terrain.dynamic_friction
"""
# host-side consumers: (path, file that reads it)
_HOST = {
    "envs.py": """
        arm.osc_kd arm.osc_kp asset.file box.added_mass_range box.box_env_origins_y_range box.randomize_base_mass control.torque_supervision
        domain_rand.added_com_range_x domain_rand.added_com_range_y domain_rand.added_com_range_z domain_rand.added_mass_range
        domain_rand.arm_motor_strength_range domain_rand.friction_range domain_rand.gripper_added_mass_range
        domain_rand.leg_motor_strength_range domain_rand.randomize_base_com domain_rand.randomize_base_mass domain_rand.randomize_friction
        domain_rand.randomize_gripper_mass domain_rand.randomize_motor env.num_envs env.reference_stale_time_outs env.send_timeouts
        goal_ee.hold_time goal_ee.traj_time terrain.curriculum terrain.max_init_terrain_level terrain.measure_heights
        terrain.measured_points_x terrain.measured_points_y terrain.mesh_type terrain.num_cols terrain.num_rows""",
    "terrain.py": """
        terrain.border_size terrain.horizontal_scale terrain.terrain_length
        terrain.terrain_proportions terrain.terrain_width terrain.tot_cols terrain.tot_rows terrain.transform_x terrain.transform_y
        terrain.transform_z terrain.vertical_scale terrain.zScale""",
}


def _build():
    table = {}

    def put(path, cls, detail):
        assert path not in table, f"{path} classified twice ({table[path][0]}, {cls})"
        table[path] = (cls, detail)
    for p in _KERNEL.split():
        put(p, KERNEL, "abi.fill_task_cfg / abi.fill_model / curriculum.make_curriculum")
    for fn, block in _HOST.items():
        for p in block.split():
            put(p, HOST, fn)
    for p in abi.KERNEL_CONSTANTS:
        put(p, CONSTANT, f"= {abi.KERNEL_CONSTANTS[p]}")
    for p, _, why in abi.UNSUPPORTED_SWITCHES:
        put(p, REFUSED, why)
    for p, why in abi.REFERENCE_NO_OPS:
        put(p, NO_EFFECT, why)
    return table


FIELDS = _build()


def field_class(path: str):
    """(class, detail) of a config leaf; reward scales (`rewards.scales.<term>`, `rewards.arm_scales.<term>`) are kernel fields:
    the term's entry of wbc_curriculum (a term this framework cannot run raises in make_curriculum)."""
    if path.startswith("rewards.scales.") or path.startswith("rewards.arm_scales."):
        return KERNEL, "curriculum.make_curriculum (the reward-scale tables of wbc_curriculum)"
    return FIELDS.get(path)


def flatten(cfg, prefix=""):
    """{dotted path: value} over the leaves of a config object (dict-valued leaves such as control.stiffness stay whole)."""
    from .config import class_to_dict
    d = cfg if isinstance(cfg, dict) else class_to_dict(cfg)
    out = {}
    for k, v in d.items():
        if isinstance(v, dict) and k not in ("default_joint_angles", "stiffness", "damping"):
            out.update(flatten(v, prefix + k + "."))
        else:
            out[prefix + k] = v
    return out


def unclassified(cfg):
    """Leaves of `cfg` this table does not know -- fields a user's subclass added for its own code (nothing here reads them)."""
    return sorted(p for p in flatten(cfg) if field_class(p) is None)
