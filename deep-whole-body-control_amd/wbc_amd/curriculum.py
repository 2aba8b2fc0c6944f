"""update_command_curriculum of the task (reference legged_gym/envs/widowGo1/widowGo1.py:675-692):
host scalars -> wbc_curriculum (command ranges, EE-goal ranges, the two scheduled reward scales and
the full reward-scale tables with zero = inactive, WG:128-131,145-148)."""
from __future__ import annotations

import numpy as np

from . import abi

# reward names the reference's config knows that cannot be offered, each with what the REFERENCE does when its scale is non-zero
# (recorded from its own code by tools/check_reference_dead_switches.py, profiles/r04_reference_switches.txt): a non-zero scale is
# an error here too, not a silent drop
UNIMPLEMENTED_REWARDS = {
    "orientation": "legged_robot.py:841-843 reads self.projected_gravity, which WidowGo1._init_buffers never creates (WG:636,885 commented out): "
                   "AttributeError on the first compute_reward",
    "arm_orientation": "no _reward_arm_orientation exists (WG:1417-1420 commented out): AttributeError in _prepare_reward_function",
    "feet_stumble": "the config key names no method (the method is _reward_stumble: use the scale name 'stumble'): AttributeError in "
                    "_prepare_reward_function",
}


def _scales(obj) -> dict:
    return {k: float(getattr(obj, k)) for k in dir(obj)
            if not k.startswith("_") and isinstance(getattr(obj, k), (int, float)) and not isinstance(getattr(obj, k), bool)}


def curriculum_value(schedule, init, final, counter):
    """_get_curriculum_value, WG:675-676."""
    init, final = np.asarray(init, dtype=np.float64), np.asarray(final, dtype=np.float64)
    return np.clip((counter - schedule[0]) / (schedule[1] - schedule[0]), 0, 1) * (final - init) + init


def make_curriculum(cfg, update_counter: int) -> abi.WbcCurriculum:
    cur = abi.WbcCurriculum()
    cmd, cr, g, ge = cfg.commands, cfg.commands.ranges, cfg.goal_ee, cfg.goal_ee.ranges
    abi._set(cur.lin_vel_x_range, curriculum_value(cmd.lin_vel_x_schedule, cr.init_lin_vel_x, cr.final_lin_vel_x, update_counter))
    abi._set(cur.ang_vel_yaw_range, curriculum_value(cmd.ang_vel_yaw_schedule, cr.init_ang_vel_yaw, cr.final_ang_vel_yaw, update_counter))
    abi._set(cur.goal_l_range, curriculum_value(g.l_schedule, ge.init_pos_l, ge.final_pos_l, update_counter))
    abi._set(cur.goal_p_range, curriculum_value(g.p_schedule, ge.init_pos_p, ge.final_pos_p, update_counter))
    abi._set(cur.goal_y_range, curriculum_value(g.y_schedule, ge.init_pos_y, ge.final_pos_y, update_counter))
    leg, arm = _scales(cfg.rewards.scales), _scales(cfg.rewards.arm_scales)
    for table in (leg, arm):
        for name, val in table.items():
            if val != 0 and name in UNIMPLEMENTED_REWARDS:
                raise NotImplementedError(f"reward term '{name}' has a non-zero scale; the reference cannot run it either: {UNIMPLEMENTED_REWARDS[name]}")
            if val != 0 and name not in abi.REWARD_TERMS:
                raise NotImplementedError(f"reward term '{name}' has a non-zero scale but no such _reward_ method exists in the reference's widowGo1 task")
            if val != 0 and name == "base_height" and bool(cfg.terrain.measure_heights):
                raise NotImplementedError("reward term 'base_height' with terrain.measure_heights = True needs the measured heights inside the fused "
                                          "step (legged_robot.py:845-848); with measure_heights = False (measured_heights = 0, WG:639) it is implemented")
    if leg.get("feet_air_time", 0) != 0 and arm.get("feet_air_time", 0) != 0:
        raise NotImplementedError("reward term 'feet_air_time' has a non-zero scale in BOTH rewards.scales and rewards.arm_scales: the reference's "
                                  "_reward_feet_air_time mutates feet_air_time / last_contacts on every call (legged_robot.py:898-909), so listing it twice "
                                  "advances that state twice per step; the fused step advances it once -- keep it in one list")
    # the reward-function lists are fixed at construction from the config's non-zero scales (WG:128-157)
    leg_active = {name for name, val in leg.items() if val != 0}
    arm_active = {name for name, val in arm.items() if val != 0}
    if update_counter >= 1:     # only update_command_curriculum overwrites the two scheduled scales (WG:683,689-692); before its
        # first call the object still carries the config's values
        leg["tracking_ang_vel_yaw_exp"] = float(curriculum_value(cmd.tracking_ang_vel_yaw_schedule, 0, cr.final_tracking_ang_vel_yaw_exp, update_counter))
        key = "tracking_ee_sphere" if "tracking_ee_sphere" in arm_active else "tracking_ee_cart"     # WG:689-692
        arm[key] = float(curriculum_value(g.tracking_ee_reward_schedule, 0, ge.final_tracking_ee_reward, update_counter))
    lm = am = 0
    for i, name in enumerate(abi.REWARD_TERMS):
        on_l, on_a = name in leg_active, name in arm_active
        cur.leg_reward_scale[i] = leg.get(name, 0.0) if on_l else 0.0
        cur.arm_reward_scale[i] = arm.get(name, 0.0) if on_a else 0.0
        lm |= int(on_l) << i
        am |= int(on_a) << i
    cur.leg_active_mask, cur.arm_active_mask = lm, am
    return cur
