#!/usr/bin/env python3
"""Generate tests/golden/wg_reference_*.npz by running the REFERENCE's own WidowGo1 class
(/root/reference/legged_gym/legged_gym/envs/widowGo1/widowGo1.py, imported read-only) on the fake
Isaac Gym of tools/ref_harness/igstub.py, whose `gym.simulate` is this framework's physics spec.

Every fixture is a trajectory of the reference's `step()`: the state before the first step, the
actions of every step, and the reference's complete state after every step, all under the tensor
names of include/wbc_sim.h. tests/test_wg_golden.py replays each step from the recorded pre-state
through the C oracle (CPU) and through the HIP step kernel (GPU) and compares with what the reference
computed: torques, observations + history, both reward channels, episode / metric sums, commands,
EE-goal state machine, reset / time-out masks (bit-exact), resets and their draws.

Run in the build container only: python tools/make_golden_wg.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_harness"))
import numpy as np
import torch

import harness
from harness import curriculum_struct, make_reference_env, reference_cfg, wbc_state

GOLD = os.path.join(HERE, "..", "tests", "golden")

INPUT_NAMES = ["ROOT_STATES", "DOF_STATE", "TORQUES", "OBS_HISTORY", "ACTION_HISTORY", "ACTIONS", "LAST_ACTIONS", "LAST_DOF_VEL",
               "LAST_ROOT_VEL", "COMMANDS", "GOAL_STATE", "EPISODE_LENGTH", "EPISODE_SUMS", "METRIC_SUMS", "FORCE_SENSOR",
               "NET_CONTACT_FORCE", "RIGID_BODY_STATE", "BASE_LIN_VEL", "BASE_ANG_VEL", "TIME_OUT_BUF", "RESET_BUF", "BOX_SLEEP_TIMER", "FEET_AIR_TIME", "LAST_CONTACTS"]
OUTPUT_NAMES = INPUT_NAMES + ["OBS_BUF", "REW_BUF", "ARM_REW_BUF"]


def cur_arrays(env):
    c = curriculum_struct(env)
    return np.array(list(c.lin_vel_x_range) + list(c.ang_vel_yaw_range) + list(c.goal_l_range) + list(c.goal_p_range) +
                    list(c.goal_y_range) + list(c.leg_reward_scale) + list(c.arm_reward_scale) +
                    [c.leg_active_mask, c.arm_active_mask], dtype=np.float64)


def episode_extras(env):
    from wbc_amd import abi
    ep = env.extras.get("episode", {})
    out = np.full(abi.NREW + abi.NMETRIC, np.nan, dtype=np.float64)
    for i, name in enumerate(abi.REWARD_TERMS):
        if "rew_" + name in ep:
            out[i] = float(ep["rew_" + name])
    for i, name in enumerate(abi.METRIC_NAMES):
        if "metric_" + name in ep:
            out[abi.NREW + i] = float(ep["metric_" + name])
    return out


def record_trajectory(env, steps, rng, tag, big_action_envs=()):
    """Step the reference `steps` times; before every step the physics backend is re-seated on the fp32 tensors the
    reference holds (so a replay from the recorded fp32 pre-state starts from the identical physics state)."""
    n = env.num_envs
    b = env._backend
    out = {"init/" + k: v for k, v in wbc_state(env).items() if k in INPUT_NAMES}
    out["step_counter0"] = np.int64(env.common_step_counter)
    acts, curs, eps = [], [], []
    for k in range(steps):
        b.push("root")
        b.push("dof")
        a = 0.5 * np.tanh(rng.standard_normal((n, 18))).astype(np.float32)
        for e in big_action_envs:
            a[e] = rng.choice([-150.0, 150.0, 3.0, -3.0], 18).astype(np.float32)      # exercises the +-100 clip (WG:1163)
        curs.append(cur_arrays(env))
        env.step(torch.from_numpy(a))
        acts.append(a)
        st = wbc_state(env)
        for name in OUTPUT_NAMES:
            out[f"s{k}/{name}"] = st[name]
        eps.append(episode_extras(env))
        out[f"s{k}/time_outs"] = env.extras["time_outs"].numpy().astype(np.uint8)
    out["actions"] = np.stack(acts)
    out["curriculum"] = np.stack(curs)
    out["episode_extras"] = np.stack(eps)
    out["steps"] = np.int64(steps)
    for k, v in env._env_params.items():
        out["param/" + k] = np.asarray(v)
    out["seed"] = np.int64(harness.CTX.seed)
    tc = env._backend.tcfg                                      # the task constants that differ between the fixtures
    out["tcfg/term_z_threshold"] = np.float64(tc.term_z_threshold)
    out["tcfg/term_contact_rb_mask"] = np.int64(tc.term_contact_rb_mask)
    out["tcfg/penalize_contact_rb_mask"] = np.int64(tc.penalize_contact_rb_mask)
    resets = sum(int(out[f"s{k}/RESET_BUF"].sum()) for k in range(steps))
    tos = sum(int(out[f"s{k}/TIME_OUT_BUF"].sum()) for k in range(steps))
    print(f"[{tag}] {steps} steps x {n} envs: {resets} resets ({tos} time-outs), draws logged: {len(harness.CTX.log)}")
    return out


def save(name, out):
    path = os.path.join(GOLD, name)
    np.savez_compressed(path, **out)
    print("wrote", path, f"{os.path.getsize(path) / 1e6:.2f} MB")


def flat_cfg():
    cfg = reference_cfg()
    cfg.terrain.tot_rows = 400                      # the reference's Terrain_Perlin is still built (host, init-time); physics
    cfg.terrain.transform_y = -400 * cfg.terrain.horizontal_scale / 2     # below runs on the flat plane of BASELINE configs[1]
    return cfg


def stage_events(env, rng):
    """Put envs next to every event the step can trigger: time-outs (episode_length > 500), command resampling
    (episode_length % 150 == 0), a global push (common_step_counter % 150 == 0), EE-goal resampling."""
    n = env.num_envs
    L = np.array([0, 1, 148, 149, 298, 299, 448, 497, 498, 499, 500, 501, 30, 77] + list(rng.integers(2, 480, max(n - 14, 0))))[:n]
    env.episode_length_buf[:] = torch.from_numpy(L)
    env.common_step_counter = 146
    env.goal_timer[:] = torch.from_numpy(rng.integers(0, 60, n).astype(np.float32))
    soon = rng.random(n) < 0.4
    env.goal_timer[torch.from_numpy(soon)] = (env.traj_total_timesteps[torch.from_numpy(soon)] - torch.from_numpy(rng.integers(0, 6, int(soon.sum())).astype(np.float32)))


def self_contact_arm_poses(backend, count, rng):
    """Arm joint angles (6) at which an arm sphere touches the trunk box or a front thigh: found by holding robots in the air
    (no terrain contact possible) in a scratch OracleSim and keeping the poses whose first substep reports a contact force."""
    from oracle import OracleSim
    m = backend.model
    n = 1024
    o = OracleSim(backend.wmodel, backend.tcfg, n, seed=1, precision="f64")
    root = np.zeros((n, 2, 13)); root[:, :, 6] = 1; root[:, 0, 2] = 1.0
    dof = np.zeros((n, 20, 2)); dof[:, :, 0] = np.array(backend.tcfg.default_dof_pos)[None]
    lo, hi = np.array(m.dof_lower[12:18]), np.array(m.dof_upper[12:18])
    lo[0], hi[0] = -1.5, 1.5                                       # the waist has no URDF limit
    dof[:, 12:18, 0] = rng.uniform(lo, hi, (n, 6))
    o.set("ROOT_STATES", root); o.set("DOF_STATE", dof); o.set("TORQUES", np.zeros((n, 20)))
    o.simulate()
    f = o.get("NET_CONTACT_FORCE")[:, :27]                         # (row 27: the box actor, resting on the ground below)
    hit = np.abs(f).sum((1, 2)) > 0
    thigh = np.abs(f[:, [3, 7]]).sum((1, 2)) > 0                   # FL_thigh, FR_thigh rows: a gripper / wrist against a thigh
    picks = list(np.nonzero(thigh)[0][:max(1, count // 3)]) + list(np.nonzero(hit & ~thigh)[0])
    assert len(picks) >= count, (hit.sum(), thigh.sum())
    return dof[picks[:count], 12:18, 0]


def main():
    os.makedirs(GOLD, exist_ok=True)
    rng = np.random.default_rng(20260925)
    # ---- A: shipped config, counter 0 (the reset() the runner issues before its first curriculum update) then counter 1
    env = make_reference_env(24, seed=7, cfg=flat_cfg(), heightfield=None)
    env._backend.ora.set_heightfield(None, 0, 0, 0, 0, 0)
    with torch.inference_mode():
        env.reset()                                                           # BT:127-131: reset_idx(all, start=True) + zero-action step
        out0 = record_trajectory(env, 3, rng, "A0 counter=0")
        env.update_command_curriculum()                                       # OPR:126
        stage_events(env, rng)
        outA = record_trajectory(env, 28, rng, "A1 counter=1", big_action_envs=(3, 11))
    save("wg_reference_counter0.npz", out0)
    save("wg_reference_default.npz", outA)
    # ---- B: every one of the 21 reward terms of WG:1352-1469 active, and the base class's _reward_collision (LR:865-867)
    cfg = flat_cfg()
    s, a = cfg.rewards.scales, cfg.rewards.arm_scales
    leg = dict(energy_square=-6e-5, survive=0.2, tracking_lin_vel_x_l1=0.5, tracking_ang_vel_yaw_exp=0.15, hip_action_l2=-0.01,
               foot_contacts_z=-1e-4, leg_energy_abs_sum=-3e-3, leg_energy_sum_abs=-2e-3, leg_action_l2=-0.02, leg_energy=-1e-3,
               tracking_lin_vel=0.3, tracking_lin_vel_x_exp=0.25, tracking_ang_vel_yaw_l1=0.1, tracking_lin_vel_y_l2=-0.4,
               tracking_lin_vel_z_l2=-0.2, torques=-1e-4, collision=-0.7,
               # the base class's terms that work in the widowGo1 task (legged_robot.py:832-922)
               lin_vel_z=-0.5, ang_vel_xy=-0.05, dof_vel=-1e-4, dof_acc=-1e-7, action_rate=-0.01, termination=-5.0, dof_pos_limits=-1.0,
               dof_vel_limits=-0.5, torque_limits=-0.01, tracking_ang_vel=0.2, feet_air_time=1.0, stumble=-0.3, stand_still=-0.1,
               feet_contact_forces=-0.01, base_height=-2.0)
    arm = dict(tracking_ee_sphere=0.55, arm_energy_abs_sum=-0.004, tracking_ee_cart=0.35, tracking_ee_orn=0.2, tracking_ee_orn_ry=0.15,
               termination=-1.0)
    cfg.rewards.soft_dof_pos_limit, cfg.rewards.soft_dof_vel_limit, cfg.rewards.soft_torque_limit = 0.9, 0.1, 0.3     # so that the limit terms bite
    cfg.rewards.max_contact_force, cfg.rewards.base_height_target = 20.0, 0.3
    for k, v in leg.items():
        setattr(s, k, v)                # instance attributes: class_to_dict (helpers.py:41-56) walks dir(obj)
    for k, v in arm.items():
        setattr(a, k, v)
    cfg.goal_ee.ranges.final_delta_orn = [[-0.3, 0.3], [-0.2, 0.4], [-0.5, 0.5]]   # non-degenerate orientation goals
    env = make_reference_env(16, seed=11, cfg=cfg)
    env._backend.ora.set_heightfield(None, 0, 0, 0, 0, 0)
    with torch.inference_mode():
        env.reset()
        env.update_command_curriculum()
        stage_events(env, rng)
        outB = record_trajectory(env, 10, rng, "B all rewards")
    outB["leg_scales"] = np.array([leg.get(nm, 0.0) for nm in __import__("wbc_amd").abi.REWARD_TERMS])
    outB["arm_scales"] = np.array([arm.get(nm, 0.0) for nm in __import__("wbc_amd").abi.REWARD_TERMS])
    outB["delta_orn"] = np.array(cfg.goal_ee.ranges.final_delta_orn)
    outB["soft_limits"] = np.array([cfg.rewards.soft_dof_pos_limit, cfg.rewards.soft_dof_vel_limit, cfg.rewards.soft_torque_limit,
                                    cfg.rewards.max_contact_force, cfg.rewards.base_height_target])
    base_terms = [i for i, nm in enumerate(__import__("wbc_amd").abi.REWARD_TERMS) if i >= 22]
    sums = np.stack([outB[f"s{k}/EPISODE_SUMS"] for k in range(10)])
    print("   base-class terms with a non-zero episode sum somewhere:", [__import__("wbc_amd").abi.REWARD_TERMS[i] for i in base_terms if np.abs(sums[:, :, i]).max() > 0])
    save("wg_reference_allrewards.npz", outB)
    # ---- C: the collision set at work. Robots dropped on their trunks (legs folded up), robots on their sides, arms swung into
    # the trunk and the front thighs by +-3 rad arm targets; episodes end through terminate_after_contacts_on (WG:940),
    # _reward_collision counts the penalised bodies in contact (LR:865-867). The height termination is moved out of the way.
    cfg = flat_cfg()
    cfg.termination.z_threshold = 0.02
    cfg.asset.terminate_after_contacts_on = ["thigh"]
    cfg.asset.penalize_contacts_on = ["thigh", "trunk", "calf"]
    cfg.rewards.scales.collision = -1.0
    env = make_reference_env(20, seed=13, cfg=cfg)
    env._backend.ora.set_heightfield(None, 0, 0, 0, 0, 0)
    with torch.inference_mode():
        env.reset()
        env.update_command_curriculum()
        n = env.num_envs
        dof = env.dof_state.view(n, 20, 2)
        low = torch.arange(n) < 7                                 # on the trunk: legs folded up beside the body
        env.root_states[low, 2] = 0.062
        env.root_states[low, 7:] = 0
        for leg in range(4):
            dof[low, 3 * leg + 1, 0] = 2.9
            dof[low, 3 * leg + 2, 0] = -2.7
        dof[low, :, 1] = 0
        side = (torch.arange(n) >= 7) & (torch.arange(n) < 10)    # on its side: thighs and trunk edges on the ground
        env.root_states[side, 2] = 0.14
        env.root_states[side, 3:7] = torch.tensor([0.7071068, 0.0, 0.0, 0.7071068])
        arm_envs = tuple(range(10, 17))
        poses = self_contact_arm_poses(env._backend, len(arm_envs), rng)
        for e, pose in zip(arm_envs, poses):                      # arm spheres start inside the trunk box / a thigh capsule
            dof[e, 12:18, 0] = torch.from_numpy(pose.astype(np.float32))
            dof[e, 12:18, 1] = 0
        kick = torch.arange(n) >= 17                               # the free box actor in the path of a front foot, the robot walking into it
        env.box_root_state[kick, 0] = env.root_states[kick, 0] + 0.19 + 0.05 + 0.02
        env.box_root_state[kick, 1] = env.root_states[kick, 1] + torch.tensor([0.13, -0.13, 0.10])
        env.box_root_state[kick, 2] = 0.05
        env.root_states[kick, 7] = torch.tensor([0.5, 0.8, 0.6])
        outC = record_trajectory(env, 12, rng, "C contacts", big_action_envs=arm_envs[:3])
    f = np.stack([outC[f"s{k}/NET_CONTACT_FORCE"] for k in range(12)])                     # [steps, n, 28, 3]
    touching = np.abs(f).sum(-1) > 0
    names = env._backend.model.rb_names
    print("   bodies in contact (steps x envs):", {names[i]: int(touching[:, :, i].sum()) for i in range(27) if touching[:, :, i].any()})
    airborne_arm = touching[:, :, 20:25].any(-1) & (np.stack([outC[f"s{k}/RIGID_BODY_STATE"] for k in range(12)])[:, :, 20:25, 2].min(-1) > 0.08)
    print("   arm links in contact above the ground (self-collision):", int(airborne_arm.sum()))
    box_pushed = (np.abs(f[:, 17:, 27, :2]).sum(-1) > 0.5).sum()
    print("   box pushed sideways by a foot (steps x envs):", int(box_pushed))
    assert touching[:, :, 1].sum() > 10 and airborne_arm.sum() > 5 and box_pushed > 3
    save("wg_reference_contacts.npz", outC)


def cart_fixture():
    """D: goal_ee.command_mode = 'cart' (WG:589-593): curr_ee_goal is the Cartesian goal -- observation entries 70..72 and the sign
    tests of the roll / pitch termination (WG:945-946). Robots tilted past the 0.2 rad threshold in both directions, with goals on both
    sides, so that the two modes decide differently (checked below); goal resampling staged as in fixture A."""
    rng = np.random.default_rng(20261001)
    cfg = flat_cfg()
    cfg.goal_ee.command_mode = "cart"
    env = make_reference_env(24, seed=17, cfg=cfg)
    env._backend.ora.set_heightfield(None, 0, 0, 0, 0, 0)
    with torch.inference_mode():
        env.reset()
        env.update_command_curriculum()
        stage_events(env, rng)
        n = env.num_envs
        tilt = torch.arange(n) < 16                               # roll / pitch of +-0.3 rad, standing height (no height termination)
        ang = torch.tensor([[0.3, 0.0], [-0.3, 0.0], [0.0, 0.3], [0.0, -0.3]]).repeat(4, 1)
        half = 0.5 * ang
        qx = torch.stack([torch.sin(half[:, 0]), torch.zeros(16), torch.zeros(16), torch.cos(half[:, 0])], 1)
        qy = torch.stack([torch.zeros(16), torch.sin(half[:, 1]), torch.zeros(16), torch.cos(half[:, 1])], 1)
        env.root_states[tilt, 3:7] = torch.where(ang[:, :1] != 0, qx, qy)
        env.root_states[tilt, 7:] = 0
        outD = record_trajectory(env, 10, rng, "D command_mode = cart")
    outD["tcfg/goal_command_cart"] = np.int64(1)
    assert int(env._backend.tcfg.goal_command_cart) == 1
    # the two bindings of curr_ee_goal must differ where it matters: observation entries and at least one termination decision
    obs = np.stack([outD[f"s{k}/OBS_BUF"] for k in range(10)])
    goal = np.stack([outD[f"s{k}/GOAL_STATE"] for k in range(10)])
    assert np.abs(obs[:, :, 70:73] - goal[:, :, 12:15]).max() < 1e-6 and np.abs(obs[:, :, 70:73] - goal[:, :, 9:12]).max() > 0.1     # the Cartesian goal, not the spherical
    roll, pitch = np.array([0.3, -0.3, 0, 0] * 4), np.array([0, 0, 0.3, -0.3] * 4)

    def would_end(c, e):
        return ((roll[e] > 0.2) and c[2] >= 0) or ((roll[e] < -0.2) and c[2] <= 0) or ((pitch[e] > 0.2) and c[1] >= 0) or ((pitch[e] < -0.2) and c[1] <= 0)
    differ = sum(would_end(goal[0, e, 9:12], e) != would_end(goal[0, e, 12:15], e) for e in range(16))
    print("   tilted envs whose first-step termination differs between the two bindings of curr_ee_goal:", differ)
    assert differ >= 3
    resets = np.stack([outD[f"s{k}/RESET_BUF"] for k in range(10)])
    print("   resets per step:", resets.sum(1).tolist())
    save("wg_reference_cart.npz", outD)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "cart":
        os.makedirs(GOLD, exist_ok=True)
        cart_fixture()
    else:
        main()
        cart_fixture()
