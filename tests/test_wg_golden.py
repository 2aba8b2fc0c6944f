"""The sim-side task logic against the REFERENCE's own code.

tests/golden/wg_reference_*.npz hold trajectories of the reference's `WidowGo1.step` (run on a fake
Isaac Gym whose `simulate` is this framework's physics spec; tools/make_golden_wg.py, tools/ref_harness/).
Each step is replayed from the recorded pre-state through (a) the C oracle here on the CPU and (b) the
HIP step kernel on the GPU (-m gpu), and everything the reference computed around physics is compared:
torques, observations and history, both reward channels, episode / metric sums, commands, the EE-goal
state machine, reset / time-out masks and episode lengths (bit-exact), reset states and their draws.
"""
import os

import numpy as np
import pytest

from wbc_amd import abi

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INPUT_NAMES = ["ROOT_STATES", "DOF_STATE", "TORQUES", "OBS_HISTORY", "ACTION_HISTORY", "ACTIONS", "LAST_ACTIONS", "LAST_DOF_VEL",
               "LAST_ROOT_VEL", "COMMANDS", "GOAL_STATE", "EPISODE_LENGTH", "EPISODE_SUMS", "METRIC_SUMS", "FORCE_SENSOR",
               "NET_CONTACT_FORCE", "RIGID_BODY_STATE", "BASE_LIN_VEL", "BASE_ANG_VEL", "TIME_OUT_BUF", "RESET_BUF", "BOX_SLEEP_TIMER", "FEET_AIR_TIME", "LAST_CONTACTS"]
EXACT = ["RESET_BUF", "TIME_OUT_BUF", "EPISODE_LENGTH"]
# (name, atol, rtol) for the fp64 oracle against the reference's fp32 torch arithmetic
CHECK_F64 = [("TORQUES", 2e-5, 1e-5), ("ROOT_STATES", 2e-5, 1e-5), ("DOF_STATE", 5e-5, 1e-5), ("OBS_BUF", 5e-5, 2e-5),
             ("OBS_HISTORY", 5e-5, 2e-5), ("ACTION_HISTORY", 0, 0), ("ACTIONS", 0, 0), ("LAST_ACTIONS", 0, 0),
             ("LAST_DOF_VEL", 5e-5, 1e-5), ("LAST_ROOT_VEL", 2e-5, 1e-5), ("COMMANDS", 1e-6, 1e-6), ("GOAL_STATE", 2e-6, 2e-6),
             ("REW_BUF", 2e-6, 2e-5), ("ARM_REW_BUF", 2e-6, 2e-5), ("EPISODE_SUMS", 2e-4, 3e-5), ("METRIC_SUMS", 2e-3, 3e-5),
             ("BASE_LIN_VEL", 2e-5, 1e-5), ("BASE_ANG_VEL", 2e-5, 1e-5),
             # physics outputs the reference only reads (the fake gym's tensors): recorded with the post-state all the same
             ("FORCE_SENSOR", 5e-4, 1e-5), ("NET_CONTACT_FORCE", 5e-4, 1e-5), ("RIGID_BODY_STATE", 1e-4, 2e-5)]
# the fp32 step kernel carries its own physics rounding through four substeps: the tolerances of test_gpu_sim_parity.py
CHECK_GPU = [("TORQUES", 3e-3, 1e-3), ("ROOT_STATES", 3e-4, 5e-4), ("DOF_STATE", 1.5e-3, 1e-3), ("OBS_BUF", 1.5e-3, 1e-3),
             ("OBS_HISTORY", 1.5e-3, 1e-3), ("ACTION_HISTORY", 0, 0), ("ACTIONS", 0, 0), ("LAST_ACTIONS", 0, 0),
             ("COMMANDS", 1e-6, 1e-6), ("GOAL_STATE", 2e-5, 2e-5), ("REW_BUF", 3e-5, 2e-3), ("ARM_REW_BUF", 3e-5, 2e-3),
             ("BASE_LIN_VEL", 5e-4, 1e-3), ("BASE_ANG_VEL", 2e-3, 1e-3),
             # every other tensor the fixture records with the post-state (the fp64 list above has them too)
             ("LAST_DOF_VEL", 1.5e-3, 1e-3), ("LAST_ROOT_VEL", 5e-4, 1e-3), ("EPISODE_SUMS", 2e-4, 2e-3), ("METRIC_SUMS", 5e-3, 2e-3),
             ("FORCE_SENSOR", 0.05, 3e-3), ("NET_CONTACT_FORCE", 0.05, 3e-3), ("RIGID_BODY_STATE", 1e-3, 1e-3)]

FIXTURES = ["wg_reference_counter0.npz", "wg_reference_default.npz", "wg_reference_allrewards.npz", "wg_reference_contacts.npz",
            "wg_reference_cart.npz"]        # (cart: goal_ee.command_mode = 'cart', robots tilted past the roll / pitch threshold)


def load(name):
    return dict(np.load(os.path.join(GOLD, name)))


def cur_from_array(a):
    c = abi.WbcCurriculum()
    abi._set(c.lin_vel_x_range, a[0:2])
    abi._set(c.ang_vel_yaw_range, a[2:4])
    abi._set(c.goal_l_range, a[4:6])
    abi._set(c.goal_p_range, a[6:8])
    abi._set(c.goal_y_range, a[8:10])
    abi._set(c.leg_reward_scale, a[10:10 + abi.NREW])
    abi._set(c.arm_reward_scale, a[10 + abi.NREW:10 + 2 * abi.NREW])
    c.leg_active_mask, c.arm_active_mask = int(a[10 + 2 * abi.NREW]), int(a[11 + 2 * abi.NREW])
    return c


def fixture_tcfg(robot, g):
    """Task config of the run that produced fixture `g`: the all-rewards run uses non-zero orientation-goal ranges, the contacts run
    a lowered height threshold and non-empty contact lists (recorded with the fixture)."""
    tc = type(robot["tcfg"]).from_buffer_copy(robot["tcfg"])
    if "delta_orn" in g:
        abi._set(tc.goal_delta_orn_range, g["delta_orn"])
    if "soft_limits" in g:
        abi.set_soft_limits(tc, robot["model"], *[float(v) for v in g["soft_limits"]])
    tc.term_z_threshold = float(g["tcfg/term_z_threshold"])
    tc.term_contact_rb_mask = int(g["tcfg/term_contact_rb_mask"])
    tc.penalize_contact_rb_mask = int(g["tcfg/penalize_contact_rb_mask"])
    tc.goal_command_cart = int(g["tcfg/goal_command_cart"]) if "tcfg/goal_command_cart" in g else 0       # WG:589-593
    return tc


def params_of(g):
    return {k[len("param/"):]: g[k] for k in g if k.startswith("param/")}


def compare(tag, got, ref, checks, exact=EXACT):
    """got(name) -> ndarray; ref: dict name -> ndarray."""
    for name in exact:
        np.testing.assert_array_equal(got(name).astype(np.int64), ref[name].astype(np.int64), err_msg=f"{tag} {name}")
    worst = {}
    for name, atol, rtol in checks:
        a, b = np.asarray(got(name), dtype=np.float64), np.asarray(ref[name], dtype=np.float64)
        if name == "GOAL_STATE":
            pass
        np.testing.assert_allclose(a, b, atol=atol, rtol=rtol, err_msg=f"{tag} {name}")
        worst[name] = float(np.max(np.abs(a - b))) if a.size else 0.0
    return worst


@pytest.mark.parametrize("fixture", FIXTURES)
def test_oracle_matches_reference_step(robot, fixture):
    """oracle/wbc_oracle.c (fp64) replays every recorded step of the reference from its recorded pre-state."""
    from oracle import OracleSim
    g = load(fixture)
    n = g["actions"].shape[1]
    o = OracleSim(robot["wmodel"], fixture_tcfg(robot, g), n, seed=int(g["seed"]), precision="f64")
    o.set_env_params(robot_model=robot["model"], **params_of(g))
    resets = 0
    for k in range(int(g["steps"])):
        pre = (lambda nm: g["init/" + nm]) if k == 0 else (lambda nm: g[f"s{k - 1}/{nm}"])
        for name in INPUT_NAMES:
            o.set(name, pre(name).astype(np.float64))
        o.step_counter = int(g["step_counter0"]) + k
        o.set_curriculum(cur_from_array(g["curriculum"][k]))
        o.step(g["actions"][k])
        ref = {nm: g[f"s{k}/{nm}"] for nm in INPUT_NAMES + ["OBS_BUF", "REW_BUF", "ARM_REW_BUF"]}
        compare(f"{fixture} step {k}", o.get, ref, CHECK_F64)
        if ref["RESET_BUF"].any():     # extras['time_outs'] is only re-bound inside reset_idx (WG:753-754): stale on a step without resets
            np.testing.assert_array_equal(o.get("TIME_OUT_BUF").astype(np.uint8), g[f"s{k}/time_outs"])
        resets += int(ref["RESET_BUF"].sum())
    if fixture != "wg_reference_counter0.npz":
        assert resets >= 10
    if fixture == "wg_reference_contacts.npz":     # the collision set at work: trunk / thigh contacts, arm self-collision, contact resets
        f = np.stack([g[f"s{k}/NET_CONTACT_FORCE"] for k in range(int(g["steps"]))])
        assert (np.abs(f[:, :, 1]).sum(-1) > 0).sum() > 10 and (np.abs(f[:, :, [3, 7, 11, 15]]).sum(-1) > 0).sum() >= 5
        assert (np.abs(f[:, :, 20:25]).sum(-1) > 0).sum() > 10


def test_oracle_f32_matches_reference_step(robot):
    """The fp32 build of the oracle (the rounding mirror of the kernels) against the same trajectory: masks stay exact."""
    from oracle import OracleSim
    g = load("wg_reference_default.npz")
    n = g["actions"].shape[1]
    o = OracleSim(robot["wmodel"], robot["tcfg"], n, seed=int(g["seed"]), precision="f32")
    o.set_env_params(robot_model=robot["model"], **params_of(g))
    for k in range(int(g["steps"])):
        pre = (lambda nm: g["init/" + nm]) if k == 0 else (lambda nm: g[f"s{k - 1}/{nm}"])
        for name in INPUT_NAMES:
            o.set(name, pre(name).astype(np.float64))
        o.step_counter = int(g["step_counter0"]) + k
        o.set_curriculum(cur_from_array(g["curriculum"][k]))
        o.step(g["actions"][k])
        ref = {nm: g[f"s{k}/{nm}"] for nm in INPUT_NAMES + ["OBS_BUF", "REW_BUF", "ARM_REW_BUF"]}
        compare(f"f32 step {k}", o.get, ref, CHECK_GPU)


def test_curriculum_tables_match_reference(robot):
    """make_curriculum (the struct the kernels read) equals the reference object's ranges / scales before the first
    update_command_curriculum call (counter 0: config scales, init ranges) and after it (counter 1, shipped schedules)."""
    from wbc_amd.curriculum import make_curriculum

    def arr(c):
        return np.array(list(c.lin_vel_x_range) + list(c.ang_vel_yaw_range) + list(c.goal_l_range) + list(c.goal_p_range) +
                        list(c.goal_y_range) + list(c.leg_reward_scale) + list(c.arm_reward_scale) +
                        [c.leg_active_mask, c.arm_active_mask])
    g0, g1 = load("wg_reference_counter0.npz"), load("wg_reference_default.npz")
    np.testing.assert_allclose(arr(make_curriculum(robot["cfg"], 0)), g0["curriculum"][0], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(arr(make_curriculum(robot["cfg"], 1)), g1["curriculum"][0], rtol=1e-6, atol=1e-7)


def test_episode_extras_match_reference(robot):
    """extras['episode'] (WG:743-750): mean over the envs reset in a step of their finished episode sums / max_episode_length_s."""
    from oracle import OracleSim
    g = load("wg_reference_default.npz")
    n = g["actions"].shape[1]
    o = OracleSim(robot["wmodel"], robot["tcfg"], n, seed=int(g["seed"]), precision="f64")
    o.set_env_params(robot_model=robot["model"], **params_of(g))
    checked = 0
    for k in range(int(g["steps"])):
        pre = (lambda nm: g["init/" + nm]) if k == 0 else (lambda nm: g[f"s{k - 1}/{nm}"])
        for name in INPUT_NAMES:
            o.set(name, pre(name).astype(np.float64))
        o.step_counter = int(g["step_counter0"]) + k
        o.set_curriculum(cur_from_array(g["curriculum"][k]))
        o.step(g["actions"][k])
        m = o.get("RESET_BUF").astype(bool)
        if m.any():
            got = np.concatenate([o.get("EPISODE_SUMS_DONE")[m].mean(0), o.get("METRIC_SUMS_DONE")[m].mean(0)]) / 10.0
            ref = g["episode_extras"][k]
            ok = ~np.isnan(ref)
            np.testing.assert_allclose(got[ok], ref[ok], rtol=3e-5, atol=2e-5, err_msg=f"step {k}")
            checked += 1
    assert checked >= 5


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("fixture", FIXTURES)
def test_hip_step_matches_reference_step(robot, fixture):
    """The fused HIP step (through the C-ABI) replays every recorded step of the reference from its recorded pre-state."""
    import torch
    from wbc_amd.sim import WbcSim
    g = load(fixture)
    n = g["actions"].shape[1]
    dev = torch.device("cuda:0")
    sim = WbcSim(robot["wmodel"], fixture_tcfg(robot, g), n, dev, seed=int(g["seed"]))
    sim.set_env_params(**params_of(g))
    worst = {}
    for k in range(int(g["steps"])):
        pre = (lambda nm: g["init/" + nm]) if k == 0 else (lambda nm: g[f"s{k - 1}/{nm}"])
        for name in INPUT_NAMES:
            t = sim.tensor(name)
            t.copy_(torch.from_numpy(np.ascontiguousarray(pre(name))).to(dev).to(t.dtype).reshape(t.shape))
        sim.step_counter = int(g["step_counter0"]) + k
        sim.set_curriculum(cur_from_array(g["curriculum"][k]))
        sim.step(torch.from_numpy(g["actions"][k]).to(dev))
        torch.cuda.synchronize()
        ref = {nm: g[f"s{k}/{nm}"] for nm in INPUT_NAMES + ["OBS_BUF", "REW_BUF", "ARM_REW_BUF"]}
        w = compare(f"{fixture} step {k}", lambda nm: sim.tensor(nm).cpu().numpy(), ref, CHECK_GPU)
        for kk, v in w.items():
            worst[kk] = max(worst.get(kk, 0.0), v)
        # extras['episode'] (WG:742-750): means over the envs that reset in this step; untouched on a step without resets
        stats = sim.episode_stats(1.0 / 10.0).cpu().numpy()
        if ref["RESET_BUF"].any():
            want = g["episode_extras"][k]
            ok = ~np.isnan(want)
            np.testing.assert_allclose(stats[ok], want[ok], rtol=3e-3, atol=2e-4, err_msg=f"{fixture} episode stats, step {k}")
        elif k > 0:
            np.testing.assert_array_equal(stats, last_stats)
        last_stats = stats
    print("max abs deviation from the reference per tensor:", {k: f"{v:.2e}" for k, v in worst.items()})


def test_stale_time_outs_option_reproduces_reference_extras():
    """Quirk Q9 (cfg.env.reference_stale_time_outs): the reference re-binds extras['time_outs'] only inside reset_idx, so a step
    without resets publishes the mask of the last step that had one. envs.stale_time_outs over the recorded reset / time-out
    masks must give the recorded extras -- including wg_reference_allrewards step 4, where no env resets, the current mask is
    empty and the reference still hands the learner one time-out."""
    import torch
    from wbc_amd.envs import stale_time_outs
    stale_steps = 0
    for fixture in FIXTURES:
        g = load(fixture)
        prev = torch.zeros(g["actions"].shape[1], dtype=torch.bool)
        for k in range(int(g["steps"])):
            cur = torch.from_numpy(g[f"s{k}/TIME_OUT_BUF"].astype(bool))
            reset = torch.from_numpy(g[f"s{k}/RESET_BUF"].astype(np.int64))
            prev = stale_time_outs(prev, cur, reset)
            np.testing.assert_array_equal(prev.numpy().astype(np.uint8), g[f"s{k}/time_outs"], err_msg=f"{fixture} step {k}")
            stale_steps += int(not reset.any() and not torch.equal(prev, cur))
    assert stale_steps >= 1
