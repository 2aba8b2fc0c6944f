"""CPU tests of the host side: config mirror vs the reference's own config classes (golden JSON made
by tools/make_golden_config.py), ABI struct sizes, exported symbols of the C-ABI library, the URDF
loader conventions, and the oracle's env-logic helpers against closed-form cases."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

from wbc_amd import abi, native
from wbc_amd.config import WidowGo1RoughCfg, WidowGo1RoughCfgPPO, class_to_dict
from wbc_amd.curriculum import make_curriculum

HERE = os.path.dirname(os.path.abspath(__file__))


def _flatten(d, prefix=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict) and k not in ("default_joint_angles", "stiffness", "damping"):
            out.update(_flatten(v, prefix + k + "."))
        else:
            out[prefix + k] = v
    return out


def _same(a, b):
    if isinstance(a, (list, tuple)) and isinstance(b, (list, tuple)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, dict) and isinstance(b, dict):
        return a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, (int, float)) and isinstance(b, (int, float)) and not isinstance(a, bool):
        return math.isclose(a, b, rel_tol=1e-12, abs_tol=1e-15)
    return a == b


@pytest.mark.parametrize("name,cls", [("WidowGo1RoughCfg", WidowGo1RoughCfg), ("WidowGo1RoughCfgPPO", WidowGo1RoughCfgPPO)])
def test_config_mirror_equals_reference(name, cls):
    gold = json.load(open(os.path.join(HERE, "golden", "widowgo1_config.json")))[name]
    mine = _flatten(class_to_dict(cls()))
    missing = [k for k in gold if k not in mine]
    assert not missing, missing
    bad = {k: (gold[k], mine[k]) for k in gold if not _same(gold[k], mine[k])}
    assert not bad, bad


def test_config_subclassing_is_dropin():
    class MyCfg(WidowGo1RoughCfg):
        class env(WidowGo1RoughCfg.env):
            num_envs = 128
    c = MyCfg()
    assert c.env.num_envs == 128 and c.env.num_observations == 860 and c.sim.dt == 0.005
    assert c.control.stiffness == {"joint": 50, "widow": 5}


def test_abi_struct_sizes_and_symbols():
    L = native.lib()        # loads libwbc_amd.so; no compute call
    sizes = (C.c_int * 3)()
    L.wbc_abi_sizes(sizes)
    assert list(sizes) == [C.sizeof(abi.WbcModel), C.sizeof(abi.WbcTaskCfg), C.sizeof(abi.WbcCurriculum)]
    header = open(os.path.join(HERE, "..", "include", "wbc_sim.h")).read()
    import re
    declared = set(re.findall(r"\b(wbc_[a-z_0-9]+)\s*\(", header))
    for sym in declared:
        assert hasattr(L, sym), f"{sym} declared in include/wbc_sim.h but not exported"
    assert set(native.EXPORTED_SYMBOLS) <= declared | {"wbc_last_error"}
    assert L.wbc_sim_arena_bytes(4096) > 4096 * 860 * 4
    # the library's tensor table (shapes, dtypes) against the binding's (abi.TENSOR_SHAPES / TENSOR_DTYPES): a mismatch shifts
    # every later tensor of the arena
    for name in abi.TENSOR_IDS:
        dims, nd, dt = (C.c_int64 * 3)(), C.c_int(), C.c_int()
        assert L.wbc_tensor_spec(abi.T[name], dims, C.byref(nd), C.byref(dt)) == 0
        assert tuple(dims[i] for i in range(nd.value)) == abi.TENSOR_SHAPES[name], name
        assert ["f32", "i64", "u8"][dt.value] == abi.TENSOR_DTYPES[name], name
    import oracle
    o = C.CDLL(os.path.join(HERE, "..", "oracle", "libwbc_oracle_f64.so"))
    o.ora_abi_sizes(sizes)
    assert list(sizes) == [C.sizeof(abi.WbcModel), C.sizeof(abi.WbcTaskCfg), C.sizeof(abi.WbcCurriculum)]


def test_model_follows_importer_conventions(robot):
    m = robot["model"]
    legs = [f"{l}_{j}_joint" for l in ("FL", "FR", "RL", "RR") for j in ("hip", "thigh", "calf")]
    arm = ["widow_waist", "widow_shoulder", "widow_elbow", "widow_forearm_roll", "widow_wrist_angle", "widow_wrist_rotate"]
    assert m.dof_names == legs + arm + ["widow_left_finger", "widow_right_finger"]
    assert m.num_rigid_bodies == 27 and m.rb_names[0] == "base" and m.rb_names[1] == "trunk"
    assert m.rb_names[-3:] == ["wx250s/ee_gripper_link", "wx250s/left_finger_link", "wx250s/right_finger_link"]
    assert [n for n in m.rb_names if "foot" in n] == ["FL_foot", "FR_foot", "RL_foot", "RR_foot"]
    np.testing.assert_allclose(m.mass.sum(), 14.151, atol=1e-3)                           # SURVEY 8c URDF facts
    np.testing.assert_allclose(m.dof_effort[:12], 23.7)
    np.testing.assert_allclose(m.dof_effort[12:18], [10, 20, 15, 2, 5, 1])
    tc = robot["tcfg"]
    assert tc.max_episode_length == 500 and tc.resample_interval == 150 and tc.push_interval == 150
    np.testing.assert_allclose(list(tc.p_gains), [50] * 12 + [5] * 6)
    np.testing.assert_allclose(list(tc.action_scale), [0.4, 0.45, 0.45] * 4 + [2.1, 0.6, 0.6, 0, 0, 0], rtol=1e-6)


def test_curriculum_saturates_on_first_call_and_rejects_unimplemented_rewards():
    cfg = WidowGo1RoughCfg()
    c0, c1 = make_curriculum(cfg, 0), make_curriculum(cfg, 1)
    assert list(c0.lin_vel_x_range) == [0, 0] and list(c1.lin_vel_x_range) == [0, pytest.approx(0.9)]
    assert c1.leg_reward_scale[abi.REWARD_TERMS.index("tracking_ang_vel_yaw_exp")] == pytest.approx(0.15)
    assert c1.arm_reward_scale[abi.REWARD_TERMS.index("tracking_ee_sphere")] == pytest.approx(0.55)
    active = [abi.REWARD_TERMS[i] for i in range(abi.NREW) if c1.leg_reward_scale[i] != 0]
    assert sorted(active) == sorted(["energy_square", "survive", "tracking_lin_vel_x_l1", "tracking_ang_vel_yaw_exp",
                                     "hip_action_l2", "foot_contacts_z"])                 # SURVEY 8a, compute_reward row
    # the base class's terms (legged_robot.py:832-922) are table entries too: a config subclass that switches one on is drop-in
    cfg.rewards.scales.feet_air_time = 1.0
    cfg.rewards.scales.termination = -5.0
    cfg.rewards.arm_scales.termination = -1.0
    c2 = make_curriculum(cfg, 1)
    ia, it = abi.REWARD_TERMS.index("feet_air_time"), abi.REWARD_TERMS.index("termination")
    assert c2.leg_reward_scale[ia] == 1.0 and (c2.leg_active_mask >> ia) & 1 and not (c2.arm_active_mask >> ia) & 1
    assert c2.leg_reward_scale[it] == -5.0 and c2.arm_reward_scale[it] == -1.0 and (c2.leg_active_mask >> it) & (c2.arm_active_mask >> it) & 1
    # ... except the ones the reference itself cannot run (recorded from its own code: profiles/r04_reference_switches.txt)
    for name, table in (("orientation", "scales"), ("arm_orientation", "arm_scales"), ("feet_stumble", "scales")):
        cfg = WidowGo1RoughCfg()
        setattr(getattr(cfg.rewards, table), name, -1.0)
        with pytest.raises(NotImplementedError, match=name):
            make_curriculum(cfg, 1)
    cfg = WidowGo1RoughCfg()
    cfg.rewards.scales.base_height = -1.0
    assert (make_curriculum(cfg, 1).leg_active_mask >> abi.REWARD_TERMS.index("base_height")) & 1
    cfg.terrain.measure_heights = True
    with pytest.raises(NotImplementedError, match="base_height"):
        make_curriculum(cfg, 1)


def test_oracle_env_logic_closed_form_cases(robot):
    """PD torque law with quirk Q2, observation layout (Appendix A), time-out and z termination."""
    import copy
    from oracle import OracleSim, default_curriculum
    tc = copy.copy(robot["tcfg"])
    tc.push_interval = 0
    o = OracleSim(robot["wmodel"], tc, 2)
    o.set_curriculum(default_curriculum(robot["cfg"]))
    dof = np.zeros((2, 20, 2))
    dof[:, :, 0] = np.array(tc.default_dof_pos)
    dof[:, 12, 0] = 3.5                 # waist beyond pi: wrapped in the observation, NOT in the PD law (Q2)
    dof[:, 0, 1] = 2.0
    o.set("DOF_STATE", dof)
    act = np.zeros((2, 18)); act[:, 1] = 1.0
    o.set("ACTIONS", act)
    ms = np.ones((2, 18)); ms[1, 1] = 1.2
    o.set("MOTOR_STRENGTH", ms)
    o.compute_torques()
    tq = o.get("TORQUES")
    assert tq[0, 0] == pytest.approx(-1.0 * 2.0)                              # -Kd*qd
    assert tq[0, 1] == pytest.approx(50 * 0.45) and tq[1, 1] == pytest.approx(23.7)   # clipped to the effort limit
    assert tq[0, 12] == pytest.approx(max(-10.0, 5 * (0 - 3.5)))                      # unwrapped waist, clipped
    assert (tq[:, 18:] == 0).all()
    # a step from high above the ground: free fall, time-out bookkeeping, observation layout
    root = np.zeros((2, 2, 13)); root[:, 0, 2] = 5.0; root[:, 0, 6] = 1; root[:, 1, 6] = 1
    o.set("ROOT_STATES", root)
    o.set("EPISODE_LENGTH", np.array([10.0, 500.0]))
    o.set("COMMANDS", np.array([[0.5, 0, 0.7]] * 2))
    o.step_counter = 7
    o.step(np.zeros((2, 18)))
    assert o.get("TIME_OUT_BUF").tolist() == [0, 1] and o.get("RESET_BUF").tolist() == [0, 1]
    assert o.get("EPISODE_LENGTH").tolist() == [11, 0]
    obs = o.get("OBS_BUF")
    assert obs.shape == (2, 860)
    q_obs = obs[0, 5:25]
    assert q_obs[12] == pytest.approx(((3.5 + math.pi) % (2 * math.pi)) - math.pi, abs=0.2)   # wrapped waist (moves a bit in 4 substeps)
    assert obs[0, 67] == pytest.approx(0.5) and obs[0, 69] == pytest.approx(0.7)              # commands * scale
    np.testing.assert_allclose(obs[0, 82:100], ms[0] - 1)                                       # priv: motor_strength - 1
    assert (obs[1, 100:] == 0).all()                                                           # reset env: history zeroed before assembly
    hist = o.get("OBS_HISTORY")
    np.testing.assert_allclose(hist[1, 0], hist[1, 9])                                          # ... then refilled 10x with the new obs


def test_perlin_terrain_matches_reference_statistics():
    """TerrainPerlin (utils/terrain.py:40-99 restated): shape, value range, octave structure and quirk Q3."""
    from wbc_amd.terrain import TerrainPerlin, perlin_2d
    cfg = WidowGo1RoughCfg().terrain
    cfg.tot_cols, cfg.tot_rows = 400, 200               # small grid: 10 m x 5 m at 0.025 m
    t = TerrainPerlin(cfg, seed=4)
    assert t.heightsamples.shape == (400, 200) and t.heightsamples.dtype == np.int16
    near = t.heightsamples[: t.flat_beyond_row].astype(np.float64) * cfg.vertical_scale
    far = t.heightsamples[t.flat_beyond_row:]
    assert (far == 0).all()                               # Q3: flat beyond row tot_cols//2 - 100
    # each octave lies in [0, zScale * amp]; two octaves with gain 0.25: total within [0, 1.25 * zScale]
    assert near.min() >= -1e-9 and near.max() <= 1.25 * cfg.zScale + 1e-9
    assert 0.3 * cfg.zScale < near.mean() < 0.95 * cfg.zScale
    # gradient noise vanishes at lattice nodes (value 0.5 after the affine map) and is smooth in between
    p = perlin_2d((64, 64), (4, 4), np.random.default_rng(0))
    np.testing.assert_allclose(p[::16, ::16], 0.5, atol=1e-12)
    assert np.abs(np.diff(p, axis=0)).max() < 0.2
    # determinism in the seed
    np.testing.assert_array_equal(TerrainPerlin(cfg, seed=4).heightsamples, t.heightsamples)
    assert (TerrainPerlin(cfg, seed=5).heightsamples != t.heightsamples).any()


def test_rollout_slot_hand_over_is_cuda_only_and_storage_skips_filled_slots():
    """The fused-rollout hand-overs (observation / reward / done slots written by the env's step) are offered only for CUDA
    storages, and RolloutStorage.add_transitions copies exactly the fields that are not already views of their slots."""
    import torch
    import golden_procedure as gp
    from wbc_amd.rsl_rl.algorithms import PPO
    from wbc_amd.rsl_rl.modules import ActorCritic
    torch.manual_seed(0)
    ac = ActorCritic(76, 76, 18, **gp.POLICY_KW)
    alg = PPO(ac, device="cpu", **gp.ALG_KW)
    alg.init_storage(6, 3, [860], [None], [18])
    assert alg.next_observation_slot() is None and alg.rollout_slots() is None          # CPU storage: the eager path
    st = alg.storage
    obs = torch.randn(6, 860)
    with torch.inference_mode():
        alg.act(obs, obs, False)
        tr = alg.transition
        assert tr.observations.data_ptr() == st.observations[0].data_ptr()              # parked at act() time (WidowGo1 reuses its buffer)
        # pretend the env filled the reward / done slots itself (extras['rollout_stored']): nothing is overwritten
        st.rewards[0].fill_(3.5)
        st.dones[0].fill_(1)
        tr_r, tr_d = st.rewards[0], st.dones[0]
        alg.transition.rewards, alg.transition.dones = tr_r, tr_d.view(-1)
        alg.storage.add_transitions(alg.transition, torque_supervision=False)
    assert st.step == 1 and (st.rewards[0] == 3.5).all() and (st.dones[0] == 1).all()
    assert torch.equal(st.observations[0], obs) and torch.isfinite(st.values[0]).all() and (st.sigma[0] > 0).all()


def test_collision_set_follows_the_urdf_collision_blocks():
    """abi.collision_set: the URDF's <collision> geometry (urdf/widowGo1.urdf) as the contact list of the physics spec: one sphere
    per foot first (the force sensors read contacts 0..3), trunk-box corners on the box surface, thigh tops / knees / mid-shanks on
    the thigh and calf rows, arm spheres, the corners of the free box actor (WG:321-325); then the pairs, each naming a partner
    body different from its own: the robot's self-collision pairs and the robot spheres against the box; fits WBC_NCP."""
    m = abi.load_default_model()
    cps = abi.collision_set(m)
    names = m.rb_names
    assert len(cps) == 47 <= abi.NCP
    feet = [i for i, n in enumerate(names) if "foot" in n]
    assert [c["rb"] for c in cps[:4]] == feet and all(c["kind"] == abi.CP_TERRAIN and c["radius"] == 0.02 for c in cps[:4])
    terrain = [c for c in cps if c["kind"] == abi.CP_TERRAIN]
    pairs = [c for c in cps if c["kind"] != abi.CP_TERRAIN]
    assert len(terrain) == 35 and len(pairs) == 12
    # slots (= wavefront lanes): the robot's terrain spheres and self pairs below 32 (the set a walking robot lives in), everything
    # that involves the free box in the 16-lane row 32..47 (summed by a row reduction in the kernel), the mid-shanks from 48
    slots = [c["slot"] for c in cps]
    assert slots == sorted(slots) and len(set(slots)) == 47 and max(slots) < abi.NCP
    low = [c for c in cps if c["slot"] < 32]
    assert [c["slot"] for c in low] == list(range(30)) and all(c["body"] != abi.BOX_BODY and c["body2"] != abi.BOX_BODY for c in low)
    row = [c for c in cps if 32 <= c["slot"] < 48]
    assert len(row) == 13 and all(c["body"] == abi.BOX_BODY or c["body2"] == abi.BOX_BODY for c in row)
    assert [c["slot"] for c in cps if c["radius"] == abi.CALF_RADIUS] == [48, 49, 50, 51]
    wm = abi.fill_model(m)
    assert wm.ncp == 52 and wm.cp_kind[31] == abi.CP_NONE and all(wm.cp_kind[c["slot"]] == c["kind"] for c in cps)
    robot_terrain = [c for c in terrain if c["body"] != abi.BOX_BODY]
    box_corners = [c for c in terrain if c["body"] == abi.BOX_BODY]
    assert len(robot_terrain) == 27 and len(box_corners) == 8 and all(c["rb"] == abi.BOX_RB for c in box_corners)
    for c in box_corners:                                # sphere surface = the 0.1 m cube (box.box_size, widowGo1_config.py:186)
        np.testing.assert_allclose(np.abs(c["pos"]) + c["radius"], 0.05, atol=1e-9)
    corners = [c for c in robot_terrain if names[c["rb"]] == "trunk"]
    assert len(corners) == 8
    for c in corners:                                    # sphere surface = the URDF box 0.3762 x 0.0935 x 0.114
        np.testing.assert_allclose(np.abs(c["pos"]) + c["radius"], np.array([0.3762, 0.0935, 0.114]) / 2, atol=1e-9)
    assert sorted(names[c["rb"]] for c in robot_terrain if "thigh" in names[c["rb"]]) == ["FL_thigh", "FR_thigh", "RL_thigh", "RR_thigh"]
    calf = [c for c in robot_terrain if "calf" in names[c["rb"]]]
    assert len(calf) == 8                                # knee (calf origin) + mid-shank (the middle of the calf box, urdf:981)
    shank = [c for c in calf if c["radius"] == abi.CALF_RADIUS]
    assert len(shank) == 4
    for c in shank:
        np.testing.assert_allclose(c["pos"] - np.asarray(m.rb_offset[c["rb"]]), [0, 0, -0.1065], atol=1e-9)
    arm = {names[c["rb"]] for c in robot_terrain if "wx250s" in names[c["rb"]]}
    assert arm == {"wx250s/ee_gripper_link", "wx250s/upper_forearm_link", "wx250s/wrist_link"}
    self_pairs = [c for c in pairs if c["body2"] != abi.BOX_BODY]
    box_pairs = [c for c in pairs if c["body2"] == abi.BOX_BODY]
    assert len(self_pairs) == 7 and len(box_pairs) == 5
    for c in self_pairs:
        assert c["body2"] >= 0 and c["body2"] != c["body"] and "wx250s" in names[c["rb"]]
        assert names[c["rb2"]] in ("trunk", "FL_thigh", "FR_thigh")
        assert (c["kind"] == abi.CP_BOX) == (names[c["rb2"]] == "trunk")
    for c in box_pairs:
        assert c["kind"] == abi.CP_BOX and c["rb2"] == abi.BOX_RB and np.allclose(c["b"], 0.05) and np.allclose(c["a"], 0)
    assert sorted(names[c["rb"]] for c in box_pairs) == sorted([names[i] for i in feet] + ["wx250s/ee_gripper_link"])
    wm0 = abi.fill_model(m, self_collisions=False)                      # no pairs at all: the box actor shares the filter (WG:384)
    assert sum(wm0.cp_kind[k] != abi.CP_NONE for k in range(wm0.ncp)) == 35 and all(wm0.cp_kind[k] <= abi.CP_TERRAIN for k in range(wm0.ncp))
    # rigid-body masks of the task config (WG:299-306: substring match)
    cfg = WidowGo1RoughCfg()
    cfg.asset.terminate_after_contacts_on = ["wx250", "base"]        # the list the reference keeps commented out (widowGo1_config.py:179)
    tc = abi.fill_task_cfg(cfg, m)
    assert tc.penalize_contact_rb_mask == sum(1 << i for i, n in enumerate(names) if "thigh" in n or "trunk" in n)
    assert tc.term_contact_rb_mask == sum(1 << i for i, n in enumerate(names) if "wx250" in n or "base" in n)


def _set_path(cfg, path, value):
    o = cfg
    parts = path.split(".")
    for name in parts[:-1]:
        o = getattr(o, name)
    setattr(o, parts[-1], value)


def test_no_config_switch_is_silently_ignored():
    """Every switch of WidowGo1RoughCfg that the reference's widowGo1 path reads is one of: implemented (flipping it changes the
    task constants / the model handed to the kernels), refused (abi.UNSUPPORTED_SWITCHES: NotImplementedError naming the switch
    and the reference line), or a no-op in the reference itself (abi.REFERENCE_NO_OPS: the code that would read it is commented
    out or overridden there; what the reference does with the doubtful ones is recorded by tools/check_reference_dead_switches.py
    in profiles/r04_reference_switches.txt)."""
    import ctypes
    m = abi.load_default_model()

    def snapshot(cfg):
        tc = abi.fill_task_cfg(cfg, m)
        wm = abi.fill_model(m, foot_name=cfg.asset.foot_name, self_collisions=int(cfg.asset.self_collisions) == 0, box_size=float(cfg.box.box_size))
        return bytes(ctypes.string_at(ctypes.addressof(tc), ctypes.sizeof(tc))) + bytes(ctypes.string_at(ctypes.addressof(wm), ctypes.sizeof(wm)))
    base = snapshot(WidowGo1RoughCfg())
    # refused
    flips = {"control.adaptive_arm_gains": True, "env.reorder_dofs": False, "domain_rand.observe_priv": False, "goal_ee.command_mode": "cart",
             "asset.fix_base_link": True, "asset.disable_gravity": True, "asset.collapse_fixed_joints": False, "asset.default_dof_drive_mode": 1,
             "asset.linear_damping": 0.1, "asset.angular_damping": 0.1, "terrain.restitution": 0.5, "sim.substeps": 2, "sim.up_axis": 0}
    assert set(flips) == {p for p, _, _ in abi.UNSUPPORTED_SWITCHES}
    for path, value in flips.items():
        cfg = WidowGo1RoughCfg()
        _set_path(cfg, path, value)
        with pytest.raises(NotImplementedError, match=path.replace(".", r"\.")):
            abi.fill_task_cfg(cfg, m)
    cfg = WidowGo1RoughCfg()
    cfg.terrain.dynamic_friction = 0.5
    with pytest.raises(NotImplementedError, match="dynamic_friction"):
        abi.fill_task_cfg(cfg, m)
    # no-ops of the reference: nothing handed to the kernels changes
    noop_values = {"noise.add_noise": True, "commands.heading_command": False, "commands.curriculum": False, "domain_rand.randomize_arm_ema": True,
                   "control.control_type": "V", "box.box_pos_obs_range": 2.0, "arm.grasp_offset": 0.2, "arm.init_target_ee_base": [0.3, 0.1, 0.1],
                   "termination.r_threshold": 0.3, "termination.p_threshold": 0.3, "asset.flip_visual_attachments": True,
                   "asset.max_linear_velocity": 500.0, "asset.max_angular_velocity": 500.0, "asset.thickness": 0.02, "asset.density": 0.01,
                   "asset.replace_cylinder_with_capsule": False, "env.env_spacing": 5.0, "viewer.pos": [0, 0, 1], "viewer.lookat": [1, 0, 0]}
    assert set(noop_values) == {p for p, _ in abi.REFERENCE_NO_OPS}
    for path, value in noop_values.items():
        cfg = WidowGo1RoughCfg()
        try:
            _set_path(cfg, path, value)
        except AttributeError:
            continue                                           # (a field the shipped config does not even define)
        assert snapshot(cfg) == base, path
    # implemented: each of these changes what the kernels receive
    effective = {"control.decimation": 2, "control.action_scale": [0.3] * 18, "control.stiffness": {"joint": 40, "widow": 4},
                 "control.damping": {"joint": 2, "widow": 1}, "env.action_delay": 1, "env.episode_length_s": 5, "normalization.clip_actions": 50.0,
                 "normalization.clip_observations": 50.0, "termination.z_threshold": 0.2, "asset.terminate_after_contacts_on": ["thigh"],
                 "asset.penalize_contacts_on": ["calf"], "asset.self_collisions": 1, "asset.armature": 0.01, "asset.foot_name": "foot",
                 "commands.resampling_time": 2.0, "commands.lin_vel_x_clip": 0.2, "commands.ang_vel_yaw_clip": 0.4, "domain_rand.push_robots": False,
                 "domain_rand.push_interval_s": 5, "domain_rand.max_push_vel_xy": 1.0, "goal_ee.underground_limit": -0.4,
                 "goal_ee.num_collision_check_samples": 5, "goal_ee.collision_upper_limits": [0.3, 0.15, 0.0], "goal_ee.sphere_error_scale": [1, 1, 1],
                 "goal_ee.orn_error_scale": [1, 1, 1], "rewards.tracking_sigma": 0.5, "rewards.tracking_ee_sigma": 0.5, "rewards.only_positive_rewards": True,
                 "init_state.pos": [0, 0, 0.5], "terrain.origin_perturb_range": 0.1, "terrain.init_vel_perturb_range": 0.3, "terrain.static_friction": 0.5,
                 "box.box_env_origins_x": 1.0, "box.box_env_origins_z": 0.3, "box.box_size": 0.2, "sim.dt": 0.0025, "sim.gravity": [0, 0, -5.0]}
    for path, value in effective.items():
        cfg = WidowGo1RoughCfg()
        _set_path(cfg, path, value)
        if path == "terrain.static_friction":
            cfg.terrain.dynamic_friction = value               # (one Coulomb coefficient: the two must agree)
        if path == "asset.foot_name":
            continue                                           # (renaming the feet away is an assertion in fill_model: covered below)
        assert snapshot(cfg) != base, f"{path} = {value!r} changed nothing"
    cfg = WidowGo1RoughCfg()
    cfg.sim.physx.contact_offset = 0.02
    assert snapshot(cfg) != base
    cfg = WidowGo1RoughCfg()
    cfg.sim.physx.max_depenetration_velocity = 2.0
    assert snapshot(cfg) != base
