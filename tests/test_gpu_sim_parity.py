"""GPU parity of the fused rollout step (libwbc_amd.so, through the C-ABI) against the oracle on the
same seeded inputs. Floating-point tolerance: the HIP path computes in fp32; against the fp64 oracle
(the spec) and its fp32 build, single-step state errors stay below 3e-4 absolute / 5e-4 relative
(velocities: the contact solve amplifies fp32 rounding of positions by 1/dt). Reset masks,
time-outs, episode lengths and every random draw are compared bit-exactly."""
import copy

import numpy as np
import pytest

import helpers
from wbc_amd import abi

pytestmark = pytest.mark.gpu

N = 64


def _rows_close(a, b, atol, rtol, msg):
    """assert_allclose for [n, rows, k] tensors that names the offending (env, row) pairs."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    bad = np.abs(a - b) > atol + rtol * np.abs(b)
    if bad.any():
        idx = np.argwhere(bad.any(-1))[:8]
        detail = "; ".join(f"env {e} row {r}: got {np.round(a[e, r], 4)} want {np.round(b[e, r], 4)}" for e, r in idx)
        raise AssertionError(f"{msg}: {int(bad.any(-1).sum())} rows differ (max err {np.abs(a - b).max():.3e}): {detail}")


def _t(g, name):
    import torch
    torch.cuda.synchronize()
    return g.tensor(name).detach().cpu().numpy().astype(np.float64)


def test_simulate_substep_matches_oracle(robot):
    import torch
    params = helpers.random_env_params(N, seed=3)
    g = helpers.make_gpu(robot, N, params)
    o = helpers.make_oracle(robot, N, params, "f64")
    rng = np.random.default_rng(11)
    root, dof = helpers.random_standing_state(N, robot["tcfg"], rng)
    tau = rng.uniform(-8, 8, (N, 20)).astype(np.float32)
    tau[:, 18:] = 0
    g.tensor("ROOT_STATES").copy_(torch.from_numpy(root))
    g.tensor("DOF_STATE").copy_(torch.from_numpy(dof))
    g.set_dof_forces(torch.from_numpy(tau).cuda())
    o.set("ROOT_STATES", root); o.set("DOF_STATE", dof); o.set("TORQUES", tau)
    n_contact = 0
    for it in range(3):
        g.simulate()
        o.simulate()
        np.testing.assert_allclose(_t(g, "DOF_STATE"), o.get("DOF_STATE"), atol=3e-4, rtol=1e-4)
        np.testing.assert_allclose(_t(g, "ROOT_STATES")[:, 0], o.get("ROOT_STATES")[:, 0], atol=3e-4, rtol=1e-4)
        fo = o.get("NET_CONTACT_FORCE")
        _rows_close(_t(g, "NET_CONTACT_FORCE"), fo, 0.05, 2e-3, f"substep {it} NET_CONTACT_FORCE")
        _rows_close(_t(g, "ROOT_STATES")[:, 1:, :], o.get("ROOT_STATES")[:, 1:, :], 2e-3, 2e-3, f"substep {it} box state")
        np.testing.assert_allclose(_t(g, "FORCE_SENSOR"), o.get("FORCE_SENSOR"), atol=0.05, rtol=2e-3)
        n_contact += (np.abs(fo).sum(-1) > 0).sum()
        helpers.sync_oracle_from_gpu(o, g, ["ROOT_STATES", "DOF_STATE"])
    assert n_contact > N      # the test states do exercise the contact solver
    g.close()


@pytest.mark.parametrize("precision,atol", [("f64", 3e-4), ("f32", 3e-4)])
def test_step_matches_oracle_from_synced_state(robot, precision, atol):
    import torch
    params = helpers.random_env_params(N, seed=5)
    g = helpers.make_gpu(robot, N, params)
    o = helpers.make_oracle(robot, N, params, precision)
    g.reset_all()
    o.reset_all()
    for name in ("ROOT_STATES", "DOF_STATE", "GOAL_STATE", "COMMANDS", "RIGID_BODY_STATE"):
        np.testing.assert_allclose(_t(g, name), o.get(name), atol=2e-5, rtol=1e-5, err_msg=name)
    rng = np.random.default_rng(17)
    resets = 0
    # a step counter that makes the push (every 150 steps) and the command resampling fire inside the window
    g.step_counter = 140
    o.step_counter = 140
    for step in range(40):
        helpers.sync_oracle_from_gpu(o, g)
        a = (0.6 * rng.normal(size=(N, 18))).astype(np.float32)
        g.step(torch.from_numpy(a).cuda())
        o.step(a)
        # integer / boolean outputs: bit-exact
        np.testing.assert_array_equal(_t(g, "RESET_BUF"), o.get("RESET_BUF"), err_msg=f"reset mask, step {step}")
        np.testing.assert_array_equal(_t(g, "TIME_OUT_BUF"), o.get("TIME_OUT_BUF"))
        np.testing.assert_array_equal(_t(g, "EPISODE_LENGTH"), o.get("EPISODE_LENGTH"))
        resets += int(o.get("RESET_BUF").sum())
        for name in ("DOF_STATE", "ROOT_STATES", "TORQUES", "COMMANDS", "GOAL_STATE", "ACTIONS", "ACTION_HISTORY",
                     "BASE_LIN_VEL", "BASE_ANG_VEL", "LAST_DOF_VEL", "LAST_ROOT_VEL", "LAST_ACTIONS"):
            np.testing.assert_allclose(_t(g, name), o.get(name), atol=atol, rtol=5e-4, err_msg=f"{name}, step {step}")
        np.testing.assert_allclose(_t(g, "RIGID_BODY_STATE"), o.get("RIGID_BODY_STATE"), atol=atol, rtol=1e-4)
        np.testing.assert_allclose(_t(g, "OBS_BUF"), o.get("OBS_BUF"), atol=5 * atol, rtol=1e-4, err_msg=f"obs, step {step}")
        np.testing.assert_allclose(_t(g, "OBS_HISTORY"), o.get("OBS_HISTORY"), atol=5 * atol, rtol=1e-4)
        np.testing.assert_allclose(_t(g, "REW_BUF"), o.get("REW_BUF"), atol=2e-4, rtol=2e-3, err_msg=f"rew, step {step}")
        np.testing.assert_allclose(_t(g, "ARM_REW_BUF"), o.get("ARM_REW_BUF"), atol=2e-5, rtol=1e-3)
        np.testing.assert_allclose(_t(g, "EPISODE_SUMS"), o.get("EPISODE_SUMS"), atol=2e-2, rtol=2e-3)
        np.testing.assert_allclose(_t(g, "FORCE_SENSOR"), o.get("FORCE_SENSOR"), atol=0.1, rtol=5e-3)
        m = o.get("RESET_BUF").astype(bool)        # the finished episode's travel / command norm (terrain curriculum inputs)
        np.testing.assert_allclose(_t(g, "RESET_TRAVEL")[m], o.get("RESET_TRAVEL")[m], atol=atol, rtol=1e-4)
    assert resets > 10     # resets (and therefore the reset path and its random draws) were exercised
    g.close()


@pytest.mark.parametrize("n", [1, 13])
def test_ragged_env_counts_match_oracle(robot, n):
    """The step launch rounds its grid up to a multiple of 8 and deals the envs to the XCDs in contiguous ranges: env counts that
    are not multiples of 8 (and a single env) run through the same kernel, every env is stepped exactly once (the episode length of
    every env advances) and the results are the oracle's."""
    import torch
    params = helpers.random_env_params(n, seed=9)
    g = helpers.make_gpu(robot, n, params)
    o = helpers.make_oracle(robot, n, params, "f64")
    g.reset_all(); o.reset_all()
    rng = np.random.default_rng(23)
    for step in range(12):
        helpers.sync_oracle_from_gpu(o, g)
        a = (0.6 * rng.normal(size=(n, 18))).astype(np.float32)
        before = _t(g, "EPISODE_LENGTH").copy()
        g.step(torch.from_numpy(a).cuda())
        o.step(a)
        np.testing.assert_array_equal(_t(g, "RESET_BUF"), o.get("RESET_BUF"))
        np.testing.assert_array_equal(_t(g, "EPISODE_LENGTH"), o.get("EPISODE_LENGTH"))
        after = _t(g, "EPISODE_LENGTH")
        assert np.all((after == before + 1) | (after == 0))            # every env took exactly one step
        for name in ("DOF_STATE", "ROOT_STATES", "TORQUES", "GOAL_STATE", "COMMANDS"):
            np.testing.assert_allclose(_t(g, name), o.get(name), atol=3e-4, rtol=5e-4, err_msg=f"{name}, step {step}")
        np.testing.assert_allclose(_t(g, "OBS_BUF"), o.get("OBS_BUF"), atol=1.5e-3, rtol=1e-4)
    g.close()


def test_free_running_rollout_stays_close(robot):
    """No state syncing: 8 policy steps (32 substeps) of the HIP path vs the fp64 oracle."""
    import torch
    n = 32
    params = helpers.random_env_params(n, seed=9)
    tc = copy.copy(robot["tcfg"])
    tc.term_z_threshold = 0.05        # keep the episodes alive: this test is about the dynamics
    tc.term_rp_threshold = 10.0
    g = helpers.make_gpu(robot, n, params, tcfg=tc)
    o = helpers.make_oracle(robot, n, params, "f64", tcfg=tc)
    g.reset_all(); o.reset_all()
    rng = np.random.default_rng(23)
    for step in range(8):
        a = (0.3 * rng.normal(size=(n, 18))).astype(np.float32)
        g.step(torch.from_numpy(a).cuda()); o.step(a)
    assert np.isfinite(_t(g, "OBS_BUF")).all()
    err = np.abs(_t(g, "DOF_STATE")[:, :, 0] - o.get("DOF_STATE")[:, :, 0]).max(1)
    # chaotic contact switching may let a few envs drift; the bulk must agree to fp32 accuracy
    assert np.median(err) < 2e-4 and np.quantile(err, 0.9) < 5e-3
    np.testing.assert_array_equal(_t(g, "EPISODE_LENGTH"), o.get("EPISODE_LENGTH"))
    g.close()


def test_state_setters_and_fk_refresh(robot):
    import torch
    params = helpers.random_env_params(N, seed=2)
    g = helpers.make_gpu(robot, N, params)
    o = helpers.make_oracle(robot, N, params, "f64")
    rng = np.random.default_rng(4)
    root, dof = helpers.random_standing_state(N, robot["tcfg"], rng)
    g.set_root_state(torch.from_numpy(root).cuda())
    g.set_dof_state(torch.from_numpy(dof).cuda())
    g.refresh_rigid_body_state()
    o.set("ROOT_STATES", root); o.set("DOF_STATE", dof); o.refresh_rigid_body_state()
    np.testing.assert_allclose(_t(g, "RIGID_BODY_STATE"), o.get("RIGID_BODY_STATE"), atol=2e-5, rtol=1e-5)
    # indexed setters touch only the listed envs
    root2 = root.copy(); root2[:, 0, 2] += 1.0
    ids = torch.tensor([1, 5, 7], device="cuda")
    g.set_root_state_indexed(torch.from_numpy(root2).cuda(), ids)
    got = _t(g, "ROOT_STATES")
    mask = np.zeros(N, bool); mask[[1, 5, 7]] = True
    np.testing.assert_allclose(got[mask, 0, 2], root2[mask, 0, 2], rtol=1e-6)
    np.testing.assert_allclose(got[~mask, 0, 2], root[~mask, 0, 2], rtol=1e-6)
    g.close()


def _make_heightfield(rows=80, cols=120, seed=0):
    """A small rough height grid (int16 samples, as Terrain builds them: utils/terrain.py:51) around the origin."""
    rng = np.random.default_rng(seed)
    x = np.linspace(0, 4 * np.pi, rows)[:, None]
    y = np.linspace(0, 6 * np.pi, cols)[None, :]
    h = 0.04 * np.sin(x) * np.cos(y) + 0.01 * rng.standard_normal((rows, cols))
    return np.round(h / 0.005).astype(np.int16)          # vertical_scale 0.005 (legged_robot_config.py:45)


def test_heightfield_contact_matches_oracle(robot):
    """BASELINE config 3's contact path: spheres against the triangulated height grid (terrain indexing by
    truncation + clipping as LR:816-829 is integer work and must agree exactly; forces to tolerance)."""
    import torch
    n = 48
    params = helpers.random_env_params(n, seed=12)
    params["env_origins"] = np.zeros((n, 3), dtype=np.float32)
    hf = _make_heightfield()
    hs, vs = 0.1, 0.005
    tx, ty, tz = -0.5 * hf.shape[0] * hs, -0.5 * hf.shape[1] * hs, 0.0
    g = helpers.make_gpu(robot, n, params)
    o = helpers.make_oracle(robot, n, params, "f64")
    g.set_heightfield(hf, hs, vs, tx, ty, tz)
    o.set_heightfield(hf, hs, vs, tx, ty, tz)
    rng = np.random.default_rng(31)
    root, dof = helpers.random_standing_state(n, robot["tcfg"], rng, height=(0.28, 0.40))
    root[:, 0, 0:2] = rng.uniform(-3.5, 3.5, (n, 2))
    root[:5, 0, 0] = -100.0              # off the grid: indices clip to the border cells
    root[5:8, 0, 1] = 100.0
    tau = rng.uniform(-5, 5, (n, 20)).astype(np.float32)
    tau[:, 18:] = 0
    g.tensor("ROOT_STATES").copy_(torch.from_numpy(root)); g.tensor("DOF_STATE").copy_(torch.from_numpy(dof))
    g.set_dof_forces(torch.from_numpy(tau).cuda())
    o.set("ROOT_STATES", root); o.set("DOF_STATE", dof); o.set("TORQUES", tau)
    touched = 0
    for it in range(4):
        g.simulate(); o.simulate()
        fo = o.get("NET_CONTACT_FORCE")
        np.testing.assert_array_equal(np.abs(_t(g, "NET_CONTACT_FORCE")).sum(-1) > 0, np.abs(fo).sum(-1) > 0)   # same active set
        _rows_close(_t(g, "NET_CONTACT_FORCE"), fo, 0.08, 3e-3, f"heightfield substep {it} NET_CONTACT_FORCE")
        np.testing.assert_allclose(_t(g, "DOF_STATE"), o.get("DOF_STATE"), atol=4e-4, rtol=2e-4)
        np.testing.assert_allclose(_t(g, "ROOT_STATES")[:, 0], o.get("ROOT_STATES")[:, 0], atol=4e-4, rtol=2e-4)
        _rows_close(_t(g, "ROOT_STATES")[:, 1:, :], o.get("ROOT_STATES")[:, 1:, :], 2e-3, 2e-3, f"heightfield substep {it} box state")
        touched += (np.abs(fo).sum(-1) > 0).sum()
        helpers.sync_oracle_from_gpu(o, g, ["ROOT_STATES", "DOF_STATE"])
    assert touched > n
    # sloped contact: some force has a horizontal component through the terrain normal
    g.set_heightfield(None)
    g.close()


@pytest.mark.parametrize("n", [1, 3, 257])
def test_ragged_env_counts(robot, n):
    """Sizes that are not multiples of anything; N=1 is the minimum."""
    import torch
    params = helpers.random_env_params(n, seed=n)
    g = helpers.make_gpu(robot, n, params)
    o = helpers.make_oracle(robot, n, params, "f64")
    g.reset_all(); o.reset_all()
    rng = np.random.default_rng(n)
    for step in range(3):
        helpers.sync_oracle_from_gpu(o, g)
        a = (0.5 * rng.normal(size=(n, 18))).astype(np.float32)
        g.step(torch.from_numpy(a).cuda()); o.step(a)
        np.testing.assert_array_equal(_t(g, "RESET_BUF"), o.get("RESET_BUF"))
        np.testing.assert_allclose(_t(g, "OBS_BUF"), o.get("OBS_BUF"), atol=1.5e-3, rtol=5e-4)
        np.testing.assert_allclose(_t(g, "DOF_STATE"), o.get("DOF_STATE"), atol=3e-4, rtol=5e-4)
    g.close()


def test_action_clip_timeout_and_episode_boundaries(robot):
    """Edge cases the reference's logic has: actions beyond +-clip_actions, episodes at max length (time-out bootstrap
    flag, command resampling only for timed-out envs), goal timers at their resample threshold."""
    import torch
    n = 32
    params = helpers.random_env_params(n, seed=40)
    g = helpers.make_gpu(robot, n, params)
    o = helpers.make_oracle(robot, n, params, "f64")
    g.reset_all(); o.reset_all()
    torch.cuda.synchronize()
    ep = np.zeros(n); ep[:8] = 500; ep[8:16] = 149; ep[16:24] = 499      # 500 -> 501 > max: time-out; 149 -> 150: command resample
    g.tensor("EPISODE_LENGTH").copy_(torch.from_numpy(ep).long())
    goal = _t(g, "GOAL_STATE"); goal[24:, 21] = goal[24:, 23]            # goal_timer at traj_total: resample next step
    g.tensor("GOAL_STATE").copy_(torch.from_numpy(goal).float())
    helpers.sync_oracle_from_gpu(o, g)
    a = np.full((n, 18), 250.0, dtype=np.float32); a[::2] *= -1          # clipped to +-100 (WGC:114)
    g.step(torch.from_numpy(a).cuda()); o.step(a)
    np.testing.assert_array_equal(_t(g, "TIME_OUT_BUF"), o.get("TIME_OUT_BUF"))
    assert _t(g, "TIME_OUT_BUF")[:8].all() and not _t(g, "TIME_OUT_BUF")[8:].any()
    np.testing.assert_array_equal(_t(g, "RESET_BUF"), o.get("RESET_BUF"))
    np.testing.assert_array_equal(_t(g, "EPISODE_LENGTH"), o.get("EPISODE_LENGTH"))
    assert np.abs(_t(g, "ACTION_HISTORY")[_t(g, "RESET_BUF") == 0]).max() <= 100.0
    for name in ("COMMANDS", "GOAL_STATE", "ACTION_HISTORY", "ROOT_STATES", "DOF_STATE"):
        np.testing.assert_allclose(_t(g, name), o.get(name), atol=5e-4, rtol=5e-4, err_msg=name)
    np.testing.assert_allclose(_t(g, "EPISODE_SUMS_DONE"), o.get("EPISODE_SUMS_DONE"), atol=2e-2, rtol=2e-3)
    g.close()


def test_collision_set_matches_oracle(robot):
    """The URDF's collision geometry beyond feet and knees: trunk-box corners and thigh tops against the
    ground, arm spheres against the trunk box and the front thighs (self-collision pairs: the impulse acts on both bodies).
    Robots dropped on their bellies with the legs folded away, robots lying on a side, robots whose arm starts inside the
    trunk / a thigh: HIP vs the fp64 oracle from a synced state, with terminate_after_contacts_on = thighs and the collision
    reward on trunk + thighs + calves active. Same active contact set, forces / state to tolerance, masks bit-exact."""
    import torch
    n = 384
    params = helpers.random_env_params(n, seed=61)
    params["env_origins"] = np.zeros((n, 3), dtype=np.float32)
    tc = type(robot["tcfg"]).from_buffer_copy(robot["tcfg"])
    names = robot["model"].rb_names
    tc.term_z_threshold = 0.02
    tc.term_contact_rb_mask = sum(1 << i for i, nm in enumerate(names) if "thigh" in nm)
    tc.penalize_contact_rb_mask = sum(1 << i for i, nm in enumerate(names) if any(k in nm for k in ("thigh", "trunk", "calf")))
    from wbc_amd.curriculum import make_curriculum
    cfg = copy.deepcopy(robot["cfg"])
    cfg.rewards.scales.collision = -1.0
    cur = make_curriculum(cfg, 1)
    g = helpers.make_gpu(robot, n, params, tcfg=tc)
    o = helpers.make_oracle(robot, n, params, "f64", tcfg=tc)
    g.set_curriculum(cur); o.set_curriculum(cur)
    g.reset_all()
    torch.cuda.synchronize()
    rng = np.random.default_rng(62)
    root = _t(g, "ROOT_STATES").astype(np.float32)
    dof = _t(g, "DOF_STATE").astype(np.float32)
    third = n // 3
    root[:third, 0, 2] = rng.uniform(0.058, 0.08, third)                     # on the belly, legs folded up
    for leg in range(4):
        dof[:third, 3 * leg + 1, 0] = 2.9
        dof[:third, 3 * leg + 2, 0] = -2.7
    dof[:third, :, 1] = 0
    root[third:2 * third, 0, 2] = rng.uniform(0.10, 0.16, third)             # on a side
    sgn = rng.choice([-1.0, 1.0], third)
    root[third:2 * third, 0, 3] = 0.7071068 * sgn
    root[third:2 * third, 0, 6] = 0.7071068
    lo, hi = np.array(robot["model"].dof_lower[12:18]), np.array(robot["model"].dof_upper[12:18])
    lo[0], hi[0] = -1.5, 1.5
    dof[2 * third:, 12:18, 0] = rng.uniform(lo, hi, (n - 2 * third, 6))      # arms anywhere in their joint ranges (standing robots)
    g.tensor("ROOT_STATES").copy_(torch.from_numpy(root)); g.tensor("DOF_STATE").copy_(torch.from_numpy(dof))
    seen = np.zeros(27, dtype=np.int64)
    resets = airborne_arm = 0
    for step in range(5):
        helpers.sync_oracle_from_gpu(o, g)
        a = (0.6 * rng.normal(size=(n, 18))).astype(np.float32)
        g.step(torch.from_numpy(a).cuda()); o.step(a)
        tag = f"collision set, step {step}"
        for name in ("RESET_BUF", "TIME_OUT_BUF", "EPISODE_LENGTH"):
            np.testing.assert_array_equal(_t(g, name), o.get(name), err_msg=f"{tag} {name}")
        fo, fg = o.get("NET_CONTACT_FORCE"), _t(g, "NET_CONTACT_FORCE")
        m = o.get("RESET_BUF").astype(bool)
        live = ~m                                                              # (a reset zeroes nothing here, but its state is re-drawn)
        strong = np.abs(fo).sum(-1) > 0.5
        assert ((np.abs(fg).sum(-1) > 0) == (np.abs(fo).sum(-1) > 0))[strong | (np.abs(fo).sum(-1) == 0)].mean() > 0.999, tag
        _assert_close_bulk(fg, fo, 0.08, 5e-3, f"{tag} NET_CONTACT_FORCE", frac=2e-3)
        for name, atol, rtol in (("DOF_STATE", 6e-4, 1e-3), ("ROOT_STATES", 4e-4, 5e-4), ("TORQUES", 4e-4, 5e-4),
                                 ("REW_BUF", 2e-4, 2e-3), ("ARM_REW_BUF", 2e-5, 1e-3), ("OBS_BUF", 3e-3, 1e-3)):
            _close(name, _t(g, name), o.get(name), atol, rtol, f"{tag} {name}", frac=2e-3)
        np.testing.assert_allclose(_t(g, "EPISODE_SUMS")[live][:, 21], o.get("EPISODE_SUMS")[live][:, 21], atol=1e-6)   # the collision counts
        seen += (np.abs(fo[:, :27]).sum(-1) > 0).sum(0)
        resets += int(m.sum())
        rbz = o.get("RIGID_BODY_STATE")[:, 20:25, 2]
        airborne_arm += int(((np.abs(fo[:, 20:25]).sum(-1) > 0) & (rbz > 0.08)).sum())
    assert seen[1] > 100 and seen[[3, 7, 11, 15]].sum() > 20 and airborne_arm > 20 and resets > 50, (seen, airborne_arm, resets)
    g.close()


def test_self_collision_candidates_match_oracle(robot):
    """Self-collision as configured (asset.self_collisions = 0: every pair of non-adjacent links): the broad phase on all 64 lanes,
    the promotion of its hits into dynamic contact slots and the exact limb-vs-limb / sphere-vs-box tests, HIP vs the fp64 oracle
    from a synced state. Staged: robots in the air with EVERY joint drawn uniformly inside its limits (legs crossing, the arm in the
    legs: up to a dozen simultaneous pairs), robots near the stance with one leg swung into its neighbour, and free boxes placed
    at knees, shins and under the trunk (the candidates beyond the five static box pairs). Same contact forces per rigid body (the
    set of bodies in contact included), state to tolerance, masks bit-exact."""
    import torch
    n = 512
    params = helpers.random_env_params(n, seed=71)
    params["env_origins"] = np.zeros((n, 3), dtype=np.float32)
    tc = type(robot["tcfg"]).from_buffer_copy(robot["tcfg"])
    tc.term_z_threshold = 0.02
    g = helpers.make_gpu(robot, n, params, tcfg=tc)
    o = helpers.make_oracle(robot, n, params, "f64", tcfg=tc)
    g.reset_all()
    torch.cuda.synchronize()
    rng = np.random.default_rng(72)
    model = robot["model"]
    names = model.rb_names
    root = _t(g, "ROOT_STATES").astype(np.float32)
    dof = _t(g, "DOF_STATE").astype(np.float32)
    lo, hi = np.array(model.dof_lower, dtype=np.float64), np.array(model.dof_upper, dtype=np.float64)
    free = ~(lo < hi)
    lo[free], hi[free] = -np.pi, np.pi
    lo[18:], hi[18:] = 0.0, 0.0
    half = n // 2
    root[:, 0, 2] = 1.2                                                        # in the air for the five steps: no terrain contact
    root[:, 0, 7:] = 0
    dof[:, :, 1] = 0
    dof[:half, :, 0] = rng.uniform(lo, hi, (half, 20))                          # every joint anywhere
    q = np.array(tc.default_dof_pos, dtype=np.float64)[None] + rng.uniform(-0.3, 0.3, (n - half, 20))
    leg = rng.integers(0, 4, n - half)
    for e in range(n - half):                                                  # one hip rolled inward, that leg swung at its neighbour
        q[e, 3 * leg[e]] = (-1 if leg[e] % 2 == 0 else 1) * rng.uniform(0.6, 1.04)
        q[e, 3 * leg[e] + 1] = rng.uniform(-0.6, 2.9)
    dof[half:, :, 0] = np.clip(q, lo, hi)
    dof[:, 18:, 0] = 0
    # boxes: at a knee / a shin / under the trunk of every third robot (else far away), at rest
    g.tensor("ROOT_STATES").copy_(torch.from_numpy(root)); g.tensor("DOF_STATE").copy_(torch.from_numpy(dof))
    g.refresh_rigid_body_state()
    torch.cuda.synchronize()
    rb = _t(g, "RIGID_BODY_STATE")
    root[:, 1, :3] = root[:, 0, :3] + np.array([3.0, 0.0, -1.15])
    root[:, 1, 3:7] = [0, 0, 0, 1]; root[:, 1, 7:] = 0
    targets = [names.index(k) for k in ("FL_calf", "FR_calf", "RL_calf", "RR_calf", "trunk")]
    for e in range(0, n, 3):
        t = targets[(e // 3) % 5]
        off = rng.uniform(-0.03, 0.03, 3) + (np.array([0, 0, -0.11]) if names[t] == "trunk" else np.array([0.05, 0, rng.choice([0.0, -0.1])]))
        root[e, 1, :3] = rb[e, t, :3] + off
        root[e, 1, 3:7] = helpers.random_quat(rng)
    g.tensor("ROOT_STATES").copy_(torch.from_numpy(root))
    seen = np.zeros(28, dtype=np.int64)
    legleg = boxhits = 0
    legs_rb = [i for i, nm in enumerate(names) if any(k in nm for k in ("thigh", "calf", "foot"))]
    for step in range(4):
        helpers.sync_oracle_from_gpu(o, g)
        a = (0.3 * rng.normal(size=(n, 18))).astype(np.float32)
        g.step(torch.from_numpy(a).cuda()); o.step(a)
        tag = f"self-collision candidates, step {step}"
        for name in ("RESET_BUF", "TIME_OUT_BUF", "EPISODE_LENGTH"):
            np.testing.assert_array_equal(_t(g, name), o.get(name), err_msg=f"{tag} {name}")
        fo, fg = o.get("NET_CONTACT_FORCE"), _t(g, "NET_CONTACT_FORCE")
        strong = np.abs(fo).sum(-1) > 0.5
        assert ((np.abs(fg).sum(-1) > 0) == (np.abs(fo).sum(-1) > 0))[strong | (np.abs(fo).sum(-1) == 0)].mean() > 0.998, tag
        _assert_close_bulk(fg, fo, 0.08, 5e-3, f"{tag} NET_CONTACT_FORCE", frac=4e-3, slack=1e3)
        _assert_close_bulk(_t(g, "FORCE_SENSOR"), o.get("FORCE_SENSOR"), 0.08, 5e-3, f"{tag} FORCE_SENSOR", frac=4e-3, slack=1e3)
        for name, atol, rtol in (("DOF_STATE", 6e-4, 1e-3), ("ROOT_STATES", 4e-4, 5e-4), ("TORQUES", 4e-4, 5e-4), ("OBS_BUF", 3e-3, 1e-3)):
            _close(name, _t(g, name), o.get(name), atol, rtol, f"{tag} {name}", frac=4e-3)
        hit = np.abs(fo).sum(-1) > 0
        seen += hit.sum(0)
        legleg += int((hit[:, legs_rb].sum(1) >= 2).sum())
        boxhits += int((hit[:, 27] & (hit[:, [names.index(k) for k in ("FL_calf", "FR_calf", "RL_calf", "RR_calf", "trunk")]].any(1))).sum())
    assert legleg > 60 and boxhits > 100 and seen[[names.index(k) for k in ("wx250s/upper_forearm_link", "wx250s/wrist_link")]].sum() > 30, (legleg, boxhits, seen)
    # hits that found no free dynamic slot: the same count on both sides, env by env (uniform joint draws CAN tangle a robot beyond
    # the 17 slots; the free-run test asserts that a running robot never does)
    dg, do = _t(g, "DROPPED_HITS"), o.get("DROPPED_HITS")
    agree = (dg == do).mean()
    print("dropped broad-phase hits: HIP", float(dg.sum()), "oracle", float(np.asarray(do).sum()), "envs agreeing", agree)
    assert agree > 0.99
    g.close()


# ---- BASELINE.json configurations ----------------------------------------------------------------------------------
def _assert_close_bulk(a, b, atol, rtol, msg, frac=2e-5, slack=10.0):
    """allclose for batches of thousands of envs: every element within slack x the tolerance, and all but a fraction `frac`
    within the tolerance itself (a contact that switches one substep earlier in fp32 than in the oracle's arithmetic moves a
    handful of velocities by a few 1e-4; with 8192 envs x 40 DoF values one such env per step is expected)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = err > tol
    assert not (err > slack * tol).any(), f"{msg}: max err {err.max():.3e}"
    assert bad.mean() <= frac, f"{msg}: {bad.sum()} of {bad.size} outside tolerance (max err {err.max():.3e})"


def _close(name, got, want, atol, rtol, msg, **kw):
    """_assert_close_bulk; ROOT_STATES split into the robot's row (the given tolerance) and the box actor's (a 1 kg cube on corner
    contacts: a corner that meets a terrain edge one substep apart in fp32 moves it by millimetres per second)."""
    if name == "ROOT_STATES":
        _assert_close_bulk(got[:, 0], want[:, 0], atol, rtol, msg + " (robot)", **kw)
        _assert_close_bulk(got[:, 1], want[:, 1], 2e-3, 2e-3, msg + " (box)", frac=3e-3, slack=1e3)
    else:
        _assert_close_bulk(got, want, atol, rtol, msg, **kw)


def _synced_steps(robot, g, o, n, steps, rng, tag, check_travel=False):
    import torch
    resets = 0
    for step in range(steps):
        helpers.sync_oracle_from_gpu(o, g)
        a = (0.6 * rng.normal(size=(n, 18))).astype(np.float32)
        g.step(torch.from_numpy(a).cuda())
        o.step(a)
        for name in ("RESET_BUF", "TIME_OUT_BUF", "EPISODE_LENGTH"):                  # integer / boolean outputs: bit-exact
            np.testing.assert_array_equal(_t(g, name), o.get(name), err_msg=f"{tag} {name}, step {step}")
        m = o.get("RESET_BUF").astype(bool)
        resets += int(m.sum())
        for name, atol, rtol in (("DOF_STATE", 4e-4, 5e-4), ("ROOT_STATES", 4e-4, 5e-4), ("TORQUES", 4e-4, 5e-4), ("COMMANDS", 1e-6, 1e-6),
                                 ("GOAL_STATE", 2e-5, 2e-5), ("OBS_BUF", 2e-3, 5e-4), ("OBS_HISTORY", 2e-3, 5e-4),
                                 ("REW_BUF", 2e-4, 2e-3), ("ARM_REW_BUF", 2e-5, 1e-3), ("EPISODE_SUMS", 2e-2, 2e-3)):
            _close(name, _t(g, name), o.get(name), atol, rtol, f"{tag} {name}, step {step}")
        if check_travel and m.any():
            np.testing.assert_allclose(_t(g, "RESET_TRAVEL")[m], o.get("RESET_TRAVEL")[m], atol=4e-4, rtol=1e-4)
    return resets


def test_box_actor_matches_oracle(robot):
    """The free box actor (widowGo1.py:321-325,384,769-771: a 0.1 m cube of 1 kg + its mass draw, eight corner spheres against the
    terrain, the robot's foot spheres and gripper tip against it). Staged: boxes dropped from their spawn height, boxes thrown and
    spinning, boxes in the path of a front foot of a robot walking into them, boxes dropped onto the gripper; on the plane and on a
    height grid. HIP vs the fp64 oracle from a synced state: both rows of ROOT_STATES, the box row of NET_CONTACT_FORCE / of
    RIGID_BODY_STATE, the foot sensors (a foot against the box shows up there), masks bit-exact."""
    import torch
    n = 512
    params = helpers.random_env_params(n, seed=71)
    rng = np.random.default_rng(72)
    params["env_origins"] = np.stack([rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), np.zeros(n)], 1).astype(np.float32)
    params["box_dmass"] = rng.uniform(-0.001, 0.05, n).astype(np.float32)
    tc = type(robot["tcfg"]).from_buffer_copy(robot["tcfg"])
    tc.term_z_threshold = 0.15
    feet = list(robot["wmodel"].feet_rb)
    for use_hf in (False, True):
        g = helpers.make_gpu(robot, n, params, tcfg=tc)
        o = helpers.make_oracle(robot, n, params, "f64", tcfg=tc)
        np.testing.assert_allclose(_t(g, "BOX_MASS"), o.get("BOX_MASS"), rtol=1e-6)
        if use_hf:
            hf = _make_heightfield(rows=200, cols=200, seed=9)
            hs, vs = 0.1, 0.005
            args = (hf, hs, vs, -0.5 * hf.shape[0] * hs, -0.5 * hf.shape[1] * hs, 0.0)
            g.set_heightfield(*args); o.set_heightfield(*args)
        g.reset_all()
        torch.cuda.synchronize()
        root = _t(g, "ROOT_STATES").astype(np.float32)
        dof = _t(g, "DOF_STATE").astype(np.float32)
        q = n // 4
        ground = root[:, 0, 2] - 0.42                                          # nominal ground height under the robot (spawn height 0.42)
        A, B, Cc, D = slice(0, q), slice(q, 2 * q), slice(2 * q, 3 * q), slice(3 * q, n)
        # (a) dropped from the spawn height (WG:769-771 puts the box at world x = 0; here: within a metre of the robot)
        root[A, 1, 0] = root[A, 0, 0] + rng.uniform(0.5, 1.0, q)
        root[A, 1, 2] = ground[A] + float(tc.box_origin_z)
        # (b) thrown and spinning
        root[B, 1, :3] = root[B, 0, :3] + np.stack([rng.uniform(0.6, 1.2, q), rng.uniform(-0.5, 0.5, q), rng.uniform(-0.3, 0.0, q)], 1)
        root[B, 1, 7:10] = rng.uniform(-1.5, 1.5, (q, 3))
        root[B, 1, 10:13] = rng.uniform(-8, 8, (q, 3))
        ang = rng.normal(size=(q, 4))
        root[B, 1, 3:7] = ang / np.linalg.norm(ang, axis=1, keepdims=True)
        # (c) in the path of a front foot (default stance: front feet about 0.19 m ahead and 0.13 m to the side of the base)
        side = rng.choice([-1.0, 1.0], q)
        root[Cc, 1, 0] = root[Cc, 0, 0] + 0.19 + 0.05 + rng.uniform(0.0, 0.04, q)
        root[Cc, 1, 1] = root[Cc, 0, 1] + side * 0.13 + rng.uniform(-0.04, 0.04, q)
        root[Cc, 1, 2] = ground[Cc] + 0.05 + (0.06 if use_hf else 0.0)
        root[Cc, 0, 7] = rng.uniform(0.3, 1.0, q)
        # (d) dropped onto the gripper: the arm stretched out in front of the robot, the box released just above its tip
        dof[D, 13, 0] = rng.uniform(1.2, 1.6, n - 3 * q)
        dof[D, 14, 0] = rng.uniform(-0.6, -0.2, n - 3 * q)
        g.tensor("DOF_STATE").copy_(torch.from_numpy(dof))
        g.tensor("ROOT_STATES").copy_(torch.from_numpy(root))
        g.refresh_rigid_body_state()
        ee = _t(g, "RIGID_BODY_STATE")[D, robot["wmodel"].gripper_rb, :3]
        root[D, 1, :3] = ee + np.stack([rng.uniform(-0.03, 0.03, n - 3 * q), rng.uniform(-0.03, 0.03, n - 3 * q),
                                        0.062 + rng.uniform(-0.004, 0.01, n - 3 * q)], 1)
        # sleeping (PhysX-style: frozen after 0.4 s at rest): half of the dropped boxes start at rest on the plane instead, asleep,
        # and the boxes in the feet's path start asleep too -- the foot wakes them
        timer = np.zeros(n, dtype=np.float32)
        if not use_hf:
            root[:q // 2, 1, 2] = ground[:q // 2] + 0.05
            timer[:q // 2] = 80
        timer[Cc] = 80
        g.tensor("ROOT_STATES").copy_(torch.from_numpy(root))
        g.tensor("BOX_SLEEP_TIMER").copy_(torch.from_numpy(timer))
        foot_hits = grip_hits = box_steps = slept = 0
        for step in range(6):
            helpers.sync_oracle_from_gpu(o, g)
            root_before = _t(g, "ROOT_STATES").astype(np.float64)
            a = (0.3 * rng.normal(size=(n, 18))).astype(np.float32)
            g.step(torch.from_numpy(a).cuda()); o.step(a)
            tag = f"box actor ({'grid' if use_hf else 'plane'}), step {step}"
            for name in ("RESET_BUF", "TIME_OUT_BUF", "EPISODE_LENGTH"):
                np.testing.assert_array_equal(_t(g, name), o.get(name), err_msg=f"{tag} {name}")
            fo, fg = o.get("NET_CONTACT_FORCE"), _t(g, "NET_CONTACT_FORCE")
            ro, rg = o.get("ROOT_STATES"), _t(g, "ROOT_STATES")
            # Rows are split by whether kernel and oracle ended the step with the SAME set of rigid bodies in contact. Same set: every
            # element within 10 x its tolerance (and all but the stated fraction within it). Different set -- a box corner or a foot that
            # touches one substep apart in fp32 and in fp64 --: a stated fraction of the rows at most, and what they may differ by is
            # bounded physically: one of the two has applied an impact the other has not yet, i.e. a velocity change of at most the
            # box's closing speed before the step (|v| + |omega| x half diagonal, the robot's base speed for the foot pair) plus one
            # depenetration step, over at most one policy step of travel.
            same = ((np.abs(fg).sum(-1) > 0) == (np.abs(fo).sum(-1) > 0)).all(1)
            ndiff = int((~same).sum())
            assert ndiff <= 0.03 * n, f"{tag}: contact sets differ in {ndiff} of {n} envs"
            vclose = (np.linalg.norm(root_before[:, 1, 7:10], axis=1) + 0.0866 * np.linalg.norm(root_before[:, 1, 10:13], axis=1)
                      + np.linalg.norm(root_before[:, 0, 7:10], axis=1) + float(tc.max_depenetration_vel) + 9.81 * 0.02)
            worst = {}

            def split(name, got, want, atol, rtol, frac, dbound):
                got, want = np.asarray(got, np.float64).reshape(n, -1), np.asarray(want, np.float64).reshape(n, -1)
                if same.any():
                    _assert_close_bulk(got[same], want[same], atol, rtol, f"{tag} {name} (same contact set)", frac=frac, slack=10.0)
                if ndiff:
                    err = np.abs(got[~same] - want[~same]).max(1)
                    lim = np.broadcast_to(np.asarray(dbound, np.float64), (n,))[~same] if np.ndim(dbound) else np.full(ndiff, float(dbound))
                    worst[name] = float((err / lim).max())
                    assert (err <= lim).all(), f"{tag} {name} (different contact set): {err.max():.3e} against the bound {lim[err.argmax()]:.3e}"
            wmax = float(np.abs(fo).max(initial=1.0))
            split("NET_CONTACT_FORCE", fg, fo, 0.08, 5e-3, 4e-3, 2.0 * wmax + 50.0)             # (a contact more or less: up to the largest force in the batch)
            _assert_close_bulk(rg[:, 0], ro[:, 0], 4e-4, 5e-4, f"{tag} ROOT_STATES robot", frac=2e-3)
            # the box: 1 kg on 0.1 m -- a corner that touches one substep apart in fp32 moves it visibly: the bulk must be tight
            split("box pose", rg[:, 1, :7], ro[:, 1, :7], 4e-4, 5e-4, 1e-2, 0.02 * vclose + 0.05)
            split("box velocity", rg[:, 1, 7:10], ro[:, 1, 7:10], 2e-3, 2e-3, 2e-2, vclose)
            split("box spin", rg[:, 1, 10:], ro[:, 1, 10:], 3e-2, 5e-3, 2e-2, vclose / 0.05 + 1.0)
            np.testing.assert_allclose(_t(g, "RIGID_BODY_STATE")[:, 27], rg[:, 1], atol=1e-6)
            split("DOF_STATE", _t(g, "DOF_STATE"), o.get("DOF_STATE"), 6e-4, 1e-3, 4e-3, 2.0)       # (a 1 kg box against a 12 kg robot's limb)
            split("FORCE_SENSOR", _t(g, "FORCE_SENSOR"), o.get("FORCE_SENSOR"), 0.08, 5e-3, 4e-3, 2.0 * wmax + 50.0)
            split("OBS_BUF", _t(g, "OBS_BUF"), o.get("OBS_BUF"), 3e-3, 1e-3, 4e-3, 2.0)
            split("REW_BUF", _t(g, "REW_BUF"), o.get("REW_BUF"), 2e-4, 2e-3, 4e-3, 0.05)
            if ndiff:
                print(f"{tag}: {ndiff} envs with different contact sets; worst error / physical bound:", {k: round(v, 3) for k, v in worst.items()})
            to, tg = o.get("BOX_SLEEP_TIMER"), _t(g, "BOX_SLEEP_TIMER")
            assert (to != tg).mean() < 0.01, f"{tag} BOX_SLEEP_TIMER: {(to != tg).sum()} differ"
            frozen = (to == 80) & (np.abs(fo[:, 27]).sum(-1) == 0)
            assert np.abs(rg[frozen & (tg == 80), 1, 7:]).max(initial=0.0) == 0.0, f"{tag}: a sleeping box moves"
            slept += int(frozen.sum())
            box_steps += int((np.abs(fo[:, 27]).sum(-1) > 0).sum())
            # a front foot pushed along x together with the box / the gripper pushed up with the box loaded: the pairs at work
            foot_hits += int(((np.abs(fo[Cc][:, feet[:2], 0]) > 1.0).any(-1) & (np.abs(fo[Cc][:, 27, 0]) > 1.0)).sum())
            grip_hits += int(((fo[D][:, robot["wmodel"].gripper_rb, 2] < -0.2) & (fo[D][:, 27, 2] > 0.2)).sum())
        assert box_steps > n and foot_hits > 20 and grip_hits > 20, (box_steps, foot_hits, grip_hits)
        assert use_hf or slept > 100, slept
        g.close()


def test_step_at_baseline_config1_4096_flat(robot):
    """BASELINE.json configs[1] (the bench workload): 4096 envs on the plane, the fused step against the fp32 oracle from a
    synced state, after a settling rollout so that the batch holds standing, falling and freshly reset robots."""
    import torch
    n = 4096
    params = helpers.random_env_params(n, seed=41)
    g = helpers.make_gpu(robot, n, params)
    o = helpers.make_oracle(robot, n, params, "f32")
    g.reset_all()
    rng = np.random.default_rng(42)
    g.tensor("EPISODE_LENGTH").copy_(torch.from_numpy(rng.integers(0, 500, n)).long())   # OPR:107-108 init_at_random_ep_len
    g.step_counter = 140
    for _ in range(12):
        g.step(torch.from_numpy((0.6 * rng.normal(size=(n, 18))).astype(np.float32)).cuda())
    o.step_counter = g.step_counter
    resets = _synced_steps(robot, g, o, n, 4, rng, "4096 flat")
    assert resets > 20
    g.close()


def test_step_at_baseline_config2_8192_heightfield(robot):
    """BASELINE.json configs[2]: 8192 envs on a height grid (sphere-vs-triangulated-grid contact, integer cell indexing) with
    the terrain curriculum's inputs (the finished episode's travel and command norm, LR:431-435) compared on every reset."""
    import torch
    n = 8192
    params = helpers.random_env_params(n, seed=43)
    rng = np.random.default_rng(44)
    hf = _make_heightfield(rows=400, cols=400, seed=5)
    hs, vs = 0.1, 0.005
    tx, ty, tz = -0.5 * hf.shape[0] * hs, -0.5 * hf.shape[1] * hs, 0.0
    params["env_origins"] = np.stack([rng.uniform(-15, 15, n), rng.uniform(-15, 15, n), np.zeros(n)], 1).astype(np.float32)
    g = helpers.make_gpu(robot, n, params)
    o = helpers.make_oracle(robot, n, params, "f32")
    g.set_heightfield(hf, hs, vs, tx, ty, tz)
    o.set_heightfield(hf, hs, vs, tx, ty, tz)
    g.reset_all()
    g.tensor("EPISODE_LENGTH").copy_(torch.from_numpy(rng.integers(0, 500, n)).long())
    for _ in range(10):
        g.step(torch.from_numpy((0.6 * rng.normal(size=(n, 18))).astype(np.float32)).cuda())
    o.step_counter = g.step_counter
    resets = _synced_steps(robot, g, o, n, 3, rng, "8192 heightfield", check_travel=True)
    assert resets > 20
    g.close()


def test_step_at_baseline_config2_8192_on_the_terrain_grid_with_curriculum(robot):
    """BASELINE.json configs[2] AS THE BENCH RUNS IT (`bench.py --terrain grid`): 8192 envs on the base class's sub-terrain grid
    (10 levels x 20 types of 8 m tiles: slopes, 5-23 cm stairs, discrete obstacles; utils/terrain.py:101-227) with
    terrain.curriculum=True, stepped through WidowGo1.step -- the fused kernel + _apply_terrain_curriculum -- against the fp32
    oracle + the rule of LR:421-441 in numpy, from a synced state. The robots are scattered over their tiles first so that feet
    meet stair edges, not only the flat platforms. Masks, episode lengths and terrain levels bit-exact; state to tolerance."""
    import torch
    from wbc_amd.config import WidowGo1RoughCfg, use_grid_terrain
    from wbc_amd.envs import WidowGo1
    from wbc_amd import abi
    from oracle import OracleSim, default_curriculum
    n = 8192
    cfg = use_grid_terrain(WidowGo1RoughCfg())
    cfg.env.num_envs = n
    env = WidowGo1(cfg, sim_device="cuda:0", seed=21)
    assert cfg.terrain.curriculum and env.terrain.heightsamples.shape == (1300, 2100)
    env.reset()
    rng = np.random.default_rng(45)
    # scatter: up to 3.5 m from the platform centre, on the terrain surface under the new position (+ the spawn clearance)
    hs_grid, t = env.terrain.heightsamples, cfg.terrain
    off = rng.uniform(-3.5, 3.5, (n, 2))
    root = env.root_states.cpu().numpy().copy()
    xy = root[:, :2] + off
    ij = np.clip(((xy + t.border_size) / t.horizontal_scale).astype(np.int64), 0, np.array(hs_grid.shape) - 2)
    patch = np.stack([hs_grid[ij[:, 0] + a, ij[:, 1] + b] for a in (0, 1) for b in (0, 1)], 1).max(1) * t.vertical_scale
    root[:, :2], root[:, 2] = xy, patch + 0.40
    env.root_states.copy_(torch.from_numpy(root).cuda())
    env.episode_length_buf = torch.from_numpy(rng.integers(0, 500, n)).cuda()
    for _ in range(10):                                   # settle: standing on edges, falling, freshly reset, levels moving
        env.step(torch.from_numpy((0.6 * rng.normal(size=(n, 18))).astype(np.float32)).cuda())
    torch.cuda.synchronize()
    o = OracleSim(env.wmodel, env.tcfg, n, seed=21, precision="f32")
    o.set_curriculum(default_curriculum(cfg, env.update_counter))
    o.set_heightfield(hs_grid, env.terrain.horizontal_scale, env.terrain.vertical_scale, *env.terrain.transform)
    names = [nm for nm in abi.TENSOR_IDS if nm != "OBS_BUF"]
    origins = env.terrain_origins.cpu().numpy()
    types, max_level, L = env.terrain_types.cpu().numpy(), env.max_terrain_level, env.max_episode_length_s
    resets = moved = on_rough = 0
    for step in range(4):
        helpers.sync_oracle_from_gpu(o, env.sim, names)
        levels0 = env.terrain_levels.cpu().numpy().copy()
        a = (0.6 * rng.normal(size=(n, 18))).astype(np.float32)
        env.step(torch.from_numpy(a).cuda())
        o.step(a)
        torch.cuda.synchronize()
        tag = f"8192 terrain grid, step {step}"
        for name in ("RESET_BUF", "TIME_OUT_BUF", "EPISODE_LENGTH"):
            np.testing.assert_array_equal(_t(env.sim, name), o.get(name), err_msg=f"{tag} {name}")
        m = o.get("RESET_BUF").astype(bool)
        travel = o.get("RESET_TRAVEL")
        up = m & (travel[:, 0] > env.terrain.env_length / 2)                                       # LR:431-435
        down = m & (travel[:, 0] < travel[:, 1] * L * 0.5) & ~up
        lv = levels0 + up.astype(np.int64) - down.astype(np.int64)
        wrapped = m & (lv >= max_level)                                                           # LR:438: a random level
        want = np.where(wrapped, -1, np.clip(lv, 0, None))
        got = env.terrain_levels.cpu().numpy()
        np.testing.assert_array_equal(got[~wrapped], want[~wrapped], err_msg=f"{tag} terrain levels")
        assert ((got[wrapped] >= 0) & (got[wrapped] < max_level)).all()
        new_origin = origins[got, types]
        np.testing.assert_array_equal(_t(env.sim, "ENV_ORIGINS"), new_origin.astype(np.float32))
        # the oracle's robots were reset relative to the OLD origin; move them as _apply_terrain_curriculum moved the kernel's
        oroot = o.get("ROOT_STATES")
        delta = new_origin - o.get("ENV_ORIGINS")
        oroot[:, 0, :3] += delta
        oroot[:, 1, 1] += delta[:, 1]
        for name, atol, rtol in (("DOF_STATE", 4e-4, 5e-4), ("TORQUES", 4e-4, 5e-4), ("COMMANDS", 1e-6, 1e-6), ("GOAL_STATE", 2e-5, 2e-5),
                                 ("OBS_BUF", 2e-3, 5e-4), ("OBS_HISTORY", 2e-3, 5e-4), ("REW_BUF", 2e-4, 2e-3), ("ARM_REW_BUF", 2e-5, 1e-3),
                                 ("EPISODE_SUMS", 2e-2, 2e-3)):
            # (stairs: a shin or a foot that meets an edge one substep apart in fp32 -- a handful of envs per step, bounded loosely)
            _assert_close_bulk(_t(env.sim, name), o.get(name), atol, rtol, f"{tag} {name}", frac=1e-4, slack=100.0)
        _close("ROOT_STATES", _t(env.sim, "ROOT_STATES"), oroot, 4e-4, 5e-4, f"{tag} ROOT_STATES", frac=1e-4, slack=100.0)
        if m.any():
            np.testing.assert_allclose(_t(env.sim, "RESET_TRAVEL")[m], travel[m], atol=4e-4, rtol=1e-4)
        resets += int(m.sum()); moved += int((got != levels0).sum())
        fz = np.abs(o.get("NET_CONTACT_FORCE")[:, :, :2]).sum((1, 2))                              # horizontal contact force: an edge or a slope
        on_rough += int((fz > 1.0).sum())
    assert resets > 40 and moved > 5 and on_rough > 200
    env.sim.close()


def test_free_running_100_policy_steps(robot):
    """state after 1, 4 and 400 substeps without any syncing (fp32 HIP vs fp64 oracle). Contact
    switching makes the trajectories diverge exponentially from fp32 rounding, so the statement is about quantiles."""
    import torch
    n = 128
    params = helpers.random_env_params(n, seed=19)
    tc = copy.copy(robot["tcfg"])
    tc.term_z_threshold = 0.05        # keep the episodes alive: this test is about the dynamics
    tc.term_rp_threshold = 10.0
    tc.max_episode_length = 100000
    g = helpers.make_gpu(robot, n, params, tcfg=tc)
    o = helpers.make_oracle(robot, n, params, "f64", tcfg=tc)
    g.reset_all(); o.reset_all()
    rng = np.random.default_rng(29)
    report = {}
    for step in range(1, 101):
        a = (0.25 * rng.normal(size=(n, 18))).astype(np.float32)
        g.step(torch.from_numpy(a).cuda()); o.step(a)
        if step in (1, 4, 25, 100):
            err = np.abs(_t(g, "DOF_STATE")[:, :, 0] - o.get("DOF_STATE")[:, :, 0]).max(1)
            zerr = np.abs(_t(g, "ROOT_STATES")[:, 0, 2] - o.get("ROOT_STATES")[:, 0, 2])
            report[step] = (np.median(err), np.quantile(err, 0.9), np.median(zerr))
    print("free run: policy step -> (median, p90 max joint error [rad], median base height error [m])",
          {k: tuple(f"{x:.2e}" for x in v) for k, v in report.items()})
    assert np.isfinite(_t(g, "OBS_BUF")).all()
    assert report[1][0] < 5e-5 and report[1][1] < 5e-4          # 4 substeps
    assert report[4][0] < 2e-4 and report[4][1] < 5e-3          # 16 substeps
    assert report[100][0] < 0.25 and report[100][2] < 0.02      # 400 substeps: same gait regime, decorrelated joint phases allowed
    np.testing.assert_array_equal(_t(g, "EPISODE_LENGTH"), o.get("EPISODE_LENGTH"))
    g.close()


def test_free_run_statistics_match_the_fp64_oracle(robot):
    """What the trajectory-level free-run test above cannot pin (contact switching decorrelates joint phases within seconds):
    the STATISTICS of a long free run. 4096 envs (the bench size) x 500 policy steps = 10 s of simulated time each, the same seeds and the same
    action stream through the fp32 HIP kernel and the fp64 oracle, nothing synced in between, episodes ending by falls
    (z threshold lowered to 0.22 m so that robots survive landing and fall from the 0.6-sigma action noise instead) and by
    150-step time-outs. Compared: finished episodes, fall rate, time-out count, mean episode length and the mean reward of
    both channels, each within 4 standard errors of the oracle's own estimate plus 2 % -- the stated confidence interval."""
    import torch
    n, steps = 4096, 500
    params = helpers.random_env_params(n, seed=23)
    tc = copy.copy(robot["tcfg"])
    tc.term_z_threshold = 0.22
    tc.max_episode_length = 150
    g = helpers.make_gpu(robot, n, params, tcfg=tc)
    o = helpers.make_oracle(robot, n, params, "f64", tcfg=tc)
    g.reset_all(); o.reset_all()
    rng = np.random.default_rng(31)
    acts = (0.6 * rng.normal(size=(steps, n, 18))).astype(np.float32)
    dev_acts = torch.from_numpy(acts).cuda()
    # device side: accumulate without host round trips
    dsum = dict(rew=torch.zeros(n, device="cuda"), arm=torch.zeros(n, device="cuda"), resets=torch.zeros(n, device="cuda"),
                touts=torch.zeros(n, device="cuda"), lens=torch.zeros(n, device="cuda"), len2=torch.zeros(n, device="cuda"))
    for k in range(steps):
        ep = g.tensor("EPISODE_LENGTH").clone()
        g.step(dev_acts[k])
        r, t = g.tensor("RESET_BUF") != 0, g.tensor("TIME_OUT_BUF") != 0
        ln = (ep + 1).float() * r
        dsum["rew"] += g.tensor("REW_BUF"); dsum["arm"] += g.tensor("ARM_REW_BUF")
        dsum["resets"] += r; dsum["touts"] += t; dsum["lens"] += ln; dsum["len2"] += ln * ln
    torch.cuda.synchronize()
    hsum = {k: np.zeros(n) for k in dsum}
    for k in range(steps):
        ep = o.get("EPISODE_LENGTH").copy()
        o.step(acts[k])
        r, t = o.get("RESET_BUF") != 0, o.get("TIME_OUT_BUF") != 0
        ln = (ep + 1) * r
        hsum["rew"] += o.get("REW_BUF"); hsum["arm"] += o.get("ARM_REW_BUF")
        hsum["resets"] += r; hsum["touts"] += t; hsum["lens"] += ln; hsum["len2"] += ln * ln

    def stats(s):
        s = {k: (v.double().cpu().numpy() if hasattr(v, "cpu") else v) for k, v in s.items()}
        n_ep = s["resets"].sum()
        mean_len = s["lens"].sum() / n_ep
        se_len = np.sqrt(max(s["len2"].sum() / n_ep - mean_len ** 2, 0.0) / n_ep)
        per_env = lambda x: (x.mean() / steps, x.std() / steps / np.sqrt(n))      # noqa: E731  (mean per step, its standard error)
        falls = s["resets"] - s["touts"]
        return dict(episodes=(n_ep, np.sqrt(n_ep)), fall_rate=per_env(falls), timeouts=(s["touts"].sum(), np.sqrt(max(s["touts"].sum(), 1.0))),
                    mean_len=(mean_len, se_len), rew=per_env(s["rew"]), arm_rew=per_env(s["arm"]))
    hip, ora = stats(dsum), stats(hsum)
    print("free-run statistics (value, standard error): HIP", {k: tuple(f"{x:.5g}" for x in v) for k, v in hip.items()},
          "fp64 oracle", {k: tuple(f"{x:.5g}" for x in v) for k, v in ora.items()})
    assert ora["episodes"][0] > 5000 and ora["timeouts"][0] > 100 and ora["fall_rate"][0] > 1e-3       # a mix of both endings
    for key in ora:
        (a, _), (b, se) = hip[key], ora[key]
        assert abs(a - b) <= 4.0 * se + 0.02 * abs(b), (key, a, b, se)
    # no broad-phase hit was ever left without a dynamic contact slot (4096 envs x 500 steps x 4 substeps, on both sides): the
    # 17 + 3 slots are enough for what robots falling about under action noise produce
    assert float(g.tensor("DROPPED_HITS").sum().item()) == 0.0
    assert float(np.asarray(o.get("DROPPED_HITS")).sum()) == 0.0
    g.close()
