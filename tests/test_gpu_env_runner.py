"""GPU tests of the drop-in layer: WidowGo1 (reference surface: WG:49, BT:41-131) and OnPolicyRunner
(OPR:46-300) over the HIP path, including the shipped Perlin-trimesh terrain and the checkpoint format."""
import os

import numpy as np
import pytest
import torch

from wbc_amd.config import WidowGo1RoughCfg, WidowGo1RoughCfgPPO, class_to_dict
from wbc_amd.envs import WidowGo1
from wbc_amd.rsl_rl.runners import OnPolicyRunner

pytestmark = pytest.mark.gpu


def _cfg(n=64, plane=True):
    cfg = WidowGo1RoughCfg()
    cfg.env.num_envs = n
    if plane:
        cfg.terrain.mesh_type = "plane"
    else:
        cfg.terrain.tot_rows = 2000          # 50 m strip instead of 250 m: same generator, faster test
        cfg.terrain.transform_y = -cfg.terrain.tot_rows * cfg.terrain.horizontal_scale / 2
    return cfg


def test_env_surface_matches_reference_contract():
    env = WidowGo1(_cfg(), sim_params=None, physics_engine=None, sim_device="cuda:0", headless=True, seed=1)
    assert (env.num_envs, env.num_obs, env.num_privileged_obs, env.num_actions) == (64, 860, None, 18)
    assert env.num_dofs == 20 and env.num_bodies == 27 and env.max_episode_length == 500
    obs, priv = env.reset()
    assert obs.shape == (64, 860) and priv is None
    out = env.step(torch.zeros(64, 18, device="cuda"))
    assert len(out) == 6                                          # WG:1199
    obs, priv, rew, arm_rew, dones, infos = out
    assert rew.shape == arm_rew.shape == (64,) and dones.dtype == torch.int64
    assert infos["time_outs"].dtype == torch.bool and "episode" in infos
    for key in ("rew_survive", "rew_tracking_ee_sphere", "metric_tracking_ee_sphere", "coeff_lin_vel_x_upper_bound"):
        assert key in infos["episode"], key
    # tensor attributes the reference code reads (WG:522-556, 619-627)
    assert env.root_states.shape == (64, 13) and env.dof_pos.shape == (64, 20) and env.dof_vel.shape == (64, 20)
    assert env.contact_forces.shape == (64, 27, 3) and env.rigid_body_state.shape == (64, 27, 13)
    assert env.force_sensor_tensor.shape == (64, 4, 6) and env.torques.shape == (64, 20) and env.commands.shape == (64, 3)
    assert env.p_gains.shape == (18,) and env.default_dof_pos.shape == (20,) and env.ee_pos.shape == (64, 3)
    assert env.obs_history_buf.shape == (64, 10, 76) and env.action_history_buf.shape == (64, 4, 18)
    # views are live: the obs block layout of Appendix A
    np.testing.assert_allclose(obs[:, 82:100].cpu().numpy(), (env.motor_strength - 1).cpu().numpy(), atol=1e-6)
    # the runner re-binds episode_length_buf (OPR:107-108): the setter must write through to the sim tensor
    env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=500)
    assert (env.sim.tensor("EPISODE_LENGTH") == env.episode_length_buf).all()
    env.update_command_curriculum()
    assert env.update_counter == 1 and env.lin_vel_x_ranges[1] == pytest.approx(0.9)


def test_step_into_a_caller_buffer_is_the_same_step():
    """wbc_sim_step_to / WidowGo1.set_obs_output: the observation rows land in the caller's buffer (the rollout storage
    slot), bit-identical to what obs_buf gets from the plain step of an identically seeded env; one-shot."""
    envs = [WidowGo1(_cfg(n=96), sim_device="cuda:0", seed=5) for _ in range(2)]
    for e in envs:
        e.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    slots = torch.full((3, 96, 860), float("nan"), device="cuda")
    for i in range(6):
        a = 0.3 * torch.randn(96, 18, device="cuda", generator=g)
        ref = envs[0].step(a)[0]
        if i % 2 == 0:
            envs[1].set_obs_output(slots[i // 2])
        got = envs[1].step(a)[0]
        assert (got.data_ptr() == slots[i // 2].data_ptr()) == (i % 2 == 0)          # redirected once, then back to obs_buf
        assert torch.equal(got, ref)
    assert torch.equal(envs[1].rew_buf, envs[0].rew_buf) and torch.equal(envs[1].obs_history_buf, envs[0].obs_history_buf)
    assert torch.isfinite(slots).all()


def test_step_fills_the_rollout_reward_and_done_slots():
    """wbc_sim_step_rollout / WidowGo1.set_rollout_output = the plain step followed by wbc_rollout_store (PPO.process_env_step's
    tensor work, ppo.py:129-141) on an identically seeded env: same slots, time-out bootstrap included."""
    from wbc_amd.native import check, lib
    cfg = _cfg(n=192)
    cfg.env.episode_length_s = 0.08                       # 4 steps: time-outs (the bootstrap) before anything else ends an episode
    envs = [WidowGo1(cfg, sim_device="cuda:0", seed=8) for _ in range(2)]
    for e in envs:
        e.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    L, stream = lib(), torch.cuda.current_stream().cuda_stream
    timeouts = 0
    for i in range(40):
        a = 0.02 * torch.randn(192, 18, device="cuda", generator=g)     # the robots keep standing: episodes end by time-out
        values = torch.randn(192, 2, device="cuda", generator=g)
        _, _, rew, arm_rew, dones, infos = envs[0].step(a)
        ref_r = torch.empty(192, 2, device="cuda"); ref_d = torch.empty(192, 1, dtype=torch.uint8, device="cuda")
        check(L.wbc_rollout_store(rew.data_ptr(), arm_rew.data_ptr(), dones.data_ptr(), infos["time_outs"].data_ptr(), values.data_ptr(), 0.99,
                                  ref_r.data_ptr(), ref_d.data_ptr(), 192, stream), "wbc_rollout_store")
        got_r = torch.full((192, 2), float("nan"), device="cuda"); got_d = torch.full((192, 1), 7, dtype=torch.uint8, device="cuda")
        envs[1].set_rollout_output(values, 0.99, got_r, got_d)
        _, _, _, _, _, infos1 = envs[1].step(a)
        assert infos1["rollout_stored"] == got_r.data_ptr() and infos.get("rollout_stored") is None
        assert torch.equal(got_d, ref_d)
        np.testing.assert_allclose(got_r.cpu().numpy(), ref_r.cpu().numpy(), rtol=0, atol=1e-7)
        timeouts += int(infos["time_outs"].sum())
    assert timeouts > 100
    assert envs[1].step(a)[5]["rollout_stored"] is None   # one-shot


def test_perlin_terrain_rollout_stays_on_the_ground():
    env = WidowGo1(_cfg(n=128, plane=False), sim_device="cuda:0", seed=2)
    assert env.terrain is not None and env.height_samples.shape == (600, 2000)
    env.reset()
    for _ in range(30):
        obs, _, rew, arm_rew, dones, infos = env.step(0.2 * torch.randn(128, 18, device="cuda"))
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    z = env.root_states[:, 2]
    assert (z > 0.15).all() and (z < 0.8).all()                  # neither fell through the terrain nor flew off
    feet = env.rigid_body_state[:, env.feet_indices, 2]
    assert (feet > -0.05).all()
    assert (env.force_sensor_tensor.norm(dim=-1) > 1.5).any()    # feet do load the rough terrain


def test_runner_learn_save_load_roundtrip(tmp_path):
    env = WidowGo1(_cfg(n=128), sim_device="cuda:0", seed=3)
    train = class_to_dict(WidowGo1RoughCfgPPO())
    train["runner"]["num_steps_per_env"] = 8
    train["runner"]["save_interval"] = 1
    runner = OnPolicyRunner(env, train, log_dir=str(tmp_path), device="cuda:0")
    runner.learn(2, init_at_random_ep_len=True)                   # iteration 0 = DAgger update, iteration 1 = PPO update
    assert len(runner.history) == 2 and all(np.isfinite(h["mean_value_loss"]) for h in runner.history)
    assert runner.alg.counter == 2                                # both kinds of update advance the schedule counter (quirk L7)
    path = os.path.join(str(tmp_path), "model_2.pt")
    assert os.path.exists(path)
    ckpt = torch.load(path, map_location="cpu")
    assert set(ckpt) >= {"model_state_dict", "optimizer_state_dict", "iter", "infos"}   # reference format (OPR:276-282)
    assert len(ckpt["model_state_dict"]) == 41 and ckpt["iter"] == 2
    env2 = WidowGo1(_cfg(n=128), sim_device="cuda:0", seed=3)
    runner2 = OnPolicyRunner(env2, train, log_dir=None, device="cuda:0")
    runner2.load(path)
    assert runner2.current_learning_iteration == 2 and runner2.alg.counter == 2
    for (k, a), (_, b) in zip(runner.alg.actor_critic.state_dict().items(), runner2.alg.actor_critic.state_dict().items()):
        assert torch.equal(a, b), k
    policy = runner2.get_inference_policy(device="cuda:0")
    act = policy(env2.get_observations())
    assert act.shape == (128, 18) and torch.isfinite(act).all()


@pytest.mark.gpu
def test_runner_with_rccl_process_group_single_rank():
    """The sharded-learner code path on the real backend: a 1-rank RCCL ("nccl") group exercises every collective
    the multi-GPU run issues (parameter broadcast, advantage statistics, per-minibatch gradient all-reduce in the
    fused and the DAgger update) on device tensors; with one rank the results must equal the group-less run."""
    import torch.distributed as dist
    if not dist.is_initialized():
        import socket
        with socket.socket() as sk:                     # any free port: the suite may run next to other jobs
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        outs = []
        for group in (None, dist.group.WORLD):
            torch.manual_seed(3)
            env = WidowGo1(_cfg(256), sim_params=None, physics_engine=None, sim_device="cuda:0", headless=True, seed=11)
            train = class_to_dict(WidowGo1RoughCfgPPO())
            train["runner"]["num_steps_per_env"] = 8
            runner = OnPolicyRunner(env, train, log_dir=None, device="cuda:0", dist_group=group)
            runner.learn(3)                    # iteration 0 is a DAgger update, 1 and 2 are PPO updates
            outs.append(torch.cat([p.detach().flatten() for p in runner.alg.actor_critic.parameters()]).cpu().numpy())
        np.testing.assert_allclose(outs[1], outs[0], rtol=0, atol=1e-6)
    finally:
        dist.destroy_process_group()


def test_device_episode_tracker_equals_the_reference_host_bookkeeping():
    """wbc_runner_track_episodes (one launch per env step, rings read once per iteration) against OPR:140-154 executed
    literally on the host: running sums, deque(maxlen=100) of finished episodes in (step, env) order, done fractions --
    with more than 100 episodes finishing in single steps and fewer than 100 in total at the start."""
    import statistics
    from collections import deque
    from wbc_amd.rsl_rl.runners.on_policy_runner import _EpisodeTracker
    n = 2500                                             # not a multiple of the kernel's 1024-env chunk
    g = torch.Generator(device="cuda").manual_seed(4)
    rew0 = torch.zeros(n, device="cuda")
    trk = _EpisodeTracker.create(None, rew0, rew0.clone(), torch.zeros(n, dtype=torch.int64, device="cuda"))
    assert trk is not None and trk.summary() == {}
    rewbuffer, armbuffer, lenbuffer, donebuffer = (deque(maxlen=100) for _ in range(4))
    cur = torch.zeros(3, n, dtype=torch.float64)
    for step in range(130):
        rew = torch.randn(n, device="cuda", generator=g)
        arm = torch.randn(n, device="cuda", generator=g)
        p = 0.0 if step < 3 else (0.001 if step < 20 else (0.2 if step % 7 == 0 else 0.01))
        dones = (torch.rand(n, device="cuda", generator=g) < p).long()
        trk.step(rew, arm, dones)
        cur[0] += rew.cpu().float().double(); cur[1] += arm.cpu().float().double(); cur[2] += 1
        ids = dones.cpu().nonzero()[:, 0]
        rewbuffer.extend(cur[0][ids].tolist()); armbuffer.extend(cur[1][ids].tolist()); lenbuffer.extend(cur[2][ids].tolist())
        donebuffer.append(len(ids) / n)
        cur[:, ids] = 0
        if step in (2, 10, 19, 21, 129):
            torch.cuda.synchronize()
            got = trk.summary()
            if len(rewbuffer) == 0:
                assert got == {}
                continue
            assert got["mean_reward"] == pytest.approx(statistics.mean(rewbuffer), abs=2e-4)
            assert got["mean_arm_reward"] == pytest.approx(statistics.mean(armbuffer), abs=2e-4)
            assert got["mean_episode_length"] == pytest.approx(statistics.mean(lenbuffer), abs=1e-6)
            assert got["dones"] == pytest.approx(statistics.mean(donebuffer), abs=1e-7)


def test_logged_runner_reports_episode_statistics_from_the_device_rings(tmp_path, capsys):
    cfg = _cfg(n=256)
    cfg.env.episode_length_s = 0.2                       # 10-step episodes: plenty of finished episodes in 2 x 8 steps
    env = WidowGo1(cfg, sim_device="cuda:0", seed=3)
    train = class_to_dict(WidowGo1RoughCfgPPO())
    train["runner"]["num_steps_per_env"] = 8
    runner = OnPolicyRunner(env, train, log_dir=str(tmp_path), device="cuda:0")
    runner.learn(2, init_at_random_ep_len=True)
    rec = runner.history[-1]
    assert 0 < rec["mean_episode_length"] <= 10 and 0 < rec["dones"] <= 1 and np.isfinite(rec["mean_reward"])
    assert "Mean episode length:" in capsys.readouterr().out
    assert env.async_episode_stats is False


def test_resume_with_terrain_levels_keeps_every_robot_on_its_platform(tmp_path):
    """OnPolicyRunner.load on the sub-terrain grid with the terrain curriculum (no sim arena in the checkpoint): the restored
    levels come with their origins, all robots are re-placed on them, and the next steps teleport nobody."""
    from wbc_amd.config import use_grid_terrain

    def make(seed):
        cfg = _cfg(n=200)
        use_grid_terrain(cfg)
        env = WidowGo1(cfg, sim_device="cuda:0", seed=seed)
        train = class_to_dict(WidowGo1RoughCfgPPO())
        train["runner"]["num_steps_per_env"] = 4
        return env, OnPolicyRunner(env, train, log_dir=None, device="cuda:0")
    env, runner = make(3)
    env.terrain_levels.copy_(torch.randint(0, env.max_terrain_level, (200,), device="cuda"))       # a curriculum that has moved
    path = os.path.join(str(tmp_path), "ck.pt")
    runner.save(path)
    env2, runner2 = make(3)
    assert not torch.equal(env2.terrain_levels, env.terrain_levels)
    runner2.load(path)
    torch.cuda.synchronize()
    assert torch.equal(env2.terrain_levels, env.terrain_levels)
    want = env2.terrain_origins[env2.terrain_levels, env2.terrain_types]
    assert torch.equal(env2.env_origins, want) and torch.equal(env2._sim_env_origins, want)
    d0 = (env2.root_states[:, :2] - want[:, :2]).norm(dim=1)
    assert (d0 < 1.5).all()                               # re-placed around the restored platform centres (WG:759-767 offsets)
    for _ in range(3):
        env2.step(torch.zeros(200, 18, device="cuda"))
    torch.cuda.synchronize()
    assert ((env2.root_states[:, :2] - env2.env_origins[:, :2]).norm(dim=1) < 2.0).all()
    assert (env2.root_states[:, 2] - env2.env_origins[:, 2] > 0.1).all()                        # nobody inside the terrain


def test_stale_time_outs_option_goes_through_the_learner():
    """cfg.env.reference_stale_time_outs (quirk Q9): the published mask only changes on steps with a reset, the in-step reward
    store is bypassed (the bootstrap must use the published mask) and the rollout's reward slots carry exactly that bootstrap."""
    cfg = _cfg(n=32)
    cfg.env.reference_stale_time_outs = True
    cfg.env.episode_length_s = 0.1                       # 5-step episodes, all envs in phase: resets only every 5th step
    env = WidowGo1(cfg, sim_device="cuda:0", seed=9)
    train = class_to_dict(WidowGo1RoughCfgPPO())
    train["runner"]["num_steps_per_env"] = 12
    runner = OnPolicyRunner(env, train, log_dir=None, device="cuda:0")
    masks, resets, rews = [], [], []
    raw = env.step

    def spy(a):
        out = raw(0.0 * a)                               # zero actions: the robots stand, episodes end by time-out only
        masks.append(out[5]["time_outs"].clone()); resets.append(out[4].clone()); rews.append(out[2].clone())
        assert out[5]["rollout_stored"] is None
        return out
    env.step = spy
    st = runner.alg.storage
    kept = {}

    def keep_rollout():
        kept["rewards"], kept["values"] = st.rewards.clone(), st.values.clone()
        st.clear()
    runner.alg.update = lambda: (keep_rollout(), (0.,) * 7)[1]
    runner.alg.update_dagger = lambda: (keep_rollout(), 0.)[1]
    runner.learn(1)
    torch.cuda.synchronize()
    stale = [t for t in range(12) if not resets[t].any() and masks[t].any()]
    assert len(stale) >= 4                                # steps without a reset that still publish the last reset step's time-outs
    for t in range(1, 12):
        if not resets[t].any():
            assert torch.equal(masks[t], masks[t - 1])
    gamma = runner.alg.gamma
    for t in range(12):                                   # PPO:133-134 with the PUBLISHED mask, stale or not
        want = rews[t] + gamma * kept["values"][t, :, 0] * masks[t].float()
        np.testing.assert_allclose(kept["rewards"][t, :, 0].cpu().numpy(), want.cpu().numpy(), rtol=0, atol=1e-6)


def test_episode_statistics_carried_by_the_policy_launch_equal_the_stand_alone_launch():
    """WidowGo1.defer_episode_stats + ActorCritic.fused_act(side_job=...): the env step's extras['episode'] reduction (and the
    runner's episode deques) executed by extra workgroups of the policy inference that follows, against the stand-alone launch
    of an identically seeded env; the inference outputs do not depend on carrying a job; a job nobody takes is run by the next step."""
    from wbc_amd.rsl_rl.modules import ActorCritic
    from wbc_amd.rsl_rl.runners.on_policy_runner import _EpisodeTracker
    cfg = _cfg(n=700)                                     # not a multiple of the 16-row tiles, nor of 256
    cfg.env.episode_length_s = 0.3                        # 15-step episodes: resets (hence statistics) in most steps
    train = class_to_dict(WidowGo1RoughCfgPPO())
    torch.manual_seed(2)
    ac = ActorCritic(76, 76, 18, **train["policy"], num_priv=24, num_hist=10, num_prop=76).cuda()
    envs = [WidowGo1(cfg, sim_device="cuda:0", seed=6) for _ in range(2)]
    for e in envs:
        e.reset()
        e.episode_length_buf = torch.randint(0, 15, (700,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
    zeros = torch.zeros(700, dtype=torch.int64, device="cuda")
    trackers = [_EpisodeTracker.create(e, e.rew_buf, e.arm_rew_buf, zeros) for e in envs]     # both envs publish statistics: attached
    assert all(t._hooked_env is not None for t in trackers)
    envs[1].defer_episode_stats = True
    g = torch.Generator(device="cuda").manual_seed(1)
    obs = [e.get_observations() for e in envs]
    with torch.inference_mode():
        for step in range(24):
            eps = torch.randn(700, 18, device="cuda", generator=g)
            job = envs[1].take_stats_job()
            assert (job is not None) == (step > 0)
            if step == 7:                                  # nobody carries this one: the next env.step() must run it
                envs[1]._stats_job, job = job, None
            a0 = ac.fused_act(obs[0], eps)
            a1 = ac.fused_act(obs[1], eps, side_job=job)
            for x, y in zip(a0, a1):
                assert torch.equal(x, y)
            out0 = envs[0].step(a0[0])
            out1 = envs[1].step(a1[0])
            obs = [out0[0], out1[0]]
            assert torch.equal(out0[0], out1[0]) and torch.equal(out0[4], out1[4])
            if step > 0:
                np.testing.assert_allclose(prev1.vector.cpu().numpy(), prev0.vector.cpu().numpy(), rtol=2e-6, atol=1e-7)
            prev0, prev1 = out0[5]["episode"], out1[5]["episode"]
    envs[1].flush_stats_job()
    torch.cuda.synchronize()
    np.testing.assert_allclose(prev1.vector.cpu().numpy(), prev0.vector.cpu().numpy(), rtol=2e-6, atol=1e-7)
    s0, s1 = trackers[0].summary(), trackers[1].summary()
    assert s0["mean_episode_length"] > 0 and s0.keys() == s1.keys()
    for k in s0:
        assert s0[k] == pytest.approx(s1[k], rel=1e-6), k
    for t in trackers:
        t.close()


def test_create_rejects_what_the_row_addressing_cannot_hold():
    """wbc_sim_create bounds num_envs (include/wbc_sim.h): the kernels address a tensor's rows with 32-bit element offsets. The
    check comes before any allocation; the message names the bound."""
    import ctypes as C
    from wbc_amd import abi
    from wbc_amd.native import lib
    L = lib()
    m = abi.load_default_model()
    cfg = _cfg()
    model, tcfg = abi.fill_model(m), abi.fill_task_cfg(cfg, m)
    h = C.c_void_p()
    for n in (0, (1 << 22) + 1):
        rc = L.wbc_sim_create(C.byref(model), C.byref(tcfg), n, 0, 1, None, 0, C.byref(h))
        assert rc == -1 and not h.value
    L.wbc_last_error.restype = C.c_char_p
    assert b"2^22" in L.wbc_last_error()
