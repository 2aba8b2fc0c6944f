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
