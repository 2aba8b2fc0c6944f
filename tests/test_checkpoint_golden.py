"""A checkpoint written by the REFERENCE's own rsl_rl classes (tests/golden/reference_checkpoint.pt, made by
tools/make_golden_checkpoint.py exactly as on_policy_runner.py:276-282 saves) loaded through wbc_amd's OnPolicyRunner.load:
the policy must reproduce the reference's act_inference / evaluate outputs on fixed observations, and the optimiser state
must arrive intact; a save -> load round trip restores the counters and generator states the reference forgets."""
import os
import types

import numpy as np
import pytest
import torch

from wbc_amd.config import WidowGo1RoughCfg, WidowGo1RoughCfgPPO, class_to_dict
from wbc_amd.rsl_rl.env import VecEnv
from wbc_amd.rsl_rl.runners import OnPolicyRunner

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class _StubEnv(VecEnv):
    """The attribute surface OnPolicyRunner reads (SURVEY.md section 8b seam 1); no simulation behind it."""

    def __init__(self, n=8, device="cpu"):
        self.cfg = WidowGo1RoughCfg()
        self.num_envs, self.num_obs, self.num_privileged_obs, self.num_actions = n, 860, None, 18
        self.device = torch.device(device)
        self.max_episode_length = 500
        self.episode_length_buf = torch.zeros(n, dtype=torch.long, device=device)
        self.obs_buf = torch.zeros(n, 860, device=device)
        self.p_gains = torch.tensor([50.0] * 12 + [5.0] * 6, device=device)
        self.d_gains = torch.tensor([1.0] * 12 + [0.5] * 6, device=device)
        self.default_dof_pos = torch.zeros(20, device=device)
        self.update_counter = 0
        self.common_step_counter = 0

    def reset(self):
        return self.obs_buf, None

    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return None

    def update_command_curriculum(self):
        self.update_counter += 1

    def step(self, actions):
        self.common_step_counter += 1
        n = self.num_envs
        z = torch.zeros(n, device=self.device)
        return self.obs_buf, None, z, z, torch.zeros(n, dtype=torch.long, device=self.device), {"time_outs": torch.zeros(n, dtype=torch.bool, device=self.device)}


def _runner(device="cpu"):
    train = class_to_dict(WidowGo1RoughCfgPPO())
    train["runner"]["num_steps_per_env"] = 4
    return OnPolicyRunner(_StubEnv(device=device), train, log_dir=None, device=device)


def _check_outputs(runner, device):
    io = np.load(os.path.join(GOLD, "reference_checkpoint_io.npz"))
    x = torch.from_numpy(io["obs"]).to(device)
    ac = runner.alg.actor_critic
    policy = runner.get_inference_policy(device=device)
    tol = dict(atol=1e-6, rtol=1e-5) if device == "cpu" else dict(atol=2e-5, rtol=1e-4)
    with torch.inference_mode():
        np.testing.assert_allclose(policy(x).cpu().numpy(), io["act_teacher"], **tol)
        np.testing.assert_allclose(ac.act_inference(x, hist_encoding=True).cpu().numpy(), io["act_student"], **tol)
        np.testing.assert_allclose(ac.evaluate(x).cpu().numpy(), io["value"], **tol)
    np.testing.assert_array_equal(ac.std.detach().cpu().numpy(), io["std"])
    return io


def test_reference_checkpoint_loads_and_reproduces_reference_outputs():
    runner = _runner("cpu")
    infos = runner.load(os.path.join(GOLD, "reference_checkpoint.pt"))
    assert infos is None and runner.current_learning_iteration == 1234
    io = _check_outputs(runner, "cpu")
    st = runner.alg.optimizer.state_dict()["state"]
    assert len(st) == 33                                                   # quirk L6: the history encoder has no state in this optimiser
    np.testing.assert_array_equal([float(v["step"]) for v in st.values()], io["adam_steps"])
    np.testing.assert_allclose([float(v["exp_avg"].abs().sum()) for v in st.values()], io["adam_exp_avg_abs_sum"], rtol=1e-6)
    runner.alg.storage.observations.normal_()                              # and training continues from it
    runner.learn(2)
    assert np.isfinite(runner.history[-1]["mean_value_loss"])


def test_save_load_roundtrip_restores_what_the_reference_forgets(tmp_path):
    r1 = _runner("cpu")
    r1.load(os.path.join(GOLD, "reference_checkpoint.pt"))
    r1.current_learning_iteration = 1239                                   # iteration 1240 is a DAgger one (every 20th)
    r1.learn(3)
    path = os.path.join(str(tmp_path), "m.pt")
    r1.save(path)
    expect = torch.rand(5)                                                 # the next CPU draws after the save
    d = torch.load(path, map_location="cpu")
    assert set(d) >= {"model_state_dict", "optimizer_state_dict", "iter", "infos"}
    assert set(d["wbc_extra"]) >= {"hist_encoder_optimizer_state_dict", "ppo_counter", "env_update_counter", "env_common_step_counter",
                                   "torch_rng_state"}
    r2 = _runner("cpu")
    r2.load(path)
    assert r2.current_learning_iteration == r1.current_learning_iteration == 1242
    assert r2.alg.counter == r1.alg.counter == 3 and r2.env.update_counter == 3 and r2.env.common_step_counter == 12
    assert torch.equal(torch.rand(5), expect)
    for (k, a), (_, b) in zip(r1.alg.actor_critic.state_dict().items(), r2.alg.actor_critic.state_dict().items()):
        assert torch.equal(a, b), k
    h1 = r1.alg.hist_encoder_optimizer.state_dict()["state"]
    h2 = r2.alg.hist_encoder_optimizer.state_dict()["state"]
    assert len(h1) == len(h2) == 8
    for k in h1:
        assert torch.equal(h1[k]["exp_avg"], h2[k]["exp_avg"])
    # a non-None infos round-trips through the weights_only load: numpy scalars / arrays are converted on save, anything that
    # could not be read back is refused at save time, not discovered at resume time
    r1.env.update_counter = np.int64(3)
    r1.save(path, infos={"note": "x", "best": np.float32(1.5), "curve": np.arange(3.0), "nested": {"k": [1, np.int64(2)]}})
    r3 = _runner("cpu")
    r3.load(path)
    inf = torch.load(path, map_location="cpu", weights_only=True)["infos"]
    assert inf["note"] == "x" and inf["best"] == 1.5 and inf["nested"]["k"] == [1, 2] and torch.equal(inf["curve"], torch.arange(3.0, dtype=torch.float64))
    with pytest.raises(TypeError, match="infos"):
        r1.save(path, infos={"env": r1.env})


@pytest.mark.gpu
def test_reference_checkpoint_on_the_gpu_and_fused_path_after_resume(tmp_path):
    """On the MI355X: the reference checkpoint through the fused inference kernels, and after load() the update still takes
    the fused clip + Adam path (the optimiser's step counters arrive on the device and are moved back to the host)."""
    from wbc_amd.envs import WidowGo1
    runner = _runner("cuda:0")
    runner.load(os.path.join(GOLD, "reference_checkpoint.pt"))
    _check_outputs(runner, "cuda:0")
    cfg = WidowGo1RoughCfg()
    cfg.env.num_envs = 256
    cfg.terrain.mesh_type = "plane"
    env = WidowGo1(cfg, sim_device="cuda:0", seed=3)
    train = class_to_dict(WidowGo1RoughCfgPPO())
    train["runner"]["num_steps_per_env"] = 8
    r = OnPolicyRunner(env, train, log_dir=None, device="cuda:0")
    r.load(os.path.join(GOLD, "reference_checkpoint.pt"))
    r.current_learning_iteration = 1                                       # a PPO iteration next (0 mod 20 would be DAgger)
    calls = {"n": 0}
    from wbc_amd.native import lib
    L = lib()
    raw = L.wbc_ppo_clip_adam_packed

    def counted(*a):
        calls["n"] += 1
        return raw(*a)
    L.wbc_ppo_clip_adam_packed = counted
    try:
        r.learn(1)
    finally:
        L.wbc_ppo_clip_adam_packed = raw
    assert calls["n"] == 20                                                # 5 epochs x 4 minibatches on the fused path
    steps = [float(v["step"]) for v in r.alg.optimizer.state_dict()["state"].values()]
    assert steps and all(s == 40.0 for s in steps)                         # 20 (checkpoint) + 20
    path = os.path.join(str(tmp_path), "g.pt")
    r.save(path, save_env_state=True)
    d = torch.load(path, map_location="cpu")
    assert d["wbc_extra"]["sim_step_counter"] == env.sim.step_counter and "cuda_rng_state" in d["wbc_extra"] and "sim_arena" in d["wbc_extra"]
    obs_before = env.obs_buf.clone()
    a = torch.randn(256, 18, device="cuda:0")
    env.step(a)
    after1 = env.obs_buf.clone()
    r.load(path)                                                            # rewinds the sim state and its draw counter
    assert torch.equal(env.obs_buf, obs_before)
    env.step(a)
    assert torch.equal(env.obs_buf, after1)
