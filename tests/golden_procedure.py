"""The seeded procedure shared by tools/make_golden_ppo.py (which runs it through the reference's
rsl_rl) and tests/test_ppo_parity.py (which runs it through wbc_amd.rsl_rl): BASELINE.json
configs[0], i.e. a synthetic 64-env x 24-step rollout with the widowGo1 PPO hyper-parameters
(widowGo1_config.py:321-366, non-RESUME schedules), then compute_returns, two update() calls and one
update_dagger() call."""
import numpy as np
import torch

POLICY_KW = dict(init_std=[[0.8, 1.0, 1.0] * 4 + [1.0] * 6], actor_hidden_dims=[128], critic_hidden_dims=[128], activation="elu",
                 leg_control_head_hidden_dims=[128, 128], arm_control_head_hidden_dims=[128, 128], priv_encoder_dims=[64, 20],
                 num_leg_actions=12, num_arm_actions=6, adaptive_arm_gains=False, adaptive_arm_gains_scale=10.0,
                 num_priv=24, num_hist=10, num_prop=76)
ALG_KW = dict(num_learning_epochs=5, num_mini_batches=4, clip_param=0.2, gamma=0.99, lam=0.95, value_loss_coef=1.0,
              entropy_coef=0.0, learning_rate=2e-4, max_grad_norm=1.0, use_clipped_value_loss=True, schedule="fixed",
              desired_kl=None, mixing_schedule=[1.0, 0, 3000], torque_supervision=False,
              torque_supervision_schedule=[0.0, 1000, 1000], adaptive_arm_gains=False,
              min_policy_std=[[0.15, 0.25, 0.25] * 4 + [0.2] * 3 + [0.05] * 3], dagger_update_freq=20,
              priv_reg_coef_schedual=[0, 0.1, 3000, 7000])
N, T = 64, 24


def synthetic_rollout(seed):
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(T + 1, N, 860, generator=g)
    rew = 0.01 * torch.randn(T, N, generator=g)
    arm = 0.01 * torch.randn(T, N, generator=g)
    dones = (torch.rand(T, N, generator=g) < 0.05).long()
    touts = torch.rand(T, N, generator=g) < 0.01
    return obs, rew, arm, dones, touts


def param_digest(module):
    sd = module.state_dict()
    return np.array([[v.double().sum().item(), v.double().abs().sum().item(), v.flatten()[0].item(), v.flatten()[-1].item()]
                     for v in sd.values()])


def run_procedure(ActorCritic, PPO, device="cpu", counter0=3500, **extra_alg_kw):
    """counter0 = 3500 puts the schedules mid-ramp: mixing beta = 1, ROA coefficient = 0.1*500/7000."""
    out = {}
    torch.manual_seed(1)
    ac = ActorCritic(76, 76, 18, **POLICY_KW)
    alg = PPO(ac, device=device, **ALG_KW, **extra_alg_kw)
    alg.counter = counter0
    out["init_digest"] = param_digest(ac)
    alg.init_storage(N, T, [860], [None], [18])
    for it, hist in enumerate([False, False, True]):
        obs, rew, arm, dones, touts = synthetic_rollout(100 + it)
        obs = obs.to(device)
        torch.manual_seed(1000 + it)
        with torch.inference_mode():
            for t in range(T):
                a = alg.act(obs[t], obs[t], hist)
                if t == 0:
                    out[f"it{it}_actions0"] = a.cpu().numpy().copy()
                alg.process_env_step(rew[t].to(device), arm[t].to(device), dones[t].to(device), {"time_outs": touts[t].to(device)})
            alg.compute_returns(obs[T])
        out[f"it{it}_returns"] = alg.storage.returns.cpu().numpy().copy()
        out[f"it{it}_advantages"] = alg.storage.advantages.cpu().numpy().copy()
        out[f"it{it}_rewards"] = alg.storage.rewards.cpu().numpy().copy()
        if hist:
            out[f"it{it}_stats"] = np.array([alg.update_dagger()])
        else:
            out[f"it{it}_stats"] = np.array([float(x) for x in alg.update()])
        out[f"it{it}_digest"] = param_digest(ac)
        out[f"it{it}_std"] = ac.std.detach().cpu().numpy().copy()
    return out


STORAGE_FIELDS = ("actions", "rewards", "dones", "values", "actions_log_prob", "mu", "sigma", "returns", "advantages")


def flat_params(module):
    return np.concatenate([v.detach().cpu().double().numpy().ravel() for v in module.state_dict().values()]).astype(np.float32)


def synthetic_storage(storage, seed):
    """Fill a RolloutStorage with a synthetic but self-consistent rollout WITHOUT any network: N(0,1) observations, a stored policy
    mean 0.05 N(0,1) (a randomly initialised actor's tanh outputs are that small), sigma = the config's init std, actions = mu +
    sigma eps, the old log-probabilities in closed form (two channels: 12 leg / 6 arm dimensions, actor_critic.py:341-345), values
    0.1 N(0,1), rewards 0.01 N(0,1), 3 % dones; returns / advantages by the storage's own compute_returns (RS:136-150). Only the CPU
    generator and elementwise torch math: the build container (through the reference's RolloutStorage) and the GPU box (through
    this package's) produce the same inputs; what the reference COMPUTES from them is what the fixture records."""
    g = torch.Generator().manual_seed(seed)
    T, N = storage.num_transitions_per_env, storage.num_envs
    dev = storage.observations.device
    obs = torch.randn(T, N, 860, generator=g)
    mu = 0.05 * torch.randn(T, N, 18, generator=g)
    sigma = torch.tensor(POLICY_KW["init_std"][0]).expand(T, N, 18).contiguous()
    act = mu + sigma * torch.randn(T, N, 18, generator=g)
    lp = -0.5 * ((act - mu) / sigma) ** 2 - torch.log(sigma) - 0.5 * np.log(2 * np.pi)
    logp = torch.stack([lp[..., :12].sum(-1), lp[..., 12:].sum(-1)], dim=-1)
    val = 0.1 * torch.randn(T, N, 2, generator=g)
    rew = 0.01 * torch.randn(T, N, 2, generator=g)
    dones = (torch.rand(T, N, 1, generator=g) < 0.03).to(storage.dones.dtype)
    last = 0.1 * torch.randn(N, 2, generator=g)
    for name, x in (("observations", obs), ("mu", mu), ("sigma", sigma), ("actions", act), ("actions_log_prob", logp), ("values", val),
                    ("rewards", rew), ("dones", dones)):
        getattr(storage, name).copy_(x.to(dev))
    storage.step = T
    return last


def gae_known_answer_inputs():
    rew = torch.tensor([[[1, .5], [0, -1]], [[.5, .25], [1, 0]], [[-1, 2], [.5, .5]], [[.25, 0], [2, 1]]], dtype=torch.float32)
    val = torch.tensor([[[.1, .2], [.3, .4]], [[.5, .6], [.7, .8]], [[.9, 1.0], [1.1, 1.2]], [[1.3, 1.4], [1.5, 1.6]]], dtype=torch.float32)
    dones = torch.tensor([[0, 0], [1, 0], [0, 0], [0, 1]], dtype=torch.uint8).view(4, 2, 1)
    last = torch.tensor([[2., 3.], [4., 5.]])
    return rew, val, dones, last
