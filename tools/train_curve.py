"""Does the whole loop learn? Train the shipped widowGo1 config (flat terrain) from scratch for a few hundred iterations and print
the per-iteration means of both reward channels, the fraction of envs resetting per step and the mean episode length at reset.
usage: python tools/train_curve.py [iterations] [envs] [z_threshold] [survive_scale] > curve.json"""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd"))
import torch
from wbc_amd.config import WidowGo1RoughCfg, WidowGo1RoughCfgPPO, class_to_dict
from wbc_amd.envs import WidowGo1
from wbc_amd.rsl_rl.runners import OnPolicyRunner

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cfg = WidowGo1RoughCfg()
cfg.env.num_envs = n
cfg.terrain.mesh_type = "plane"
if len(sys.argv) > 3:                      # optional: termination height (shipped: 0.325, 6 mm under the default stance's 0.331)
    cfg.termination.z_threshold = float(sys.argv[3])
if len(sys.argv) > 4:                      # optional: survive reward scale (shipped: 0.2, below the foot-contact penalty of a standing robot)
    cfg.rewards.scales.survive = float(sys.argv[4])
train = class_to_dict(WidowGo1RoughCfgPPO())
torch.manual_seed(train["seed"])
env = WidowGo1(cfg, sim_device="cuda:0", seed=train["seed"])
runner = OnPolicyRunner(env, train, log_dir=None, device="cuda:0")
T = runner.num_steps_per_env
rows = []
raw_step = env.step
acc = {"rew": torch.zeros((), device="cuda:0"), "arm": torch.zeros((), device="cuda:0"), "resets": torch.zeros((), device="cuda:0"),
       "len": torch.zeros((), device="cuda:0"), "steps": 0}


def step(a):
    ep_before = env.episode_length_buf.clone()
    out = raw_step(a)
    m = env.reset_buf > 0
    acc["rew"] += env.rew_buf.mean(); acc["arm"] += env.arm_rew_buf.mean(); acc["resets"] += m.float().mean()
    acc["len"] += ((ep_before + 1) * m).sum().float() / m.sum().clamp(min=1).float()
    acc["steps"] += 1
    return out


env.step = step
t0 = time.time()
for it in range(iters):
    for k in ("rew", "arm", "resets", "len"):
        acc[k].zero_()
    acc["steps"] = 0
    runner.learn(1, init_at_random_ep_len=(it == 0))
    s = acc["steps"]
    rows.append(dict(it=it, rew=acc["rew"].item() / s, arm_rew=acc["arm"].item() / s, reset_frac=acc["resets"].item() / s,
                     ep_len_at_reset=acc["len"].item() / s, **{k: runner.history[-1][k] for k in ("mean_value_loss", "mean_surrogate_loss", "mean_hist_latent_loss")}))
wall = time.time() - t0
for r in rows[::max(1, iters // 30)]:
    print(json.dumps({k: (round(v, 5) if isinstance(v, float) else v) for k, v in r.items()}))
print(json.dumps({"iterations": iters, "envs": n, "wall_s": round(wall, 2), "env_steps": iters * n * T, "first10_rew": sum(r["rew"] for r in rows[:10]) / 10,
                  "last10_rew": sum(r["rew"] for r in rows[-10:]) / 10, "first10_reset_frac": sum(r["reset_frac"] for r in rows[:10]) / 10,
                  "last10_reset_frac": sum(r["reset_frac"] for r in rows[-10:]) / 10, "first10_arm": sum(r["arm_rew"] for r in rows[:10]) / 10,
                  "last10_arm": sum(r["arm_rew"] for r in rows[-10:]) / 10}))
