"""wbc_policy_act16_kernel alone: rows sweep, with and without a carried side job (the launch's own time: HIP events over 200 launches)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd"))
import torch
from wbc_amd.config import WidowGo1RoughCfg, WidowGo1RoughCfgPPO, class_to_dict
from wbc_amd.envs import WidowGo1
from wbc_amd.rsl_rl.runners import OnPolicyRunner
cfg = WidowGo1RoughCfg(); cfg.env.num_envs = 4096; cfg.terrain.mesh_type = "plane"
tc = WidowGo1RoughCfgPPO(); torch.manual_seed(tc.seed)
env = WidowGo1(cfg, sim_device="cuda:0", seed=tc.seed)
runner = OnPolicyRunner(env, class_to_dict(tc), log_dir=None, device="cuda:0")
ac = runner.alg.actor_critic
for n in (256, 1024, 2048, 4096, 8192, 16384):
    obs = torch.randn(n, 860, device="cuda"); eps = torch.randn(n, 18, device="cuda")
    out = tuple(torch.empty(n, w, device="cuda") for w in (18, 18, 2, 2))
    for _ in range(20): ac.fused_act(obs, eps, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): ac.fused_act(obs, eps, out)
    e1.record(); torch.cuda.synchronize()
    print(f"rows {n:6d}: {e0.elapsed_time(e1) / 200 * 1000:.1f} us per launch (no side job)")

# the same launch behind 64 MB of unrelated traffic (as behind the step kernel in a rollout): cold obs + cold weights; then with the weight
# pack touched again after the traffic (what an L2 prefetch at the end of the step kernel could give)
n = 4096
obs = torch.randn(n, 860, device="cuda"); eps = torch.randn(n, 18, device="cuda")
out = tuple(torch.empty(n, w, device="cuda") for w in (18, 18, 2, 2))
junk = torch.empty(16 * 1024 * 1024, device="cuda")
for mode in ("warm", "after 64 MB of traffic", "after the traffic + weight pack re-read"):
    ts = []
    for _ in range(30):
        if mode != "warm": junk.add_(1.0)
        if mode.endswith("re-read"): ac._wpack.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ac.fused_act(obs, eps, out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000)
    ts.sort()
    print(f"4096 rows, {mode}: median {ts[len(ts) // 2]:.1f} us (single launches between events)")
