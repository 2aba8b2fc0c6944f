"""How far ahead of the GPU is the host in the fused update? Wall time of PPO._update_fused's Python side WITHOUT a final
synchronize (= time to enqueue its ~160 launches) next to the synchronised time."""
import sys, os, time
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
alg = runner.alg
raw = alg._update_fused
rec = []
def timed():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = raw()                       # returns after .tolist() of the loss sums: a sync at the very end
    rec.append(time.perf_counter() - t0)
    return out
alg._update_fused = timed
import wbc_amd.native as nat
L = nat.lib()
g = L.wbc_ppo_minibatch_grad
enq = []
def wrapped(*a):
    t0 = time.perf_counter(); rc = g(*a); enq.append(time.perf_counter() - t0); return rc
L.wbc_ppo_minibatch_grad = wrapped
runner.learn(8, init_at_random_ep_len=True)
print("update wall (synchronised at its end) ms:", [round(x * 1e3, 2) for x in rec[-5:]])
print("host time inside wbc_ppo_minibatch_grad (6 launches) us: mean", round(sum(enq[-100:]) / 100 * 1e6, 1), "max", round(max(enq[-100:]) * 1e6, 1))
