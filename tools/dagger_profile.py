"""Kernel mix of the DAgger iterations alone (every iteration a DAgger update): python tools/dagger_profile.py [iters]"""
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
train = class_to_dict(tc); train["algorithm"]["dagger_update_freq"] = 1
runner = OnPolicyRunner(env, train, log_dir=None, device="cuda:0")
runner.learn(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
for h in runner.history[1:]:
    print(f"it {h['it']:3d}  collect {h['collection_time']*1e3:7.2f} ms  learn {h['learn_time']*1e3:7.2f} ms")
