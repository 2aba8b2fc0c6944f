"""Per-iteration collection / learn times of the benchmark loop (which iterations are DAgger updates, what they cost)."""
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
group = None
if os.environ.get("WBC_DIST"):          # 1-rank RCCL group: the collectives of the multi-GPU path on one GPU
    import torch.distributed as dist
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29534", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    group = dist.group.WORLD
runner = OnPolicyRunner(env, class_to_dict(tc), log_dir=None, device="cuda:0", dist_group=group)
runner.learn(int(sys.argv[1]) if len(sys.argv) > 1 else 24, init_at_random_ep_len=True)
for h in runner.history:
    print(f"it {h['it']:3d}  collect {h['collection_time']*1e3:7.2f} ms  learn {h['learn_time']*1e3:7.2f} ms  {'DAgger' if h['it'] % 20 == 0 else ''}")
