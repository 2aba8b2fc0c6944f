"""Quick timing of the fused step kernel alone (no policy, no learner)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers
from wbc_amd import abi
from wbc_amd.config import WidowGo1RoughCfg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
m = abi.load_default_model(); cfg = WidowGo1RoughCfg()
robot = dict(model=m, wmodel=abi.fill_model(m), cfg=cfg, tcfg=abi.fill_task_cfg(cfg, m))
g = helpers.make_gpu(robot, n, helpers.random_env_params(n, 0))
g.reset_all()
acts = [torch.randn(n, 18, device="cuda") * 0.5 for _ in range(8)]
for i in range(20): g.step(acts[i % 8])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(steps): g.step(acts[i % 8])
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
print(f"N={n} step kernel {ms*1000:.1f} us/step -> {n/ms*1000:.3e} env-steps/s (sim only); resets/step {g.tensor('RESET_BUF').float().mean().item():.3f}")
