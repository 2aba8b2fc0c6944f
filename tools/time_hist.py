"""wbc_hist_latent at the benchmark's size (the ROA target of every stored row: 163840 rows) and at one env step's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_procedure as gp
from wbc_amd.rsl_rl.modules import ActorCritic
torch.manual_seed(0)
ac = ActorCritic(76, 76, 18, **gp.POLICY_KW).cuda()
for rows in (4096, 163840):
    obs = torch.randn(rows, 860, device="cuda")
    with torch.inference_mode():
        for _ in range(3): ac.actor.infer_hist_latent(obs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ac.actor.infer_hist_latent(obs)
        e1.record(); torch.cuda.synchronize()
    print(f"hist latent rows={rows}: {e0.elapsed_time(e1) / 20 * 1000:.1f} us")
