"""Per-layer clock stamps of wbc_policy_act16_kernel's workgroup (0, 0) (build: tools/build_variant.py ppotiming -DWBC_PPO_TIMING):
where the 2.3 us per layer of a small-batch launch go.  usage: python tools/time_act_layers.py [rows]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd"))
os.environ["WBC_AMD_LIB"] = os.path.join(ROOT, "deep-whole-body-control_amd", "wbc_amd", "libwbc_amd_ppotiming.so")
import numpy as np, torch
from wbc_amd.config import WidowGo1RoughCfg, WidowGo1RoughCfgPPO, class_to_dict
from wbc_amd.envs import WidowGo1
from wbc_amd.rsl_rl.runners import OnPolicyRunner
from wbc_amd.native import lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = WidowGo1RoughCfg(); cfg.env.num_envs = 512; cfg.terrain.mesh_type = "plane"
tc = WidowGo1RoughCfgPPO(); torch.manual_seed(tc.seed)
env = WidowGo1(cfg, sim_device="cuda:0", seed=tc.seed)
ac = OnPolicyRunner(env, class_to_dict(tc), log_dir=None, device="cuda:0").alg.actor_critic
obs = torch.randn(n, 860, device="cuda"); eps = torch.randn(n, 18, device="cuda")
out = tuple(torch.empty(n, w, device="cuda") for w in (18, 18, 2, 2))
for _ in range(10): ac.fused_act(obs, eps, out)
L = lib(); L.wbc_debug_set_policy_timing.argtypes = [C.c_void_p]
buf = torch.zeros(160, dtype=torch.int64, device="cuda")
L.wbc_debug_set_policy_timing(buf.data_ptr())
ac.fused_act(obs, eps, out); torch.cuda.synchronize()
t = buf.cpu().numpy()
L.wbc_debug_set_policy_timing(None)
print(f"rows {n}: entry -> inputs in LDS {t[1] - t[0]}, layers {t[2] - t[1]} cycles")
for l in range(16):
    s = t[32 + 4 * l: 36 + 4 * l]
    if s[0]: print(f"   layer {l:2d}: MFMA chain {s[1] - s[0]:5d}, epilogue {s[2] - s[1]:5d}, barrier {s[3] - s[2]:5d}; to the next layer's start {(t[32 + 4 * (l + 1)] - s[3]) if l < 15 and t[32 + 4 * (l + 1)] else 0}")
