"""Substep phase stamps (block 0, timing build) and the step-kernel time for two populations: robots standing after a reset, and
robots that have fallen and stay down (terminations off, random actions): where do the heavy-contact states spend their time?"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["WBC_AMD_LIB"] = os.path.join(ROOT, "deep-whole-body-control_amd", "wbc_amd", "libwbc_amd_timing.so")
import numpy as np, torch
import helpers
from wbc_amd import abi
from wbc_amd.config import WidowGo1RoughCfg
from wbc_amd.native import lib
n = 4096
m = abi.load_default_model(); cfg = WidowGo1RoughCfg()
L = lib(); L.wbc_debug_set_step_timing.argtypes = [C.c_void_p]
names = ["fk", "S,v,c", "inertia", "pass2", "root inv", "pass3+K", "contact detect", "contact iters", "outputs", "integrate"]
for scen in ("standing", "fallen"):
    tc = abi.fill_task_cfg(cfg, m)
    tc.term_z_threshold = -10.0; tc.term_rp_threshold = 100.0
    g = helpers.make_gpu(dict(model=m, wmodel=abi.fill_model(m), cfg=cfg, tcfg=tc), n, helpers.random_env_params(n, 0))
    g.reset_all()
    scale = 0.0 if scen == "standing" else 1.0
    acts = [torch.randn(n, 18, device="cuda") * scale for _ in range(8)]
    for i in range(40 if scen == "standing" else 200): g.step(acts[i % 8])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(50): g.step(acts[i % 8])
    e1.record(); torch.cuda.synchronize()
    z = g.tensor("ROOT_STATES")[:, 0, 2].float().mean().item()
    ncon = (g.tensor("NET_CONTACT_FORCE").norm(dim=-1) > 0.1).float().sum(1).mean().item()
    buf = torch.zeros(32, dtype=torch.int64, device="cuda")
    L.wbc_debug_set_step_timing(buf.data_ptr())
    g.simulate(); torch.cuda.synchronize()
    t = buf.cpu().numpy()
    L.wbc_debug_set_step_timing(None)
    print(f"{scen}: step kernel {e0.elapsed_time(e1) / 50 * 1000:.1f} us; mean base height {z:.3f}, bodies in contact {ncon:.1f}; env 0 z {g.tensor('ROOT_STATES')[0, 0, 2].item():.3f}")
    print("   substep phase cycles (block 0):", {names[i]: int(t[i + 1] - t[i]) for i in range(10)}, "total", int(t[10] - t[0]))
    print("   contact detail: narrow phase", int(t[18] - t[6]), "set-up", int(t[7] - t[18]), "| first iteration: solve", int(t[20] - t[7]), "gather", int(t[21] - t[20]),
          "inward", int(t[22] - t[21]), "root", int(t[23] - t[22]), "outward", int(t[24] - t[23]))
    g.close()
