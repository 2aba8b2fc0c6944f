"""Timing of the fused step kernel alone (no policy, no learner), with knobs to attribute the time."""
import sys, os, time, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
_timing_lib = os.path.join(ROOT, "deep-whole-body-control_amd", "wbc_amd", "libwbc_amd_timing.so")   # tools/build_variant.py timing -DWBC_STEP_TIMING
if os.path.exists(_timing_lib) and os.environ.get("WBC_STAMPS"):
    os.environ["WBC_AMD_LIB"] = _timing_lib
import numpy as np, torch
import helpers
from wbc_amd import abi
from wbc_amd.config import WidowGo1RoughCfg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
variants = sys.argv[3:] or ["base"]
m = abi.load_default_model(); cfg = WidowGo1RoughCfg()
for var in variants:
    tc = abi.fill_task_cfg(cfg, m)
    if var.startswith("dec"): tc.decimation = int(var[3:])
    if var.startswith("it"): tc.contact_iters = int(var[2:])
    if var == "nocontact": tc.contact_margin = -1e9
    if var == "noreset": tc.term_z_threshold = -10.0; tc.term_rp_threshold = 100.0
    wm = abi.fill_model(m)
    for k in range(abi.NCP):      # table-driven ablations of the self-collision broad phase: no candidates / no box candidates / no limb pairs
        if (var == "nocand" and wm.pr_kind[k] in (abi.PR_LIMBS, abi.PR_SPHERE_BOX)) or (var == "noboxcand" and wm.pr_kind[k] == abi.PR_SPHERE_BOX) or \
           (var == "nolimbcand" and wm.pr_kind[k] == abi.PR_LIMBS):
            wm.pr_kind[k] = abi.PR_NONE
    robot = dict(model=m, wmodel=wm, cfg=cfg, tcfg=tc)
    g = helpers.make_gpu(robot, n, helpers.random_env_params(n, 0))
    g.reset_all()
    acts = [torch.randn(n, 18, device="cuda") * float(os.environ.get("WBC_ACT_SCALE", "0.5")) for _ in range(8)]
    for i in range(int(os.environ.get("WBC_WARM_STEPS", "20"))): g.step(acts[i % 8])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): g.step(acts[i % 8])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    print(f"{var:10s} N={n} step kernel {ms*1000:.1f} us/step -> {n/ms*1000:.3e} env-steps/s (sim only); resets/step {g.tensor('RESET_BUF').float().mean().item():.3f}")
    g.close()

# simulate-only timing (one substep per launch, no PD / post-physics, no scratch in that kernel)
tc = abi.fill_task_cfg(cfg, m)
robot = dict(model=m, wmodel=abi.fill_model(m), cfg=cfg, tcfg=tc)
g = helpers.make_gpu(robot, n, helpers.random_env_params(n, 0))
g.reset_all()
for i in range(30): g.step(acts[i % 8])
for i in range(20): g.simulate()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(100): g.simulate()
e1.record(); torch.cuda.synchronize()
print(f"simulate kernel {e0.elapsed_time(e1)/100*1000:.1f} us/substep")
e0.record()
for i in range(100): g.refresh_rigid_body_state()
e1.record(); torch.cuda.synchronize()
print(f"fk kernel {e0.elapsed_time(e1)/100*1000:.1f} us")

import ctypes as C
from wbc_amd.native import lib
L = lib(); L.wbc_debug_set_step_timing.argtypes = [C.c_void_p]
buf = torch.zeros(32, dtype=torch.int64, device="cuda")
L.wbc_debug_set_step_timing(buf.data_ptr())
g.simulate(); torch.cuda.synchronize()
t = buf.cpu().numpy()
names = ["fk", "S,v,c", "inertia", "pass2", "root inv", "pass3+K", "contact detect", "contact iters", "outputs", "integrate"]
print("substep phase cycles (block 0):", {names[i]: int(t[i+1]-t[i]) for i in range(10)}, "total", int(t[10]-t[0]))
if os.environ.get("WBC_XSTAMPS") == "walk": print("  walk stamps: 25 before, 27 after prefetch, 28 before level 3, 26 after:", [int(t[i] - t[25]) for i in (27, 28, 26)])
elif os.environ.get("WBC_XSTAMPS"): print("  second solver sweep (stamps 25..31: start | solve+adds | box row | inward | root | outward | response), cycles between:", [int(t[i + 1] - t[i]) for i in range(25, 31)])
print("  contact detail: narrow phase", int(t[18]-t[6]), "set-up", int(t[7]-t[18]), "| first iteration: solve", int(t[20]-t[7]), "gather", int(t[21]-t[20]),
      "inward", int(t[22]-t[21]), "root", int(t[23]-t[22]), "outward", int(t[24]-t[23]))
g.step(acts[0]); torch.cuda.synchronize()
t = buf.cpu().numpy()
n2 = ["setup+load+action", "4 substeps", "rigid bodies", "lane-0 task logic + rewards", "reset", "observe+store"]
print("step phase cycles (block 0):", {n2[i]: int(t[12+i]-t[11+i]) for i in range(6)}, "total", int(t[17]-t[11]))
L.wbc_debug_set_step_timing(None)
