"""Timing of one fused PPO minibatch (forward+backward kernel, weight-gradient kernel) at the benchmark's size."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
_lib = os.path.join(ROOT, "deep-whole-body-control_amd", "wbc_amd", os.environ.get("WBC_TIMING_LIB", "libwbc_amd_ppotiming.so"))   # tools/build_variant.py ppotiming -DWBC_PPO_TIMING
if os.path.exists(_lib) and os.environ.get("WBC_STAMPS"):
    os.environ["WBC_AMD_LIB"] = _lib
import torch
import golden_procedure as gp
from wbc_amd.rsl_rl.modules import ActorCritic
from wbc_amd.native import lib, check
torch.manual_seed(0)
L = lib()
ac = ActorCritic(76, 76, 18, **gp.POLICY_KW).cuda()
TN = 163840
B = int(sys.argv[1]) if len(sys.argv) > 1 else 40960
dev = "cuda"
obs = torch.randn(TN, 860, device=dev); actions = torch.randn(TN, 18, device=dev); values = torch.randn(TN, 2, device=dev)
adv = torch.randn(TN, 2, device=dev); ret = torch.randn(TN, 2, device=dev); logp = -torch.rand(TN, 2, device=dev) * 20
hist = torch.randn(TN, 20, device=dev); idx = torch.randperm(TN, device=dev)[:B].contiguous()
table = ac.fused_param_table()
ws = torch.empty(L.wbc_ppo_workspace_floats(B), device=dev); grad = torch.zeros(L.wbc_ppo_grad_floats(), device=dev)
stream = torch.cuda.current_stream().cuda_stream
def run():
    check(L.wbc_ppo_minibatch_grad(table, obs.data_ptr(), actions.data_ptr(), values.data_ptr(), adv.data_ptr(), ret.data_ptr(), logp.data_ptr(),
                                   hist.data_ptr(), idx.data_ptr(), B, 0.2, 1.0, 0.5, 0.1, 1, ws.data_ptr(), grad.data_ptr(), None, stream), "grad")
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
NIT = int(os.environ.get("WBC_ITERS", "100"))
for _ in range(NIT): run()
e1.record(); torch.cuda.synchronize()
print(f"minibatch B={B}: {e0.elapsed_time(e1)/NIT*1000:.1f} us per call (pack + fwd_bwd + wgrad + reducers)")
if os.environ.get("WBC_STAMPS"):
    L.wbc_debug_set_ppo_timing.argtypes = [C.c_void_p]
    buf = torch.zeros(128 + 4 * 2 * ((B + 15) // 16) + 64, dtype=torch.int64, device=dev)
    L.wbc_debug_set_ppo_timing(buf.data_ptr())
    run(); torch.cuda.synchronize()
    t = buf.cpu().numpy()
    names = ["ring init + gather", "priv0/priv2 (actor)", "backbone", "h0 L0", "h0 L2", "h0 head", "h1 L0", "h1 L2", "h1 head", "(pad)", "loss", "h0 headT", "h0 L2T", "h0 L0T",
             "h1 headT", "h1 L2T", "h1 L0T", "backbone ELU'", "latent/priv2T (actor)"]
    for base, who in ((0, "actor unit 0"), (64, "critic unit 1")):
        q = t[base:base + 20]
        d = {names[i]: int(q[i + 1] - q[i]) for i in range(19) if q[i + 1] > 0 and q[i] > 0}
        print(who, "cycles:", d, "total", int(max(q) - q[0]))
    import numpy as np
    nu = 2 * ((B + 15) // 16)
    u = t[128:128 + 4 * nu].reshape(nu, 4).astype(np.float64)
    if u[:, 1].max() > 0:
        t0 = u[:, 0].min()
        st, en = (u[:, 0] - t0) / 100.0, (u[:, 1] - t0) / 100.0                      # us (100 MHz wall clock)
        dur = en - st
        ghz = (u[:, 3] - u[:, 2]) / (dur * 1000.0)
        print(f"units {nu}: kernel span {en.max():.1f} us; unit duration actor {dur[0::2].mean():.1f} us critic {dur[1::2].mean():.1f} us; shader clock {np.median(ghz):.2f} GHz")
        order = np.argsort(st)
        for q in (0.0, 0.25, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0):
            i = order[min(nu - 1, int(q * (nu - 1)))]
            print(f"   start quantile {q:.2f}: unit {i} starts {st[i]:.1f} us, runs {dur[i]:.1f} us")
        edges = np.linspace(0, en.max(), 11)
        active = [(int(((st < b) & (en > a)).sum())) for a, b in zip(edges[:-1], edges[1:])]
        print("   units alive per tenth of the span:", active)
    L.wbc_debug_set_ppo_timing(None)
