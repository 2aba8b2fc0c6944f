"""Timing of one fused PPO minibatch (forward+backward kernel, weight-gradient kernel) at the benchmark's size."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
_lib = os.path.join(ROOT, "deep-whole-body-control_amd", "wbc_amd", "libwbc_amd_ppotiming.so")   # tools/build_variant.py ppotiming -DWBC_PPO_TIMING
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
    buf = torch.zeros(128, dtype=torch.int64, device=dev)
    L.wbc_debug_set_ppo_timing(buf.data_ptr())
    run(); torch.cuda.synchronize()
    t = buf.cpu().numpy()
    names = ["load x", "forward 16 layers", "z stash", "losses"] + [f"bwd stage {i}" for i in range(14)]
    d = [int(t[i + 1] - t[i]) for i in range(18)]
    print("workgroup 0 cycles:", dict(zip(names, d)), "total", int(t[18] - t[0]))
    for l in range(16):
        q = t[32 + 4 * l: 36 + 4 * l]
        nxt = t[32 + 4 * (l + 1)] if l < 15 else q[3]
        print(f"  fwd layer {l:2d}: mfma chain {int(q[1]-q[0]):6d}  epilogue {int(q[2]-q[1]):6d}  barrier {int(q[3]-q[2]):6d}  to next layer start {int(nxt-q[3]):6d}")
    L.wbc_debug_set_ppo_timing(None)
