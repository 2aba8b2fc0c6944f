import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_procedure as gp
from wbc_amd.rsl_rl.modules import ActorCritic
torch.manual_seed(0)
ac = ActorCritic(76, 76, 18, **gp.POLICY_KW).cuda()
for n in (4096, 8192, 16384, 40960):
    obs = torch.randn(n, 860, device="cuda"); eps = torch.randn(n, 18, device="cuda")
    with torch.inference_mode():
        for _ in range(5): ac.fused_act(obs, eps)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): ac.fused_act(obs, eps)
        e1.record(); torch.cuda.synchronize()
    print(f"fused_act rows={n}: {e0.elapsed_time(e1)/50*1000:.1f} us")

# stage timing of block 0 (clock64 stamps)
import ctypes as C
from wbc_amd.native import lib
L = lib()
L.wbc_debug_set_policy_timing.argtypes = [C.c_void_p]
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
L.wbc_debug_set_policy_timing(buf.data_ptr())
obs = torch.randn(4096, 860, device="cuda"); eps = torch.randn(4096, 18, device="cuda")
with torch.inference_mode():
    ac.fused_act(obs, eps); torch.cuda.synchronize()
t = buf.cpu().numpy()
n = int((t != 0).sum())
d = (t[1:n] - t[:n-1])
print("stamps", n, "total cycles", t[n-1] - t[0])
print("stage cycles:", d.tolist())
L.wbc_debug_set_policy_timing(None)
