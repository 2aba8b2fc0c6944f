"""wbc_hist_train_grad (one DAgger minibatch: history-encoder forward + loss + backward + weight gradients) and wbc_priv_latent
at the benchmark's size: 40960 gathered rows of 163840."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
_timing_lib = os.path.join(ROOT, "deep-whole-body-control_amd", "wbc_amd", "libwbc_amd_histtiming.so")   # tools/build_variant.py histtiming -DWBC_HIST_TIMING
if os.path.exists(_timing_lib) and os.environ.get("WBC_STAMPS"):
    os.environ["WBC_AMD_LIB"] = _timing_lib
import torch
import golden_procedure as gp
from wbc_amd.native import check, lib
from wbc_amd.rsl_rl.modules import ActorCritic
torch.manual_seed(0)
L = lib()
ac = ActorCritic(76, 76, 18, **gp.POLICY_KW).cuda()
TN, B = 163840, int(sys.argv[1]) if len(sys.argv) > 1 else 40960
iters = int(os.environ.get("WBC_ITERS", "20"))
obs = torch.randn(TN, 860, device="cuda")
idx = torch.randperm(TN, device="cuda")[:B].contiguous()
he, pe = ac.actor.history_encoder, ac.actor.priv_encoder
hp = [he.encoder[0].weight, he.encoder[0].bias, he.conv_layers[0].weight, he.conv_layers[0].bias, he.conv_layers[2].weight,
      he.conv_layers[2].bias, he.linear_output[0].weight, he.linear_output[0].bias]
pp = [pe[0].weight, pe[0].bias, pe[2].weight, pe[2].bias]
table, ptable = (C.c_void_p * 8)(*[p.data_ptr() for p in hp]), (C.c_void_p * 4)(*[p.data_ptr() for p in pp])
priv = torch.empty(TN, 20, device="cuda")
grad = torch.zeros(L.wbc_hist_train_grad_floats(), device="cuda")
ws = torch.empty(L.wbc_hist_train_workspace_floats(), device="cuda")
st = torch.cuda.current_stream().cuda_stream


def ev(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000


t_priv = ev(lambda: check(L.wbc_priv_latent(ptable, obs.data_ptr(), priv.data_ptr(), TN, st)), iters)
t_grad = ev(lambda: check(L.wbc_hist_train_grad(table, obs.data_ptr(), priv.data_ptr(), idx.data_ptr(), B, ws.data_ptr(), grad.data_ptr(), st)), iters)
flops = 2 * 80e3 * B
print(f"priv_latent {TN} rows: {t_priv:.1f} us; hist_train_grad (+reduce) {B} rows: {t_grad:.1f} us = {flops / t_grad / 1e6:.1f} TFLOP/s, "
      f"{B * 3040 / t_grad / 1e3:.1f} GB/s of history rows")

if os.environ.get("WBC_STAMPS") and hasattr(L, "wbc_debug_set_hist_timing"):
    L.wbc_debug_set_hist_timing.argtypes = [C.c_void_p]
    buf = torch.zeros(16, dtype=torch.int64, device="cuda")
    L.wbc_debug_set_hist_timing(buf.data_ptr())
    check(L.wbc_hist_train_grad(table, obs.data_ptr(), priv.data_ptr(), idx.data_ptr(), B, ws.data_ptr(), grad.data_ptr(), st)); torch.cuda.synchronize()
    t = buf.cpu().numpy()
    names = ["projection fwd", "conv1 fwd", "conv2 fwd", "linear fwd", "loss", "linear bwd", "conv2 bwd", "conv1 wgrad", "dz1", "projection wgrad"]
    print("hist_train phase cycles (workgroup 0, first group):", {names[i]: int(t[i + 1] - t[i]) for i in range(10)}, "total", int(t[10] - t[0]))
    L.wbc_debug_set_hist_timing(None)
