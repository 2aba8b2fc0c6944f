"""Instruction counts per phase of wbc_step_kernel: launches whose waves all end at stamp k (build: tools/build_variant.py phaseexit
-DWBC_PHASE_EXIT), run under rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES (tools/r06_phase_counts.sh, which takes
the differences between consecutive launches).  usage: python tools/phase_counts.py [N]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ["WBC_AMD_LIB"] = os.path.join(ROOT, "deep-whole-body-control_amd", "wbc_amd", "libwbc_amd_phaseexit.so")
import torch
import helpers
from wbc_amd import abi
from wbc_amd.config import WidowGo1RoughCfg
from wbc_amd.native import lib
SEQ = [11, 12, 0, 25, 27, 28, 26, 1, 2, 3, 4, 5, 6, 18, 7, 8, 9, 10, 13, 14, 15, -1]
if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    m = abi.load_default_model(); cfg = WidowGo1RoughCfg(); tc = abi.fill_task_cfg(cfg, m)
    g = helpers.make_gpu(dict(model=m, wmodel=abi.fill_model(m), cfg=cfg, tcfg=tc), n, helpers.random_env_params(n, 0))
    g.reset_all()
    acts = [torch.randn(n, 18, device="cuda") * 0.5 for _ in range(8)]
    L = lib(); L.wbc_debug_set_phase_exit.argtypes = [C.c_int]
    L.wbc_debug_set_phase_exit(-1)
    for i in range(77): g.step(acts[i % 8])
    torch.cuda.synchronize()
    for k in SEQ:
        L.wbc_debug_set_phase_exit(k)
        g.step(acts[0]); torch.cuda.synchronize()
