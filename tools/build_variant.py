"""Build a kernel variant of libwbc_amd.so for A/B runs: python tools/build_variant.py <name> [-DFLAG ...]
-> deep-whole-body-control_amd/wbc_amd/libwbc_amd_<name>.so; select it with WBC_AMD_LIB=<path>."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(os.path.dirname(g.LIB), f"libwbc_amd_{name}.so")
srcs = [os.path.join(g.CSRC, s) for s in g.SOURCES]
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                       "-Wno-unused-result", "-fno-slp-vectorize", "-o", out] + flags + srcs)
print(out)
