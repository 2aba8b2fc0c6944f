"""Build a kernel variant of libwbc_amd.so for A/B runs: python tools/build_variant.py <name> [-DFLAG ...]
-> deep-whole-body-control_amd/wbc_amd/libwbc_amd_<name>.so; select it with WBC_AMD_LIB=<path>. The flags are added to every
source's own flags (__graft_entry__.EXTRA_FLAGS); NOSTEPFLAGS=1 drops the step kernel's."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
name, flags = sys.argv[1], sys.argv[2:]
if os.environ.get("NOSTEPFLAGS"):
    g.EXTRA_FLAGS.pop("wbc_step_kernel.hip", None)
out = os.path.join(os.path.dirname(g.LIB), f"libwbc_amd_{name}.so")
objs = g.compile_objects(force=True, extra=flags, objdir=os.path.join(ROOT, "build", "obj_" + name))
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
