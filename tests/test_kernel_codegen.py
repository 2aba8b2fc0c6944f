"""Code-generation invariants of wbc_step_kernel that the design rests on (DESIGN.md sections 2.1, 7.1b), checked on the assembly hipcc
emits for gfx950 (cross-compiles without a GPU; ~40 s):
  * 128 VGPRs and 10 KB of LDS per wavefront-workgroup -> 16 robots per CU, all 4096 envs of the bench resident at once;
  * no scratch memory (a register spill in this kernel cost 5-10 % twice: rounds 2 and 4);
  * no flat memory instructions: the constant block and the tensor table are read through the constant address space (scalar loads
    whatever wavefront fences surround them) and tensor bases are typed global -- through generic pointers every constant read after
    the first fence was a vector load on the dependent path (148 -> 137 us when that was fixed)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "deep-whole-body-control_amd", "csrc", "wbc_step_kernel.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def step_kernel_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = str(tmp_path_factory.mktemp("asm") / "step.s")
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    flags = [f for f in g.COMMON_FLAGS if f != "-fPIC"] + g.EXTRA_FLAGS.get("wbc_step_kernel.hip", [])     # the build's flags for this source
    subprocess.check_call([HIPCC] + flags + ["-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), "-o", out, SRC],
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
    a = text.index("\nwbc_step_kernel:")
    body = text[a:text.index("s_endpgm", a)]
    meta = _entry(text, "wbc_step_kernel")
    return body, meta, text


def _entry(text, kernel):
    """The amdhsa.kernels metadata entry of `kernel` (entries start at '- .agpr_count', keys in alphabetical order)."""
    entries = text[text.index("amdhsa.kernels:"):].split("\n  - .agpr_count")
    return next(e for e in entries if re.search(r"\.name:\s+%s\n" % re.escape(kernel), e))


def _meta(meta, key):
    return int(re.search(r"\.%s:\s+(\d+)" % key, meta).group(1))


def test_step_kernel_keeps_sixteen_robots_per_cu(step_kernel_asm):
    body, meta, text = step_kernel_asm
    assert _meta(meta, "vgpr_count") <= 128                       # 4 waves per SIMD
    assert _meta(meta, "group_segment_fixed_size") <= 10240       # 16 workgroups in 160 KB of LDS
    assert _meta(meta, "private_segment_fixed_size") == 0         # no scratch
    assert "scratch_" not in body


def test_step_kernel_reads_constants_through_the_scalar_path(step_kernel_asm):
    body, meta, text = step_kernel_asm
    assert not re.search(r"\bflat_(load|store|atomic)", body)
    assert len(re.findall(r"\bs_load_dword", body)) > 60          # constants and tensor pointers arrive by scalar loads
    # the per-env rows are (scalar base) + (32-bit lane offset), or ONE 64-bit lane address per row with immediate offsets for the
    # row's accesses (which of the two the instruction selector picks changed with the build flags in round 6): either way there is
    # well under one 64-bit address computation per global access
    accesses = re.findall(r"\bglobal_(?:store|load)_\w+", body)
    addr64 = re.findall(r"\bv_lshl_add_u64\b|\bv_add_co_u32\b", body)
    assert len(accesses) > 100 and len(addr64) <= 0.7 * len(accesses)


@pytest.mark.parametrize("src, kernel, max_vgpr", [
    ("wbc_ppo_kernel.hip", "ppo_chain_kernel", 256),            # two units per SIMD (section 2.3)
    ("wbc_ppo_kernel.hip", "ppo_wgrad_kernel", 256),            # two workgroups per CU, three operand sets in flight
    ("wbc_policy_kernel.hip", "wbc_policy_act16_kernel", 256),  # an actor and a critic tile per CU
    ("wbc_hist_train_kernel.hip", "hist_train_kernel", 256),
])
def test_mfma_kernels_keep_their_occupancy_and_do_not_spill(tmp_path, src, kernel, max_vgpr):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = str(tmp_path / "k.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                           "-I" + os.path.join(ROOT, "include"), "-o", out, os.path.join(ROOT, "deep-whole-body-control_amd", "csrc", src)],
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
    meta = _entry(text, kernel)
    assert _meta(meta, "private_segment_fixed_size") == 0
    assert _meta(meta, "vgpr_count") <= max_vgpr
    body = text[text.index("\n%s:" % kernel):]
    body = body[:body.index("s_endpgm")]
    assert not re.search(r"\bflat_(load|store)", body) and "scratch_" not in body
