"""bench.py's host-side pieces that need no GPU: the CPU-baseline leg on a small stand-in rollout (always this package's eager
path, which tests/test_ppo_parity.py pins to the reference: the same classes on every box, nothing outside the repository
executed inside the bench), the usable-core count, and the launcher's argument handling."""
import os
import subprocess

import pytest
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from wbc_amd.config import WidowGo1RoughCfgPPO, class_to_dict  # noqa: E402
from wbc_amd.rsl_rl.algorithms import PPO  # noqa: E402
from wbc_amd.rsl_rl.modules import ActorCritic  # noqa: E402


def _stand_in_runner(n=32, t=8):
    tr = class_to_dict(WidowGo1RoughCfgPPO())
    torch.manual_seed(0)
    ac = ActorCritic(76, 76, 18, **tr["policy"], num_priv=24, num_hist=10, num_prop=76)
    alg = PPO(ac, device="cpu", **tr["algorithm"])
    alg.init_storage(n, t, [860], [None], [18])
    st = alg.storage
    for x in (st.observations, st.actions, st.values, st.actions_log_prob, st.mu):
        x.normal_()
    st.rewards.normal_().mul_(0.01)
    st.sigma.fill_(1.0)
    return types.SimpleNamespace(alg=alg)


def test_cpu_baseline_leg_reports_the_contract_fields():
    out = bench.cpu_baseline(_stand_in_runner(), sim_sample_envs=16)
    assert set(out) >= {"value", "unit", "cores", "kind", "sample", "update_s", "compute_returns_s", "update_dagger_s"}
    assert out["kind"] == "port" and "learner only" in out["scope"]
    assert out["unit"] == "env-steps/s" and out["value"] > 0 and 1 <= out["cores"] <= 16
    assert out["sim_port"].get("value", 0) > 0


def test_cpu_baseline_never_imports_the_reference_tree():
    before = set(sys.modules)
    bench.cpu_baseline(_stand_in_runner(16, 4), sim_sample_envs=8)
    new = [m for m in set(sys.modules) - before if getattr(sys.modules[m], "__file__", None) and "/root/reference" in (sys.modules[m].__file__ or "")]
    assert not new and not any("/root/reference" in p for p in sys.path)


@pytest.mark.skipif(not os.path.isdir(bench.REFERENCE_RSL_RL), reason="the reference tree exists in the build container only")
def test_cpu_baseline_can_time_the_reference_classes_in_the_build_container():
    """`bench.py --cpu-baseline-only`: kind 'reference' = the reference's own rsl_rl PPO / ActorCritic on the same storage contents
    (profiles/r05_cpu_baseline_reference_vs_port.json holds both lines from one box)."""
    r = bench._synthetic_runner(32, 4)
    try:
        out = bench.cpu_baseline(r, sim_sample_envs=8, learner="reference")
    finally:
        while bench.REFERENCE_RSL_RL in sys.path:
            sys.path.remove(bench.REFERENCE_RSL_RL)
    assert out["kind"] == "reference" and bench.REFERENCE_RSL_RL in out["sample"] and out["value"] > 0
    port = bench.cpu_baseline(r, sim_sample_envs=8, learner="port")
    assert port["kind"] == "port" and port["value"] > 0


def test_world_size_mismatch_is_an_error_message_not_an_assert():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout)
