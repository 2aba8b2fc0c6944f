#!/usr/bin/env python3
"""Generate tests/golden/ppo_reference.npz by running the REFERENCE's own rsl_rl code
(/root/reference/rsl_rl, imported read-only) on seeded synthetic rollouts.

Run in the build container only (the reference tree does not exist on the GPU box); the tests
replay the same seeded procedure through wbc_amd.rsl_rl and compare with the stored outputs.
Covers BASELINE.json configs[0]: PPO.update() on a synthetic 64-env x 24-step RolloutStorage
 with the widowGo1 hyper-parameters, plus update_dagger and the
GAE known-answer of SURVEY.md section 8c."""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/rsl_rl")
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import io
import contextlib

import numpy as np
import torch

from golden_procedure import POLICY_KW, ALG_KW, run_procedure, gae_known_answer_inputs  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    from rsl_rl.algorithms import PPO
    from rsl_rl.modules import ActorCritic
    from rsl_rl.storage import RolloutStorage

out = {}
with contextlib.redirect_stdout(io.StringIO()):
    res = run_procedure(ActorCritic, PPO, device="cpu")
for k, v in res.items():
    out[k] = v
# GAE known answer
rew, val, dones, last = gae_known_answer_inputs()
st = RolloutStorage(2, 4, [3], [None], [1])
st.rewards.copy_(rew); st.values.copy_(val); st.dones.copy_(dones)
st.compute_returns(last, 0.99, 0.95)
out["gae_returns"] = st.returns.numpy().copy()
out["gae_advantages"] = st.advantages.numpy().copy()
path = os.path.join(HERE, "..", "tests", "golden", "ppo_reference.npz")
np.savez_compressed(path, **out)
print("wrote", path, {k: np.asarray(v).shape for k, v in out.items()})
print("gae returns", out["gae_returns"].flatten())
