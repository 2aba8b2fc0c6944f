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
# the permutation mini_batch_generator draws (RS:163) and the storage contents at the first update() are recorded next to what
# the reference computes from them, so that a different implementation (the fused GPU learner) can be fed EXACTLY the reference's
# inputs and compared with the reference's outputs directly (tests/test_ppo_parity.py, one hop)
import golden_procedure as gp   # noqa: E402

_real_randperm = torch.randperm
perms = []


def _recording_randperm(n, *a, **kw):
    p = _real_randperm(n, *a, **kw)
    perms.append(p.clone())
    return p


captured = {}
_real_update = PPO.update


def _capturing_update(self):
    if not captured:
        st = self.storage
        for name in gp.STORAGE_FIELDS:
            captured[name] = getattr(st, name).detach().cpu().numpy().copy()
        captured["params_before"] = gp.flat_params(self.actor_critic)
    return _real_update(self)


captured_dagger = {}
_real_update_dagger = PPO.update_dagger


def _capturing_update_dagger(self):
    st = self.storage
    for name in gp.STORAGE_FIELDS:
        captured_dagger[name] = getattr(st, name).detach().cpu().numpy().copy()
    captured_dagger["params_before"] = gp.flat_params(self.actor_critic)
    captured_dagger["perm_index"] = len(perms)              # the permutation this call is about to draw
    r = _real_update_dagger(self)
    captured_dagger["params_after"] = gp.flat_params(self.actor_critic)
    return r


torch.randperm = _recording_randperm
PPO.update = _capturing_update
PPO.update_dagger = _capturing_update_dagger
with contextlib.redirect_stdout(io.StringIO()):
    res = run_procedure(ActorCritic, PPO, device="cpu")
PPO.update = _real_update
PPO.update_dagger = _real_update_dagger
for k, v in res.items():
    out[k] = v
for k, v in captured.items():
    out["it0_storage_" + k] = v
out["it0_perm"] = perms[0].numpy()
# the DAgger iteration (it2 of the procedure: student rollouts, then update_dagger, ppo.py:265-291): its storage, permutation, the
# parameters before and after
for k in gp.STORAGE_FIELDS:
    out["it2_storage_" + k] = captured_dagger[k]
out["it2_params_before"] = captured_dagger["params_before"]
out["it2_params_after"] = captured_dagger["params_after"]
out["it2_perm"] = perms[captured_dagger["perm_index"]].numpy()

# bench-shaped minibatch: N = 1024, T = 40 with ONE minibatch per epoch -> 40 960 rows per minibatch call, the shape of the bench's
# update (4096 envs x 40 steps / 4 minibatches); synthetic storage (golden_procedure.synthetic_storage), 2 epochs
BN, BT = 1024, 40
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    ac = ActorCritic(76, 76, 18, **POLICY_KW)
    kw = dict(ALG_KW, num_mini_batches=1, num_learning_epochs=2)
    alg = PPO(ac, device="cpu", **kw)
alg.counter = 3500
alg.init_storage(BN, BT, [860], [None], [18])
last = gp.synthetic_storage(alg.storage, 777)
alg.storage.compute_returns(last, ALG_KW["gamma"], ALG_KW["lam"])
out["bench_returns"] = alg.storage.returns.numpy().copy()
out["bench_advantages"] = alg.storage.advantages.numpy().copy()
perms.clear()
with contextlib.redirect_stdout(io.StringIO()):
    stats = alg.update()
torch.randperm = _real_randperm
out["bench_perm"] = perms[0].numpy()
out["bench_stats"] = np.array([float(x) for x in stats])
out["bench_params"] = gp.flat_params(ac)
out["bench_std"] = ac.std.detach().numpy().copy()
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
