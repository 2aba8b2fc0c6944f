#!/usr/bin/env python3
"""Write tests/golden/reference_checkpoint.pt + reference_checkpoint_io.npz with the REFERENCE's own rsl_rl classes
(/root/reference/rsl_rl, imported read-only): an ActorCritic + Adam after one PPO.update() on a seeded synthetic rollout,
saved exactly as OnPolicyRunner.save does (on_policy_runner.py:276-282: model_state_dict, optimizer_state_dict, iter, infos),
and the reference's act_inference outputs (teacher and student latent) on fixed observations.
tests/test_checkpoint_golden.py loads the file through wbc_amd's OnPolicyRunner.load and must reproduce those outputs.

The file is 2 MB: 168,698 fp32 parameters plus Adam's two moments for the 33 tensors update() trains (quirk L6: the history
encoder's parameters have no state in this optimiser).

Run in the build container only: python tools/make_golden_checkpoint.py"""
import contextlib
import io
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/rsl_rl")
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import numpy as np
import torch

from golden_procedure import ALG_KW, N, POLICY_KW, T, synthetic_rollout  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    from rsl_rl.algorithms import PPO
    from rsl_rl.modules import ActorCritic
    torch.manual_seed(4)
    ac = ActorCritic(76, 76, 18, **POLICY_KW)
    alg = PPO(ac, device="cpu", **ALG_KW)
alg.counter = 3500
alg.init_storage(N, T, [860], [None], [18])
obs, rew, arm, dones, touts = synthetic_rollout(300)
torch.manual_seed(5)
with torch.inference_mode():
    for t in range(T):
        alg.act(obs[t], obs[t], False)
        alg.process_env_step(rew[t], arm[t], dones[t], {"time_outs": touts[t]})
    alg.compute_returns(obs[T])
alg.update()
gold = os.path.join(HERE, "..", "tests", "golden")
path = os.path.join(gold, "reference_checkpoint.pt")
torch.save({"model_state_dict": ac.state_dict(), "optimizer_state_dict": alg.optimizer.state_dict(), "iter": 1234,
            "infos": None}, path)                                                     # OPR:276-282
g = torch.Generator().manual_seed(11)
x = torch.randn(32, 860, generator=g)
ac.eval()
with torch.inference_mode():
    out = {"obs": x.numpy(), "act_teacher": ac.act_inference(x, hist_encoding=False).numpy(),
           "act_student": ac.act_inference(x, hist_encoding=True).numpy(), "value": ac.evaluate(x).numpy(),
           "std": ac.std.detach().numpy()}
st = alg.optimizer.state_dict()["state"]
out["adam_steps"] = np.array([float(v["step"]) for v in st.values()])
out["adam_exp_avg_abs_sum"] = np.array([float(v["exp_avg"].abs().sum()) for v in st.values()])
np.savez_compressed(os.path.join(gold, "reference_checkpoint_io.npz"), **out)
print("wrote", path, f"{os.path.getsize(path) / 1e6:.2f} MB; optimizer entries with state: {len(st)}")
