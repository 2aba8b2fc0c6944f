"""Export / play path: the deployed modules reproduce the training actor's student path."""
import os

import numpy as np
import pytest
import torch

import golden_procedure as gp
from wbc_amd import deploy
from wbc_amd.rsl_rl.modules import ActorCritic


def _ac(seed=2):
    torch.manual_seed(seed)
    return ActorCritic(76, 76, 18, **gp.POLICY_KW)


def test_deploy_actor_state_dict_equals_training_actor():
    ac = _ac()
    dep = deploy.DeployActor(ac.actor)
    a, b = ac.actor.state_dict(), dep.state_dict()
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_traced_modules_reproduce_student_path(tmp_path):
    ac = _ac()
    obs = torch.randn(7, 860)
    with torch.no_grad():
        want = ac.act_inference(obs, hist_encoding=True)
        want_latent = ac.actor.infer_hist_latent(obs)
    p_actor, p_enc = deploy.trace_actor_and_hist_encoder(ac, str(tmp_path / "traced"), "run_100")
    assert os.path.basename(p_actor) == "run_100_actor_jit.pt" and os.path.basename(p_enc) == "run_100_hist_encoder_jit.pt"
    actor, enc = torch.jit.load(p_actor), torch.jit.load(p_enc)
    with torch.no_grad():
        latent = enc(obs[:, 100:])                                         # play.py:119
        got = actor(torch.cat((obs[:, :76], latent), dim=1))               # play.py:120
    np.testing.assert_allclose(latent.numpy(), want_latent.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=1e-6)
    assert got.shape == (7, 18) and latent.shape == (7, 20)


def test_scripted_policy_and_actor_state(tmp_path):
    ac = _ac(3)
    path = deploy.export_policy_as_jit(ac, str(tmp_path / "exported" / "policies"))
    assert path.endswith("policy_1.pt")
    mod = torch.jit.load(path)
    x = torch.randn(4, 96)
    with torch.no_grad():
        ref = deploy.DeployActor(ac.actor)(x)
        np.testing.assert_allclose(mod(x).numpy(), ref.numpy(), rtol=1e-6, atol=1e-6)
    p = deploy.save_actor_state(ac, str(tmp_path / "exported"), "model_100")
    sd = torch.load(p)
    fresh = _ac(4)
    fresh.actor.load_state_dict(sd)
    with torch.no_grad():
        obs = torch.randn(3, 860)
        np.testing.assert_allclose(fresh.act_inference(obs).numpy(), ac.act_inference(obs).numpy(), rtol=0, atol=0)


@pytest.mark.gpu
def test_play_loop_traced_equals_policy_on_gpu(tmp_path):
    from wbc_amd.config import WidowGo1RoughCfg
    from wbc_amd.envs import WidowGo1
    ac = _ac(5).cuda()
    p_actor, p_enc = deploy.trace_actor_and_hist_encoder(ac, str(tmp_path), "t")
    actor, enc = torch.jit.load(p_actor, map_location="cuda:0"), torch.jit.load(p_enc, map_location="cuda:0")
    logs = []
    for use_jit in (False, True):
        cfg = deploy.play_cfg_overrides(WidowGo1RoughCfg())
        cfg.terrain.mesh_type = "plane"
        assert cfg.env.num_envs == 5
        env = WidowGo1(cfg, sim_device="cuda:0", seed=3)
        logs.append(deploy.play(env, ac.act_inference, steps=30, traced_actor=actor if use_jit else None,
                                traced_hist_encoder=enc if use_jit else None).cpu().numpy())
    assert logs[0].shape == (30, 2) and np.isfinite(logs[0]).all()
    np.testing.assert_allclose(logs[1], logs[0], rtol=1e-3, atol=1e-4)       # same rollout up to fp32 round-off of the two forward paths
