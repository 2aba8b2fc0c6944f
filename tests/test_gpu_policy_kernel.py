"""GPU parity of the fused ActorCritic inference kernel (wbc_policy_act, fp32 MFMA) against the plain
PyTorch fp32 modules (which tests/test_ppo_parity.py pins to the reference's rsl_rl). Tolerance 2e-5
absolute on means/values (both are fp32 GEMM chains with different summation orders), 2e-4 on the summed
log-probabilities."""
import numpy as np
import pytest
import torch

import golden_procedure as gp
from wbc_amd.rsl_rl.modules import ActorCritic

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows", [1, 15, 16, 17, 33, 1000, 2048, 4096, 8224])
def test_fused_act_matches_torch_modules(rows):
    torch.manual_seed(3)
    ac = ActorCritic(76, 76, 18, **gp.POLICY_KW).cuda()
    with torch.no_grad():                      # non-trivial biases / std so that a dropped bias or a transposed weight shows
        for p in ac.parameters():
            p.add_(0.05 * torch.randn_like(p))
        ac.std.copy_(0.3 + torch.rand_like(ac.std))
    obs = torch.randn(rows, 860, device="cuda")
    eps = torch.randn(rows, 18, device="cuda")
    assert ac.fused_act_supported(obs)
    with torch.inference_mode():
        actions, mean, logp, values = ac.fused_act(obs, eps)
        ref_mean = ac.act_inference(obs)
        ref_values = ac.evaluate(obs)
        ac.update_distribution(obs, False)
        ref_actions = ref_mean + ac.std * eps
        ref_logp = ac.get_actions_log_prob(ref_actions)
    torch.cuda.synchronize()
    np.testing.assert_allclose(mean.cpu().numpy(), ref_mean.cpu().numpy(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(values.cpu().numpy(), ref_values.cpu().numpy(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(actions.cpu().numpy(), ref_actions.cpu().numpy(), atol=3e-5, rtol=1e-5)
    np.testing.assert_allclose(logp.cpu().numpy(), ref_logp.cpu().numpy(), atol=2e-4, rtol=1e-5)
    # acting on the mean
    with torch.inference_mode():
        a0, m0, _, _ = ac.fused_act(obs, None)
    np.testing.assert_allclose(a0.cpu().numpy(), m0.cpu().numpy(), atol=0, rtol=0)


def test_fused_act_is_what_ppo_act_uses():
    from wbc_amd.rsl_rl.algorithms import PPO
    torch.manual_seed(1)
    ac = ActorCritic(76, 76, 18, **gp.POLICY_KW)
    alg = PPO(ac, device="cuda:0", **gp.ALG_KW)
    alg.init_storage(64, 4, [860], [None], [18])
    obs = torch.randn(64, 860, device="cuda")
    with torch.inference_mode():
        torch.manual_seed(5)
        a = alg.act(obs, obs, False)
        tr = alg.transition
        fused = [t.clone() for t in (a, tr.values, tr.actions_log_prob, tr.action_mean, tr.action_sigma)]
        alg.fused_rollout = False
        torch.manual_seed(5)
        b = alg.act(obs, obs, False)
        eager = [b, tr.values, tr.actions_log_prob, tr.action_mean, tr.action_sigma]
    for f, e, tol in zip(fused[1:], eager[1:], (2e-5, 1e-3, 2e-5, 0)):
        # values / mean / sigma agree; log-probs are of differently sampled actions so only compare shapes there
        if tol == 1e-3:
            assert f.shape == e.shape
        else:
            np.testing.assert_allclose(f.cpu().numpy(), e.cpu().numpy(), atol=tol, rtol=1e-5)
    assert fused[0].shape == eager[0].shape == (64, 18)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 24, 25, 4096, 5000])
def test_fused_history_encoder_matches_torch(n):
    """wbc_hist_latent (csrc/wbc_hist_kernel.hip) against the module's own torch forward (AC:39-84): same weights, random
    history blocks; fp32 sums in a different order -> 2e-6 absolute on O(1) outputs."""
    torch.manual_seed(5)
    ac = ActorCritic(76, 76, 18, **gp.POLICY_KW).cuda()
    obs = torch.randn(n, 860, device="cuda")
    with torch.no_grad():
        assert ac.actor._fused_hist_supported(obs)
        fused = ac.actor.infer_hist_latent(obs)
        hist = obs[:, -760:]
        ref = ac.actor.history_encoder(hist.view(-1, 10, 76))
    assert fused.shape == ref.shape == (n, 20)
    np.testing.assert_allclose(fused.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=2e-6)


@pytest.mark.gpu
def test_fused_act_student_path_matches_torch():
    """hist_encoding=True (AC:206-209): the history latent (wbc_hist_latent) replaces the privileged encoder's output in
    the fused inference kernel; mean action and values against the module's eager forward."""
    torch.manual_seed(9)
    ac = ActorCritic(76, 76, 18, **gp.POLICY_KW).cuda()
    obs = torch.randn(1000, 860, device="cuda")
    with torch.no_grad():
        latent = ac.actor.infer_hist_latent(obs)
        actions, mean, logp, values = ac.fused_act(obs, None, latent=latent)
        ref_mean = ac.actor(obs, hist_encoding=True)
        ref_val = ac.evaluate(obs)
    np.testing.assert_allclose(mean.cpu().numpy(), ref_mean.cpu().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(actions.cpu().numpy(), mean.cpu().numpy(), rtol=0, atol=0)      # eps=None acts on the mean
    np.testing.assert_allclose(values.cpu().numpy(), ref_val.cpu().numpy(), rtol=2e-5, atol=2e-5)
