"""GPU parity of the fused PPO minibatch kernels (wbc_ppo_minibatch_grad: forward + losses + backward +
split-K weight gradients on fp32 MFMA) against PyTorch autograd over the eager update() path, which
tests/test_ppo_parity.py pins to the reference's rsl_rl. Same storage, same permutation, same initial
weights: gradients after the clip, loss statistics and post-step parameters must agree to fp32 round-off
(5e-5 relative to each tensor's largest entry; the reduction length is 40960 rows)."""
import unittest.mock as mock

import numpy as np
import pytest
import torch

import golden_procedure as gp
from wbc_amd.rsl_rl.algorithms import PPO
from wbc_amd.rsl_rl.modules import ActorCritic

pytestmark = pytest.mark.gpu


def _make(N, T, epochs, mbs, fused, counter=3500, **kw):
    torch.manual_seed(1)
    ac = ActorCritic(76, 76, 18, **gp.POLICY_KW)
    akw = dict(gp.ALG_KW)
    akw.update(num_learning_epochs=epochs, num_mini_batches=mbs)
    akw.update(kw)
    alg = PPO(ac, device="cuda:0", **akw)
    alg.fused_update = fused
    alg.counter = counter
    alg.init_storage(N, T, [860], [None], [18])
    return ac, alg


def _fill(alg, N, T, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    st, ac = alg.storage, alg.actor_critic
    with torch.inference_mode():
        st.observations.copy_(torch.randn(T, N, 860, generator=g, device="cuda"))
        for t in range(T):
            o = st.observations[t]
            ac.update_distribution(o, False)
            a = ac.distribution.mean + ac.std * torch.randn(N, 18, generator=g, device="cuda")
            st.actions[t].copy_(a)
            st.values[t].copy_(ac.evaluate(o))
            st.actions_log_prob[t].copy_(ac.get_actions_log_prob(a) + 0.3 * torch.randn(N, 2, generator=g, device="cuda"))  # ratios off 1: clip branches hit
            st.mu[t].copy_(ac.action_mean)
            st.sigma[t].copy_(ac.action_std)
        st.returns.copy_(st.values + 0.5 * torch.randn(T, N, 2, generator=g, device="cuda"))   # outside the value clip for many rows
        st.advantages.copy_(torch.randn(T, N, 2, generator=g, device="cuda"))
        st.step = T


@pytest.mark.parametrize("N,T,mbs,kw", [(64, 8, 1, {}), (100, 7, 1, {}), (31, 1, 1, {}), (11, 3, 1, {}), (4096, 10, 1, {}), (1500, 9, 1, {}), (512, 8, 2, {"use_clipped_value_loss": False, "entropy_coef": 0.01})])
def test_fused_minibatch_gradients_match_autograd(N, T, mbs, kw):
    ac_e, alg_e = _make(N, T, 1, mbs, fused=False, **kw)
    ac_f, alg_f = _make(N, T, 1, mbs, fused=True, **kw)
    _fill(alg_e, N, T)
    for name in ("observations", "actions", "values", "actions_log_prob", "mu", "sigma", "returns", "advantages"):
        getattr(alg_f.storage, name).copy_(getattr(alg_e.storage, name))
    alg_f.storage.step = T
    assert alg_f._fused_update_supported()
    perm = torch.randperm((N * T // mbs) * mbs, device="cuda")
    with mock.patch("torch.randperm", lambda n, **k: perm):
        out_e = alg_e.update()
        out_f = alg_f.update()
    np.testing.assert_allclose(out_f, out_e, rtol=2e-4, atol=1e-6)
    ge, gf = dict(ac_e.named_parameters()), dict(ac_f.named_parameters())
    for name, p in ge.items():
        q = gf[name]
        if p.grad is None:
            assert q.grad is None, name
            continue
        scale = p.grad.abs().max().item() + 1e-12
        err = (p.grad - q.grad).abs().max().item()
        assert err <= 5e-5 * scale + 1e-9, (name, err, scale)
        # after one Adam step: elements whose gradient is ~1e-8 (Adam's eps) take a step lr*g/(|g|+eps) that is
        # sensitive to round-off in g, so the bound is a fraction of lr = 2e-4, not of the weight
        np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().cpu().numpy(), atol=5e-5, rtol=1e-5, err_msg=name)


def test_fused_update_full_iteration_tracks_eager():
    """5 epochs x 4 minibatches of 40960 rows (the benchmark's learner workload): parameters after the whole
    update stay within 2e-4 of the eager path (20 Adam steps of accumulated fp32 round-off)."""
    N, T = 4096, 40
    ac_e, alg_e = _make(N, T, 5, 4, fused=False)
    ac_f, alg_f = _make(N, T, 5, 4, fused=True)
    _fill(alg_e, N, T, seed=3)
    for name in ("observations", "actions", "values", "actions_log_prob", "mu", "sigma", "returns", "advantages"):
        getattr(alg_f.storage, name).copy_(getattr(alg_e.storage, name))
    alg_f.storage.step = T
    perm = torch.randperm(N * T, device="cuda")
    with mock.patch("torch.randperm", lambda n, **k: perm):
        out_e = alg_e.update()
        out_f = alg_f.update()
    np.testing.assert_allclose(out_f, out_e, rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(gp.param_digest(ac_f)[:, :2], gp.param_digest(ac_e)[:, :2], rtol=2e-4, atol=2e-4)
    assert alg_f.counter == alg_e.counter == 3501


@pytest.mark.parametrize("batch,rows", [(64, 64), (700, 333), (31, 31), (5, 1), (40960, 40960), (163840, 40960)])
def test_fused_history_encoder_gradients_match_autograd(batch, rows):
    """wbc_hist_train_grad (one launch: history-encoder forward, loss = mean ||priv - hist||_2, backward, weight gradients;
    csrc/wbc_hist_train_kernel.hip) and wbc_priv_latent against PyTorch autograd over the module path that
    tests/test_ppo_parity.py pins to the reference (PPO:265-291, AC:39-84), on a gathered subset of the stored rows."""
    import ctypes as C
    from wbc_amd.native import check, lib
    torch.manual_seed(3)
    ac = ActorCritic(76, 76, 18, **gp.POLICY_KW).to("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(batch + rows)
    obs = torch.randn(batch, 860, generator=g, device="cuda")
    idx = torch.randperm(batch, generator=g, device="cuda")[:rows].contiguous()
    he, pe = ac.actor.history_encoder, ac.actor.priv_encoder
    hp = [he.encoder[0].weight, he.encoder[0].bias, he.conv_layers[0].weight, he.conv_layers[0].bias,
          he.conv_layers[2].weight, he.conv_layers[2].bias, he.linear_output[0].weight, he.linear_output[0].bias]
    assert [id(p) for p in hp] == [id(p) for p in he.parameters()]           # flat gradient layout = parameters() order
    pp = [pe[0].weight, pe[0].bias, pe[2].weight, pe[2].bias]
    L = lib()
    stream = torch.cuda.current_stream().cuda_stream
    priv = torch.empty(batch, 20, device="cuda")
    check(L.wbc_priv_latent((C.c_void_p * 4)(*[p.data_ptr() for p in pp]), obs.data_ptr(), priv.data_ptr(), batch, stream))
    with torch.no_grad():
        priv_ref = ac.actor.infer_priv_latent(obs)
    np.testing.assert_allclose(priv.cpu().numpy(), priv_ref.cpu().numpy(), atol=2e-6, rtol=2e-5)
    ng = L.wbc_hist_train_grad_floats()
    grad = torch.zeros(ng, device="cuda")
    ws = torch.empty(L.wbc_hist_train_workspace_floats(), device="cuda")
    check(L.wbc_hist_train_grad((C.c_void_p * 8)(*[p.data_ptr() for p in hp]), obs.data_ptr(), priv_ref.data_ptr(), idx.data_ptr(), rows,
                                ws.data_ptr(), grad.data_ptr(), stream))
    obs_b = obs[idx]
    hist = he(obs_b[:, -760:].view(-1, 10, 76))
    loss = (priv_ref[idx].detach() - hist).norm(p=2, dim=1).mean()
    loss.backward()
    torch.cuda.synchronize()
    assert abs(grad[ng - 1].item() / rows - loss.item()) <= 2e-5 * max(1.0, abs(loss.item()))
    flat = torch.cat([p.grad.reshape(-1) for p in hp])
    off = 0
    for p in hp:
        n = p.numel()
        a, b = grad[off:off + n], flat[off:off + n]
        scale = b.abs().max().item() + 1e-12
        err = (a - b).abs().max().item()
        assert err <= 1e-4 * scale + 1e-9, (tuple(p.shape), err, scale)
        off += n
    # deterministic: a second launch gives the same bits
    grad2 = torch.zeros(ng, device="cuda")
    check(L.wbc_hist_train_grad((C.c_void_p * 8)(*[p.data_ptr() for p in hp]), obs.data_ptr(), priv_ref.data_ptr(), idx.data_ptr(), rows,
                                ws.data_ptr(), grad2.data_ptr(), stream))
    torch.cuda.synchronize()
    assert torch.equal(grad, grad2)


@pytest.mark.parametrize("B", [40960, 1000])
def test_fused_minibatch_is_bitwise_reproducible(B):
    """wbc_ppo_minibatch_grad has no atomics and a fixed reduction order: two calls on the same inputs (into workspaces filled with
    NaN: nothing uninitialised is read) give the same gradient bit for bit, at the bench's minibatch size and at a ragged one."""
    from wbc_amd.native import check, lib
    L = lib()
    torch.manual_seed(0)
    ac = ActorCritic(76, 76, 18, **gp.POLICY_KW).cuda()
    TN, dev = 2 * B, "cuda"
    obs = torch.randn(TN, 860, device=dev); actions = torch.randn(TN, 18, device=dev); values = torch.randn(TN, 2, device=dev)
    adv = torch.randn(TN, 2, device=dev); ret = torch.randn(TN, 2, device=dev); logp = -torch.rand(TN, 2, device=dev) * 20
    hist = torch.randn(TN, 20, device=dev); idx = torch.randperm(TN, device=dev)[:B].contiguous()
    table = ac.fused_param_table()
    stream = torch.cuda.current_stream().cuda_stream
    grads = []
    for _ in range(2):
        ws = torch.full((L.wbc_ppo_workspace_floats(B),), float("nan"), device=dev)
        grad = torch.zeros(L.wbc_ppo_grad_floats(), device=dev)
        check(L.wbc_ppo_minibatch_grad(table, obs.data_ptr(), actions.data_ptr(), values.data_ptr(), adv.data_ptr(), ret.data_ptr(), logp.data_ptr(),
                                       hist.data_ptr(), idx.data_ptr(), B, 0.2, 1.0, 0.5, 0.1, 1, ws.data_ptr(), grad.data_ptr(), None, stream), "grad")
        torch.cuda.synchronize()
        grads.append(grad.clone())
    assert torch.isfinite(grads[0]).all()
    assert torch.equal(grads[0], grads[1])


@pytest.mark.parametrize("B", [4096, 1000])
def test_adam_keeps_the_weight_streams_current(B):
    """wbc_ppo_clip_adam_packed writes every updated parameter to its copies in the chain kernel's weight streams, so
    wbc_ppo_minibatch_grad_packed (no pack launch) must see exactly what a fresh pack would: three optimiser steps along both routes
    from the same start give the same gradients and the same parameters, bit for bit. The entry weight of every layer moves in every
    step (Adam's first steps are +-lr), so a stale or misplaced copy cannot hide."""
    import copy
    import ctypes as C
    from wbc_amd.native import check, lib
    L = lib()
    torch.manual_seed(0)
    ac0 = ActorCritic(76, 76, 18, **gp.POLICY_KW).cuda()
    TN, dev = 3 * B, "cuda"
    obs = torch.randn(TN, 860, device=dev); actions = torch.randn(TN, 18, device=dev); values = torch.randn(TN, 2, device=dev)
    adv = torch.randn(TN, 2, device=dev); ret = torch.randn(TN, 2, device=dev); logp = -torch.rand(TN, 2, device=dev) * 20
    hist = torch.randn(TN, 20, device=dev)
    idx = torch.randperm(TN, device=dev).contiguous()
    stream = torch.cuda.current_stream().cuda_stream
    ng = L.wbc_ppo_grad_floats()
    results = []
    for route in ("fresh pack every minibatch", "streams kept by Adam"):
        ac = copy.deepcopy(ac0)
        table = ac.fused_param_table()
        nparam = sum(p.numel() for p in ac.fused_params())
        ws = torch.full((L.wbc_ppo_workspace_floats(B),), float("nan"), device=dev)
        grad = torch.zeros(ng, device=dev); m = torch.zeros(nparam, device=dev); v = torch.zeros(nparam, device=dev)
        aws = torch.empty(int(L.wbc_ppo_clip_adam_workspace_floats()), device=dev)
        sq = ws.data_ptr() + 4 * int(L.wbc_ppo_sq_partials_offset(B))
        grads = []
        for k in range(3):
            f = L.wbc_ppo_minibatch_grad if (route.startswith("fresh") or k == 0) else L.wbc_ppo_minibatch_grad_packed
            check(f(table, obs.data_ptr(), actions.data_ptr(), values.data_ptr(), adv.data_ptr(), ret.data_ptr(), logp.data_ptr(), hist.data_ptr(),
                    idx[k * B:(k + 1) * B].data_ptr(), B, 0.2, 1.0, 0.5, 0.1, 1, ws.data_ptr(), grad.data_ptr(), None, stream), "grad")
            grads.append(grad.clone())
            t = k + 1.0
            args = (table, grad.data_ptr(), m.data_ptr(), v.data_ptr(), 1.0, 0.9, 0.999, 1e-8, 1e-3 / (1.0 - 0.9 ** t), (1.0 - 0.999 ** t) ** 0.5, 1.0, sq, aws.data_ptr())
            if route.startswith("fresh"):
                check(L.wbc_ppo_clip_adam(*args, stream), "adam")
            else:
                check(L.wbc_ppo_clip_adam_packed(*args, ws.data_ptr(), B, stream), "adam_packed")
        torch.cuda.synchronize()
        results.append((grads, [p.detach().clone() for p in ac.fused_params()]))
    (g0, p0), (g1, p1) = results
    assert all(torch.isfinite(g).all() for g in g0)
    assert not torch.equal(g0[0], g0[1])                     # the parameters did move between the minibatches
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)


def test_packed_call_with_stale_streams_packs_afresh():
    """wbc_ppo_minibatch_grad_packed trusts the weight streams of a workspace only while the library's record (workspace, B,
    params[0]) says they are current: after a plain wbc_ppo_clip_adam step (which moves the parameters without touching the
    streams), on a workspace nobody packed, or for another parameter set, it packs afresh -- the gradients equal those of
    wbc_ppo_minibatch_grad bit for bit instead of being computed against stale weights."""
    import copy
    from wbc_amd.native import check, lib
    L = lib()
    B, dev = 1024, "cuda"
    torch.manual_seed(0)
    ac = ActorCritic(76, 76, 18, **gp.POLICY_KW).cuda()
    TN = 2 * B
    obs = torch.randn(TN, 860, device=dev); actions = torch.randn(TN, 18, device=dev); values = torch.randn(TN, 2, device=dev)
    adv = torch.randn(TN, 2, device=dev); ret = torch.randn(TN, 2, device=dev); logp = -torch.rand(TN, 2, device=dev) * 20
    hist = torch.randn(TN, 20, device=dev)
    idx = torch.randperm(TN, device=dev).contiguous()
    stream = torch.cuda.current_stream().cuda_stream
    ng = L.wbc_ppo_grad_floats()
    nparam = sum(p.numel() for p in ac.fused_params())

    def grad_of(f, model, ws, k):
        g = torch.zeros(ng, device=dev)
        check(f(model.fused_param_table(), obs.data_ptr(), actions.data_ptr(), values.data_ptr(), adv.data_ptr(), ret.data_ptr(), logp.data_ptr(), hist.data_ptr(),
                idx[k * B:(k + 1) * B].data_ptr(), B, 0.2, 1.0, 0.5, 0.1, 1, ws.data_ptr(), g.data_ptr(), None, stream), "grad")
        torch.cuda.synchronize()
        return g
    ws = torch.zeros(L.wbc_ppo_workspace_floats(B), device=dev)
    grad_of(L.wbc_ppo_minibatch_grad, ac, ws, 0)                               # packs; the record says current
    g = torch.randn(ng, device=dev) * 1e-2
    m = torch.zeros(nparam, device=dev); v = torch.zeros(nparam, device=dev)
    aws = torch.empty(int(L.wbc_ppo_clip_adam_workspace_floats()), device=dev)
    check(L.wbc_ppo_clip_adam(ac.fused_param_table(), g.data_ptr(), m.data_ptr(), v.data_ptr(), 1.0, 0.9, 0.999, 1e-8, 1e-2, 0.03, 1.0, None, aws.data_ptr(), stream), "adam")
    want = grad_of(L.wbc_ppo_minibatch_grad, copy.deepcopy(ac), torch.zeros_like(ws), 1)
    got = grad_of(L.wbc_ppo_minibatch_grad_packed, ac, ws, 1)                   # streams are one Adam step behind: must not be used
    assert torch.equal(got, want)
    got = grad_of(L.wbc_ppo_minibatch_grad_packed, ac, torch.zeros_like(ws), 1)  # a workspace nobody packed
    assert torch.equal(got, want)
    check(L.wbc_ppo_pack_invalidate(ws.data_ptr()), "invalidate")
    assert torch.equal(grad_of(L.wbc_ppo_minibatch_grad_packed, ac, ws, 1), want)
