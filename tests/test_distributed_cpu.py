"""world_size-2 gloo test of the sharded learner: env shards per rank, pooled
advantage statistics (collective 2), one flat gradient all-reduce per minibatch (collective 1).
Checked against a single-process learner over the union of the shards with the matching minibatches."""
import os
import socket
import unittest.mock as mock

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_procedure as gp
from wbc_amd.rsl_rl.algorithms import PPO
from wbc_amd.rsl_rl.modules import ActorCritic

N, T, W = 32, 8, 2          # global envs, steps, world size (the module-level default: world 2; the world-8 test passes its own)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(n_envs, dist_group=None, **over):
    torch.manual_seed(1)
    ac = ActorCritic(76, 76, 18, **gp.POLICY_KW)
    kw = dict(gp.ALG_KW)
    kw["num_mini_batches"] = 2
    kw["num_learning_epochs"] = 2
    kw.update(over)
    alg = PPO(ac, device="cpu", dist_group=dist_group, **kw)
    alg.counter = 3500
    alg.init_storage(n_envs, T, [860], [None], [18])
    return ac, alg


def _data(n=N):
    g = torch.Generator().manual_seed(7)
    return dict(obs=torch.randn(T + 1, n, 860, generator=g), act=torch.randn(T, n, 18, generator=g),
                rew=0.1 * torch.randn(T, n, 2, generator=g), dones=(torch.rand(T, n, 1, generator=g) < 0.1).to(torch.uint8))


def _fill(alg, d, sl):
    """Storage contents computed with the (identical) initial policy on the env slice `sl`."""
    st, ac = alg.storage, alg.actor_critic
    with torch.inference_mode():
        for t in range(T):
            o = d["obs"][t, sl]
            ac.update_distribution(o, False)
            st.observations[t].copy_(o)
            st.actions[t].copy_(d["act"][t, sl])
            st.values[t].copy_(ac.evaluate(o))
            st.actions_log_prob[t].copy_(ac.get_actions_log_prob(d["act"][t, sl]))
            st.mu[t].copy_(ac.action_mean)
            st.sigma[t].copy_(ac.action_std)
        st.rewards.copy_(d["rew"][:, sl])
        st.dones.copy_(d["dones"][:, sl])
        st.step = T
        alg.compute_returns(d["obs"][T, sl])


def _local_perm(rank, n=N, w=W):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randperm(T * (n // w), generator=g)


def _worker(rank, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    torch.set_num_threads(1)
    ac, alg = _make(N // W, dist.group.WORLD)
    sl = slice(rank * (N // W), (rank + 1) * (N // W))
    _fill(alg, _data(), sl)
    adv = alg.storage.advantages.clone()
    perm = _local_perm(rank)
    with mock.patch("torch.randperm", lambda n, **kw: perm):
        stats = alg.update()
    q.put((rank, adv.numpy(), gp.param_digest(ac), [float(x) for x in stats]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_learner_matches_single_process():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(W)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in range(W)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process over the union
    ac, alg = _make(N)
    _fill(alg, _data(), slice(0, N))
    adv_full = alg.storage.advantages.numpy()
    for rank, adv, _, _ in results:
        np.testing.assert_allclose(adv, adv_full[:, rank * (N // W):(rank + 1) * (N // W)], atol=2e-6)     # pooled normalisation
    # global minibatch i = union of the ranks' local minibatches i (flat index = t * n_envs + env)
    nl = N // W
    mb_local = T * nl // 2
    parts = []
    for i in range(2):
        for rank in range(W):
            loc = _local_perm(rank)[i * mb_local:(i + 1) * mb_local]
            parts.append((loc // nl) * N + rank * nl + (loc % nl))
    perm = torch.cat(parts)
    with mock.patch("torch.randperm", lambda n, **kw: perm):
        stats_full = alg.update()
    d_full = gp.param_digest(ac)
    np.testing.assert_allclose(results[0][2], results[1][2], rtol=0, atol=0)       # replicas stay bit-identical
    np.testing.assert_allclose(results[0][2][:, :2], d_full[:, :2], rtol=2e-5, atol=2e-5)
    mean_stats = np.mean([r[3] for r in results], axis=0)
    np.testing.assert_allclose(mean_stats[:2], [float(x) for x in stats_full][:2], rtol=1e-4, atol=1e-6)


# ---- world size 8 (what `bench.py --gpus 8` runs), including the two collectives the world-2 test does not reach: the KL all-reduce
# ---- of the adaptive learning-rate schedule (ppo.py: schedule == "adaptive") and update_dagger's gradient all-reduce
N8, W8 = 64, 8
_ADAPTIVE = dict(schedule="adaptive", desired_kl=0.01)


def _global_perm(n, w, nmb=2):
    """The single-process minibatch order that matches the ranks' local ones: global minibatch i = the union of every rank's local
    minibatch i (flat index = t * n_envs + env)."""
    nl = n // w
    mb_local = T * nl // nmb
    parts = []
    for i in range(nmb):
        for rank in range(w):
            loc = _local_perm(rank, n, w)[i * mb_local:(i + 1) * mb_local]
            parts.append((loc // nl) * n + rank * nl + (loc % nl))
    return torch.cat(parts)


def _worker8(rank, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=W8)
        torch.set_num_threads(1)
        nl = N8 // W8
        ac, alg = _make(nl, dist.group.WORLD, **_ADAPTIVE)
        sl = slice(rank * nl, (rank + 1) * nl)
        d = _data(N8)
        _fill(alg, d, sl)
        perm = _local_perm(rank, N8, W8)
        with mock.patch("torch.randperm", lambda n, **kw: perm):
            stats = alg.update()
        lr_after = alg.learning_rate
        _fill(alg, d, sl)                                   # a second rollout's worth of storage for the DAgger update
        with mock.patch("torch.randperm", lambda n, **kw: perm):
            dag = alg.update_dagger()
        q.put((rank, gp.param_digest(ac), [float(x) for x in stats], lr_after, float(dag), None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:      # noqa: BLE001  (the parent prints every rank's failure instead of timing out on the queue)
        import traceback
        q.put((rank, None, None, None, None, traceback.format_exc()))
        raise e


def test_eight_rank_learner_matches_single_process_with_adaptive_kl_and_dagger():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, port, q)) for r in range(W8)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=600) for _ in range(W8)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
    errs = [r[5] for r in results if r[5]]
    assert not errs, "\n".join(errs)
    assert all(p.exitcode == 0 for p in procs)
    # one process over the union of the shards, the matching global minibatches
    ac, alg = _make(N8, **_ADAPTIVE)
    d = _data(N8)
    _fill(alg, d, slice(0, N8))
    perm = _global_perm(N8, W8)
    with mock.patch("torch.randperm", lambda n, **kw: perm):
        stats_full = alg.update()
    lr_full = alg.learning_rate
    _fill(alg, d, slice(0, N8))
    with mock.patch("torch.randperm", lambda n, **kw: perm):
        dag_full = alg.update_dagger()
    d_full = gp.param_digest(ac)
    for r in results[1:]:
        np.testing.assert_array_equal(r[1], results[0][1])                          # the eight replicas stay bit-identical
        assert r[3] == results[0][3]                                                # ... and agree on the adapted learning rate
    assert lr_full != gp.ALG_KW["learning_rate"], "the adaptive branch must have fired"
    np.testing.assert_allclose(results[0][3], lr_full, rtol=1e-12)                  # the pooled KL moved the rate as the single process's did
    np.testing.assert_allclose(results[0][1][:, :2], d_full[:, :2], rtol=5e-5, atol=5e-5)
    mean_stats = np.mean([r[2] for r in results], axis=0)
    np.testing.assert_allclose(mean_stats[:2], [float(x) for x in stats_full][:2], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(np.mean([r[4] for r in results]), dag_full, rtol=2e-4)
