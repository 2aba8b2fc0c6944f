"""world_size-2 tests of the sharded learner's FUSED GPU path (wbc_ppo_minibatch_grad -> ONE flat gradient all-reduce ->
wbc_ppo_clip_adam with 1/world_size folded in; the fused DAgger path; the pooled advantage statistics between wbc_gae_compute
and wbc_gae_normalize; OnPolicyRunner(dist_group=...)), against a single-GPU learner over the union of the shards with the
matching minibatches (tests/test_distributed_cpu.py is the eager counterpart on the CPU).

Two rigs: (a) two processes on cuda:0 with a gloo group whose collectives are staged through the host (wbc_amd.collectives)
-- runs on the 1-GPU test box, same learner code, world_size really 2; (b) one rank per GPU over RCCL (skipped where fewer
than two devices are visible)."""
import os
import unittest.mock as mock

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_procedure as gp
from test_distributed_cpu import _data, _free_port, _local_perm
from wbc_amd.rsl_rl.algorithms import PPO
from wbc_amd.rsl_rl.modules import ActorCritic

pytestmark = pytest.mark.gpu
N, T, W = 32, 8, 2


def _make(n_envs, device, dist_group=None):
    torch.manual_seed(1)
    ac = ActorCritic(76, 76, 18, **gp.POLICY_KW)
    kw = dict(gp.ALG_KW)
    kw["num_mini_batches"] = 2
    kw["num_learning_epochs"] = 2
    alg = PPO(ac, device=device, dist_group=dist_group, **kw)
    alg.counter = 3500
    alg.init_storage(n_envs, T, [860], [None], [18])
    return ac, alg


def _fill(alg, d, sl, device):
    st, ac = alg.storage, alg.actor_critic
    with torch.inference_mode():
        for t in range(T):
            o = d["obs"][t, sl].to(device).contiguous()
            a = d["act"][t, sl].to(device)
            ac.update_distribution(o, False)
            st.observations[t].copy_(o)
            st.actions[t].copy_(a)
            st.values[t].copy_(ac.evaluate(o))
            st.actions_log_prob[t].copy_(ac.get_actions_log_prob(a))
            st.mu[t].copy_(ac.action_mean)
            st.sigma[t].copy_(ac.action_std)
        st.rewards.copy_(d["rew"][:, sl])
        st.dones.copy_(d["dones"][:, sl])
        st.step = T
        alg.compute_returns(d["obs"][T, sl].to(device).contiguous())


def _init(rank, port, backend):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev_index = rank if backend == "nccl" else 0          # gloo rig: both ranks share cuda:0
    torch.cuda.set_device(dev_index)
    dev = f"cuda:{dev_index}"
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=W, device_id=torch.device(dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=W)
    return dev


def _worker(rank, port, q, backend="nccl"):
    dev = _init(rank, port, backend)
    ac, alg = _make(N // W, dev, dist.group.WORLD)
    alg.warm_up_collectives()
    sl = slice(rank * (N // W), (rank + 1) * (N // W))
    d = _data()
    _fill(alg, d, sl, dev)
    assert alg._fused_update_supported() and alg._fused_dagger_supported()
    adv = alg.storage.advantages.cpu().clone()
    perm = _local_perm(rank)
    with mock.patch("torch.randperm", lambda n, **kw: perm.to(kw.get("device", "cpu"))):
        stats = alg.update()
        _fill(alg, d, sl, dev)
        dag = alg.update_dagger()
    torch.cuda.synchronize()
    q.put((rank, adv.numpy(), gp.param_digest(ac), [float(x) for x in stats], dag))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(target, *args):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, port, q) + args) for r in range(W)]
    for p in procs:
        p.start()
    try:
        results = sorted([q.get(timeout=600) for _ in range(W)], key=lambda r: r[0])
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    for p in procs:
        assert p.exitcode == 0
    return results


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the driver's multi-GPU node)")
def test_two_rank_rccl_fused_learner_matches_single_gpu():
    _check_fused_learner(_spawn(_worker, "nccl"))


def test_two_rank_same_device_fused_learner_matches_single_gpu():
    """world_size 2 on ONE GPU: both ranks on cuda:0, gloo collectives staged through the host."""
    _check_fused_learner(_spawn(_worker, "gloo"))


def _check_fused_learner(results):
    dev = "cuda:0"
    ac, alg = _make(N, dev)
    d = _data()
    _fill(alg, d, slice(0, N), dev)
    adv_full = alg.storage.advantages.cpu().numpy()
    for rank, adv, _, _, _ in results:
        np.testing.assert_allclose(adv, adv_full[:, rank * (N // W):(rank + 1) * (N // W)], atol=5e-6)     # pooled normalisation
    nl = N // W
    mb_local = T * nl // 2
    parts = []
    for i in range(2):
        for rank in range(W):
            loc = _local_perm(rank)[i * mb_local:(i + 1) * mb_local]
            parts.append((loc // nl) * N + rank * nl + (loc % nl))
    perm = torch.cat(parts)
    with mock.patch("torch.randperm", lambda n, **kw: perm.to(kw.get("device", "cpu"))):
        stats_full = alg.update()
        _fill(alg, d, slice(0, N), dev)
        dag_full = alg.update_dagger()
    d_full = gp.param_digest(ac)
    np.testing.assert_array_equal(results[0][2], results[1][2])                     # replicas stay bit-identical
    np.testing.assert_allclose(results[0][2][:, :2], d_full[:, :2], rtol=1e-4, atol=1e-4)
    mean_stats = np.mean([r[3] for r in results], axis=0)
    np.testing.assert_allclose(mean_stats[:2], [float(x) for x in stats_full][:2], rtol=1e-3, atol=1e-6)
    assert abs(np.mean([r[4] for r in results]) - dag_full) < 1e-3 * max(1.0, abs(dag_full))


def _runner_worker(rank, port, q, backend="gloo"):
    """OnPolicyRunner(dist_group=...) on an env shard: parameter broadcast, 3 iterations (DAgger, PPO, PPO) with every
    collective of the sharded loop on device tensors."""
    dev = _init(rank, port, backend)
    from wbc_amd.config import WidowGo1RoughCfg, WidowGo1RoughCfgPPO, class_to_dict
    from wbc_amd.envs import WidowGo1
    from wbc_amd.rsl_rl.runners import OnPolicyRunner
    cfg = WidowGo1RoughCfg()
    cfg.env.num_envs = 128
    cfg.terrain.mesh_type = "plane"
    torch.manual_seed(3 + rank)                         # DIFFERENT initial replicas: the broadcast must make them equal
    env = WidowGo1(cfg, sim_device=dev, seed=11 + rank)
    train = class_to_dict(WidowGo1RoughCfgPPO())
    train["runner"]["num_steps_per_env"] = 8
    runner = OnPolicyRunner(env, train, log_dir=None, device=dev, dist_group=dist.group.WORLD)
    torch.manual_seed(100 + rank)                       # per-rank exploration noise / permutations
    runner.learn(3)
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().flatten() for p in runner.alg.actor_critic.parameters()]).cpu().numpy()
    adam_steps = [float(s["step"]) for s in runner.alg.optimizer.state.values()]
    q.put((rank, flat, [h["mean_value_loss"] for h in runner.history], adam_steps[0],
           bool(runner.alg._fused is not None), bool(runner.alg.__dict__.get("_fused_hist") is not None)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_same_device_runner_replicas_stay_identical():
    res = _spawn(_runner_worker, "gloo")
    (_, p0, vl0, st0, f0, h0), (_, p1, vl1, st1, f1, h1) = res
    assert f0 and f1 and h0 and h1                      # both the fused PPO update and the fused DAgger update ran on both ranks
    np.testing.assert_array_equal(p0, p1)               # identical steps from identical reduced gradients
    assert np.isfinite(p0).all() and np.isfinite(vl0).all() and np.isfinite(vl1).all()
    assert st0 == st1 == 40.0                           # 2 PPO updates x 5 epochs x 4 minibatches
    assert vl0 != vl1                                   # the shards really differ (local losses), only the gradients are pooled
