#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the widowGo1 rollout + PPO hot path on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one PPO iteration of the hot path: T=40 fused rollout steps over the env batch
(policy inference + the HIP step kernel: 4 physics substeps + post-physics each), GAE, and
PPO.update() (5 epochs x 4 minibatches; every 20th iteration is the DAgger update instead, as in
the reference's learn loop, on_policy_runner.py:129,166-169). Workload = BASELINE.json configs[1]:
widowGo1, flat terrain, 4096 envs per GPU, fp32, domain randomisation as shipped. The metric is the
reference's own fps definition (on_policy_runner.py:206), aggregated over ranks (weak scaling: 4096
envs per GPU, one gradient all-reduce per minibatch and one 3-scalar advantage-statistics all-reduce
per iteration over RCCL).

Rank 0 prints ONE JSON line with the contract fields plus `roofline` (the fused step kernel: launch
durations measured live with HIP events on torch's current stream over the timed region) and, at
N=1, `cpu_baseline` (the CPU oracle of the same loop, timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd"))

ENVS_PER_GPU = 4096
T_STEPS = 40
ALGO_BYTES_PER_ENV_STEP = 11.1e3      # fused sim step, SURVEY.md section 8(d): 947 f32 read + 1822 f32 written
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32-input MFMA = the fp32 vector rate (155 measured)
# one PPO minibatch row through the teacher networks (actor_critic.py layer sizes): forward 2*in*out per layer, the input
# gradient of every layer but the two that read the observation, the weight gradient of every layer
_LAYERS = [(24, 64), (64, 20), (96, 128), (128, 128), (128, 128), (128, 12), (128, 128), (128, 128), (128, 6),
           (100, 128), (128, 128), (128, 128), (128, 1), (128, 128), (128, 128), (128, 1)]
UPDATE_FLOPS_PER_ROW = sum(2 * i * o * (3 if k not in (0, 9) else 2) for k, (i, o) in enumerate(_LAYERS))


CPU_BASELINE_ENVS = 4096             # bounded sample: one full iteration of the workload (40 s guard on the rollout part for slow hosts)
CPU_BASELINE_THREADS = 8


def cpu_baseline(num_envs=CPU_BASELINE_ENVS, T=T_STEPS):
    """A bounded sample of the same workload on the host: the C oracle steps `num_envs` envs T times
    (scalar, 1 core) with CPU policy inference in between, then the functional PPO oracle does GAE +
    one update() (torch CPU, CPU_BASELINE_THREADS threads: more threads only add overhead at these
    layer sizes)."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as ora
    import ppo_oracle as po
    from wbc_amd import abi
    from wbc_amd.config import WidowGo1RoughCfg, WidowGo1RoughCfgPPO, class_to_dict
    from wbc_amd.rsl_rl.modules import ActorCritic
    ora.build()
    nthreads = min(CPU_BASELINE_THREADS, os.cpu_count() or 1)
    torch.set_num_threads(nthreads)
    m = abi.load_default_model()
    cfg = WidowGo1RoughCfg()
    wm, tc = abi.fill_model(m), abi.fill_task_cfg(cfg, m)
    sim = ora.OracleSim(wm, tc, num_envs, seed=1, precision="f64")
    sim.set_curriculum(ora.default_curriculum(cfg, 1))
    rng = np.random.default_rng(0)
    n = num_envs
    tt = rng.uniform(1, 3, n) / 0.02
    sim.set_env_params(rng.uniform(-0.5, 3, n), rng.uniform(-0.5, 2.5, n), rng.uniform(-.15, .15, (n, 3)), rng.uniform(0, .1, n),
                       rng.uniform(.7, 1.3, (n, 18)), np.stack([rng.uniform(-3.75, -3, n), rng.uniform(-115, 115, n), np.zeros(n)], 1),
                       rng.uniform(.1, .3, n), tt, tt + rng.uniform(.5, 2, n) / 0.02, m)
    sim.reset_all()
    sim.step(np.zeros((n, 18)))
    pol = class_to_dict(WidowGo1RoughCfgPPO())["policy"]
    torch.manual_seed(1)
    ac = ActorCritic(76, 76, 18, **pol, num_priv=24, num_hist=10, num_prop=76)
    sd = {k: v.detach().clone() for k, v in ac.state_dict().items()}
    obs_l, act_l, val_l, lp_l, rew_l, done_l = [], [], [], [], [], []
    t0 = time.time()
    obs = torch.from_numpy(sim.get("OBS_BUF")).float()
    with torch.no_grad():
        for _ in range(T):
            if time.time() - t0 > 40.0:          # hard bound on a slow host: use the steps done so far
                break
            mean = po.actor_mean(sd, obs)
            std = mean * 0 + sd["std"]
            a = torch.normal(mean, std)
            obs_l.append(obs); act_l.append(a); val_l.append(po.critic_value(sd, obs)); lp_l.append(po.log_prob2(mean, std, a))
            sim.step(a.numpy().astype(np.float64))
            obs = torch.from_numpy(sim.get("OBS_BUF")).float()
            rew = torch.from_numpy(np.stack([sim.get("REW_BUF"), sim.get("ARM_REW_BUF")], -1)).float()
            tout = torch.from_numpy(sim.get("TIME_OUT_BUF")).float()
            rew_l.append(rew + 0.99 * val_l[-1] * tout[:, None])
            done_l.append(torch.from_numpy(sim.get("RESET_BUF")).to(torch.uint8)[:, None])
        last_v = po.critic_value(sd, obs)
    t_roll = time.time() - t0
    rewards, values, dones = torch.stack(rew_l), torch.stack(val_l), torch.stack(done_l)
    returns, adv = po.gae(rewards, values, dones, last_v, 0.99, 0.95)
    learner = po.PPOOracle(sd, min_std=torch.tensor(class_to_dict(WidowGo1RoughCfgPPO())["algorithm"]["min_policy_std"]))
    f = lambda x: x.flatten(0, 1)   # noqa: E731
    learner.update(f(torch.stack(obs_l)), f(torch.stack(act_l)), f(values), f(adv), f(returns), f(torch.stack(lp_l)), beta=1.0, roa_coef=0.0)
    t_all = time.time() - t0
    T = len(obs_l)
    return {"value": n * T / t_all, "unit": "env-steps/s", "cores": nthreads, "kind": "port",
            "sample": f"{n} envs x {T} steps of the same workload (1/{ENVS_PER_GPU // n} of one iteration): C oracle sim on 1 core "
                      f"incl. torch-CPU policy inference {t_roll:.1f}s, then torch-CPU GAE + PPO.update() 5x4 minibatches "
                      f"on {nthreads} threads {t_all - t_roll:.1f}s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="create the RCCL process group even for one rank (exercises the multi-GPU code path on one GPU)")
    ap.add_argument("--terrain", choices=["plane", "trimesh"], default="plane",
                    help="plane = BASELINE.json configs[1] (the bench line); trimesh = the shipped fractal-Perlin terrain (configs[2])")
    ap.add_argument("--terrain-curriculum", action="store_true",
                    help="with --terrain trimesh: terrain.curriculum=True (the base class's terrain levels, LR:421-441; configs[2] names it)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"launch with torch.distributed.run --nproc-per-node {args.gpus} (WORLD_SIZE={world})"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the hot path only exists as HIP kernels")
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    group = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        group = dist.group.WORLD

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if use_dist:
        dist.barrier()
    from wbc_amd.config import WidowGo1RoughCfg, WidowGo1RoughCfgPPO, class_to_dict
    from wbc_amd.envs import WidowGo1
    from wbc_amd.rsl_rl.runners import OnPolicyRunner

    cfg = WidowGo1RoughCfg()
    cfg.env.num_envs = args.envs_per_gpu
    if args.terrain == "plane":
        cfg.terrain.mesh_type = "plane"               # BASELINE.json configs[1]: flat terrain (the shipped default is the Perlin trimesh)
    elif args.terrain_curriculum:
        cfg.terrain.curriculum = True
    train_cfg = WidowGo1RoughCfgPPO()
    torch.manual_seed(train_cfg.seed)                 # identical replicas; env RNG differs per rank
    env = WidowGo1(cfg, sim_device=device, seed=train_cfg.seed + rank)
    train = class_to_dict(train_cfg)
    runner = OnPolicyRunner(env, train, log_dir=None, device=device, dist_group=group)
    torch.manual_seed(train_cfg.seed + 1000 * rank)   # replicas are identical (broadcast at construction); exploration noise is per rank
    T = runner.num_steps_per_env

    # HIP-event timing of every fused-step launch in the timed region (torch's current stream is the
    # stream the kernel is launched on)
    events = []
    raw_step = env.sim.step
    timing_on = {"v": False}

    nstep = {"n": 0}

    def timed_step(*a, **kw):                        # (whatever signature WbcSim.step has)
        nstep["n"] += 1
        if timing_on["v"] and nstep["n"] % 4 == 0:       # every 4th launch: an event pair around each one cost 0.34 ms per iteration
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            raw_step(*a, **kw)
            e1.record()
            events.append((e0, e1))
        else:
            raw_step(*a, **kw)
    env.sim.step = timed_step

    # The fused PPO minibatch step (weight pack + ppo_fwd_bwd16 + ppo_wgrad + reducers: one C-ABI call) is timed AFTER the
    # timed region by replaying the last call 20 times between two events: an event pair around every call inside the
    # region cost 4.5 ms per iteration (measured), i.e. it would have changed the number being reported.
    from wbc_amd.native import lib as _lib
    _L = _lib()
    raw_grad = _L.wbc_ppo_minibatch_grad
    last_grad_args = {}

    def remember_grad(*a):
        last_grad_args["a"] = a
        return raw_grad(*a)
    _L.wbc_ppo_minibatch_grad = remember_grad

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Two priming iterations before the W warm-up steps: iteration 0 (a DAgger update: rocBLAS heuristics, ~1 s) and the
    # first PPO update (first launches of every update kernel, the 0.6 GB workspace: ~110 ms instead of 13) are one-off
    # set-up costs that a small W would otherwise push into the timed region.
    runner.learn(2, init_at_random_ep_len=True)
    barrier()
    runner.learn(max(args.warmup, 0)) if args.warmup > 0 else None
    barrier()
    timing_on["v"] = rank == 0
    t0 = time.perf_counter()
    runner.learn(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    timing_on["v"] = False
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    total_env_steps = args.envs_per_gpu * world * T * args.steps
    value = total_env_steps / elapsed

    if rank == 0:
        hist = runner.history[-args.steps:]
        kern_ms = sum(a.elapsed_time(b) for a, b in events) / max(len(events), 1)
        algo_bytes = ALGO_BYTES_PER_ENV_STEP * args.envs_per_gpu
        achieved = algo_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "step_kernel_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec whole node, widowGo1 4096-env PPO",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": f"synthetic (random-init policy, seeded domain randomisation, {'flat' if args.terrain == 'plane' else 'fractal-Perlin trimesh'} terrain)",
            "config": {"workload": f"widowGo1 {'flat' if args.terrain == 'plane' else 'trimesh (Perlin)'} terrain, {args.envs_per_gpu} envs per GPU, PPO fp32 "
                                   f"(BASELINE.json configs[{1 if args.terrain == 'plane' else 2}]{', terrain-level curriculum on' if args.terrain_curriculum else ''}); T={T} steps/iteration, 5 epochs x 4 minibatches, "
                                   f"DAgger every 20th iteration", "envs_per_gpu": args.envs_per_gpu,
                       "global_envs": args.envs_per_gpu * world, "steps_per_env": T,
                       "parallelism": f"env-shard x{world}, 1 grad all-reduce/minibatch" if world > 1 else "single GPU",
                       "collection_ms": 1e3 * sum(h["collection_time"] for h in hist) / len(hist),
                       "learn_ms": 1e3 * sum(h["learn_time"] for h in hist) / len(hist)},
            "roofline": {"kernel": "wbc_step_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "launch_ms": kern_ms,
                         "algorithmic_bytes_per_launch": algo_bytes, "launches_timed": len(events)},
        }
        if "a" in last_grad_args:
            a = last_grad_args["a"]
            reps = 20
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                raw_grad(*a)
            e1.record()
            torch.cuda.synchronize()
            upd_ms = e0.elapsed_time(e1) / reps
            rows = int(a[9])
            flops = UPDATE_FLOPS_PER_ROW * rows
            tf = flops / (upd_ms * 1e-3) / 1e12
            upd_traffic = None
            upath = os.path.join(ROOT, "profiles", "ppo_update_traffic.json")
            if os.path.exists(upath) and rows == 40960:          # counters were collected at this minibatch size
                try:
                    upd_traffic = json.load(open(upath)).get("hbm_bytes_per_launch")
                except Exception:
                    upd_traffic = None
            out["roofline_update"] = {"kernel": "wbc_ppo_minibatch_grad = wbc_pack16 + ppo_fwd_bwd16 + ppo_wgrad + reducers", "bound": "mfma",
                                      "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
                                      "traffic": upd_traffic, "launch_ms": upd_ms, "algorithmic_flops_per_launch": flops, "rows_per_launch": rows,
                                      "launches_timed": reps, "timed": "after the timed region (replay of the last minibatch call)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
