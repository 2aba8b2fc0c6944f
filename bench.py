#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the widowGo1 rollout + PPO hot path on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one PPO iteration of the hot path: T=40 fused rollout steps over the env batch
(two launches each: the policy inference -- which also carries the previous step's episode statistics as a
side job -- and the HIP step kernel: 4 physics substeps with the URDF's collision set + post-physics), GAE, and
PPO.update() (5 epochs x 4 minibatches; every 20th iteration is the DAgger update instead, as in
the reference's learn loop, on_policy_runner.py:129,166-169). Workload = BASELINE.json configs[1]:
widowGo1, flat terrain, 4096 envs per GPU, fp32, domain randomisation as shipped. The metric is the
reference's own fps definition (on_policy_runner.py:206), aggregated over ranks (weak scaling: 4096
envs per GPU, one gradient all-reduce per minibatch and one 3-scalar advantage-statistics all-reduce
per iteration over RCCL). --log runs the loop as train.py does (log_dir set); --backend gloo --same-device
puts all ranks on cuda:0 (the sharded learner with world_size > 1 on a 1-GPU box).

Rank 0 prints ONE JSON line with the contract fields plus `roofline` (the fused step kernel: launch
durations measured live with HIP events on torch's current stream over the timed region) and, at
N=1, `cpu_baseline` (the rsl_rl PPO-update path on this box's host cores over the same rollout).
`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run.
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


def _code_only(text):
    """C / C++ source without comments and without whitespace: what the compiler sees (string literals in these sources hold no
    comment markers)."""
    import re
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return re.sub(r"\s+", "", text)


def step_kernel_sha16():
    """First 16 hex digits of sha256 over the CODE of the step kernel's sources (comments and whitespace stripped: documentation may be
    corrected without orphaning the committed counters) and the flags it is built with: ties profiles/step_kernel_counters.json to
    the build it measured."""
    import hashlib
    h = hashlib.sha256()
    for fn in ("deep-whole-body-control_amd/csrc/wbc_step_kernel.hip", "deep-whole-body-control_amd/csrc/wbc_device.h", "include/wbc_sim.h"):
        h.update(_code_only(open(os.path.join(ROOT, fn)).read()).encode())
    try:
        import __graft_entry__ as g
        h.update(" ".join(g.COMMON_FLAGS + g.EXTRA_FLAGS.get("wbc_step_kernel.hip", [])).encode())
    except Exception:      # noqa: BLE001
        pass
    return h.hexdigest()[:16]


def _usable_cores(cap=16):
    """Cores this process may actually use: affinity mask and cgroup CPU quota (a container on a big host reports the host's
    cpu_count), capped: these layers (<= 128 wide) stop scaling long before a big host runs out of cores."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


CPU_BASELINE_THREADS = _usable_cores()


REFERENCE_RSL_RL = "/root/reference/rsl_rl"     # exists in the build container only (never on the GPU box)


def _cpu_learner_classes(kind="port"):
    """(kind, PPO, ActorCritic) of the CPU baseline. "port" (what every bench line on a GPU box reports: /root/reference does not
    exist there): this package's eager torch path -- the same op sequence as the reference's rsl_rl, pinned to it seed for seed by
    tests/test_ppo_parity.py. "reference": the reference's own rsl_rl classes, importable in the build container only -- used by
    `bench.py --cpu-baseline-only`, which times BOTH on one box so that the port's cost is shown next to the reference's
    (profiles/r05_cpu_baseline_reference_vs_port.json)."""
    if kind == "reference":
        import contextlib
        import io
        sys.dont_write_bytecode = True              # the reference tree is read-only
        if REFERENCE_RSL_RL not in sys.path:
            sys.path.insert(0, REFERENCE_RSL_RL)
        with contextlib.redirect_stdout(io.StringIO()):
            from rsl_rl.algorithms import PPO
            from rsl_rl.modules import ActorCritic
        return "reference", PPO, ActorCritic
    from wbc_amd.rsl_rl.algorithms import PPO
    from wbc_amd.rsl_rl.modules import ActorCritic
    return "port", PPO, ActorCritic


def _synthetic_runner(n=ENVS_PER_GPU, t=T_STEPS):
    """A stand-in for the GPU runner on a box without a GPU: a randomly initialised policy and a synthetic rollout storage of the
    bench shape (SURVEY.md section 8d config 1's recipe at 4096 x 40: N(0,1) observations, 0.01 N(0,1) rewards, 5 % dones)."""
    import contextlib
    import io
    import types
    import torch
    from wbc_amd.config import WidowGo1RoughCfgPPO, class_to_dict
    from wbc_amd.rsl_rl.algorithms import PPO
    from wbc_amd.rsl_rl.modules import ActorCritic
    train = class_to_dict(WidowGo1RoughCfgPPO())
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        ac = ActorCritic(76, 76, 18, **train["policy"], num_priv=24, num_hist=10, num_prop=76)
        alg = PPO(ac, device="cpu", **train["algorithm"])
        alg.init_storage(n, t, [860], [None], [18])
    st = alg.storage
    for x in (st.observations, st.actions, st.values, st.actions_log_prob, st.mu):
        x.normal_()
    st.rewards.normal_().mul_(0.01)
    st.dones.copy_((torch.rand(st.dones.shape) < 0.05).to(st.dones.dtype))
    st.sigma.fill_(1.0)
    return types.SimpleNamespace(alg=alg)


def cpu_baseline(runner, sim_sample_envs=512, learner="port"):
    """The north star's CPU baseline: the rsl_rl PPO-update path on this box's host cores, on the SAME rollout the GPU
    learner has just consumed (its storage copied to the host): compute_returns + update() (median of 3) and
    update_dagger() (once), torch CPU with all cores. `value` = N*T / (returns + update): the learner-only ceiling of a
    CPU run in env-steps/s (the reference has no CPU simulator: Isaac Gym is CUDA-only). For context, `sim_port` times
    this framework's scalar C oracle of the sim step (OpenMP over the envs: the thread count is reported) over a bounded sample."""
    import contextlib
    import io
    import statistics
    import numpy as np
    import torch
    from wbc_amd.config import WidowGo1RoughCfgPPO, class_to_dict
    kind, PPO, ActorCritic = _cpu_learner_classes(learner)
    nthreads = CPU_BASELINE_THREADS
    torch.set_num_threads(nthreads)
    gst = runner.alg.storage
    T, N = gst.num_transitions_per_env, gst.num_envs
    train = class_to_dict(WidowGo1RoughCfgPPO())
    alg_kw = dict(train["algorithm"])
    with contextlib.redirect_stdout(io.StringIO()):
        ac = ActorCritic(76, 76, 18, **train["policy"], num_priv=24, num_hist=10, num_prop=76)
        ac.load_state_dict({k: v.detach().cpu() for k, v in runner.alg.actor_critic.state_dict().items()})
        alg = PPO(ac, device="cpu", **alg_kw)
        alg.init_storage(N, T, [860], [None], [18])
    host = {k: getattr(gst, k).detach().cpu().clone() for k in
            ("observations", "actions", "rewards", "dones", "values", "actions_log_prob", "mu", "sigma")}
    last_obs = host["observations"][-1]

    def fill():
        st = alg.storage
        for k, v in host.items():
            getattr(st, k).copy_(v.view_as(getattr(st, k)))
        st.step = T
    t_ret, t_upd = [], []
    for rep in range(3):
        if rep > 0 and t_upd[0] > 15.0:       # a slow host: one measurement instead of the median of three (bounded sample)
            break
        fill()
        t0 = time.perf_counter()
        with torch.inference_mode():
            alg.compute_returns(last_obs)
        t1 = time.perf_counter()
        alg.update()
        t2 = time.perf_counter()
        t_ret.append(t1 - t0)
        t_upd.append(t2 - t1)
    t_dag = float("nan")
    if t_upd[0] <= 30.0:
        fill()
        with torch.inference_mode():
            alg.compute_returns(last_obs)
        t0 = time.perf_counter()
        alg.update_dagger()
        t_dag = time.perf_counter() - t0
    ret_s, upd_s = statistics.median(t_ret), statistics.median(t_upd)
    out = {"value": N * T / (ret_s + upd_s), "unit": "env-steps/s", "cores": nthreads, "kind": kind,
           "scope": "learner only: compute_returns + PPO.update on the host; a ceiling for a CPU run, not an end-to-end rate "
                    "(the reference has no CPU simulator; sim_port below is this framework's scalar C oracle)",
           "sample": f"rsl_rl PPO-update path on the host ({'the REFERENCE rsl_rl classes imported from ' + REFERENCE_RSL_RL if kind == 'reference' else 'this package eager torch CPU path, pinned to the reference by tests/test_ppo_parity.py'}): "
                     f"compute_returns {ret_s * 1e3:.1f} ms + update() {upd_s:.2f} s (median of {len(t_upd)}; 5 epochs x 4 minibatches over the "
                     f"{N}x{T} rollout the GPU learner consumed), update_dagger() {t_dag:.2f} s (once); learner only, no CPU sim exists",
           "compute_returns_s": ret_s, "update_s": upd_s, "update_dagger_s": t_dag,
           "sample_epochs_per_s": N * T * 5 / upd_s}
    # context: this framework's own scalar CPU oracle of the sim step (test infrastructure), OpenMP over the envs
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as ora
        from wbc_amd import abi
        from wbc_amd.config import WidowGo1RoughCfg
        ora.build()
        m = abi.load_default_model()
        cfg = WidowGo1RoughCfg()
        n = sim_sample_envs
        sim = ora.OracleSim(abi.fill_model(m), abi.fill_task_cfg(cfg, m), n, seed=1, precision="f64")
        sim.set_curriculum(ora.default_curriculum(cfg, 1))
        sim.reset_all()
        rng = np.random.default_rng(0)
        t0 = time.perf_counter()
        k = 0
        while k < 10 and time.perf_counter() - t0 < 8.0:
            sim.step(0.3 * rng.standard_normal((n, 18)))
            k += 1
        out["sim_port"] = {"value": n * k / (time.perf_counter() - t0), "unit": "env-steps/s", "cores": sim.threads,
                           "sample": f"oracle/wbc_oracle.c (fp64, OpenMP over envs, {sim.threads} threads), {n} envs x {k} steps"}
    except Exception as e:      # the checker is optional here
        out["sim_port"] = {"error": repr(e)}
    return out


def _self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU."""
    import subprocess
    port = os.environ.get("MASTER_PORT", str(29500 + (os.getpid() % 2000)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def _phase(hist, pred):
    sel = [h for h in hist if pred(h)]
    if not sel:
        return None
    return {"count": len(sel), "collection_ms": 1e3 * sum(h["collection_time"] for h in sel) / len(sel),
            "learn_ms": 1e3 * sum(h["learn_time"] for h in sel) / len(sel)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--global-envs", type=int, default=0,
                    help="strong scaling: this many envs in total, split evenly over the ranks (BASELINE.json configs[3]: 16384 over 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="create the RCCL process group even for one rank (exercises the multi-GPU code path on one GPU)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend: nccl = RCCL over xGMI (production); gloo = host-staged collectives (single-device rigs)")
    ap.add_argument("--same-device", action="store_true",
                    help="all ranks on cuda:0 (needs --backend gloo): runs the sharded learner with world_size > 1 on a 1-GPU box; "
                         "the ranks share the device, so the value is not a scaling figure")
    ap.add_argument("--log", action="store_true",
                    help="run the loop as train.py does (log_dir set: per-step episode bookkeeping, per-iteration log text, checkpoints)")
    ap.add_argument("--terrain", choices=["plane", "trimesh", "grid"], default="plane",
                    help="plane = BASELINE.json configs[1] (the bench line); trimesh = the shipped fractal-Perlin terrain; "
                         "grid = the base class's sub-terrain grid with the terrain-level curriculum (configs[2])")
    ap.add_argument("--contact-iters", type=int, default=0,
                    help="override sim.physx.num_position_iterations (the contact solver's sweeps per substep; the reference ships 4 = the default)")
    ap.add_argument("--dagger-every", type=int, default=0,
                    help="override algorithm.dagger_update_freq (widowGo1_config.py:365 ships 20): every N-th iteration collects with the history "
                         "encoder's latent and runs PPO.update_dagger() (ppo.py:265-291) instead of update(); 1 = the teacher -> student "
                         "distillation regime of BASELINE.json configs[4]")
    ap.add_argument("--regime", choices=["default", "standing"], default="default",
                    help="default = random-init policy under the shipped thresholds (13 %% of the envs reset per step, most robots airborne); "
                         "standing = what a trained run looks like to the step kernel: every robot on its feet, 500-step episodes "
                         "(z_threshold 0.25 as tools/train_walk.py, near-zero mean actions, exploration std at its floor)")
    ap.add_argument("--terrain-curriculum", action="store_true",
                    help="with --terrain trimesh: terrain.curriculum=True on the Perlin field (grid always has it)")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="no GPU needed: time the rsl_rl PPO-update path on this box's host cores over a synthetic 4096 x 40 rollout -- the "
                         "REFERENCE's own classes where /root/reference/rsl_rl exists (kind 'reference') and this package's eager port "
                         "(kind 'port'), both on the same storage contents")
    args = ap.parse_args()

    if args.cpu_baseline_only:
        r = _synthetic_runner(args.envs_per_gpu, T_STEPS)
        out = {"metric": "env-steps/sec, rsl_rl PPO-update path on the host (learner only)", "unit": "env-steps/s", "n_gpus": 0,
               "config": {"workload": f"synthetic {args.envs_per_gpu} x {T_STEPS} rollout storage, widowGo1 hyper-parameters"}}
        if os.path.isdir(REFERENCE_RSL_RL):
            out["cpu_baseline"] = cpu_baseline(r, sim_sample_envs=64, learner="reference")
        out["cpu_baseline_port"] = cpu_baseline(r, sim_sample_envs=64, learner="port")
        if "cpu_baseline" in out:
            out["port_over_reference_update_time"] = out["cpu_baseline_port"]["update_s"] / out["cpu_baseline"]["update_s"]
        else:
            out["cpu_baseline"] = out["cpu_baseline_port"]
        print(json.dumps(out))
        return

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        _self_launch(args)
    # stdout carries ONE JSON line and nothing else: libraries that print to the C-level stdout (RCCL's version banner on
    # communicator creation) are sent to stderr by re-pointing fd 1; the JSON line goes out through a private duplicate
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}, or run `python bench.py --gpus {args.gpus}` without a launcher")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the hot path only exists as HIP kernels")
    if args.global_envs:
        assert args.global_envs % world == 0, "--global-envs must divide evenly over the ranks"
        args.envs_per_gpu = args.global_envs // world
    if args.same_device:
        assert args.backend == "gloo", "--same-device needs --backend gloo (RCCL cannot put two ranks on one device)"
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    group = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        group = dist.group.WORLD

    import __graft_entry__ as ge
    if rank == 0:
        ge.build_product()
    if use_dist:
        dist.barrier()
    from wbc_amd.config import WidowGo1RoughCfg, WidowGo1RoughCfgPPO, class_to_dict
    from wbc_amd.envs import WidowGo1
    from wbc_amd.rsl_rl.runners import OnPolicyRunner

    cfg = WidowGo1RoughCfg()
    cfg.env.num_envs = args.envs_per_gpu
    if args.terrain == "plane":
        cfg.terrain.mesh_type = "plane"               # BASELINE.json configs[1]: flat terrain (the shipped default is the Perlin trimesh)
        terrain_name, cfg_idx = "flat", 1
    elif args.terrain == "grid":
        from wbc_amd.config import use_grid_terrain
        use_grid_terrain(cfg)                         # LeggedRobotCfg.terrain (LRC:43-66): 10 levels x 20 types of 8 m tiles, curriculum on
        terrain_name, cfg_idx = "sub-terrain grid (10 x 20 tiles, terrain-level curriculum)", 2
    else:
        cfg.terrain.curriculum = bool(args.terrain_curriculum)
        terrain_name, cfg_idx = "fractal-Perlin trimesh" + (", terrain-level curriculum on" if args.terrain_curriculum else ""), 2
    if args.contact_iters:
        cfg.sim.physx.num_position_iterations = args.contact_iters
    if args.regime == "standing":
        cfg.termination.z_threshold = 0.25            # (the shipped 0.325 ends every episode at touchdown)
    train_cfg = WidowGo1RoughCfgPPO()
    torch.manual_seed(train_cfg.seed)                 # identical replicas; env RNG differs per rank
    env = WidowGo1(cfg, sim_device=device, seed=train_cfg.seed + rank)
    train = class_to_dict(train_cfg)
    if args.dagger_every:
        train["algorithm"]["dagger_update_freq"] = args.dagger_every
    dagger_freq = int(train["algorithm"]["dagger_update_freq"])
    log_dir = None
    if args.log:                                       # the logged loop (train.py): text goes to /dev/null, checkpoints to a scratch directory
        import tempfile
        log_dir = tempfile.mkdtemp(prefix=f"wbc_bench_log_r{rank}_")
    runner = OnPolicyRunner(env, train, log_dir=log_dir, device=device, dist_group=group)
    env.collect_episode_stats = True                  # extras['episode'] is filled on every step as the reference does (WG:743-750)
    if args.regime == "standing":                     # a policy that keeps the robots up: zero mean action (the PD law holds the default stance), std at its floor
        ac = runner.alg.actor_critic
        with torch.no_grad():
            for name, p in ac.named_parameters():
                if name.endswith("std"):
                    p.copy_(torch.as_tensor(train["algorithm"]["min_policy_std"], device=p.device).view_as(p))
            for head in (ac.actor.actor_leg_control_head, ac.actor.actor_arm_control_head):
                last = [m for m in head if hasattr(m, "weight")][-1]
                last.weight.mul_(0.02)
                last.bias.zero_()
    torch.manual_seed(train_cfg.seed + 1000 * rank)   # replicas are identical (broadcast at construction); exploration noise is per rank
    T = runner.num_steps_per_env

    # HIP-event timing of fused-step launches in the timed region (torch's current stream is the stream the kernel is
    # launched on): every 16th launch and at most 128 pairs. (Round 5 measured what the pairs themselves cost: at every 4th launch --
    # 500 pairs = 1000 live events over 50 iterations -- the collection phase took 8.6 ms instead of 7.2; at every 16th nothing.)
    events, upd_events = [], []
    raw_step = env.sim.step
    timing_on = {"v": False}
    nstep = {"n": 0, "g": 0}

    stride = int(os.environ.get("WBC_BENCH_EVENT_STRIDE", "16"))
    MAX_PAIRS = 128

    def timed_step(*a, **kw):                        # (whatever signature WbcSim.step has)
        nstep["n"] += 1
        if timing_on["v"] and stride and nstep["n"] % stride == 0 and len(events) < MAX_PAIRS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            raw_step(*a, **kw)
            e1.record()
            events.append((e0, e1))
        else:
            raw_step(*a, **kw)
    env.sim.step = timed_step

    # The fused PPO minibatch step (ppo_chain + ppo_wgrad + reducers: one C-ABI call) is timed INSIDE the
    # timed region on every 10th call (2 of an update's 20 minibatches: an event pair around every call cost 4.5 ms per
    # iteration, around 1 in 10 it is below the run-to-run noise).
    from wbc_amd.native import lib as _lib
    _L = _lib()
    last_grad_args = {}

    def timed(raw_grad):
        def timed_grad(*a):
            last_grad_args["a"] = a
            nstep["g"] += 1
            if timing_on["v"] and nstep["g"] % 10 == 0 and len(upd_events) < MAX_PAIRS:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = raw_grad(*a)
                e1.record()
                upd_events.append((e0, e1))
                return r
            return raw_grad(*a)
        return timed_grad
    # (the first minibatch call of an update packs the weight streams, the other 19 find them kept current by the fused Adam)
    _L.wbc_ppo_minibatch_grad = timed(_L.wbc_ppo_minibatch_grad)
    _L.wbc_ppo_minibatch_grad_packed = timed(_L.wbc_ppo_minibatch_grad_packed)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Two priming iterations before the W warm-up steps: iteration 0 (a DAgger update) and the first PPO update (first
    # launches of every update kernel, the 0.6 GB workspace: ~110 ms instead of 13) are one-off set-up costs that a small W
    # would otherwise push into the timed region.
    def note(msg):
        if rank == 0 and os.environ.get("WBC_BENCH_VERBOSE"):
            print(f"[bench +{time.perf_counter() - t_start:.1f}s] {msg}", file=sys.stderr, flush=True)
    import contextlib
    t_start = time.perf_counter()
    sink = open(os.devnull, "w") if args.log else None
    with (contextlib.redirect_stdout(sink) if sink is not None else contextlib.nullcontext()):
        runner.learn(2, init_at_random_ep_len=True)
        barrier()
        note("primed")
        runner.learn(max(args.warmup, 0)) if args.warmup > 0 else None
        barrier()
        timing_on["v"] = rank == 0
        t0 = time.perf_counter()
        runner.learn(args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
    timing_on["v"] = False
    note(f"timed region done: {elapsed:.3f}s")
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    total_env_steps = args.envs_per_gpu * world * T * args.steps
    value = total_env_steps / elapsed

    # the gradient all-reduce of one minibatch, measured after the timed region: the flat 0.67 MB bucket, 50 back-to-back calls
    allreduce_us = None
    if use_dist:
        nparam = sum(p.numel() for p in runner.alg.actor_critic.parameters())
        buf = torch.zeros(nparam, device=device)
        from wbc_amd import collectives
        for _ in range(5):
            collectives.all_reduce(buf, group)
        torch.cuda.synchronize()
        if args.backend == "nccl":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                collectives.all_reduce(buf, group)
            e1.record()
            torch.cuda.synchronize()
            allreduce_us = e0.elapsed_time(e1) * 1e3 / 50
        else:                                             # host-staged: wall clock (device -> host, gloo, host -> device)
            tw = time.perf_counter()
            for _ in range(50):
                collectives.all_reduce(buf, group)
            torch.cuda.synchronize()
            allreduce_us = (time.perf_counter() - tw) * 1e6 / 50

    checkpoint_ms = None
    if args.log and rank == 0:         # what the one end-of-learn() checkpoint (OPR:182) inside the timed region cost
        torch.cuda.synchronize()
        tc0 = time.perf_counter()
        runner.save(os.path.join(log_dir, "bench_probe.pt"))
        torch.cuda.synchronize()
        checkpoint_ms = (time.perf_counter() - tc0) * 1e3
    if rank == 0:
        hist = runner.history[-args.steps:]
        kern_ms = sum(a.elapsed_time(b) for a, b in events) / max(len(events), 1)
        algo_bytes = ALGO_BYTES_PER_ENV_STEP * args.envs_per_gpu
        achieved = algo_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0

        def counter_file(name, want_rows=None):
            path = os.path.join(ROOT, "profiles", name)
            try:
                d = json.load(open(path))
                if want_rows is not None and d.get("rows_per_launch", want_rows) != want_rows:
                    return None
                return d.get("hbm_bytes_per_launch")
            except Exception:
                return None
        # Counter-derived fields are NOT collected by this run (rocprofv3 --pmc needs its own passes): they are read from the
        # committed summary of the same bench command, and the line says so (roofline.counters_source); null when the summary
        # does not match this configuration.
        traffic = active_lanes = valu_per_wave = counters_source = None
        if args.envs_per_gpu == 4096 and args.terrain == "plane" and args.regime == "default":
            try:
                cj = json.load(open(os.path.join(ROOT, "profiles", "step_kernel_counters.json")))
                sk = cj["step_kernel"]
                if cj.get("kernel_sha16") == step_kernel_sha16() and cj.get("contact_iters") == int(env.tcfg.contact_iters):
                    traffic, active_lanes, valu_per_wave = sk["hbm_bytes_per_launch"], sk["active_lanes"], sk["valu_insts_per_wave"]
                    counters_source = {"file": "profiles/step_kernel_counters.json", "collected_with": cj["collected_with"], "date": cj["date"],
                                       "kernel_sha16": cj["kernel_sha16"],
                                       "note": "separate rocprofv3 --pmc passes over the same bench loop on the same kernel source; not measured by this run"}
                else:      # counters of another kernel build say nothing about this one: the derived fields stay null
                    counters_source = {"file": "profiles/step_kernel_counters.json", "stale": True,
                                       "note": f"collected on kernel source {cj.get('kernel_sha16')} with {cj.get('contact_iters')} solver sweeps; this run: "
                                               f"{step_kernel_sha16()} with {int(env.tcfg.contact_iters)}"}
            except Exception:
                pass
        strong = bool(args.global_envs)
        out = {
            "metric": "env-steps/sec whole node, widowGo1 4096-env PPO",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": f"synthetic (random-init policy, seeded domain randomisation, {terrain_name} terrain)",
            "config": {"workload": f"widowGo1 {terrain_name} terrain, {args.envs_per_gpu} envs per GPU, PPO fp32 "
                                   f"(BASELINE.json configs[{cfg_idx}]); T={T} steps/iteration, 5 epochs x 4 minibatches, "
                                   f"DAgger (update_dagger) every {dagger_freq}{'th' if dagger_freq != 1 else 'st'} iteration; contact solver {int(env.tcfg.contact_iters)} sweeps per substep "
                                   f"(sim.physx.num_position_iterations, reference: 4); regime: "
                                   + ("random-init policy, shipped thresholds" if args.regime == "default" else
                                      "STANDING (z_threshold 0.25, zero-mean policy at the std floor: every robot on its feet, 500-step episodes)"),
                       "envs_per_gpu": args.envs_per_gpu, "contact_iters": int(env.tcfg.contact_iters), "regime": args.regime,
                       "global_envs": args.envs_per_gpu * world, "steps_per_env": T,
                       "parallelism": (f"env-shard x{world}, 1 {'RCCL' if args.backend == 'nccl' else 'gloo (host-staged)'} grad all-reduce/minibatch"
                                       f" + 1 three-scalar all-reduce/iteration" + (", ALL RANKS ON ONE DEVICE (functional run, not a scaling figure)" if args.same_device else "")
                                       if use_dist else "single GPU"),
                       "rccl_ranks": dist.get_world_size(group) if use_dist and args.backend == "nccl" else 0,
                       "backend": args.backend if use_dist else None, "same_device": bool(args.same_device),
                       "logged": bool(args.log), "end_of_learn_checkpoint_ms": checkpoint_ms,
                       "grad_allreduce_us": allreduce_us,
                       "dagger_update_freq": dagger_freq,
                       "collection_ms": 1e3 * sum(h["collection_time"] for h in hist) / len(hist),
                       "learn_ms": 1e3 * sum(h["learn_time"] for h in hist) / len(hist),
                       # the two kinds of iteration apart (a DAgger iteration: rollout through the history encoder, update_dagger)
                       "ppo_iterations": _phase(hist, lambda h: h["it"] % dagger_freq != 0),
                       "dagger_iterations": _phase(hist, lambda h: h["it"] % dagger_freq == 0)},
            "roofline": {"kernel": "wbc_step_kernel", "bound": "hbm",
                         "limiter": "vector-instruction issue + the slowest wave's dependent chain, not HBM (SURVEY.md section 8d: the 40 % HBM target is "
                                    "not meaningful for this kernel at this size; 'bound' = the roofline the contract prices it against)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "launch_ms": kern_ms,
                         "algorithmic_bytes_per_launch": algo_bytes, "launches_timed": len(events),
                         # what actually bounds this kernel (one wavefront per env, 4 per SIMD at 4096 envs): vector-instruction issue
                         # and the dependent chain of the slowest wave. active_lanes = SQ_THREAD_CYCLES_VALU /
                         # SQ_INSTS_VALU; valu_issue_frac = the share of the launch the SIMDs spend issuing vector instructions =
                         # insts per wave x waves per SIMD x 4 cycles / (launch time x 2.4 GHz) -- 4 cycles is what SQ_ACTIVE_INST_VALU
                         # reports (1.01 quad-cycles per instruction); a plain fp32 v_fma issues in 2 (MI355X_MICROARCH.md): _2cyc
                         "active_lanes": active_lanes,
                         "valu_issue_frac": (valu_per_wave * (args.envs_per_gpu / 1024.0) * 4.0 / (kern_ms * 1e-3 * 2.4e9)
                                             if valu_per_wave and kern_ms > 0 else None),
                         "valu_issue_frac_2cyc": (valu_per_wave * (args.envs_per_gpu / 1024.0) * 2.0 / (kern_ms * 1e-3 * 2.4e9)
                                                  if valu_per_wave and kern_ms > 0 else None),
                         "counters_source": counters_source},
        }
        if upd_events:
            upd_ms = sum(a.elapsed_time(b) for a, b in upd_events) / len(upd_events)
            rows = int(last_grad_args["a"][9])
            flops = UPDATE_FLOPS_PER_ROW * rows
            tf = flops / (upd_ms * 1e-3) / 1e12
            out["roofline_update"] = {"kernel": "wbc_ppo_minibatch_grad(_packed) = ppo_chain + ppo_wgrad + reducers (+ chain_pack on the first call of an update)", "bound": "mfma",
                                      "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
                                      "traffic": counter_file("ppo_update_traffic.json") if rows == 40960 else None,
                                      "launch_ms": upd_ms, "algorithmic_flops_per_launch": flops, "rows_per_launch": rows,
                                      "launches_timed": len(upd_events), "timed": "inside the timed region, every 10th minibatch call"}
        if world == 1 and not args.no_cpu_baseline:
            note("cpu baseline ...")
            out["cpu_baseline"] = cpu_baseline(runner)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if log_dir is not None:
        import shutil
        shutil.rmtree(log_dir, ignore_errors=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
