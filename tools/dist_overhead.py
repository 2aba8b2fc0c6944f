"""What a process group costs the loop on ONE GPU, separated: (a) no group; (b) a 1-rank RCCL group exists but the learner does
not use it (watchdog / heartbeat threads only); (c) the learner uses it (parameter broadcast, 20 gradient all-reduces + 1
statistics all-reduce per iteration). Prints mean collection / learn ms over the non-DAgger iterations of each leg."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd"))
import torch
import torch.distributed as dist
from wbc_amd.config import WidowGo1RoughCfg, WidowGo1RoughCfgPPO, class_to_dict
from wbc_amd.envs import WidowGo1
from wbc_amd.rsl_rl.runners import OnPolicyRunner

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def leg(tag, group):
    cfg = WidowGo1RoughCfg(); cfg.env.num_envs = N; cfg.terrain.mesh_type = "plane"
    tc = WidowGo1RoughCfgPPO(); torch.manual_seed(tc.seed)
    env = WidowGo1(cfg, sim_device="cuda:0", seed=tc.seed)
    runner = OnPolicyRunner(env, class_to_dict(tc), log_dir=None, device="cuda:0", dist_group=group)
    env.collect_episode_stats = True
    if os.environ.get("WBC_ASYNC_STATS"):              # experiment: episode statistics on a side stream (what rounds 1-2 shipped)
        env.async_episode_stats = True
    runner.learn(2, init_at_random_ep_len=True)
    runner.learn(4)
    torch.cuda.synchronize()
    runner.history.clear()
    # where inside a rollout step the time goes: events around the policy inference and around the env step of every 5th step
    ev = []
    raw_act, raw_step, k = runner.alg.act, env.step, {"n": 0}

    def act(obs, critic_obs, hist_encoding=False, side_job=None):       # (the runner looks for the side_job parameter)
        k["n"] += 1
        if k["n"] % 5 == 0:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record(); out = raw_act(obs, critic_obs, hist_encoding, side_job=side_job); e[1].record()
            ev.append(e)
            return out
        return raw_act(obs, critic_obs, hist_encoding, side_job=side_job)

    def step(a):
        out = raw_step(a)
        if k["n"] % 5 == 0:
            ev[-1][2].record()
        return out
    runner.alg.act, env.step = act, step
    runner.learn(13)                                   # iterations 6..18: no DAgger iteration among them
    torch.cuda.synchronize()
    t_act = sum(e[0].elapsed_time(e[1]) for e in ev) / len(ev) * 1e3
    t_step = sum(e[1].elapsed_time(e[2]) for e in ev) / len(ev) * 1e3
    c = sum(h["collection_time"] for h in runner.history) / len(runner.history) * 1e3
    l = sum(h["learn_time"] for h in runner.history) / len(runner.history) * 1e3
    print(f"{tag:34s} collect {c:6.3f} ms  learn {l:6.3f} ms  sum {c + l:6.3f} ms | per rollout step: act {t_act:6.1f} us, env.step (+ stats) {t_step:6.1f} us", flush=True)
    env.sim.close()


leg("(a) no process group", None)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1, device_id=torch.device("cuda:0"))
dist.all_reduce(torch.zeros(4, device="cuda:0")); torch.cuda.synchronize()
if os.environ.get("WBC_ONLY_C") is None:
    leg("(b) 1-rank RCCL group, unused", None)
leg("(c) 1-rank RCCL group, used", dist.group.WORLD)
if os.environ.get("WBC_ONLY_C") is None:
    leg("(a'') no group use again", None)
dist.destroy_process_group()
