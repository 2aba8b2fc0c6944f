"""Host time to ENQUEUE one 40-step rollout (no synchronisation inside) next to its synchronised wall time: is the collection phase
bound by the host's launch rate or by the GPU?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd"))
import torch
from wbc_amd.config import WidowGo1RoughCfg, WidowGo1RoughCfgPPO, class_to_dict
from wbc_amd.envs import WidowGo1
from wbc_amd.rsl_rl.runners import OnPolicyRunner
cfg = WidowGo1RoughCfg(); cfg.env.num_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096; cfg.terrain.mesh_type = "plane"
tc = WidowGo1RoughCfgPPO(); torch.manual_seed(tc.seed)
env = WidowGo1(cfg, sim_device="cuda:0", seed=tc.seed)
runner = OnPolicyRunner(env, class_to_dict(tc), log_dir=None, device="cuda:0")
runner.learn(3, init_at_random_ep_len=True)
alg = runner.alg
obs = env.get_observations()
enq, wall = [], []
with torch.inference_mode():
    for rep in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40):
            actions = alg.act(obs, obs, False)
            slot = alg.next_observation_slot()
            if slot is not None: env.set_obs_output(slot)
            slots = alg.rollout_slots()
            if slots is not None: env.set_rollout_output(*slots)
            obs, priv, rewards, arm_rewards, dones, infos = env.step(actions)
            alg.process_env_step(rewards, arm_rewards, dones, infos)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        enq.append((t1 - t0) * 1e3); wall.append((t2 - t0) * 1e3)
        alg.storage.clear()
print("rollout of 40 steps: host enqueue ms", [round(x, 2) for x in enq], " synchronised wall ms", [round(x, 2) for x in wall])
