"""OnPolicyRunner: the rollout / learn loop with the reference's constructor, `learn`, `save`,
`load` and `get_inference_policy` (rsl_rl/runners/on_policy_runner.py:46-300). The throughput
figure is the reference's own definition, fps = T*N / (collection_time + learn_time) (OPR:206).

Differences that are deliberate: no wandb/tensorboard dependency (stats go to `self.history` and,
with a log_dir, to stdout); per-step reward bookkeeping stays on the device (the reference pulls
lists to the host every step, OPR:147-151); optional `dist_group` for the sharded multi-GPU learner.
"""
from __future__ import annotations

import os
import statistics
import time
from collections import deque

import torch

from ... import collectives
from ..algorithms import PPO
from ..env import VecEnv
from ..modules import ActorCritic

_POLICIES = {"ActorCritic": ActorCritic}
_ALGORITHMS = {"PPO": PPO}


def _accepts(fn, name) -> bool:
    import inspect
    try:
        return name in inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return False


class _EpisodeTracker:
    """learn()'s episode bookkeeping (OPR:140-154) as device state: running (reward, arm reward, length) per env and the
    deque(maxlen=100) trio + the done-fraction deque as device rings, advanced by one launch per env step
    (wbc_runner_track_episodes) and read back once per iteration."""
    CAP = 100                                            # the reference's deque maxlen (OPR:103-106)

    @classmethod
    def create(cls, env, rewards, arm_rewards, dones):
        ok = all(t.is_cuda and t.is_contiguous() for t in (rewards, arm_rewards, dones)) and rewards.dtype == torch.float32 \
            and arm_rewards.dtype == torch.float32 and dones.dtype == torch.int64 and rewards.dim() == 1
        return cls(env, rewards.shape[0], rewards.device) if ok else None

    def __init__(self, env, n, device):
        from ...native import check, lib
        self._check, self._L = check, lib()
        self.n, self.device = n, device
        self.state = torch.zeros(self._L.wbc_runner_track_state_floats(n, self.CAP), device=device)
        # An env that publishes extras['episode'] with one launch per step (WidowGo1.attach_episode_tracker) carries the
        # bookkeeping in that launch as an extra workgroup: no launch and no stream hop of its own.
        self._hooked_env, self._env_from = None, None
        if hasattr(env, "attach_episode_tracker") and getattr(env, "collect_episode_stats", False):
            self._env_from = env.attach_episode_tracker(self.state, self.CAP)     # first env step (its common_step_counter) the env accounts for
            self._hooked_env = env

    def _launch(self, rewards, arm_rewards, dones):
        self._check(self._L.wbc_runner_track_episodes(rewards.data_ptr(), arm_rewards.data_ptr(), dones.data_ptr(), self.n, self.CAP,
                                                      self.state.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream),
                    "wbc_runner_track_episodes")

    def step(self, rewards, arm_rewards, dones):
        """Account for the env step that has just produced these tensors. An attached env does it inside its own step() -- from
        the step index attach_episode_tracker returned on; a step before that (the one during which the tracker was created) is
        launched here. Gated on the env's step counter, not on call order: no step is counted twice or dropped."""
        env = self._hooked_env
        if env is None or self._env_from is None or int(env.common_step_counter) < self._env_from:
            self._launch(rewards, arm_rewards, dones)

    def close(self):
        if self._hooked_env is not None:
            self._hooked_env.attach_episode_tracker(None, 0)
        self._hooked_env = None

    def device_tail(self):
        """The rings and their header as one device tensor (4 cap + 4 floats), for a caller that packs several read-backs into one copy."""
        return self.state[3 * self.n:]

    def summary(self, tail=None):
        """{mean_reward, mean_arm_reward, mean_episode_length, dones} over the rings (empty before the first finished episode,
        as the reference's `if len(rewbuffer) > 0`). `tail`: a host copy of device_tail(); default: copy it now. The caller has
        synchronised the device."""
        tail = self.device_tail().cpu() if tail is None else tail
        cap = self.CAP
        ring, done_ring = tail[:3 * cap].view(cap, 3).double(), tail[3 * cap:4 * cap].double()
        hdr = tail[4 * cap:4 * cap + 4].contiguous().view(torch.int32)
        fill, dfill = int(hdr[1]), int(hdr[3])
        if fill == 0:
            return {}
        m = ring[:fill].mean(0) if fill < cap else ring.mean(0)
        d = done_ring[:dfill].mean() if dfill < cap else done_ring.mean()
        return dict(mean_reward=float(m[0]), mean_arm_reward=float(m[1]), mean_episode_length=float(m[2]), dones=float(d))


def _plain(x, where):
    """`x` as something torch.load(weights_only=True) reads back: tensors, None, bool / int / float / str, and lists / tuples / dicts
    of those; numpy scalars and arrays are converted (to Python numbers / tensors); anything else is an error naming its place."""
    import numpy as _np
    if x is None or isinstance(x, (bool, int, float, str)) or torch.is_tensor(x):
        return x
    if isinstance(x, _np.generic):
        return x.item()
    if isinstance(x, _np.ndarray):
        return torch.from_numpy(_np.ascontiguousarray(x))
    if isinstance(x, dict):
        return {str(k): _plain(v, f"{where}[{k!r}]") for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v, f"{where}[{i}]") for i, v in enumerate(x)]
    raise TypeError(f"OnPolicyRunner.save: {where} is a {type(x).__name__}; a checkpoint holds tensors, numbers, strings and lists / dicts of them "
                    "(OnPolicyRunner.load reads it with weights_only=True)")


class OnPolicyRunner:
    def __init__(self, env: VecEnv, train_cfg, log_dir=None, device="cpu", dist_group=None):
        self.cfg, self.alg_cfg, self.policy_cfg = train_cfg["runner"], train_cfg["algorithm"], train_cfg["policy"]
        self.device, self.env = device, env
        e = env.cfg.env
        policy_cls = _POLICIES[self.cfg["policy_class_name"]]
        actor_critic = policy_cls(e.num_proprio, e.num_proprio, env.num_actions, **self.policy_cfg, num_priv=e.num_priv,
                                  num_hist=e.history_len, num_prop=e.num_proprio).to(self.device)
        self.dist_group = dist_group
        if dist_group is not None:      # identical initial replicas on every rank: ONE flat broadcast (41 tensors, 675 kB), not one per parameter
            params = list(actor_critic.parameters())
            flat = torch.cat([p.data.reshape(-1) for p in params])
            collectives.broadcast(flat, 0, dist_group)
            off = 0
            for p in params:
                p.data.copy_(flat[off:off + p.numel()].view_as(p.data))
                off += p.numel()
        self.alg: PPO = _ALGORITHMS[self.cfg["algorithm_class_name"]](actor_critic, device=self.device, dist_group=dist_group,
                                                                      **self.alg_cfg)
        if hasattr(self.alg, "warm_up_collectives"):
            self.alg.warm_up_collectives()       # RCCL's first-collective set-up here, not inside the first update
        self.num_steps_per_env = self.cfg["num_steps_per_env"]
        self.save_interval = self.cfg["save_interval"]
        self.alg.init_storage(env.num_envs, self.num_steps_per_env, [env.num_obs], [env.num_privileged_obs], [env.num_actions])
        self.log_dir = log_dir
        self.writer = None
        self.tot_timesteps = 0
        self.tot_time = 0
        self.current_learning_iteration = 0
        self.dagger_update_freq = self.alg_cfg["dagger_update_freq"]
        self.history = []
        if hasattr(env, "collect_episode_stats"):
            env.collect_episode_stats = log_dir is not None
        env.reset()
        # OPR:91 passes default_dof_pos[-7:-2] (5 values): arm_fk then fails with a shape error the moment torque
        # supervision is switched on (quirk Q7: the path is dead in the reference). The 6 arm DoFs are [-8:-2].
        self.alg.set_arm_default_coeffs(env.p_gains[12:], env.d_gains[12:], env.default_dof_pos[-8:-2])

    def learn(self, num_learning_iterations, init_at_random_ep_len=False):
        env, alg = self.env, self.alg
        if init_at_random_ep_len:
            env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
        obs = env.get_observations()
        priv = env.get_privileged_observations()
        critic_obs = priv if priv is not None else obs
        obs, critic_obs = obs.to(self.device), critic_obs.to(self.device)
        alg.actor_critic.train()
        logging = self.log_dir is not None
        rewbuffer, armrewbuffer, lenbuffer, donebuffer = (deque(maxlen=100) for _ in range(4))
        n = env.num_envs
        cur_rew = torch.zeros(n, device=self.device)
        cur_arm = torch.zeros(n, device=self.device)
        cur_len = torch.zeros(n, device=self.device)
        ep_infos = []
        is_cuda = torch.device(self.device).type == "cuda"
        sync = (lambda: torch.cuda.synchronize(self.device)) if is_cuda else (lambda: None)
        loss_stats = dict(mean_value_loss=0., mean_surrogate_loss=0., mean_arm_torques_loss=0., value_mixing_ratio=0.,
                          torque_supervision_weight=0., mean_hist_latent_loss=0., mean_priv_reg_loss=0., priv_reg_coef=0.)
        tot_iter = self.current_learning_iteration + num_learning_iterations
        redirect_obs = hasattr(env, "set_obs_output") and hasattr(alg, "next_observation_slot") and getattr(alg, "fused_rollout", False)
        store_in_step = hasattr(env, "set_rollout_output") and hasattr(alg, "rollout_slots") and getattr(alg, "fused_rollout", False)
        # (WidowGo1.async_episode_stats -- extras['episode'] computed on a side stream under the next policy inference -- stays off:
        # measured on the MI355X (tools/dist_overhead.py), the two cross-stream hand-overs per env step cost more than the 8 us
        # kernel they hide (151 vs 155 us per rollout step), and with RCCL's streams in the process the extra stream can end up
        # sharing a hardware queue with the compute stream: +16 us per step)
        tracker = None                                   # device-side deques (wbc_runner_track_episodes), set up on the first logged step
        # the env's per-step episode statistics ride on the policy inference that follows each step (extra workgroups of that launch)
        carry = (hasattr(env, "take_stats_job") and getattr(alg, "fused_rollout", False) and is_cuda
                 and _accepts(alg.act, "side_job") and _accepts(alg.compute_returns, "side_job"))
        if carry:
            env.defer_episode_stats = True
        try:
            for it in range(self.current_learning_iteration, tot_iter):
                env.update_command_curriculum()
                sync()
                start = time.time()
                hist_encoding = it % self.dagger_update_freq == 0
                with torch.inference_mode():
                    for _ in range(self.num_steps_per_env):
                        actions = alg.act(obs, critic_obs, hist_encoding, side_job=env.take_stats_job()) if carry else alg.act(obs, critic_obs, hist_encoding)
                        slot = alg.next_observation_slot() if redirect_obs else None
                        if slot is not None:
                            env.set_obs_output(slot)              # the env writes the next observation where act() would copy it
                        if store_in_step:
                            slots = alg.rollout_slots()
                            if slots is not None:
                                env.set_rollout_output(*slots)    # ... and this transition's reward / done slots
                        obs, priv, rewards, arm_rewards, dones, infos = env.step(actions)
                        critic_obs = priv if priv is not None else obs
                        obs, critic_obs, rewards, arm_rewards, dones = (x.to(self.device) for x in (obs, critic_obs, rewards, arm_rewards, dones))
                        alg.process_env_step(rewards, arm_rewards, dones, infos)
                        if logging:
                            if "episode" in infos:
                                ep_infos.append(infos["episode"])
                            if tracker is None:
                                tracker = _EpisodeTracker.create(env, rewards, arm_rewards, dones) or False
                            if tracker:                           # OPR:140-154 in one launch, no host copies
                                tracker.step(rewards, arm_rewards, dones)
                                continue
                            cur_rew += rewards                    # the reference's host path (CPU envs / foreign env classes)
                            cur_arm += arm_rewards
                            cur_len += 1
                            new_ids = (dones > 0).nonzero(as_tuple=False)
                            rewbuffer.extend(cur_rew[new_ids][:, 0].cpu().numpy().tolist())
                            armrewbuffer.extend(cur_arm[new_ids][:, 0].cpu().numpy().tolist())
                            lenbuffer.extend(cur_len[new_ids][:, 0].cpu().numpy().tolist())
                            donebuffer.append(len(new_ids) / n)
                            cur_rew[new_ids] = 0
                            cur_arm[new_ids] = 0
                            cur_len[new_ids] = 0
                    sync()
                    stop = time.time()
                    collection_time = stop - start
                    start = stop
                    if carry:
                        alg.compute_returns(critic_obs, side_job=env.take_stats_job())    # the last step's statistics ride on this inference
                    else:
                        alg.compute_returns(critic_obs)
                if hist_encoding:
                    loss_stats["mean_hist_latent_loss"] = alg.update_dagger()
                else:
                    (loss_stats["mean_value_loss"], loss_stats["mean_surrogate_loss"], loss_stats["mean_arm_torques_loss"],
                     loss_stats["value_mixing_ratio"], loss_stats["torque_supervision_weight"], loss_stats["mean_priv_reg_loss"],
                     loss_stats["priv_reg_coef"]) = alg.update()
                sync()
                stop = time.time()
                learn_time = stop - start
                fps = int(self.num_steps_per_env * n / (collection_time + learn_time))       # OPR:206
                rec = dict(it=it, collection_time=collection_time, learn_time=learn_time, fps=fps, **loss_stats)
                self.tot_timesteps += self.num_steps_per_env * n
                self.tot_time += collection_time + learn_time
                if logging:
                    if not tracker and len(rewbuffer) > 0:
                        rec.update(mean_reward=statistics.mean(rewbuffer), mean_arm_reward=statistics.mean(armrewbuffer),
                                   mean_episode_length=statistics.mean(lenbuffer), dones=statistics.mean(donebuffer))
                    self.log(rec, ep_infos, tot_iter, tracker=tracker or None)     # (the tracker's rings travel in log()'s one host copy)
                    if it % self.save_interval == 0:
                        self.save(os.path.join(self.log_dir, f"model_{it}.pt"))
                self.history.append(rec)
                ep_infos.clear()
        finally:                                         # also on an exception / KeyboardInterrupt
            if carry:
                env.defer_episode_stats = False
                env.flush_stats_job()
            if tracker:
                tracker.close()
            if getattr(env, "async_episode_stats", False):   # a caller-enabled side stream: its results are complete from here on
                sync()
        self.current_learning_iteration += num_learning_iterations
        if logging:
            self.save(os.path.join(self.log_dir, f"model_{self.current_learning_iteration}.pt"))

    def log(self, rec, ep_infos, tot_iter, width=80, pad=35, tracker=None):
        """OPR:187-274 without its ~40 device synchronisations per line: everything the line needs from the device -- the means of
        extras['episode'] over the rollout, the two noise levels, the episode deques -- comes over in ONE copy."""
        first = ep_infos[0] if ep_infos else None
        index = getattr(first, "vector_index", None)
        std = self.alg.actor_critic.std.detach().reshape(-1)
        parts = [std[:12].mean().reshape(1), std[12:].mean().reshape(1)]
        vectorised = index is not None and all(getattr(i, "vector", None) is not None for i in ep_infos)
        if vectorised:
            parts.append(torch.stack([i.vector for i in ep_infos]).mean(0))
        if tracker is not None:
            parts.append(tracker.device_tail())
        host = torch.cat(parts).cpu()
        leg_std, arm_std = host[0].item(), host[1].item()
        off = 2
        ep_means = None
        if vectorised:
            ep_means = host[off:off + first.vector.numel()]
            off += first.vector.numel()
        if tracker is not None:
            rec.update(tracker.summary(host[off:]))
        lines = ["#" * width, f" Learning iteration {rec['it']}/{tot_iter} ".center(width), ""]
        lines.append(f"{'Computation:':>{pad}} {rec['fps']:.0f} steps/s (collection: {rec['collection_time']:.3f}s, "
                     f"learning {rec['learn_time']:.3f}s)")
        for key, label in (("mean_value_loss", "Value function loss:"), ("mean_surrogate_loss", "Surrogate loss:"),
                           ("mean_hist_latent_loss", "History latent supervision loss:"),
                           ("mean_priv_reg_loss", "Privileged info regularizer loss:"),
                           ("priv_reg_coef", "Privileged info regularizer lambda:"), ("mean_reward", "Mean reward:"),
                           ("mean_episode_length", "Mean episode length:"), ("dones", "Dones:")):
            if key in rec:
                lines.append(f"{label:>{pad}} {rec[key]:.4f}")
        lines.append(f"{'Leg mean action noise std:':>{pad}} {leg_std:.2f}")
        lines.append(f"{'Arm mean action noise std:':>{pad}} {arm_std:.2f}")
        if ep_infos:
            for key in first:
                if ep_means is not None and key in index:
                    val = ep_means[index[key]].item()
                elif ep_means is not None and not torch.is_tensor(first[key]):
                    val = sum(float(info[key]) for info in ep_infos) / len(ep_infos)
                else:                                    # the reference's way (OPR:214-226): one cat + mean + item per key
                    vals = [torch.as_tensor(info[key], dtype=torch.float32, device=self.device).reshape(-1) for info in ep_infos]
                    val = torch.cat(vals).mean().item()
                lines.append(f"{'Mean episode ' + key + ':':>{pad}} {val:.4f}")
        lines += ["-" * width, f"{'Total timesteps:':>{pad}} {self.tot_timesteps}", f"{'Total time:':>{pad}} {self.tot_time:.2f}s"]
        print("\n".join(lines))

    def save(self, path, infos=None, save_env_state=False):
        """Reference checkpoint format (OPR:276-282: the four keys a reference-side loader reads) plus, under 'wbc_extra', what
        the reference forgets: the history-encoder optimiser, the schedule counter, the env's
        curriculum counter and common step counter (the sim's draw counter), terrain levels, torch's CPU and device generator
        states, and -- with save_env_state -- every device tensor of the sim (64 MB at 4096 envs), so that a resumed run
        continues the same trajectories."""
        env = self.env
        infos = _plain(infos, "infos")          # load() reads with weights_only=True: what it could not read back is refused here, not at resume time
        extra = {"hist_encoder_optimizer_state_dict": self.alg.hist_encoder_optimizer.state_dict(),
                 "ppo_counter": int(self.alg.counter),
                 "env_update_counter": int(getattr(env, "update_counter", 0)),
                 "env_common_step_counter": int(getattr(env, "common_step_counter", 0)),
                 "torch_rng_state": torch.get_rng_state()}
        sim = getattr(env, "sim", None)
        if sim is not None and hasattr(sim, "step_counter"):
            extra["sim_step_counter"] = int(sim.step_counter)
        if torch.device(self.device).type == "cuda":
            extra["cuda_rng_state"] = torch.cuda.get_rng_state(self.device)
        if getattr(env, "terrain_levels", None) is not None and torch.is_tensor(getattr(env, "terrain_levels", None)):
            extra["terrain_levels"] = env.terrain_levels.detach().cpu()
        if save_env_state and sim is not None and hasattr(sim, "arena"):
            extra["sim_arena"] = sim.arena.detach().cpu()
        torch.save({
            "model_state_dict": self.alg.actor_critic.state_dict(),
            "optimizer_state_dict": self.alg.optimizer.state_dict(),
            "iter": int(self.current_learning_iteration),
            "infos": infos,
            "wbc_extra": extra,
        }, path)

    def load(self, path, load_optimizer=True, restore_rng=True):
        """OPR:284-290; a checkpoint written by the reference (no 'wbc_extra') loads the same way."""
        # weights_only: a checkpoint is tensors, numbers and plain containers (what save() writes and what the reference writes,
        # OPR:276-282); nothing in it needs the unpickler to run code
        d = torch.load(path, map_location=self.device, weights_only=True)
        self.alg.actor_critic.load_state_dict(d["model_state_dict"])
        if hasattr(self.alg.actor_critic, "mark_params_changed"):
            self.alg.actor_critic.mark_params_changed()          # the fused kernels re-pack the weights
        if load_optimizer:
            self.alg.optimizer.load_state_dict(d["optimizer_state_dict"])
        self.current_learning_iteration = d["iter"]
        extra = d.get("wbc_extra")
        if extra is not None:
            env = self.env
            if load_optimizer:
                self.alg.hist_encoder_optimizer.load_state_dict(extra["hist_encoder_optimizer_state_dict"])
            self.alg.counter = extra["ppo_counter"]
            if hasattr(env, "update_counter"):
                env.update_counter = extra["env_update_counter"]
                if hasattr(env, "_refresh_ranges") and hasattr(env, "sim"):
                    env.sim.set_curriculum(env._refresh_ranges())
            if "env_common_step_counter" in extra and hasattr(env, "common_step_counter"):
                env.common_step_counter = extra["env_common_step_counter"]
            sim = getattr(env, "sim", None)
            arena_restored = False
            if "sim_arena" in extra and sim is not None and hasattr(sim, "arena"):
                if sim.arena.shape == extra["sim_arena"].shape:
                    sim.arena.copy_(extra["sim_arena"])
                    arena_restored = True
                else:      # another build's tensor layout (enum wbc_tensor_id / WBC_NREW changed): the env state cannot be adopted
                    import warnings
                    warnings.warn(f"checkpoint {path}: its simulator arena ({tuple(extra['sim_arena'].shape)} bytes) does not match this build's "
                                  f"({tuple(sim.arena.shape)}); networks, optimisers and counters are restored, the env state is NOT "
                                  f"(robots are re-placed by a reset)", stacklevel=2)
            if "terrain_levels" in extra and torch.is_tensor(getattr(env, "terrain_levels", None)):
                if hasattr(env, "restore_terrain_levels"):      # origins follow the levels; robots re-placed unless the arena came back
                    env.restore_terrain_levels(extra["terrain_levels"], arena_restored)
                else:
                    env.terrain_levels.copy_(extra["terrain_levels"].to(env.terrain_levels.device))
            if "sim_step_counter" in extra and sim is not None:
                sim.step_counter = int(extra["sim_step_counter"])
            if restore_rng:
                torch.set_rng_state(extra["torch_rng_state"].cpu())
                if "cuda_rng_state" in extra and torch.device(self.device).type == "cuda":
                    torch.cuda.set_rng_state(extra["cuda_rng_state"].cpu(), self.device)
        return d["infos"]

    def get_inference_policy(self, device=None, stochastic=False):
        self.alg.actor_critic.eval()
        if device is not None:
            self.alg.actor_critic.to(device)
        return self.alg.actor_critic.act if stochastic else self.alg.actor_critic.act_inference
