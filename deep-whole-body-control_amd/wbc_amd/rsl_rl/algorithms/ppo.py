"""PPO with two-channel (leg / arm) advantages, Advantage Mixing, Regularized Online Adaptation
(privileged-latent regulariser + DAgger step for the history encoder) and a minimum policy std
(reference rsl_rl/algorithms/ppo.py:39-324; the quirks that decide seed-identical parity are
SURVEY.md section 8a L1-L10). They are all kept by the eager paths (CPU, and the GPU fallback), which
tests/test_ppo_parity.py pins to the reference seed for seed. The fused GPU paths keep L2-L10 and drop
the random-stream bookkeeping (L1: update() / update_dagger() sample-and-discard one [mb, 18] normal per
minibatch; the rollout noise is drawn per rollout, not per step): a device generator's stream is not the
reference's CPU stream in the first place, and nothing computed depends on the discarded draws.

Multi-GPU (not in the reference): with `dist_group` set, every rank owns a shard of the envs and a
replica of the networks; gradients are flattened into ONE bucket and all-reduced (RCCL over xGMI on
ROCm, gloo in the CPU tests) once per minibatch between backward() and clip_grad_norm_, so the
clip acts on the reduced gradient and all ranks take the identical step.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.optim as optim

from ... import collectives
from ..modules import ActorCritic
from ..storage import RolloutStorage


class PPO:
    actor_critic: ActorCritic

    def __init__(self, actor_critic, num_learning_epochs=1, num_mini_batches=1, clip_param=0.2, gamma=0.998, lam=0.95,
                 value_loss_coef=1.0, entropy_coef=0.0, learning_rate=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True,
                 schedule="fixed", desired_kl=0.01, device="cpu", mixing_schedule=[0.5, 2000, 4000], torque_supervision=True,
                 torque_supervision_schedule=[0.1, 1000, 1000], adaptive_arm_gains=True, min_policy_std=None,
                 dagger_update_freq=20, priv_reg_coef_schedual=[0, 0, 0], dist_group=None):
        self.device = device
        self.desired_kl, self.schedule, self.learning_rate = desired_kl, schedule, learning_rate
        self.actor_critic = actor_critic
        self.actor_critic.to(self.device)
        self.storage = None
        self.optimizer = optim.Adam(self.actor_critic.parameters(), lr=learning_rate)
        self.transition = RolloutStorage.Transition()
        self.hist_encoder_optimizer = optim.Adam(self.actor_critic.actor.history_encoder.parameters(), lr=learning_rate)
        self.priv_reg_coef_schedual = priv_reg_coef_schedual
        self.clip_param, self.num_learning_epochs, self.num_mini_batches = clip_param, num_learning_epochs, num_mini_batches
        self.value_loss_coef, self.entropy_coef = value_loss_coef, entropy_coef
        self.gamma, self.lam, self.max_grad_norm = gamma, lam, max_grad_norm
        self.use_clipped_value_loss = use_clipped_value_loss
        self.min_policy_std = torch.tensor(min_policy_std, device=self.device)
        self.mixing_schedule = mixing_schedule
        self.torque_supervision, self.torque_supervision_schedule = torque_supervision, torque_supervision_schedule
        self.adaptive_arm_gains = adaptive_arm_gains
        self.counter = 0
        self.arm_fk = self.arm_fk_adaptive_gains if adaptive_arm_gains else self.arm_fk_fixed_gains
        self.fused_rollout = True        # use the fused HIP inference kernel in act() where it applies
        self._eps_all = None
        self.fused_update = True         # use the fused HIP minibatch kernels in update() where they apply
        self._fused = None
        self.dist_group = dist_group
        self.world_size = torch.distributed.get_world_size(dist_group) if dist_group is not None else 1
        self._buckets = {}

    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape):
        self.storage = RolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape, self.device)
        self.storage.dist_group = self.dist_group

    def test_mode(self):
        self.actor_critic.eval()

    def train_mode(self):
        self.actor_critic.train()

    # ---- rollout side ----------------------------------------------------------------------
    def act(self, obs, critic_obs, hist_encoding=False, side_job=None):
        """`side_job` (sim.SideJob, optional): work the policy launch carries along (WidowGo1.take_stats_job); executed stand-alone
        when the fused inference kernel is not the path taken."""
        tr, ac = self.transition, self.actor_critic
        if (self.fused_rollout and critic_obs is obs and not torch.is_grad_enabled() and ac.fused_act_supported(obs)
                and (not hist_encoding or ac.actor._fused_hist_supported(obs))):
            # one HIP launch for actor + critic + sample + log-prob (csrc/wbc_policy_kernel.hip)
            if self.storage is not None and self.storage.step == 0:
                ac.mark_params_changed()      # start of a rollout: re-pack once, whoever touched the weights since
            st, out = self.storage, None
            if st is not None and st.step < st.num_transitions_per_env and st.actions.is_cuda and st.actions.shape[1:] == (obs.shape[0], 18):
                if st.step == 0 or self._eps_all is None or self._eps_all.shape != st.actions.shape or self._eps_all.device != obs.device:
                    self._eps_all = torch.randn(st.actions.shape, device=obs.device)      # the rollout's noise in one launch
                eps = self._eps_all[st.step]
            else:
                eps = torch.randn(obs.shape[0], 18, device=obs.device)
            if (st is not None and st.step < st.num_transitions_per_env and st.actions.is_cuda and st.actions.shape[1:] == (obs.shape[0], 18)):
                i = st.step                       # write straight into this step's storage slots
                out = (st.actions[i], st.mu[i], st.actions_log_prob[i], st.values[i])
            latent = ac.actor.infer_hist_latent(obs) if hist_encoding else None      # student rollouts (DAgger iterations)
            tr.actions, tr.action_mean, tr.actions_log_prob, tr.values = ac.fused_act(obs, eps, out, latent, side_job)
            if out is not None:                   # std is constant over a rollout: fill the storage's sigma slab once, not per step
                if st.step == 0:
                    st.sigma.copy_(ac.std.detach().reshape(1, 1, -1).expand_as(st.sigma))
                tr.action_sigma = st.sigma[st.step]
            else:
                tr.action_sigma = ac.std.detach().expand_as(tr.action_mean)
        else:
            if side_job is not None:
                side_job.run(torch.cuda.current_stream(obs.device).cuda_stream)
            tr.actions = ac.act(obs, hist_encoding).detach()
            tr.values = ac.evaluate(critic_obs).detach()
            tr.actions_log_prob = ac.get_actions_log_prob(tr.actions).detach()
            tr.action_mean = ac.action_mean.detach()
            tr.action_sigma = ac.action_std.detach()
        # The env may hand out views it overwrites in place on the next step (WidowGo1.obs_buf is one):
        # park the acting observation in its storage slot now instead of at process_env_step time.
        st = self.storage
        if st is not None and st.step < st.num_transitions_per_env:
            obs_slot = st.observations[st.step]
            if obs.data_ptr() != obs_slot.data_ptr():         # (already there when the env wrote into the slot: next_observation_slot)
                obs_slot.copy_(obs)
            if st.privileged_observations is not None:
                st.privileged_observations[st.step].copy_(critic_obs)
                critic_slot = st.privileged_observations[st.step]
            else:
                critic_slot = obs_slot
            tr.observations, tr.critic_observations = obs_slot, critic_slot
        else:
            tr.observations, tr.critic_observations = obs, critic_obs
        return tr.actions

    def next_observation_slot(self):
        """Storage slot the observation produced by the coming env.step() will be parked in by the next act(), or None
        (last step of the rollout: slot 0 still holds this rollout's first observation, which update() reads)."""
        st = self.storage
        if st is None or st.privileged_observations is not None or not st.observations.is_cuda:
            return None
        i = st.step + 1
        return st.observations[i] if i < st.num_transitions_per_env else None

    def rollout_slots(self):
        """(values, gamma, rewards slot, dones slot) of the transition act() has just started, for an env that can fill them
        inside its step (WidowGo1.set_rollout_output), or None."""
        st, tr = self.storage, self.transition
        if (st is None or st.step >= st.num_transitions_per_env or not st.rewards.is_cuda or tr.values is None or not tr.values.is_cuda
                or tr.values.dtype != torch.float32 or not tr.values.is_contiguous() or tr.values.shape != (st.num_envs, 2)):
            return None
        return tr.values, self.gamma, st.rewards[st.step], st.dones[st.step]

    def _process_env_step_fused(self, rewards, arm_rewards, dones, infos):
        """rewards (+ time-out bootstrap) and dones of this step into their storage slots with one launch."""
        st, tr = self.storage, self.transition
        if not (self.fused_rollout and st is not None and st.step < st.num_transitions_per_env and st.rewards.is_cuda
                and rewards.is_cuda and rewards.dtype == torch.float32 and arm_rewards.dtype == torch.float32
                and dones.dtype == torch.int64 and rewards.is_contiguous() and arm_rewards.is_contiguous() and dones.is_contiguous()
                and tr.values is not None and tr.values.is_cuda and tr.values.is_contiguous() and tr.values.dtype == torch.float32
                and tr.values.shape == (rewards.shape[0], 2)):
            return False
        if infos.get("rollout_stored") is not None and infos["rollout_stored"] == st.rewards[st.step].data_ptr():
            tr.rewards, tr.dones = st.rewards[st.step], st.dones[st.step]      # the env's step has already filled the slots
            return True
        to = infos.get("time_outs")
        if to is not None:
            if not (to.is_cuda and to.is_contiguous() and to.dtype in (torch.bool, torch.uint8) and to.shape == rewards.shape):
                return False
        from ...native import check, lib
        i = st.step
        check(lib().wbc_rollout_store(rewards.data_ptr(), arm_rewards.data_ptr(), dones.data_ptr(), to.data_ptr() if to is not None else None,
                                      tr.values.data_ptr(), float(self.gamma), st.rewards[i].data_ptr(), st.dones[i].data_ptr(),
                                      rewards.shape[0], torch.cuda.current_stream(rewards.device).cuda_stream), "wbc_rollout_store")
        tr.rewards, tr.dones = st.rewards[i], st.dones[i]
        return True

    def process_env_step(self, rewards, arm_rewards, dones, infos):
        tr = self.transition
        if not self._process_env_step_fused(rewards, arm_rewards, dones, infos):
            tr.rewards = torch.stack([rewards.clone(), arm_rewards.clone()], dim=-1)
            tr.dones = dones
            if "time_outs" in infos:       # bootstrap both channels on time-outs (PPO:133-134)
                tr.rewards += self.gamma * torch.squeeze(tr.values * infos["time_outs"].unsqueeze(1).to(self.device), 1)
        supervised = "target_arm_torques" in infos
        if supervised:
            tr.target_arm_torques = infos["target_arm_torques"].detach()
            tr.current_arm_dof_pos = infos["current_arm_dof_pos"].detach()
            tr.current_arm_dof_vel = infos["current_arm_dof_vel"].detach()
        self.storage.add_transitions(tr, torque_supervision=supervised)
        tr.clear()
        self.actor_critic.reset(dones)

    def compute_returns(self, last_critic_obs, side_job=None):
        ac = self.actor_critic
        if self.fused_rollout and not torch.is_grad_enabled() and ac.fused_act_supported(last_critic_obs):
            # (the weight pack follows ac.param_version: update() / update_dagger() / load_state_dict bump it; nothing steps the
            # weights between the rollout's first act() and here, so no re-pack is forced)
            last_values = ac.fused_act(last_critic_obs, side_job=side_job)[3]     # the critic half of the inference kernel (one launch)
        else:
            if side_job is not None:
                side_job.run(torch.cuda.current_stream(last_critic_obs.device).cuda_stream)
            last_values = ac.evaluate(last_critic_obs).detach()
        self.storage.compute_returns(last_values, self.gamma, self.lam)

    # ---- gradient exchange -----------------------------------------------------------------
    def warm_up_collectives(self):
        """RCCL sets up its channels / loads its kernels on a communicator's first collectives (measured: the first PPO
        update with a process group took 104 ms instead of 13): do that at construction, not inside a timed iteration.
        One all-reduce per payload size the learner uses: the flat policy gradient, the history encoder's, a few scalars."""
        if self.dist_group is None:
            return
        dev = next(self.actor_critic.parameters()).device
        sizes = {sum(p.numel() for p in self.actor_critic.parameters()),
                 sum(p.numel() for p in self.actor_critic.actor.history_encoder.parameters()), 3, 1}
        for n in sorted(sizes):
            t = torch.zeros(n, device=dev)
            collectives.all_reduce(t, self.dist_group)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    def _allreduce_grads(self, params):
        """One flat bucket, one all-reduce, mean over ranks."""
        if self.dist_group is None:
            return
        params = [p for p in params if p.grad is not None]
        key = tuple(id(p) for p in params)
        if key not in self._buckets:
            n = sum(p.numel() for p in params)
            self._buckets[key] = torch.empty(n, dtype=params[0].dtype, device=params[0].device)
        flat = self._buckets[key]
        off = 0
        for p in params:
            flat[off:off + p.numel()].copy_(p.grad.reshape(-1))
            off += p.numel()
        collectives.all_reduce(flat, self.dist_group)
        flat.div_(self.world_size)
        off = 0
        for p in params:
            p.grad.copy_(flat[off:off + p.numel()].view_as(p.grad))
            off += p.numel()

    # ---- learner side ----------------------------------------------------------------------
    def _generator(self):
        if self.actor_critic.is_recurrent:
            raise NotImplementedError("recurrent policies are unreachable in the reference as well (SURVEY.md inventory #24)")
        return self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs)

    # ---- fused update (csrc/wbc_ppo_kernel.hip) -------------------------------------------------
    def _fused_update_supported(self):
        st, ac = self.storage, self.actor_critic
        return (self.fused_update and st.observations.is_cuda and st.privileged_observations is None and not ac.is_recurrent
                and not self.torque_supervision and not (self.desired_kl is not None and self.schedule == "adaptive")
                and ac.fused_act_supported(st.observations[0]))

    def _update_fused(self):
        """update() with each minibatch's forward, losses and backward in the HIP kernels; the optimiser, the
        gradient clip and (multi-GPU) the all-reduce stay in torch and act on one flat gradient buffer."""
        import ctypes as C
        from ...native import check, lib
        L, st, ac, dev = lib(), self.storage, self.actor_critic, self.storage.observations.device
        T, N = st.num_transitions_per_env, st.num_envs
        batch = T * N
        mb = batch // self.num_mini_batches
        params = ac.fused_params()
        if self._fused is None or self._fused["mb"] != mb:
            ng = L.wbc_ppo_grad_floats()
            grad = torch.zeros(ng, device=dev)
            off = 0
            for p in params:                       # parameter gradients are views of the flat buffer
                p.grad = grad[off:off + p.numel()].view_as(p)
                off += p.numel()
            assert off == ng - 3
            self._fused = dict(mb=mb, grad=grad, nparam=off, ws=torch.empty(L.wbc_ppo_workspace_floats(mb), device=dev),
                               hist=torch.empty(batch, 20, device=dev), m=torch.zeros(off, device=dev), v=torch.zeros(off, device=dev),
                               adam_ws=torch.empty(int(L.wbc_ppo_clip_adam_workspace_floats()), device=dev), sq_off=int(L.wbc_ppo_sq_partials_offset(mb)))
        F = self._fused
        for p, g in zip(params, torch.split(F["grad"][:F["nparam"]], [p.numel() for p in params])):
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g.view_as(p)
        for p in ac.actor.history_encoder.parameters():
            p.grad = None                          # untouched by update() (PPO:175-176)
        table = ac.fused_param_table()
        obs = st.observations.view(batch, -1)
        with torch.inference_mode():               # the regulariser's target: history latent of every stored row
            if ac.actor._fused_hist_supported(obs):
                F["hist"] = ac.actor.infer_hist_latent(obs)          # one launch (csrc/wbc_hist_kernel.hip)
            else:
                for s0 in range(0, batch, 32768):
                    F["hist"][s0:s0 + 32768] = ac.actor.infer_hist_latent(obs[s0:s0 + 32768])
        flat = lambda x: x.view(batch, -1)         # noqa: E731
        actions, values, adv, returns, logp = (flat(x) for x in (st.actions, st.values, st.advantages, st.returns, st.actions_log_prob))
        value_mixing_ratio = self.get_value_mixing_ratio()
        s = self.priv_reg_coef_schedual
        priv_reg_coef = min(max((self.counter - s[2]), 0) / s[3], 1) * (s[1] - s[0]) + s[0]
        indices = torch.randperm(self.num_mini_batches * mb, requires_grad=False, device=dev)
        sums = torch.zeros(3, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        vcoef = self.value_loss_coef
        steps = self._bind_adam_state(params, F)   # once per update: nothing re-binds the optimiser's state inside it
        packed = False     # the workspace's weight streams are current: packed by the update's first minibatch call, kept by the fused Adam
        for _ in range(self.num_learning_epochs):
            for i in range(self.num_mini_batches):
                idx = indices[i * mb:(i + 1) * mb]
                check((L.wbc_ppo_minibatch_grad_packed if packed else L.wbc_ppo_minibatch_grad)(table, obs.data_ptr(), actions.data_ptr(), values.data_ptr(), adv.data_ptr(),
                                               returns.data_ptr(), logp.data_ptr(), F["hist"].data_ptr(), idx.data_ptr(), mb,
                                               float(self.clip_param), float(vcoef), float(value_mixing_ratio), float(priv_reg_coef),
                                               int(self.use_clipped_value_loss), F["ws"].data_ptr(), F["grad"].data_ptr(), sums.data_ptr(), stream),
                      "wbc_ppo_minibatch_grad")            # (the three loss sums are accumulated into `sums` by the reduction kernel)
                grad_is_fresh = True                   # still exactly what the kernels wrote: their partial sums of squares give the norm
                if self.entropy_coef != 0.0:       # -coef * entropy.mean(): d/d sigma_j of (1/2) sum_j log sigma_j
                    ac.std.grad.sub_(self.entropy_coef * 0.5 / ac.std.detach())
                    grad_is_fresh = False
                if self.dist_group is not None:    # ONE flat bucket, SUM over ranks; the 1/world_size mean is folded into the clip + Adam kernel
                    collectives.all_reduce(F["grad"][:F["nparam"]], self.dist_group)
                    grad_is_fresh = False
                if steps is None:                  # optimiser options the fused kernel does not cover
                    if self.world_size > 1:
                        F["grad"][:F["nparam"]].div_(self.world_size)
                    nn.utils.clip_grad_norm_(params, self.max_grad_norm)
                    self.optimizer.step()
                    packed = False
                else:                              # clip_grad_norm_ + Adam.step in two launches on the flat buffers
                    g0 = self.optimizer.param_groups[0]
                    t = float(steps[0]) + 1.0
                    b1, b2 = g0["betas"]
                    check(L.wbc_ppo_clip_adam_packed(table, F["grad"].data_ptr(), F["m"].data_ptr(), F["v"].data_ptr(),
                                                     float(self.max_grad_norm), b1, b2, g0["eps"], g0["lr"] / (1.0 - b1 ** t),
                                                     (1.0 - b2 ** t) ** 0.5, 1.0 / self.world_size,
                                                     F["ws"].data_ptr() + 4 * F["sq_off"] if grad_is_fresh else None,
                                                     F["adam_ws"].data_ptr(), F["ws"].data_ptr(), mb, stream), "wbc_ppo_clip_adam_packed")
                    packed = True
                    torch._foreach_add_(steps, 1.0)
        num_updates = self.num_learning_epochs * self.num_mini_batches
        surr, vls, preg = (sums / num_updates).tolist()
        self.storage.clear()
        self.update_counter()
        self.enforce_min_std()
        return (vls / (2 * mb), surr / (2 * mb), 0.0, value_mixing_ratio, 0, preg / mb, priv_reg_coef)

    def _bind_adam_state(self, params, F, opt=None):
        """Make the optimiser's (default: self.optimizer) Adam moments of the fused parameters views of the flat buffers
        F['m'], F['v'] (so that state_dict()/load_state_dict() and the eager step keep working on the same memory). Returns
        the list of the parameters' step counters, or None if the optimiser is not a plain Adam the fused kernel reproduces."""
        opt = self.optimizer if opt is None else opt
        if type(opt) is not optim.Adam or len(opt.param_groups) != 1:
            return None
        g0 = opt.param_groups[0]
        if (g0.get("weight_decay", 0) != 0 or g0.get("amsgrad", False) or g0.get("maximize", False) or g0.get("capturable", False)
                or g0.get("fused", False) or isinstance(g0["lr"], torch.Tensor)):
            return None
        steps, off = [], 0
        for p in params:
            n = p.numel()
            mv, vv = F["m"][off:off + n].view_as(p), F["v"][off:off + n].view_as(p)
            st = opt.state[p]
            if len(st) == 0:                      # as Adam._init_group
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"], st["exp_avg_sq"] = mv.zero_(), vv.zero_()
            elif st["exp_avg"].data_ptr() != mv.data_ptr() or st["exp_avg_sq"].data_ptr() != vv.data_ptr():
                mv.copy_(st["exp_avg"]); vv.copy_(st["exp_avg_sq"])      # eager steps / load_state_dict happened in between
                st["exp_avg"], st["exp_avg_sq"] = mv, vv
            if st["step"].is_cuda:                 # a checkpoint loaded with map_location=cuda leaves the counters on the device
                st["step"] = st["step"].detach().cpu()     # (non-capturable Adam): keep them on the host, where the fused path reads them
            steps.append(st["step"])
            off += n
        if any(float(x) != float(steps[0]) for x in steps):
            return None
        return steps

    def update(self):
        if self._fused_update_supported():
            return self._update_fused()
        ac = self.actor_critic
        sums = torch.zeros(4, device=self.device)      # value, surrogate, arm-torque, priv-reg loss accumulators
        value_mixing_ratio = self.get_value_mixing_ratio()
        torque_supervision_weight = self.get_torque_supervision_weight() if self.torque_supervision else 0
        priv_reg_coef = 0.0
        for (obs_b, critic_obs_b, actions_b, target_values_b, adv_b, returns_b, old_logp_b, old_mu_b, old_sigma_b,
             target_arm_torques, cur_arm_pos, cur_arm_vel, hid_b, masks_b) in self._generator():
            ac.act(obs_b, hist_encoding=False, masks=masks_b, hidden_states=hid_b[0])   # samples and discards (quirk L1)
            logp_b = ac.get_actions_log_prob(actions_b)
            value_b = ac.evaluate(critic_obs_b, masks=masks_b, hidden_states=hid_b[1])
            mu_b, sigma_b, entropy_b = ac.action_mean, ac.action_std, ac.entropy

            # Regularized Online Adaptation: pull the privileged latent towards the history latent
            priv_latent = ac.actor.infer_priv_latent(obs_b)
            with torch.inference_mode():
                hist_latent = ac.actor.infer_hist_latent(obs_b)
            priv_reg_loss = (priv_latent - hist_latent.detach()).norm(p=2, dim=1).mean()
            s = self.priv_reg_coef_schedual
            stage = min(max((self.counter - s[2]), 0) / s[3], 1)
            priv_reg_coef = stage * (s[1] - s[0]) + s[0]

            if self.desired_kl is not None and self.schedule == "adaptive":
                with torch.inference_mode():
                    kl = torch.sum(torch.log(sigma_b / old_sigma_b + 1.e-5) +
                                   (torch.square(old_sigma_b) + torch.square(old_mu_b - mu_b)) / (2.0 * torch.square(sigma_b)) - 0.5,
                                   axis=-1)
                    kl_mean = torch.mean(kl)
                    if self.dist_group is not None:
                        collectives.all_reduce(kl_mean, self.dist_group)
                        kl_mean /= self.world_size
                    if kl_mean > self.desired_kl * 2.0:
                        self.learning_rate = max(1e-5, self.learning_rate / 1.5)
                    elif kl_mean < self.desired_kl / 2.0 and kl_mean > 0.0:
                        self.learning_rate = min(1e-2, self.learning_rate * 1.5)
                    for group in self.optimizer.param_groups:
                        group["lr"] = self.learning_rate

            # Advantage Mixing: each channel's surrogate sees its own advantage plus beta x the other's
            mixed = torch.stack([adv_b[..., 0] + value_mixing_ratio * adv_b[..., 1],
                                 adv_b[..., 1] + value_mixing_ratio * adv_b[..., 0]], dim=-1)
            ratio = torch.exp(logp_b - old_logp_b)
            surrogate_loss = torch.max(-mixed * ratio, -mixed * torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param)).mean()

            if self.use_clipped_value_loss:
                value_clipped = target_values_b + (value_b - target_values_b).clamp(-self.clip_param, self.clip_param)
                value_loss = torch.max((value_b - returns_b).pow(2), (value_clipped - returns_b).pow(2)).mean()
            else:
                value_loss = (returns_b - value_b).pow(2).mean()

            loss = surrogate_loss + self.value_loss_coef * value_loss - self.entropy_coef * entropy_b.mean() + priv_reg_coef * priv_reg_loss

            if self.torque_supervision:
                mean_actions = ac.act_inference(obs_b)
                if self.adaptive_arm_gains:
                    target_arm_dof_pos, delta_arm_p_gains = mean_actions[:, 12:-6], mean_actions[:, -6:]
                else:
                    target_arm_dof_pos, delta_arm_p_gains = mean_actions[:, -6:], None
                arm_torques = self.arm_fk(delta_arm_p_gains, target_arm_dof_pos, cur_arm_pos, cur_arm_vel)
                arm_torques_loss = (arm_torques - target_arm_torques).pow(2).mean()
                torque_supervision_weight = self.get_torque_supervision_weight()
                loss = loss + arm_torques_loss * torque_supervision_weight
                sums[2] += arm_torques_loss.detach()

            self.optimizer.zero_grad()
            loss.backward()
            self._allreduce_grads(list(ac.parameters()))
            nn.utils.clip_grad_norm_(ac.parameters(), self.max_grad_norm)
            self.optimizer.step()
            sums[0] += value_loss.detach()
            sums[1] += surrogate_loss.detach()
            sums[3] += priv_reg_loss.detach()

        num_updates = self.num_learning_epochs * self.num_mini_batches
        mean_value_loss, mean_surrogate_loss, mean_arm_torques_loss, mean_priv_reg_loss = (sums / num_updates).tolist()
        self.storage.clear()
        self.update_counter()
        self.enforce_min_std()
        return (mean_value_loss, mean_surrogate_loss, mean_arm_torques_loss, value_mixing_ratio, torque_supervision_weight,
                mean_priv_reg_loss, priv_reg_coef)

    def _fused_dagger_supported(self):
        st, ac = self.storage, self.actor_critic
        return (self.fused_update and st.observations.is_cuda and st.privileged_observations is None and not ac.is_recurrent
                and ac.actor._fused_hist_supported(st.observations[0]) and ac.fused_act_supported(st.observations[0]))

    def _update_dagger_fused(self):
        """update_dagger() with each minibatch's history-encoder forward, loss, backward and weight gradients in ONE HIP
        launch (csrc/wbc_hist_train_kernel.hip) + a fixed-order reduction + clip/Adam on flat buffers; the privileged
        latents of all stored rows (constant: only the history encoder moves) come from one wbc_priv_latent launch."""
        import ctypes as C
        from ...native import check, lib
        L, st, ac, dev = lib(), self.storage, self.actor_critic, self.storage.observations.device
        batch = st.num_transitions_per_env * st.num_envs
        mb = batch // self.num_mini_batches
        he, pe = ac.actor.history_encoder, ac.actor.priv_encoder
        hp = [he.encoder[0].weight, he.encoder[0].bias, he.conv_layers[0].weight, he.conv_layers[0].bias,
              he.conv_layers[2].weight, he.conv_layers[2].bias, he.linear_output[0].weight, he.linear_output[0].bias]
        pp = [pe[0].weight, pe[0].bias, pe[2].weight, pe[2].bias]
        for p in hp + pp:
            assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
        F = self.__dict__.get("_fused_hist")
        if F is None or F["batch"] != batch:
            ng = L.wbc_hist_train_grad_floats()
            F = dict(batch=batch, ng=ng, grad=torch.zeros(ng, device=dev), ws=torch.empty(L.wbc_hist_train_workspace_floats(), device=dev),
                     m=torch.zeros(ng - 1, device=dev), v=torch.zeros(ng - 1, device=dev), priv=torch.empty(batch, 20, device=dev))
            self._fused_hist = F
        nparam = F["ng"] - 1
        for p, g in zip(hp, torch.split(F["grad"][:nparam], [p.numel() for p in hp])):
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g.view_as(p)               # parameter gradients are views of the flat buffer
        table = (C.c_void_p * 8)(*[p.data_ptr() for p in hp])
        ptable = (C.c_void_p * 4)(*[p.data_ptr() for p in pp])
        obs = st.observations.view(batch, -1)
        stream = torch.cuda.current_stream(dev).cuda_stream
        check(L.wbc_priv_latent(ptable, obs.data_ptr(), F["priv"].data_ptr(), batch, stream), "wbc_priv_latent")
        indices = torch.randperm(self.num_mini_batches * mb, requires_grad=False, device=dev)      # one per update (RS:163)
        total = torch.zeros((), device=dev)
        opt = self.hist_encoder_optimizer
        steps = self._bind_adam_state(hp, F, opt)
        for _ in range(self.num_learning_epochs):
            for i in range(self.num_mini_batches):
                idx = indices[i * mb:(i + 1) * mb]
                check(L.wbc_hist_train_grad(table, obs.data_ptr(), F["priv"].data_ptr(), idx.data_ptr(), mb, F["ws"].data_ptr(),
                                            F["grad"].data_ptr(), stream), "wbc_hist_train_grad")
                total += F["grad"][nparam]
                reduced = 0
                if self.dist_group is not None:    # one flat 23 kB bucket, SUM; the mean is folded into the clip + Adam kernel
                    collectives.all_reduce(F["grad"][:nparam], self.dist_group)
                    reduced = 1
                if steps is None:
                    if self.world_size > 1:
                        F["grad"][:nparam].div_(self.world_size)
                    nn.utils.clip_grad_norm_(hp, self.max_grad_norm)
                    opt.step()
                else:
                    g0 = opt.param_groups[0]
                    t = float(steps[0]) + 1.0
                    b1, b2 = g0["betas"]
                    check(L.wbc_hist_clip_adam(table, F["grad"].data_ptr(), F["m"].data_ptr(), F["v"].data_ptr(), float(self.max_grad_norm),
                                               b1, b2, g0["eps"], g0["lr"] / (1.0 - b1 ** t), (1.0 - b2 ** t) ** 0.5, 1.0 / self.world_size,
                                               reduced, F["ws"].data_ptr(), stream), "wbc_hist_clip_adam")
                    torch._foreach_add_(steps, 1.0)
        num_updates = self.num_learning_epochs * self.num_mini_batches
        self.storage.clear()
        self.update_counter()
        ac.mark_params_changed()
        return (total / (num_updates * mb)).item()

    def update_dagger(self):
        if self._fused_dagger_supported():
            return self._update_dagger_fused()
        ac = self.actor_critic
        total = torch.zeros((), device=self.device)
        hist_params = list(ac.actor.history_encoder.parameters())
        if self.fused_rollout and self.storage.observations.is_cuda and self.storage.privileged_observations is None:
            batches = self._dagger_batches_light()
        else:
            batches = ((obs_b, None) for (obs_b, *_rest) in self._generator())
        for obs_b, priv_latent in batches:
            with torch.inference_mode():
                ac.act(obs_b, hist_encoding=True, masks=None, hidden_states=None)       # samples and discards (quirk L1)
                if priv_latent is None:
                    priv_latent = ac.actor.infer_priv_latent(obs_b)
            hist_latent = ac.actor.infer_hist_latent(obs_b)
            loss = (priv_latent.detach() - hist_latent).norm(p=2, dim=1).mean()
            self.hist_encoder_optimizer.zero_grad()
            loss.backward()
            self._allreduce_grads(hist_params)
            nn.utils.clip_grad_norm_(hist_params, self.max_grad_norm)
            self.hist_encoder_optimizer.step()
            total += loss.detach()
        num_updates = self.num_learning_epochs * self.num_mini_batches
        self.storage.clear()
        self.update_counter()
        ac.mark_params_changed()
        return (total / num_updates).item()

    def _dagger_batches_light(self):
        """The minibatches update_dagger needs -- observations and the (constant: only the history encoder is trained)
        privileged latent -- with the generator's permutation (one randperm per update, RS:163) but without gathering the
        ten rollout tensors it does not read; the privileged latent of every stored row is computed once."""
        st = self.storage
        batch = st.num_envs * st.num_transitions_per_env
        mb = batch // self.num_mini_batches
        indices = torch.randperm(self.num_mini_batches * mb, requires_grad=False, device=st.observations.device)
        obs = st.observations.flatten(0, 1)
        with torch.inference_mode():
            priv_all = self.actor_critic.actor.infer_priv_latent(obs)
        for _ in range(self.num_learning_epochs):
            for i in range(self.num_mini_batches):
                idx = indices[i * mb:(i + 1) * mb]
                yield obs[idx], priv_all[idx]

    def enforce_min_std(self):
        with torch.no_grad():           # in place: kernels hold the parameter's address
            self.actor_critic.std.copy_(torch.max(self.actor_critic.std, self.min_policy_std))
        self.actor_critic.mark_params_changed()

    def update_counter(self):
        self.counter += 1

    def get_value_mixing_ratio(self):
        s = self.mixing_schedule
        return min(max((self.counter - s[1]) / s[2], 0), 1) * s[0]

    def get_torque_supervision_weight(self):
        s = self.torque_supervision_schedule
        return (1 - min(max((self.counter - s[1]) / s[2], 0), 1)) * s[0]

    def set_arm_default_coeffs(self, default_arm_p_gains, default_arm_d_gains, default_arm_dof_pos):
        self.default_arm_p_gains, self.default_arm_d_gains = default_arm_p_gains, default_arm_d_gains
        self.default_arm_dof_pos = default_arm_dof_pos

    def arm_fk_adaptive_gains(self, delta_arm_p_gains, target_arm_dof_pos, current_arm_dof_pos, current_arm_dof_vel):
        p = self.default_arm_p_gains + delta_arm_p_gains
        d = 2 * (p ** 0.5)
        return p * (target_arm_dof_pos + self.default_arm_dof_pos - current_arm_dof_pos) - d * current_arm_dof_vel

    def arm_fk_fixed_gains(self, _, target_arm_dof_pos, current_arm_dof_pos, current_arm_dof_vel):
        return (self.default_arm_p_gains * (target_arm_dof_pos + self.default_arm_dof_pos - current_arm_dof_pos)
                - self.default_arm_d_gains * current_arm_dof_vel)
