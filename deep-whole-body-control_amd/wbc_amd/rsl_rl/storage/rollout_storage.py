"""Rollout buffers [T, N, ...] with two-channel (leg, arm) rewards/values, GAE and the
random-permutation minibatch gather (reference rsl_rl/storage/rollout_storage.py:36-205).

On a ROCm device `compute_returns` is two HIP launches (wbc_gae_compute + wbc_gae_normalize,
csrc/wbc_gae_kernel.hip) instead of T sequential eager steps; with `dist_group` set the three
advantage statistics are all-reduced between them, so a sharded learner normalises exactly like a
single-GPU learner over the union of the shards (collective 2). For CPU
tensors (the gloo multi-process tests and the CPU restatement in oracle/) the same recurrence runs
as torch ops.
"""
from __future__ import annotations

import torch

from ... import collectives


class RolloutStorage:
    class Transition:
        _FIELDS = ("observations", "critic_observations", "actions", "rewards", "dones", "values", "actions_log_prob",
                   "action_mean", "action_sigma", "hidden_states", "target_arm_torques", "current_arm_dof_pos",
                   "current_arm_dof_vel")

        def __init__(self):
            for f in self._FIELDS:
                setattr(self, f, None)

        def clear(self):
            self.__init__()

    def __init__(self, num_envs, num_transitions_per_env, obs_shape, privileged_obs_shape, actions_shape, device="cpu"):
        self.device = device
        self.obs_shape, self.privileged_obs_shape, self.actions_shape = obs_shape, privileged_obs_shape, actions_shape
        T, N = num_transitions_per_env, num_envs
        z = lambda *s, **kw: torch.zeros(T, N, *s, device=self.device, **kw)   # noqa: E731
        self.observations = z(*obs_shape)
        self.privileged_observations = z(*privileged_obs_shape) if privileged_obs_shape[0] is not None else None
        self.rewards = z(2)
        self.actions = z(*actions_shape)
        self.dones = z(1, dtype=torch.uint8)
        self.actions_log_prob, self.values, self.returns, self.advantages = z(2), z(2), z(2), z(2)
        self.mu, self.sigma = z(*actions_shape), z(*actions_shape)
        self.target_arm_torques, self.current_arm_dof_pos, self.current_arm_dof_vel = z(6), z(6), z(6)
        self.num_transitions_per_env, self.num_envs = T, N
        self.saved_hidden_states_a = self.saved_hidden_states_c = None
        self.step = 0
        self.dist_group = None          # set by the multi-GPU learner
        self._gae_ws = None

    def add_transitions(self, transition: "RolloutStorage.Transition", torque_supervision):
        if self.step >= self.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        i = self.step
        if transition.observations.data_ptr() != self.observations[i].data_ptr():      # PPO.act may have parked it already
            self.observations[i].copy_(transition.observations)
        if self.privileged_observations is not None and \
                transition.critic_observations.data_ptr() != self.privileged_observations[i].data_ptr():
            self.privileged_observations[i].copy_(transition.critic_observations)
        def put(dst, src):                     # the fused rollout path writes the slots directly: nothing to copy then
            if src.data_ptr() != dst.data_ptr() or src.dtype != dst.dtype:
                dst.copy_(src)
        put(self.actions[i], transition.actions)
        put(self.rewards[i], transition.rewards)
        put(self.dones[i], transition.dones.view(-1, 1))
        put(self.values[i], transition.values)
        put(self.actions_log_prob[i], transition.actions_log_prob)
        put(self.mu[i], transition.action_mean)
        put(self.sigma[i], transition.action_sigma)
        if torque_supervision:
            self.target_arm_torques[i].copy_(transition.target_arm_torques)
            self.current_arm_dof_pos[i].copy_(transition.current_arm_dof_pos)
            self.current_arm_dof_vel[i].copy_(transition.current_arm_dof_vel)
        self.step += 1

    def clear(self):
        self.step = 0

    # ---- GAE -------------------------------------------------------------------------------
    def compute_returns(self, last_values, gamma, lam):
        if self.rewards.is_cuda:
            self._compute_returns_hip(last_values, gamma, lam)
        else:
            self._compute_returns_torch(last_values, gamma, lam)

    def _compute_returns_hip(self, last_values, gamma, lam):
        from ...native import check, lib
        L = lib()
        T, N = self.num_transitions_per_env, self.num_envs
        if self._gae_ws is None:
            self._gae_ws = torch.zeros(L.wbc_gae_workspace_doubles(N), dtype=torch.float64, device=self.rewards.device)
        lv = last_values.detach().to(torch.float32).contiguous()
        stream = torch.cuda.current_stream(self.rewards.device).cuda_stream
        check(L.wbc_gae_compute(self.rewards.data_ptr(), self.values.data_ptr(), self.dones.data_ptr(), lv.data_ptr(),
                                self.returns.data_ptr(), self.advantages.data_ptr(), self._gae_ws.data_ptr(), T, N,
                                float(gamma), float(lam), stream), "wbc_gae_compute")
        if self.dist_group is not None:
            collectives.all_reduce(self._gae_ws[:3], self.dist_group)
        check(L.wbc_gae_normalize(self.advantages.data_ptr(), self._gae_ws.data_ptr(), T * N * 2, stream), "wbc_gae_normalize")

    def _compute_returns_torch(self, last_values, gamma, lam):
        advantage = 0
        for step in reversed(range(self.num_transitions_per_env)):
            next_values = last_values if step == self.num_transitions_per_env - 1 else self.values[step + 1]
            not_terminal = 1.0 - self.dones[step].float()
            delta = self.rewards[step] + not_terminal * gamma * next_values - self.values[step]
            advantage = delta + not_terminal * gamma * lam * advantage
            self.returns[step] = advantage + self.values[step]
        adv = self.returns - self.values
        if self.dist_group is None:
            self.advantages = (adv - adv.mean()) / (adv.std() + 1e-8)
        else:   # pooled mean / unbiased std over all ranks
            a64 = adv.double()
            st = torch.stack([torch.tensor(float(adv.numel()), dtype=torch.float64), a64.sum(), (a64 * a64).sum()])
            torch.distributed.all_reduce(st, group=self.dist_group)
            mean = st[1] / st[0]
            var = torch.clamp((st[2] - st[1] * st[1] / st[0]) / (st[0] - 1.0), min=0.0)
            self.advantages = ((adv - mean.float()) / (var.sqrt().float() + 1e-8))

    def get_statistics(self):
        done = self.dones
        done[-1] = 1
        flat = done.permute(1, 0, 2).reshape(-1, 1)
        idx = torch.cat((flat.new_tensor([-1], dtype=torch.int64), flat.nonzero(as_tuple=False)[:, 0]))
        return (idx[1:] - idx[:-1]).float().mean(), self.rewards.mean()

    # ---- minibatches -----------------------------------------------------------------------
    def mini_batch_generator(self, num_mini_batches, num_epochs=8):
        """One randperm for all epochs (RS:163), contiguous slices of it as minibatches; yields the
        reference's 14-tuple."""
        batch = self.num_envs * self.num_transitions_per_env
        mb = batch // num_mini_batches
        indices = torch.randperm(num_mini_batches * mb, requires_grad=False, device=self.device)
        flat = lambda x: x.flatten(0, 1)   # noqa: E731
        obs = flat(self.observations)
        critic_obs = flat(self.privileged_observations) if self.privileged_observations is not None else obs
        cols = [flat(t) for t in (self.actions, self.values, self.advantages, self.returns, self.actions_log_prob, self.mu,
                                  self.sigma, self.target_arm_torques, self.current_arm_dof_pos, self.current_arm_dof_vel)]
        for _ in range(num_epochs):
            for i in range(num_mini_batches):
                idx = indices[i * mb:(i + 1) * mb]
                o = obs[idx]
                co = o if critic_obs is obs else critic_obs[idx]
                a, v, adv, ret, lp, mu, sg, tq, dp, dv = (c[idx] for c in cols)
                yield o, co, a, v, adv, ret, lp, mu, sg, tq, dp, dv, (None, None), None
