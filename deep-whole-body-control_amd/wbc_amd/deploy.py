"""Deployment export and the play loop.

What the reference does with a trained checkpoint, restated on this framework's classes:
  * `export_policy_as_jit(actor_critic, path)`      legged_gym/utils/helpers.py:187-197  -> <path>/policy_1.pt
  * `save_actor_state(actor_critic, path, name)`    legged_gym/scripts/play.py:86-93     -> <path>/<name>_actor.pt
  * `trace_actor_and_hist_encoder(...)`             legged_gym/scripts/save_jit.py:197-236
        -> <dir>/<tag>_actor_jit.pt (96 -> 18: [proprio 76, latent 20] -> mean action) and
           <dir>/<tag>_hist_encoder_jit.pt (760 -> 20: flat proprio history -> latent), the two modules the robot runs
  * `play_cfg_overrides(cfg)`                       play.py:45-77 (5 envs, relaxed termination, curricula at their end)
  * `play(env, policy | traced modules, steps)`     play.py:112-124: the student-latent rollout

The exported modules are plain `torch.nn` stacks (`DeployActor`, `DeployHistoryEncoder`) holding copies of the weights: the
training classes route large batches through custom autograd functions and HIP kernels that have no place on the robot.
State-dict keys equal the reference's `Actor` (save_jit.py:82-195), so `<name>_actor.pt` files are interchangeable.
"""
from __future__ import annotations

import copy
import os
from typing import Callable, Optional

import torch
import torch.nn as nn


def _plain(seq: nn.Sequential) -> nn.Sequential:
    """Same layers as plain nn.Linear / nn.Conv1d / activations with copied weights (keys and indices unchanged)."""
    out = []
    for m in seq:
        if isinstance(m, nn.Linear):
            lin = nn.Linear(m.in_features, m.out_features, bias=m.bias is not None)
            lin.load_state_dict(m.state_dict())
            out.append(lin)
        else:
            out.append(copy.deepcopy(m))
    return nn.Sequential(*out)


class DeployHistoryEncoder(nn.Module):
    """StateHistoryEncoder as save_jit.py:39-79 traces it: input [n, T*num_prop] (flat history, oldest first)."""

    def __init__(self, enc):
        super().__init__()
        self.tsteps = enc.tsteps
        self.encoder = _plain(enc.encoder)
        self.conv_layers = _plain(enc.conv_layers)
        self.linear_output = _plain(enc.linear_output)

    def forward(self, obs):
        nd = obs.shape[0]
        projection = self.encoder(obs.reshape([nd * self.tsteps, -1]))
        output = self.conv_layers(projection.reshape([nd, self.tsteps, -1]).permute((0, 2, 1)))
        return self.linear_output(output)


class DeployActor(nn.Module):
    """The deployed actor (save_jit.py:82-195): forward([proprio, latent]) -> mean action; the privileged encoder and
    the history encoder ride along so that the state dict equals the training actor's."""

    def __init__(self, actor):
        super().__init__()
        self.num_prop, self.num_priv, self.num_hist = actor.num_prop, actor.num_priv, actor.num_hist
        self.adaptive_arm_gains = bool(actor.adaptive_arm_gains)
        self.adaptive_arm_gains_scale = float(getattr(actor, "adaptive_arm_gains_scale", 1.0))
        self.num_arm_actions = int(actor.num_arm_actions)
        self.priv_encoder = _plain(actor.priv_encoder) if isinstance(actor.priv_encoder, nn.Sequential) else nn.Identity()
        self.history_encoder = DeployHistoryEncoder(actor.history_encoder)
        self.actor_backbone = _plain(actor.actor_backbone) if isinstance(actor.actor_backbone, nn.Sequential) else nn.Identity()
        self.actor_leg_control_head = _plain(actor.actor_leg_control_head)
        self.actor_arm_control_head = _plain(actor.actor_arm_control_head)

    def forward(self, obs_prop_and_latent):
        trunk = self.actor_backbone(obs_prop_and_latent)
        leg = self.actor_leg_control_head(trunk)
        arm = self.actor_arm_control_head(trunk)
        if self.adaptive_arm_gains:
            half = self.num_arm_actions // 2
            arm = torch.cat([arm[:, :half], self.adaptive_arm_gains_scale * arm[:, half:]], dim=-1)
        return torch.cat([leg, arm], dim=-1)

    def infer_priv_latent(self, obs):
        return self.priv_encoder(obs[:, self.num_prop: self.num_prop + self.num_priv])

    def infer_hist_latent(self, obs):
        return self.history_encoder(obs[:, -self.num_hist * self.num_prop:])


def export_policy_as_jit(actor_critic, path: str) -> str:
    """helpers.py:187-197: script the actor on the CPU -> <path>/policy_1.pt (forward([proprio, latent]))."""
    os.makedirs(path, exist_ok=True)
    out = os.path.join(path, "policy_1.pt")
    model = DeployActor(actor_critic.actor).to("cpu").eval()
    torch.jit.script(model).save(out)
    return out


def save_actor_state(actor_critic, path: str, model_name: str) -> str:
    """play.py:86-93: the actor's state dict alone (input of save_jit.py)."""
    os.makedirs(path, exist_ok=True)
    out = os.path.join(path, model_name + "_actor.pt")
    torch.save(actor_critic.actor.state_dict(), out)
    return out


def trace_actor_and_hist_encoder(actor_critic, out_dir: str, tag: str):
    """save_jit.py:197-236: trace the actor (96 -> 18) and the history encoder (760 -> 20) on the CPU."""
    os.makedirs(out_dir, exist_ok=True)
    actor = DeployActor(actor_critic.actor).to("cpu").eval()
    latent_dim = actor.history_encoder.linear_output[0].out_features
    with torch.no_grad():
        traced_policy = torch.jit.trace(actor, torch.zeros(1, latent_dim + actor.num_prop))
        traced_encoder = torch.jit.trace(actor.history_encoder, torch.zeros(1, actor.num_hist * actor.num_prop))
    p_actor = os.path.join(out_dir, tag + "_actor_jit.pt")
    p_enc = os.path.join(out_dir, tag + "_hist_encoder_jit.pt")
    traced_policy.save(p_actor)
    traced_encoder.save(p_enc)
    return p_actor, p_enc


def play_cfg_overrides(env_cfg):
    """play.py:45-77: the evaluation-time configuration (in place; returns env_cfg)."""
    env_cfg.env.num_envs = min(env_cfg.env.num_envs, 5)
    env_cfg.terrain.tot_rows = 600
    env_cfg.terrain.tot_cols = 600
    env_cfg.termination.r_threshold = 1.0
    env_cfg.termination.p_threshold = 1.0
    env_cfg.termination.z_threshold = 0.0
    env_cfg.domain_rand.randomize_friction = True
    env_cfg.domain_rand.randomize_base_mass = True
    env_cfg.domain_rand.randomize_base_com = True
    env_cfg.domain_rand.randomize_motor = True
    env_cfg.domain_rand.push_robots = True
    env_cfg.commands.lin_vel_x_schedule = [0, 1]
    env_cfg.commands.ang_vel_yaw_schedule = [0, 1]
    env_cfg.commands.tracking_ang_vel_yaw_schedule = [0, 1]
    env_cfg.goal_ee.l_schedule = [0, 1]
    env_cfg.goal_ee.p_schedule = [0, 1]
    env_cfg.goal_ee.y_schedule = [0, 1]
    env_cfg.goal_ee.arm_action_scale_schedule = [0, 1]
    env_cfg.goal_ee.tracking_ee_reward_schedule = [0, 1]
    env_cfg.goal_ee.underground_limit = -0.57
    return env_cfg


def play(env, policy: Optional[Callable] = None, steps: int = 100, traced_actor=None, traced_hist_encoder=None,
         on_step: Optional[Callable] = None):
    """play.py:112-124. With traced modules (use_jit): latent = hist_encoder(obs[:, num_prop+num_priv:]);
    actions = actor(cat(obs[:, :num_prop], latent)). Otherwise actions = policy(obs, hist_encoding=True).
    Returns the per-step mean rewards (leg, arm) as a [steps, 2] tensor."""
    use_jit = traced_actor is not None and traced_hist_encoder is not None
    if not use_jit and policy is None:
        raise ValueError("play() needs either a policy callable or the two traced modules")
    n_prop, n_priv = env.cfg.env.num_proprio, env.cfg.env.num_priv
    env.update_command_curriculum()
    env.reset()
    obs = env.get_observations()
    log = []
    with torch.inference_mode():
        for i in range(steps):
            if use_jit:
                latent = traced_hist_encoder(obs[:, n_prop + n_priv:])
                actions = traced_actor(torch.cat((obs[:, :n_prop], latent), dim=1))
            else:
                actions = policy(obs.detach(), hist_encoding=True)
            obs, _, rews, arm_rews, dones, infos = env.step(actions.detach())
            log.append(torch.stack([rews.mean(), arm_rews.mean()]))
            if on_step is not None:
                on_step(i, obs, rews, arm_rews, dones, infos)
    return torch.stack(log)
