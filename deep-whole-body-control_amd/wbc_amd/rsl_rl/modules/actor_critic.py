"""ActorCritic of the widowGo1 policy: privileged-latent / history-latent actor with separate leg
and arm heads, two-head critic, learnable per-action std.

Same constructor arguments, attribute names, state_dict keys/shapes and
module creation order as the reference (rsl_rl/modules/actor_critic.py:39-353), so that a
reference checkpoint loads unchanged and the same torch seed yields the same initial weights.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.nn as nn
from torch.distributions import Normal

_ACTIVATIONS = {"elu": nn.ELU, "selu": nn.SELU, "relu": nn.ReLU, "crelu": nn.ReLU, "lrelu": nn.LeakyReLU,
                "tanh": nn.Tanh, "sigmoid": nn.Sigmoid}


def get_activation(name: str):
    cls = _ACTIVATIONS.get(name)
    if cls is None:
        print("invalid activation function!")
        return None
    return cls()


class _SplitKLinearFn(torch.autograd.Function):
    """y = x W^T + b whose weight gradient dW = dY^T X (output <= 128x128, reduction length = batch,
    40960 in PPO.update) is computed as SPLIT_K independent partial GEMMs + one sum. The library GEMM
    for that shape uses only out*in/1024 workgroups of a 256-CU chip (measured 106-145 us per layer on
    MI355X; 26 us split 32 ways, same result to fp32 round-off)."""
    SPLIT_K = 32
    MIN_BATCH = 4096

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = gy @ weight
        b, s = x.shape[0], _SplitKLinearFn.SPLIT_K
        while b // s > 2048 and b % (2 * s) == 0:       # longer reductions (history-encoder rows: batch x 10): more splits, ~1-2 k rows each
            s *= 2
        if ctx.needs_input_grad[1]:
            gyc, xc = gy.contiguous(), x.contiguous()
            gw = torch.bmm(gyc.view(s, b // s, -1).transpose(1, 2), xc.view(s, b // s, -1)).sum(0)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum(0)
        return gx, gw, gb


class Linear(nn.Linear):
    """nn.Linear (same parameters, same state_dict keys) that routes large training batches through
    the split-K weight-gradient path."""

    def forward(self, x):
        if (x.is_cuda and torch.is_grad_enabled() and x.dim() == 2 and x.shape[0] >= _SplitKLinearFn.MIN_BATCH
                and x.shape[0] % _SplitKLinearFn.SPLIT_K == 0 and self.weight.requires_grad):
            return _SplitKLinearFn.apply(x, self.weight, self.bias)
        return super().forward(x)


def _stack(sizes: Sequence[int], act: nn.Module, last_act=None, act_after_last=True) -> nn.Sequential:
    """Linear(sizes[0], sizes[1]), act, Linear(...), act, ...; Sequential indices 0,2,4,... are the
    Linear layers, which is what the checkpoint key names encode."""
    layers: List[nn.Module] = []
    n = len(sizes) - 1
    for i in range(n):
        layers.append(Linear(sizes[i], sizes[i + 1]))
        if i < n - 1:
            layers.append(act)
        elif last_act is not None:
            layers.append(last_act)
        elif act_after_last:
            layers.append(act)
    return nn.Sequential(*layers)


class StateHistoryEncoder(nn.Module):
    """Per-step projection + two 1-D convolutions over the proprio history (AC:39-84)."""
    _CONVS = {   # tsteps -> [(in_mult, out_mult, kernel, stride), ...] in units of channel_size
        50: [(3, 2, 8, 4), (2, 1, 5, 1), (1, 1, 5, 1)],
        10: [(3, 2, 4, 2), (2, 1, 2, 1)],
        20: [(3, 2, 6, 2), (2, 1, 4, 2)],
    }

    def __init__(self, activation_fn, input_size, tsteps, output_size, tanh_encoder_output=False):
        super().__init__()
        if tsteps not in self._CONVS:
            raise ValueError("tsteps must be 10, 20 or 50")
        self.activation_fn = activation_fn
        self.tsteps = tsteps
        ch = 10
        self.encoder = nn.Sequential(Linear(input_size, 3 * ch), self.activation_fn)
        conv: List[nn.Module] = []
        for cin, cout, ksz, stride in self._CONVS[tsteps]:
            conv += [nn.Conv1d(cin * ch, cout * ch, kernel_size=ksz, stride=stride), self.activation_fn]
        conv.append(nn.Flatten())
        self.conv_layers = nn.Sequential(*conv)
        self.linear_output = nn.Sequential(Linear(ch * 3, output_size), self.activation_fn)

    @staticmethod
    def _conv1d_as_gemm(x, conv: nn.Conv1d):
        """x [B, L, Cin] (time-major) -> [B, Lout, Cout]. Same arithmetic as nn.Conv1d on the permuted
        input, but as ONE rocBLAS GEMM of M = B*Lout rows: MIOpen's 1-D convolution path launches an
        im2col + GEMM pair per sample at these sizes (measured: 49 % of GPU time of a DAgger iteration)."""
        k, s = conv.kernel_size[0], conv.stride[0]
        patches = x.unfold(1, k, s)                                   # [B, Lout, Cin, k]
        b, lout = patches.shape[0], patches.shape[1]
        w = conv.weight.reshape(conv.out_channels, -1)                # [Cout, Cin*k], (cin, k) order = patches' last dims
        flat = patches.reshape(b * lout, -1)
        if (flat.is_cuda and torch.is_grad_enabled() and flat.shape[0] >= _SplitKLinearFn.MIN_BATCH
                and flat.shape[0] % _SplitKLinearFn.SPLIT_K == 0 and conv.weight.requires_grad):
            return _SplitKLinearFn.apply(flat, w, conv.bias).reshape(b, lout, -1)
        return torch.nn.functional.linear(flat, w, conv.bias).reshape(b, lout, -1)

    def forward(self, obs):          # [B, T, n_proprio]
        b, t = obs.shape[0], self.tsteps
        x = self.encoder(obs.reshape(b * t, -1)).reshape(b, t, -1)    # [B, T, 30], time-major
        for layer in self.conv_layers:
            if isinstance(layer, nn.Conv1d):
                x = self._conv1d_as_gemm(x, layer)
            elif isinstance(layer, nn.Flatten):
                x = x.permute(0, 2, 1).flatten(1)                     # channel-major flatten, as Conv1d output [B, C, L]
            else:
                x = layer(x)
        return self.linear_output(x)


class _Actor(nn.Module):
    def __init__(self, num_prop, hidden, act, leg_dims, arm_dims, n_leg, n_arm, adaptive_arm_gains, gains_scale,
                 num_priv, num_hist, priv_dims):
        super().__init__()
        self.adaptive_arm_gains = adaptive_arm_gains
        self.adaptive_arm_gains_scale = gains_scale
        self.num_arm_actions = n_arm
        self.num_priv, self.num_hist, self.num_prop = num_priv, num_hist, num_prop
        if len(priv_dims) > 0:
            self.priv_encoder = _stack([num_priv] + list(priv_dims), act)
            latent = priv_dims[-1]
        else:
            self.priv_encoder = nn.Identity()
            latent = num_priv
        self.history_encoder = StateHistoryEncoder(act, num_prop, num_hist, latent)
        if len(hidden) > 0:
            self.actor_backbone = _stack([num_prop + latent] + list(hidden), act)
            trunk = hidden[-1]
        else:
            self.actor_backbone = nn.Identity()
            trunk = num_prop + latent
        self.actor_leg_control_head = _stack([trunk] + list(leg_dims) + [n_leg], act, last_act=nn.Tanh())
        self.actor_arm_control_head = _stack([trunk] + list(arm_dims) + [n_arm], act, last_act=nn.Tanh())

    def infer_priv_latent(self, obs):
        return self.priv_encoder(obs[:, self.num_prop: self.num_prop + self.num_priv])

    def infer_hist_latent(self, obs):
        if not torch.is_grad_enabled() and self._fused_hist_supported(obs):
            return self._fused_hist_latent(obs)
        hist = obs[:, -self.num_hist * self.num_prop:]
        return self.history_encoder(hist.view(-1, self.num_hist, self.num_prop))

    def _fused_hist_supported(self, obs):
        """csrc/wbc_hist_kernel.hip covers the shipped encoder (10 x 76 history, ELU) on a ROCm device, no gradient."""
        if not (obs.is_cuda and obs.dtype == torch.float32 and obs.dim() == 2 and obs.shape[1] == 860 and obs.is_contiguous()):
            return False
        ok = self.__dict__.get("_fused_hist_ok")
        if ok is None:
            he = self.history_encoder
            ok = (self.num_hist == 10 and self.num_prop == 76 and self.num_priv == 24 and isinstance(he.activation_fn, nn.ELU)
                  and he.activation_fn.alpha == 1.0 and len(he.conv_layers) == 5
                  and tuple(he.encoder[0].weight.shape) == (30, 76) and tuple(he.conv_layers[0].weight.shape) == (20, 30, 4)
                  and tuple(he.conv_layers[2].weight.shape) == (10, 20, 2) and he.conv_layers[0].stride == (2,)
                  and he.conv_layers[2].stride == (1,) and tuple(he.linear_output[0].weight.shape) == (20, 30))
            self.__dict__["_fused_hist_ok"] = ok
        return ok

    def _fused_hist_latent(self, obs):
        import ctypes as C
        from ...native import check, lib
        he = self.history_encoder
        ps = [he.encoder[0].weight, he.encoder[0].bias, he.conv_layers[0].weight, he.conv_layers[0].bias,
              he.conv_layers[2].weight, he.conv_layers[2].bias, he.linear_output[0].weight, he.linear_output[0].bias]
        for p in ps:
            assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
        table = (C.c_void_p * 8)(*[p.data_ptr() for p in ps])
        out = torch.empty(obs.shape[0], 20, device=obs.device)
        check(lib().wbc_hist_latent(table, obs.data_ptr(), out.data_ptr(), obs.shape[0],
                                    torch.cuda.current_stream(obs.device).cuda_stream), "wbc_hist_latent")
        return out

    def forward(self, obs, hist_encoding=False):
        latent = self.infer_hist_latent(obs) if hist_encoding else self.infer_priv_latent(obs)
        trunk = self.actor_backbone(torch.cat([obs[:, :self.num_prop], latent], dim=1))
        leg = self.actor_leg_control_head(trunk)
        arm = self.actor_arm_control_head(trunk)
        if self.adaptive_arm_gains:
            half = self.num_arm_actions // 2
            arm = torch.cat([arm[:, :half], self.adaptive_arm_gains_scale * arm[:, half:]], dim=-1)
        return torch.cat([leg, arm], dim=-1)


class _Critic(nn.Module):
    def __init__(self, in_dim, hidden, act, leg_dims, arm_dims, num_priv, num_hist, num_prop):
        super().__init__()
        self.num_priv, self.num_hist, self.num_prop = num_priv, num_hist, num_prop
        if len(hidden) > 0:
            self.critic_backbone = _stack([in_dim] + list(hidden), act)
            trunk = hidden[-1]
        else:
            self.critic_backbone = nn.Identity()
            trunk = in_dim
        self.critic_leg_control_head = _stack([trunk] + list(leg_dims) + [1], act, act_after_last=False)
        self.critic_arm_control_head = _stack([trunk] + list(arm_dims) + [1], act, act_after_last=False)

    def forward(self, obs):
        trunk = self.critic_backbone(obs[:, :self.num_prop + self.num_priv])
        return torch.cat([self.critic_leg_control_head(trunk), self.critic_arm_control_head(trunk)], dim=-1)


class ActorCritic(nn.Module):
    is_recurrent = False

    def __init__(self, num_actor_obs, num_critic_obs, num_actions, actor_hidden_dims=[256, 256, 256],
                 critic_hidden_dims=[256, 256, 256], priv_encoder_dims=[64, 20], activation="elu", init_std=1, **kwargs):
        super().__init__()
        self.num_leg_actions = kwargs["num_leg_actions"]
        self.num_arm_actions = kwargs["num_arm_actions"]
        adaptive = kwargs["adaptive_arm_gains"]
        if adaptive:
            self.num_arm_actions *= 2
        num_priv, num_hist, num_prop = kwargs["num_priv"], kwargs["num_hist"], kwargs["num_prop"]
        act = get_activation(activation)
        self.actor = _Actor(num_actor_obs, actor_hidden_dims, act, kwargs["leg_control_head_hidden_dims"],
                            kwargs["arm_control_head_hidden_dims"], self.num_leg_actions, self.num_arm_actions, adaptive,
                            kwargs["adaptive_arm_gains_scale"], num_priv, num_hist, priv_encoder_dims)
        self.critic = _Critic(num_critic_obs + num_priv, critic_hidden_dims, act, kwargs["leg_control_head_hidden_dims"],
                              kwargs["arm_control_head_hidden_dims"], num_priv, num_hist, num_prop)
        if kwargs.get("verbose", False):
            print(f"Actor MLP: {self.actor}")
            print(f"Critic MLP: {self.critic}")
        self.std = nn.Parameter(torch.tensor(init_std))          # [1, num_actions]
        self.distribution = None
        Normal.set_default_validate_args = False

    def reset(self, dones=None):
        pass

    def forward(self):
        raise NotImplementedError

    @property
    def action_mean(self):
        return self.distribution.mean

    @property
    def action_std(self):
        return self.distribution.stddev

    def _split_sum(self, x):
        n = self.num_leg_actions
        return torch.cat([x[:, :n].sum(dim=-1, keepdim=True), x[:, n:].sum(dim=-1, keepdim=True)], dim=-1)

    @property
    def entropy(self):
        return self._split_sum(self.distribution.entropy())

    def update_distribution(self, observations, hist_encoding):
        mean = self.actor(observations, hist_encoding)
        self.distribution = Normal(mean, mean * 0. + self.std)

    def act(self, observations, hist_encoding, **kwargs):
        self.update_distribution(observations, hist_encoding)
        return self.distribution.sample()

    def get_actions_log_prob(self, actions):
        """Log-probability summed separately over the leg and the arm action dims -> [B, 2]."""
        return self._split_sum(self.distribution.log_prob(actions))

    # ---- fused rollout inference (csrc/wbc_policy_kernel.hip) -----------------------------------
    _FUSED_LAYERS = ("actor.priv_encoder.0", "actor.priv_encoder.2", "actor.actor_backbone.0",
                     "actor.actor_leg_control_head.0", "actor.actor_leg_control_head.2", "actor.actor_leg_control_head.4",
                     "actor.actor_arm_control_head.0", "actor.actor_arm_control_head.2", "actor.actor_arm_control_head.4",
                     "critic.critic_backbone.0",
                     "critic.critic_leg_control_head.0", "critic.critic_leg_control_head.2", "critic.critic_leg_control_head.4",
                     "critic.critic_arm_control_head.0", "critic.critic_arm_control_head.2", "critic.critic_arm_control_head.4")

    def fused_act_supported(self, observations) -> bool:
        """The fused kernel covers the shipped widowGo1 architecture on a ROCm device (teacher latent)."""
        if not (observations.is_cuda and observations.dtype == torch.float32 and observations.dim() == 2
                and observations.shape[1] == 860 and observations.is_contiguous()):
            return False
        if getattr(self, "_fused_ok", None) is None:
            sd = dict(self.named_parameters())
            want = {"actor.priv_encoder.0.weight": (64, 24), "actor.priv_encoder.2.weight": (20, 64),
                    "actor.actor_backbone.0.weight": (128, 96), "actor.actor_leg_control_head.0.weight": (128, 128),
                    "actor.actor_leg_control_head.2.weight": (128, 128), "actor.actor_leg_control_head.4.weight": (12, 128),
                    "actor.actor_arm_control_head.4.weight": (6, 128), "critic.critic_backbone.0.weight": (128, 100),
                    "critic.critic_leg_control_head.4.weight": (1, 128), "critic.critic_arm_control_head.2.weight": (128, 128)}
            self._fused_ok = all(k in sd and tuple(sd[k].shape) == v for k, v in want.items()) and \
                isinstance(self.actor.actor_leg_control_head[1], nn.ELU) and not self.actor.adaptive_arm_gains
        return self._fused_ok

    param_version = 0      # bump (mark_params_changed) whenever parameters are modified: the fused kernels cache a packed copy

    def mark_params_changed(self):
        """Bump the version the packed weight copies follow AND tell the library that whatever weight streams it keeps current for
        the update kernels (wbc_ppo_minibatch_grad_packed trusts them by pointer, csrc/wbc_ppo_kernel.hip) are stale: an in-place
        write -- load_state_dict, enforce_min_std, a broadcast -- keeps every pointer, so only the caller can know."""
        self.param_version += 1
        if any(p.is_cuda for p in self.parameters()):
            from ... import native
            native.lib().wbc_ppo_pack_invalidate(None)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.mark_params_changed()
        return out

    def fused_params(self):
        """The 33 parameters the fused kernels read, in struct PolicyParams order."""
        cached = self.__dict__.get("_fused_param_list")
        if cached is None:
            sd = dict(self.named_parameters())
            cached = []
            for name in self._FUSED_LAYERS:
                cached += [sd[name + ".weight"], sd[name + ".bias"]]
            cached.append(self.std)
            self.__dict__["_fused_param_list"] = cached       # Parameter objects survive .to()/load_state_dict (data is swapped)
        return cached

    def fused_param_table(self):
        import ctypes as C
        ps = self.fused_params()
        for p in ps:
            assert p.is_contiguous() and p.dtype == torch.float32 and p.is_cuda
        return (C.c_void_p * len(ps))(*[p.data_ptr() for p in ps])

    def fused_act(self, observations, eps=None, out=None, latent=None, side_job=None):
        """PPO.act's policy side in one launch: returns (actions, mean, log_prob[.,2], values[.,2]).
        `eps` = standard normals [B,18] (drawn by the caller from torch's generator); None acts on the mean.
        `out`: optional 4-tuple of contiguous float32 destination tensors (e.g. rollout-storage slots).
        `latent` [B,20]: student path (hist_encoding=True), replaces the privileged encoder's output.
        `side_job` (sim.SideJob): a small reduction the launch carries as extra workgroups (the env step's episode statistics)."""
        from ...native import check, lib
        table = self.fused_param_table()
        dev = observations.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        if getattr(self, "_wpack", None) is None or self._wpack.device != dev:
            self._wpack = torch.empty(lib().wbc_policy_pack_floats(), device=dev)
            self._wpack_version = -1
        if self._wpack_version != self.param_version:
            check(lib().wbc_policy_pack(table, self._wpack.data_ptr(), stream), "wbc_policy_pack")
            self._wpack_version = self.param_version
        n = observations.shape[0]
        if out is not None:
            actions, mean, logp, values = out
            for t, w in ((actions, 18), (mean, 18), (logp, 2), (values, 2)):
                assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape == (n, w)
        else:
            actions = torch.empty(n, 18, device=dev)
            mean = torch.empty(n, 18, device=dev)
            logp = torch.empty(n, 2, device=dev)
            values = torch.empty(n, 2, device=dev)
        if latent is not None:
            assert latent.is_cuda and latent.dtype == torch.float32 and latent.is_contiguous() and latent.shape == (n, 20)
        import ctypes as C
        carried = side_job is not None and not side_job.done
        check(lib().wbc_policy_act_job(table, self._wpack.data_ptr(), observations.data_ptr(), latent.data_ptr() if latent is not None else None,
                                       eps.data_ptr() if eps is not None else None,
                                       actions.data_ptr(), mean.data_ptr(), logp.data_ptr(), values.data_ptr(), n,
                                       C.byref(side_job.c) if carried else None, stream), "wbc_policy_act")
        if carried:
            side_job.done = True
        return actions, mean, logp, values

    def act_inference(self, observations, hist_encoding=False):
        return self.actor(observations, hist_encoding)

    def evaluate(self, critic_observations, **kwargs):
        return self.critic(critic_observations)
