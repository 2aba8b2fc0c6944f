"""The environment contract OnPolicyRunner relies on (reference rsl_rl/env/vec_env.py:36-60; the
runner in fact uses the wider surface listed in SURVEY.md section 8b, seam 1)."""
from abc import ABC, abstractmethod
from typing import Optional, Tuple

import torch


class VecEnv(ABC):
    num_envs: int
    num_obs: int
    num_privileged_obs: Optional[int]
    num_actions: int
    max_episode_length: int
    obs_buf: torch.Tensor
    privileged_obs_buf: Optional[torch.Tensor]
    rew_buf: torch.Tensor
    reset_buf: torch.Tensor
    episode_length_buf: torch.Tensor
    extras: dict
    device: torch.device

    @abstractmethod
    def step(self, actions: torch.Tensor) -> Tuple:
        """-> (obs, privileged_obs | None, leg_reward, arm_reward, dones, infos)"""

    @abstractmethod
    def reset(self):
        """-> (obs, privileged_obs | None)"""

    @abstractmethod
    def get_observations(self) -> torch.Tensor:
        ...

    @abstractmethod
    def get_privileged_observations(self) -> Optional[torch.Tensor]:
        ...
