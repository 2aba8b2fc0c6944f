"""Learner side of the hot path with the reference's rsl_rl API surface
(rsl_rl/rsl_rl/{modules,storage,algorithms,runners,env}): ActorCritic, RolloutStorage, PPO,
OnPolicyRunner, VecEnv. PyTorch-ROCm carries the networks and the optimiser; GAE and the
rollout step are HIP kernels behind the C-ABI of include/wbc_sim.h."""
