from .actor_critic import ActorCritic, StateHistoryEncoder, get_activation  # noqa: F401
