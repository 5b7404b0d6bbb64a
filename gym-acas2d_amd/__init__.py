"""gym-acas2d_amd -- MI355X-native batched step engine for the ACAS2D environment.

Drop-in for the per-step hot path of Christos-14/gym-ACAS2D (gym_ACAS2D/envs): the reference's
``ACAS2DEnv.reset()/step()`` surface over one hand-written HIP kernel per step.  Import name:
``gym_acas2d_amd`` (the directory name carries a hyphen; ``gym_acas2d_amd.py`` at the repository
root is the import shim).
"""
from .config import ACAS2DConfig, OUTCOME_NAMES                      # noqa: F401
from .sharding import shard_range                                    # noqa: F401
from .spaces import Box                                              # noqa: F401
from . import native, records, render, reset_parity, sharding        # noqa: F401
from .vec_env import ACAS2DVecEnv, LazyInfos                         # noqa: F401
from .env import ACAS2DEnv, GameView, register_with_gym              # noqa: F401
from .policy import SB3ActorPolicy, load_sb3_policy, evaluate_policy, evaluate_policy_fused  # noqa: F401
from .ppo import ActorCritic, FusedUpdate, PPOConfig, PPOTrainer, compute_gae, ppo_loss   # noqa: F401

register_with_gym()


def make(env_id="ACAS2D-v0", **kwargs):
    """gym.make("ACAS2D-v0") equivalent (gym_ACAS2D/__init__.py:3-6)."""
    if env_id != "ACAS2D-v0":
        raise ValueError("unknown env id %r" % (env_id,))
    return ACAS2DEnv(**kwargs)
