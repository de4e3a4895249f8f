"""gym_d2d_amd - MI355X-native implementation of GymD2D's per-step SINR / interference path behind the reference's
gym.make('D2DEnv-v0') / reset() / step() / ObsFunction / RewardFunction / PathLoss surface.

    from gym_d2d_amd import make                 # gym / gymnasium are optional
    env = make('D2DEnv-v0', env_config={...})    # or gym.make(...) when gym is installed

The simulation runs in hand-written HIP kernels (gym_d2d_amd/csrc) through a C ABI (include/d2d_hip.h); there is no
CPU fallback.
"""
from .envs import D2DEnv, VecD2DEnv  # noqa: F401  (lazy)
from .spaces import HAVE_GYM

__all__ = ['D2DEnv', 'VecD2DEnv', 'make', 'ENV_ID']
__version__ = '0.1.0'
ENV_ID = 'D2DEnv-v0'


def make(env_id: str = ENV_ID, **kwargs):
    """gym.make stand-in for boxes without gym: make('D2DEnv-v0', env_config={...})."""
    if env_id != ENV_ID:
        raise ValueError(f'unknown environment id {env_id!r}')
    from .envs.d2d_env import D2DEnv as _Env
    return _Env(**kwargs)


def __getattr__(name):
    if name in ('D2DEnv', 'VecD2DEnv'):
        from . import envs
        return getattr(envs, name)
    raise AttributeError(name)


if HAVE_GYM:                                             # pragma: no cover - gym is not installed on the build boxes
    try:
        from gym.envs.registration import register      # type: ignore
    except Exception:
        from gymnasium.envs.registration import register  # type: ignore
    try:
        register(id=ENV_ID, entry_point='gym_d2d_amd.envs:D2DEnv')
    except Exception:
        pass    # already registered (e.g. by the reference package)
