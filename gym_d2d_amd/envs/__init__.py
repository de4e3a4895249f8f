"""Environment front-ends.  Attribute access is lazy so that `gym_d2d_amd.simulator` can import
`gym_d2d_amd.envs.env_config` without dragging in d2d_env (which imports the simulator) - the reference has exactly
this cycle and fails on `import gym_d2d.simulator` first."""

__all__ = ['D2DEnv', 'VecD2DEnv']


def __getattr__(name):
    if name == 'D2DEnv':
        from .d2d_env import D2DEnv
        return D2DEnv
    if name == 'VecD2DEnv':
        from .vec_env import VecD2DEnv
        return VecD2DEnv
    raise AttributeError(name)
