"""A throw-away stand-in for the `gym` package (Env, Space, spaces.{Discrete, Box, Dict}, envs.registration.{register, make}):
neither gym nor gymnasium is installed on the build or GPU boxes, and gym contributes no arithmetic to the path.  Written to
a temporary directory and put on sys.path of a CHILD process, so that `import gym_d2d_amd` there takes its gym branch -
spaces from `gym`, `register(id='D2DEnv-v0', ...)` at import - and `gym.make('D2DEnv-v0', env_config=...)` can be exercised
(reference: gym_d2d/__init__.py:8-11, README.md:59)."""
import textwrap
from pathlib import Path

FILES = {
    'gym/__init__.py': '''
        from . import spaces
        from .spaces import Space
        from .envs.registration import register, make
        class Env:
            metadata = {}
            def reset(self): raise NotImplementedError
            def step(self, action): raise NotImplementedError
            def render(self, mode='human'): raise NotImplementedError
            def close(self): pass
    ''',
    'gym/spaces.py': '''
        import numpy as np
        _rng = np.random.Generator(np.random.PCG64(0))
        class Space:
            pass
        class Discrete(Space):
            def __init__(self, n): self.n = int(n)
            def sample(self): return int(_rng.integers(0, self.n))
            def contains(self, x): return 0 <= int(x) < self.n
        class Box(Space):
            def __init__(self, low, high, shape=None, dtype=np.float32):
                self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype
        class Dict(Space):
            def __init__(self, spaces): self.spaces = dict(spaces)
            def __getitem__(self, k): return self.spaces[k]
    ''',
    'gym/envs/__init__.py': '',
    'gym/envs/registration.py': '''
        import importlib
        registry = {}
        def register(id, entry_point, **kw): registry[id] = entry_point
        def make(id, **kwargs):
            mod, cls = registry[id].split(':')
            return getattr(importlib.import_module(mod), cls)(**kwargs)
    ''',
}

CHILD = '''
import json, sys
sys.path[:0] = [{stub!r}, {root!r}]
import gym
import gym_d2d_amd
from gym.envs import registration
out = {{'have_gym': gym_d2d_amd.spaces.HAVE_GYM, 'entry_point': registration.registry.get('D2DEnv-v0'),
       'space_is_gym': gym_d2d_amd.spaces.Discrete is gym.spaces.Discrete}}
try:
    env = gym.make('D2DEnv-v0', env_config={{'num_rbs': 5, 'num_cues': 3, 'num_due_pairs': 4}})
except Exception as exc:
    out['make'] = type(exc).__name__
else:
    out['make'] = 'ok'
    out['is_gym_env'] = isinstance(env, gym.Env)
    obs = env.reset()
    acts = {{k: env.action_space['due' if k.startswith('due') else 'cue'].sample() for k in obs}}
    obs, rewards, done, info = env.step(acts)
    out['agents'] = len(obs)
    out['obs_width'] = int(next(iter(obs.values())).shape[0])
    out['obs_space'] = list(env.observation_space.shape)
    out['done'] = done
    out['info_keys'] = sorted(next(iter(info.values())))
    env.close()
print(json.dumps(out))
'''


def write_stub(directory) -> str:
    for rel, body in FILES.items():
        p = Path(directory) / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(textwrap.dedent(body))
    return str(directory)


def run_gym_make(tmp_path, root):
    """Child process: stub gym on sys.path, import gym_d2d_amd, gym.make('D2DEnv-v0', ...).  Returns the child's report."""
    import json
    import subprocess
    import sys
    stub = write_stub(Path(tmp_path) / 'gymstub')
    r = subprocess.run([sys.executable, '-c', CHILD.format(stub=stub, root=str(root))], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])
