"""Minimal stand-ins for gym.spaces, used only when neither `gym` nor `gymnasium` is importable (they are not
installed on the build or GPU boxes).  Same attribute names as the real classes for the parts GymD2D touches:
Discrete(n).sample()/.n, Box(low, high, shape), Dict[...]."""
from __future__ import annotations

import numpy as np

try:                                   # pragma: no cover - depends on the environment
    from gym import spaces as _real    # type: ignore
    from gym import Space, Env         # type: ignore
    Discrete, Box, Dict = _real.Discrete, _real.Box, _real.Dict
    HAVE_GYM = True
except Exception:                      # pragma: no cover
    try:
        from gymnasium import spaces as _real   # type: ignore
        from gymnasium import Space, Env        # type: ignore
        Discrete, Box, Dict = _real.Discrete, _real.Box, _real.Dict
        HAVE_GYM = True
    except Exception:
        HAVE_GYM = False

if not HAVE_GYM:
    _rng = np.random.default_rng()

    def seed(value) -> None:
        global _rng
        _rng = np.random.default_rng(value)

    class Space:
        def sample(self):
            raise NotImplementedError

        def contains(self, x) -> bool:
            raise NotImplementedError

    class Discrete(Space):
        def __init__(self, n: int) -> None:
            self.n = int(n)
            self.shape = ()
            self.dtype = np.int64

        def sample(self) -> int:
            return int(_rng.integers(0, self.n))

        def contains(self, x) -> bool:
            return isinstance(x, (int, np.integer)) and 0 <= int(x) < self.n

        def __repr__(self) -> str:
            return f'Discrete({self.n})'

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32) -> None:
            self.shape = tuple(shape) if shape is not None else np.shape(low)
            self.dtype = np.dtype(dtype)
            self.low = np.full(self.shape, low, dtype=self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype)

        def sample(self):
            return _rng.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x) -> bool:
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self) -> str:
            return f'Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})'

    class Dict(Space):
        def __init__(self, spaces) -> None:
            self.spaces = dict(spaces)

        def __getitem__(self, key):
            return self.spaces[key]

        def sample(self):
            return {k: s.sample() for k, s in self.spaces.items()}

        def contains(self, x) -> bool:
            return isinstance(x, dict) and all(k in self.spaces and self.spaces[k].contains(v) for k, v in x.items())

        def __repr__(self) -> str:
            return f'Dict({self.spaces})'

    class Env:
        metadata: dict = {}

        def reset(self):
            raise NotImplementedError

        def step(self, action):
            raise NotImplementedError

        def render(self, mode='human'):
            raise NotImplementedError
