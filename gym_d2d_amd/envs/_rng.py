"""Counter-based integers for the random actions of reset() (d2d_env.py:54-60), keyed by GLOBAL env index.

The reference samples its initial actions from gym's Discrete spaces (global NumPy state).  A batch that may be
sharded over several GPUs needs a stream that does not depend on how it is sharded: value(seed, episode, global env,
column) is a pure function (splitmix64 finaliser over a Weyl sequence), evaluated with the same 64-bit wrap-around
arithmetic in NumPy (uint64) and in torch (int64 bit patterns), so the NumPy and torch paths of VecD2DEnv - and any
split of the env axis over ranks - produce identical actions.
"""
from __future__ import annotations

import numpy as np

_MASK = (1 << 64) - 1
_M1, _M2, _GOLDEN = 0xBF58476D1CE4E5B9, 0x94D049BB133111EB, 0x9E3779B97F4A7C15


def _mix_int(x: int) -> int:
    x &= _MASK
    x = ((x ^ (x >> 30)) * _M1) & _MASK
    x = ((x ^ (x >> 27)) * _M2) & _MASK
    return x ^ (x >> 31)


def stream_key(seed: int, episode: int) -> int:
    return _mix_int(_mix_int(int(seed)) ^ ((int(episode) + 1) * _GOLDEN))


def _signed(v: int) -> int:
    v &= _MASK
    return v - (1 << 64) if v >> 63 else v


def uniform_ints_numpy(seed: int, episode: int, first_env: int, num_envs: int, num_cols: int, high) -> np.ndarray:
    """int32 [num_envs, num_cols]; column c uniform in [0, high[c]) (modulo bias < 2^-40, irrelevant here)."""
    high = np.broadcast_to(np.asarray(high, dtype=np.uint64), (num_cols,))
    env = (np.arange(num_envs, dtype=np.uint64) + np.uint64(first_env))[:, None]
    col = np.arange(num_cols, dtype=np.uint64)[None, :]
    with np.errstate(over='ignore'):
        x = np.uint64(stream_key(seed, episode)) + (env * np.uint64(num_cols) + col + np.uint64(1)) * np.uint64(_GOLDEN)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(_M1)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(_M2)
        x = x ^ (x >> np.uint64(31))
    return ((x >> np.uint64(11)) % high[None, :]).astype(np.int32)


def uniform_ints_torch(torch, seed: int, episode: int, first_env: int, num_envs: int, num_cols: int, high, device):
    """Same values as uniform_ints_numpy, computed on `device` (int64 holds the uint64 bit patterns; logical right
    shifts are arithmetic shifts with the sign extension masked off)."""
    i64 = torch.int64
    high_t = torch.as_tensor(np.broadcast_to(np.asarray(high, dtype=np.int64), (num_cols,)).copy(), device=device)
    env = torch.arange(first_env, first_env + num_envs, dtype=i64, device=device)[:, None]
    col = torch.arange(num_cols, dtype=i64, device=device)[None, :]

    def shr(v, s):
        return (v >> s) & ((1 << (64 - s)) - 1)

    x = (env * num_cols + col + 1) * _signed(_GOLDEN) + _signed(stream_key(seed, episode))
    x = (x ^ shr(x, 30)) * _signed(_M1)
    x = (x ^ shr(x, 27)) * _signed(_M2)
    x = x ^ shr(x, 31)
    return (shr(x, 11) % high_t[None, :]).to(torch.int32)
