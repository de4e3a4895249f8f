"""Reward plugins (mirrors gym_d2d/envs/reward_fn.py).  The three built-ins are evaluated inside the HIP step
kernel (csrc/d2d_step.hip, pass 3); `native_id` / `native_param` select the branch.  Calling a built-in with a
NativeState just re-keys the kernel's per-agent output.  A user subclass that overrides __call__ runs as Python."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Dict

from .. import _native


class RewardFunction(ABC):
    native_id = _native.REWARD_NONE

    @property
    def native_param(self) -> float:
        return 0.0

    @abstractmethod
    def __call__(self, actions, state: dict) -> Dict[str, float]:
        """Reward per 'tx:rx' agent id."""

    def _from_kernel(self, actions, state) -> Dict[str, float]:
        got = getattr(state, 'native_reward', None)
        if got is None or getattr(state, 'native_reward_key', None) != (self.native_id, float(self.native_param)):
            raise RuntimeError(f'{type(self).__name__} is evaluated by the HIP step kernel; `state` must be the '
                               'NativeState of an env configured with this reward function')
        return dict(zip(map(':'.join, getattr(actions, 'data', actions).keys()), got.tolist()))


class SystemCapacityRewardFunction(RewardFunction):
    """Mean capacity over all links, or -1 for everyone if a D2D link shares an RB with a non-D2D link whose
    capacity is <= min_capacity_mbps (reward_fn.py:22-44)."""
    native_id = _native.REWARD_SYSTEM_CAPACITY

    def __init__(self, min_capacity_mbps=0.0) -> None:
        self.min_capacity_mbps = float(min_capacity_mbps)

    @property
    def native_param(self) -> float:
        return self.min_capacity_mbps

    def __call__(self, actions, state):
        return self._from_kernel(actions, state)


class ShannonRewardFunction(RewardFunction):
    """log2(1 + SINR) per agent, -1 below min_sinr dB (reward_fn.py:47-57)."""
    native_id = _native.REWARD_SHANNON

    def __init__(self, min_sinr=-70.0) -> None:
        self.min_sinr = float(min_sinr)

    @property
    def native_param(self) -> float:
        return self.min_sinr

    def __call__(self, actions, state):
        return self._from_kernel(actions, state)


class CueSinrShannonRewardFunction(RewardFunction):
    """Own Shannon rate unless a non-D2D link on my RB has SINR below the threshold, then -1 (reward_fn.py:60-78)."""
    native_id = _native.REWARD_CUE_SINR_SHANNON

    def __init__(self, sinr_threshold_dB=0.0) -> None:
        self.sinr_threshold_dB = float(sinr_threshold_dB)

    @property
    def native_param(self) -> float:
        return self.sinr_threshold_dB

    def __call__(self, actions, state):
        return self._from_kernel(actions, state)
