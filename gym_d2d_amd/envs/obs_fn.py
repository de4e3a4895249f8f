"""Observation plugins (mirrors gym_d2d/envs/obs_fn.py).

Two levels:
  * `ObsFunction` - the reference's ABC, dict-in / dict-out, for single-env drop-in use.  Built-ins are executed
    by the HIP obs kernel; their get_state() only re-keys the kernel's output.  A user subclass that overrides
    get_state() runs as ordinary Python on dict views of the GPU results.
  * `ArrayObsFunction` - array-native variant for the batched env: receives the step's arrays (device tensors when
    torch is available) and returns an array; `native_mode` tells the kernel what to materialise for it.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Dict

import numpy as np

from .. import _native
from ..spaces import Box, Space


class ObsFunction(ABC):
    native_mode = _native.OBS_TABLE     # what a Python-side get_state needs from the GPU: the base table

    @abstractmethod
    def get_obs_space(self, env_config) -> Space:
        """Observation space of one agent."""

    @abstractmethod
    def get_state(self, actions, state: dict, devices) -> Dict[str, np.ndarray]:
        """Observations per 'tx:rx' agent id for the step described by `actions` / `state`."""


class LinearObsFunction(ObsFunction):
    """Every agent sees the whole network: its own (tx_x, tx_y, rx_x, rx_y, sinr_dB, snr_dB) first, then every other
    link's six values in agent order (obs_fn.py:43-61).  Materialised by csrc/d2d_obs.hip."""
    native_mode = _native.OBS_LINEAR
    FEATURES = 6

    def get_obs_space(self, env_config) -> Space:
        r = env_config.cell_radius_m
        width = self.FEATURES * (env_config.num_cues + env_config.num_due_pairs)
        return Box(low=-r, high=r, shape=(width,))

    def get_state(self, actions, state, devices) -> Dict[str, np.ndarray]:
        obs = getattr(state, 'linear_obs', None)
        if obs is None:
            raise RuntimeError('LinearObsFunction is evaluated by the HIP obs kernel; `state` must be the NativeState '
                               'returned by gym_d2d_amd.Simulator.step (there is no host implementation)')
        return dict(zip(map(':'.join, getattr(actions, 'data', actions).keys()), obs))     # rows of the [N, 6N] block


class ArrayObsFunction(ABC):
    """Batched plugin: compute(view) -> [B, N, width] array/tensor.  `view` has pos_x/pos_y [B,D], rb, pwr, sinr_db,
    snr_db, rate_bps, capacity_mbps [B,N], table [B,N,6], link_tx/link_rx/link_type [N]."""
    native_mode = _native.OBS_TABLE

    @abstractmethod
    def get_obs_space(self, env_config) -> Space:
        pass

    @abstractmethod
    def compute(self, view):
        pass


class OwnLinkObsFunction(ArrayObsFunction):
    """Example custom plugin (BASELINE.json config 4): each agent observes only its own six values."""

    def get_obs_space(self, env_config) -> Space:
        r = env_config.cell_radius_m
        return Box(low=-r, high=r, shape=(6,))

    def compute(self, view):
        return view.table


class SignalPlanesObsFunction(ArrayObsFunction):
    """No observation array is materialised at all (D2D_OBS_NONE: the step writes neither the [B,N,6N] block nor the
    compact [B,N,6] table): a learner that builds its own features reads the step's (sinr_dB, snr_dB) planes - the only
    columns of the table that change within an episode (obs_fn.py:57-60) - and takes the four position columns once per
    reset from `VecD2DEnv.link_positions()` ([B,N,4], simulator.py:61-75 only moves devices in reset()).  compute()
    returns the pair of planes; `distributed.StepGatherer(mode='planes')` ships exactly these across GPUs."""
    native_mode = _native.OBS_NONE

    def get_obs_space(self, env_config) -> Space:
        return Box(low=-np.inf, high=np.inf, shape=(2,))

    def compute(self, view):
        return view.sinr_db, view.snr_db
