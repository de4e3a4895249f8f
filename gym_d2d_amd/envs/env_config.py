"""EnvConfig: the reference's configuration surface (gym_d2d/envs/env_config.py:10-37; documented in its
README.md:106-125) with the same field names and defaults, so an env_config dict written for the reference is accepted
unchanged and an unknown key raises TypeError exactly like the dataclass there.  Four extra keys (num_envs,
device_ordinal, seed, obs_dtype) describe the batch, the GPU and the observation dtype.  `num_pwr_actions` centralises the action-space arithmetic that
the reference keeps in D2DEnv.__init__."""
from __future__ import annotations

import json
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, Optional, Type

from ..path_loss import LogDistancePathLoss, PathLoss
from ..traffic_model import TrafficModel, UplinkTrafficModel


@dataclass
class EnvConfig:
    # ---- network size
    num_rbs: int = 25                                   # resource blocks agents can choose from (R)
    num_cues: int = 25                                  # cellular users, one uplink each (C)
    num_due_pairs: int = 25                             # D2D transmitter/receiver pairs (P)
    # ---- geometry
    cell_radius_m: float = 500.0                        # devices are placed inside this disc around the BS
    d2d_radius_m: float = 20.0                          # max distance of a DUE receiver from its transmitter
    # ---- transmit power ranges (dBm); they size the discrete action spaces
    due_min_tx_power_dBm: int = 0
    due_max_tx_power_dBm: int = 20
    cue_max_tx_power_dBm: int = 23
    mbs_max_tx_power_dBm: int = 46
    # ---- plugins: CLASSES, instantiated by the simulator as path_loss_model(carrier_freq_GHz), traffic_model(num_rbs)
    path_loss_model: Type[PathLoss] = LogDistancePathLoss
    traffic_model: Type[TrafficModel] = UplinkTrafficModel
    # ---- radio
    carrier_freq_GHz: float = 2.1
    num_subcarriers: int = 12                           # per resource block
    subcarrier_spacing_kHz: int = 15                    # 12 x 15 kHz = 180 kHz per RB
    channel_bandwidth_MHz: float = 20.0
    # ---- optional JSON of fixed positions / per-device link budgets (D2DEnv.save_device_config writes one)
    device_config_file: Optional[Path] = None
    # ---- additions (not in the reference)
    num_envs: int = 1                                   # B: independent environments stepped together on one GPU
    device_ordinal: int = 0                             # which GPU
    seed: Optional[int] = None                          # seed of the device-side RNG streams (reset, shadowing)
    obs_dtype: str = 'float32'                          # 'float64' = the reference's observation dtype (obs_fn.py:51)

    def __post_init__(self) -> None:
        if self.obs_dtype not in ('float32', 'float64'):
            raise ValueError("obs_dtype must be 'float32' or 'float64'")
        self.devices: Dict[str, dict] = self.load_device_config()

    def load_device_config(self) -> dict:
        """{device id: {'position': [x, y], 'config': {...}}} parsed from device_config_file, or {} when unset."""
        path = self.device_config_file
        if not isinstance(path, Path):
            return {}
        with path.open(mode='r') as handle:
            return json.load(handle)

    @property
    def num_pwr_actions(self) -> Dict[str, int]:
        """Power levels per transmitter class (d2d_env.py:31-35; +1 because the top level is included)."""
        return {
            'due': self.due_max_tx_power_dBm - self.due_min_tx_power_dBm + 1,
            'cue': self.cue_max_tx_power_dBm + 1,
            'mbs': self.mbs_max_tx_power_dBm + 1,
        }
