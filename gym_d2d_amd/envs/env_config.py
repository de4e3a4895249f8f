"""EnvConfig (mirrors gym_d2d/envs/env_config.py:10-37): same field names and defaults, so an env_config dict
written for the reference is accepted unchanged; an unknown key raises TypeError exactly like the dataclass there.
Three extra keys (num_envs, device_ordinal, seed) describe the batch and the GPU."""
from __future__ import annotations

import json
from dataclasses import dataclass
from pathlib import Path
from typing import Optional, Type

from ..path_loss import LogDistancePathLoss, PathLoss
from ..traffic_model import TrafficModel, UplinkTrafficModel


@dataclass
class EnvConfig:
    num_rbs: int = 25
    num_cues: int = 25
    num_due_pairs: int = 25
    cell_radius_m: float = 500.0
    d2d_radius_m: float = 20.0
    due_min_tx_power_dBm: int = 0
    due_max_tx_power_dBm: int = 20
    cue_max_tx_power_dBm: int = 23
    mbs_max_tx_power_dBm: int = 46
    path_loss_model: Type[PathLoss] = LogDistancePathLoss
    traffic_model: Type[TrafficModel] = UplinkTrafficModel
    carrier_freq_GHz: float = 2.1
    num_subcarriers: int = 12
    subcarrier_spacing_kHz: int = 15
    channel_bandwidth_MHz: float = 20.0
    device_config_file: Optional[Path] = None
    # ---- additions (not in the reference)
    num_envs: int = 1               # B: independent environments stepped together on one GPU
    device_ordinal: int = 0         # which GPU
    seed: Optional[int] = None      # seed of the device-side reset stream (batched env)

    def __post_init__(self) -> None:
        self.devices = self.load_device_config()

    def load_device_config(self) -> dict:
        """{device id: {'position': [x, y], 'config': {...}}} from the JSON file, or {}."""
        if isinstance(self.device_config_file, Path):
            with self.device_config_file.open(mode='r') as fid:
                return json.load(fid)
        return {}

    @property
    def num_pwr_actions(self) -> dict:
        """Power levels per transmitter class (d2d_env.py:31-35; +1 because the max level is included)."""
        return {
            'due': self.due_max_tx_power_dBm - self.due_min_tx_power_dBm + 1,
            'cue': self.cue_max_tx_power_dBm + 1,
            'mbs': self.mbs_max_tx_power_dBm + 1,
        }
