"""Traffic models (mirrors gym_d2d/traffic_model.py).  The reference constructs one (simulator.py:58) but never
calls it (simulator.py:78 is commented out); the classes are kept so env_config['traffic_model'] keeps working and
`assignments()` exposes the same round-robin rule as arrays for the batched env."""
from __future__ import annotations

import numpy as np

from .actions import Action, Actions
from .devices import Devices
from .link_type import LinkType


class TrafficModel:
    def __init__(self, num_rbs: int) -> None:
        self.num_rbs: int = num_rbs

    def get_traffic(self, devices: Devices) -> Actions:
        pass

    def assignments(self, devices: Devices):
        """(rb[C], pwr_dBm[C]) for the CUEs in index order: round-robin RBs at each CUE's max power."""
        cues = list(devices.cues.values())
        rb = np.arange(len(cues), dtype=np.int32) % self.num_rbs
        pwr = np.array([int(c.max_tx_power_dBm) for c in cues], dtype=np.int32)
        return rb, pwr


class UplinkTrafficModel(TrafficModel):
    """Every CUE transmits to the base station (traffic_model.py:15-22)."""

    def get_traffic(self, devices: Devices) -> Actions:
        traffic = Actions()
        for k, (cue_id, cue) in enumerate(devices.cues.items()):
            traffic[(cue_id, devices.bs.id)] = Action(cue, devices.bs, LinkType.UPLINK, k % self.num_rbs,
                                                      cue.max_tx_power_dBm)
        return traffic


class DownlinkTrafficModel(TrafficModel):
    """The base station transmits to every CUE (traffic_model.py:25-32)."""

    def get_traffic(self, devices: Devices) -> Actions:
        traffic = Actions()
        for k, (cue_id, cue) in enumerate(devices.cues.items()):
            traffic[(devices.bs.id, cue_id)] = Action(devices.bs, cue, LinkType.DOWNLINK, k % self.num_rbs,
                                                      cue.max_tx_power_dBm)
        return traffic
