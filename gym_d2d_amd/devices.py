"""The device set of one environment (mirrors gym_d2d/devices.py) with a stable integer index per device.

Index order is the reference's iteration order (devices.py:20-25): base station, CUEs, then every DUE pair as
(tx, rx).  The HIP side addresses devices by this index.
"""
from __future__ import annotations

from collections.abc import Mapping
from typing import Dict, Iterator, Tuple

from .device import BaseStation, Device, UserEquipment
from .id import Id


class Devices(Mapping):
    def __init__(self, bs: BaseStation, cues: Dict[Id, UserEquipment],
                 due_pairs: Dict[Tuple[Id, Id], Tuple[UserEquipment, UserEquipment]]) -> None:
        self.bs = bs
        self.cues = cues
        self.dues = due_pairs
        self.due_pairs: Dict[Id, Id] = {}       # tx id -> rx id
        self.due_pairs_inv: Dict[Id, Id] = {}   # rx id -> tx id
        self._by_id: Dict[Id, Device] = {bs.id: bs}
        self._by_id.update(cues)
        for (tx_id, rx_id), (tx, rx) in due_pairs.items():
            self._by_id[tx_id] = tx
            self._by_id[rx_id] = rx
            self.due_pairs[tx_id] = rx_id
            self.due_pairs_inv[rx_id] = tx_id
        self.index: Dict[Id, int] = {dev_id: k for k, dev_id in enumerate(self._by_id)}

    def __getitem__(self, key: Id) -> Device:
        return self._by_id[key]

    def __len__(self) -> int:
        return len(self._by_id)

    def __iter__(self) -> Iterator[Id]:
        return iter(self._by_id)

    def index_of(self, dev_id) -> int:
        """Integer index of a device id (KeyError for unknown ids, like the reference's lookup)."""
        return self.index[dev_id]
