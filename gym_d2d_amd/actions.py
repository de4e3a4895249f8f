"""Actions of one step (mirrors gym_d2d/actions.py): who transmits to whom, on which RB, at what power."""
from __future__ import annotations

from collections import UserDict
from dataclasses import dataclass
from typing import Dict, Set

from .device import Device
from .link_type import LinkType


@dataclass(frozen=True)
class Action:
    tx: Device
    rx: Device
    link_type: LinkType
    rb: int
    tx_pwr_dBm: float


class Actions(UserDict):
    """{(tx_id, rx_id): Action}, insertion-ordered = agent order, with a lazily built per-RB index."""

    def __init__(self, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._by_rb: Dict[int, Set[Action]] = {}

    def clear(self) -> None:
        super().clear()
        self._by_rb = {}

    def get_actions_by_rb(self, rb: int) -> Set[Action]:
        """All actions sharing resource block `rb` (index built on first use, kept until clear())."""
        if not self._by_rb:
            for act in self.data.values():
                self._by_rb.setdefault(act.rb, set()).add(act)
        return self._by_rb.setdefault(rb, set())
