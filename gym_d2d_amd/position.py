"""Planar positions and the reference's two samplers (gym_d2d/position.py).

Host samplers consume Python's global `random` in the same order as the reference (theta first, then radius), so
`random.seed(k)` reproduces the reference's layout for a single env.  Batches are sampled on the GPU instead
(csrc/d2d_reset.hip) from a counter-based stream.
"""
from __future__ import annotations

import math
import random
from dataclasses import dataclass
from typing import Tuple

_TWO_PI = 2 * math.pi


@dataclass
class Position:
    x: float
    y: float

    def distance(self, other: 'Position') -> float:
        # written as (dx^2 + dy^2)^0.5 rather than math.hypot so it rounds exactly like position.py:12
        return ((self.x - other.x) ** 2 + (self.y - other.y) ** 2) ** 0.5

    def as_tuple(self) -> Tuple[float, float]:
        return self.x, self.y


def _polar_draw(radius: float) -> Tuple[float, float]:
    angle = _TWO_PI * random.random()
    rho = radius * math.sqrt(random.random())     # sqrt: uniform over the disc's area
    return rho * math.cos(angle), rho * math.sin(angle)


def get_random_position(radius: float) -> Position:
    """Uniform point in the disc of `radius` centred on the origin (position.py:18-28)."""
    return Position(*_polar_draw(radius))


def get_random_position_nearby(radius: float, anchor_pos: Position, anchor_radius: float) -> Position:
    """Uniform point within `anchor_radius` of `anchor_pos`, re-drawn until it lies inside the cell disc
    (position.py:31-45)."""
    limit = radius * radius
    while True:
        dx, dy = _polar_draw(anchor_radius)
        x, y = anchor_pos.x + dx, anchor_pos.y + dy
        if not (x * x + y * y > limit):
            return Position(x, y)
