"""Result records of the reference API (``/root/reference/src/core.rs:94-122``)."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Any, Generic, Optional, TypeVar

T = TypeVar("T")
E = TypeVar("E")


@dataclass(frozen=True)
class ActionReward(Generic[T, E]):
    """``ActionReward<T, E>`` — core.rs:94-106: what ``Env::step`` returns."""

    observation: T
    reward: float
    done: bool
    truncated: bool
    info: Optional[Any]


@dataclass(frozen=True)
class RewardRange:
    """``RewardRange`` — core.rs:109-122; the default is (-inf, +inf) (core.rs:16-19)."""

    lower_bound: float = -math.inf
    upper_bound: float = math.inf
