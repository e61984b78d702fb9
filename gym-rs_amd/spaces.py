"""Value-space descriptors (``/root/reference/src/spaces``)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Generic, TypeVar

T = TypeVar("T")


@dataclass(frozen=True)
class Discrete:
    """``Discrete(pub usize)`` — spaces/discrete.rs:12: the integers ``0..n``."""

    n: int

    def contains(self, value: int) -> bool:
        """spaces/discrete.rs:14-19: ``value < upper_bound`` (usize: negatives do not exist)."""
        return 0 <= int(value) < self.n


@dataclass(frozen=True)
class BoxR(Generic[T]):
    """``BoxR<T>{low, high}`` — spaces/box_r.rs:5-13: a plain pair of bounds (no ``contains``)."""

    low: T
    high: T
