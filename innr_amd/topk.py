"""Host-side mirror of innr's `topk::TopK` (reference: src/topk.rs:47-187): fixed-capacity tracker of the K
smallest (id, distance) pairs for graph traversal loops. Pure host data structure in the reference and here
(a few dozen entries, called per candidate); the device equivalent for scans is the fused candidate list
(innr_amd/csrc/topk_dev.h)."""
from __future__ import annotations

import math
import struct
from typing import List, Tuple

from ._lib import InnrPanic


def _key(x: float) -> int:
    """f32::total_cmp key."""
    b = struct.unpack("<i", struct.pack("<f", x))[0]
    return b ^ ((b >> 31) & 0x7FFFFFFF)


class TopK:
    def __init__(self, k: int):
        if k <= 0:
            raise InnrPanic("innr::TopK: k must be >= 1")  # topk.rs:65
        self.k = k
        self._d: List[float] = []  # sorted DESCENDING: worst at index 0 (topk.rs:49-50)
        self._i: List[int] = []

    def threshold(self) -> float:  # topk.rs:80-87
        return math.inf if len(self._d) < self.k else self._d[0]

    def _pos(self, distance: float, length: int) -> int:
        # topk.rs:171-186 via core::slice::binary_search_by (branch-light form): the last not-Greater probe decides
        if length == 0:
            return 0
        kd = _key(distance)
        size, base = length, 0
        while size > 1:
            half = size // 2
            mid = base + half
            if not (_key(self._d[mid]) < kd):
                base = mid
            size -= half
        ke = _key(self._d[base])
        if ke == kd:
            return base
        return base + (1 if ke > kd else 0)

    def insert(self, id_: int, distance: float) -> None:  # topk.rs:96-121
        distance = struct.unpack("<f", struct.pack("<f", distance))[0]
        if len(self._d) < self.k:
            p = self._pos(distance, len(self._d))
            self._d.insert(p, distance)
            self._i.insert(p, id_)
        elif _key(distance) < _key(self._d[0]):  # strict less, total order (NaN sorts greatest)
            del self._d[0]
            del self._i[0]
            p = self._pos(distance, self.k - 1)
            self._d.insert(p, distance)
            self._i.insert(p, id_)

    def __len__(self) -> int:
        return len(self._d)

    def is_empty(self) -> bool:
        return not self._d

    def into_sorted(self) -> List[Tuple[int, float]]:  # topk.rs:140-145: ascending, best first
        return list(zip(reversed(self._i), reversed(self._d)))
