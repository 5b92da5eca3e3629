"""Optional per-launch HIP-event timing of libm324 calls (used by bench.py for the roofline object).

When a Recorder is active, ops.gemm / ops.attention bracket their C-ABI call with events on torch's
current stream -- the stream the kernels are launched on -- and tag them with the launch's algorithmic
FLOPs.  Durations are read after the timed region has been synchronised.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Optional, Tuple

import torch

_active: Optional["Recorder"] = None


class Recorder:
    def __init__(self):
        self.items: List[Tuple[str, float, float, torch.cuda.Event, torch.cuda.Event, str]] = []

    def __enter__(self):
        global _active
        _active = self
        return self

    def __exit__(self, *exc):
        global _active
        _active = None

    def summary(self) -> Dict[str, dict]:
        """kernel class -> {launches, total_ms, avg_ms, flops, bytes} (call after torch.cuda.synchronize())."""
        acc = defaultdict(lambda: {"launches": 0, "total_ms": 0.0, "flops": 0.0, "bytes": 0.0})
        for name, flops, nbytes, e0, e1, _tag in self.items:
            a = acc[name]
            a["launches"] += 1
            a["total_ms"] += e0.elapsed_time(e1)
            a["flops"] += flops
            a["bytes"] += nbytes
        for a in acc.values():
            a["avg_ms"] = a["total_ms"] / max(a["launches"], 1)
        return dict(acc)


    def by_tag(self) -> Dict[Tuple[str, str], dict]:
        """(kernel class, tag) -> {launches, total_ms, flops}: per-shape breakdown for tuning."""
        acc = defaultdict(lambda: {"launches": 0, "total_ms": 0.0, "flops": 0.0, "bytes": 0.0})
        for name, flops, nbytes, e0, e1, tag in self.items:
            a = acc[(name, tag)]
            a["launches"] += 1
            a["total_ms"] += e0.elapsed_time(e1)
            a["flops"] += flops
            a["bytes"] += nbytes
        return dict(acc)


def active() -> bool:
    """True while a Recorder is collecting (callers skip building tags otherwise)."""
    return _active is not None


class span:
    """with span(name, flops, bytes, tag): <one C-ABI launch>; `tag` (e.g. the GEMM shape) only refines by_tag()."""
    __slots__ = ("name", "flops", "nbytes", "e0", "tag")

    def __init__(self, name: str, flops: float, nbytes: float = 0.0, tag: str = ""):
        self.name, self.flops, self.nbytes, self.e0, self.tag = name, flops, nbytes, None, tag

    def __enter__(self):
        if _active is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def cancel(self) -> None:
        """Nothing was launched inside (the library refused the call): record no row."""
        self.e0 = None

    def __exit__(self, *exc):
        if self.e0 is not None and _active is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _active.items.append((self.name, self.flops, self.nbytes, self.e0, e1, self.tag))
