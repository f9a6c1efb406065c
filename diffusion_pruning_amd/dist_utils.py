"""Small helpers shared by the data-parallel code paths (train_step.py, quantizer.py): world / rank that work without an
initialised process group, and an optional device-side stopwatch around the collectives."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist


def _world() -> int:
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def _rank() -> int:
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


class CollectiveTimer:
    """Optional device-side stopwatch around the step's collectives (bench.py --config train reports their share of a
    step).  Events are recorded on the current stream right before and after each collective, so a span covers the
    hand-off to RCCL's stream, the collective and the hand-back.  Inactive (None) by default: no events, no overhead."""

    def __init__(self):
        self.spans = []

    def span(self, name):
        return _Span(self, name)

    def total_ms(self, reset: bool = True) -> Dict[str, float]:
        out: Dict[str, float] = {}
        for name, e0, e1 in self.spans:
            out[name] = out.get(name, 0.0) + (e0.elapsed_time(e1) if e0 is not None else e1)
        if reset:
            self.spans = []
        return out


class _Span:
    def __init__(self, timer, name):
        self.timer, self.name = timer, name

    def __enter__(self):
        if torch.cuda.is_available():
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        else:
            import time
            self.e0, self.t0 = None, time.perf_counter()
        return self

    def __exit__(self, *exc):
        if self.e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.timer.spans.append((self.name, self.e0, e1))
        else:
            import time
            self.timer.spans.append((self.name, None, (time.perf_counter() - self.t0) * 1e3))
        return False


class _NoSpan:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


COLLECTIVE_TIMER: Optional[CollectiveTimer] = None


def _span(name):
    return COLLECTIVE_TIMER.span(name) if COLLECTIVE_TIMER is not None else _NoSpan()


