"""
Optional per-kernel-family timing (HIP events on the current stream).  Disabled by default;
bench.py enables it to attribute the step time and to compute the roofline of the dominant kernel.
"""
import contextlib
from collections import defaultdict

import torch

_enabled = False
_events = defaultdict(list)


def enable(flag: bool = True):
    global _enabled
    _enabled = flag
    _events.clear()


def enabled() -> bool:
    return _enabled


@contextlib.contextmanager
def section(name: str):
    if not _enabled:
        yield
        return
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    start.record()
    try:
        yield
    finally:
        end.record()
        _events[name].append((start, end))


def summary():
    """name -> (count, mean milliseconds). Call after torch.cuda.synchronize()."""
    out = {}
    for name, pairs in _events.items():
        ms = [s.elapsed_time(e) for s, e in pairs]
        out[name] = (len(ms), sum(ms) / max(1, len(ms)))
    return out
