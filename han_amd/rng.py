"""Seeds for the counter-based dropout RNG of the kernels (han_common.h).

Each dropout site draws ``keep = top24(hash(seed, stream, a, b)) < keep_prob*2^24``
from global ids, so a forward, its backward and every node partition see the same
mask.  A fresh 64-bit seed is taken per (training step, meta-path).
"""
from __future__ import annotations

_state = {"seed": 0x243F6A8885A308D3, "counter": 0}
_MASK = (1 << 64) - 1


def manual_seed(seed: int) -> None:
    _state["seed"] = int(seed) & _MASK
    _state["counter"] = 0


def _splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _MASK
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
    return z ^ (z >> 31)


def next_seed() -> int:
    """Deterministic stream of 64-bit seeds; identical on every rank that made
    the same number of draws."""
    _state["counter"] += 1
    return _splitmix64(_state["seed"] ^ _splitmix64(_state["counter"]))
