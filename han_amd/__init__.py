"""han_amd -- MI355X (gfx950) native HAN layer: node-level attention, semantic-level
attention and the HeteGAT_multi surface of CG-Labs/HAN on hand-written HIP kernels.

Importing the package does not touch the GPU; the first compute call loads
``han_amd/libhan_hip.so`` and fails loudly if it is missing (no CPU fallback).
"""
from . import rng  # noqa: F401
from .graph import CSRGraph, as_graph  # noqa: F401

__all__ = ["CSRGraph", "as_graph", "rng"]
__version__ = "0.1.0"
