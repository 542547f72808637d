"""meshclust2_amd: MI355X-native (gfx950) drop-in for MeShClust2's alignment-free pairwise-identity hot path.

The product is libmeshclust2_hip.so (hand-written HIP kernels behind the C ABI in include/meshclust2_hip.h).
This package holds its sources (csrc/), a C++ header mirroring the reference's classes (host/), and the thin
ctypes mirror the tests and bench.py drive it through. There is no CPU fallback anywhere in this package.
"""
from ._capi import FEAT, FEAT_DIV, FEAT_FAST, FEAT_SLOW, LIB_PATH, MscError, load_library  # noqa: F401

__all__ = ["FEAT", "FEAT_FAST", "FEAT_DIV", "FEAT_SLOW", "LIB_PATH", "MscError", "load_library"]
