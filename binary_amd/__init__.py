"""binary_amd — MI355X-native interval-overlap engine behind ylab-hi/BINARY's IntervalTree API.

Layout: csrc/ (hand-written HIP kernels + the C ABI of include/bivx.h -> libbivx.so), capi.py (ctypes
binding), interval_index.py (host mirror of the reference's IntervalTree interface), sharding.py
(per-chromosome multi-GPU sharding), synth.py (portable synthetic workloads).
"""
from .capi import BIVX_NO_HIT, BivxError  # noqa: F401
from .interval_index import IntervalIndex  # noqa: F401

__all__ = ["IntervalIndex", "BivxError", "BIVX_NO_HIT"]
