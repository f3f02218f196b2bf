"""oracle/ivtree_oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes wrapper over oracle/liboracle_ivtree.so (ivtree.c: the CPU restatement of the reference's
red-black interval tree, rb_tree.hpp + interval_tree.hpp). Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg import this; nothing under binary_amd/ or include/ does.

Parity-pin status: pinned by the reference's own known-answer tests and SURVEY.md §8c vectors
(tests/test_oracle_pins.py); no oracle/_ref build exists (reference needs spdlog: unbuildable here).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_ivtree.so")

_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile). Building the checker is not using it."""
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("ivtree.c", "ivtree.h"))
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < src_m:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "all"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.ivt_create.restype = C.c_void_p
        L.ivt_destroy.argtypes = [C.c_void_p]
        L.ivt_insert.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.ivt_insert.restype = C.c_int32
        L.ivt_insert_many.argtypes = [C.c_void_p, _u32p, _u32p, C.c_size_t]
        L.ivt_size.argtypes = [C.c_void_p]
        L.ivt_size.restype = C.c_size_t
        L.ivt_root.argtypes = [C.c_void_p]
        L.ivt_root.restype = C.c_int32
        for name in ("ivt_low", "ivt_high", "ivt_max"):
            f = getattr(L, name)
            f.argtypes = [C.c_void_p, C.c_int32]
            f.restype = C.c_uint32
        for name in ("ivt_left", "ivt_right", "ivt_parent", "ivt_minimum", "ivt_maximum",
                     "ivt_successor", "ivt_predecessor", "ivt_black_height", "ivt_is_red"):
            f = getattr(L, name)
            f.argtypes = [C.c_void_p, C.c_int32]
            f.restype = C.c_int32
        L.ivt_search.argtypes = [C.c_void_p, C.c_uint32]
        L.ivt_search.restype = C.c_int32
        L.ivt_check_max.argtypes = [C.c_void_p]
        L.ivt_check_max.restype = C.c_int
        L.ivt_preorder.argtypes = [C.c_void_p, _i32p, C.c_size_t]
        L.ivt_preorder.restype = C.c_size_t
        L.ivt_find_overlap.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.ivt_find_overlap.restype = C.c_int32
        L.ivt_find_overlaps.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]
        L.ivt_find_overlaps.restype = C.c_size_t
        L.ivt_find_overlaps_batch.argtypes = [C.c_void_p, _u32p, _u32p, C.c_size_t, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_void_p]
        L.ivt_find_overlaps_batch.restype = C.c_uint64
        L.ivt_brute_overlaps.argtypes = [_u32p, _u32p, C.c_size_t, C.c_uint32, C.c_uint32,
                                         C.c_void_p, C.c_size_t]
        L.ivt_brute_overlaps.restype = C.c_size_t
        L.ivt_delete.argtypes = [C.c_void_p, C.c_int32]
        _lib = L
    return _lib


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint32))


class OracleTree:
    """The reference IntervalTree<UIntIntervalNode>, restated. Node index == insertion index."""

    def __init__(self, low=None, high=None):
        self._L = lib()
        self._t = C.c_void_p(self._L.ivt_create())
        self.n = 0
        if low is not None:
            self.insert_many(low, high)

    def __del__(self):
        t, self._t = getattr(self, "_t", None), None
        if t:
            self._L.ivt_destroy(t)

    # -- build ------------------------------------------------------------------------------------
    def insert(self, low: int, high: int) -> int:
        self.n += 1
        return self._L.ivt_insert(self._t, low, high)

    def insert_many(self, low, high) -> None:
        low, high = _u32(low), _u32(high)
        assert low.shape == high.shape and low.ndim == 1
        self._L.ivt_insert_many(self._t, low, high, low.size)
        self.n += low.size

    # -- structure --------------------------------------------------------------------------------
    def size(self) -> int:
        return self._L.ivt_size(self._t)

    def root(self) -> int:
        return self._L.ivt_root(self._t)

    def node(self, i: int) -> dict:
        L, t = self._L, self._t
        return dict(low=L.ivt_low(t, i), high=L.ivt_high(t, i), max=L.ivt_max(t, i),
                    left=L.ivt_left(t, i), right=L.ivt_right(t, i), parent=L.ivt_parent(t, i),
                    red=bool(L.ivt_is_red(t, i)))

    def black_height(self, i: int | None = None) -> int:
        return self._L.ivt_black_height(self._t, self.root() if i is None else i)

    def check_max(self) -> bool:
        return bool(self._L.ivt_check_max(self._t))

    def preorder(self) -> np.ndarray:
        out = np.empty(max(self.n, 1), dtype=np.int32)
        k = self._L.ivt_preorder(self._t, out, out.size)
        return out[:k]

    def minimum(self, i): return self._L.ivt_minimum(self._t, i)
    def maximum(self, i): return self._L.ivt_maximum(self._t, i)
    def successor(self, i): return self._L.ivt_successor(self._t, i)
    def predecessor(self, i): return self._L.ivt_predecessor(self._t, i)
    def search(self, key): return self._L.ivt_search(self._t, key)
    def delete(self, i): self._L.ivt_delete(self._t, i)

    # -- queries ----------------------------------------------------------------------------------
    def find_overlap(self, qlow: int, qhigh: int) -> int:
        """Node index of the reference's single-descent hit, or -1."""
        return self._L.ivt_find_overlap(self._t, qlow, qhigh)

    def find_overlaps(self, qlow: int, qhigh: int) -> np.ndarray:
        """Node (insertion) indices of all hits, in the reference's pre-order."""
        k = self._L.ivt_find_overlaps(self._t, qlow, qhigh, None, 0)
        out = np.empty(max(k, 1), dtype=np.int32)
        self._L.ivt_find_overlaps(self._t, qlow, qhigh, out.ctypes.data, out.size)
        return out[:k]

    def find_overlaps_batch(self, qlow, qhigh, nthreads: int = 1, want_hits: bool = True):
        """CSR (offsets u64[q+1], hits i32[H]) — per query the reference's pre-order hit list."""
        qlow, qhigh = _u32(qlow), _u32(qhigh)
        q = qlow.size
        counts = np.zeros(q, dtype=np.uint32)
        total = self._L.ivt_find_overlaps_batch(self._t, qlow, qhigh, q, nthreads,
                                                counts.ctypes.data, None, None)
        offsets = np.zeros(q + 1, dtype=np.uint64)
        np.cumsum(counts, dtype=np.uint64, out=offsets[1:])
        assert int(offsets[-1]) == total
        if not want_hits:
            return offsets, None
        hits = np.empty(max(total, 1), dtype=np.int32)
        self._L.ivt_find_overlaps_batch(self._t, qlow, qhigh, q, nthreads, None,
                                        offsets.ctypes.data, hits.ctypes.data)
        return offsets, hits[:total]


def brute_overlaps(low, high, qlow: int, qhigh: int) -> np.ndarray:
    """is_overlap over every interval (interval_tree.hpp:119-121), ascending insertion index."""
    low, high = _u32(low), _u32(high)
    k = lib().ivt_brute_overlaps(low, high, low.size, qlow, qhigh, None, 0)
    out = np.empty(max(k, 1), dtype=np.int32)
    lib().ivt_brute_overlaps(low, high, low.size, qlow, qhigh, out.ctypes.data, out.size)
    return out[:k]


def sorted_csr(offsets: np.ndarray, hits: np.ndarray) -> np.ndarray:
    """Per-query ascending hit ids (the order-free canonical form parity tests compare)."""
    offsets = np.asarray(offsets, dtype=np.int64)
    if hits.size == 0:
        return hits.astype(np.int64)
    qid = np.repeat(np.arange(offsets.size - 1, dtype=np.int64), np.diff(offsets))
    order = np.lexsort((hits.astype(np.int64), qid))
    return hits.astype(np.int64)[order]


def count_overlaps_numpy(low, high, qlow, qhigh) -> np.ndarray:
    """Per-query hit counts from the predicate alone, for sets where every low <= high and every
    qlow <= qhigh: #(low <= qhigh) - #(high < qlow). Sort + searchsorted, so it runs at any size."""
    low, high = np.sort(_u32(low)), np.sort(_u32(high))
    qlow, qhigh = _u32(qlow), _u32(qhigh)
    return (np.searchsorted(low, qhigh, side="right") - np.searchsorted(high, qlow, side="left")).astype(np.int64)
