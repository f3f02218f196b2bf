"""Host-side mirror of the reference's IntervalTree interface over the C ABI (include/bivx.h).

The reference API (library/include/binary/algorithm/interval_tree.hpp:140-187, rb_tree.hpp:97-171) is
per-record: insert_node(...), find_overlaps(interval) -> vector, find_overlap(interval) -> optional,
size(), empty(). This class keeps those names and meanings but takes whole arrays, because one device
batch replaces the reference's per-record calls. PyTorch is used only as the owner of device buffers and
streams; every result comes from the hand-written HIP kernels in libbivx.so. No CPU path exists here.
"""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np

from . import capi

try:  # torch is plumbing (device memory, streams); the host-pointer API works without it
    import torch
except Exception:  # pragma: no cover
    torch = None


def _u32(a) -> np.ndarray:
    a = np.asarray(a)
    if a.dtype == np.int32:
        a = a.view(np.uint32)
    return np.ascontiguousarray(a, dtype=np.uint32)


def _ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _tptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _check_dev_tensor(t, name, n=None, itemsize=4):
    if not (torch is not None and isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous()):
        raise TypeError(f"{name}: expected a contiguous device tensor")
    if t.element_size() != itemsize:
        raise TypeError(f"{name}: expected {itemsize}-byte elements, got {t.dtype}")
    if n is not None and t.numel() != n:
        raise ValueError(f"{name}: expected {n} elements, got {t.numel()}")


class IntervalIndex:
    """Batched drop-in for IntervalTree<UIntIntervalNode> (+ a chromosome id per interval/query)."""

    def __init__(self, device=0):
        """device: a HIP device ordinal, or a sequence of them for an index sharded by chromosome over several GPUs
        of the node (bivx_create_sharded; host-array entry points only)."""
        self._L = capi.load()
        h = C.c_void_p()
        if isinstance(device, (list, tuple)):
            devs = (C.c_int * len(device))(*[int(d) for d in device])
            capi.check(self._L.bivx_create_sharded(C.byref(h), devs, len(device)))
            self.device = int(device[0])
        else:
            capi.check(self._L.bivx_create(C.byref(h), int(device)))
            self.device = int(device)
        self._h = h

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.bivx_destroy(h)

    __del__ = close

    @staticmethod
    def release_pooled() -> None:
        """Frees the index objects bivx_destroy parked for the next bivx_create (their device blocks, pinned blocks and
        streams; include/bivx.h, bivx_release_pooled). The library does this itself when a device allocation fails."""
        capi.load().bivx_release_pooled()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- build side (RbTree::insert_node, rb_tree.hpp:111-117,145-149) ---------------------------------
    def insert_node(self, low, high, chrom=None, svtype=None) -> None:
        """Append intervals [low[i], high[i]] (closed); ids continue in append order. svtype (uint8, 1..255) labels
        the intervals (the svtype column of the VCF-record intervals, vcf.hpp:598-639): queries may then ask for one
        type (svtype= of the query methods) and meet only those."""
        if torch is not None and isinstance(low, torch.Tensor) and low.is_cuda:
            n = low.numel()
            _check_dev_tensor(low, "low")
            _check_dev_tensor(high, "high", n)
            if chrom is not None:
                _check_dev_tensor(chrom, "chrom", n)
            s = torch.cuda.current_stream(low.device).cuda_stream
            if svtype is not None:
                _check_dev_tensor(svtype, "svtype", n, 1)
                capi.check(self._L.bivx_append_typed_dev(self._h, _tptr(chrom), _tptr(low), _tptr(high), _tptr(svtype),
                                                         n, C.c_void_p(s)))
            else:
                capi.check(self._L.bivx_append_dev(self._h, _tptr(chrom), _tptr(low), _tptr(high), n, C.c_void_p(s)))
            torch.cuda.current_stream(low.device).synchronize()  # the copy reads caller memory
            return
        low, high = _u32(low).ravel(), _u32(high).ravel()
        if low.shape != high.shape:
            raise ValueError("low and high must have the same length")
        c = None if chrom is None else _u32(chrom).ravel()
        if c is not None and c.shape != low.shape:
            raise ValueError("chrom must have the same length as low")
        if svtype is not None:
            t = np.ascontiguousarray(svtype, dtype=np.uint8).ravel()
            if t.shape != low.shape:
                raise ValueError("svtype must have the same length as low")
            capi.check(self._L.bivx_append_typed(self._h, _ptr(c), _ptr(low), _ptr(high), _ptr(t), low.size))
        else:
            capi.check(self._L.bivx_append(self._h, _ptr(c), _ptr(low), _ptr(high), low.size))

    def build(self) -> None:
        """Makes the appended set searchable (the reference does this incrementally per insert)."""
        capi.check(self._L.bivx_build(self._h))

    def clear(self) -> None:
        capi.check(self._L.bivx_clear(self._h))

    def size(self) -> int:  # RbTree::size rb_tree.hpp:173-180
        return int(self._L.bivx_size(self._h))

    def empty(self) -> bool:  # RbTree::empty rb_tree.hpp:182-184
        return self.size() == 0

    def num_chroms(self) -> int:
        return int(self._L.bivx_num_chroms(self._h))

    def query_kernel_name(self, q: int, hit_capacity: int, sort_by_id: bool = False, flt=None) -> str:
        """Name of the kernel query_device runs for a batch of q queries into a buffer of hit_capacity ids."""
        return self._L.bivx_query_kernel_name(self._h, int(q), int(hit_capacity), 1 if sort_by_id else 0,
                                              None if flt is None else C.byref(flt)).decode()

    def num_devices(self) -> int:
        return int(self._L.bivx_num_devices(self._h))

    def device_of_chrom(self, chrom: int) -> int:
        return int(self._L.bivx_device_of_chrom(self._h, int(chrom)))

    def num_types(self) -> int:
        return int(self._L.bivx_num_types(self._h))

    def get_svtypes(self, ids) -> np.ndarray:
        ids = _u32(ids).ravel()
        out = np.empty(ids.size, np.uint8)
        capi.check(self._L.bivx_get_svtypes(self._h, _ptr(ids), ids.size, _ptr(out)))
        return out

    @staticmethod
    def type_filter(svtype: int):
        """bivx_filter that only selects the intervals of one svtype (no post-filter)."""
        return capi.Filter(capi.FILTER_NONE, 0, 0, int(svtype), None, None)

    def stats(self) -> dict:
        st = capi.Stats()
        capi.check(self._L.bivx_get_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def get_intervals(self, ids):
        ids = _u32(ids).ravel()
        c, lo, hi = (np.empty(ids.size, np.uint32) for _ in range(3))
        capi.check(self._L.bivx_get_intervals(self._h, _ptr(ids), ids.size, _ptr(c), _ptr(lo), _ptr(hi)))
        return c, lo, hi

    def stream_status(self, stream=None) -> None:
        """Synchronises `stream` (default: torch's current stream on the index's device) and raises BivxError
        (code capi.E_TIMEOUT) if a single-pass kernel of this index reported an invalid result since the last check."""
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream if torch is not None else 0
        capi.check(self._L.bivx_stream_status(self._h, C.c_void_p(stream)))

    def _ensure_built(self):
        if not self._L.bivx_is_built(self._h):
            self.build()

    # ---- query side, host arrays (IntervalTree::find_overlaps interval_tree.hpp:161-168,306-334) -------
    def find_overlaps(self, qlow, qhigh, qchrom=None, sort_by_id: bool = True, svtype: int = 0):
        """CSR of all overlaps: (offsets uint64[q+1], hit_ids uint32[H]). svtype != 0: only intervals of that type."""
        self._ensure_built()
        qlow, qhigh = _u32(qlow).ravel(), _u32(qhigh).ravel()
        if qlow.shape != qhigh.shape:
            raise ValueError("qlow and qhigh must have the same length")
        qc = None if qchrom is None else _u32(qchrom).ravel()
        q = qlow.size
        offsets = np.zeros(q + 1, dtype=np.uint64)
        hp = C.POINTER(C.c_uint32)()
        flt = self.type_filter(svtype) if svtype else None
        capi.check(self._L.bivx_find_overlaps(self._h, _ptr(qc), _ptr(qlow), _ptr(qhigh), q,
                                              None if flt is None else C.byref(flt),
                                              1 if sort_by_id else 0, _ptr(offsets), C.byref(hp)))
        total = int(offsets[-1])
        if total == 0:
            return offsets, np.zeros(0, dtype=np.uint32)
        # a zero-copy view of the library's buffer: the ctypes array is the numpy base of `hits` and of every view
        # taken from it, and bivx_free runs when the last of them is gone
        addr = C.cast(hp, C.c_void_p).value
        buf = (C.c_uint32 * total).from_address(addr)
        weakref.finalize(buf, self._L.bivx_free, C.c_void_p(addr))
        return offsets, np.ctypeslib.as_array(buf)

    def find_overlap(self, qlow, qhigh, qchrom=None) -> np.ndarray:
        """Per query the smallest overlapping id, or capi.BIVX_NO_HIT (find_overlap, interval_tree.hpp:290-304)."""
        self._ensure_built()
        qlow, qhigh = _u32(qlow).ravel(), _u32(qhigh).ravel()
        qc = None if qchrom is None else _u32(qchrom).ravel()
        first = np.empty(qlow.size, dtype=np.uint32)
        capi.check(self._L.bivx_any(self._h, _ptr(qc), _ptr(qlow), _ptr(qhigh), qlow.size, _ptr(first)))
        return first

    # ---- query side, device tensors (no host round trip except the hit total) --------------------------
    def count_overlaps_device(self, qlow, qhigh, qchrom=None, offsets=None):
        """offsets int64[q+1] on the device (exclusive prefix of hit counts); asynchronous."""
        self._ensure_built()
        q = qlow.numel()
        _check_dev_tensor(qlow, "qlow")
        _check_dev_tensor(qhigh, "qhigh", q)
        if qchrom is not None:
            _check_dev_tensor(qchrom, "qchrom", q)
        if offsets is None:
            offsets = torch.empty(q + 1, dtype=torch.int64, device=qlow.device)
        _check_dev_tensor(offsets, "offsets", q + 1, 8)
        s = torch.cuda.current_stream(qlow.device).cuda_stream
        capi.check(self._L.bivx_count_dev(self._h, _tptr(qchrom), _tptr(qlow), _tptr(qhigh), q, _tptr(offsets),
                                          C.c_void_p(s)))
        return offsets

    def fill_overlaps_device(self, qlow, qhigh, offsets, hits, qchrom=None, sort_by_id: bool = False):
        """Writes hit ids into `hits` (int32/uint32 device tensor with >= offsets[-1] elements); asynchronous."""
        q = qlow.numel()
        _check_dev_tensor(offsets, "offsets", q + 1, 8)
        _check_dev_tensor(hits, "hits")
        s = C.c_void_p(torch.cuda.current_stream(qlow.device).cuda_stream)
        capi.check(self._L.bivx_fill_dev(self._h, _tptr(qchrom), _tptr(qlow), _tptr(qhigh), q, _tptr(offsets),
                                         _tptr(hits), s))
        if sort_by_id:
            capi.check(self._L.bivx_sort_hits_dev(self._h, _tptr(offsets), _tptr(hits), q, s))
        return hits

    def query_workspace_bytes(self, q: int) -> int:
        return int(self._L.bivx_query_workspace_bytes(int(q)))

    @staticmethod
    def device_filter(kind: int, max_dist: int, use_strand: bool = True, query_aux=None, interval_aux=None,
                      svtype: int = 0):
        """bivx_filter over DEVICE tensors (fused sv2nl check_condition, include/bivx.h). Keep the tensors alive."""
        for name, t in (("query_aux", query_aux), ("interval_aux", interval_aux)):
            if t is not None:
                _check_dev_tensor(t, name)
        return capi.Filter(kind, max_dist, 1 if use_strand else 0, int(svtype),
                           None if query_aux is None else query_aux.data_ptr(),
                           None if interval_aux is None else interval_aux.data_ptr())

    def query_device(self, qlow, qhigh, offsets, hits, workspace=None, qchrom=None, sort_by_id: bool = False,
                     flt=None):
        """Single-pass count+prefix+fill into caller-owned buffers (bivx_query_dev_s); asynchronous.
        sort_by_id: ids ascend inside every query (ordered by the same kernel), else index order.
        offsets[-1] is the true hit total even if it exceeds hits.numel() (then only a prefix was written).
        flt: optional device_filter(...)."""
        self._ensure_built()
        q = qlow.numel()
        _check_dev_tensor(qlow, "qlow")
        _check_dev_tensor(qhigh, "qhigh", q)
        if qchrom is not None:
            _check_dev_tensor(qchrom, "qchrom", q)
        _check_dev_tensor(offsets, "offsets", q + 1, 8)
        _check_dev_tensor(hits, "hits")
        if workspace is not None:
            _check_dev_tensor(workspace, "workspace", None, 1)
        s = C.c_void_p(torch.cuda.current_stream(qlow.device).cuda_stream)
        capi.check(self._L.bivx_query_dev_s(self._h, _tptr(qchrom), _tptr(qlow), _tptr(qhigh), q,
                                            None if flt is None else C.byref(flt), 1 if sort_by_id else 0,
                                            _tptr(offsets), _tptr(hits), hits.numel(), _tptr(workspace),
                                            0 if workspace is None else workspace.numel(), s))
        return offsets, hits

    def query_device_unordered(self, qlow, qhigh, begin, count, hits, total, workspace=None, qchrom=None, flt=None):
        """Single pass without the canonical CSR (bivx_query_dev_u); asynchronous. Query i's hits are
        hits[begin[i] : begin[i] + count[i]] (index order); workgroups reserve their output ranges in arrival
        order, so the layout differs from call to call while the sets do not. total (int64[1]) receives the number
        of ids reserved; if it exceeds hits.numel() only a prefix of the buffer was written."""
        self._ensure_built()
        q = qlow.numel()
        _check_dev_tensor(qlow, "qlow")
        _check_dev_tensor(qhigh, "qhigh", q)
        if qchrom is not None:
            _check_dev_tensor(qchrom, "qchrom", q)
        _check_dev_tensor(begin, "begin", q, 8)
        _check_dev_tensor(count, "count", q)
        _check_dev_tensor(hits, "hits")
        _check_dev_tensor(total, "total", 1, 8)
        if workspace is not None:
            _check_dev_tensor(workspace, "workspace", None, 1)
        s = C.c_void_p(torch.cuda.current_stream(qlow.device).cuda_stream)
        capi.check(self._L.bivx_query_dev_u(self._h, _tptr(qchrom), _tptr(qlow), _tptr(qhigh), q,
                                            None if flt is None else C.byref(flt), _tptr(begin), _tptr(count),
                                            _tptr(hits), hits.numel(), _tptr(total), _tptr(workspace),
                                            0 if workspace is None else workspace.numel(), s))
        return begin, count, hits, total

    def self_overlaps_device(self, offsets, hits, sort_by_id: bool = False):
        """The index overlapped with itself (bivx_self_overlaps_dev): query i is appended interval i. offsets int64[n+1]
        and hits are caller-owned device tensors; offsets[-1] is the true total even if it exceeds hits.numel().
        Same CSR as query_device(low, high, qchrom=chrom) of the appended columns; asynchronous."""
        self._ensure_built()
        n = self.size()
        _check_dev_tensor(offsets, "offsets", n + 1, 8)
        _check_dev_tensor(hits, "hits")
        s = C.c_void_p(torch.cuda.current_stream(offsets.device).cuda_stream)
        capi.check(self._L.bivx_self_overlaps_dev(self._h, 1 if sort_by_id else 0, _tptr(offsets), _tptr(hits),
                                                  hits.numel(), s))
        return offsets, hits

    def query_sharded_device(self, qlow, qhigh, qchrom=None, sort_by_id: bool = False):
        """A sharded handle's batch with the result left on the device (bivx_query_sharded_dev): every device answers its
        chromosomes, RCCL gathers the CSRs into devices[0]. Host arrays in; returns torch tensors on devices[0]
        (offsets int64[rows + 1], hit ids int32[total] — global ids —, query_of_row int32[rows]) COPIED out of the handle's
        buffers, and whether the blocks travelled through RCCL."""
        qlow, qhigh = _u32(qlow), _u32(qhigh)
        qchrom = _u32(qchrom) if qchrom is not None else None
        res = capi.ShardedResult()
        capi.check(self._L.bivx_query_sharded_dev(self._h, _ptr(qchrom), _ptr(qlow), _ptr(qhigh), qlow.size, int(sort_by_id),
                                                  C.byref(res)))
        dev = torch.device("cuda", res.device)
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

        def take(ptr, n, dtype, itemsize):
            t = torch.empty(n, dtype=dtype, device=dev)
            if n:
                with torch.cuda.device(dev):
                    rc = hip.hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(ptr), n * itemsize, 3)  # device to device
                if rc != 0:
                    raise RuntimeError(f"hipMemcpy of the gathered result failed: {rc}")
            return t

        off = take(res.d_offsets, res.rows + 1, torch.int64, 8)
        hits = take(res.d_hit_ids, res.total, torch.int32, 4)
        rows = take(res.d_query_of_row, res.rows, torch.int32, 4)
        return off, hits, rows, bool(res.used_rccl)

    def find_overlaps_device(self, qlow, qhigh, qchrom=None, sort_by_id: bool = False):
        """(offsets int64[q+1], hits int32[H]) as device tensors. One host sync to size the hit buffer."""
        offsets = self.count_overlaps_device(qlow, qhigh, qchrom)
        total = int(offsets[-1].item())
        hits = torch.empty(max(total, 1), dtype=torch.int32, device=qlow.device)
        self.fill_overlaps_device(qlow, qhigh, offsets, hits, qchrom, sort_by_id)
        return offsets, hits[:total]

    def find_overlap_device(self, qlow, qhigh, qchrom=None):
        self._ensure_built()
        q = qlow.numel()
        first = torch.empty(q, dtype=torch.int32, device=qlow.device)
        s = C.c_void_p(torch.cuda.current_stream(qlow.device).cuda_stream)
        capi.check(self._L.bivx_any_dev(self._h, _tptr(qchrom), _tptr(qlow), _tptr(qhigh), q, _tptr(first), s))
        return first
