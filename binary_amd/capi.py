"""ctypes binding of the C ABI in include/bivx.h (libbivx.so, hand-written HIP for gfx950).

There is no fallback of any kind: if the shared library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

from ._build import LIB_PATH

BIVX_NO_HIT = 0xFFFFFFFF
E_INVALID, E_HIP, E_NOMEM, E_STATE, E_RANGE, E_TIMEOUT, E_COMM = -1, -2, -3, -4, -5, -6, -7
ABI_VERSION = 0x00020003

EXPORTS = (
    "bivx_abi_version", "bivx_last_error", "bivx_create", "bivx_create_sharded", "bivx_num_devices", "bivx_device_of_chrom", "bivx_destroy", "bivx_device", "bivx_append",
    "bivx_append_dev", "bivx_append_typed", "bivx_append_typed_dev", "bivx_num_types", "bivx_get_svtypes", "bivx_clear", "bivx_build", "bivx_is_built", "bivx_size", "bivx_num_chroms",
    "bivx_get_intervals", "bivx_count", "bivx_fill", "bivx_count_dev",
    "bivx_fill_dev", "bivx_query_workspace_bytes", "bivx_query_dev", "bivx_sort_hits_dev", "bivx_any", "bivx_any_dev", "bivx_get_stats",
    "bivx_count_f", "bivx_fill_f", "bivx_count_dev_f", "bivx_fill_dev_f", "bivx_query_dev_f", "bivx_query_dev_s", "bivx_query_dev_u",
    "bivx_find_overlaps", "bivx_free", "bivx_self_overlaps_dev", "bivx_stream_status", "bivx_query_kernel_name", "bivx_debug_corrupt_workspace", "bivx_release_pooled",
    "bivx_query_sharded_dev",
)


class BivxError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"libbivx error {code}: {text}")
        self.code = code


FILTER_NONE, FILTER_SV2NL_DUP, FILTER_SV2NL_INV, FILTER_SV2NL_TRA = 0, 1, 2, 3


class Filter(C.Structure):
    """bivx_filter (include/bivx.h): fused sv2nl check_condition."""
    _fields_ = [("kind", C.c_uint32), ("max_dist", C.c_uint32), ("use_strand", C.c_uint32), ("svtype", C.c_uint32),
                ("query_aux", C.c_void_p), ("interval_aux", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("n_intervals", C.c_uint64), ("n_chroms", C.c_uint32), ("n_segments", C.c_uint32),
                ("n_cells", C.c_uint64), ("index_bytes", C.c_uint64), ("staging_bytes", C.c_uint64),
                ("build_ms", C.c_double), ("prefix_timeouts", C.c_uint64)]


class ShardedResult(C.Structure):
    """bivx_sharded_result (include/bivx.h): the gathered device-resident CSR of bivx_query_sharded_dev."""
    _fields_ = [("d_offsets", C.c_void_p), ("d_hit_ids", C.c_void_p), ("d_query_of_row", C.c_void_p),
                ("rows", C.c_uint64), ("total", C.c_uint64), ("device", C.c_int), ("used_rccl", C.c_int)]


_lib = None


def load() -> C.CDLL:
    """Loads binary_amd/libbivx.so; raises if it is not there (build it with __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: the HIP extension has not been built "
                          "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, sz, u32p, u64p = C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p
    L.bivx_abi_version.restype = C.c_uint32
    L.bivx_last_error.restype = C.c_char_p
    L.bivx_create.argtypes = [C.POINTER(vp), C.c_int]
    L.bivx_create_sharded.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int]
    L.bivx_num_devices.argtypes = [vp]
    L.bivx_device_of_chrom.argtypes = [vp, C.c_uint32]
    L.bivx_destroy.argtypes = [vp]
    L.bivx_destroy.restype = None
    L.bivx_device.argtypes = [vp]
    L.bivx_append.argtypes = [vp, u32p, u32p, u32p, sz]
    L.bivx_append_dev.argtypes = [vp, u32p, u32p, u32p, sz, vp]
    L.bivx_append_typed.argtypes = [vp, u32p, u32p, u32p, vp, sz]
    L.bivx_append_typed_dev.argtypes = [vp, u32p, u32p, u32p, vp, sz, vp]
    L.bivx_num_types.argtypes = [vp]
    L.bivx_num_types.restype = C.c_uint32
    L.bivx_get_svtypes.argtypes = [vp, u32p, sz, vp]
    L.bivx_clear.argtypes = [vp]
    L.bivx_build.argtypes = [vp]
    L.bivx_is_built.argtypes = [vp]
    L.bivx_size.argtypes = [vp]
    L.bivx_size.restype = sz
    L.bivx_num_chroms.argtypes = [vp]
    L.bivx_num_chroms.restype = C.c_uint32
    L.bivx_get_intervals.argtypes = [vp, u32p, sz, u32p, u32p, u32p]
    L.bivx_count.argtypes = [vp, u32p, u32p, u32p, sz, u64p]
    L.bivx_fill.argtypes = [vp, u32p, u32p, u32p, sz, u64p, u32p, C.c_int]
    L.bivx_count_dev.argtypes = [vp, u32p, u32p, u32p, sz, u64p, vp]
    L.bivx_stream_status.argtypes = [vp, vp]
    L.bivx_query_kernel_name.argtypes = [vp, sz, C.c_uint64, C.c_int, vp]
    L.bivx_query_kernel_name.restype = C.c_char_p
    L.bivx_debug_corrupt_workspace.argtypes = [vp, vp]
    L.bivx_fill_dev.argtypes = [vp, u32p, u32p, u32p, sz, u64p, u32p, vp]
    L.bivx_query_workspace_bytes.argtypes = [sz]
    L.bivx_query_workspace_bytes.restype = sz
    L.bivx_query_dev.argtypes = [vp, u32p, u32p, u32p, sz, u64p, u32p, C.c_uint64, vp, sz, vp]
    L.bivx_sort_hits_dev.argtypes = [vp, u64p, u32p, sz, vp]
    L.bivx_any.argtypes = [vp, u32p, u32p, u32p, sz, u32p]
    L.bivx_any_dev.argtypes = [vp, u32p, u32p, u32p, sz, u32p, vp]
    L.bivx_get_stats.argtypes = [vp, C.POINTER(Stats)]
    fp = C.POINTER(Filter)
    L.bivx_find_overlaps.argtypes = [vp, u32p, u32p, u32p, sz, fp, C.c_int, u64p, C.POINTER(C.POINTER(C.c_uint32))]
    L.bivx_free.argtypes = [vp]
    L.bivx_free.restype = None
    L.bivx_count_f.argtypes = [vp, u32p, u32p, u32p, sz, fp, u64p]
    L.bivx_fill_f.argtypes = [vp, u32p, u32p, u32p, sz, fp, u64p, u32p, C.c_int]
    L.bivx_count_dev_f.argtypes = [vp, u32p, u32p, u32p, sz, fp, u64p, vp]
    L.bivx_fill_dev_f.argtypes = [vp, u32p, u32p, u32p, sz, fp, u64p, u32p, vp]
    L.bivx_query_dev_f.argtypes = [vp, u32p, u32p, u32p, sz, fp, u64p, u32p, C.c_uint64, vp, sz, vp]
    L.bivx_query_dev_s.argtypes = [vp, u32p, u32p, u32p, sz, fp, C.c_int, u64p, u32p, C.c_uint64, vp, sz, vp]
    L.bivx_self_overlaps_dev.argtypes = [vp, C.c_int, u64p, u32p, C.c_uint64, vp]
    L.bivx_query_dev_u.argtypes = [vp, u32p, u32p, u32p, sz, fp, u64p, u32p, u32p, C.c_uint64, u64p, vp, sz, vp]
    L.bivx_query_sharded_dev.argtypes = [vp, u32p, u32p, u32p, sz, C.c_int, C.POINTER(ShardedResult)]
    if L.bivx_abi_version() >> 16 != ABI_VERSION >> 16:
        raise ImportError("libbivx.so ABI major version mismatch")
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != 0:
        raise BivxError(rc, load().bivx_last_error().decode("utf-8", "replace"))
