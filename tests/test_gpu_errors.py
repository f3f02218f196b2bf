"""Error surfacing of the single-pass kernels (include/bivx.h: BIVX_E_TIMEOUT, bivx_stream_status) and the ordering
of device-side appends. A CSR computed with a wrong cross-workgroup prefix must never come back with rc 0
(every status surfaced). The error paths are driven once each through an injection knob — the
bound of a prefix wait shrunk to nothing (BIVX_PREFIX_WAIT_LOG2) behind a deliberately slow first tile, and a
corrupted ticket word (bivx_debug_corrupt_workspace) — not by repeating runs."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _heavy_then_light(n=200_000, light=100_000, seed=5):
    """1024 chromosome-wide queries (all of tile 0: each walks the whole index with its wavefront) followed by point
    queries (tiles that finish their counting long before tile 0 does)."""
    rng = np.random.default_rng(seed)
    low = rng.integers(0, 1_000_000, n).astype(np.uint32)
    high = (low + rng.integers(1, 2_000, n)).astype(np.uint32)
    p = rng.integers(0, 1_002_000, light).astype(np.uint32)
    qlo = np.concatenate([np.zeros(1024, np.uint32), p])
    qhi = np.concatenate([np.full(1024, 2_000_000, np.uint32), p])
    return low, high, qlo, qhi


def test_heavy_first_tile_default_bound_is_exact_and_reports_nothing(oracle):
    import torch
    from binary_amd import IntervalIndex
    low, high, qlo, qhi = _heavy_then_light()
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        dev = torch.device("cuda:0")
        to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
        off = idx.count_overlaps_device(to(qlo), to(qhi))
        idx.stream_status()  # synchronises; raises if any wait gave up
        cnt = np.diff(off.cpu().numpy())
        assert np.array_equal(cnt, oracle.count_overlaps_numpy(low, high, qlo, qhi))
        assert np.all(cnt[:1024] == low.size)
        assert idx.stats()["prefix_timeouts"] == 0


def test_expired_prefix_wait_is_an_error_not_a_result(oracle):
    import torch
    from binary_amd import IntervalIndex, capi
    low, high, qlo, qhi = _heavy_then_light()
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        dev = torch.device("cuda:0")
        to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
        d_qlo, d_qhi = to(qlo), to(qhi)
        off = idx.count_overlaps_device(d_qlo, d_qhi)       # allocates the workspace, normal bound
        idx.stream_status()
        os.environ["BIVX_PREFIX_WAIT_LOG2"] = "1"            # every wait that is not satisfied at once expires
        try:
            idx.count_overlaps_device(d_qlo, d_qhi, offsets=off)
            with pytest.raises(capi.BivxError) as e:
                idx.stream_status()
            assert e.value.code == capi.E_TIMEOUT
            # the host-pointer entry points make the same check before they return
            offsets = np.zeros(qlo.size + 1, dtype=np.uint64)
            rc = idx._L.bivx_count(idx._h, None, qlo.ctypes.data_as(C.c_void_p), qhi.ctypes.data_as(C.c_void_p),
                                   qlo.size, offsets.ctypes.data_as(C.c_void_p))
            assert rc == capi.E_TIMEOUT
        finally:
            del os.environ["BIVX_PREFIX_WAIT_LOG2"]
        assert idx.stats()["prefix_timeouts"] == 2
        # reported once; the index stays usable and the repeated call is exact
        idx.count_overlaps_device(d_qlo, d_qhi, offsets=off)
        idx.stream_status()
        assert np.array_equal(np.diff(off.cpu().numpy()), oracle.count_overlaps_numpy(low, high, qlo, qhi))
        assert idx.stats()["prefix_timeouts"] == 2


def test_inconsistent_workspace_is_an_error_and_is_repaired(oracle):
    import torch
    from binary_amd import IntervalIndex, capi, synth
    low, high = synth.gen_intervals(50_000, 10_000_000, 1000)
    qlo, qhi = synth.gen_range_queries(70_000, 10_000_000, 1000)
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        dev = torch.device("cuda:0")
        to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
        d_qlo, d_qhi = to(qlo), to(qhi)
        off, hits = idx.find_overlaps_device(d_qlo, d_qhi, sort_by_id=True)
        idx.stream_status()
        s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        capi.check(idx._L.bivx_debug_corrupt_workspace(idx._h, s))
        off2 = torch.full_like(off, -1)
        hits2 = torch.full_like(hits, -1)
        idx.query_device(d_qlo, d_qhi, off2, hits2, sort_by_id=True)
        with pytest.raises(capi.BivxError) as e:
            idx.stream_status()
        assert e.value.code == capi.E_TIMEOUT
        idx.query_device(d_qlo, d_qhi, off2, hits2, sort_by_id=True)   # the workspace is cleared before this launch
        idx.stream_status()
        assert torch.equal(off, off2) and torch.equal(hits, hits2)
        t = oracle.OracleTree(low, high)
        off_o, hits_o = t.find_overlaps_batch(qlo, qhi)
        assert np.array_equal(off2.cpu().numpy().astype(np.uint64), off_o)
        assert np.array_equal(hits2.cpu().numpy().view(np.uint32).astype(np.int64), oracle.sorted_csr(off_o, hits_o))


def test_append_dev_back_to_back_growth_without_host_sync(oracle):
    """bivx_append_dev is asynchronous on the caller's stream: a later append that grows the arrays must not lose an
    earlier one that is still in flight (the growth copy used to run on the index's own, unordered stream)."""
    import torch
    from binary_amd import IntervalIndex, capi
    rng = np.random.default_rng(11)
    n1, n2 = 1000, 3_000_000   # the first fits the initial capacity, the second forces a growth
    low = rng.integers(0, 50_000_000, n1 + n2).astype(np.uint32)
    high = (low + rng.integers(0, 500, n1 + n2)).astype(np.uint32)
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream(dev)
    d_low = torch.from_numpy(low.view(np.int32)).to(dev)
    d_high = torch.from_numpy(high.view(np.int32)).to(dev)
    torch.cuda.synchronize()
    with IntervalIndex(0) as idx:
        s = C.c_void_p(side.cuda_stream)
        with torch.cuda.stream(side):
            junk = torch.empty(64 << 20, dtype=torch.int32, device=dev)
            for _ in range(8):
                junk.add_(1)           # keeps the side stream busy so that the first append is still queued
            p = lambda t, o: C.c_void_p(t.data_ptr() + 4 * o)
            capi.check(idx._L.bivx_append_dev(idx._h, None, p(d_low, 0), p(d_high, 0), n1, s))
            capi.check(idx._L.bivx_append_dev(idx._h, None, p(d_low, n1), p(d_high, n1), n2, s))
        side.synchronize()
        idx.build()
        assert idx.size() == n1 + n2
        ids = np.arange(0, n1 + n2, 997, dtype=np.uint32)
        ids[:n1 // 997 + 1] = np.arange(n1 // 997 + 1)
        _, lo_b, hi_b = idx.get_intervals(np.arange(n1, dtype=np.uint32))
        assert np.array_equal(lo_b, low[:n1]) and np.array_equal(hi_b, high[:n1])
        _, lo_b, hi_b = idx.get_intervals(ids)
        assert np.array_equal(lo_b, low[ids]) and np.array_equal(hi_b, high[ids])
        q = rng.integers(0, 50_000_000, 5000).astype(np.uint32)
        off, _ = idx.find_overlaps(q, q)
        assert np.array_equal(np.diff(off.astype(np.int64)), oracle.count_overlaps_numpy(low, high, q, q))
