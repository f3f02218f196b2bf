"""bivx_self_overlaps_dev — the index overlapped with itself (BASELINE config 5: a whole-genome SV call set against
itself; reference pattern: one tree from a file's records, find_overlaps for each of the same records,
mapper.hpp:199-218). The library answers in the index's own order through k_query_pipe_dense (counts, offsets, lists)
and writes every list where its interval's id puts it. It must give the SAME CSR as the general call with the appended
columns as the batch — offsets and ids, list by list in the same (index) order — and the oracle's sets."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _general(idx, torch, low, high, chrom, sort_by_id):
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    d_lo, d_hi, d_c = to(low), to(high), to(chrom)
    n = low.size
    off = idx.count_overlaps_device(d_lo, d_hi, d_c)
    H = int(off[-1].item())
    hits = torch.full((max(H, 1),), -1, dtype=torch.int32, device=dev)
    off2 = torch.empty(n + 1, dtype=torch.int64, device=dev)
    idx.query_device(d_lo, d_hi, off2, hits, qchrom=d_c, sort_by_id=sort_by_id)
    idx.stream_status()
    assert torch.equal(off, off2)
    return off2, hits, H


def _self(idx, torch, n, cap, sort_by_id):
    dev = torch.device("cuda:0")
    off = torch.full((n + 1,), -1, dtype=torch.int64, device=dev)
    hits = torch.full((max(cap, 1),), -1, dtype=torch.int32, device=dev)
    idx.self_overlaps_device(off, hits[:cap] if cap else hits[:0], sort_by_id=sort_by_id)
    idx.stream_status()
    return off, hits


def _data(seed, n, nchrom=3, span=40_000_000, lmax=1000, inverted=0.0, long_every=0):
    rng = np.random.default_rng(seed)
    chrom = np.sort(rng.integers(0, nchrom, n)).astype(np.uint32)
    low = rng.integers(0, span, n).astype(np.uint32)
    ln = rng.integers(0, lmax, n)
    if long_every:
        ln[::long_every] = rng.integers(50_000, 2_000_000, ln[::long_every].size)
    high = (low + ln).astype(np.uint32)
    if inverted:
        sw = rng.random(n) < inverted
        low[sw], high[sw] = high[sw], low[sw]
    return chrom, low, high


@pytest.mark.parametrize("case", ["packed", "inverted entries", "dense", "lists of a hundred", "lists of thousands"])
def test_self_overlaps_equals_the_general_call(case, oracle):
    import torch
    from binary_amd import IntervalIndex
    if case == "packed":
        chrom, low, high = _data(1, 400_000)
    elif case == "inverted entries":          # no packed records: every slice goes through the general enumeration
        chrom, low, high = _data(2, 200_000, inverted=0.05)
    elif case == "dense":                      # ~40 ids per query: windows of more than 32 slots
        chrom, low, high = _data(3, 300_000, span=4_000_000)
    elif case == "lists of a hundred":         # 60-180 ids per query: beyond the sorting networks, runs of a few lists
        chrom, low, high = _data(5, 70_000, span=600_000, nchrom=1)
        high = (low + np.random.default_rng(5).integers(600, 1000, low.size)).astype(np.uint32)    # (one length class)
    else:                                      # ~1 600 ids per query, no crowded cell: lists longer than k_permute_lines
        chrom, low, high = _data(4, 80_000, span=5_000_000, nchrom=2)                  # puts together in LDS
        high = (low + np.random.default_rng(4).integers(70_000, 100_000, low.size)).astype(np.uint32)  # (one length class)
    n = low.size
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high, chrom)
        idx.build()
        assert idx.stats()["n_segments"] == len(np.unique(chrom))    # one length class: the fast path's territory
        for sort_by_id in (False, True):
            g_off, g_hits, H = _general(idx, torch, low, high, chrom, sort_by_id)
            s_off, s_hits = _self(idx, torch, n, H, sort_by_id)
            assert torch.equal(s_off, g_off), (case, sort_by_id)
            assert torch.equal(s_hits[:H], g_hits[:H]), (case, sort_by_id)
        # offsets only
        o_off, _ = _self(idx, torch, n, 0, False)
        assert torch.equal(o_off, g_off)
        # a buffer smaller than the result: the true total in offsets, nothing written at or beyond the capacity (what
        # lies below it is unspecified for this entry point: the lists pass through a scratch of the same capacity)
        g_off, g_hits, H = _general(idx, torch, low, high, chrom, False)
        cap = H // 3
        dev = torch.device("cuda:0")
        off = torch.empty(n + 1, dtype=torch.int64, device=dev)
        buf = torch.full((cap + 64,), -7, dtype=torch.int32, device=dev)
        idx.self_overlaps_device(off, buf[:cap], sort_by_id=False)
        idx.stream_status()
        assert torch.equal(off, g_off)
        assert bool((buf[cap:] == -7).all())
        # and the sets are the oracle's (a sample of queries, per chromosome trees)
        off_h = g_off.cpu().numpy()
        hits_h = g_hits.cpu().numpy().view(np.uint32).astype(np.int64)
        for c in np.unique(chrom):
            ids = np.nonzero(chrom == c)[0]
            t = oracle.OracleTree(low[ids], high[ids])
            for i in ids[:: max(1, ids.size // 150)]:
                exp = np.sort(ids[t.find_overlaps(int(low[i]), int(high[i]))])
                assert np.array_equal(np.sort(hits_h[off_h[i]:off_h[i + 1]]), exp), (case, int(i))


def test_self_overlaps_small_and_multiclass_take_the_general_path(oracle):
    import torch
    from binary_amd import IntervalIndex
    for chrom, low, high in (_data(5, 5_000), _data(6, 120_000, long_every=25)):
        n = low.size
        with IntervalIndex(0) as idx:
            idx.insert_node(low, high, chrom)
            idx.build()
            g_off, g_hits, H = _general(idx, torch, low, high, chrom, True)
            s_off, s_hits = _self(idx, torch, n, H, True)
            assert torch.equal(s_off, g_off) and torch.equal(s_hits[:H], g_hits[:H])
            cnt = np.diff(s_off.cpu().numpy())
            assert (cnt >= 1).all() or (low > high).any()   # an interval with low <= high meets itself


def test_self_overlaps_after_rebuild_and_on_two_streams():
    import torch
    from binary_amd import IntervalIndex
    chrom, low, high = _data(8, 300_000)
    n = low.size
    half = n // 2
    with IntervalIndex(0) as idx:
        idx.insert_node(low[:half], high[:half], chrom[:half])
        idx.build()
        g_off, g_hits, H = _general(idx, torch, low[:half], high[:half], chrom[:half], False)
        s_off, s_hits = _self(idx, torch, half, H, False)
        assert torch.equal(s_off, g_off) and torch.equal(s_hits[:H], g_hits[:H])
        idx.insert_node(low[half:], high[half:], chrom[half:])     # the cached query batch is stale now
        idx.build()
        g_off, g_hits, H = _general(idx, torch, low, high, chrom, False)
        res = []
        for st in (torch.cuda.Stream(), torch.cuda.Stream()):
            with torch.cuda.stream(st):
                res.append(_self(idx, torch, n, H, False))
        torch.cuda.synchronize()
        for s_off, s_hits in res:
            assert torch.equal(s_off, g_off) and torch.equal(s_hits[:H], g_hits[:H])


def test_a_total_beyond_2_to_the_38_leaves_the_offsets_exact():
    """The slot-order pass keeps a list's length above its 38-bit position in one word (query_pipe.hip, kSelfPosBits). With
    640 000 intervals that nearly all overlap each other the call has 3.6e11 pairs — more than 2^38 — and a caller that
    offers a small buffer must still get every offset, and the true total in offsets[n], to come back with a larger one
    (bivx.h): a position beyond the buffer may not run into the length bits. Counts against sort + searchsorted
    (interval_tree.hpp:119-121)."""
    import torch
    from binary_amd import IntervalIndex
    rng = np.random.default_rng(38)
    n = 640_000
    low = rng.permutation(np.arange(n, dtype=np.uint64) * 1500 + 7).astype(np.uint32)   # distinct, spread lows: no crowded cell
    high = (low.astype(np.uint64) + 960_000_000 + rng.integers(0, 1000, n)).astype(np.uint32)
    cnt = np.searchsorted(np.sort(low), high, "right") - np.searchsorted(np.sort(high), low, "left")
    total = int(cnt.sum())
    assert total > (1 << 38)
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        assert idx.stats()["n_segments"] == 1
        off = torch.full((n + 1,), -1, dtype=torch.int64, device="cuda:0")
        hits = torch.full((4096,), -1, dtype=torch.int32, device="cuda:0")
        idx.self_overlaps_device(off, hits)
        idx.stream_status()
        off = off.cpu().numpy()
    assert int(off[-1]) == total
    assert np.array_equal(np.diff(off), cnt)
    first = np.flatnonzero((low <= high[0]) & (high >= low[0]))[:4096 if cnt[0] >= 4096 else int(cnt[0])]
    assert np.array_equal(np.sort(hits.cpu().numpy().view(np.uint32)[: first.size]), np.sort(first)) or cnt[0] > 4096
