"""k_query_pipe_ms (binary_amd/csrc/query_pipe.hip): the pipelined single-pass kernel for several length classes per
chromosome, type partitions, fused filters and many ids per query — every lane walks its own windows, once to count and
once more, when the slice's place in the output is known, to lay the ids out in the wavefront's stage.
Against k_query_fused (BIVX_PIPE=0 sends the same call there) bit for bit — offsets AND ids, index order and ascending —
and against the closed-interval predicate itself (interval_tree.hpp:119-121; the filters: mapper.cpp:50-79,144-156 as
restated in tests/test_sv2nl.py). BIVX_PIPE=2 lifts the batch-size threshold so that small batches take the kernel too."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = 248_956_422


class _env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            os.environ[k] = str(v)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _to(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to("cuda:0")


def _run(idx, mode, dq, dqh, dqc, H, by_id, flt=None, cap=None):
    import torch
    q = dq.numel()
    with _env(BIVX_PIPE=mode):
        off = torch.full((q + 1,), -1, dtype=torch.int64, device=dq.device)
        hits = torch.full((max(H if cap is None else cap, 1),), -1, dtype=torch.int32, device=dq.device)
        idx.query_device(dq, dqh, off, hits, qchrom=dqc, sort_by_id=by_id, flt=flt)
        idx.stream_status()
        return off.cpu().numpy(), hits.cpu().numpy().view(np.uint32)


def _sv_like(maxlen, n, q, seed=5):
    """tools/skewed_bench.py's workload: log-uniform lengths 50 bp .. maxlen on one chromosome, point queries"""
    rng = np.random.default_rng(seed)
    low = rng.integers(0, G - 10_000_001, size=n).astype(np.uint32)
    high = low + np.exp(rng.uniform(np.log(50), np.log(maxlen), size=n)).astype(np.uint32)
    qlo = rng.integers(0, G, size=q).astype(np.uint32)
    return low, high, qlo


@pytest.mark.parametrize("maxlen,q", [(1e4, 1_000_000), (1e5, 300_000), (1e6, 60_000)])
def test_sv_like_spectra_equal_the_fused_kernel_and_the_predicate(oracle, maxlen, q):
    """several length classes per chromosome; 1e5: the longest class is not packed; 1e6: windows beyond the lane limit
    (the slices go through the general enumeration and k_fill_slices)"""
    from binary_amd import IntervalIndex
    low, high, qlo = _sv_like(maxlen, 1_000_000, q)
    qlo[:7] = [0, 1, G - 1, low[3], high[3], high[3] + 1, 0xFFFFFFFF]
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        assert idx.stats()["n_segments"] >= 2
        dq = _to(qlo)
        H = int(idx.count_overlaps_device(dq, dq)[-1].item())
        with _env(BIVX_PIPE=2):
            assert idx.query_kernel_name(q, H, False) == "k_query_pipe_ms"
        with _env(BIVX_PIPE=0):
            assert idx.query_kernel_name(q, H, False) == "k_query_fused"
        # default routing: batches of 0.8 M queries and more on an index whose windows the kernel walks itself
        assert idx.query_kernel_name(1_000_000, H, False) == ("k_query_pipe_ms" if maxlen <= 1e6 else "k_query_fused")
        assert idx.query_kernel_name(100_000, H, False) == "k_query_fused"
        for by_id in (False, True):
            off_p, hits_p = _run(idx, 2, dq, dq, None, H, by_id)
            off_f, hits_f = _run(idx, 0, dq, dq, None, H, by_id)
            assert np.array_equal(off_p, off_f) and int(off_p[-1]) == H
            assert np.array_equal(hits_p[:H], hits_f[:H]), by_id
    assert np.array_equal(np.diff(off_p), oracle.count_overlaps_numpy(low, high, qlo, qlo))
    for k in np.r_[np.arange(7), np.arange(7, q, q // 300)]:
        exp = np.flatnonzero((low <= qlo[k]) & (high >= qlo[k]))
        assert np.array_equal(hits_p[off_p[k]:off_p[k + 1]], exp), k     # (ascending ids from the last run)


def test_ragged_batches_capacity_prefix_and_unknown_chromosomes():
    from binary_amd import IntervalIndex
    low, high, qlo = _sv_like(3e4, 300_000, 70_001, seed=9)      # 73 tiles of 960 queries, the last one ragged
    rng = np.random.default_rng(1)
    chrom = rng.integers(0, 3, low.size).astype(np.uint32)
    qc = rng.integers(0, 5, qlo.size).astype(np.uint32)           # chromosomes 3 and 4 do not exist: no hits
    qhi = (qlo + rng.integers(0, 3000, qlo.size) * (rng.random(qlo.size) < 0.5)).astype(np.uint32)
    qhi[::97] = qlo[::97] - np.minimum(qlo[::97], 5)              # inverted queries: legal, mostly empty
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high, chrom)
        idx.build()
        dq, dqh, dqc = _to(qlo), _to(qhi), _to(qc)
        H = int(idx.count_overlaps_device(dq, dqh, dqc)[-1].item())
        off_p, hits_p = _run(idx, 2, dq, dqh, dqc, H, False)
        off_f, hits_f = _run(idx, 0, dq, dqh, dqc, H, False)
        assert np.array_equal(off_p, off_f) and np.array_equal(hits_p[:H], hits_f[:H])
        assert np.all(np.diff(off_p)[qc >= 3] == 0)
        for k in range(0, qlo.size, 211):
            exp = np.flatnonzero((chrom == qc[k]) & (low <= qhi[k]) & (high >= qlo[k]))
            assert np.array_equal(np.sort(hits_p[off_p[k]:off_p[k + 1]]), exp), k
        # a buffer smaller than the result: every offset as before, the ids up to the capacity as before, nothing beyond
        cap = H // 3
        off_c, hits_c = _run(idx, 2, dq, dqh, dqc, H, False, cap=cap)
        assert np.array_equal(off_c, off_p) and np.array_equal(hits_c[:cap], hits_p[:cap])
        # and no buffer at all: the offsets alone
        import torch
        with _env(BIVX_PIPE=2):
            off0 = idx.count_overlaps_device(dq, dqh, dqc)
            assert np.array_equal(off0.cpu().numpy(), off_p)


def test_type_partitions_all_types_and_one_type():
    """a typed index keeps one segment range per (chromosome, type): a query for every type walks them all (several
    segments per query), a query for one type only that type's"""
    from binary_amd import IntervalIndex
    rng = np.random.default_rng(3)
    n, q = 400_000, 120_000
    chrom = rng.integers(0, 4, n).astype(np.uint32)
    typ = rng.integers(1, 4, n).astype(np.uint8)
    low = rng.integers(0, 30_000_000, n).astype(np.uint32)
    high = (low + np.exp(rng.uniform(np.log(30), np.log(40_000), n))).astype(np.uint32)
    qc = rng.integers(0, 4, q).astype(np.uint32)
    qlo = rng.integers(0, 30_000_000, q).astype(np.uint32)
    qhi = (qlo + rng.integers(0, 2000, q)).astype(np.uint32)
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high, chrom, svtype=typ)
        idx.build()
        dq, dqh, dqc = _to(qlo), _to(qhi), _to(qc)
        for t in (0, 2):
            flt = IntervalIndex.type_filter(t) if t else None
            import torch
            off0 = torch.empty(q + 1, dtype=torch.int64, device=dq.device)
            with _env(BIVX_PIPE=0):
                idx.query_device(dq, dqh, off0, torch.empty(1, dtype=torch.int32, device=dq.device), qchrom=dqc, flt=flt)
            H = int(off0[-1].item())
            off_p, hits_p = _run(idx, 2, dq, dqh, dqc, H, True, flt=flt)
            off_f, hits_f = _run(idx, 0, dq, dqh, dqc, H, True, flt=flt)
            assert np.array_equal(off_p, off_f) and np.array_equal(hits_p[:H], hits_f[:H]), t
            sel = np.ones(n, bool) if t == 0 else typ == t
            for k in range(0, q, 401):
                exp = np.flatnonzero(sel & (chrom == qc[k]) & (low <= qhi[k]) & (high >= qlo[k]))
                assert np.array_equal(hits_p[off_p[k]:off_p[k + 1]], exp), (t, k)


@pytest.mark.parametrize("kind_name", ["DUP", "INV", "TRA"])
def test_fused_filters_equal_the_fused_kernel_and_the_host_predicates(kind_name):
    from binary_amd import IntervalIndex, capi
    kind = getattr(capi, "FILTER_SV2NL_" + kind_name)
    rng = np.random.default_rng(17)
    n, q, d = 200_000, 90_000, 3000
    low = rng.integers(0, 6_000_000, n).astype(np.uint32)
    high = low + rng.integers(0, 6000, n).astype(np.uint32)
    longer = rng.random(n) < 0.03                                # a few long SVs: several length classes
    high[longer] = low[longer] + rng.integers(100_000, 900_000, int(longer.sum())).astype(np.uint32)
    if kind == capi.FILTER_SV2NL_TRA:                            # TRA trees hold low > high records too
        inv = rng.random(n) < 0.1
        low, high = np.where(inv, high, low).astype(np.uint32), np.where(inv, low, high).astype(np.uint32)
    qlo = rng.integers(0, 6_000_000, q).astype(np.uint32)
    qhi = qlo + rng.integers(0, 6000, q).astype(np.uint32)
    iaux = ((rng.integers(0, 6, n) << 1) | rng.integers(0, 2, n)).astype(np.uint32)
    qaux = (((rng.integers(0, 6, q) << 1) | rng.integers(0, 2, q)) if kind == capi.FILTER_SV2NL_TRA
            else rng.integers(0, 4, q)).astype(np.uint32)
    ad = lambda a, b: np.where(a >= b, a - b, b - a)
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        assert idx.stats()["n_segments"] >= 2
        dq, dqh = _to(qlo), _to(qhi)
        t_qaux, t_iaux = _to(qaux), _to(iaux)
        flt = IntervalIndex.device_filter(kind, d, True, t_qaux, t_iaux)
        H0 = int(idx.count_overlaps_device(dq, dqh)[-1].item())
        off_p, hits_p = _run(idx, 2, dq, dqh, None, H0, True, flt=flt)
        off_f, hits_f = _run(idx, 0, dq, dqh, None, H0, True, flt=flt)
        H = int(off_p[-1])
        assert H > 100 and np.array_equal(off_p, off_f) and np.array_equal(hits_p[:H], hits_f[:H])
        off_u, hits_u = _run(idx, 2, dq, dqh, None, H0, True)     # unfiltered, then the reference's predicates on the host
    qid = np.repeat(np.arange(q), np.diff(off_u))
    h = hits_u[: int(off_u[-1])]
    a_lo, a_hi = qlo[qid].astype(np.int64), qhi[qid].astype(np.int64)
    b_lo, b_hi = low[h].astype(np.int64), high[h].astype(np.int64)
    if kind == capi.FILTER_SV2NL_DUP:
        keep = (b_lo <= a_lo) & (b_hi >= a_hi) & (ad(a_lo, b_lo) <= d) & (ad(a_hi, b_hi) <= d)
    elif kind == capi.FILTER_SV2NL_INV:
        c1, c2 = (b_lo <= a_lo) & (b_hi >= a_hi), (a_lo <= b_lo) & (a_hi >= b_hi)
        near = (ad(a_lo, b_lo) <= d) & (ad(a_hi, b_hi) <= d)
        s1, s2 = (qaux[qid] & 1) != 0, (qaux[qid] & 2) != 0
        keep = ~c1 & ~c2 & near & np.where(a_lo <= b_lo, s1 & ~s2, ~s1 & s2)
    else:
        qa, ia = qaux[qid], iaux[h]
        q1, q2 = np.where(qa & 1, a_hi, a_lo), np.where(qa & 1, a_lo, a_hi)
        i1, i2 = np.where(ia & 1, b_hi, b_lo), np.where(ia & 1, b_lo, b_hi)
        keep = ((qa >> 1) == (ia >> 1)) & (ad(q1, i1) <= d) & (ad(q2, i2) <= d)
    assert np.array_equal(hits_p[:H], h[keep])
    assert np.array_equal(np.diff(off_p), np.bincount(qid[keep], minlength=q))


@pytest.mark.parametrize("order", ["generated", "sorted"])
def test_many_ids_per_query_on_one_length_class(oracle, order):
    """one segment per chromosome but ~17 ids per query (config 5's density): generation order is this kernel's, a
    position-sorted batch k_query_pipe_dense's (this one is launched behind it and returns)"""
    from binary_amd import IntervalIndex, synth
    n = 600_000
    low, high = synth.gen_intervals(n, 37_000_000, 1000, 0)
    if order == "sorted":
        o = np.argsort(low, kind="stable")
        low, high = low[o], high[o]
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        assert idx.stats()["n_segments"] == 1
        dq, dqh = _to(low), _to(high)
        H = int(idx.count_overlaps_device(dq, dqh)[-1].item())
        assert H > 6 * n
        with _env(BIVX_PIPE=2):
            assert idx.query_kernel_name(n, H, False) == "k_query_pipe_dense|k_query_pipe_ms"
        for by_id in (False, True):
            off_p, hits_p = _run(idx, 2, dq, dqh, None, H, by_id)
            off_f, hits_f = _run(idx, 0, dq, dqh, None, H, by_id)
            assert np.array_equal(off_p, off_f) and np.array_equal(hits_p[:H], hits_f[:H]), by_id
    assert np.array_equal(np.diff(off_p), oracle.count_overlaps_numpy(low, high, low, high))


@pytest.mark.parametrize("many", [False, True])
def test_short_windows_in_packed_and_plain_classes(many):
    """Segments whose 64 windows all end within 48 slots are evaluated lane by lane (group_scan's own-window trips: 16 slots
    per trip, windows that begin at odd slots, a second and third trip), in classes that store 16 + 16-bit records and in a
    class that does not (lengths beyond 65 535: ids from id[]). `many`: piles that put some lists beyond the sixteen keep slots,
    so that slices are walked again with the ids going straight to the output. Against k_query_fused bit for bit and against
    the predicate (interval_tree.hpp:119-121)."""
    from binary_amd import IntervalIndex
    rng = np.random.default_rng(31 + many)
    n = 120_000
    low = rng.integers(0, G - 2_000_000, size=n).astype(np.uint32)
    ln = np.where(rng.random(n) < 0.7, rng.integers(50, 3000, n), rng.integers(70_000, 900_000, n))
    if many:  # forty piles of 30 short intervals each: lists of 30+ ids for the queries that fall there
        at = rng.integers(0, G - 2_000_000, size=40)
        low[:1200] = (at[:, None] + rng.integers(0, 40, (40, 30))).ravel()
        ln[:1200] = 500
    high = (low + ln).astype(np.uint32)
    q = 60_000
    qlo = rng.integers(0, G, size=q).astype(np.uint32)
    if many:
        qlo[:4000] = (at[rng.integers(0, 40, 4000)] + rng.integers(0, 300, 4000)).astype(np.uint32)
        qlo = qlo[rng.permutation(q)]
    qhi = (qlo + rng.integers(0, 2000, q) * (rng.random(q) < 0.5)).astype(np.uint32)
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        assert idx.stats()["n_segments"] >= 2
        dq, dqh = _to(qlo), _to(qhi)
        H = int(idx.count_overlaps_device(dq, dqh)[-1].item())
        with _env(BIVX_PIPE=2):
            assert idx.query_kernel_name(q, H, False) == "k_query_pipe_ms"
        for by_id in (False, True):
            off_p, hits_p = _run(idx, 2, dq, dqh, None, H, by_id)
            off_f, hits_f = _run(idx, 0, dq, dqh, None, H, by_id)
            assert np.array_equal(off_p, off_f) and int(off_p[-1]) == H
            assert np.array_equal(hits_p[:H], hits_f[:H]), by_id
        cnt = np.diff(off_p)
        if many:
            assert cnt.max() > 16
        for k in np.r_[np.argsort(cnt)[-20:], np.arange(0, q, q // 200)]:
            exp = np.flatnonzero((low <= qhi[k]) & (high >= qlo[k]))
            assert np.array_equal(hits_p[off_p[k]:off_p[k + 1]], exp), k
        cap = H // 2   # a buffer smaller than the result (index order): the ids up to the capacity as before, nothing beyond
        off_i, hits_i = _run(idx, 2, dq, dqh, None, H, False)
        off_c, hits_c = _run(idx, 2, dq, dqh, None, H, False, cap=cap)
        assert np.array_equal(off_c, off_i) and np.array_equal(hits_c[:cap], hits_i[:cap])
