"""the (chrom, start, end, svtype) SoA. Intervals appended with an svtype are partitioned by
(chromosome, svtype) in the built index; a query that names a type meets exactly the intervals of that type — what the
reference gets by filtering records before insertion (mapper.hpp:153-156) into one tree per (mapper, chromosome).
Checked against brute force over the typed subset and against one separate index per type."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(seed, n=60_000, q=40_000, nchrom=5, ntypes=5, span=2_000_000, long_every=0):
    rng = np.random.default_rng(seed)
    chrom = rng.integers(0, nchrom, n).astype(np.uint32)
    low = rng.integers(0, span, n).astype(np.uint32)
    ln = rng.integers(0, 800, n)
    if long_every:
        ln[::long_every] = rng.integers(10_000, 400_000, ln[::long_every].size)  # several length classes per partition
    high = (low + ln).astype(np.uint32)
    typ = rng.integers(0, ntypes, n).astype(np.uint8)   # type 0 = untyped
    qc = rng.integers(0, nchrom + 1, q).astype(np.uint32)  # one chromosome id the index does not have
    qlo = rng.integers(0, span, q).astype(np.uint32)
    qhi = (qlo + rng.integers(0, 1500, q)).astype(np.uint32)
    return chrom, low, high, typ, qc, qlo, qhi


def _brute_counts(chrom, low, high, sel, qc, qlo, qhi):
    """per-query hit counts over the intervals selected by the boolean mask `sel`, chromosome by chromosome"""
    from oracle import ivtree_oracle as oracle
    cnt = np.zeros(qlo.size, np.int64)
    for c in np.unique(qc):
        m = sel & (chrom == c)
        qm = qc == c
        if m.any():
            cnt[qm] = oracle.count_overlaps_numpy(low[m], high[m], qlo[qm], qhi[qm])
    return cnt


@pytest.mark.parametrize("long_every", [0, 50])
def test_type_selection_host_and_device(oracle, long_every):
    import torch
    from binary_amd import IntervalIndex
    chrom, low, high, typ, qc, qlo, qhi = _data(3 + long_every, long_every=long_every)
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
    with IntervalIndex(0) as idx:
        half = low.size // 2                     # two appends; the second from device memory
        idx.insert_node(low[:half], high[:half], chrom[:half], svtype=typ[:half])
        idx.insert_node(to(low[half:]), to(high[half:]), to(chrom[half:]),
                        svtype=torch.from_numpy(typ[half:]).to(dev))
        idx.build()
        assert idx.num_types() == int(typ.max()) + 1
        ids = np.arange(0, low.size, 37, dtype=np.uint32)
        assert np.array_equal(idx.get_svtypes(ids), typ[ids])
        d_qc, d_qlo, d_qhi = to(qc), to(qlo), to(qhi)
        for t in (0, 1, 2, 3, 4, 9):
            sel = np.ones(low.size, bool) if t == 0 else typ == t
            exp_cnt = _brute_counts(chrom, low, high, sel, qc, qlo, qhi)
            off, hits = idx.find_overlaps(qlo, qhi, qc, svtype=t)               # host entry point
            assert np.array_equal(np.diff(off.astype(np.int64)), exp_cnt), t
            h = hits.astype(np.int64)
            qid = np.repeat(np.arange(qlo.size), exp_cnt)
            assert np.all(sel[h]) and np.all(chrom[h] == qc[qid])
            assert np.all(low[h] <= qhi[qid]) and np.all(high[h] >= qlo[qid])
            for k in range(0, qlo.size, 997):                                   # ids ascend: sets are equal
                assert np.all(np.diff(h[off[k]:off[k + 1]]) > 0)
            # device entry points: single pass with a type filter equals the host path bit for bit
            d_off = torch.empty(qlo.size + 1, dtype=torch.int64, device=dev)
            d_hits = torch.empty(max(int(off[-1]), 1), dtype=torch.int32, device=dev)
            idx.query_device(d_qlo, d_qhi, d_off, d_hits, qchrom=d_qc, sort_by_id=True,
                             flt=IntervalIndex.type_filter(t) if t else None)
            idx.stream_status()
            assert np.array_equal(d_off.cpu().numpy().astype(np.uint64), off)
            assert np.array_equal(d_hits.cpu().numpy()[:int(off[-1])].view(np.uint32), hits)


def test_one_typed_index_equals_one_index_per_type():
    from binary_amd import IntervalIndex
    chrom, low, high, typ, qc, qlo, qhi = _data(11, n=30_000, q=20_000, ntypes=4)
    typ = np.maximum(typ, 1).astype(np.uint8)    # every interval typed, as in sv2nl (DUP / INV / BND)
    with IntervalIndex(0) as one:
        one.insert_node(low, high, chrom, svtype=typ)
        one.build()
        for t in (1, 2, 3):
            m = typ == t
            ids_of = np.nonzero(m)[0]
            with IntervalIndex(0) as sep:
                sep.insert_node(low[m], high[m], chrom[m])
                off_s, hits_s = sep.find_overlaps(qlo, qhi, qc)
            off_o, hits_o = one.find_overlaps(qlo, qhi, qc, svtype=t)
            assert np.array_equal(off_o, off_s)
            assert np.array_equal(hits_o, ids_of[hits_s.astype(np.int64)].astype(np.uint32))


def test_untyped_appends_keep_working_and_mix():
    from binary_amd import IntervalIndex
    rng = np.random.default_rng(5)
    low = rng.integers(0, 100_000, 5000).astype(np.uint32)
    high = (low + rng.integers(0, 300, 5000)).astype(np.uint32)
    q = rng.integers(0, 100_000, 3000).astype(np.uint32)
    with IntervalIndex(0) as idx:
        idx.insert_node(low[:2000], high[:2000])                                  # untyped
        idx.build()
        assert idx.num_types() == 1
        off0, _ = idx.find_overlaps(q, q)
        off7, hits7 = idx.find_overlaps(q, q, svtype=7)                           # a type the index does not hold
        assert int(off7[-1]) == 0 and hits7.size == 0
        idx.insert_node(low[2000:], high[2000:], svtype=np.full(3000, 2, np.uint8))  # typed append afterwards
        idx.build()
        assert idx.num_types() == 3
        assert np.array_equal(idx.get_svtypes(np.array([0, 1999, 2000, 4999, 5000], np.uint32)),
                              np.array([0, 0, 2, 2, 0xFF], np.uint8))
        off_any, _ = idx.find_overlaps(q, q)
        off2, hits2 = idx.find_overlaps(q, q, svtype=2)
        assert np.all(hits2 >= 2000)
        assert np.array_equal(np.diff(off_any.astype(np.int64)),
                              np.diff(off0.astype(np.int64)) + np.diff(off2.astype(np.int64)))
