"""GPU parity: the HIP path (through the C ABI of include/bivx.h) against the CPU oracle and the golden
vectors. Bit-exact bar: per query the hit SET (ascending ids) equals the reference tree's hit set.

All tests here need a real MI355X (-m gpu). The oracle is only the checker.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))
RT, FX = GOLD["reference_tests"], GOLD["fixture10"]
M32 = 0xFFFFFFFF


@pytest.fixture(scope="module")
def IntervalIndex():
    from binary_amd import IntervalIndex as cls
    return cls


def gpu_csr(IntervalIndex, low, high, qlow, qhigh, chrom=None, qchrom=None, sort_by_id=True):
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high, chrom)
        idx.build()
        return idx.find_overlaps(qlow, qhigh, qchrom, sort_by_id=sort_by_id)


def oracle_csr_sorted(oracle, low, high, qlow, qhigh):
    t = oracle.OracleTree(low, high)
    off, hits = t.find_overlaps_batch(qlow, qhigh, nthreads=4)
    return off, oracle.sorted_csr(off, hits)


def assert_same(oracle, low, high, qlow, qhigh, IntervalIndex):
    off_g, hits_g = gpu_csr(IntervalIndex, low, high, qlow, qhigh)
    off_o, hits_o = oracle_csr_sorted(oracle, low, high, qlow, qhigh)
    assert np.array_equal(off_g, off_o)
    assert np.array_equal(hits_g.astype(np.int64), hits_o)


# ---- the reference's own known answers, through the C ABI ----------------------------------------------

def test_reference_fixture10(IntervalIndex, oracle):
    with IntervalIndex(0) as idx:
        idx.insert_node(FX["low"], FX["high"])
        assert idx.size() == RT["fixture10_size"]["value"]
        q1, q2 = RT["find_overlaps_7_25_count"], RT["find_overlaps_15_25_count"]
        off, hits = idx.find_overlaps([q1["q"][0], q2["q"][0]], [q1["q"][1], q2["q"][1]])
        assert np.diff(off.astype(np.int64)).tolist() == [q1["count"], q2["count"]]
        for k, name in enumerate(("find_overlaps_7_25_preorder", "find_overlaps_15_25_preorder")):
            got = sorted([FX["low"][i], FX["high"][i]] for i in hits[int(off[k]):int(off[k + 1])])
            assert got == sorted(GOLD["survey_8c"][name])
        # find_overlap: existence must match (22,25) -> some hit, (100,111) -> none
        first = idx.find_overlap([22, 100], [25, 111])
        assert first[0] != M32 and first[1] == M32
        # (22,25) overlaps exactly [15,23] and [25,30]; the reference returns [15,23] (id 5), the smaller id
        assert [FX["low"][first[0]], FX["high"][first[0]]] == RT["find_overlap_22_25"]["hit"]


def test_reference_duplicates(IntervalIndex):
    d = RT["dup4"]
    off, hits = gpu_csr(IntervalIndex, d["low"], d["high"], [d["q"][0]], [d["q"][1]])
    assert int(off[1]) == d["count"] and hits.tolist() == [0, 1, 2, 3]


def test_reference_seq500(IntervalIndex, oracle):
    s = RT["seq500"]
    low = np.arange(s["start"], s["stop"], s["step"], dtype=np.uint32)
    q = np.arange(0, 1010, 7, dtype=np.uint32)
    assert_same(oracle, low, low + s["len"], q, q + 5, IntervalIndex)


# ---- edge cases ------------------------------------------------------------------------------------------

def test_empty_index_and_empty_batch(IntervalIndex):
    with IntervalIndex(0) as idx:
        assert idx.empty()
        off, hits = idx.find_overlaps([0, 5], [10, M32])
        assert off.tolist() == [0, 0, 0] and hits.size == 0
        assert idx.find_overlap([3], [4]).tolist() == [M32]
        idx.insert_node([1], [2])
        off, hits = idx.find_overlaps(np.zeros(0, np.uint32), np.zeros(0, np.uint32))
        assert off.tolist() == [0] and hits.size == 0
        off, hits = idx.find_overlaps([2, 3], [2, 3])  # one node
        assert off.tolist() == [0, 1, 1] and hits.tolist() == [0]


def test_boundary_touching(IntervalIndex, oracle):
    low = np.array([10, 20, 20, 30, 31], dtype=np.uint32)
    high = np.array([20, 20, 25, 30, 40], dtype=np.uint32)
    qlo = np.array([0, 9, 20, 21, 25, 26, 30, 41, 0], dtype=np.uint32)
    qhi = np.array([9, 10, 20, 24, 30, 29, 31, 50, M32], dtype=np.uint32)
    assert_same(oracle, low, high, qlo, qhi, IntervalIndex)


def test_u32_extremes(IntervalIndex, oracle):
    low = np.array([0, 0, M32, M32 - 1, 5, M32 - 1000], dtype=np.uint32)
    high = np.array([0, M32, M32, M32, 5, M32 - 1], dtype=np.uint32)
    qlo = np.array([0, M32, 0, 6, 1, M32 - 500], dtype=np.uint32)
    qhi = np.array([0, M32, M32, M32 - 2, 4, M32 - 400], dtype=np.uint32)
    assert_same(oracle, low, high, qlo, qhi, IntervalIndex)


def test_low_greater_than_high_intervals_and_queries(IntervalIndex, oracle):
    """TraMapper inserts unvalidated BND records (mapper.cpp:158-170): low > high nodes are legal."""
    rng = np.random.default_rng(5)
    n = 3000
    low = rng.integers(0, 20000, size=n).astype(np.uint32)
    high = low + rng.integers(0, 400, size=n).astype(np.uint32)
    swap = rng.random(n) < 0.2
    low2, high2 = np.where(swap, high, low).astype(np.uint32), np.where(swap, low, high).astype(np.uint32)
    qlo = rng.integers(0, 20000, size=2000).astype(np.uint32)
    qhi = qlo + rng.integers(0, 400, size=2000).astype(np.uint32)
    qswap = rng.random(2000) < 0.1  # and a few inverted queries
    qlo2, qhi2 = np.where(qswap, qhi, qlo).astype(np.uint32), np.where(qswap, qlo, qhi).astype(np.uint32)
    assert_same(oracle, low2, high2, qlo2, qhi2, IntervalIndex)


@pytest.mark.parametrize("n,q", [(1, 1), (2, 2), (63, 65), (64, 64), (65, 63), (1000, 1000), (100000, 100000)])
def test_random_sets_vs_oracle(IntervalIndex, oracle, n, q):
    from binary_amd import synth
    L = 1 << 20 if n <= 1000 else 248956422 // 10
    low, high = synth.gen_intervals(n, L, 1000, chrom_index=n % 7)
    qlo, qhi = synth.gen_range_queries(q, L, 1000, chrom_index=q % 5)
    assert_same(oracle, low, high, qlo, qhi, IntervalIndex)


def test_mixed_lengths_force_length_classes_and_heavy_windows(IntervalIndex, oracle):
    """A few chromosome-scale intervals among many short ones: several length classes, windows that
    take the wavefront-cooperative path, hit lists long enough for every tier of the id sort."""
    rng = np.random.default_rng(9)
    L = 5_000_000
    n_short = 60000
    low = rng.integers(0, L, size=n_short).astype(np.uint32)
    high = low + rng.integers(0, 200, size=n_short).astype(np.uint32)
    big_low = rng.integers(0, L // 2, size=40).astype(np.uint32)
    big_high = big_low + rng.integers(L // 4, L // 2, size=40).astype(np.uint32)
    mid_low = rng.integers(0, L, size=3000).astype(np.uint32)
    mid_high = mid_low + rng.integers(5000, 60000, size=3000).astype(np.uint32)
    low = np.concatenate([low, big_low, mid_low])
    high = np.concatenate([high, big_high, mid_high])
    perm = rng.permutation(low.size)
    low, high = low[perm], high[perm]
    qlo = rng.integers(0, L, size=3000).astype(np.uint32)
    qhi = qlo + rng.integers(0, 300, size=3000).astype(np.uint32)
    # long queries: hundreds to tens of thousands of hits each
    qlo[:40] = rng.integers(0, L // 2, size=40)
    qhi[:40] = qlo[:40] + rng.integers(10_000, L // 2, size=40).astype(np.uint32)
    qlo[40], qhi[40] = 0, M32  # everything
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        assert idx.stats()["n_segments"] >= 2
        off_g, hits_g = idx.find_overlaps(qlo, qhi)
        off_n, hits_n = idx.find_overlaps(qlo, qhi, sort_by_id=False)
        first = idx.find_overlap(qlo, qhi)
    off_o, hits_o = oracle_csr_sorted(oracle, low, high, qlo, qhi)
    assert np.array_equal(off_g, off_o)
    assert np.array_equal(hits_g.astype(np.int64), hits_o)
    assert int(np.diff(off_o.astype(np.int64)).max()) > 2048
    # index-order output is the same set
    assert np.array_equal(off_n, off_o)
    assert np.array_equal(oracle.sorted_csr(off_n, hits_n.astype(np.int32)), hits_o)
    # find_overlap: existence exact, choice = smallest id
    cnt = np.diff(off_o.astype(np.int64))
    exp_first = np.where(cnt > 0, hits_o[np.minimum(off_o[:-1].astype(np.int64), max(hits_o.size - 1, 0))], M32)
    assert np.array_equal(first.astype(np.int64), exp_first)


def test_chromosomes_partition_the_index(IntervalIndex, oracle):
    """One tree per chromosome in sv2nl (mapper.hpp:147-162,199): equal chrom ids meet, others never."""
    rng = np.random.default_rng(21)
    nchrom, n, q = 7, 20000, 15000
    chrom = rng.integers(0, nchrom, size=n).astype(np.uint32)
    chrom[chrom == 3] = 4  # chromosome 3 stays empty
    low = rng.integers(0, 200000, size=n).astype(np.uint32)
    high = low + rng.integers(0, 500, size=n).astype(np.uint32)
    qchrom = rng.integers(0, nchrom + 3, size=q).astype(np.uint32)  # ids 7..9 do not exist in the index
    qlo = rng.integers(0, 200000, size=q).astype(np.uint32)
    qhi = qlo + rng.integers(0, 500, size=q).astype(np.uint32)
    off_g, hits_g = gpu_csr(IntervalIndex, low, high, qlo, qhi, chrom, qchrom)
    cnt_g = np.diff(off_g.astype(np.int64))
    for c in range(nchrom + 3):
        sel_i = np.nonzero(chrom == c)[0]
        sel_q = np.nonzero(qchrom == c)[0]
        if sel_i.size == 0:
            assert cnt_g[sel_q].sum() == 0
            continue
        t = oracle.OracleTree(low[sel_i], high[sel_i])
        off_o, hits_o = t.find_overlaps_batch(qlo[sel_q], qhi[sel_q])
        assert np.array_equal(cnt_g[sel_q], np.diff(off_o.astype(np.int64)))
        for k in (0, sel_q.size // 2, sel_q.size - 1):
            qi = sel_q[k]
            got = hits_g[int(off_g[qi]):int(off_g[qi + 1])].astype(np.int64)
            exp = np.sort(sel_i[hits_o[int(off_o[k]):int(off_o[k + 1])]])
            assert np.array_equal(got, exp)
        got_all = np.concatenate([hits_g[int(off_g[qi]):int(off_g[qi + 1])] for qi in sel_q]).astype(np.int64)
        exp_all = sel_i[oracle.sorted_csr(off_o, hits_o)] if hits_o.size else np.zeros(0, np.int64)
        # oracle.sorted_csr sorts local ids; local order == global order because sel_i is ascending
        assert np.array_equal(got_all, exp_all)


def test_incremental_append_rebuilds(IntervalIndex, oracle):
    rng = np.random.default_rng(2)
    low = rng.integers(0, 50000, size=4000).astype(np.uint32)
    high = low + rng.integers(0, 100, size=4000).astype(np.uint32)
    qlo = rng.integers(0, 50000, size=1000).astype(np.uint32)
    qhi = qlo + 50
    with IntervalIndex(0) as idx:
        idx.insert_node(low[:1500], high[:1500])
        o1, h1 = idx.find_overlaps(qlo, qhi)
        idx.insert_node(low[1500:], high[1500:])  # ids continue
        o2, h2 = idx.find_overlaps(qlo, qhi)
        c, lo_back, hi_back = idx.get_intervals(np.array([0, 1499, 1500, 3999], dtype=np.uint32))
        assert lo_back.tolist() == low[[0, 1499, 1500, 3999]].tolist()
        assert hi_back.tolist() == high[[0, 1499, 1500, 3999]].tolist()
    off_a, hits_a = oracle_csr_sorted(oracle, low[:1500], high[:1500], qlo, qhi)
    off_b, hits_b = oracle_csr_sorted(oracle, low, high, qlo, qhi)
    assert np.array_equal(o1, off_a) and np.array_equal(h1.astype(np.int64), hits_a)
    assert np.array_equal(o2, off_b) and np.array_equal(h2.astype(np.int64), hits_b)


def test_device_resident_path_matches_host_path(IntervalIndex):
    import torch
    from binary_amd import synth
    low, high = synth.gen_intervals(50000, 10_000_000, 1000)
    qlo, qhi = synth.gen_range_queries(30000, 10_000_000, 1000)
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
    with IntervalIndex(0) as idx:
        idx.insert_node(to(low), to(high))  # device append
        idx.build()
        off_h, hits_h = idx.find_overlaps(qlo, qhi, sort_by_id=True)
        off_d, hits_d = idx.find_overlaps_device(to(qlo), to(qhi), sort_by_id=True)
        first_d = idx.find_overlap_device(to(qlo), to(qhi))
        first_h = idx.find_overlap(qlo, qhi)
        torch.cuda.synchronize()
    assert np.array_equal(off_d.cpu().numpy().astype(np.uint64), off_h)
    assert np.array_equal(hits_d.cpu().numpy().view(np.uint32), hits_h)
    assert np.array_equal(first_d.cpu().numpy().view(np.uint32), first_h)


# ---- full BASELINE sizes through size-independent properties ------------------------------------------------

def _check_full(IntervalIndex, oracle, data, point):
    """count == predicate count (sort+searchsorted), every reported pair satisfies the predicate, ids
    strictly ascending inside a query  =>  the hit set is exactly the reference's."""
    with IntervalIndex(0) as idx:
        idx.insert_node(data["low"], data["high"], data["chrom"])
        idx.build()
        off, hits = idx.find_overlaps(data["qlow"], data["qhigh"], data["qchrom"], sort_by_id=True)
    cnt = np.diff(off.astype(np.int64))
    exp = np.zeros_like(cnt)
    for c in np.unique(data["qchrom"]):
        qi = data["qchrom"] == c
        ii = data["chrom"] == c
        exp[qi] = oracle.count_overlaps_numpy(data["low"][ii], data["high"][ii], data["qlow"][qi], data["qhigh"][qi])
    assert np.array_equal(cnt, exp)
    qid = np.repeat(np.arange(cnt.size), cnt)
    assert np.all(data["chrom"][hits] == data["qchrom"][qid])
    assert np.all(data["qlow"][qid] <= data["high"][hits]) and np.all(data["low"][hits] <= data["qhigh"][qid])
    same_q = qid[1:] == qid[:-1]
    assert np.all(hits[1:][same_q].astype(np.int64) > hits[:-1][same_q].astype(np.int64))
    return int(off[-1])


def test_config2_full_size_1M_x_1M_point(IntervalIndex, oracle):
    from binary_amd import synth
    L = int(synth.HG38_LENGTHS[0])
    low, high = synth.gen_intervals(1_000_000, L, 1000, 0)
    qlo, qhi = synth.gen_point_queries(1_000_000, L, 0)
    z = np.zeros(1_000_000, np.uint32)
    H = _check_full(IntervalIndex, oracle, dict(chrom=z, low=low, high=high, qchrom=z, qlow=qlo, qhigh=qhi), True)
    # and against the reference-algorithm oracle itself on the full batch (a few seconds on 4 threads)
    off_g, hits_g = gpu_csr(IntervalIndex, low, high, qlo, qhi)
    off_o, hits_o = oracle_csr_sorted(oracle, low, high, qlo, qhi)
    assert int(off_o[-1]) == H
    assert np.array_equal(off_g, off_o) and np.array_equal(hits_g.astype(np.int64), hits_o)


def test_config3_full_size_24_chroms_10M_x_10M_range(IntervalIndex, oracle):
    from binary_amd import synth
    data = synth.gen_genome(10_000_000, 10_000_000, 1000, point_queries=False)
    H = _check_full(IntervalIndex, oracle, data, False)
    assert 25_000_000 < H < 40_000_000  # SURVEY §8d expects about 32 M


# ---- single-pass kernel (bivx_query_dev) ---------------------------------------------------------------------

def _fused(idx, torch, qlo_t, qhi_t, cap, qchrom_t=None):
    q = qlo_t.numel()
    off = torch.empty(q + 1, dtype=torch.int64, device=qlo_t.device)
    hits = torch.full((max(cap, 1),), -1, dtype=torch.int32, device=qlo_t.device)
    ws = torch.empty(idx.query_workspace_bytes(q), dtype=torch.uint8, device=qlo_t.device)
    idx.query_device(qlo_t, qhi_t, off, hits, ws, qchrom=qchrom_t, sort_by_id=False)
    torch.cuda.synchronize()
    return off, hits


def test_single_pass_equals_two_pass_and_oracle(IntervalIndex, oracle):
    import torch
    from binary_amd import synth
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    data = synth.gen_genome(300_000, 200_003, 1000)  # 24 chromosomes, ragged last tile
    # a few long intervals and long queries so that both the lane path and the wavefront path run
    rng = np.random.default_rng(4)
    data["high"][:50] = data["low"][:50] + rng.integers(1_000_000, 50_000_000, size=50).astype(np.uint32)
    data["qhigh"][:30] = data["qlow"][:30] + rng.integers(100_000, 5_000_000, size=30).astype(np.uint32)
    with IntervalIndex(0) as idx:
        idx.insert_node(data["low"], data["high"], data["chrom"])
        idx.build()
        off2, hits2 = idx.find_overlaps(data["qlow"], data["qhigh"], data["qchrom"], sort_by_id=False)
        H = int(off2[-1])
        off1, hits1 = _fused(idx, torch, to(data["qlow"]), to(data["qhigh"]), H, to(data["qchrom"]))
        # same CSR, same index order, bit for bit
        assert np.array_equal(off1.cpu().numpy().astype(np.uint64), off2)
        assert np.array_equal(hits1.cpu().numpy().view(np.uint32)[:H], hits2)
        # capacity smaller than H: offsets still exact, the written prefix is exact, nothing beyond it is touched
        cap = H // 3
        off3, hits3 = _fused(idx, torch, to(data["qlow"]), to(data["qhigh"]), cap, to(data["qchrom"]))
        assert np.array_equal(off3.cpu().numpy().astype(np.uint64), off2)
        assert np.array_equal(hits3.cpu().numpy().view(np.uint32)[:cap], hits2[:cap])
        # zero capacity: a pure count
        off4, _ = _fused(idx, torch, to(data["qlow"]), to(data["qhigh"]), 0, to(data["qchrom"]))
        assert int(off4[-1].item()) == H
    # and the set is the reference's, chromosome by chromosome
    for c in (0, 7, 23):
        ii, qi = data["chrom"] == c, data["qchrom"] == c
        t = oracle.OracleTree(data["low"][ii], data["high"][ii])
        off_o, hits_o = t.find_overlaps_batch(data["qlow"][qi], data["qhigh"][qi])
        assert np.array_equal(np.diff(off2.astype(np.int64))[qi], np.diff(off_o.astype(np.int64)))
        sel = np.nonzero(ii)[0]
        got = np.concatenate([np.sort(hits2[int(off2[k]):int(off2[k + 1])]) for k in np.nonzero(qi)[0]])
        assert np.array_equal(got.astype(np.int64), sel[oracle.sorted_csr(off_o, hits_o)])


@pytest.mark.parametrize("span,max_len", [(60_000_000, 1000), (600_000, 3000)])  # ~2 and ~40 ids per query
def test_single_pass_sorted_ids_any_capacity(IntervalIndex, span, max_len):
    """sort_by_id in the single pass: ids ascend inside every query whichever way they are ordered (inside the
    kernel when the buffer says few ids per query, by the follow-up pass otherwise), a buffer smaller than the
    result keeps exact offsets, and nothing past the buffer's capacity is ever written."""
    import torch
    from binary_amd import synth
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    low, high = synth.gen_intervals(200_000, span, max_len)
    low[:40] = 0                                   # a few chromosome-long intervals: wavefront-path queries too
    high[:40] = span
    qlo, qhi = synth.gen_range_queries(150_001, span, max_len)
    qhi[:20] = qlo[:20] + span // 50
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        ref_off, ref_hits = idx.find_overlaps(qlo, qhi, sort_by_id=True)   # two-pass + k_sort_hits
        H = int(ref_off[-1])
        guard = 4096
        for cap in (H, 5 * qlo.size + 7, H + 5 * qlo.size, H // 2, 1000):
            off = torch.empty(qlo.size + 1, dtype=torch.int64, device=dev)
            buf = torch.full((cap + guard,), -2, dtype=torch.int32, device=dev)
            idx.query_device(to(qlo), to(qhi), off, buf[:cap], sort_by_id=True)
            torch.cuda.synchronize()
            assert np.array_equal(off.cpu().numpy().astype(np.uint64), ref_off), cap
            assert bool((buf[cap:] == -2).all()), cap                  # the guard words are untouched
            got = buf[:cap].cpu().numpy().view(np.uint32)
            if cap >= H:
                assert np.array_equal(got[:H], ref_hits), cap
            else:  # every list that fits entirely is exact and ascending; the one the capacity cuts is unspecified
                last = int(np.searchsorted(ref_off, cap, side="right")) - 1  # queries [0, last) fit entirely
                n = int(ref_off[last])
                assert np.array_equal(got[:n], ref_hits[:n]), cap


def test_single_pass_index_owned_workspace(IntervalIndex):
    """workspace=None: the index keeps a self-cleaning prefix workspace per stream; results must not change over
    repeated calls, different batch sizes, two streams, or a rebuild in between."""
    import torch
    from binary_amd import synth
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
    low, high = synth.gen_intervals(300_000, 30_000_000, 1000)
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        side = torch.cuda.Stream()
        for rep, nq in enumerate((1_300_003, 70_000, 1024, 1, 1_300_003)):
            qlo, qhi = synth.gen_range_queries(nq, 30_000_000, 1000, chrom_index=rep)
            ref_off, ref_hits = idx.find_overlaps(qlo, qhi, sort_by_id=False)
            for stream in (torch.cuda.current_stream(), side):
                with torch.cuda.stream(stream):
                    off = torch.empty(nq + 1, dtype=torch.int64, device=dev)
                    hits = torch.empty(max(int(ref_off[-1]), 1), dtype=torch.int32, device=dev)
                    for _ in range(3):
                        idx.query_device(to(qlo), to(qhi), off, hits)
                    stream.synchronize()
                assert np.array_equal(off.cpu().numpy().astype(np.uint64), ref_off)
                assert np.array_equal(hits.cpu().numpy().view(np.uint32)[: int(ref_off[-1])], ref_hits)
            assert idx.stats()["prefix_timeouts"] == 0  # no cross-workgroup wait ever gave up
            if rep == 2:
                idx.insert_node(low[:1000], high[:1000])  # forces a rebuild; workspaces are reset with it
                idx.build()


def test_single_pass_under_hip_graph_capture(IntervalIndex):
    """The device entry points allocate nothing and never synchronise once their buffers exist, so a step can be
    captured into a HIP graph and replayed (torch.cuda.CUDAGraph is a hipGraph on ROCm)."""
    import torch
    from binary_amd import synth
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
    low, high = synth.gen_intervals(200_000, 20_000_000, 1000)
    qlo, qhi = synth.gen_range_queries(100_000, 20_000_000, 1000)
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        ref_off, ref_hits = idx.find_overlaps(qlo, qhi, sort_by_id=True)
        d_qlo, d_qhi = to(qlo), to(qhi)
        off = torch.zeros(qlo.size + 1, dtype=torch.int64, device=dev)
        hits = torch.zeros(int(ref_off[-1]), dtype=torch.int32, device=dev)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(3):  # warm up on the capture stream: creates the index-owned workspace of that stream
                idx.query_device(d_qlo, d_qhi, off, hits, sort_by_id=True)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            idx.query_device(d_qlo, d_qhi, off, hits, sort_by_id=True)
        for _ in range(3):
            off.zero_()
            hits.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert np.array_equal(off.cpu().numpy().astype(np.uint64), ref_off)
            assert np.array_equal(hits.cpu().numpy().view(np.uint32), ref_hits)
        assert idx.stats()["prefix_timeouts"] == 0


def test_single_pass_repeated_calls_are_deterministic(IntervalIndex):
    import torch
    from binary_amd import synth
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
    low, high = synth.gen_intervals(200_000, 20_000_000, 1000)
    qlo, qhi = synth.gen_range_queries(150_001, 20_000_000, 1000)
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        ref = None
        for _ in range(5):
            off, hits = _fused(idx, torch, to(qlo), to(qhi), 1_000_000)
            cur = (off.cpu().numpy().copy(), hits.cpu().numpy()[: int(off[-1].item())].copy())
            if ref is None:
                ref = cur
            assert np.array_equal(cur[0], ref[0]) and np.array_equal(cur[1], ref[1])


def test_clustered_intervals_long_cells_are_trimmed(IntervalIndex, oracle):
    """Positional hotspots: hundreds of thousands of intervals start inside a few directory cells, so directory
    windows are long and the wavefront path trims them with its 64-ary search before scanning."""
    rng = np.random.default_rng(33)
    hot = rng.integers(5_000_000, 5_020_000, size=300_000).astype(np.uint32)          # 20 kbp hotspot
    hot2 = rng.integers(90_000_000, 90_000_400, size=50_000).astype(np.uint32)        # 400 bp hotspot
    bg = rng.integers(0, 200_000_000, size=50_000).astype(np.uint32)
    low = np.concatenate([hot, hot2, bg])
    high = low + rng.integers(0, 300, size=low.size).astype(np.uint32)
    perm = rng.permutation(low.size)
    low, high = low[perm], high[perm]
    qlo = np.concatenate([rng.integers(4_999_000, 5_021_000, size=3000), rng.integers(89_999_900, 90_000_500, size=1000),
                          rng.integers(0, 200_000_000, size=2000)]).astype(np.uint32)
    qhi = qlo + rng.integers(0, 50, size=qlo.size).astype(np.uint32)
    qhi[:5] = qlo[:5] + 15_000   # a few long ones across the hotspot
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        off, hits = idx.find_overlaps(qlo, qhi)
        first = idx.find_overlap(qlo, qhi)
    off_o, hits_o = oracle_csr_sorted(oracle, low, high, qlo, qhi)
    assert np.array_equal(off, off_o) and np.array_equal(hits.astype(np.int64), hits_o)
    cnt = np.diff(off_o.astype(np.int64))
    assert cnt.max() > 50_000
    exp_first = np.where(cnt > 0, hits_o[np.minimum(off_o[:-1].astype(np.int64), hits_o.size - 1)], M32)
    assert np.array_equal(first.astype(np.int64), exp_first)


@pytest.mark.parametrize("max_len", [10_000, 100_000, 1_000_000])
def test_single_pass_over_several_length_classes(IntervalIndex, oracle, max_len):
    """Log-uniform lengths (an SV-like spectrum) make the planner cut 2 .. 7 length classes on one chromosome, so a
    query meets several segments: the single pass records up to three windows per query for its output phase
    and enumerates again beyond that; all of it must give the two-pass CSR and the reference's sets."""
    import torch
    rng = np.random.default_rng(max_len)
    span, n, q = 25_000_000, 100_000, 60_001
    low = rng.integers(0, span - max_len - 1, size=n).astype(np.uint32)
    high = low + np.exp(rng.uniform(np.log(50), np.log(max_len), size=n)).astype(np.uint32)
    qlo = rng.integers(0, span, size=q).astype(np.uint32)
    qhi = qlo + rng.integers(0, 200, size=q).astype(np.uint32)
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        assert idx.stats()["n_segments"] >= 2
        ref_off, ref_hits = idx.find_overlaps(qlo, qhi, sort_by_id=False)          # two-pass, index order
        H = int(ref_off[-1])
        for sort in (False, True):
            off = torch.empty(q + 1, dtype=torch.int64, device=dev)
            hits = torch.full((H,), -1, dtype=torch.int32, device=dev)
            idx.query_device(to(qlo), to(qhi), off, hits, sort_by_id=sort)
            torch.cuda.synchronize()
            assert np.array_equal(off.cpu().numpy().astype(np.uint64), ref_off)
            got = hits.cpu().numpy().view(np.uint32)
            if not sort:
                assert np.array_equal(got, ref_hits)
        assert idx.stats()["prefix_timeouts"] == 0
    # `got` is the id-sorted CSR now: the reference's sets
    t = oracle.OracleTree(low, high)
    off_o, hits_o = t.find_overlaps_batch(qlo, qhi)
    assert np.array_equal(ref_off, off_o.astype(np.uint64))
    assert np.array_equal(got.astype(np.int64), oracle.sorted_csr(off_o, hits_o))


def _check_unordered(begin, count, hits, total, ref_off, ref_hits_sorted, cap):
    """begin/count output against an id-sorted reference CSR: same sets, disjoint ranges, exact total."""
    begin, count = begin.astype(np.int64), count.astype(np.int64)
    ref_cnt = np.diff(ref_off.astype(np.int64))
    assert np.array_equal(count, ref_cnt)
    assert total == int(ref_cnt.sum())
    nz = count > 0
    o = np.argsort(begin[nz], kind="stable")
    b, c = begin[nz][o], count[nz][o]
    assert b.size == 0 or (b[0] == 0 and np.array_equal(b[1:], (b + c)[:-1]))   # the ranges tile [0, total) exactly
    if total <= cap:
        idx = np.repeat(begin - np.cumsum(count) + count, count) + np.arange(int(count.sum()))
        got = hits[idx]                                   # lists gathered in query order
        seg = np.repeat(np.arange(count.size), count)
        order = np.lexsort((got, seg))
        assert np.array_equal(got[order].astype(np.int64), ref_hits_sorted.astype(np.int64))


@pytest.mark.parametrize("nq", [1, 1000, 1024, 70_001, 1_300_003])
def test_unordered_single_pass_same_sets(IntervalIndex, nq):
    """bivx_query_dev_u: per-query (begin, count) instead of a CSR; sets, counts and the total must be those of the
    ordered paths, ranges must tile the buffer, repeated calls and both workspace kinds must agree, and a buffer
    that is too small is never overrun."""
    import torch
    from binary_amd import synth
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    data = synth.gen_genome(200_000, nq, 1000)
    rng = np.random.default_rng(nq)
    data["high"][:30] = data["low"][:30] + rng.integers(1_000_000, 50_000_000, size=30).astype(np.uint32)  # classes
    if nq > 100:
        data["qhigh"][:20] = data["qlow"][:20] + rng.integers(100_000, 5_000_000, size=20).astype(np.uint32)
    with IntervalIndex(0) as idx:
        idx.insert_node(data["low"], data["high"], data["chrom"])
        idx.build()
        ref_off, ref_hits = idx.find_overlaps(data["qlow"], data["qhigh"], data["qchrom"], sort_by_id=True)
        H = int(ref_off[-1])
        ql, qh, qc = to(data["qlow"]), to(data["qhigh"]), to(data["qchrom"])
        guard = 1024
        for rep, (cap, own_ws) in enumerate([(H, False), (H, True), (H + 77, False), (H // 2, False), (0, True)]):
            begin = torch.empty(nq, dtype=torch.int64, device=dev)
            count = torch.empty(nq, dtype=torch.int32, device=dev)
            total = torch.full((1,), -1, dtype=torch.int64, device=dev)
            buf = torch.full((cap + guard,), -2, dtype=torch.int32, device=dev)
            ws = torch.empty(idx.query_workspace_bytes(nq), dtype=torch.uint8, device=dev) if own_ws else None
            for _ in range(2):
                idx.query_device_unordered(ql, qh, begin, count, buf[:cap], total, workspace=ws, qchrom=qc)
            torch.cuda.synchronize()
            assert bool((buf[cap:] == -2).all())
            _check_unordered(begin.cpu().numpy(), count.cpu().numpy(), buf[:cap].cpu().numpy().view(np.uint32),
                             int(total.item()), ref_off, ref_hits, cap)
        # the ordered single pass still works on the same stream afterwards (they share the index's workspace)
        off = torch.empty(nq + 1, dtype=torch.int64, device=dev)
        hits = torch.empty(max(H, 1), dtype=torch.int32, device=dev)
        idx.query_device(ql, qh, off, hits, qchrom=qc, sort_by_id=True)
        assert np.array_equal(off.cpu().numpy().astype(np.uint64), ref_off)
        assert np.array_equal(hits.cpu().numpy().view(np.uint32)[:H], ref_hits)
        assert idx.stats()["prefix_timeouts"] == 0


def test_thousands_of_chromosomes_descriptors_from_global_memory(IntervalIndex):
    """A reference header with alternate contigs has > 3 000 chromosomes (hg38 with alts: 3 366): more chromosomes
    and segments than the kernels stage in LDS, so they read the descriptors from global memory. Every entry
    point against the predicate evaluated per chromosome."""
    import torch
    rng = np.random.default_rng(3366)
    nchrom, n, q = 3366, 120_000, 60_000
    chrom = rng.integers(0, nchrom, size=n).astype(np.uint32)
    low = rng.integers(0, 3_000_000, size=n).astype(np.uint32)
    high = low + rng.integers(0, 2_000, size=n).astype(np.uint32)
    big = rng.choice(n, size=400, replace=False)                      # long intervals: extra length classes
    high[big] = low[big] + rng.integers(200_000, 2_000_000, size=400).astype(np.uint32)
    qchrom = rng.integers(0, nchrom + 50, size=q).astype(np.uint32)   # some ids beyond the index
    qlo = rng.integers(0, 3_000_000, size=q).astype(np.uint32)
    qhi = qlo + rng.integers(0, 3_000, size=q).astype(np.uint32)
    # expected lists, chromosome by chromosome
    exp = [np.zeros(0, np.int64)] * q
    iord, qord = np.argsort(chrom, kind="stable"), np.argsort(qchrom, kind="stable")
    ib = np.searchsorted(chrom[iord], np.arange(nchrom + 51))
    qb = np.searchsorted(qchrom[qord], np.arange(nchrom + 51))
    L64, H64 = low.astype(np.int64), high.astype(np.int64)
    for c in range(nchrom):
        ii, qi = iord[ib[c]:ib[c + 1]], qord[qb[c]:qb[c + 1]]
        if ii.size == 0 or qi.size == 0:
            continue
        hit = (qlo[qi].astype(np.int64)[:, None] <= H64[ii][None, :]) & (L64[ii][None, :] <= qhi[qi].astype(np.int64)[:, None])
        for k, qq in enumerate(qi):
            exp[qq] = np.sort(ii[hit[k]])
    exp_cnt = np.array([e.size for e in exp], dtype=np.int64)
    exp_off = np.concatenate([[0], np.cumsum(exp_cnt)]).astype(np.uint64)
    exp_hits = np.concatenate(exp).astype(np.uint32)
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high, chrom)
        idx.build()
        st = idx.stats()
        assert st["n_chroms"] == nchrom and st["n_segments"] > 128
        off, hits = idx.find_overlaps(qlo, qhi, qchrom, sort_by_id=True)           # count + fill + sort
        assert np.array_equal(off, exp_off) and np.array_equal(hits, exp_hits)
        first = idx.find_overlap(qlo, qhi, qchrom)
        assert np.array_equal(first, np.array([e[0] if e.size else M32 for e in exp], dtype=np.uint32))
        H = int(exp_off[-1])
        d_off = torch.empty(q + 1, dtype=torch.int64, device=dev)
        d_hits = torch.empty(H, dtype=torch.int32, device=dev)
        for sort in (True, False):
            idx.query_device(to(qlo), to(qhi), d_off, d_hits, qchrom=to(qchrom), sort_by_id=sort)
            torch.cuda.synchronize()
            assert np.array_equal(d_off.cpu().numpy().astype(np.uint64), exp_off)
            got = d_hits.cpu().numpy().view(np.uint32)
            if sort:
                assert np.array_equal(got, exp_hits)
            else:
                seg = np.repeat(np.arange(q), exp_cnt)
                assert np.array_equal(got[np.lexsort((got, seg))], exp_hits)
        beg = torch.empty(q, dtype=torch.int64, device=dev)
        cnt = torch.empty(q, dtype=torch.int32, device=dev)
        tot = torch.zeros(1, dtype=torch.int64, device=dev)
        idx.query_device_unordered(to(qlo), to(qhi), beg, cnt, d_hits, tot, qchrom=to(qchrom))
        torch.cuda.synchronize()
        _check_unordered(beg.cpu().numpy(), cnt.cpu().numpy(), d_hits.cpu().numpy().view(np.uint32), int(tot.item()),
                         exp_off, exp_hits, H)
        assert idx.stats()["prefix_timeouts"] == 0


@pytest.mark.parametrize("q,span", [(1, 50_000), (5, 600_000), (2048, 3_000), (2049, 3_000)])
def test_small_batches_take_one_round_trip_or_fall_back(IntervalIndex, oracle, q, span):
    """Up to 2048 unfiltered queries go through one upload, one single-pass launch and one download; a result
    larger than that path's buffer (few queries, very many hits each) falls back to count-then-fill. Same answers."""
    rng = np.random.default_rng(q)
    low = rng.integers(0, 1_000_000, size=60_000).astype(np.uint32)
    high = low + rng.integers(0, 1_000, size=60_000).astype(np.uint32)
    qlo = rng.integers(0, 1_000_000 - span, size=q).astype(np.uint32)
    qhi = qlo + np.uint32(span)
    for sort in (True, False):
        off_g, hits_g = gpu_csr(IntervalIndex, low, high, qlo, qhi, sort_by_id=sort)
        off_o, hits_o = oracle_csr_sorted(oracle, low, high, qlo, qhi)
        assert np.array_equal(off_g, off_o)
        if sort:
            assert np.array_equal(hits_g.astype(np.int64), hits_o)
        else:
            seg = np.repeat(np.arange(q), np.diff(off_o.astype(np.int64)))
            assert np.array_equal(hits_g[np.lexsort((hits_g, seg))].astype(np.int64), hits_o)


@pytest.mark.parametrize("max_tiles", [3, 1100])
def test_chained_launches_of_the_single_pass(IntervalIndex, max_tiles):
    """One launch covers 64 M queries (ordered) or more (unordered); larger batches run as consecutive launches that
    chain through offsets[q_begin] / the workspace's running total. BIVX_MAX_TILES_PER_LAUNCH shrinks a launch so that
    the chain runs at test size: 3 tiles (flat sweep) and 1100 tiles (two-level sweep) per launch."""
    import torch
    from binary_amd import synth
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    nq = 2_500_001 if max_tiles > 1000 else 20_001
    data = synth.gen_genome(300_000, nq, 1000)
    with IntervalIndex(0) as idx:
        idx.insert_node(data["low"], data["high"], data["chrom"])
        idx.build()
        ql, qh, qc = to(data["qlow"]), to(data["qhigh"]), to(data["qchrom"])
        ref_off = torch.empty(nq + 1, dtype=torch.int64, device=dev)
        idx.count_overlaps_device(ql, qh, qc, offsets=ref_off)                     # one launch
        H = int(ref_off[-1].item())
        ref_hits = torch.empty(H, dtype=torch.int32, device=dev)
        idx.query_device(ql, qh, ref_off, ref_hits, qchrom=qc, sort_by_id=True)
        torch.cuda.synchronize()
        os.environ["BIVX_MAX_TILES_PER_LAUNCH"] = str(max_tiles)
        try:
            for own_ws in (False, True):
                ws = torch.empty(idx.query_workspace_bytes(nq), dtype=torch.uint8, device=dev) if own_ws else None
                off = torch.empty(nq + 1, dtype=torch.int64, device=dev)
                hits = torch.empty(H, dtype=torch.int32, device=dev)
                idx.count_overlaps_device(ql, qh, qc, offsets=off)
                assert torch.equal(off, ref_off)
                off.zero_()
                idx.query_device(ql, qh, off, hits, workspace=ws, qchrom=qc, sort_by_id=True)
                assert torch.equal(off, ref_off) and torch.equal(hits, ref_hits)
                beg = torch.empty(nq, dtype=torch.int64, device=dev)
                cnt = torch.empty(nq, dtype=torch.int32, device=dev)
                tot = torch.full((1,), -1, dtype=torch.int64, device=dev)
                for _ in range(2):                                                  # the running total restarts per call
                    idx.query_device_unordered(ql, qh, beg, cnt, hits, tot, workspace=ws, qchrom=qc)
                torch.cuda.synchronize()
                _check_unordered(beg.cpu().numpy(), cnt.cpu().numpy(), hits.cpu().numpy().view(np.uint32),
                                 int(tot.item()), ref_off.cpu().numpy().astype(np.uint64),
                                 ref_hits.cpu().numpy().view(np.uint32), H)
        finally:
            del os.environ["BIVX_MAX_TILES_PER_LAUNCH"]
        assert idx.stats()["prefix_timeouts"] == 0


def test_tile_totals_beyond_32_bits(IntervalIndex):
    """1 100 chromosome-wide queries over 4.2 M nested intervals: every query has 4.2 M hits (> 2^22, the bound
    up to which a tile's 32-bit sums are exact) and one tile of 1 024 queries 4.3 G (> 2^32). Offsets, counts and the
    total must be exact; the hit ids (18 GB) are not materialised: zero-capacity calls."""
    import torch
    dev = torch.device("cuda:0")
    n, q = 4_200_000, 1_100
    low = np.zeros(n, np.uint32)
    high = np.arange(1000, 1000 + n, dtype=np.uint32)
    qlo = np.zeros(q, np.uint32)
    qhi = np.full(q, 1000, np.uint32)
    qhi[7] = 0          # still everything (every interval starts at 0)
    qlo[9] = qhi[9] = 1000 + n   # beyond every interval: nothing
    to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
    exp = np.full(q, n, np.int64)
    exp[9] = 0
    exp_off = np.concatenate([[0], np.cumsum(exp)])
    assert exp_off[1024] > 2 ** 32
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        off = idx.count_overlaps_device(to(qlo), to(qhi))
        assert np.array_equal(off.cpu().numpy(), exp_off)
        beg = torch.empty(q, dtype=torch.int64, device=dev)
        cnt = torch.empty(q, dtype=torch.int32, device=dev)
        tot = torch.zeros(1, dtype=torch.int64, device=dev)
        none = torch.empty(0, dtype=torch.int32, device=dev)
        idx.query_device_unordered(to(qlo), to(qhi), beg, cnt, none, tot)
        torch.cuda.synchronize()
        assert int(tot.item()) == int(exp_off[-1])
        assert np.array_equal(cnt.cpu().numpy().view(np.uint32).astype(np.int64), exp)
        b = beg.cpu().numpy()
        o = np.argsort(b[exp > 0], kind="stable")
        bs, cs = b[exp > 0][o], exp[exp > 0][o]
        assert bs[0] == 0 and np.array_equal(bs[1:], (bs + cs)[:-1])
