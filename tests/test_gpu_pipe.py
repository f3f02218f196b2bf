"""The pipelined single-pass kernel (binary_amd/csrc/query_pipe.hip: persistent workgroups, worker and service
wavefronts, output deferred by two iterations) against k_query_fused and the oracle. By default it only takes batches of
0.8 M queries and more; BIVX_PIPE=2 sends every eligible batch through it, BIVX_PIPE=0 none, so the two kernels can be
compared bit for bit on the same inputs: offsets AND ids in index order must be identical."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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


def _both(idx, qlo, qhi, qc=None, cap=None, workspace=False, sort_by_id=False):
    """(offsets, hits) from the pipelined kernel and from k_query_fused, same buffers sizes"""
    import torch
    dev = qlo.device
    q = qlo.numel()
    off0 = idx.count_overlaps_device(qlo, qhi, qc)
    H = int(off0[-1].item())
    out = []
    for mode in (2, 0):
        with _env(BIVX_PIPE=mode):
            off = torch.full((q + 1,), -1, dtype=torch.int64, device=dev)
            hits = torch.full((max(H if cap is None else cap, 1),), -1, dtype=torch.int32, device=dev)
            ws = torch.empty(idx.query_workspace_bytes(q), dtype=torch.uint8, device=dev) if workspace else None
            idx.query_device(qlo, qhi, off, hits, workspace=ws, qchrom=qc, sort_by_id=sort_by_id)
            idx.stream_status()
            out.append((off.cpu().numpy(), hits.cpu().numpy()))
    return H, out


@pytest.mark.parametrize("order", ["generated", "sorted", "nearly"])
@pytest.mark.parametrize("workspace,by_id", [(False, False), (True, False), (False, True)])
def test_same_csr_as_the_fused_kernel_and_the_oracle(oracle, order, workspace, by_id):
    import torch
    from binary_amd import IntervalIndex, synth
    d = synth.gen_genome(400_000, 300_007, 1000)       # 24 chromosomes; 313 tiles of 960 queries, a ragged last one
    qc, qlo, qhi = d["qchrom"], d["qlow"], d["qhigh"]
    if order != "generated":
        p = np.lexsort((qlo, qc))
        if order == "nearly":                            # sorted with every 50th query displaced
            rng = np.random.default_rng(0)
            sw = rng.permutation(p.size)[: p.size // 50]
            p[sw] = p[np.roll(sw, 1)]
        qc, qlo, qhi = qc[p], qlo[p], qhi[p]
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    with IntervalIndex(0) as idx:
        idx.insert_node(d["low"], d["high"], d["chrom"])
        idx.build()
        H, ((off_p, hits_p), (off_f, hits_f)) = _both(idx, to(qlo), to(qhi), to(qc), workspace=workspace,
                                                       sort_by_id=by_id)
        assert np.array_equal(off_p, off_f) and np.array_equal(hits_p[:H], hits_f[:H])
        if by_id:   # ascending inside every query: no descent anywhere except at a query's first id
            desc = np.nonzero(np.diff(hits_p[:H].astype(np.int64)) < 0)[0] + 1
            assert np.isin(desc, off_p).all()
    # the oracle on a few chromosomes
    for c in (0, 7, 23):
        m = d["chrom"] == c
        base = int(np.nonzero(m)[0][0])
        t = oracle.OracleTree(d["low"][m], d["high"][m])
        qm = np.nonzero(qc == c)[0][:4000]
        off_o, hits_o = t.find_overlaps_batch(qlo[qm], qhi[qm])
        got = np.concatenate([np.sort(hits_p[off_p[i]:off_p[i + 1]]) for i in qm]).astype(np.int64)
        assert np.array_equal(got, oracle.sorted_csr(off_o, hits_o) + base)


def test_slices_that_cannot_be_staged_are_filled_behind_the_kernel(oracle):
    """Wavefront-cooperative windows (a few chromosome-wide queries among point queries) and lists beyond a stage: the
    slice is listed, k_fill_slices writes its ids; everything else goes through the stages."""
    import torch
    from binary_amd import IntervalIndex
    rng = np.random.default_rng(3)
    n = 150_000
    low = rng.integers(0, 3_000_000, n).astype(np.uint32)
    high = (low + rng.integers(1, 400, n)).astype(np.uint32)
    q = 60_000
    qlo = rng.integers(0, 3_000_000, q).astype(np.uint32)
    qhi = qlo.copy()
    wide = rng.permutation(q)[:40]
    qlo[wide] = rng.integers(0, 1_000_000, 40)
    qhi[wide] = qlo[wide] + rng.integers(20_000, 200_000, 40)   # hundreds to thousands of hits each
    dense = np.arange(5000, 5400)                                # neighbouring queries with ~40 hits each: lists overflow a stage
    qlo[dense] = 1_500_000 + np.arange(400)
    qhi[dense] = qlo[dense] + 800
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        H, ((off_p, hits_p), (off_f, hits_f)) = _both(idx, to(qlo), to(qhi))
        assert np.array_equal(off_p, off_f) and np.array_equal(hits_p[:H], hits_f[:H])
        # ascending ids: the staged lists are ordered in the stage (lists of 40 ids go through the keep slots), the
        # listed slices by the conditional pass behind k_fill_slices
        H1, ((off_s, hits_s), (off_fs, hits_fs)) = _both(idx, to(qlo), to(qhi), sort_by_id=True)
        assert np.array_equal(off_s, off_f) and np.array_equal(hits_s[:H], hits_fs[:H])
        want = np.concatenate([np.sort(hits_f[off_f[i]:off_f[i + 1]]) for i in range(q)])
        assert np.array_equal(hits_s[:H], want)
        # a buffer that is too small: offsets stay exact, nothing is written beyond the capacity
        cap = H // 3
        H2, ((off_c, hits_c), _) = _both(idx, to(qlo), to(qhi), cap=cap)
        assert np.array_equal(off_c, off_f) and np.array_equal(hits_c[:cap], hits_f[:cap])
    assert np.array_equal(np.diff(off_p), oracle.count_overlaps_numpy(low, high, qlo, qhi))


def test_ascending_ids_for_lists_of_every_staged_length():
    """Lists of 0..60 ids next to each other in a wavefront's stage: up to eight ids are ranked in registers, longer
    ones through the keep slots; the slices whose 64 lists exceed a stage are listed and ordered behind the kernel."""
    import torch
    from binary_amd import IntervalIndex
    rng = np.random.default_rng(11)
    # piles of k intervals over the same point, k = 0..60, 5000 positions apart; ids shuffled so index order != id order
    pos = np.arange(1, 4001, dtype=np.uint32) * 5000
    k = rng.integers(0, 61, pos.size)
    few = rng.random(pos.size) < 0.8
    k[few] = rng.integers(0, 5, int(few.sum()))
    centre = np.repeat(pos, k)
    low = (centre - rng.integers(1, 2000, centre.size)).astype(np.uint32)
    high = (centre + rng.integers(1, 2000, centre.size)).astype(np.uint32)
    p = rng.permutation(low.size)
    low, high = low[p], high[p]
    q = 120_000
    qlo = pos[rng.integers(0, pos.size, q)].astype(np.uint32)
    qhi = qlo.copy()
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        H, ((off_s, hits_s), (off_f, hits_f)) = _both(idx, to(qlo), to(qhi), sort_by_id=True)
        H0, ((off_u, hits_u), _) = _both(idx, to(qlo), to(qhi))
    assert np.array_equal(off_s, off_f) and np.array_equal(hits_s[:H], hits_f[:H]) and np.array_equal(off_u, off_s)
    cnt = np.diff(off_s)
    assert cnt.max() >= 40 and (cnt == 0).any() and ((cnt > 8) & (cnt < 30)).any()
    seg = np.repeat(np.arange(q), cnt)
    order = np.lexsort((hits_u[:H], seg))
    assert np.array_equal(hits_s[:H], hits_u[:H][order])


@pytest.mark.parametrize("order", ["sorted", "nearly", "generated"])
def test_many_ids_per_query_position_sorted_batches(oracle, order):
    """Config-5 density (an index overlapped with itself, ~17 ids per query): the regenerating form of the pipeline
    (k_query_pipe_dense) does position-sorted batches, k_query_fused everything else — a device-side probe of the
    query order decides, so the same call must give the same CSR whatever the order and whichever kernel ran."""
    import torch
    from binary_amd import IntervalIndex, synth
    n = 600_000
    low, high = synth.gen_intervals(n, 37_000_000, 1000, 5)
    # a few chromosome-wide intervals' worth of long windows and a pile of 200 intervals over one point: slices that
    # cannot come out of a slab are listed and filled behind the kernel
    rng = np.random.default_rng(2)
    pile = np.full(200, 20_000_000, dtype=np.uint32)
    low = np.concatenate([low, pile - rng.integers(1, 500, 200).astype(np.uint32)])
    high = np.concatenate([high, pile + rng.integers(1, 500, 200).astype(np.uint32)])
    qlo, qhi = low.copy(), high.copy()
    if order != "generated":
        p = np.argsort(qlo, kind="stable")
        if order == "nearly":
            sw = rng.permutation(p.size)[: p.size // 100]
            p[sw] = p[np.roll(sw, 1)]
        qlo, qhi = qlo[p], qhi[p]
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        H, ((off_p, hits_p), (off_f, hits_f)) = _both(idx, to(qlo), to(qhi))
        assert H > 10 * qlo.size
        assert np.array_equal(off_p, off_f) and np.array_equal(hits_p[:H], hits_f[:H])
        # ascending ids: k_sort_hits orders the CSR behind whichever kernel wrote it
        H1, ((off_s, hits_s), (off_fs, hits_fs)) = _both(idx, to(qlo), to(qhi), sort_by_id=True)
        assert np.array_equal(off_s, off_f) and np.array_equal(hits_s[:H], hits_fs[:H])
        seg = np.repeat(np.arange(qlo.size), np.diff(off_f))
        assert np.array_equal(hits_s[:H], hits_f[:H][np.lexsort((hits_f[:H], seg))])
        cap = H // 2
        H2, ((off_c, hits_c), _) = _both(idx, to(qlo), to(qhi), cap=cap, workspace=True)
        assert np.array_equal(off_c, off_f) and np.array_equal(hits_c[:cap], hits_f[:cap])
        assert idx.stats()["prefix_timeouts"] == 0
    assert np.array_equal(np.diff(off_p), oracle.count_overlaps_numpy(low, high, qlo, qhi))
    t = oracle.OracleTree(low, high)
    sel = np.arange(0, qlo.size, 997)
    off_o, hits_o = t.find_overlaps_batch(qlo[sel], qhi[sel])
    got = np.concatenate([np.sort(hits_p[off_p[i]:off_p[i + 1]]) for i in sel]).astype(np.int64)
    assert np.array_equal(got, oracle.sorted_csr(off_o, hits_o))


@pytest.mark.parametrize("by_id", [False, True])
def test_edge_coordinates_through_the_pipelined_kernel(oracle, by_id):
    """The corners test_gpu_parity.py walks through the host entry points, sent through k_query_pipe: coordinates at both
    ends of uint32, low > high intervals and queries (TraMapper's BND records), touching ends, queries on a chromosome
    the index does not have, an empty chromosome in the middle."""
    import torch
    from binary_amd import IntervalIndex
    rng = np.random.default_rng(9)
    M = 0xFFFFFFFF
    n = 40_000
    chrom = rng.integers(0, 5, n).astype(np.uint32)
    chrom[chrom == 2] = 3                                           # chromosome 2 stays empty
    low = rng.integers(0, 3_000_000, n).astype(np.uint32)
    high = (low + rng.integers(0, 900, n)).astype(np.uint32)
    top = rng.permutation(n)[:4000]                                 # a cluster at the very top of the range
    low[top] = (M - rng.integers(0, 500_000, 4000)).astype(np.uint32)
    high[top] = np.minimum(low[top].astype(np.int64) + rng.integers(0, 900, 4000), M).astype(np.uint32)
    inv = rng.permutation(n)[:3000]                                 # inverted records
    low[inv], high[inv] = high[inv].copy(), low[inv].copy()
    low[:4] = [0, 0, M, M - 1]
    high[:4] = [0, M, M, M]
    chrom[:4] = 0
    q = 30_000
    qc = rng.integers(0, 7, q).astype(np.uint32)                    # 5 and 6 do not exist in the index
    qlo = rng.integers(0, 3_000_000, q).astype(np.uint32)
    qhi = (qlo + rng.integers(0, 900, q)).astype(np.uint32)
    t = rng.permutation(q)[:5000]
    qlo[t] = (M - rng.integers(0, 500_000, 5000)).astype(np.uint32)
    qhi[t] = np.minimum(qlo[t].astype(np.int64) + rng.integers(0, 900, 5000), M).astype(np.uint32)
    qi = rng.permutation(q)[:1500]
    qlo[qi], qhi[qi] = qhi[qi].copy(), qlo[qi].copy()
    qlo[:6] = [0, M, 0, 6, 1, M - 500]
    qhi[:6] = [0, M, M, M - 2, 4, M - 400]
    qc[:6] = 0
    touch = rng.permutation(n)[:2000]                               # queries that end exactly where an interval begins
    qhi[6:2006] = low[touch]
    qlo[6:2006] = np.minimum(qhi[6:2006], qlo[6:2006])
    qc[6:2006] = chrom[touch]
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high, chrom)
        idx.build()
        H, ((off_p, hits_p), (off_f, hits_f)) = _both(idx, to(qlo), to(qhi), to(qc), sort_by_id=by_id)
    assert np.array_equal(off_p, off_f) and np.array_equal(hits_p[:H], hits_f[:H])
    # the predicate itself (interval_tree.hpp:119-121), per chromosome, inverted records and queries included
    want = np.zeros(q, dtype=np.int64)
    for c in range(5):
        mi, mq = np.nonzero(chrom == c)[0], np.nonzero(qc == c)[0]
        for k in range(0, mq.size, 500):
            sel = mq[k:k + 500]
            want[sel] = ((low[mi][None, :] <= qhi[sel][:, None]) & (high[mi][None, :] >= qlo[sel][:, None])).sum(axis=1)
    assert np.array_equal(np.diff(off_p), want)
    for i in list(range(8)) + list(rng.integers(0, q, 300)):
        m = (chrom == qc[i]) & (low <= qhi[i]) & (high >= qlo[i])
        assert np.array_equal(np.sort(hits_p[off_p[i]:off_p[i + 1]]), np.nonzero(m)[0])


def test_slice_totals_beyond_32_bits_in_the_pipelined_kernels():
    """Chromosome-wide queries over 4.2 M nested intervals among point queries: a lane with more than 2^22 hits sends its
    slice through the 64-bit scan, and a tile's total passes 2^32. Counts and offsets must be exact through
    k_query_pipe (zero capacity) and, for the many-ids form, through k_query_pipe_dense on the same batch sorted."""
    import torch
    from binary_amd import IntervalIndex
    dev = torch.device("cuda:0")
    n, q = 4_200_000, 20_000
    low = np.zeros(n, np.uint32)
    high = np.arange(1000, 1000 + n, dtype=np.uint32)
    rng = np.random.default_rng(1)
    qlo = rng.integers(2000, 1000 + n, q).astype(np.uint32)       # a point query at x hits the intervals with high >= x
    qhi = qlo.copy()
    wide = np.sort(rng.permutation(q)[:1100])
    qlo[wide] = 0
    qhi[wide] = 1000                                               # everything: 4.2 M hits each, 4.6 G in all
    exp = np.where(qlo == 0, n, 1000 + n - qlo.astype(np.int64))
    exp_off = np.concatenate([[0], np.cumsum(exp)])
    assert exp_off[-1] > 2 ** 32
    to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        for mode in (2, 0):
            with _env(BIVX_PIPE=mode):
                off = idx.count_overlaps_device(to(qlo), to(qhi))
                idx.stream_status()
            assert np.array_equal(off.cpu().numpy(), exp_off), mode
        # many ids per query, position-sorted, a hit buffer far too small: offsets stay exact, nothing beyond it is written
        p = np.argsort(qlo, kind="stable")
        with _env(BIVX_PIPE=2):
            assert idx.query_kernel_name(q, 10 * q).startswith("k_query_pipe_dense")
            o = torch.empty(q + 1, dtype=torch.int64, device=dev)
            h = torch.full((10 * q + 64,), -1, dtype=torch.int32, device=dev)
            idx.query_device(to(qlo[p]), to(qhi[p]), o, h[:10 * q])
            idx.stream_status()
        assert np.array_equal(o.cpu().numpy(), np.concatenate([[0], np.cumsum(exp[p])]))
        assert (h[10 * q:] == -1).all() and (h[:10 * q] >= 0).all()
        assert idx.stats()["prefix_timeouts"] == 0


def test_begin_count_output_through_the_pipelined_kernel():
    """bivx_query_dev_u asks for per-query (begin, count) in any layout; the ordered CSR is one, and on batches the
    pipelined kernel takes it is the faster answer: begin[q] = offsets[q], count[q] = the list's length, total = H —
    staged slices, listed slices (a stage overflow) and a caller workspace included."""
    import torch
    from binary_amd import IntervalIndex, synth
    d = synth.gen_genome(300_000, 200_003, 1000)
    qc, qlo, qhi = d["qchrom"], d["qlow"], d["qhigh"]
    qlo[5000:5400] = 1_500_000 + np.arange(400)      # neighbouring long queries: their wavefronts' lists overflow a stage
    qhi[5000:5400] = qlo[5000:5400] + 40_000
    qc[5000:5400] = 0
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    with IntervalIndex(0) as idx:
        idx.insert_node(d["low"], d["high"], d["chrom"])
        idx.build()
        d_qlo, d_qhi, d_qc = to(qlo), to(qhi), to(qc)
        with _env(BIVX_PIPE=0):
            ref_off, ref_hits = idx.find_overlaps_device(d_qlo, d_qhi, d_qc)
        H, q = int(ref_off[-1].item()), qlo.size
        for workspace in (False, True):
            with _env(BIVX_PIPE=2):
                beg = torch.full((q,), -1, dtype=torch.int64, device=dev)
                cnt = torch.full((q,), -1, dtype=torch.int32, device=dev)
                tot = torch.full((1,), -1, dtype=torch.int64, device=dev)
                hits = torch.full((H,), -1, dtype=torch.int32, device=dev)
                ws = torch.empty(idx.query_workspace_bytes(q), dtype=torch.uint8, device=dev) if workspace else None
                for _ in range(2):
                    idx.query_device_unordered(d_qlo, d_qhi, beg, cnt, hits, tot, workspace=ws, qchrom=d_qc)
                idx.stream_status()
            assert int(tot.item()) == H
            assert torch.equal(beg, ref_off[:-1]) and torch.equal(cnt.to(torch.int64), ref_off[1:] - ref_off[:-1])
            assert torch.equal(hits, ref_hits)


def test_two_streams_query_the_same_index_at_once():
    """const calls are mutually thread-safe (the reference's mappers query their trees concurrently,
    mapper.cpp:130-141): two host threads, a stream each, pipelined launches that compete for the CUs — a workgroup of
    one launch may have to wait for the other launch's workgroups to leave before it can start. Every result must be
    the single-stream one."""
    import threading
    import torch
    from binary_amd import IntervalIndex, synth
    d = synth.gen_genome(500_000, 400_000, 1000)
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    with IntervalIndex(0) as idx, _env(BIVX_PIPE=2):
        idx.insert_node(d["low"], d["high"], d["chrom"])
        idx.build()
        halves = []
        for sl in (slice(0, 200_000), slice(200_000, 400_000)):
            qc, ql, qh = to(d["qchrom"][sl]), to(d["qlow"][sl]), to(d["qhigh"][sl])
            off, hits = idx.find_overlaps_device(ql, qh, qc)
            halves.append((qc, ql, qh, off, hits))
        torch.cuda.synchronize()
        errors = []

        def work(k):
            try:
                qc, ql, qh, ref_off, ref_hits = halves[k]
                st = torch.cuda.Stream(device=dev)
                with torch.cuda.stream(st):
                    off = torch.empty_like(ref_off)
                    hits = torch.empty_like(ref_hits)
                    for _ in range(25):
                        off.fill_(-1)
                        hits.fill_(-1)
                        idx.query_device(ql, qh, off, hits, qchrom=qc)
                        st.synchronize()
                        idx.stream_status()
                        if not (torch.equal(off, ref_off) and torch.equal(hits, ref_hits)):
                            errors.append(f"thread {k}: result differs")
                            return
            except Exception as e:  # noqa: BLE001
                errors.append(f"thread {k}: {type(e).__name__}: {e}")

        ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors
        assert idx.stats()["prefix_timeouts"] == 0


def test_every_slice_of_a_large_batch_listed():
    """4 584 tiles whose slices all overflow their stage (5.6 ids per query: ~360 per wavefront against 320) while the
    capacity still says "at most 6 per query": 68 760 listed slices — the list must hold 15 entries per tile, not one
    (it was sized per tile at first)."""
    import torch
    from binary_amd import IntervalIndex, synth
    low, high = synth.gen_intervals(1_000_000, 89_000_000, 1000, 3)
    qlo, qhi = synth.gen_point_queries(4_400_000, 89_000_000, 4)
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        d_qlo, d_qhi = to(qlo), to(qhi)
        off = idx.count_overlaps_device(d_qlo, d_qhi)
        H = int(off[-1].item())
        assert 5.2 * qlo.size < H <= 6 * qlo.size
        if os.environ.get("BIVX_PIPE", "1") != "0":   # (a suite run with the pipelined kernels switched off)
            assert idx.query_kernel_name(qlo.size, H) == "k_query_pipe"
        res = []
        for mode in (1, 0):
            with _env(BIVX_PIPE=mode):
                o = torch.empty(qlo.size + 1, dtype=torch.int64, device=dev)
                h = torch.full((H,), -1, dtype=torch.int32, device=dev)
                idx.query_device(d_qlo, d_qhi, o, h)
                idx.stream_status()
                res.append((o, h))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        assert torch.equal(res[0][0], off)
        # and the next call on the same workspace is clean
        o = torch.empty(qlo.size + 1, dtype=torch.int64, device=dev)
        h = torch.full((H,), -1, dtype=torch.int32, device=dev)
        idx.query_device(d_qlo, d_qhi, o, h)
        idx.stream_status()
        assert torch.equal(o, off) and torch.equal(h, res[1][1])


def test_chained_launches_and_few_workgroups():
    import torch
    from binary_amd import IntervalIndex, synth
    low, high = synth.gen_intervals(200_000, 50_000_000, 1000)
    qlo, qhi = synth.gen_range_queries(100_001, 50_000_000, 1000)
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high)
        idx.build()
        H, ((off_ref, hits_ref), _) = _both(idx, to(qlo), to(qhi))
        for knobs in (dict(BIVX_MAX_TILES_PER_LAUNCH=7), dict(BIVX_PIPE_WGS=1), dict(BIVX_PIPE_WGS=3)):
            with _env(**knobs):
                H2, ((off_p, hits_p), _) = _both(idx, to(qlo), to(qhi))
            assert np.array_equal(off_p, off_ref) and np.array_equal(hits_p[:H], hits_ref[:H]), knobs
        # a pure count through the pipelined kernel
        with _env(BIVX_PIPE=2):
            off = idx.count_overlaps_device(to(qlo), to(qhi))
            idx.stream_status()
        assert np.array_equal(off.cpu().numpy(), off_ref)
        assert idx.stats()["prefix_timeouts"] == 0


def test_errors_are_surfaced_by_the_pipelined_kernel_too():
    import ctypes as C
    import torch
    from binary_amd import IntervalIndex, capi, synth
    low, high = synth.gen_intervals(100_000, 20_000_000, 1000)
    qlo, qhi = synth.gen_range_queries(200_000, 20_000_000, 1000)
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
    d_qlo, d_qhi = to(qlo), to(qhi)
    with IntervalIndex(0) as idx, _env(BIVX_PIPE=2):
        idx.insert_node(low, high)
        idx.build()
        off, hits = idx.find_overlaps_device(d_qlo, d_qhi)
        idx.stream_status()
        ref_off, ref_hits = off.clone(), hits.clone()
        idx.query_device(d_qlo, d_qhi, off, hits)
        idx.stream_status()
        assert torch.equal(off, ref_off) and torch.equal(hits, ref_hits)
        # an inconsistent workspace (the ticket word of a launch that died half-way)
        s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        capi.check(idx._L.bivx_debug_corrupt_workspace(idx._h, s))
        idx.query_device(d_qlo, d_qhi, off, hits)
        with pytest.raises(capi.BivxError) as e:
            idx.stream_status()
        assert e.value.code == capi.E_TIMEOUT
        idx.query_device(d_qlo, d_qhi, off, hits)       # cleared before this launch
        idx.stream_status()
        assert torch.equal(off, ref_off) and torch.equal(hits, ref_hits)


def test_two_processes_share_the_gpu():
    """Two PROCESSES launch the pipelined kernels on one card at the same time (what `bench.py --gpus 2` rehearsed on a
    one-GPU box does): persistent workgroups of two launches compete for the CUs and a launch may run for a long time with
    only part of its grid resident. Round 4's sharded first tickets deadlocked exactly here (a workgroup was handed a
    smaller tile after a larger one: the flush of the larger waits for a sweep over the smaller, which the same workers
    would count later) — every call must finish promptly, with the same totals in both processes."""
    import ast
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "two_process_stress.py"), "1e7", "80"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = ast.literal_eval([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    res = out["results (rank, ids, slowest round s)"]
    assert out["exit codes"] == [0, 0] and len(res) == 2 and res[0][1] == res[1][1] and max(x[2] for x in res) < 5.0, out
