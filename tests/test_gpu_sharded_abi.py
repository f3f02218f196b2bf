"""SURVEY.md §8 rows b / e: several GPUs behind ONE C-ABI handle (bivx_create_sharded). The test box has one GPU, so
the devices are {0, 0}: two shards, two host threads, the same routing and id mapping as on a node with several cards
(only the peer placement differs). The sharded handle must give the single-index CSR bit for bit, and the oracle's."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_by_chromosome_equals_single_index_and_oracle(oracle):
    from binary_amd import IntervalIndex, synth, capi
    d = synth.gen_genome(300_000, 200_000, 1000)                      # 24 chromosomes, grouped by chromosome
    rng = np.random.default_rng(1)
    perm = rng.permutation(d["qlow"].size)                            # queries in arbitrary order across chromosomes
    qc, qlo, qhi = d["qchrom"][perm], d["qlow"][perm], d["qhigh"][perm]
    with IntervalIndex(0) as one, IntervalIndex([0, 0, 0]) as sh:
        for idx in (one, sh):
            half = d["low"].size // 2
            idx.insert_node(d["low"][:half], d["high"][:half], d["chrom"][:half])
            idx.insert_node(d["low"][half:], d["high"][half:], d["chrom"][half:])
            idx.build()
        assert sh.num_devices() == 3 and one.num_devices() == 1 and sh.size() == one.size()
        assert sh.num_chroms() == 24 and all(sh.device_of_chrom(c) == 0 for c in range(24))
        assert sh.device_of_chrom(99) == -1
        for sort_by_id in (True, False):
            off1, hits1 = one.find_overlaps(qlo, qhi, qc, sort_by_id=sort_by_id)
            off2, hits2 = sh.find_overlaps(qlo, qhi, qc, sort_by_id=sort_by_id)
            assert np.array_equal(off1, off2) and np.array_equal(hits1, hits2)
        assert np.array_equal(one.find_overlap(qlo, qhi, qc), sh.find_overlap(qlo, qhi, qc))
        ids = np.arange(0, d["low"].size, 101, dtype=np.uint32)
        for a, b in zip(one.get_intervals(ids), sh.get_intervals(ids)):
            assert np.array_equal(a, b)
        # the two-call form through the raw ABI
        q = qlo.size
        off3 = np.zeros(q + 1, np.uint64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        capi.check(sh._L.bivx_count(sh._h, p(qc), p(qlo), p(qhi), q, p(off3)))
        assert np.array_equal(off3, off1)
        hits3 = np.zeros(int(off3[-1]), np.uint32)
        capi.check(sh._L.bivx_fill(sh._h, p(qc), p(qlo), p(qhi), q, p(off3), p(hits3), 1))
        off1s, hits1s = one.find_overlaps(qlo, qhi, qc, sort_by_id=True)
        assert np.array_equal(hits3, hits1s)
        # device-pointer entry points belong to one device
        rc = sh._L.bivx_count_dev(sh._h, None, None, None, 0, None, None)
        assert rc == capi.E_STATE
    # the oracle, chromosome by chromosome
    off_h = off1s.astype(np.int64)
    for c in range(24):
        m = d["chrom"] == c
        base = int(np.nonzero(m)[0][0])
        t = oracle.OracleTree(d["low"][m], d["high"][m])
        qm = np.nonzero(qc == c)[0][:3000]
        off_o, hits_o = t.find_overlaps_batch(qlo[qm], qhi[qm])
        exp = oracle.sorted_csr(off_o, hits_o) + base
        got = np.concatenate([hits1s[off_h[i]:off_h[i + 1]] for i in qm]) if qm.size else np.zeros(0)
        assert np.array_equal(got.astype(np.int64), exp)


def test_single_chromosome_is_replicated_and_queries_are_split(oracle):
    from binary_amd import IntervalIndex, synth
    low, high = synth.gen_intervals(100_000, 30_000_000, 2000)
    qlo, qhi = synth.gen_range_queries(50_001, 30_000_000, 2000)
    with IntervalIndex([0, 0]) as sh:
        sh.insert_node(low, high, svtype=np.where(np.arange(low.size) % 3 == 0, 2, 1).astype(np.uint8))
        off, hits = sh.find_overlaps(qlo, qhi)
        off2, hits2 = sh.find_overlaps(qlo, qhi, svtype=2)
    t = oracle.OracleTree(low, high)
    off_o, hits_o = t.find_overlaps_batch(qlo, qhi)
    assert np.array_equal(off, off_o)
    assert np.array_equal(hits.astype(np.int64), oracle.sorted_csr(off_o, hits_o))
    assert np.all(hits2 % 3 == 0)
    cnt2 = np.diff(off2.astype(np.int64))
    h = hits.astype(np.int64)
    qid = np.repeat(np.arange(qlo.size), np.diff(off.astype(np.int64)))
    assert np.array_equal(cnt2, np.bincount(qid[h % 3 == 0], minlength=qlo.size))


def test_sv2nl_tool_on_a_sharded_index(tmp_path):
    """sv2nl --devices 0,0: the typed index of the three mappers sharded by chromosome, TRA's per-interval filter words
    remapped to shard-local ids; same lines as the restatement."""
    sv2nl = os.path.join(ROOT, "binary_amd", "sv2nl", "sv2nl")
    subprocess.check_call(["make", "-C", os.path.dirname(sv2nl), "-s", "all"])
    vcf = os.path.join(ROOT, "tests", "golden", "vcf")
    out = str(tmp_path / "o.tsv")
    r = subprocess.run([sv2nl, os.path.join(vcf, "pair_sv.vcf"), os.path.join(vcf, "pair_nl.vcf"), "-o", out,
                        "--devices", "0,0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for k in ("dup", "inv", "tra"):
        got = sorted(open(out + "." + k).read().splitlines()[1:])
        exp = sorted(open(os.path.join(vcf, f"pair_expected.{k}.tsv")).read().splitlines()[1:])
        assert got == exp, k


# ---- bivx_query_sharded_dev: the gathered device-resident CSR (RCCL gatherv below the C ABI) ---------------------------

def _rows_as_lists(off, hits, rows, q):
    """per batch query: its list of the gathered CSR (rows are grouped by device: query_of_row says whose row it is)"""
    off, hits, rows = off.cpu().numpy(), hits.cpu().numpy().view(np.uint32), rows.cpu().numpy()
    assert rows.size == q and np.array_equal(np.sort(rows), np.arange(q)), "every query has exactly one row"
    out = [None] * q
    for r, qi in enumerate(rows):
        out[qi] = hits[off[r]:off[r + 1]]
    return out


@pytest.mark.parametrize("sort_by_id", [False, True])
def test_sharded_dev_one_device_communicator_equals_the_single_index_bit_for_bit(sort_by_id):
    """devices = [0]: ncclCommInitAll over one device, ncclAllGather of the sizes on a one-rank communicator; the root's
    block is a local copy. The gathered CSR must be bivx_query_dev_s's, bit for bit."""
    import torch
    from binary_amd import IntervalIndex, synth
    d = synth.gen_genome(400_000, 250_000, 1000)
    rng = np.random.default_rng(3)
    perm = rng.permutation(d["qlow"].size)
    qc, qlo, qhi = d["qchrom"][perm], d["qlow"][perm], d["qhigh"][perm]
    qc[:50] = 77                                   # queries on a chromosome nobody holds: empty rows
    dev = torch.device("cuda:0")
    to = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int32)).to(dev)
    with IntervalIndex(0) as one, IntervalIndex([0]) as sh:
        for idx in (one, sh):
            idx.insert_node(d["low"], d["high"], d["chrom"])
            idx.build()
        off1 = one.count_overlaps_device(to(qlo), to(qhi), to(qc))
        hits1 = torch.empty(int(off1[-1].item()), dtype=torch.int32, device=dev)
        one.query_device(to(qlo), to(qhi), off1, hits1, qchrom=to(qc), sort_by_id=sort_by_id)
        off2, hits2, rows, used_rccl = sh.query_sharded_device(qlo, qhi, qc, sort_by_id=sort_by_id)
        assert used_rccl, "one distinct device: the RCCL path (communicator of one rank)"
        assert torch.equal(rows, torch.arange(qlo.size, dtype=torch.int32, device=dev))
        assert torch.equal(off1, off2) and torch.equal(hits1, hits2)
        # a second call reuses the handle's buffers and the communicator; a smaller batch, after a rebuild
        sh.insert_node(d["low"][:1000], d["high"][:1000], d["chrom"][:1000])
        sh.build()
        one.insert_node(d["low"][:1000], d["high"][:1000], d["chrom"][:1000])
        one.build()
        off1 = one.count_overlaps_device(to(qlo[:9999]), to(qhi[:9999]), to(qc[:9999]))
        hits1 = torch.empty(int(off1[-1].item()), dtype=torch.int32, device=dev)
        one.query_device(to(qlo[:9999]), to(qhi[:9999]), off1, hits1, qchrom=to(qc[:9999]), sort_by_id=sort_by_id)
        off2, hits2, rows, _ = sh.query_sharded_device(qlo[:9999], qhi[:9999], qc[:9999], sort_by_id=sort_by_id)
        assert torch.equal(off1, off2) and torch.equal(hits1, hits2)


@pytest.mark.parametrize("sort_by_id", [False, True])
def test_sharded_dev_three_shards_same_lists_as_the_single_index(sort_by_id, oracle):
    """devices = [0, 0, 0]: three shards on the one card — the routing, the per-shard device CSRs, the local -> global id
    mapping on the device, the displacement arithmetic and the rebasing kernel of a three-device gather; the blocks move by
    device-to-device copies because RCCL refuses a device twice in one communicator (used_rccl == False)."""
    from binary_amd import IntervalIndex, synth
    d = synth.gen_genome(300_000, 200_000, 1000)
    rng = np.random.default_rng(4)
    perm = rng.permutation(d["qlow"].size)
    qc, qlo, qhi = d["qchrom"][perm], d["qlow"][perm], d["qhigh"][perm]
    qc[::1000] = 31                                # no device holds chromosome 31
    with IntervalIndex(0) as one, IntervalIndex([0, 0, 0]) as sh:
        for idx in (one, sh):
            idx.insert_node(d["low"], d["high"], d["chrom"])
            idx.build()
        off1, hits1 = one.find_overlaps(qlo, qhi, qc, sort_by_id=sort_by_id)
        off2, hits2, rows, used_rccl = sh.query_sharded_device(qlo, qhi, qc, sort_by_id=sort_by_id)
        assert not used_rccl
        assert int(off2[-1].item()) == int(off1[-1]) == hits2.numel()
        lists = _rows_as_lists(off2, hits2, rows, qlo.size)
        for qi in range(qlo.size):
            a = hits1[int(off1[qi]):int(off1[qi + 1])]
            # (index order of a shard is index order of the single index restricted to its chromosomes, in global ids only
            # after sorting: compare as sets unless ascending ids were asked for)
            b = lists[qi]
            assert np.array_equal(a, b) if sort_by_id else np.array_equal(np.sort(a), np.sort(b)), qi
        # rows are grouped by device, in batch order inside a group
        r = rows.cpu().numpy()
        devs = np.array([0 if c >= 24 else sh.device_of_chrom(int(c)) for c in qc])
        assert (devs == 0).all()                   # (one card)
        # a replicated handle (one populated chromosome, three shards): the queries are split instead
        sel = d["chrom"] == 0
        with IntervalIndex([0, 0, 0]) as rep, IntervalIndex(0) as one0:
            for idx in (rep, one0):
                idx.insert_node(d["low"][sel], d["high"][sel])
                idx.build()
            q0 = d["qchrom"] == 0
            o1, h1 = one0.find_overlaps(d["qlow"][q0], d["qhigh"][q0], sort_by_id=True)
            o2, h2, rows2, _ = rep.query_sharded_device(d["qlow"][q0], d["qhigh"][q0], sort_by_id=True)
            assert np.array_equal(rows2.cpu().numpy(), np.arange(int(q0.sum())))
            assert np.array_equal(o2.cpu().numpy().astype(np.uint64), o1) and np.array_equal(h2.cpu().numpy().view(np.uint32), h1)


def test_sharded_dev_rejects_a_plain_handle_and_an_unbuilt_one():
    from binary_amd import IntervalIndex, capi
    q = np.array([5], np.uint32)
    with IntervalIndex(0) as one:
        one.insert_node(q, q)
        one.build()
        with pytest.raises(capi.BivxError) as e:
            one.query_sharded_device(q, q)
        assert e.value.code == capi.E_STATE
    with IntervalIndex([0]) as sh:
        sh.insert_node(q, q)
        with pytest.raises(capi.BivxError) as e:
            sh.query_sharded_device(q, q)
        assert e.value.code == capi.E_STATE
        sh.build()
        off, hits, rows, _ = sh.query_sharded_device(q, q)
        assert off.tolist() == [0, 1] and hits.tolist() == [0] and rows.tolist() == [0]
        off, hits, rows, _ = sh.query_sharded_device(q[:0], q[:0])
        assert off.tolist() == [0] and hits.numel() == 0 and rows.numel() == 0
