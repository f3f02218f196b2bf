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
