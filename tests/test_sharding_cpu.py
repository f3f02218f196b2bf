"""Multi-rank path on CPU (gloo, world_size 2 and 3): per-chromosome LPT sharding and the gatherv of
per-rank CSR hit lists to rank 0. No GPU: the per-rank CSR comes from the oracle here, which is exactly
what each rank's HIP index would hand to gatherv_csr on the GPU box."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from binary_amd import sharding, synth


def test_lpt_assignment_hg38_matches_survey():
    w = synth.HG38_LENGTHS.astype(float)  # work ~ chromosome length for the synthetic sets
    for n, expect in ((2, 1.001), (4, 1.005), (8, 1.036)):
        a = sharding.lpt_assign(w, n)
        assert sorted(sum(a, [])) == list(range(24))
        assert sharding.imbalance(w, a) <= expect + 5e-4
    a8 = sharding.lpt_assign(w, 8)
    assert all(len(x) == 3 for x in a8)
    # deterministic
    assert a8 == sharding.lpt_assign(w, 8)
    # degenerate shapes
    assert sharding.lpt_assign([5.0], 4) == [[0], [], [], []]
    assert sharding.lpt_assign([], 2) == [[], []]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_csr(oracle_mod, data, chroms):
    """CSR of this rank's chromosomes (query order: by chromosome, then generation order)."""
    offs, hits, base = [np.zeros(1, np.int64)], [], 0
    ids_global = []
    for c in chroms:
        ii = np.nonzero(data["chrom"] == c)[0]
        qi = np.nonzero(data["qchrom"] == c)[0]
        t = oracle_mod.OracleTree(data["low"][ii], data["high"][ii])
        off, h = t.find_overlaps_batch(data["qlow"][qi], data["qhigh"][qi])
        offs.append(off[1:].astype(np.int64) + base)
        base += int(off[-1])
        hits.append(ii[h].astype(np.int32))
        ids_global.append(qi)
    return (np.concatenate(offs), np.concatenate(hits) if hits else np.zeros(0, np.int32),
            np.concatenate(ids_global) if ids_global else np.zeros(0, np.int64))


def _worker(rank, world, port, tmpdir, self_overlap=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ivtree_oracle as oracle_mod
        if self_overlap:
            # BASELINE config 5 in miniature, as bench.py --config 5 shards it under WORLD_SIZE > 1: queries = the
            # intervals, chromosome sizes skewed (proportional to length), LPT on Q_c log2 N_c + E[H_c]
            data = synth.gen_genome(9000, 0, 1000)
            data["qchrom"], data["qlow"], data["qhigh"] = data["chrom"], data["low"], data["high"]
            ni = synth.split_by_length(9000)
            assign = sharding.lpt_assign(sharding.chrom_work(ni, ni, ni * ni * 1002.0 / synth.HG38_LENGTHS), world)
        else:
            data = synth.gen_genome(6000, 4000, 1000)
            assign = sharding.lpt_assign(sharding.chrom_work(synth.split_by_length(6000), synth.split_by_length(4000)), world)
        off, hits, qids = _rank_csr(oracle_mod, data, assign[rank])
        res = sharding.gatherv_csr(torch.from_numpy(off), torch.from_numpy(hits), dst=0)
        if rank == 0:
            g_off, g_hits, qs, hs = res
            # expected: the concatenation of every rank's CSR in rank order
            exp_off, exp_hits, exp_q = [np.zeros(1, np.int64)], [], []
            base = 0
            for r in range(world):
                o, h, qi = _rank_csr(oracle_mod, data, assign[r])
                exp_off.append(o[1:] + base)
                base += int(o[-1])
                exp_hits.append(h)
                exp_q.append(qi)
                assert qs[r] == o.size - 1 and hs[r] == h.size
            assert np.array_equal(g_off.numpy(), np.concatenate(exp_off))
            assert np.array_equal(g_hits.numpy(), np.concatenate(exp_hits))
            # and it is the whole-genome answer: every query appears exactly once
            assert np.array_equal(np.sort(np.concatenate(exp_q)), np.arange(data["qlow"].size))
            open(os.path.join(tmpdir, "ok"), "w").write("ok")
        else:
            assert res is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gatherv_csr_gloo(world, tmp_path, oracle):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "ok").exists()


def test_gatherv_csr_gloo_self_overlap_skewed(tmp_path, oracle):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), True), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_gatherv_single_rank(tmp_path):
    port = _free_port()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        off = torch.tensor([0, 2, 2, 5], dtype=torch.int64)
        hits = torch.tensor([4, 9, 1, 2, 3], dtype=torch.int32)
        g_off, g_hits, qs, hs = sharding.gatherv_csr(off, hits)
        assert torch.equal(g_off, off) and torch.equal(g_hits, hits) and qs.tolist() == [3] and hs.tolist() == [5]
    finally:
        dist.destroy_process_group()
