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


# ---- the bench's tree-free verification and the checksum that vouches for a gathered CSR (binary_amd/csr_check.py) -----

def _dense_genome():
    # the bench's generator on short "chromosomes", so that lists are a few ids long
    lengths = np.array([40_000, 25_000, 9_000, 31_000, 5_000], dtype=np.int64)
    return synth.gen_genome(3000, 2500, 1000, lengths=lengths), lengths


def _oracle_csr(oracle_mod, data, chroms):
    off, hits, _ = _rank_csr(oracle_mod, data, chroms)
    return off, hits


def test_csr_check_accepts_the_oracle_and_rejects_every_kind_of_damage(oracle):
    from binary_amd import csr_check
    data, lengths = _dense_genome()
    off, hits = _oracle_csr(oracle, data, range(len(lengths)))
    assert hits.size > 5000
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int32))
    args = (T(data["chrom"]), T(data["low"]), T(data["high"]), T(data["qchrom"]), T(data["qlow"]), T(data["qhigh"]))
    v = csr_check.verify_shard(*args, torch.from_numpy(off), torch.from_numpy(hits), chunk=700)
    assert v["ok"] and v["pairs"] == hits.size
    gq, gi = np.arange(data["qlow"].size), np.arange(data["low"].size)
    a = csr_check.checksum_csr(torch.from_numpy(off), torch.from_numpy(hits), torch.from_numpy(gq), torch.from_numpy(gi), chunk=333)
    assert a == csr_check.checksum_csr_np(off, hits, gq, gi)
    # order inside a list does not matter, content does
    q = int(np.argmax(np.diff(off)))
    h2 = hits.copy()
    h2[off[q]:off[q + 1]] = h2[off[q]:off[q + 1]][::-1]
    assert csr_check.checksum_csr(torch.from_numpy(off), torch.from_numpy(h2), torch.from_numpy(gq), torch.from_numpy(gi)) == a
    assert csr_check.verify_shard(*args, torch.from_numpy(off), torch.from_numpy(h2))["ok"]
    h3 = hits.copy()
    h3[off[q]] = h3[off[q] + 1]          # an id twice (and one missing): counts still right
    v3 = csr_check.verify_shard(*args, torch.from_numpy(off), torch.from_numpy(h3))
    assert v3["counts_ok"] and not v3["distinct_ok"] and not v3["ok"]
    assert csr_check.checksum_csr(torch.from_numpy(off), torch.from_numpy(h3), torch.from_numpy(gq), torch.from_numpy(gi)) != a
    h4 = hits.copy()
    h4[off[q]] = (h4[off[q]] + 1500) % data["low"].size   # a non-overlapping id
    assert not csr_check.verify_shard(*args, torch.from_numpy(off), torch.from_numpy(h4))["pairs_ok"]
    o5 = off.copy()
    o5[q + 1] -= 1                        # a hit moved to the next query
    assert not csr_check.verify_shard(*args, torch.from_numpy(o5), torch.from_numpy(hits))["counts_ok"]


def _checksum_worker(rank, world, port, tmpdir):
    """bench.py's N > 1 verification flow in miniature: every rank checksums its shard with GLOBAL ids, the sums are
    all-reduced, rank 0 recomputes the checksum over the gathered CSR block by block and compares."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ivtree_oracle as oracle_mod
        from binary_amd import csr_check
        data, lengths = _dense_genome()
        per_i, per_q = synth.split_by_length(3000, lengths), synth.split_by_length(2500, lengths)
        assign = sharding.lpt_assign(sharding.chrom_work(per_i, per_q), world)
        # a rank's shard as bench.py holds it: only its chromosomes, ids local to the shard
        mine = synth.gen_genome(3000, 2500, 1000, lengths=lengths, chrom_ids=assign[rank])
        off, hits = _oracle_csr(oracle_mod, mine, assign[rank])
        gq = torch.from_numpy(csr_check.global_id_maps(assign[rank], per_q))
        gi = torch.from_numpy(csr_check.global_id_maps(assign[rank], per_i))
        s, x, n = csr_check.checksum_csr(torch.from_numpy(off), torch.from_numpy(hits), gq, gi)
        t = torch.tensor([s & 0xFFFFFFFF, s >> 32, n], dtype=torch.int64)
        dist.all_reduce(t)
        xs = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(xs, torch.tensor([x & 0xFFFFFFFF, x >> 32], dtype=torch.int64))
        tot_s = (int(t[0]) + (int(t[1]) << 32)) & 0xFFFFFFFFFFFFFFFF
        tot_x = 0
        for v in xs:
            tot_x ^= int(v[0]) | (int(v[1]) << 32)
        res = sharding.gatherv_csr(torch.from_numpy(off), torch.from_numpy(hits), dst=0)
        if rank == 0:
            g_off, g_hits, qs, hs = res
            qd = np.concatenate([[0], np.cumsum(qs)])
            gs = gx = gp = 0
            for r in range(world):
                rs, rx, rp = csr_check.checksum_csr(g_off[int(qd[r]):int(qd[r + 1]) + 1], g_hits,
                                                    torch.from_numpy(csr_check.global_id_maps(assign[r], per_q)),
                                                    torch.from_numpy(csr_check.global_id_maps(assign[r], per_i)))
                gs, gx, gp = (gs + rs) & 0xFFFFFFFFFFFFFFFF, gx ^ rx, gp + rp
            assert (gs, gx, gp) == (tot_s, tot_x, int(t[2]))
            # ... and it is the whole genome's checksum, whatever the sharding
            off_w, hits_w = _oracle_csr(oracle_mod, data, range(len(lengths)))
            whole = csr_check.checksum_csr_np(off_w, hits_w, np.arange(2500), np.arange(3000))
            assert whole == (gs, gx, gp)
            # a damaged block is noticed
            bad = g_hits.clone()
            bad[-1] = bad[-2]
            r = world - 1
            rs, rx, _ = csr_check.checksum_csr(g_off[int(qd[r]):int(qd[r + 1]) + 1], bad,
                                               torch.from_numpy(csr_check.global_id_maps(assign[r], per_q)),
                                               torch.from_numpy(csr_check.global_id_maps(assign[r], per_i)))
            assert (rs, rx) != csr_check.checksum_csr(g_off[int(qd[r]):int(qd[r + 1]) + 1], g_hits,
                                                      torch.from_numpy(csr_check.global_id_maps(assign[r], per_q)),
                                                      torch.from_numpy(csr_check.global_id_maps(assign[r], per_i)))[:2]
            open(os.path.join(tmpdir, "ok"), "w").write("ok")
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gathered_checksum_equals_the_ranks_gloo(world, tmp_path, oracle):
    port = _free_port()
    mp.spawn(_checksum_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "ok").exists()
