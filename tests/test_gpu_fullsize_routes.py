"""Full BASELINE sizes through exactly the kernel routes bench.py times.

  config 3 (24 hg38 chromosomes, 10 M intervals x 10 M range queries), bivx_query_dev_s with DEFAULT routing
      -> k_query_pipe: generation order AND position-sorted queries, index order AND ascending ids, every
      (query, hit) pair against the tree oracle (the port of the reference's red-black interval tree, one tree per
      chromosome built by sequential inserts in generation order, find_overlaps per query: interval_tree.hpp:306-334).
  config 5 (50 M self-overlap) in POSITION order -> k_query_pipe_dense (the device-side order probe picks it):
      every count against the tree-free sort + searchsorted count, every 50th hit list against the tree oracle.

The routed kernel's name is asserted (bivx_query_kernel_name), so a change of the routing thresholds cannot silently
move these checks to another kernel."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_csr(data, stride=1):
    """(offsets, GLOBAL hit ids in pre-order, selected query indices) of the tree oracle, chromosomes over threads."""
    import bench
    ncores = min(len(os.sched_getaffinity(0)), 16)
    off_o, hits_o, sel, _ = bench.cpu_baseline(data, ncores, stride)
    return off_o, hits_o, sel


def _pairs_sorted(torch, qid_of_slot, hits):
    """sorted (original query id << 32 | hit id) keys of a CSR given every slot's query id"""
    key = (qid_of_slot << 32) | (hits.to(torch.int64) & 0xFFFFFFFF)
    return torch.sort(key)[0]


def test_config3_full_size_pipelined_route_all_pairs_vs_tree_oracle(oracle):
    import torch
    from binary_amd import IntervalIndex, synth
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    data = synth.gen_genome(10_000_000, 10_000_000, 1000, point_queries=False)
    Q = int(data["qlow"].size)
    off_o, hits_o, sel = _oracle_csr(data)
    assert sel.size == Q
    cnt_o = np.diff(off_o.astype(np.int64))
    H = int(off_o[-1])
    assert 25_000_000 < H < 40_000_000            # SURVEY §8d expects about 32 M
    # the oracle's pairs, sorted (query ids are the generation-order ones)
    exp = torch.from_numpy((np.repeat(np.arange(Q, dtype=np.int64), cnt_o) << 32) | hits_o.astype(np.int64)).to(dev)
    exp = torch.sort(exp)[0]
    perm = np.lexsort((data["qlow"], data["qchrom"]))
    batches = {
        "generation order": (np.arange(Q, dtype=np.int64), data["qchrom"], data["qlow"], data["qhigh"]),
        "position-sorted": (perm.astype(np.int64), data["qchrom"][perm], data["qlow"][perm], data["qhigh"][perm]),
    }
    with IntervalIndex(0) as idx:
        idx.insert_node(to(data["low"]), to(data["high"]), to(data["chrom"]))
        idx.build()
        for name, (orig, qc, qlo, qhi) in batches.items():
            d_qc, d_qlo, d_qhi = to(qc), to(qlo), to(qhi)
            d_orig = torch.from_numpy(orig).to(dev)
            for sort_by_id in (False, True):
                assert idx.query_kernel_name(Q, H, sort_by_id) == "k_query_pipe", (name, sort_by_id)
                d_off = torch.empty(Q + 1, dtype=torch.int64, device=dev)
                d_hits = torch.full((H,), -1, dtype=torch.int32, device=dev)
                idx.query_device(d_qlo, d_qhi, d_off, d_hits, qchrom=d_qc, sort_by_id=sort_by_id)
                idx.stream_status()
                cnt = d_off[1:] - d_off[:-1]
                assert int(d_off[0].item()) == 0 and int(d_off[-1].item()) == H, (name, sort_by_id)
                assert np.array_equal(cnt.cpu().numpy(), cnt_o[orig]), (name, sort_by_id)       # every count
                slot_q = torch.repeat_interleave(torch.arange(Q, device=dev), cnt)
                got = _pairs_sorted(torch, d_orig[slot_q], d_hits)
                assert torch.equal(got, exp), (name, sort_by_id)                                # every pair
                if sort_by_id:  # ... and, as written, ids ascend inside every query
                    raw = (slot_q << 32) | (d_hits.to(torch.int64) & 0xFFFFFFFF)
                    assert bool((raw[1:] > raw[:-1]).all()), name
                del d_off, d_hits, slot_q, got
            del d_qc, d_qlo, d_qhi


def test_config5_full_size_position_sorted_dense_route(oracle):
    import torch
    from binary_amd import IntervalIndex, synth
    N = 50_000_000
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    data = synth.gen_genome(N, 0, 1000)
    perm = np.lexsort((data["low"], data["chrom"]))
    qc, qlo, qhi = data["chrom"][perm], data["low"][perm], data["high"][perm]
    with IntervalIndex(0) as idx:
        idx.insert_node(to(data["low"]), to(data["high"]), to(data["chrom"]))
        idx.build()
        d_qc, d_qlo, d_qhi = to(qc), to(qlo), to(qhi)
        d_off = torch.empty(N + 1, dtype=torch.int64, device=dev)
        idx.count_overlaps_device(d_qlo, d_qhi, d_qc, offsets=d_off)
        H = int(d_off[-1].item())
        assert 0.7e9 < H < 1.0e9                  # SURVEY §8d expects about 0.81 G
        # many ids per query: both kernels are launched and the order probe gives the batch to the dense one
        assert idx.query_kernel_name(N, H, False) == "k_query_pipe_dense|k_query_fused"
        d_hits = torch.full((H,), -1, dtype=torch.int32, device=dev)
        d_off.fill_(-1)
        idx.query_device(d_qlo, d_qhi, d_off, d_hits, qchrom=d_qc, sort_by_id=False)
        idx.stream_status()
        off = d_off.cpu().numpy()
        hits = d_hits.cpu().numpy().view(np.uint32)
        del d_hits, d_off
    assert off[0] == 0 and off[-1] == H
    cnt = np.diff(off)
    # every count against the tree-free count, chromosome by chromosome (queries are grouped: position order)
    b = np.searchsorted(qc, np.arange(25))
    for c in range(24):
        s = data["chrom"] == c
        exp = oracle.count_overlaps_numpy(data["low"][s], data["high"][s], qlo[b[c]:b[c + 1]], qhi[b[c]:b[c + 1]])
        assert np.array_equal(cnt[b[c]:b[c + 1]], exp), f"chromosome {c}"
    # every 50th list against the tree oracle (trees of all 50 M intervals, sequential inserts in generation order)
    batch = dict(chrom=data["chrom"], low=data["low"], high=data["high"], qchrom=qc, qlow=qlo, qhigh=qhi)
    off_o, hits_o, sel = _oracle_csr(batch, stride=50)
    assert np.array_equal(cnt[sel], np.diff(off_o.astype(np.int64)))
    cs = cnt[sel]
    loc = np.zeros(sel.size + 1, np.int64)
    np.cumsum(cs, out=loc[1:])
    flat = np.arange(loc[-1], dtype=np.int64) + np.repeat(off[sel] - loc[:-1], cs)   # the sampled lists' slots
    kq = np.repeat(np.arange(sel.size, dtype=np.int64), cs) << 32
    got = np.sort(kq | hits[flat].astype(np.int64))
    exp = np.sort(kq | hits_o.astype(np.int64))
    assert np.array_equal(got, exp)
    assert bool((hits != 0xFFFFFFFF).all())      # every slot of the CSR was written
