"""BASELINE.json configs[4] on ONE GPU: whole-genome self-overlap (queries = the intervals), N = Q = 50 M over 24
chromosomes (counts proportional to length), H ~ 0.86 G hit ids. Checked through size-independent properties:
every per-query count equals the sort+searchsorted count, and a strided sample of hit lists satisfies the
predicate, ascends, stays on its chromosome and contains the query itself. Prints the timings it observed."""
import json
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config5_whole_genome_self_overlap_50M(oracle):
    import torch
    from binary_amd import IntervalIndex, synth
    N = 50_000_000
    dev = torch.device("cuda:0")
    data = synth.gen_genome(N, 0, 1000)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    d_c, d_lo, d_hi = to(data["chrom"]), to(data["low"]), to(data["high"])
    with IntervalIndex(0) as idx:
        t0 = time.perf_counter()
        idx.insert_node(d_lo, d_hi, d_c)
        idx.build()
        build_s = time.perf_counter() - t0
        st = idx.stats()
        off = torch.empty(N + 1, dtype=torch.int64, device=dev)
        idx.count_overlaps_device(d_lo, d_hi, d_c, offsets=off)
        H = int(off[-1].item())
        assert 0.7e9 < H < 1.0e9          # SURVEY §8d expects about 0.81 G
        hits = torch.empty(H, dtype=torch.int32, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        idx.query_device(d_lo, d_hi, off, hits, qchrom=d_c, sort_by_id=False)
        e1.record()
        torch.cuda.synchronize()
        single_ms = e0.elapsed_time(e1)
        e0.record()
        idx.query_device(d_lo, d_hi, off, hits, qchrom=d_c, sort_by_id=True)
        e1.record()
        torch.cuda.synchronize()
        sorted_ms = e0.elapsed_time(e1)
        # the self-overlap entry point (index order inside, lists written at their ids' offsets): same CSR, bit for bit
        idx.query_device(d_lo, d_hi, off, hits, qchrom=d_c, sort_by_id=False)
        off_s = torch.full_like(off, -1)
        hits_s = torch.full_like(hits, -1)
        idx.self_overlaps_device(off_s, hits_s, sort_by_id=False)   # (first call: builds the slot-order query batch)
        e0.record()
        idx.self_overlaps_device(off_s, hits_s, sort_by_id=False)
        e1.record()
        torch.cuda.synchronize()
        self_ms = e0.elapsed_time(e1)
        idx.stream_status()
        assert torch.equal(off_s, off) and torch.equal(hits_s, hits)
        # ... and with ascending ids inside every list (k_permute_lines<true> orders the lists while the piece sits in LDS)
        idx.query_device(d_lo, d_hi, off, hits, qchrom=d_c, sort_by_id=True)
        off_s.fill_(-1)
        hits_s.fill_(-1)
        e0.record()
        idx.self_overlaps_device(off_s, hits_s, sort_by_id=True)
        e1.record()
        torch.cuda.synchronize()
        self_sorted_ms = e0.elapsed_time(e1)
        idx.stream_status()
        assert torch.equal(off_s, off) and torch.equal(hits_s, hits)
        del off_s, hits_s
        torch.cuda.synchronize()
        off_h = off.cpu().numpy()
        hits_h = hits.cpu().numpy().view(np.uint32)
    cnt = np.diff(off_h)
    for c in range(24):
        sel = data["chrom"] == c
        exp = oracle.count_overlaps_numpy(data["low"][sel], data["high"][sel], data["low"][sel], data["high"][sel])
        assert np.array_equal(cnt[sel], exp), f"chromosome {c}"
    for q in np.arange(0, N, N // 20_000):
        h = hits_h[off_h[q]:off_h[q + 1]]
        assert np.all(data["chrom"][h] == data["chrom"][q])
        assert np.all(data["low"][h] <= data["high"][q]) and np.all(data["high"][h] >= data["low"][q])
        assert np.all(np.diff(h.astype(np.int64)) > 0) and q in h
    print(json.dumps({"N": N, "H": H, "append+build_s": build_s, "build_ms": st["build_ms"], "segments": st["n_segments"],
                      "index_bytes": st["index_bytes"], "single_pass_ms": single_ms, "self_overlaps_ms": self_ms, "self_overlaps_sorted_ids_ms": self_sorted_ms, "single_pass_sorted_ids_ms": sorted_ms,
                      "gqps_single_pass": N / single_ms / 1e6}))
