"""Tree-free checks of a CSR overlap result with torch tensor ops (for bench.py and the tests; nothing on the query
path calls this, and it uses no CPU tree): every per-query count against a sort + searchsorted count, every reported pair against
the predicate, pairs distinct — together: the exact hit SET of every query (reference semantics:
interval_tree.hpp:119-121, 306-334) — and SURVEY.md §8d's order-independent 64-bit checksum, sum and XOR of
hash(global query id, global interval id), which lets ranks that hold different chromosomes vouch for a gathered CSR.

Global ids: the position in the whole-genome set grouped by chromosome in generation order (synth.gen_genome), whatever
subset of chromosomes a rank holds."""
from __future__ import annotations

import numpy as np
import torch

_GAMMA = 0x9E3779B97F4A7C15
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB


def _s64(x: int) -> int:
    """the uint64 constant as the int64 with the same bits"""
    x &= 0xFFFFFFFFFFFFFFFF
    return x - (1 << 64) if x >= (1 << 63) else x


def _lsr(x: torch.Tensor, k: int) -> torch.Tensor:
    return (x >> k) & ((1 << (64 - k)) - 1)


def pair_hash(gq: torch.Tensor, gi: torch.Tensor) -> torch.Tensor:
    """splitmix64's finalizer of (gq << 32 | gi) + gamma, in wrapping int64 arithmetic (same bits as pair_hash_np)."""
    z = ((gq.to(torch.int64) << 32) | gi.to(torch.int64)) + _s64(_GAMMA)
    z = (z ^ _lsr(z, 30)) * _s64(_M1)
    z = (z ^ _lsr(z, 27)) * _s64(_M2)
    return z ^ _lsr(z, 31)


def pair_hash_np(gq: np.ndarray, gi: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = ((gq.astype(np.uint64) << np.uint64(32)) | gi.astype(np.uint64)) + np.uint64(_GAMMA)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_M1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_M2)
        return z ^ (z >> np.uint64(31))


def _xor_reduce(x: torch.Tensor) -> int:
    while x.numel() > 1:
        n = x.numel()
        h = n // 2
        y = x[:h] ^ x[h:2 * h]
        x = torch.cat([y, x[2 * h:]]) if n & 1 else y
    return int(x.item()) & 0xFFFFFFFFFFFFFFFF if x.numel() else 0


def checksum_csr(offsets: torch.Tensor, hits: torch.Tensor, gq_of_q: torch.Tensor, gi_of_i: torch.Tensor,
                 chunk: int = 4_000_000) -> tuple[int, int, int]:
    """(sum mod 2^64, xor, pairs) of pair_hash over every (query, hit) pair of the CSR. `gq_of_q[q]` / `gi_of_i[id]` map
    the CSR's own query positions / hit ids to global ids. Works in chunks of queries (config 5 has 0.86 G pairs)."""
    Q = offsets.numel() - 1
    s, x, n = 0, 0, 0
    for q0 in range(0, Q, chunk):
        q1 = min(Q, q0 + chunk)
        o = offsets[q0:q1 + 1].to(torch.int64)
        b, e = int(o[0].item()), int(o[-1].item())
        if e == b:
            continue
        cnt = o[1:] - o[:-1]
        gq = torch.repeat_interleave(gq_of_q[q0:q1].to(torch.int64), cnt)
        gi = gi_of_i[hits[b:e].to(torch.int64) & 0xFFFFFFFF]
        h = pair_hash(gq, gi)
        s = (s + (int(h.sum().item()) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
        x ^= _xor_reduce(h)
        n += e - b
    return s, x, n


def checksum_csr_np(offsets: np.ndarray, hits: np.ndarray, gq_of_q: np.ndarray, gi_of_i: np.ndarray):
    cnt = np.diff(offsets.astype(np.int64))
    gq = np.repeat(gq_of_q.astype(np.int64), cnt)
    h = pair_hash_np(gq, gi_of_i[hits.astype(np.int64) & 0xFFFFFFFF])
    with np.errstate(over="ignore"):
        return int(h.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(h)) if h.size else 0, int(h.size)


def global_id_maps(chroms, per_chrom_counts) -> np.ndarray:
    """Global ids of a shard that holds `chroms` (ascending, grouped, generation order inside each): the shard's k-th
    element of chromosome c is global element base[c] + k, base = exclusive prefix of the whole genome's counts."""
    per = np.asarray(per_chrom_counts, dtype=np.int64)
    base = np.concatenate([[0], np.cumsum(per)])
    parts = [np.arange(base[c], base[c] + per[c], dtype=np.int64) for c in chroms]
    return np.concatenate(parts) if parts else np.zeros(0, np.int64)


def verify_shard(chrom, low, high, qchrom, qlow, qhigh, offsets: torch.Tensor, hits: torch.Tensor,
                 chunk: int = 8_000_000) -> dict:
    """Exact check of one CSR against the closed-interval predicate, without a tree: (a) offsets[0] == 0 and every count
    equals #(low <= q.high) - #(high < q.low) on the query's chromosome (valid for low <= high intervals, which is what
    the synthetic sets hold); (b) every reported id lies on the query's chromosome and overlaps it; (c) no id twice in a
    list. (a) + (b) + (c) => every list is exactly the reference's hit set. All arguments device tensors (coordinates as
    the int32 views the library takes); chrom / qchrom may be None (one chromosome)."""
    dev = offsets.device
    u = lambda t: t.to(torch.int64) & 0xFFFFFFFF
    lo, hi, ql, qh = u(low), u(high), u(qlow), u(qhigh)
    c = u(chrom) if chrom is not None else torch.zeros_like(lo)
    qc = u(qchrom) if qchrom is not None else torch.zeros_like(ql)
    off = offsets.to(torch.int64)
    Q = ql.numel()
    res = {"queries": Q, "pairs": int(off[-1].item()) if off.numel() else 0}
    assert bool((lo <= hi).all().item()) and bool((ql <= qh).all().item()), "verify_shard needs low <= high"
    kl = torch.sort((c << 32) | lo)[0]
    kh = torch.sort((c << 32) | hi)[0]
    exp = torch.searchsorted(kl, (qc << 32) | qh, right=True) - torch.searchsorted(kh, (qc << 32) | ql, right=False)
    del kl, kh
    cnt = off[1:] - off[:-1]
    res["counts_ok"] = bool(off[0].item() == 0) and bool(torch.equal(cnt, exp))
    pairs_ok, distinct_ok = True, True
    for q0 in range(0, Q, chunk):
        q1 = min(Q, q0 + chunk)
        b, e = int(off[q0].item()), int(off[q1].item())
        if e == b:
            continue
        qid = torch.repeat_interleave(torch.arange(q0, q1, device=dev), cnt[q0:q1])
        iid = u(hits[b:e])
        if bool((iid >= lo.numel()).any().item()):
            pairs_ok = False
            break
        ok = (c[iid] == qc[qid]) & (lo[iid] <= qh[qid]) & (hi[iid] >= ql[qid])
        pairs_ok = pairs_ok and bool(ok.all().item())
        key = torch.sort((qid << 32) | iid)[0]
        distinct_ok = distinct_ok and bool((key[1:] != key[:-1]).all().item())
        del qid, iid, ok, key
    res["pairs_ok"], res["distinct_ok"] = pairs_ok, distinct_ok
    res["ok"] = res["counts_ok"] and pairs_ok and distinct_ok
    return res
