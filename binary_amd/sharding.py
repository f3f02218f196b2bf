"""Per-chromosome sharding of the overlap path across the GPUs of one node.

The reference's only decomposition is one task per chromosome (standalone/sv2nl/include/mapper.hpp:238-246):
an interval on chromosome c can only meet a query on chromosome c, so whole chromosomes are independent
units. One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for
tests); no collective on the data path. The single exchange step is the final gatherv of the per-chromosome
CSR hit lists to rank 0: an all_gather of two integers per rank, then one variable-length point-to-point
transfer per peer (RCCL has no native gatherv; each peer->root transfer rides its own direct xGMI link).
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch
import torch.distributed as dist


def lpt_assign(weights: Sequence[float], nranks: int) -> list[list[int]]:
    """Longest-processing-time-first: heaviest unit to the least loaded rank. Returns unit ids per rank
    (each list ascending). Deterministic: ties go to the lower rank / lower unit id."""
    w = np.asarray(weights, dtype=np.float64)
    order = sorted(range(w.size), key=lambda i: (-w[i], i))
    load = [0.0] * nranks
    out: list[list[int]] = [[] for _ in range(nranks)]
    for i in order:
        r = min(range(nranks), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += float(w[i])
    return [sorted(x) for x in out]


def imbalance(weights: Sequence[float], assignment: list[list[int]]) -> float:
    """max rank load / mean rank load."""
    w = np.asarray(weights, dtype=np.float64)
    loads = np.array([w[a].sum() if a else 0.0 for a in assignment])
    return float(loads.max() / loads.mean()) if loads.mean() > 0 else 1.0


def chrom_work(n_intervals: Sequence[int], n_queries: Sequence[int], n_hits: Sequence[int] | None = None):
    """Estimated work of a chromosome: Q_c * log2(N_c) + H_c."""
    n = np.maximum(np.asarray(n_intervals, dtype=np.float64), 2.0)
    q = np.asarray(n_queries, dtype=np.float64)
    h = np.zeros_like(q) if n_hits is None else np.asarray(n_hits, dtype=np.float64)
    return q * np.log2(n) + h


def gatherv_csr(offsets: torch.Tensor, hits: torch.Tensor, dst: int = 0, group=None):
    """Gathers every rank's CSR (offsets int64[q_r + 1], hits int32[H_r]) on `dst`.

    Returns on dst: (offsets int64[sum q_r + 1], hits[sum H_r], q_per_rank, h_per_rank) with rank blocks in
    rank order and offsets rebased to the concatenated hit array; on other ranks: None.
    Message shapes: one all_gather of 2 x int64 per rank, then per peer one send of its offsets (q_r + 1) x 8 B
    and one of its hits H_r x 4 B, all posted as one batch so the peer->root links work in parallel."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = offsets.device
    q_r = offsets.numel() - 1
    h_r = int(hits.numel())
    mine = torch.tensor([q_r, h_r], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, mine, group=group)
    sizes_h = torch.stack(sizes).cpu().numpy()
    qs, hs = sizes_h[:, 0].astype(np.int64), sizes_h[:, 1].astype(np.int64)
    if world == 1:
        return offsets.clone(), hits.clone(), qs, hs
    if rank != dst:
        ops = [dist.P2POp(dist.isend, offsets, dst, group=group)]
        if h_r:
            ops.append(dist.P2POp(dist.isend, hits, dst, group=group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        return None
    out_off = torch.empty(int(qs.sum()) + 1, dtype=torch.int64, device=dev)
    out_hits = torch.empty(int(hs.sum()), dtype=hits.dtype, device=dev)
    q_disp = np.concatenate([[0], np.cumsum(qs)])
    h_disp = np.concatenate([[0], np.cumsum(hs)])
    staged = {}
    ops = []
    for r in range(world):
        if r == dst:
            continue
        staged[r] = torch.empty(int(qs[r]) + 1, dtype=torch.int64, device=dev)
        ops.append(dist.P2POp(dist.irecv, staged[r], r, group=group))
        if hs[r]:
            ops.append(dist.P2POp(dist.irecv, out_hits[int(h_disp[r]):int(h_disp[r + 1])], r, group=group))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    for r in range(world):
        src_off = offsets if r == dst else staged[r]
        out_off[int(q_disp[r]):int(q_disp[r + 1])] = src_off[:-1] + int(h_disp[r])
        if r == dst and h_r:
            out_hits[int(h_disp[r]):int(h_disp[r + 1])] = hits
    out_off[-1] = int(hs.sum())
    return out_off, out_hits, qs, hs
