"""Portable synthetic interval / query generator: splitmix64 + multiply-high uniform.

Bit-identical in numpy here and in any C/C++ restatement:
    state += 0x9E3779B97F4A7C15; z = state
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9; z = (z ^ (z >> 27)) * 0x94D049BB133111EB; z ^= z >> 31
    uniform(n) = (z * n) >> 64
Interval i of chromosome c draws start = uniform(L_c), then len = 1 + uniform(Lmax);
end = min(start + len, 2^32 - 1). Generation order is insertion order (unsorted).
"""
from __future__ import annotations

import numpy as np

GAMMA = 0x9E3779B97F4A7C15
SEED_INTERVALS = 0xB1A40000
SEED_QUERIES = 0xC0FE0000
U32_MAX = 0xFFFFFFFF

# hg38 primary contigs chr1..chr22, chrX, chrY (lengths as in the reference fixture header,
# test/data/debug_uncom.vcf:5-63,457-458); sum = 3,088,269,832
HG38 = (
    ("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555),
    ("chr5", 181538259), ("chr6", 170805979), ("chr7", 159345973), ("chr8", 145138636),
    ("chr9", 138394717), ("chr10", 133797422), ("chr11", 135086622), ("chr12", 133275309),
    ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189), ("chr16", 90338345),
    ("chr17", 83257441), ("chr18", 80373285), ("chr19", 58617616), ("chr20", 64444167),
    ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415),
)
HG38_LENGTHS = np.array([l for _, l in HG38], dtype=np.int64)


def splitmix64_stream(seed: int, count: int) -> np.ndarray:
    """The first `count` outputs of splitmix64 started at `seed` (uint64 array)."""
    with np.errstate(over="ignore"):
        k = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + k * np.uint64(GAMMA)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def mulhi_u64_u32(z: np.ndarray, n: int) -> np.ndarray:
    """(z * n) >> 64 for uint64 z and 0 < n <= 2^32, without 128-bit integers."""
    assert 0 < n <= (1 << 32)
    if n == (1 << 32):
        return z >> np.uint64(32)
    n64 = np.uint64(n)
    zh, zl = z >> np.uint64(32), z & np.uint64(U32_MAX)
    return (zh * n64 + ((zl * n64) >> np.uint64(32))) >> np.uint64(32)


def gen_intervals(n: int, length: int, lmax: int = 1000, chrom_index: int = 0, seed_base: int = SEED_INTERVALS):
    """n intervals on a chromosome of `length` bp: (low u32[n], high u32[n])."""
    z = splitmix64_stream(seed_base + chrom_index, 2 * n)
    start = mulhi_u64_u32(z[0::2], length)
    ln = np.uint64(1) + mulhi_u64_u32(z[1::2], lmax)
    end = np.minimum(start + ln, np.uint64(U32_MAX))
    return start.astype(np.uint32), end.astype(np.uint32)


def gen_point_queries(q: int, length: int, chrom_index: int = 0, seed_base: int = SEED_QUERIES):
    z = splitmix64_stream(seed_base + chrom_index, q)
    p = mulhi_u64_u32(z, length).astype(np.uint32)
    return p, p.copy()


def gen_range_queries(q: int, length: int, lmax: int = 1000, chrom_index: int = 0, seed_base: int = SEED_QUERIES):
    return gen_intervals(q, length, lmax, chrom_index, seed_base)


def split_by_length(total: int, lengths=HG38_LENGTHS) -> np.ndarray:
    """Per-chromosome counts proportional to chromosome length, remainder to the first (chr1)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    cnt = (total * lengths) // int(lengths.sum())
    cnt[0] += total - int(cnt.sum())
    return cnt


def gen_genome(total_intervals: int, total_queries: int, lmax: int = 1000, point_queries: bool = False,
               lengths=HG38_LENGTHS, chrom_ids=None):
    """Config-3 style set: intervals and queries over all chromosomes, grouped by chromosome.

    Returns dict of uint32 arrays: chrom, low, high, qchrom, qlow, qhigh. `chrom_ids` restricts the
    output to those chromosomes (per-chromosome sharding) without changing any chromosome's data."""
    lengths = np.asarray(lengths, dtype=np.int64)
    ni, nq = split_by_length(total_intervals, lengths), split_by_length(total_queries, lengths)
    ids = range(len(lengths)) if chrom_ids is None else chrom_ids
    out = {k: [] for k in ("chrom", "low", "high", "qchrom", "qlow", "qhigh")}
    for c in ids:
        lo, hi = gen_intervals(int(ni[c]), int(lengths[c]), lmax, c)
        if point_queries:
            qlo, qhi = gen_point_queries(int(nq[c]), int(lengths[c]), c)
        else:
            qlo, qhi = gen_range_queries(int(nq[c]), int(lengths[c]), lmax, c)
        out["chrom"].append(np.full(lo.size, c, dtype=np.uint32))
        out["low"].append(lo)
        out["high"].append(hi)
        out["qchrom"].append(np.full(qlo.size, c, dtype=np.uint32))
        out["qlow"].append(qlo)
        out["qhigh"].append(qhi)
    return {k: (np.concatenate(v) if v else np.zeros(0, np.uint32)) for k, v in out.items()}
