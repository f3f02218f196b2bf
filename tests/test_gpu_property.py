"""Property-based parity (hypothesis): small adversarial interval sets — coordinates clustered around 0, 2^16, 2^31
and 2^32-1, lengths straddling the packed-record limit (65535/65536) and the length-class cuts, low > high entries,
several chromosomes — against the brute-force closed-interval predicate (interval_tree.hpp:119-121) in numpy.
Every entry point must give the same hit sets: two-pass, single-pass, existence, and ids sorted or not."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

pytestmark = pytest.mark.gpu

M32 = 0xFFFFFFFF
ANCHORS = [0, 1, 255, 65535, 65536, 65537, 1 << 20, (1 << 31) - 1, 1 << 31, M32 - 65536, M32 - 1, M32]
LENGTHS = [0, 0, 1, 2, 7, 100, 1000, 65534, 65535, 65536, 70000, 1 << 20, 1 << 31, M32]


@st.composite
def coord(draw):
    a = draw(st.sampled_from(ANCHORS))
    d = draw(st.integers(-300, 300))
    return min(max(a + d, 0), M32)


@st.composite
def interval(draw, allow_inverted=True):
    lo = draw(coord())
    ln = draw(st.sampled_from(LENGTHS)) + draw(st.integers(0, 3))
    hi = min(lo + ln, M32)
    if allow_inverted and draw(st.integers(0, 9)) == 0:
        lo, hi = hi, lo
    return lo, hi


@st.composite
def problem(draw):
    n = draw(st.integers(0, 120))
    q = draw(st.integers(1, 80))
    nchrom = draw(st.sampled_from([1, 1, 2, 5]))
    iv = [draw(interval()) for _ in range(n)]
    qs = [draw(interval(allow_inverted=draw(st.booleans()))) for _ in range(q)]
    ic = [draw(st.integers(0, nchrom - 1)) for _ in range(n)]
    qc = [draw(st.integers(0, nchrom)) for _ in range(q)]  # nchrom itself = a chromosome the index has never seen
    return (np.array(iv, dtype=np.uint32).reshape(-1, 2), np.array(ic, dtype=np.uint32),
            np.array(qs, dtype=np.uint32).reshape(-1, 2), np.array(qc, dtype=np.uint32), nchrom)


@settings(max_examples=int(__import__("os").environ.get("BIVX_HYPOTHESIS_EXAMPLES", "300")), deadline=None,
          suppress_health_check=list(HealthCheck), derandomize=__import__("os").environ.get("BIVX_HYPOTHESIS_RANDOM") is None)
@given(problem())
def test_all_entry_points_equal_brute_force(p):
    import torch
    from binary_amd import IntervalIndex
    iv, ic, qs, qc, nchrom = p
    low, high = iv[:, 0].copy(), iv[:, 1].copy()
    qlo, qhi = qs[:, 0].copy(), qs[:, 1].copy()
    use_chrom = nchrom > 1
    # brute force, int64 arithmetic
    L, H, QL, QH = (a.astype(np.int64) for a in (low, high, qlo, qhi))
    hit = (QL[:, None] <= H[None, :]) & (L[None, :] <= QH[:, None])
    if use_chrom:
        hit &= qc[:, None] == ic[None, :]
    exp = [np.nonzero(hit[i])[0] for i in range(qlo.size)]
    exp_off = np.concatenate([[0], np.cumsum([e.size for e in exp])]).astype(np.uint64)
    exp_hits = np.concatenate(exp).astype(np.uint32) if exp_off[-1] else np.zeros(0, np.uint32)
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high, ic if use_chrom else None)
        idx.build()
        qcc = qc if use_chrom else None
        off, hits = idx.find_overlaps(qlo, qhi, qcc, sort_by_id=True)           # two-pass + sort
        assert np.array_equal(off, exp_off) and np.array_equal(hits, exp_hits)
        off_u, hits_u = idx.find_overlaps(qlo, qhi, qcc, sort_by_id=False)       # index order: same sets
        assert np.array_equal(off_u, exp_off)
        for i in range(qlo.size):
            assert np.array_equal(np.sort(hits_u[int(off_u[i]):int(off_u[i + 1])]), exp[i])
        first = idx.find_overlap(qlo, qhi, qcc)                                   # existence + smallest id
        assert np.array_equal(first, np.array([e[0] if e.size else M32 for e in exp], dtype=np.uint32))
        d_off = torch.empty(qlo.size + 1, dtype=torch.int64, device=dev)          # single pass (+ sort)
        d_hits = torch.empty(max(int(exp_off[-1]), 1), dtype=torch.int32, device=dev)
        idx.query_device(to(qlo), to(qhi), d_off, d_hits, qchrom=to(qc) if use_chrom else None, sort_by_id=True)
        torch.cuda.synchronize()
        assert np.array_equal(d_off.cpu().numpy().astype(np.uint64), exp_off)
        assert np.array_equal(d_hits.cpu().numpy().view(np.uint32)[: int(exp_off[-1])], exp_hits)
        nq = qlo.size                                                              # unordered single pass
        beg = torch.empty(nq, dtype=torch.int64, device=dev)
        cnt = torch.empty(nq, dtype=torch.int32, device=dev)
        tot = torch.full((1,), -1, dtype=torch.int64, device=dev)
        idx.query_device_unordered(to(qlo), to(qhi), beg, cnt, d_hits, tot, qchrom=to(qc) if use_chrom else None)
        torch.cuda.synchronize()
        assert int(tot.item()) == int(exp_off[-1])
        b, c, hu = beg.cpu().numpy(), cnt.cpu().numpy(), d_hits.cpu().numpy().view(np.uint32)
        for i in range(nq):
            assert c[i] == exp[i].size and np.array_equal(np.sort(hu[b[i]:b[i] + c[i]]), exp[i])
