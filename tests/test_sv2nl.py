"""sv2nl "next" rows: f1 the VCF record reader, f2 the command line tool, f3 the device-side check_condition.

Reader pins are the reference's own (test/source/test_parser/test_vcf.cpp:94-100,163,173 on its fixtures, copied as
data under tests/golden/vcf/). Tool outputs are compared, as sorted line sets, with TSVs derived from the CPU
restatement (oracle/sv2nl_oracle.py): sv2nl-level parity against the reference itself is UNPINNED — the reference
holds neither a delly-style fixture nor an expected output."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VCF = os.path.join(ROOT, "tests", "golden", "vcf")
SVDIR = os.path.join(ROOT, "binary_amd", "sv2nl")


@pytest.fixture(scope="module")
def tools():
    from binary_amd import _build
    _build.build_lib()
    subprocess.run(["make", "-C", SVDIR, "-s", "all"], check=True)
    return os.path.join(SVDIR, "sv2nl"), os.path.join(SVDIR, "vcf_dump")


def dump(tools, path, source):
    out = subprocess.run([tools[1], path, source], capture_output=True, text=True, check=True).stdout.splitlines()
    contigs = [l[2:] for l in out if l.startswith("C ")]
    recs = [l[2:].split(" ") for l in out if l.startswith("R ")]
    err = [l[2:] for l in out if l.startswith("E ")]
    return contigs, recs, (err[0] if err else None)


# ---- f1: reader -------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", ["debug_uncom.vcf", "debug.vcf.gz"])
def test_reader_reference_fixture_pins(tools, name):
    contigs, recs, err = dump(tools, os.path.join(VCF, name), "nls")
    assert err is None
    assert len(contigs) == 455 and contigs[0] == "chr1" and contigs[1] == "chr10"      # header order (vcf.hpp:577-589)
    assert sum("_" not in c for c in contigs) == 25
    assert len(recs) == 6                                                                 # test_vcf.cpp:163
    assert recs[0][:3] == ["chr10", str(93567288 - 1), "TRA"]                             # test_vcf.cpp:94-100 (0-based)
    assert recs[0][3] == "7705262" and recs[0][4] == "chr17"
    assert sum(r[2] == "TRA" for r in recs) == 2                                          # test_vcf.cpp:173
    assert [r[2] for r in recs] == ["TRA", "TRA", "INS", "TDUP", "TDUP", "TDUP"]


def test_reader_matches_python_restatement(tools, oracle):
    from oracle import sv2nl_oracle
    for name, source in (("pair_sv.vcf", "delly"), ("pair_nl.vcf", "nls"), ("debug_uncom.vcf", "nls")):
        path = os.path.join(VCF, name)
        contigs, recs, err = dump(tools, path, source)
        ocontigs, orecs, oerr = sv2nl_oracle.read_vcf(path, source)
        assert contigs == ocontigs and err == oerr and len(recs) == len(orecs)
        for r, o in zip(recs, orecs):
            assert r == [o.chrom, str(o.pos), o.svtype, str(o.svend), o.chr2 or ".", str(int(o.strand1)), str(int(o.strand2))]


def test_reader_error_behaviour(tools):
    # the reference fixture read as a delly file: the first record has no END -> "Failed to get info END"
    # (vcf_info.cpp:39-41); reading stops there, like every chromosome task of the reference does
    _, recs, err = dump(tools, os.path.join(VCF, "debug_uncom.vcf"), "delly")
    assert recs == [] and err == "Failed to get info END"


# ---- f2: command line surface (no GPU needed for these) -----------------------------------------------------------

def test_cli_help_and_argument_errors(tools, tmp_path):
    r = subprocess.run([tools[0], "-h"], capture_output=True, text=True)
    assert r.returncode == 0 and "--dis" in r.stdout and "-s, --short" in r.stdout and "-m, --merge" in r.stdout
    r = subprocess.run([tools[0]], capture_output=True, text=True)
    assert r.returncode == 1 and "has no value" in r.stderr          # cxxopts::option_has_no_value_exception path
    r = subprocess.run([tools[0], "nope.vcf", "nope2.vcf"], capture_output=True, text=True)
    assert r.returncode == 1                                           # check_file_path -> exit(1)


# ---- f2 + f3 on the GPU -------------------------------------------------------------------------------------------

def run_tool(tools, tmp_path, extra=(), merge=False):
    out = str(tmp_path / "out.tsv")
    cmd = [tools[0], os.path.join(VCF, "pair_sv.vcf"), os.path.join(VCF, "pair_nl.vcf"), "-o", out, *extra]
    if merge:
        cmd.append("-m")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return out


def lines_of(path):
    with open(path) as f:
        ls = f.read().splitlines()
    assert ls[0] == "chrom\tpos\tend\tsvtype\tchrom\tpos\tend\tsvtype"
    return sorted(ls[1:])


@pytest.mark.gpu
@pytest.mark.parametrize("extra,tag", [((), ""), (("--dis", "50000", "-s"), "_short_dis50000")])
@pytest.mark.parametrize("host_filter", [False, True])
@pytest.mark.parametrize("per_mapper", [False, True])
def test_tool_outputs_match_restatement(tools, tmp_path, extra, tag, host_filter, per_mapper):
    """Default: ONE device index with an svtype column for the three mappers; --index-per-mapper:
    round 1's three host-filtered indexes. Both must give the restatement's lines."""
    out = run_tool(tools, tmp_path, extra + (("--host-filter",) if host_filter else ())
                   + (("--index-per-mapper",) if per_mapper else ()))
    for k in ("dup", "inv", "tra"):
        assert lines_of(out + "." + k) == lines_of(os.path.join(VCF, f"pair_expected{tag}.{k}.tsv")), k


@pytest.mark.gpu
def test_tool_merge(tools, tmp_path):
    out = run_tool(tools, tmp_path, merge=True)
    assert not any(os.path.exists(out + e) for e in (".dup", ".inv", ".tra"))   # parts are deleted (utils.hpp:60-63)
    exp = sorted(sum((lines_of(os.path.join(VCF, f"pair_expected.{k}.tsv")) for k in ("dup", "inv", "tra")), []))
    assert lines_of(out) == exp


@pytest.mark.gpu
def test_tool_sv_file_error(tools, tmp_path):
    out = str(tmp_path / "o.tsv")
    r = subprocess.run([tools[0], os.path.join(VCF, "debug_uncom.vcf"), os.path.join(VCF, "pair_nl.vcf"), "-o", out],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "Failed to get info END" in r.stderr
    assert lines_of(out + ".dup") == [] and lines_of(out + ".inv") == []          # header-only parts


@pytest.mark.gpu
def test_device_filters_equal_host_predicates_on_random_data():
    """f3: bivx_*_f with each filter kind == unfiltered hits post-filtered by the reference predicates."""
    import ctypes as C
    import numpy as np
    from binary_amd import IntervalIndex, capi
    rng = np.random.default_rng(17)
    n, q, d = 40000, 30000, 3000
    low = rng.integers(0, 2_000_000, n).astype(np.uint32)
    high = low + rng.integers(0, 6000, n).astype(np.uint32)
    longer = rng.random(n) < 0.03                                # a few long SVs: several length classes per index
    high[longer] = low[longer] + rng.integers(100_000, 900_000, int(longer.sum())).astype(np.uint32)
    inv = rng.random(n) < 0.1                                    # TRA trees hold low > high records too
    qlo = rng.integers(0, 2_000_000, q).astype(np.uint32)
    qhi = qlo + rng.integers(0, 6000, q).astype(np.uint32)
    iaux = ((rng.integers(0, 6, n) << 1) | rng.integers(0, 2, n)).astype(np.uint32)
    qaux_t = ((rng.integers(0, 6, q) << 1) | rng.integers(0, 2, q)).astype(np.uint32)
    qaux_s = rng.integers(0, 4, q).astype(np.uint32)
    ad = lambda a, b: np.where(a >= b, a - b, b - a)
    for kind in (capi.FILTER_SV2NL_DUP, capi.FILTER_SV2NL_INV, capi.FILTER_SV2NL_TRA):
        lo_i = np.where(inv & (kind == capi.FILTER_SV2NL_TRA), high, low).astype(np.uint32)
        hi_i = np.where(inv & (kind == capi.FILTER_SV2NL_TRA), low, high).astype(np.uint32)
        with IntervalIndex(0) as idx:
            idx.insert_node(lo_i, hi_i)
            idx.build()
            off0, hits0 = idx.find_overlaps(qlo, qhi)
            L = capi.load()
            qaux = qaux_t if kind == capi.FILTER_SV2NL_TRA else qaux_s
            flt = capi.Filter(kind, d, 1, 0, qaux.ctypes.data, iaux.ctypes.data)
            off = np.zeros(q + 1, np.uint64)
            p = lambda a: a.ctypes.data_as(C.c_void_p)
            capi.check(L.bivx_count_f(idx._h, None, p(qlo), p(qhi), q, C.byref(flt), p(off)))
            hits = np.empty(int(off[-1]), np.uint32)
            capi.check(L.bivx_fill_f(idx._h, None, p(qlo), p(qhi), q, C.byref(flt), p(off), p(hits), 1))
            # the single-pass kernel with the same filter (device pointers): same CSR, index order -> compare sorted
            import torch
            dev = torch.device("cuda:0")
            to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
            t_qaux, t_iaux = to(qaux), to(iaux)
            dflt = IntervalIndex.device_filter(kind, d, True, t_qaux, t_iaux)
            d_off = torch.empty(q + 1, dtype=torch.int64, device=dev)
            d_hits = torch.empty(max(int(off[-1]), 1), dtype=torch.int32, device=dev)
            idx.query_device(to(qlo), to(qhi), d_off, d_hits, flt=dflt, sort_by_id=True)
            torch.cuda.synchronize()
            assert np.array_equal(d_off.cpu().numpy().astype(np.uint64), off), kind
            assert np.array_equal(d_hits.cpu().numpy().view(np.uint32)[: int(off[-1])], hits), kind
            assert idx.stats()["n_segments"] >= 2
            # and the unordered single pass with the filter: the same sets per query
            beg = torch.empty(q, dtype=torch.int64, device=dev)
            cnt = torch.empty(q, dtype=torch.int32, device=dev)
            tot = torch.zeros(1, dtype=torch.int64, device=dev)
            idx.query_device_unordered(to(qlo), to(qhi), beg, cnt, d_hits, tot, flt=dflt)
            torch.cuda.synchronize()
            assert int(tot.item()) == int(off[-1]), kind
            b, c, hu = beg.cpu().numpy(), cnt.cpu().numpy().astype(np.int64), d_hits.cpu().numpy().view(np.uint32)
            assert np.array_equal(c, np.diff(off.astype(np.int64))), kind
            gather = np.repeat(b - np.cumsum(c) + c, c) + np.arange(int(c.sum()))
            got = hu[gather]
            order = np.lexsort((got, np.repeat(np.arange(q), c)))
            assert np.array_equal(got[order], hits), kind
        qid = np.repeat(np.arange(q), np.diff(off0.astype(np.int64)))
        a_lo, a_hi = qlo[qid].astype(np.int64), qhi[qid].astype(np.int64)
        b_lo, b_hi = lo_i[hits0].astype(np.int64), hi_i[hits0].astype(np.int64)
        if kind == capi.FILTER_SV2NL_DUP:
            keep = (b_lo <= a_lo) & (b_hi >= a_hi) & (ad(a_lo, b_lo) <= d) & (ad(a_hi, b_hi) <= d)
        elif kind == capi.FILTER_SV2NL_INV:
            c1 = (b_lo <= a_lo) & (b_hi >= a_hi)
            c2 = (a_lo <= b_lo) & (a_hi >= b_hi)
            near = (ad(a_lo, b_lo) <= d) & (ad(a_hi, b_hi) <= d)
            s1, s2 = (qaux_s[qid] & 1) != 0, (qaux_s[qid] & 2) != 0
            keep = ~c1 & ~c2 & near & np.where(a_lo <= b_lo, s1 & ~s2, ~s1 & s2)
        else:
            qa, ia = qaux_t[qid], iaux[hits0]
            q1, q2 = np.where(qa & 1, a_hi, a_lo), np.where(qa & 1, a_lo, a_hi)
            i1, i2 = np.where(ia & 1, b_hi, b_lo), np.where(ia & 1, b_lo, b_hi)
            keep = ((qa >> 1) == (ia >> 1)) & (ad(q1, i1) <= d) & (ad(q2, i2) <= d)
        exp_cnt = np.bincount(qid[keep], minlength=q)
        assert np.array_equal(np.diff(off.astype(np.int64)), exp_cnt), kind
        assert np.array_equal(hits, hits0[keep]), kind
        assert 0 < keep.sum() < keep.size
