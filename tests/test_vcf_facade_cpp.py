"""the boundary TYPE of the path — BaseVcfInterval<Record>, VcfInterval, VcfIntervalNode and sv2nl's
Sv2nl* aliases (include/binary/parser/vcf.hpp, binary_amd/sv2nl/vcf_info.hpp) — against the reference's own VCF -> tree
tests (test/source/test_parser/test_vcf.cpp:47-187), re-expressed in tests/cpp/test_vcf_facade.cpp on the reference's
own data files (tests/golden/vcf/debug*.vcf*).

CPU: compiles with -Wall -Wextra -Werror and runs the parser / node cases (chroms, first record chr10 / TRA /
93567287, ranges and views, node from record and from (end, start, record)). GPU: the tree cases (insert_node(range)
-> 6, filtered TRA view -> 2, range-for inserts -> 6) and find_overlaps on that 6-record tree — which holds three
low > high nodes (POS > SVEND records) — against the tree oracle: same hits in the same pre-order in
HitOrder::ReferencePreorder, same set in the default order."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURES = os.path.join(ROOT, "tests", "golden", "vcf")

# (pos, svend) of the six records of debug.vcf.gz, in file order — pinned by binary_amd/sv2nl/vcf_dump in test_sv2nl.py
RECORDS = [(93567287, 7705262, "chr10", "TRA"), (93567288, 7705202, "chr10", "TRA"), (93567288, 93567289, "chr10", "INS"),
           (29927247, 29929790, "chr14", "TDUP"), (7708249, 7701656, "chr17", "TDUP"), (7708249, 7701656, "chr17", "TDUP")]


@pytest.fixture(scope="module")
def vcf_facade_binary(tmp_path_factory):
    from binary_amd import _build
    _build.build_lib()
    out = str(tmp_path_factory.mktemp("vcf_facade") / "test_vcf_facade")
    libdir = os.path.join(ROOT, "binary_amd")
    cmd = ["g++", "-std=c++20", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "test_vcf_facade.cpp"), "-o", out, "-L", libdir, "-lbivx", "-lz", "-pthread",
           f"-Wl,-rpath,{libdir}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_vcf_boundary_types_compile_and_parser_cases_pass(vcf_facade_binary):
    r = subprocess.run([vcf_facade_binary, FIXTURES, "--no-gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 failed" in r.stdout


@pytest.mark.gpu
def test_vcf_tree_cases_on_gpu(vcf_facade_binary):
    r = subprocess.run([vcf_facade_binary, FIXTURES], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 failed" in r.stdout


@pytest.mark.gpu
def test_find_overlaps_on_the_six_record_tree_vs_oracle(vcf_facade_binary, tmp_path, oracle):
    low = np.array([r[0] for r in RECORDS], np.uint32)
    high = np.array([r[1] for r in RECORDS], np.uint32)
    t = oracle.OracleTree(low, high)      # sequential inserts in file order, as insert_node(vcf_ranges) does
    rng = np.random.default_rng(7)
    qs = [(7701000, 7709000), (93567288, 93567288), (93567289, 93567289), (29927000, 29928000), (0, 0xFFFFFFFF),
          (7705262, 93567287), (7701656, 7708249), (93567290, 93567300), (0, 0), (29929790, 29929790), (29929791, 4_000_000_000)]
    for _ in range(40):
        a = int(rng.choice(np.r_[low, high])) + int(rng.integers(-3, 4))
        b = int(rng.choice(np.r_[low, high])) + int(rng.integers(-3, 4))
        qs.append((max(a, 0), max(b, 0)))            # unordered pairs too: inverted queries are legal input
    qf = tmp_path / "queries.txt"
    qf.write_text("".join(f"{a} {b}\n" for a, b in qs))
    r = subprocess.run([vcf_facade_binary, FIXTURES, "--overlaps", str(qf)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.splitlines()
    assert len(lines) == len(qs)
    fmt = lambda i: f"{RECORDS[i][0]}-{RECORDS[i][1]}:{RECORDS[i][2]}:{RECORDS[i][3]}"
    nonempty = 0
    for (a, b), line in zip(qs, lines):
        pre, dflt = (x.split() for x in line.split("|"))
        exp = [int(i) for i in t.find_overlaps(a, b)]
        assert pre == [fmt(i) for i in exp], (a, b)                     # the reference's hits in the reference's order
        assert dflt == [fmt(i) for i in sorted(exp)], (a, b)            # default: ascending insertion index
        brute = [i for i in range(6) if low[i] <= b and a <= high[i]]   # and both equal the predicate itself
        assert sorted(exp) == brute
        nonempty += bool(exp)
    assert nonempty >= 10


@pytest.mark.gpu
def test_per_record_caller_written_against_the_dropin_types(tmp_path):
    """tests/cpp/dropin_dup_mapper.cpp: the reference's DUP mapper pattern (a tree per chromosome from a filter | transform
    view, one find_overlaps(record) per NL record, check_condition, the duplicate-key rule, the line format:
    mapper.hpp:147-162,194-236, mapper.cpp:50-55) written against nothing but the drop-in headers. Its lines equal the
    expected TSV of the authored pair fixture — the lines the oracle-derived golden file and the batched tool give."""
    from binary_amd import _build
    _build.build_lib()
    exe = str(tmp_path / "dropin_dup_mapper")
    libdir = os.path.join(ROOT, "binary_amd")
    r = subprocess.run(["g++", "-std=c++20", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "cpp", "dropin_dup_mapper.cpp"), "-o", exe, "-L", libdir, "-lbivx", "-lz",
                        "-pthread", f"-Wl,-rpath,{libdir}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for dis, expected in (("1000000", "pair_expected.dup.tsv"), ("50000", "pair_expected_short_dis50000.dup.tsv")):
        r = subprocess.run([exe, os.path.join(FIXTURES, "pair_sv.vcf"), os.path.join(FIXTURES, "pair_nl.vcf"), dis],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        got = sorted(l for l in r.stdout.splitlines()[1:] if l)
        exp = sorted(l for l in open(os.path.join(FIXTURES, expected)).read().splitlines()[1:] if l)
        assert got == exp and len(exp) > 20, (dis, len(got), len(exp))
