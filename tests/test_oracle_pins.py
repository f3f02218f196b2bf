"""Pins the CPU oracle (oracle/ivtree.c) to the reference's own known answers.

Every expected value comes from tests/golden/reference_known_answers.json, i.e. from the assertions in
the reference's test_interval_tree.cpp / test_rb_tree.cpp and from the reference outputs recorded in
SURVEY.md §8c. Runs on CPU (no GPU marker).
"""
import itertools
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))
RT = GOLD["reference_tests"]
FX = GOLD["fixture10"]


def key_tree(oracle, keys):
    """RbTree<IntNode> in the reference == interval tree whose interval is [k, k] as far as shape goes."""
    return oracle.OracleTree(keys, keys)


def test_fixture10_shape(oracle):
    t = oracle.OracleTree(FX["low"], FX["high"])
    assert t.size() == RT["fixture10_size"]["value"]
    assert t.node(t.root())["low"] == RT["fixture10_root_key"]["value"]
    assert t.black_height() >= 0
    assert t.check_max()


def test_fixture10_tree_dump_matches_survey(oracle):
    t = oracle.OracleTree(FX["low"], FX["high"])
    dump = [[t.node(i)["low"], t.node(i)["high"], t.node(i)["max"], "R" if t.node(i)["red"] else "B"]
            for i in t.preorder()]
    assert dump == GOLD["survey_8c"]["tree_dump_preorder"]


def test_seq500_balanced(oracle):
    s = RT["seq500"]
    low = np.arange(s["start"], s["stop"], s["step"], dtype=np.uint32)
    t = oracle.OracleTree(low, low + s["len"])
    assert t.size() == s["size"]
    assert t.black_height() >= 0
    assert t.check_max()


def test_find_overlap_single(oracle):
    t = oracle.OracleTree(FX["low"], FX["high"])
    h = t.find_overlap(*RT["find_overlap_22_25"]["q"])
    assert h >= 0
    assert [t.node(h)["low"], t.node(h)["high"]] == RT["find_overlap_22_25"]["hit"]
    assert t.find_overlap(*RT["find_overlap_100_111"]["q"]) == -1


def test_find_overlaps_counts_and_order(oracle):
    t = oracle.OracleTree(FX["low"], FX["high"])
    for name, ordered in (("find_overlaps_7_25", "find_overlaps_7_25_preorder"),
                          ("find_overlaps_15_25", "find_overlaps_15_25_preorder")):
        q = RT[name + "_count"]["q"]
        hits = t.find_overlaps(*q)
        assert len(hits) == RT[name + "_count"]["count"]
        got = [[FX["low"][i], FX["high"][i]] for i in hits]
        assert got == GOLD["survey_8c"][ordered]
        # and as a set it is the brute-force predicate
        assert sorted(hits.tolist()) == oracle.brute_overlaps(FX["low"], FX["high"], *q).tolist()


def test_duplicates_kept(oracle):
    d = RT["dup4"]
    t = oracle.OracleTree(d["low"], d["high"])
    assert t.size() == d["size"]
    assert len(t.find_overlaps(*d["q"])) == d["count"]


@pytest.mark.parametrize("name", ["rb_keys_1_2_4", "rb_keys_1_2_7_4_10", "rb_keys21", "rb_keys20"])
def test_rb_roots(oracle, name):
    c = RT[name]
    t = key_tree(oracle, c["keys"])
    assert t.size() == c["size"]
    if "root_key" in c:
        assert t.node(t.root())["low"] == c["root_key"]
    if "black_height" in c:
        assert t.black_height() == c["black_height"]
    assert t.black_height() >= 0
    if "search" in c:
        assert t.node(t.search(c["search"]))["low"] == c["search"]


def test_rb_succ_pred(oracle):
    c = RT["rb_succ_pred"]
    t = key_tree(oracle, c["keys"])
    r = t.root()
    assert t.node(t.successor(r))["low"] == c["succ_of_root"]
    assert t.node(t.predecessor(r))["low"] == c["pred_of_root"]
    assert t.node(t.node(r)["left"])["parent"] == r and t.node(t.node(r)["right"])["parent"] == r


def test_rb_permutations_balanced(oracle):
    # test_rb_tree.cpp:175-186: 42 successive permutations of the 21-key array stay balanced
    keys = sorted(RT["rb_keys21"]["keys"])
    for _, perm in zip(range(42), itertools.permutations(keys)):
        t = key_tree(oracle, list(perm))
        assert t.size() == 21 and t.black_height() >= 0


def test_rb_random_balanced(oracle):
    # test_rb_tree.cpp:188-207 (the reference draws from an unseeded random_device; any seed will do)
    rng = np.random.default_rng(7)
    for _ in range(10):
        keys = rng.integers(1, 100001, size=5000).astype(np.uint32)
        t = key_tree(oracle, keys)
        assert t.size() == keys.size and t.black_height() >= 0


def test_rb_delete_root(oracle):
    keys = RT["rb_keys21"]["keys"]
    t = key_tree(oracle, keys)
    t.delete(t.root())
    t.delete(t.root())
    assert t.size() == RT["rb_delete_root_twice"]["size_after"]
    for _ in range(19):
        assert t.black_height() >= 0
        t.delete(t.root())
    assert t.size() == 0
    # the double-free regression key set (test_rb_tree.cpp:285-299)
    t = key_tree(oracle, RT["rb_keys20"]["keys"])
    for _ in range(20):
        t.delete(t.root())
    assert t.size() == 0


def test_tree_equals_brute_force_incl_invalid_intervals(oracle):
    """The tree's two prunes are exact even with low > high nodes (TraMapper inserts unvalidated BND
    records, mapper.cpp:158-170); SURVEY §7 'Unvalidated intervals'."""
    rng = np.random.default_rng(11)
    for trial in range(60):
        n = int(rng.integers(1, 200))
        low = rng.integers(0, 500, size=n).astype(np.uint32)
        high = low + rng.integers(0, 60, size=n).astype(np.uint32)
        swap = rng.random(n) < 0.2
        low2 = np.where(swap, high, low).astype(np.uint32)
        high2 = np.where(swap, low, high).astype(np.uint32)
        t = oracle.OracleTree(low2, high2)
        assert t.check_max() and t.black_height() >= 0
        for _ in range(40):
            a, b = sorted(rng.integers(0, 600, size=2).tolist())
            got = sorted(t.find_overlaps(a, b).tolist())
            assert got == oracle.brute_overlaps(low2, high2, a, b).tolist()


def test_u32_extremes_and_empty(oracle):
    t = oracle.OracleTree()
    assert t.size() == 0 and t.find_overlap(0, 10) == -1 and len(t.find_overlaps(0, 0xFFFFFFFF)) == 0
    M = 0xFFFFFFFF
    low = [0, 0, M, M - 1, 5]
    high = [0, M, M, M, 5]
    t = oracle.OracleTree(low, high)
    assert sorted(t.find_overlaps(0, 0).tolist()) == [0, 1]
    assert sorted(t.find_overlaps(M, M).tolist()) == [1, 2, 3]
    assert sorted(t.find_overlaps(0, M).tolist()) == [0, 1, 2, 3, 4]
    assert sorted(t.find_overlaps(6, M - 2).tolist()) == [1]


def test_batch_matches_single(oracle):
    rng = np.random.default_rng(3)
    low = rng.integers(0, 100000, size=5000).astype(np.uint32)
    high = low + rng.integers(0, 300, size=5000).astype(np.uint32)
    t = oracle.OracleTree(low, high)
    qlo = rng.integers(0, 100000, size=700).astype(np.uint32)
    qhi = qlo + rng.integers(0, 300, size=700).astype(np.uint32)
    off1, hits1 = t.find_overlaps_batch(qlo, qhi, nthreads=1)
    off4, hits4 = t.find_overlaps_batch(qlo, qhi, nthreads=4)
    assert np.array_equal(off1, off4) and np.array_equal(hits1, hits4)
    for i in (0, 1, 350, 699):
        assert np.array_equal(hits1[int(off1[i]):int(off1[i + 1])], t.find_overlaps(int(qlo[i]), int(qhi[i])))
