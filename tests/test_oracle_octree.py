"""Quadtree distribution: the oracle's literal std::list restatement vs the library's array-based host code, plus
structural properties that follow from reference src/ORBextractor.cc:540-764.  CPU only."""
import numpy as np
import pytest
import oracle
from multi_orb_slam_amd import synth
from multi_orb_slam_amd.extractor import distribute_octree


def _cands(n, w, h, seed, max_resp=200):
    r = synth.hash32(np.arange(3 * n, dtype=np.uint64) + np.uint64(seed * 1315423911 & 0xFFFFFFFF)).astype(np.int64)
    k = np.zeros(n, oracle.KP_DTYPE)
    k["x"] = 3 + r[0::3] % (w - 6); k["y"] = 3 + r[1::3] % (h - 6); k["response"] = 7 + r[2::3] % max_resp
    k["size"] = 7; k["angle"] = -1; k["class_id"] = -1
    # FAST emits unique positions; drop duplicates keeping order
    _, first = np.unique(k["y"].astype(np.int64) * 10000 + k["x"].astype(np.int64), return_index=True)
    return k[np.sort(first)]


@pytest.mark.parametrize("w,h", [(608, 448), (501, 368), (147, 102), (1888, 1048), (300, 300), (200, 500)])
@pytest.mark.parametrize("n", [0, 1, 2, 17, 500, 5000])
def test_library_octree_equals_oracle(w, h, n):
    c = _cands(n, w, h, n + w) if n else np.zeros(0, oracle.KP_DTYPE)
    for N in (1, 31, 60, 217, 869, 3000):
        if h > w and round(w / h) == 0:
            continue  # nIni = 0: division by zero in the reference itself
        a = oracle.distribute_octree(c, 16, 16 + w, 16, 16 + h, N)
        b = distribute_octree(c, 16, 16 + w, 16, 16 + h, N)
        assert np.array_equal(a, b), (w, h, n, N)


def test_octree_properties():
    w, h = 608, 448
    c = _cands(4000, w, h, 3)
    for N in (60, 217, 869):
        out = oracle.distribute_octree(c, 16, 16 + w, 16, 16 + h, N)
        assert N <= len(out) <= N + 3                      # careful phase can overshoot by at most 3 (one split)
        pos = set(zip(out["x"].tolist(), out["y"].tolist()))
        assert len(pos) == len(out)                        # one keypoint per node, nodes are disjoint
        allpos = set(zip(c["x"].tolist(), c["y"].tolist()))
        assert pos <= allpos
    few = _cands(40, w, h, 5)
    out = oracle.distribute_octree(few, 16, 16 + w, 16, 16 + h, 217)
    assert len(out) == len(few)                            # fewer candidates than quota: everyone survives
    assert sorted(zip(out["x"].tolist(), out["y"].tolist())) == sorted(zip(few["x"].tolist(), few["y"].tolist()))


def test_octree_equal_responses_first_candidate_wins_and_list_order():
    k = np.zeros(4, oracle.KP_DTYPE)
    k["x"] = [10, 12, 500, 502]; k["y"] = [10, 12, 400, 402]; k["response"] = [50, 50, 30, 60]
    out = oracle.distribute_octree(k, 16, 16 + 608, 16, 16 + 448, 2)
    # the root splits once into UL {0,1} and BR {2,3}; children are pushed to the list FRONT, so BR comes first;
    # inside UL the responses tie and the first candidate wins (strict '>', reference src/ORBextractor.cc:753)
    assert list(zip(out["x"].tolist(), out["y"].tolist())) == [(502.0, 402.0), (10.0, 10.0)]
    assert np.array_equal(out, distribute_octree(k, 16, 16 + 608, 16, 16 + 448, 2))


def test_octree_stops_when_a_pass_does_not_grow_the_list():
    # all four points fall into the same child: the node count stays 1 == prevSize and the loop finishes with ONE
    # keypoint although N = 2 (reference src/ORBextractor.cc:667-671)
    k = np.zeros(4, oracle.KP_DTYPE)
    k["x"] = [10, 12, 300, 302]; k["y"] = [10, 12, 200, 202]; k["response"] = [50, 50, 30, 60]
    out = oracle.distribute_octree(k, 16, 16 + 608, 16, 16 + 448, 2)
    assert len(out) == 1 and out["response"][0] == 60
    assert np.array_equal(out, distribute_octree(k, 16, 16 + 608, 16, 16 + 448, 2))


def test_octree_refuses_regions_the_reference_leaves_undefined():
    """nIni = round(width / height) root nodes (reference src/ORBextractor.cc:544): 0 for a region more than twice as high as it is
    wide -- the reference then divides by zero and indexes an empty vector.  The oracle restates nothing there: it raises (the
    product defines this case as ONE root, DESIGN.md section 5, without a reference to hold it against)."""
    k = np.zeros(4, oracle.KP_DTYPE)
    k["x"] = [10, 12, 50, 52]; k["y"] = [10, 12, 400, 402]; k["response"] = [50, 50, 30, 60]
    with pytest.raises(ValueError):
        oracle.distribute_octree(k, 16, 16 + 100, 16, 16 + 448, 2)
    from multi_orb_slam_amd import synth
    with pytest.raises(ValueError):
        oracle.extract(synth.image(1, 0, 200, 600), nfeatures=100)
