"""The BowVector / FeatureVector half of the oracle's BoW row against the REFERENCE's own compiled classes.

tests/golden/dbow2_ref_vectors.npz was produced by oracle/_ref/libdbow2_ref.so (the reference's BowVector.cpp and
FeatureVector.cpp compiled where they lie, tests/golden/make_dbow2_golden.py); when that library is present (build
container, and the GPU box: it travels with the snapshot) fresh cases are compared live as well."""
import os
import sys

import numpy as np
import pytest

import oracle
from multi_orb_slam_amd import synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_dbow2_golden as gold  # noqa: E402  (case list + input construction; loads nothing of the reference by itself)

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dbow2_ref_vectors.npz"))
REF_SO = os.path.join(gold.ROOT, "oracle", "_ref", "libdbow2_ref.so")


def _same(got, bid, bval, fn, fs, fi):
    (gid, gval), (gn, gs, gi) = got
    assert np.array_equal(gid, bid) and np.array_equal(gn, fn) and np.array_equal(gs, fs) and np.array_equal(gi, fi)
    assert gval.dtype == np.float64 and np.array_equal(gval, bval)  # bit-identical doubles: same accumulation and division order


@pytest.mark.parametrize("i", range(int(GOLD["n_cases"])))
def test_oracle_matches_reference_classes_golden(i):
    c = gold.CASES[i]
    assert all(int(GOLD["c%d_%s" % (i, k)]) == int(v) for k, v in c.items())
    voc, feats = gold.case_inputs(c)
    V = oracle.Vocabulary(voc)
    word, node, weight = V.transform(feats, c["levelsup"])
    assert np.array_equal(word, GOLD["c%d_word" % i]) and np.array_equal(node, GOLD["c%d_node" % i])
    assert np.array_equal(weight, GOLD["c%d_weight" % i])
    _same(V.bow_vectors(feats, c["levelsup"]), *[GOLD["c%d_%s" % (i, k)] for k in ("bow_id", "bow_val", "fv_node", "fv_start", "fv_items")])
    # the L1 norm the reference's normalize() produced
    assert abs(GOLD["c%d_bow_val" % i].sum() - 1.0) < 1e-12


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libdbow2_ref.so not built (make -C oracle ref needs /root/reference)")
@pytest.mark.parametrize("seed", [101, 102, 103, 104])
def test_oracle_matches_reference_classes_live(seed):
    L = gold.ref_lib()
    voc = synth.vocabulary(k=3 + seed % 7, L=3 + seed % 2, seed=seed, ragged=bool(seed & 1), stop_every=(0, 5, 2, 9)[seed % 4])
    feats = synth.vocabulary_words(voc, 700 + 100 * (seed % 5), seed=seed, flip_p=0.1)
    V = oracle.Vocabulary(voc)
    for levelsup in (0, 1, 2, 6):
        word, node, weight = V.transform(feats, levelsup)
        ref = gold.ref_build(L, np.ascontiguousarray(word), np.ascontiguousarray(weight), np.ascontiguousarray(node))
        _same(V.bow_vectors(feats, levelsup), *ref)
