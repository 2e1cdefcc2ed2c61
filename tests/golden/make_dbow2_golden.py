#!/usr/bin/env python3
"""Writes tests/golden/dbow2_ref_vectors.npz: BowVector / FeatureVector outputs of the REFERENCE's own compiled classes
(oracle/_ref/libdbow2_ref.so = /root/reference/Thirdparty/DBoW2/DBoW2/{BowVector,FeatureVector}.cpp + our driver, built by
`make -C oracle ref` in the build container) for seeded synthetic inputs.  The inputs are regenerated from the seeds by
multi_orb_slam_amd.synth at test time; the per-feature (word id, weight, node id) fed to the reference classes come from the
oracle's descent (the descent itself cannot be built from the reference here), and are stored too.
Run from the repo root:  python tests/golden/make_dbow2_golden.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from multi_orb_slam_amd import synth  # noqa: E402

CASES = [dict(k=10, L=3, seed=11, ragged=False, stop_every=0, n=1500, fseed=3, levelsup=2),
         dict(k=6, L=4, seed=5, ragged=True, stop_every=7, n=800, fseed=9, levelsup=2),
         dict(k=4, L=5, seed=23, ragged=False, stop_every=3, n=2500, fseed=1, levelsup=4),
         dict(k=10, L=2, seed=2, ragged=False, stop_every=0, n=40, fseed=4, levelsup=1),
         dict(k=3, L=3, seed=8, ragged=True, stop_every=2, n=300, fseed=6, levelsup=5)]


def ref_lib():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdbow2_ref.so"))
    L.ref_bow_build.restype = C.c_int
    return L


def ref_build(L, word, weight, node):
    n = len(word)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    bid, bval = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.float64)
    fn, fs, fi = np.zeros(max(n, 1), np.uint32), np.zeros(n + 1, np.int32), np.zeros(max(n, 1), np.uint32)
    nn = C.c_int()
    nw = L.ref_bow_build(p(word), p(weight), p(node), n, p(bid), p(bval), p(fn), p(fs), p(fi), C.byref(nn))
    return bid[:nw].copy(), bval[:nw].copy(), fn[:nn.value].copy(), fs[:nn.value + 1].copy(), fi[:fs[nn.value]].copy()


def case_inputs(c):
    voc = synth.vocabulary(k=c["k"], L=c["L"], seed=c["seed"], ragged=c["ragged"], stop_every=c["stop_every"])
    feats = synth.vocabulary_words(voc, c["n"], seed=c["fseed"], flip_p=0.08)
    return voc, feats


def main():
    L = ref_lib()
    out = {"n_cases": np.int32(len(CASES))}
    for i, c in enumerate(CASES):
        voc, feats = case_inputs(c)
        word, node, weight = oracle.Vocabulary(voc).transform(feats, c["levelsup"])
        bid, bval, fn, fs, fi = ref_build(L, np.ascontiguousarray(word), np.ascontiguousarray(weight), np.ascontiguousarray(node))
        for k, v in c.items():
            out["c%d_%s" % (i, k)] = np.int64(v)
        out.update({"c%d_word" % i: word, "c%d_node" % i: node, "c%d_weight" % i: weight, "c%d_bow_id" % i: bid, "c%d_bow_val" % i: bval,
                    "c%d_fv_node" % i: fn, "c%d_fv_start" % i: fs, "c%d_fv_items" % i: fi})
        print("case %d: %d features -> %d words, %d nodes" % (i, c["n"], len(bid), len(fn)))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dbow2_ref_vectors.npz"), **out)


if __name__ == "__main__":
    main()
