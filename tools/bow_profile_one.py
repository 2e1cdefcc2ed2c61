#!/usr/bin/env python3
"""One size of the BoW join for rocprofv3 (kernel times of k_bow_init / k_bow_join / k_bow_finish)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth
import helpers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
voc = synth.vocabulary(10, 6, seed=3)
V = m.Vocabulary(voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"], voc["L"])
S = m.BowSearch()


class PV:   # bow_vectors through the product (no oracle in this script)
    def bow_vectors(self, f, levelsup):
        (b, fv) = V.bow_vectors(f, levelsup)
        return b, (fv.node_id, fv.node_start, fv.items)


a, b = helpers.make_bow_pair(voc, PV(), n, n, seed=7, levelsup=4)
fv = lambda s: m.FeatureVector(s["node_id"], s["node_start"], s["items"])
KA = S.keyframe(m.BowSide(a["desc"], a["angle"], fv(a), a["flags"])); KB = S.keyframe(m.BowSide(b["desc"], b["angle"], fv(b), b["flags"]))
for _ in range(10):
    nm, _m = S.search_by_bow_resident(KA, KB, 1, a["flags"], b["flags"])
print("matches", nm, "largest node", int(np.diff(b["node_start"]).max()), flush=True)
KA.close(); KB.close(); S.close(); V.close()
