#!/usr/bin/env python3
"""Extractor against the oracle over odd parameter sets (thresholds, level counts, scale factors, image shapes): keypoint records and
descriptors must be bit-identical.  Exercises the packed FAST quick test at other thresholds and cell shapes, the pyramid at other
steps, the device quadtree at other quotas."""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth
import oracle

cases = [(640, 480, dict(nfeatures=1000, ini_th_fast=12, min_th_fast=3)),
         (640, 480, dict(nfeatures=500, ini_th_fast=40, min_th_fast=20)),
         (700, 333, dict(nfeatures=800, scale_factor=1.1, nlevels=10)),
         (512, 512, dict(nfeatures=1500, scale_factor=1.5, nlevels=4)),
         (1280, 720, dict(nfeatures=3000, nlevels=6, ini_th_fast=25, min_th_fast=5)),
         (777, 333, dict(nfeatures=600, scale_factor=1.3, nlevels=5, ini_th_fast=15, min_th_fast=7)),
         (1920, 1080, dict(nfeatures=4000)),
         (160, 120, dict(nfeatures=100, nlevels=3))]
import sys
for (w, h, kw) in cases:
    p = m.ExtractorParams(**kw)
    ex = m.Extractor([p, p], w, h)
    for t in range(2):
        imgs = [synth.image(7 + c, t, w, h) for c in range(2)]
        out = ex.extract(imgs)
        for c in range(2):
            ok, od = oracle.extract(imgs[c], nfeatures=p.nfeatures, scale_factor=p.scale_factor, nlevels=p.nlevels,
                                    ini_th=p.ini_th_fast, min_th=p.min_th_fast)
            k, d = out[c]
            assert k.tobytes() == ok.tobytes() and np.array_equal(d, od), (w, h, kw, t, c, len(k), len(ok))
    print("ok", w, h, kw, [len(o[0]) for o in out], "path", ex.last_path(), flush=True)
    ex.close()
# a level more than twice as high as wide: round(width / height) == 0 root nodes, undefined in the reference (the oracle refuses);
# the product takes ONE root there and must simply run
p = m.ExtractorParams(nfeatures=600)
ex = m.Extractor([p], 333, 777)
out = ex.extract([synth.image(3, 0, 333, 777)])
print("portrait 333 x 777:", len(out[0][0]), "keypoints, path", ex.last_path())
try:
    oracle.extract(synth.image(3, 0, 333, 777), nfeatures=600); raise SystemExit("the oracle should have refused")
except ValueError as e:
    print("oracle:", e)
ex.close()
print("ALL OK")
