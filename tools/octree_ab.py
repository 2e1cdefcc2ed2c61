#!/usr/bin/env python3
"""Quadtree stage A/B (round 6): GPU time of the DistributeOctTree stage (stage events around k_octree / k_octree_reg) per input,
keys in registers (default) against keys in LDS (MORB_OCT_REG=0) -- run once per form:
    python tools/octree_ab.py            MORB_OCT_REG=0 python tools/octree_ab.py
Inputs: the synthetic rig of configs[1] / [2] / [4] and the photograph rigs of tests/natural.py (1080p: 15 000 - 33 000 candidates on
the first levels, more than the LDS form holds)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth
import natural

form = "LDS keys (k_octree)" if os.environ.get("MORB_OCT_REG") == "0" else "register keys (k_octree_reg)"
print("quadtree form:", form)
cases = [("synthetic", 640, 480, 1000, 2), ("synthetic", 1280, 720, 2000, 2), ("synthetic", 1920, 1080, 4000, 8),
         ("china", 640, 480, 1000, 2), ("china", 1280, 720, 2000, 2), ("china", 1920, 1080, 4000, 2), ("hopper", 1920, 1080, 4000, 2),
         ("flower", 1920, 1080, 4000, 2)]
for name, w, h, nf, nc in cases:
    ex = m.Extractor([m.ExtractorParams(nfeatures=nf)] * nc, w, h)
    imgs = [synth.image(c, 0, w, h) for c in range(nc)] if name == "synthetic" else natural.rig(name, 0, w, h, n_cams=nc)
    ex.set_profiling(True)
    ts = []
    for it in range(12):
        ex.extract(imgs)
        ts.append(ex.stage_times_us())
    med = {k: float(np.median([t[k] for t in ts[2:]])) for k in ts[0]}
    cand = max(len(ex.debug_candidates(0, 0)), len(ex.debug_candidates(nc - 1, 0)))
    print("%-9s %4dx%-4d x%d  level-0 candidates %6d  path %d  quadtree %7.1f us  fast %6.1f  describe %6.1f  pyramid %6.1f"
          % (name, w, h, nc, cand, ex.last_path(), med["quadtree"], med["fast_cells"], med["describe"], med["pyramid"]), flush=True)
    ex.close()
