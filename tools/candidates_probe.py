#!/usr/bin/env python3
"""Candidates per pyramid level (what the device quadtree has to hold, limit 4096) for the synthetic streams at several sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth
for (w, h, nf) in [(640, 480, 1000), (1280, 720, 2000), (1920, 1080, 4000)]:
    ex = m.Extractor([m.ExtractorParams(nfeatures=nf)], w, h)
    img = synth.image(0, 1, w, h)
    kps, desc = ex.extract([img])[0]
    print(w, h, nf, "kept", len(kps), "candidates per level", [len(ex.debug_candidates(0, l)) for l in range(8)], "stages", {k: round(v, 1) for k, v in ex.stage_times_us().items()})
    ex.close()
