#!/usr/bin/env python3
"""Rounds of the frame search's resolve on the benchmark stream: orbm_debug_last_resolve() = (status, matches, rounds + 1, longest
candidate list).  k_resolve_mono (default): rounds of wave-local fixed points, 2-3 here; MORB_RESOLVE_MONO=0: Jacobi sweeps, 9-10."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline
W, H = 640, 480
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=1000)] * 2, W, H)
for t in range(8):
    r = fe.step([synth.image(c, t, W, H) for c in range(2)])
    print(t, sum(r["counts"]), r["n_temporal"], fe.mt.last_resolve())
fe.close()
