#!/usr/bin/env python3
"""How many sweeps does the multi-workgroup resolve of configs[4] (8 x 1080p @4000, 32 000 queries) need?  Prints
orbm_debug_last_resolve() = (status, matches, sweeps, longest candidate list) after every step of a short stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline
W, H, NF, NC = 1920, 1080, 4000, 8
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=NF)] * NC, W, H)
for t in range(8):
    r = fe.step([synth.image(c, t, W, H) for c in range(NC)])
    print(t, sum(r["counts"]), r["n_temporal"], fe.mt.last_resolve())
fe.close()
