#!/usr/bin/env python3
"""Throughput of the native stream loop with 0 / 1 / 2 timesteps announced ahead (how much look-ahead pays)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline, rt
from multi_orb_slam_amd.matcher import TH_LOW
W, H, RING = 640, 480, 8
dev = [[rt.DeviceBuffer(W * H) for c in range(2)] for t in range(RING)]
for t in range(RING):
    for c in range(2):
        dev[t][c].upload(synth.image(c, t, W, H))
ring = [[(dev[t][c].ptr, W, H, W, 1) for c in range(2)] for t in range(RING)]
motion = (pipeline.MOTION[0], pipeline.MOTION[1], pipeline.TH_PROJ)
out = {}
for ahead in (0, 1, 2, 1, 2):
    fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=1000)] * 2, W, H)
    upto = -1
    st, upto = fe.fe.run_stream(ring, 0, 300, ahead, upto, motion, TH_LOW, pipeline.BOW_RATIO)
    best = 1e9
    for rep in range(5):
        st, upto = fe.fe.run_stream(ring, 300 + 2000 * rep, 2000, ahead, upto, motion, TH_LOW, pipeline.BOW_RATIO)
        best = min(best, st["seconds"] / 2000)
    out.setdefault(ahead, []).append(round(1e6 * best, 1))
    fe.close()
print(json.dumps({"us_per_step_by_ahead": out}))
