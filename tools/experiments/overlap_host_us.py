#!/usr/bin/env python3
"""Overlapped configs[1] loop (two steps announced ahead): where a step's wall time goes as the library sees it --
orbf_result::host_us = [before begin, begin (everything enqueued), wait for the results, collect] -- medians over the loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline, rt
W, H = 640, 480
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=1000)] * 2, W, H)
dev = [[rt.DeviceBuffer(W * H) for c in range(2)] for t in range(8)]
for t in range(8):
    for c in range(2):
        dev[t][c].upload(synth.image(c, t, W, H))
rt.device_sync()
fe.copy_results = False
args = lambda t: [(dev[t % 8][c].ptr, W) for c in range(2)]
AHEAD = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for k in range(1, AHEAD):
    fe.announce(args(k), resident=True)
hs, ts = [], []
for it in range(3000):
    t0 = time.perf_counter()
    r = fe.step(args(it), resident=True, next_images=args(it + AHEAD))
    ts.append(time.perf_counter() - t0); hs.append(r["host_us"])
hs = np.array(hs[500:]); ts = np.array(ts[500:]) * 1e6
print("ahead", AHEAD, "step wall median %.1f us; host_us medians: pre %.1f, begin %.1f, wait %.1f, collect %.1f; python around the two calls %.1f" %
      (np.median(ts), *np.median(hs, axis=0), np.median(ts - hs[:, 1] - hs[:, 2] - hs[:, 3])))
fe.close()
