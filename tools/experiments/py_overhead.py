#!/usr/bin/env python3
"""Where the Python binding spends its ~9 us per overlapped step (cProfile over 3000 steps)."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline, rt
W, H = 640, 480
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=1000)] * 2, W, H)
dev = [[rt.DeviceBuffer(W * H) for c in range(2)] for t in range(8)]
for t in range(8):
    for c in range(2):
        dev[t][c].upload(synth.image(c, t, W, H))
fe.copy_results = False
AHEAD = 3
args = [fe.prepare([(dev[t][c].ptr, W) for c in range(2)], True) for t in range(8)]   # (as bench.py: marshalled once)
for k in range(1, AHEAD):
    fe.announce(args[k], resident=True)
def loop(n):
    for it in range(n):
        fe.step(args[it % 8], resident=True, next_images=args[(it + AHEAD) % 8])
loop(200)
pr = cProfile.Profile(); pr.enable(); loop(3000); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:4000])
