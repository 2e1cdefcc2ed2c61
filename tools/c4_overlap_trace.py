#!/usr/bin/env python3
"""Overlapped loop of one configuration (default configs[4]: 8 x 1080p @4000), a few dozen steps, for a kernel trace
(rocprofv3 --kernel-trace) of the steady state; tools/experiments/trace_timeline.py prints one period of it per stream.
usage: c4_overlap_trace.py [steps] [width height cameras features]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import pipeline, rt, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
W, H, NC, NF = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (1920, 1080, 8, 4000)
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=NF)] * NC, W, H)
RING = 6
dev = [[rt.DeviceBuffer(W * H) for c in range(NC)] for t in range(RING)]
for t in range(RING):
    for c in range(NC):
        dev[t][c].upload(synth.image(c, t, W, H))
rt.device_sync()
fe.copy_results = False
prep = [fe.prepare([(dev[t][c].ptr, W) for c in range(NC)]) for t in range(RING)]
fe.announce(prep[1]); fe.announce(prep[2])
hs, tt = [], []
for t in range(N):
    t0 = time.perf_counter()
    r = fe.step(prep[t % RING], resident=True, next_images=prep[(t + 3) % RING])
    tt.append(time.perf_counter() - t0)
    hs.append(r["host_us"])
hs = np.array(hs[N // 3:])
print("host_us medians [prep, enqueue, gpu_wait, post]:", np.median(hs, axis=0).round(1), "step us", round(float(np.median(tt[N // 3:])) * 1e6, 1))
fe.close()
