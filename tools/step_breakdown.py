#!/usr/bin/env python3
"""Host-view timing of FrontEnd.step on the native orbf path: total wall, time blocked on the GPU, query building."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline, rt

W, H = 640, 480
params = [m.ExtractorParams(nfeatures=1000)] * 2
fe = pipeline.FrontEnd(params, W, H)
frames = [[synth.image(c, t, W, H) for c in range(2)] for t in range(8)]
dev = [[rt.DeviceBuffer(W * H) for c in range(2)] for t in range(8)]
for t in range(8):
    for c in range(2):
        dev[t][c].upload(frames[t][c])
rt.device_sync()
N = 300
OVERLAP = os.environ.get("OVERLAP", "1") != "0"
fe.copy_results = False
tot = wait = 0.0
host = np.zeros(4)
for it in range(N + 20):
    if it == 20:
        tot = wait = 0.0; host[:] = 0
    t = it % 8
    t0 = time.perf_counter()
    if OVERLAP and it == 0:
        fe.announce([(dev[1][c].ptr, W) for c in range(2)], resident=True)
    r = fe.step([(dev[t][c].ptr, W) for c in range(2)], resident=True,
                next_images=[(dev[(t + 2) % 8][c].ptr, W) for c in range(2)] if OVERLAP else None)
    tot += time.perf_counter() - t0; wait += r["gpu_wait_us"]; host += np.array(r["host_us"])
fe.ex.set_profiling(True); fe.step([(dev[1][c].ptr, W) for c in range(2)], resident=True)
print(json.dumps({"step_us": round(tot / N * 1e6, 1), "final_sync_wait_us": round(wait / N, 1), "host_us[query_prep,enqueue,wait,post]": [round(x / N, 1) for x in host], "extractor": fe.ex.stage_times_us()}))
