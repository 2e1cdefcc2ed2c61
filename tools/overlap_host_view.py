#!/usr/bin/env python3
"""The overlapped loop of configs[1] as bench.py runs it (prepared image arrays, three steps announced ahead), with the library's own
clocks summed: where a 40 us step goes on the HOST -- orbf_result::host_us = [entry -> begin, begin -> everything enqueued, blocked on
the GPU, after the wait].  `enqueue` >> `wait` means the loop is bound by the stepping thread, not by the device."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline, rt

W, H, NF, NC = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (640, 480, 1000, 2)   # e.g. 640 480 1000 4 = configs[3]
RING = 8
AHEAD = int(os.environ.get("AHEAD", "3"))
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=NF)] * NC, W, H)
fe.copy_results = False
dev = []
for t in range(RING):
    row = []
    for c in range(NC):
        b = rt.DeviceBuffer(W * H); b.upload(synth.image(c, t, W, H)); row.append(b)
    dev.append(row)
rt.device_sync()
prep = [fe.prepare([(dev[t][c].ptr, W) for c in range(NC)], True) for t in range(RING)]
for k in range(1, AHEAD):
    fe.announce(prep[k % RING], resident=True)
N, WARM = 4000, 400
host = np.zeros(4); wait = 0.0
for i in range(N + WARM):
    if i == WARM:
        host[:] = 0; wait = 0.0; t0 = time.perf_counter()
    r = fe.step(prep[i % RING], resident=True, next_images=prep[(i + AHEAD) % RING])
    host += np.array(r["host_us"]); wait += r["gpu_wait_us"]
dt = time.perf_counter() - t0
print(json.dumps({"ahead": AHEAD, "step_us": round(dt / N * 1e6, 2), "host_us[entry,enqueue,wait,post]": [round(x / N, 2) for x in host],
                  "outside_the_library_us": round(dt / N * 1e6 - host.sum() / N, 2)}))
