#!/usr/bin/env python3
"""Where a configs[4] timestep (8 x 1920x1080 @4000 on one GPU) spends its time: host timeline + extractor stages."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline, rt
W, H, NF, NC = 1920, 1080, 4000, 8
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=NF)] * NC, W, H)
fe.copy_results = False
host = [[synth.image(c, t, W, H) for c in range(NC)] for t in range(3)]
dev = []
for t in range(3):
    row = []
    for c in range(NC):
        b = rt.DeviceBuffer(W * H); b.upload(host[t][c]); row.append(b)
    dev.append(row)
rt.device_sync()
arg = lambda t: [(dev[t % 3][c].ptr, W) for c in range(NC)]
OV = os.environ.get('OVERLAP', '1') == '1'
for i in range(5):
    fe.step(arg(i), resident=True, next_images=arg(i + 1) if OV else None)
hs = np.zeros(4); n = 20
t0 = time.perf_counter()
for i in range(5, 5 + n):
    r = fe.step(arg(i), resident=True, next_images=arg(i + 1) if OV else None); hs += np.array(r["host_us"])
dt = (time.perf_counter() - t0) / n
fe.reset(); fe.ex.set_profiling(True); fe.step(arg(0), resident=True)
print(json.dumps({"ms_per_step": round(dt * 1e3, 3), "host_us[prep,enqueue,wait,post]": [round(x / n, 1) for x in hs],
                  "extractor_us": {k: round(v, 1) for k, v in fe.ex.stage_times_us().items()}, "path": fe.ex.last_path(),
                  "resolve": fe.mt.last_resolve()}))
