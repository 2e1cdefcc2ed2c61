#!/usr/bin/env python3
"""Timestep time of the other BASELINE.json configurations on ONE GPU (informational; bench.py times configs[1])."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline, rt
for name, (W, H, NF, NC) in {"configs[1] 2x640x480@1000": (640, 480, 1000, 2), "configs[2] 2x1280x720@2000": (1280, 720, 2000, 2), "configs[3] 4x640x480@1000": (640, 480, 1000, 4),
                              "configs[4] 8x1920x1080@4000": (1920, 1080, 4000, 8)}.items():
    fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=NF)] * NC, W, H)
    fe.copy_results = False
    R = 4
    host = [[synth.image(c, t, W, H) for c in range(NC)] for t in range(R)]
    dev = []
    for t in range(R):
        row = []
        for c in range(NC):
            b = rt.DeviceBuffer(W * H); b.upload(host[t][c]); row.append(b)
        dev.append(row)
    rt.device_sync()
    arg = lambda t: [(dev[t % R][c].ptr, W) for c in range(NC)]
    for ov in (0, 1, 2):        # timesteps announced ahead
        fe.reset()
        n = 400 if W <= 1280 else 40
        if ov == 2:
            fe.announce(arg(1), resident=True)
        for i in range(6):
            fe.step(arg(i), resident=True, next_images=arg(i + ov) if ov else None)
        t0 = time.perf_counter()
        for i in range(6, 6 + n):
            r = fe.step(arg(i), resident=True, next_images=arg(i + ov) if ov else None)
        dt = (time.perf_counter() - t0) / n
        print(json.dumps({"config": name, "announced_ahead": ov, "ms_per_step": round(dt * 1e3, 3), "steps_per_s": round(1 / dt, 1),
                          "keypoints": r["counts"], "temporal_matches": r["n_temporal"]}))
    fe.close()
