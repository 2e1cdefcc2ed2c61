#!/usr/bin/env python3
"""Isolated timesteps back to back, with and without the camera-pair top-2 (whose fork onto the side stream puts an event
record between the frame build and the projection kernel): the difference bounds what removing the fork could save."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline, rt
from multi_orb_slam_amd.frontend import SKIP_CROSS
W, H = 640, 480
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=1000)] * 2, W, H)
dev = [[rt.DeviceBuffer(W * H) for c in range(2)] for t in range(8)]
for t in range(8):
    for c in range(2):
        dev[t][c].upload(synth.image(c, t, W, H))
out = {}
for name, flags in (("with_cross", 0), ("skip_cross", SKIP_CROSS), ("with_cross_again", 0), ("skip_cross_again", SKIP_CROSS)):
    ts, cabi = [], []
    for it in range(700):
        imgs = [(dev[it % 8][c].ptr, W, H, W, 1) for c in range(2)]
        t0 = time.perf_counter()
        r = fe.fe.step(imgs, None, flags, copy=False, motion=(pipeline.MOTION[0], pipeline.MOTION[1], pipeline.TH_PROJ))
        ts.append(time.perf_counter() - t0)
        h = r["host_us"]; cabi.append(h[1] + h[2] + h[3])
    ts = sorted(ts[100:]); cabi = sorted(cabi[100:])
    out[name] = {"python_us": round(1e6 * ts[len(ts) // 2], 1), "c_abi_us": round(cabi[len(cabi) // 2], 1)}
print(json.dumps(out))
fe.close()
