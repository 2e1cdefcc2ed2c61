#!/usr/bin/env python3
"""Timestep time on PHOTOGRAPHS (informational; bench.py's contract is the synthetic stream): overlapping camera rigs sliding over the
three photographs of tests/natural.py (content moving by (3, 1) px per step, as the synthetic stream's), images resident in HBM, isolated
steps and three announced ahead.  A photograph puts 5-30 x the candidates of the synthetic rectangles on a level (10 000 at 640x480,
33 000 at 1080p): the quadtree and FAST stages weigh more.  One JSON line per (photo, configuration, look-ahead)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import multi_orb_slam_amd as m
from multi_orb_slam_amd import pipeline, rt
import natural

CONFIGS = {"configs[1] 2x640x480@1000": (640, 480, 1000, 2), "configs[2] 2x1280x720@2000": (1280, 720, 2000, 2),
           "configs[3] 4x640x480@1000": (640, 480, 1000, 4), "2x1920x1080@4000": (1920, 1080, 4000, 2)}
photos = sys.argv[1:] or ["china", "hopper"]
for photo in photos:
    for name, (W, H, NF, NC) in CONFIGS.items():
        fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=NF)] * NC, W, H)
        fe.copy_results = False
        R = 8
        dev = []
        for t in range(R):
            row = []
            for im in natural.rig(photo, t, W, H, n_cams=NC):
                b = rt.DeviceBuffer(W * H); b.upload(im); row.append(b)
            dev.append(row)
        rt.device_sync()
        arg = lambda t: [(dev[t % R][c].ptr, W) for c in range(NC)]
        for ahead in (0, 3):
            fe.reset()
            n = 300 if W <= 1280 else 100
            for k in range(1, ahead):
                fe.announce(arg(k), resident=True)
            for i in range(8):
                r = fe.step(arg(i), resident=True, next_images=arg(i + ahead) if ahead else None)
            t0 = time.perf_counter()
            for i in range(8, 8 + n):
                r = fe.step(arg(i), resident=True, next_images=arg(i + ahead) if ahead else None)
            dt = (time.perf_counter() - t0) / n
            print(json.dumps({"photo": photo, "config": name, "announced_ahead": ahead, "ms_per_step": round(dt * 1e3, 4), "steps_per_s": round(1 / dt, 1),
                              "keypoints": r["counts"], "temporal_matches": r["n_temporal"], "cross_accepted": r["n_cross"]}), flush=True)
        fe.close()
