#!/usr/bin/env python3
"""Wall-time breakdown of one FrontEnd.step (host view), averaged over many steps."""
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
acc = {}
def tick(name, t0):
    t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1
N = 300
ex, mt = fe.ex, fe.mt
for it in range(N + 20):
    if it == 20:
        acc.clear()
    t = it % 8
    t0 = time.perf_counter()
    for c in range(2):
        ex.upload_device(c, dev[t][c].ptr, W, H, W)
    t0 = tick("upload_device", t0)
    ex.run()
    t0 = tick("extract_run", t0)
    counts = [ex.count(c) for c in range(2)]
    cams = [(ex.device_keypoints(c), ex.device_descriptors(c), counts[c], fe.depth_dev[c].ptr, W) for c in range(2)]
    frame = mt.frame_from_device(cams, pipeline.MBF, (0.0, 0.0, float(W), float(H)))
    t0 = tick("frame_from_device(enqueue)", t0)
    kps, desc, uright, depth = frame.download()
    t0 = tick("frame.download(sync)", t0)
    cam_of = np.repeat(np.arange(2, dtype=np.int32), counts)
    if fe.prev is not None:
        q = pipeline.make_queries(fe.prev, fe.scale)
        t0 = tick("make_queries_py", t0)
        n, mo = mt.SearchByProjection(frame, q)
        t0 = tick("search_by_projection", t0)
    fe.prev = (kps, desc, depth, cam_of)
    bi, bd, sd = mt.cross_top2(frame)
    t0 = tick("cross_top2", t0)
    nc = int(pipeline.accept_cross(bd, sd).sum())
    frame.close()
    t0 = tick("accept+close", t0)
import ctypes
from multi_orb_slam_amd import _lib
st = (ctypes.c_int * 4)(); _lib.lib().orbm_debug_last_resolve(mt._h, st); print('last resolve {status, nmatches, sweeps, longest}:', list(st), 'nq', len(q))
tot = sum(acc.values())
print(json.dumps({k: round(v / N * 1e6, 1) for k, v in acc.items()} | {"total_us": round(tot / N * 1e6, 1)}))
