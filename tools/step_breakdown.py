#!/usr/bin/env python3
"""Wall-time breakdown of one FrontEnd.step (host view), averaged over many steps."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline, rt
from multi_orb_slam_amd.matcher import FrameData, Matcher

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
fe.ex.set_profiling(False)
for it in range(N + 20):
    if it == 20:
        acc.clear()
    t = it % 8
    ex, mt = fe.ex, fe.mt
    t0 = time.perf_counter()
    for c in range(2):
        ex.upload_device(c, dev[t][c].ptr, W, H, W)
    t0 = tick("upload_device", t0)
    ex.run()
    t0 = tick("extract_run", t0)
    per_cam = [ex.download(c) for c in range(2)]
    t0 = tick("download", t0)
    uright = np.concatenate([pipeline.synth_uright(k) for k, _ in per_cam])
    fd = FrameData.from_cameras(per_cam, W, H, uright)
    t0 = tick("frame_merge_py", t0)
    if fe.prev is not None:
        frame = mt.frame(fd)
        t0 = tick("frame_create", t0)
        q = pipeline.make_queries(fe.prev, fe.scale)
        t0 = tick("make_queries_py", t0)
        n, mo = mt.SearchByProjection(frame, q)
        t0 = tick("search_by_projection", t0)
        frame.close()
        t0 = tick("frame_destroy", t0)
    fe.prev = per_cam
    for c in range(2):
        o = 1 - c
        nq, nr = len(per_cam[c][0]), len(per_cam[o][0])
        Matcher.hamming_top2_device(ex.device_descriptors(c), nq, ex.device_descriptors(o), nr, fe.d_res[0].ptr,
                                    fe.d_res[1].ptr, fe.d_res[2].ptr, fe.d_scratch.ptr, fe.stream)
        bi = fe.d_res[0].download(np.int32, nq, fe.stream); bd = fe.d_res[1].download(np.int32, nq, fe.stream)
        sd = fe.d_res[2].download(np.int32, nq, fe.stream)
    t0 = tick("cross_top2", t0)
tot = sum(acc.values())
print(json.dumps({k: round(v / N * 1e6, 1) for k, v in acc.items()} | {"total_us": round(tot / N * 1e6, 1)}))
fe.ex.set_profiling(True); fe.step([(dev[1][c].ptr, W) for c in range(2)], resident=True); print(fe.ex.stage_times_us())
