#!/usr/bin/env python3
"""A/B: the generic matrix-core top-2 (32000 x 32000) against the cross-camera form (8 cameras x 4000 in one list, own camera
excluded), alternating, a few launches each; run under `rocprofv3 --kernel-trace --stats` for kernel-only durations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import rt, synth
N = 32000
mt = m.Matcher(); st = mt.stream
d = synth.descriptors(N, 4242)
dq = rt.DeviceBuffer(N * 32); dr = rt.DeviceBuffer(N * 32)
dq.upload(d); dr.upload(synth.perturbed_queries(d, 9))
res = [rt.DeviceBuffer(N * 4) for _ in range(3)]
scr = rt.DeviceBuffer(max(m.Matcher.top2_scratch_bytes(N, N), 16))
cams = [synth.perturbed_queries(d[:4000], 100 + c, 0.06) for c in range(8)]
frc = dict(un_x=np.zeros(N, np.float32), un_y=np.zeros(N, np.float32), octave=np.zeros(N, np.int32), angle=np.zeros(N, np.float32),
           uright=np.full(N, -1, np.float32), cam_of=np.repeat(np.arange(8, dtype=np.int32), 4000),
           local_of=np.tile(np.arange(4000, dtype=np.int32), 8), descs=cams, bounds=(0.0, 0.0, 1920.0, 1080.0))
Fc = mt.frame(m.FrameData(**frc))
for rep in range(12):
    m.Matcher.hamming_top2_device(dq.ptr, N, dr.ptr, N, res[0].ptr, res[1].ptr, res[2].ptr, scr.ptr, st)
    rt.stream_sync(st)
    mt.cross_top2(Fc)
print("done")
