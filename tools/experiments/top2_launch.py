#!/usr/bin/env python3
"""A handful of launches of the exhaustive top-2 on the matrix cores (Q = R = 32000) for `rocprofv3 --pmc ...` passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, rt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
d = synth.descriptors(n, 777); qh = synth.perturbed_queries(d, 11)
dq = rt.DeviceBuffer(n * 32); dr = rt.DeviceBuffer(n * 32); dq.upload(qh); dr.upload(d)
o = [rt.DeviceBuffer(n * 4) for _ in range(3)]
sb = m.Matcher.top2_scratch_bytes(n, n); scratch = rt.DeviceBuffer(max(sb, 16))
for _ in range(iters):
    m.Matcher.hamming_top2_device(dq.ptr, n, dr.ptr, n, o[0].ptr, o[1].ptr, o[2].ptr, scratch.ptr if sb else None, None)
rt.device_sync()
