#!/usr/bin/env python3
"""Workload for rocprofv3 passes over the matcher kernels (run under `rocprofv3 ... -- python3 tools/profile_matcher.py`).

Launches, on one stream: a 1 GiB hipMemset (known write bytes: WRITE_SIZE calibration), a 1 GiB device-to-device
copy (known read + write bytes: FETCH_SIZE calibration), then k_hamming_matrix at Q = R = 32000 and k_hamming_top2 at
4000 x 4000 and 32000 x 32000, a few launches each."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import rt, synth

N = int(os.environ.get("MATRIX_N", "32000"))
GIB = 1 << 30
mt = m.Matcher()
st = mt.stream
a = rt.DeviceBuffer(GIB); b = rt.DeviceBuffer(GIB)
for _ in range(3):
    rt._L().orb_memset(a.ptr, 1, GIB, st)
    rt._L().orb_memcpy_d2d(b.ptr, a.ptr, GIB, st)
d = synth.descriptors(N, 4242)
dq = rt.DeviceBuffer(N * 32); dr = rt.DeviceBuffer(N * 32); dout = rt.DeviceBuffer(N * N * 2)
dq.upload(d); dr.upload(synth.perturbed_queries(d, 9))
res = [rt.DeviceBuffer(N * 4) for _ in range(3)]
scr = rt.DeviceBuffer(max(m.Matcher.top2_scratch_bytes(N, N), m.Matcher.top2_scratch_bytes(4000, 4000), 16))
for _ in range(5):
    m.Matcher.hamming_matrix_device(dq.ptr, N, dr.ptr, N, dout.ptr, st)
for _ in range(5):
    m.Matcher.hamming_top2_device(dq.ptr, 4000, dr.ptr, 4000, res[0].ptr, res[1].ptr, res[2].ptr, scr.ptr, st)
for _ in range(5):
    m.Matcher.hamming_top2_device(dq.ptr, N, dr.ptr, N, res[0].ptr, res[1].ptr, res[2].ptr, scr.ptr, st)
rt.stream_sync(st)
print("done")
