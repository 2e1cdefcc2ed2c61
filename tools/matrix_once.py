#!/usr/bin/env python3
"""Counter workload of bench.py's live `roofline.traffic`: a 1 GiB hipMemset and a 1 GiB device copy (known bytes: the
calibration of WRITE_SIZE / FETCH_SIZE, MI355X_MICROARCH.md section HBM) and three launches of the distance-matrix kernel at
Q = R = n.  Run by bench.py under `rocprofv3 --pmc FETCH_SIZE` and, in a second pass, `--pmc WRITE_SIZE`."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_orb_slam_amd as m
from multi_orb_slam_amd import rt, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
GIB = 1 << 30
mt = m.Matcher(); st = mt.stream
a = rt.DeviceBuffer(GIB); b = rt.DeviceBuffer(GIB)
rt._L().orb_memset(a.ptr, 1, GIB, st)
rt._L().orb_memcpy_d2d(b.ptr, a.ptr, GIB, st)
d = synth.descriptors(n, 4242)
dq = rt.DeviceBuffer(n * 32); dr = rt.DeviceBuffer(n * 32); dout = rt.DeviceBuffer(n * n * 2)
dq.upload(d); dr.upload(synth.perturbed_queries(d, 9))
for _ in range(3):
    m.Matcher.hamming_matrix_device(dq.ptr, n, dr.ptr, n, dout.ptr, st)
rt.stream_sync(st)
print("matrix_once: n = %d" % n)
