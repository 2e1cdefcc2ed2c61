#!/usr/bin/env python3
"""The distance matrix's two per-process bands, round 6: tools/m2_bands.py (round 5) re-allocated the OUTPUT matrix inside a process
and found the band a property of the process.  What it never moved are the two 1 MB INPUT tables, allocated once per process: this times
the settled kernel on fresh input allocations (filler allocations of varying size in between, the output kept), then on fresh streams.
One JSON line per measurement."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_orb_slam_amd as m
from multi_orb_slam_amd import rt, synth
import bench

n = 32000
mt = m.Matcher(); st = mt.stream
d = synth.descriptors(n, 4242); q = synth.perturbed_queries(d, 9)
out = rt.DeviceBuffer(n * n * 2)


def measure(leg, dq, dr, stream):
    ms, used, curve = bench._settled_launches(rt, lambda: m.Matcher.hamming_matrix_device(dq.ptr, n, dr.ptr, n, out.ptr, stream), stream, 100)
    print(json.dumps({"leg": leg, "us": round(ms * 1e3, 1), "dq": hex(dq.ptr), "dr": hex(dr.ptr), "settle_launches": used}), flush=True)


keep = []
for i in range(8):
    if i:
        keep.append(rt.DeviceBuffer((1 << 20) * (1 + 7 * i) + 4096 * i))     # filler: moves the next allocations
    dq = rt.DeviceBuffer(n * 32); dr = rt.DeviceBuffer(n * 32)
    dq.upload(d); dr.upload(q)
    measure("fresh_inputs_%d" % i, dq, dr, st)
    keep += [dq, dr]
print("m2_inputs: done")
