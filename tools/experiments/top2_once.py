#!/usr/bin/env python3
"""A few launches of the exhaustive top-2 (FP4 form unless argv[2] == int8) for counter passes.  usage: top2_once.py [n] [fp4|int8] [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multi_orb_slam_amd as m
from multi_orb_slam_amd import rt, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
form = sys.argv[2] if len(sys.argv) > 2 else "fp4"
L = int(sys.argv[3]) if len(sys.argv) > 3 else 5
mt = m.Matcher(); st = mt.stream
d = synth.descriptors(n, 4242)
dq = rt.DeviceBuffer(n * 32); dr = rt.DeviceBuffer(n * 32)
dq.upload(d); dr.upload(synth.perturbed_queries(d, 9))
res = [rt.DeviceBuffer(n * 4) for _ in range(3)]
m.Matcher.use_fp4_top2(-1 if form == "fp4" else 0)
scr = rt.DeviceBuffer(max(m.Matcher.top2_scratch_bytes(n, n), 16))
for _ in range(L):
    m.Matcher.hamming_top2_device(dq.ptr, n, dr.ptr, n, res[0].ptr, res[1].ptr, res[2].ptr, scr.ptr, st)
rt.stream_sync(st)
print("done")
