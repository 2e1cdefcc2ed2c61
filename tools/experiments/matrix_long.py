#!/usr/bin/env python3
"""The distance matrix launched 4000 times in a row (1.5 s), average per 200 launches: does a 'slow' process stay slow?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, rt
n = 32000
d = synth.descriptors(n, 777); qh = synth.perturbed_queries(d, 11)
dq = rt.DeviceBuffer(n * 32); dr = rt.DeviceBuffer(n * 32); dq.upload(qh); dr.upload(d)
dout = rt.DeviceBuffer(n * n * 2)
run = lambda: m.Matcher.hamming_matrix_device(dq.ptr, n, dr.ptr, n, dout.ptr, None)
ev = [rt.Event() for _ in range(21)]
ev[0].record(None)
for k in range(20):
    for _ in range(200): run()
    ev[k + 1].record(None)
rt.device_sync()
print([round(ev[k].elapsed_ms(ev[k + 1]) * 1e3 / 200, 1) for k in range(20)])
