#!/usr/bin/env python3
"""M1 (exhaustive top-2, Q = R = 32 000 by default) timed behind the clock transient: 300 untimed launches, then 200 timed with HIP
events; FP4 and int8 forms.  usage: python3 tools/top2_quick.py [n]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_orb_slam_amd as m
from multi_orb_slam_amd import rt, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
mt = m.Matcher(); st = mt.stream
d = synth.descriptors(n, 4242)
dq = rt.DeviceBuffer(n * 32); dr = rt.DeviceBuffer(n * 32)
dq.upload(d); dr.upload(synth.perturbed_queries(d, 9))
res = [rt.DeviceBuffer(n * 4) for _ in range(3)]
out = {}
for name, fp4 in (("fp4", -1), ("int8", 0)):
    prev = m.Matcher.use_fp4_top2(fp4)
    scr = rt.DeviceBuffer(max(m.Matcher.top2_scratch_bytes(n, n), 16))
    run = lambda: m.Matcher.hamming_top2_device(dq.ptr, n, dr.ptr, n, res[0].ptr, res[1].ptr, res[2].ptr, scr.ptr, st)
    for _ in range(300):
        run()
    rt.stream_sync(st)
    e0, e1 = rt.Event(), rt.Event()
    e0.record(st)
    for _ in range(200):
        run()
    e1.record(st)
    us = e0.elapsed_ms(e1) * 1e3 / 200
    out[name] = dict(us=round(us, 1), frac_of_peak=round(512.0 * n * n / us / 1e6 / (10000.0 if fp4 else 5000.0), 3))
    m.Matcher.use_fp4_top2(prev)
    scr.free()
print(json.dumps(out))
