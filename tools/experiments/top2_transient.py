#!/usr/bin/env python3
"""Launch duration of the exhaustive top-2 (matrix-core form, Q = R = 32000) over time: averages of consecutive groups of 10
launches after an idle period -- how long the clocks take to settle, i.e. how long bench.py has to time for a sustained figure."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, rt
n = 32000
d = synth.descriptors(n, 777); qh = synth.perturbed_queries(d, 11)
dq = rt.DeviceBuffer(n * 32); dr = rt.DeviceBuffer(n * 32); dq.upload(qh); dr.upload(d)
o = [rt.DeviceBuffer(n * 4) for _ in range(3)]
sb = m.Matcher.top2_scratch_bytes(n, n); scratch = rt.DeviceBuffer(max(sb, 16))
st = rt.Stream() if hasattr(rt, "Stream") else None
s = st.ptr if st else None
run = lambda: m.Matcher.hamming_top2_device(dq.ptr, n, dr.ptr, n, o[0].ptr, o[1].ptr, o[2].ptr, scratch.ptr if sb else None, s)
for trial in range(2):
    time.sleep(0.5)
    ev = [rt.Event() for _ in range(31)]
    ev[0].record(s)
    for g in range(30):
        for _ in range(10):
            run()
        ev[g + 1].record(s)
    rt.device_sync()
    print("trial %d: us per launch, groups of 10:" % trial, [round(ev[g].elapsed_ms(ev[g + 1]) * 100, 1) for g in range(30)])
# the same for the uint16 distance matrix (k_hamming_matrix_mfma)
dout = rt.DeviceBuffer(n * n * 2)
runm = lambda: m.Matcher.hamming_matrix_device(dq.ptr, n, dr.ptr, n, dout.ptr, s)
for trial in range(2):
    time.sleep(0.5)
    ev = [rt.Event() for _ in range(31)]
    ev[0].record(s)
    for g in range(30):
        for _ in range(10):
            runm()
        ev[g + 1].record(s)
    rt.device_sync()
    print("matrix trial %d: us per launch, groups of 10:" % trial, [round(ev[g].elapsed_ms(ev[g + 1]) * 100, 1) for g in range(30)])
