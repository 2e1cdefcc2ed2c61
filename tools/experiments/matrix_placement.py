#!/usr/bin/env python3
"""Does the distance matrix's launch time depend on WHERE its 2 GB result lies?  Several result buffers allocated in one process (kept
alive, so that every one is a different piece of memory), the same kernel timed on each after the clocks have settled."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, rt
n = 32000
d = synth.descriptors(n, 777); qh = synth.perturbed_queries(d, 11)
dq = rt.DeviceBuffer(n * 32); dr = rt.DeviceBuffer(n * 32); dq.upload(qh); dr.upload(d)
sizes = [n * n * 2, n * n * 2, n * n * 2 + (1 << 20), 1 << 32, n * n * 2, (3 << 30)]
bufs = [rt.DeviceBuffer(s) for s in sizes]
def timed(ptr, reps=100):
    run = lambda: m.Matcher.hamming_matrix_device(dq.ptr, n, dr.ptr, n, ptr, None)
    for _ in range(60): run()
    e0, e1 = rt.Event(), rt.Event()
    e0.record(None)
    for _ in range(reps): run()
    e1.record(None)
    return e0.elapsed_ms(e1) * 1e3 / reps
for _ in range(250): m.Matcher.hamming_matrix_device(dq.ptr, n, dr.ptr, n, bufs[0].ptr, None)
for rnd in range(2):
    for b, s in zip(bufs, sizes):
        for off in (0, 1 << 21, (1 << 30) - (b.ptr & ((1 << 30) - 1)) if s >= (3 << 30) else 0):
            if off and off + n * n * 2 > s: continue
            us = timed(b.ptr + off)
            print("round %d  buffer %#x (+%#x, size %.2f GiB, ptr mod 1 GiB = %#x): %.1f us = %.3f" % (rnd, b.ptr, off, s / 2**30, (b.ptr + off) & ((1 << 30) - 1), us, 2.050048e9 / us / 8e6), flush=True)
