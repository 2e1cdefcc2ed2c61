#!/usr/bin/env python3
"""Kernel micro-benchmarks with HIP events on the launch stream: matrix-mode and top-2 Hamming at several sizes."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import rt, synth


def time_it(fn, st, iters=20, warm=3):
    for _ in range(warm):
        fn()
    rt.stream_sync(st)
    e0, e1 = rt.Event(), rt.Event()
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    return e0.elapsed_ms(e1) * 1e3 / iters


def main():
    sizes = [int(x) for x in os.environ.get("SIZES", "32000").split(",")]
    mt = m.Matcher(); st = mt.stream
    out = {}
    for n in sizes:
        d = synth.descriptors(n, 4242)
        dq = rt.DeviceBuffer(n * 32); dr = rt.DeviceBuffer(n * 32); dout = rt.DeviceBuffer(n * n * 2)
        dq.upload(d); dr.upload(synth.perturbed_queries(d, 9))
        us = time_it(lambda: m.Matcher.hamming_matrix_device(dq.ptr, n, dr.ptr, n, dout.ptr, st), st)
        alg = 64.0 * n + 2.0 * n * n
        out["matrix_%d" % n] = dict(us=round(us, 1), GBs=round(alg / us / 1e3, 1))
        res = [rt.DeviceBuffer(n * 4) for _ in range(3)]
        scr = rt.DeviceBuffer(max(m.Matcher.top2_scratch_bytes(n, n), 16))
        us = time_it(lambda: m.Matcher.hamming_top2_device(dq.ptr, n, dr.ptr, n, res[0].ptr, res[1].ptr, res[2].ptr, scr.ptr, st), st)
        out["top2_%d" % n] = dict(us=round(us, 1), Gpairs_s=round(n * n / us / 1e3, 1), valu_Tops=round(18.0 * n * n / us / 1e6, 2))
        for b in [dq, dr, dout, scr] + res:
            b.free()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
