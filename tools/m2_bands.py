#!/usr/bin/env python3
"""The distance-matrix kernel's two per-process bands (366-368 / 397-401 us at Q = R = 32 000): what decides
which one a process lands in?  Within ONE process this times the settled kernel
  (a) on the same three buffers several times in a row (is a band a property of the process or of the moment?),
  (b) on a freshly allocated output matrix each time, with filler allocations of varying size in between (placement in HBM),
  (c) on the same allocation entered at different byte offsets (alignment of the 64 000-byte rows against channels / pages),
  (d) after an idle pause and after a stretch of other work (power / clock state),
and prints one JSON line per measurement: {"leg", "us", "out_ptr", "settle"}.  Usage: python tools/m2_bands.py [n]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_orb_slam_amd as m
from multi_orb_slam_amd import rt, synth
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
LEGS = sys.argv[2] if len(sys.argv) > 2 else "abcd"   # which legs to run; "ef": the clock legs below
mt = m.Matcher(); st = mt.stream
d = synth.descriptors(n, 4242)
dq = rt.DeviceBuffer(n * 32); dr = rt.DeviceBuffer(n * 32)
dq.upload(d); dr.upload(synth.perturbed_queries(d, 9))


def smi():
    s = bench.device_state() or {}
    return {k.split(" ")[0].lower(): v for k, v in s.items() if any(w in k.lower() for w in ("sclk clock speed", "power (w)", "junction"))}


def measure(leg, out_ptr, iters=100):
    ms, used, curve = bench._settled_launches(rt, lambda: m.Matcher.hamming_matrix_device(dq.ptr, n, dr.ptr, n, out_ptr, st), st, iters)
    print(json.dumps({"leg": leg, "us": round(ms * 1e3, 1), "out_ptr": hex(out_ptr), "settle_launches": used,
                      "first_group_us": curve[0], "smi": smi()}), flush=True)
    return ms


SZ = n * n * 2


def clock_legs():
    """(e) the same launches with a host pause between them (the part then runs at a lower average power): per-launch duration from
    an event pair around every launch; (f) a sustained stretch with rocm-smi sampled WHILE it runs (clocks and power under the load
    itself, not after it)."""
    import subprocess
    import threading
    out = rt.DeviceBuffer(SZ)
    run = lambda: m.Matcher.hamming_matrix_device(dq.ptr, n, dr.ptr, n, out.ptr, st)
    for _ in range(200):
        run()
    rt.stream_sync(st)
    for gap_us in (0, 100, 400, 1000, 3000):
        ev = [(rt.Event(), rt.Event()) for _ in range(150)]
        for a, b in ev:
            a.record(st); run(); b.record(st)
            if gap_us:
                rt.stream_sync(st)
                t0 = time.perf_counter()
                while (time.perf_counter() - t0) * 1e6 < gap_us:
                    pass
        rt.stream_sync(st)
        us = sorted(a.elapsed_ms(b) * 1e3 for a, b in ev[50:])
        print(json.dumps({"leg": "e_gap_%dus" % gap_us, "median_us": round(us[len(us) // 2], 1), "p10_us": round(us[len(us) // 10], 1),
                          "p90_us": round(us[9 * len(us) // 10], 1), "smi": smi()}), flush=True)
    samples = []
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(smi())
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); launches = 0
    e0, e1 = rt.Event(), rt.Event()
    e0.record(st)
    while time.perf_counter() - t0 < 4.0:
        for _ in range(50):
            run()
        launches += 50
        rt.stream_sync(st)
    e1.record(st); rt.stream_sync(st)
    stop.set(); th.join()
    print(json.dumps({"leg": "f_sustained_4s", "us": round(e0.elapsed_ms(e1) * 1e3 / launches, 1), "smi_during": samples[1:-1]}), flush=True)
    try:
        cap = subprocess.run(["rocm-smi", "-d", "0", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=20).stdout.strip()
    except Exception as e:
        cap = repr(e)
    print(json.dumps({"leg": "power_cap", "rocm_smi": cap}), flush=True)
    out.free()


def stream_legs():
    """(g) the same launches on eight streams created one after the other (HIP maps streams onto a few hardware queues round robin, the
    driver spreads queues over the command processor's pipes): is the band a property of the queue the launches go through?"""
    out = rt.DeviceBuffer(SZ)
    handles = [m.Matcher() for _ in range(8)]   # (every matcher handle owns a stream)
    streams = [h.stream for h in handles]
    for rnd in range(2):
        for i, s in enumerate(streams):
            run = lambda: m.Matcher.hamming_matrix_device(dq.ptr, n, dr.ptr, n, out.ptr, s)
            ms, used, curve = bench._settled_launches(rt, run, s, 60)
            print(json.dumps({"leg": "g_stream_%d_round_%d" % (i, rnd), "us": round(ms * 1e3, 1), "settle_launches": used}), flush=True)
    out.free()


def code_legs():
    """(h) the same kernel from SECOND and THIRD copies of the library loaded into this process (other code objects at other addresses, the
    same buffers and stream): is the band a property of where the kernel's code lies?"""
    import ctypes as C, shutil, tempfile
    out = rt.DeviceBuffer(SZ)
    libs = [("first", None)]
    d = tempfile.mkdtemp()
    for k in range(3):
        pth = os.path.join(d, "libmorb_copy%d.so" % k)
        shutil.copy(m.LIB_PATH, pth)
        libs.append(("copy%d" % k, C.CDLL(pth)))
    for rnd in range(2):
        for name, L in libs:
            if L is None:
                run = lambda: m.Matcher.hamming_matrix_device(dq.ptr, n, dr.ptr, n, out.ptr, st)
            else:
                fn = L.orbm_hamming_matrix_device
                fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
                run = (lambda f: (lambda: f(dq.ptr, n, dr.ptr, n, out.ptr, st)))(fn)
            ms, used, curve = bench._settled_launches(rt, run, st, 60)
            print(json.dumps({"leg": "h_%s_round_%d" % (name, rnd), "us": round(ms * 1e3, 1), "settle_launches": used}), flush=True)
    out.free()


def input_legs():
    """(i) fresh copies of the two 1 MB descriptor arrays (the kernel's inputs) at other addresses and offsets, the same result buffer"""
    out = rt.DeviceBuffer(SZ)
    keep = []
    qd = synth.perturbed_queries(d, 9)
    for k in range(8):
        keep.append(rt.DeviceBuffer((3 + 17 * k) << 16))
        q2 = rt.DeviceBuffer(n * 32 + 4096); r2 = rt.DeviceBuffer(n * 32 + 4096)
        q2.upload(d); r2.upload(qd)
        run = lambda: m.Matcher.hamming_matrix_device(q2.ptr, n, r2.ptr, n, out.ptr, st)
        ms, used, curve = bench._settled_launches(rt, run, st, 60)
        print(json.dumps({"leg": "i_inputs_%d" % k, "us": round(ms * 1e3, 1), "q_ptr": hex(q2.ptr), "r_ptr": hex(r2.ptr)}), flush=True)
        keep += [q2, r2]
    out.free()


if "e" in LEGS or "f" in LEGS:
    clock_legs()
if "i" in LEGS:
    input_legs()
if "h" in LEGS:
    code_legs()
if "g" in LEGS:
    stream_legs()
if not any(c in LEGS for c in "abcd"):
    print("m2_bands: done")
    sys.exit(0)
# (a) the same buffers, five times
out = rt.DeviceBuffer(SZ + (4 << 20))
for i in range(5):
    measure("a_same_buffers_%d" % i, out.ptr)
# (c) offsets into the same allocation
for off in (64, 128, 256, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 128):
    measure("c_offset_%d" % off, out.ptr + off)
out.free()
# (b) fresh allocations, fillers of growing size kept alive in between
fillers = []
for i in range(6):
    fillers.append(rt.DeviceBuffer((37 + 101 * i) << 20))
    o = rt.DeviceBuffer(SZ)
    measure("b_fresh_alloc_%d" % i, o.ptr)
    o.free()
for f in fillers:
    f.free()
# (d) idle, then busy
out = rt.DeviceBuffer(SZ)
measure("d_before_pause", out.ptr)
time.sleep(5.0)
measure("d_after_5s_idle", out.ptr)
# a stretch of the matrix-core top-2 (other work, the matrix pipe hot), then the matrix kernel at once
idx = rt.DeviceBuffer(n * 4); b1 = rt.DeviceBuffer(n * 4); b2 = rt.DeviceBuffer(n * 4)
sb = m.Matcher.top2_scratch_bytes(n, n)
scratch = rt.DeviceBuffer(max(sb, 16))
for _ in range(300):
    m.Matcher.hamming_top2_device(dq.ptr, n, dr.ptr, n, idx.ptr, b1.ptr, b2.ptr, scratch.ptr if sb else None, st)
measure("d_after_top2_stretch", out.ptr)
rt.stream_sync(st)
print("m2_bands: done")
