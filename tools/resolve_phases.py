#!/usr/bin/env python3
"""Phase clocks of the frame search's resolve on the instrumented build (make PHASES=1; MORB_LIB_PATH=.../libmorb_phases.so):
configs[2] / [3] isolated steps, the single-workgroup form (MORB_RS_CAMS_MIN_Q=100000) or the per-camera form (default)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline, _lib
lib = _lib.lib()
out = (C.c_uint64 * 64)()
CFGS = {"configs[2]": (1280, 720, 2000, 2), "configs[3]": (640, 480, 1000, 4), "configs[1]": (640, 480, 1000, 2), "configs[4]": (1920, 1080, 4000, 8)}
if len(sys.argv) > 1:
    CFGS = {k: v for k, v in CFGS.items() if k in sys.argv[1:]}
for name, (W, H, NF, NC) in CFGS.items():
    fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=NF)] * NC, W, H)
    for t in range(6):
        fe.step([synth.image(c, t, W, H) for c in range(NC)])
    lib.morb_debug_phases_matcher(0, out); v = list(out)
    it = int(v[62])
    d = lambda idx: [round((v[b] - v[a]) / 100.0, 2) for a, b in zip(idx[:-1], idx[1:])]
    if v[1] == 2:
        print(name, "per-camera (camera 0): count, gather, set-up, rounds x%d, owners->meeting, tail:" % (it - 1), d([0, 2, 3, 4, 5, 6, 7]), "total", (v[7] - v[0]) / 100.0)
    elif v[1] == 0:
        idx = [0, 2] + list(range(3, 2 + it)) + [60, 61]
        print(name, "one workgroup: set-up + round 0, rounds x%d, tail, write:" % (it - 1), d(idx), "total", (v[61] - v[0]) / 100.0,
              " tail: owners+hist, maxima, reject:", d([1 + it, 52, 53, 54, 60]))
    if v[1] == 0:
        out[0] = 0xC4A26E
        lib.morb_debug_phases_matcher(0, out); c = list(out)
        print("   round 1, per wave: (passes, us since kernel start when it left):", [(int(x) >> 32, round((int(x) & 0xffffffff) / 100.0, 1)) for x in c[48:64]])
        print("   displaced per round (sum over 6 steps):", [int(x) for x in c[:8]], " rescanned:", [int(x) for x in c[16:24]], " passes of the busiest wave per round (max):", [int(x) for x in c[32:40]])
    fe.close()
