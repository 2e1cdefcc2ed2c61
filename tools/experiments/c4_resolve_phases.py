#!/usr/bin/env python3
"""Phase clocks of k_rs_mono_cam (camera 0's workgroup) on configs[4] isolated steps; needs the PHASES build:
make -C multi_orb_slam_amd/csrc PHASES=1; MORB_LIB_PATH=multi_orb_slam_amd/lib/libmorb_phases.so python3 tools/experiments/c4_resolve_phases.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import pipeline, rt, synth, _lib
W, H, NC, NF = 1920, 1080, 8, 4000
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=NF)] * NC, W, H)
frames = [[synth.image(c, t, W, H) for c in range(NC)] for t in range(4)]
acc = []
for t in range(8):
    r = fe.step(frames[t % 4])
    out = (C.c_uint64 * 64)()
    _lib.lib().morb_debug_phases_matcher(0, out); v = list(out)
    if v[1] == 2 and t >= 2:
        acc.append([(v[b] - v[a]) / 100.0 for a, b in ((0, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7))] + [(v[7] - v[0]) / 100.0, v[62]])
a = np.array(acc)
print("k_rs_mono_cam, camera 0 (us): count pass, gather pass, set-up + round 0, rounds, owners + histogram, write-out | total, rounds:")
print(np.median(a, axis=0).round(2))
fe.close()
