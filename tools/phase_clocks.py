#!/usr/bin/env python3
"""Phase clocks of the single-workgroup kernels (instrumented build: make -C multi_orb_slam_amd/csrc PHASES=1, run with
MORB_LIB_PATH=multi_orb_slam_amd/lib/libmorb_phases.so).  Prints microseconds between consecutive stamps (100 MHz clock)."""
import os, sys, ctypes as C, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline, rt, _lib

W, H = 640, 480
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=1000)] * 2, W, H)
frames = [[synth.image(c, t, W, H) for c in range(2)] for t in range(6)]
for t in range(6):
    r = fe.step(frames[t])
lib = _lib.lib()
out = (C.c_uint64 * 64)()


def deltas(v, idx):
    return [round((v[b] - v[a]) / 100.0, 2) for a, b in zip(idx[:-1], idx[1:])]


lib.morb_debug_phases_extractor(0, out); v = list(out)
npass = int(v[61])
idx = list(range(0, npass)) + [62]
print("octree (cam0 level0): load-scan, limits, dense-load, roots, passes..., select:", deltas(v, idx), "total", (v[62] - v[0]) / 100.0)
lib.morb_debug_phases_extractor(4, out); v = list(out)
print("pyramid tile (3, 3) of camera 0: spans, tables issued, level 0 in LDS, levels 1..7:", deltas(v, list(range(0, 11))), "total", (v[10] - v[0]) / 100.0)
lib.morb_debug_phases_extractor(2, out); v = list(out)
print("describe, one wave (block 40): slot bookkeeping, patch fetch, moments + angle, horizontal blur, vertical blur, rBRIEF, output:",
      deltas(v, [0, 1, 2, 3, 4, 5, 6, 7]), "total", (v[7] - v[0]) / 100.0)
lib.morb_debug_phases_matcher(0, out); v = list(out)
it = int(v[62])
if v[1] == 0:   # k_resolve_mono (the default of the frame search): set-up incl. round 0, rounds 1.., tail
    idx = [0, 2] + list(range(3, 2 + it)) + [60, 61]
    print("resolve (monotone): set-up + round 0, rounds x%d, tail, write:" % (it - 1), deltas(v, idx), "total", (v[61] - v[0]) / 100.0)
    print("  tail: owners + histogram, three maxima, reject:", deltas(v, [1 + it, 52, 53, 54, 60]))
    print("  shader clock over the kernel: %.0f MHz" % ((v[59] - v[58]) / ((v[61] - v[0]) / 100.0)))
else:           # k_resolve (Jacobi sweeps; MORB_RESOLVE_MONO=0, top-2 searches, states beyond LDS)
    idx = [0, 1, 2] + list(range(3, 3 + it)) + [60, 61]
    print("resolve: init, ldsq-load, sweeps x%d, tail, write:" % it, deltas(v, idx), "total", (v[61] - v[0]) / 100.0)
    print("  queries rescanned per sweep:", [int(x) for x in v[40:40 + it]])
    print("  sweep 0: thread-0 loop (incl. its wave's rescans), closing barrier:", deltas(v, [2, 20, 3]))
    print("  tail: reset, owners + histogram, three maxima, reject:", deltas(v, [2 + it, 52, 53, 54, 60]))
    if it > 5:
        print("  sweep 5: thread-0 loop (incl. its wave's rescans), closing barrier:", deltas(v, [7, 24, 8]))
out[0] = 0xC4A26E
lib.morb_debug_phases_matcher(0, out); c = list(out)
print("  queries that changed their choice per sweep / were displaced per round (sum over %d steps):" % 6, [int(x) for x in c[:12]])
print("  (monotone) queries rescanned per round (sum over %d steps):" % 6, [int(x) for x in c[16:28]])
print("  (monotone) passes with displaced queries of the busiest wave per round (max over the steps):", [int(x) for x in c[32:44]])
lib.morb_debug_phases_matcher(1, out); v = list(out)
print("frame_build: counts, fill, scan, scatter, sort:", deltas(v, [0, 1, 2, 3, 4, 5]), "total", (v[5] - v[0]) / 100.0)
