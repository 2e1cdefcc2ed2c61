#!/usr/bin/env python3
"""Phase clocks of k_octree (workgroup 0 = camera 0, level 0) on the instrumented build:
    make -C multi_orb_slam_amd/csrc PHASES=1 && MORB_LIB_PATH=multi_orb_slam_amd/lib/libmorb_phases.so python tools/octree_phases.py
Microseconds (100 MHz clock) between the stamps: cell scan, limits, key load, roots, every pass, selection."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, _lib
import natural

lib = _lib.lib()
if os.environ.get("OCT_DBG"):
    lib.morb_debug_oct_dbg(int(os.environ["OCT_DBG"]))
    print("OCT_DBG =", os.environ["OCT_DBG"], "(results are garbage: timing of the first passes only)")
out = (C.c_uint64 * 64)()
for name, w, h, nf in (("synthetic", 640, 480, 1000), ("synthetic", 1280, 720, 2000), ("synthetic", 1920, 1080, 4000), ("china", 640, 480, 1000),
                       ("china", 1920, 1080, 4000), ("hopper", 1280, 720, 2000)):
    ex = m.Extractor([m.ExtractorParams(nfeatures=nf)], w, h)
    img = synth.image(0, 0, w, h) if name == "synthetic" else natural.frame(name, 0, 0, w, h)
    for _ in range(3):
        ex.extract([img])
    lib.morb_debug_phases_extractor(0, out); v = list(out)
    npass = int(v[61])
    idx = list(range(0, npass)) + [62]
    d = [round((v[b] - v[a]) / 100.0, 2) for a, b in zip(idx[:-1], idx[1:])]
    print("%-9s %4dx%-4d level-0 candidates %6d: scan %.2f, limits %.2f, load %.2f, roots %.2f, passes %s, select %.2f; total %.1f us"
          % (name, w, h, len(ex.debug_candidates(0, 0)), d[0], d[1], d[2], d[3] if len(d) > 3 else 0, d[4:-1], d[-1], (v[62] - v[0]) / 100.0), flush=True)
    ex.close()
