#!/usr/bin/env python3
"""Start / end of every wave of one k_describe launch (instrumented build, see tools/phase_clocks.py)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline, _lib
W, H = 640, 480
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=1000)] * 2, W, H)
for t in range(6):
    fe.step([synth.image(c, t, W, H) for c in range(2)])
out = (C.c_uint64 * 8192)()
_lib.lib().morb_debug_phases_extractor(3, out)
v = np.array(list(out), dtype=np.int64).reshape(2, 4096)
ok = (v[0] > 0) & (v[1] >= v[0])
t0 = v[0][ok].min()
st = (v[0][ok] - t0) / 100.0; en = (v[1][ok] - t0) / 100.0
print("waves with a keypoint: %d; start us: min %.2f median %.2f p95 %.2f max %.2f" % (ok.sum(), st.min(), np.median(st), np.percentile(st, 95), st.max()))
print("duration us: min %.2f median %.2f p95 %.2f max %.2f;  end us: median %.2f p95 %.2f max %.2f" % ((en - st).min(), np.median(en - st), np.percentile(en - st, 95), (en - st).max(), np.median(en), np.percentile(en, 95), en.max()))
idx = np.nonzero(ok)[0]
slow = np.argsort(en)[-8:]
print("last finishers (slot, start, end):", [(int(idx[i]), round(float(st[i]), 2), round(float(en[i]), 2)) for i in slow])
