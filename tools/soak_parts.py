#!/usr/bin/env python3
"""Soak diagnostic (round 6): N isolated (or overlapped) steps on a ring of 8 frames; every part of every step's result is hashed on its
own, and a step whose part differs from the same frame one lap earlier is reported with the part's name -- which result was wrong,
not just that one was.   python tools/soak_parts.py [steps] [isolated=1] [poll=1]"""
import hashlib
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
isolated = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
os.environ["MORB_POLL"] = sys.argv[3] if len(sys.argv) > 3 else "1"
import multi_orb_slam_amd as m
from multi_orb_slam_amd import pipeline, rt, synth

W, H, RING, AHEAD = 640, 480, 8, 3
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=1000)] * 2, W, H)
dev = []
for t in range(RING):
    row = []
    for c in range(2):
        b = rt.DeviceBuffer(W * H); b.upload(synth.image(c, t, W, H)); row.append(b)
    dev.append(row)
rt.device_sync()
arg = lambda t: [(dev[t % RING][c].ptr, W) for c in range(2)]
fe.copy_results = False
PARTS = ("match_of_feature", "kps", "desc", "uright", "cross0", "cross1", "cross2", "counts")
hist = []
keep = {}
if not isolated:
    for k in range(1, AHEAD):
        fe.announce(arg(k), resident=True)
bad = 0
for t in range(steps):
    r = fe.step(arg(t), resident=True, next_images=None if isolated else arg(t + AHEAD))
    vals = [r["match_of_feature"], r["kps"], r["desc"], r["uright"], r["cross"][0], r["cross"][1], r["cross"][2],
            np.array(list(r["counts"]) + [r["n_temporal"], r["n_cross"]])]
    d = [hashlib.blake2b(np.ascontiguousarray(a).tobytes(), digest_size=8).digest() for a in vals]
    hist.append(d)
    if t >= 24:
        for i, name in enumerate(PARTS):
            if d[i] != hist[t - 8][i] and hist[t - 8][i] == hist[t - 16][i]:
                bad += 1
                ref = keep.get((t % 8, i))
                cur = np.ascontiguousarray(vals[i]).copy()
                where = ""
                if ref is not None and ref.shape == cur.shape:
                    diff = np.flatnonzero(ref.view(np.uint8).ravel() != cur.view(np.uint8).ravel())
                    where = " %d bytes differ, first at %d (of %d)" % (len(diff), diff[0] if len(diff) else -1, cur.nbytes)
                print("step %d: %s differs from the lap before.%s" % (t, name, where), flush=True)
    if 16 <= t < 24:
        for i in range(len(PARTS)):
            keep[(t % 8, i)] = np.ascontiguousarray(vals[i]).copy()
fe.close()
print("soak_parts: %d steps, isolated=%s poll=%s: %d deviating parts" % (steps, isolated, os.environ["MORB_POLL"], bad))
