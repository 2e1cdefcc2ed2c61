#!/usr/bin/env python3
"""Four host threads, one extractor handle each, extracting different images over and over; every result is compared with the one the
same handle produced alone at the start.  Prints the number of mismatching calls per thread (keypoints / descriptors)."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth
W, H, NT, IT = 640, 480, int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 300
exs = [m.Extractor([m.ExtractorParams(nfeatures=1000)], W, H) for _ in range(NT)]
imgs = [[synth.image(t, k, W, H) for k in range(4)] for t in range(NT)]
ref = [[exs[t].extract([imgs[t][k]])[0] for k in range(4)] for t in range(NT)]
bad = [[0, 0] for _ in range(NT)]
first = [None] * NT
def work(t):
    for it in range(IT):
        k = it % 4
        kp, d = exs[t].extract([imgs[t][k]])[0]
        rk, rd = ref[t][k]
        if kp.tobytes() != rk.tobytes(): bad[t][0] += 1
        elif not np.array_equal(d, rd):
            bad[t][1] += 1
            if first[t] is None:
                rows = np.flatnonzero((d != rd).any(axis=1))
                first[t] = (it, len(rows), rows[:8].tolist(), kp["octave"][rows[:8]].tolist(), kp["x"][rows[:8]].tolist(), kp["y"][rows[:8]].tolist())
th = [threading.Thread(target=work, args=(t,)) for t in range(NT)]
[x.start() for x in th]; [x.join() for x in th]
print("mismatching calls per thread [keypoints, descriptors]:", bad)
for t in range(NT):
    if first[t]: print(" thread", t, "first: call %d, %d rows differ, rows %s octaves %s x %s y %s" % first[t])
