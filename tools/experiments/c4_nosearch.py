import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import pipeline, rt, synth
W, H, NC, NF = 1920, 1080, 8, 4000
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=NF)] * NC, W, H)
RING = 6
dev = [[rt.DeviceBuffer(W * H) for c in range(NC)] for t in range(RING)]
for t in range(RING):
    for c in range(NC):
        dev[t][c].upload(synth.image(c, t, W, H))
rt.device_sync()
prep = [fe.prepare([(dev[t][c].ptr, W) for c in range(NC)]) for t in range(RING)]
nf = fe.fe
for mode in ("search", "nosearch"):
    nf.reset() if hasattr(nf, "reset") else None
    nf.prefetch(prep[1]); nf.prefetch(prep[2])
    tt = []; hs = []
    for t in range(80):
        t0 = time.perf_counter()
        nf.prefetch(prep[(t + 3) % RING])
        if mode == "search":
            r = nf.step(prep[t % RING], None, 0, copy=False, motion=(3.0, 1.0, 15.0))
        else:
            r = nf.step(prep[t % RING], None, 0, copy=False)
        tt.append(time.perf_counter() - t0); hs.append(r["host_us"])
    hs = np.array(hs[30:])
    print(mode, "step us", round(float(np.median(tt[30:])) * 1e6, 1), "host_us [prep, enqueue, wait, post]", np.median(hs, axis=0).round(1))
fe.close()
