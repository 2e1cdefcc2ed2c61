import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline, rt, _lib
lib = _lib.lib()
W, H, NF, NC = (int(x) for x in sys.argv[1:5])
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=NF)] * NC, W, H)
fe.copy_results = False
dev = []
for t in range(8):
    row = []
    for c in range(NC):
        b = rt.DeviceBuffer(W * H); b.upload(synth.image(c, t, W, H)); row.append(b)
    dev.append(row)
rt.device_sync()
prep = [fe.prepare([(dev[t][c].ptr, W) for c in range(NC)], True) for t in range(8)]
for k in range(1, 3):
    fe.announce(prep[k], resident=True)
out = (C.c_uint64 * 64)()
tot = []
for i in range(3000):
    fe.step(prep[i % 8], resident=True, next_images=prep[(i + 3) % 8])
    if i > 500 and i % 25 == 0:
        lib.morb_debug_phases_matcher(0, out); v = list(out)
        if v[1] == 0 and v[61] > v[0]:
            tot.append(((v[61] - v[0]) / 100.0, (v[2] - v[0]) / 100.0, (v[60] - v[2]) / 100.0))
a = np.array(tot)
print("resolve in the overlapped loop (%dx%d @%d x %d, reserve=%s): total median %.1f us (p5 %.1f p95 %.1f), set-up %.1f, rounds+tail %.1f; n=%d" %
      (W, H, NF, NC, os.environ.get("MORB_RESERVE_CUS", "0"), np.median(a[:, 0]), np.percentile(a[:, 0], 5), np.percentile(a[:, 0], 95), np.median(a[:, 1]), np.median(a[:, 2]), len(a)))
