#!/usr/bin/env python3
"""Soak: N overlapped steps (ISOLATED=1: N isolated steps, no look-ahead); prints a digest of every step's results.  Run twice
(MORB_POLL=1 / 0) and compare: the status-word polling of orbf_step_end must never hand out anything but the finished arrays --
including, for isolated steps, the camera-pair top-2 and the result mirrors that ride in the projection kernel's launch."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline, rt
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
W, H, RING = 640, 480, 8
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=1000)] * 2, W, H)
dev = []
for t in range(RING):
    row = []
    for c in range(2):
        b = rt.DeviceBuffer(W * H); b.upload(synth.image(c, t, W, H)); row.append(b)
    dev.append(row)
rt.device_sync()
arg = lambda t: [(dev[t % RING][c].ptr, W) for c in range(2)]
h = hashlib.sha256()
ISOLATED = os.environ.get("ISOLATED") == "1"
if not ISOLATED:
    fe.announce(arg(1), resident=True)
log = open(sys.argv[2], "w") if len(sys.argv) > 2 else None
d = lambda a: hashlib.md5(a.tobytes()).hexdigest()[:8]
for t in range(N):
    r = fe.step(arg(t), resident=True, next_images=None if ISOLATED else arg(t + 2))
    h.update(r["match_of_feature"].tobytes()); h.update(r["kps"].tobytes()); h.update(r["cross"][0].tobytes())
    h.update(r["cross"][1].tobytes()); h.update(r["cross"][2].tobytes()); h.update(r["desc"].tobytes()); h.update(r["uright"].tobytes())
    h.update(str((r["counts"], r["n_temporal"])).encode())
    if log:
        log.write("%d mof=%s kps=%s desc=%s x0=%s x1=%s ur=%s cnt=%s nt=%d\n" % (t, d(r["match_of_feature"]), d(r["kps"]), d(r["desc"]), d(r["cross"][0]),
                                                                             d(r["cross"][1]), d(r["uright"]), r["counts"], r["n_temporal"]))
print("digest", N, h.hexdigest(), flush=True)
if log:
    log.close()
os._exit(0)
