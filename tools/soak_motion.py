#!/usr/bin/env python3
"""A long overlapped stream whose motion (du, dv, th) changes every step, hashed: the digest must not depend on WHERE the queries are
built (MORB_MOTION_ON_DEVICE=1 / 0), on the top-2's arithmetic (MORB_TOP2_FP4=1 / 0) or on the resolve (MORB_RESOLVE_MONO=1 / 0).
Usage: soak_motion.py [steps] -- prints one line with the digest; run it under each setting and compare."""
import os, sys, zlib
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline, rt
from multi_orb_slam_amd.frontend import NO_QUERY_RECORDS
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
W, H, RING, AHEAD = 640, 480, 16, 3
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=1000), m.ExtractorParams(nfeatures=700)], W, H)
dev = [[rt.DeviceBuffer(W * H) for c in range(2)] for t in range(RING)]
for t in range(RING):
    for c in range(2):
        dev[t][c].upload(synth.image(c, t, W, H))
rt.device_sync()
args = [fe.prepare([(dev[t][c].ptr, W) for c in range(2)], True) for t in range(RING)]
rng = np.random.RandomState(5)
for k in range(1, AHEAD):
    fe.announce(args[k % RING], resident=True)
crc = 0; tot = [0, 0, 0]
for it in range(N):
    mo = (float(np.float32(rng.uniform(-6, 6))), float(np.float32(rng.uniform(-4, 4))), float(np.float32(rng.choice([7.0, 15.0, 15.0, 30.0]))))
    r = fe.fe.step_ahead(args[it % RING], args[(it + AHEAD) % RING], mo, 50, 0.7, flags=NO_QUERY_RECORDS if it % 3 else 0, copy=False)
    mof = r["match_of_feature"]
    crc = zlib.crc32(np.ascontiguousarray(mof).tobytes(), crc)
    if "cross" in r:
        for a in r["cross"]:
            crc = zlib.crc32(np.ascontiguousarray(a).tobytes(), crc)
    tot[0] += r["n_total"]; tot[1] += r["n_temporal"]; tot[2] += (r["n_cross"] or 0)
print("soak_motion: %d steps, features %d, temporal %d, cross %d, digest %08x" % (N, tot[0], tot[1], tot[2], crc & 0xffffffff))
fe.close()
