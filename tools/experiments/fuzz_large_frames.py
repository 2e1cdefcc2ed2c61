#!/usr/bin/env python3
"""Frame searches of frames beyond one workgroup's LDS (per-camera resolve / per-sweep form) against the oracle: random rigs of 2-8
cameras with 20 000-36 000 features (balanced, unbalanced, empty cameras), random query counts, windows, blocking patterns, occupied
flags, both orientation settings.  usage: fuzz_large_frames.py [seed] [iterations]"""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import multi_orb_slam_amd as m
import oracle, helpers

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
mt = m.Matcher(0.8, True)
for it in range(N):
    ncam = int(rng.choice([2, 3, 4, 6, 8]))
    total = int(rng.choice([20000, 24000, 32000, 36000]))
    w = rng.rand(ncam) ** float(rng.choice([0.2, 1.0, 3.0])); w /= w.sum()
    per = [int(total * x) for x in w]
    if rng.rand() < 0.3: per[int(rng.randint(ncam))] = 0
    nq = int(rng.choice([300, 4000, 12000, 24000, 40000]))
    th = float(rng.choice([5.0, 15.0, 30.0]))
    blocks = int(rng.choice([0, 1, 1, 2])); dup = float(rng.choice([0.3, 0.9]))
    seed = int(rng.randint(1, 10000))
    fr = helpers.make_frame_arrays(per, 1920, 1080, seed)
    if sum(per) < 19000: continue
    q = helpers.make_queries(fr, nq, seed + 40, th=th, blocks=blocks, dup_prob=dup)
    if rng.rand() < 0.5:
        q = np.ascontiguousarray(q[np.argsort(q["cam"], kind="stable")])
    occ = (helpers.rand_unit(sum(per), seed + 7) < 0.25).astype(np.uint8) if rng.rand() < 0.4 else None
    F = mt.frame(m.FrameData(**fr)); OF = oracle.FrameData(**fr)
    for check_ori in (True, False):
        mt.check_orientation = check_ori
        n, mo = mt.SearchByProjection(F, q, occupied=occ)
        on, omo = oracle.search_by_projection_frames(OF, q, 100, check_ori, occ)
        assert n == on and np.array_equal(mo, omo), ("frames", it, per, seed, nq, th, blocks, dup, check_ori, n, on, mt.last_resolve())
    print("ok", it, per, "nq", nq, "th", th, "blocks", blocks, "occ", occ is not None, "matches", n, mt.last_resolve(), flush=True)
    F.close()
mt.close()
print("ALL OK")
