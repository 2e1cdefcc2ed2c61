#!/usr/bin/env python3
"""SearchByProjection(Frame, Frame) and (Frame, points) against the oracle over random frames / query sets: camera counts, feature
counts (incl. beyond the 2048 queries a workgroup keeps in registers and dense frames that contest heavily), window sizes, blocking
patterns, both orientation settings.  Any mismatch prints the configuration and stops."""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import multi_orb_slam_amd as m
import oracle, helpers

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
mt = m.Matcher(0.8, True)
for it in range(N):
    ncam = int(rng.choice([1, 2, 2, 3, 4]))
    n_per_cam = [int(rng.choice([40, 200, 700, 1000, 2000])) for _ in range(ncam)]
    w, h = [(640, 480), (320, 240), (1280, 720)][rng.randint(3)]
    seed = int(rng.randint(1, 10000))
    nq = int(rng.choice([50, 500, 1500, 2500, 5000]))
    th = float(rng.choice([3.0, 7.0, 15.0, 30.0, 60.0]))
    blocks = int(rng.choice([0, 1, 1, 2]))
    dup = float(rng.choice([0.2, 0.5, 0.9]))
    fr = helpers.make_frame_arrays(n_per_cam, w, h, seed)
    q = helpers.make_queries(fr, nq, seed + 40, th=th, blocks=blocks, dup_prob=dup)
    F = mt.frame(m.FrameData(**fr)); OF = oracle.FrameData(**fr)
    for check_ori in (True, False):
        mt.check_orientation = check_ori
        n, mo = mt.SearchByProjection(F, q)
        on, omo = oracle.search_by_projection_frames(OF, q, 100, check_ori)
        assert n == on and np.array_equal(mo, omo), ("frames", it, n_per_cam, w, h, seed, nq, th, blocks, dup, check_ori, n, on)
    mt.check_orientation = True
    cntp, mop = mt.SearchByProjectionPoints(F, q)
    onp, omop = oracle.search_by_projection_points(OF, q, None, 0.8, 100)
    assert cntp == onp and np.array_equal(mop, omop), ("points", it, n_per_cam, w, h, seed, nq, th, blocks, dup)
    print("ok", it, n_per_cam, (w, h), "nq", nq, "th", th, "blocks", blocks, "matches", n, cntp, mt.last_resolve(), flush=True)
    F.close()
mt.close()
print("ALL OK")
