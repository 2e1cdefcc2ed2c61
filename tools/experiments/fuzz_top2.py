#!/usr/bin/env python3
"""Exhaustive and camera-pair top-2 against the oracle's brute force on TIE-HEAVY data: descriptors drawn from a few base vectors with
0-3 flipped bits (equal distances everywhere, exact duplicates, best == second), random sizes across the tile / slice boundaries, all
three forms (FP4, int8, popcount).  Any mismatch prints the configuration and stops.  usage: fuzz_top2.py [seed] [iterations]"""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import multi_orb_slam_amd as m
import oracle, helpers

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30


def tie_heavy(n, nbase, maxflip, far_p):
    base = rng.randint(0, 256, (nbase, 32)).astype(np.uint8)
    d = base[rng.randint(0, nbase, n)].copy()
    for i in range(n):
        for _ in range(rng.randint(0, maxflip + 1)):
            b = rng.randint(0, 256); d[i, b >> 3] ^= np.uint8(1 << (b & 7))
    far = rng.rand(n) < far_p                       # a few all-different rows: distance 256 exists
    d[far] = ~d[far]
    return np.ascontiguousarray(d)


mt = m.Matcher()
for it in range(N):
    nq = int(rng.choice([1, 63, 64, 65, 200, 257, 1000, 3000, 6000]))
    nr = int(rng.choice([1, 2, 63, 64, 100, 129, 1000, 4097, 9000]))
    nbase = int(rng.choice([1, 2, 5, 40])); maxflip = int(rng.choice([0, 1, 3])); far_p = float(rng.choice([0.0, 0.02]))
    q = tie_heavy(nq, nbase, maxflip, far_p); r = tie_heavy(nr, nbase, maxflip, far_p)
    if rng.rand() < 0.5: r[: min(nr, nq)] = q[: min(nr, nq)]   # exact copies
    e = oracle.bf_top2(q, r)
    for form, (mc, fp4) in (("fp4", (1, -1)), ("int8", (1, 0)), ("popcount", (0, -1))):
        pm = m.Matcher.use_matrix_cores(mc); pf = m.Matcher.use_fp4_top2(fp4)
        g = mt.hamming_top2(q, r)
        m.Matcher.use_matrix_cores(pm); m.Matcher.use_fp4_top2(pf)
        assert all(np.array_equal(a, b) for a, b in zip(g, e)), ("top2", form, it, nq, nr, nbase, maxflip, far_p)
    # camera-pair form: the same rows as a frame of 2-4 cameras
    ncam = int(rng.choice([2, 3, 4]))
    per = [int(rng.choice([1, 60, 64, 300, 1100, 2500])) for _ in range(ncam)]
    fr = helpers.make_frame_arrays(per, 640, 480, int(rng.randint(1, 10000)))
    fr["descs"] = [tie_heavy(n, nbase, maxflip, far_p) for n in per]
    F = mt.frame(m.FrameData(**fr))
    for form, (mc, fp4) in (("fp4", (1, -1)), ("int8", (1, 0)), ("popcount", (0, -1))):
        pm = m.Matcher.use_matrix_cores(mc); pf = m.Matcher.use_fp4_top2(fp4)
        bi, bd, sd = mt.cross_top2(F)
        m.Matcher.use_matrix_cores(pm); m.Matcher.use_fp4_top2(pf)
        off = 0
        for c in range(ncam):
            others = [fr["descs"][o] for o in range(ncam) if o != c]
            ebi, ebd, esd = oracle.bf_top2(fr["descs"][c], np.concatenate(others))
            sl = slice(off, off + per[c])
            assert np.array_equal(bi[sl], ebi) and np.array_equal(bd[sl], ebd) and np.array_equal(sd[sl], esd), ("cross", form, it, per, c, nbase, maxflip, far_p)
            off += per[c]
    F.close()
    print("ok", it, "nq", nq, "nr", nr, "bases", nbase, "flips", maxflip, "far", far_p, "cams", per, flush=True)
mt.close()
print("ALL OK")
