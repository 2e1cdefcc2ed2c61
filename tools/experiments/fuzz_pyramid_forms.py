#!/usr/bin/env python3
"""Every pyramid level against the oracle over random geometries (image sizes 40 .. 2000 x 40 .. 1200, scale factors 1.05 .. 1.6, 2 .. 11
levels, one to three cameras of different sizes), in whatever pyramid form the process's environment selects -- run it under
MORB_PYR_CHAIN=1 MORB_PYRAMID_PAIRS=0 with different MORB_PYR_T4_W / _H / MORB_PYR_SPLIT for the tile launches (tile sizes and the
split are read once per process).  usage: fuzz_pyramid_forms.py [cases] [seed]"""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth
import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
forms = {}
for case in range(N):
    n_cams = int(rng.integers(1, 4))
    sf = float(rng.choice([1.05, 1.1, 1.2, 1.2, 1.25, 1.3, 1.5, 1.6]))
    nl = int(rng.integers(2, 12))
    W = int(rng.integers(64, 2001)); H = int(rng.integers(64, 1201))
    # the smallest level must keep the 2 x 19 pixel border plus a cell: cap the level count
    while nl > 2 and min(W, H) / sf ** (nl - 1) < 72:
        nl -= 1
    if min(W, H) / sf ** (nl - 1) < 72:
        W = max(W, int(72 * sf ** (nl - 1)) + 1); H = max(H, int(72 * sf ** (nl - 1)) + 1)
    lo = int(72 * sf ** (nl - 1)) + 1
    sizes = [(W, H)] + [(int(rng.integers(min(lo, W), W + 1)), int(rng.integers(min(lo, H), H + 1))) for _ in range(n_cams - 1)]
    p = m.ExtractorParams(nfeatures=300, scale_factor=sf, nlevels=nl)
    ex = m.Extractor([p] * n_cams, W, H)
    imgs = [synth.image(case + c, 0, w, h) for c, (w, h) in enumerate(sizes)]
    try:
        ex.extract(imgs)
    except Exception as e:   # (reported, not fatal for this fuzz: the levels are compared anyway when the run got that far)
        print("extract failed:", case, sizes, sf, nl, repr(e)[:160], flush=True)
        ex.close(); continue
    f = ex.pyramid_form(); forms[f] = forms.get(f, 0) + 1
    for c in range(n_cams):
        for l, ref in enumerate(oracle.pyramid(imgs[c], sf, nl)):
            got = ex.debug_level(c, l)
            assert got.shape == ref.shape and np.array_equal(got, ref), ("level differs", case, sizes, sf, nl, c, l, f)
    ex.close()
print("ok: %d cases, pyramid forms used %s" % (N, forms))
