#!/usr/bin/env python3
"""The whole extraction (keypoint records and descriptors) against the oracle over RANDOM parameter sets: image sizes 120 .. 900 x 100 .. 700
(aspect below 4), scale factors 1.1 .. 1.5, 2 .. 9 levels, FAST thresholds that cover both forms of the packed quick test (min_th 1 .. 127:
the byte form; 128 and above: the unpacked form) and ini_th at and above min_th, feature counts 50 .. 2500, the image families of
synth.FAMILIES.  usage: fuzz_extractor_random.py [cases] [seed]"""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth
import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
done = 0
for case in range(N):
    sf = float(rng.choice([1.1, 1.2, 1.2, 1.3, 1.5]))
    nl = int(rng.integers(2, 10))
    W = int(rng.integers(120, 901)); H = int(rng.integers(100, 701))
    if W > 3.9 * H: W = int(3.9 * H)
    if H > 3.9 * W: H = int(3.9 * W)
    while nl > 2 and min(W, H) / sf ** (nl - 1) < 72:
        nl -= 1
    if min(W, H) / sf ** (nl - 1) < 72:
        continue
    min_th = int(rng.choice([1, 2, 3, 7, 7, 7, 20, 60, 127, 128, 129, 200]))   # (orbx_create: 1 <= min_th <= ini_th <= 255)
    ini_th = min(255, max(min_th, int(rng.choice([min_th, min_th + 5, 20, 40, 250]))))
    nf = int(rng.choice([50, 300, 1000, 2500]))
    kind = synth.FAMILIES[int(rng.integers(0, len(synth.FAMILIES)))] if rng.random() < 0.5 else None
    p = m.ExtractorParams(nfeatures=nf, scale_factor=sf, nlevels=nl, ini_th_fast=ini_th, min_th_fast=min_th)
    ex = m.Extractor([p, p], W, H)
    imgs = [synth.family_image(kind, case + c, 0, W, H) if kind else synth.image(case + c, 0, W, H) for c in range(2)]
    try:
        out = ex.extract(imgs)
    except Exception as e:
        print("extract refused:", case, (W, H), sf, nl, ini_th, min_th, nf, kind, repr(e)[:120], flush=True)
        ex.close(); continue
    for c in range(2):
        try:
            ok, od = oracle.extract(imgs[c], nfeatures=nf, scale_factor=sf, nlevels=nl, ini_th=ini_th, min_th=min_th)
        except ValueError as e:   # (inputs the reference leaves undefined: the oracle says so)
            print("oracle refuses:", case, repr(e)[:100]); continue
        k, d = out[c]
        assert k.tobytes() == ok.tobytes() and np.array_equal(d, od), ("differs", case, (W, H), sf, nl, ini_th, min_th, nf, kind, c, len(k), len(ok))
    done += 1
    ex.close()
print("ok: %d of %d cases compared" % (done, N))
