import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth
import oracle
for (w, h) in [(640, 480), (641, 479), (1920, 1080), (333, 777)]:
    ex = m.Extractor([m.ExtractorParams(nfeatures=500)], w, h)
    img = synth.image(0, 0, w, h)
    ex(img)
    print(w, h, "form", ex.pyramid_form())
    bad = 0
    for l, ref in enumerate(oracle.pyramid(img)):
        got = ex.debug_level(0, l)
        eq = got.shape == ref.shape and np.array_equal(got, ref)
        if not eq:
            d = np.argwhere(got != ref) if got.shape == ref.shape else None
            print("  level", l, "DIFFERS", got.shape, ref.shape, None if d is None else (len(d), d[:5].tolist()))
            bad += 1
    print("  levels differing:", bad)
    ex.close()
