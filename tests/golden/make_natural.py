#!/usr/bin/env python3
"""Regenerates tests/golden/natural_photos.npz and tests/golden/natural_expected.npz (BUILD CONTAINER ONLY).

  python tests/golden/make_natural.py photos      decode the three photographs bundled with this image's Python packages
                                                  (scikit-learn: china.jpg, flower.jpg; matplotlib: grace_hopper.jpg), grey with
                                                  cvtColor's 8-bit formula (tests/natural.py: rgb_to_grey) -> natural_photos.npz
  python tests/golden/make_natural.py expected    the CPU oracle's outputs on frames cut from those planes (tests/natural.py) ->
                                                  natural_expected.npz

The reference (C++ + OpenCV) cannot run in this image, so the expected outputs are the oracle's: they pin it against drift and let
the GPU tests check the HIP path against committed vectors without the oracle.  What is stored per (photo, size):
  640x480: camera 0 and 1 of frame 0 in full (keypoints, descriptors); larger sizes: SHA-256 of the same bytes + counts;
  a 4-step two-camera sequence through the whole front-end step: per step SHA-256 of keypoints / descriptors / stereo / temporal
  match list / cross-camera top-2, and the counts a reader can eyeball (features, temporal matches, accepted cross-camera matches).
"""
import os
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np


def make_photos():
    import natural
    from PIL import Image
    from sklearn.datasets import load_sample_images
    import matplotlib
    d = load_sample_images()
    named = {os.path.basename(f).split(".")[0]: im for f, im in zip(d.filenames, d.images)}
    hop = np.asarray(Image.open(os.path.join(os.path.dirname(matplotlib.__file__), "mpl-data", "sample_data", "grace_hopper.jpg")).convert("RGB"))
    out = {"china": natural.rgb_to_grey(named["china"]), "flower": natural.rgb_to_grey(named["flower"]), "hopper": natural.rgb_to_grey(hop)}
    np.savez_compressed(natural.PHOTOS_NPZ, **out)
    print({k: v.shape for k, v in out.items()}, os.path.getsize(natural.PHOTOS_NPZ), "bytes")


def make_expected():
    import natural, oracle
    from natural import sha, step_digests, SEQ_STEPS
    import multi_orb_slam_amd as m
    from oracle_pipeline import OracleFrontEnd
    out = {"photo_sums": np.array([int(natural.photos()[k].astype(np.int64).sum()) for k in natural.PHOTOS], np.int64)}
    for photo in natural.PHOTOS:
        for (w, h, nf) in natural.SIZES:
            tag = "%s_%d" % (photo, w)
            for c in range(2):
                k, d = oracle.extract(natural.frame(photo, c, 0, w, h), nfeatures=nf)
                if w == 640:
                    out["%s_kps%d" % (tag, c)] = k; out["%s_desc%d" % (tag, c)] = d
                out["%s_sha%d" % (tag, c)] = sha(k, d); out["%s_n%d" % (tag, c)] = np.array([len(k)], np.int32)
            params = [m.ExtractorParams(nfeatures=nf)] * 2
            ofe = OracleFrontEnd(params, w, h, cam_threads=True)
            for t in range(SEQ_STEPS):
                r = ofe.step(natural.rig(photo, t, w, h))
                for key, v in step_digests(r).items():
                    out["%s_step%d_%s" % (tag, t, key)] = v
            print(tag, "features", r["counts"], "temporal", r["n_temporal"], "cross", r["n_cross"], flush=True)
    p = os.path.join(HERE, "natural_expected.npz")
    np.savez_compressed(p, **out)
    print(os.path.getsize(p), "bytes")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("photos", "all"):
        make_photos()
    if what in ("expected", "all"):
        make_expected()
