#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

The reference (C++ + OpenCV) cannot be built or run in this image, so these fixtures are produced by the CPU oracle
(oracle/orb_oracle.cpp) on seeded synthetic inputs; they pin the oracle against drift and give the GPU tests a
fixture-based check that does not need the oracle at all.  Inputs are regenerated from multi_orb_slam_amd.synth (pure
integer hashing), only the expected outputs are stored.
"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
import oracle, helpers
from multi_orb_slam_amd import synth


def main():
    # extractor: 320x240, 300 features (cam 0, frame 0) and 150 features (cam 1, frame 2)
    out = {}
    for tag, cam, t, nf in (("a", 0, 0, 300), ("b", 1, 2, 150)):
        img = synth.image(cam, t, 320, 240)
        k, d = oracle.extract(img, nfeatures=nf)
        out["kps_" + tag] = k; out["desc_" + tag] = d
        out["img_crc_" + tag] = np.array([np.uint32(np.bitwise_xor.reduce(img.astype(np.uint32).ravel() * np.arange(1, img.size + 1, dtype=np.uint32)))])
    np.savez_compressed(os.path.join(HERE, "extract_320x240.npz"), **out)
    # matcher: top-2 and projection search
    r = synth.descriptors(700, 42); q = synth.perturbed_queries(synth.descriptors(500, 42), 7)
    bi, bd, sd = oracle.bf_top2(q, r)
    fr = helpers.make_frame_arrays([600, 300], 640, 480, 21)
    qs = helpers.make_queries(fr, 700, 61, th=15.0)
    n, mo = oracle.search_by_projection_frames(oracle.FrameData(**fr), qs, 100, True)
    qp = qs.copy(); qp["cam"] = 0; qp["max_level"] = np.maximum(qp["max_level"], 0); qp["min_level"] = qp["max_level"] - 1
    n2, mo2 = oracle.search_by_projection_points(oracle.FrameData(**fr), qp, None, 0.8, 100)
    np.savez_compressed(os.path.join(HERE, "matcher.npz"), best_idx=bi, best_dist=bd, second_dist=sd, n_frames=np.array([n]),
                        match_frames=mo, n_points=np.array([n2]), match_points=mo2)
    print("extract:", len(out["kps_a"]), len(out["kps_b"]), "top2 acc:", int((bd <= 50).sum()), "proj:", n, n2)


if __name__ == "__main__":
    main()
