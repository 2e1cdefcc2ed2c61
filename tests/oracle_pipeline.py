"""CPU mirror of multi_orb_slam_amd.pipeline.FrontEnd.step built on the oracle.  TEST INFRASTRUCTURE ONLY:
used as the checker in tests / smoke() and as the timed `cpu_baseline` leg of bench.py."""
import numpy as np
import oracle
from multi_orb_slam_amd import pipeline as P


class OracleFrontEnd:
    def __init__(self, params_per_cam, width, height):
        self.params = list(params_per_cam); self.n_cams = len(self.params)
        self.width, self.height = width, height
        p = self.params[0]
        self.scale = oracle.tables(p.nfeatures, p.scale_factor, p.nlevels, p.ini_th_fast, p.min_th_fast)["scale"]
        self.prev = None

    def extract(self, images):
        return [oracle.extract(im, p.nfeatures, p.scale_factor, p.nlevels, p.ini_th_fast, p.min_th_fast)
                for im, p in zip(images, self.params)]

    def step(self, images, other_descs=None):
        """other_descs: descriptor arrays of cameras owned by OTHER ranks, as {global_cam: desc} (multi-GPU mirror)."""
        per_cam = self.extract(images)
        counts = [len(k) for k, _ in per_cam]
        n_total = sum(counts)
        n_temporal = 0; match_of = np.zeros(0, np.int32)
        if self.prev is not None and n_total > 0:
            cat = np.concatenate
            uright = cat([P.synth_uright(k) for k, _ in per_cam])
            fr = oracle.FrameData(cat([k["x"] for k, _ in per_cam]), cat([k["y"] for k, _ in per_cam]),
                                  cat([k["octave"] for k, _ in per_cam]), cat([k["angle"] for k, _ in per_cam]), uright,
                                  cat([np.full(len(k), c, np.int32) for c, (k, _) in enumerate(per_cam)]),
                                  cat([np.arange(len(k), dtype=np.int32) for k, _ in per_cam]),
                                  [d for _, d in per_cam], (0, 0, self.width, self.height))
            q = P.make_queries(self.prev, self.scale)
            n_temporal, match_of = oracle.search_by_projection_frames(fr, q, 100, True)
        self.prev = per_cam
        cross = []; n_cross = 0
        for c in range(self.n_cams):
            others = [per_cam[o][1] for o in range(self.n_cams) if o != c]
            if other_descs is not None:
                others = other_descs(c, per_cam)
            refs = np.concatenate(others) if others else np.zeros((0, 32), np.uint8)
            bi, bd, sd = oracle.bf_top2(per_cam[c][1], refs)
            cross.append((bi, bd, sd))
            n_cross += int(P.accept_cross(bd, sd).sum())
        return dict(per_cam=per_cam, counts=counts, n_temporal=n_temporal, match_of_feature=match_of, cross=cross,
                    n_cross=n_cross)


def assert_same_step(a, b):
    """Bit-exact equality of two step results (keypoint records, descriptors, match pairs)."""
    assert a["counts"] == b["counts"], (a["counts"], b["counts"])
    for (ka, da), (kb, db) in zip(a["per_cam"], b["per_cam"]):
        assert ka.tobytes() == kb.tobytes(), "keypoints differ"
        assert np.array_equal(da, db), "descriptors differ"
    assert a["n_temporal"] == b["n_temporal"], (a["n_temporal"], b["n_temporal"])
    assert np.array_equal(a["match_of_feature"], b["match_of_feature"]), "temporal matches differ"
    for x, y in zip(a["cross"], b["cross"]):
        for u, v in zip(x, y):
            assert np.array_equal(u, v), "cross-camera top-2 differs"
    assert a["n_cross"] == b["n_cross"]
