"""CPU mirror of multi_orb_slam_amd.pipeline.FrontEnd.step built on the oracle.  TEST INFRASTRUCTURE ONLY:
used as the checker in tests / smoke() and as the timed `cpu_baseline` leg of bench.py."""
import numpy as np
import oracle
from multi_orb_slam_amd import pipeline as P


class OracleFrontEnd:
    def __init__(self, params_per_cam, width, height, global_cams=None, cam_threads=False, calib=None):
        """cam_threads: extract the cameras (and match them across cameras) on one host thread each -- the variant ORB-SLAM2
        upstream uses for stereo (reference src/Frame.cc:106-109, commented out there); results are identical."""
        self.params = list(params_per_cam); self.n_cams = len(self.params)
        self.calib = calib
        self.pool = None
        if cam_threads and self.n_cams > 1:
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(self.n_cams)   # the oracle's C entry points release the GIL (ctypes)
        self.width, self.height = width, height
        p = self.params[0]
        self.scale = oracle.tables(p.nfeatures, p.scale_factor, p.nlevels, p.ini_th_fast, p.min_th_fast)["scale"]
        self.global_cams = global_cams or list(range(self.n_cams))
        self.depth = [P.synth_depth_image(g, width, height) for g in self.global_cams]
        self.prev = None

    def step(self, images, other_descs=None, on_extracted=None):
        """other_descs(cam_index) -> list of descriptor arrays of every OTHER camera of the rig in global camera order
        (multi-GPU mirror); None = the rig is just this process' cameras.  on_extracted(per_cam) is called with this front end's
        own [(keypoints, descriptors)] before the cross-camera matching asks for the others."""
        ext = lambda a: oracle.extract(a[0], a[1].nfeatures, a[1].scale_factor, a[1].nlevels, a[1].ini_th_fast, a[1].min_th_fast)
        per_cam = list(self.pool.map(ext, zip(images, self.params))) if self.pool else [ext(a) for a in zip(images, self.params)]
        if on_extracted is not None:
            on_extracted(per_cam)
        counts = [len(k) for k, _ in per_cam]
        cat = np.concatenate
        kps = cat([k for k, _ in per_cam]); desc = cat([d for _, d in per_cam])
        un_x, un_y = oracle.undistort_points(self.calib, kps["x"], kps["y"])           # Frame::UndistortKeyPoints
        bounds = oracle.image_bounds(self.calib, self.width, self.height)               # Frame::ComputeImageBounds
        off = np.concatenate([[0], np.cumsum(counts)])
        st = [oracle.stereo_from_depth(k, self.depth[c], P.MBF, un_x[off[c]:off[c + 1]]) for c, (k, _) in enumerate(per_cam)]
        uright = cat([a for a, _ in st]); depth = cat([b for _, b in st])
        cam_of = np.repeat(np.arange(self.n_cams, dtype=np.int32), counts)
        n_temporal = 0; match_of = np.zeros(0, np.int32)
        if self.prev is not None and len(kps) > 0:
            fr = oracle.FrameData(un_x, un_y, kps["octave"], kps["angle"], uright, cam_of,
                                  cat([np.arange(n, dtype=np.int32) for n in counts]), [d for _, d in per_cam], bounds)
            q = P.make_queries(self.prev, self.scale)
            n_temporal, match_of = oracle.search_by_projection_frames(fr, q, 100, True)
        self.prev = (kps, desc, depth, cam_of, un_x, un_y)
        def cross(c):
            others = other_descs(c) if other_descs is not None else [per_cam[o][1] for o in range(self.n_cams) if o != c]
            refs = cat(others) if others else np.zeros((0, 32), np.uint8)
            return oracle.bf_top2(per_cam[c][1], refs)
        res = list(self.pool.map(cross, range(self.n_cams))) if self.pool else [cross(c) for c in range(self.n_cams)]
        bi, bd, sd = cat([r[0] for r in res]), cat([r[1] for r in res]), cat([r[2] for r in res])
        return dict(kps=kps, desc=desc, uright=uright, depth=depth, un_x=un_x, un_y=un_y, counts=counts, n_temporal=n_temporal,
                    match_of_feature=match_of, cross=(bi, bd, sd), n_cross=int(P.accept_cross(bd, sd).sum()))


def assert_same_step(a, b):
    """Bit-exact equality of two step results (keypoint records, descriptors, stereo, match pairs)."""
    assert a["counts"] == b["counts"], (a["counts"], b["counts"])
    assert a["kps"].tobytes() == b["kps"].tobytes(), "keypoints differ"
    assert np.array_equal(a["desc"], b["desc"]), "descriptors differ"
    assert a["uright"].tobytes() == b["uright"].tobytes() and a["depth"].tobytes() == b["depth"].tobytes(), "stereo differs"
    assert a["un_x"].tobytes() == b["un_x"].tobytes() and a["un_y"].tobytes() == b["un_y"].tobytes(), "undistorted positions differ"
    assert a["n_temporal"] == b["n_temporal"], (a["n_temporal"], b["n_temporal"])
    assert np.array_equal(a["match_of_feature"], b["match_of_feature"]), "temporal matches differ"
    for u, v in zip(a["cross"], b["cross"]):
        assert np.array_equal(u, v), "cross-camera top-2 differs"
    assert a["n_cross"] == b["n_cross"]
